"""TEST INFRASTRUCTURE.  Pins the HEADLINE workload (bench.py, BASELINE.json configs[1]) and a sweep of scan-sized
ICP runs against oracle/_ref — the real reference compiled in place from /root/reference.  Run in the build
container only; the fixtures are committed so the GPU box, which has no /root/reference, can check against them.

  tests/golden/bench_seed11.npz, bench_seed23.npz   for bench.build_inputs(1_000_000, seed=11 | 23) (11 is what bench.py runs):
        icp_pose / icp_err / icp_n_corrs[10] / icp_errs[10]   10 FIXED iterations composed from the reference's own
              icp_find_corrs + icp_estimate_rigid_xform_pt2pl (lib/rs/icp.h:306-412,210-298) with the loop of
              icp_align (:433-497, radius schedule :493) minus the stop test (oracle/ref_driver.cpp: ref_icp_iterate)
        scores[256]                mgs_compute_object_alignment_score (apps/pose_proposal/pose_proposal.cpp:93-158), K = 64
        labels / min_dists / ids   rspf_arrangement_to_labels + rspf__assign_temporary_labels (lib/rs/rs_pointcloud_filters.cpp:
              738-879) by the REFERENCE's own text (oracle/_ref/libref_filters.so, round 4; `labels_source` says so) — labels
              int8[n] stored whole, min_dists / class ids / instance ids as sha256 (+ a strided sample of min_dists)
        input digests              sha256 of the generated inputs, so that a test on another machine notices if the
              generator (numpy, rescan_amd/synth.py) no longer produces the arrays the fixture was made for

  tests/golden/sweep_icp.npz      24 scan-to-scan icp_align runs (seeds 1..24) on >= 100 k-point scans, the
        reference's own icp_align (stop test included): final pose, error; iteration counts from ref_icp_iterate
        with the stop test on (same loop; asserted to end at the same pose bit for bit)

  tests/golden/bench_seed31.npz … bench_seed38.npz   (round 3) eight more rooms of the headline's size, the same
        fields — labels as a sha256 + a strided sample instead of whole, to keep the files at a few KB — plus
        stop_pose / stop_err / stop_iters / stop_params: the reference's own icp_align, STOP TEST ON (lib/rs/icp.h:489),
        on the same two ~0.98 M-point scans, the three call sites' parameter sets in turn and start poses from
        5 mm / 0.3 deg to 3 cm / 2 deg (MORE_SEEDS, more_stop_case)

  tests/golden/bench_seed11_centre.npz, bench_seed31_centre.npz, bench_seed32_centre.npz   (round 5) bench.py --centre's inputs: the same
        rooms moved so that the scan's median point is the origin — the regime in which the reference's fp32 centroid sums hover
        around zero — with the same fields as --more's; bench_seed11_t1.npz, _t2.npz: the further scan pairs of bench.py --timesteps 4

Usage:  python oracle/gen_golden_bench.py [--bench-only [--seed N] | --sweep-only | --more [--seed N] | --labels-only [--seed N] | --strong-only | --static-labels
                                          | --centre [--seed N] | --pair T0 [--seed N] | --units]
        --units (round 6) -> tests/golden/bench_seed11_units.npz: the further unit lists of bench.py --gpus N (weak scaling), N <= 8
        --labels-only recomputes the label fields of every bench_seed*.npz from the reference build and leaves the rest as it is
"""
import ctypes as C
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, Ref, RefFilters, build, f32p, i32p  # noqa: E402
from rescan_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
I4 = np.eye(4, dtype=np.float32).ravel()
BENCH_SEEDS = [11, 23]            # 11: the bench's own workload; 23: a second room of the same size
MORE_SEEDS = list(range(31, 39))  # round 3: further rooms of the headline's size, with a stop-test-on icp_align run each
SWEEP_SEEDS = list(range(1, 25))
SWEEP_POINTS = 120_000


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_iterate(R, p1, n1, p2, n2, T0, T2, max_dist, max_angle, n_iters, stop_test):
    f = R.lib.ref_icp_iterate
    f.restype = C.c_int32
    f.argtypes = [f32p, f32p, C.c_int32, f32p, f32p, C.c_int32, f32p, f32p, C.c_float, C.c_float, C.c_int32, C.c_int32,
                  i32p, f32p, C.POINTER(C.c_float)]
    T = np.ascontiguousarray(T0, np.float32).ravel().copy()
    nc = np.zeros(n_iters, np.int32); errs = np.zeros(n_iters, np.float32); err = C.c_float()
    done = f(p1, n1, len(p1), p2, n2, len(p2), T, np.ascontiguousarray(T2, np.float32).ravel(), float(max_dist), float(max_angle),
             int(n_iters), int(stop_test), nc, errs, C.byref(err))
    return T, np.float32(err.value), done, nc, errs


def sweep_inputs(seed):
    """Scan-to-scan case `seed` of the sweep: two timesteps of one room of ~SWEEP_POINTS points, a start pose and the
    icp_align parameters of the reference's three call sites in turn."""
    rng = np.random.default_rng(1000 + seed)
    s0 = synth.scene_for_point_count(int(SWEEP_POINTS * 0.84), seed=seed, timestep=0)
    s1 = synth.scene_for_point_count(int(SWEEP_POINTS * 0.84), seed=seed, timestep=1)
    max_dist, max_angle = [(0.10, 60.0), (0.075, 50.0), (0.05, 10.0)][seed % 3]
    T0 = synth.perturbed_pose(I4, rng, [0.005, 0.02, 0.05][(seed // 3) % 3], [0.005, 0.02, 0.04][(seed // 9) % 3])
    return s0, s1, T0, np.float32(max_dist), np.float32(np.deg2rad(max_angle))


def more_stop_case(seed):
    """The stop-test-on icp_align case of room `seed`: parameters of the reference's three call sites in turn
     and a start pose whose size cycles independently."""
    rng = np.random.default_rng(7000 + seed)
    max_dist, max_angle = [(0.10, 60.0), (0.075, 50.0), (0.05, 10.0)][seed % 3]
    T0 = synth.perturbed_pose(I4, rng, [0.005, 0.015, 0.03][(seed // 2) % 3], [0.005, 0.01, 0.03][(seed // 3) % 3])
    return T0, np.float32(max_dist), np.float32(np.deg2rad(max_angle))


def gen_bench(R, O, seed=11, more=False, centre=False, t0=0):
    """centre: bench.py --centre's inputs (the world moved so that the scan's median point is the origin: coordinates of both signs, the
    regime real scans are in) -> bench_seed<seed>_centre.npz; t0: the pair (t0, t0 + 1) of bench.py --timesteps -> bench_seed<seed>_t<t0>.npz."""
    import bench
    t = time.time()
    w = bench.build_inputs(1_000_000, seed=seed, centre=centre, t0=t0)
    s0, s1 = w["s0"], w["s1"]
    print(f"inputs: {w['n_scan0']} / {w['n_scan1']} scan points, {w['n_obj']} object points ({time.time()-t:.1f} s)", flush=True)
    t = time.time()
    T, err, done, nc, errs = ref_iterate(R, s1["points"], s1["normals"], s0["points"], s0["normals"], w["icp_T0"], I4,
                                         0.10, np.float32(np.deg2rad(60.0)), bench.ICP_ITERS, 0)
    print(f"icp: {done} iterations, err {err}, n_corrs {nc.tolist()} ({time.time()-t:.1f} s)", flush=True)
    assert done == bench.ICP_ITERS
    t = time.time()
    op, on = w["obj_score_np"]
    scores = R.alignment_scores(s1["points"], s1["normals"], op, on, w["score_poses"], 64)
    print(f"scores: mean {scores.mean():.4f} ({time.time()-t:.1f} s)", flush=True)
    t = time.time()
    lab = ref_labels(w)
    print(f"labels: {int((lab['labels'] > 0).sum())} labelled ({time.time()-t:.1f} s)", flush=True)
    extra = dict(labels=lab["labels"])
    if more:
        t = time.time()
        T0s, md, ma = more_stop_case(seed)
        e, Ts, _ = R.icp_align(s1["points"], s1["normals"], s0["points"], s0["normals"], T0s, I4, md, ma)
        Ts2, e2, sdone, _, _ = ref_iterate(R, s1["points"], s1["normals"], s0["points"], s0["normals"], T0s, I4, md, ma, 100, 1)
        assert (Ts2 == Ts).all() and np.float32(e) == e2, "ref_icp_iterate (stop test on) must BE icp_align"
        print(f"icp_align (stop test on): r {md:.3f} iters {sdone} err {e:.6f} ({time.time()-t:.1f} s)", flush=True)
        extra = dict(labels_sha=sha(lab["labels"]), labels_sample=lab["labels"][::257].copy(), n_labelled=int((lab["labels"] > 0).sum()),
                     stop_T0=T0s, stop_pose=Ts, stop_err=np.float32(e), stop_iters=np.int32(sdone), stop_params=np.array([md, ma], np.float32))
    if centre or t0:      # (the further fixtures keep the labels as a digest + sample, like --more's)
        extra = {k: v for k, v in extra.items() if k != "labels"}
        extra.update(labels_sha=sha(lab["labels"]), labels_sample=lab["labels"][::257].copy(), n_labelled=int((lab["labels"] > 0).sum()))
    np.savez_compressed(
        os.path.join(OUT, "bench_seed%d%s%s.npz" % (seed, "_centre" if centre else "", "_t%d" % t0 if t0 else "")), **extra,
        n_points=1_000_000, seed=seed, n_scan0=w["n_scan0"], n_scan1=w["n_scan1"], n_obj=w["n_obj"],
        in_sha=np.array([sha(s0["points"]), sha(s0["normals"]), sha(s1["points"]), sha(s1["normals"]), sha(op), sha(on),
                         sha(w["score_poses"]), sha(w["plc_poses"]), sha(w["icp_T0"])]),
        icp_T0=w["icp_T0"], icp_pose=T, icp_err=err, icp_n_corrs=nc, icp_errs=errs,
        scores=scores, **label_fields(lab))


LABEL_SOURCE = "reference: rs_pointcloud_filters.cpp:738-879 via oracle/_ref/libref_filters.so"


def ref_labels(w):
    """The headline's label transfer (8 placements of ~50 k-point models against the ~0.98 M-point scan, radius 0.05) by the
    reference's own loops."""
    objs = [dict(pos=p["np"][0], nor=p["np"][1], class_idx=p["cls"], is_static=0) for p in w["plc"]]
    plcs = [dict(pose=p["pose"], object_idx=k, uidx=k) for k, p in enumerate(w["plc"])]
    RF = RefFilters(synth.CLASS_IDX)
    lab = RF.arrangement_to_labels(w["s1"]["points"], w["s1"]["normals"], objs, plcs, 0.05, 0, synth.CLASS_IDX["unlabelled"])
    RF.close()
    return lab


def label_fields(lab):
    return dict(labels_source=LABEL_SOURCE, order=lab["order"], min_dists_sha=sha(lab["min_dists"]),
                min_dists_sample=lab["min_dists"][::257].copy(), class_ids_sha=sha(lab["class_ids"]),
                instance_ids_sha=sha(lab["instance_ids"]))


def restrong(seed=11):
    """--strong-only: the ICP units of `bench.py --scaling strong` — each placement's ~50 k-point model refined against the scan with
    rsdb_refine_alignment_of_objects_to_scene's parameters (lib/rs/rs_database.h:220-230: 0.075, 50 deg), ten fixed iterations — by the
    reference (ref_icp_iterate, stop test off), added to bench_seed<seed>.npz; everything else is kept."""
    import bench
    path = os.path.join(OUT, "bench_seed%d.npz" % seed)
    g = dict(np.load(path))
    w = bench.build_inputs(1_000_000, seed=seed)
    R = Ref()
    si = w["strong_icp"]
    poses, errs = [], []
    for k, p in enumerate(w["plc"][:bench.N_PLACEMENTS]):
        t = time.time()
        T, err, done, nc, _ = ref_iterate(R, p["np"][0], p["np"][1], w["s1"]["points"], w["s1"]["normals"], si["T0s"][k], I4,
                                          np.float32(si["max_dist"]), np.float32(si["max_angle"]), bench.ICP_ITERS, 0)
        assert done == bench.ICP_ITERS
        poses.append(T); errs.append(err)
        print(f"strong icp unit {k}: {len(p['np'][0])} source points, err {err:.6f}, n_corrs {nc.tolist()} ({time.time()-t:.1f} s)", flush=True)
    g.update(strong_icp_pose=np.stack(poses), strong_icp_err=np.array(errs, np.float32),
             strong_icp_sha=np.array([sha(p["np"][0]) for p in w["plc"][:bench.N_PLACEMENTS]] + [sha(si["T0s"])]))
    np.savez_compressed(path, **g)


def gen_units(seed=11, units=8):
    """--units: tests/golden/bench_seed<seed>_units.npz — what the WEAK-scaling line of bench.py --gpus N computes beyond the first rank's
    unit list (bench.build_inputs( ..., units = N ): per further unit one more ICP start pose, 256 more score poses, 8 more placements of
    the same scene), by the reference build: the ten fixed iterations of every start pose, every score, and — per N in {2, 4, 8} — the
    label transfer of the 8 N placements (digests + a strided sample).  The unit lists are prefixes of one another, so one fixture
    serves every N <= units; the labels depend on the whole arrangement and are kept per N."""
    import bench
    from multiprocessing.pool import ThreadPool
    w = bench.build_inputs(1_000_000, seed=seed, units=units)
    s0, s1 = w["s0"], w["s1"]
    R = Ref()
    t = time.time()

    def one(u):
        Ru = Ref()      # (ctypes releases the GIL inside the call; every thread its own handle)
        T, err, done, nc, errs = ref_iterate(Ru, s1["points"], s1["normals"], s0["points"], s0["normals"], w["icp_T0s"][u], I4,
                                             0.10, np.float32(np.deg2rad(60.0)), bench.ICP_ITERS, 0)
        assert done == bench.ICP_ITERS
        print(f"unit {u}: icp err {err}, n_corrs {nc.tolist()} ({time.time()-t:.1f} s)", flush=True)
        return T, err
    res = ThreadPool(min(units, 7)).map(one, range(units))
    poses = np.stack([r[0] for r in res]); errs = np.array([r[1] for r in res], np.float32)
    g0 = dict(np.load(os.path.join(OUT, "bench_seed%d.npz" % seed)))
    assert (poses[0] == g0["icp_pose"]).all(), "unit 0 must be the headline fixture's problem"
    op, on = w["obj_score_np"]
    t = time.time()
    scores = R.alignment_scores(s1["points"], s1["normals"], op, on, w["score_poses"], 64)
    assert (scores[:bench.N_POSES] == g0["scores"]).all()
    print(f"scores: {len(scores)} poses ({time.time()-t:.1f} s)", flush=True)
    out = dict(seed=seed, n_points=1_000_000, units=units, icp_T0s=w["icp_T0s"], icp_pose=poses, icp_err=errs, scores=scores,
               in_sha=np.array([sha(s0["points"]), sha(s1["points"]), sha(w["icp_T0s"]), sha(w["score_poses"]), sha(w["plc_poses"])]),
               labels_source=LABEL_SOURCE)
    for n in (2, 4, 8):
        if n > units:
            continue
        t = time.time()
        plc = w["plc"][:n * bench.N_PLACEMENTS]
        objs = [dict(pos=p["np"][0], nor=p["np"][1], class_idx=p["cls"], is_static=0) for p in plc]
        plcs = [dict(pose=p["pose"], object_idx=k, uidx=k) for k, p in enumerate(plc)]
        RF = RefFilters(synth.CLASS_IDX)
        lab = RF.arrangement_to_labels(s1["points"], s1["normals"], objs, plcs, 0.05, 0, synth.CLASS_IDX["unlabelled"])
        RF.close()
        print(f"labels, {len(plc)} placements: {int((lab['labels'] > 0).sum())} labelled ({time.time()-t:.1f} s)", flush=True)
        out.update({f"labels_sha_u{n}": sha(lab["labels"]), f"labels_sample_u{n}": lab["labels"][::257].copy(), f"min_dists_sha_u{n}": sha(lab["min_dists"]),
                    f"order_u{n}": lab["order"], f"n_labelled_u{n}": int((lab["labels"] > 0).sum())})
    np.savez_compressed(os.path.join(OUT, "bench_seed%d_units.npz" % seed), **out)


def static_arrangement(w, seed):
    """An arrangement of the headline's size WITH static placements, for the label transfer's second pass (rs_pointcloud_filters.cpp:
    841-848: 1.5 x radius, shared min_dists) and its ordering (:724-736,823-835): the bench's eight ~50 k-point dynamic placements
    plus three static objects cut from the scan itself — a third of the floor, half of one wall, a quarter of the other (0.1 - 0.3 M
    points each: "static wall/floor objects up to scan-sized") — shuffled.  Returns (objects, placements) in the
    oracle's dict form; static objects carry `sub`, their indices into scan 1."""
    rng = np.random.default_rng(9100 + seed)
    s1 = w["s1"]
    objs = [dict(pos=p["np"][0], nor=p["np"][1], class_idx=p["cls"], is_static=0) for p in w["plc"][:8]]
    plcs = [dict(pose=p["pose"], object_idx=k, uidx=10 + k) for k, p in enumerate(w["plc"][:8])]
    inst = s1["instance_idx"]
    for cls_name, which, step in (("floor", 0, 3), ("wall", 1, 2), ("wall", 2, 4)):
        idx = np.nonzero(inst == which)[0]
        sub = np.sort(rng.permutation(idx)[::step]).astype(np.int32)
        objs.append(dict(pos=np.ascontiguousarray(s1["points"][sub]), nor=np.ascontiguousarray(s1["normals"][sub]),
                         class_idx=synth.CLASS_IDX[cls_name], is_static=1, sub=sub))
        plcs.append(dict(pose=I4 if which != 2 else synth.perturbed_pose(I4, rng, 0.003, 0.002), object_idx=len(objs) - 1, uidx=100 + which))
    order = rng.permutation(len(plcs))
    return objs, [plcs[i] for i in order]


def gen_static_labels(seed=11):
    """--static-labels: bench_labels_static_seed<seed>.npz — the reference's label loops (oracle/_ref/libref_filters.so) on
    static_arrangement(): digests of labels / min_dists / class ids / instance ids, the visiting order, and the arrangement itself."""
    import bench
    w = bench.build_inputs(1_000_000, seed=seed)
    objs, plcs = static_arrangement(w, seed)
    t = time.time()
    RF = RefFilters(synth.CLASS_IDX)
    out = {}
    for prio in (0, 1):
        lab = RF.arrangement_to_labels(w["s1"]["points"], w["s1"]["normals"], objs, plcs, 0.05, prio, synth.CLASS_IDX["unlabelled"])
        out[prio] = lab
        print(f"static arrangement, prioritize_static {prio}: {int((lab['labels'] > 0).sum())} of {len(lab['labels'])} labelled, order {lab['order'].tolist()} ({time.time()-t:.1f} s)", flush=True)
    RF.close()
    np.savez_compressed(
        os.path.join(OUT, "bench_labels_static_seed%d.npz" % seed), labels_source=LABEL_SOURCE, seed=seed, n_points=1_000_000,
        scan_sha=sha(w["s1"]["points"]), n_obj=len(objs), n_plc=len(plcs),
        **{f"obj{i}_sub_sha": sha(o["sub"]) for i, o in enumerate(objs) if "sub" in o},
        obj_class=np.array([o["class_idx"] for o in objs], np.int32), obj_static=np.array([o["is_static"] for o in objs], np.int32),
        plc_pose=np.stack([np.asarray(p["pose"], np.float32) for p in plcs]), plc_obj=np.array([p["object_idx"] for p in plcs], np.int32),
        plc_uidx=np.array([p["uidx"] for p in plcs], np.int32),
        **{f"{k}_prio{pr}": (v["order"] if k == "order" else sha(v[k])) for pr, v in out.items() for k in ("order", "labels", "min_dists", "class_ids", "instance_ids")},
        n_labelled=np.array([int((out[pr]["labels"] > 0).sum()) for pr in (0, 1)]))


def relabel(seed):
    """--labels-only: the label fields of bench_seed<seed>.npz again, from the reference build; everything else is kept."""
    import bench
    path = os.path.join(OUT, "bench_seed%d.npz" % seed)
    g = dict(np.load(path))
    w = bench.build_inputs(1_000_000, seed=seed)
    op, on = w["obj_score_np"]
    got = [sha(w["s0"]["points"]), sha(w["s0"]["normals"]), sha(w["s1"]["points"]), sha(w["s1"]["normals"]), sha(op), sha(on),
           sha(w["score_poses"]), sha(w["plc_poses"]), sha(w["icp_T0"])]
    assert got == [str(x) for x in g["in_sha"]], "the generator no longer produces the inputs the fixture was made for"
    t = time.time()
    lab = ref_labels(w)
    had = (str(g["labels_sha"]) if "labels_sha" in g else sha(g["labels"]), str(g["min_dists_sha"]))
    now = (sha(lab["labels"]), sha(lab["min_dists"]))
    print(f"seed {seed}: {int((lab['labels'] > 0).sum())} labelled ({time.time()-t:.1f} s); the fixture's earlier digests "
          f"{'ARE' if had == now else 'are NOT'} the reference's", flush=True)
    if "labels_sha" in g:
        g.update(labels_sha=sha(lab["labels"]), labels_sample=lab["labels"][::257].copy(), n_labelled=int((lab["labels"] > 0).sum()))
    else:
        g.update(labels=lab["labels"])
    g.update(label_fields(lab))
    np.savez_compressed(path, **g)


def gen_sweep(R):
    poses, errs, iters, T0s, params, sizes, digests = [], [], [], [], [], [], []
    for seed in SWEEP_SEEDS:
        t = time.time()
        s0, s1, T0, md, ma = sweep_inputs(seed)
        e, T, _ = R.icp_align(s1["points"], s1["normals"], s0["points"], s0["normals"], T0, I4, md, ma)
        T2, e2, done, _, _ = ref_iterate(R, s1["points"], s1["normals"], s0["points"], s0["normals"], T0, I4, md, ma, 100, 1)
        assert (T2 == T).all() and np.float32(e) == e2, "ref_icp_iterate (stop test on) must BE icp_align"
        poses.append(T); errs.append(np.float32(e)); iters.append(done); T0s.append(T0); params.append([md, ma])
        sizes.append([len(s1["points"]), len(s0["points"])]); digests.append(sha(s1["points"]) + sha(s0["points"]))
        print(f"sweep seed {seed:2d}: n {len(s1['points'])} r {md:.3f} iters {done} err {e:.6f} ({time.time()-t:.1f} s)", flush=True)
    np.savez_compressed(os.path.join(OUT, "sweep_icp.npz"), seeds=np.array(SWEEP_SEEDS), n_points=SWEEP_POINTS,
                        pose=np.stack(poses), err=np.array(errs, np.float32), iters=np.array(iters, np.int32),
                        T0=np.stack(T0s), params=np.array(params, np.float32), sizes=np.array(sizes), in_sha=np.array(digests))


if __name__ == "__main__":
    build(ref=True)
    if "--static-labels" in sys.argv:
        gen_static_labels(11)
        sys.exit(0)
    if "--units" in sys.argv:
        gen_units(11, 8)
        sys.exit(0)
    if "--strong-only" in sys.argv:
        restrong(11)
        sys.exit(0)
    if "--labels-only" in sys.argv:
        for seed in BENCH_SEEDS + MORE_SEEDS:
            if "--seed" not in sys.argv or str(seed) == sys.argv[sys.argv.index("--seed") + 1]:
                relabel(seed)
        sys.exit(0)
    R, O = Ref(), Oracle()
    if "--centre" in sys.argv or "--pair" in sys.argv:      # --centre [--seed N] | --pair T0 [--seed N]
        seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 11
        gen_bench(R, O, seed, more="--centre" in sys.argv, centre="--centre" in sys.argv, t0=int(sys.argv[sys.argv.index("--pair") + 1]) if "--pair" in sys.argv else 0)
        sys.exit(0)
    if "--more" in sys.argv:
        for seed in MORE_SEEDS:
            if "--seed" not in sys.argv or str(seed) == sys.argv[sys.argv.index("--seed") + 1]:
                gen_bench(R, O, seed, more=True)
        sys.exit(0)
    if "--sweep-only" not in sys.argv:
        for seed in BENCH_SEEDS:
            if "--seed" not in sys.argv or str(seed) == sys.argv[sys.argv.index("--seed") + 1]:
                gen_bench(R, O, seed)
    if "--bench-only" not in sys.argv:
        gen_sweep(R)
