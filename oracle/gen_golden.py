"""TEST INFRASTRUCTURE.  Generates tests/golden/*.npz from oracle/_ref — the real reference
compiled in place from /root/reference (run in the build container only; the fixtures are
committed so the GPU box, which has no /root/reference, can check against them).

Each fixture holds the inputs and the reference's outputs for one hot-path function:

  rows_*.npz     msh_hash_grid_radius_search rows (lib/msh/msh_hash_grid.h:1090-1259)
  corrs_*.npz    one icp_find_corrs call (lib/rs/icp.h:306-412)
  icp_*.npz      icp_align final pose + error (lib/rs/icp.h:416-500) for the three call-site
                 parameter sets (pose_proposal main.cpp:195, rs_database.h:227, database_update.cpp:65)
  scores_*.npz   mgs_compute_object_alignment_score for a list of poses (pose_proposal.cpp:93-158)
  mat4.npz       msh_mat4_inverse / msh_mat4_mul / msh_translate / msh_rotate samples
  gates.npz      accept/reject of the label normal gate over a sweep of dot values
  labels_*.npz   rspf_arrangement_to_labels + rspf__assign_temporary_labels (rs_pointcloud_filters.cpp:738-879) by the
                 REFERENCE's own text: oracle/_ref/libref_filters.so = that file's lines 1-14 + 16-879 (everything but
                 the include of the un-vendored gco header and rspf_smooth_labels, its one user), oracle/Makefile
  edge_cost.npz  the neighbourhood edge weight (rs_pointcloud_filters.cpp:706-708) from the
                 reference-toolchain TU oracle/ref_label_gate.cpp
  neighborhood_*.npz  rspf_compute_neighborhood (:674-722) — by the same reference build, sorted by pair key
  coverage.npz   rsao__compute_scene_coverage_score + grids (arrangement_optimization.cpp:344-373,1064-1106) from the
                 reference TU compiled in place (oracle/_ref/libref_ao.so)

Usage:  python oracle/gen_golden.py                       (everything)
        python oracle/gen_golden.py --neighborhood-only   (adds those two without rewriting the rest)
        python oracle/gen_golden.py --labels-only         (labels_*.npz again, from the inputs of scene.npz)
        python oracle/gen_golden.py --coverage-only       (likewise)
        python oracle/gen_golden.py --level-only          (likewise: level.npz)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, Ref, RefFilters, build, edge_digest  # noqa: E402
from rescan_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
I4 = np.eye(4, dtype=np.float32).ravel()


def scene_small(seed=7, density=1200.0, timestep=1):
    return synth.make_scene(seed=seed, density=density, timestep=timestep)


def main():
    build(ref=True)
    R = Ref()
    O = Oracle()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(2024)
    s = scene_small()
    pts, nor = s["points"], s["normals"]
    # the shared inputs are stored once: scene scan + the objects' model clouds and poses
    np.savez_compressed(os.path.join(OUT, "scene.npz"), points=pts, normals=nor, instance_idx=s["instance_idx"],
                        class_idx=s["class_idx"], n_obj=len(s["objects"]),
                        **{f"obj{i}_pos": o["pos"] for i, o in enumerate(s["objects"])},
                        **{f"obj{i}_nor": o["nor"] for i, o in enumerate(s["objects"])},
                        obj_pose=np.stack([o["pose"] for o in s["objects"]]),
                        obj_class=np.array([o["class_idx"] for o in s["objects"]], np.int32),
                        obj_uidx=np.array([o["uidx"] for o in s["objects"]], np.int32))

    # ---- radius-search rows --------------------------------------------------------
    cases = [("k1_r005", 0.05, 1, 0.05), ("k16_r010", 0.10, 16, 0.10), ("k16_r006", 0.10, 16, 0.06),
             ("k64_r010", 0.05, 64, 0.10), ("k8_r005", 0.05, 8, 0.05), ("k1_r0075", 0.05, 1, 0.075)]
    for name, grid_radius, k, r in cases:
        q = pts[rng.integers(0, len(pts), 1500)] + rng.normal(0, 0.02, (1500, 3)).astype(np.float32)
        q[:40] += 5.0                      # outside the grid
        q[40:80] -= np.float32(0.3)        # straddling the lower corner (negative grid coordinates)
        g = R.grid_create(pts, grid_radius)
        d, i, nn, tot = R.radius_search(g, q, r, k, 1)
        R.grid_destroy(g)
        np.savez_compressed(os.path.join(OUT, f"rows_{name}.npz"), query=q.astype(np.float32),
                            grid_radius=np.float32(grid_radius), radius=np.float32(r), k=k,
                            dists=d, inds=i, nn=nn, total=tot)

    # ---- ICP -----------------------------------------------------------------------
    params = [("pp", 0.10, 60.0), ("refine", 0.075, 50.0), ("fuse", 0.05, 10.0)]
    for oi, o in enumerate(s["objects"]):
        T0 = synth.perturbed_pose(o["pose"], rng)
        for pname, md, deg in params:
            ma = np.float32(np.deg2rad(np.float32(deg)))
            c = R.icp_find_corrs(o["pos"], o["nor"], pts, nor, T0, I4, md, ma)
            np.savez_compressed(os.path.join(OUT, f"corrs_{o['kind']}{oi}_{pname}.npz"),
                                obj=oi, T1=T0, T2=I4,
                                max_dist=np.float32(md), max_angle=ma,
                                c_pts1=c[0], c_nor1=c[1], c_pts2=c[2], c_nor2=c[3], weights=c[4])
            err, T, _ = R.icp_align(o["pos"], o["nor"], pts, nor, T0, I4, md, ma)
            _, _, iters = O.icp_align(o["pos"], o["nor"], pts, nor, T0, I4, md, ma)
            np.savez_compressed(os.path.join(OUT, f"icp_{o['kind']}{oi}_{pname}.npz"),
                                obj=oi, T1=T0, T2=I4,
                                max_dist=np.float32(md), max_angle=ma, err=np.float32(err), T_out=T, iters=iters)

    # ---- scores --------------------------------------------------------------------
    for oi, o in enumerate(s["objects"]):
        poses = np.stack([synth.perturbed_pose(o["pose"], rng, 0.5, 0.15) for _ in range(30)] + [o["pose"]] +
                         [synth.pose_matrix(a, (1.0 + 0.3 * a, 0.0, 2.0)) for a in (0.0, 1.0, 2.0)])
        for K in (64, 32):
            sc = R.alignment_scores(pts, nor, o["pos"], o["nor"], poses, K)
            np.savez_compressed(os.path.join(OUT, f"scores_{o['kind']}{oi}_k{K}.npz"), obj=oi, poses=poses, k=K, scores=sc)

    # ---- matrix helpers ------------------------------------------------------------
    ms = rng.normal(size=(64, 16)).astype(np.float32)
    ms[:8] = [synth.pose_matrix(a, rng.normal(size=3)) for a in rng.uniform(0, 6.28, 8)]
    bs = rng.normal(size=(64, 16)).astype(np.float32)
    ts = rng.normal(size=(64, 3)).astype(np.float32)
    ang = rng.normal(size=64).astype(np.float32)
    axes = np.tile(np.eye(3, dtype=np.float32), (22, 1))[:64]
    np.savez_compressed(os.path.join(OUT, "mat4.npz"), m=ms, b=bs, t=ts, angle=ang, axis=axes,
                        inverse=np.stack([R.mat4_inverse(m) for m in ms]),
                        mul=np.stack([R.mat4_mul(m, b) for m, b in zip(ms, bs)]),
                        translate=np.stack([R.translate(m, t) for m, t in zip(ms, ts)]),
                        rotate=np.stack([R.rotate(m, a, ax) for m, a, ax in zip(ms, ang, axes)]))

    # ---- label gate ----------------------------------------------------------------
    c70 = np.float32(np.cos(np.deg2rad(70.0)))
    xs = np.concatenate([np.linspace(-1.1, 1.1, 401).astype(np.float32),
                         np.array([np.nextafter(c70, np.float32(0)), c70, np.nextafter(c70, np.float32(1))], np.float32)])
    for _ in range(12):
        xs = np.concatenate([xs, np.nextafter(xs[-3:], np.float32(1)), np.nextafter(xs[-6:-3], np.float32(0))])
    np.savez_compressed(os.path.join(OUT, "gates.npz"), dot=xs,
                        label_accept=np.array([R.label_gate_dot(x) for x in xs], np.int8))

    gen_labels(s)

    gen_neighborhood(O, R, s)
    gen_coverage(s)
    gen_level(s)

    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"wrote {len(os.listdir(OUT))} fixtures, {total/1e6:.2f} MB")


LABEL_CASES = (          # tag, static placements, prioritize_static, every dynamic placement three times (one an exact copy)
    ("mixed", ("floor", "wall"), 0, False),
    ("nostatic", (), 0, False),                          # first_static = 0: one pass at 1.5 * radius (:830-848)
    ("prio", ("floor", "wall"), 1, False),               # prioritize_static: min_dists reset, same radius (:841-845)
    ("ties", ("floor",), 0, True),                       # equal distances between placements: the earlier one wins (strict <)
    ("multistatic", ("floor", "wall", "wall", "floor"), 0, True),
)


def gen_labels(s):
    """labels_*.npz from the reference's own label loops (oracle/_ref/libref_filters.so, RefFilters): the temporary labels and
    min_dists of rspf__assign_temporary_labels, the class / instance ids of rspf_arrangement_to_labels.  Every case has its
    own generator, so that cases can be added without changing the others."""
    pts, nor = s["points"], s["normals"]
    for ci, (tag, statics, prio, dup) in enumerate(LABEL_CASES):
        rng = np.random.default_rng(4100 + ci)
        objs, plcs = [], []
        for oi, o in enumerate(s["objects"]):
            objs.append(dict(pos=o["pos"], nor=o["nor"], class_idx=o["class_idx"], is_static=0))
            plcs.append(dict(pose=synth.perturbed_pose(o["pose"], rng, 0.02, 0.01), object_idx=oi, uidx=o["uidx"]))
            if dup:
                plcs.append(dict(pose=synth.perturbed_pose(o["pose"], rng, 0.01, 0.005), object_idx=oi, uidx=40 + o["uidx"]))
                plcs.append(dict(pose=plcs[-2]["pose"], object_idx=oi, uidx=80 + o["uidx"]))
        for k, cls_name in enumerate(statics):
            m = s["instance_idx"] == (0 if cls_name == "floor" else 1)
            sub = rng.permutation(np.nonzero(m)[0])[::2]
            objs.append(dict(pos=pts[sub], nor=nor[sub], class_idx=synth.CLASS_IDX[cls_name], is_static=1, sub=sub))
            plcs.append(dict(pose=I4 if k < 2 else synth.perturbed_pose(I4, rng, 0.004, 0.002), object_idx=len(objs) - 1, uidx=120 + k))
        order = rng.permutation(len(plcs))
        plcs = [plcs[i] for i in order]
        RF = RefFilters(synth.CLASS_IDX)
        res = RF.arrangement_to_labels(pts, nor, objs, plcs, 0.05, prio, synth.CLASS_IDX["unlabelled"])
        RF.close()
        np.savez_compressed(
            os.path.join(OUT, f"labels_{tag}.npz"), source="reference: rs_pointcloud_filters.cpp:738-879 via oracle/_ref/libref_filters.so",
            n_obj=len(objs), n_plc=len(plcs), prioritize_static=prio,
            # objects beyond the scene's own are subsets of the scan, stored as index lists
            **{f"obj{i}_sub": o["sub"].astype(np.int32) for i, o in enumerate(objs) if "sub" in o},
            obj_class=np.array([o["class_idx"] for o in objs], np.int32),
            obj_static=np.array([o["is_static"] for o in objs], np.int32),
            plc_pose=np.stack([np.asarray(p["pose"], np.float32) for p in plcs]),
            plc_obj=np.array([p["object_idx"] for p in plcs], np.int32),
            plc_uidx=np.array([p["uidx"] for p in plcs], np.int32),
            labels=res["labels"], min_dists=res["min_dists"], order=res["order"],
            class_ids=res["class_ids"], instance_ids=res["instance_ids"])
        print(f"labels_{tag}: {len(plcs)} placements, {int((res['labels'] > 0).sum())} of {len(pts)} labelled, order {res['order'].tolist()}")


def by_pair_key(a, b, w, n):
    """The reference returns its edges in hash-table order (insertion order); fixtures keep them sorted by the pair key."""
    o = np.argsort(np.maximum(a, b).astype(np.int64) * n + np.minimum(a, b), kind="stable")
    return a[o], b[o], w[o]


def gen_neighborhood(O, R, s):
    """edge_cost.npz: rs_pointcloud_filters.cpp:706-708 evaluated by the reference-toolchain TU
    (ref_label_gate.cpp: same math.h preamble).  neighborhood_*.npz: rspf_compute_neighborhood by the REFERENCE's own text
    (oracle/_ref/libref_filters.so), full edge lists for the object clouds (below the int32 key wrap, so the pair key is
    unique) and a digest for the scan."""
    rng = np.random.default_rng(77)
    d2 = np.concatenate([rng.uniform(0, 0.0025, 4000), [0.0, 0.0025, 0.0024999, 1e-12, 0.01]]).astype(np.float32)
    dot = np.concatenate([rng.uniform(-0.2, 1.1, 4000), [1.0, 0.0, -1.0, 0.5, 0.999999]]).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "edge_cost.npz"), d2=d2, dot=dot,
                        cost=np.array([R.edge_cost(a, b) for a, b in zip(d2, dot)], np.float32))
    RF = RefFilters(synth.CLASS_IDX)
    src = "reference: rs_pointcloud_filters.cpp:674-722 via oracle/_ref/libref_filters.so"
    for oi, o in enumerate(s["objects"]):
        a, b, w = by_pair_key(*RF.compute_neighborhood(o["pos"], o["nor"]), len(o["pos"]))
        np.savez_compressed(os.path.join(OUT, f"neighborhood_obj{oi}.npz"), source=src, obj=oi, idx1=a, idx2=b, weight=w)
    a, b, w = RF.compute_neighborhood(s["points"], s["normals"])
    np.savez_compressed(os.path.join(OUT, "neighborhood_scene.npz"), source=src, digest=edge_digest(a, b, w))
    RF.close()


def gen_coverage(s):
    """coverage.npz: the scene-coverage term by the REFERENCE itself (oracle/_ref/libref_ao.so =
    arrangement_optimization.cpp compiled in place): grid geometry, the rasterised scene grid and the
    scores of 24 arrangements (perturbed object poses, subsets, one static placement)."""
    from oracle.pyoracle import RefAO
    pts = s["points"]
    bmin, bmax = pts.min(0), pts.max(0)
    rng = np.random.default_rng(99)
    quality = rng.uniform(0.3, 1.0, len(pts)).astype(np.float32)        # some points fall below the 0.5 threshold
    R = RefAO(synth.CLASS_IDX, pts, bmin, bmax, quality=quality)
    objs = s["objects"]
    idx = [R.add_object(o["pos"], o["class_idx"], o["uidx"]) for o in objs]
    wall = pts[s["instance_idx"] == 1][::3].copy()
    idx.append(R.add_object(wall, synth.CLASS_IDX["wall"], 1))          # a static object: must be skipped
    arr_obj, arr_pose, arr_first, scores = [], [], [0], []
    for a in range(24):
        members = [k for k in range(len(idx)) if rng.uniform() < 0.7] or [0]
        poses = [synth.perturbed_pose(objs[k]["pose"], rng, 0.02 * (a % 5), 0.01 * (a % 3)) if k < len(objs) else I4 for k in members]
        scores.append(R.coverage([idx[k] for k in members], poses))
        arr_obj += members; arr_pose += poses; arr_first.append(len(arr_obj))
    np.savez_compressed(os.path.join(OUT, "coverage.npz"), bbox_min=bmin, bbox_max=bmax, quality=quality, wall=wall,
                        res=R.res, origin=R.origin, n_cells=R.n_cells, scene_grid=np.packbits(R.scene_grid()),
                        arr_obj=np.array(arr_obj, np.int32), arr_pose=np.array(arr_pose, np.float32).reshape(-1, 16),
                        arr_first=np.array(arr_first, np.int32), scores=np.array(scores, np.float32))


def gen_level(s):
    """level.npz: rs_pointcloud__compute_level_poisson by the REFERENCE itself (oracle/_ref/libref_ao.so carries
    rs_pointcloud.h, oracle/ref_ao_driver.cpp: ref_level_poisson) on the golden scene's points, levels 1-4, in the
    scene's own point order and in a raster (z, y, x) order; the fixture stores the permutation and the sample indices."""
    from oracle.pyoracle import ref_level_poisson
    pts = s["points"]
    raster = np.lexsort((pts[:, 0], pts[:, 1], pts[:, 2])).astype(np.int32)
    out = dict(raster=raster)
    for level in (1, 2, 3, 4):
        out[f"own_l{level}"] = ref_level_poisson(pts, level)
        out[f"raster_l{level}"] = ref_level_poisson(np.ascontiguousarray(pts[raster]), level)
    np.savez_compressed(os.path.join(OUT, "level.npz"), **out)


def main_level_only():
    d = dict(np.load(os.path.join(OUT, "scene.npz")))
    gen_level(dict(points=d["points"]))


def main_coverage_only():
    d = dict(np.load(os.path.join(OUT, "scene.npz")))
    s = dict(points=d["points"], normals=d["normals"], instance_idx=d["instance_idx"],
             objects=[dict(pos=d[f"obj{i}_pos"], nor=d[f"obj{i}_nor"], pose=d["obj_pose"][i], class_idx=int(d["obj_class"][i]),
                           uidx=int(d["obj_uidx"][i])) for i in range(int(d["n_obj"]))])
    gen_coverage(s)


def scene_from_fixture():
    d = dict(np.load(os.path.join(OUT, "scene.npz")))
    return dict(points=d["points"], normals=d["normals"], instance_idx=d["instance_idx"],
                objects=[dict(pos=d[f"obj{i}_pos"], nor=d[f"obj{i}_nor"], pose=d["obj_pose"][i], class_idx=int(d["obj_class"][i]),
                              uidx=int(d["obj_uidx"][i])) for i in range(int(d["n_obj"]))])


def main_neighborhood_only():
    """Adds the neighbourhood fixtures without rewriting the others (zip timestamps would churn them)."""
    from oracle.pyoracle import Oracle, Ref
    d = dict(np.load(os.path.join(OUT, "scene.npz")))
    s = dict(points=d["points"], normals=d["normals"],
             objects=[dict(pos=d[f"obj{i}_pos"], nor=d[f"obj{i}_nor"]) for i in range(int(d["n_obj"]))])
    gen_neighborhood(Oracle(), Ref(), s)


if __name__ == "__main__":
    if "--neighborhood-only" in sys.argv:
        main_neighborhood_only()
    elif "--labels-only" in sys.argv:
        build(ref=True)
        gen_labels(scene_from_fixture())
    elif "--level-only" in sys.argv:
        main_level_only()
    elif "--coverage-only" in sys.argv:
        main_coverage_only()
    else:
        main()
