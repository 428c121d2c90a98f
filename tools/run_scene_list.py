#!/usr/bin/env python3
"""BASELINE.json configs[0] / configs[4]: a scene_list.txt batch through the reference's own binaries, end to end, with
the hot path on the CPU (the reference's build) and on the GPU (the same sources linked against librescan_dropin.so).

Follows scripts/run_segmentation_pipeline.py:62-72 and scripts/common.py:6-16 of the reference: per sequence of
scene_list.txt (one name per line), the sorted stems of <sequence>/gt_segmentation/*.ply; the first goes through
`seg2rsdb <ply> nyu40_classes.txt <seq>/<t0>.rsdb -v`, every later one through
`pose_proposal <prev.rsdb> <ply> <seq>/<t>_pp.rsdb -v`, then (round 5) `segment_transfer <t>_pp.rsdb -o <t>.rsdb -v` MINUS its
graph-cut smoothing: gco-v3.0 is not vendored, so oracle/Makefile builds the app's own text less its two gco
includes and its one call of rspf_smooth_labels — no stand-in — the reference's way and against the shim.  NOT run: the model
fusion that follows (PoissonRecon, external).  Sequences have two timesteps.

The sequences are synthetic (rescan_amd/synth.py, seeds 100, 101, ...).  Builds compared (oracle/Makefile, all from the
unmodified sources under /root/reference, prebuilt in the build container):
    ref     _ref/pose_proposal          as shipped: one thread
    omp     _ref/pose_proposal_omp      the reference's optional OpenMP path, all host cores
    icp     _ref/pose_proposal_hip      shadow/icp: icp_align on the GPU
    grid    _ref/pose_proposal_hip2     shadow/icp + shadow/grid: also the app's score loop, level builder and level grids
                                        on the shim's msh_hash_grid_* (device kernels for batched searches)
    batched _ref/pose_proposal_hip3     grid + the app's grid search (mgs_propose_poses) bound to shadow/apps/pose_proposal_batched.cpp:
                                        one rsd_alignment_scores call (k_score) per object and level
--jobs J runs J sequences of a GPU build at a time on each GPU (the apps' host code is single-threaded and the GPU idles most of
a run; a box admits few GPU processes, so J <= 4).
One sequence per GPU (--gpus N: N worker processes, each bound to its device through HIP_VISIBLE_DEVICES); the CPU builds
run one sequence at a time (omp uses every core by itself).  Per sequence and build: wall-clock of the process, the app's
own "Computed poses in" (apps/pose_proposal/main.cpp:208), and the distance of its proposal .bin from the ref build's.

Usage: python tools/run_scene_list.py [--sequences 8] [--density 6400] [--gpus 1] [--out profiles/r02/scene_list.json]
"""
import argparse
import json
import os
import re
import struct
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref")
BUILDS = {"ref": "pose_proposal", "omp": "pose_proposal_omp", "icp": "pose_proposal_hip", "grid": "pose_proposal_hip2", "batched": "pose_proposal_hip3"}
GPU_BUILDS = ("icp", "grid", "batched")


def list_sequences(list_filename):
    with open(list_filename) as f:
        return [line.rstrip() for line in f if line.strip()]


def list_subsequences(folder):
    return sorted(os.path.splitext(f)[0] for f in os.listdir(folder) if os.path.splitext(f)[1] == ".ply")


def read_pose_bin(path):
    b = open(path, "rb").read()
    n = struct.unpack("<i", b[:4])[0]
    counts = struct.unpack("<%di" % n, b[4:4 + 4 * n])
    off = 4 + 4 * n
    out = []
    for c in counts:
        out.append(np.frombuffer(b[off:off + 68 * c], np.float32).reshape(c, 17))
        off += 68 * c
    return out


def run(cmd, cwd, env=None, timeout=240):
    t = time.perf_counter()
    try:
        r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=timeout, env=env)
    except subprocess.TimeoutExpired as e:
        r = subprocess.CompletedProcess(cmd, returncode=-9, stdout=(e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""),
                                        stderr="timed out after %d s" % timeout)
    return r, time.perf_counter() - t


def computed_in(stdout):
    m = re.search(r"Computed poses in\s*([0-9.]+)", stdout)
    return float(m.group(1)) if m else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sequences", type=int, default=8)
    ap.add_argument("--density", type=float, default=6400.0)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--builds", default="ref,omp,icp,grid,batched")
    ap.add_argument("--jobs", type=int, default=1, help="sequences of a GPU build in flight per GPU")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "scene_list.json"))
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--no-segment-transfer", action="store_true", help="skip the segment_transfer (minus smoothing) leg")
    args = ap.parse_args()
    from rescan_amd import synth
    builds = [b for b in args.builds.split(",") if os.path.exists(os.path.join(REF, BUILDS[b]))]
    missing = [b for b in args.builds.split(",") if b not in builds]
    work = args.workdir or tempfile.mkdtemp(prefix="scene_list_")
    names = [f"seq{100 + k}" for k in range(args.sequences)]
    n_pts = {}
    for k, name in enumerate(names):
        os.makedirs(os.path.join(work, name, "gt_segmentation"), exist_ok=True)
        for t in (0, 1):
            s = synth.make_scene(seed=100 + k, density=args.density, timestep=t)
            synth.write_ply(os.path.join(work, name, "gt_segmentation", f"t{t}.ply"), s)
            n_pts[name] = len(s["points"])
    with open(os.path.join(work, "scene_list.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    synth.write_class_table(os.path.join(work, "nyu40_classes.txt"))

    rows = []
    # stage 1: seg2rsdb for the first timestep of every sequence (its exit code is unreliable: it crashes in cleanup after
    # writing a correct file)
    for name in list_sequences(os.path.join(work, "scene_list.txt")):
        stems = list_subsequences(os.path.join(work, name, "gt_segmentation"))
        r, dt = run([os.path.join(REF, "seg2rsdb"), os.path.join(name, "gt_segmentation", stems[0] + ".ply"), "nyu40_classes.txt",
                     os.path.join(name, stems[0] + ".rsdb"), "-v"], work)
        assert os.path.exists(os.path.join(work, name, stems[0] + ".rsdb")), r.stdout[-400:] + r.stderr[-400:]
        rows.append(dict(sequence=name, stage="seg2rsdb", build="ref", wall_s=dt))
        print(f"[scene_list] {name} seg2rsdb: {dt:.1f} s", file=sys.stderr, flush=True)

    def pose_proposal(name, build, device):
        stems = list_subsequences(os.path.join(work, name, "gt_segmentation"))
        prev = os.path.join(name, stems[0] + ".rsdb")
        out = []
        for stem in stems[1:]:
            env = dict(os.environ)
            if build in GPU_BUILDS:
                env["HIP_VISIBLE_DEVICES"] = str(device)
            if build == "omp":
                # The reference's OpenMP split (msh_hash_grid.h:1119-1133) underflows `high_lim - low_lim` whenever
                # (T - 1) * ceil(n / T) > n for a call's n query points — with the 128 threads of this host that is every
                # object level of a few hundred points, and the app then loops (practically) forever.  8 threads is what
                # SURVEY.md §6 measured (4.2x, byte-identical output); a run that still hangs is cut off and reported as failed.
                env["OMP_NUM_THREADS"] = os.environ.get("RS_SCENE_LIST_OMP_THREADS", "8")
            dst = os.path.join(name, f"{stem}_pp_{build}.rsdb")
            r, dt = run([os.path.join(REF, BUILDS[build]), prev, os.path.join(name, "gt_segmentation", stem + ".ply"), dst, "-v"], work, env)
            ok = r.returncode == 0 and "[rescan_hip]" not in r.stderr
            print(f"[scene_list] {name} {build}: {dt:.1f} s (computed poses in {computed_in(r.stdout)}) ok={ok}", file=sys.stderr, flush=True)
            out.append(dict(sequence=name, stage="pose_proposal", build=build, wall_s=dt, computed_poses_s=computed_in(r.stdout), ok=ok,
                            bin=os.path.join(work, name, f"{stem}_pp_{build}", f"{stem}_pp_{build}.bin"), err=(r.stderr[-300:] if not ok else "")))
            # (the next timestep would read segment_transfer's output of this one: not available, see the docstring)
        return out

    t_total = {}
    for build in builds:
        t = time.perf_counter()
        if build in GPU_BUILDS and args.gpus * args.jobs > 1:
            with ThreadPoolExecutor(max_workers=args.gpus * min(args.jobs, 4)) as pool:           # one sequence per GPU (x --jobs)
                futs = [pool.submit(pose_proposal, name, build, k % args.gpus) for k, name in enumerate(names)]
                for f in futs:
                    rows += f.result()
        else:
            for name in names:
                rows += pose_proposal(name, build, 0)
        t_total[build] = time.perf_counter() - t

    # stage 3 (round 5): segment_transfer MINUS its graph-cut smoothing (oracle/Makefile: the reference's text less two gco includes and
    # main.cpp's one call of rspf_smooth_labels; no stand-in for gco) — the reference build on the ref build's proposals, the shim build
    # (shadow/icp + shadow/grid) on the batched build's (bit-identical) proposals: wall-clock, the app's own stage timers
    # (apps/segment_transfer/main.cpp:362-400) and whether the output .rsdb's pose lines are the reference build's.
    st_bins = {"ref": os.path.join(REF, "segment_transfer_nosmooth"), "shim": os.path.join(REF, "segment_transfer_nosmooth_hip2")}
    st_rows, st_total = [], {}
    if all(os.path.exists(b) for b in st_bins.values()) and not args.no_segment_transfer:
        def seg_transfer(name, which, device):
            stems = list_subsequences(os.path.join(work, name, "gt_segmentation"))
            src_build = "ref" if which == "ref" or "batched" not in builds else "batched"
            src = os.path.join(name, f"{stems[1]}_pp_{src_build}.rsdb")
            if not os.path.exists(os.path.join(work, src)):
                return []
            env = dict(os.environ)
            if which == "shim":
                env["HIP_VISIBLE_DEVICES"] = str(device)
            dst = os.path.join(name, f"{stems[1]}_st_{which}.rsdb")
            r, dt = run([st_bins[which], src, "-o", dst, "-v"], work, env)
            ok = r.returncode == 0 and "[rescan_hip]" not in r.stderr and os.path.exists(os.path.join(work, dst))
            stages = {k: float(v) for k, v in re.findall(r"(Optimization finished|Refining optimized poses done|Segmentation finished|Database augmentation finished) in ([0-9.]+)s", r.stdout)}
            poses = [l for l in open(os.path.join(work, dst)).read().splitlines() if l.strip().startswith("pose")] if ok else []
            print(f"[scene_list] {name} segment_transfer (no smoothing) {which}: {dt:.1f} s {stages} ok={ok}", file=sys.stderr, flush=True)
            return [dict(sequence=name, stage="segment_transfer_nosmooth", build=which, wall_s=dt, ok=ok, stages=stages, poses=poses)]
        for which in ("ref", "shim"):
            t = time.perf_counter()
            for k, name in enumerate(names):
                st_rows += seg_transfer(name, which, k % args.gpus)
            st_total[which] = time.perf_counter() - t
        ref_poses = {r["sequence"]: r["poses"] for r in st_rows if r["build"] == "ref"}
        for r in st_rows:
            if r["build"] != "ref":
                r["pose_lines_identical"] = bool(r["ok"] and r["poses"] == ref_poses.get(r["sequence"]))
            r["n_pose_lines"] = len(r.pop("poses"))

    # distance of every build's proposals from the ref build's
    ref_bins = {r["sequence"]: r["bin"] for r in rows if r.get("build") == "ref" and r["stage"] == "pose_proposal"}
    for r in rows:
        if r["stage"] != "pose_proposal" or r["build"] == "ref" or r["sequence"] not in ref_bins or not r["ok"]:
            continue
        try:
            a, b = read_pose_bin(ref_bins[r["sequence"]]), read_pose_bin(r["bin"])
            same_shape = len(a) == len(b) and all(len(x) == len(y) for x, y in zip(a, b))
            r["same_proposals"] = bool(same_shape)
            if same_shape:
                d = [float(np.linalg.norm(x[:16].astype(np.float64) - y[:16])) for pa, pb in zip(a, b) for x, y in zip(pa, pb)]
                good = [float(np.linalg.norm(x[:16].astype(np.float64) - y[:16])) for pa, pb in zip(a, b) for x, y in zip(pa, pb) if 0.9 < x[16] < 9.0]
                r["n_proposals"] = len(d); r["identical"] = int(sum(v == 0.0 for v in d))
                r["max_pose_delta"] = max(d) if d else 0.0
                r["max_pose_delta_good"] = max(good) if good else 0.0
        except Exception as e:  # a missing .bin is a failed run, reported as such
            r["compare_error"] = str(e)
    for r in rows:
        r.pop("bin", None)

    summary = {}
    for build in builds:
        pr = [r for r in rows if r["stage"] == "pose_proposal" and r["build"] == build]
        summary[build] = dict(sequences=len(pr), ok=int(sum(r["ok"] for r in pr)), batch_wall_s=t_total[build],
                              sum_wall_s=sum(r["wall_s"] for r in pr),
                              sum_computed_poses_s=sum(r["computed_poses_s"] or 0.0 for r in pr),
                              max_pose_delta_good=max([r.get("max_pose_delta_good", 0.0) for r in pr] or [0.0]),
                              identical=int(sum(r.get("identical", 0) for r in pr)), proposals=int(sum(r.get("n_proposals", 0) for r in pr)))
    out = dict(config="BASELINE.json configs[0]/[4]: scene_list batch, seg2rsdb -> pose_proposal (segment_transfer not run: gco-v3.0 is not vendored)",
               sequences=names, points_per_scan=n_pts, density=args.density, gpus=args.gpus, jobs_per_gpu=args.jobs, host_cores=len(os.sched_getaffinity(0)),
               omp_threads=int(os.environ.get("RS_SCENE_LIST_OMP_THREADS", "8")),
               builds_missing=missing, summary=summary, rows=rows,
               segment_transfer_nosmooth=dict(note="apps/segment_transfer minus its gco lines (no smoothing: says nothing about rspf_smooth_labels); ref build on the ref proposals, shim build on the batched build's",
                                              batch_wall_s=st_total, rows=st_rows,
                                              sum_stage_s={w: {k: sum(r["stages"].get(k, 0.0) for r in st_rows if r["build"] == w) for k in ("Optimization finished", "Refining optimized poses done", "Segmentation finished")} for w in st_total},
                                              pose_lines_identical=int(sum(bool(r.get("pose_lines_identical")) for r in st_rows)), sequences=len(names)))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print("| build | sequences ok | batch wall-clock s | sum of 'Computed poses in' s | proposals identical to ref | max pose delta (score > 0.9) |")
    print("|---|---|---|---|---|---|")
    for b, v in summary.items():
        print(f"| {b} | {v['ok']}/{v['sequences']} | {v['batch_wall_s']:.2f} | {v['sum_computed_poses_s']:.2f} | {v['identical']}/{v['proposals']} | {v['max_pose_delta_good']:.2e} |")
    if st_rows:
        st = out["segment_transfer_nosmooth"]
        print()
        print("| segment_transfer minus its smoothing | batch wall-clock s | sum 'Optimization' s | sum 'Refining optimized poses' s | sum 'Segmentation' s | .rsdb pose lines identical to the ref build's |")
        print("|---|---|---|---|---|---|")
        for w in st_total:
            ss = st["sum_stage_s"][w]
            print(f"| {w} | {st_total[w]:.2f} | {ss['Optimization finished']:.2f} | {ss['Refining optimized poses done']:.3f} | {ss['Segmentation finished']:.3f} | " + ("—" if w == "ref" else f"{st['pose_lines_identical']}/{len(names)}") + " |")


if __name__ == "__main__":
    main()
