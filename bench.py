#!/usr/bin/env python3
"""bench.py — point-pairs/sec of the Rescan hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the three consumers over one synthetic 2-scan scene held in HBM
(BASELINE.json configs[1]):

  ICP-NN    whole-scan point-to-plane ICP, scan t1 (~1M pts) -> scan t0 (~1M pts), K=16,
            r = 0.10 with the reference's 0.95 shrink schedule, 60°, 10 fixed iterations
            (search, weights, normal-equation reduction, 6x6 solve and pose update all on the GPU)
  score-NN  256 poses x 10k-point object against the 1M-point scan, K=64, r=0.10, 35° gate
  label-NN  8 placements of ~50k-point objects against the 1M scene points, K=1, r=0.05

point-pairs = sum of query points over all searches = 10*n_scan + 256*n_obj + 8*n_scan.
Inputs are resident in HBM when the timed region starts; poses / scores / labels return to the
host inside the timed region (they are the outputs the reference's callers consume).

`--scaling strong` (N >= 1): ONE FIXED scene whatever N is — 8 ICP problems (every placement's ~50k-point model refined against
the 1M-point scan, the loop of lib/rs/rs_database.h:220-230 with the reference's (0.075, 50 deg) and 10 fixed iterations), 256
score poses, 8 placements — its units sharded over the ranks like below; total work does not grow with N ("scaling": "strong").

N > 1 (torchrun, one rank per GPU, RCCL) — BASELINE.json configs[3], SURVEY.md §8e: ONE scene, replicated on every
GPU; the units of the three consumers (ICP start poses, score poses, placements of the sorted arrangement) are
sharded across the ranks with rescan_amd.dist.shard_range; every rank's label kernel writes its per-placement unary
rows straight into the RCCL send buffer, next to its poses / errors / scores; ONE all-gather per step; every rank
folds the gathered rows in the arrangement's sorted order on the device (rs_hip_fold_label_rows_device).  Weak
scaling: the unit lists grow with N (N x {1 ICP problem, 256 poses, 8 placements}), so per-GPU work is fixed;
value = pairs of all ranks / max-over-ranks time.  `--shard` runs the same route at N = 1 (a world of one: the
"gather" is the send buffer itself); `--replicas` is the other multi-GPU shape (configs[4]: every rank its own
scene, results all-gathered).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The live profile of the ICP chain is SAMPLED: every 4th icp_align call of the timed region carries the HIP events its kernels
# are timed with (RS_HIP_PROF_EVERY=1 in the environment times every call).  An event recorded between two dependent launches is
# a packet of its own: two per ICP iteration cost 2.3 % of a step (3.05 against 2.98 ms, same box, interleaved repeats:
# profiles/r02/ab_profiling_events.txt; events carried inside the kernels' dispatch packets — hipExtLaunchKernelGGL — cost more:
# 3.11).  The score and label kernels are timed on every call.
os.environ.setdefault("RS_HIP_PROF_EVERY", "4")
# completion waits busy-wait instead of sleeping (the library reads this in rs_hip_init; opt-in, see rs_api.hip)
os.environ.setdefault("RS_HIP_SCHEDULE", "spin")

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)
ICP_ITERS = 10
N_POSES = 256
N_PLACEMENTS = 8
I4 = np.eye(4, dtype=np.float32).ravel()


def build_inputs(n_points, seed, units=1, centre=False, t0=0):
    """The step's inputs as numpy arrays only (no device): tests/golden/bench_seed11.npz pins the reference's
    outputs for exactly these (oracle/gen_golden_bench.py imports this function in the build container).
    units > 1 (sharded multi-GPU route): the unit lists are `units` times as long — further ICP start poses, score
    poses and placements of the same scene and objects, drawn after the first unit's, which stays what it is."""
    from rescan_amd import synth
    # (t0 > 0, --timesteps T: the pair (t0, t0 + 1) of the same sequence — the room after the furniture was moved again; the random
    #  draws below are the pair's own.  t0 = 0 is the workload the fixtures pin.)
    s0 = synth.scene_for_point_count(int(n_points * 0.84), seed=seed, timestep=t0)
    s1 = synth.scene_for_point_count(int(n_points * 0.84), seed=seed, timestep=t0 + 1)
    if centre:
        # the whole world moved so that the scan's median point is the origin: coordinates of both signs — the reference's fp32
        # centroid sums then hover around zero instead of growing, the regime the grid chains give up in
        sh = -np.median(s1["points"], axis=0).astype(np.float32)
        for sc in (s0, s1):
            sc["points"] = sc["points"] + sh
            for q in sc["objects"]:
                q["pose"] = q["pose"].copy(); q["pose"][12:15] += sh
    w = {}
    w["s0"], w["s1"] = s0, s1
    rng = np.random.default_rng(seed + 5 + 100000 * t0)
    w["icp_T0"] = synth.perturbed_pose(I4, rng, 0.01, 0.01)
    # score object: a table model resampled to ~10k points
    op, on = synth.make_object("table", seed * 13 + 1, density=3800.0)
    w["obj_score_np"] = (op, on)
    tbl = [o for o in s1["objects"] if o["kind"] == "table"][0]
    w["score_poses"] = np.stack([synth.perturbed_pose(tbl["pose"], rng, 0.6, 0.25) for _ in range(N_POSES)])
    # label placements: 8 scene objects with dense (~50k-point) model clouds, slightly mis-posed
    plc = []
    for k, o in enumerate(s1["objects"][:N_PLACEMENTS]):
        dens = 50000.0 / max(1, len(o["pos"])) * synth.DENSITY
        lp, ln = synth.make_object(o["kind"], seed * 7919 + k, density=dens)
        plc.append(dict(np=(lp, ln), pose=synth.perturbed_pose(o["pose"], rng, 0.01, 0.005), cls=o["class_idx"]))
    w["icp_T0s"] = [w["icp_T0"]]
    for u in range(1, units):
        rng_u = np.random.default_rng(seed + 5 + 1000 * u + 100000 * t0)
        w["icp_T0s"].append(synth.perturbed_pose(I4, rng_u, 0.01, 0.01))
        w["score_poses"] = np.concatenate([w["score_poses"], np.stack([synth.perturbed_pose(tbl["pose"], rng_u, 0.6, 0.25) for _ in range(N_POSES)])])
        for k in range(N_PLACEMENTS):
            plc.append(dict(np=plc[k]["np"], same_as=k, pose=synth.perturbed_pose(s1["objects"][k]["pose"], rng_u, 0.01, 0.005), cls=plc[k]["cls"]))
    w["icp_T0s"] = np.stack(w["icp_T0s"])
    # --scaling strong: one ICP problem per placement — the placement's model cloud against the scan, from the placement's (slightly
    # wrong) pose, with rsdb_refine_alignment_of_objects_to_scene's parameters (lib/rs/rs_database.h:220-230)
    w["strong_icp"] = dict(T0s=np.stack([p["pose"] for p in plc[:N_PLACEMENTS]]), max_dist=0.075, max_angle=np.deg2rad(50.0))
    w["plc"] = plc
    w["plc_poses"] = np.stack([p["pose"] for p in plc])
    w["units"] = units
    w["n_scan0"], w["n_scan1"], w["n_obj"] = len(s0["points"]), len(s1["points"]), len(op)
    w["pairs"] = dict(icp=ICP_ITERS * w["n_scan1"], score=N_POSES * w["n_obj"], label=N_PLACEMENTS * w["n_scan1"])   # per unit
    return w


def build_workload(n_points, seed, knn, units=1, centre=False, t0=0):
    """build_inputs + the device-resident clouds (inputs are in HBM before the timed region starts)."""
    from rescan_amd import capi
    w = build_inputs(n_points, seed, units, centre, t0)
    cell = float(os.environ.get("RS_BENCH_CELL", "-1")) if knn == "hash" else 0.0
    s0, s1 = w["s0"], w["s1"]
    w["scan0"] = capi.Cloud(s0["points"], s0["normals"], cell_size=cell)      # ICP target
    w["scan1"] = capi.Cloud(s1["points"], s1["normals"], cell_size=cell)      # ICP source, score + label scene
    op, on = w["obj_score_np"]
    w["obj_score"] = capi.Cloud(op, on, cell_size=cell if cell != 0 else -1.0)
    for p in w["plc"]:
        # (--knn brute: the placed models — the label pass's TARGETS — are one brute tile each too)
        p["cloud"] = w["plc"][p["same_as"]]["cloud"] if "same_as" in p else capi.Cloud(p["np"][0], p["np"][1], cell_size=cell)
    return w


_POOL = None


def _pool():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=3)
    return _POOL


_ROLE_POOLS = None
_ROLE_TIMES = [] if os.environ.get("RS_BENCH_PRINT_STEPS") else None      # per step: host-side ms of the three consumers' calls


class RoleRunner:
    """Runs the three consumers of a step side by side, each always on the same host thread (i.e. on the same HIP stream and CU
    set), and joins them.

    spin (default): the longest of the three — the ICP chain (RS_BENCH_MAIN_ROLE=icp|score|label; it was the score batch
    until that got single-wave workgroups) — runs on the CALLING thread, the two others on worker threads that never sleep:
    between steps they busy-wait in native code (tools/benchaux: rsb_spin_wait, no interpreter lock held) for a flag that the
    caller stores WITHOUT releasing the interpreter lock right before its own consumer's call (spin_post_holding_gil: the
    waiting thread can only go on once the poster is inside its native call — the longest consumer is under way before the
    others are released, the second-longest first), and the caller busy-waits for theirs.  These helpers are the harness's own
    (tools/benchaux/librs_benchaux.so), not part of the product ABI.  With executors every step is three submissions and three future waits, i.e. six wake-ups of sleeping
    threads by the host scheduler — usually tens of microseconds each, but on the shared 256-thread hosts of this pool one
    step in ~35 lost 1.5-3 ms there while all three library calls took their usual time (cpu.stat before / after: no cgroup
    throttling in the region; the time is between the calls).  RS_BENCH_SPIN=0: three single-thread executors as before."""

    ICP, SCORE, LABEL = 0, 1, 2

    def __init__(self, masks, spin, main_role=0):
        import threading
        from concurrent.futures import ThreadPoolExecutor
        from rescan_amd import capi
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import benchaux
        self.aux = benchaux
        self.capi, self.spin, self.k, self.stop, self.masked = capi, spin, 0, False, bool(masks)
        self.fns, self.out, self.dt = [None] * 3, [None] * 3, [0.0] * 3
        self.main = main_role
        # release order of the workers: the score batch before the label pass
        self.workers = [r for r in (self.SCORE, self.ICP, self.LABEL) if r != main_role]
        if not spin:
            self.pools = [ThreadPoolExecutor(max_workers=1) for _ in range(3)]
            if masks:
                for ex, m in zip(self.pools, masks):
                    ex.submit(capi.stream_cu_mask, m).result()
            return
        self.flags = np.zeros(16, np.int32)           # [0..2] go, [4..6] done, [8..10] ready
        self.addr = lambda kind, r: self.flags.ctypes.data + 4 * (4 * kind + r)
        self.failed = None
        if masks:
            capi.stream_cu_mask(masks[self.main])     # the calling thread's stream takes the main role's CUs
        self.threads = [threading.Thread(target=self._worker, args=(r, masks[r] if masks else None), daemon=True) for r in self.workers]
        for t in self.threads:
            t.start()
        for r in self.workers:
            self.aux.spin_wait(self.addr(2, r), 1, 60.0)
        if self.failed is not None:
            try:
                self.close()                       # (park the workers that did start, give the calling thread its unmasked stream back)
            finally:
                raise self.failed                  # the original error, whatever close() ran into

    def _worker(self, r, mask):
        capi, aux = self.capi, self.aux
        try:
            if mask:
                capi.stream_cu_mask(mask)
        except Exception as e:               # a runtime without CU masks
            self.failed = e
        aux.spin_post(self.addr(2, r), 1)
        k = 0
        while True:
            k += 1
            aux.spin_wait(self.addr(0, r), k, 0.0)
            if self.stop:
                if mask:
                    try:
                        capi.stream_cu_mask(None)     # (a profiler's finalisation does not survive masked streams)
                    except Exception:
                        pass
                aux.spin_post(self.addr(1, r), 1 << 30)
                return
            if r == self.workers[0]:
                aux.spin_post_holding_gil(self.addr(0, self.workers[1]), k)  # the second worker goes once this one is inside the library
            t = time.perf_counter()
            try:
                out = self.fns[r]()
            except BaseException as e:        # handed to the caller
                out = e
            self.out[r], self.dt[r] = out, time.perf_counter() - t
            aux.spin_post(self.addr(1, r), k)

    def run3(self, icp, score, label):
        """-> (icp(), score(), label()), run concurrently."""
        fns = (icp, score, label)
        if not self.spin:
            def timed(r):
                def run():
                    t = time.perf_counter(); o = fns[r](); self.dt[r] = time.perf_counter() - t
                    return o
                return run
            f = [ex.submit(timed(r)) for r, ex in enumerate(self.pools)]
            outs = [x.result() for x in f]
        else:
            aux = self.aux
            self.k += 1
            self.fns = fns
            aux.spin_post_holding_gil(self.addr(0, self.workers[0]), self.k)   # the first worker goes once this thread is inside the library
            t = time.perf_counter()
            mine = fns[self.main]()
            self.dt[self.main] = time.perf_counter() - t
            for r in self.workers:
                aux.spin_wait(self.addr(1, r), self.k, 120.0)
            outs = [mine if r == self.main else self.out[r] for r in range(3)]
            for o in outs:
                if isinstance(o, BaseException):
                    raise o
        if _ROLE_TIMES is not None:
            _ROLE_TIMES.append([x * 1e3 for x in self.dt])
        return outs

    def close(self):
        """Parks the spinning workers (they would otherwise keep two cores busy) and gives every thread an unmasked stream again."""
        if self.stop:
            return
        self.stop = True
        if not self.spin:
            if self.masked:
                for ex in self.pools:
                    ex.submit(self.capi.stream_cu_mask, None).result()
            return
        for r in self.workers:
            self.aux.spin_post(self.addr(0, r), 1 << 30)
        for r in self.workers:
            if self.threads[self.workers.index(r)].is_alive():        # a worker that died has nothing to acknowledge
                self.aux.spin_wait(self.addr(1, r), 1 << 30, 30.0)
        if self.masked:
            self.capi.stream_cu_mask(None)


def _roles():
    """The RoleRunner of this process.  A consumer (ICP chain, score batch, label pass) always runs on the same host thread,
    i.e. on the same HIP stream — and the streams are confined to disjoint sets of CUs (rs_hip_stream_cu_mask).  A bit
    of the mask is not "a CU of the chip in order": bit i is CU slot i / 8 of XCD i % 8 and slot j is a CU of shader engine
    j % 4 (tools/cu_mask_probe.py, profiles/r02/cu_mask_probe.txt), so bits [0,160) | [160,256) give the ICP chain 5 CUs of every
    shader engine of every XCD and the score batch the other 3; the shader engines hand out workgroups evenly, so splits that
    leave them unequal run at the smallest one's pace (granularity: 32 bits).  The chain (short latency-bound kernels) gets CUs
    no long-lived score wave sits on; the label pass, 0.4 ms, shares the chain's side.  3.07 ms per step unpartitioned,
    2.95 with 5 + 3 and the label pass beside the score batch, 2.80 with it beside the chain (interleaved repeats on one box:
    profiles/r02/ab_cu_split*.txt, ab_label_on_chain_and_no_partition_experiments.txt).
    RS_BENCH_CU_SPLIT=<fraction of the mask bits for the chain> overrides (0 = no partition); RS_BENCH_LABEL_ON=chain|batch|all."""
    global _ROLE_POOLS
    if _ROLE_POOLS is None:
        from rescan_amd import capi
        import torch
        n_cu = int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count)
        # (round 3: 0.75 — 6 + 2 CUs of every shader engine.  With the reference's centroid chains the ICP chain is the longer side by a
        #  millisecond (3.25 ms against a 1.97 ms score batch at 5 + 3), so it gets the sixth CU: 3.03 ms, the batch 2.85 ms; at 7 + 1 the
        #  batch would take 5.7 ms — profiles/r03/ab_cu_split.txt)
        split = float(os.environ.get("RS_BENCH_CU_SPLIT", "0.75" if n_cu == 256 else "0"))
        # spinning threads need their cores: 3 busy threads per rank; a container whose CPU quota does not cover that for all the
        # node's ranks would be throttled (every thread of the cgroup stalls for the rest of the 100 ms period)
        st = host_cpu_stat()
        cpus = None
        if st and st[0] and st[0].split()[0] != "max":
            q, per = st[0].split()
            cpus = float(q) / float(per)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None      # (a cpuset limits the cores like a quota does)
        cpus = aff if cpus is None else (min(cpus, aff) if aff else cpus)
        spin_default = "0" if (cpus is not None and cpus < 4.0 * local_world) else "1"
        spin = os.environ.get("RS_BENCH_SPIN", spin_default) != "0"
        masks, note = None, "none"
        if split > 0.0:
            k = int(round(split * n_cu))
            lo = int(round(float(os.environ.get("RS_BENCH_CU_BATCH_LO", split)) * n_cu))     # experiment: overlapping partitions
            label_on = os.environ.get("RS_BENCH_LABEL_ON", "chain")
            chain, batch = [1] * k + [0] * (n_cu - k), [0] * lo + [1] * (n_cu - lo)
            masks = [chain, batch, {"chain": chain, "batch": batch, "all": [1] * n_cu}[label_on]]
            f_chain, f_batch = k / n_cu, (n_cu - lo) / n_cu
            f_label = {"chain": f_chain, "batch": f_batch, "all": 1.0}[label_on]
            note = "CU mask bits [0,%d) (%d CUs of every shader engine of every XCD) ICP chain%s, [%d,%d) score batch%s" % (
                k, k // 32, " + label pass" if label_on == "chain" else "", lo, n_cu, " + label pass" if label_on == "batch" else "")
        main_role = {"icp": 0, "score": 1, "label": 2}.get(os.environ.get("RS_BENCH_MAIN_ROLE", "icp"), 0)
        try:
            runner = RoleRunner(masks, spin, main_role)
            if masks:
                _CU_SHARES.update({"nn_icp": f_chain, "icp_moments": f_chain, "nn_score": f_batch, "nn_label": f_label})
        except Exception as e:               # a runtime without CU masks: plain streams
            runner, note = RoleRunner(None, spin, main_role), "none (%s)" % e
        _ROLE_POOLS = [runner, note + ("; consumers joined by spinning (%s on the calling thread)" % os.environ.get("RS_BENCH_MAIN_ROLE", "icp") if spin else "; consumers on three executors")]
    return _ROLE_POOLS[0]


def close_roles():
    """Parks the process's RoleRunner (its spinning workers, its CU-masked streams); the next concurrent step makes a new one."""
    global _ROLE_POOLS
    if _ROLE_POOLS:
        _ROLE_POOLS[0].close()
        _ROLE_POOLS = None


def host_cpu_stat():
    """(quota string, throttled periods, throttled microseconds) of this container's CPU cgroup, or None: a step that falls into
    a throttled period stalls with every thread of the process, whatever the GPU does."""
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None
        f = "/sys/fs/cgroup/cpu.stat" if os.path.exists("/sys/fs/cgroup/cpu.stat") else "/sys/fs/cgroup/cpu/cpu.stat"
        d = dict(line.split() for line in open(f).read().strip().splitlines())
        return quota, int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", d.get("throttled_time", 0)))
    except Exception:
        return None


def cu_partition_note():
    return _ROLE_POOLS[1] if _ROLE_POOLS else "none"


_CU_SHARES = {}


def cu_shares():
    """Fraction of the chip's CUs each domain's stream may use (1.0 without a partition)."""
    return dict(_CU_SHARES)


def run_step(w, dist_ctx=None, concurrent=True):
    """One pass of the hot path (single GPU, or one replica of --replicas).  Returns the outputs (poses, scores, labels).

    The three consumers are independent (in the reference they even run in different processes),
    so they are issued from three host threads; the library gives every thread its own HIP stream
    and workspaces, and the GPU overlaps the ICP chain with the score batch and the label pass."""
    from rescan_amd import capi

    def icp():
        return capi.icp_align(w["scan1"], w["scan0"], w["icp_T0"], I4, 0.10, np.deg2rad(60.0),
                              max_iter=ICP_ITERS, fixed_iters=True)

    def score():
        return capi.alignment_scores(w["obj_score"], w["scan1"], w["score_poses"], 0.1, 64)

    def label():
        return capi.arrangement_to_labels(w["scan1"], w["plc_poses"], [p["cloud"] for p in w["plc"]],
                                          [0] * len(w["plc"]), [p["cls"] for p in w["plc"]], 0.05, False)

    if concurrent:
        (err, T, it), scores, res = _roles().run3(icp, score, label)
    else:
        (err, T, it), scores, res = icp(), score(), label()
    if dist_ctx is not None:
        # The exchange of step s overlaps step s + 1 (its consumers are host code downstream — the graph cut — not the next
        # step's kernels): one exchange in flight, the last one is waited for before the timed region closes (exchange_wait).
        exchange_wait()
        _XCH["pending"] = _exchange_pool().submit(exchange_results, dist_ctx[0], dist_ctx[1], T, err, scores, res)
    return dict(err=err, T=T, scores=scores, labels=res["labels"], min_dists=res["min_dists"])


def _exchange_pool():
    if "pool" not in _XCH:
        from concurrent.futures import ThreadPoolExecutor
        _XCH["pool"] = ThreadPoolExecutor(max_workers=1)
    return _XCH["pool"]


def exchange_wait():
    f = _XCH.pop("pending", None)
    return f.result() if f is not None else None


_XCH = {}


def all_gather_flat(dist, recv, send):
    """recv = the concatenation of every rank's send (device tensors).  RCCL: one all_gather_into_tensor.  Under the one-GPU
    rehearsal (gloo) the buffers pass through the host."""
    if dist.get_backend() == "gloo":
        import torch
        parts = [torch.empty(send.numel(), dtype=send.dtype) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, send.cpu())
        recv.copy_(torch.cat(parts))
    else:
        dist.all_gather_into_tensor(recv, send)


def exchange_results(dist, dev, T, err, scores, res):
    """--replicas: every rank receives every rank's pose + error + scores and label partials (labels int8 + min_dists
    f32) in ONE all-gather per step.  Scene sizes differ slightly across ranks: the label arrays are padded to the common
    maximum, agreed on once.  Packed per rank: int64 n | f32 pose, err, scores | f32 min_dists[nmax] |
    int8 labels[nmax].  Returns per-rank views (small, labels, min_dists) of the gathered buffer."""
    import torch
    if dev.type == "cuda":
        torch.cuda.set_device(dev)                 # (called from the exchange thread)
    world = dist.get_world_size()
    lab = np.ascontiguousarray(res["labels"], np.int8)
    mind = np.ascontiguousarray(res["min_dists"], np.float32)
    small = np.concatenate([np.asarray(T, np.float32).ravel(), [np.float32(err)], np.asarray(scores, np.float32)]).astype(np.float32)
    st = _XCH.get("st")
    if st is None or st["n"] != len(lab) or st["n_small"] != len(small) or st["world"] != world:
        n = torch.tensor([len(lab)], device=dev if dist.get_backend() != "gloo" else "cpu", dtype=torch.int64)
        dist.all_reduce(n, op=dist.ReduceOp.MAX)
        nmax = int(n.item())
        o_small, o_mind = 8, 8 + 4 * len(small)
        o_lab = o_mind + 4 * nmax
        nbytes = (o_lab + nmax + 15) // 16 * 16
        host = torch.zeros(nbytes, dtype=torch.uint8)
        if dev.type == "cuda":
            host = host.pin_memory()
        st = dict(n=len(lab), n_small=len(small), world=world, nmax=nmax, nbytes=nbytes, off=(o_small, o_mind, o_lab), host=host,
                  send=torch.empty(nbytes, dtype=torch.uint8, device=dev), recv=torch.empty(world * nbytes, dtype=torch.uint8, device=dev))
        _XCH["st"] = st
    o_small, o_mind, o_lab = st["off"]
    nmax, nbytes = st["nmax"], st["nbytes"]
    h = st["host"].numpy()
    h[0:8].view(np.int64)[0] = len(lab)
    h[o_small:o_mind].view(np.float32)[:] = small
    hm = h[o_mind:o_lab].view(np.float32); hm[: len(mind)] = mind; hm[len(mind):] = 1e9
    hl = h[o_lab:o_lab + nmax].view(np.int8); hl[: len(lab)] = lab; hl[len(lab):] = 0
    if dev.type == "cuda":
        if "stream" not in _XCH:
            _XCH["stream"] = torch.cuda.Stream(device=dev)      # not the legacy default stream: it would wait for the consumers' (blocking, CU-masked) streams
        with torch.cuda.stream(_XCH["stream"]):
            st["send"].copy_(st["host"], non_blocking=True)
            all_gather_flat(dist, st["recv"], st["send"])
        _XCH["stream"].synchronize()
    else:
        st["send"].copy_(st["host"])
        all_gather_flat(dist, st["recv"], st["send"])
    out, gl, gm = [], [], []
    for r in range(world):
        b = st["recv"][r * nbytes:(r + 1) * nbytes]
        out.append(b[o_small:o_mind].view(torch.float32))
        gm.append(b[o_mind:o_lab].view(torch.float32))
        gl.append(b[o_lab:o_lab + nmax].view(torch.int8))
    return out, gl, gm


# ---- the sharded route (north_star / configs[3]) ----------------------------------------------------------------

class Sharded:
    """One scene, units sharded over the ranks, rows gathered on the device (rescan_amd/dist.py: shard_*).
    Two send / receive buffer sets: the exchange of step s (copy of the small results, all-gather, fold, download of the
    folded labels) runs on its own host thread while step s + 1 computes into the other set."""

    _xs = None
    _xs_bound = False

    def __init__(self, w, dist, dev, rank, world, strong=False, strong_problems=N_PLACEMENTS):
        import torch
        from rescan_amd import dist as rd
        self.w, self.dist, self.dev, self.rank, self.world = w, dist, dev, rank, world
        n_plc = len(w["plc"])
        n_icp = strong_problems if strong else len(w["icp_T0s"])
        # every rank sends the (min_dist, label) partial of its own run of the sorted arrangement (5 B per scene point) instead of its unary
        # rows (4 B per point and placement: 31 MB per rank at 8 placements); RS_BENCH_PREFOLD=0 sends the rows.  Same bits either way.
        self.lay = rd.ShardLayout(world, n_icp, len(w["score_poses"]), n_plc, w["n_scan1"], prefold=os.environ.get("RS_BENCH_PREFOLD", "1") != "0")
        order, _, radii = rd.arrangement_plan([0] * n_plc, [p["cls"] for p in w["plc"]], 0.05)
        self.order = order
        si = w["strong_icp"]
        if strong and strong_problems > N_PLACEMENTS:
            # further refines of the same eight models from further (slightly wrong) start poses, drawn after the fixture's eight
            from rescan_amd import synth
            rng_s = np.random.default_rng(4242)
            more = [synth.perturbed_pose(si["T0s"][k % N_PLACEMENTS], rng_s, 0.01, 0.005) for k in range(N_PLACEMENTS, strong_problems)]
            si = dict(si, T0s=np.concatenate([si["T0s"], np.stack(more)]))
        self.units = dict(
            icp=([w["plc"][k % N_PLACEMENTS]["cloud"] for k in range(strong_problems)], w["scan1"], si["T0s"], si["max_dist"], si["max_angle"], ICP_ITERS) if strong else
                (w["scan1"], w["scan0"], w["icp_T0s"], 0.10, np.deg2rad(60.0), ICP_ITERS),
            score=(w["obj_score"], w["scan1"], w["score_poses"], 0.1, 64),
            label=(w["scan1"], w["plc_poses"][order], [w["plc"][i]["cloud"] for i in order], radii))
        self.bufs = []
        for _ in range(2):
            send = torch.zeros(self.lay.words, dtype=torch.float32, device=dev)
            recv = send if world == 1 and dist is None else torch.zeros(world * self.lay.words, dtype=torch.float32, device=dev)
            small = torch.zeros(self.lay.small_words, dtype=torch.float32).pin_memory()
            # the folded labels / min_dists land in caller-owned pinned arrays (the reference's callers own them too)
            out = (torch.zeros(w["n_scan1"], dtype=torch.int8).pin_memory().numpy(), torch.zeros(w["n_scan1"], dtype=torch.float32).pin_memory().numpy())
            self.bufs.append((send, recv, small, out))
        # the exchange thread's stream: non-blocking (see exchange) and of high priority — its few kernels (copy, collective, fold) are
        # dispatched ahead of the next step's tens of thousands of queued workgroups instead of behind them
        # (one such stream per process: --timesteps > 2 makes one Sharded per pair of scans, and they share the exchange thread)
        if Sharded._xs is None:
            Sharded._xs = torch.cuda.Stream(device=dev, priority=-1) if not os.environ.get("RS_BENCH_XCH_DEFAULT_STREAM") else torch.cuda.default_stream(dev)   # (the variable: A/B of the above)
        self.xs = Sharded._xs
        self.step_index = 0
        self.stat_from = 0
        self.t_compute = self.t_wait = 0.0
        self.t_gather = self.t_fold = 0.0; self.n_exchanges = 0      # host-side durations of the exchange thread's two halves

    def reset_stats(self):
        """(the averages in the bench line are over the timed steps, not the warm-up's first-use allocations)"""
        self.stat_from = self.step_index
        self.t_compute = self.t_wait = self.t_gather = self.t_fold = 0.0
        self.n_exchanges = 0

    def exchange(self, b):
        import torch
        from rescan_amd import capi, dist as rd
        torch.cuda.set_device(self.dev)                # (exchange thread)
        send, recv, small, outbuf = b
        t0 = time.perf_counter()
        # On a stream of its own: torch's default stream is the legacy null stream, which waits for — and holds up — every
        # "blocking" stream of the device, and the CU-masked streams of the consumers are such streams
        # (hipExtStreamCreateWithCUMask takes no flags): on the default stream this copy and the collective ran only once the NEXT
        # step's kernels had drained (2.4 ms of "gather" per step, 0.3 ms of it waited for by the main thread).
        with torch.cuda.stream(self.xs):
            rd.shard_publish(self.lay, send, small)
            if self.dist is not None:
                all_gather_flat(self.dist, recv, send)
        self.xs.synchronize()
        if not Sharded._xs_bound:
            capi.set_stream(self.xs.cuda_stream)         # the library's work of this thread (the fold) goes to the same stream
            Sharded._xs_bound = True
        t1 = time.perf_counter()
        with torch.cuda.stream(self.xs):                 # (its read-back of the small blocks is a torch copy: same stream, same reason)
            out = rd.shard_fold(capi, self.lay, recv, outbuf, scene=self.w["scan1"])
        self.t_gather += t1 - t0; self.t_fold += time.perf_counter() - t1; self.n_exchanges += 1
        return out

    def step(self, concurrent=True):
        from rescan_amd import capi, dist as rd
        b = self.bufs[self.step_index & 1]
        self.step_index += 1
        t0 = time.perf_counter()
        rd.shard_compute(capi, self.lay, self.rank, self.units, b[0], b[2], threads=_roles() if concurrent else None)
        t1 = time.perf_counter()
        prev = exchange_wait()                          # at most one exchange in flight, so the other buffer set is free again
        _XCH["pending"] = _exchange_pool().submit(self.exchange, b)
        self.t_compute += t1 - t0; self.t_wait += time.perf_counter() - t1
        return prev


def parity_block(out, n_points, seed, knn, units, strong=False, centre=False, t0=0):
    """Distance of the LAST step's outputs from tests/golden/bench_seed11.npz — what the compiled reference computes for
    the same inputs (oracle/gen_golden_bench.py: pose / error after the 10 fixed iterations composed from the reference's
    icp_find_corrs + icp_estimate_rigid_xform_pt2pl, the 256 scores of mgs_compute_object_alignment_score; labels and min_dists
    of rspf_arrangement_to_labels by the reference's own text, oracle/_ref/libref_filters.so).  Computed outside the timed region."""
    import hashlib
    # (--centre and the further scan pairs of --timesteps have fixtures of their own, made by the same reference build:
    #  oracle/gen_golden_bench.py --centre / --pair)
    name = "bench_seed%d%s%s.npz" % (seed, "_centre" if centre else "", "_t%d" % t0 if t0 else "")
    path = os.path.join(ROOT, "tests", "golden", name)
    if out is None or not os.path.exists(path) or n_points != 1_000_000:
        return None if not (centre or t0) else "not compared: no tests/golden/%s" % name
    g = np.load(path)
    blk = {"fixture": "tests/golden/" + name, "knn": knn}
    if strong and "strong_icp_pose" in g and out.get("Ts") is not None:
        # --scaling strong refines the 8 placements' models against the scan (object-sized sources: the reference-order estimator); the
        # fixture holds the reference's own ten iterations of each (oracle/gen_golden_bench.py --strong-only)
        d = np.linalg.norm(np.asarray(out["Ts"], np.float64).reshape(-1, 16)[:N_PLACEMENTS] - g["strong_icp_pose"].astype(np.float64).reshape(-1, 16), axis=1)
        blk["icp"] = "8 per-placement refines (lib/rs/rs_database.h:220-230) against the reference's"
        blk["pose_dist"] = float(d.max()); blk["poses_bit_identical"] = int((d == 0.0).sum())
        blk["err_abs_diff"] = float(np.abs(np.asarray(out["errs"], np.float64)[:N_PLACEMENTS] - g["strong_icp_err"].astype(np.float64)).max())
    elif strong:
        blk["icp"] = "not compared: the fixture holds no --scaling strong units"
    else:
        blk["pose_dist"] = float(np.linalg.norm(np.asarray(out["T"], np.float64).ravel() - g["icp_pose"].astype(np.float64)))
        blk["err_abs_diff"] = float(abs(float(out["err"]) - float(g["icp_err"])))
    sc = np.asarray(out["scores"], np.float64)[:N_POSES]
    blk["score_max_abs_err"] = float(np.abs(sc - g["scores"].astype(np.float64)).max())
    if units == 1:
        if "labels" in g:
            blk["label_mismatches"] = int((out["labels"] != g["labels"]).sum())
        else:      # (the further fixtures hold the labels as a digest and a strided sample)
            same = hashlib.sha256(np.ascontiguousarray(out["labels"], np.int8).tobytes()).hexdigest() == str(g["labels_sha"])
            blk["label_mismatches"] = 0 if same else max(1, int((np.asarray(out["labels"])[::257] != g["labels_sample"]).sum()))
        blk["min_dists_identical"] = bool(hashlib.sha256(np.ascontiguousarray(out["min_dists"], np.float32).tobytes()).hexdigest() == str(g["min_dists_sha"]))
    else:
        # Weak scaling at N = units ranks (round 6): the FURTHER unit lists — one more ICP start pose, 256 more score poses, 8 more
        # placements per rank — against the reference build's results for them (tests/golden/bench_seed11_units.npz,
        # oracle/gen_golden_bench.py --units): every pose, every score, and the labels / min_dists of the whole 8 N-placement arrangement.
        upath = os.path.join(ROOT, "tests", "golden", "bench_seed%d_units.npz" % seed)
        gu = np.load(upath) if (os.path.exists(upath) and not strong and not centre and not t0) else None
        if gu is None or ("labels_sha_u%d" % units) not in gu or out.get("Ts") is None:
            blk["labels"] = "not compared: %d placements instead of the fixture's %d (tests/golden/bench_seed%d_units.npz holds N = 2, 4, 8)" % (units * N_PLACEMENTS, N_PLACEMENTS, seed)
        else:
            blk["units_fixture"] = "tests/golden/bench_seed%d_units.npz" % seed
            Ts = np.asarray(out["Ts"], np.float64).reshape(-1, 16)[:units]
            blk["pose_dist_all_units"] = float(np.linalg.norm(Ts - gu["icp_pose"][:units].astype(np.float64).reshape(-1, 16), axis=1).max())
            blk["err_abs_diff_all_units"] = float(np.abs(np.asarray(out["errs"], np.float64)[:units] - gu["icp_err"][:units].astype(np.float64)).max())
            sc_all = np.asarray(out["scores"], np.float64)[:units * N_POSES]
            blk["score_max_abs_err_all_units"] = float(np.abs(sc_all - gu["scores"][:units * N_POSES].astype(np.float64)).max())
            same = hashlib.sha256(np.ascontiguousarray(out["labels"], np.int8).tobytes()).hexdigest() == str(gu["labels_sha_u%d" % units])
            blk["label_mismatches"] = 0 if same else max(1, int((np.asarray(out["labels"])[::257] != gu["labels_sample_u%d" % units]).sum()))
            blk["min_dists_identical"] = bool(hashlib.sha256(np.ascontiguousarray(out["min_dists"], np.float32).tobytes()).hexdigest() == str(gu["min_dists_sha_u%d" % units]))
            blk["placements"] = units * N_PLACEMENTS
    blk["tolerance"] = {"pose_dist": 1e-4, "score_max_abs_err": 2e-6, "label_mismatches": 0}
    return blk


def dropin_block(w):
    """"device time + H2D/D2H that the reference-equivalent call would incur" — the same three consumers issued
    through librescan_dropin.so with HOST arrays in and results out (the reference's own entry points: icp_align with its stop
    test, the flat score / label entries), outside the timed region: the first call pays hashing, upload and the device index
    build of every array; a repeated call pays the hashes (arrays above 4 MB: a ~1 KB sample) and the results' way back."""
    import ctypes as C
    from rescan_amd import capi
    lib = C.CDLL(os.path.join(ROOT, "rescan_amd", "librescan_dropin.so"))

    class Mat4(C.Structure):
        _fields_ = [("data", C.c_float * 16)]
    lib.icp_align.restype = C.c_float
    lib.icp_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(Mat4), Mat4, C.c_float, C.c_float, C.c_bool]
    lib.rsd_alignment_scores.restype = C.c_int
    lib.rsd_alignment_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_void_p]
    lib.rsd_arrangement_to_labels.restype = C.c_int
    lib.rsd_arrangement_to_labels.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                              C.c_float, C.c_bool, C.c_void_p, C.c_void_p]
    lib.rsd_cache_clear.restype = None
    s0, s1 = w["s0"], w["s1"]
    op, on = w["obj_score_np"]
    plc = w["plc"][:N_PLACEMENTS]
    T2 = Mat4(); T2.data[:] = [float(x) for x in I4]
    poses = np.ascontiguousarray(w["score_poses"][:N_POSES], np.float32)
    scores = np.zeros(N_POSES, np.float32)
    pp = (C.c_void_p * len(plc))(*[p["np"][0].ctypes.data for p in plc]); pn = (C.c_void_p * len(plc))(*[p["np"][1].ctypes.data for p in plc])
    ns = np.array([len(p["np"][0]) for p in plc], np.int32)
    pl_poses = np.ascontiguousarray(w["plc_poses"][:N_PLACEMENTS], np.float32)
    st = np.zeros(len(plc), np.int32); cls = np.array([p["cls"] for p in plc], np.int32)
    labels = np.zeros(w["n_scan1"], np.int8); order = np.zeros(len(plc), np.int32)

    def step():
        if os.environ.get("RS_BENCH_MARK"):
            capi.profile_marker()                   # (a kernel trace of this block: tools/profile.sh trace_dropin)
        T = Mat4(); T.data[:] = [float(x) for x in w["icp_T0"]]
        t = [time.perf_counter()]
        lib.icp_align(s1["points"].ctypes.data, s1["normals"].ctypes.data, len(s1["points"]), s0["points"].ctypes.data, s0["normals"].ctypes.data,
                      len(s0["points"]), C.byref(T), T2, 0.1, float(np.deg2rad(60.0)), False)
        t.append(time.perf_counter())
        lib.rsd_alignment_scores(op.ctypes.data, on.ctypes.data, len(op), s1["points"].ctypes.data, s1["normals"].ctypes.data, len(s1["points"]),
                                 poses.ctypes.data, N_POSES, 0.1, 64, scores.ctypes.data)
        t.append(time.perf_counter())
        lib.rsd_arrangement_to_labels(s1["points"].ctypes.data, s1["normals"].ctypes.data, len(s1["points"]), C.addressof(pp), C.addressof(pn), ns.ctypes.data,
                                      pl_poses.ctypes.data, st.ctypes.data, cls.ctypes.data, len(plc), 0.05, False, labels.ctypes.data, order.ctypes.data)
        t.append(time.perf_counter())
        return [1e3 * (b - a) for a, b in zip(t[:-1], t[1:])]

    lib.rsd_cache_clear()
    capi.cloud_build_seconds(reset=True)
    first = step()
    build_s, n_built = capi.cloud_build_seconds(reset=True)
    rep = [step() for _ in range(3)]
    rep = [min(r[k] for r in rep) for k in range(3)]
    lib.rsd_cache_clear()
    # the same icp_align call through the library's own entry point on resident clouds (stop test on): what the boundary adds is the difference
    t = time.perf_counter()
    e_n, T_n, it_n = capi.icp_align(w["scan1"], w["scan0"], w["icp_T0"], I4, 0.10, np.deg2rad(60.0))
    native_icp_ms = 1e3 * (time.perf_counter() - t)
    t = time.perf_counter()
    e_n, T_n, it_n = capi.icp_align(w["scan1"], w["scan0"], w["icp_T0"], I4, 0.10, np.deg2rad(60.0))
    native_icp_ms = min(native_icp_ms, 1e3 * (time.perf_counter() - t))
    return {"first_call_ms": sum(first), "repeated_ms": sum(rep),
            "first_call_cloud_builds": {"clouds": n_built, "host_copy_ms": 1e3 * build_s[0], "upload_and_bounds_ms": 1e3 * build_s[1], "cell_index_ms": 1e3 * build_s[2],
                                        "hilbert_order_and_tiles_ms": 1e3 * build_s[3],
                                        "note": "of first_call_ms; the rest is hashing the arrays for the cache's keys and the calls themselves"},
            "icp_align_same_call_on_resident_clouds_ms": native_icp_ms, "icp_align_iterations": int(it_n),
            "first_call_ms_by_consumer": dict(icp_align=first[0], alignment_scores=first[1], arrangement_to_labels=first[2]),
            "repeated_ms_by_consumer": dict(icp_align=rep[0], alignment_scores=rep[1], arrangement_to_labels=rep[2]),
            "note": "librescan_dropin.so, host arrays in / results out, the consumers one after the other; icp_align runs the reference's own loop "
                    "(stop test on: it converges where the timed step runs 10 fixed iterations)"}


def cpu_baseline(w, budget_s=20.0):
    """The reference itself (oracle/_ref, built from /root/reference in the build container and
    shipped as a .so), timed on this host on a bounded sample of the same workload.  Grids are
    prebuilt outside the timed region, mirroring 'inputs resident' on the GPU side."""
    import ctypes
    from oracle.pyoracle import Ref
    out = {}
    for omp in (True, False):
        if not Ref.available(omp=omp):
            continue
        nthr = 1
        if omp:
            # The reference's OpenMP split (msh_hash_grid.h:1122-1133) underflows `high_lim - low_lim`
            # when thread_idx * ceil(n/threads) > n, i.e. for many threads and query counts that are
            # not a multiple of the thread count.  Use a power-of-two thread count and trim every
            # sampled query set to a multiple of it, so the reference runs as its author intended.
            avail = len(os.sched_getaffinity(0))
            nthr = 1
            while nthr * 2 <= min(avail, 128):
                nthr *= 2
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(nthr)
        R = Ref(omp=omp)
        cores = R.num_threads()
        trim = lambda n: (n // nthr) * nthr  # noqa: E731
        s0, s1 = w["s0"], w["s1"]
        rng = np.random.default_rng(0)
        # ICP-NN: one find_corrs iteration on a query sample against the full target
        n_q = trim(min(len(s1["points"]), 200_000 if omp else 60_000))
        sel = np.sort(rng.choice(len(s1["points"]), n_q, replace=False))
        qp, qn = np.ascontiguousarray(s1["points"][sel]), np.ascontiguousarray(s1["normals"][sel])
        g0 = R.grid_create(s0["points"], 0.10)                     # icp.h:437 (radius = max_dist)
        t = time.perf_counter()
        R.icp_find_corrs_grid(g0, qp, qn, s0["points"], s0["normals"], w["icp_T0"], I4, 0.10, np.float32(np.deg2rad(60.0)))
        t_icp = (time.perf_counter() - t) / n_q
        R.grid_destroy(g0)
        # score-NN: a few poses of the same object against the full scene (grid radius 0.05)
        scn = R.scene_create(s1["points"], s1["normals"])
        n_p = 24 if omp else 8
        op, on = w["obj_score_np"]
        op, on = np.ascontiguousarray(op[: trim(len(op))]), np.ascontiguousarray(on[: trim(len(on))])
        t = time.perf_counter()
        R.alignment_scores(s1["points"], s1["normals"], op, on, w["score_poses"][:n_p], 64, scene=scn)
        t_score = (time.perf_counter() - t) / (n_p * len(op))
        R.scene_destroy(scn)
        # label-NN: the reference's own placement loop (rspf__assign_temporary_labels, rs_pointcloud_filters.cpp:738-778: transform,
        # K=1 search in the object's grid, gate, running minimum) for a sample of scene points, 2 placements
        from oracle.pyoracle import RefFilters
        from rescan_amd import synth
        n_l = trim(min(len(s1["points"]), 400_000))
        sel = np.sort(rng.choice(len(s1["points"]), n_l, replace=False))
        sp, sn = np.ascontiguousarray(s1["points"][sel]), np.ascontiguousarray(s1["normals"][sel])
        t_label = 0.0
        label_how = "rspf__assign_temporary_labels"
        if RefFilters.available(omp=omp):
            RF = RefFilters(synth.CLASS_IDX, omp=omp)
            plc2 = [dict(pose=p["pose"], object_idx=RF.add_object(p["np"][0], p["np"][1], p["cls"], 1000 + k), uidx=k)
                    for k, p in enumerate(w["plc"][:2])]
            lab = np.zeros(n_l, np.int8); mind = np.full(n_l, 1e9, np.float32)
            t = time.perf_counter()
            RF.assign(sp, sn, plc2, 0, 2, 0.05, lab, mind)
            t_label = time.perf_counter() - t
            RF.close()
        else:          # an oracle/_ref from before round 4: the loop's search alone (its gate loop omitted, which favours the CPU)
            label_how = "transform + K=1 search only"
            for p in w["plc"][:2]:
                g = R.grid_create(p["np"][0], 0.05)
                inv = R.mat4_inverse(p["pose"])
                t = time.perf_counter()
                q = R.xform_points(inv, sp, 1)
                R.radius_search(g, q, 0.05, 1, 0)
                t_label += time.perf_counter() - t
                R.grid_destroy(g)
        t_label /= 2 * n_l
        pr = w["pairs"]
        total = pr["icp"] + pr["score"] + pr["label"]
        est = pr["icp"] * t_icp + pr["score"] * t_score + pr["label"] * t_label
        out["omp" if omp else "single"] = dict(
            value=total / est, unit="point-pairs/s", cores=cores, kind="reference",
            sample=f"1 icp_find_corrs on {n_q} queries + {n_p} score poses x {len(op)} pts + 2 label placements x {n_l} pts ({label_how}); "
                   f"per-pair times ICP {t_icp*1e9:.0f} ns, score {t_score*1e9:.0f} ns, label {t_label*1e9:.0f} ns, "
                   f"combined in the GPU workload's mix")
    return out


def launch_ranks(n, argv, dry_run=False):
    """Starts bench.py once per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as under torch.distributed.run, rendezvous on
    127.0.0.1), relays rank 0's stdout — the ONE JSON line — and every rank's stderr, and returns the worst exit code.  The children
    are new processes started from a parent that never initialised the GPU; a rank that fails takes the others down with it
    (by PID), so a hung collective cannot outlive the failure."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    argv = [a for a in argv if a != "--launch-dry-run"]
    plans = []
    for r in range(n):
        env = dict(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"),
                   MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        plans.append(dict(cmd=[sys.executable, os.path.abspath(__file__)] + argv, env=env))
    if dry_run:
        print(json.dumps(dict(launcher="bench.py", ranks=plans)))
        return 0
    procs = []
    for r, pl in enumerate(plans):
        procs.append(subprocess.Popen(pl["cmd"], env=dict(os.environ, **pl["env"]), stdout=None if r == 0 else sys.stderr))
    worst, live, killed_at = 0, set(range(n)), 0.0
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                worst = worst or rc
                print("bench.py launcher: rank %d exited with %d, stopping the others" % (r, rc), file=sys.stderr)
                for q in live:
                    procs[q].terminate()
                killed_at = time.time()
        # a rank blocked inside a collective (a native call) may sit on SIGTERM: after 5 s it is killed by PID
        if worst and live and time.time() - killed_at > 5.0:
            for q in live:
                procs[q].kill()
        time.sleep(0.05)
    return worst if worst >= 0 else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)        # a step is ~3 ms: 20 of them average out the host-side jitter of a short run
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=1_000_000, help="points per scan")
    ap.add_argument("--knn", choices=["hash", "brute"], default="hash",
                    help="candidate layout: LDS spatial-hash cells (default) or one brute tile")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the warm-up and the timed steps: no serial re-timing, no drop-in block afterwards (profiling runs: "
                    "every kernel the process launches after its clouds are built then belongs to a step; RS_BENCH_MARK=1 starts each step with the "
                    "library's marker kernel)")
    ap.add_argument("--serial", action="store_true", help="issue the three consumers one after another")
    ap.add_argument("--shard", action="store_true", help="the sharded route (default for --gpus > 1) also at N = 1")
    ap.add_argument("--replicas", action="store_true", help="--gpus > 1: every rank its own scene (configs[4]) instead of one sharded scene")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="strong: ONE fixed scene (8 per-placement ICP problems, 256 score poses, 8 placements) sharded over the ranks; total work does not grow with --gpus")
    ap.add_argument("--centre", action="store_true",
                    help="the same scene moved so that its median point is the origin (coordinates of both signs, as real scans have): the reference's fp32 centroid sums hover around zero instead of growing; parity against tests/golden/bench_seed11_centre.npz")
    ap.add_argument("--strong-problems", type=int, default=N_PLACEMENTS,
                    help="--scaling strong: ICP problems of the fixed unit list (default 8: one refine per placement; the app's shape is many object-sized "
                         "refines per step — 43 icp_align per sequence — so e.g. 512: further start poses of the same eight models)")
    ap.add_argument("--timesteps", type=int, default=2,
                    help="scans of the sequence (BASELINE configs[3]: 4): a step then covers every consecutive pair (t, t + 1) — T - 1 times the unit lists, "
                         "each pair its own two scans, object poses and placements")
    ap.add_argument("--launch-dry-run", action="store_true",
                    help="--gpus N > 1 without a launcher: print the N ranks' command and environment as JSON instead of starting them")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver may type it: this process becomes the launcher — it has not touched the GPU (no
        # torch, no HIP yet) — and starts one FRESH process per GPU with the environment torch.distributed.run would give them.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], dry_run=args.launch_dry_run))

    if os.environ.get("RS_BENCH_LAUNCH_SELFTEST"):
        # CPU rehearsal of the launcher (tests/test_bench_launcher.py): the ranks it started find each other (gloo over 127.0.0.1) and
        # sum their ranks; "fail<r>": rank r dies first, the launcher must stop the others and hand its exit code on
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        if os.environ["RS_BENCH_LAUNCH_SELFTEST"] == "fail%d" % rank:
            sys.exit(3)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([rank + 1])
        dist.all_reduce(t)
        if os.environ["RS_BENCH_LAUNCH_SELFTEST"].startswith("die"):
            # "die<r>": the group has formed and exchanged once; rank r then dies WITHOUT entering the next collective while the others
            # block in it — the launcher must notice the death, stop the blocked ranks and hand the exit code on (no hang)
            if os.environ["RS_BENCH_LAUNCH_SELFTEST"] == "die%d" % rank:
                os._exit(5)
            big = torch.zeros(1 << 20)
            dist.all_reduce(big)          # never completes: a rank is gone
            time.sleep(600)               # (gloo may return with an error instead of blocking: the launcher still has to end this process)
            return
        if rank == 0:
            print(json.dumps({"selftest": int(t.item()), "world": world, "gpus": args.gpus}))
        dist.destroy_process_group()
        return

    # stdout carries the ONE JSON line and nothing else: whatever libraries print there (RCCL's version banner at the first
    # collective) goes to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start it plainly — it launches its own ranks — or under torch.distributed.run with --nproc-per-node equal to --gpus)" % (args.gpus, world))
    # (RS_BENCH_ONE_DEVICE=1: rehearsal of the multi-rank plumbing on a ONE-GPU box — every rank on device 0, exchange over gloo)
    one_device = bool(os.environ.get("RS_BENCH_ONE_DEVICE"))
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # RS_BENCH_FORCE_DIST=1: rehearse the RCCL exchange step on a one-GPU box (world size 1 under torch.distributed.run)
    if world > 1 or os.environ.get("RS_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if one_device:
            dist.init_process_group("gloo")            # RCCL refuses two ranks on one device
        else:
            dist.init_process_group("nccl", device_id=dev)

    from rescan_amd import capi
    capi.init(local_rank)

    strong = args.scaling == "strong"
    sharded = args.shard or strong or (world > 1 and not args.replicas)
    seed = 11 if sharded else 11 + rank
    units = 1 if strong else (world if sharded else 1)
    n_pairs = max(1, args.timesteps - 1)
    W = [build_workload(args.points, seed=seed, knn=args.knn, units=units, centre=args.centre, t0=k) for k in range(n_pairs)]
    n_strong = max(N_PLACEMENTS, args.strong_problems) if strong else N_PLACEMENTS
    for wk in W:
        if strong:      # the step's ICP units are the per-placement problems: model points x iterations
            wk["pairs"] = dict(icp=ICP_ITERS * sum(len(wk["plc"][k % N_PLACEMENTS]["np"][0]) for k in range(n_strong)), score=wk["pairs"]["score"], label=wk["pairs"]["label"])
    w = W[0]
    dist_ctx = (dist, dev) if dist is not None else None
    # RS_BENCH_SIM_WORLD=W (one process, no exchange): rank 0's share of a W-rank world, for the per-rank compute time of a split
    sim_world = int(os.environ.get("RS_BENCH_SIM_WORLD", "0")) if world == 1 else 0
    SH = [Sharded(wk, dist, dev, rank, sim_world or world, strong=strong, strong_problems=n_strong) for wk in W] if sharded else None
    sh = SH[0] if sharded else None

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    conc = not args.serial
    last = None            # the newest results of the FIRST pair
    last_of = {}           # ... and of every pair (each has its fixture: tests/golden/bench_seed11[_t<k>].npz)
    in_flight = []         # sharded route: pair whose exchange is under way

    mark_steps = bool(os.environ.get("RS_BENCH_MARK"))

    def one_step():
        """One pass over the sequence: every consecutive pair of scans in turn (one pair at --timesteps 2)."""
        nonlocal last
        if mark_steps:
            capi.profile_marker()                   # (rs::k_step_marker: where this step begins in a rocprofv3 kernel trace)
        for k in range(n_pairs):
            if SH is not None:
                r = SH[k].step(conc)                # (returns the results of the exchange that was in flight: the pair before)
                if r is not None and in_flight:
                    last_of[in_flight[-1]] = r
                in_flight[:] = [k]
            else:
                last_of[k] = run_step(W[k], dist_ctx, conc)
        last = last_of.get(0, last)

    def drain():
        nonlocal last
        r = exchange_wait()                         # the last step's exchange belongs to the timed region
        if SH is not None and r is not None and in_flight:
            last_of[in_flight[-1]] = r
        in_flight[:] = []
        last = last_of.get(0, last)

    # (rehearsal of the failure path, tools/r06_bench_modes.sh: RS_BENCH_DIE="<rank>:<warm-up step>" — that rank exits hard in the middle of
    #  the run, with the others' exchange of the step before in flight; the launcher must end them and return its code)
    die_rank, die_step = ([int(x) for x in os.environ["RS_BENCH_DIE"].split(":")] if os.environ.get("RS_BENCH_DIE") else (-1, -1))
    for k_w in range(args.warmup):
        if rank == die_rank and k_w == die_step:
            os._exit(7)
        one_step()
    drain()
    for x in (SH or []):
        x.reset_stats()
    capi.profile_enable(True)
    capi.profile_reset()
    import gc
    gc.collect(); gc.disable()                  # no collector pauses inside the timed region
    barrier()
    cpu_before = host_cpu_stat()
    t0 = time.perf_counter()
    marks = []
    for _ in range(args.steps):
        one_step()
        marks.append(time.perf_counter())       # (a step ends with its results on the host: no extra synchronisation)
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    cpu_after = host_cpu_stat()
    gc.enable()
    step_ms = np.diff(np.array([t0] + marks)) * 1e3
    if os.environ.get("RS_BENCH_PRINT_STEPS") and rank == 0:
        slow = [int(k) for k, v in enumerate(step_ms) if v > 1.3 * np.median(step_ms)]
        print("slow steps (index, ms):", [(k, round(float(step_ms[k]), 2)) for k in slow], file=sys.stderr)
        if _ROLE_TIMES:
            rt = np.array(_ROLE_TIMES[-len(step_ms):])
            print("  consumers' calls, ms (icp, score, label): median", np.round(np.median(rt, axis=0), 2).tolist(),
                  "| slow steps:", [(k, np.round(rt[k], 2).tolist()) for k in slow if k < len(rt)], file=sys.stderr)
    capi.profile_enable(False)
    if _ROLE_POOLS:
        _ROLE_POOLS[0].close()

    pairs_unit = sum(sum(wk["pairs"].values()) for wk in W)       # per step: every pair of the sequence
    pairs_total = float(pairs_unit * args.steps * (world if not sharded else units))
    if dist is not None and not sharded:
        # --replicas: every rank has its own scene (its own seed, its own point counts): the pairs of all ranks, summed
        tp = torch.tensor([float(pairs_unit * args.steps)], device=dev if dist.get_backend() != "gloo" else "cpu", dtype=torch.float64)
        dist.all_reduce(tp, op=dist.ReduceOp.SUM)
        pairs_total = float(tp.item())
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if dist.get_backend() != "gloo" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        if sh is not None:
            errs, Ts, its, scores, labels, mind = last
            out = dict(err=errs[0], T=Ts[0], scores=scores, labels=labels, min_dists=mind, Ts=Ts, errs=errs)
        else:
            out = last
        prof = {k: capi.profile_read(k) for k in ("nn_icp", "icp_moments", "nn_score", "nn_label")}
        per_step = {k: v[1] / max(1, v[0]) * (ICP_ITERS if k in ("nn_icp", "icp_moments") else 1) * n_pairs for k, v in prof.items()}
        shares = cu_shares() if conc else {}
        mean = lambda f: float(np.mean([f(wk) for wk in W]))
        # algorithmic bytes per launch (SURVEY.md §8d / BASELINE.md §3.5).  The label pass is ONE fused launch over all
        # placements: a scene point (24 B) is read once, not once per placement — 24 + n_placements x 22 B per point.
        alg_bytes = {"nn_icp": mean(lambda x: x["n_scan1"] * 56 + x["n_scan0"] * 16),
                     "nn_score": mean(lambda x: N_POSES * x["n_obj"] * 40 + x["n_scan1"] * 16),
                     "nn_label": mean(lambda x: x["n_scan1"] * (24 + N_PLACEMENTS * 22) + sum(len(p["np"][0]) for p in x["plc"][:N_PLACEMENTS]) * 16),
                     "icp_moments": mean(lambda x: x["n_scan1"] * 56)}
        by_kernel = {}
        for k, (n_k, ms_k) in prof.items():
            avg = (ms_k / max(1, n_k)) * 1e-3
            tr, _ = read_traffic(k)
            by_kernel[k] = {"avg_launch_ms": avg * 1e3, "launches": n_k, "alg_bytes_per_launch": alg_bytes[k], "cu_share": shares.get(k, 1.0),
                            "achieved_GBs": alg_bytes[k] / avg / 1e9 if avg > 0 else 0.0, "frac": (alg_bytes[k] / avg / 1e9 / HBM_PEAK_GBS) if avg > 0 else 0.0,
                            # HBM-side bytes per launch by the PMC counters (profiles/pmc_traffic.json, same kernel sources only) and the
                            # fraction of the HBM peak those bytes make of this run's launch time
                            "traffic": tr, "traffic_frac": (tr / avg / 1e9 / HBM_PEAK_GBS) if (tr and avg > 0) else None}
        # roofline.kernel: the domain with the LONGEST time per step (the top row of a rocprofv3 --stats summary of this command).
        # Round 4 reported the domain with the largest time x CU share instead; the rule now is fixed: longest interval.
        dom = max(per_step, key=lambda k: per_step[k])
        worst = min((k for k in by_kernel if by_kernel[k]["frac"] > 0), key=lambda k: by_kernel[k]["frac"], default=dom)
        n_l, ms = prof[dom]
        bytes_launch = alg_bytes[dom]
        avg_s = (ms / max(1, n_l)) * 1e-3
        achieved = bytes_launch / avg_s / 1e9 if avg_s > 0 else 0.0
        traffic, traffic_note = read_traffic(dom)
        if strong:
            wl = ("configs[3], strong scaling: ONE FIXED scene (~1M-pt scans) replicated per rank, its units sharded over the ranks: " + str(n_strong) + " ICP-NN problems "
                  "(each placement's ~50k-pt model -> scan, 0.075 / 50 deg, 10 it: lib/rs/rs_database.h:220-230), score-NN 256 poses x 10k, label-NN 8 placements; "
                  "per-placement rows all-gathered on the device, ordered fold")
        elif sharded:
            wl = ("configs[3]: ONE scene (~1M-pt scans) replicated per rank, units sharded over the ranks: %d x {ICP-NN 10 it x scan->scan, "
                  "score-NN 256 poses x 10k, label-NN 8 placements}; per-placement rows all-gathered on the device, ordered fold" % units)
        else:
            wl = ("configs[1]: single scene, 2 timesteps, ~1M-pt scans on 1xMI355X per rank: "
                  "ICP-NN 10 it x scan->scan + score-NN 256 poses x 10k + label-NN 8 placements")
        line = {
            "metric": "point-pairs/sec (ICP-NN + score-NN + segment-NN) per scene",
            "value": pairs_total / elapsed, "unit": "point-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl + (" [--timesteps %d: every step covers the %d consecutive scan pairs of the sequence, each pair its own scans, poses and placements]" % (args.timesteps, n_pairs) if n_pairs > 1 else "")
                                   + (" [--centre: the scene moved so that its median point is the origin]" if args.centre else ""),
                       "timesteps": args.timesteps, "strong_problems": n_strong if strong else None, "simulated_world": sim_world or None,
                       "rank0_compute_ms_per_step": (sum(x.t_compute for x in SH) / max(1, SH[0].step_index - SH[0].stat_from) * 1e3) if SH else None,
                       "knn": "lds-hash-cells" if args.knn == "hash" else "brute-tile",
                       "route": "sharded" if sharded else ("replicas" if world > 1 else "single"),
                       "n_scan0": w["n_scan0"], "n_scan1": w["n_scan1"], "n_obj": w["n_obj"],
                       "pairs_per_step": pairs_unit * (units if sharded else world), "pairs_split_per_unit": w["pairs"],
                       "issue": ("3 host threads / 3 HIP streams (ICP chain | score batch | label pass); CU partition: " + cu_partition_note()) if conc else "serial",
                       "exchange": ("one all_gather of the per-rank send buffers (poses, errors, scores, " + ("the rank's (min_dist, label) partial of its own run" if sh.lay.prefold else "per-placement rows") + ": %.1f MB per rank) per step, "
                                    "overlapped with the next step; ordered fold of the rows on the device; on the exchange thread: publish + all_gather %.3f ms, "
                                    "fold + download of poses / scores / labels %.3f ms per step; main thread: compute %.3f ms, waiting for the previous exchange %.3f ms per step"
                                    % (sh.lay.words * 4 / 1e6, sh.t_gather / max(1, sh.n_exchanges) * 1e3, sh.t_fold / max(1, sh.n_exchanges) * 1e3,
                                       sh.t_compute / max(1, sh.step_index - sh.stat_from) * 1e3, sh.t_wait / max(1, sh.step_index - sh.stat_from) * 1e3)) if sharded
                                   else ("one fused all_gather(poses, scores, label partials) per step, overlapped with the next step" if dist is not None else "none")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "avg_launch_ms": ms / max(1, n_l), "launches": n_l, "alg_bytes_per_launch": bytes_launch,
                         "dominant_by": "longest time per step", "ms_per_step": per_step[dom], "cu_share": shares.get(dom, 1.0),
                         "note": "achieved / frac compare a kernel confined to cu_share of the CUs with the whole chip's peak; roofline_by_kernel has every domain, each with its counter traffic",
                         # the longest SINGLE launch (the top row of `rocprofv3 --stats` of this command goes to its kernel) — named beside the
                         # longest domain so that neither reading of "dominant" hides the other
                         "longest_launch": (lambda k: {"kernel": k, "avg_launch_ms": by_kernel[k]["avg_launch_ms"], "frac": by_kernel[k]["frac"],
                                                        "traffic_frac": by_kernel[k]["traffic_frac"], "cu_share": by_kernel[k]["cu_share"]})(max(by_kernel, key=lambda k: by_kernel[k]["avg_launch_ms"])),
                         "furthest_below_roofline": {"kernel": worst, "frac": by_kernel[worst]["frac"], "instructions_per_launch": read_instructions(worst)}},
            "roofline_by_kernel": by_kernel,
            "parity": parity_block(out, args.points, seed, args.knn, units, strong, centre=args.centre) if not sim_world else "not compared: RS_BENCH_SIM_WORLD runs one rank's share without the exchange",
            "icp_chains_gave_up_calls": int(capi.icp_chains_gave_up()),     # calls whose centroid chains gave a problem up and were run again by the replay (0 on scenes in one octant)
            "ms_per_step_spread": {"min": float(step_ms.min()), "median": float(np.median(step_ms)), "max": float(step_ms.max()),
                                   "steps_over_1.3x_median": int((step_ms > 1.3 * np.median(step_ms)).sum())},
            # the container's CPU cgroup around the timed region: a throttled period stalls every host thread of the process
            "host": ({"cpu_max": cpu_before[0], "throttled_periods_in_timed_region": cpu_after[1] - cpu_before[1],
                      "throttled_usec_in_timed_region": cpu_after[2] - cpu_before[2]} if cpu_before and cpu_after else None),
            # per-launch averages x launches per step (the ICP loop's events are sampled: one call in RS_HIP_PROF_EVERY)
            "kernel_ms_per_step": per_step,
            "profile_sampling": "ICP chain: events on every %s-th call (%d launches timed); score / label: every call" % (os.environ.get("RS_HIP_PROF_EVERY", "1"), n_l),
        }
        if n_pairs > 1:      # the further scan pairs against their own reference fixtures
            def as_out(x):
                if x is None or SH is None:
                    return x
                errs, Ts, its, scores, labels, mind = x
                return dict(err=errs[0], T=Ts[0], scores=scores, labels=labels, min_dists=mind, Ts=Ts, errs=errs)
            line["parity_pairs"] = [parity_block(as_out(last_of.get(k)), args.points, seed, args.knn, units, strong, centre=args.centre, t0=k) for k in range(1, n_pairs)]
        # "the real limiter is candidate evaluation ... so also report candidate-evals/s"
        # (candidates staged in LDS x the 64 query lanes that test each of them)
        cand = capi.profile_read("candidates")[0]
        line["candidate_evals"] = {"per_step": cand * 64 / args.steps, "per_s": cand * 64 / elapsed,
                                   "candidates_staged_per_step": cand / args.steps,
                                   "per_point_pair": cand * 64 / args.steps / max(1, pairs_unit)}
        if world == 1 and not sharded and not args.no_extras:
            # outside the timed region: the same step with the consumers issued one after the other, and through the drop-in boundary
            try:
                capi.profile_enable(False)
                t_ser = []
                for _ in range(5):
                    t = time.perf_counter(); run_step(w, None, False); t_ser.append(time.perf_counter() - t)
                line["serial_ms_per_step"] = float(np.median(t_ser) * 1e3)
                line["dropin"] = dropin_block(w)
            except Exception as e:
                line["dropin"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                cb = cpu_baseline(w)
                if "omp" in cb:
                    line["cpu_baseline"] = cb["omp"]
                    if "single" in cb:
                        line["cpu_baseline_single_thread"] = cb["single"]
                elif "single" in cb:
                    line["cpu_baseline"] = cb["single"]
            except Exception as e:  # the baseline is reported, never required for the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "point-pairs/s", "cores": 0, "kind": "reference",
                                        "sample": f"unavailable: {e}"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


def read_instructions(kernel):
    """Wave-level instruction counts per launch of `kernel` (SQ_INSTS_VALU / SALU / LDS, SQ_WAVES, busy cycles) from the PMC pass
    (tools/profile.sh pmc -> profiles/pmc_instructions.json), or why they are not reported: like the traffic, only counters collected
    for the kernel sources of THIS library are."""
    ipath = os.path.join(ROOT, "profiles", "pmc_instructions.json")
    if not os.path.exists(ipath):
        return "no profiles/pmc_instructions.json"
    try:
        t = json.load(open(ipath))
    except Exception as e:
        return f"unreadable: {e}"
    from rescan_amd.build import sources_sha
    want = sources_sha()
    if t.get("kernels_sha") != want:
        return "profiles/pmc_instructions.json was collected for other kernel sources (kernels_sha %s, now %s): stale, not reported" % (t.get("kernels_sha"), want)
    return t.get(kernel)


def read_traffic(kernel):
    """HBM bytes per launch of `kernel` from the PMC passes (tools/profile.sh traffic -> profiles/pmc_traffic.json).
    Counters cannot be collected inside this process (rocprofv3 wraps the command), so the file is a separate run of the
    same command; it is REFUSED when it is older than the library it claims to describe."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tpath):
        return None, "no profiles/pmc_traffic.json"
    try:
        t = json.load(open(tpath))
    except Exception as e:
        return None, f"unreadable: {e}"
    from rescan_amd.build import sources_sha
    want = sources_sha()
    if t.get("kernels_sha") != want:
        return None, "profiles/pmc_traffic.json was collected for other kernel sources (kernels_sha %s, now %s): stale, not reported" % (t.get("kernels_sha"), want)
    return t.get(kernel), "profiles/pmc_traffic.json (tools/profile.sh traffic, same kernel sources: %s)" % want


if __name__ == "__main__":
    main()
