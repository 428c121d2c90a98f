"""Repeat / stress tests of the places where a bit-exact contract can break only sometimes (round 2 found a 1-in-25 race in
the parallel reference-order estimator and withdrew a polled zero-copy path whose rows arrived late once in ~60 suite runs):
a single pass of the suite cannot see such failures, so these run the same inputs many times — from one thread and from
several — and demand identical bits every time.  Sized to finish within about a minute on the GPU box.

* the multi-source ICP batch (a different source per problem, bound per problem on the device) against the single runs, 32 times;
* the batched ICP with the parallel reference-order estimator ("replay") against the one-problem-at-a-time runs, 32 times;
* replay against the sequential chains on a scan-sized source, 32 times;
* small radius searches through the zero-copy (pinned-block) route of k_rows_wave from three host threads at once, each
  thread with its own queries, 300 calls per thread, every call compared with the first answer;
* the three consumers of bench.py's step issued from three threads against the same step issued serially;
* the score batch's two routes (object tiles / queries sorted by scene block) on random batches: every object, near and far poses, three grids;
* the time of a whole-scan ICP on a room moved to the origin (centroid sums that hover around zero: the grid chains give up, the
  replay takes over) against the same room as generated: a bounded multiple, and the attempt is not paid again on the next calls.
"""
import threading

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

I4 = np.eye(4, dtype=np.float32).ravel()
REPEATS = 32


@pytest.fixture(scope="module")
def capi():
    from rescan_amd import capi
    capi.init(0)
    return capi


@pytest.fixture(scope="module")
def clouds(capi, gscene):
    scn = capi.Cloud(gscene["points"], gscene["normals"], cell_size=0.2)
    objs = [capi.Cloud(o["pos"], o["nor"], cell_size=0.1) for o in gscene["objects"]]
    return scn, objs


def _key(err, T, it):
    return np.float32(err).tobytes() + np.asarray(T, np.float32).tobytes() + np.int32(it).tobytes()


def test_multi_source_batch_matches_single_repeatedly(capi, gscene, clouds):
    """rs_hip_icp_align_multi — every problem its own source cloud, bound on the device per problem (queues, slow-tile lists,
    certificates and records at per-problem offsets) — REPEATS times against the problems run alone: identical bits every time, also
    when the same workspace has just served a batch of another shape."""
    from rescan_amd import synth
    scn, objs = clouds
    rng = np.random.default_rng(8)
    order = [1, 0, 2, 0, 1]
    T0s = np.stack([synth.perturbed_pose(gscene["objects"][k]["pose"], rng) for k in order])
    single = [_key(*capi.icp_align(objs[k], scn, T0s[j], I4, 0.1, np.deg2rad(60.0))) for j, k in enumerate(order)]
    bad = []
    for rep in range(REPEATS):
        errs, Ts, its = capi.icp_align_multi([objs[k] for k in order], scn, T0s, I4, 0.1, np.deg2rad(60.0))
        bad += [(rep, j) for j in range(len(order)) if _key(errs[j], Ts[j], its[j]) != single[j]]
        if rep % 4 == 3:          # another shape in between: two problems, the other way round
            e2, T2, i2 = capi.icp_align_multi([objs[order[4]], objs[order[1]]], scn, T0s[[4, 1]], I4, 0.1, np.deg2rad(60.0))
            bad += [(rep, -1) for a, j in ((0, 4), (1, 1)) if _key(e2[a], T2[a], i2[a]) != single[j]]
    assert not bad, f"(repeat, problem) that differed: {bad}"


def test_parallel_chains_batch_matches_single_repeatedly(capi, gscene, clouds):
    """tests/test_gpu_parity.py::test_icp_batch_matches_single[parallel chains], REPEATS times: the batch, whose problems
    advance side by side through the replay kernels, returns the bits of the problems run alone — every time."""
    from rescan_amd import synth
    scn, objs = clouds
    o = gscene["objects"][1]
    rng = np.random.default_rng(3)
    T0s = np.stack([synth.perturbed_pose(o["pose"], rng) for _ in range(5)])
    prev, prev_r = capi.icp_reference_order_below(-1), capi.icp_replay_below(-1)
    try:
        capi.icp_reference_order_below(0); capi.icp_replay_below(1 << 30)
        single = [_key(*capi.icp_align(objs[1], scn, T0s[k], I4, 0.1, np.deg2rad(60.0))) for k in range(5)]
        bad = []
        for rep in range(REPEATS):
            errs, Ts, its = capi.icp_align_batch(objs[1], scn, T0s, I4, 0.1, np.deg2rad(60.0))
            bad += [(rep, k) for k in range(5) if _key(errs[k], Ts[k], its[k]) != single[k]]
            if rep % 8 == 7:                                     # the single runs must not drift either
                bad += [(rep, -1 - k) for k in range(5) if _key(*capi.icp_align(objs[1], scn, T0s[k], I4, 0.1, np.deg2rad(60.0))) != single[k]]
        assert not bad, f"(repeat, problem) pairs that differ: {bad}"
        capi.icp_reference_order_below(1 << 30)                  # and they are the sequential chains' bits
        assert [_key(*capi.icp_align(objs[1], scn, T0s[k], I4, 0.1, np.deg2rad(60.0))) for k in range(5)] == single
    finally:
        capi.icp_reference_order_below(prev); capi.icp_replay_below(prev_r)


def test_replay_matches_sequential_chains_repeatedly(capi):
    """A scan-sized source (the replay's own regime: thousands of segments, hundreds of superblocks per accumulator):
    REPEATS runs of the parallel chains return the one result of the sequential chains."""
    from rescan_amd import synth
    s0 = synth.scene_for_point_count(90_000, seed=29, timestep=0)
    s1 = synth.scene_for_point_count(90_000, seed=29, timestep=1)
    a, b = capi.Cloud(s0["points"], s0["normals"]), capi.Cloud(s1["points"], s1["normals"])
    T0 = synth.perturbed_pose(I4, np.random.default_rng(4), 0.02, 0.02)
    prev, prev_r = capi.icp_reference_order_below(-1), capi.icp_replay_below(-1)
    try:
        capi.icp_reference_order_below(1 << 30)
        ref = _key(*capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=8, fixed_iters=True))
        capi.icp_reference_order_below(0); capi.icp_replay_below(1 << 30)
        bad = [rep for rep in range(REPEATS) if _key(*capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=8, fixed_iters=True)) != ref]
        assert not bad, f"repeats that differ from the sequential chains: {bad}"
    finally:
        capi.icp_reference_order_below(prev); capi.icp_replay_below(prev_r)
        a.close(); b.close()


def test_small_searches_from_three_threads(capi, gscene):
    """The unchanged apps' call pattern (~130 queries per call, K = 64: one launch on device-visible pinned blocks, one
    synchronisation — DESIGN.md §3 k_rows_wave) from three host threads at once, each with its own stream, pinned blocks
    and queries: 300 calls per thread, every answer identical to the thread's first, and the first identical to the same
    call made alone."""
    scn = capi.Cloud(gscene["points"], None, cell_size=0.1)
    rng = np.random.default_rng(12)
    n_threads, n_calls = 3, 300
    queries = []
    for t in range(n_threads):
        base = gscene["points"][rng.integers(0, len(gscene["points"]), 130)]
        queries.append(np.ascontiguousarray(base + rng.normal(0, 0.01, base.shape).astype(np.float32), np.float32))
    alone = [capi.radius_search(scn, q, 0.1, 64) for q in queries]
    errors = []

    def worker(t):
        try:
            for c in range(n_calls):
                d, i, nn, _ = capi.radius_search(scn, queries[t], 0.1, 64)
                if not (np.array_equal(nn, alone[t][2]) and np.array_equal(d, alone[t][0]) and np.array_equal(i, alone[t][1])):
                    errors.append((t, c)); return
        except Exception as e:                                  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    scn.close()
    assert not errors, f"(thread, call) that differed: {errors}"


def test_three_consumers_side_by_side_repeatedly(capi):
    """bench.py's step (ICP chain, score batch, label
    pass) issued from three threads returns the bits of the step issued serially, 12 times in a row."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    w = bench.build_workload(200_000, 11, "hash")
    ref = bench.run_step(w, concurrent=False)

    def same(got):
        return (np.float32(got["err"]) == np.float32(ref["err"]) and (np.asarray(got["T"]) == np.asarray(ref["T"])).all()
                and (got["scores"] == ref["scores"]).all() and (got["labels"] == ref["labels"]).all()
                and (got["min_dists"] == ref["min_dists"]).all())

    try:
        bad = [k for k in range(12) if not same(bench.run_step(w, concurrent=True))]
        bad += [-1 - k for k in range(2) if not same(bench.run_step(w, concurrent=False))]
    finally:
        bench.close_roles()                                      # (the runner's worker threads busy-wait between steps)
    assert not bad, f"steps that differ: {bad}"


def test_score_routes_agree_on_random_batches(capi, gscene, clouds):
    """The scene-space score route (queries of every pose sorted by scene block and normal direction; round 5) against the object-space
    launch on batches it was not tuned on: every object of the scene, poses from a hair off the true one to metres away and rotated about
    all three axes, radii / K of the reference's three scoring call sites, three scene grids (the reference's 2 r cells, a finer one, the
    density-derived one), batch sizes that end inside a wave — bit for bit, and repeated (the route's sort is not stable against ties in
    the KEY, which must not matter)."""
    from rescan_amd import synth
    scn_ref, objs = clouds
    scenes = [scn_ref, capi.Cloud(gscene["points"], gscene["normals"], cell_size=0.05), capi.Cloud(gscene["points"], gscene["normals"])]
    rng = np.random.default_rng(2025)
    prev = capi.score_scene_space_from(-1)
    bad = []
    try:
        for trial in range(24):
            k = trial % len(objs)
            o = gscene["objects"][k]
            n_poses = int(rng.choice([1, 3, 17, 64, 129]))
            poses = []
            for _ in range(n_poses):
                scale = float(rng.choice([0.01, 0.1, 0.5, 3.0]))
                P = synth.perturbed_pose(o["pose"], rng, 0.6 * min(1.0, scale * 4), 0.25 * scale).reshape(4, 4).T.astype(np.float64)
                ax = rng.normal(size=3); ax /= np.linalg.norm(ax); a = rng.uniform(-0.5, 0.5) * min(1.0, scale * 4)
                Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
                R = np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * Kx @ Kx
                P[:3, :3] = P[:3, :3] @ R
                poses.append(np.ascontiguousarray(P.T.astype(np.float32).ravel()))
            poses = np.stack(poses)
            radius, K = [(0.1, 64), (0.1, 32), (0.05, 64)][trial % 3]
            scn = scenes[trial % 3]
            capi.score_scene_space_from(1 << 60)
            a = capi.alignment_scores(objs[k], scn, poses, radius, K)
            capi.score_scene_space_from(0)
            b = capi.alignment_scores(objs[k], scn, poses, radius, K)
            c = capi.alignment_scores(objs[k], scn, poses, radius, K)
            if not ((a.view(np.uint32) == b.view(np.uint32)).all() and (b.view(np.uint32) == c.view(np.uint32)).all()):
                bad.append((trial, k, n_poses, radius, K, float(np.abs(a - b).max())))
        assert not bad, bad
    finally:
        capi.score_scene_space_from(prev)
        for c in scenes[1:]:
            c.close()


def test_centroid_chains_sweep_vs_replay(capi):
    """The grid chains (the default estimator of scan-sized sources) against pass 2 of the replay on whole icp_align runs: two rooms,
    each as generated (sums that grow) and moved so that its median point is the origin (sums that wander around zero: many
    more binade changes, ties in long runs of blocks, now and then a walk that gives up and hands the call to the replay), five
    start poses each.  Bit-identical poses, errors and iteration counts throughout — a run of 37 blocks whose tie corrections
    outgrew their four bits was once taken whole with the corrections clamped: 4 ulps in one centroid sum, 7e-8 in the pose,
    in four of five start poses of the centred room and in none of the others."""
    from rescan_amd import synth
    prev, prev_r, prev_c, prev_w = capi.icp_reference_order_below(-1), capi.icp_replay_below(-1), capi.icp_exact_centroids(-1), capi.icp_chains_retry_after(-1)
    bad, gave_up = [], 0
    try:
        capi.icp_reference_order_below(0); capi.icp_replay_below(0)
        # every mode-1 call TRIES the chains (a source that gave up would otherwise go straight to the replay on its next calls, and the
        # centred clouds' later start poses would compare the replay with itself)
        capi.icp_chains_retry_after(0)
        for seed in (22, 23):
            s0 = synth.scene_for_point_count(330_000, seed=seed, timestep=0)
            s1 = synth.scene_for_point_count(330_000, seed=seed, timestep=1)
            for centred in (True, False):
                shift = -np.median(s1["points"], axis=0).astype(np.float32) if centred else np.zeros(3, np.float32)
                a, b = capi.Cloud(s0["points"] + shift, s0["normals"]), capi.Cloud(s1["points"] + shift, s1["normals"])
                rng = np.random.default_rng(5)
                for trial in range(5):
                    T0 = synth.perturbed_pose(I4, rng, 0.02, 0.01)
                    res = {}
                    for mode in (2, 1):
                        capi.icp_exact_centroids(mode)
                        g = capi.icp_chains_gave_up()
                        res[mode] = capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=6, fixed_iters=True)
                        gave_up += capi.icp_chains_gave_up() - g
                    if _key(*res[1]) != _key(*res[2]):
                        bad.append((seed, centred, trial))
                a.close(); b.close()
        print(f"calls the chains gave up: {gave_up} of 20")
        assert not bad, f"(seed, centred, start pose) that differ: {bad}"
        assert gave_up < 20          # (some of the twenty calls were decided by the chains themselves)
    finally:
        capi.icp_reference_order_below(prev); capi.icp_replay_below(prev_r); capi.icp_exact_centroids(prev_c); capi.icp_chains_retry_after(prev_w)


def test_centred_room_costs_a_bounded_multiple(capi):
    """DESIGN.md §4, "the cliff": on a scan whose coordinates straddle the origin two of the seven centroid sums hover around zero —
    thousands of their 18 000 segments change binade, and the records kept from the last iteration are for the wrong binades — so
    the walk gives up and the call is run again with those sums by pass 2 of the replay: same bits (test_centroid_chains_sweep_vs_
    replay), 3.6 times the time per iteration at 1.15 M points.  Held here: the multiple stays below 5, and a source that gave up
    does not pay for the attempt again on its next calls (rs_hip_icp_chains_gave_up stands still)."""
    from rescan_amd import synth
    s0 = synth.scene_for_point_count(980_000, seed=11, timestep=0)
    s1 = synth.scene_for_point_count(980_000, seed=11, timestep=1)
    T0 = synth.perturbed_pose(I4, np.random.default_rng(1), 0.01, 0.01)
    per_iter = {}
    n_timed = min(4, max(1, capi.icp_chains_retry_after(-1)))          # (within the calls a source that gave up skips the attempt)
    for name in ("as generated", "centred"):
        shift = -np.median(s1["points"], axis=0).astype(np.float32) if name == "centred" else np.zeros(3, np.float32)
        a, b = capi.Cloud(s0["points"] + shift, s0["normals"]), capi.Cloud(s1["points"] + shift, s1["normals"])
        capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=10, fixed_iters=True)
        g0 = capi.icp_chains_gave_up()
        # DEVICE time of the calls' kernels (the library's own events), not the host's clock: the hosts of this pool lose milliseconds
        # to scheduling now and then
        capi.profile_enable(True); capi.profile_reset()
        for _ in range(n_timed):
            capi.icp_align(b, a, T0, I4, 0.1, np.deg2rad(60.0), max_iter=10, fixed_iters=True)
        ms = sum(capi.profile_read(k)[1] for k in ("nn_icp", "icp_moments"))
        capi.profile_enable(False)
        per_iter[name] = (ms * 1e-3 / (10 * n_timed), capi.icp_chains_gave_up() - g0)
        a.close(); b.close()
    print("us per iteration (device):", {k: round(v[0] * 1e6, 1) for k, v in per_iter.items()}, "give-ups in the timed calls:", {k: v[1] for k, v in per_iter.items()})
    assert per_iter["as generated"][1] == 0 and per_iter["centred"][1] == 0
    assert per_iter["centred"][0] < 5.0 * per_iter["as generated"][0], per_iter
