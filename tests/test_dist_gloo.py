"""CPU, world_size 2, gloo: the N>1 sharding of the hot path's independent units and the ordered
fold of label partials (rescan_amd/dist.py).  The per-rank compute is played by the CPU oracle
(tests may use it); on the GPU box the same functions are driven by the HIP path (bench.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import label_case, load_golden
        from oracle.pyoracle import Oracle
        from rescan_amd import dist as rd
        O = Oracle()
        d = load_golden("scene.npz")
        gscene = dict(points=d["points"], normals=d["normals"], objects=[
            dict(pos=d[f"obj{i}_pos"], nor=d[f"obj{i}_nor"], pose=d["obj_pose"][i]) for i in range(int(d["n_obj"]))])
        to_t = torch.from_numpy
        I4 = np.eye(4, dtype=np.float32).ravel()

        # --- labels: contiguous runs of the SORTED arrangement per rank, rank-ordered fold
        lab, objs, plcs = label_case(gscene, "labels_mixed.npz")
        order = lab["order"]
        sorted_plcs = [plcs[i] for i in order]
        first_static = next((k for k, p in enumerate(sorted_plcs) if objs[p["object_idx"]]["is_static"]), 0)

        def partial(lo, hi):
            # each placement alone against fresh state gives its unary row; fold the run in order
            mind = np.full(len(gscene["points"]), 1e9, np.float32); labels = np.zeros(len(mind), np.int8)
            for k in range(lo, hi):
                o = objs[sorted_plcs[k]["object_idx"]]
                one = O.arrangement_to_labels(gscene["points"], gscene["normals"],
                                              [dict(pos=o["pos"], nor=o["nor"], class_idx=0, is_static=0)],
                                              [dict(pose=sorted_plcs[k]["pose"], object_idx=0, uidx=0)],
                                              # a lone dynamic placement is searched at 1.5*radius (first_static = 0)
                                              (0.05 if k < first_static else 0.075) / 1.5, 0, 0)
                take = (one["labels"] > 0) & (one["min_dists"] < mind)
                mind[take] = one["min_dists"][take]; labels[take] = k + 1
            return mind, labels

        mind, labels = rd.sharded_label_transfer(dist, rank, world, len(gscene["points"]), len(sorted_plcs), partial, to_t)
        ok_labels = bool((labels == lab["labels"]).all() and (mind == lab["min_dists"]).all())

        # --- ICP problems and score poses: sharded, gathered in order
        g = load_golden("icp_chair1_pp.npz")
        o = gscene["objects"][int(g["obj"])]
        rng = np.random.default_rng(0)
        from rescan_amd import synth
        T0s = np.stack([g["T1"]] + [synth.perturbed_pose(o["pose"], rng) for _ in range(2)])

        def run_batch(Ts):
            out = [O.icp_align(o["pos"], o["nor"], gscene["points"], gscene["normals"], T, I4, 0.1, g["max_angle"]) for T in Ts]
            return (np.array([x[0] for x in out], np.float32), np.stack([x[1] for x in out]),
                    np.array([x[2] for x in out], np.int32))

        errs, Ts, its = rd.sharded_icp(dist, rank, world, T0s, run_batch, to_t)
        ok_icp = bool((Ts[0] == g["T_out"]).all() and errs[0] == g["err"] and its[0] == int(g["iters"]) and len(Ts) == 3)

        sg = load_golden("scores_chair1_k64.npz")
        so = gscene["objects"][int(sg["obj"])]
        sc = rd.sharded_scores(dist, rank, world, sg["poses"],
                               lambda P: O.alignment_scores(gscene["points"], gscene["normals"], so["pos"], so["nor"], P, 64), to_t)
        ok_sc = bool((sc == sg["scores"]).all())
        q.put((rank, ok_labels, ok_icp, ok_sc))
    finally:
        dist.destroy_process_group()


def test_shard_range_covers_everything():
    from rescan_amd.dist import shard_range
    for n in (0, 1, 5, 8, 17):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_fold_is_order_dependent_like_the_reference():
    from rescan_amd.dist import fold_label_partials
    pm = np.array([[1.0, 2.0, 1e9], [1.0, 1.0, 5.0]], np.float32)
    pl = np.array([[1, 1, 0], [2, 2, 2]], np.int8)
    mind, lab = fold_label_partials(pm, pl)
    assert lab.tolist() == [1, 2, 2] and mind.tolist() == [1.0, 1.0, 5.0]     # tie -> earlier rank keeps it


@pytest.mark.timeout(600)
def test_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok_labels, ok_icp, ok_sc in res:
        assert ok_labels, f"rank {rank}: folded labels differ from the sequential loop"
        assert ok_icp, f"rank {rank}: gathered ICP results differ"
        assert ok_sc, f"rank {rank}: gathered scores differ"


def test_shard_layout_and_arrangement_plan():
    """Pure bookkeeping of the HIP-driven sharded route (rescan_amd/dist.py): every unit is owned by exactly one rank, the send
    buffers have one fixed size, and the arrangement plan is rspf_arrangement_to_labels' order (stable sort by
    (is_static << 10 | class), first static entry or 0, 1.5 x radius from there on)."""
    from rescan_amd.dist import ShardLayout, arrangement_plan, shard_range
    for world in (1, 2, 3, 8):
        lay = ShardLayout(world, n_icp=5, n_score=257, n_plc=11, n_scene=1000)
        seen = [[0] * 5, [0] * 257, [0] * 11]
        for r in range(world):
            for k, (lo, hi) in enumerate(lay.slices(r)):
                for u in range(lo, hi):
                    seen[k][u] += 1
                cap = (lay.icp_cap, lay.score_cap, lay.plc_cap)[k]
                assert hi - lo <= cap
        assert all(all(v == 1 for v in s) for s in seen)
        assert lay.words == lay.off_rows + lay.plc_cap * 1000 and lay.off_rows % 64 == 0 and lay.off_score == 18 * lay.icp_cap
    order, first_static, radii = arrangement_plan([0, 1, 0, 1, 0], [7, 1, 5, 2, 5], 0.05)
    assert order == [2, 4, 0, 1, 3] and first_static == 3
    assert [float(r) for r in radii] == [float(np.float32(0.05))] * 3 + [float(np.float32(1.5) * np.float32(0.05))] * 2
    order, first_static, radii = arrangement_plan([0, 0], [3, 2], 0.05)           # no static placement: everything at 1.5 x radius
    assert order == [1, 0] and first_static == 0 and all(float(r) == float(np.float32(1.5) * np.float32(0.05)) for r in radii)
