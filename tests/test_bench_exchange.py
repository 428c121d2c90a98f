"""CPU, world_size 2, gloo: bench.py's multi-GPU exchange step (all-gather of poses/scores and of
label partials of unequal length) runs and delivers every rank's data to every rank."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch
        import bench
        n = 1000 + 37 * rank                                   # scenes differ in size across ranks
        T = np.arange(16, dtype=np.float32) + rank
        scores = np.full(8, 0.5 + rank, np.float32)
        res = dict(labels=np.full(n, rank + 1, np.int8), min_dists=np.full(n, 0.25 * (rank + 1), np.float32))
        out, gl, gm = bench.exchange_results(dist, torch.device("cpu"), T, 0.125 * (rank + 1), scores, res)
        ok = len(out) == world and len(gl) == world
        for r in range(world):
            o = out[r].numpy()
            ok &= bool((o[:16] == np.arange(16) + r).all() and o[16] == np.float32(0.125 * (r + 1)) and (o[17:] == 0.5 + r).all())
            nr = 1000 + 37 * r
            ok &= bool((gl[r].numpy()[:nr] == r + 1).all() and (gm[r].numpy()[:nr] == np.float32(0.25 * (r + 1))).all())
            ok &= gl[r].numel() == 1000 + 37 * (world - 1)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_exchange_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_role_runner_joins_three_consumers_without_sleeping():
    """bench.py's RoleRunner in its spinning form (no GPU needed: the hand-off flags are host memory): 200 steps of three
    callables, each always on its own thread, results in order; an exception raised by a worker's callable reaches the caller;
    close() parks the workers."""
    import threading
    sys.path.insert(0, ROOT)
    import bench
    from rescan_amd import build
    build.build()
    r = bench.RoleRunner(None, True)
    assert r.workers == [bench.RoleRunner.SCORE, bench.RoleRunner.LABEL]        # released in this order
    names = {}

    def fn(tag, k):
        def run():
            names.setdefault(tag, set()).add(threading.current_thread().name)
            return (tag, k)
        return run

    for k in range(200):
        out = r.run3(fn("icp", k), fn("score", k), fn("label", k))
        assert out == [("icp", k), ("score", k), ("label", k)]
    assert all(len(v) == 1 for v in names.values()) and len({next(iter(v)) for v in names.values()}) == 3
    assert next(iter(names["icp"])) == threading.current_thread().name            # the longest consumer (role 0 by default) runs on the caller

    def boom():
        raise ValueError("from the label thread")
    with pytest.raises(ValueError):
        r.run3(fn("icp", -1), fn("score", -1), boom)
    assert r.run3(fn("icp", 7), fn("score", 7), fn("label", 7))[2] == ("label", 7)   # still usable
    r.close()
    for t in r.threads:
        t.join(10.0)
        assert not t.is_alive()
