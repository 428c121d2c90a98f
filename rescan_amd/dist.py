"""Multi-GPU sharding of the hot path's independent units.

One process per GPU.  The three consumers produce independent units — (object, start pose) ICP
problems, (object, pose) score evaluations, per-placement label rows — so ranks work on disjoint
contiguous slices with no data-path collective; the only exchange is an all-gather of the small
results (4x4 poses + errors, score vectors, per-rank label partials).  `torch.distributed` is the
plumbing (backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests).

The label combine keeps the reference's order dependence
(lib/rs/rs_pointcloud_filters.cpp:763: strict `<`, so the earlier placement wins a tie): ranks own
contiguous runs of the *sorted* arrangement, each reduces its run to a (min_dist, label) partial,
and partials are folded in rank order with the same strict `<`.
"""
import numpy as np


def shard_range(n_units, rank, world):
    """Contiguous slice [lo, hi) of n_units owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_units), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_ragged(dist, local, counts, device=None):
    """All-gather rows of a 2-D float tensor whose leading sizes differ per rank (counts[r] rows)."""
    import torch
    world = len(counts)
    width = local.shape[1]
    m = max(max(counts), 1)
    pad = torch.zeros((m, width), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: counts[r]] for r, o in enumerate(out)], dim=0)


def fold_label_partials(part_min, part_label):
    """part_min: (world, n) float32, part_label: (world, n) int8 — rank-ordered partials of the
    label loop.  Returns the (min_dists, labels) the sequential loop would have produced."""
    mind = np.full(part_min.shape[1], 1e9, np.float32)
    labels = np.zeros(part_min.shape[1], np.int8)
    for r in range(part_min.shape[0]):
        take = part_min[r] < mind
        mind[take] = part_min[r][take]
        labels[take] = part_label[r][take]
    return mind, labels


def sharded_label_transfer(dist, rank, world, n_scene, n_placements, compute_partial, to_tensor):
    """Each rank reduces its contiguous run of the sorted arrangement with
    compute_partial(lo, hi) -> (min_dists float32[n_scene], labels int8[n_scene]) (labels already
    offset by lo), all-gathers the partials and folds them in rank order."""
    import torch
    lo, hi = shard_range(n_placements, rank, world)
    if hi > lo:
        pmin, plab = compute_partial(lo, hi)
    else:
        pmin, plab = np.full(n_scene, 1e9, np.float32), np.zeros(n_scene, np.int8)
    tmin = to_tensor(np.ascontiguousarray(pmin, np.float32))
    tlab = to_tensor(np.ascontiguousarray(plab, np.int8))
    gmin = [torch.empty_like(tmin) for _ in range(world)]
    glab = [torch.empty_like(tlab) for _ in range(world)]
    dist.all_gather(gmin, tmin)
    dist.all_gather(glab, tlab)
    part_min = np.stack([t.cpu().numpy() for t in gmin])
    part_label = np.stack([t.cpu().numpy() for t in glab])
    return fold_label_partials(part_min, part_label)


def sharded_icp(dist, rank, world, T0s, run_batch, to_tensor):
    """T0s: (n,16) start poses.  run_batch(T0s[lo:hi]) -> (errs, Ts, iters).  Returns the gathered
    (errs, Ts, iters) for all n problems on every rank."""
    import torch
    n = len(T0s)
    counts = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
    lo, hi = shard_range(n, rank, world)
    if hi > lo:
        errs, Ts, its = run_batch(T0s[lo:hi])
        local = np.concatenate([Ts.reshape(-1, 16), errs.reshape(-1, 1), its.reshape(-1, 1).astype(np.float32)], axis=1)
    else:
        local = np.zeros((0, 18), np.float32)
    g = all_gather_ragged(dist, to_tensor(np.ascontiguousarray(local, np.float32)), counts).cpu().numpy()
    return g[:, 16].copy(), g[:, :16].copy(), g[:, 17].astype(np.int32)


def sharded_scores(dist, rank, world, poses, run_scores, to_tensor):
    """poses: (n,16).  run_scores(poses[lo:hi]) -> float32[hi-lo].  Returns all n scores."""
    n = len(poses)
    counts = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
    lo, hi = shard_range(n, rank, world)
    local = run_scores(poses[lo:hi]).reshape(-1, 1) if hi > lo else np.zeros((0, 1), np.float32)
    g = all_gather_ragged(dist, to_tensor(np.ascontiguousarray(local, np.float32)), counts)
    return g.cpu().numpy()[:, 0].copy()
