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


# ---------------------------------------------------------------------------------------------
# The same split driven by the HIP path (bench.py --shard, tests/test_gpu_parity.py): one scene
# replicated on every GPU, the units of the three consumers sharded, per-placement label rows
# written by the kernel straight into the rank's send buffer, ONE all-gather, ordered fold.
# ---------------------------------------------------------------------------------------------

def arrangement_plan(is_static, class_idx, radius):
    """The order and per-placement radii of rspf_arrangement_to_labels with prioritize_static = 0 (its only call
    site, apps/segment_transfer/main.cpp:389): stable sort by (is_static << 10 | class_idx)
    (lib/rs/rs_pointcloud_filters.cpp:724-736,823-827), first static entry or 0 (:830-835), radius for the
    dynamic run and 1.5 x radius from the first static entry on (:837-848).  Folding the rows of the sorted
    placements in order with strict `<` from (label 0, 1e9) is then exactly the two passes."""
    key = [(int(s) << 10) | int(c) for s, c in zip(is_static, class_idx)]
    order = sorted(range(len(key)), key=lambda i: key[i])          # Python's sort is stable
    first_static = next((k for k, i in enumerate(order) if is_static[i]), 0)
    radii = [np.float32(radius) if k < first_static else np.float32(1.5) * np.float32(radius) for k in range(len(order))]
    return order, first_static, radii


class ShardLayout:
    """Where a rank's results sit in its send buffer (float32 words), identical on every rank:
         [ icp: n_icp_max x 18 (pose 16, err, iterations) | scores: n_score_max | rows: n_plc_max x n_scene ]
    n_*_max = the largest slice any rank owns, so the all-gather is one fixed-size collective."""

    def __init__(self, world, n_icp, n_score, n_plc, n_scene, prefold=False):
        self.world, self.n_icp, self.n_score, self.n_plc, self.n_scene = int(world), int(n_icp), int(n_score), int(n_plc), int(n_scene)
        cap = lambda n: max(shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world))  # noqa: E731
        self.icp_cap, self.score_cap, self.plc_cap = cap(n_icp), cap(n_score), cap(n_plc)
        self.off_icp = 0
        self.off_score = self.off_icp + 18 * self.icp_cap
        self.off_rows = (self.off_score + self.score_cap + 63) // 64 * 64          # rows start 256-byte aligned
        # prefold: instead of one row per placement, the rank's PARTIAL of the loop over its own run — min_dists float32[n_scene],
        # then labels int8[n_scene] — 5 bytes per scene point whatever the number of placements (rs_hip_label_partial_device)
        self.prefold = bool(prefold)
        self.words = self.off_rows + ((self.n_scene + (self.n_scene + 3) // 4 + 63) // 64 * 64 if self.prefold else self.plc_cap * self.n_scene)
        self.small_words = self.off_rows

    def slices(self, rank):
        return (shard_range(self.n_icp, rank, self.world), shard_range(self.n_score, rank, self.world),
                shard_range(self.n_plc, rank, self.world))


def shard_compute(capi, lay, rank, units, send, small_host, threads=None):
    """This rank's share of one step, results into `send` (a float32 device tensor of lay.words):
    units = dict(icp=(source, target, T0s, max_dist, max_angle, iters), score=(object, scene, poses, radius, K),
                 label=(scene, poses_sorted, clouds_sorted, radii_sorted)).
    The label kernel writes its rows into the send buffer itself (rs_hip_label_rows, rows_device = 1); poses / errors /
    scores (a few KB) pass through `small_host` (pinned float32[lay.small_words]).  Returns nothing: call
    shard_publish afterwards from the thread that owns torch's stream."""
    (i0, i1), (s0, s1), (p0, p1) = lay.slices(rank)
    src, tgt, T0s, max_dist, max_angle, iters = units["icp"]
    obj, scn, poses, radius, K = units["score"]
    lscene, lposes, lclouds, lradii = units["label"]
    sh = small_host.numpy() if hasattr(small_host, "numpy") else small_host
    sh[:] = 0.0

    def icp():
        if i1 > i0:
            blk = sh[lay.off_icp: lay.off_icp + 18 * (i1 - i0)].reshape(-1, 18)
            if isinstance(src, (list, tuple)):
                # one source cloud per problem (the placements of a scene, each refined against the scan: lib/rs/rs_database.h:220-230),
                # all of this rank's problems in ONE call: their sequential chains run side by side (rs_hip_icp_align_multi)
                errs, Ts, its = capi.icp_align_multi(list(src[i0:i1]), tgt, np.asarray(T0s)[i0:i1], max_dist=max_dist, max_angle=max_angle,
                                                     max_iter=iters, fixed_iters=True)
                blk[:, :16] = Ts.reshape(-1, 16); blk[:, 16] = errs; blk[:, 17] = its
            else:
                errs, Ts, its = capi.icp_align_batch(src, tgt, T0s[i0:i1], max_dist=max_dist, max_angle=max_angle, max_iter=iters, fixed_iters=True)
                blk[:, :16] = Ts.reshape(-1, 16); blk[:, 16] = errs; blk[:, 17] = its

    def score():
        if s1 > s0:
            sh[lay.off_score: lay.off_score + (s1 - s0)] = capi.alignment_scores(obj, scn, poses[s0:s1], radius, K)

    def label():
        if lay.prefold:
            # (every rank sends a partial, an empty run's being the loop's initial state)
            base = send.data_ptr() + 4 * lay.off_rows
            capi.label_partial_device(lscene, lposes[p0:p1], lclouds[p0:p1], lradii[p0:p1], p0, base, base + 4 * lay.n_scene)
            capi.synchronize()
        elif p1 > p0:
            # rows in the scene cloud's query order: what the kernel writes (coalesced); every rank builds the same cloud from
            # the same scene, so the gathered rows share that order and shard_fold hands the cloud to the fold
            capi.label_rows(lscene, lposes[p0:p1], lclouds[p0:p1], lradii[p0:p1], out_device_ptr=send.data_ptr() + 4 * lay.off_rows, query_order=True)
            capi.synchronize()            # the rows are complete before the collective (another stream) reads them

    if threads is not None and hasattr(threads, "run3"):      # bench.py's RoleRunner: each consumer on its own thread / stream / CU partition
        threads.run3(icp, score, label)
    elif threads is not None:          # three executors, one per consumer, or one pool
        subs = [ex.submit(fn) for ex, fn in zip(threads, (icp, score, label))] if isinstance(threads, (list, tuple)) else [threads.submit(fn) for fn in (icp, score, label)]
        for f in subs:
            f.result()
    else:
        icp(); score(); label()


def shard_publish(lay, send, small_host):
    """Copies the small results next to the rows (torch's current stream)."""
    import torch
    t = small_host if isinstance(small_host, torch.Tensor) else torch.from_numpy(small_host)
    send[: lay.small_words].copy_(t, non_blocking=True)


def shard_fold(capi, lay, recv, out=None, scene=None):
    """recv = the all-gathered send buffers (float32 device tensor, lay.world x lay.words), complete (the caller has
    waited for the collective).  Returns (errs, Ts, iters, scores, labels, min_dists) for ALL units, in unit order:
    the small blocks are read back, the rows are folded on the device in the sorted placement order."""
    W = lay.words
    small = recv.view(lay.world, W)[:, : lay.small_words].cpu().numpy()
    Ts, errs, its, scores, offsets = [], [], [], [], []
    for r in range(lay.world):
        (i0, i1), (s0, s1), (p0, p1) = lay.slices(r)
        blk = small[r, lay.off_icp: lay.off_icp + 18 * (i1 - i0)].reshape(-1, 18)
        Ts.append(blk[:, :16]); errs.append(blk[:, 16]); its.append(blk[:, 17].astype(np.int32))
        scores.append(small[r, lay.off_score: lay.off_score + (s1 - s0)])
        offsets += [r * W + lay.off_rows + k * lay.n_scene for k in range(p1 - p0)]
    if lay.prefold:
        mo = [r * W + lay.off_rows for r in range(lay.world)]
        lo = [4 * (r * W + lay.off_rows + lay.n_scene) for r in range(lay.world)]
        labels, mind = capi.fold_label_partials_device(recv.data_ptr(), mo, lo, lay.n_scene, *(out if out is not None else (None, None)), query_order_of=scene)
        return (np.concatenate(errs), np.concatenate(Ts).copy(), np.concatenate(its), np.concatenate(scores), labels, mind)
    # from (0, 1e9): rs_pointcloud_filters.cpp:799-802,820; out = (labels int8[n_scene], min_dists float32[n_scene]) to receive the result
    # scene: the Cloud in whose query order shard_compute wrote the rows
    labels, mind = capi.fold_label_rows_device(recv.data_ptr(), offsets, lay.n_scene, *(out if out is not None else (None, None)), fresh=True,
                                               query_order_of=scene)
    return (np.concatenate(errs), np.concatenate(Ts).copy(), np.concatenate(its), np.concatenate(scores), labels, mind)
