"""ctypes binding of include/rescan_hip.h (librescan_hip.so) and a thin Python mirror of the
reference's operator names for the hot path (icp_align, alignment scores, arrangement_to_labels).

There is deliberately no CPU fallback here: if the HIP extension is missing, or no HIP device
is usable, every call raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librescan_hip.so")

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")

# name -> (restype, argtypes); kept in one table so the symbol test can walk it.
SIGNATURES = {
    "rs_hip_init": (C.c_int, [C.c_int]),
    "rs_hip_last_error": (C.c_char_p, []),
    "rs_hip_set_stream": (C.c_int, [C.c_void_p]),
    "rs_hip_synchronize": (C.c_int, []),
    "rs_hip_stream_cu_mask": (C.c_int, [C.c_void_p, C.c_int32]),
    "rs_hip_get_stream": (C.c_void_p, []),
    "rs_hip_version": (C.c_char_p, []),
    "rs_hip_profile_enable": (C.c_int, [C.c_int]),
    "rs_hip_profile_reset": (C.c_int, []),
    "rs_hip_profile_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "rs_hip_cloud_create": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float]),
    "rs_hip_cloud_destroy": (None, [C.c_void_p]),
    "rs_hip_profile_marker": (C.c_int, []),
    "rs_hip_cloud_size": (C.c_int32, [C.c_void_p]),
    "rs_hip_cloud_bytes": (C.c_int64, [C.c_void_p]),
    "rs_hip_cloud_build_seconds": (C.c_int64, [C.POINTER(C.c_double), C.c_int32]),
    "rs_hip_radius_search": (C.c_int, [C.c_void_p, f32p, C.c_int64, C.c_float, C.c_int32, f32p, i32p, u64p,
                                       C.POINTER(C.c_uint64)]),
    "rs_hip_icp_align": (C.c_int, [C.c_void_p, C.c_void_p, f32p, f32p, C.c_float, C.c_float, C.c_int32, C.c_int32,
                                   C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "rs_hip_icp_align_traced": (C.c_int, [C.c_void_p, C.c_void_p, f32p, f32p, C.c_float, C.c_float, C.c_int32, C.c_int32,
                                          C.POINTER(C.c_float), C.POINTER(C.c_int32), f32p]),
    "rs_hip_icp_reference_order_below": (C.c_int32, [C.c_int32]),
    "rs_hip_icp_replay_below": (C.c_int32, [C.c_int32]),
    "rs_hip_icp_lane_chains_below": (C.c_int32, [C.c_int32]),
    "rs_hip_icp_lane_chains_sequential": (C.c_int64, []),
    "rs_hip_icp_stop_guard_redone": (C.c_int64, []),
    "rs_hip_icp_stop_guard": (C.c_float, [C.c_float]),
    "rs_hip_icp_early_plain": (C.c_int32, [C.c_int32]),
    "rs_hip_icp_replay_redone": (C.c_int32, []),
    "rs_hip_icp_faith_redone": (C.c_int32, []),
    "rs_hip_icp_faith_guess": (C.c_int32, [C.c_int32]),
    "rs_hip_icp_exact_centroids": (C.c_int32, [C.c_int32]),
    "rs_hip_score_scene_space_from": (C.c_int64, [C.c_int64]),
    "rs_hip_icp_chains_retry_after": (C.c_int32, [C.c_int32]),
    "rs_hip_icp_chains_gave_up": (C.c_int32, []),
    "rs_hip_icp_align_batch": (C.c_int, [C.c_void_p, C.c_void_p, f32p, C.c_int32, f32p, C.c_float, C.c_float,
                                         C.c_int32, C.c_int32, f32p, i32p]),
    "rs_hip_icp_align_multi": (C.c_int, [C.c_void_p, C.c_void_p, f32p, C.c_int32, f32p, C.c_float, C.c_float,
                                         C.c_int32, C.c_int32, f32p, i32p]),
    "rs_hip_icp_find_corrs": (C.c_int, [C.c_void_p, C.c_void_p, f32p, f32p, C.c_float, C.c_float,
                                        f32p, f32p, f32p, f32p, f32p, C.POINTER(C.c_int32)]),
    "rs_hip_alignment_scores": (C.c_int, [C.c_void_p, C.c_void_p, f32p, C.c_int32, C.c_float, C.c_int32, f32p]),
    "rs_hip_assign_labels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, i8p, f32p]),
    "rs_hip_label_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int]),
    "rs_hip_combine_label_rows": (None, [f32p, C.c_int32, C.c_int64, C.c_int32, i8p, f32p]),
    "rs_hip_label_partial_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "rs_hip_fold_label_partials_device": (C.c_int, [C.c_void_p, np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS"), np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS"),
                                                    C.c_int32, C.c_int64, i8p, f32p, C.c_void_p]),
    "rs_hip_fold_label_rows_device": (C.c_int, [C.c_void_p, np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS"), C.c_int32,
                                                C.c_int64, C.c_int32, i8p, f32p, C.c_int32, C.c_void_p]),
    "rs_hip_arrangement_to_labels": (C.c_int, [C.c_void_p, f32p, C.c_void_p, i32p, i32p, C.c_int32, C.c_float,
                                               C.c_int, i8p, f32p, i32p]),
    "rs_hip_arrangement_to_ids": (C.c_int, [C.c_void_p, f32p, C.c_void_p, i32p, i32p, i32p, C.c_int32, C.c_float, C.c_int, C.c_int32,
                                            i32p, i32p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rs_hip_gather_attributes": (C.c_int, [i32p, C.c_int32, C.c_int32, C.c_void_p, i32p, C.c_void_p, C.c_int32]),
    "rs_hip_compute_neighborhood": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, i32p, i32p, f32p,
                                              C.c_int64, C.POINTER(C.c_int64)]),
    "rs_hip_coverage_create": (C.c_void_p, [f32p, f32p, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_float]),
    "rs_hip_coverage_destroy": (None, [C.c_void_p]),
    "rs_hip_coverage_info": (C.c_int, [C.c_void_p, i32p, f32p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rs_hip_coverage_scene_grid": (C.c_int, [C.c_void_p, np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")]),
    "rs_hip_coverage_scores": (C.c_int, [C.c_void_p, C.c_void_p, f32p, i32p, i32p, C.c_int32, f32p, C.c_void_p]),
    "rs_hip_cloud_create_level": (C.c_void_p, [C.c_void_p, C.c_float, C.c_int32, C.c_float, i32p, C.POINTER(C.c_int32)]),
    "rs_hip_level_samples": (C.c_int, [C.c_void_p, C.c_float, C.c_int32, i32p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rs_hip_mat4_inverse": (None, [f32p, f32p]),
    "rs_hip_sincosf_model": (None, [f32p, C.c_int64, f32p, f32p]),
    "rs_hip_mat4_mul": (None, [f32p, f32p, f32p]),
    "rs_hip_icp_estimate_pt2pl": (C.c_int, [f32p, f32p, f32p, f32p, C.c_int32, f32p, C.POINTER(C.c_float)]),
}


class RescanHipError(RuntimeError):
    pass


class Placement(C.Structure):
    _fields_ = [("pose", C.c_float * 16), ("object", C.c_void_p), ("radius", C.c_float)]


_lib = None


def load():
    """dlopen librescan_hip.so and attach signatures.  Raises if the extension was not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("RS_HIP_LIB", LIB_PATH)       # kernel A/B experiments (tools/variant.sh)
    if not os.path.exists(path):
        raise RescanHipError(
            f"{path} is missing: build it with `python -m rescan_amd.build` "
            "(there is no CPU fallback for the hot path)")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if "RS_HIP_LIB" in os.environ:        # an A/B build of an older tree may predate an entry point
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise RescanHipError(f"librescan_hip error {rc}: {load().rs_hip_last_error().decode()}")


def init(device=0):
    _check(load().rs_hip_init(int(device)))


def set_stream(stream_handle):
    _check(load().rs_hip_set_stream(C.c_void_p(stream_handle) if stream_handle else None))


def synchronize():
    _check(load().rs_hip_synchronize())


def stream_cu_mask(bits):
    """Restrict the calling thread's stream to the CUs whose entries in `bits` (sequence of 0/1, CU 0 first) are set; None: back
    to an unrestricted stream."""
    if bits is None:
        _check(load().rs_hip_stream_cu_mask(None, 0))
        return
    b = np.asarray(bits, np.uint8)
    words = np.zeros((len(b) + 31) // 32, np.uint32)
    for k, v in enumerate(b):
        if v:
            words[k // 32] |= np.uint32(1 << (k % 32))
    _check(load().rs_hip_stream_cu_mask(words.ctypes.data, len(words)))


def get_stream():
    """The HIP stream (an integer handle) the calling thread's launches go to."""
    return load().rs_hip_get_stream()


def profile_enable(on=True):
    load().rs_hip_profile_enable(1 if on else 0)


def profile_reset():
    load().rs_hip_profile_reset()


def profile_read(name):
    n = C.c_int64(); ms = C.c_double()
    load().rs_hip_profile_read(name.encode(), C.byref(n), C.byref(ms))
    return n.value, ms.value


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


IDENTITY = np.eye(4, dtype=np.float32).ravel()


class Cloud:
    """A device-resident cloud level + its grid index (rs_hip_cloud_t)."""

    def __init__(self, pos, nor=None, cell_size=-1.0):
        """cell_size > 0: explicit grid cell; < 0 (default): from the sampling density; 0: brute-tile layout."""
        lib = load()
        pos = _f32(pos).reshape(-1, 3)
        self.n = len(pos)
        self._pos = pos
        self._nor = None if nor is None else _f32(nor).reshape(-1, 3)
        self.handle = lib.rs_hip_cloud_create(
            pos.ctypes.data_as(C.c_void_p),
            None if self._nor is None else self._nor.ctypes.data_as(C.c_void_p),
            self.n, float(cell_size))
        if not self.handle:
            raise RescanHipError("rs_hip_cloud_create failed: " + lib.rs_hip_last_error().decode())

    @classmethod
    def level_of(cls, base, radius, max_n_neigh, cell_size=-1.0):
        """The level of `base` as a cloud of its own, built on the device (rs_hip_cloud_create_level).
        Returns (cloud, sample indices)."""
        lib = load()
        idx = np.zeros(max(base.n, 1), np.int32); m = C.c_int32()
        h = lib.rs_hip_cloud_create_level(base.handle, float(radius), int(max_n_neigh), float(cell_size), idx, C.byref(m))
        if not h:
            raise RescanHipError("rs_hip_cloud_create_level failed: " + lib.rs_hip_last_error().decode())
        self = cls.__new__(cls)
        self.handle = h; self.n = m.value
        idx = idx[:m.value].copy()
        self._pos = base._pos[idx]
        self._nor = None if base._nor is None else base._nor[idx]
        return self, idx

    @property
    def nbytes(self):
        return load().rs_hip_cloud_bytes(self.handle)

    def close(self):
        if getattr(self, "handle", None):
            load().rs_hip_cloud_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def radius_search(target, query, radius, k):
    """msh_hash_grid_radius_search: rows of the k nearest within radius, ascending."""
    query = _f32(query).reshape(-1, 3)
    nq = len(query)
    d = np.zeros((nq, k), np.float32); i = np.zeros((nq, k), np.int32); nn = np.zeros(nq, np.uint64)
    tot = C.c_uint64()
    _check(load().rs_hip_radius_search(target.handle, query, nq, float(radius), int(k), d, i, nn, C.byref(tot)))
    return d, i, nn.astype(np.int64), tot.value


def profile_marker():
    """A named no-op kernel on the calling thread's stream (marks a step's start in a rocprofv3 kernel trace)."""
    _check(load().rs_hip_profile_marker())


def cloud_build_seconds(reset=False):
    """(diagnostics) seconds spent building clouds since the last reset: (host copy, upload + bounds, cell index, Hilbert order + tiles), clouds counted."""
    out = (C.c_double * 4)()
    n = load().rs_hip_cloud_build_seconds(out, 1 if reset else 0)
    return [float(x) for x in out], int(n)


def icp_reference_order_below(n_points=-1):
    """Sources of at most n_points points run the estimator in the reference's accumulation order (bit-identical
    results); -1 only reads.  Returns the previous threshold."""
    return int(load().rs_hip_icp_reference_order_below(int(n_points)))


def icp_replay_below(n_points=-1):
    """Threshold up to which sources above the reference-order threshold get the reference's sums computed in parallel
    (same bits); -1 only reads.  Returns the previous threshold."""
    return int(load().rs_hip_icp_replay_below(int(n_points)))


def icp_lane_chains_below(n_points=-1):
    """Sources above the two thresholds before and of at most n_points points: the reference's centroid chains by one wave per chain
    + fp64 moments (any number of problems side by side); -1 only reads.  Returns the previous threshold."""
    return int(load().rs_hip_icp_lane_chains_below(int(n_points)))


def icp_lane_chains_sequential():
    """(diagnostics) addends the lane chains have added one by one in fp32 since init."""
    return int(load().rs_hip_icp_lane_chains_sequential())


def icp_stop_guard_redone():
    """(diagnostics) problems run again in the reference's order because a stop test was decided inside the guard, since init."""
    return int(load().rs_hip_icp_stop_guard_redone())


def icp_early_plain(on=-1):
    """Plain (chain-free) early iterations of the lane / grid chain estimators (default on); -1 only reads.  Returns the previous setting."""
    return int(load().rs_hip_icp_early_plain(int(on)))


def icp_stop_guard(guard=-1.0):
    """Width of the stop test's guard (0: off; default 1.5e-6); < 0 only reads.  Returns the previous width."""
    return float(load().rs_hip_icp_stop_guard(float(guard)))


def icp_exact_centroids(on=-1):
    """Sources above both thresholds: centre the fp64 step on the reference's own fp32 centroid chains (default on); -1 only
    reads.  Returns the previous setting."""
    return int(load().rs_hip_icp_exact_centroids(int(on)))


def icp_chains_retry_after(calls=-1):
    """After a source's centroid chains gave up, its next `calls` ICP calls go straight to the replay (default 15; 0: every call tries
    the chains first); -1 only reads.  Returns the previous value."""
    return int(load().rs_hip_icp_chains_retry_after(int(calls)))


def score_scene_space_from(n_queries=-1):
    """Score batches of at least n_queries (poses x object points) take the scene-space route (queries sorted by scene block);
    -1 only reads.  Returns the previous threshold."""
    return int(load().rs_hip_score_scene_space_from(int(n_queries)))


def icp_chains_gave_up():
    """Calls the grid chains gave up and the replay's pass 2 redid (include/rescan_hip.h)."""
    return int(load().rs_hip_icp_chains_gave_up())


def icp_replay_redone():
    return int(load().rs_hip_icp_replay_redone())


def icp_faith_guess(permille=-1):
    """Test switch of the sequential estimator's guessed cut (1000: as made, 0: three passes, else scaled); returns the previous value."""
    return int(load().rs_hip_icp_faith_guess(int(permille)))


def icp_faith_redone():
    """Iterations of the sequential estimator whose one-pass statistics + centroids had to be summed again (cumulative)."""
    return int(load().rs_hip_icp_faith_redone())


def icp_align(source, target, T1, T2=IDENTITY, max_dist=0.1, max_angle=np.deg2rad(60.0), max_iter=100,
              fixed_iters=False):
    """icp_align (lib/rs/icp.h:416-500).  Returns (err, T1_new, n_iters)."""
    T = _f32(T1).ravel().copy()
    err = C.c_float(); it = C.c_int32()
    _check(load().rs_hip_icp_align(source.handle, target.handle, T, _f32(T2).ravel(), float(max_dist),
                                   float(np.float32(max_angle)), int(max_iter), int(bool(fixed_iters)),
                                   C.byref(err), C.byref(it)))
    return err.value, T, it.value


def icp_align_traced(source, target, T1, T2=IDENTITY, max_dist=0.1, max_angle=np.deg2rad(60.0), max_iter=100, fixed_iters=False):
    """icp_align, also returning the error after every iteration (what the reference prints with verbose = true).
    Returns (err, T1_new, n_iters, errs[n_iters])."""
    T = _f32(T1).ravel().copy()
    err = C.c_float(); it = C.c_int32()
    errs = np.zeros(max(1, int(max_iter)), np.float32)
    _check(load().rs_hip_icp_align_traced(source.handle, target.handle, T, _f32(T2).ravel(), float(max_dist),
                                          float(np.float32(max_angle)), int(max_iter), int(bool(fixed_iters)),
                                          C.byref(err), C.byref(it), errs))
    return err.value, T, it.value, errs[: it.value].copy()


def icp_align_batch(source, target, T1s, T2=IDENTITY, max_dist=0.1, max_angle=np.deg2rad(60.0), max_iter=100,
                    fixed_iters=False):
    T = _f32(T1s).reshape(-1, 16).copy()
    n = len(T)
    errs = np.zeros(n, np.float32); its = np.zeros(n, np.int32)
    _check(load().rs_hip_icp_align_batch(source.handle, target.handle, T, n, _f32(T2).ravel(), float(max_dist),
                                         float(np.float32(max_angle)), int(max_iter), int(bool(fixed_iters)),
                                         errs, its))
    return errs, T, its


def icp_align_multi(sources, target, T1s, T2=IDENTITY, max_dist=0.1, max_angle=np.deg2rad(60.0), max_iter=100, fixed_iters=False):
    """rs_hip_icp_align_multi: problem p aligns sources[p] (a Cloud each) to `target` from T1s[p] — one call."""
    T = _f32(T1s).reshape(-1, 16).copy()
    n = len(T)
    assert n == len(sources)
    handles = (C.c_void_p * max(1, n))(*[s_.handle for s_ in sources])
    errs = np.zeros(n, np.float32); its = np.zeros(n, np.int32)
    _check(load().rs_hip_icp_align_multi(C.addressof(handles), target.handle, T, n, _f32(T2).ravel(), float(max_dist),
                                         float(np.float32(max_angle)), int(max_iter), int(bool(fixed_iters)), errs, its))
    return errs, T, its


def icp_find_corrs(source, target, T1, T2=IDENTITY, max_dist=0.1, max_angle=np.deg2rad(60.0)):
    n1 = source.n
    out = [np.zeros((max(n1, 1), 3), np.float32) for _ in range(4)]
    w = np.zeros(max(n1, 1), np.float32)
    nc = C.c_int32()
    _check(load().rs_hip_icp_find_corrs(source.handle, target.handle, _f32(T1).ravel(), _f32(T2).ravel(),
                                        float(max_dist), float(np.float32(max_angle)), *out, w, C.byref(nc)))
    return [o[:nc.value] for o in out] + [w[:nc.value]]


def icp_estimate_pt2pl(p1, p2, n2, w, T1):
    T = _f32(T1).ravel().copy(); err = C.c_float()
    _check(load().rs_hip_icp_estimate_pt2pl(_f32(p1), _f32(p2), _f32(n2), _f32(w), len(w), T, C.byref(err)))
    return err.value, T


def alignment_scores(obj, scene, poses, radius=0.1, max_n_neigh=64):
    """mgs_compute_object_alignment_score for a batch of poses."""
    poses = _f32(poses).reshape(-1, 16)
    out = np.zeros(len(poses), np.float32)
    _check(load().rs_hip_alignment_scores(obj.handle, scene.handle, poses, len(poses), float(radius),
                                          int(max_n_neigh), out))
    return out


def _placements(poses, objects, radii):
    n = len(objects)
    arr = (Placement * max(1, n))()
    poses = _f32(poses).reshape(-1, 16)
    for i in range(n):
        arr[i].pose[:] = [float(x) for x in poses[i]]
        arr[i].object = objects[i].handle
        arr[i].radius = float(radii[i])
    return arr


def assign_labels(scene, poses, objects, radii, labels, min_dists, label_base=0):
    arr = _placements(poses, objects, radii)
    _check(load().rs_hip_assign_labels(scene.handle, C.addressof(arr), len(objects), int(label_base), labels, min_dists))
    return labels, min_dists


def label_rows(scene, poses, objects, radii, out_device_ptr=None, query_order=False):
    """Per-placement unary rows.  With out_device_ptr the rows stay on the GPU (for an all-gather): indexed by scene point
    in input order, or — query_order — by the scene cloud's query slot (what the kernel writes, no re-ordering pass)."""
    arr = _placements(poses, objects, radii)
    n = len(objects)
    if out_device_ptr is not None:
        _check(load().rs_hip_label_rows(scene.handle, C.addressof(arr), n, C.c_void_p(out_device_ptr), 2 if query_order else 1))
        return None
    rows = np.zeros((n, scene.n), np.float32)
    _check(load().rs_hip_label_rows(scene.handle, C.addressof(arr), n, rows.ctypes.data_as(C.c_void_p), 0))
    return rows


def label_partial_device(scene, poses, objects, radii, label_base, min_dists_ptr, labels_ptr):
    """The (min_dist, label) partial of a contiguous run of the sorted arrangement, written to device memory (query order)."""
    arr = _placements(poses, objects, radii)
    _check(load().rs_hip_label_partial_device(scene.handle, C.addressof(arr), len(objects), int(label_base), C.c_void_p(min_dists_ptr), C.c_void_p(labels_ptr)))


def fold_label_partials_device(base_ptr, min_offsets, label_offsets, scene_n, labels=None, min_dists=None, query_order_of=None):
    """Ordered fold of gathered per-rank partials (min_dists at base + 4*min_offsets[r], int8 labels at base + label_offsets[r])."""
    mo = np.ascontiguousarray(min_offsets, np.int64); lo = np.ascontiguousarray(label_offsets, np.int64)
    if labels is None or min_dists is None:
        labels = np.empty(int(scene_n), np.int8); min_dists = np.empty(int(scene_n), np.float32)
    _check(load().rs_hip_fold_label_partials_device(C.c_void_p(base_ptr), mo, lo, len(mo), int(scene_n), labels, min_dists,
                                                    query_order_of.handle if query_order_of is not None else None))
    return labels, min_dists


def combine_label_rows(rows, labels, min_dists, label_base=0):
    rows = _f32(rows)
    load().rs_hip_combine_label_rows(rows, rows.shape[0], rows.shape[1], int(label_base), labels, min_dists)
    return labels, min_dists


def fold_label_rows_device(rows_device_ptr, row_offsets, scene_n, labels=None, min_dists=None, label_base=0, fresh=None, query_order_of=None):
    """Ordered arg-min over rows that sit in device memory (row k at rows_device_ptr + 4*row_offsets[k]).
    fresh (default: when no labels / min_dists are given): the fold starts from the loop's initial state (label 0, 1e9)
    on the device; labels / min_dists, if given, only receive the result (e.g. pinned buffers that are reused).
    query_order_of: the scene Cloud whose query order the rows are in (label_rows(..., query_order=True)); the result is
    returned in input order either way."""
    off = np.ascontiguousarray(row_offsets, np.int64)
    if fresh is None:
        fresh = labels is None or min_dists is None
    if labels is None or min_dists is None:
        labels = np.empty(int(scene_n), np.int8); min_dists = np.empty(int(scene_n), np.float32)
    _check(load().rs_hip_fold_label_rows_device(C.c_void_p(rows_device_ptr), off, len(off), int(scene_n), int(label_base),
                                                labels, min_dists, 1 if fresh else 0,
                                                query_order_of.handle if query_order_of is not None else None))
    return labels, min_dists


def arrangement_to_labels(scene, poses, objects, is_static, class_idx, radius=0.05, prioritize_static=False):
    """rspf_arrangement_to_labels ordering + both passes.  Returns dict(labels, min_dists, order)."""
    n = len(objects)
    handles = (C.c_void_p * max(1, n))(*[o.handle for o in objects])
    labels = np.zeros(scene.n, np.int8); mind = np.zeros(scene.n, np.float32); order = np.zeros(max(1, n), np.int32)
    _check(load().rs_hip_arrangement_to_labels(
        scene.handle, _f32(poses).reshape(-1, 16), C.addressof(handles),
        np.ascontiguousarray(is_static, np.int32), np.ascontiguousarray(class_idx, np.int32), n,
        float(radius), int(bool(prioritize_static)), labels, mind, order))
    return dict(labels=labels, min_dists=mind, order=order[:n])


def arrangement_to_ids(scene, poses, objects, is_static, class_idx, uidx, radius=0.05, prioritize_static=False, unlabelled_class_idx=0):
    """rspf_arrangement_to_labels including its tail (:851-869).  Returns dict(class_ids, instance_ids, labels, min_dists, order)."""
    n = len(objects)
    handles = (C.c_void_p * max(1, n))(*[o.handle for o in objects])
    cls = np.zeros(scene.n, np.int32); inst = np.zeros(scene.n, np.int32)
    labels = np.zeros(scene.n, np.int8); mind = np.zeros(scene.n, np.float32); order = np.zeros(max(1, n), np.int32)
    _check(load().rs_hip_arrangement_to_ids(
        scene.handle, _f32(poses).reshape(-1, 16), C.addressof(handles), np.ascontiguousarray(is_static, np.int32),
        np.ascontiguousarray(class_idx, np.int32), np.ascontiguousarray(uidx, np.int32), n, float(radius), int(bool(prioritize_static)),
        int(unlabelled_class_idx), cls, inst, labels.ctypes.data_as(C.c_void_p), mind.ctypes.data_as(C.c_void_p), order.ctypes.data_as(C.c_void_p)))
    return dict(class_ids=cls, instance_ids=inst, labels=labels, min_dists=mind, order=order[:n])


def gather_attributes(sample_idx, arrays):
    """dst[a][i] = arrays[a][sample_idx[i]] on the device (the level builder's attribute gathers); arrays: list of 2-D or 1-D
    float32 / int32 arrays over the base level's points."""
    idx = np.ascontiguousarray(sample_idx, np.int32)
    srcs = [np.ascontiguousarray(a) for a in arrays]
    for a in srcs:
        assert a.dtype.itemsize == 4
    words = np.array([int(np.prod(a.shape[1:])) if a.ndim > 1 else 1 for a in srcs], np.int32)
    outs = [np.empty((len(idx),) + a.shape[1:], a.dtype) for a in srcs]
    sp = (C.c_void_p * len(srcs))(*[a.ctypes.data for a in srcs]); dp = (C.c_void_p * len(srcs))(*[o.ctypes.data for o in outs])
    _check(load().rs_hip_gather_attributes(idx, len(idx), len(srcs[0]) if srcs else 0, C.addressof(sp), words, C.addressof(dp), len(srcs)))
    return outs


def mat4_inverse(m):
    o = np.empty(16, np.float32); load().rs_hip_mat4_inverse(_f32(m).ravel(), o); return o


def level_samples(cloud, radius, max_n_neigh):
    """rs_pointcloud__compute_level_poisson (lib/rs/rs_pointcloud.h:984-1106): (sample indices, rounds)."""
    out = np.zeros(max(cloud.n, 1), np.int32)
    n = C.c_int32(); r = C.c_int32()
    _check(load().rs_hip_level_samples(cloud.handle, float(radius), int(max_n_neigh), out, C.byref(n), C.byref(r)))
    return out[:n.value].copy(), r.value


def sincosf_model(x):
    """The device's sinf/cosf evaluated on the host (tests: must equal the machine's libm)."""
    x = np.ascontiguousarray(x, np.float32).ravel()
    s = np.empty_like(x); c = np.empty_like(x)
    load().rs_hip_sincosf_model(x, len(x), s, c)
    return s, c


def mat4_mul(a, b):
    o = np.empty(16, np.float32); load().rs_hip_mat4_mul(_f32(a).ravel(), _f32(b).ravel(), o); return o


def compute_neighborhood(cloud, max_nn=8, radius_sq=0.05 * 0.05, dist_exp=15.0, angle_exp=16.0):
    """rspf_compute_neighborhood: unique weighted edges (idx1, idx2, weight) of the K-nearest self-search."""
    cap = max(1, cloud.n * max_nn)
    a = np.zeros(cap, np.int32); b = np.zeros(cap, np.int32); w = np.zeros(cap, np.float32)
    m = C.c_int64()
    _check(load().rs_hip_compute_neighborhood(cloud.handle, int(max_nn), float(np.float32(radius_sq)), float(dist_exp),
                                              float(angle_exp), a, b, w, cap, C.byref(m)))
    return a[:m.value].copy(), b[:m.value].copy(), w[:m.value].copy()


class Coverage:
    """Scene voxel grid + coverage scores (rsao__compute_scene_coverage_score)."""

    def __init__(self, bbox_min, bbox_max, scene_pos, quality=None, voxel_size=0.05, threshold=0.5):
        pos = np.ascontiguousarray(scene_pos, np.float32)
        q = None if quality is None else np.ascontiguousarray(quality, np.float32)
        self.handle = load().rs_hip_coverage_create(np.ascontiguousarray(bbox_min, np.float32), np.ascontiguousarray(bbox_max, np.float32),
                                                    float(voxel_size), pos.ctypes.data if len(pos) else None,
                                                    None if q is None else q.ctypes.data, len(pos), float(threshold))
        if not self.handle:
            raise RescanHipError(f"coverage_create failed: {load().rs_hip_last_error().decode()}")
        res = np.zeros(3, np.int32); org = np.zeros(3, np.float32); n = C.c_int64(); v = C.c_int64()
        _check(load().rs_hip_coverage_info(self.handle, res, org, C.byref(n), C.byref(v)))
        self.res, self.origin, self.n_cells, self.valid_cells = res, org, n.value, v.value

    def scene_grid(self):
        data = np.zeros(self.n_cells, np.uint8)
        _check(load().rs_hip_coverage_scene_grid(self.handle, data))
        return data

    def scores(self, arrangements):
        """arrangements: list of lists of (Cloud, pose16, is_static).  Returns (scores f32, agree i32)."""
        flat = [p for a in arrangements for p in a]
        first = np.zeros(len(arrangements) + 1, np.int32)
        first[1:] = np.cumsum([len(a) for a in arrangements])
        n = len(flat)
        objs = (C.c_void_p * max(1, n))(*[p[0].handle for p in flat])
        poses = np.ascontiguousarray(np.array([np.asarray(p[1], np.float32).ravel() for p in flat], np.float32).reshape(-1, 16)) if n else np.zeros((1, 16), np.float32)
        stat = np.array([int(p[2]) for p in flat] or [0], np.int32)
        sc = np.zeros(len(arrangements), np.float32); ag = np.zeros(len(arrangements), np.int32)
        _check(load().rs_hip_coverage_scores(self.handle, C.addressof(objs), poses, stat, first, len(arrangements), sc, ag.ctypes.data))
        return sc, ag

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                load().rs_hip_coverage_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
