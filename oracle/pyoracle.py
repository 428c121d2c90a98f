"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes bindings for the two CPU checkers:

* ``Oracle``  -> oracle/librs_oracle.so   (plain-C restatement, rs_oracle.c)
* ``Ref``     -> oracle/_ref/libref.so    (the real reference compiled in place; only
                 buildable where /root/reference exists, the .so travels to the GPU box)
* ``RefAO``   -> oracle/_ref/libref_ao.so (the reference's arrangement_optimization.cpp: scene-coverage term)

Both expose the same method names so tests can run one function against the other.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")


def build(ref=True):
    """make the oracle (and oracle/_ref when the reference tree is present)."""
    subprocess.check_call(["make", "-s", "-f", os.path.join(HERE, "Makefile"),
                           os.path.join(HERE, "librs_oracle.so")])
    if ref:
        subprocess.check_call(["make", "-s", "-f", os.path.join(HERE, "Makefile"), "ref"])


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class _Base:
    prefix = ""

    def _fn(self, name, restype, argtypes):
        f = getattr(self.lib, self.prefix + name)
        f.restype = restype
        f.argtypes = argtypes
        return f

    def _common(self):
        self._mat4_inverse = self._fn("mat4_inverse", None, [f32p, f32p])
        self._mat4_mul = self._fn("mat4_mul", None, [f32p, f32p, f32p])
        self._translate = self._fn("translate", None, [f32p, f32p, f32p])
        self._rotate = self._fn("rotate", None, [f32p, C.c_float, f32p, f32p])
        self._xform_points = self._fn("xform_points", None, [f32p, f32p, C.c_int64, C.c_int, f32p])
        self._normalize = self._fn("normalize", None, [f32p, C.c_int64, f32p])
        self._mean = self._fn("mean", C.c_float, [f32p, C.c_int])
        self._stddev = self._fn("stddev", C.c_float, [C.c_float, f32p, C.c_int])
        self._grid_create = self._fn("grid_create", C.c_void_p, [f32p, C.c_int32, C.c_float])
        self._grid_destroy = self._fn("grid_destroy", None, [C.c_void_p])
        self._grid_info = self._fn("grid_info", None, [C.c_void_p, i64p, C.POINTER(C.c_double), f32p,
                                                       C.POINTER(C.c_uint32)])
        self._radius_search = self._fn("radius_search", C.c_uint64,
                                       [C.c_void_p, f32p, C.c_int64, C.c_float, C.c_int64, C.c_int,
                                        f32p, i32p, i64p])
        self._icp_estimate = self._fn("icp_estimate_pt2pl", C.c_float, [f32p, f32p, f32p, f32p, C.c_int32, f32p])

    # -- helpers ---------------------------------------------------------------------
    def mat4_inverse(self, m):
        o = np.empty(16, np.float32); self._mat4_inverse(_f32(m).ravel(), o); return o

    def mat4_mul(self, a, b):
        o = np.empty(16, np.float32); self._mat4_mul(_f32(a).ravel(), _f32(b).ravel(), o); return o

    def translate(self, m, t):
        o = np.empty(16, np.float32); self._translate(_f32(m).ravel(), _f32(t), o); return o

    def rotate(self, m, angle, axis):
        o = np.empty(16, np.float32); self._rotate(_f32(m).ravel(), float(angle), _f32(axis), o); return o

    def xform_points(self, m, pts, is_point=1):
        pts = _f32(pts); o = np.empty_like(pts)
        self._xform_points(_f32(m).ravel(), pts, len(pts), int(is_point), o); return o

    def normalize(self, v):
        v = _f32(v); o = np.empty_like(v); self._normalize(v, len(v), o); return o

    def mean(self, v):
        v = _f32(v); return self._mean(v, len(v))

    def stddev(self, mean, v):
        v = _f32(v); return self._stddev(float(mean), v, len(v))

    # -- grid ------------------------------------------------------------------------
    def grid_create(self, pts, radius):
        pts = _f32(pts)
        return self._grid_create(pts, len(pts), float(radius))

    def grid_destroy(self, g):
        self._grid_destroy(g)

    def grid_info(self, g):
        dims = np.zeros(3, np.int64); cell = C.c_double(); minp = np.zeros(3, np.float32); mb = C.c_uint32()
        self._grid_info(g, dims, C.byref(cell), minp, C.byref(mb))
        return dims, cell.value, minp, mb.value

    def radius_search(self, g, query, radius, k, sort=1):
        query = _f32(query); nq = len(query)
        d = np.zeros((nq, k), np.float32); i = np.zeros((nq, k), np.int32); nn = np.zeros(nq, np.int64)
        total = self._radius_search(g, query, nq, float(radius), int(k), int(sort), d, i, nn)
        return d, i, nn, total

    def icp_estimate_pt2pl(self, p1, p2, n2, w, T1):
        T = _f32(T1).ravel().copy()
        err = self._icp_estimate(_f32(p1), _f32(p2), _f32(n2), _f32(w), len(w), T)
        return err, T


class Oracle(_Base):
    prefix = "orc_"

    def __init__(self):
        path = os.path.join(HERE, "librs_oracle.so")
        if not os.path.exists(path):
            build(ref=False)
        self.lib = C.CDLL(path)
        self._common()
        self._find_corrs = self._fn("icp_find_corrs", C.c_int32,
                                    [f32p, f32p, C.c_int32, f32p, f32p, C.c_int32, C.c_void_p, f32p, f32p,
                                     C.c_float, C.c_float, f32p, f32p, f32p, f32p, f32p])
        self._icp_align = self._fn("icp_align", C.c_float,
                                   [f32p, f32p, C.c_int32, f32p, f32p, C.c_int32, f32p, f32p, C.c_float, C.c_float,
                                    C.POINTER(C.c_int32)])
        self._scores = self._fn("alignment_scores", None,
                                [C.c_void_p, f32p, f32p, f32p, C.c_int32, f32p, C.c_int32, C.c_int32, f32p])
        self._icp_gate = self._fn("icp_gate", C.c_int, [C.c_float, C.c_float])
        self._score_gate = self._fn("score_gate", C.c_int, [C.c_float])
        self._label_gate = self._fn("label_gate", C.c_int, [C.c_float])
        self.lib.orc_arrangement_to_labels.restype = None
        self.lib.orc_assign_labels.restype = None

    def icp_find_corrs(self, pts1, nor1, pts2, nor2, T1, T2, max_dist, max_angle):
        pts1, nor1, pts2, nor2 = map(_f32, (pts1, nor1, pts2, nor2))
        n1 = len(pts1)
        g = self.grid_create(pts2, max_dist)
        out = [np.zeros((n1, 3), np.float32) for _ in range(4)]
        w = np.zeros(n1, np.float32)
        nc = self._find_corrs(pts1, nor1, n1, pts2, nor2, len(pts2), g, _f32(T1).ravel(), _f32(T2).ravel(),
                              float(max_dist), float(max_angle), *out, w)
        self.grid_destroy(g)
        return [o[:nc] for o in out] + [w[:nc]]

    def icp_align(self, pts1, nor1, pts2, nor2, T1, T2, max_dist, max_angle):
        pts1, nor1, pts2, nor2 = map(_f32, (pts1, nor1, pts2, nor2))
        T = _f32(T1).ravel().copy(); it = C.c_int32()
        err = self._icp_align(pts1, nor1, len(pts1), pts2, nor2, len(pts2), T, _f32(T2).ravel(),
                              float(max_dist), float(max_angle), C.byref(it))
        return err, T, it.value

    def alignment_scores(self, scene_pos, scene_nor, obj_pos, obj_nor, poses, k, scene_grid=None):
        scene_pos, scene_nor, obj_pos, obj_nor = map(_f32, (scene_pos, scene_nor, obj_pos, obj_nor))
        poses = _f32(poses).reshape(-1, 16)
        g = scene_grid if scene_grid is not None else self.grid_create(scene_pos, 0.05)
        out = np.zeros(len(poses), np.float32)
        self._scores(g, scene_nor, obj_pos, obj_nor, len(obj_pos), poses, len(poses), int(k), out)
        if scene_grid is None:
            self.grid_destroy(g)
        return out

    class _Obj(C.Structure):
        _fields_ = [("pos", C.c_void_p), ("nor", C.c_void_p), ("n", C.c_int32), ("grid", C.c_void_p),
                    ("class_idx", C.c_int32), ("is_static", C.c_int32)]

    class _Plc(C.Structure):
        _fields_ = [("pose", C.c_float * 16), ("object_idx", C.c_int32), ("uidx", C.c_int32)]

    def arrangement_to_labels(self, scene_pos, scene_nor, objects, placements, radius=0.05,
                              prioritize_static=0, unlabelled_class_idx=0):
        """objects: list of dict(pos, nor, class_idx, is_static); placements: list of dict(pose, object_idx, uidx)."""
        scene_pos, scene_nor = _f32(scene_pos), _f32(scene_nor)
        n = len(scene_pos)
        keep = []
        objs = (self._Obj * max(1, len(objects)))()
        grids = []
        for i, o in enumerate(objects):
            p, nn_ = _f32(o["pos"]), _f32(o["nor"]); keep += [p, nn_]
            g = self.grid_create(p, 0.05); grids.append(g)
            objs[i].pos = p.ctypes.data; objs[i].nor = nn_.ctypes.data; objs[i].n = len(p); objs[i].grid = g
            objs[i].class_idx = int(o["class_idx"]); objs[i].is_static = int(o["is_static"])
        plcs = (self._Plc * max(1, len(placements)))()
        for i, pl in enumerate(placements):
            plcs[i].pose[:] = [float(x) for x in _f32(pl["pose"]).ravel()]
            plcs[i].object_idx = int(pl["object_idx"]); plcs[i].uidx = int(pl["uidx"])
        labels = np.zeros(n, np.int8); mind = np.zeros(n, np.float32)
        order = np.zeros(max(1, len(placements)), np.int32)
        cls = np.zeros(n, np.int32); inst = np.zeros(n, np.int32)
        self.lib.orc_arrangement_to_labels.argtypes = [f32p, f32p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                                       C.c_float, C.c_int, C.c_int32, i8p, f32p, i32p, i32p, i32p]
        self.lib.orc_arrangement_to_labels(scene_pos, scene_nor, n, C.addressof(objs), C.addressof(plcs),
                                           len(placements), float(radius), int(prioritize_static),
                                           int(unlabelled_class_idx), labels, mind, order, cls, inst)
        for g in grids:
            self.grid_destroy(g)
        return dict(labels=labels, min_dists=mind, order=order[:len(placements)], class_ids=cls, instance_ids=inst)

    def compute_neighborhood(self, pos, nor, max_nn=8, radius_sq=0.0025, dist_exp=15.0, angle_exp=16.0):
        """rspf_compute_neighborhood on a level-1 style cloud (grid radius 0.05).  Returns (idx1, idx2, weight)."""
        pos, nor = _f32(pos), _f32(nor)
        n = len(pos)
        g = self.grid_create(pos, 0.05)
        a = np.zeros(n * max_nn, np.int32); b = np.zeros(n * max_nn, np.int32); w = np.zeros(n * max_nn, np.float32)
        f = self.lib.orc_compute_neighborhood
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, f32p, f32p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, i32p, i32p, f32p]
        m = f(g, pos, nor, n, int(max_nn), float(radius_sq), float(dist_exp), float(angle_exp), a, b, w)
        self.grid_destroy(g)
        return a[:m].copy(), b[:m].copy(), w[:m].copy()

    def edge_cost(self, d2, dot, radius_sq=0.0025, dist_exp=15.0, angle_exp=16.0):
        f = self.lib.orc_edge_cost
        f.restype = C.c_float
        f.argtypes = [C.c_float] * 5
        return f(float(d2), float(dot), float(radius_sq), float(dist_exp), float(angle_exp))

    # ---- scene-coverage term ----
    class VoxGrid(C.Structure):
        _fields_ = [("x_res", C.c_int32), ("y_res", C.c_int32), ("z_res", C.c_int32), ("n_cells", C.c_int32),
                    ("voxel_size", C.c_float), ("origin", C.c_float * 3)]

    def voxgrid(self, bbox_min, bbox_max, voxel_size=0.05):
        g = self.VoxGrid()
        f = self.lib.orc_voxgrid_init
        f.restype = None
        f.argtypes = [C.POINTER(self.VoxGrid), f32p, f32p, C.c_float]
        f(C.byref(g), _f32(bbox_min), _f32(bbox_max), float(voxel_size))
        return g

    def rasterize_scene(self, g, pos, quality=None, threshold=0.5):
        pos = _f32(pos)
        data = np.zeros(g.n_cells, np.uint8)
        f = self.lib.orc_rasterize_scene
        f.restype = None
        f.argtypes = [C.POINTER(self.VoxGrid), f32p, C.c_void_p, C.c_int64, C.c_float, np.ctypeslib.ndpointer(np.uint8)]
        q = None if quality is None else _f32(quality)
        f(C.byref(g), pos, None if q is None else q.ctypes.data, len(pos), float(threshold), data)
        return data

    def rasterize_arrangement(self, g, obj_pos, poses, is_static):
        n = len(obj_pos)
        keep = [_f32(p) for p in obj_pos]
        ptrs = (C.c_void_p * n)(*[p.ctypes.data for p in keep])
        ns = np.array([len(p) for p in keep], np.int64)
        data = np.zeros(g.n_cells, np.uint8)
        f = self.lib.orc_rasterize_arrangement
        f.restype = None
        f.argtypes = [C.POINTER(self.VoxGrid), C.c_void_p, i64p, f32p, i32p, C.c_int32, np.ctypeslib.ndpointer(np.uint8)]
        f(C.byref(g), C.addressof(ptrs), ns, _f32(np.asarray(poses).reshape(-1, 16)), np.asarray(is_static, np.int32), n, data)
        return data

    def level_poisson(self, pos, radius, max_n_neigh):
        """rs_pointcloud__compute_level_poisson: indices of the samples (increasing)."""
        pos = _f32(pos)
        out = np.zeros(max(len(pos), 1), np.int32)
        f = self.lib.orc_level_poisson
        f.restype = C.c_int32
        f.argtypes = [f32p, C.c_int32, C.c_float, C.c_int32, i32p]
        m = f(pos, len(pos), float(radius), int(max_n_neigh), out)
        return out[:m].copy()

    def coverage_score(self, scene_data, arr_data):
        f = self.lib.orc_coverage_score
        f.restype = C.c_float
        u8 = np.ctypeslib.ndpointer(np.uint8)
        f.argtypes = [u8, u8, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        a, v = C.c_int32(), C.c_int32()
        s = f(scene_data, arr_data, len(scene_data), C.byref(a), C.byref(v))
        return np.float32(s), a.value, v.value

    def icp_gate(self, dot, max_angle):
        return self._icp_gate(float(dot), float(max_angle))

    def score_gate(self, dot):
        return self._score_gate(float(dot))

    def label_gate(self, dot):
        return self._label_gate(float(dot))


class Ref(_Base):
    prefix = "ref_"

    @staticmethod
    def available(omp=False):
        return os.path.exists(os.path.join(HERE, "_ref", "libref_omp.so" if omp else "libref.so"))

    def __init__(self, omp=False):
        path = os.path.join(HERE, "_ref", "libref_omp.so" if omp else "libref.so")
        self.lib = C.CDLL(path)
        self._common()
        self._find_corrs = self._fn("icp_find_corrs", C.c_int32,
                                    [f32p, f32p, C.c_int32, f32p, f32p, C.c_int32, f32p, f32p,
                                     C.c_float, C.c_float, f32p, f32p, f32p, f32p, f32p])
        self._icp_align = self._fn("icp_align", C.c_float,
                                   [f32p, f32p, C.c_int32, f32p, f32p, C.c_int32, f32p, f32p, C.c_float, C.c_float])
        self._scene_create = self._fn("scene_create", C.c_void_p, [f32p, f32p, C.c_int64])
        self._scene_destroy = self._fn("scene_destroy", None, [C.c_void_p])
        self._scene_grid = self._fn("scene_grid", C.c_void_p, [C.c_void_p])
        self._scores = self._fn("alignment_scores", None,
                                [C.c_void_p, f32p, f32p, C.c_int32, f32p, C.c_int32, C.c_int32, f32p])
        self._find_corrs_grid = self._fn("icp_find_corrs_grid", C.c_int32,
                                         [C.c_void_p, f32p, f32p, C.c_int32, f32p, f32p, C.c_int32, f32p, f32p,
                                          C.c_float, C.c_float])
        self._label_gate = self._fn("label_gate", C.c_int, [f32p, f32p, f32p])
        self._label_gate_dot = self._fn("label_gate_dot", C.c_int, [C.c_float])
        self._num_threads = self._fn("num_threads", C.c_int, [])

    def num_threads(self):
        return self._num_threads()

    def icp_find_corrs(self, pts1, nor1, pts2, nor2, T1, T2, max_dist, max_angle):
        pts1, nor1, pts2, nor2 = map(_f32, (pts1, nor1, pts2, nor2))
        n1 = len(pts1)
        out = [np.zeros((n1, 3), np.float32) for _ in range(4)]
        w = np.zeros(n1, np.float32)
        nc = self._find_corrs(pts1, nor1, n1, pts2, nor2, len(pts2), _f32(T1).ravel(), _f32(T2).ravel(),
                              float(max_dist), float(max_angle), *out, w)
        return [o[:nc] for o in out] + [w[:nc]]

    def icp_find_corrs_grid(self, grid2, pts1, nor1, pts2, nor2, T1, T2, max_dist, max_angle):
        """One icp_find_corrs against a prebuilt grid of pts2; returns n_corrs only."""
        return self._find_corrs_grid(grid2, pts1, nor1, len(pts1), pts2, nor2, len(pts2), _f32(T1).ravel(),
                                     _f32(T2).ravel(), float(max_dist), float(max_angle))

    def icp_align(self, pts1, nor1, pts2, nor2, T1, T2, max_dist, max_angle):
        pts1, nor1, pts2, nor2 = map(_f32, (pts1, nor1, pts2, nor2))
        T = _f32(T1).ravel().copy()
        err = self._icp_align(pts1, nor1, len(pts1), pts2, nor2, len(pts2), T, _f32(T2).ravel(),
                              float(max_dist), float(max_angle))
        return err, T, None

    def alignment_scores(self, scene_pos, scene_nor, obj_pos, obj_nor, poses, k, scene=None):
        scene_pos, scene_nor, obj_pos, obj_nor = map(_f32, (scene_pos, scene_nor, obj_pos, obj_nor))
        poses = _f32(poses).reshape(-1, 16)
        s = scene if scene is not None else self._scene_create(scene_pos, scene_nor, len(scene_pos))
        out = np.zeros(len(poses), np.float32)
        self._scores(s, obj_pos, obj_nor, len(obj_pos), poses, len(poses), int(k), out)
        if scene is None:
            self._scene_destroy(s)
        return out

    def scene_create(self, scene_pos, scene_nor):
        return self._scene_create(scene_pos, scene_nor, len(scene_pos))

    def scene_destroy(self, s):
        self._scene_destroy(s)

    def label_gate(self, pose, n_scene, n_obj):
        return self._label_gate(_f32(pose).ravel(), _f32(n_scene), _f32(n_obj))

    def label_gate_dot(self, dot):
        return self._label_gate_dot(float(dot))

    def edge_cost(self, d2, dot, radius_sq=0.0025, dist_exp=15.0, angle_exp=16.0):
        f = self.lib.ref_edge_cost
        f.restype = C.c_float
        f.argtypes = [C.c_float] * 5
        return f(float(d2), float(dot), float(radius_sq), float(dist_exp), float(angle_exp))


def edge_digest(a, b, w):
    """Order-independent digest of an edge list: count, sum of pair keys, sum of weight bit patterns."""
    hi, lo = np.maximum(a, b).astype(np.int64), np.minimum(a, b).astype(np.int64)
    return np.array([len(a), int((hi * 1000003 + lo).sum() % (1 << 61)), int(np.sort(w.view(np.uint32)).astype(np.int64).sum())],
                    np.int64)


class RefAO:
    """The reference's scene-coverage term (apps/segment_transfer/arrangement_optimization.cpp) behind
    oracle/ref_ao_driver.cpp.  One class table per process (the reference caches class ids in statics)."""
    PATH = os.path.join(HERE, "_ref", "libref_ao.so")

    @staticmethod
    def available():
        return os.path.exists(RefAO.PATH)

    def __init__(self, class_table, scene_pos, bbox_min, bbox_max, quality=None, voxel_size=0.05, threshold=0.5):
        self.lib = C.CDLL(self.PATH)
        names = list(class_table)
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        ids = np.array([class_table[n] for n in names], np.int32)
        self._pos = _f32(scene_pos)
        self._q = _f32(np.ones(len(self._pos)) if quality is None else quality)
        self._objs = []
        f = self.lib.ref_ao_create
        f.restype = C.c_void_p
        f.argtypes = [C.c_void_p, i32p, C.c_int32, f32p, f32p, C.c_int64, f32p, f32p, C.c_float, C.c_float]
        self.h = f(C.addressof(arr), ids, len(names), self._pos, self._q, len(self._pos), _f32(bbox_min), _f32(bbox_max),
                   float(voxel_size), float(threshold))
        res = np.zeros(3, np.int32); org = np.zeros(3, np.float32); n = C.c_int64()
        g = self.lib.ref_ao_info
        g.restype = None
        g.argtypes = [C.c_void_p, i32p, f32p, C.POINTER(C.c_int64)]
        g(self.h, res, org, C.byref(n))
        self.res, self.origin, self.n_cells = res, org, n.value

    def add_object(self, pos, class_idx, uidx):
        p = _f32(pos)
        self._objs.append(p)
        f = self.lib.ref_ao_add_object
        f.restype = C.c_int32
        f.argtypes = [C.c_void_p, f32p, C.c_int64, C.c_int32, C.c_int32]
        return f(self.h, p, len(p), int(class_idx), int(uidx))

    def coverage(self, object_idx, poses):
        f = self.lib.ref_ao_coverage
        f.restype = C.c_float
        f.argtypes = [C.c_void_p, i32p, f32p, C.c_int32]
        return np.float32(f(self.h, np.asarray(object_idx, np.int32), _f32(np.asarray(poses).reshape(-1, 16)), len(object_idx)))

    def _grid(self, name):
        f = getattr(self.lib, name)
        f.restype = C.POINTER(C.c_uint8)
        f.argtypes = [C.c_void_p]
        return np.ctypeslib.as_array(f(self.h), shape=(self.n_cells,)).copy()

    def scene_grid(self):
        return self._grid("ref_ao_scene_grid")

    def arrangement_grid(self):
        return self._grid("ref_ao_arrangement_grid")


class RefFilters:
    """The reference's label transfer and neighbourhood graph (lib/rs/rs_pointcloud_filters.cpp:674-879, the file's own text
    compiled by oracle/Makefile -> oracle/_ref/libref_filters.so) behind oracle/ref_filters_driver.cpp.
    One class table per process (the reference caches class ids in function statics, rs_database.h:260-271)."""
    PATH = os.path.join(HERE, "_ref", "libref_filters.so")
    PATH_OMP = os.path.join(HERE, "_ref", "libref_filters_omp.so")     # the same text with the reference's optional OpenMP path
    _libs = {}
    _table = None

    @staticmethod
    def available(omp=False):
        return os.path.exists(RefFilters.PATH_OMP if omp else RefFilters.PATH)

    def __init__(self, class_table, omp=False):
        if omp not in RefFilters._libs:
            RefFilters._libs[omp] = C.CDLL(self.PATH_OMP if omp else self.PATH)
        if RefFilters._table is None:
            RefFilters._table = dict(class_table)
        assert dict(class_table) == RefFilters._table, "one class table per process"
        self.lib = RefFilters._libs[omp]
        names = list(class_table)
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        ids = np.array([class_table[n] for n in names], np.int32)
        f = self.lib.ref_filters_create
        f.restype = C.c_void_p
        f.argtypes = [C.c_void_p, i32p, C.c_int32]
        self.h = f(C.addressof(arr), ids, len(names))
        self._keep = []
        self.unlabelled = int(class_table["unlabelled"])

    def close(self):
        if self.h:
            f = self.lib.ref_filters_destroy
            f.restype = None
            f.argtypes = [C.c_void_p]
            f(self.h)
            self.h = None

    def add_object(self, pos, nor, class_idx, uidx):
        p, n_ = _f32(pos).copy(), _f32(nor).copy()
        self._keep += [p, n_]
        f = self.lib.ref_filters_add_object
        f.restype = C.c_int32
        f.argtypes = [C.c_void_p, f32p, f32p, C.c_int64, C.c_int32, C.c_int32]
        return f(self.h, p, n_, len(p), int(class_idx), int(uidx))

    def is_object_static(self, idx):
        f = self.lib.ref_filters_is_object_static
        f.restype = C.c_int32
        f.argtypes = [C.c_void_p, C.c_int32]
        return int(f(self.h, int(idx)))

    def _plc_arrays(self, placements):
        n = len(placements)
        oi = np.array([int(p["object_idx"]) for p in placements] + [0] * (n == 0), np.int32)
        ui = np.array([int(p["uidx"]) for p in placements] + [0] * (n == 0), np.int32)
        po = _f32(np.stack([_f32(p["pose"]).ravel() for p in placements]) if n else np.zeros((1, 16)))
        return oi, ui, po

    def whole(self, scene_pos, scene_nor, placements, radius=0.05, prioritize_static=0):
        """rspf_arrangement_to_labels itself -> (class_ids, instance_ids) of the scene's level 1."""
        sp, sn = _f32(scene_pos).copy(), _f32(scene_nor).copy()
        oi, ui, po = self._plc_arrays(placements)
        cls = np.zeros(max(len(sp), 1), np.int32); inst = np.zeros(max(len(sp), 1), np.int32)
        f = self.lib.ref_filters_arrangement_to_labels
        f.restype = None
        f.argtypes = [C.c_void_p, f32p, f32p, C.c_int64, i32p, i32p, f32p, C.c_int32, C.c_float, C.c_int32, i32p, i32p]
        f(self.h, sp, sn, len(sp), oi, ui, po, len(placements), float(radius), int(prioritize_static), cls, inst)
        return cls[:len(sp)], inst[:len(sp)]

    def assign(self, scene_pos, scene_nor, placements, start, end, radius, labels, min_dists):
        """rspf__assign_temporary_labels over placements[start:end] (visited in the order given), in place."""
        sp, sn = _f32(scene_pos).copy(), _f32(scene_nor).copy()
        oi, ui, po = self._plc_arrays(placements)
        f = self.lib.ref_filters_assign_labels
        f.restype = None
        f.argtypes = [C.c_void_p, f32p, f32p, C.c_int64, i32p, i32p, f32p, C.c_int32, C.c_int32, C.c_int32, C.c_float, i8p, f32p]
        f(self.h, sp, sn, len(sp), oi, ui, po, len(placements), int(start), int(end), float(radius), labels, min_dists)

    def arrangement_to_labels(self, scene_pos, scene_nor, objects, placements, radius=0.05, prioritize_static=0,
                              unlabelled_class_idx=0):
        """Same arguments and result dict as Oracle.arrangement_to_labels, every array from the reference's compiled text:
        class / instance ids from rspf_arrangement_to_labels as a whole; labels / min_dists from the reference's
        rspf__assign_temporary_labels driven the way :837-848 drive it.  The visiting order that second part needs is the
        stable order of the comparator's key (:724-736; glibc's qsort is a merge sort) and is PROVEN against the whole
        function: mapping the labels through it must give exactly the whole function's ids."""
        assert int(unlabelled_class_idx) == self.unlabelled
        base = len(self._keep)          # object uidx must be unique per database (rsdb_add_object, rs_database.h:648-658); the loops never read it
        idx = [self.add_object(o["pos"], o["nor"], o["class_idx"], 100000 + base + i) for i, o in enumerate(objects)]
        for o, k in zip(objects, idx):
            assert int(o["is_static"]) == self.is_object_static(k), "is_static must follow the class table (rs_database.h:257-288)"
        plc = [dict(pose=p["pose"], object_idx=idx[int(p["object_idx"])], uidx=p["uidx"]) for p in placements]
        cls, inst = self.whole(scene_pos, scene_nor, plc, radius, prioritize_static)
        stat = [self.is_object_static(p["object_idx"]) for p in plc]
        key = [(s << 10) | int(objects[int(p["object_idx"])]["class_idx"]) for s, p in zip(stat, placements)]
        order = sorted(range(len(plc)), key=lambda i: key[i])
        srt = [plc[i] for i in order]
        first_static = next((i for i, k in enumerate(order) if stat[k]), 0)                      # :830-835
        n = len(scene_pos)
        labels = np.zeros(max(n, 1), np.int8); mind = np.full(max(n, 1), 1e9, np.float32)       # :799-802, :820
        self.assign(scene_pos, scene_nor, srt, 0, first_static, radius, labels, mind)            # :837-839
        if prioritize_static:
            mind[:] = 1e9                                                                        # :841-844
        r2 = np.float32(radius) if prioritize_static else np.float32(1.5) * np.float32(radius)   # :845
        self.assign(scene_pos, scene_nor, srt, first_static, len(srt), r2, labels, mind)         # :846-848
        labels, mind = labels[:n], mind[:n]
        if len(srt):
            sc = np.array([objects[int(placements[i]["object_idx"])]["class_idx"] for i in order], np.int32)
            su = np.array([int(placements[i]["uidx"]) for i in order], np.int32)
            li = np.maximum(labels.astype(np.int64) - 1, 0)
            want_c = np.where(labels == 0, self.unlabelled, sc[li]); want_i = np.where(labels == 0, 1024, su[li])
        else:
            want_c = np.full(n, self.unlabelled); want_i = np.full(n, 1024)
        assert (want_c == cls).all() and (want_i == inst).all(), "driver order differs from rspf_arrangement_to_labels' own"
        return dict(labels=labels, min_dists=mind, order=np.array(order, np.int32), class_ids=cls, instance_ids=inst)

    def compute_neighborhood(self, pos, nor, max_nn=8, radius_sq=0.0025, dist_exp=15.0, angle_exp=16.0):
        """rspf_compute_neighborhood on a bare level-1 cloud -> (idx1, idx2, weight) in the reference's own output order."""
        p, n_ = _f32(pos).copy(), _f32(nor).copy()
        cap = max(len(p), 1) * max_nn
        a = np.zeros(cap, np.int32); b = np.zeros(cap, np.int32); w = np.zeros(cap, np.float32)
        f = self.lib.ref_filters_compute_neighborhood
        f.restype = C.c_int64
        f.argtypes = [f32p, f32p, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_float, i32p, i32p, f32p]
        m = f(p, n_, len(p), int(max_nn), float(radius_sq), float(dist_exp), float(angle_exp), a, b, w)
        return a[:m].copy(), b[:m].copy(), w[:m].copy()


def ref_level_poisson(pos, level, voxel_size=0.0):
    """The REAL reference's level builder (oracle/_ref/libref_ao.so: ref_level_poisson) -> sample indices."""
    lib = C.CDLL(RefAO.PATH)
    pos = _f32(pos)
    out = np.zeros(max(len(pos), 1), np.int32)
    f = lib.ref_level_poisson
    f.restype = C.c_int32
    f.argtypes = [f32p, C.c_int32, C.c_int32, C.c_float, i32p]
    m = f(pos, len(pos), int(level), float(voxel_size), out)
    return out[:m].copy()


LEVEL_VOXEL = (0.005, 0.01, 0.02, 0.04, 0.08)          # rs_pointcloud.h:148


def level_max_n_neigh(level):
    """rs_pointcloud.h:995-996: size_t max_n_neigh = 1024 * (level / (float)(RSPC_N_LEVELS-1)); 256 if that is 0."""
    m = int(np.float32(1024) * (np.float32(level) / np.float32(4)))
    return m if m else 256
