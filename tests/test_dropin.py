"""The drop-in boundary: librescan_dropin.so exports the reference's own symbol names
(icp_align, icp_find_corrs, msh_hash_grid_*) with the reference's struct layouts."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden_files, load_golden

DROPIN = os.path.join(ROOT, "rescan_amd", "librescan_dropin.so")


class Mat4(C.Structure):
    _fields_ = [("data", C.c_float * 16)]


class HashGrid(C.Structure):                       # msh_hash_grid_t, lib/msh/msh_hash_grid.h:248-269
    _fields_ = [("width", C.c_size_t), ("height", C.c_size_t), ("depth", C.c_size_t), ("cell_size", C.c_double),
                ("min_pt", C.c_float * 3), ("max_pt", C.c_float * 3), ("bin_table", C.c_void_p),
                ("data_buffer", C.c_void_p), ("offsets", C.c_void_p), ("_slab_size", C.c_int32),
                ("_inv_cell_size", C.c_double), ("_pts_dim", C.c_uint8), ("_num_threads", C.c_uint16),
                ("_dont_use_omp", C.c_int32), ("max_n_pts_in_bin", C.c_uint32), ("_n_pts", C.c_size_t)]


class SearchDesc(C.Structure):                     # msh_hash_grid_search_desc_t, :196-216
    _fields_ = [("query_pts", C.c_void_p), ("n_query_pts", C.c_size_t), ("distances_sq", C.c_void_p),
                ("indices", C.c_void_p), ("n_neighbors", C.c_void_p), ("radius", C.c_float),
                ("max_n_neigh", C.c_size_t), ("sort", C.c_int)]


def declared():
    hdr = open(os.path.join(ROOT, "include", "rescan_dropin.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b((?:icp|msh_hash_grid|rsd)_[a-z0-9_]+)\s*\(", hdr)))


def test_symbols_and_layout():
    from rescan_amd import build
    build.build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", DROPIN], text=True)
    exported = set(re.findall(r" T ([a-z0-9_]+)", out))
    names = declared()
    assert {"icp_align", "icp_find_corrs", "icp_estimate_rigid_xform_pt2pl", "msh_hash_grid_init_3d",
            "msh_hash_grid_term", "msh_hash_grid_radius_search"} <= set(names)
    assert set(names) <= exported
    assert C.sizeof(HashGrid) == 120 and HashGrid.data_buffer.offset == 64 and HashGrid._n_pts.offset == 112
    assert C.sizeof(SearchDesc) == 64 and C.sizeof(Mat4) == 64


def test_layout_matches_the_reference_headers():
    from oracle.pyoracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref not built")
    lib = Ref().lib
    lib.ref_layout.restype = C.c_int64
    want = [12, 64, C.sizeof(HashGrid), HashGrid.data_buffer.offset, HashGrid._n_pts.offset, C.sizeof(SearchDesc),
            SearchDesc.radius.offset, SearchDesc.sort.offset, HashGrid.cell_size.offset]
    assert [lib.ref_layout(i) for i in range(9)] == want


@pytest.fixture(scope="module")
def dropin():
    from rescan_amd import capi
    capi.init(0)
    lib = C.CDLL(DROPIN)
    lib.icp_align.restype = C.c_float
    lib.icp_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                              C.POINTER(Mat4), Mat4, C.c_float, C.c_float, C.c_bool]
    lib.msh_hash_grid_init_3d.restype = None
    lib.msh_hash_grid_init_3d.argtypes = [C.POINTER(HashGrid), C.c_void_p, C.c_int32, C.c_float]
    lib.msh_hash_grid_term.restype = None
    lib.msh_hash_grid_term.argtypes = [C.POINTER(HashGrid)]
    lib.msh_hash_grid_radius_search.restype = C.c_size_t
    lib.msh_hash_grid_radius_search.argtypes = [C.POINTER(HashGrid), C.POINTER(SearchDesc)]
    lib.rsd_alignment_scores.restype = C.c_int
    lib.rsd_alignment_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_void_p]
    return lib


@pytest.mark.gpu
@pytest.mark.parametrize("fname", golden_files("icp_")[:4])
def test_icp_align_by_reference_name(dropin, gscene, fname):
    g = load_golden(fname)
    o = gscene["objects"][int(g["obj"])]
    T1 = Mat4(); T1.data[:] = [float(x) for x in g["T1"]]
    T2 = Mat4(); T2.data[:] = [float(x) for x in g["T2"]]
    pts2, nor2 = gscene["points"], gscene["normals"]
    for _ in range(2):                       # second call hits the device-cloud cache
        T = Mat4(); T.data[:] = T1.data[:]
        err = dropin.icp_align(o["pos"].ctypes.data, o["nor"].ctypes.data, len(o["pos"]), pts2.ctypes.data,
                               nor2.ctypes.data, len(pts2), C.byref(T), T2, float(g["max_dist"]),
                               float(g["max_angle"]), False)
        got = np.array(T.data[:], np.float32)
        assert np.linalg.norm(got.astype(np.float64) - g["T_out"]) < 1e-4 and abs(err - float(g["err"])) < 1e-5


def _search(dropin, hg, q, radius, k, sort=1):
    q = np.ascontiguousarray(q, np.float32).reshape(-1, 3)
    d = np.zeros((len(q), k), np.float32); i = np.zeros((len(q), k), np.int32); nn = np.zeros(len(q), np.uint64)
    sd = SearchDesc(q.ctypes.data, len(q), d.ctypes.data, i.ctypes.data, nn.ctypes.data, float(radius), k, sort)
    tot = dropin.msh_hash_grid_radius_search(C.byref(hg), C.byref(sd))
    return d, i, nn.astype(np.int64), int(tot)


@pytest.mark.gpu
@pytest.mark.parametrize("fname", golden_files("rows_"))
def test_hash_grid_by_reference_name(dropin, gscene, fname, monkeypatch):
    """msh_hash_grid_init_3d / _radius_search / _term under the reference's names, on the reference's own rows: the
    batched call through both device kernels (one wave per query; Hilbert tiles + successive minima) and the same
    queries asked one at a time (the host path the level builder's one-query loop takes)."""
    from conftest import rows_equal_up_to_ties
    g = load_golden(fname)
    hg = HashGrid()
    pts = gscene["points"]
    dropin.msh_hash_grid_init_3d(C.byref(hg), pts.ctypes.data, len(pts), float(g["grid_radius"]))
    assert hg.data_buffer and hg._n_pts == len(pts)
    q = g["query"]; k = int(g["k"])
    for route in ("wave", "tiled"):
        monkeypatch.setenv("RS_HIP_ROWS_TILED_FROM", "1000000000" if route == "wave" else "0")
        if route == "tiled":
            monkeypatch.setenv("RS_HIP_NO_ROWS_WAVE", "1")
        d, i, nn, tot = _search(dropin, hg, q, g["radius"], k)
        assert tot == int(g["total"]), route
        rows_equal_up_to_ties(d, i, nn, g["dists"], g["inds"], g["nn"])
    monkeypatch.delenv("RS_HIP_NO_ROWS_WAVE")
    for r in range(0, len(q), 7):                                # one query per call: answered from the grid's host copy
        d, i, nn, tot = _search(dropin, hg, q[r], g["radius"], k, sort=0)
        rows_equal_up_to_ties(d, i, nn, g["dists"][r:r + 1], g["inds"][r:r + 1], g["nn"][r:r + 1])
        assert tot == int(g["nn"][r])
    dropin.msh_hash_grid_term(C.byref(hg))
    assert not hg.data_buffer and hg.width == 0


@pytest.mark.gpu
def test_level_builder_loop_through_the_shim(dropin, gscene):
    """rs_pointcloud__compute_level_poisson (lib/rs/rs_pointcloud.h:984-1038), the loop as the reference runs it — one
    one-query search per sample against a grid of radius 2.5 x voxel, sort = 0, max_n_neigh = 1024 * level / 4 — on top of
    the shim's msh_hash_grid_*: the samples are the reference's (tests/golden/level.npz, made by its own function)."""
    g = load_golden("level.npz")
    pts = gscene["points"]
    for level, voxel in ((2, 0.02), (4, 0.08)):
        hg = HashGrid()
        dropin.msh_hash_grid_init_3d(C.byref(hg), pts.ctypes.data, len(pts), 2.5 * voxel)            # :988-990
        k = int(1024 * (level / 4.0)) or 256                                                         # :995-996
        unmarked = np.ones(len(pts), bool)
        samples = []
        d = np.zeros(k, np.float32); i = np.zeros(k, np.int32)
        sd = SearchDesc(0, 1, d.ctypes.data, i.ctypes.data, None, float(voxel), k, 0)
        idx = 0
        n_marked = 0
        while n_marked < len(pts):
            while not unmarked[idx]:
                idx += 1
            samples.append(idx)
            sd.query_pts = pts[idx:idx + 1].ctypes.data
            n = dropin.msh_hash_grid_radius_search(C.byref(hg), C.byref(sd))
            hit = i[:n]
            n_marked += int(unmarked[hit].sum())
            unmarked[hit] = False
        dropin.msh_hash_grid_term(C.byref(hg))
        assert np.array_equal(np.array(samples, np.int32), g[f"own_l{level}"]), level


@pytest.mark.gpu
def test_upload_cache_follows_the_arrays(dropin, gscene):
    """The shim's device-cloud cache: same arrays -> same result (cached upload); arrays rebuilt in place -> noticed by the
    sampled fingerprint; a few points edited in place -> noticed after rsd_cache_invalidate (the explicit route)."""
    g = load_golden(golden_files("icp_")[0])
    o = gscene["objects"][int(g["obj"])]
    pos, nor = o["pos"].copy(), o["nor"].copy()
    pts2, nor2 = gscene["points"], gscene["normals"]
    T2 = Mat4(); T2.data[:] = [float(x) for x in g["T2"]]
    dropin.rsd_cache_invalidate.restype = None
    dropin.rsd_cache_invalidate.argtypes = [C.c_void_p]

    def run():
        T = Mat4(); T.data[:] = [float(x) for x in g["T1"]]
        dropin.icp_align(pos.ctypes.data, nor.ctypes.data, len(pos), pts2.ctypes.data, nor2.ctypes.data, len(pts2),
                         C.byref(T), T2, float(g["max_dist"]), float(g["max_angle"]), False)
        return np.array(T.data[:], np.float32)

    a = run()
    assert (run() == a).all() and np.linalg.norm(a.astype(np.float64) - g["T_out"]) < 1e-4
    saved = pos.copy()
    pos += np.float32(0.01)                                      # the whole array changes behind the same pointer
    b = run()
    assert not (b == a).all()
    pos[:] = saved
    assert (run() == a).all()
    pos[len(pos) // 3] += np.float32(0.05)                       # one point: invisible to a sample, so the caller says so
    dropin.rsd_cache_invalidate(pos.ctypes.data)
    c = run()
    pos[:] = saved
    dropin.rsd_cache_invalidate(pos.ctypes.data)
    assert (run() == a).all() and not (c == a).all()


@pytest.mark.gpu
def test_scores_by_flat_entry(dropin, gscene):
    g = load_golden("scores_table0_k64.npz")
    o = gscene["objects"][int(g["obj"])]
    poses = np.ascontiguousarray(g["poses"], np.float32)
    out = np.zeros(len(poses), np.float32)
    rc = dropin.rsd_alignment_scores(o["pos"].ctypes.data, o["nor"].ctypes.data, len(o["pos"]),
                                     gscene["points"].ctypes.data, gscene["normals"].ctypes.data, len(gscene["points"]),
                                     poses.ctypes.data, len(poses), 0.1, 64, out.ctypes.data)
    assert rc == 0 and np.abs(out.astype(np.float64) - g["scores"]).max() < 2e-6


@pytest.mark.gpu
def test_neighborhood_by_flat_entry(dropin, gscene):
    d = load_golden("neighborhood_obj1.npz")
    o = gscene["objects"][int(d["obj"])]
    n = len(o["pos"])
    a = np.zeros(n * 8, np.int32); b = np.zeros(n * 8, np.int32); w = np.zeros(n * 8, np.float32)
    dropin.rsd_compute_neighborhood.restype = C.c_int64
    dropin.rsd_compute_neighborhood.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float,
                                                C.c_void_p, C.c_void_p, C.c_void_p]
    m = dropin.rsd_compute_neighborhood(o["pos"].ctypes.data, o["nor"].ctypes.data, n, 8, 0.05 * 0.05, 15.0, 16.0,
                                        a.ctypes.data, b.ctypes.data, w.ctypes.data)
    assert m == len(d["idx1"])
    key = lambda x, y: np.sort(np.maximum(x, y).astype(np.int64) * n + np.minimum(x, y))
    assert (key(a[:m], b[:m]) == key(d["idx1"], d["idx2"])).all()


@pytest.mark.gpu
def test_level_poisson_by_flat_entry(dropin, gscene):
    g = load_golden("level.npz")
    pts = gscene["points"]
    out = np.zeros(len(pts), np.int32)
    dropin.rsd_level_poisson.restype = C.c_int32
    dropin.rsd_level_poisson.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_void_p]
    for level, voxel in ((2, 0.02), (4, 0.08)):
        m = dropin.rsd_level_poisson(pts.ctypes.data, len(pts), voxel, level, out.ctypes.data)
        assert m == len(g[f"own_l{level}"]) and (out[:m] == g[f"own_l{level}"]).all()


@pytest.mark.gpu
def test_coverage_by_flat_entry(dropin, gscene):
    from conftest import coverage_case
    d, objs, static, arrangements = coverage_case(gscene)
    dropin.rsd_coverage_create.restype = C.c_void_p
    dropin.rsd_coverage_create.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_float]
    dropin.rsd_coverage_score.restype = C.c_float
    dropin.rsd_coverage_score.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    dropin.rsd_coverage_destroy.argtypes = [C.c_void_p]
    bmin, bmax = np.ascontiguousarray(d["bbox_min"], np.float32), np.ascontiguousarray(d["bbox_max"], np.float32)
    pts, q = gscene["points"], np.ascontiguousarray(d["quality"], np.float32)
    h = dropin.rsd_coverage_create(bmin.ctypes.data, bmax.ctypes.data, 0.05, pts.ctypes.data, q.ctypes.data, len(pts), 0.5)
    assert h
    keep = [np.ascontiguousarray(o, np.float32) for o in objs]
    for a in (0, 7, 23):
        plc = arrangements[a]
        ptrs = (C.c_void_p * len(plc))(*[keep[k].ctypes.data for k, _ in plc])
        ns = np.array([len(keep[k]) for k, _ in plc], np.int32)
        poses = np.ascontiguousarray(np.array([p for _, p in plc], np.float32))
        st = np.array([static[k] for k, _ in plc], np.int32)
        s = dropin.rsd_coverage_score(h, C.addressof(ptrs), ns.ctypes.data, poses.ctypes.data, st.ctypes.data, len(plc))
        assert np.float32(s) == d["scores"][a]
    dropin.rsd_coverage_destroy(h)
