"""CPU: the plain-C oracle (oracle/rs_oracle.c) against the golden vectors generated from the
compiled reference (oracle/gen_golden.py -> tests/golden/).  Everything here is bit-exact."""
import numpy as np
import pytest

from conftest import coverage_case, golden_files, label_case, load_golden

I4 = np.eye(4, dtype=np.float32).ravel()


@pytest.mark.parametrize("fname", golden_files("rows_"))
def test_rows(oracle, gscene, fname):
    g = load_golden(fname)
    grid = oracle.grid_create(gscene["points"], float(g["grid_radius"]))
    d, i, nn, tot = oracle.radius_search(grid, g["query"], float(g["radius"]), int(g["k"]), 1)
    oracle.grid_destroy(grid)
    assert (nn == g["nn"]).all() and tot == int(g["total"])
    valid = np.arange(int(g["k"]))[None, :] < nn[:, None]
    assert (d[valid] == g["dists"][valid]).all()
    assert (i[valid] == g["inds"][valid]).all()          # including the reference's order among ties


@pytest.mark.parametrize("fname", golden_files("corrs_"))
def test_find_corrs(oracle, gscene, fname):
    g = load_golden(fname)
    o = gscene["objects"][int(g["obj"])]
    out = oracle.icp_find_corrs(o["pos"], o["nor"], gscene["points"], gscene["normals"], g["T1"], g["T2"],
                                g["max_dist"], g["max_angle"])
    for a, name in zip(out, ("c_pts1", "c_nor1", "c_pts2", "c_nor2", "weights")):
        assert a.shape == g[name].shape and (a == g[name]).all(), name


@pytest.mark.parametrize("fname", golden_files("icp_"))
def test_icp_align(oracle, gscene, fname):
    g = load_golden(fname)
    o = gscene["objects"][int(g["obj"])]
    err, T, it = oracle.icp_align(o["pos"], o["nor"], gscene["points"], gscene["normals"], g["T1"], g["T2"],
                                  g["max_dist"], g["max_angle"])
    assert np.float32(err) == g["err"] and (T == g["T_out"]).all() and it == int(g["iters"])


@pytest.mark.parametrize("fname", golden_files("scores_"))
def test_scores(oracle, gscene, fname):
    g = load_golden(fname)
    o = gscene["objects"][int(g["obj"])]
    sc = oracle.alignment_scores(gscene["points"], gscene["normals"], o["pos"], o["nor"], g["poses"], int(g["k"]))
    assert (sc == g["scores"]).all()


def test_mat4(oracle):
    g = load_golden("mat4.npz")
    for k in range(len(g["m"])):
        assert (oracle.mat4_inverse(g["m"][k]) == g["inverse"][k]).all()
        assert (oracle.mat4_mul(g["m"][k], g["b"][k]) == g["mul"][k]).all()
        assert (oracle.translate(g["m"][k], g["t"][k]) == g["translate"][k]).all()
        assert (oracle.rotate(g["m"][k], g["angle"][k], g["axis"][k]) == g["rotate"][k]).all()


def test_label_gate(oracle):
    g = load_golden("gates.npz")
    got = np.array([oracle.label_gate(x) for x in g["dot"]], np.int8)
    assert (got == g["label_accept"]).all()


@pytest.mark.parametrize("fname", golden_files("labels_"))
def test_labels_reproducible(oracle, gscene, fname):
    """labels_*.npz hold the REFERENCE's outputs (rs_pointcloud_filters.cpp:738-879 compiled from its own text,
    oracle/_ref/libref_filters.so, oracle/gen_golden.py): temporary labels, min_dists, visiting order, class / instance ids —
    the restatement must reproduce all of them bit for bit (mixed, no static placement, prioritize_static, exact ties
    between placements, several static placements)."""
    d, objs, plcs = label_case(gscene, fname)
    assert "reference" in str(d["source"])
    res = oracle.arrangement_to_labels(gscene["points"], gscene["normals"], objs, plcs, 0.05, int(d["prioritize_static"]), 0)
    for k in ("labels", "min_dists", "order", "class_ids", "instance_ids"):
        assert (res[k] == d[k]).all(), k
    assert (res["labels"] > 0).mean() > 0.1


def test_labels_semantics(oracle, gscene):
    """Order dependence of the label loop: with equal distances the earlier placement wins, static
    placements come after dynamic ones, and 'no static placement' runs everything at 1.5*radius
    (rs_pointcloud_filters.cpp:830-848)."""
    pts, nor = gscene["points"], gscene["normals"]
    o = gscene["objects"][1]
    objs = [dict(pos=o["pos"], nor=o["nor"], class_idx=5, is_static=0)]
    twice = [dict(pose=o["pose"], object_idx=0, uidx=10), dict(pose=o["pose"], object_idx=0, uidx=11)]
    res = oracle.arrangement_to_labels(pts, nor, objs, twice, 0.05, 0, 0)
    assert set(np.unique(res["labels"])) <= {0, 1}                       # the duplicate never wins (strict <)
    # no static object -> first_static = 0 -> the only pass uses 1.5*radius
    one = oracle.arrangement_to_labels(pts, nor, objs, twice[:1], 0.05, 0, 0)
    wide = oracle.arrangement_to_labels(pts, nor, objs, twice[:1], 0.075, 1, 0)   # prioritize_static keeps `radius`
    assert (one["labels"] == wide["labels"]).all()
    # empty arrangement
    none = oracle.arrangement_to_labels(pts, nor, objs, [], 0.05, 0, 0)
    assert (none["labels"] == 0).all() and (none["instance_ids"] == 1024).all()


def test_edge_cost(oracle):
    """rs_pointcloud_filters.cpp:706-708 — restatement vs the reference-toolchain values."""
    d = load_golden("edge_cost.npz")
    got = np.array([oracle.edge_cost(a, b) for a, b in zip(d["d2"], d["dot"])], np.float32)
    assert (got.view(np.uint32) == d["cost"].view(np.uint32)).all()


def test_neighborhood_reproducible(oracle, gscene):
    """neighborhood_*.npz hold the REFERENCE's edges (rspf_compute_neighborhood, rs_pointcloud_filters.cpp:674-722, compiled
    from its own text), sorted by pair key."""
    from oracle.pyoracle import edge_digest
    for fname in golden_files("neighborhood_obj"):
        d = load_golden(fname)
        o = gscene["objects"][int(d["obj"])]
        a, b, w = oracle.compute_neighborhood(o["pos"], o["nor"])
        assert (a == d["idx1"]).all() and (b == d["idx2"]).all() and (w.view(np.uint32) == d["weight"].view(np.uint32)).all()
    a, b, w = oracle.compute_neighborhood(gscene["points"], gscene["normals"])
    assert (edge_digest(a, b, w) == load_golden("neighborhood_scene.npz")["digest"]).all()


def test_neighborhood_semantics(oracle, gscene):
    """every point has its self edge; pairs are unique; each edge joins points within the radius;
    at most max_nn directed entries per point."""
    o = gscene["objects"][0]
    n = len(o["pos"])
    a, b, w = oracle.compute_neighborhood(o["pos"], o["nor"])
    assert ((a == b).sum() == n)
    key = np.maximum(a, b).astype(np.int64) * n + np.minimum(a, b)
    assert len(np.unique(key)) == len(key)
    d2 = ((o["pos"][a] - o["pos"][b]) ** 2).sum(1)
    assert (d2 < 0.0025 * 1.0001).all()
    assert (w >= 0).all() and (w <= 1).all()
    assert np.bincount(a, minlength=n).max() <= 8


def test_coverage(oracle, gscene):
    """Scene-coverage term: restatement vs the values the reference TU produced (coverage.npz)."""
    from conftest import coverage_case
    d, objs, static, arrangements = coverage_case(gscene)
    g = oracle.voxgrid(d["bbox_min"], d["bbox_max"])
    assert (g.x_res, g.y_res, g.z_res) == tuple(d["res"]) and g.n_cells == int(d["n_cells"])
    assert (np.array(list(g.origin), np.float32) == d["origin"]).all()
    sd = oracle.rasterize_scene(g, gscene["points"], d["quality"], 0.5)
    assert (sd == np.unpackbits(d["scene_grid"])[:g.n_cells]).all()
    for a, plc in enumerate(arrangements):
        ad = oracle.rasterize_arrangement(g, [objs[k] for k, _ in plc], [p for _, p in plc], [static[k] for k, _ in plc])
        s, agree, valid = oracle.coverage_score(sd, ad)
        assert s == d["scores"][a]
    # empty scene grid -> score 0, no NaN (:367)
    s, agree, valid = oracle.coverage_score(np.zeros(g.n_cells, np.uint8), ad)
    assert s == 0 and valid == 0


def test_level_poisson(oracle, gscene):
    """rs_pointcloud__compute_level_poisson: the restatement against the sample indices the REFERENCE
    itself produced (tests/golden/level.npz, oracle/gen_golden.py: gen_level), for the scene's own point order and
    a raster order, levels 1-4."""
    from oracle.pyoracle import LEVEL_VOXEL, level_max_n_neigh
    g = load_golden("level.npz")
    pts = gscene["points"]
    ras = np.ascontiguousarray(pts[g["raster"]])
    assert [level_max_n_neigh(l) for l in range(5)] == [256, 256, 512, 768, 1024]
    for level in (1, 2, 3, 4):
        for name, p in (("own", pts), ("raster", ras)):
            got = oracle.level_poisson(p, LEVEL_VOXEL[level], level_max_n_neigh(level))
            want = g[f"{name}_l{level}"]
            assert len(got) == len(want) and (got == want).all(), (name, level)
            assert (np.diff(got) > 0).all() and got[0] == 0            # increasing, and the first point is always a sample


def test_headline_static_arrangement_restatement(oracle):
    """The restatement's label loops at the headline's size, with static placements (0.1 - 0.3 M-point objects, the 1.5 x radius pass,
    both values of prioritize_static), against the reference's digests (tests/golden/bench_labels_static_seed11.npz, generated by
    the reference's own text: oracle/gen_golden_bench.py --static-labels)."""
    import hashlib
    import sys
    from conftest import ROOT, static_label_case
    sys.path.insert(0, ROOT)
    import bench
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    w = bench.build_inputs(1_000_000, seed=11)
    g, objs, plcs = static_label_case(w)
    for prio in (0, 1):
        res = oracle.arrangement_to_labels(w["s1"]["points"], w["s1"]["normals"], objs, plcs, 0.05, prio, 0)
        assert (res["order"] == g[f"order_prio{prio}"]).all()
        for k in ("labels", "min_dists", "class_ids", "instance_ids"):
            assert sha(res[k]) == str(g[f"{k}_prio{prio}"]), (k, prio)
