"""CPU: the C restatement against the real reference (oracle/_ref, compiled in place from
/root/reference) on fresh seeded inputs.  Runs wherever oracle/_ref/libref.so exists (the build
container; the .so also travels to the GPU box)."""
import numpy as np
import pytest

from oracle.pyoracle import Ref

pytestmark = pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not built (needs /root/reference)")
I4 = np.eye(4, dtype=np.float32).ravel()


@pytest.fixture(scope="module")
def ref():
    return Ref()


@pytest.fixture(scope="module")
def scene():
    from rescan_amd import synth
    return synth.make_scene(seed=13, density=1800.0, timestep=0, objects=("chair", "shelf", "table"))


@pytest.mark.parametrize("grid_radius,k,r", [(0.05, 64, 0.1), (0.05, 1, 0.05), (0.1, 16, 0.1), (0.1, 16, 0.0663),
                                              (0.05, 8, 0.05), (0.05, 1, 0.075), (0.02, 32, 0.1), (0.3, 4, 0.05)])
def test_radius_search(oracle, ref, scene, grid_radius, k, r):
    rng = np.random.default_rng(int(grid_radius * 1000) + k)
    pts = scene["points"]
    go, gr = oracle.grid_create(pts, grid_radius), ref.grid_create(pts, grid_radius)
    io, ir = oracle.grid_info(go), ref.grid_info(gr)
    assert (io[0] == ir[0]).all() and io[1] == ir[1] and (io[2] == ir[2]).all() and io[3] == ir[3]
    q = pts[rng.integers(0, len(pts), 2000)] + rng.normal(0, 0.02, (2000, 3)).astype(np.float32)
    q[:50] += 5.0
    q[50:100] -= np.float32(0.4)
    for sort in (1, 0):
        a = oracle.radius_search(go, q, r, k, sort)
        b = ref.radius_search(gr, q, r, k, sort)
        assert (a[2] == b[2]).all() and a[3] == b[3]
        m = np.arange(k)[None, :] < a[2][:, None]
        assert (a[0][m] == b[0][m]).all() and (a[1][m] == b[1][m]).all()
    oracle.grid_destroy(go); ref.grid_destroy(gr)


def test_bin_cap_quirk(oracle, ref):
    """radius >> cell: both stop collecting bins at 512 (msh_hash_grid.h:1101,1213)."""
    rng = np.random.default_rng(0)
    pts = rng.uniform(0, 1, (400, 3)).astype(np.float32)
    q = rng.uniform(0, 1, (40, 3)).astype(np.float32)
    go, gr = oracle.grid_create(pts, 0.02), ref.grid_create(pts, 0.02)
    a, b = oracle.radius_search(go, q, 0.6, 32, 1), ref.radius_search(gr, q, 0.6, 32, 1)
    assert (a[2] == b[2]).all() and (a[0] == b[0]).all() and (a[1] == b[1]).all()
    oracle.grid_destroy(go); ref.grid_destroy(gr)


def test_math(oracle, ref):
    rng = np.random.default_rng(1)
    for _ in range(200):
        m = rng.normal(size=16).astype(np.float32); b = rng.normal(size=16).astype(np.float32)
        t = rng.normal(size=3).astype(np.float32); v = rng.normal(size=(7, 3)).astype(np.float32)
        assert (oracle.mat4_inverse(m) == ref.mat4_inverse(m)).all()
        assert (oracle.mat4_mul(m, b) == ref.mat4_mul(m, b)).all()
        assert (oracle.translate(m, t) == ref.translate(m, t)).all()
        a = float(rng.normal())
        for ax in np.eye(3, dtype=np.float32):
            assert (oracle.rotate(m, a, ax) == ref.rotate(m, a, ax)).all()
        ax = rng.normal(size=3).astype(np.float32)
        assert (oracle.rotate(m, a, ax) == ref.rotate(m, a, ax)).all()
        assert (oracle.xform_points(m, v, 1) == ref.xform_points(m, v, 1)).all()
        assert (oracle.xform_points(m, v, 0) == ref.xform_points(m, v, 0)).all()
        assert (oracle.normalize(v) == ref.normalize(v)).all()
    x = rng.uniform(0, 0.01, 5000).astype(np.float32)
    assert oracle.mean(x) == ref.mean(x) and oracle.stddev(oracle.mean(x), x) == ref.stddev(ref.mean(x), x)


def test_icp_and_scores(oracle, ref, scene):
    from rescan_amd import synth
    rng = np.random.default_rng(2)
    pts, nor = scene["points"], scene["normals"]
    for o in scene["objects"]:
        T0 = synth.perturbed_pose(o["pose"], rng)
        for md, deg in ((0.10, 60.0), (0.075, 50.0), (0.05, 10.0)):
            ma = np.float32(np.deg2rad(np.float32(deg)))
            a = oracle.icp_find_corrs(o["pos"], o["nor"], pts, nor, T0, I4, md, ma)
            b = ref.icp_find_corrs(o["pos"], o["nor"], pts, nor, T0, I4, md, ma)
            assert all(x.shape == y.shape and (x == y).all() for x, y in zip(a, b))
            e1, T1 = oracle.icp_estimate_pt2pl(a[0], a[2], a[3], a[4], T0)
            e2, T2 = ref.icp_estimate_pt2pl(b[0], b[2], b[3], b[4], T0)
            assert e1 == e2 and (T1 == T2).all()
            ea, Ta, _ = oracle.icp_align(o["pos"], o["nor"], pts, nor, T0, I4, md, ma)
            eb, Tb, _ = ref.icp_align(o["pos"], o["nor"], pts, nor, T0, I4, md, ma)
            assert ea == eb and (Ta == Tb).all()
        poses = np.stack([synth.perturbed_pose(o["pose"], rng, 0.5, 0.15) for _ in range(6)])
        for K in (64, 32):
            assert (oracle.alignment_scores(pts, nor, o["pos"], o["nor"], poses, K) ==
                    ref.alignment_scores(pts, nor, o["pos"], o["nor"], poses, K)).all()


def test_label_gate_matches_reference_tu(oracle, ref):
    """The gate expression compiled with rs_pointcloud_filters.cpp's include preamble (acosf on
    the float |dot|) vs the restatement, around the 70 degree threshold and over a sweep."""
    c70 = np.float32(np.cos(np.deg2rad(70.0)))
    xs = [np.float32(x) for x in np.linspace(-1.2, 1.2, 3001)]
    x = c70
    for _ in range(200):
        x = np.nextafter(x, np.float32(1)); xs.append(x)
    x = c70
    for _ in range(200):
        x = np.nextafter(x, np.float32(0)); xs.append(x)
    for v in xs:
        assert oracle.label_gate(v) == ref.label_gate_dot(v)
    rng = np.random.default_rng(4)
    from rescan_amd import synth
    for _ in range(300):
        pose = synth.pose_matrix(rng.uniform(0, 6.28), rng.normal(size=3))
        n1 = rng.normal(size=3).astype(np.float32); n2 = rng.normal(size=3).astype(np.float32)
        nm = oracle.xform_points(np.ascontiguousarray(pose.reshape(4, 4).T.ravel()), n1[None], 0)[0]
        u1, u2 = oracle.normalize(nm[None])[0], oracle.normalize(n2[None])[0]
        dot = np.float32(np.float32(np.float32(u1[0] * u2[0]) + np.float32(u1[1] * u2[1])) + np.float32(u1[2] * u2[2]))
        assert oracle.label_gate(dot) == ref.label_gate(pose, n1, n2)


def _label_arrangement(scene, rng, statics, dup):
    """Objects + a shuffled arrangement on `scene`: every dynamic object placed once (three times with `dup`, one of them an
    exact copy of another placement: ties between placements), `statics` = classes of static placements cut from the scan."""
    from rescan_amd import synth
    pts, nor = scene["points"], scene["normals"]
    objs, plcs = [], []
    for oi, o in enumerate(scene["objects"]):
        objs.append(dict(pos=o["pos"], nor=o["nor"], class_idx=o["class_idx"], is_static=0))
        plcs.append(dict(pose=synth.perturbed_pose(o["pose"], rng, 0.02, 0.01), object_idx=oi, uidx=o["uidx"]))
        if dup:
            plcs.append(dict(pose=synth.perturbed_pose(o["pose"], rng, 0.01, 0.005), object_idx=oi, uidx=40 + o["uidx"]))
            plcs.append(dict(pose=plcs[-2]["pose"], object_idx=oi, uidx=80 + o["uidx"]))
    for k, cls_name in enumerate(statics):
        sub = rng.permutation(np.nonzero(scene["instance_idx"] == (0 if cls_name == "floor" else 1))[0])[::2]
        objs.append(dict(pos=np.ascontiguousarray(pts[sub]), nor=np.ascontiguousarray(nor[sub]), class_idx=synth.CLASS_IDX[cls_name], is_static=1))
        plcs.append(dict(pose=I4 if k < 2 else synth.perturbed_pose(I4, rng, 0.004, 0.002), object_idx=len(objs) - 1, uidx=120 + k))
    return objs, [plcs[i] for i in rng.permutation(len(plcs))]


@pytest.mark.parametrize("seed,statics,prio,dup,radius", [
    (1, ("floor", "wall"), 0, False, 0.05),                  # mixed
    (2, (), 0, False, 0.05),                                 # no static placement: first_static = 0, one pass at 1.5 r
    (3, ("floor", "wall"), 1, False, 0.05),                  # prioritize_static
    (4, ("floor",), 0, True, 0.05),                          # ties between placements
    (5, ("wall", "floor", "wall", "floor"), 0, True, 0.05),  # more than one static placement per class
    (6, (), 1, True, 0.03),                                  # no static + prioritize + ties, another radius
    (7, ("wall",), 1, True, 0.08),
    (8, ("floor", "wall"), 0, False, 0.025),                 # the header's default radius
])
def test_labels_vs_reference_text(oracle, scene, seed, statics, prio, dup, radius):
    """The label loops of lib/rs/rs_pointcloud_filters.cpp:738-879 as the reference's OWN compiled text
    (oracle/_ref/libref_filters.so: the file's lines 1-14 + 16-879, oracle/Makefile) against the restatement: temporary
    labels, min_dists, visiting order, class and instance ids, bit for bit."""
    from oracle.pyoracle import RefFilters
    from rescan_amd import synth
    if not RefFilters.available():
        pytest.skip("oracle/_ref/libref_filters.so not built")
    objs, plcs = _label_arrangement(scene, np.random.default_rng(900 + seed), statics, dup)
    pts, nor = scene["points"], scene["normals"]
    a = oracle.arrangement_to_labels(pts, nor, objs, plcs, radius, prio, synth.CLASS_IDX["unlabelled"])
    R = RefFilters(synth.CLASS_IDX)
    b = R.arrangement_to_labels(pts, nor, objs, plcs, radius, prio, synth.CLASS_IDX["unlabelled"])
    R.close()
    for k in ("order", "labels", "min_dists", "class_ids", "instance_ids"):
        assert (a[k] == b[k]).all(), k
    assert (a["labels"] > 0).mean() > 0.05
    if dup:
        assert len(np.unique(a["labels"])) > 3


def test_labels_vs_reference_text_edge_cases(oracle, scene):
    """A single static placement (first_static = 0 again: everything at 1.5 r), a one-point scene, a scene moved out of every
    object's reach.  (An EMPTY arrangement or an empty scene is outside the reference's domain — it dereferences the null
    msh_array, resp. trips msh_hash_grid.h:1097's assert — so those two stay with the restatement's own tests.)"""
    from oracle.pyoracle import RefFilters
    from rescan_amd import synth
    if not RefFilters.available():
        pytest.skip("oracle/_ref/libref_filters.so not built")
    pts, nor = scene["points"], scene["normals"]
    objs, plcs = _label_arrangement(scene, np.random.default_rng(77), ("floor",), False)
    only_static = [p for p in plcs if objs[p["object_idx"]]["is_static"]]
    far = (pts + np.float32(40.0)).astype(np.float32)
    for P, S, N in ((only_static, pts, nor), (plcs[:1], pts[:1], nor[:1]), (plcs, far, nor)):
        a = oracle.arrangement_to_labels(S, N, objs, P, 0.05, 0, 0)
        R = RefFilters(synth.CLASS_IDX)
        b = R.arrangement_to_labels(S, N, objs, P, 0.05, 0, 0)
        R.close()
        for k in ("order", "labels", "min_dists", "class_ids", "instance_ids"):
            assert (a[k][:len(S)] == b[k][:len(S)]).all(), k


def test_neighborhood_vs_reference_text(oracle, scene):
    """rspf_compute_neighborhood (rs_pointcloud_filters.cpp:674-722) as the reference's own compiled text against the
    restatement: the same edges with the same weights — object clouds, the scan (above the int32 key wrap at 46 340 points,
    which both reproduce), other K / exponents."""
    from oracle.pyoracle import RefFilters, edge_digest
    from rescan_amd import synth
    if not RefFilters.available():
        pytest.skip("oracle/_ref/libref_filters.so not built")
    R = RefFilters(synth.CLASS_IDX)
    big = synth.make_scene(seed=3, density=2600.0, timestep=0)
    assert len(big["points"]) > 46340
    clouds = [(o["pos"], o["nor"]) for o in scene["objects"]] + [(scene["points"], scene["normals"]), (big["points"], big["normals"])]
    for pos, nor in clouds:
        for args in ((8, 0.0025, 15.0, 16.0), (4, 0.0016, 2.0, 3.0)):
            a, b, w = oracle.compute_neighborhood(pos, nor, *args)
            ra, rb, rw = R.compute_neighborhood(pos, nor, *args)
            assert len(a) == len(ra)
            n = len(pos)
            if n < 46340:       # unique pair keys: compare edge by edge
                o = np.argsort(np.maximum(ra, rb).astype(np.int64) * n + np.minimum(ra, rb), kind="stable")
                assert (a == ra[o]).all() and (b == rb[o]).all() and (w.view(np.uint32) == rw[o].view(np.uint32)).all()
            else:
                assert (edge_digest(a, b, w) == edge_digest(ra, rb, rw)).all()
                ka = np.sort(a.astype(np.int64) * n + b); kb = np.sort(ra.astype(np.int64) * n + rb)
                assert (ka == kb).all()
    R.close()


def test_neighborhood_composition(oracle, ref, scene):
    """rspf_compute_neighborhood (rs_pointcloud_filters.cpp:674-722) once more, from the reference's pieces composed in
    Python (kept from the rounds in which that TU was not compiled; test_neighborhood_vs_reference_text runs the function
    itself): the unsorted K=8 search of msh_hash_grid.h, the edge weight from the reference-toolchain TU, and the
    first-insertion-wins de-duplication on max*n+min."""
    pts, nor = scene["points"][::3].copy(), scene["normals"][::3].copy()
    n = len(pts)
    assert n < 46340          # no int32 key wrap in this case
    gr = ref.grid_create(pts, 0.05)
    radius = np.float32(np.sqrt(np.float64(np.float32(0.05 * 0.05))))
    d, idx, nn, _ = ref.radius_search(gr, pts, float(radius), 8, 0)
    ref.grid_destroy(gr)
    seen, e1, e2, ew = set(), [], [], []
    for i in range(n):
        for t in range(int(nn[i])):
            j = int(idx[i, t])
            key = max(i, j) * n + min(i, j)
            if key in seen:
                continue
            seen.add(key)
            dot = np.float32(np.float32(nor[i, 0] * nor[j, 0]) + np.float32(nor[i, 1] * nor[j, 1])) + np.float32(nor[i, 2] * nor[j, 2])
            e1.append(i); e2.append(j); ew.append(ref.edge_cost(d[i, t], dot))
    key = np.array([max(a, b) * n + min(a, b) for a, b in zip(e1, e2)], np.int64)
    o = np.argsort(key, kind="stable")
    a, b, w = oracle.compute_neighborhood(pts, nor)
    assert len(a) == len(o)
    assert (a == np.array(e1, np.int32)[o]).all() and (b == np.array(e2, np.int32)[o]).all()
    assert (w.view(np.uint32) == np.array(ew, np.float32)[o].view(np.uint32)).all()


def test_edge_cost(oracle, ref):
    rng = np.random.default_rng(9)
    for _ in range(5000):
        d2, dot = np.float32(rng.uniform(0, 0.0026)), np.float32(rng.uniform(-0.1, 1.05))
        assert oracle.edge_cost(d2, dot) == ref.edge_cost(d2, dot)
    for de, ae in ((2.0, 3.0), (1.5, 0.5), (15.0, 16.0)):
        for _ in range(300):
            d2, dot = np.float32(rng.uniform(0, 0.0026)), np.float32(rng.uniform(0, 1))
            assert oracle.edge_cost(d2, dot, 0.0025, de, ae) == ref.edge_cost(d2, dot, 0.0025, de, ae)


def test_coverage_vs_reference(oracle, scene):
    """Fresh inputs through the reference's arrangement_optimization.cpp (oracle/_ref/libref_ao.so)."""
    from oracle.pyoracle import RefAO
    from rescan_amd import synth
    if not RefAO.available():
        pytest.skip("oracle/_ref/libref_ao.so not built")
    pts = scene["points"]
    rng = np.random.default_rng(4)
    bmin, bmax = pts.min(0), pts.max(0)
    for voxel in (0.05, 0.08):
        R = RefAO(synth.CLASS_IDX, pts, bmin, bmax, voxel_size=voxel)
        g = oracle.voxgrid(bmin, bmax, voxel)
        assert (g.x_res, g.y_res, g.z_res) == tuple(R.res) and (np.array(list(g.origin), np.float32) == R.origin).all()
        sd = oracle.rasterize_scene(g, pts)
        assert (sd == R.scene_grid()).all()
        idx = [R.add_object(o["pos"], o["class_idx"], o["uidx"]) for o in scene["objects"]]
        for _ in range(6):
            poses = [synth.perturbed_pose(o["pose"], rng, 0.3, 0.2) for o in scene["objects"]]     # some points leave the grid
            want = R.coverage(idx, poses)
            ad = oracle.rasterize_arrangement(g, [o["pos"] for o in scene["objects"]], poses, [0] * len(idx))
            assert (ad == R.arrangement_grid()).all()
            assert oracle.coverage_score(sd, ad)[0] == want


def test_level_poisson_vs_reference(oracle):
    """The level builder's restatement against the reference's own function on fresh clouds (seeded scene in three
    point orders, a tiny cloud, coincident points)."""
    from oracle.pyoracle import RefAO, ref_level_poisson, LEVEL_VOXEL, level_max_n_neigh
    if not RefAO.available():
        pytest.skip("oracle/_ref/libref_ao.so not built")
    from rescan_amd import synth
    pts = synth.make_scene(seed=5, density=1500.0, timestep=0)["points"]
    rng = np.random.default_rng(2)
    clouds = [pts, np.ascontiguousarray(pts[np.lexsort((pts[:, 0], pts[:, 1], pts[:, 2]))]), np.ascontiguousarray(pts[rng.permutation(len(pts))]),
              np.ascontiguousarray(pts[:7]), np.repeat(pts[:3], 5, axis=0)]
    for p in clouds:
        for level in (1, 2, 4):
            a = oracle.level_poisson(p, LEVEL_VOXEL[level], level_max_n_neigh(level))
            b = ref_level_poisson(p, level)
            assert len(a) == len(b) and (a == b).all()
