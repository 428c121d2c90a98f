"""Seeded synthetic indoor scenes (own code; stands in for the Rescan dataset, which is not
available offline — SURVEY.md §8d).  y is up, the floor is y = 0 (the reference's pose grid
search rotates about +y at height 0, apps/pose_proposal/pose_proposal.cpp:219-222).

Everything here is plain numpy and deterministic for a given seed.  Surfaces are sampled
uniformly at `density` points per square metre (the reference resamples meshes at
6400 pts/m², lib/rs/rs_pointcloud.h:1132-1227) with Gaussian jitter along all axes; normals
are the unit normals of the generating faces.
"""
import numpy as np

DENSITY = 6400.0
JITTER = 0.0007
# Normals of real scans are estimated from noisy points; perfectly axis-aligned synthetic normals
# would make the ICP normal equations exactly rank-deficient whenever one face orientation drops
# out of the correspondence set, and the reference's unpivoted LDL^T (lib/rs/lineqn.h:153-196) then
# returns rounding noise.  A small angular perturbation keeps the systems well-posed.
NORMAL_NOISE = 0.03


def rot_y(angle):
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def pose_matrix(angle_y, t):
    """4x4 column-major float32[16] pose: rotation about +y then translation."""
    m = np.eye(4, dtype=np.float64)
    m[:3, :3] = rot_y(angle_y)
    m[:3, 3] = t
    return np.ascontiguousarray(m.T.astype(np.float32).ravel())   # column-major


def apply_pose(pose16, pts, is_point=True):
    m = np.asarray(pose16, np.float64).reshape(4, 4).T
    out = pts.astype(np.float64) @ m[:3, :3].T
    if is_point:
        out += m[:3, 3]
    return out.astype(np.float32)


def _sample_rect(rng, origin, eu, ev, normal, density):
    """Uniform samples on the parallelogram origin + a*eu + b*ev, a,b in [0,1)."""
    area = np.linalg.norm(np.cross(eu, ev))
    n = int(rng.poisson(area * density)) if area * density < 50 else int(round(area * density))
    ab = rng.random((n, 2))
    pts = origin[None, :] + ab[:, :1] * eu[None, :] + ab[:, 1:] * ev[None, :]
    nor = np.repeat(np.asarray(normal, np.float64)[None, :], n, axis=0)
    return pts, nor


def sample_box(rng, center, size, density, skip_bottom=False):
    """Six faces of an axis-aligned box, outward normals."""
    c = np.asarray(center, np.float64)
    h = np.asarray(size, np.float64) / 2
    P, N = [], []
    for ax in range(3):
        u, v = (ax + 1) % 3, (ax + 2) % 3
        for sgn in (-1.0, 1.0):
            if skip_bottom and ax == 1 and sgn < 0:
                continue
            o = c.copy(); o[ax] += sgn * h[ax]; o[u] -= h[u]; o[v] -= h[v]
            eu = np.zeros(3); eu[u] = 2 * h[u]
            ev = np.zeros(3); ev[v] = 2 * h[v]
            nrm = np.zeros(3); nrm[ax] = sgn
            p, n = _sample_rect(rng, o, eu, ev, nrm, density)
            P.append(p); N.append(n)
    return np.concatenate(P), np.concatenate(N)


def make_chair(rng, density=DENSITY, scale=1.0):
    s = scale
    parts = [((0, 0.45 * s, 0), (0.45 * s, 0.04 * s, 0.45 * s)),            # seat
             ((0, 0.72 * s, -0.205 * s), (0.45 * s, 0.50 * s, 0.04 * s))]    # backrest
    for sx in (-1, 1):
        for sz in (-1, 1):
            parts.append(((sx * 0.19 * s, 0.215 * s, sz * 0.19 * s), (0.04 * s, 0.43 * s, 0.04 * s)))
    P, N = zip(*[sample_box(rng, c, sz, density) for c, sz in parts])
    return np.concatenate(P), np.concatenate(N)


def make_table(rng, density=DENSITY, scale=1.0):
    s = scale
    parts = [((0, 0.74 * s, 0), (1.2 * s, 0.04 * s, 0.8 * s))]
    for sx in (-1, 1):
        for sz in (-1, 1):
            parts.append(((sx * 0.55 * s, 0.36 * s, sz * 0.35 * s), (0.05 * s, 0.72 * s, 0.05 * s)))
    P, N = zip(*[sample_box(rng, c, sz, density) for c, sz in parts])
    return np.concatenate(P), np.concatenate(N)


def make_shelf(rng, density=DENSITY, scale=1.0):
    """An L-shaped, asymmetric object (well-conditioned for ICP in all 6 dof)."""
    s = scale
    parts = [((0, 0.5 * s, 0), (0.8 * s, 1.0 * s, 0.05 * s)),
             ((-0.375 * s, 0.5 * s, 0.2 * s), (0.05 * s, 1.0 * s, 0.35 * s)),
             ((0.1 * s, 0.3 * s, 0.2 * s), (0.9 * s, 0.04 * s, 0.35 * s)),
             ((0.2 * s, 0.8 * s, 0.15 * s), (0.4 * s, 0.04 * s, 0.25 * s))]
    P, N = zip(*[sample_box(rng, c, sz, density) for c, sz in parts])
    return np.concatenate(P), np.concatenate(N)


OBJECT_MAKERS = {"chair": make_chair, "table": make_table, "shelf": make_shelf}
# class indices in the style of an nyu40 class file (wall/floor are static, rs_database.h:257-288)
CLASS_IDX = {"unlabelled": 0, "wall": 1, "floor": 2, "chair": 5, "table": 7, "shelf": 15}
STATIC_CLASSES = ("wall", "floor", "unlabelled")


def noisy_normals(rng, N, sigma=NORMAL_NOISE):
    N = N + rng.normal(0.0, sigma, N.shape)
    return N / np.linalg.norm(N, axis=1, keepdims=True)


def _finish(rng, P, N, jitter):
    P = P + rng.normal(0.0, jitter, P.shape)
    return P.astype(np.float32), noisy_normals(rng, N).astype(np.float32)


def make_object(kind, seed, density=DENSITY, scale=1.0, jitter=JITTER):
    rng = np.random.default_rng(seed)
    P, N = OBJECT_MAKERS[kind](rng, density, scale)
    return _finish(rng, P, N, jitter)


def make_room_shell(rng, width, depth, height, density):
    """Floor (y=0, normal +y) and two walls (x=0 normal +x, z=0 normal +z)."""
    parts = [
        _sample_rect(rng, np.zeros(3), np.array([width, 0, 0.]), np.array([0, 0, depth]), (0, 1, 0), density),
        _sample_rect(rng, np.zeros(3), np.array([0, height, 0.]), np.array([0, 0, depth]), (1, 0, 0), density),
        _sample_rect(rng, np.zeros(3), np.array([width, 0, 0.]), np.array([0, height, 0.]), (0, 0, 1), density),
    ]
    return parts


def make_scene(seed=7, width=3.2, depth=3.2, height=1.2, density=DENSITY, objects=("table", "chair", "chair"),
               timestep=0, jitter=JITTER, shuffle=True):
    """A room with movable objects.  Returns a dict with the scan (points/normals/class/instance)
    and, per object, its local-frame cloud and its pose in this timestep.

    Object local clouds depend only on (seed, object number); poses depend on the timestep, so
    timestep 1 is 'the same room after somebody moved the furniture'."""
    rng_shell = np.random.default_rng([seed, 1000 + timestep])
    rng_pose = np.random.default_rng([seed, 2000 + timestep])
    floor, wall_x, wall_z = make_room_shell(rng_shell, width, depth, height, density)
    P = [floor[0], wall_x[0], wall_z[0]]
    N = [floor[1], wall_x[1], wall_z[1]]
    cls = [np.full(len(floor[0]), CLASS_IDX["floor"]), np.full(len(wall_x[0]), CLASS_IDX["wall"]),
           np.full(len(wall_z[0]), CLASS_IDX["wall"])]
    inst = [np.full(len(floor[0]), 0), np.full(len(wall_x[0]), 1), np.full(len(wall_z[0]), 2)]
    objs = []
    n_obj = len(objects)
    # objects sit on a jittered lattice so they never overlap
    cols = int(np.ceil(np.sqrt(n_obj)))
    for k, kind in enumerate(objects):
        lp, ln = make_object(kind, seed * 7919 + k, density, 1.0, 0.0)
        cx = (k % cols + 0.5) / cols * (width - 1.4) + 0.7
        cz = (k // cols + 0.5) / max(1, int(np.ceil(n_obj / cols))) * (depth - 1.4) + 0.7
        ang = rng_pose.uniform(0, 2 * np.pi)
        tx = cx + rng_pose.uniform(-0.15, 0.15)
        tz = cz + rng_pose.uniform(-0.15, 0.15)
        pose = pose_matrix(ang, (tx, 0.0, tz))
        # the scan sees a fresh sampling of the same surfaces (not the model's own points)
        sp, sn = make_object(kind, seed * 104729 + 31 * k + 977 * timestep, density, 1.0, 0.0)
        P.append(apply_pose(pose, sp).astype(np.float64)); N.append(apply_pose(pose, sn, False).astype(np.float64))
        cls.append(np.full(len(sp), CLASS_IDX[kind])); inst.append(np.full(len(sp), 3 + k))
        objs.append(dict(kind=kind, class_idx=CLASS_IDX[kind], uidx=3 + k, pos=lp, nor=ln, pose=pose,
                         is_static=0))
    P = np.concatenate(P); N = np.concatenate(N)
    cls = np.concatenate(cls).astype(np.int32); inst = np.concatenate(inst).astype(np.int32)
    rng_j = np.random.default_rng([seed, 3000 + timestep])
    P = P + rng_j.normal(0.0, jitter, P.shape)
    N = noisy_normals(rng_j, N)
    if shuffle:
        perm = rng_j.permutation(len(P))
        P, N, cls, inst = P[perm], N[perm], cls[perm], inst[perm]
    return dict(points=np.ascontiguousarray(P, np.float32), normals=np.ascontiguousarray(N, np.float32),
                class_idx=cls, instance_idx=inst, objects=objs, dims=(width, depth, height))


def scene_for_point_count(n_target, seed=11, timestep=0, density=DENSITY, n_objects=None):
    """Scale the room so the scan holds about n_target points (SURVEY §8d config 2:
    1 M points ~ 156 m² at 6400 pts/m²)."""
    area = n_target / density
    height = 2.4 if area > 40 else 1.2
    # floor w*d + walls (w+d)*h = area, with w = 1.25 d
    # 1.25 d² + 2.25 d h - area = 0
    d = (-2.25 * height + np.sqrt((2.25 * height) ** 2 + 5.0 * area)) / 2.5
    w = 1.25 * d
    if n_objects is None:
        n_objects = max(3, int(area / 12))
    kinds = ["table", "chair", "chair", "shelf"]
    objects = tuple(kinds[i % len(kinds)] for i in range(n_objects))
    return make_scene(seed=seed, width=float(w), depth=float(d), height=height, density=density,
                      objects=objects, timestep=timestep)


def perturbed_pose(pose16, rng, max_angle=0.08, max_shift=0.04):
    """pose * small random rigid motion (for ICP start poses)."""
    ang = rng.uniform(-max_angle, max_angle)
    t = rng.uniform(-max_shift, max_shift, 3); t[1] *= 0.25
    d = np.eye(4); d[:3, :3] = rot_y(ang); d[:3, 3] = t
    m = np.asarray(pose16, np.float64).reshape(4, 4).T @ d
    return np.ascontiguousarray(m.T.astype(np.float32).ravel())


def write_ply(path, s):
    """A scene as the binary-LE PLY the reference's seg2rsdb / pose_proposal read (positions, normals, colour, radius,
    class_idx, instance_idx per vertex; no faces, so points are used as they are: lib/rs/rs_pointcloud.h:1268-1281)."""
    n = len(s["points"])
    hdr = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\n"
           "property float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
           "property uchar red\nproperty uchar green\nproperty uchar blue\nproperty float radius\n"
           "property int class_idx\nproperty int instance_idx\nend_header\n") % n
    dt = np.dtype([("p", "<f4", 3), ("n", "<f4", 3), ("c", "u1", 3), ("r", "<f4"), ("cls", "<i4"), ("inst", "<i4")])
    a = np.zeros(n, dt)
    a["p"] = s["points"]; a["n"] = s["normals"]; a["c"] = 128; a["r"] = 0.01
    a["cls"] = s["class_idx"]; a["inst"] = s["instance_idx"]
    with open(path, "wb") as f:
        f.write(hdr.encode()); f.write(a.tobytes())


def write_class_table(path):
    """The class file of the pipeline ('nyu40_classes.txt': rsdb syntax, lib/rs/rs_database.h:301-316)."""
    with open(path, "w") as f:
        f.write("rsdb 0.1\n")
        for k, v in CLASS_IDX.items():
            f.write(f"class {k} {v}\n")
