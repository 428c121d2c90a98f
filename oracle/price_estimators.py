"""TEST INFRASTRUCTURE (CPU, build container).  Prices the estimator choices for icp_estimate_rigid_xform_pt2pl
(lib/rs/icp.h:210-298) against the bar: for every fixture whose reference result is committed — the nine icp_* object fixtures,
the eight `strong_icp_*` refine units of bench_seed11.npz, the 24 scan-sized sweep runs — the pose distance (Frobenius) from the
REFERENCE's pose and the iteration count of oracle/rs_oracle.c: orc_icp_iterate_variant in

   mode 0  the reference's arithmetic (must reproduce the fixture bit for bit: asserted)
   mode 1  reference cut + reference centroid chains, normal equations summed exactly   ("(ii)" of VERDICT r05 item 2)
   mode 2  as 1, the 2.5 sigma cut from exactly summed statistics                       (what the GPU's integer statistics give)
   mode 3  as 2, centroids from fp64 sums too                                          ("(iii)": nothing of the reference's drift left)

Usage: python oracle/price_estimators.py [--no-sweep] [--no-strong]  ->  profiles/r06/estimator_policy_cpu.txt"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from oracle.pyoracle import Oracle, f32p  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
I4 = np.eye(4, dtype=np.float32).ravel()
MODES = (0, 1, 2, 3)


def variant(O, p1, n1, p2, n2, T0, T2, md, ma, n_iters, stop, mode, errs=False):
    f = O.lib.orc_icp_iterate_variant
    f.restype = C.c_float
    f.argtypes = [f32p, f32p, C.c_int32, f32p, f32p, C.c_int32, f32p, f32p, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_int32,
                  C.POINTER(C.c_int32), f32p]
    c = lambda a: np.ascontiguousarray(a, np.float32)  # noqa: E731
    T = c(T0).ravel().copy(); it = C.c_int32()
    eo = np.zeros(max(1, int(n_iters)), np.float32)
    e = f(c(p1), c(n1), len(p1), c(p2), c(n2), len(p2), T, c(T2).ravel(), float(md), float(ma), int(n_iters), int(stop), int(mode), C.byref(it), eo)
    if errs:
        return np.float32(e), T, it.value, eo[:it.value]
    return np.float32(e), T, it.value


def main():
    O = Oracle()
    out = []
    say = lambda s: (print(s, flush=True), out.append(s))  # noqa: E731
    say("# pose distance (Frobenius) from the reference's pose | iterations, per estimator variant (oracle/price_estimators.py)")
    say("# case, source points, reference iterations | mode 0 | mode 1 (ref cut + ref centroid chains + exact moments) | mode 2 (exact cut too) | mode 3 (fp64 centroids too)")
    worst = {m: 0.0 for m in MODES}; itdiff = {m: 0 for m in MODES}; n_cases = 0

    def run(name, p1, n1, p2, n2, T0, md, ma, n_iters, stop, T_ref, it_ref):
        nonlocal n_cases
        cells = []
        for m in MODES:
            e, T, it = variant(O, p1, n1, p2, n2, T0, I4, md, ma, n_iters, stop, m)
            d = float(np.linalg.norm(T.astype(np.float64) - np.asarray(T_ref, np.float64).ravel()))
            if m == 0:
                assert d == 0.0 and (it_ref is None or it == it_ref), f"{name}: mode 0 is not the reference ({d}, {it} vs {it_ref})"
            worst[m] = max(worst[m], d); itdiff[m] += int(it_ref is not None and it != it_ref)
            cells.append(f"{d:.2e} ({it} it)")
        n_cases += 1
        say(f"{name:28s} n {len(p1):7d} ref {it_ref if it_ref is not None else n_iters:3d} it | " + " | ".join(cells))

    d = dict(np.load(os.path.join(GOLDEN, "scene.npz")))
    objs = [(d[f"obj{i}_pos"], d[f"obj{i}_nor"]) for i in range(int(d["n_obj"]))]
    for fn in sorted(f for f in os.listdir(GOLDEN) if f.startswith("icp_")):
        g = dict(np.load(os.path.join(GOLDEN, fn)))
        p, n = objs[int(g["obj"])]
        run(fn[:-4], p, n, d["points"], d["normals"], g["T1"], float(g["max_dist"]), float(g["max_angle"]), 100, 1, g["T_out"], int(g["iters"]))
    if "--no-strong" not in sys.argv:
        import bench
        g = dict(np.load(os.path.join(GOLDEN, "bench_seed11.npz")))
        t = time.time()
        w = bench.build_inputs(1_000_000, seed=11)
        say(f"# bench inputs generated in {time.time() - t:.1f} s")
        si = w["strong_icp"]
        for k, p in enumerate(w["plc"][:bench.N_PLACEMENTS]):
            run(f"strong_icp_{k}", p["np"][0], p["np"][1], w["s1"]["points"], w["s1"]["normals"], si["T0s"][k], float(si["max_dist"]), float(si["max_angle"]),
                bench.ICP_ITERS, 0, g["strong_icp_pose"][k], None)
    if "--no-sweep" not in sys.argv:
        from gen_golden_bench import sweep_inputs
        g = dict(np.load(os.path.join(GOLDEN, "sweep_icp.npz")))
        for k, seed in enumerate(g["seeds"]):
            s0, s1, T0, md, ma = sweep_inputs(int(seed))
            run(f"sweep_{int(seed):02d}", s1["points"], s1["normals"], s0["points"], s0["normals"], T0, float(md), float(ma), 100, 1, g["pose"][k], int(g["iters"][k]))
    say(f"# {n_cases} cases; worst pose distance per mode: " + ", ".join(f"mode {m}: {worst[m]:.2e}" for m in MODES))
    say("# cases whose iteration count differs from the reference's: " + ", ".join(f"mode {m}: {itdiff[m]}" for m in MODES))
    os.makedirs(os.path.join(ROOT, "profiles", "r06"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r06", "estimator_policy_cpu.txt"), "w") as f:
        f.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
