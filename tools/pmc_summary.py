"""Summaries of rocprofv3 --pmc CSVs (per kernel, averaged over launches)."""
import collections
import csv
import glob
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
    return {k: {c: (v / cnt[k][c], cnt[k][c]) for c, v in d.items()} for k, d in agg.items()}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--traffic":
        out = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            f = glob.glob(f"gpurun_out/pmc_{c}/*/*counter_collection.csv")[0]
            out[c] = {k: v[c] for k, v in per_kernel(f).items() if c in v}
        import os
        sys.path.insert(0, os.getcwd())
        from rescan_amd.build import sources_sha
        out["kernels_sha"] = sources_sha()          # the sources of the library these counters were collected with
        json.dump(out, open("gpurun_out/pmc_traffic_raw.json", "w"), indent=1)
        for k in sorted(out["FETCH_SIZE"]):
            print(k, "FETCH_SIZE avg KB", round(out["FETCH_SIZE"][k][0], 1),
                  "WRITE_SIZE avg KB", round(out["WRITE_SIZE"].get(k, (0, 0))[0], 1))
    elif len(sys.argv) > 1 and sys.argv[1] == "--fold":
        # gpurun_out/pmc_traffic_raw.json -> profiles/pmc_traffic.json (what bench.py reports as roofline.traffic):
        # HBM-side bytes per launch of each bench domain = sum over its kernels of (2 x FETCH_SIZE + WRITE_SIZE) KB x calls,
        # divided by the domain's launches (one ICP search = k_icp_corr + whichever cooperative kernel followed it).
        raw = json.load(open("gpurun_out/pmc_traffic_raw.json"))
        F, W = raw["FETCH_SIZE"], raw["WRITE_SIZE"]
        def total(names):
            return sum((2 * F[k][0] + W.get(k, (0, 0))[0]) * F[k][1] for k in F if any(n in k for n in names)) * 1024
        # domain -> (kernels that belong to it, kernels whose launches count as ONE unit of it)
        doms = {"nn_icp": (["k_icp_corr"], ["k_icp_corr<"]), "icp_moments": (["k_icp_moments", "k_icp_update", "k_chain_"], ["k_icp_update"]),       # the estimator: one k_icp_update per iteration (round 3: + the centroid chains' kernels)
                "nn_score": (["k_score"], ["rs::k_score<"]), "nn_label": (["k_label"], ["rs::k_label("])}
        out = {"_note": "HBM-side bytes per launch from rocprofv3 PMC (separate FETCH_SIZE and WRITE_SIZE passes of `bench.py --steps 2 "
                        "--warmup 1 --serial`, tools/profile.sh traffic, folded by tools/pmc_summary.py --fold), FETCH_SIZE doubled as "
                        "MI355X_MICROARCH.md \u00a7HBM prescribes for 16-B-per-lane reads on gfx950; WRITE_SIZE as reported.  nn_icp = "
                        "k_icp_corr<cold|warm> + k_icp_corr_coop<4|8> (one search); nn_label = k_label + the gather to input order.  "
                        "Atomics (statistics, queues, the profiling counters) execute at the memory side and count 64 B each."}
        def launches(keys):
            return sum(F[k][1] for k in F if any(q in (k + "(") for q in keys) and "coop" not in k)
        for d, (names, per) in doms.items():
            out[d] = total(names) / max(1, launches(per))
        out["kernels_sha"] = raw.get("kernels_sha")
        out["raw_avg_KB"] = {c: {k: v[0] for k, v in raw[c].items()} for c in ("FETCH_SIZE", "WRITE_SIZE")}
        json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
        print({d: round(out[d] / 1e6, 1) for d in doms}, "MB per launch")
    elif len(sys.argv) > 2 and sys.argv[1] == "--trace":
        rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
        rows = [r for r in rows if "rs::" in r["Kernel_Name"]]
        # bench.py --steps 2 --warmup 1 --serial with RS_HIP_PROF_EVERY=1000: cloud construction, then three identical steps; show the
        # LAST one — the first timed step carries the live profile's event records between its kernels (a ~6 us bubble each: what
        # round 1's trace showed as "launch gaps"), the later ones do not
        last_build = max([k for k, r in enumerate(rows) if "k_build_" in r["Kernel_Name"]], default=-1)
        rows = rows[last_build + 1:]
        n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
        start = len(rows) * (n_steps - 1) // n_steps
        lines = []
        for r in rows[start:]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            lines.append(f'{(s - int(rows[start]["Start_Timestamp"])) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {r["Kernel_Name"].split("(")[0].replace("void ", "")}')
        open("gpurun_out/trace_last_step.txt", "w").write("\n".join(lines) + "\n")
        print("\n".join(lines[:80]))
    else:
        for k, d in per_kernel(sys.argv[1]).items():
            if k.startswith("rs::"):
                print(k, {c: round(v[0] / 1e6, 2) for c, v in d.items()}, "(millions per launch)")
