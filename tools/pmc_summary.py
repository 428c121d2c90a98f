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


DOMAINS = ("nn_icp", "icp_moments", "nn_score", "nn_label")


def per_domain(path, counters=None):
    """Totals per bench domain of every counter in a rocprofv3 --pmc CSV, and the number of launch UNITS of each domain, taken dispatch
    by dispatch in launch order: a score batch is everything from k_score_keys to k_score_gather (its radix-sort passes and fills are
    rocprim / runtime kernels without our name: only their place in the sequence says whose they are) or one k_score< / k_score_coop /
    k_score_final group; an ICP search is k_icp_corr< + the k_icp_corr_coop< behind it; the estimator is one k_icp_update*; a label
    pass one k_label( with its gather to input order.  -> {domain: {"units": n, counter: total}}"""
    rows = {}
    for r in csv.DictReader(open(path)):
        d = rows.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "c": {}})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out = {k: collections.defaultdict(float) for k in DOMAINS}
    in_score = False
    for _, d in sorted(rows.items()):
        n = d["name"]
        dom = None
        if "k_score_keys" in n:
            in_score = True
        if in_score or "k_score" in n:
            dom = "nn_score"
            if "k_score_gather" in n or "k_score_final" in n:
                out[dom]["units"] += 1
                in_score = False
        elif "k_icp_corr" in n:
            dom = "nn_icp"
            if "coop" not in n:
                out[dom]["units"] += 1
        elif any(x in n for x in ("k_icp_moments", "k_icp_update", "k_chain_", "k_replay_", "k_icp_faith", "k_lane_")):
            dom = "icp_moments"
            if "k_icp_update" in n or "k_icp_faithful" in n or "k_replay_finish" in n or "k_lane_walk_and_update" in n:
                out[dom]["units"] += 1
        elif "k_label" in n:
            dom = "nn_label"
            if "rs::k_label(" in n:
                out[dom]["units"] += 1
        if dom:
            for c, v in d["c"].items():
                if counters is None or c in counters:
                    out[dom][c] += v
    return {k: dict(v) for k, v in out.items()}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--fold-domains":
        # the three PMC passes of tools/profile.sh (pmc, traffic) -> profiles/pmc_traffic.json + profiles/pmc_instructions.json, per launch
        # unit of each bench domain, stamped with the kernel sources' digest (bench.py refuses files made for other sources)
        import os
        sys.path.insert(0, os.getcwd())
        from rescan_amd.build import sources_sha
        sha = sources_sha()
        F = per_domain(glob.glob("gpurun_out/pmc_FETCH_SIZE/*/*counter_collection.csv")[0])
        Wr = per_domain(glob.glob("gpurun_out/pmc_WRITE_SIZE/*/*counter_collection.csv")[0])
        tr = {"_note": "HBM-side bytes per launch unit from rocprofv3 PMC (separate FETCH_SIZE and WRITE_SIZE passes of `bench.py --steps 2 --warmup 1 "
                       "--serial`, tools/profile.sh traffic, folded by tools/pmc_summary.py --fold-domains), FETCH_SIZE (KB) doubled as MI355X_MICROARCH.md "
                       "\u00a7HBM prescribes for gfx950; WRITE_SIZE as reported.  Units: nn_icp = k_icp_corr<cold|warm> + its k_icp_corr_coop; icp_moments = one "
                       "estimator step (k_chain_* + k_icp_update_wide); nn_score = one batch: k_score_keys + the radix sort's passes + k_score_scene + "
                       "k_score_gather; nn_label = k_label + the gather to input order.  Atomics execute at the memory side and count 64 B each.",
              "kernels_sha": sha}
        for d in DOMAINS:
            u = max(1.0, F[d].get("units", 0))
            tr[d] = (2.0 * F[d].get("FETCH_SIZE", 0.0) + Wr[d].get("WRITE_SIZE", 0.0)) * 1024.0 / u
            tr[d + "_fetch_x2_write_MB"] = [round(2.0 * F[d].get("FETCH_SIZE", 0.0) * 1024 / u / 1e6, 2), round(Wr[d].get("WRITE_SIZE", 0.0) * 1024 / u / 1e6, 2)]
        json.dump(tr, open("profiles/pmc_traffic.json", "w"), indent=1)
        print("traffic, MB per launch unit:", {d: round(tr[d] / 1e6, 1) for d in DOMAINS})
        ipath = glob.glob("gpurun_out/prof_pmc/*/*counter_collection.csv")
        if ipath:
            I = per_domain(ipath[0])
            ins = {"_note": "wave-level instruction counts per launch unit (SQ_INSTS_VALU / SALU / LDS, SQ_WAVES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAIT_*), one "
                            "rocprofv3 --pmc pass of `bench.py --steps 1 --warmup 0 --serial` (tools/profile.sh pmc), units as in pmc_traffic.json",
                   "kernels_sha": sha}
            for d in DOMAINS:
                u = max(1.0, I[d].get("units", 0))
                ins[d] = {c: round(v / u) for c, v in I[d].items() if c != "units"}
                ins[d]["units_in_the_pass"] = int(I[d].get("units", 0))
            json.dump(ins, open("profiles/pmc_instructions.json", "w"), indent=1)
            print("instructions, millions per launch unit:", {d: {c: round(v / 1e6, 1) for c, v in ins[d].items() if c.startswith("SQ_INSTS")} for d in DOMAINS})
        sys.exit(0)
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
        doms = {"nn_icp": (["k_icp_corr"], ["k_icp_corr<"]), "icp_moments": (["k_icp_moments", "k_icp_update", "k_chain_", "k_lane_"], ["k_icp_update", "k_lane_walk_and_update"]),       # the estimator: one k_icp_update per iteration (round 3: + the centroid chains' kernels)
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
        # A rocprofv3 --kernel-trace of `RS_BENCH_MARK=1 python bench.py --serial --no-cpu-baseline --no-extras ...`: every step of the
        # run starts with the library's marker kernel (rs::k_step_marker) and nothing follows the last timed step, so the rows from
        # the LAST marker to the end are exactly one step of the timed workload (round 5 took "the last third of the rows", which
        # landed in the drop-in block that used to follow the timed steps).  Written whole.  `--trace <csv> <out> [all]`: every
        # marked step, separated (the drop-in trace: tools/profile.sh trace_dropin).
        rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
        # (every kernel of the process: the library's rs::, the radix sort's rocprim:: passes, the runtime's fills and copies; the marker
        #  sits in an extern "C" block, so its name carries no namespace)
        marks = [k for k, r in enumerate(rows) if "k_step_marker" in r["Kernel_Name"]]
        out_path = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/trace_last_step.txt"
        every = len(sys.argv) > 4 and sys.argv[4] == "all"
        if not marks:
            sys.exit("no rs::k_step_marker in the trace: run bench.py with RS_BENCH_MARK=1")
        spans = [(marks[i] + 1, marks[i + 1] if i + 1 < len(marks) else len(rows)) for i in range(len(marks))]
        lines = []
        for n, (a, b) in enumerate(spans if every else spans[-1:]):
            if every:
                lines.append(f"# marked unit {n}")
            t0 = int(rows[a]["Start_Timestamp"]) if a < b else 0
            for r in rows[a:b]:
                s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                lines.append(f'{(s_ - t0) / 1e3:9.1f} us  +{(e_ - s_) / 1e3:8.1f} us  {r["Kernel_Name"].split("(")[0].replace("void ", "")[:90]}')
            if a < b:
                lines.append(f"# {b - a} kernels, first start to last end {(int(rows[b - 1]['End_Timestamp']) - t0) / 1e3:.1f} us")
        open(out_path, "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))
    else:
        for k, d in per_kernel(sys.argv[1]).items():
            if k.startswith("rs::"):
                print(k, {c: round(v[0] / 1e6, 2) for c, v in d.items()}, "(millions per launch)")
