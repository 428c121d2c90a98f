#!/bin/bash
# The round's committed profile set (run on the GPU box through gpurun): bench lines, rocprofv3 kernel stats of the bench command (three
# streams) AND of the serial bench (every kernel alone on the chip), SQ instruction counts, FETCH / WRITE traffic in separate PMC
# passes (folded per bench domain into profiles/pmc_traffic.json and profiles/pmc_instructions.json, stamped with the kernel
# sources' digest), ONE serial step of the timed workload kernel by kernel (marker kernels delimit the steps), the drop-in block's
# calls kernel by kernel, the score batch alone -> gpurun_out/r06/, to be copied into profiles/r06/.
set -e
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
R=gpurun_out/r06; mkdir -p $R profiles
python bench.py > $R/bench_final.json 2> $R/bench_final.err
python bench.py --serial --no-cpu-baseline > $R/bench_serial.json 2>> $R/bench_final.err
bash tools/profile.sh stats > $R/stats_head.txt 2>&1
cp "$(find gpurun_out/prof_stats -name '*kernel_stats.csv' | head -1)" $R/kernel_stats_bench.csv
cp gpurun_out/prof_stats_bench.json $R/bench_under_rocprof.json
bash tools/profile.sh stats_serial > $R/stats_serial_head.txt 2>&1
cp "$(find gpurun_out/prof_stats_serial -name '*kernel_stats.csv' | head -1)" $R/kernel_stats_serial.csv
bash tools/profile.sh pmc > $R/pmc_instruction_counts.txt 2>&1
bash tools/profile.sh traffic > $R/pmc_traffic_per_kernel.txt 2>&1
python tools/pmc_summary.py --fold-domains > $R/pmc_fold_domains.txt 2>&1
cp profiles/pmc_traffic.json $R/pmc_traffic.json; cp profiles/pmc_instructions.json $R/pmc_instructions.json
bash tools/profile.sh trace > /dev/null 2>&1; cp gpurun_out/trace_last_step.txt $R/trace_one_serial_step.txt
bash tools/profile.sh trace_dropin > /dev/null 2>&1; cp gpurun_out/trace_dropin_calls.txt $R/trace_dropin_calls.txt
bash tools/score_scene_profile.sh > /dev/null 2>&1; cp gpurun_out/sq_profile.txt $R/score_scene_profile.txt
python bench.py > $R/bench_final2.json 2>> $R/bench_final.err
cat $R/pmc_fold_domains.txt; tail -3 $R/trace_one_serial_step.txt; tail -c 600 $R/bench_final2.json
