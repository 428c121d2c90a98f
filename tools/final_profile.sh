#!/bin/bash
# The round's committed profile set (run on the GPU box through gpurun): bench lines, rocprofv3 kernel stats of the bench command, SQ
# instruction counts, FETCH / WRITE traffic in separate PMC passes, one serial step kernel by kernel -> gpurun_out/r04/, to be copied
# into profiles/r04/ (and folded: python tools/pmc_summary.py --fold).
set -e
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out/r04
python bench.py > gpurun_out/r04/bench_final.json 2> gpurun_out/r04/bench_final.err
python bench.py --serial --no-cpu-baseline > gpurun_out/r04/bench_serial.json 2>> gpurun_out/r04/bench_final.err
bash tools/profile.sh stats > gpurun_out/r04/stats_head.txt 2>&1
cp "$(find gpurun_out/prof_stats -name '*kernel_stats.csv' | head -1)" gpurun_out/r04/kernel_stats_bench.csv
cp gpurun_out/prof_stats_bench.json gpurun_out/r04/bench_under_rocprof.json
bash tools/profile.sh pmc > gpurun_out/r04/pmc_instruction_counts.txt 2>&1
bash tools/profile.sh traffic > gpurun_out/r04/pmc_traffic_per_kernel.txt 2>&1
cp gpurun_out/pmc_traffic_raw.json gpurun_out/r04/pmc_traffic_raw.json
bash tools/profile.sh trace > gpurun_out/r04/trace_one_serial_step.txt 2>&1
python bench.py > gpurun_out/r04/bench_final2.json 2>> gpurun_out/r04/bench_final.err
