#!/bin/bash
# Profiling recipes used for profiles/ (run on the GPU box through gpurun):
#   bash tools/profile.sh stats     -> rocprofv3 --kernel-trace --stats of the bench command
#   bash tools/profile.sh pmc       -> SQ instruction / wait counters per kernel
#   bash tools/profile.sh trace     -> per-dispatch durations of ONE serial step of the timed workload (marker kernels delimit the steps), whole
#   bash tools/profile.sh trace_dropin -> the drop-in block's calls, kernel by kernel (its own file)
#   bash tools/profile.sh stats_serial -> kernel stats of the serial bench (every kernel alone on the chip), beside `stats` (three streams)
#   bash tools/profile.sh traffic   -> FETCH_SIZE and WRITE_SIZE in separate passes -> gpurun_out/pmc_traffic_raw.json
#                                      (BENCH_ARGS="--knn brute --points 100000" for the other layout / size)
# rocprofv3 is given the program itself after `--` (python ...), never a shell or env wrapper.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
case "$1" in
stats)
  rm -rf gpurun_out/prof_stats
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_stats_bench.json 2> gpurun_out/prof_stats.err
  head -14 "$(find gpurun_out/prof_stats -name '*kernel_stats.csv' | head -1)" | cut -c1-150 ;;
pmc)
  rm -rf gpurun_out/prof_pmc
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/prof_pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --serial > /dev/null 2> gpurun_out/prof_pmc.err
  python tools/pmc_summary.py "$(find gpurun_out/prof_pmc -name '*counter_collection.csv' | head -1)" ;;
traffic)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --serial $BENCH_ARGS > /dev/null 2> gpurun_out/pmc_$c.err
  done
  python tools/pmc_summary.py --traffic ;;
trace)   # per-dispatch timeline of ONE serial step of the timed workload, whole: which ICP iteration costs what
  rm -rf gpurun_out/prof_trace
  export RS_HIP_PROF_EVERY=1000 RS_BENCH_MARK=1
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --serial > /dev/null 2> gpurun_out/prof_trace.err
  python tools/pmc_summary.py --trace "$(find gpurun_out/prof_trace -name '*kernel_trace.csv' | head -1)" gpurun_out/trace_last_step.txt ;;
trace_dropin)   # the same three consumers through librescan_dropin.so (host arrays in, results out): first call and a repeated one, kernel by kernel
  rm -rf gpurun_out/prof_trace_dropin
  export RS_HIP_PROF_EVERY=1000 RS_BENCH_MARK=1      # unit 0: the one timed (serial) step; units 1..4: the drop-in block's first call and its three repeated ones
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace_dropin -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --serial > gpurun_out/prof_trace_dropin.log 2> gpurun_out/prof_trace_dropin.err
  python tools/pmc_summary.py --trace "$(find gpurun_out/prof_trace_dropin -name '*kernel_trace.csv' | head -1)" gpurun_out/trace_dropin_calls.txt all ;;
stats_serial)
  rm -rf gpurun_out/prof_stats_serial
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats_serial -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --serial > gpurun_out/prof_stats_serial_bench.json 2> gpurun_out/prof_stats_serial.err
  head -14 "$(find gpurun_out/prof_stats_serial -name '*kernel_stats.csv' | head -1)" | cut -c1-150 ;;
*) echo "usage: $0 stats|pmc|traffic|trace" ;;
esac
