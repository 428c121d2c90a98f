#!/bin/bash
# Is the periodic slow step (about one in 35, every ~100 ms) the container's CPU quota?  cgroup throttling counters around the timed region
# (the bench line's "host" block) next to the slow steps and what the three consumers' calls took in them.
cd "$(dirname "$0")/.."
for f in /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us; do [ -f $f ] && echo "$f: $(cat $f)"; done
echo "nproc $(nproc)"
for i in 1 2 3; do
  RS_BENCH_PRINT_STEPS=1 python bench.py --steps ${STEPS:-200} --warmup 5 --no-cpu-baseline 2> gpurun_out/throttle_check.err > gpurun_out/throttle_check.json
  grep -A1 "slow steps" gpurun_out/throttle_check.err
  python -c "import json; d=json.load(open('gpurun_out/throttle_check.json')); print('  mean', round(d['ms_per_step'],3), d['ms_per_step_spread'], d['host'])"
done
