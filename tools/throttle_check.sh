#!/bin/bash
# Is the periodic slow step (about one in 35, every ~100 ms) the container's CPU quota?  cgroup throttling counters around a long bench run.
cd "$(dirname "$0")/.."
for f in /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us; do [ -f $f ] && echo "$f: $(cat $f)"; done
stat() { cat /sys/fs/cgroup/cpu.stat 2>/dev/null || cat /sys/fs/cgroup/cpu/cpu.stat 2>/dev/null; }
echo "nproc $(nproc)"; echo "--- before"; stat
RS_BENCH_PRINT_STEPS=1 python bench.py --steps 200 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep "slow steps"
echo "--- after"; stat
