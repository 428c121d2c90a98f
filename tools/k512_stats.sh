#!/bin/bash
# rocprofv3 kernel stats of bench.py --scaling strong --strong-problems 512 (the many-refines batch): where its 43 ms go
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
rm -rf gpurun_out/prof_k512
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k512 -- python bench.py --scaling strong --strong-problems 512 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --serial > gpurun_out/prof_k512_bench.json 2> gpurun_out/prof_k512.err
head -16 "$(find gpurun_out/prof_k512 -name '*kernel_stats.csv' | head -1)" | cut -c1-170
