#!/bin/bash
# kernel time of k_icp_faithful by source size (rocprofv3 kernel trace of tools/faithful_timing.py)
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = default ]; then unset RS_HIP_LIB; else export RS_HIP_LIB=$GRAFT_REPO_ROOT/rescan_amd/librescan_hip_$lib.so; fi
  rm -rf /tmp/ft_$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ft_$lib -- python3 $GRAFT_REPO_ROOT/tools/faithful_timing.py 8000 > /tmp/ft_$lib.log 2>&1
  echo "== $lib"; grep "^n " /tmp/ft_$lib.log
  f=$(find /tmp/ft_$lib -name "*kernel_stats.csv" | head -1)
  grep -E "faith|Name" "$f" | cut -c1-200
done
