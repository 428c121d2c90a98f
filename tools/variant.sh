#!/bin/bash
# Kernel A/B experiments: build librescan_hip_<tag>.so with extra -D flags and select it at run time with
# RS_HIP_LIB=<path> (rescan_amd/capi.py).   usage: tools/variant.sh <tag> [-DNAME=VALUE ...]
set -e
cd "$(dirname "$0")/../rescan_amd"
tag=$1; shift
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function"
objs=""
for tu in rs_icp_search rs_icp_estimate rs_score rs_rows rs_api; do
  /opt/rocm/bin/hipcc $F "$@" -c csrc/$tu.hip -o /tmp/${tu}_$tag.o &
  objs="$objs /tmp/${tu}_$tag.o"
done
wait
[ -f csrc/rs_build.o ] || /opt/rocm/bin/hipcc $F -c csrc/rs_build.hip -o csrc/rs_build.o      # (no experiment flags in the index build; hipCUB: ~20 s)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs csrc/rs_build.o -o librescan_hip_$tag.so
echo "$(pwd)/librescan_hip_$tag.so"
