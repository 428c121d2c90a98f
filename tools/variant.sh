#!/bin/bash
# Kernel A/B experiments: build librescan_hip_<tag>.so with extra -D flags and select it at run time with
# RS_HIP_LIB=<path> (rescan_amd/capi.py).   usage: tools/variant.sh <tag> [-DNAME=VALUE ...]
set -e
cd "$(dirname "$0")/../rescan_amd"
tag=$1; shift
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c csrc/rs_kernels.hip -o /tmp/rs_kernels_$tag.o
/opt/rocm/bin/hipcc $F "$@" -c csrc/rs_api.hip -o /tmp/rs_api_$tag.o
[ -f csrc/rs_build.o ] || /opt/rocm/bin/hipcc $F -c csrc/rs_build.hip -o csrc/rs_build.o      # (no experiment flags in the index build; hipCUB: ~20 s)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/rs_kernels_$tag.o csrc/rs_build.o /tmp/rs_api_$tag.o -o librescan_hip_$tag.so
echo "$(pwd)/librescan_hip_$tag.so"
