#!/bin/bash
# bash tools/build_variant.sh <name> <extra hipcc flags...>: rescan_amd/librescan_hip_<name>.so with rs_icp_search.hip compiled under the extra flags
# (for tools/ab_lib.sh: an A/B of one compile-time choice against the shipped library)
cd "$(dirname "$0")/.."
name=$1; shift
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c rescan_amd/csrc/rs_icp_search.hip -o /tmp/rs_icp_search_$name.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/rs_icp_search_$name.o rescan_amd/csrc/rs_icp_estimate.o rescan_amd/csrc/rs_score.o rescan_amd/csrc/rs_rows.o rescan_amd/csrc/rs_build.o rescan_amd/csrc/rs_api.o -o rescan_amd/librescan_hip_$name.so && echo rescan_amd/librescan_hip_$name.so
