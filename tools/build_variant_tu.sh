#!/bin/bash
# bash tools/build_variant_tu.sh <name> <tu> <extra hipcc flags...>: rescan_amd/librescan_hip_<name>.so with csrc/<tu>.hip compiled under the extra flags
cd "$(dirname "$0")/.."
name=$1; tu=$2; shift; shift
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c rescan_amd/csrc/$tu.hip -o /tmp/${tu}_$name.o || exit 1
objs=""; for t in rs_icp_search rs_icp_estimate rs_score rs_rows rs_build rs_api; do if [ $t = $tu ]; then objs="$objs /tmp/${tu}_$name.o"; else objs="$objs rescan_amd/csrc/$t.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o rescan_amd/librescan_hip_$name.so && echo rescan_amd/librescan_hip_$name.so
