#!/bin/bash
# build an experimental variant of the library: tools/build_variant.sh <name> [-DFLAG ...]
# -> zvec_amd/_variants/libzvec_hip_<name>.so ; select it with ZVEC_HIP_LIBRARY=<path>
set -e
cd "$(dirname "$0")/.."
mkdir -p zvec_amd/_variants
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DZVEC_HIP_TUNING "$@" -o zvec_amd/_variants/libzvec_hip_$name.so zvec_amd/csrc/zvec_hip_api.hip
echo zvec_amd/_variants/libzvec_hip_$name.so
