#!/bin/bash
# A/B builds of the library (never the product): bash tools/diag/build_variant.sh <name> <hipcc flags...>
# -> tools/diag/lib/<name>/libcalib_lm.so, picked up through CALIB_LM_LIBRARY (tools/ab_variant.sh).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
mkdir -p $R/tools/diag/lib/$name
cd $R/camera-calibration_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wno-unused-function "$@" \
    -o $R/tools/diag/lib/$name/libcalib_lm.so calib_lm.hip
