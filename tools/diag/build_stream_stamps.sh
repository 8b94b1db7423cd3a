#!/bin/bash
# Diagnostic build of the library (never the product) with per-phase s_memtime stamps in fused_stream_kernel:
# tools/diag/lib/stream_stamps/libcalib_lm.so, picked up through CALIB_LM_LIBRARY (tools/diag/stream_stamps.py).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/tools/diag/lib/stream_stamps
cd $R/camera-calibration_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wno-unused-function -DCALIB_STREAM_STAMPS \
    -o $R/tools/diag/lib/stream_stamps/libcalib_lm.so calib_lm.hip
