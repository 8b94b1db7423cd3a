#!/bin/bash
# Diagnostic builds of the library (never the product): the fused kernel with per-phase s_memtime stamps, and with
# parts compiled out (in-situ ablation). The patch is applied to a scratch copy of csrc/, the product sources stay
# as they are. Outputs: tools/diag/lib/<name>/libcalib_lm.so, picked up through CALIB_LM_LIBRARY.
#   bash tools/diag/build_diag.sh            (here or on the GPU box; ~1 min per variant, built in parallel)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
W=$(mktemp -d)
mkdir -p $W/pkg/csrc $W/include
cp $R/camera-calibration_amd/csrc/*.hpp $R/camera-calibration_amd/csrc/*.hip $W/pkg/csrc/
cp $R/include/*.h $W/include/
(cd $W/pkg/csrc && patch -s -p0 < $R/tools/diag/fused_diag.patch)
build() {   # name, flags
  mkdir -p $R/tools/diag/lib/$1
  (cd $W/pkg/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wno-unused-function $2 \
      -o $R/tools/diag/lib/$1/libcalib_lm.so calib_lm.hip)
}
A="-DCALIB_IGNORE_DONE"
build stamps "-DCALIB_STAMPS" &
build noV "$A -DCALIB_ABLATE_VALU" &
build noVM "$A -DCALIB_ABLATE_VALU -DCALIB_ABLATE_MFMA" &
build noVMW "$A -DCALIB_ABLATE_VALU -DCALIB_ABLATE_MFMA -DCALIB_ABLATE_LDSW" &
build noVMWG "$A -DCALIB_ABLATE_VALU -DCALIB_ABLATE_MFMA -DCALIB_ABLATE_LDSW -DCALIB_ABLATE_GLD" &
wait
rm -rf $W
ls $R/tools/diag/lib/
