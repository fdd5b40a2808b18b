#!/bin/bash
# Timing ablations of conv3d_k3_split_kernel (results are WRONG for any mask != 0; timing only).
#   bit 1: no per-tap barrier   2: no A-fragment LDS reads   4: no B-fragment LDS reads   8: no weight hand-over
#   16: no ring refill / slice prefetch   32: no epilogue
# build here:  tools/ablate_split.sh build ;  run on the GPU box:  tools/ablate_split.sh run
cd "$(dirname "$0")/.."
MASKS="${MASKS:-0 1 2 4 6 8 16 32 63}"
if [ "$1" = build ]; then
  for m in $MASKS; do ESTD_LIB_SUFFIX=_sabl$m ESTD_BUILD_DEFS="-DESTD_SABL=$m" python -c "from estdepth_amd import build; build.build()" >/dev/null || exit 1; done
else
  for m in $MASKS; do echo -n "SABL=$m  "; ESTD_LIB=estdepth_amd/lib/libestd_hip_sabl$m.so ESTD_CONV3D_ARITH=bf16x3 timeout 120 python tools/conv_bench.py ${N:-3} 20; done
fi
