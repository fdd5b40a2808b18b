# parameter variants of csrc/conv3d_wino3.hip, one library per variant: built here with  ESTD_LIB_SUFFIX=_w3_<name> ESTD_BUILD_DEFS="-DESTD_W3..." python -m estdepth_amd.build,
# run on the GPU box with  W3_VARIANTS="<name> ..." bash tools/w3_variants.sh  (how the table of profiles/r5_wino3_table.txt was measured)
for v in base $W3_VARIANTS; do
  lib="X=1"; [ "$v" != base ] && lib="ESTD_LIB=$PWD/estdepth_amd/lib/libestd_hip_w3_$v.so"
  echo "== $v"; env $lib ESTD_BINDING=ctypes W3_NOCHECK=1 python tools/w3_bench.py 3 30 2>&1 | grep "plain\|accumulate\|resid" | tail -4 | cut -c1-120
done
