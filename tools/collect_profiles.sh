#!/bin/bash
# Profile collection on the GPU box (writes under gpurun_out/profiles_<round>/; copy what is kept into profiles/).
#   bash tools/collect_profiles.sh [round prefix, default r6]
# (default build: the superseded A/B kernels -- depth-only / row-only Winograd, bf16 operand split -- are not part of it; their lines
#  of the round-4 collection are gone)
set -u
R=$PWD
P=${1:-r6}
OUT=$R/gpurun_out/profiles_$P
mkdir -p $OUT tools/bin
cd /tmp && export TMPDIR=/tmp
for wl in joint estm cfg5; do
  rm -rf /tmp/prof_$wl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o p -- python $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads > $OUT/${P}_bench_${wl}_profiled.json 2> $OUT/${P}_bench_${wl}_profiled.err
  T=$(find /tmp/prof_$wl -name "p_kernel_trace.csv" | head -1)
  python $R/tools/prof_summary.py $T $OUT/${P}_bench_${wl}_kernel_stats.csv 5
  python $R/tools/prof_timeline.py $T 5 --gaps > $OUT/${P}_bench_${wl}_timeline.txt 2>&1
  python $R/tools/prof_seq.py $T -1 60 5 > $OUT/${P}_bench_${wl}_seq.txt 2>&1
  S=$(find /tmp/prof_$wl -name "p_kernel_stats.csv" | head -1)
  [ -n "$S" ] && head -41 $S > $OUT/${P}_bench_${wl}_rocprof_stats_top40.csv
done
# PMC: dominant kernel (conv_bench: N = 3 volumes) per algorithm, separate --pmc passes
# (wino3 = ESTD_CONV3D_ALGO=wino2 with ESTD_W3=1, the default: the 32 -> 32 instances on csrc/conv3d_wino3.hip; wino2 = ESTD_W3=0)
algo_env() { case $1 in wino3) echo "ESTD_CONV3D_ALGO=wino2 ESTD_W3=1";; wino2) echo "ESTD_CONV3D_ALGO=wino2 ESTD_W3=0";; *) echo "ESTD_CONV3D_ALGO=$1";; esac; }
for algo in wino3 wino2 direct; do
  env $(algo_env $algo) bash $R/tools/pmc_collect.sh "FETCH_SIZE WRITE_SIZE" $OUT/${P}_conv3d_${algo}_pmc.csv -- python $R/tools/conv_bench.py 3 10 > /dev/null 2>&1
done
python $R/tools/pmc_json.py $OUT $P          # ${P}_conv3d_pmc.json: what bench.py reads for roofline.traffic (from profiles/)
cp $OUT/${P}_conv3d_pmc.json $R/profiles/ 2>/dev/null      # (this run's copy of the tree: the bench lines below then carry THIS round's traffic figure)
bash $R/tools/pmc_collect.sh "FETCH_SIZE WRITE_SIZE" $OUT/${P}_hbm_kernels_pmc.csv -- python $R/tools/hbm_bench.py > /dev/null 2>&1
# the hardware's own matrix-pipe utilisation counter of every convolution kernel, stand-alone benches
for b in "conv_bench.py 3 10" "conv_bench.py 1 10" "kv_bench.py" "head_bench.py" "conv2d_bench.py" "conv1x1_bench.py" "taps_bench.py"; do
  bash $R/tools/pmc_collect.sh "MfmaUtil SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" /tmp/mfma_one.csv -- python $R/tools/$b > /dev/null 2>&1
  echo "# python tools/$b" >> $OUT/${P}_mfma_util_pmc.csv; grep -v "at::native\|rocclr" /tmp/mfma_one.csv >> $OUT/${P}_mfma_util_pmc.csv
done
cd $R
python tools/hbm_bench.py > $OUT/${P}_hbm_bench.txt 2>&1
for algo in wino3 wino2 direct; do
  echo "# $algo: $(algo_env $algo)" >> $OUT/${P}_conv_bench.txt
  env $(algo_env $algo) CB_EPI=1 python tools/conv_bench.py 3 30 2>&1 | grep -v amdgpu >> $OUT/${P}_conv_bench.txt
  env $(algo_env $algo) python tools/conv_bench.py 1 30 2>&1 | grep -v amdgpu >> $OUT/${P}_conv_bench.txt
done
python tools/head_bench.py 2>&1 | grep -v amdgpu >> $OUT/${P}_conv_bench.txt
python tools/kv_bench.py 2>&1 | grep -v amdgpu >> $OUT/${P}_conv_bench.txt
python tools/w3_bench.py 3 30 2>&1 | grep -v amdgpu > $OUT/${P}_w3_bench.txt
python tools/gate_bench.py 2>&1 | grep -v amdgpu > $OUT/${P}_gate_bench.txt
python tools/conv2d_bench.py 2>&1 | grep -v amdgpu > $OUT/${P}_conv2d_bench.txt
python tools/psm_small_bench.py 2>&1 | grep -v amdgpu > $OUT/${P}_psm_small_bench.txt
python tools/conv1x1_bench.py 2>&1 | grep -v amdgpu > $OUT/${P}_conv1x1_bench.txt
python tools/conv1x1_cfg_sweep.py 2>&1 | grep -v amdgpu > $OUT/${P}_conv1x1_sweep.txt
python tools/overlap_probe.py 40 2>&1 | grep -v amdgpu > $OUT/${P}_overlap_probe.txt
# what writing only h in warp_attention would buy (timing ablations: variant libraries built before the gpurun call, ctypes binding) + the texture-address unit's account of it
for v in "" _waabl1 _waabl2; do
  [ -f estdepth_amd/lib/libestd_hip$v.so ] || continue
  echo "# libestd_hip$v.so (ESTD_WA_ABL: '' = default [V_t | h] record, 1 = h only as a compact 16-channel volume, 2 = h only into its half of the record); cfg2 size, then cfg5 size" >> $OUT/${P}_warp_attention_honly.txt
  ESTD_BINDING=ctypes ESTD_LIB=$R/estdepth_amd/lib/libestd_hip$v.so python tools/hbm_bench.py 2>&1 | grep "warp_att" >> $OUT/${P}_warp_attention_honly.txt
  ESTD_BINDING=ctypes ESTD_LIB=$R/estdepth_amd/lib/libestd_hip$v.so python tools/hbm_bench.py 128 240 320 2>&1 | grep "warp_att" >> $OUT/${P}_warp_attention_honly.txt
done
for v in "" _waabl1; do
  [ -f estdepth_amd/lib/libestd_hip$v.so ] || continue
  ESTD_BINDING=ctypes ESTD_LIB=$R/estdepth_amd/lib/libestd_hip$v.so bash $R/tools/pmc_collect.sh "TA_TA_BUSY_sum SQ_BUSY_CU_CYCLES TCP_TCC_READ_REQ_sum FETCH_SIZE WRITE_SIZE" /tmp/wa_pmc.csv -- python $R/tools/hbm_bench.py > /dev/null 2>&1
  echo "# libestd_hip$v.so" >> $OUT/${P}_warp_attention_honly_pmc.csv; grep "kernel,\|warp_attention" /tmp/wa_pmc.csv >> $OUT/${P}_warp_attention_honly_pmc.csv
done
cd $R
python tools/taps_bench.py 2>&1 | grep -v amdgpu > $OUT/${P}_taps_bench.txt
# (built here, before the gpurun call: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/src/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap)
[ -x tools/bin/mfma_valu_overlap ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/src/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap
tools/bin/mfma_valu_overlap > $OUT/${P}_mfma_valu_overlap.txt 2>&1
# (ESTD_COLLECT_LINES=0: stop here -- copy ${P}_conv3d_pmc.json into profiles/ first, so that the bench lines of tools/collect_bench_lines.sh carry this round's roofline.traffic)
[ "${ESTD_COLLECT_LINES:-1}" = 1 ] || { ls -la $OUT; exit 0; }
# default bench lines (with cpu_baseline + parity) of every workload; algorithm A/B; the world-size-1 RCCL run
last() { grep "^{" | tail -1; }
python bench.py --steps 20 --warmup 5 2>/dev/null | last > $OUT/${P}_bench_joint.json
python bench.py --steps 20 --warmup 5 --sustained-s 60 --no-cpu-baseline --no-other-workloads --no-replay-profile 2>/dev/null | last > $OUT/${P}_bench_joint_sustained60.json
python bench.py --workload estm 2>/dev/null | last > $OUT/${P}_bench_estm.json
python bench.py --workload cfg1 2>/dev/null | last > $OUT/${P}_bench_cfg1.json
python bench.py --workload cfg5 --steps 5 --warmup 2 2>/dev/null | last > $OUT/${P}_bench_cfg5.json
python bench.py --workload stream --steps 20 2>/dev/null | last > $OUT/${P}_bench_stream.json
ESTD_W3=0 python bench.py --no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads 2>/dev/null | last > $OUT/${P}_bench_joint_two_axis.json
ESTD_GATE_IN_CONV=0 python bench.py --no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads 2>/dev/null | last > $OUT/${P}_bench_joint_gate_pass.json
python bench.py --conv3d-algo direct --no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads 2>/dev/null | last > $OUT/${P}_bench_joint_direct_conv.json
python bench.py --no-graph --no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads 2>/dev/null | last > $OUT/${P}_bench_joint_eager.json
ESTD_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-alt --no-other-workloads 2>/dev/null | last > $OUT/${P}_bench_joint_rccl_world1.json
ESTD_FORCE_DIST=1 python bench.py --workload estm --no-cpu-baseline --no-alt --no-other-workloads 2>/dev/null | last > $OUT/${P}_bench_estm_rccl_world1.json
python bench.py --gpus 2 --workload cfg1 --steps 5 --warmup 2 2>/dev/null | last > $OUT/${P}_bench_gpus2_codepath.json
ESTD_HIP_1X1=0 python bench.py --no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads 2>/dev/null | last > $OUT/${P}_bench_joint_lib1x1.json
ESTD_HIP_TAPS=0 ESTD_HIP_POOL=0 ESTD_HIP_STEM7=0 python bench.py --no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads 2>/dev/null | last > $OUT/${P}_bench_joint_lib2d.json
python bench.py --graph-memory copy --no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads 2>/dev/null | last > $OUT/${P}_bench_joint_graph_copy.json
python bench.py --workload estm --graph-memory copy --no-cpu-baseline --no-alt --no-replay-profile --no-other-workloads 2>/dev/null | last > $OUT/${P}_bench_estm_graph_copy.json
timeout 2400 python -m pytest tests/ -q -m gpu > $OUT/${P}_gputests.log 2>&1
ls -la $OUT
