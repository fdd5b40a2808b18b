mkdir -p gpurun_out/r3
C="TA_TA_BUSY_sum SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU TCP_TOTAL_CACHE_ACCESSES_sum"
R=$PWD
bash tools/pmc_collect.sh "$C" $R/gpurun_out/r3/wa_base_pmc.csv -- python $R/tools/hbm_bench.py | grep "kernel,\|warp_attention_kernel<3>"
ESTD_BINDING=ctypes ESTD_LIB=$R/estdepth_amd/lib/libestd_hip_share.so bash tools/pmc_collect.sh "$C" $R/gpurun_out/r3/wa_share_pmc.csv -- python $R/tools/hbm_bench.py | grep "warp_attention_kernel<3>"
