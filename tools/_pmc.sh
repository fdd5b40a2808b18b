#!/bin/bash
# tools/_pmc.sh <outdir> <binary args...>: one rocprofv3 --pmc pass per counter group over a stand-alone harness
OUT=$1; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_$i -o p -- "$@" > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "p_counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' >> $GRAFT_REPO_ROOT/$OUT/pmc.txt
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "wino2" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("%-40s %16.1f  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
  [ -z "$f" ] && tail -3 /tmp/pmc_$i.log >> $GRAFT_REPO_ROOT/$OUT/pmc.txt
done
cat $GRAFT_REPO_ROOT/$OUT/pmc.txt
