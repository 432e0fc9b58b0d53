#!/bin/bash
# PMC passes over tools/attn_only.py (separate runs per counter group, kernel-trace only): usage tools/pmc_attn.sh <out-dir>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/${1:-pmc_attn}; mkdir -p $O; rm -f $O/counters.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*\|TCP_[A-Z_0-9]*" | sort -u > $O/avail.txt
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" \
           "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  rm -rf /tmp/pa_$i
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pa_$i -o p -- python $R/tools/attn_only.py > /tmp/pa_$i.log 2>&1
  f=$(find /tmp/pa_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" >> $O/counters.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "attn_d64" in k:
        acc["attn_d64_mx8_kernel" if "mx8" in k else "attn_d64_v2_kernel"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    for c, v in d.items():
        print(f"{k:42s} {c:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
PY
  else echo "group $i failed: $(tail -2 /tmp/pa_$i.log)" >> $O/counters.txt; fi
done
cat $O/counters.txt
