#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "1024 1024 50 68 20 1" "1024 1024 50 68 20 0" "2048 2048 25 34 20 1" "256 256 200 272 20 1"; do
  rm -rf /tmp/prof; timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o t -- python $R/tools/r6/c1b3_run.py $cfg > /tmp/kt.log 2>&1 || tail -3 /tmp/kt.log
  echo "== $cfg"; grep -h "k_conv1x1_b3" $(find /tmp/prof -name "*kernel_stats.csv") | cut -c1-200
done
cfg="1024 1024 50 68 10 1"
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAVES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  rm -rf /tmp/prof; timeout 150 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/prof -o t -- python $R/tools/r6/c1b3_run.py $cfg > /tmp/pmc.log 2>&1 || tail -3 /tmp/pmc.log
  f=$(find /tmp/prof -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "k_conv1x1_b3" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items(): print("%-32s per launch %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
done
