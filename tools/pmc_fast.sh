#!/bin/bash
# SQ counters of k_fast_strips (tools/prof_frontend_batch.py) for one library build: tools/pmc_fast.sh <lib.so> <outdir>
lib=$1; out=/root/repo/$2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift
  VIDO_LIB_PATH=$lib timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_$tag -o fast -- python /root/repo/tools/prof_frontend_batch.py > $out/pmc_$tag.log 2>&1 || tail -3 $out/pmc_$tag.log
}
run a SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
cd /root/repo
python - "$out" <<'PY'
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/pmc_*/**/*counter_collection.csv', recursive=True)):
    agg = {}
    for r in csv.DictReader(open(f)):
        if 'fast_strips' in r['Kernel_Name']:
            agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()): print("%-26s %3d launches  avg %.4g" % (k, len(v), sum(v) / len(v)))
PY
find $out -name "*.csv" -size +1M -delete
