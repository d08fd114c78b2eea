#!/bin/bash
mkdir -p gpurun_out/r2e
cd /root/repo
timeout 300 python tools/dbg_nms.py 2>&1 | grep -v "^\[vido\]" | tail -40
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d /root/repo/gpurun_out/r2e/pmc -o fast -- python /root/repo/tools/dbg_fast_batch.py > /root/repo/gpurun_out/r2e/pmc.log 2>&1; echo "pmc rc $?"
cd /root/repo
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r2e/pmc/**/*counter_collection.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    agg={}
    for r in rows:
        if 'fast_strips' in r['Kernel_Name'] or 'k_resize' in r['Kernel_Name'] or 'quadtree' in r['Kernel_Name']:
            k=(r['Kernel_Name'][:30], r['Counter_Name']); agg.setdefault(k,[]).append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print(k, len(v), sum(v)/len(v))
PY
find gpurun_out/r2e/pmc -name "*.csv" -size +2M -delete
