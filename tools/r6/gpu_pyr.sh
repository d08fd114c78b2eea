#!/bin/bash
# pyramid kernel: round-5 item order vs XCD-dealt items — time (kernel trace) and memory-side traffic (separate --pmc passes)
cd /tmp; export TMPDIR=/tmp; REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/pyr; mkdir -p $OUT
for X in 0 1; do
  VIDO_PYR_XCD=$X timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$X -o p -- python $REPO/tools/prof_frontend_batch.py > $OUT/kt$X.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    VIDO_PYR_XCD=$X timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc${X}_$C -o p -- python $REPO/tools/prof_frontend_batch.py > $OUT/pmc${X}_$C.log 2>&1
  done
done
python - <<'P'
import csv, glob, os
OUT=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/pyr'
for X in (0,1):
    for f in glob.glob(f'{OUT}/kt{X}/**/*kernel_stats.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'pyramid' in r['Name'] or 'fast_strips' in r['Name']: print('xcd',X,r['Name'][:40],r['Calls'],r['AverageNs'],r['MinNs'])
    for C in ('FETCH_SIZE','WRITE_SIZE'):
        for f in glob.glob(f'{OUT}/pmc{X}_{C}/**/*counter_collection.csv', recursive=True):
            v={}
            for r in csv.DictReader(open(f)):
                if r['Counter_Name']==C: v.setdefault(r['Kernel_Name'][:30],[]).append(float(r['Counter_Value']))
            for k,l in v.items():
                if 'pyramid' in k or 'fast_strips' in k or 'blur' in k: print('xcd',X,C,k,len(l),sum(l)/len(l))
P
