#!/bin/bash
# memory-side traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, kernel trace only) of the split-fp16 kernels on their headline shapes
cd /tmp; export TMPDIR=/tmp; REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/pmc_split; mkdir -p $OUT
run() { tag=$1; shift; for C in FETCH_SIZE WRITE_SIZE; do timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${tag}_$C -o p -- "$@" > $OUT/${tag}_$C.log 2>&1; done; }
run c1_layer3 python $REPO/tools/r6/c1b3_run.py 1024 1024 50 68 20
run c1_layer1 python $REPO/tools/r6/c1b3_run.py 256 256 200 272 20
run c3_p2 python $REPO/tools/r6/c3h_run.py 1 256 256 200 272 20
run c3_mask python $REPO/tools/r6/c3h_run.py 100 256 256 14 14 20
run fc6 python $REPO/tools/r6/fch_run.py 1000 12544 1024 20
python - <<'P'
import csv, glob, os
OUT=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/pmc_split'
for tag in ('c1_layer3','c1_layer1','c3_p2','c3_mask','fc6'):
    res={}
    for C in ('FETCH_SIZE','WRITE_SIZE'):
        for f in glob.glob(f'{OUT}/{tag}_{C}/**/*counter_collection.csv', recursive=True):
            v={}
            for r in csv.DictReader(open(f)):
                if r['Counter_Name']==C: v.setdefault(r['Kernel_Name'][:48],[]).append(float(r['Counter_Value']))
            for k,l in v.items():
                if any(s in k for s in ('conv1x1_b3','conv3x3_h','k_fc_h')): res.setdefault(k,{})[C]=(len(l), sum(l[2:])/max(len(l[2:]),1))
    for k,d in res.items(): print(tag, k, {c:(n, round(x,1)) for c,(n,x) in d.items()})
P
