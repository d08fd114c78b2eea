#!/bin/bash
# round 4, call AC: HBM traffic of k_wino3x3 / k_conv1x1 / k_gconv3x3_m32d (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE in separate passes, MI355X_MICROARCH.md recipe)
cd /tmp; export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r4ac; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  WINO_ONLY=1 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/wn_$C -o p -- python $REPO/tools/prof_wino.py > $OUT/wn_$C.log 2>&1
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/c1_$C -o p -- python $REPO/tools/prof_conv1x1.py > $OUT/c1_$C.log 2>&1
done
cd $REPO
python - "$OUT" <<'PY' | tee $OUT/pmc_traffic_nets_kernels.txt
import csv, glob, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for tag in ("wn", "c1"):
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, tag, C), recursive=True):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                n = r["Kernel_Name"]
                if "k_wino3x3" in n or "k_conv1x1" in n:
                    import re
                    m = re.search(r"(k_wino3x3<[^>]*>|k_conv1x1<[^>]*>)", n)
                    key = (m.group(1) if m else n[:40], r.get("Grid_Size", ""))
                    agg[key].append(float(r["Counter_Value"]))
            for k, v in agg.items(): res[k][C] = (sum(v) / len(v), len(v))
print("kernel | grid | launches | FETCH_SIZE KB mean | WRITE_SIZE KB mean | HBM bytes per launch = 2 x FETCH (gfx950 correction) + WRITE")
for k in sorted(res):
    f = res[k].get("FETCH_SIZE", (0, 0)); w = res[k].get("WRITE_SIZE", (0, 0))
    print("%-28s grid %-9s x%-3d  fetch %10.1f KB  write %10.1f KB  -> %8.1f MB" % (k[0], k[1], f[1], f[0], w[0], (2 * f[0] + w[0]) * 1024 / 1e6))
PY
find $OUT -name "*.csv" -delete
