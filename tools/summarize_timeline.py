"""Cuts a rocprofv3 kernel-trace CSV at the marker kernels (name contains 'erfinv') and prints per phase: span, busy time, launches, top kernels."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
phases = []; cur = None
for s, e, n in rows:
    if "erfinv" in n:
        if cur: phases.append(cur)
        cur = []
    elif cur is not None:
        cur.append((s, e, n))
names = sys.argv[2].split(",") if len(sys.argv) > 2 else []
for i, ph in enumerate(phases):
    if not ph: continue
    span = (ph[-1][1] - ph[0][0]) / 1e3; busy = sum(e - s for s, e, _ in ph) / 1e3
    agg = collections.Counter(); cnt = collections.Counter()
    for s, e, n in ph:
        k = n[:70]; agg[k] += (e - s) / 1e3; cnt[k] += 1
    print("phase %d %s: span %.1f us busy %.1f us launches %d" % (i, names[i] if i < len(names) else "", span, busy, len(ph)))
    for k, v in agg.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 12):
        print("    %8.1f us %4d x  %s" % (v, cnt[k], k))
