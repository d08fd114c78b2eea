"""Summary of a rocprofv3 --kernel-trace of the pipelined bench for one network-stream schedule: which (hardware queue, HIP stream) pairs carried which kernels, how busy
each was inside the steady state, how much of that time two of them ran side by side, and how long the tracker's launches sat between the end of their predecessor and their
own start.  The raw trace (20 MB per run) stays on the GPU box; this prints the few hundred bytes DESIGN.md section 9 argues from.
usage: python tools/sched_timeline.py <kernel_trace.csv> [label]"""
import csv, json, sys
from collections import Counter, defaultdict

def union_len(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot

def overlap_len(a, b):
    a = sorted(a); b = sorted(b); i = j = 0; tot = 0
    while i < len(a) and j < len(b):
        s = max(a[i][0], b[j][0]); e = min(a[i][1], b[j][1])
        if e > s: tot += e - s
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return tot

def merge(iv):
    iv = sorted(iv); out = []
    for s, e in iv:
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out

def main():
    path = sys.argv[1]; label = sys.argv[2] if len(sys.argv) > 2 else path
    rows = list(csv.DictReader(open(path)))
    for r in rows: r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    pose_pairs = Counter((r["Queue_Id"], r["Stream_Id"]) for r in rows if r["Kernel_Name"].startswith("k_pose_opt"))
    tq = pose_pairs.most_common(1)[0][0] if pose_pairs else None                          # the tracker's (queue, stream): the one that carries k_pose_opt
    marks = [r["s"] for r in rows if r["Kernel_Name"].startswith("k_depth_prescale") and (r["Queue_Id"], r["Stream_Id"]) == tq]      # one per tracked frame
    if len(marks) < 14:
        print(json.dumps({"label": label, "error": "fewer than 14 tracked frames in the trace"})); return
    t0, t1, nframes = marks[-12], marks[-2], 10
    reg = [r for r in rows if r["s"] >= t0 and r["e"] <= t1]
    by = defaultdict(list); names = defaultdict(Counter)
    for r in reg:
        k = "q%s/s%s" % (r["Queue_Id"], r["Stream_Id"]); by[k].append((r["s"], r["e"])); names[k][r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "")[:36]] += 1
    keys = sorted(by, key=lambda k: -sum(e - s for s, e in by[k]))[:6]
    out = {"label": label, "frames": nframes, "ms_per_frame": round((t1 - t0) / nframes / 1e6, 3), "queues": {}}
    merged = {k: merge(by[k]) for k in keys}
    for k in keys:
        out["queues"][k] = {"kernels_per_frame": round(len(by[k]) / nframes, 1), "busy_ms_per_frame": round(union_len(by[k]) / nframes / 1e6, 3), "top": names[k].most_common(3)}
    out["all_queues_union_busy_ms_per_frame"] = round(union_len([iv for k in by for iv in by[k]]) / nframes / 1e6, 3)
    out["pair_overlap_ms_per_frame"] = {"%s & %s" % (a, b): round(overlap_len(merged[a], merged[b]) / nframes / 1e6, 3) for i, a in enumerate(keys) for b in keys[i + 1:]}
    # the tracker's stream: the pair that carries k_pose_opt
    trk = [k for k in by if names[k].get("k_pose_opt", 0)]
    if trk:
        tk = trk[0]; seq = sorted([r for r in reg if "q%s/s%s" % (r["Queue_Id"], r["Stream_Id"]) == tk], key=lambda r: r["s"])
        gaps = [b["s"] - a["e"] for a, b in zip(seq, seq[1:]) if b["s"] - a["e"] < 400000]      # (longer gaps are host work / the wait for the next frame)
        stretch = Counter()
        for r in seq: stretch[r["Kernel_Name"].split("(")[0][:24]] += r["e"] - r["s"]
        out["tracker_stream"] = {"pair": tk, "launches_per_frame": round(len(seq) / nframes, 1), "kernel_ms_per_frame": round(sum(r["e"] - r["s"] for r in seq) / nframes / 1e6, 3),
                                 "mean_gap_us_under_400": round(sum(gaps) / max(len(gaps), 1) / 1e3, 1), "top_time": [(n, round(v / nframes / 1e3, 1)) for n, v in stretch.most_common(5)]}
    print(json.dumps(out))

if __name__ == "__main__":
    main()
