"""One forward pass of each network node (fp32, random-init weights, 1242x375 like bench.py's `nets_fp32_1242x375`) for a `rocprofv3 --pmc` run:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> -- python tools/nets_pmc.py
`python tools/nets_pmc.py --summarise <dir>` turns the counter csv into profiles/<round>/nets_mfma.json: per kernel, MFMA-busy cycles over the
cycles the kernel had the GPU (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs' worth of matrix pipes)."""
import os, sys, json, glob, csv, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import numpy as np, torch
    import vido_slam_amd as V
    from vido_slam_amd import nets
    ctx = V.Context(width=640, height=480, max_batch=1)
    ops = nets.HipOps(ctx)
    rgb = np.random.RandomState(0).randint(0, 256, size=(375, 1242, 3)).astype(np.uint8)
    lfn = nets.fill_deterministic(nets.LiteFlowNet(ops.correlation, epilogue=ops.bias_act_), 1).eval().cuda()
    md = nets.fill_deterministic(nets.MonoDepth2(), 2).eval().cuda()
    mr = nets.fill_maskrcnn(nets.MaskRCNN(ops), 3).eval().cuda()
    for _ in range(2):
        nets.analyse_flow(lfn, rgb, rgb); nets.analyse_depth(md, rgb); nets.analyse_image(mr, rgb)
    torch.cuda.synchronize()


def summarise(d, out):
    rows = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); dur = collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]; rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE": calls[k] += 1; dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot_busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for v in rows.values()); tot_act = sum(v.get("GRBM_GUI_ACTIVE", 0) for v in rows.values())
    ks = []
    for k, v in rows.items():
        act = v.get("GRBM_GUI_ACTIVE", 0.0); busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if act <= 0: continue
        ks.append(dict(kernel=k[:120], launches=calls[k], total_ms=round(dur[k] / 1e6, 3), mfma_busy_cycles=busy, gui_active_cycles=act,
                       mfma_util_pct=round(100.0 * busy / (act * 256 * 4), 2)))          # busy cycles are summed over the 1024 SIMDs of the chip
    ks.sort(key=lambda r: -r["total_ms"])
    res = dict(command="rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python tools/nets_pmc.py (2 passes of LiteFlowNet, MonoDepth2, Mask R-CNN X-101-FPN, fp32, 1242x375)",
               definition="mfma_util_pct = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs)",
               all_kernels_mfma_util_pct=round(100.0 * tot_busy / max(tot_act * 1024, 1), 2), kernels=ks[:25])
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(dict(all=res["all_kernels_mfma_util_pct"], top=[(r["kernel"][:60], r["total_ms"], r["mfma_util_pct"]) for r in ks[:8]]), indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise": summarise(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "nets_mfma.json")
    else: run()
