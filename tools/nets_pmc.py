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
    # Normalisation.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1 024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs and also counts the dispatch ramps around a
    # short kernel (19 600 .. 89 000 "cycles" per microsecond of kernel time in one run), so the denominator used here is kernel time x 2.4 GHz x 1 024 SIMDs.  Calibration:
    # k_gconv3x3_m32 issues 516 096 v_mfma_f32_32x32x2 (64 cycles each) in 26.9 us = 50 % by construction and reads 49.5 % on this scale; the rocBLAS fp32 GEMMs read
    # 62..67 % at their measured ~100 TFLOP/s of the 157.3 peak.
    CLK_GHZ = 2.4
    tot_busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for v in rows.values()); tot_ns = sum(dur.values())
    ks = []
    for k, v in rows.items():
        act = v.get("GRBM_GUI_ACTIVE", 0.0); busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if act <= 0 or dur[k] <= 0: continue
        ks.append(dict(kernel=k[:120], launches=calls[k], total_ms=round(dur[k] / 1e6, 3), mfma_busy_cycles=busy, gui_active_cycles=act,
                       mfma_util_pct=round(100.0 * busy / (dur[k] * CLK_GHZ * 1024), 2)))
    ks.sort(key=lambda r: -r["total_ms"])
    res = dict(command=os.environ.get("NETS_PMC_COMMAND", "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python tools/nets_pmc.py (2 passes of LiteFlowNet, MonoDepth2, Mask R-CNN X-101-FPN, fp32, 1242x375)"),
               definition="mfma_util_pct = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x 2.4 GHz x 1024 SIMDs); see tools/nets_pmc.py for the calibration",
               all_kernels_mfma_util_pct=round(100.0 * tot_busy / max(tot_ns * CLK_GHZ * 1024, 1), 2),
               note="the Winograd convolutions (miopenSp3AsmConv*) read 0: they run on the vector ALUs (v_fma / v_pk_fma), not on the matrix cores", kernels=ks[:30])
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(dict(all=res["all_kernels_mfma_util_pct"], top=[(r["kernel"][:60], r["total_ms"], r["mfma_util_pct"]) for r in ks[:8]]), indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise": summarise(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "nets_mfma.json")
    else: run()
