"""1x1 convolutions of the detector body as GEMMs: MIOpen's conv2d vs torch.mm (default heuristic) vs torch.mm with TunableOp.  python tools/exp_gemm.py [tune]"""
import os, sys, time
tune = len(sys.argv) > 1 and sys.argv[1] == "tune"
if tune:
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"; os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
    os.environ["PYTORCH_TUNABLEOP_FILENAME"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "tunableop_exp.csv")
    os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "30"); os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS", "5")
import torch, torch.nn.functional as F
dev = "cuda"
shapes = [(256, 64, 54400), (256, 256, 54400), (512, 256, 13600), (512, 512, 13600), (1024, 512, 3400), (1024, 1024, 3400), (2048, 1024, 850), (2048, 2048, 850), (256, 2048, 850), (256, 1024, 3400)]
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
tot_c = tot_m = 0
for co, ci, hw in shapes:
    h = {54400: (200, 272), 13600: (100, 136), 3400: (50, 68), 850: (25, 34)}[hw]
    x = torch.randn(1, ci, *h, device=dev); w = torch.randn(co, ci, 1, 1, device=dev)
    x2 = x.view(ci, hw); w2 = w.view(co, ci); out = torch.empty(co, hw, device=dev)
    tc = timeit(lambda: F.conv2d(x, w)); tm = timeit(lambda: torch.mm(w2, x2, out=out))
    err = float((F.conv2d(x, w).view(co, hw) - torch.mm(w2, x2)).abs().max())
    gf = 2.0 * co * ci * hw / 1e9
    print("Cout %5d Cin %5d HW %6d  conv2d %7.1f us (%5.1f TF/s)  mm %7.1f us (%5.1f TF/s)  maxdiff %.2e" % (co, ci, hw, tc, gf / tc * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 * 1e3, tm, gf / tm * 1e3 / 1e3, err))
    tot_c += tc; tot_m += tm
print("sum conv2d %.1f us  mm %.1f us  (tunableop %s)" % (tot_c, tot_m, tune))
