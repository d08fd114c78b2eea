#!/bin/bash
# round 4, call M: wino.hip wired into the networks: network tests, then the headline with and without it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4m; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_wino_gpu.py tests/test_nets_modules_gpu.py tests/test_maskrcnn_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r4m/pytest_nets.txt
cat gpurun_out/r4m/pytest_nets.txt
timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/r4m/bench_wino.json 2>gpurun_out/r4m/bench_wino.err; tail -c 600 gpurun_out/r4m/bench_wino.err
VIDO_NO_WINO=1 timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/r4m/bench_nowino.json 2>gpurun_out/r4m/bench_nowino.err
python - <<'P'
import json
for n in ("wino", "nowino"):
    try:
        d = json.loads(open("gpurun_out/r4m/bench_%s.json" % n).read().strip().splitlines()[-1])
        s = d["stage_ms"]
        print(n, d["value"], d["ms_per_step"], {k: s[k] for k in ("liteflownet_ms", "monodepth2_ms", "maskrcnn_x101_fpn_ms", "tracker_thread_ms", "tracker_wait_for_nets_ms")})
    except Exception as e:
        print(n, "failed", e)
P
