#!/bin/bash
# round 5, call y: two ranks on the one GPU (--oversubscribe: gloo all-reduce through host memory): the sharded global BA and the replica bench still run end to end
set -u
OUT=gpurun_out/r5y; mkdir -p $OUT
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --oversubscribe --steps 20 --warmup 3 --cpu-baseline 0 > $OUT/bench2.json 2> $OUT/bench2.err
echo "rc $?"; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5y/bench2.json").read().strip().splitlines()[-1]); print(d["value"], d["n_gpus"], d.get("global_ba_iters_per_s")); print(d["extra"]["global_ba"])
except Exception as e:
    print("parse failed", e); print(open("gpurun_out/r5y/bench2.err").read()[-1500:])
PY
