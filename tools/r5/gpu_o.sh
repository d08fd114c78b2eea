#!/bin/bash
for lib in vido-slam_amd/libvido_slam_hip.so vido-slam_amd/variants/pt1.so vido-slam_amd/variants/pt4.so; do echo "== $lib"
  VIDO_LIB_PATH=/root/repo/$lib timeout 120 python tools/prof_frontend_batch.py 2>&1 | grep pyramid_ms | cut -c1-100; done
timeout 300 python -m pytest tests/test_orb_gpu.py -m gpu -q -x 2>&1 | tail -2
