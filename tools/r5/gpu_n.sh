#!/bin/bash
for lib in vido-slam_amd/libvido_slam_hip.so vido-slam_amd/variants/qt128.so vido-slam_amd/variants/qt512.so vido-slam_amd/variants/qt1024.so; do echo "== $lib"
  VIDO_LIB_PATH=/root/repo/$lib timeout 120 python tools/prof_frontend_batch.py 2>&1 | grep pyramid_ms | cut -c1-330; done
VIDO_LIB_PATH=/root/repo/vido-slam_amd/variants/qt512.so timeout 300 python -m pytest tests/test_orb_gpu.py -m gpu -q -x 2>&1 | tail -2
