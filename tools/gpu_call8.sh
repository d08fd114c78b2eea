#!/bin/bash
# FAST loop-trip profile (debug build with -DFS_PROF), then restore the normal library
cd /root/repo
mkdir -p gpurun_out/r2i
cp vido-slam_amd/libvido_slam_hip.so /tmp/lib_ok.so; cp vido-slam_amd/libvido_slam_hip.so.stamp /tmp/lib_ok.stamp
VIDO_EXTRA_FLAGS="-DFS_PROF" python -c "import sys; sys.path.insert(0,'vido-slam_amd'); import build; build.build(force=True)" 2>&1 | grep -v warning | tail -3
python tools/dbg_fast_batch.py 2>&1 | tail -6 | tee gpurun_out/r2i/fs_prof.txt
cp /tmp/lib_ok.so vido-slam_amd/libvido_slam_hip.so; cp /tmp/lib_ok.stamp vido-slam_amd/libvido_slam_hip.so.stamp
