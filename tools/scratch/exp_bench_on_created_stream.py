"""bench.py's headline with the CALLER's stream a created (non-default) stream instead of the process' default stream: does the detector's queue matter?"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
sys.argv = ["bench.py", "--steps", "100", "--warmup", "5", "--no-extra", "--cpu-baseline", "0"] + sys.argv[1:]
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
