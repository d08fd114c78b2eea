import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vido_slam_amd as V
from oracle import pyoracle as O
ctx = V.Context(width=640, height=480, max_batch=2)
g = V.synth.make_frame(640, 480, seed=3)
kps, desc = ctx.orb_extract(g)
print("gpu kps", len(kps))
p = O.orb_params()
rk, rd = O.orb_extract(p, g)
print("ref kps", len(rk))
n = min(len(kps), len(rk))
bad = [i for i in range(n) if (kps[i]["x"], kps[i]["y"], kps[i]["octave"]) != (rk[i]["x"], rk[i]["y"], rk[i]["octave"])]
print("mismatch count", len(bad), bad[:10])
