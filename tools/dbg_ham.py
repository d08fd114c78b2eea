import sys; sys.path.insert(0,'.')
import numpy as np, vido_slam_amd as V
ctx = V.Context()
rng = np.random.RandomState(1)
def ref(a,b):
    d = np.unpackbits(a[:,None,:]^b[None,:,:],axis=2).sum(2)
    return d.argmin(1), d.min(1)
for na,nb in ((7,300),(8,300),(7,64),(7,128),(9,300),(33,300)):
    a = rng.randint(0,256,size=(na,32)).astype(np.uint8); b = rng.randint(0,256,size=(nb,32)).astype(np.uint8)
    i,d = ctx.hamming_match(a,b); ri,rd = ref(a,b)
    print(na,nb, 'ok' if (i==ri).all() and (d==rd).all() else 'BAD', i[:10], d[:10], ri[:10], rd[:10])
