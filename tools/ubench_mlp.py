"""Per-call GPU time (graph replay) of the small dense layers around the ContinuousConv layers at N = 16 384:
encoder MLP 4 -> 32 -> 64 -> 128 (tanh), LayerNorm(256), decoder 256 -> 64 -> 32 -> 3.   python tools/ubench_mlp.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT]
import json
import torch
from nbd import nnops


def t(fn, it=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(it): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


n = 16384
out = {}
for (k, m, act) in [(4, 32, "tanh"), (32, 64, "tanh"), (64, 128, "tanh"), (256, 64, "tanh"), (64, 32, "tanh"), (32, 3, None)]:
    x, w, b = torch.randn(n, k).cuda(), torch.randn(m, k).cuda(), torch.randn(m).cuda()
    o = torch.empty(n, m).cuda()
    out[f"linear_{k}_to_{m}_us"] = t(lambda: nnops.linear(x, w, b, act=act, out=o))
x = torch.randn(n, 256).cuda(); g = torch.ones(256).cuda(); o = torch.empty(n, 256).cuda()
out["layernorm_256_us"] = t(lambda: nnops.layernorm(x, g, g, 1e-5, out=o))
print(json.dumps(out))
