"""Diagnostic: per-call GPU time of nbd_linear_f32 / aggregate / layernorm at the GNN's shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT]
import torch
from nbd import nnops, graphops, _lib

def t(fn, it=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(it): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3

n = 4096
for (k, m) in [(4, 128), (64, 64), (64, 128), (68, 3), (128, 128), (256, 64)]:
    x, w, b = torch.randn(n, k).cuda(), torch.randn(m, k).cuda(), torch.randn(m).cuda()
    out = torch.empty(n, m).cuda()
    print(f"linear n={n} k={k} m={m}: {t(lambda: nnops.linear(x, w, b, out=out)):.2f} us")
pq = torch.randn(n, 128).cuda(); pos = torch.randn(n, 3).cuda()
for kk in (32, 50):
    ei = graphops.knn_graph(pos, kk); src = ei[0].contiguous(); s = torch.empty(n, 64).cuda()
    print(f"aggregate k={kk}: {t(lambda: nnops.edgeconv_aggregate(pq, 64, None, src, kk, 'mean', out=s)):.2f} us")
    print(f"knn k={kk}: {t(lambda: graphops.knn_graph(pos, kk), 50):.2f} us")
x = torch.randn(n, 68).cuda(); g = torch.ones(68).cuda(); o = torch.empty(n, 68).cuda()
print(f"layernorm: {t(lambda: nnops.layernorm(x, g, g, 1e-5, out=o)):.2f} us")
a = torch.randn(n, 3).cuda(); v = torch.randn(n, 3).cuda()
from nbd import direct
print(f"kick: {t(lambda: direct.kick(v, a, 0.1)):.2f} us")

# whole GNN predict (published shape) under graph replay: pure GPU time of the kernel sequence
import gnn
torch.manual_seed(0)
model = gnn.GraphModel(input_dim=4, gnn_dim=64, message_passing_steps=2, aggr="mean", device="cuda", neighbors=10).eval()
feat = torch.randn(n, 4).cuda()
for kk in (32, 50):
    for fused in (True, False):
        model.use_fused = fused
        with torch.no_grad():
            print(f"GNN predict k={kk} fused={fused}: {t(lambda: model.predict(pos, feat, neighbors=kk), 20):.1f} us")
from nbd.data import Data
ei = graphops.knn_graph(pos, 32)
d = Data(x=torch.cat([pos, feat], 1), edge_index=ei); d._regular_k = 32
for fused in (True, False):
    model.use_fused = fused
    with torch.no_grad():
        print(f"GNN forward only (graph given, k=32) fused={fused}: {t(lambda: model.forward(d), 20):.1f} us")
