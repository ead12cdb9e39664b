"""Phase stamps inside gnn_layer_kernel (probe build: hipcc ... -DNBD_GNN_TRACE -c csrc/gnn_fused.hip, linked into a
copy of the library -- see tools/build_contconv_trace.sh for the recipe): per wave its start, the end of the weight
staging, the end of the edge loop. Round-2 reading at N = 4096, k = 50, last layer: staging 1.3 us, edge loop 7.9 us
(four rounds of 16 gathered rows), then ~7 us of mat-vec + LayerNorm + head, kernel 18.7 us."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT):
    sys.path.insert(0, _p)
import numpy as np, torch
import gnn
from nbd import _lib
from nbd.plummer import generate_plummer
n = 4096
torch.manual_seed(0)
model = gnn.GraphModel(input_dim=4, node_encoder_dims=None, gnn_dim=64, message_passing_steps=2, aggr="mean",
                       output_hiddens=None, device="cuda", neighbors=10, scale_factor=1e6)
p, v, m = generate_plummer(n, seed=1234)
pos = torch.tensor(p, dtype=torch.float32, device="cuda"); vel = torch.tensor(v, dtype=torch.float32, device="cuda")
m1 = torch.tensor(m * n, dtype=torch.float32, device="cuda")[:, None]
feat = torch.cat([vel, m1], 1)
for _ in range(3): model.predict(pos, feat)
L = _lib.lib(); L.nbd_debug_gnn_trace.argtypes = [ctypes.c_void_p]; L.nbd_debug_gnn_trace.restype = ctypes.c_int
res = {}
for name, epi in (("first_layer_next_pq_folded", 4), ("last_layer_final_head", 2)):
    tr = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
    assert L.nbd_debug_gnn_trace_epilogue(epi) == 0
    assert L.nbd_debug_gnn_trace(tr.data_ptr()) == 0
    model.predict(pos, feat); torch.cuda.synchronize()
    assert L.nbd_debug_gnn_trace(None) == 0
    t = tr.view(-1, 8).cpu().numpy(); t = t[t[:, 0] != 0]
    if len(t) == 0:
        continue
    t0 = t[:, 0].min()
    out = {"waves": int(len(t)), "start_us_p50_max": [float(np.percentile((t[:,0]-t0)/100,50)), float(((t[:,0]-t0)/100).max())],
           "staging_us_mean": float(((t[:,1]-t[:,0])/100).mean()), "edges_us_mean": float(((t[:,2]-t[:,1])/100).mean()),
           "edges_done_at_us_p50_max": [float(np.percentile((t[:,2]-t0)/100,50)), float(((t[:,2]-t0)/100).max())]}
    if (t[:, 6] != 0).any():
        out["matvec_us_mean"] = float(((t[:,4]-t[:,2])/100).mean()); out["layernorm_us_mean"] = float(((t[:,5]-t[:,4])/100).mean())
        out["head_us_mean"] = float(((t[:,6]-t[:,5])/100).mean()); out["end_at_us_p50_max"] = [float(np.percentile((t[:,6]-t0)/100,50)), float(((t[:,6]-t0)/100).max())]
    if (t[:, 3] != 0).any():
        out["matvec_folded_us_mean"] = float(((t[:,3]-t[:,2])/100).mean()); out["end_at_us_p50_max"] = [float(np.percentile((t[:,3]-t0)/100,50)), float(((t[:,3]-t0)/100).max())]
    res[name] = out
print(json.dumps(res))
