"""Host-side profile of GraphModel.predict in a rollout (one C-ABI call per prediction): cProfile over 2000 calls.
python tools/profile_predict_host.py"""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT]
import torch
import gnn
from nbd.plummer import generate_plummer
n = 4096
torch.manual_seed(0)
model = gnn.GraphModel(input_dim=4, node_encoder_dims=None, gnn_dim=64, message_passing_steps=2, aggr="mean",
                       output_hiddens=None, device="cuda", neighbors=10, scale_factor=1e6)
p, v, m = generate_plummer(n, seed=1234)
pos = torch.tensor(p, dtype=torch.float32, device="cuda"); vel = torch.tensor(v, dtype=torch.float32, device="cuda")
m1 = torch.tensor(m * n, dtype=torch.float32, device="cuda")[:, None]
feat = torch.cat([vel, m1], 1)
for _ in range(10): model.predict(pos, feat)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): model.predict(pos, feat)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue())
