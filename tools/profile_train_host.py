"""Host-side profile of GraphModel.train_graph_batch (the reference's training configuration, tools/bench_train.py):
cProfile of 30 steps, top functions by cumulative and by own time.   python tools/profile_train_host.py"""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch
import gnn
from bench_train import Batches, make_batch

torch.manual_seed(0)
model = gnn.GraphModel(input_dim=4, gnn_dim=64, message_passing_steps=2, aggr="mean", neighbors=10, device="cuda", scale_factor=1e6)
opt = torch.optim.Adam(model.parameters(), lr=0.01)
loader = Batches(make_batch(64, 10))
for _ in range(5):
    model.train_graph_batch(opt, loader.next())
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    model.train_graph_batch(opt, loader.next())
torch.cuda.synchronize()
pr.disable()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(22)
    print(s.getvalue()[:5200])
