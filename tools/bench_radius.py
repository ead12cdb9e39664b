"""radius_lists (search + merge + O(E) transpose) at BASELINE configs[3]'s shape; for --kernel-trace runs."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT):
    sys.path.insert(0, _p)
import torch
from nbd import graphops
from nbd.plummer import generate_plummer
n = 16384
p, v, m = generate_plummer(n, seed=1234)
pos = torch.tensor(p * 4.599349753792708, dtype=torch.float32, device="cuda")
for _ in range(5):
    graphops.radius_lists(pos, 1.0, loop=True, max_num_neighbors=32)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    lists = graphops.radius_lists(pos, 1.0, loop=True, max_num_neighbors=32)
e1.record(); torch.cuda.synchronize()
print(json.dumps({"radius_lists_ms": e0.elapsed_time(e1) / 50, "edges": int(lists.rowptr[-1])}))
