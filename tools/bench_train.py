"""Training-step benchmark: ms per `train_graph_batch` (forward + backward on the HIP
kernels + torch.optim.Adam step + the two .item() syncs of the reference's API) at the reference's
training configuration:
  GNN      published shape (gnn_experiment.py:61-79), k = 10, batch_size = 64 graphs of
           n-bodies cycling through 3, 25, 50, 100, 250, 500 (gnn_experiment.py:33, 87-93)
  ContConv published shape (contconv_experiment.py:62-78), batch_size = 16 graphs (:90)
and, beside it, the same step on the CPU oracle (torch autograd, all host threads).
usage: bench_train.py [iters] [--cpu]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT]
import numpy as np
import torch
import gnn, contconv
from nbd import graphops
from nbd.data import Data, collate
from nbd.plummer import generate_plummer

SIZES = [3, 25, 50, 100, 250, 500]


def make_batch(n_graphs, k, seed=0):
    graphs = []
    for g in range(n_graphs):
        n = SIZES[g % len(SIZES)]
        p, v, m = generate_plummer(n, seed=seed + g)
        x = torch.tensor(np.concatenate([p, v, (m * n)[:, None]], 1), dtype=torch.float32, device="cuda")
        from galaxify import simulation
        sim = simulation.LeapFrogSimulator(positions=p, velocities=v, masses=m, calc_energy=False, device="cuda")
        ei = graphops.knn_graph(x[:, :3].contiguous(), k=k, loop=False) if k > 0 else \
            torch.zeros((2, 0), dtype=torch.int64, device="cuda")
        graphs.append(Data(x=x, edge_index=ei, y=sim.accelerations.clone()))
    return graphs


class Batches:
    """What the reference's DataLoader does per step: a fresh collate of the (rotated) graph list, so
    nothing derived from a batch (CSR, transposed adjacency) survives from one step to the next."""

    def __init__(self, graphs):
        self.graphs, self.i = graphs, 0

    def next(self):
        self.i += 1
        r = self.i % len(self.graphs)
        return collate(self.graphs[r:] + self.graphs[:r])


def time_steps(step, iters, warm=3):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def run(iters=20, cpu=False):
    out = {}
    torch.manual_seed(0)
    # ---- GNN
    model = gnn.GraphModel(input_dim=4, gnn_dim=64, message_passing_steps=2, aggr="mean", neighbors=10, device="cuda",
                           scale_factor=1e6)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    loader = Batches(make_batch(64, 10))
    ms = time_steps(lambda: model.train_graph_batch(opt, loader.next()), iters)
    data = loader.next()
    model.eval()
    with torch.no_grad():
        fwd = time_steps(lambda: model.forward(data), iters)
    out["gnn"] = {"nodes": int(data.x.shape[0]), "edges": int(data.edge_index.shape[1]), "train_step_ms": ms,
                  "inference_forward_ms": fwd, "batch_graphs": 64, "k": 10}
    if cpu:
        from oracle import surrogate_oracle as so
        ora = so.GraphModelOracle(input_dim=4, gnn_dim=64, message_passing_steps=2, aggr="mean", neighbors=10)
        oo = torch.optim.Adam(ora.parameters(), lr=0.01)
        x7, ei, y = data.x.cpu(), data.edge_index.cpu(), data.y.cpu()

        def cpu_step():
            oo.zero_grad()
            pred = ora.forward_graph(x7, ei)
            loss = torch.sqrt(torch.nn.functional.mse_loss(pred * 1e6, y * 1e6))
            loss.backward(); oo.step(); loss.item()
        cpu_step()
        t0 = time.perf_counter()
        for _ in range(5):
            cpu_step()
        out["gnn"]["cpu_oracle_train_step_ms"] = (time.perf_counter() - t0) / 5 * 1e3
        out["gnn"]["cpu_threads"] = torch.get_num_threads()
    # ---- ContinuousConv
    cmodel = contconv.ContinuousConvModel(in_channels=4, out_channels=3, filter_resolution=[6, 4], radius=1.0, agg="mean",
                                          self_loops=True, continuous_conv_layers=2, continuous_conv_dim=128,
                                          encoder_hiddens=[32, 64], decoder_hiddens=[64, 32], device="cuda",
                                          scale_factor=1e6)
    copt = torch.optim.Adam(cmodel.parameters(), lr=0.01)
    cloader = Batches(make_batch(16, 0, seed=100))
    cms = time_steps(lambda: cmodel.train_graph_batch(copt, cloader.next()), iters)
    cdata = cloader.next()
    out["contconv"] = {"nodes": int(cdata.x.shape[0]), "train_step_ms": cms, "batch_graphs": 16}
    if cpu:
        from oracle import surrogate_oracle as so
        cora = so.ContinuousConvModelOracle(in_channels=4, out_channels=3, filter_resolution=[6, 4], radius=1.0,
                                            agg="mean", self_loops=True, continuous_conv_layers=2,
                                            continuous_conv_dim=128, encoder_hiddens=[32, 64], decoder_hiddens=[64, 32])
        co = torch.optim.Adam(cora.parameters(), lr=0.01)
        x7, b, y = cdata.x.cpu(), cdata.batch.cpu(), cdata.y.cpu()

        def ccpu_step():
            co.zero_grad()
            loss = torch.sqrt(torch.nn.functional.mse_loss(cora.forward_x(x7, batch=b) * 1e6, y * 1e6))
            loss.backward(); co.step(); loss.item()
        ccpu_step()
        t0 = time.perf_counter()
        for _ in range(2):
            ccpu_step()
        out["contconv"]["cpu_oracle_train_step_ms"] = (time.perf_counter() - t0) / 2 * 1e3
    return out


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20, "--cpu" in sys.argv)))
