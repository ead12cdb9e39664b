#!/usr/bin/env python3
"""Fits the published GNN shape (gnn_experiment.py:61-72: input 4, gnn_dim 64, 2 EdgeConv layers, mean,
k = 10, RMSE of 1e6-scaled accelerations, Adam lr 1e-2) for a few CPU-minutes and stores the
state_dict as tests/golden/gnn_small_trained.pt, so that the rollout-MSE legs of the tests and of
bench.py run a model that has actually learned the force law instead of random weights (the reference
publishes no weights: .gitignore:19-20).

Test-infrastructure only: training uses the CPU oracle modules (oracle/surrogate_oracle.py are plain
torch nn.Modules, hence differentiable) and oracle-integrated spiral galaxies with the dataset-CLI
defaults (G = 4.5e-6, softening 0.05, dt = 1e-4). Run in the build container:
    python tests/golden/train_small_gnn.py
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path[:0] = [os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT]
from galaxify import galaxies                      # noqa: E402  (this repo's seed-compatible generators)
from oracle import galaxify_oracle as go           # noqa: E402
from oracle import surrogate_oracle as so          # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)
G, EPS, DT, K, SCALE = 4.5e-6, 0.05, 1e-4, 10, 1e6
GAL = dict(total_mass=1.0, radial_scale=3.0, height_scale=0.3, g_const=G, black_hole_mass=0.01)


def scenes(seeds, sizes, steps, every):
    out = []
    for seed in seeds:
        for n in sizes:
            p, v, m = galaxies.generate_spiral(n_bodies=n, seed=seed, **GAL)
            sim = go.OracleSimulator(positions=p, velocities=v, masses=m, g_const=G, softening=EPS, dt=DT)
            m1 = sim.masses[:, None]
            for s in range(steps):
                sim.leapfrog_step()
                if s % every == 0:
                    x7 = torch.cat([sim.positions, sim.velocities, m1], 1).clone()
                    out.append((x7, so.knn_graph(sim.positions, K), sim.accelerations.clone()))
    return out


def main(minutes=4.0):
    data = scenes(seeds=range(40), sizes=(25, 50, 100, 250), steps=100, every=25)
    held = scenes(seeds=[100], sizes=(100,), steps=50, every=10)
    model = so.GraphModelOracle(input_dim=4, gnn_dim=64, message_passing_steps=2, aggr="mean", neighbors=K)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.25, patience=8)

    # The loss is the reference's (RMSE of 1e6-scaled accelerations, gnn.py:150-161), but the model is trained
    # in the scaled output space (its head predicts a * 1e6) and the factor is folded into the final Linear
    # when saving: the function is the same, the optimisation is far better conditioned than asking Adam
    # (lr 1e-2) for head weights of order 1e-7.
    def loss_on(batch):
        return torch.sqrt(torch.stack([torch.nn.functional.mse_loss(model.forward_graph(x, ei), y * SCALE)
                                       for x, ei, y in batch]).mean())
    t0, epoch = time.time(), 0
    while time.time() - t0 < minutes * 60:
        perm = torch.randperm(len(data)).tolist()
        tot = 0.0
        for b in range(0, len(perm), 16):
            opt.zero_grad()
            loss = loss_on([data[i] for i in perm[b:b + 16]])
            loss.backward()
            opt.step()
            tot += loss.item()
        epoch += 1
        train = tot / ((len(perm) + 15) // 16)
        sched.step(train)
        with torch.no_grad():
            val = loss_on(held).item()
        print(f"epoch {epoch:3d}  train rmse*1e6 {train:10.3f}  held-out {val:10.3f}  lr {opt.param_groups[0]['lr']:.1e}", flush=True)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd["output.weight"] /= SCALE
    sd["output.bias"] /= SCALE
    torch.save(sd, os.path.join(HERE, "gnn_small_trained.pt"))
    print("saved", os.path.join(HERE, "gnn_small_trained.pt"))


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 4.0)
