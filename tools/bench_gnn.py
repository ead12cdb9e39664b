"""GNN rollout step at BASELINE configs[2] (published GraphModel shape, N = 4 096, predict's k = 50): the captured
hipGraph step and the eager step, plus a 200-step Trainer.evaluate_rollout (wall per step vs the captured step's GPU
time). The workload of the --pmc / --kernel-trace passes for the GNN kernels.   python tools/bench_gnn.py [iters]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT):
    sys.path.insert(0, _p)
import pandas as pd
import torch
import gnn
import trainer
from nbd.data import Data
from nbd.plummer import generate_plummer


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, (time.perf_counter() - t0) / iters * 1e3


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    n = 4096
    torch.manual_seed(0)
    model = gnn.GraphModel(input_dim=4, node_encoder_dims=None, gnn_dim=64, message_passing_steps=2, aggr="mean",
                           output_hiddens=None, device="cuda", neighbors=10, scale_factor=1e6)
    tr = trainer.Trainer(model, None, device="cuda", dt=1e-4)
    p, v, m = generate_plummer(n, seed=1234)
    pos = torch.tensor(p, dtype=torch.float32, device="cuda")
    vel = torch.tensor(v, dtype=torch.float32, device="cuda")
    m1 = torch.tensor(m * n, dtype=torch.float32, device="cuda")[:, None]
    acc = model.predict(pos, torch.cat([vel, m1], 1))
    out = {"n": n, "k": 50}
    feat = torch.cat([vel, m1], 1)
    out["eager_predict_ms_gpu"], out["eager_predict_ms_wall"] = timeit(lambda: model.predict(pos, feat), iters)   # one C-ABI call
    out["eager_step_ms_gpu"], out["eager_step_ms_wall"] = timeit(lambda: tr.step(pos, vel, m1, acc, 1e-4), iters)
    adv = tr._capture_step(pos, vel, m1, acc, 1e-4)
    out["captured_step_ms_gpu"], out["captured_step_ms_wall"] = timeit(lambda: adv(clone=False), iters)
    # evaluate_rollout over 200 steps: ground truth = the initial state repeated (the harness cost is what is timed)
    steps = 200
    x = torch.cat([pos, vel, m1], 1).repeat(steps, 1)
    y = acc.repeat(steps, 1)
    step = torch.arange(steps, device="cuda").repeat_interleave(n)
    data = Data(x=x, y=y, step=step)
    tr.evaluate_rollout("f.csv", data, 0, steps, 1e-4, pd.DataFrame())          # warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    df = tr.evaluate_rollout("f.csv", data, 0, steps, 1e-4, pd.DataFrame())
    wall = time.perf_counter() - t0
    gpu_step = float(df.groupby("step")["step_time"].first().iloc[1:].mean() * 1e3)
    t1 = time.perf_counter()
    lt = tr.last_rollout_timing
    out["evaluate_rollout_200_steps"] = {"wall_ms_total_incl_frame": wall * 1e3, "frame_rows": len(df),
                                         "mean_step_time_ms_from_events": gpu_step,
                                         "loop_wall_ms_per_step": lt["loop_wall_s"] / lt["steps"] * 1e3,
                                         "loop_wall_over_captured_gpu": lt["loop_wall_s"] / lt["steps"] * 1e3 / out["captured_step_ms_gpu"]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
