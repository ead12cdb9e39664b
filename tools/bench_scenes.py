"""Rollout of SEVERAL scenes at the reference's sizes (N = 3 .. 500, gnn_experiment.py:34; 1000 steps, dt = 1e-4): scene by
scene, as /root/reference/trainer.py:171-175 runs them, against all scenes of the file advanced together as one batched
system (Trainer.evaluate_rollout_scenes). Ground truth = the HIP direct-force integrator; model = the briefly trained GNN
fixture (tests/golden/gnn_small_trained.pt) or, with `contconv`, a random-weight ContinuousConvModel of the published
shape. Prints one JSON line.   python tools/bench_scenes.py [gnn|contconv] [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT]
import numpy as np
import pandas as pd
import torch
import contconv, gnn, trainer
from galaxify import galaxies, simulation
from nbd.data import Data

SIZES = [3, 25, 64, 150, 300, 500]


def scene(n, steps, seed):
    p, v, m = galaxies.generate_spiral(n_bodies=n, total_mass=1.0, radial_scale=3.0, height_scale=0.3, g_const=4.5e-6,
                                       black_hole_mass=0.01, seed=seed)
    sim = simulation.LeapFrogSimulator(positions=p, velocities=v, masses=m, g_const=4.5e-6, softening=0.05, dt=1e-4,
                                       calc_energy=False, device="cuda")
    m1 = sim.masses[:, None]
    xs, ys, st = [], [], []
    for s in range(steps):
        sim.step()
        xs.append(torch.cat([sim.positions, sim.velocities, m1], 1)); ys.append(sim.accelerations.clone())
        st.append(torch.full((n,), s, device="cuda"))
    return Data(x=torch.cat(xs), y=torch.cat(ys), step=torch.cat(st))


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "gnn"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    torch.manual_seed(0)
    if kind == "gnn":
        model = gnn.GraphModel(input_dim=4, gnn_dim=64, message_passing_steps=2, aggr="mean", device="cuda", neighbors=10,
                               scale_factor=1e6)
        w = os.path.join(ROOT, "tests", "golden", "gnn_small_trained.pt")
        if os.path.exists(w):
            model.load_state_dict(torch.load(w, map_location="cuda"))
    else:
        model = contconv.ContinuousConvModel(in_channels=4, out_channels=3, filter_resolution=[6, 4], radius=1.0, agg="mean",
                                             self_loops=True, continuous_conv_layers=2, continuous_conv_dim=128,
                                             encoder_hiddens=[32, 64], decoder_hiddens=[64, 32], device="cuda").eval()
    scenes = [scene(n, steps, 700 + i) for i, n in enumerate(SIZES)]
    tr = trainer.Trainer(model, None, device="cuda", dt=1e-4)
    out = {"model": kind, "scene_sizes": SIZES, "steps": steps}
    for name in ("scene_by_scene", "scenes_together"):
        for rep in range(2):                                  # second pass timed (allocator pools, weight caches warm)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            loop = 0.0
            if name == "scene_by_scene":
                df = pd.DataFrame(columns=trainer.ROLLOUT_COLUMNS)
                for i, d in enumerate(scenes):
                    df = tr.evaluate_rollout("f.csv", d, i, steps, 1e-4, df)
                    loop += tr.last_rollout_timing["loop_wall_s"]
            else:
                df = tr.evaluate_rollout_scenes("f.csv", scenes, steps, 1e-4, None)
                loop = tr.last_rollout_timing["loop_wall_s"]
            torch.cuda.synchronize(); wall = time.perf_counter() - t0
        mse = trainer.rollout_mse(df)
        out[name] = {"wall_s_whole_call": wall, "stepping_loops_wall_s": loop,
                     "ms_per_step_all_scenes": loop / (steps - 1) * 1e3, "captured": tr.last_rollout_timing["captured"],
                     "path": getattr(model, "last_path", None), "capture": tr.last_capture,
                     "pos_mse_last_step_by_scene": [float(x) for x in mse["pos_mse"].groupby(level=1).last()]}
    out["speedup_stepping"] = out["scene_by_scene"]["stepping_loops_wall_s"] / out["scenes_together"]["stepping_loops_wall_s"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
