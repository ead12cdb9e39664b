"""Wall time of the dataset CLI (s01-dataset-generation.py) on the reference's own sizes: scenes of 100..500
bodies x 1000 steps, energies on (the CLI hard-codes them).   python tools/cli_timing.py"""
import importlib.util, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "nbody-deep-sim_amd")
sys.path[:0] = [PKG, ROOT]
import torch
spec = importlib.util.spec_from_file_location("s01", f"{PKG}/s01-dataset-generation.py")
cli = importlib.util.module_from_spec(spec); spec.loader.exec_module(cli)
out = {}
with tempfile.TemporaryDirectory() as tmp:
    for label, env in (("captured_chunks", "1"), ("eager", "0")):
        os.environ["NBD_RUN_GRAPH"] = env
        args = ["--integrator", "leapfrog", "--n-bodies", "100", "300", "500", "--sim-type", "spiral", "--steps", "1000",
                "--dt", "1e-4", "--g", "4.5e-6", "--softening", "0.05", "--seed", "3", "--output", f"{tmp}/{label}.csv", "--device", "cuda"]
        cli.main(args[:-6] + ["--seed", "4", "--output", f"{tmp}/warm.csv", "--device", "cuda"]) if label == "captured_chunks" else None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cli.main(args)
        torch.cuda.synchronize()
        out[label + "_seconds_3_scenes_x_1000_steps"] = time.perf_counter() - t0
        out[label + "_csv_bytes"] = os.path.getsize(f"{tmp}/{label}.csv")
    import datautils
    for rep in range(2):                                  # second pass: page cache and kernels warm
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ds = datautils.ParticleGraphDataset(f"{tmp}/captured_chunks.csv", k=10, device="cuda")
        torch.cuda.synchronize()
        out["dataset_load_seconds_k10"] = time.perf_counter() - t0
        out["dataset_graphs"] = len(ds)
    # the per-group construction the reference spells out (datautils.py:23-48) with this build's kNN kernel
    import pandas as pd
    from nbd import graphops
    torch.cuda.synchronize(); t0 = time.perf_counter()
    df = pd.read_csv(f"{tmp}/captured_chunks.csv"); graphs = []
    for (scene, step), g in df.groupby(["scene", "step"]):
        x = torch.tensor(g[["x", "y", "z", "vx", "vy", "vz", "mass"]].values, dtype=torch.float, device="cuda")
        y = torch.tensor(g[["ax", "ay", "az"]].values, dtype=torch.float, device="cuda")
        graphs.append((x, y, graphops.knn_graph(x[:, :3].contiguous(), k=10, loop=False)))
    torch.cuda.synchronize()
    out["per_group_construction_seconds_k10"] = time.perf_counter() - t0
# the writer alone: this build's native formatter against the reference's csv.DictWriter loop (s01...py:218-241)
import csv, io
import numpy as np
from galaxify import galaxies, simulation
pos, vel, m = galaxies.generate_spiral(n_bodies=500, total_mass=1.0, radial_scale=3.0, height_scale=0.3, g_const=4.5e-6,
                                       black_hole_mass=0.01, seed=3)
states = simulation.LeapFrogSimulator(positions=pos, velocities=vel, masses=m, g_const=4.5e-6, softening=0.05, dt=1e-4,
                                      calc_energy=True, device="cuda").run(200)
f = io.BytesIO(); t0 = time.perf_counter(); cli.write_states(f, 0, "spiral", states, m); t_native = time.perf_counter() - t0
g = io.StringIO(newline=""); w = csv.DictWriter(g, fieldnames=cli.FIELDNAMES); t0 = time.perf_counter()
for st in states:
    p_, v_, a_ = st.positions.cpu().numpy(), st.velocities.cpu().numpy(), st.accelerations.cpu().numpy()
    for i in range(p_.shape[0]):
        w.writerow({"scene": 0, "scene_type": "spiral", "step": st.step, "step_time": st.step_time, "mass": m[i],
                    "x": p_[i, 0], "y": p_[i, 1], "z": p_[i, 2], "vx": v_[i, 0], "vy": v_[i, 1], "vz": v_[i, 2],
                    "ax": a_[i, 0], "ay": a_[i, 1], "az": a_[i, 2], "u": st.u_energy, "k": st.k_energy})
t_dict = time.perf_counter() - t0
assert f.getvalue() == g.getvalue().encode()
out["writer_rows"] = 500 * 200
out["writer_native_rows_per_s"] = 500 * 200 / t_native
out["writer_dictwriter_rows_per_s"] = 500 * 200 / t_dict
print(json.dumps(out))
