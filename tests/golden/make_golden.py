#!/usr/bin/env python3
"""Generates tests/golden/direct_*.npz by running the REAL reference integrator.

Run in the build container only (it imports /root/reference/src/galaxify, which never travels
to the GPU box):   python tests/golden/make_golden.py

Each .npz holds data only: fp32 inputs (pos, vel, mass; float64 `mass64` as the generators
returned it), the scalar parameters, and the reference's outputs
  acc0                         BaseSimulator.__init__ force (simulation.py:69)
  lf1_{pos,vel,acc}            after 1 LeapFrogSimulator.step()
  lf10_{pos,vel,acc}           after 10 steps
  eu1_{pos,vel,acc}            after 1 EulerSimulator.step()
  energy0 = (U, K)             compute_energies() on the initial state
Input generators: the reference's generate_spiral/generate_disk with the dataset-CLI defaults
(s01-dataset-generation.py:44-50), and this repo's Plummer generator with the simulator's class
defaults (simulation.py:28-30).
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, "/root/reference/src")  # `galaxify` below is the REFERENCE package
warnings.filterwarnings("ignore")

import importlib.util  # noqa: E402

import torch  # noqa: E402
from galaxify import galaxies, simulation  # noqa: E402

_spec = importlib.util.spec_from_file_location(
    "nbd_plummer", os.path.join(ROOT, "nbody-deep-sim_amd", "nbd", "plummer.py"))
_plummer = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_plummer)
generate_plummer = _plummer.generate_plummer

assert simulation.__file__.startswith("/root/reference"), simulation.__file__
torch.set_num_threads(8)


def run_case(name, pos, vel, mass, g, eps, dt, long_steps=10):
    kw = dict(positions=pos, velocities=vel, masses=mass, g_const=g, softening=eps, dt=dt,
              calc_energy=True, device="cpu")
    out = {
        "pos": np.asarray(pos, dtype=np.float32), "vel": np.asarray(vel, dtype=np.float32),
        "mass": np.asarray(mass, dtype=np.float32), "mass64": np.asarray(mass, dtype=np.float64),
        "g_const": np.float64(g), "softening": np.float64(eps), "dt": np.float64(dt),
    }
    sim = simulation.LeapFrogSimulator(**kw)
    out["acc0"] = sim.accelerations.numpy().copy()
    out["energy0"] = np.array(sim.compute_energies(), dtype=np.float64)
    sim.step()
    for k, t in (("pos", sim.positions), ("vel", sim.velocities), ("acc", sim.accelerations)):
        out[f"lf1_{k}"] = t.numpy().copy()
    if long_steps:
        for _ in range(long_steps - 1):
            sim.step()
        for k, t in (("pos", sim.positions), ("vel", sim.velocities), ("acc", sim.accelerations)):
            out[f"lf{long_steps}_{k}"] = t.numpy().copy()
    eu = simulation.EulerSimulator(**kw)
    eu.step()
    for k, t in (("pos", eu.positions), ("vel", eu.velocities), ("acc", eu.accelerations)):
        out[f"eu1_{k}"] = t.numpy().copy()
    path = os.path.join(HERE, f"direct_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name:>22}: n={len(mass):5d}  {os.path.getsize(path)/1024:7.1f} KiB")


CLI = dict(g=4.5e-6, eps=0.05, dt=1e-4)          # s01-dataset-generation.py:44-50
GAL = dict(total_mass=1.0, radial_scale=3.0, height_scale=0.3, g_const=4.5e-6, black_hole_mass=0.01)

for n in (3, 25, 64, 1024):
    p, v, m = galaxies.generate_spiral(n_bodies=n, n_arms=2, pitch_angle=-np.pi / 6, arm_strength=0.3,
                                       seed=42, **GAL)
    run_case(f"spiral_n{n}", p, v, m, **CLI)
for n in (25, 1024):
    p, v, m = galaxies.generate_disk(n_bodies=n, seed=42, **GAL)
    run_case(f"disk_n{n}", p, v, m, **CLI)
for n in (64, 1000, 1024):
    p, v, m = generate_plummer(n, seed=123)
    run_case(f"plummer_n{n}", p, v, m, g=1.0, eps=0.1, dt=0.01)
p, v, m = generate_plummer(4096, seed=123)
run_case("plummer_n4096", p, v, m, g=1.0, eps=0.1, dt=0.01, long_steps=0)
# unequal masses + a massless body, in N-body units (exercises the m_j broadcast)
p, v, m = generate_plummer(300, seed=7)
rng = np.random.default_rng(7)
m = m * rng.uniform(0.1, 5.0, m.shape); m[17] = 0.0
run_case("plummer_n300_ragged_mass", p, v, m, g=1.0, eps=0.1, dt=0.01)
# softening = 0: the diagonal is removed only by fill_diagonal_(0) (simulation.py:85)
p, v, m = generate_plummer(64, seed=5)
run_case("plummer_n64_eps0", p, v, m, g=1.0, eps=0.0, dt=0.001)
