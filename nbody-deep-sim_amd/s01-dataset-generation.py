#!/usr/bin/env python3
"""Dataset-generation CLI with the reference's flags and CSV wire format
(src/s01-dataset-generation.py:12-91 flags, :93-104 cartesian product over list-valued flags,
:108-125 columns, :218-241 one row per particle per step), running the simulation on the MI355X
through galaxify.simulation. Rows are formatted state by state in native code (the reference builds one
dict per particle per step and hands it to csv.DictWriter). Extension: --sim-type plummer (the reference has disk | spiral only).

  python s01-dataset-generation.py --integrator leapfrog --n-bodies 3 25 50 --sim-type spiral \\
         --steps 1000 --seed 7 --output data/train/output_file_1.csv
"""
import argparse
import ctypes
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from galaxify import galaxies, simulation  # noqa: E402
from nbd import _lib  # noqa: E402
from nbd.plummer import generate_plummer  # noqa: E402

FIELDNAMES = ["scene", "scene_type", "step", "step_time", "mass", "x", "y", "z", "vx", "vy", "vz",
              "ax", "ay", "az", "u", "k"]


def write_states(f, scene_id, scene_type, states, masses):
    """Append the rows of one scene to the binary file `f`: for every state, for every particle, the 16 columns
    above, each value printed as Python's csv module prints it (str() of the fp32 / float64 / float value). The
    per-state constants and the float64 masses are printed here, once; the nine fp32 columns of every row by
    nbd_csv_format_state (csrc/csv_format.hip), which reproduces str(np.float32(x)) digit for digit."""
    L = _lib.lib()
    m = [s.encode() for s in np.asarray(masses).astype(str).tolist()]
    n = len(m)
    if n == 0:
        return
    mass_chars = b"".join(m)
    mass_off = np.zeros(n + 1, dtype=np.int32)
    np.cumsum([len(s) for s in m], out=mass_off[1:])
    buf, cap = None, 0
    for st in states:
        prefix = f"{scene_id},{scene_type},{st.step},{st.step_time},".encode()
        suffix = (f",{'' if st.u_energy is None else st.u_energy},{'' if st.k_energy is None else st.k_energy}"
                  "\r\n").encode()                                     # csv.writer's default line terminator
        need = L.nbd_csv_state_bound(n, len(prefix), len(mass_chars), len(suffix))
        if need > cap:
            buf, cap = ctypes.create_string_buffer(need), need
        p, v, a = (t.cpu().contiguous() for t in (st.positions, st.velocities, st.accelerations))
        if not (p.dtype == v.dtype == a.dtype == torch.float32 and p.shape == v.shape == a.shape == (n, 3)):
            raise ValueError("write_states: estados (n,3) float32 esperados")
        wrote = L.nbd_csv_format_state(buf, cap, prefix, len(prefix), mass_chars, mass_off.ctypes.data, p.data_ptr(),
                                       v.data_ptr(), a.data_ptr(), n, suffix, len(suffix))
        if wrote < 0:
            raise _lib.NbdError("nbd_csv_format_state: argumentos rechazados")
        f.write(memoryview(buf)[:wrote])


def build_parser():
    p = argparse.ArgumentParser(description="Generación de dataset de simulaciones de galaxias (MI355X)")
    p.add_argument("--n-bodies", type=int, nargs="+", required=True)
    p.add_argument("--integrator", type=str, default="leapfrog", choices=["leapfrog", "euler"], required=True)
    p.add_argument("--output", type=str, required=True)
    p.add_argument("--sim-type", type=str, nargs="+", choices=["disk", "spiral", "plummer"], default=["disk"])
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--dt", type=float, default=0.0001)
    p.add_argument("--softening", type=float, default=0.05)
    p.add_argument("--g", type=float, default=4.5e-6)
    p.add_argument("--total-mass", type=float, default=1.0)
    p.add_argument("--radial-scale", type=float, default=3.0)
    p.add_argument("--height-scale", type=float, default=0.3)
    p.add_argument("--black-hole-mass", type=float, default=0.01)
    p.add_argument("--n-arms", type=int, default=2)
    p.add_argument("--pitch-angle", type=float, default=-np.pi / 6)
    p.add_argument("--arm-strength", type=float, default=0.3)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--device", type=str, choices=["cuda", "cpu"], default=None)
    return p


def initial_conditions(c):
    common = dict(n_bodies=c["n_bodies"], total_mass=c["total_mass"], radial_scale=c["radial_scale"],
                  height_scale=c["height_scale"], g_const=c["g"], seed=c["seed"])
    if c["sim_type"] == "disk":
        return galaxies.generate_disk(black_hole_mass=c["black_hole_mass"], **common)
    if c["sim_type"] == "spiral":
        return galaxies.generate_spiral(black_hole_mass=c["black_hole_mass"], n_arms=c["n_arms"],
                                        pitch_angle=c["pitch_angle"], arm_strength=c["arm_strength"], **common)
    return generate_plummer(c["n_bodies"], seed=c["seed"] or 0, total_mass=c["total_mass"],
                            scale=c["radial_scale"], g_const=c["g"])


def main(argv=None):
    args = build_parser().parse_args(argv)
    params = {k: (v if isinstance(v, list) else [v]) for k, v in vars(args).items() if k not in ("output", "device")}
    keys = list(params)
    combos = list(itertools.product(*(params[k] for k in keys)))
    print(f"Generando {len(combos)} escenarios -> {args.output}")
    cls = simulation.EulerSimulator if args.integrator == "euler" else simulation.LeapFrogSimulator
    with open(args.output, "wb") as f:
        f.write((",".join(FIELDNAMES) + "\r\n").encode())
        for scene_id, combo in enumerate(combos):
            c = dict(zip(keys, combo))
            pos, vel, masses = initial_conditions(c)
            sim = cls(positions=pos, velocities=vel, masses=masses, g_const=c["g"], softening=c["softening"],
                      dt=c["dt"], calc_energy=True, device=args.device)
            states = sim.run(c["steps"])
            write_states(f, scene_id, c["sim_type"], states, masses)
            print(f"  escena {scene_id + 1}/{len(combos)}: n={c['n_bodies']} {c['sim_type']} {c['steps']} pasos "
                  f"({1e3 * sum(s.step_time for s in states) / max(len(states), 1):.3f} ms/paso en GPU)")


if __name__ == "__main__":
    main()
