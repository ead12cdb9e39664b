#!/usr/bin/env python3
"""Dataset-generation CLI with the reference's flags and CSV wire format
(src/s01-dataset-generation.py:12-91 flags, :93-104 cartesian product over list-valued flags,
:108-125 columns, :218-241 one row per particle per step), running the simulation on the MI355X
through galaxify.simulation. Rows are formatted array-wise (the reference builds one dict per
particle per step). Extension: --sim-type plummer (the reference has disk | spiral only).

  python s01-dataset-generation.py --integrator leapfrog --n-bodies 3 25 50 --sim-type spiral \\
         --steps 1000 --seed 7 --output data/train/output_file_1.csv
"""
import argparse
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from galaxify import galaxies, simulation  # noqa: E402
from nbd.plummer import generate_plummer  # noqa: E402

FIELDNAMES = ["scene", "scene_type", "step", "step_time", "mass", "x", "y", "z", "vx", "vy", "vz",
              "ax", "ay", "az", "u", "k"]


def write_states(f, scene_id, scene_type, states, masses):
    """Append the rows of one scene: for every state, for every particle, the 16 columns above, each
    value printed as Python's csv module prints it (str() of the fp32 / float64 / float value)."""
    m = np.asarray(masses).astype(str)
    n = m.shape[0]
    for st in states:
        cols = [np.full(n, str(scene_id)), np.full(n, scene_type), np.full(n, str(st.step)),
                np.full(n, str(st.step_time)), m]
        for t in (st.positions, st.velocities, st.accelerations):
            a = t.cpu().numpy()
            cols += [a[:, 0].astype(str), a[:, 1].astype(str), a[:, 2].astype(str)]
        cols += [np.full(n, "" if st.u_energy is None else str(st.u_energy)),
                 np.full(n, "" if st.k_energy is None else str(st.k_energy))]
        lines = cols[0]
        for c in cols[1:]:
            lines = np.char.add(np.char.add(lines, ","), c)
        f.write("\r\n".join(lines.tolist()) + "\r\n")          # csv.writer's default line terminator


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
    with open(args.output, "w", newline="") as f:
        f.write(",".join(FIELDNAMES) + "\r\n")
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
