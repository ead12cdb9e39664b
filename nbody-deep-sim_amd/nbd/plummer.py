"""Seeded Plummer-sphere initial conditions (host side, numpy).

The reference has no Plummer generator (only generate_disk / generate_spiral,
src/galaxify/galaxies.py:54,195), so BASELINE.json's "synthetic Plummer-sphere particle
sets" come from here. Output matches what the reference's generators hand to the simulator:
(positions (N,3), velocities (N,3), masses (N,)) float64 ndarrays (galaxies.py:54-67).

Sampling: Aarseth, Henon & Wielen (1974). Units G = M = a = 1, equal masses 1/N, radii
re-drawn beyond `rmax`, isotropic velocities with speed q*v_esc where q is drawn by
rejection from g(q) = q^2 (1-q^2)^(7/2); centre-of-mass position and velocity removed.
"""
from __future__ import annotations

import numpy as np


def _unit_vectors(rng: np.random.Generator, n: int) -> np.ndarray:
    z = rng.uniform(-1.0, 1.0, n)
    phi = rng.uniform(0.0, 2.0 * np.pi, n)
    s = np.sqrt(1.0 - z * z)
    return np.stack([s * np.cos(phi), s * np.sin(phi), z], axis=1)


def generate_plummer(n_bodies: int, seed: int = 1234, total_mass: float = 1.0,
                     scale: float = 1.0, g_const: float = 1.0, rmax: float = 50.0):
    rng = np.random.default_rng(seed)
    n = int(n_bodies)
    # radii from the cumulative mass M(r)/M = r^3 (1+r^2)^(-3/2)
    r = np.empty(n)
    todo = np.arange(n)
    while todo.size:
        u = rng.uniform(1e-10, 1.0, todo.size)
        rr = 1.0 / np.sqrt(u ** (-2.0 / 3.0) - 1.0)
        ok = rr <= rmax
        r[todo[ok]] = rr[ok]
        todo = todo[~ok]
    pos = _unit_vectors(rng, n) * r[:, None]
    # speeds: rejection sampling of q in [0,1] under g(q) <= 0.1
    q = np.empty(n)
    todo = np.arange(n)
    while todo.size:
        x = rng.uniform(0.0, 1.0, todo.size)
        y = rng.uniform(0.0, 0.1, todo.size)
        ok = y < x * x * (1.0 - x * x) ** 3.5
        q[todo[ok]] = x[ok]
        todo = todo[~ok]
    v_esc = np.sqrt(2.0) * (1.0 + r * r) ** -0.25
    vel = _unit_vectors(rng, n) * (q * v_esc)[:, None]
    # physical units
    pos *= scale
    vel *= np.sqrt(g_const * total_mass / scale)
    masses = np.full(n, total_mass / n)
    pos -= pos.mean(axis=0)
    vel -= vel.mean(axis=0)
    return pos, vel, masses
