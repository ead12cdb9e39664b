"""CPU oracle for the direct-force path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module. The product path (nbody-deep-sim_amd/) never imports anything under oracle/.

What it is: a row-blocked torch-CPU restatement of the reference integrator, written from
the reference's documented operation sequence (cited per function). "Row-blocked" means
targets are processed in blocks of <= `block` rows so the (rows, N, 3) temporaries stay
small; inside a block the operations and their order are those of the reference, so the
result is the reference's up to the fp32 summation order of `sum(dim=1)`.

Pinning: tests/test_oracle_golden.py checks every function here against the golden
vectors in tests/golden/*.npz, which tests/golden/make_golden.py produced by importing
the real reference (/root/reference/src/galaxify) in the build container.
"""
from __future__ import annotations

import time

import numpy as np
import torch


def _f32(x, device="cpu"):
    # reference: simulation.py:58-65 -- torch.tensor(..., dtype=float32) copies
    return torch.tensor(np.asarray(x), dtype=torch.float32, device=device)


def accelerations(pos: torch.Tensor, mass: torch.Tensor, g_const: float, softening: float,
                  block: int = 1024, tgt_slice: slice | None = None) -> torch.Tensor:
    """a_i = G sum_j m_j (r_j - r_i) (|r_j - r_i|^2 + eps^2)^-3/2, diagonal zeroed.

    Follows simulation.py:71-89 op for op: diff (:80), (diff**2).sum(2)+eps**2 (:82),
    pow(-1.5) (:83), fill_diagonal_(0) (:85), (diff*inv*m).sum(1) then G* (:86-88).
    `tgt_slice` restricts the target rows (used by the range-partitioned multi-rank tests).
    """
    n = pos.shape[0]
    lo, hi = (0, n) if tgt_slice is None else (tgt_slice.start, tgt_slice.stop)
    out = torch.empty((hi - lo, 3), dtype=torch.float32)
    eps2 = softening ** 2  # python double, cast to fp32 by torch's scalar add (:82)
    for r0 in range(lo, hi, block):
        r1 = min(r0 + block, hi)
        diff = pos.unsqueeze(0) - pos[r0:r1].unsqueeze(1)          # (rows, N, 3) = r_j - r_i
        dist_sq = (diff ** 2).sum(dim=2) + eps2
        inv = dist_sq.pow(-1.5)
        rows = torch.arange(r0, r1)
        inv[rows - r0, rows] = 0.0                                   # the diagonal entries
        acc = g_const * (diff * inv.unsqueeze(2) * mass.unsqueeze(0).unsqueeze(2)).sum(dim=1)
        out[r0 - lo:r1 - lo] = acc
    return out


def accelerations_f64(pos, mass, g_const, softening, block: int = 1024) -> np.ndarray:
    """Same formula evaluated in float64 (numpy): the accuracy yardstick, not the reference."""
    p = np.asarray(pos, dtype=np.float64)
    m = np.asarray(mass, dtype=np.float64)
    n = p.shape[0]
    out = np.empty((n, 3))
    for r0 in range(0, n, block):
        r1 = min(r0 + block, n)
        d = p[None, :, :] - p[r0:r1, None, :]
        q = (d * d).sum(2) + float(softening) ** 2
        with np.errstate(divide="ignore"):
            inv = q ** -1.5
        inv[np.arange(r1 - r0), np.arange(r0, r1)] = 0.0
        out[r0:r1] = g_const * (d * (inv * m[None, :])[:, :, None]).sum(1)
    return out


def energies(pos: torch.Tensor, vel: torch.Tensor, mass: torch.Tensor, g_const: float,
             softening: float, block: int = 1024) -> tuple[float, float]:
    """(U, K) as simulation.py:91-115: K = sum 0.5 m v^2 (:100-101); U = sum_{i<j}
    -G m_i m_j / (|r_ij| + eps) (:104-113). The reference sums one fp32 (N,N) matrix;
    the row-blocked partial sums here are accumulated in float64, so agreement is to
    fp32 summation error, not bitwise."""
    kinetic = 0.5 * mass * (vel ** 2).sum(dim=1)
    k_energy = kinetic.sum().item()
    n = pos.shape[0]
    u = 0.0
    for r0 in range(0, n, block):
        r1 = min(r0 + block, n)
        diff = pos.unsqueeze(0) - pos[r0:r1].unsqueeze(1)
        dist = (diff ** 2).sum(dim=2).sqrt() + softening
        pot = -g_const * (mass.unsqueeze(0) * mass[r0:r1].unsqueeze(1)) / dist
        cols = torch.arange(n).unsqueeze(0)
        rows = torch.arange(r0, r1).unsqueeze(1)
        u += pot.masked_fill(cols <= rows, 0.0).sum(dtype=torch.float64).item()
    return u, k_energy


class OracleSimulator:
    """State holder mirroring BaseSimulator (simulation.py:21-69): fp32 copies, eager
    initial force evaluation."""

    def __init__(self, *, positions, velocities, masses, g_const=1.0, softening=0.1, dt=0.01,
                 block=1024, initial_accelerations=None):
        self.dt, self.g_const, self.softening, self.block = dt, g_const, softening, block
        self.positions = _f32(positions)
        self.velocities = _f32(velocities)
        self.masses = _f32(masses)
        self.n = self.positions.shape[0]
        # `initial_accelerations`: a timing harness that wants ONE force evaluation per timed step hands in a(t0)
        # (bench.py's whole-step CPU baseline); the reference always evaluates it here (simulation.py:69)
        self.accelerations = self.compute_accelerations() if initial_accelerations is None \
            else _f32(initial_accelerations)

    def compute_accelerations(self):
        return accelerations(self.positions, self.masses, self.g_const, self.softening, self.block)

    def compute_energies(self):
        return energies(self.positions, self.velocities, self.masses, self.g_const,
                        self.softening, self.block)

    def leapfrog_step(self):
        # simulation.py:164-170 (KDK; scalar*tensor then in-place add: two fp32 roundings)
        self.velocities += 0.5 * self.dt * self.accelerations
        self.positions += self.dt * self.velocities
        self.accelerations = self.compute_accelerations()
        self.velocities += 0.5 * self.dt * self.accelerations

    def euler_step(self):
        # simulation.py:183-187
        self.accelerations = self.compute_accelerations()
        self.velocities += self.dt * self.accelerations
        self.positions += self.dt * self.velocities


def time_leapfrog(pos, vel, mass, *, g_const=1.0, softening=0.1, dt=0.01, steps=1, block=1024,
                  threads=None):
    """cpu_baseline helper for bench.py: pair-interactions/s of `steps` leapfrog steps."""
    if threads:
        torch.set_num_threads(threads)
    sim = OracleSimulator(positions=pos, velocities=vel, masses=mass, g_const=g_const,
                          softening=softening, dt=dt, block=block)
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.leapfrog_step()
    dt_s = time.perf_counter() - t0
    return sim.n * sim.n * steps / dt_s, dt_s
