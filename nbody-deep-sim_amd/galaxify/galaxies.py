"""Initial-condition generators with the reference's names and signatures
(src/galaxify/galaxies.py:11-51, 54-67, 195-207) so that `from galaxify import galaxies` keeps
working when this package replaces the reference's on sys.path (s01-dataset-generation.py:7).

Two paths, one random stream. The draws always come from NumPy's legacy global RNG on the host, in exactly
the order the reference consumes it, so a given seed yields the same galaxy (tests/test_galaxies.py checks this
against the golden inputs the real reference generated). What follows the draws runs
  * on the host (default; float64 numpy, the reference's return types): vectorised -- the reference's O(N^2)
    enclosed-mass loop (galaxies.py:143-152) becomes a sort + prefix sum, its per-body Python arithmetic
    (:245-294) array expressions;
  * or on the MI355X with `device="cuda"` (csrc/generators.hip through the C-ABI): the same
    arithmetic in fp64 kernels, the enclosed mass by radix sort + prefix sum + lower-bound search; returns torch
    CUDA float64 tensors that the simulators take as they are. Agrees with the host path to fp64 rounding.
"""
from __future__ import annotations

import enum

import numpy as np


class BodyType(enum.Enum):
    BLACK_HOLE = "black hole"
    STAR = "star"


def spherical_hernquist_distribution(*, r, r0: float = 1, total_mass: float = 1, avoid_distance_zero: bool = True):
    """Hernquist density rho(r) = M/(2 pi) * r0 / (r (r0 + r)^3)  (galaxies.py:11-51)."""
    r = np.asarray(r)
    if avoid_distance_zero:
        r = np.where(r == 0, np.finfo(np.float32).eps, r)
    elif np.any(r == 0):
        raise ValueError("r contiene cero(s) y avoid_distance_zero es False")
    return (total_mass / (2 * np.pi)) * (r0 / (r * (r0 + r) ** 3))


def _euler_rotation(angle) -> np.ndarray:
    """R such that rows transform as v @ R  ==  v @ Rx.T @ Ry.T @ Rz.T  (galaxies.py:160-186)."""
    a, b, c = (float(t) for t in angle)
    rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    return rx.T @ ry.T @ rz.T


def generate_disk(*, n_bodies: int, total_mass: float, radial_scale: float, height_scale: float, g_const: float,
                  black_hole_mass: float, offset=(0, 0, 0), initial_vel=(0, 0, 0), clockwise=True,
                  angle=(0, 0, 0), seed: int = None, device=None):
    """Exponential disc around a central black hole (body 0); returns (positions, velocities, masses)."""
    np.random.seed(seed)
    n = int(n_bodies)
    if device is not None:
        return _disk_on_device(n, total_mass, radial_scale, height_scale, g_const, black_hole_mass, offset, initial_vel,
                               clockwise, angle, device)
    star = np.ones(n, dtype=bool)
    star[0] = False
    # three vector draws, in this order: radius, height, azimuth
    dist = -radial_scale * np.log(1 - np.random.uniform(low=np.finfo(np.float32).eps, high=1.0, size=n))
    dist[0] = 0.0
    z = np.random.uniform(-1.0, 1.0, size=n) * height_scale * (1 - np.sqrt(dist))
    z[0] = 0.0
    phi = np.random.rand(n) * 2 * np.pi
    pos = np.stack((np.cos(phi) * dist, np.sin(phi) * dist, z), axis=1)

    # masses: black hole = fraction of the total, stars weighted by a Hernquist profile of their radius
    m_bh = total_mass * black_hole_mass
    masses = np.empty(n)
    masses[0] = m_bh
    w = spherical_hernquist_distribution(r=dist[star], r0=1, total_mass=total_mass)
    masses[star] = w * ((total_mass - m_bh) / w.sum())

    # circular speed from the mass strictly inside each star's radius: sort once, prefix-sum
    order = np.argsort(dist, kind="stable")
    d_sorted = dist[order]
    csum = np.concatenate(([0.0], np.cumsum(masses[order])))
    m_enc = csum[np.searchsorted(d_sorted, dist, side="left")]
    vel = np.zeros((n, 3))
    with np.errstate(divide="ignore", invalid="ignore"):
        v = np.sqrt(g_const * m_enc[star] / dist[star])
    vel[star, 0] = v * np.cos(phi[star] + np.pi / 2)
    vel[star, 1] = v * np.sin(phi[star] + np.pi / 2)
    if clockwise:
        vel[:, :2] = -vel[:, :2]

    rot = _euler_rotation(angle)
    pos = pos @ rot + np.array(offset)
    vel = vel @ rot + np.array(initial_vel)
    return pos, vel, masses


def generate_spiral(*, n_bodies: int, total_mass: float, radial_scale: float, height_scale: float, g_const: float,
                    black_hole_mass: float, n_arms: int = 2, pitch_angle: float = -np.pi / 6,
                    arm_strength: float = 0.3, seed: int = None, device=None):
    """Spiral-perturbed exponential disc around a central black hole (body 0), equal-mass stars."""
    np.random.seed(seed)
    n = int(n_bodies)
    if device is not None:
        return _spiral_on_device(n, total_mass, radial_scale, height_scale, g_const, black_hole_mass, n_arms,
                                 pitch_angle, arm_strength, device)
    m_bh = total_mass * black_hole_mass
    masses = np.empty(n)
    masses[0] = m_bh
    if n > 1:
        masses[1:] = (total_mass - m_bh) / (n - 1)
    pos = np.zeros((n, 3))
    vel = np.zeros((n, 3))
    if n <= 1:
        return pos, vel, masses

    raw = _spiral_draws(n, radial_scale)
    r, phi = raw[:, 0], 2 * np.pi * raw[:, 1]
    g_z, g_r, g_phi, g_vz = raw[:, 2], raw[:, 3], raw[:, 4], raw[:, 5]

    with np.errstate(divide="ignore", invalid="ignore"):
        swirl = arm_strength * np.sin(n_arms * (phi - np.log(r / radial_scale) / np.tan(pitch_angle)))
    ang = np.where(r > 0, phi + swirl, phi)
    cos_a, sin_a = np.cos(ang), np.sin(ang)
    pos[1:] = np.stack((r * cos_a, r * sin_a, 0.0 + height_scale * g_z), axis=1)

    # circular speed of an exponential disc's enclosed mass, with small anisotropic dispersions
    m_enc = total_mass * (1 - np.exp(-r / radial_scale) * (1 + r / radial_scale))
    with np.errstate(divide="ignore", invalid="ignore"):
        v_circ = np.where(r < 1e-8, 0.0, np.sqrt(g_const * m_enc / r))
    v_r = 0.0 + (0.1 * v_circ) * g_r
    v_phi = v_circ + (0.0 + (0.07 * v_circ) * g_phi)
    v_z = 0.0 + (0.05 * v_circ) * g_vz
    vel[1:] = np.stack((v_r * cos_a - v_phi * sin_a, v_r * sin_a + v_phi * cos_a, v_z), axis=1)
    return pos, vel, masses


def _spiral_draws(n: int, radial_scale: float) -> np.ndarray:
    """The legacy global RNG has to be consumed star by star (gamma is a rejection sampler, so the number of
    underlying draws varies): radius, azimuth, then four unit normals (z, v_R, v_phi, v_z) -- galaxies.py:245-262."""
    raw = np.empty((max(n - 1, 0), 6))
    for row in raw:
        row[0] = np.random.gamma(shape=2, scale=radial_scale)
        row[1] = np.random.rand()
        row[2:] = (np.random.normal(), np.random.normal(), np.random.normal(), np.random.normal())
    return raw


def _device_lib(device):
    import torch
    from nbd import _lib
    if str(device) not in ("cuda",) and not str(device).startswith("cuda:"):
        raise ValueError("device debe ser 'cuda' o None")
    if not torch.cuda.is_available():
        raise RuntimeError("galaxies(device='cuda'): no GPU visible")
    return torch, _lib, torch.device(device)


def _disk_on_device(n, total_mass, radial_scale, height_scale, g_const, black_hole_mass, offset, initial_vel, clockwise,
                    angle, device):
    import ctypes
    torch, _lib, dev = _device_lib(device)
    # the three vector draws, in the reference's order (galaxies.py:98-112)
    u_r = np.random.uniform(low=np.finfo(np.float32).eps, high=1.0, size=n)
    u_z = np.random.uniform(-1.0, 1.0, size=n)
    u_phi = np.random.rand(n)
    d = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
    u_r_d, u_z_d, u_phi_d, rot_d = d(u_r), d(u_z), d(u_phi), d(_euler_rotation(angle))
    pos = torch.empty((n, 3), dtype=torch.float64, device=dev)
    vel = torch.empty((n, 3), dtype=torch.float64, device=dev)
    mass = torch.empty(n, dtype=torch.float64, device=dev)
    if n == 0:
        return pos, vel, mass
    L = _lib.lib()
    ws = torch.empty(L.nbd_disk_workspace_bytes(n), dtype=torch.uint8, device=dev)
    off = (ctypes.c_double * 3)(*[float(t) for t in offset])
    v0 = (ctypes.c_double * 3)(*[float(t) for t in initial_vel])
    with _lib.on_device(dev):
        _lib.check(L.nbd_disk_from_draws_f64(u_r_d.data_ptr(), u_z_d.data_ptr(), u_phi_d.data_ptr(), n, float(total_mass),
                                             float(radial_scale), float(height_scale), float(g_const),
                                             float(black_hole_mass), int(bool(clockwise)), rot_d.data_ptr(), off, v0,
                                             pos.data_ptr(), vel.data_ptr(), mass.data_ptr(), ws.data_ptr(), ws.numel(),
                                             _lib.current_stream(dev)), "nbd_disk_from_draws_f64")
    return pos, vel, mass


def _spiral_on_device(n, total_mass, radial_scale, height_scale, g_const, black_hole_mass, n_arms, pitch_angle,
                      arm_strength, device):
    torch, _lib, dev = _device_lib(device)
    raw = torch.tensor(_spiral_draws(n, radial_scale), dtype=torch.float64, device=dev)
    pos = torch.empty((n, 3), dtype=torch.float64, device=dev)
    vel = torch.empty((n, 3), dtype=torch.float64, device=dev)
    mass = torch.empty(n, dtype=torch.float64, device=dev)
    if n == 0:
        return pos, vel, mass
    with _lib.on_device(dev):
        _lib.check(_lib.lib().nbd_spiral_from_draws_f64(raw.data_ptr() if n > 1 else None, n, float(total_mass),
                                                        float(radial_scale), float(height_scale), float(g_const),
                                                        float(black_hole_mass), int(n_arms), float(pitch_angle),
                                                        float(arm_strength), pos.data_ptr(), vel.data_ptr(),
                                                        mass.data_ptr(), _lib.current_stream(dev)),
                   "nbd_spiral_from_draws_f64")
    return pos, vel, mass
