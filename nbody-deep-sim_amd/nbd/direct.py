"""Typed wrappers over the direct-force entry points of libnbd_hip.so.

Every function takes torch CUDA(=HIP) tensors, checks dtype/shape/contiguity on the host (a
wrong shape must never reach a hand-written kernel) and launches on torch's current stream.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib


def _chk(t: torch.Tensor, shape, name: str, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.NbdError(f"{name}: expected a CUDA/HIP tensor (no CPU fallback on this path)")
    if t.dtype != dtype or not t.is_contiguous():
        raise _lib.NbdError(f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise _lib.NbdError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")


def padded_len(n: int) -> int:
    return _lib.lib().nbd_posm_padded_len(int(n))


def accel_plan(n_src: int, n_tgt: int) -> dict:
    g, s, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(_lib.lib().nbd_accel_plan(n_src, n_tgt, g, s, c), "nbd_accel_plan")
    return {"groups": g.value, "slabs": s.value, "chunks_per_wave": c.value}


def alloc_posm(n: int, device) -> torch.Tensor:
    return torch.empty((padded_len(n), 4), dtype=torch.float32, device=device)


def alloc_bytes(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def pack_posm(pos: torch.Tensor, mass: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    n = pos.shape[0]
    _chk(pos, (n, 3), "pos"); _chk(mass, (n,), "mass")
    if out is None:
        out = alloc_posm(n, pos.device)
    _chk(out, (padded_len(n), 4), "posm")
    with _lib.on_device(pos.device):
        _lib.check(_lib.lib().nbd_pack_posm_f32(pos.data_ptr(), mass.data_ptr(), n, out.data_ptr(),
                                                _lib.current_stream(pos.device)), "nbd_pack_posm_f32")
    return out


def accel_workspace(n_src: int, n_tgt: int, device) -> torch.Tensor:
    return alloc_bytes(_lib.lib().nbd_accel_workspace_bytes(n_src, n_tgt), device)


def accel(posm_src: torch.Tensor, n_src: int, posm_tgt: torch.Tensor, n_tgt: int, tgt_offset: int,
          softening_sq: float, g_const: float, out: torch.Tensor | None = None,
          workspace: torch.Tensor | None = None) -> torch.Tensor:
    """acc (n_tgt,3) of targets posm_tgt[:n_tgt] under sources posm_src[:n_src] (simulation.py:71-89)."""
    _chk(posm_src, (padded_len(n_src), 4), "posm_src") if n_src > 0 else None
    _chk(posm_tgt, None, "posm_tgt")
    if posm_tgt.dim() != 2 or posm_tgt.shape[1] != 4 or posm_tgt.shape[0] < n_tgt:
        raise _lib.NbdError(f"posm_tgt: need >= {n_tgt} rows of 4, got {tuple(posm_tgt.shape)}")
    dev = posm_tgt.device
    if out is None:
        out = torch.empty((n_tgt, 3), dtype=torch.float32, device=dev)
    _chk(out, (n_tgt, 3), "acc_out")
    need = _lib.lib().nbd_accel_workspace_bytes(n_src, n_tgt)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = alloc_bytes(need, dev)
    with _lib.on_device(dev):
        _lib.check(_lib.lib().nbd_accel_f32(
            posm_src.data_ptr() if n_src > 0 else None, n_src, posm_tgt.data_ptr(), n_tgt, tgt_offset,
            float(softening_sq), float(g_const), out.data_ptr(), workspace.data_ptr(),
            workspace.numel() * workspace.element_size(), _lib.current_stream(dev)), "nbd_accel_f32")
    return out


def accel_tuned(posm_src: torch.Tensor, n_src: int, posm_tgt: torch.Tensor, n_tgt: int, tgt_offset: int,
                softening_sq: float, g_const: float, slabs: int, variant: int = 0, exclude=(0, 0),
                out: torch.Tensor | None = None, workspace: torch.Tensor | None = None) -> torch.Tensor:
    """`accel` with an explicit launch geometry and an optional excluded source range (tuning hook)."""
    _chk(posm_src, (padded_len(n_src), 4), "posm_src")
    _chk(posm_tgt, None, "posm_tgt")
    if posm_tgt.dim() != 2 or posm_tgt.shape[1] != 4 or posm_tgt.shape[0] < n_tgt:
        raise _lib.NbdError(f"posm_tgt: need >= {n_tgt} rows of 4, got {tuple(posm_tgt.shape)}")
    dev = posm_tgt.device
    if out is None:
        out = torch.empty((n_tgt, 3), dtype=torch.float32, device=dev)
    _chk(out, (n_tgt, 3), "acc_out")
    need = _lib.lib().nbd_accel_tuned_workspace_bytes(n_tgt, slabs)
    if workspace is None or _nbytes(workspace) < need:
        workspace = alloc_bytes(need, dev)
    with _lib.on_device(dev):
        _lib.check(_lib.lib().nbd_accel_tuned_f32(
            posm_src.data_ptr(), n_src, int(exclude[0]), int(exclude[1]), posm_tgt.data_ptr(), n_tgt, tgt_offset,
            float(softening_sq), float(g_const), out.data_ptr(), workspace.data_ptr(), _nbytes(workspace),
            int(slabs), int(variant), _lib.current_stream(dev)), "nbd_accel_tuned_f32")
    return out


def shard_plan(n_total: int, lo: int, n_local: int) -> dict:
    a, b, c, d = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(_lib.lib().nbd_shard_plan(n_total, lo, n_local, a, b, c, d), "nbd_shard_plan")
    return {"slabs_local": a.value, "chunks_per_wave_local": b.value, "slabs_remote": c.value,
            "chunks_per_wave_remote": d.value}


def shard_workspace(n_total: int, lo: int, n_local: int, device) -> torch.Tensor:
    return alloc_bytes(_lib.lib().nbd_shard_workspace_bytes(n_total, lo, n_local), device)


def shard_force_local(posm_local: torch.Tensor, n_local: int, n_total: int, lo: int, softening_sq: float,
                      workspace: torch.Tensor, uniform=None) -> None:
    """First launch of the sharded force: own bodies as sources (runs while the all-gather is in flight).
    uniform: the bodies' common mass (uniform_mass) -> the kernel without its per-pair mass multiply."""
    _chk(posm_local, (padded_len(n_local), 4), "posm_local")
    with _lib.on_device(posm_local.device):
        fn = _lib.lib().nbd_shard_force_local_f32 if uniform is None else _lib.lib().nbd_shard_force_local_uniform_f32
        _lib.check(fn(
            posm_local.data_ptr(), n_local, float(softening_sq), workspace.data_ptr(), _nbytes(workspace),
            n_total, lo, _lib.current_stream(posm_local.device)), "nbd_shard_force_local_f32")


def shard_force_remote(posm_all: torch.Tensor, n_total: int, posm_local: torch.Tensor, n_local: int, lo: int,
                       softening_sq: float, g_const: float, acc_out: torch.Tensor, vel: torch.Tensor | None,
                       c_kick: float, workspace: torch.Tensor, uniform=None) -> None:
    """Second launch: every other body as a source, then acc = G * sum(slabs) and v += c_kick * acc."""
    _chk(posm_all, (padded_len(n_total), 4), "posm_all")
    _chk(posm_local, (padded_len(n_local), 4), "posm_local")
    _chk(acc_out, (n_local, 3), "acc_out")
    if vel is not None:
        _chk(vel, (n_local, 3), "vel")
    with _lib.on_device(posm_all.device):
        if uniform is not None:
            _lib.check(_lib.lib().nbd_shard_force_remote_uniform_f32(
                posm_all.data_ptr(), n_total, posm_local.data_ptr(), n_local, lo, float(softening_sq),
                float(g_const), float(uniform), acc_out.data_ptr(), _lib.ptr(vel), float(c_kick), workspace.data_ptr(),
                _nbytes(workspace), _lib.current_stream(posm_all.device)), "nbd_shard_force_remote_uniform_f32")
            return
        _lib.check(_lib.lib().nbd_shard_force_remote_f32(
            posm_all.data_ptr(), n_total, posm_local.data_ptr(), n_local, lo, float(softening_sq),
            float(g_const), acc_out.data_ptr(), _lib.ptr(vel), float(c_kick), workspace.data_ptr(),
            _nbytes(workspace), _lib.current_stream(posm_all.device)), "nbd_shard_force_remote_f32")


def f32(x: float) -> float:
    """The value torch uses when a Python double scalar meets an fp32 tensor."""
    return float(np.float32(x))


def _nbytes(t: torch.Tensor) -> int:
    return t.numel() * t.element_size()


def step_workspace(n: int, device) -> torch.Tensor:
    return alloc_bytes(_lib.lib().nbd_step_workspace_bytes(n), device)


def kick_drift(pos, vel, acc, mass, c_kick: float, c_drift: float, posm=None) -> None:
    """v += c_kick*a ; x += c_drift*v ; posm[:n] = {x,m} (simulation.py:164,166). In place."""
    n = pos.shape[0]
    _chk(pos, (n, 3), "pos"); _chk(vel, (n, 3), "vel")
    if acc is not None:
        _chk(acc, (n, 3), "acc")
    if posm is not None:
        _chk(mass, (n,), "mass")
        if posm.dtype != torch.float32 or not posm.is_contiguous() or posm.shape[0] < padded_len(n):
            raise _lib.NbdError("posm: need contiguous fp32 (>= padded_len(n), 4)")
    with _lib.on_device(pos.device):
        _lib.check(_lib.lib().nbd_kick_drift_f32(pos.data_ptr(), vel.data_ptr(), _lib.ptr(acc),
                                                 _lib.ptr(mass), n, c_kick, c_drift, _lib.ptr(posm),
                                                 _lib.current_stream(pos.device)), "nbd_kick_drift_f32")


def kick(vel, acc, c: float) -> None:
    n = vel.shape[0]
    _chk(vel, (n, 3), "vel"); _chk(acc, (n, 3), "acc")
    with _lib.on_device(vel.device):
        _lib.check(_lib.lib().nbd_kick_f32(vel.data_ptr(), acc.data_ptr(), n, c,
                                           _lib.current_stream(vel.device)), "nbd_kick_f32")


def drift(pos, vel, c: float) -> None:
    n = pos.shape[0]
    _chk(pos, (n, 3), "pos"); _chk(vel, (n, 3), "vel")
    with _lib.on_device(pos.device):
        _lib.check(_lib.lib().nbd_drift_f32(pos.data_ptr(), vel.data_ptr(), n, c,
                                            _lib.current_stream(pos.device)), "nbd_drift_f32")


def snapshot(pos, vel, acc, out) -> None:
    """out (3, n, 3) = [pos, vel, acc]: one launch (a slot of run()'s device ring)."""
    n = pos.shape[0]
    _chk(pos, (n, 3), "pos"); _chk(vel, (n, 3), "vel"); _chk(acc, (n, 3), "acc"); _chk(out, (3, n, 3), "out")
    with _lib.on_device(pos.device):
        _lib.check(_lib.lib().nbd_snapshot_f32(pos.data_ptr(), vel.data_ptr(), acc.data_ptr(), n, out.data_ptr(),
                                               _lib.current_stream(pos.device)), "nbd_snapshot_f32")


def uniform_mass(mass: torch.Tensor):
    """The common mass as a Python float when every body has the same, finite, positive mass -- what
    nbd_leapfrog_step_uniform_f32 asks its caller to vouch for -- else None. One device read-back: call it once."""
    if mass.numel() == 0:
        return None
    lo, hi = torch.aminmax(mass)
    lo, hi = float(lo), float(hi)
    return lo if (lo == hi and lo > 0.0 and lo != float("inf")) else None


def leapfrog_step(pos, vel, acc_in, acc_out, mass, dt_half: float, dt: float, softening_sq: float,
                  g_const: float, posm, workspace, ev_begin=None, ev_end=None, uniform=None) -> None:
    """One fused step; ev_begin/ev_end: optional torch.cuda.Event (already recorded once, so the
    handle exists) recorded around the force kernel -- bench.py's roofline hook. uniform: the bodies' common mass
    (uniform_mass(mass)) -> the force kernel without its per-pair mass multiply (nbd_leapfrog_step_uniform_f32)."""
    n = pos.shape[0]
    for t, nm in ((pos, "pos"), (vel, "vel"), (acc_in, "acc_in"), (acc_out, "acc_out")):
        _chk(t, (n, 3), nm)
    _chk(mass, (n,), "mass"); _chk(posm, (padded_len(n), 4), "posm")
    with _lib.on_device(pos.device):
        if uniform is not None:
            _lib.check(_lib.lib().nbd_leapfrog_step_uniform_f32(
                pos.data_ptr(), vel.data_ptr(), acc_in.data_ptr(), acc_out.data_ptr(), mass.data_ptr(), float(uniform), n,
                dt_half, dt, softening_sq, g_const, posm.data_ptr(), workspace.data_ptr(), _nbytes(workspace),
                _lib.current_stream(pos.device),
                None if ev_begin is None else ev_begin.cuda_event,
                None if ev_end is None else ev_end.cuda_event), "nbd_leapfrog_step_uniform_f32")
            return
        _lib.check(_lib.lib().nbd_leapfrog_step_ev_f32(
            pos.data_ptr(), vel.data_ptr(), acc_in.data_ptr(), acc_out.data_ptr(), mass.data_ptr(), n,
            dt_half, dt, softening_sq, g_const, posm.data_ptr(), workspace.data_ptr(), _nbytes(workspace),
            _lib.current_stream(pos.device),
            None if ev_begin is None else ev_begin.cuda_event,
            None if ev_end is None else ev_end.cuda_event), "nbd_leapfrog_step_f32")


def euler_step(pos, vel, acc_out, mass, dt: float, softening_sq: float, g_const: float, posm,
               workspace) -> None:
    n = pos.shape[0]
    for t, nm in ((pos, "pos"), (vel, "vel"), (acc_out, "acc_out")):
        _chk(t, (n, 3), nm)
    _chk(mass, (n,), "mass"); _chk(posm, (padded_len(n), 4), "posm")
    with _lib.on_device(pos.device):
        _lib.check(_lib.lib().nbd_euler_step_f32(
            pos.data_ptr(), vel.data_ptr(), acc_out.data_ptr(), mass.data_ptr(), n, dt, softening_sq,
            g_const, posm.data_ptr(), workspace.data_ptr(), _nbytes(workspace),
            _lib.current_stream(pos.device)), "nbd_euler_step_f32")


def energy(posm, vel, n: int, softening: float, g_const: float, out_uk=None, workspace=None):
    """Device double[2] = {U, K} (simulation.py:91-115); asynchronous."""
    _chk(posm, (padded_len(n), 4), "posm"); _chk(vel, (n, 3), "vel")
    dev = vel.device
    if out_uk is None:
        out_uk = torch.empty(2, dtype=torch.float64, device=dev)
    _chk(out_uk, (2,), "out_uk", torch.float64)
    need = _lib.lib().nbd_energy_workspace_bytes(n)
    if workspace is None or _nbytes(workspace) < need:
        workspace = alloc_bytes(need, dev)
    with _lib.on_device(dev):
        _lib.check(_lib.lib().nbd_energy_f32(posm.data_ptr(), vel.data_ptr(), n, softening, g_const,
                                             out_uk.data_ptr(), workspace.data_ptr(), _nbytes(workspace),
                                             _lib.current_stream(dev)), "nbd_energy_f32")
    return out_uk
