"""MI355X drop-in for the reference integrator module (src/galaxify/simulation.py).

Same public surface -- SimulationState, BaseSimulator, LeapFrogSimulator, EulerSimulator with
the kw-only constructor, attributes (positions, velocities, accelerations, masses, n, dt,
g_const, softening, calc_energy, device) and methods step()/run()/compute_accelerations()/
compute_energies() -- but every O(N^2)/O(N) update runs in hand-written HIP kernels
(csrc/direct_force.hip) through the C-ABI of include/nbd.h. There is no CPU path: device="cpu"
or a missing libnbd_hip.so raises.

Addition over the reference (which has no distributed code): `process_group=` range-partitions
the bodies over the ranks of a torch.distributed group (one process per GPU, RCCL over xGMI);
positions/velocities/accelerations then hold the local shard [lo, hi) and `gather()` assembles
the global arrays. See nbd/dist.py and DESIGN.md.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass

import numpy as np
import torch

from nbd import _lib, direct
from nbd import dist as nbd_dist


@dataclass
class SimulationState:
    """Snapshot of one step; field names and order as simulation.py:8-18."""

    step: int
    step_time: float
    positions: torch.Tensor
    velocities: torch.Tensor
    accelerations: torch.Tensor
    u_energy: float = None
    k_energy: float = None


def _to_f32(x, device) -> torch.Tensor:
    # simulation.py:58-65 makes fp32 device copies with torch.tensor(...)
    if isinstance(x, torch.Tensor):
        return x.detach().to(device=device, dtype=torch.float32, copy=True).contiguous()
    return torch.tensor(np.asarray(x), dtype=torch.float32, device=device).contiguous()


def _resolve_device(device) -> torch.device:
    """Device rule of simulation.py:46-51 ("cuda" is PyTorch-ROCm's name for the MI355X), minus
    the CPU branch: this build has no CPU compute path."""
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("galaxify (MI355X build): no GPU visible and there is no CPU path")
        return torch.device("cuda", torch.cuda.current_device())
    if device == "cuda":
        return torch.device("cuda", torch.cuda.current_device())
    if device == "cpu":
        raise RuntimeError("galaxify (MI355X build): device='cpu' is not provided by this build; "
                           "the HIP kernels are the only compute path (use the reference for CPU)")
    raise ValueError("device debe ser 'cuda', 'cpu' o None")


class BaseSimulator:
    def __init__(self, *, positions, velocities, masses, g_const: float = 1.0, softening: float = 0.1,
                 dt: float = 0.01, calc_energy: bool = True, device: str = None, process_group=None):
        self.device = _resolve_device(device)
        _lib.lib()  # fail now, loudly, if the extension is not built

        self.dt = dt
        self.g_const = g_const
        self.softening = softening
        self.calc_energy = calc_energy
        # fp32 scalars exactly as torch forms them from the Python doubles (simulation.py:82,88,164)
        self._eps2 = direct.f32(softening ** 2)
        self._g = direct.f32(g_const)

        full_pos = _to_f32(positions, self.device)
        full_vel = _to_f32(velocities, self.device)
        self.masses = _to_f32(masses, self.device)
        self.n = full_pos.shape[0]
        if full_pos.shape != (self.n, 3) or full_vel.shape != (self.n, 3) or self.masses.shape != (self.n,):
            raise ValueError("positions/velocities must be (n,3) and masses (n,)")

        world, rank = nbd_dist.group_info(process_group) if process_group is not None else (1, 0)
        self.process_group = process_group
        self.part = nbd_dist.RangePartition(self.n, world, rank)
        lo, hi = self.part.lo, self.part.hi
        # the range-sharded code path: more than one rank, or a one-rank group with NBD_FORCE_SHARDED=1 (rehearsal
        # of the RCCL calls, the asynchronous gather and the split force on a single GPU)
        self._sharded = world > 1 or (process_group is not None and os.environ.get("NBD_FORCE_SHARDED") == "1")
        self.positions = full_pos[lo:hi].clone() if self._sharded else full_pos
        self.velocities = full_vel[lo:hi].clone() if self._sharded else full_vel
        self.accelerations = None

        # scratch owned by the simulator: packed sources (all ranks' bodies), slabs, energy partials
        self._posm = direct.alloc_posm(self.n, self.device)
        self._posm.zero_()
        # equal masses (the published configurations): the force kernel without its per-pair mass multiply, the common
        # factor applied once to the finished sum -- in the fused leapfrog step and in both launches of the
        # range-sharded force. Checked here, once (the masses are replicated on every rank: all ranks decide alike);
        # NBD_UNIFORM_MASS=0 keeps the general kernel. The un-sharded Euler step and compute_accelerations() use the
        # general kernel.
        self._uniform = (direct.uniform_mass(self.masses)
                         if ((self._sharded or isinstance(self, LeapFrogSimulator))
                             and os.environ.get("NBD_UNIFORM_MASS", "1") != "0") else None)
        if not self._sharded:
            self._ws = direct.step_workspace(max(self.n, 1), self.device)
            self._posm_local, self._mass_local, self._gather = self._posm, self.masses, None
        else:
            # the rank's own packed bodies: source of its local force block and send buffer of the gather
            # (max_count rows so that ragged shards send equal, zero-padded pieces)
            self._posm_local = direct.alloc_posm(self.part.max_count, self.device)
            self._posm_local.zero_()
            self._mass_local = self.masses[lo:hi].contiguous()
            self._ws = direct.shard_workspace(self.n, lo, self.part.n_local, self.device) \
                if self.part.n_local else None
            self._gather = nbd_dist.RowGather(self.part, 4, torch.float32, self.device, process_group, collective=True)

        self.accelerations = self.compute_accelerations()

    # ------------------------------------------------------------------ force
    def _refresh_sources(self):
        """posm[:n] = {x,y,z,m} of ALL bodies in global order (one all-gather when sharded)."""
        if not self._sharded:
            direct.pack_posm(self.positions, self.masses, out=self._posm)
        else:
            self._pack_local()
            self._gather.finish(self._gather.start(self._posm_local, self._posm), self._posm)

    def _pack_local(self):
        n_loc = self.part.n_local
        if n_loc:
            direct.pack_posm(self.positions, self._mass_local, out=self._posm_local[:direct.padded_len(n_loc)])

    def _force_sharded(self, vel=None, c_kick: float = 0.0) -> torch.Tensor:
        """Force on the rank's bodies from `_posm_local` (already packed): start the all-gather, run the
        local x local block while it is in flight, then the remote block + slab sum (+ fused kick)."""
        p = self.part
        handle = self._gather.start(self._posm_local, self._posm)
        acc = torch.empty((p.n_local, 3), dtype=torch.float32, device=self.device)
        if p.n_local:
            local = self._posm_local[:direct.padded_len(p.n_local)]
            direct.shard_force_local(local, p.n_local, self.n, p.lo, self._eps2, self._ws, uniform=self._uniform)
        self._gather.finish(handle, self._posm)
        if p.n_local:
            direct.shard_force_remote(self._posm, self.n, local, p.n_local, p.lo, self._eps2, self._g, acc,
                                      vel, c_kick, self._ws, uniform=self._uniform)
        return acc

    def compute_accelerations(self) -> torch.Tensor:
        """a_i = G sum_{j!=i} m_j (r_j - r_i)/(|r_j - r_i|^2 + eps^2)^(3/2) -> new (n_local,3) tensor
        (simulation.py:71-89)."""
        if self.n == 0:
            return torch.zeros((0, 3), dtype=torch.float32, device=self.device)
        if not self._sharded:
            direct.pack_posm(self.positions, self.masses, out=self._posm)
            return direct.accel(self._posm, self.n, self._posm, self.n, 0, self._eps2, self._g,
                                workspace=self._ws)
        self._pack_local()
        return self._force_sharded()

    def compute_energies(self):
        """(U, K) as Python floats (simulation.py:91-115). Sharded: every rank evaluates the
        global sums from the gathered state (velocities are gathered for this call)."""
        if self.n == 0:
            return 0.0, 0.0
        self._refresh_sources()
        vel = self.gather("velocities") if self._sharded else self.velocities
        uk = direct.energy(self._posm, vel, self.n, direct.f32(self.softening), self._g)
        u, k = uk.cpu().tolist()
        return u, k

    def gather(self, name: str) -> torch.Tensor:
        """Global (n,3) copy of a sharded state array on every rank ('positions', ...)."""
        local = getattr(self, name)
        if not self._sharded:
            return local
        out = torch.empty((self.n, 3), dtype=torch.float32, device=self.device)
        return nbd_dist.allgather_rows(local, self.part, out, group=self.process_group)

    # ------------------------------------------------------------------ run loop
    def run(self, steps: int) -> list[SimulationState]:
        """Run `steps` steps and return one SimulationState per step (simulation.py:117-146):
        CPU clones of the state after the step, the step's time and (if calc_energy) U and K.
        step_time is the GPU time of step() from HIP events (the reference's un-synchronised
        time.time() bracket would only time the launches)."""
        states = []
        if steps <= 0:
            return states
        if self._graph_run_ok(steps):
            return self._run_graphed(steps)
        return self._run_eager(steps, 0)

    def _run_eager(self, steps: int, first_index: int) -> list[SimulationState]:
        states = []
        n_loc = self.part.n_local
        per_step = 3 * n_loc * 3 * 4
        chunk = max(1, min(max(steps, 32), (64 << 20) // max(per_step, 1)))
        # pinned staging is expensive to create (page-locking): keep it across run() calls
        cached = getattr(self, "_run_stage", None)
        if cached is None or cached[0].shape[0] < chunk:
            cached = (torch.empty((chunk, 3, n_loc, 3), dtype=torch.float32).pin_memory(),
                      torch.empty((chunk, 2), dtype=torch.float64).pin_memory())
            self._run_stage = cached
        stage, uk_host = cached
        chunk = stage.shape[0]
        uk_dev = torch.zeros((chunk, 2), dtype=torch.float64, device=self.device)
        done = 0
        while done < steps:
            m = min(chunk, steps - done)
            events = []
            for s in range(m):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.step()
                e1.record()
                events.append((e0, e1))
                if self.calc_energy:
                    if not self._sharded:
                        # energies of the state AFTER the step (simulation.py:131-133). The leapfrog step
                        # leaves posm = current positions; the Euler step packs before its drift, so repack.
                        if not isinstance(self, LeapFrogSimulator):
                            direct.pack_posm(self.positions, self.masses, out=self._posm)
                        direct.energy(self._posm, self.velocities, self.n, direct.f32(self.softening),
                                      self._g, out_uk=uk_dev[s])
                    else:
                        u, k = self.compute_energies()
                        uk_dev[s, 0], uk_dev[s, 1] = u, k
                stage[s, 0].copy_(self.positions, non_blocking=True)
                stage[s, 1].copy_(self.velocities, non_blocking=True)
                stage[s, 2].copy_(self.accelerations, non_blocking=True)
            uk_host[:m].copy_(uk_dev[:m], non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
            # one pageable copy of the whole chunk (the pinned staging is reused); the states' tensors are
            # views into it -- 3 m small clones cost several times more in allocation and page faults
            host = stage[:m].clone()
            uk = uk_host[:m].tolist()
            for s in range(m):
                u, k = (uk[s][0], uk[s][1]) if self.calc_energy else (None, None)
                states.append(SimulationState(
                    positions=host[s, 0], velocities=host[s, 1],
                    accelerations=host[s, 2], step=first_index + done + s,
                    step_time=events[s][0].elapsed_time(events[s][1]) * 1e-3, u_energy=u, k_energy=k))
            done += m
        return states

    # ------------------------------------------------------------------ run(): captured chunks for small systems
    # Below ~16 k bodies a step is a handful of microseconds of GPU work and run() is bound by its per-step host
    # work (events, seven ctypes launches, three staged copies, a state object): 110-280 us per step wall for 15-25 us
    # of GPU time at the reference's dataset sizes (100-2000 bodies x 1000 steps, s01-dataset-generation.py:192-214).
    # Those systems run in CHUNKS captured into a hipGraph: per step the integrator's launches, the energy launches
    # and ONE snapshot launch into a device ring; per chunk one replay, one sync, one device->host copy. Same
    # kernels in the same order as step(): bit-identical states (tested).
    GRAPH_RUN_MAX_BODIES = 16384
    GRAPH_RUN_CHUNK = 32

    def _graph_run_ok(self, steps: int) -> bool:
        return (not self._sharded and 0 < self.n <= self.GRAPH_RUN_MAX_BODIES and steps >= 8 and
                type(self).step in (LeapFrogSimulator.step, EulerSimulator.step) and
                os.environ.get("NBD_RUN_GRAPH", "1") != "0")

    def _step_in_place(self, acc):
        """One integrator step on (positions, velocities, acc) without rebinding anything (capturable)."""
        dt = direct.f32(self.dt)
        if isinstance(self, LeapFrogSimulator):
            direct.leapfrog_step(self.positions, self.velocities, acc, acc, self.masses, direct.f32(0.5 * self.dt), dt,
                                 self._eps2, self._g, self._posm, self._ws, uniform=self._uniform)
        else:
            direct.euler_step(self.positions, self.velocities, acc, self.masses, dt, self._eps2, self._g, self._posm,
                              self._ws)

    def _chunk_graph(self, m: int):
        """(graph, ring, uk) for a chunk of m steps; captured once per m and kept."""
        cache = self.__dict__.setdefault("_run_graphs", {})
        # a graph bakes in buffer addresses and scalar arguments: anything the caller may have changed is in the key
        key = (m, self.positions.data_ptr(), self.velocities.data_ptr(), self.masses.data_ptr(), float(self.dt),
               bool(self.calc_energy))
        if key in cache:
            return cache[key]
        if len(cache) > 8:
            cache.clear()
        n, dev = self.n, self.device
        if getattr(self, "_acc_g", None) is None:
            self._acc_g = torch.empty((n, 3), dtype=torch.float32, device=dev)
            self._energy_ws = direct.alloc_bytes(_lib.lib().nbd_energy_workspace_bytes(n), dev)
        ring = torch.empty((m, 3, n, 3), dtype=torch.float32, device=dev)
        uk = torch.zeros((m, 2), dtype=torch.float64, device=dev)
        soft = direct.f32(self.softening)
        leap = isinstance(self, LeapFrogSimulator)

        def body(count=m):
            for s_ in range(count):
                self._step_in_place(self._acc_g)
                if self.calc_energy:
                    if not leap:     # the Euler step packs before its drift: energies need the moved positions
                        direct.pack_posm(self.positions, self.masses, out=self._posm)
                    direct.energy(self._posm, self.velocities, n, soft, self._g, out_uk=uk[s_], workspace=self._energy_ws)
                direct.snapshot(self.positions, self.velocities, self._acc_g, ring[s_])
        # capture on a side stream; the state is saved and restored around the (executed) warm-up pass
        keep = (self.positions.clone(), self.velocities.clone(), self._acc_g.clone())
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            body(1)                                          # every kernel of a step once (lazy initialisations)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.positions.copy_(keep[0]); self.velocities.copy_(keep[1]); self._acc_g.copy_(keep[2])
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body()
        self.positions.copy_(keep[0]); self.velocities.copy_(keep[1]); self._acc_g.copy_(keep[2])
        cache[key] = (graph, ring, uk)
        return cache[key]

    def _run_graphed(self, steps: int) -> list[SimulationState]:
        n, dev = self.n, self.device
        states, done = [], 0
        first = True
        while steps - done >= 8:                             # chunks of 32, then of 8; the last < 8 steps run eagerly
            m = self.GRAPH_RUN_CHUNK if steps - done >= self.GRAPH_RUN_CHUNK else 8
            graph, ring, uk = self._chunk_graph(m)          # (capture leaves the state untouched)
            if first:                                        # a caller's handle on the old accelerations stays valid
                self._acc_g.copy_(self.accelerations)
                first = False
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            graph.replay()
            e1.record()
            host = ring.cpu()                                # one device->host copy per chunk (synchronises)
            uk_h = uk.cpu().tolist() if self.calc_energy else None
            t_step = e0.elapsed_time(e1) * 1e-3 / m          # GPU time of the chunk, spread over its steps
            for s_ in range(m):
                u, k = (uk_h[s_][0], uk_h[s_][1]) if self.calc_energy else (None, None)
                states.append(SimulationState(step=done + s_, step_time=t_step, positions=host[s_, 0],
                                              velocities=host[s_, 1], accelerations=host[s_, 2], u_energy=u, k_energy=k))
            done += m
        self.accelerations = self._acc_g.clone()             # rebound, as step() does (simulation.py:168)
        if done < steps:
            states += self._run_eager(steps - done, done)
        return states

    def step(self):
        raise NotImplementedError("El método step debe ser implementado en la subclase")


class LeapFrogSimulator(BaseSimulator):
    def step(self):
        """Kick-drift-kick (simulation.py:153-170); one force evaluation per step; positions and
        velocities are updated in place, `accelerations` is rebound to a new tensor (:168)."""
        if self.n == 0:
            return
        half = direct.f32(0.5 * self.dt)
        dt = direct.f32(self.dt)
        if not self._sharded:
            new_acc = torch.empty_like(self.accelerations)
            direct.leapfrog_step(self.positions, self.velocities, self.accelerations, new_acc, self.masses,
                                 half, dt, self._eps2, self._g, self._posm, self._ws, uniform=self._uniform)
            self.accelerations = new_acc
            return
        # sharded: kick+drift+pack of the own bodies -> all-gather in flight || own x own force block ->
        # own x remote block + slab sum + second kick. Four launches and one collective.
        if self._step_graph is not None:                 # capture_step(): the same launches and the collective, replayed
            self._step_graph.replay()
            return
        if self.part.n_local:
            direct.kick_drift(self.positions, self.velocities, self.accelerations, self._mass_local, half, dt,
                              posm=self._posm_local)
        self.accelerations = self._force_sharded(self.velocities, half)

    _step_graph = None

    def capture_step(self, warmup: int = 3) -> bool:
        """Capture the range-sharded leapfrog step -- kick-drift, the all-gather, both force blocks, the second kick -- into
        ONE hipGraph (torch's NCCL backend = RCCL records its collective into a capturing stream): a rank's step at the
        strong-scaled size is ~136 us of kernels behind ~28 us of host enqueue (profiles/r02_shard_rank_of_8.json), and the
        enqueue is what a replay removes. `accelerations` becomes a static buffer (the graph copies the new values into
        it: 12 B per body). Returns False -- and step() stays eager -- when the runtime refuses the capture; every rank
        must call this together (the warm-up steps run the collective)."""
        if not self._sharded or self.n == 0 or self._step_graph is not None:
            return self._step_graph is not None
        try:
            for _ in range(warmup):                      # communicator, allocator pools, hipFuncSetAttribute calls
                self.step()
            torch.cuda.synchronize(self.device)
            half, dt = direct.f32(0.5 * self.dt), direct.f32(self.dt)
            acc_static = self.accelerations.clone()
            self.accelerations = acc_static
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    if self.part.n_local:
                        direct.kick_drift(self.positions, self.velocities, acc_static, self._mass_local, half, dt,
                                          posm=self._posm_local)
                    new_acc = self._force_sharded(self.velocities, half)
                    acc_static.copy_(new_acc)
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            self._step_graph, self._step_graph_keep = graph, (new_acc, acc_static, side)
            return True
        except Exception as exc:                          # pragma: no cover - depends on the runtime's capture support
            import warnings
            warnings.warn(f"hipGraph capture of the range-sharded step failed ({exc}); the step stays eager")
            try:
                torch.cuda.synchronize(self.device)
            except Exception:
                pass
            self._step_graph = None
            return False

    def step_phases(self):
        """One EAGER range-sharded step with HIP events between its phases: {"local_force_ms", "gather_wait_ms",
        "remote_force_ms", "kick_drift_ms", "host_enqueue_ms"} -- gather_wait = the launch stream idle between the end of
        the own x own block and the completion of the all-gather (what the overlap does not cover). Synchronises."""
        if not self._sharded or self.n == 0:
            raise _lib.NbdError("step_phases(): only the range-sharded step has phases")
        half, dt = direct.f32(0.5 * self.dt), direct.f32(self.dt)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        p = self.part
        t0 = time.perf_counter()
        ev[0].record()
        if p.n_local:
            direct.kick_drift(self.positions, self.velocities, self.accelerations, self._mass_local, half, dt,
                              posm=self._posm_local)
        handle = self._gather.start(self._posm_local, self._posm)
        acc = torch.empty((p.n_local, 3), dtype=torch.float32, device=self.device)
        ev[1].record()
        local = self._posm_local[:direct.padded_len(p.n_local)]
        if p.n_local:
            direct.shard_force_local(local, p.n_local, self.n, p.lo, self._eps2, self._ws, uniform=self._uniform)
        ev[2].record()
        self._gather.finish(handle, self._posm)
        ev[3].record()
        if p.n_local:
            direct.shard_force_remote(self._posm, self.n, local, p.n_local, p.lo, self._eps2, self._g, acc,
                                      self.velocities, half, self._ws, uniform=self._uniform)
        ev[4].record()
        host = (time.perf_counter() - t0) * 1e3
        if self._step_graph is not None:
            self.accelerations.copy_(acc)                # the captured step's static buffer
        else:
            self.accelerations = acc
        torch.cuda.synchronize(self.device)
        return {"kick_drift_ms": ev[0].elapsed_time(ev[1]), "local_force_ms": ev[1].elapsed_time(ev[2]),
                "gather_wait_ms": ev[2].elapsed_time(ev[3]), "remote_force_ms": ev[3].elapsed_time(ev[4]),
                "host_enqueue_ms": host}


class EulerSimulator(BaseSimulator):
    def step(self):
        """a(t) -> v += dt a -> x += dt v (simulation.py:173-187)."""
        if self.n == 0:
            return
        dt = direct.f32(self.dt)
        if not self._sharded:
            new_acc = torch.empty_like(self.accelerations)
            direct.euler_step(self.positions, self.velocities, new_acc, self.masses, dt, self._eps2,
                              self._g, self._posm, self._ws)
            self.accelerations = new_acc
            return
        self._pack_local()
        self.accelerations = self._force_sharded(self.velocities, dt)        # a(t), v += dt a fused
        if self.part.n_local:
            direct.drift(self.positions, self.velocities, dt)
