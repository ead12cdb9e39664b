"""Parity of the HIP direct-force path (through the C-ABI) against (i) the golden vectors the
real reference produced and (ii) the pinned CPU oracle on seeded inputs; plus size-independent
properties at BASELINE.json's full size. Tolerance (BASELINE.json north_star): positions and
velocities within 1e-5 relative after one step, measured per particle on vector norms
(SURVEY 8c); accelerations are held to the same bound and to 1e-6 globally."""
import os

import numpy as np
import pytest
import torch

from conftest import global_rel, golden_cases, load_golden, row_rel

pytestmark = pytest.mark.gpu

TOL = 1e-5          # per-particle relative (north_star)
TOL_ACC_GLOBAL = 1e-6


def _mk(cls, g, **over):
    from galaxify import simulation
    kw = dict(positions=g["pos"], velocities=g["vel"], masses=g["mass"], g_const=float(g["g_const"]),
              softening=float(g["softening"]), dt=float(g["dt"]), calc_energy=True, device="cuda")
    kw.update(over)
    return getattr(simulation, cls)(**kw)


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("name", golden_cases())
def test_golden_initial_acceleration(name, gpu_device):
    g = load_golden(name)
    sim = _mk("LeapFrogSimulator", g)
    acc = _np(sim.accelerations)
    assert np.isfinite(acc).all()
    assert row_rel(acc, g["acc0"]) < TOL
    assert global_rel(acc, g["acc0"]) < TOL_ACC_GLOBAL


@pytest.mark.parametrize("name", golden_cases())
def test_golden_leapfrog_one_step(name, gpu_device):
    g = load_golden(name)
    sim = _mk("LeapFrogSimulator", g)
    old_acc = sim.accelerations
    pos_ptr = sim.positions.data_ptr()
    sim.step()
    assert sim.positions.data_ptr() == pos_ptr            # in place (simulation.py:166)
    assert sim.accelerations is not old_acc               # rebound (simulation.py:168)
    for key, t in (("pos", sim.positions), ("vel", sim.velocities), ("acc", sim.accelerations)):
        assert row_rel(_np(t), g[f"lf1_{key}"]) < TOL, key


@pytest.mark.parametrize("name", [n for n in golden_cases() if "4096" not in n])
def test_golden_leapfrog_ten_steps(name, gpu_device):
    g = load_golden(name)
    sim = _mk("LeapFrogSimulator", g)
    for _ in range(10):
        sim.step()
    for key, t in (("pos", sim.positions), ("vel", sim.velocities), ("acc", sim.accelerations)):
        assert row_rel(_np(t), g[f"lf10_{key}"]) < 10 * TOL, key


@pytest.mark.parametrize("name", golden_cases())
def test_golden_euler_one_step(name, gpu_device):
    g = load_golden(name)
    sim = _mk("EulerSimulator", g)
    sim.step()
    for key, t in (("pos", sim.positions), ("vel", sim.velocities), ("acc", sim.accelerations)):
        assert row_rel(_np(t), g[f"eu1_{key}"]) < TOL, key


@pytest.mark.parametrize("name", golden_cases())
def test_golden_energies(name, gpu_device):
    g = load_golden(name)
    u, k = _mk("LeapFrogSimulator", g).compute_energies()
    u0, k0 = g["energy0"]
    assert abs(u - u0) <= 2e-5 * abs(u0) + 1e-30
    assert abs(k - k0) <= 2e-6 * abs(k0) + 1e-30


def test_kick_drift_bit_exact_given_same_acceleration(gpu_device):
    """The O(N) updates are two-rounding mul+add like torch eager: given identical a(t), the
    drifted positions and half-kicked velocities match the reference BIT FOR BIT."""
    from nbd import direct
    g = load_golden("direct_plummer_n1000")
    dt = float(g["dt"])
    pos, vel, acc = (torch.tensor(g[k]) for k in ("pos", "vel", "acc0"))
    v_ref = vel + 0.5 * dt * acc          # torch CPU eager = the reference's arithmetic
    x_ref = pos + dt * v_ref
    p, v, a = pos.cuda(), vel.cuda(), acc.cuda()
    m = torch.tensor(g["mass"]).cuda()
    posm = direct.alloc_posm(1000, gpu_device)
    direct.kick_drift(p, v, a, m, direct.f32(0.5 * dt), direct.f32(dt), posm=posm)
    assert torch.equal(v.cpu(), v_ref) and torch.equal(p.cpu(), x_ref)
    assert torch.equal(posm[:1000, :3].cpu(), x_ref) and torch.equal(posm[:1000, 3].cpu(), m.cpu())
    assert (posm[1000:] == 0).all()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 129, 500, 2048, 5000])
def test_ragged_sizes_against_oracle(n, gpu_device):
    from nbd.plummer import generate_plummer
    from oracle import galaxify_oracle as go
    p, v, m = generate_plummer(n, seed=100 + n)
    rng = np.random.default_rng(n)
    m = m * rng.uniform(0.5, 2.0, n)
    g = dict(pos=p, vel=v, mass=m, g_const=1.0, softening=0.1, dt=0.01)
    sim = _mk("LeapFrogSimulator", g)
    ora = go.OracleSimulator(positions=p, velocities=v, masses=m)
    sim.step(); ora.leapfrog_step()
    for a, b in ((sim.positions, ora.positions), (sim.velocities, ora.velocities),
                 (sim.accelerations, ora.accelerations)):
        assert row_rel(_np(a), b.numpy()) < TOL


def test_empty_system(gpu_device):
    z = np.zeros((0, 3))
    sim = _mk("LeapFrogSimulator", dict(pos=z, vel=z, mass=np.zeros(0), g_const=1.0, softening=0.1, dt=0.01))
    sim.step()
    assert sim.accelerations.shape == (0, 3) and sim.run(2)[1].positions.shape == (0, 3)


def test_eps0_coincident_bodies_give_nan_like_reference(gpu_device):
    """softening = 0: only the diagonal is masked (fill_diagonal_, simulation.py:85); two distinct
    bodies at the same point give 0*inf = NaN in the reference, and here."""
    pos = np.array([[0., 0, 0], [1, 0, 0], [1, 0, 0], [0, 2, 0]])
    sim = _mk("LeapFrogSimulator", dict(pos=pos, vel=np.zeros((4, 3)), mass=np.ones(4), g_const=1.0,
                                        softening=0.0, dt=0.01))
    acc = _np(sim.accelerations)
    assert np.isnan(acc[1]).any() and np.isnan(acc[2]).any()
    assert np.isfinite(acc[0]).all() and np.isfinite(acc[3]).all()


def test_tiny_softening_uses_index_mask(gpu_device):
    """softening**2 underflows/overflows rsq^3 on the diagonal: must still match the oracle."""
    from nbd.plummer import generate_plummer
    from oracle import galaxify_oracle as go
    p, v, m = generate_plummer(200, seed=3)
    for eps in (1e-14, 1e-30):
        sim = _mk("LeapFrogSimulator", dict(pos=p, vel=v, mass=m, g_const=1.0, softening=eps, dt=0.01))
        ref = go.accelerations(torch.tensor(p, dtype=torch.float32), torch.tensor(m, dtype=torch.float32), 1.0, eps)
        assert row_rel(_np(sim.accelerations), ref.numpy()) < TOL


def test_deterministic_and_partition_invariant(gpu_device):
    """Same inputs -> bit-identical output; a target sub-range with tgt_global_offset equals the
    same rows of the full evaluation bit for bit when the slab plan is the same, and to rounding
    otherwise (this is the multi-GPU range partition on one device)."""
    from nbd import direct
    from nbd.plummer import generate_plummer
    n = 4096
    p, v, m = generate_plummer(n, seed=11)
    pos = torch.tensor(p, dtype=torch.float32).cuda(); mass = torch.tensor(m, dtype=torch.float32).cuda()
    posm = direct.pack_posm(pos, mass)
    a1 = direct.accel(posm, n, posm, n, 0, 0.01, 1.0)
    a2 = direct.accel(posm, n, posm, n, 0, 0.01, 1.0)
    assert torch.equal(a1, a2)
    parts = []
    for r in range(4):
        lo = r * 1024
        parts.append(direct.accel(posm, n, posm[lo:], 1024, lo, 0.01, 1.0))
    assert row_rel(_np(torch.cat(parts)), _np(a1)) < 2e-6
    # masked kernel: the offset decides which pair is the diagonal
    a0 = direct.accel(posm, n, posm, n, 0, 0.0, 1.0)
    b = torch.cat([direct.accel(posm, n, posm[r * 1024:], 1024, r * 1024, 0.0, 1.0) for r in range(4)])
    assert torch.isfinite(a0).all() and row_rel(_np(b), _np(a0)) < 2e-6


def test_uniform_mass_leapfrog_step_matches_general_step_and_f64(gpu_device, monkeypatch):
    """nbd_leapfrog_step_uniform_f32 (equal masses: the factor applied once to the finished sum, no per-pair multiply)
    against the general fused step on the same state and against an fp64 evaluation; N a multiple of 64 and not (the
    padded tail chunk takes the masked path); tiny softening (index-masked kernel) too; bit-identical run to run; the
    simulator picks it only for equal, positive masses and NBD_UNIFORM_MASS=0 switches it off."""
    from galaxify import simulation
    from nbd.plummer import generate_plummer
    for n, eps in ((4096, 0.05), (1000, 0.01), (193, 0.1), (130, 0.0)):
        p, v, m = generate_plummer(n, seed=21 + n)
        assert m.min() == m.max()
        kw = dict(positions=p, velocities=v, masses=m, g_const=0.7, softening=eps, dt=0.01, calc_energy=False, device="cuda")
        a = simulation.LeapFrogSimulator(**kw)
        a2 = simulation.LeapFrogSimulator(**kw)
        monkeypatch.setenv("NBD_UNIFORM_MASS", "0")
        b = simulation.LeapFrogSimulator(**kw)
        monkeypatch.delenv("NBD_UNIFORM_MASS")
        assert a._uniform == float(np.float32(m[0])) and b._uniform is None
        for _ in range(3):
            a.step(); a2.step(); b.step()
        for key in ("positions", "velocities", "accelerations"):
            assert torch.equal(getattr(a, key), getattr(a2, key)), key
            assert row_rel(_np(getattr(a, key)), _np(getattr(b, key))) < 2e-6, (n, key)
        x = a.positions.double().cpu().numpy()
        d = x[None, :, :] - x[:, None, :]
        r2 = (d ** 2).sum(-1) + eps ** 2
        np.fill_diagonal(r2, 1.0)
        w = float(m[0]) * r2 ** -1.5
        np.fill_diagonal(w, 0.0)
        ref = 0.7 * (w[:, :, None] * d).sum(1)
        assert row_rel(_np(a.accelerations), ref) < 2e-6 and row_rel(_np(b.accelerations), ref) < 2e-6
    p, v, m = generate_plummer(256, seed=2)
    m2 = m.copy(); m2[3] *= 2
    assert simulation.LeapFrogSimulator(positions=p, velocities=v, masses=m2, device="cuda", calc_energy=False)._uniform is None
    assert simulation.LeapFrogSimulator(positions=p, velocities=v, masses=0 * m, device="cuda", calc_energy=False)._uniform is None
    assert simulation.EulerSimulator(positions=p, velocities=v, masses=m, device="cuda", calc_energy=False)._uniform is None


def test_linearity_in_mass_and_g(gpu_device):
    """a is linear in the source masses and in G: size-independent property, no oracle needed."""
    from nbd import direct
    from nbd.plummer import generate_plummer
    n = 65536
    p, v, m = generate_plummer(n, seed=1234)
    pos = torch.tensor(p, dtype=torch.float32).cuda(); m1 = torch.tensor(m, dtype=torch.float32).cuda()
    a = direct.accel(direct.pack_posm(pos, m1), n, direct.pack_posm(pos, m1), n, 0, 0.01, 1.0)
    b = direct.accel(direct.pack_posm(pos, 2 * m1), n, direct.pack_posm(pos, m1), n, 0, 0.01, 0.5)
    assert torch.equal(a, b)       # scaling by powers of two is exact in fp32


def test_full_size_momentum_and_rows_against_f64(gpu_device):
    """BASELINE config 2 (N = 65 536 Plummer): total momentum change sum_i m_i a_i vanishes
    (Newton's third law) to fp32 summation noise, and 64 sampled rows match an fp64 evaluation."""
    from nbd.plummer import generate_plummer
    n = 65536
    p, v, m = generate_plummer(n, seed=1234)
    sim = _mk("LeapFrogSimulator", dict(pos=p, vel=v, mass=m, g_const=1.0, softening=0.1, dt=0.01))
    acc = _np(sim.accelerations).astype(np.float64)
    pf, mf = p.astype(np.float32).astype(np.float64), m.astype(np.float32).astype(np.float64)
    net = (mf[:, None] * acc).sum(0)
    assert np.abs(net).max() < 1e-6 * (mf[:, None] * np.abs(acc)).sum(0).max()
    rows = np.random.default_rng(0).choice(n, 64, replace=False)
    d = pf[None, :, :] - pf[rows, None, :]
    q = (d * d).sum(2) + float(np.float32(0.1 ** 2))
    inv = q ** -1.5
    inv[np.arange(64), rows] = 0.0
    ref = (d * (inv * mf[None, :])[:, :, None]).sum(1)
    assert row_rel(acc[rows], ref) < 2e-6
    # one full step stays finite and moves positions by O(dt*v)
    x0 = _np(sim.positions).copy()
    sim.step()
    dx = np.linalg.norm(_np(sim.positions) - x0, axis=1)
    assert np.isfinite(dx).all() and 1e-4 < np.median(dx) < 1e-2


def test_run_returns_reference_shaped_states(gpu_device):
    g = load_golden("direct_plummer_n64")
    sim = _mk("LeapFrogSimulator", g)
    states = sim.run(10)
    assert len(states) == 10 and [s.step for s in states] == list(range(10))
    s = states[-1]
    assert s.positions.device.type == "cpu" and s.positions.dtype == torch.float32
    assert isinstance(s.u_energy, float) and isinstance(s.k_energy, float) and s.step_time > 0
    assert row_rel(s.positions.numpy(), g["lf10_pos"]) < 10 * TOL
    assert row_rel(states[0].positions.numpy(), g["lf1_pos"]) < TOL
    # energy is that of the state after the step; drift of total energy over 10 steps is small
    e = [st.u_energy + st.k_energy for st in states]
    assert abs(e[-1] - e[0]) < 1e-3 * abs(e[0])
    sim2 = _mk("LeapFrogSimulator", g, calc_energy=False)
    assert sim2.run(2)[0].u_energy is None


def test_base_step_not_implemented(gpu_device):
    g = load_golden("direct_spiral_n3")
    with pytest.raises(NotImplementedError):
        _mk("BaseSimulator", g).step()


def test_tensor_inputs_are_copied(gpu_device):
    g = load_golden("direct_spiral_n25")
    pos = torch.tensor(g["pos"], device="cuda")
    sim = _mk("LeapFrogSimulator", g, positions=pos)
    sim.step()
    assert torch.equal(pos.cpu(), torch.tensor(g["pos"]))       # caller's tensor untouched


def _sharded_worker(rank, world, port, n, steps, out_dir):
    import os
    import sys
    import torch.distributed as dist
    from conftest import PKG, ROOT
    for p in (PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        from galaxify import simulation
        from nbd.plummer import generate_plummer
        p, v, m = generate_plummer(n, seed=77)
        m = m * np.random.default_rng(1).uniform(0.5, 2.0, n)
        sim = simulation.LeapFrogSimulator(positions=p, velocities=v, masses=m, dt=0.01, calc_energy=True,
                                           device="cuda", process_group=dist.group.WORLD)
        for _ in range(steps):
            sim.step()
        u, k = sim.compute_energies()
        full = {key: sim.gather(key).cpu().numpy() for key in ("positions", "velocities", "accelerations")}
        if rank == 0:
            np.savez(os.path.join(out_dir, "sharded.npz"), u=u, k=k, **full)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [1024, 1001])
def test_two_rank_range_partition_on_gpu_matches_single_rank(n, tmp_path, gpu_device):
    """The real sharded path (HIP kernels, tgt_global_offset, one all-gather per step) with two
    processes sharing this GPU over gloo (RCCL needs distinct devices) vs the un-sharded simulator."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    steps = 3
    mp.spawn(_sharded_worker, args=(2, port, n, steps, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "sharded.npz")
    from nbd.plummer import generate_plummer
    p, v, m = generate_plummer(n, seed=77)
    m = m * np.random.default_rng(1).uniform(0.5, 2.0, n)
    sim = _mk("LeapFrogSimulator", dict(pos=p, vel=v, mass=m, g_const=1.0, softening=0.1, dt=0.01))
    for _ in range(steps):
        sim.step()
    for key, ref in (("positions", sim.positions), ("velocities", sim.velocities), ("accelerations", sim.accelerations)):
        assert row_rel(got[key], _np(ref)) < 2e-6, key
    u, k = sim.compute_energies()
    assert abs(got["u"] - u) < 1e-6 * abs(u) and abs(got["k"] - k) < 1e-6 * abs(k)


def test_bench_py_multi_rank_json_end_to_end(gpu_device, tmp_path):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, one process per rank), on THIS box:
    the ranks share the one GPU and talk over gloo (NBD_BENCH_SHARE_GPU / NBD_DIST_BACKEND; RCCL needs distinct
    devices). Four ranks, not eight: the GPU boxes of this pool kill a run with more than six processes on the card
    (the eight-rank control flow is covered on CPU stand-ins in test_dist_gloo.py). Checks the ONE JSON line: rank
    count as launched and as seen by the collective, weak-scaling bookkeeping, the strong leg, sane numbers."""
    import json
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world, per = 4, 2048
    env = dict(os.environ, NBD_BENCH_SHARE_GPU="1", NBD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world),
           "--steps", "5", "--warmup", "2", "--particles-per-gpu", str(per), "--prewarm-seconds", "0.05",
           "--min-timed-seconds", "0.05", "--repeats", "3", "--cpu-seconds", "0", "--no-surrogates"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                     # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["ranks_seen"] == world
    assert out["scaling"] == "weak" and out["steps"] == 5 and out["warmup"] == 2 and out["repeats"] >= 3
    assert out["config"]["n_particles"] == world * per and out["config"]["particles_per_gpu"] == per
    assert out["value"] > 0 and abs(out["value"] - (world * per) ** 2 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    lo, med, hi = out["ms_per_step_min_median_max"]
    assert lo <= med <= hi and med == out["ms_per_step"]
    strong = out["strong_scaling_n65536"]                          # the single-GPU problem size split over all ranks
    assert strong["n_particles"] == per and strong["scaling"] == "strong" and strong["value"] > 0
    assert out["roofline"]["pairs_per_launch"] == per * world * per and 0 < out["roofline"]["frac"] < 1
    assert out["cpu_baseline"] is None
    sh = out["sharded_step"]                                       # where a rank's step goes, and which form of the step ran
    for k in ("local_force_ms", "gather_wait_ms", "remote_force_ms", "host_enqueue_ms", "kick_drift_ms", "host_enqueue_ms_eager"):
        assert sh[k] >= 0.0, k
    assert sh["captured"] is False and sh["step_ran"] == "eager launches"      # gloo rehearsal: nothing to capture


def test_captured_sharded_step_is_bit_identical_to_eager_one_rank_rccl(gpu_device):
    """LeapFrogSimulator.capture_step(): the range-sharded step -- kick-drift, the all-gather, both force blocks, the second
    kick -- as ONE hipGraph replay, on a one-rank RCCL group (tools/shard_capture_check.py, started from
    torch.distributed.run before the GPU is touched): 20 steps bit-identical to the eager sharded step from the same state.
    Capture refused by the runtime is reported, not hidden: the simulator then stays eager and must still be identical."""
    import json
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, NBD_FORCE_SHARDED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "shard_capture_check.py"), "8192", "20"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["bit_identical"] and out["backend"] == "nccl"
    assert out["captured"], "the runtime refused to capture the collective: " + res.stderr[-1500:]
    assert out["host_enqueue_ms_captured"] < out["host_enqueue_ms_eager"]
    assert set(out["phases_ms"]) >= {"local_force_ms", "gather_wait_ms", "remote_force_ms", "host_enqueue_ms"}


def test_full_size_all_rows_against_c_oracle_f64(gpu_device):
    """BASELINE config 2, every one of the 65 536 rows: HIP force vs the C oracle in fp64 (OpenMP on
    the host, ~10 s). Tolerance = the reference's own fp32-vs-fp64 distance (8.7e-7 measured)
    with margin; one leapfrog step then moves positions/velocities within the 1e-5 bar."""
    from nbd.plummer import generate_plummer
    from oracle import c_oracle
    n = 65536
    p, v, m = generate_plummer(n, seed=1234)
    sim = _mk("LeapFrogSimulator", dict(pos=p, vel=v, mass=m, g_const=1.0, softening=0.1, dt=0.01))
    ref = c_oracle.acc_f64(p, m, 1.0, 0.1)
    assert row_rel(_np(sim.accelerations), ref) < 3e-6
    assert global_rel(_np(sim.accelerations), ref) < 5e-7
    # one KDK step in fp64 from the fp32 inputs, force at the drifted positions from the C oracle
    p32, v32 = p.astype(np.float32).astype(np.float64), v.astype(np.float32).astype(np.float64)
    vh = v32 + 0.005 * ref
    x1 = p32 + 0.01 * vh
    a1 = c_oracle.acc_f64(x1.astype(np.float32), m, 1.0, 0.1)
    v1 = vh + 0.005 * a1
    sim.step()
    assert row_rel(_np(sim.positions), x1) < TOL and row_rel(_np(sim.velocities), v1) < TOL


def test_euler_run_reports_post_step_energies(gpu_device):
    """run() must report U, K of the state AFTER each step for every integrator (simulation.py:131-133);
    the Euler step packs its sources before the drift, so the energy pass has to repack."""
    from oracle import galaxify_oracle as go
    g = load_golden("direct_plummer_n64")
    for cls, adv in (("EulerSimulator", "euler_step"), ("LeapFrogSimulator", "leapfrog_step")):
        states = _mk(cls, g).run(3)
        ora = go.OracleSimulator(positions=g["pos"], velocities=g["vel"], masses=g["mass"], g_const=float(g["g_const"]),
                                 softening=float(g["softening"]), dt=float(g["dt"]))
        for st in states:
            getattr(ora, adv)()
            u, k = ora.compute_energies()
            assert abs(st.u_energy - u) <= 2e-5 * abs(u) and abs(st.k_energy - k) <= 2e-6 * abs(k), cls
            assert row_rel(st.positions.numpy(), ora.positions.numpy()) < TOL


def test_dataset_cli_reproduces_the_reference_csv(tmp_path, gpu_device):
    """tests/golden/ref_cli_spiral_n5_n8.csv was written by the REFERENCE's CLI on its CPU path
    (tests/golden/make_golden_cli.py). Same arguments through this build's CLI on the GPU: same header, rows,
    integer / string columns and float formatting; float values within fp32 tolerance (step_time is a clock)."""
    import csv
    import importlib.util
    import os
    import sys
    from conftest import PKG
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden_cli import ARGS
    spec = importlib.util.spec_from_file_location("s01", f"{PKG}/s01-dataset-generation.py")
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    out = str(tmp_path / "out.csv")
    cli.main([*ARGS, "--device", "cuda", "--output", out])
    ref_raw = open(os.path.join(os.path.dirname(__file__), "golden", "ref_cli_spiral_n5_n8.csv"), newline="").read()
    got_raw = open(out, newline="").read()
    assert ref_raw.split("\r\n")[0] == got_raw.split("\r\n")[0] and ref_raw.endswith("\r\n") == got_raw.endswith("\r\n")
    ref = list(csv.DictReader(ref_raw.splitlines()))
    got = list(csv.DictReader(got_raw.splitlines()))
    assert len(ref) == len(got) == (5 + 8) * 3
    f32_cols = ["x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az"]
    for a, b in zip(ref, got):
        assert (a["scene"], a["scene_type"], a["step"]) == (b["scene"], b["scene_type"], b["step"])
        assert a["mass"] == b["mass"]                                   # float64 from the generator: same seed, same string
        for c in f32_cols:
            x, y = float(a[c]), float(b[c])
            assert str(np.float32(y)) == b[c]                           # fp32 values printed the way numpy prints them
            assert abs(x - y) <= 1e-5 * max(abs(x), 1e-30) + 1e-12, (c, a[c], b[c])
        for c in ("u", "k"):
            assert abs(float(a[c]) - float(b[c])) <= 2e-5 * abs(float(a[c])), (c, a[c], b[c])
        assert float(b["step_time"]) >= 0.0


def test_config5_size_range_partition_on_one_gpu(gpu_device):
    """BASELINE configs[4]: 524 288 bodies range-partitioned over 8 ranks. On one GPU: the eight target shards
    (65 536 targets x all 524 288 sources, tgt_global_offset = lo -- exactly what each rank launches) against 64
    sampled rows of an fp64 evaluation and against momentum conservation of the assembled result."""
    from nbd import direct
    from nbd.plummer import generate_plummer
    n, ranks = 524288, 8
    p, v, m = generate_plummer(n, seed=1234)
    pos = torch.tensor(p, dtype=torch.float32).cuda(); mass = torch.tensor(m, dtype=torch.float32).cuda()
    posm = direct.pack_posm(pos, mass)
    per = n // ranks
    eps2 = direct.f32(0.1 ** 2)
    acc = torch.cat([direct.accel(posm, n, posm[r * per:], per, r * per, eps2, 1.0) for r in range(ranks)])
    a64 = _np(acc).astype(np.float64)
    pf, mf = p.astype(np.float32).astype(np.float64), m.astype(np.float32).astype(np.float64)
    net = (mf[:, None] * a64).sum(0)
    assert np.abs(net).max() < 1e-6 * (mf[:, None] * np.abs(a64)).sum(0).max()
    rows = np.random.default_rng(5).choice(n, 64, replace=False)
    d = pf[None, :, :] - pf[rows, None, :]
    q = (d * d).sum(2) + float(np.float32(0.1 ** 2))
    inv = q ** -1.5
    inv[np.arange(64), rows] = 0.0
    ref = (d * (inv * mf[None, :])[:, :, None]).sum(1)
    assert row_rel(a64[rows], ref) < 2e-6


def test_random_rectangular_blocks_against_oracle(gpu_device):
    """25 random (n, target range, softening) blocks -- what a rank of an arbitrary range partition launches --
    against the oracle's rows: ragged n, ranges that start / end anywhere, tiny softening (index-masked kernel),
    massless and heavy bodies."""
    from nbd import direct
    from oracle import galaxify_oracle as go
    rng = np.random.default_rng(99)
    for trial in range(25):
        n = int(rng.choice([2, 3, 63, 64, 65, 130, 500, 1000, 1537, 4099]))
        lo = int(rng.integers(0, n))
        cnt = int(rng.integers(1, n - lo + 1))
        eps = float(rng.choice([0.0, 1e-13, 1e-3, 0.05, 0.1, 1.0]))
        pos = torch.tensor(rng.normal(size=(n, 3)) * rng.choice([0.1, 1.0, 30.0]), dtype=torch.float32)
        mass = torch.tensor(rng.random(n) * rng.choice([1e-6, 1.0, 1e3]), dtype=torch.float32)
        mass[rng.integers(0, n)] = 0.0
        g = float(rng.choice([1.0, 4.5e-6]))
        ref = go.accelerations(pos, mass, g, eps, tgt_slice=slice(lo, lo + cnt))
        posm = direct.pack_posm(pos.cuda(), mass.cuda())
        eps2 = float(torch.tensor(eps ** 2, dtype=torch.float32))
        got = direct.accel(posm, n, posm[lo:], cnt, lo, eps2, direct.f32(g)).cpu()
        assert got.shape == (cnt, 3) and torch.isfinite(got).all(), (trial, n, lo, cnt, eps)
        scale = float(ref.norm(dim=1).max())
        err = float((got - ref).norm(dim=1).max())
        assert err <= 2e-6 * max(scale, 1e-30), (trial, n, lo, cnt, eps, err, scale)


# ---- range-sharded force: two launches (own bodies | every other body) and the tuning hook

def _posm_case(n, seed=5):
    from nbd import direct
    from nbd.plummer import generate_plummer
    p, v, m = generate_plummer(n, seed=seed)
    m = (m * np.random.default_rng(seed).uniform(0.25, 4.0, n)).astype(np.float32)
    pos = torch.tensor(p, dtype=torch.float32, device="cuda")
    mass = torch.tensor(m, dtype=torch.float32, device="cuda")
    return pos, mass, direct.pack_posm(pos, mass)


def _rows_f64(pos, mass, lo, cnt, g):
    """fp64 acceleration (eps = 0, diagonal dropped) of the rows [lo, lo + cnt) against all bodies: numpy, row-blocked."""
    p, m = pos.cpu().numpy().astype(np.float64), mass.cpu().numpy().astype(np.float64)
    out = np.empty((cnt, 3))
    for r0 in range(lo, lo + cnt, 256):
        r1 = min(r0 + 256, lo + cnt)
        d = p[None, :, :] - p[r0:r1, None, :]
        q = (d * d).sum(2)
        with np.errstate(divide="ignore"):
            inv = q ** -1.5
        inv[np.arange(r1 - r0), np.arange(r0, r1)] = 0.0
        out[r0 - lo:r1 - lo] = g * (d * (inv * m[None, :])[:, :, None]).sum(1)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("n,lo,n_loc", [(1024, 256, 256), (1001, 100, 333), (4096, 0, 512), (4096, 3584, 512),
                                        (300, 10, 5), (1000, 0, 1000), (130, 64, 66), (130, 1, 128),
                                        (5000, 4937, 63), (8192, 1024, 1024), (65536, 3 * 8192, 8192)])
@pytest.mark.parametrize("eps", [0.1, 0.0])
def test_split_force_matches_one_launch(n, lo, n_loc, eps, gpu_device):
    """nbd_shard_force_local_f32 + nbd_shard_force_remote_f32 (any lo / n_local alignment, incl. the
    index-masked eps = 0 kernel) against nbd_accel_f32 over all sources; fused kick bit-exact given acc."""
    from nbd import direct
    pos, mass, posm = _posm_case(n)
    eps2, g = direct.f32(eps ** 2), direct.f32(0.7)
    ref = direct.accel(posm, n, posm[lo:], n_loc, lo, eps2, g)
    posm_local = direct.pack_posm(pos[lo:lo + n_loc].contiguous(), mass[lo:lo + n_loc].contiguous())
    ws = direct.shard_workspace(n, lo, n_loc, "cuda")
    acc = torch.empty((n_loc, 3), device="cuda")
    vel = torch.full((n_loc, 3), 0.5, device="cuda")
    direct.shard_force_local(posm_local, n_loc, n, lo, eps2, ws)
    direct.shard_force_remote(posm, n, posm_local, n_loc, lo, eps2, g, acc, vel, 0.25, ws)
    assert torch.isfinite(acc).all()
    # two valid fp32 summation orders of the same pairs; without softening close pairs dominate single rows
    assert row_rel(_np(acc), _np(ref)) < (2e-6 if eps > 0 else 5e-6)
    if eps == 0.0 and n_loc <= 8192:
        # why 5e-6 and not 2e-6 (round 2 saw 2.07e-6 at 65536 / 8192): against an fp64 evaluation of the same rows BOTH
        # orders sit within 3e-6 of the truth -- neither is "the wrong one", their mutual distance is bounded by the sum
        from oracle import galaxify_oracle as go
        want = go.accelerations_f64(pos.cpu().numpy(), mass.cpu().numpy(), 0.7, 0.0, block=256)[lo:lo + n_loc] \
            if n <= 8192 else _rows_f64(pos, mass, lo, n_loc, 0.7)
        e_split, e_one = row_rel(_np(acc), want), row_rel(_np(ref), want)
        assert e_split < 3e-6 and e_one < 3e-6, (e_split, e_one)
    assert torch.equal(vel, torch.full_like(vel, 0.5) + 0.25 * acc)
    # determinism of the split path
    acc2 = torch.empty_like(acc)
    direct.shard_force_local(posm_local, n_loc, n, lo, eps2, ws)
    direct.shard_force_remote(posm, n, posm_local, n_loc, lo, eps2, g, acc2, None, 0.0, ws)
    assert torch.equal(acc, acc2)
    plan = direct.shard_plan(n, lo, n_loc)
    assert plan["slabs_local"] >= 1 and (plan["slabs_remote"] >= 1) == (n_loc < n)


@pytest.mark.gpu
def test_split_force_excludes_exactly_the_own_range(gpu_device):
    """Remote block alone == force from the sources outside [lo, hi): giving the own bodies a huge mass in the
    gathered array must not change the result (they are skipped or masked, never multiplied by zero)."""
    from nbd import direct
    n, lo, n_loc = 1000, 130, 301                      # both ends inside a 64-chunk
    pos, mass, posm = _posm_case(n)
    eps2, g = direct.f32(0.01), 1.0
    posm_local = direct.pack_posm(pos[lo:lo + n_loc].contiguous(), mass[lo:lo + n_loc].contiguous())
    ws = direct.shard_workspace(n, lo, n_loc, "cuda")
    out = []
    for scale in (1.0, 1e30):
        pm = posm.clone()
        pm[lo:lo + n_loc, 3] *= scale
        acc = torch.empty((n_loc, 3), device="cuda")
        direct.shard_force_local(posm_local, n_loc, n, lo, eps2, ws)
        direct.shard_force_remote(pm, n, posm_local, n_loc, lo, eps2, g, acc, None, 0.0, ws)
        out.append(acc)
    assert torch.equal(out[0], out[1])


@pytest.mark.gpu
@pytest.mark.parametrize("n_src,n_tgt", [(4096, 4096), (5000, 777), (65536, 8192)])
def test_tuned_geometries_agree(n_src, n_tgt, gpu_device):
    """Every launch geometry of the tuning hook (slab count, register variant) computes the same force."""
    from nbd import direct
    _, _, posm = _posm_case(n_src)
    eps2 = direct.f32(0.01)
    ref = direct.accel(posm, n_src, posm, n_tgt, 0, eps2, 1.0)
    for variant in (0, 1):
        for slabs in (1, 3, 20, 64):
            if slabs * 4 > (n_src + 63) // 64 * 4 and slabs > 16:
                continue
            got = direct.accel_tuned(posm, n_src, posm, n_tgt, 0, eps2, 1.0, slabs, variant)
            # one wave sums n_src / (4 slabs) sources in ONE sequential fp32 chain: with a single slab over 65 536
            # sources that chain is 16 384 long and its rounding shows (the library's plans keep chains <= 1024)
            chain = n_src / (4 * slabs)
            assert row_rel(_np(got), _np(ref)) < (2e-6 if chain <= 4096 else 2e-5), (variant, slabs)
    # an excluded source range == the same sources with zero mass
    lo, hi = 100, 100 + n_src // 3
    pz = posm.clone(); pz[lo:hi, 3] = 0
    ref_ex = direct.accel(pz, n_src, posm, n_tgt, 0, eps2, 1.0)
    got = direct.accel_tuned(posm, n_src, posm, n_tgt, 0, eps2, 1.0, 7, 0, exclude=(lo, hi))
    assert row_rel(_np(got), _np(ref_ex)) < 2e-6
    with pytest.raises(Exception):
        direct.accel_tuned(posm, n_src, posm, n_tgt, 0, eps2, 1.0, 65, 0)


@pytest.mark.gpu
def test_sharded_simulator_single_rank_group_matches_plain(gpu_device, tmp_path, monkeypatch):
    """A one-rank process group with NBD_FORCE_SHARDED=1: the range-sharded code path end to end in this process
    (collective all-gather with an async handle, split force, fused kick, energies from gathered state) against
    the plain simulator."""
    import torch.distributed as dist
    from galaxify import simulation
    from nbd.plummer import generate_plummer
    dist.init_process_group("gloo", init_method=f"file://{tmp_path}/pg", rank=0, world_size=1)
    try:
        p, v, m = generate_plummer(1500, seed=3)
        kw = dict(positions=p, velocities=v, masses=m, dt=0.01, calc_energy=False, device="cuda")
        a = simulation.LeapFrogSimulator(**kw)
        monkeypatch.setenv("NBD_FORCE_SHARDED", "1")
        b = simulation.LeapFrogSimulator(process_group=dist.group.WORLD, **kw)
        e = simulation.EulerSimulator(process_group=dist.group.WORLD, **kw)
        monkeypatch.delenv("NBD_FORCE_SHARDED")
        assert b._sharded and not a._sharded and b._gather.collective
        e0 = simulation.EulerSimulator(**kw)
        for _ in range(3):
            a.step(); b.step(); e.step(); e0.step()
        for key in ("positions", "velocities", "accelerations"):
            assert row_rel(_np(getattr(b, key)), _np(getattr(a, key))) < 2e-6, key
            assert row_rel(_np(getattr(e, key)), _np(getattr(e0, key))) < 2e-6, key
        ua, ka = a.compute_energies(); ub, kb = b.compute_energies()
        assert abs(ua - ub) < 1e-6 * abs(ua) and abs(ka - kb) < 1e-6 * abs(ka)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_config5_size_on_one_gpu_properties_and_rank_split(gpu_device):
    """BASELINE configs[4]: 524 288 bodies (the 8-GPU shape) on ONE GPU. (a) Newton's third law on the full force,
    (b) 32 sampled rows against an fp64 direct sum, (c) what rank 3 of 8 computes -- own block + remote block over
    the gathered array, nbd_shard_force_* at n_local = 65 536, lo = 196 608 -- equals its rows of the one-launch
    force."""
    from nbd import direct
    from nbd.plummer import generate_plummer
    n = 524288
    p, v, m = generate_plummer(n, seed=1234)
    pos = torch.tensor(p, dtype=torch.float32, device="cuda")
    mass = torch.tensor(m, dtype=torch.float32, device="cuda")
    posm = direct.pack_posm(pos, mass)
    eps2, g = direct.f32(0.1 ** 2), 1.0
    acc = direct.accel(posm, n, posm, n, 0, eps2, g)
    assert torch.isfinite(acc).all()
    a64, m64 = acc.double(), mass.double()
    net = (m64[:, None] * a64).sum(0).norm() / (m64[:, None] * a64).norm(dim=1).sum()
    assert float(net) < 2e-6                                                       # sum_i m_i a_i = 0
    rows = np.random.default_rng(0).choice(n, 32, replace=False)
    p64, m64n = p.astype(np.float32).astype(np.float64), m.astype(np.float32).astype(np.float64)
    ref = np.empty((32, 3))
    for k, i in enumerate(rows):
        d = p64 - p64[i]
        w = m64n * (np.einsum("ij,ij->i", d, d) + float(np.float32(0.01))) ** -1.5
        w[i] = 0.0
        ref[k] = (d * w[:, None]).sum(0)
    assert row_rel(_np(acc[torch.tensor(rows, device="cuda")]), ref) < 3e-6
    lo, n_loc = 3 * 65536, 65536
    posm_local = direct.pack_posm(pos[lo:lo + n_loc].contiguous(), mass[lo:lo + n_loc].contiguous())
    ws = direct.shard_workspace(n, lo, n_loc, "cuda")
    acc_r = torch.empty((n_loc, 3), device="cuda")
    direct.shard_force_local(posm_local, n_loc, n, lo, eps2, ws)
    direct.shard_force_remote(posm, n, posm_local, n_loc, lo, eps2, g, acc_r, None, 0.0, ws)
    assert row_rel(_np(acc_r), _np(acc[lo:lo + n_loc])) < 2e-6
    plan = direct.shard_plan(n, lo, n_loc)
    assert plan["slabs_local"] >= 1 and plan["slabs_remote"] >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("cls,n,steps,energy", [("LeapFrogSimulator", 300, 45, True), ("EulerSimulator", 129, 40, True),
                                                ("LeapFrogSimulator", 1024, 33, False), ("LeapFrogSimulator", 2, 9, True)])
def test_run_in_captured_chunks_equals_eager_run(cls, n, steps, energy, gpu_device, monkeypatch):
    """run() on launch-bound systems replays hipGraph chunks (step + energies + one snapshot launch per step, one
    copy per chunk): every SimulationState, the final device state and the bookkeeping (accelerations rebound,
    earlier handles untouched, a second run() and step() continuing from it) must equal the eager run() bit for bit."""
    from galaxify import galaxies
    p, v, m = galaxies.generate_spiral(n_bodies=n, total_mass=1.0, radial_scale=3.0, height_scale=0.3, g_const=4.5e-6,
                                       black_hole_mass=0.01, seed=5)
    g = dict(pos=p, vel=v, mass=m, g_const=4.5e-6, softening=0.05, dt=1e-4)
    a = _mk(cls, g, calc_energy=energy)
    b = _mk(cls, g, calc_energy=energy)
    assert a._graph_run_ok(steps)
    acc_before = a.accelerations
    acc_before_copy = acc_before.clone()
    sa = a.run(steps)
    monkeypatch.setenv("NBD_RUN_GRAPH", "0")
    assert not b._graph_run_ok(steps)
    sb = b.run(steps)
    monkeypatch.delenv("NBD_RUN_GRAPH")
    assert len(sa) == len(sb) == steps
    for x, y in zip(sa, sb):
        assert x.step == y.step and x.step_time > 0
        assert torch.equal(x.positions, y.positions) and torch.equal(x.velocities, y.velocities)
        assert torch.equal(x.accelerations, y.accelerations)
        assert (x.u_energy, x.k_energy) == (y.u_energy, y.k_energy)
        assert (x.u_energy is not None) == energy
    for key in ("positions", "velocities", "accelerations"):
        assert torch.equal(getattr(a, key), getattr(b, key)), key
    assert torch.equal(acc_before, acc_before_copy) and a.accelerations.data_ptr() != acc_before.data_ptr()
    # continue: a second (cached-graph) run and a plain step
    a.dt = 2e-4; b.dt = 2e-4                              # a changed dt must not replay the old graph
    sa2 = a.run(10); sb2 = [None] * 10
    monkeypatch.setenv("NBD_RUN_GRAPH", "0")
    sb2 = b.run(10)
    monkeypatch.delenv("NBD_RUN_GRAPH")
    assert torch.equal(sa2[-1].positions, sb2[-1].positions) and torch.equal(sa2[-1].accelerations, sb2[-1].accelerations)
    a.step(); b.step()
    assert torch.equal(a.positions, b.positions) and torch.equal(a.velocities, b.velocities)
