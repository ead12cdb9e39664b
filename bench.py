#!/usr/bin/env python3
"""Headline benchmark: pair-interactions/s of the direct O(N^2) leapfrog step on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one LeapFrogSimulator.step() (kick-drift, one all-pairs force evaluation, kick) on a
seeded Plummer sphere already resident in HBM. EXACTLY --steps steps are timed per window (barrier +
synchronize on both sides, MAX over ranks); the window is repeated >= --repeats times (>= 1 s of timed
steps in all) and `value` / `ms_per_step` are the MEDIAN window's. Workload: 65 536 particles PER GPU, i.e.
BASELINE.json configs[1] at N=1 (65 536 bodies, one MI355X) and configs[4] at N=8 (524 288 bodies
range-sharded over 8 GPUs with one RCCL all-gather of positions per step). pairs/step = n_total^2
(the reference evaluates every (i,j) incl. i=j, simulation.py:80-88), one force evaluation per step.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this stack needs dmabuf IPC; must be in the environment before the HIP runtime
# comes up (i.e. before torch touches the device), so it is set at import time
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

FLOP_PER_PAIR = 20.0            # customary all-pairs count
PEAK_FP32_TFLOPS = 157.3        # MI355X_MICROARCH.md: fp32 vector peak (= fp32 MFMA peak)
PARTICLES_PER_GPU = 65536


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(n, seed, budget_s):
    """The reference's torch-CPU algorithm (row-blocked port, oracle/galaxify_oracle.py) timed on this
    host, SURVEY 8(d): whole leapfrog steps at N = 1 024 (configs[0]) and N = 16 384, and at N = 65 536 ONE
    WHOLE step when the thread scan says it fits 0.75 x the --cpu-seconds budget, else a bounded sample of
    the force (the first R target rows against all n sources; `sample` says which). The headline `value`
    is the N = 65 536 figure at the thread count that measured fastest."""
    import numpy as np
    import torch
    from nbd.plummer import generate_plummer
    from oracle import galaxify_oracle as go
    affinity = len(os.sched_getaffinity(0))
    out = {"unit": "pair-interactions/s", "kind": "port", "cores_total": os.cpu_count(),
           "cores_affinity": affinity, "cpu_model": _cpu_model(), "sizes": {}}

    p, v, m = generate_plummer(n, seed=seed)
    pos = torch.tensor(p, dtype=torch.float32)
    mass = torch.tensor(m, dtype=torch.float32)

    def sample_rate(threads, rows):
        torch.set_num_threads(threads)
        go.accelerations(pos, mass, 1.0, 0.1, block=256, tgt_slice=slice(0, 256))       # warm-up
        t0 = time.perf_counter()
        go.accelerations(pos, mass, 1.0, 0.1, block=512, tgt_slice=slice(0, rows))
        return rows * n / (time.perf_counter() - t0)

    # thread scan on a short probe: torch's intra-op pool does not always scale to every core of the box
    scan = {}
    for t in sorted({min(16, affinity), min(32, affinity), min(64, affinity), affinity}):
        scan[t] = sample_rate(t, 2048)
    best_t = max(scan, key=scan.get)
    out["thread_scan_pairs_per_s"] = {str(k): v_ for k, v_ in scan.items()}
    whole_step_s = float(n) * n / scan[best_t]
    out["whole_step_estimate_s"] = whole_step_s
    if whole_step_s <= 0.75 * budget_s:
        # SURVEY 8(d) / BASELINE.md 3: ONE WHOLE leapfrog step (kick, drift, all n^2 pairs, kick) of the port at n
        torch.set_num_threads(best_t)
        ora = go.OracleSimulator(positions=p, velocities=v, masses=m, g_const=1.0, softening=0.1, dt=0.01, block=512,
                                 initial_accelerations=np.zeros_like(p))     # a(t0) is not part of a step's cost
        t0 = time.perf_counter()
        ora.leapfrog_step()
        dt = time.perf_counter() - t0
        rate = float(n) * n / dt
        rows = n
        out.update({"value": rate, "cores": best_t,
                    "sample": f"ONE WHOLE leapfrog step at N = {n} (all {n}^2 pairs; Plummer, fp32), row-blocked "
                              f"torch-CPU port of simulation.py:80-88,153-170, {best_t} threads, {dt:.1f} s"})
    else:
        rows = int(min(n, max(2048, 0.5 * budget_s * scan[best_t] / n)))
        rows = max(512, (rows // 512) * 512)
        t0 = time.perf_counter()
        rate = sample_rate(best_t, rows)
        dt = time.perf_counter() - t0
        out.update({"value": rate, "cores": best_t,
                    "sample": f"ROW SAMPLE, not a whole step (a whole step would take ~{whole_step_s:.0f} s > 0.75 x the "
                              f"--cpu-seconds budget of {budget_s:.0f} s): force on the first {rows} of {n} targets x all "
                              f"{n} sources (Plummer, fp32), row-blocked torch-CPU port of simulation.py:80-88, "
                              f"{best_t} threads, {dt:.1f} s"})

    # whole leapfrog steps at the smaller SURVEY 8(d) sizes (OracleSimulator = simulation.py:153-170)
    torch.set_num_threads(best_t)
    for n_s, steps in ((1024, 50), (16384, 3)):
        ps, vs, ms = generate_plummer(n_s, seed=seed)
        ora = go.OracleSimulator(positions=ps, velocities=vs, masses=ms)
        ora.leapfrog_step()
        t0 = time.perf_counter()
        for _ in range(steps):
            ora.leapfrog_step()
        dt = time.perf_counter() - t0
        out["sizes"][str(n_s)] = {"pairs_per_s": float(n_s) * n_s * steps / dt, "ms_per_step": dt / steps * 1e3,
                                  "steps": steps, "threads": best_t}
    out["sizes"][str(n)] = {"pairs_per_s": rate, "threads": best_t, "sample_rows": rows}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles-per-gpu", type=int, default=PARTICLES_PER_GPU)
    ap.add_argument("--n-total", type=int, default=0, help="fix the TOTAL particle count (strong scaling)")
    ap.add_argument("--cpu-seconds", type=float, default=75.0,
                    help="CPU baseline budget for the N = 65 536 leg (one whole step if it fits 0.75 x this, else a row sample); 0 disables")
    ap.add_argument("--prewarm-seconds", type=float, default=0.5,
                    help="untimed steps run for this long before --warmup (clock ramp); 0 disables")
    ap.add_argument("--repeats", type=int, default=5, help="minimum number of timed windows of --steps steps each")
    ap.add_argument("--min-timed-seconds", type=float, default=1.0,
                    help="keep adding windows until this much time has been spent in timed steps (<= 400 windows)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--no-surrogates", action="store_true", help="skip the secondary GNN / ContConv rollout timings")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from galaxify import simulation
    from nbd import direct
    from nbd.plummer import generate_plummer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with `python -m torch.distributed.run "
                     "--nproc-per-node N ...` (one process per GPU)")
        args.gpus = world
    # NBD_BENCH_SHARE_GPU=1 + NBD_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a 1-GPU box
    share = os.environ.get("NBD_BENCH_SHARE_GPU") == "1"
    local_rank = 0 if share else local_rank
    torch.cuda.set_device(local_rank)
    group = None
    force_dist = os.environ.get("NBD_FORCE_SHARDED") == "1"      # one-rank rehearsal of the RCCL path
    if world > 1 or force_dist:
        backend = os.environ.get("NBD_DIST_BACKEND", "nccl")                           # nccl = RCCL over xGMI
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    n_total = args.n_total or args.particles_per_gpu * world
    strong = bool(args.n_total)
    p, v, m = generate_plummer(n_total, seed=args.seed)
    sim = simulation.LeapFrogSimulator(positions=p, velocities=v, masses=m, g_const=1.0, softening=0.1,
                                       dt=0.01, calc_energy=False, device="cuda", process_group=group)

    def barrier():
        if group is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # steady state: the chip's clocks ramp over the first ~0.2 s of load (the first launches run 15-30 %
    # slow), so pre-warm by TIME before the caller's --warmup steps; reported in the JSON line
    def prewarm(s_, seconds):
        t_, k_ = time.perf_counter(), 0
        while True:
            for _ in range(10):
                s_.step()
            k_ += 10
            torch.cuda.synchronize()
            flag = torch.tensor([time.perf_counter() - t_ >= seconds], dtype=torch.int32, device="cuda")
            if group is not None:              # all ranks must agree on the step count (collective inside step)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if flag.item():
                return k_, time.perf_counter() - t_
    pre_steps, pre_s = prewarm(sim, args.prewarm_seconds) if args.prewarm_seconds > 0 else (0, 0.0)
    for _ in range(args.warmup):
        sim.step()

    def host_enqueue_ms(s_, k_=20):
        """wall time the host spends enqueueing one step (no synchronisation inside the loop)"""
        barrier()
        t_ = time.perf_counter()
        for _ in range(k_):
            s_.step()
        dt_ = (time.perf_counter() - t_) / k_ * 1e3
        barrier()
        return dt_
    # The range-sharded step as ONE hipGraph, collective included (LeapFrogSimulator.capture_step); eager stays the
    # fallback when the runtime refuses the capture (gloo rehearsals always do), and the JSON says which ran.
    sharded_info = None
    if sim._sharded:
        sharded_info = {"host_enqueue_ms_eager": host_enqueue_ms(sim)}
        want = os.environ.get("NBD_CAPTURE_SHARDED", "1") != "0" and (group is None or dist.get_backend() == "nccl")
        ok_t = torch.tensor([1 if (want and sim.capture_step()) else 0], dtype=torch.int32, device="cuda")
        if group is not None:                      # all ranks replay or none does
            dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        if not ok_t.item():
            sim._step_graph = None
        sharded_info["captured"] = bool(ok_t.item())
        sharded_info["step_ran"] = "hipGraph replay (kick-drift, all-gather, both force blocks)" if ok_t.item() else "eager launches"
        if ok_t.item():
            sharded_info["host_enqueue_ms_captured"] = host_enqueue_ms(sim)

    def timed_windows(s_, min_repeats, min_total_s):
        """EXACTLY --steps steps per window, each window bracketed by barrier + synchronize on both sides and the MAX
        over ranks taken per window; repeated (median of >= 5 repeats) until >= min_total_s of timed
        steps have run -- one 20 ms window is inside the 4 % box-to-box / clock noise it cannot report."""
        def one():
            barrier()
            t_ = time.perf_counter()
            for _ in range(args.steps):
                s_.step()
            barrier()
            return time.perf_counter() - t_
        first = one()
        rep = torch.tensor([max(min_repeats, min(int(min_total_s / max(first, 1e-6)) + 1, 400))], dtype=torch.int64, device="cuda")
        if group is not None:                      # every rank must run the same number of windows
            dist.all_reduce(rep, op=dist.ReduceOp.MAX)
        ts = [first] + [one() for _ in range(int(rep.item()) - 1)]
        t = torch.tensor(ts, dtype=torch.float64, device="cuda")
        if group is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return sorted(t.tolist())
    windows = timed_windows(sim, args.repeats, args.min_timed_seconds)
    elapsed = windows[len(windows) // 2] if len(windows) % 2 else 0.5 * (windows[len(windows) // 2 - 1] + windows[len(windows) // 2])

    # roofline leg (every rank runs it to stay in lock-step; rank 0 reports): HIP events on the
    # launch stream around the force kernel of K further steps
    def kernel_event_ms(sim):
        evs = []
        for _ in range(args.steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record()                       # creates the hipEvent_t handles
            if not sim._sharded:
                new_acc = torch.empty_like(sim.accelerations)
                direct.leapfrog_step(sim.positions, sim.velocities, sim.accelerations, new_acc, sim.masses,
                                     direct.f32(0.5 * sim.dt), direct.f32(sim.dt), sim._eps2, sim._g,
                                     sim._posm, sim._ws, ev_begin=e0, ev_end=e1, uniform=getattr(sim, "_uniform", None))
                sim.accelerations = new_acc
            else:                                          # sharded step: events around BOTH force launches
                half, dt = direct.f32(0.5 * sim.dt), direct.f32(sim.dt)
                pt = sim.part
                direct.kick_drift(sim.positions, sim.velocities, sim.accelerations, sim._mass_local, half, dt,
                                  posm=sim._posm_local)
                handle = sim._gather.start(sim._posm_local, sim._posm)
                sim._gather.finish(handle, sim._posm)      # measurement leg: no overlap, the kernels alone
                loc = sim._posm_local[:direct.padded_len(pt.n_local)]
                acc = torch.empty_like(sim.accelerations)
                e0.record()
                direct.shard_force_local(loc, pt.n_local, sim.n, pt.lo, sim._eps2, sim._ws, uniform=getattr(sim, "_uniform", None))
                direct.shard_force_remote(sim._posm, sim.n, loc, pt.n_local, pt.lo, sim._eps2, sim._g, acc,
                                          sim.velocities, half, sim._ws, uniform=getattr(sim, "_uniform", None))
                e1.record()
                if sim._step_graph is not None:
                    sim.accelerations.copy_(acc)           # the captured step's static buffer
                else:
                    sim.accelerations = acc
            evs.append((e0, e1))
        barrier()
        return sum(a.elapsed_time(b) for a, b in evs) / len(evs)
    k_ms = kernel_event_ms(sim)
    assert torch.isfinite(sim.positions).all()

    # where a rank's step goes (HIP events between the phases of EAGER steps, MAX over ranks): own x own block, launch
    # stream idle until the all-gather has landed, own x remote block; beside the host's enqueue time
    if sharded_info is not None:
        ph = [sim.step_phases() for _ in range(max(args.steps, 5))]
        keys = ("kick_drift_ms", "local_force_ms", "gather_wait_ms", "remote_force_ms", "host_enqueue_ms")
        t_ph = torch.tensor([sorted(x[k_] for x in ph)[len(ph) // 2] for k_ in keys], dtype=torch.float64, device="cuda")
        if group is not None:
            dist.all_reduce(t_ph, op=dist.ReduceOp.MAX)
        sharded_info.update({k_: float(v_) for k_, v_ in zip(keys, t_ph.tolist())})
        sharded_info["phases_note"] = ("median over eager steps of HIP-event intervals on the launch stream, MAX over ranks; "
                                       "gather_wait_ms = stream idle between the end of the own x own block and the all-gather's "
                                       "completion (what the overlap does not cover)")

    # the general-mass kernel beside the equal-mass headline (the reference multiplies by m_j per pair,
    # simulation.py:86-88; the Plummer set's equal masses let the headline kernel factor the mass out of the sum)
    general = None
    if world == 1 and not strong and getattr(sim, "_uniform", None) is not None:
        os.environ["NBD_UNIFORM_MASS"] = "0"
        try:
            sim_g = simulation.LeapFrogSimulator(positions=p, velocities=v, masses=m, g_const=1.0, softening=0.1,
                                                 dt=0.01, calc_energy=False, device="cuda")
        finally:
            del os.environ["NBD_UNIFORM_MASS"]
        for _ in range(args.warmup):
            sim_g.step()
        w_g = timed_windows(sim_g, args.repeats, min(args.min_timed_seconds, 0.5))
        el_g = w_g[len(w_g) // 2]
        kg_ms = kernel_event_ms(sim_g)
        ach_g = float(n_total) * float(n_total) * FLOP_PER_PAIR / (kg_ms * 1e-3) / 1e12
        general = {"kernel": "accel_kernel<false,8>", "why": "NBD_UNIFORM_MASS=0: the per-pair multiply by m_j kept (any unequal-mass caller, "
                   "e.g. generate_disk's star masses, runs this kernel)", "ms_per_step": el_g / args.steps * 1e3, "repeats": len(w_g),
                   "value": float(n_total) * float(n_total) * args.steps / el_g, "kernel_ms": kg_ms, "achieved": ach_g,
                   "frac": ach_g / PEAK_FP32_TFLOPS}
        del sim_g

    # BASELINE.json's metric string reads "N = 65 536 ... 1/2/4/8 MI355X": next to the weak series above
    # (fixed bodies per GPU, configs[4] at 8 GPUs), time the SAME 65 536-body problem split over all ranks.
    strong_leg = None
    if world > 1 and not strong:
        n_s = args.particles_per_gpu
        p2, v2, m2 = generate_plummer(n_s, seed=args.seed)
        sim2 = simulation.LeapFrogSimulator(positions=p2, velocities=v2, masses=m2, g_const=1.0, softening=0.1,
                                            dt=0.01, calc_energy=False, device="cuda", process_group=group)
        prewarm(sim2, min(args.prewarm_seconds, 0.2))
        for _ in range(args.warmup):
            sim2.step()
        w2 = timed_windows(sim2, args.repeats, min(args.min_timed_seconds, 0.5))
        el2 = w2[len(w2) // 2]
        strong_leg = {"n_particles": n_s, "value": float(n_s) * float(n_s) * args.steps / el2,
                      "unit": "pair-interactions/s", "ms_per_step": el2 / args.steps * 1e3, "scaling": "strong", "repeats": len(w2),
                      "launch_plan": direct.shard_plan(n_s, sim2.part.lo, sim2.part.n_local),
                      "note": "same run, the single-GPU problem size split over all ranks (one all-gather per step, "
                              "overlapped with the own-bodies force block)"}

    if rank != 0:
        if group is not None:
            dist.destroy_process_group()
        return
    pairs_per_step = float(n_total) * float(n_total)
    value = pairs_per_step * args.steps / elapsed
    n_loc = sim.part.n_local
    pairs_per_launch = float(n_loc) * float(n_total)
    achieved = pairs_per_launch * FLOP_PER_PAIR / (k_ms * 1e-3) / 1e12
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and world == 1 and not strong:
        tj = json.load(open(tpath))
        traffic = tj.get("accel_kernel_hbm_bytes_per_launch")
        traffic_source = ("profiles/traffic.json (" + str(tj.get("source", "rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE passes of an earlier run of this command")) +
                          "): a constant read from the repository, NOT measured by this run")
    plan = direct.accel_plan(n_total, n_loc) if not sim._sharded else direct.shard_plan(n_total, sim.part.lo, n_loc)
    out = {
        "metric": "pair-interactions/sec, direct O(N^2) leapfrog N-body, fp32",
        "value": value, "unit": "pair-interactions/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "prewarm": {"seconds": pre_s, "steps": pre_steps,
                                           "why": "clock ramp: untimed steps run by time before --warmup"},
        "ms_per_step": elapsed / args.steps * 1e3, "repeats": len(windows),
        "ms_per_step_min_median_max": [windows[0] / args.steps * 1e3, elapsed / args.steps * 1e3, windows[-1] / args.steps * 1e3],
        "timed_seconds_total": sum(windows), "ranks_seen": (dist.get_world_size() if group is not None else 1),
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"Plummer sphere, {n_total} particles, direct all-pairs leapfrog step "
                        f"(BASELINE configs[{1 if world == 1 else 4}] shape: {args.particles_per_gpu} bodies per GPU)"
                        if not strong else f"Plummer sphere, {n_total} particles total (strong scaling)",
            "n_particles": n_total, "particles_per_gpu": n_loc, "g_const": 1.0, "softening": 0.1, "dt": 0.01,
            "seed": args.seed, "pairs_per_step": pairs_per_step,
            "parallelism": "single GPU" if world == 1 else
                           f"range partition x{world}, one all-gather of float4[{n_loc}] per rank per step "
                           f"(backend {dist.get_backend() if group is not None else 'none'}: nccl = RCCL over xGMI), "
                           f"overlapped with the own-bodies force block",
            "launch_plan": plan,
        },
        "roofline": {
            # compute roofline ("mfma" in the harness's two-way hbm|mfma taxonomy): the kernel itself is fp32
            # VALU -- all-pairs gravity has no dense contraction for MFMA -- and on MI355X the fp32 vector
            # peak and the fp32 MFMA peak are the same 157.3 TFLOP/s, so the fraction is unambiguous
            "bound": "mfma", "compute_unit": "fp32 VALU (v_pk_fma_f32 / v_rsq_f32), no MFMA instructions",
            "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
            "kernel": ("accel_kernel<false,8,uniform mass>" if getattr(sim, "_uniform", None) is not None
                       else "accel_kernel<false,8>"),
            "kernel_ms": k_ms, "flop_per_pair": FLOP_PER_PAIR,
            "pairs_per_launch": pairs_per_launch,
            "note": "compute-bound on fp32 VALU issue (no dense contraction: MFMA not applicable); peak = fp32 "
                    "vector peak = fp32 MFMA peak. 20 flop/pair accounting (SURVEY 8d; what the kernel executes is "
                    "fewer: equal masses factor out of the sum, 11 packed ops + 2 rsq per source and pair of targets). "
                    "The instruction stream's own ceiling is 128 pairs per 60 issue cycles per SIMD = 66% of this peak "
                    "at 2.4 GHz (general masses: 64 cycles, 62%)",
        },
    }
    if general is not None:
        out["roofline"]["general_mass"] = general
    if sharded_info is not None:
        out["sharded_step"] = sharded_info
    if strong_leg is not None:
        out["strong_scaling_n65536"] = strong_leg
    if args.cpu_seconds > 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(min(n_total, 65536), args.seed, args.cpu_seconds)
    elif world > 1:
        out["cpu_baseline"] = None
    if world == 1 and not strong and not args.no_surrogates:
        # secondary legs (BASELINE configs[2], [3]): ms per surrogate rollout step + rollout MSE
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_surrogates
        out["secondary"] = bench_surrogates.run(10)
        import bench_train          # ms per training step at the reference's batch sizes
        out["secondary"]["training"] = bench_train.run(10)
    if force_dist:
        out["rehearsal"] = "NBD_FORCE_SHARDED=1: the range-sharded code path (RCCL group, async all-gather, split force) on one rank"
    print(json.dumps(out), flush=True)
    if group is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
