#!/usr/bin/env python3
"""Headline benchmark: pair-interactions/s of the direct O(N^2) leapfrog step on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one LeapFrogSimulator.step() (kick-drift, one all-pairs force evaluation, kick) on a
seeded Plummer sphere already resident in HBM. Workload: 65 536 particles PER GPU, i.e.
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

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

FLOP_PER_PAIR = 20.0            # customary all-pairs count
PEAK_FP32_TFLOPS = 157.3        # MI355X_MICROARCH.md: fp32 vector peak (= fp32 MFMA peak)
PARTICLES_PER_GPU = 65536


def cpu_baseline(n, seed, budget_s, threads):
    """The reference's torch-CPU algorithm (row-blocked port, oracle/galaxify_oracle.py) timed on
    this host on a bounded sample: the force on the first R target rows against all n sources."""
    import torch
    from nbd.plummer import generate_plummer
    from oracle import galaxify_oracle as go
    torch.set_num_threads(threads)
    p, v, m = generate_plummer(n, seed=seed)
    pos = torch.tensor(p, dtype=torch.float32)
    mass = torch.tensor(m, dtype=torch.float32)
    go.accelerations(pos, mass, 1.0, 0.1, block=256, tgt_slice=slice(0, 256))       # warm-up
    t0 = time.perf_counter()
    go.accelerations(pos, mass, 1.0, 0.1, block=512, tgt_slice=slice(0, 1024))
    t_probe = time.perf_counter() - t0
    rows = int(min(n, max(1024, 1024 * (budget_s / max(t_probe, 1e-6)))))
    rows = max(512, (rows // 512) * 512)
    t0 = time.perf_counter()
    go.accelerations(pos, mass, 1.0, 0.1, block=512, tgt_slice=slice(0, rows))
    dt = time.perf_counter() - t0
    return {"value": rows * n / dt, "unit": "pair-interactions/s", "cores": threads, "kind": "port",
            "sample": f"force on the first {rows} of {n} targets x all {n} sources (Plummer, fp32), "
                      f"row-blocked torch-CPU port of simulation.py:80-88, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles-per-gpu", type=int, default=PARTICLES_PER_GPU)
    ap.add_argument("--n-total", type=int, default=0, help="fix the TOTAL particle count (strong scaling)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget; 0 disables")
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
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sim.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sim.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # roofline leg (every rank runs it to stay in lock-step; rank 0 reports): HIP events on the
    # launch stream around the force kernel of K further steps
    evs = []
    for _ in range(args.steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()                       # creates the hipEvent_t handles
        if world == 1:
            new_acc = torch.empty_like(sim.accelerations)
            direct.leapfrog_step(sim.positions, sim.velocities, sim.accelerations, new_acc, sim.masses,
                                 direct.f32(0.5 * sim.dt), direct.f32(sim.dt), sim._eps2, sim._g,
                                 sim._posm, sim._ws, ev_begin=e0, ev_end=e1)
            sim.accelerations = new_acc
        else:                                          # sharded step: events around force(+finish)
            half, dt = direct.f32(0.5 * sim.dt), direct.f32(sim.dt)
            direct.kick_drift(sim.positions, sim.velocities, sim.accelerations, sim._mass_local, half, dt,
                              posm=sim._posm_local)
            sim._exchange()
            e0.record()
            sim.accelerations = sim._force()
            e1.record()
            direct.kick(sim.velocities, sim.accelerations, half)
        evs.append((e0, e1))
    barrier()
    k_ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
    assert torch.isfinite(sim.positions).all()

    # BASELINE.json's metric string reads "N = 65 536 ... 1/2/4/8 MI355X": next to the weak series above
    # (fixed bodies per GPU, configs[4] at 8 GPUs), time the SAME 65 536-body problem split over all ranks.
    strong_leg = None
    if world > 1 and not strong:
        n_s = args.particles_per_gpu
        p2, v2, m2 = generate_plummer(n_s, seed=args.seed)
        sim2 = simulation.LeapFrogSimulator(positions=p2, velocities=v2, masses=m2, g_const=1.0, softening=0.1,
                                            dt=0.01, calc_energy=False, device="cuda", process_group=group)
        for _ in range(args.warmup):
            sim2.step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            sim2.step()
        barrier()
        el2 = time.perf_counter() - t1
        t = torch.tensor([el2], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el2 = t.item()
        strong_leg = {"n_particles": n_s, "value": float(n_s) * float(n_s) * args.steps / el2,
                      "unit": "pair-interactions/s", "ms_per_step": el2 / args.steps * 1e3, "scaling": "strong",
                      "note": "same run, the single-GPU problem size split over all ranks (one all-gather per step)"}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    pairs_per_step = float(n_total) * float(n_total)
    value = pairs_per_step * args.steps / elapsed
    n_loc = sim.part.n_local
    pairs_per_launch = float(n_loc) * float(n_total)
    achieved = pairs_per_launch * FLOP_PER_PAIR / (k_ms * 1e-3) / 1e12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and world == 1 and not strong:
        traffic = json.load(open(tpath)).get("accel_kernel_hbm_bytes_per_launch")
    plan = direct.accel_plan(n_total, n_loc)
    out = {
        "metric": "pair-interactions/sec, direct O(N^2) leapfrog N-body, fp32",
        "value": value, "unit": "pair-interactions/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"Plummer sphere, {n_total} particles, direct all-pairs leapfrog step "
                        f"(BASELINE configs[{1 if world == 1 else 4}] shape: {args.particles_per_gpu} bodies per GPU)"
                        if not strong else f"Plummer sphere, {n_total} particles total (strong scaling)",
            "n_particles": n_total, "particles_per_gpu": n_loc, "g_const": 1.0, "softening": 0.1, "dt": 0.01,
            "seed": args.seed, "pairs_per_step": pairs_per_step,
            "parallelism": "single GPU" if world == 1 else
                           f"range partition x{world}, one RCCL all-gather of float4[{n_loc}] per rank per step",
            "launch_plan": plan,
        },
        "roofline": {
            # compute roofline ("mfma" in the harness's two-way hbm|mfma taxonomy): the kernel itself is fp32
            # VALU -- all-pairs gravity has no dense contraction for MFMA -- and on MI355X the fp32 vector
            # peak and the fp32 MFMA peak are the same 157.3 TFLOP/s, so the fraction is unambiguous
            "bound": "mfma", "compute_unit": "fp32 VALU (v_pk_fma_f32 / v_rsq_f32), no MFMA instructions",
            "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic,
            "kernel": "accel_kernel<false>", "kernel_ms": k_ms, "flop_per_pair": FLOP_PER_PAIR,
            "pairs_per_launch": pairs_per_launch,
            "note": "compute-bound on fp32 VALU issue (no dense contraction: MFMA not applicable); peak = fp32 "
                    "vector peak = fp32 MFMA peak. 20 flop/pair accounting; the instruction stream's own ceiling "
                    "is 2 pairs/clk/SIMD = 62% of this peak at 2.4 GHz",
        },
    }
    if strong_leg is not None:
        out["strong_scaling_n65536"] = strong_leg
    if args.cpu_seconds > 0 and world == 1:
        threads = min(len(os.sched_getaffinity(0)), 16)
        out["cpu_baseline"] = cpu_baseline(min(n_total, 65536), args.seed, args.cpu_seconds, threads)
    elif world > 1:
        out["cpu_baseline"] = None
    if world == 1 and not strong and not args.no_surrogates:
        # secondary legs (BASELINE configs[2], [3]): ms per surrogate rollout step + rollout MSE
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_surrogates
        out["secondary"] = bench_surrogates.run(10)
        import bench_train          # ms per training step at the reference's batch sizes
        out["secondary"]["training"] = bench_train.run(10)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
