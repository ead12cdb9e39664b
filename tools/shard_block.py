"""One rank's share of the range-sharded leapfrog step, emulated on ONE GPU (strong scaling evidence without
the 8-GPU node): rank `--rank` of `--world` owns n/world bodies of an n-body Plummer sphere and runs, per
step, exactly the launches the sharded LeapFrogSimulator issues --

    kick_drift(local) | [gather: here a device copy of the own piece into the gathered array]
    force(own x own)  | force(own x remote) + slab sum + kick

-- with HIP-event times per phase. Run under `rocprofv3 --kernel-trace` for per-kernel durations
(tools/summarize_trace.py groups them by kernel and grid).   python tools/shard_block.py --world 8 --rank 3"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT):
    sys.path.insert(0, _p)
import torch
from nbd import direct
from nbd.plummer import generate_plummer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--general", action="store_true", help="the general-mass kernels (default: the equal-mass ones, as the simulator picks them for a Plummer sphere)")
    args = ap.parse_args()
    n, world, rank = args.n, args.world, args.rank
    n_loc = n // world
    lo = rank * n_loc
    p, v, m = generate_plummer(n, seed=1234)
    dev = "cuda"
    pos = torch.tensor(p[lo:lo + n_loc], dtype=torch.float32, device=dev)
    vel = torch.tensor(v[lo:lo + n_loc], dtype=torch.float32, device=dev)
    mass_all = torch.tensor(m, dtype=torch.float32, device=dev)
    mass = mass_all[lo:lo + n_loc].contiguous()
    posm = direct.pack_posm(torch.tensor(p, dtype=torch.float32, device=dev), mass_all)
    posm_local = direct.alloc_posm(n_loc, dev); posm_local.zero_()
    ws = direct.shard_workspace(n, lo, n_loc, dev)
    acc = torch.zeros((n_loc, 3), device=dev)
    eps2, g = direct.f32(0.01), 1.0
    half, dt = direct.f32(0.005), direct.f32(0.01)
    uni = None if args.general else direct.uniform_mass(mass_all)
    direct.pack_posm(pos, mass, out=posm_local)
    direct.shard_force_local(posm_local, n_loc, n, lo, eps2, ws, uniform=uni)
    direct.shard_force_remote(posm, n, posm_local, n_loc, lo, eps2, g, acc, None, 0.0, ws, uniform=uni)

    def step(events=None):
        if events: events[0].record()
        direct.kick_drift(pos, vel, acc, mass, half, dt, posm=posm_local)
        if events: events[1].record()
        posm[lo:lo + n_loc].copy_(posm_local[:n_loc])            # stands in for the all-gather's arrival
        if events: events[2].record()
        direct.shard_force_local(posm_local, n_loc, n, lo, eps2, ws, uniform=uni)
        if events: events[3].record()
        direct.shard_force_remote(posm, n, posm_local, n_loc, lo, eps2, g, acc, vel, half, ws, uniform=uni)
        if events: events[4].record()

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:                        # clock ramp
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    # whole steps, back to back (what the GPU does when the host runs ahead)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    t_host = time.perf_counter() - t0                            # host time to ENQUEUE the steps
    torch.cuda.synchronize()
    ms_step = e0.elapsed_time(e1) / args.steps
    # per phase
    phases = [0.0] * 4
    for _ in range(50):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        step(ev)
        torch.cuda.synchronize()
        for i in range(4):
            phases[i] += ev[i].elapsed_time(ev[i + 1]) / 50
    out = {"n": n, "world": world, "rank": rank, "n_local": n_loc, "plan": direct.shard_plan(n, lo, n_loc),
           "ms_per_step_gpu": ms_step, "host_enqueue_ms_per_step": t_host / args.steps * 1e3,
           "phase_ms": dict(zip(["kick_drift", "gather_stand_in_copy", "force_local", "force_remote_finish_kick"], phases)),
           "pairs_per_s_this_rank": float(n_loc) * n / (ms_step * 1e-3),
           "implied_speedup_vs_1gpu_ms": None}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
