"""Leapfrog step time and pair-interactions/s of the single-GPU path across problem sizes (diagnostic)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT]
import torch
from galaxify import simulation
from nbd import direct
from nbd.plummer import generate_plummer

rows = []
for n in (1024, 4096, 16384, 65536, 131072, 262144, 524288):
    p, v, m = generate_plummer(n, seed=1234)
    sim = simulation.LeapFrogSimulator(positions=p, velocities=v, masses=m, calc_energy=False, device="cuda")
    steps = max(5, min(2000, int(3e11 / (n * n))))
    tw = time.perf_counter()                      # warm up by time: the clocks ramp over the first ~0.2 s of load
    while time.perf_counter() - tw < 0.4:
        for _ in range(10):
            sim.step()
        torch.cuda.synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        sim.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    plan = direct.accel_plan(n, n)
    rows.append({"n": n, "ms_per_step": dt * 1e3, "pairs_per_s": n * n / dt, "frac_fp32_peak": n * n / dt * 20 / 157.3e12,
                 "plan": plan})
    print(json.dumps(rows[-1]), flush=True)
