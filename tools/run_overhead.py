"""Diagnostic: cost of run() (state capture to host + energies) vs bare steps, per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT]
import torch
from galaxify import simulation
from nbd.plummer import generate_plummer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
p, v, m = generate_plummer(n, seed=1234)
for ce in (False, True):
    sim = simulation.LeapFrogSimulator(positions=p, velocities=v, masses=m, calc_energy=ce, device="cuda")
    sim.run(40)                                   # warm: pinned staging allocated once
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = sim.run(40)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
    print(f"n={n} calc_energy={ce}: run() {dt*1e3:.3f} ms/step wall (36 N bytes/step to the host = "
          f"{36*n/dt/1e9:.2f} GB/s); GPU step_time {sum(s.step_time for s in st)/40*1e3:.3f} ms")
