"""Diagnostic: cost of run() with calc_energy (energy kernel + state capture) vs bare steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT]
import torch
from galaxify import simulation
from nbd import direct
from nbd.plummer import generate_plummer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
p, v, m = generate_plummer(n, seed=1234)
for ce in (False, True):
    sim = simulation.LeapFrogSimulator(positions=p, velocities=v, masses=m, calc_energy=ce, device="cuda")
    sim.run(3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = sim.run(20)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"n={n} calc_energy={ce}: run() {dt*1e3:.3f} ms/step wall; mean step_time {sum(s.step_time for s in st)/20*1e3:.3f} ms; E={st[-1].u_energy} {st[-1].k_energy}")
sim = simulation.LeapFrogSimulator(positions=p, velocities=v, masses=m, calc_energy=True, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = torch.empty(2, dtype=torch.float64, device="cuda")
direct.energy(sim._posm, sim.velocities, n, 0.1, 1.0, out_uk=out); torch.cuda.synchronize()
e0.record()
for _ in range(10): direct.energy(sim._posm, sim.velocities, n, 0.1, 1.0, out_uk=out)
e1.record(); torch.cuda.synchronize()
print("energy kernel ms:", e0.elapsed_time(e1) / 10, out.tolist())
