"""Diagnostic: where does a leapfrog step's time go beyond the force kernel? (GPU box only)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT]
import torch
from galaxify import simulation
from nbd import direct, _lib
from nbd.plummer import generate_plummer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = 200
p, v, m = generate_plummer(n, seed=1234)
sim = simulation.LeapFrogSimulator(positions=p, velocities=v, masses=m, calc_energy=False, device="cuda")

def timed(fn, k=K):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(k): fn()
    t_host = time.perf_counter() - t0
    e1.record(); torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    return e0.elapsed_time(e1) / k * 1e3, t_host / k * 1e6, t_wall / k * 1e6

print("n =", n, "plan", direct.accel_plan(n, n))
print("sim.step()            gpu %.1f us/step  host-issue %.1f us  wall %.1f us" % timed(sim.step))
posm = sim._posm
acc = torch.empty_like(sim.accelerations)
ws = sim._ws
print("accel (+finish) only  gpu %.1f us  host %.1f us  wall %.1f" % timed(lambda: direct.accel(posm, n, posm, n, 0, sim._eps2, sim._g, out=acc, workspace=ws)))
L = _lib.lib(); st = _lib.current_stream()
half, dt = direct.f32(0.005), direct.f32(0.01)
args = (sim.positions.data_ptr(), sim.velocities.data_ptr(), sim.accelerations.data_ptr(), sim.accelerations.data_ptr(),
        sim.masses.data_ptr(), n, half, dt, sim._eps2, sim._g, posm.data_ptr(), ws.data_ptr(), ws.numel(), st)
print("raw C step call       gpu %.1f us  host %.1f us  wall %.1f" % timed(lambda: L.nbd_leapfrog_step_f32(*args)))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    a2 = (args[:-1]) + (s.cuda_stream,)
    L.nbd_leapfrog_step_f32(*a2); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(10):
            L.nbd_leapfrog_step_f32(*a2)
print("hipGraph of 10 steps  gpu %.1f us/10  host %.1f  wall %.1f" % timed(g.replay, 20))
