"""Secondary benchmark (BASELINE configs[2] and [3]): ms per surrogate rollout step on one MI355X.
  config 3: GraphModel (published shape: input 4, gnn_dim 64, 2 EdgeConv layers, mean) on N = 4096,
            k = 32 (and k = 50, the reference predict() default)
  config 4: ContinuousConvModel (published shape: dim 128, filter_resolution [6,4], R = 1, mean,
            self loops, encoder [32,64], decoder [64,32]) on N = 16384, positions scaled so that the
            mean radius-1 degree is ~32
Weights: torch.manual_seed(0) default initialisers (no trained weights are published)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT]
import numpy as np
import torch
import gnn, contconv, trainer
from nbd import graphops
from nbd.plummer import generate_plummer


def state(n, seed, scale=1.0):
    p, v, m = generate_plummer(n, seed=seed)
    pos = torch.tensor(p * scale, dtype=torch.float32, device="cuda")
    vel = torch.tensor(v, dtype=torch.float32, device="cuda")
    m1 = torch.tensor(m * n, dtype=torch.float32, device="cuda")[:, None]
    return pos, vel, m1


def _trace(msg):
    if os.environ.get("NBD_TRACE"):
        torch.cuda.synchronize(); print("trace:", msg, file=sys.stderr, flush=True)


def timeit(fn, iters, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, (time.perf_counter() - t0) / iters * 1e3


def degree_scale_for(n, target=32.0, seed=1234):
    """Scale factor on Plummer positions so the mean number of bodies within radius 1 is ~target."""
    pos, _, _ = state(n, seed)
    lo, hi = 0.05, 50.0
    for _ in range(30):
        mid = (lo * hi) ** 0.5
        lists = graphops.radius_lists(pos * mid, 1.0, loop=True, max_num_neighbors=1 << 20 if n <= 4096 else 4096,
                                      transpose=False)
        deg = lists.deg.float().mean().item()
        lo, hi = (mid, hi) if deg > target else (lo, mid)
    return mid, deg


def rollout_mse_vs_direct(model, n, k, steps, dt=0.01, seed=1234):
    """BASELINE 'rollout MSE': surrogate rollout (Trainer.step) against the direct-force HIP integrator
    from the same initial state, true MSE over particles x xyz at the last step. With the random-init
    weights used here this exercises the harness, not a trained model's quality."""
    from galaxify import simulation
    p, v, m = generate_plummer(n, seed=seed)
    sim = simulation.LeapFrogSimulator(positions=p, velocities=v, masses=m, dt=dt, calc_energy=False, device="cuda")
    pos, vel, m1 = state(n, seed)
    tr = trainer.Trainer(model, None, device="cuda", dt=dt)
    kw = {"neighbors": k} if k else {}
    acc = model.predict(pos, torch.cat([vel, m1], 1), **kw)
    for _ in range(steps):
        sim.step()
        pos, vel, acc = tr.step(pos, vel, m1, acc, dt)
    return {"steps": steps, "pos_mse": ((pos - sim.positions) ** 2).mean().item(),
            "vel_mse": ((vel - sim.velocities) ** 2).mean().item(),
            "acc_mse": ((acc - sim.accelerations) ** 2).mean().item(), "weights": "random init (seed 0)"}


def rollout_with_trained_gnn(n=500, steps=200, seed=777):
    """Rollout of the briefly-trained GNN fixture (tests/golden/gnn_small_trained.pt, published shape) on a
    held-out spiral galaxy with the reference's test settings (N = 500, G = 4.5e-6, softening 0.05,
    dt = 1e-4; gnn_experiment.py:24-49,74-78) against the direct-force HIP integrator: true per-step MSE and
    the reference's own statistic (trainer.py:179-195) at the last step."""
    import pandas as pd
    from galaxify import galaxies, simulation
    from nbd.data import Data
    path = os.path.join(ROOT, "tests", "golden", "gnn_small_trained.pt")
    if not os.path.exists(path):
        return None
    model = gnn.GraphModel(input_dim=4, gnn_dim=64, message_passing_steps=2, aggr="mean", device="cuda", neighbors=10,
                           scale_factor=1e6)
    model.load_state_dict(torch.load(path, map_location="cuda"))
    p, v, m = galaxies.generate_spiral(n_bodies=n, total_mass=1.0, radial_scale=3.0, height_scale=0.3, g_const=4.5e-6,
                                       black_hole_mass=0.01, seed=seed)
    sim = simulation.LeapFrogSimulator(positions=p, velocities=v, masses=m, g_const=4.5e-6, softening=0.05, dt=1e-4,
                                       calc_energy=False, device="cuda")
    m1 = sim.masses[:, None]
    xs, ys, st = [], [], []
    for s in range(steps):                       # ground truth: states after each direct-force step
        sim.step()
        xs.append(torch.cat([sim.positions, sim.velocities, m1], 1)); ys.append(sim.accelerations.clone())
        st.append(torch.full((n,), s, device="cuda"))
    data = Data(x=torch.cat(xs), y=torch.cat(ys), step=torch.cat(st))
    tr = trainer.Trainer(model, None, device="cuda", dt=1e-4)
    df = tr.evaluate_rollout("held_out.csv", data, 0, steps, 1e-4, pd.DataFrame(columns=trainer.ROLLOUT_COLUMNS))
    mse = trainer.rollout_mse(df).iloc[-1]
    last = df[df["step"] == steps - 1]
    ref_stat = {}
    for name, trio in (("pos", ["x", "y", "z"]), ("vel", ["vx", "vy", "vz"]), ("acc", ["ax", "ay", "az"])):
        mean_err = [(last[c].astype(float) - last[f"pred_{c}"].astype(float)).mean() for c in trio]
        ref_stat[name] = float(np.sqrt(np.mean(np.square(mean_err))))
    acc_rms = float(torch.cat(ys).pow(2).mean().sqrt())
    return {"n": n, "steps": steps, "weights": "tests/golden/gnn_small_trained.pt (minutes of CPU training)",
            "true_mse_last_step": {"pos": float(mse["pos_mse"]), "vel": float(mse["vel_mse"]), "acc": float(mse["acc_mse"])},
            "reference_statistic_last_step": ref_stat, "ground_truth_acc_rms": acc_rms,
            "mean_step_time_ms": float(df.groupby("step")["step_time"].first().iloc[1:].mean() * 1e3),
            "reference_published_at_step_999": {"pos": 1.24e-12, "vel": 5.37e-10, "acc": 1.63e-8}}


def run(iters=20):
    out = {}
    torch.manual_seed(0)
    model = gnn.GraphModel(input_dim=4, node_encoder_dims=None, gnn_dim=64, message_passing_steps=2, aggr="mean",
                           output_hiddens=None, device="cuda", neighbors=10, scale_factor=1e6)
    tr = trainer.Trainer(model, None, device="cuda", dt=1e-4)
    pos, vel, m1 = state(4096, 1234)
    _trace("gnn model built")
    for k in (32, 50):
        acc = model.predict(pos, torch.cat([vel, m1], 1), neighbors=k)
        st = [pos, vel, acc]
        def step():
            p_, v_ = st[0].clone(), st[1].clone()
            from nbd import direct
            direct.kick_drift(p_, v_, st[2], None, direct.f32(0.5e-4), direct.f32(1e-4))
            a_ = model.predict(p_, torch.cat([v_, m1], 1), neighbors=k)
            direct.kick(v_, a_, direct.f32(0.5e-4))
            st[:] = [p_, v_, a_]
        g_ms, w_ms = timeit(step, iters)
        kn_ms, _ = timeit(lambda: graphops.knn_graph(pos, k), iters)
        out[f"gnn_n4096_k{k}"] = {"rollout_step_ms_gpu": g_ms, "rollout_step_ms_wall": w_ms, "knn_graph_ms": kn_ms,
                                 "edges": 4096 * k}
    _trace("gnn k loops done")
    acc = model.predict(pos, torch.cat([vel, m1], 1))
    g_ms, w_ms = timeit(lambda: tr.step(pos, vel, m1, acc, 1e-4), iters)
    _trace("gnn trainer step timed")
    out["gnn_n4096_trainer_step_k50"] = {"ms_gpu": g_ms, "ms_wall": w_ms}
    adv = tr._capture_step(pos, vel, m1, acc, 1e-4)
    if adv is not None:
        g_ms, w_ms = timeit(lambda: adv(clone=False), iters)
        out["gnn_n4096_trainer_step_k50_hipgraph"] = {"ms_gpu": g_ms, "ms_wall": w_ms}

    _trace("gnn graph timed")
    torch.manual_seed(0)
    cc = contconv.ContinuousConvModel(in_channels=4, out_channels=3, filter_resolution=[6, 4], radius=1.0, agg="mean",
                                      self_loops=True, continuous_conv_layers=2, continuous_conv_dim=128,
                                      encoder_hiddens=[32, 64], encoder_dropout=0.0, decoder_hiddens=[64, 32],
                                      device="cuda", scale_factor=1e6).eval()
    n = 16384
    _trace("contconv model built")
    scale, deg = degree_scale_for(n)
    _trace("degree scale found")
    pos, vel, m1 = state(n, 1234, scale)
    lists = graphops.radius_lists(pos, 1.0, loop=True, max_num_neighbors=32)
    tr2 = trainer.Trainer(cc, None, device="cuda", dt=1e-4)
    acc = cc.predict(pos, torch.cat([vel, m1], 1))
    _trace("contconv first predict")
    g_ms, w_ms = timeit(lambda: tr2.step(pos, vel, m1, acc, 1e-4), max(iters // 2, 3))
    _trace("contconv step timed")
    adv = tr2._capture_step(pos, vel, m1, acc, 1e-4)
    cc_graph = timeit(lambda: adv(clone=False), max(iters // 2, 3)) if adv is not None else (None, None)
    _trace("contconv graph timed")
    r_ms, _ = timeit(lambda: graphops.radius_lists(pos, 1.0, loop=True, max_num_neighbors=32), max(iters // 2, 3))
    out["contconv_n16384"] = {"rollout_step_ms_gpu": g_ms, "rollout_step_ms_wall": w_ms, "rollout_step_ms_gpu_hipgraph": cc_graph[0], "radius_lists_ms": r_ms,
                              "position_scale": scale, "mean_uncapped_degree": deg,
                              "edges_capped": int(lists.rowptr[-1]), "max_in_degree": int((lists.rowptr[1:] - lists.rowptr[:-1]).max())}
    _trace("radius lists timed")
    out["gnn_n4096_rollout_mse_vs_direct"] = rollout_mse_vs_direct(model, 4096, None, 10)
    _trace("random-weight rollout done")
    out["gnn_trained_rollout_n500"] = rollout_with_trained_gnn()
    return out


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 20), indent=1))
