"""Where a training step's HOST time goes (tools/bench_train.py's configurations): wall time of each phase of
`train_graph_batch` as the reference's API sequences them -- collate, zero_grad, forward, loss, backward, optimizer
step, the two .item() read-backs -- enqueue only (no sync inside a phase; one sync at the end of the step), plus the
GPU time of the whole step from HIP events.   python tools/train_phases.py [iters]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch
import gnn, contconv
from bench_train import Batches, make_batch


def phases(model, opt, loader, iters):
    acc = {}
    def lap(name, t0):
        t1 = time.perf_counter()
        acc[name] = acc.get(name, 0.0) + (t1 - t0)
        return t1
    for it in range(iters + 5):
        if it == 5:
            acc.clear()
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        t = time.perf_counter()
        data = loader.next(); t = lap("collate", t)
        opt.zero_grad(); t = lap("zero_grad", t)
        pred = model.forward(data); t = lap("forward", t)
        loss = torch.sqrt(torch.nn.functional.mse_loss(pred * model.scale_factor, data.y * model.scale_factor))
        mse = torch.nn.functional.mse_loss(pred, data.y); t = lap("loss", t)
        loss.backward(); t = lap("backward", t)
        opt.step(); t = lap("optimizer", t)
        loss.item(); mse.item(); t = lap("item_syncs", t)
    ev1.record(); torch.cuda.synchronize()
    out = {k: round(v / iters * 1e3, 4) for k, v in acc.items()}
    out["sum_ms"] = round(sum(acc.values()) / iters * 1e3, 4)
    out["gpu_ms_per_step_incl_idle"] = round(ev0.elapsed_time(ev1) / iters, 4)
    return out


def main(iters=30):
    torch.manual_seed(0)
    res = {}
    m = gnn.GraphModel(input_dim=4, gnn_dim=64, message_passing_steps=2, aggr="mean", neighbors=10, device="cuda", scale_factor=1e6)
    m.train()
    res["gnn"] = phases(m, torch.optim.Adam(m.parameters(), lr=0.01), Batches(make_batch(64, 10)), iters)
    c = contconv.ContinuousConvModel(in_channels=4, out_channels=3, filter_resolution=[6, 4], radius=1.0, agg="mean",
                                     self_loops=True, continuous_conv_layers=2, continuous_conv_dim=128,
                                     encoder_hiddens=[32, 64], decoder_hiddens=[64, 32], device="cuda", scale_factor=1e6)
    res["contconv"] = phases(c, torch.optim.Adam(c.parameters(), lr=0.001), Batches(make_batch(16, 0, seed=100)), iters)
    print(json.dumps(res))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
