"""ContinuousConv rollout at BASELINE configs[3] (published model, N = 16 384, dt = 0.01, positions scaled to a mean
radius-1 degree of 32): `steps` captured Trainer steps on an advancing trajectory. The workload of the
--kernel-trace / --pmc passes for configs[3].   python tools/cc_rollout.py [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT, os.path.join(ROOT, "tools")]
import torch
import contconv, trainer
import bench_surrogates as bs

SCALE = 4.599349753792708


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    torch.manual_seed(0)
    cc = contconv.ContinuousConvModel(in_channels=4, out_channels=3, filter_resolution=[6, 4], radius=1.0, agg="mean",
                                      self_loops=True, continuous_conv_layers=2, continuous_conv_dim=128,
                                      encoder_hiddens=[32, 64], encoder_dropout=0.0, decoder_hiddens=[64, 32],
                                      device="cuda", scale_factor=1e6).eval()
    n = 16384
    pos, vel, m1 = bs.state(n, 1234, SCALE)
    tr = trainer.Trainer(cc, None, device="cuda", dt=0.01)
    acc = cc.predict(pos, torch.cat([vel, m1], 1))
    adv = tr._capture_step(pos, vel, m1, acc, 0.01)
    rb0 = cc._radius_cache.rebuilds()
    ms, wall = bs.time_rollout(lambda: adv(clone=False), steps)
    print(json.dumps({"n": n, "steps": steps, "captured_step_ms_gpu": ms, "captured_step_ms_wall": wall,
                      "radius_cache_rebuilds_per_step": (cc._radius_cache.rebuilds() - rb0) / (steps + 3)}))


if __name__ == "__main__":
    main()
