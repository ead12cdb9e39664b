"""LayerNorm + decoder at the published ContinuousConv shape (N = 16 384, 256 -> 64 -> 32 -> 3): the one-launch kernel
(nbd_ln_mlp_head_f32) against the four launches it replaces. HIP events, graph replay not used.   python tools/ubench_head.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT]
import torch
from gnn import head_chain, run_chain
from nbd import nnops


def timeit(fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    out = {}
    for n, dims in ((16384, [256, 64, 32, 3]), (4096, [68, 3]), (4096, [128, 64, 3])):
        torch.manual_seed(0)
        x = torch.randn(n, dims[0], device="cuda")
        ln = torch.nn.LayerNorm(dims[0]).cuda()
        layers = []
        for i in range(len(dims) - 1):
            layers.append(torch.nn.Linear(dims[i], dims[i + 1]))
            if i < len(dims) - 2:
                layers.append(torch.nn.Tanh())
        mlp = (torch.nn.Sequential(*layers) if len(layers) > 1 else layers[0]).cuda()
        head = head_chain(mlp)
        g, b = ln.weight.detach(), ln.bias.detach()
        plan = nnops.ln_mlp_head_plan(dims[0], head, g, b)
        buf = torch.empty(n, dims[-1], device="cuda")
        fused = timeit(lambda: nnops.ln_mlp_head(x, g, b, ln.eps, plan, out=buf))
        sep = timeit(lambda: run_chain(nnops.layernorm(x, g, b, ln.eps), head))
        out[f"n{n}_" + "x".join(map(str, dims))] = {"one_launch_us": fused, "separate_launches_us": sep}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
