"""ContinuousConv layer timings at BASELINE configs[3] (N = 16 384, mean radius-1 degree 32, 128 -> 128 channels,
D = 6 and D = 4): pair lists, fused block-sparse layer, and the round-1 formulation (dense binned matrix + GEMM)
on the same graph. HIP events, one JSON line.   python tools/bench_contconv.py [iters] [given|morton|random|dealt|dealt_shuffled|strided|indeg_sorted]
(the body order: as generated, sorted along a Morton curve, or shuffled -- the radius graph's "first 32 by index" rule
makes the edge set depend on it slightly; the question the orders answer is what spatial locality of the tiles buys)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT):
    sys.path.insert(0, _p)
import numpy as np
import torch
import contconv
from nbd import graphops, nnops
from nbd.plummer import generate_plummer

SCALE = 4.599349753792708


def ordered_bodies(p, order):
    """The generated bodies `p` (n, 3) in another order (see the module docstring)."""
    n = p.shape[0]
    if order == "morton":
        q = ((p - p.min(0)) / (p.max(0) - p.min(0)) * 1023.0).astype(np.uint64)
        key = np.zeros(n, dtype=np.uint64)
        for b in range(10):
            for a in range(3):
                key |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
        p = p[np.argsort(key, kind="stable")]
    elif order == "random":
        p = p[np.random.default_rng(5).permutation(n)]
    elif order in ("dealt", "dealt_shuffled"):       # bodies sorted by in-degree and dealt round-robin over the tiles of 128: every tile the same mix
        rp = graphops.radius_lists(torch.tensor(p * SCALE, dtype=torch.float32, device="cuda"), 1.0, loop=True,
                                   max_num_neighbors=32).rowptr.cpu().numpy()
        deg = rp[1:] - rp[:-1]                       # the in-degree: the rows the layers aggregate over (uncapped: up to ~200)
        rank = np.argsort(-deg, kind="stable")
        tiles = n // 128
        new_index = (np.arange(n) % tiles) * 128 + np.arange(n) // tiles      # rank r -> tile r mod tiles, slot r div tiles
        perm = np.empty(n, dtype=np.int64); perm[new_index] = rank
        if order == "dealt_shuffled":                # ... and in random order inside every tile (steps of mixed rows again)
            rs = np.random.default_rng(7)
            perm = np.concatenate([rs.permutation(perm[t * 128:(t + 1) * 128]) for t in range(tiles)])
        p = p[perm]
    return p


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n, c = 16384, 128
    p, v, m = generate_plummer(n, seed=1234)
    order = sys.argv[2] if len(sys.argv) > 2 else "given"
    p = ordered_bodies(p, order if order not in ("strided", "indeg_sorted") else "given")
    pos = torch.tensor(p * SCALE, dtype=torch.float32, device="cuda")
    torch.manual_seed(0)
    feat = torch.randn(n, c, device="cuda")
    lists = graphops.radius_lists(pos, 1.0, loop=True, max_num_neighbors=32)
    edges = int(lists.rowptr[-1])
    if order in ("strided", "indeg_sorted"):
        # the SAME graph (searched in the given labels: the "first 32 by index" rule sees the caller's order) with the
        # bodies relabelled i -> (i mod M) 128 + i div M, M = n / 128: tile t of the layers = bodies t, t + M, t + 2 M, ...
        # (the rule lists low indices most -- in-edges per tile 690 ... 4878 in the given labels -- and a strided tile
        # takes one body from every 128th of the index range)
        M = n // 128
        i = torch.arange(n, device="cuda")
        pi = torch.where(i < 128 * M, (i % M) * 128 + i // M, i)
        if order == "indeg_sorted":                  # ... or by descending in-degree: dense tiles denser, sparse sparser
            rp0 = lists.rowptr.long()
            rank = torch.argsort(rp0[:-1] - rp0[1:], stable=True)       # ascending -(in-degree)
            pi = torch.empty_like(i); pi[rank] = i
        inv = torch.empty_like(pi); inv[pi] = i
        rp = lists.rowptr.long()
        tgt = torch.repeat_interleave(i, rp[1:] - rp[:-1])
        src = lists.centres[:edges].long()
        key = pi[tgt]
        o = torch.argsort(key, stable=True)
        lists.centres = torch.cat([pi[src][o].int(), lists.centres[edges:]]).contiguous()
        cnt = torch.bincount(key, minlength=n)
        lists.rowptr = torch.cat([torch.zeros(1, dtype=torch.long, device="cuda"), torch.cumsum(cnt, 0)]).int().contiguous()
        pos, feat = pos[inv].contiguous(), feat[inv].contiguous()
    out = {"n": n, "edges": edges, "channels": c, "order": order}
    for d in (6, 4):
        layer = contconv.ContinuousConv(c, c, d, radius=1.0, agg="mean").cuda()
        _, cmap, n_cells = layer.cells()
        r2 = float(np.float32(1.0))
        wf, wt = layer.weight_fused(), layer.weight_t()
        pairs = nnops.contconv_pairs(pos, lists.rowptr, lists.centres, lists.centres.numel(), d, r2, cmap, n_cells)
        with torch.no_grad():
            t_pairs = timeit(lambda: nnops.contconv_pairs(pos, lists.rowptr, lists.centres, lists.centres.numel(), d, r2,
                                                          cmap, n_cells), iters)
            t_fused = timeit(lambda: layer(pos, feat, lists=lists, act="tanh", wt=wf, pairs=pairs), iters)
            y_f = layer(pos, feat, lists=lists, act="tanh", wt=wf, pairs=pairs)
            layer.use_fused = False
            t_old = timeit(lambda: layer(pos, feat, lists=lists, act="tanh", wt=wt), max(iters // 4, 3))
            y_o = layer(pos, feat, lists=lists, act="tanh", wt=wt)
        st = nnops.contconv_pairs_stats(pairs[0], n, pairs[1], n_cells)
        out[f"D{d}"] = {"cells": n_cells, "steps": st["steps"], "cost": st["cost"], "pairs_ms": t_pairs, "fused_layer_ms": t_fused, "binned_gemm_layer_ms": t_old,
                        "fused_vs_binned_max_abs_diff": float((y_f - y_o).abs().max())}
    jobs = []
    for d in (6, 4):
        layer = contconv.ContinuousConv(c, c, d, radius=1.0, agg="mean").cuda()
        _, cmap, n_cells = layer.cells()
        jobs.append((d, cmap, n_cells))
    out["pairs_both_resolutions_one_launch_ms"] = timeit(
        lambda: nnops.contconv_pairs_batch(pos, lists.rowptr, lists.centres, lists.centres.numel(), 1.0, jobs), iters)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
