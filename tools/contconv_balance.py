"""Work distribution of the fused ContinuousConv launch at BASELINE configs[3]: MFMA steps (32 packed rows x 128 x 128)
per workgroup from the pair lists' descriptors, and a list-scheduling model of the launch (256 CUs, one workgroup per
CU, workgroups taken in dispatch order) for the fixed cells-per-chunk split and for a steps-balanced split.
    python tools/contconv_balance.py"""
import heapq, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT):
    sys.path.insert(0, _p)
import numpy as np
import torch
import contconv
from nbd import graphops, nnops
from nbd.plummer import generate_plummer
SCALE = 4.599349753792708


def makespan(items, cus=256, fixed=4.0):
    """items: steps per workgroup in dispatch order; cost = fixed + steps (in step-times)."""
    heap = [0.0] * cus
    for s in items:
        t = heapq.heappop(heap)
        heapq.heappush(heap, t + fixed + s)
    return max(heap)


def main():
    n, c = 16384, 128
    p, v, m = generate_plummer(n, seed=1234)
    pos = torch.tensor(p * SCALE, dtype=torch.float32, device="cuda")
    lists = graphops.radius_lists(pos, 1.0, loop=True, max_num_neighbors=32)
    out = {}
    for d in (6, 4):
        layer = contconv.ContinuousConv(c, c, d, radius=1.0, agg="mean").cuda()
        _, cmap, n_cells = layer.cells()
        buf, cap = nnops.contconv_pairs(pos, lists.rowptr, lists.centres, lists.centres.numel(), d, 1.0, cmap, n_cells)
        tiles = (n + 127) // 128
        desc = buf[:tiles * n_cells * 8].view(torch.int32).view(tiles, n_cells, 2).cpu().numpy()
        steps = (desc[:, :, 1] + 31) // 32                      # (tile, cell)
        total = int(steps.sum())
        chunks = min(16, max(1, -(-1024 // tiles)))
        cpc = -(-n_cells // chunks); chunks = -(-n_cells // cpc)
        fixed = [int(steps[t, y * cpc:(y + 1) * cpc].sum()) for y in range(chunks) for t in range(tiles)]   # x fastest
        target = max(8, min(44, -(-total // 1024)))
        bal = []
        for t in range(tiles):
            st = int(steps[t].sum()); ct = min(16, max(1, -(-st // target)))
            cum = np.cumsum(steps[t]); bounds = [0]
            for q in range(1, ct):
                bounds.append(int(np.searchsorted(cum, q * st / ct, side="left")) + 1)
            bounds.append(n_cells)
            bal += [int(steps[t, a:b].sum()) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
        res = {"cells": n_cells, "total_steps": total, "ideal_per_cu": total / 256}
        for name, items in (("fixed_split", fixed), ("balanced_split", bal), ("balanced_heaviest_first", sorted(bal, reverse=True))):
            a = np.array(items)
            res[name] = {"workgroups": len(items), "steps_mean": float(a.mean()), "steps_max": int(a.max()), "steps_min": int(a.min()),
                         "model_makespan_over_ideal": makespan(items) / (total / 256)}
        out[f"D{d}"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
