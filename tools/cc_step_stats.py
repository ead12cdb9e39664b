"""Pairs per 16-row step of the fused ContinuousConv kernel at BASELINE configs[3] (N = 16 384, D = 6 and D = 4): the
distribution behind the stream kernel's producer / consumer balance (heavy steps = the hub cells around the grid centre in
dense tiles). Reads the pair lists back once; prints one JSON line.   python tools/cc_step_stats.py [out.npz]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT):
    sys.path.insert(0, _p)
import numpy as np
import torch
import contconv
from nbd import graphops, nnops, _lib
from nbd.plummer import generate_plummer

SCALE = 4.599349753792708


def main():
    n, c = 16384, 128
    p, v, m = generate_plummer(n, seed=1234)
    pos = torch.tensor(p * SCALE, dtype=torch.float32, device="cuda")
    lists = graphops.radius_lists(pos, 1.0, loop=True, max_num_neighbors=32)
    L = _lib.lib()
    out, save = {}, {}
    for d in (6, 4):
        layer = contconv.ContinuousConv(c, c, d, radius=1.0, agg="mean").cuda()
        _, cmap, n_cells = layer.cells()
        buf, cap = nnops.contconv_pairs(pos, lists.rowptr, lists.centres, lists.centres.numel(), d, 1.0, cmap, n_cells)
        off = (ctypes.c_size_t * 9)()
        _lib.check(L.nbd_contconv_pairs_layout(n, cap, n_cells, off), "layout")
        tiles = (n + 127) // 128
        raw = buf.cpu().numpy()
        nsteps = raw[off[5]:off[5] + 4 * tiles].view(np.int32)
        rowptr = lists.rowptr.cpu().numpy()
        steps = raw[off[4]:off[5]].view(np.int32).reshape(-1, 4)
        pairs, cells, rows = [], [], []
        for t in range(tiles):
            e_t = int(rowptr[t * 128])
            base = (e_t >> 1) + t * (n_cells + 2)
            rec = steps[base:base + nsteps[t] + 1]
            pairs.append(np.diff(rec[:, 2])); cells.append(rec[:-1, 1] & 0xff); rows.append((rec[:-1, 1] >> 8) & 31)
        pairs, cells, rows = np.concatenate(pairs), np.concatenate(cells), np.concatenate(rows)
        q = [50, 75, 90, 95, 99, 100]
        out[f"D{d}"] = {"steps": int(pairs.size), "pairs": int(pairs.sum()), "mean_pairs_per_step": float(pairs.mean()),
                        "percentiles": dict(zip(map(str, q), [int(x) for x in np.percentile(pairs, q)])),
                        "steps_over_64_pairs_frac": float((pairs > 64).mean()), "pairs_in_steps_over_64_frac": float(pairs[pairs > 64].sum() / pairs.sum()),
                        "mean_rows_per_step": float(rows.mean())}
        save[f"pairs_D{d}"] = pairs; save[f"cells_D{d}"] = cells; save[f"rows_D{d}"] = rows
    if len(sys.argv) > 1:
        np.savez_compressed(sys.argv[1], **save)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
