"""Fits the fused ContinuousConv kernel's per-workgroup duration (probe build, tools/build_contconv_trace.sh) to what the
workgroup's range holds -- steps, pairs, pairs above a threshold, cell changes -- at BASELINE configs[3] (N = 16 384,
128 -> 128 channels, D = 6 and D = 4), and evaluates per-step cost functions for the plan kernel's cuts by the makespan
the fitted model predicts for them.    NBD_LIB_OVERRIDE=tools/_trace/libnbd_trace.so python tools/contconv_cost_fit.py

Written for round 3's single step sequence (workgroup w = the w-th contiguous range). Since round 4 a layer with >= 32 filter
cells is cut into cell groups (workgroup w works in group w mod groups): the per-workgroup features below are then attributed
to the wrong steps -- use it on layers with one group only, or read tools/contconv_timeline.py instead."""
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
REC = 72


def step_pairs(pairs_buf, rowptr, n, cap_e, n_cells):
    """pairs of every step of the global step sequence (tile-major), and the cell of every step"""
    L = _lib.lib()
    off = (ctypes.c_size_t * 9)()
    _lib.check(L.nbd_contconv_pairs_layout(n, cap_e, n_cells, off), "layout")
    tiles = (n + 127) // 128
    raw = pairs_buf.cpu().numpy()
    nsteps = raw[off[5]:off[5] + 4 * tiles].view(np.int32)
    steps = raw[off[4]:off[5]].view(np.int32).reshape(-1, 4)
    rp = rowptr.cpu().numpy()
    out_p, out_c = [], []
    for t in range(tiles):
        e_t = int(rp[t * 128])
        base = (e_t >> 1) + t * (n_cells + 2)
        ns = int(nsteps[t])
        rec = steps[base:base + ns + 1]
        out_p.append(np.diff(rec[:, 2]))
        out_c.append(rec[:ns, 1] & 0xff)
    return np.concatenate(out_p), np.concatenate(out_c), nsteps


def main():
    n, c = 16384, 128
    p, v, m = generate_plummer(n, seed=1234)
    pos = torch.tensor(p * SCALE, dtype=torch.float32, device="cuda")
    torch.manual_seed(0)
    feat = torch.randn(n, c, device="cuda")
    lists = graphops.radius_lists(pos, 1.0, loop=True, max_num_neighbors=32)
    L = _lib.lib()
    if not hasattr(L, "nbd_debug_cc_trace"):
        raise SystemExit("needs the probe build: tools/build_contconv_trace.sh, NBD_LIB_OVERRIDE=tools/_trace/libnbd_trace.so")
    L.nbd_debug_cc_trace.argtypes = [ctypes.c_void_p]; L.nbd_debug_cc_trace.restype = ctypes.c_int
    out = {}
    for d in (6, 4):
        layer = contconv.ContinuousConv(c, c, d, radius=1.0, agg="mean").cuda()
        _, cmap, n_cells = layer.cells()
        wf = layer.weight_fused()
        pairs = nnops.contconv_pairs(pos, lists.rowptr, lists.centres, lists.centres.numel(), d, 1.0, cmap, n_cells)
        with torch.no_grad():
            for _ in range(3):
                layer(pos, feat, lists=lists, act="tanh", wt=wf, pairs=pairs)
            durs = []
            for rep in range(5):
                tr = torch.zeros(1024 * REC, dtype=torch.int64, device="cuda")
                assert L.nbd_debug_cc_trace(tr.data_ptr()) == 0
                layer(pos, feat, lists=lists, act="tanh", wt=wf, pairs=pairs)
                torch.cuda.synchronize()
                assert L.nbd_debug_cc_trace(None) == 0
                t = tr.view(-1, REC).cpu().numpy()[:256]
                durs.append((t[:, 2] - t[:, 0]) / 100.0)
                wg_steps = t[:, 1].astype(np.int64)
        dur = np.median(np.stack(durs), 0)                     # us per workgroup, median of 5 launches
        sp, sc, nsteps = step_pairs(pairs[0], lists.rowptr, n, pairs[1], n_cells)
        assert wg_steps.sum() == sp.size, (wg_steps.sum(), sp.size)
        cuts = np.concatenate([[0], np.cumsum(wg_steps)])
        feats = []
        for w in range(256):
            a, b = cuts[w], cuts[w + 1]
            q = sp[a:b].astype(np.float64)
            cells = sc[a:b]
            feats.append([1.0, b - a, q.sum(), np.maximum(q - 64, 0).sum(), np.maximum(q - 128, 0).sum(),
                          np.maximum(q - 256, 0).sum(), float((np.diff(cells) != 0).sum())])
        X = np.array(feats)
        names = ["fixed_us", "per_step", "per_pair", "per_pair_above_64", "per_pair_above_128", "per_pair_above_256", "per_cell_change"]
        res = {}
        for cols in ([0, 1, 2], [0, 1, 2, 6], [0, 1, 2, 4], [0, 1, 2, 3, 4, 5, 6]):
            coef, *_ = np.linalg.lstsq(X[:, cols], dur, rcond=None)
            pred = X[:, cols] @ coef
            res["+".join(names[k] for k in cols)] = {"coef": {names[k]: float(x) for k, x in zip(cols, coef)},
                                                      "rms_residual_us": float(np.sqrt(((pred - dur) ** 2).mean()))}
        # step-level model from the full fit (per-step time as a function of its pairs), then candidate cost functions
        coef, *_ = np.linalg.lstsq(X[:, [0, 1, 2, 4, 6]], dur, rcond=None)
        f0, a1, a2, a4, a6 = coef
        change = np.concatenate([[0], (np.diff(sc) != 0).astype(np.float64)])
        t_step = a1 + a2 * sp + a4 * np.maximum(sp - 128, 0) + a6 * change        # predicted us of every step
        def makespan(cost):
            cum = np.concatenate([[0], np.cumsum(cost)])
            tot = cum[-1]
            cc = [0]
            for w in range(1, 256):
                B = tot * w // 256
                cc.append(int(np.searchsorted(cum[:-1], B, side="left")))
            cc.append(sp.size)
            tw = np.array([t_step[cc[w]:cc[w + 1]].sum() for w in range(256)])
            return float(tw.max() + f0), float(tw.mean() + f0)
        cands = {"max(80,p)": np.maximum(80, sp), "max(96,p)": np.maximum(96, sp), "max(64,p)": np.maximum(64, sp),
                 "fitted_t_step": np.maximum(t_step, 1e-3) * 1000, "max(80,p)+0.5*max(p-128,0)": np.maximum(80, sp) + 0.5 * np.maximum(sp - 128, 0),
                 "max(80,p)+max(p-128,0)": np.maximum(80, sp) + np.maximum(sp - 128, 0),
                 "max(72,p)+8*cell_change": np.maximum(72, sp) + 8 * change}
        out[f"D{d}"] = {"workgroup_us_mean_min_max": [float(dur.mean()), float(dur.min()), float(dur.max())], "fits": res,
                        "predicted_makespan_and_mean_us": {k: makespan(v.astype(np.int64)) for k, v in cands.items()},
                        "pairs_per_step_percentiles_50_90_99_max": [float(np.percentile(sp, q)) for q in (50, 90, 99, 100)]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
