"""In-kernel trace of the fused ContinuousConv launch at BASELINE configs[3] (N = 16 384, 128 -> 128 channels,
D = 6 and D = 4). Needs the probe build of the library (tools/build_contconv_trace.sh, -DNBD_CC_TRACE): every
workgroup records its start / body end / end (s_memrealtime, 100 MHz), its MFMA steps and (edge, corner) pairs,
the CU it ran on, the time consumer waves 0 / 4 waited on `full` and producer waves 8 / 12 on `done`, and producer
phase times. Prints per layer: launch span, per-workgroup duration and its least-squares fit
fixed + per_step * steps + per_pair * pairs, wait fractions, CU busy fraction.
    tools/build_contconv_trace.sh && gpurun -- 'cp tools/_trace/libnbd_hip_trace.so nbody-deep-sim_amd/csrc/libnbd_hip.so
                                                 && python tools/contconv_trace.py'"""
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
REC = 16                                  # int64 per workgroup record (csrc/contconv_fused.hip, NBD_CC_TRACE)


def main():
    n, c = 16384, 128
    p, v, m = generate_plummer(n, seed=1234)
    pos = torch.tensor(p * SCALE, dtype=torch.float32, device="cuda")
    torch.manual_seed(0)
    feat = torch.randn(n, c, device="cuda")
    lists = graphops.radius_lists(pos, 1.0, loop=True, max_num_neighbors=32)
    L = _lib.lib()
    if not hasattr(L, "nbd_debug_cc_trace"):
        raise SystemExit("libnbd_hip.so was built without -DNBD_CC_TRACE: see tools/build_contconv_trace.sh")
    L.nbd_debug_cc_trace.argtypes = [ctypes.c_void_p]
    L.nbd_debug_cc_trace.restype = ctypes.c_int
    out = {}
    for d in (6, 4):
        layer = contconv.ContinuousConv(c, c, d, radius=1.0, agg="mean").cuda()
        _, cmap, n_cells = layer.cells()
        wf = layer.weight_fused()
        pairs = nnops.contconv_pairs(pos, lists.rowptr, lists.centres, lists.centres.numel(), d, 1.0, cmap, n_cells)
        with torch.no_grad():
            for _ in range(3):
                layer(pos, feat, lists=lists, act="tanh", wt=wf, pairs=pairs)
            tr = torch.zeros(4096 * REC, dtype=torch.int64, device="cuda")
            assert L.nbd_debug_cc_trace(tr.data_ptr()) == 0
            layer(pos, feat, lists=lists, act="tanh", wt=wf, pairs=pairs)
            torch.cuda.synchronize()
            assert L.nbd_debug_cc_trace(None) == 0
        t = tr.view(-1, REC).cpu().numpy()
        idx = np.nonzero(t[:, 0] != 0)[0]
        t = t[idx]
        xcc = (t[:, 5] >> 32) & 0xf
        tab = np.zeros((8, 16), dtype=int)
        for a, b in zip(idx % 8, xcc):
            tab[a, b] += 1
        t0 = t[:, 0].min()
        start, body_end, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0   # us
        steps, prs = t[:, 3], t[:, 4]
        hw = t[:, 5] & 0xffffffff
        cu = ((xcc & 0xf) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
        dur = body_end - start
        A = np.stack([np.ones_like(dur), steps, prs], 1).astype(np.float64)
        coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
        res = {"workgroups": int(len(t)), "launch_span_us": float(end.max()), "dur_mean_us": float(dur.mean()),
               "dur_min_max_us": [float(dur.min()), float(dur.max())], "epilogue_mean_us": float((end - body_end).mean()),
               "fit_us": {"fixed": coef[0], "per_step": coef[1], "per_pair": coef[2]},
               "fit_resid_rms_us": float(np.sqrt(((A @ coef - dur) ** 2).mean())),
               "steps_mean": float(steps.mean()), "pairs_mean": float(prs.mean()), "pairs_max": int(prs.max()),
               "wg_index_mod8_to_xcc_consistency": float(tab.max(axis=1).sum() / len(idx))}
        busy = {}
        for k, a, b in zip(cu, start, end):
            busy.setdefault(int(k), []).append((a, b))
        res["distinct_cus"] = len(busy)
        res["cu_busy_frac_mean"] = float(np.mean([sum(b - a for a, b in v) for v in busy.values()]) / end.max())
        res["cu_last_end_us_p10_p50_p90_max"] = [float(x) for x in np.percentile([max(b for a, b in v) for v in busy.values()],
                                                                                  [10, 50, 90, 100])]
        heavy = prs > np.percentile(prs, 90)
        for name, col in (("consumer0_wait", 6), ("consumer4_wait", 7), ("producer0_wait", 8), ("producer4_wait", 9)):
            wt = t[:, col] / 100.0
            res[name + "_frac_of_dur"] = float((wt / dur).mean())
            res[name + "_frac_densest_10pct"] = float((wt[heavy] / dur[heavy]).mean())
        nlat = max(t[:, 11].sum(), 1)
        res["producer_first_batch_issue_to_summed_us"] = float((t[:, 10] / 100.0).sum() / nlat)
        res["producer_phase_us_prologue__first_stage"] = [float((t[:, 12] / 100.0).sum() / nlat),
                                                           float((t[:, 13] / 100.0).sum() / nlat)]
        out[f"D{d}"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
