"""Per-step timeline of ONE workgroup of the fused ContinuousConv kernel (probe build, tools/build_contconv_trace.sh):
consumer waves: step entered / buffer full / step done; producer waves: step begun / buffer asked for / got / published.
Prints where the time between consecutive consumer steps goes and what the ring looked like.
    NBD_LIB_OVERRIDE=tools/_trace/libnbd_hip_trace.so python tools/contconv_timeline.py [D] [workgroup] [out.npz]"""
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
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    wg = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    n, c = 16384, 128
    p, v, m = generate_plummer(n, seed=1234)
    pos = torch.tensor(p * SCALE, dtype=torch.float32, device="cuda")
    torch.manual_seed(0)
    feat = torch.randn(n, c, device="cuda")
    lists = graphops.radius_lists(pos, 1.0, loop=True, max_num_neighbors=32)
    L = _lib.lib()
    L.nbd_debug_cc_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.nbd_debug_cc_timeline.restype = ctypes.c_int
    layer = contconv.ContinuousConv(c, c, d, radius=1.0, agg="mean").cuda()
    _, cmap, n_cells = layer.cells()
    wf = layer.weight_fused()
    pairs = nnops.contconv_pairs(pos, lists.rowptr, lists.centres, lists.centres.numel(), d, 1.0, cmap, n_cells)
    with torch.no_grad():
        for _ in range(3):
            layer(pos, feat, lists=lists, act="tanh", wt=wf, pairs=pairs)
        tl = torch.zeros(16 * 512 * 4, dtype=torch.int64, device="cuda")
        assert L.nbd_debug_cc_timeline(tl.data_ptr(), wg) == 0
        layer(pos, feat, lists=lists, act="tanh", wt=wf, pairs=pairs)
        torch.cuda.synchronize()
        assert L.nbd_debug_cc_timeline(None, -1) == 0
    t = tl.view(16, 512, 4).cpu().numpy().astype(np.float64)
    ns = int((t[0, :, 0] > 0).sum())
    t0 = t[t > 0].min()
    t = np.where(t > 0, (t - t0) / 100.0, np.nan)              # us
    out = {"D": d, "workgroup": wg, "steps": ns, "span_us": float(np.nanmax(t))}
    for w in (0, 4):
        enter, full, done = t[w, :ns, 0], t[w, :ns, 1], t[w, :ns, 2]
        out[f"consumer{w}"] = {"wait_full_us": float(np.nansum(full - enter)), "in_step_us": float(np.nansum(done - full)),
                               "between_steps_us": float(np.nansum(enter[1:] - done[:-1])),
                               "step_us_p50_p90_max": [float(np.nanpercentile(done - full, q)) for q in (50, 90, 100)],
                               "wait_us_p50_p90_max": [float(np.nanpercentile(full - enter, q)) for q in (50, 90, 100)]}
    prod = {}
    for pw in range(8, 16):
        q = np.arange(pw - 8, ns, 8)
        beg, ask, got, pub = (t[pw, q, k] for k in range(4))
        prod[pw] = {"steps": int(len(q)), "gather_before_claim_us": float(np.nanmean(ask - beg)), "claim_wait_us": float(np.nanmean(got - ask)),
                    "after_claim_us": float(np.nanmean(pub - got)), "idle_between_us": float(np.nanmean(beg[1:] - pub[:-1]))}
    out["producers_mean_per_step"] = {k: float(np.mean([prod[pw][k] for pw in prod])) for k in ("gather_before_claim_us", "claim_wait_us", "after_claim_us", "idle_between_us")}
    # how far ahead of the consumer were the published steps when consumer wave 0 entered step q?
    pub_all = np.full(ns, np.nan)
    for pw in range(8, 16):
        q = np.arange(pw - 8, ns, 8); pub_all[q] = t[pw, q, 3]
    ahead = [int(np.sum(pub_all[q:q + 8] <= t[0, q, 0])) for q in range(ns)]
    out["published_ahead_when_consumer0_enters_hist"] = np.bincount(ahead, minlength=8).tolist()
    lag = t[0, :ns, 1] - pub_all                                # consumer sees `full` this long after publication
    out["full_seen_after_publish_us_p50_p90"] = [float(np.nanpercentile(lag, q)) for q in (50, 90)]
    if len(sys.argv) > 3:
        np.savez_compressed(sys.argv[3], t=t)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
