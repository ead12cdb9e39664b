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
REC = 72                                  # int64 per workgroup record (csrc/contconv_fused.hip, NBD_CC_TRACE)


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
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                layer(pos, feat, lists=lists, act="tanh", wt=wf, pairs=pairs)
            e1.record(); torch.cuda.synchronize()
            layer_ms = e0.elapsed_time(e1) / 10
            tr = torch.zeros(1024 * REC, dtype=torch.int64, device="cuda")
            assert L.nbd_debug_cc_trace(tr.data_ptr()) == 0
            layer(pos, feat, lists=lists, act="tanh", wt=wf, pairs=pairs)
            torch.cuda.synchronize()
            assert L.nbd_debug_cc_trace(None) == 0
        t = tr.view(-1, REC).cpu().numpy()
        t = t[t[:, 0] != 0]
        t0 = t[:, 0].min()
        start, end = (t[:, 0] - t0) / 100.0, (t[:, 2] - t0) / 100.0            # us (s_memrealtime: 100 MHz)
        steps = t[:, 1]
        dur = end - start
        res = {"layer_ms_untraced_loop": layer_ms, "workgroups": int(len(t)), "launch_span_us": float(end.max()),
               "start_spread_us": float(start.max()), "dur_mean_us": float(dur.mean()),
               "dur_min_max_us": [float(dur.min()), float(dur.max())], "steps_total": int(steps.sum()),
               "steps_per_wg_min_max": [int(steps.min()), int(steps.max())],
               "us_per_step_mean": float((dur / np.maximum(steps, 1)).mean()),
               "mfma_bound_us_per_step_at_2.1GHz": 16 * 128 * 128 * 2 / (256 * 2.1e9) * 1e6}
        w = t[:, 8:72].reshape(-1, 16, 4) / 100.0          # us: [wave][flags, steps, table loads, whole role]
        slow = dur > np.percentile(dur, 90)
        for name, sel in (("all", slice(None)), ("slowest_10pct", slow)):
            ws = w[sel]
            res[name] = {"dur_us": float(dur[sel].mean()),
                         "consumer_wave_us_flags_steps_tables_role": [[float(x) for x in ws[:, k].mean(0)] for k in (0, 3, 4, 7)],
                         "producer_wave_us_flags_steps_tables_role": [[float(x) for x in ws[:, k].mean(0)] for k in (8, 11, 12, 15)]}
        out[f"D{d}"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
