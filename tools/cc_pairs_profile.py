"""TEMPORARY: phase times of contconv_pairs_kernel per workgroup."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT):
    sys.path.insert(0, _p)
import numpy as np, torch, contconv
from nbd import graphops, nnops, _lib
from nbd.plummer import generate_plummer
n = 16384
p, v, m = generate_plummer(n, seed=1234)
pos = torch.tensor(p * 4.599349753792708, dtype=torch.float32, device="cuda")
lists = graphops.radius_lists(pos, 1.0, loop=True, max_num_neighbors=32)
L = ctypes.CDLL(_lib.LIB_PATH)
for d in (6, 4):
    layer = contconv.ContinuousConv(128, 128, d, radius=1.0, agg="mean").cuda()
    _, cmap, n_cells = layer.cells()
    for _ in range(3):
        nnops.contconv_pairs(pos, lists.rowptr, lists.centres, lists.centres.numel(), d, 1.0, cmap, n_cells)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (128 * 8))()
    assert L.nbd_debug_cc_read3(buf, 128 * 8) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(128, 8).astype(np.float64)
    us = a[:, :5] * 0.01
    names = ["A_count", "B_scan_nodes", "B2_scan_cells", "B3_rows", "C_place"]
    print(json.dumps({"D": d, "phase_us_mean": dict(zip(names, us.mean(0).round(2).tolist())), "phase_us_max": dict(zip(names, us.max(0).round(2).tolist())),
                      "wg_total_us": {"mean": float(us.sum(1).mean()), "max": float(us.sum(1).max())}, "edges_per_tile": {"mean": float(a[:, 5].mean()), "max": float(a[:, 5].max())},
                      "kernel_span_us": float((a[:, 7].max() - a[:, 6].min()) * 0.01), "corr_edges_C": float(np.corrcoef(a[:, 5], us[:, 4])[0, 1])}))
