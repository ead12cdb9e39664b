"""Phase stamps of contconv_pairs_kernel (probe build: -DNBD_PAIRS_TRACE, linked as tools/build_contconv_trace.sh
does): per workgroup the time of setup / A counts / B scans / B2 / B3 row records / C placement, mean and max over the
128 tiles of each filter resolution at BASELINE configs[3]. Round-2 reading (D = 6): total 42 us mean / 67 max (the
launch lasts as long as its densest tile: one workgroup per CU), placement 23 / 39, counts 7 / 13, row records 5.5 / 9."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT):
    sys.path.insert(0, _p)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import contconv
from nbd import graphops, nnops, _lib
from nbd.plummer import generate_plummer
n, c = 16384, 128
p, v, m = generate_plummer(n, seed=1234)
if len(sys.argv) > 1:                                   # body order, as tools/bench_contconv.py's
    from bench_contconv import ordered_bodies
    p = ordered_bodies(p, sys.argv[1])
pos = torch.tensor(p * 4.599349753792708, dtype=torch.float32, device="cuda")
lists = graphops.radius_lists(pos, 1.0, loop=True, max_num_neighbors=32)
jobs = []
for d in (6, 4):
    layer = contconv.ContinuousConv(c, c, d, radius=1.0, agg="mean").cuda()
    _, cmap, n_cells = layer.cells()
    jobs.append((d, cmap, n_cells))
L = _lib.lib(); L.nbd_debug_pairs_trace.argtypes = [ctypes.c_void_p]; L.nbd_debug_pairs_trace.restype = ctypes.c_int
for _ in range(3): nnops.contconv_pairs_batch(pos, lists.rowptr, lists.centres, lists.centres.numel(), 1.0, jobs)
tr = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
assert L.nbd_debug_pairs_trace(tr.data_ptr()) == 0
nnops.contconv_pairs_batch(pos, lists.rowptr, lists.centres, lists.centres.numel(), 1.0, jobs); torch.cuda.synchronize()
assert L.nbd_debug_pairs_trace(None) == 0
t = tr.view(-1, 8).cpu().numpy().astype(np.float64) / 100.0
names = ["setup", "A_counts", "B_scans", "B2_cells", "B3_rows", "C_place"]
out = {}
for half, sl in (("D6", slice(0, 128)), ("D4", slice(128, 256))):
    tt = t[sl]
    out[half] = {nm: [round(float((tt[:, i + 1] - tt[:, i]).mean()), 2), round(float((tt[:, i + 1] - tt[:, i]).max()), 2)] for i, nm in enumerate(names)}
    out[half]["total_mean_max"] = [round(float((tt[:, 6] - tt[:, 0]).mean()), 2), round(float((tt[:, 6] - tt[:, 0]).max()), 2)]
if os.environ.get("NBD_PAIRS_TRACE_DUMP"):             # per tile: total us, in-edges of the tile
    rp = lists.rowptr.cpu().numpy()
    out["per_tile_D6_total_us"] = [round(float(x), 1) for x in (t[:128, 6] - t[:128, 0])]
    out["per_tile_edges"] = [int(rp[min(n, 128 * (i + 1))] - rp[128 * i]) for i in range(128)]
    out["per_tile_D6_start_us"] = [round(float(x), 1) for x in (t[:128, 0] - t[:, 0].min())]
print(json.dumps(out))
