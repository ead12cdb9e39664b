"""Phase stamps inside knn_select_staged_kernel (probe build: tools/build_probe.sh graph.hip libnbd_knn_trace.so
-DNBD_KNN_TRACE; run with NBD_LIB_OVERRIDE=tools/_trace/libnbd_knn_trace.so): per wave the kernel entry, the end of its
share of the staging, the barrier, the hint's bound, the end of the scan, of the ranking, of the writes -- for the hinted
search of a rollout step at N = 4096, k = 50."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT):
    sys.path.insert(0, _p)
import numpy as np, torch
from nbd import _lib, graphops
from nbd.plummer import generate_plummer
n, k = 4096, 50
p, v, m = generate_plummer(n, seed=1234)
pos = torch.tensor(p, dtype=torch.float32, device="cuda")
ei = graphops.knn_graph(pos, k, loop=True)
buf = ei.clone()
L = _lib.lib(); L.nbd_debug_knn_trace.argtypes = [ctypes.c_void_p]; L.nbd_debug_knn_trace.restype = ctypes.c_int
pos2 = pos + 1e-4 * torch.randn_like(pos)
WITH_PQ = "--pq" in sys.argv          # the rollout's form: the search kernel also writes the first layer's table rows
x = torch.randn(n, 4, device="cuda"); wpq = torch.randn(128, 4, device="cuda") * 0.3; bpq = torch.randn(64, device="cuda") * 0.1
epq = torch.empty(n, 128, device="cuda")
pq = _lib.KnnPqArgs(); pq.x, pq.ldx, pq.f, pq.h = x.data_ptr(), 4, 4, 64
pq.wpq, pq.bpq, pq.epq, pq.ldepq = wpq.data_ptr(), bpq.data_ptr(), epq.data_ptr(), 128
def hinted():
    if WITH_PQ:
        _lib.check(L.nbd_knn_graph_hint_pq_f32(pos2.data_ptr(), n, k, 1, n * k, buf.data_ptr(), buf.data_ptr(), ctypes.byref(pq), None), "knn")
    else:
        _lib.check(L.nbd_knn_graph_hint_f32(pos2.data_ptr(), n, k, 1, None, None, None, n * k, buf.data_ptr(), buf.data_ptr(), None), "knn")
for _ in range(5): hinted()
torch.cuda.synchronize()
tr = torch.zeros(n * 8, dtype=torch.int64, device="cuda")
assert L.nbd_debug_knn_trace(tr.data_ptr()) == 0
hinted(); torch.cuda.synchronize()
assert L.nbd_debug_knn_trace(None) == 0
t = tr.view(-1, 8).cpu().numpy().astype(np.float64) / 100.0      # s_memrealtime: 100 MHz -> us
t0 = t[:, 0].min()
names = ["entry", "staged", "barrier", "bound", "scan", "rank", "written"]
out = {"at_us_mean": {nm: float((t[:, q] - t0).mean()) for q, nm in enumerate(names)},
       "at_us_max": {nm: float((t[:, q] - t0).max()) for q, nm in enumerate(names)},
       "phase_us_mean": {names[q + 1]: float((t[:, q + 1] - t[:, q]).mean()) for q in range(6)}}
if (t[:, 7] != 0).any():                      # staged kernel: end of the scan loop proper (the rest of "scan" is the expansion)
    out["scan_loop_us_mean"] = float((t[:, 7] - t[:, 3]).mean()); out["expand_us_mean"] = float((t[:, 4] - t[:, 7]).mean())
print(json.dumps(out))
