"""TEMPORARY: per-phase cycle totals of the fused ContinuousConv kernel (NBD_CC_ABLATE=5 instantiation)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT):
    sys.path.insert(0, _p)
os.environ["NBD_CC_ABLATE"] = "5"
import numpy as np, torch, contconv
from nbd import graphops, nnops, _lib
from nbd.plummer import generate_plummer
n, c = 16384, 128
p, v, m = generate_plummer(n, seed=1234)
pos = torch.tensor(p * 4.599349753792708, dtype=torch.float32, device="cuda")
torch.manual_seed(0)
feat = torch.randn(n, c, device="cuda")
lists = graphops.radius_lists(pos, 1.0, loop=True, max_num_neighbors=32)
L = ctypes.CDLL(_lib.LIB_PATH)
for d in (6, 4):
    layer = contconv.ContinuousConv(c, c, d, radius=1.0, agg="mean").cuda()
    _, cmap, n_cells = layer.cells()
    wf = layer.weight_fused()
    pairs = nnops.contconv_pairs(pos, lists.rowptr, lists.centres, lists.centres.numel(), d, 1.0, cmap, n_cells)
    with torch.no_grad():
        for _ in range(3):
            layer(pos, feat, lists=lists, act="tanh", wt=wf, pairs=pairs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        e0.record()
        for _ in range(5):
            layer(pos, feat, lists=lists, act="tanh", wt=wf, pairs=pairs)
        e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"D": d, "event_ms_per_layer_call_incl_finish": e0.elapsed_time(e1) / 5}))
    nwg = 128 * 8
    buf = (ctypes.c_ulonglong * (nwg * 128))()
    assert L.nbd_debug_cc_read(buf, nwg * 128) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 16, 8).astype(np.float64)
    cons, prod = a[:, :8, :], a[:, 8:, :]
    steps = cons[:, 0, 5].sum()
    out = {"D": d, "workgroups": nwg, "steps_total": steps,
           "consumer_cycles_per_step": {"loop_top+B_swap": cons[:, :, 0].sum() / cons[:, :, 5].sum(), "wait_full": cons[:, :, 1].sum() / cons[:, :, 5].sum(),
                                        "mfma": cons[:, :, 2].sum() / cons[:, :, 5].sum(), "scatter": cons[:, :, 3].sum() / cons[:, :, 5].sum()},
           "producer_cycles_per_own_step": {"records_arrive": prod[:, :, 0].sum() / prod[:, :, 5].sum(), "issue_pairs+next_records": prod[:, :, 1].sum() / prod[:, :, 5].sum(),
                                            "wait_done": prod[:, :, 2].sum() / prod[:, :, 5].sum(), "gather_sum": prod[:, :, 3].sum() / prod[:, :, 5].sum(),
                                            "pairs_per_step": prod[:, :, 4].sum() / prod[:, :, 5].sum()},
           "per_wg_consumer_total_cycles": {"mean": float(cons[:, 0, :4].sum(1).mean()), "max": float(cons[:, 0, :4].sum(1).max())},
           "steps_per_wg": {"mean": float(cons[:, 0, 5].mean()), "max": float(cons[:, 0, 5].max())}}
    print(json.dumps(out))
    buf2 = (ctypes.c_ulonglong * (nwg * 8))()
    assert L.nbd_debug_cc_read2(buf2, nwg * 8) == 0
    w = np.frombuffer(buf2, dtype=np.uint64).reshape(nwg, 8).astype(np.float64)
    t0 = w[:, 0].min()
    tick = 1e-2                                   # s_memrealtime: 100 MHz -> 0.01 us per tick
    st, su, le, en = [(w[:, k] - t0) * tick for k in range(4)]
    hw = w[:, 4].astype(np.int64); xcc = w[:, 5].astype(np.int64) & 0xf
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
    cuid = xcc * 1000 + se * 100 + sh * 50 + cu
    print(json.dumps({"D": d, "kernel_span_us": float(en.max()), "wg_setup_us_mean": float((su - st).mean()), "wg_loop_us_mean": float((le - su).mean()),
                      "wg_loop_us_max": float((le - su).max()), "wg_writeout_us_mean": float((en - le).mean()), "wg_total_us_mean": float((en - st).mean()),
                      "distinct_cus": int(len(set(cuid.tolist()))), "wgs_per_cu_min_max": [int(np.bincount(np.unique(cuid, return_inverse=True)[1]).min()), int(np.bincount(np.unique(cuid, return_inverse=True)[1]).max())],
                      "last_wg_start_us": float(st.max()), "first_round_start_spread_us": float(np.sort(st)[255]),
                      "steps_vs_loop_us_corr": float(np.corrcoef(w[:, 6], le - su)[0, 1])}))
    # per-CU busy time and idle gaps
    busy, gaps = [], []
    for cid in set(cuid.tolist()):
        m = cuid == cid
        o = np.argsort(st[m]); s_, e_ = st[m][o], en[m][o]
        busy.append(float((e_ - s_).sum())); gaps.append(float((s_[1:] - e_[:-1]).sum()) if len(s_) > 1 else 0.0)
    print(json.dumps({"per_cu_busy_us": {"mean": float(np.mean(busy)), "min": float(np.min(busy)), "max": float(np.max(busy))},
                      "per_cu_gap_between_wgs_us_mean": float(np.mean(gaps))}))
