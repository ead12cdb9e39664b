"""Backward kernels (csrc/train.hip) and the training step of the surrogates (gnn.py:150-191,
trainer.py:20-92) against torch autograd on the CPU oracle (oracle/surrogate_oracle.py), through the
C-ABI. Gradients: within 1e-5 of the oracle's measured on the whole tensor (fp32, different summation
order); the transposed adjacency is index-exact."""
import numpy as np
import pytest
import torch

from conftest import global_rel

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _plummer(n, seed):
    from nbd.plummer import generate_plummer
    p, v, m = generate_plummer(n, seed=seed)
    return (torch.tensor(p, dtype=torch.float32), torch.tensor(v, dtype=torch.float32),
            torch.tensor(m, dtype=torch.float32))


# ------------------------------------------------------------------ building blocks
@pytest.mark.parametrize("n,e,dups", [(1, 0, False), (7, 30, True), (500, 5000, False), (3000, 40000, True)])
def test_csr_by_key_is_the_sorted_transpose(n, e, dups, gpu_device):
    from nbd import graphops
    g = torch.Generator().manual_seed(n + e)
    key = torch.randint(0, n, (e,), generator=g)
    val = torch.randint(0, n if dups else 10 ** 6, (e,), generator=g)
    rowptr, out = graphops.csr_by_key(key.cuda(), val.cuda(), n)
    order = np.lexsort((val.numpy(), key.numpy()))
    assert torch.equal(out.cpu().to(torch.int64), val[order])
    counts = torch.bincount(key, minlength=n)
    assert torch.equal(rowptr.cpu().to(torch.int64), torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)]))
    with pytest.raises(Exception):
        graphops.csr_by_key(torch.tensor([0, n], dtype=torch.int64).cuda(), torch.tensor([0, 0]).cuda(), n)


@pytest.mark.parametrize("n,m,k", [(1, 3, 4), (100, 3, 68), (4096, 64, 64), (5000, 128, 8), (333, 70, 130),
                                   (70000, 128, 128), (0, 5, 6)])
def test_linear_backward_matches_torch(n, m, k, gpu_device):
    from nbd import autograd as ag
    g = torch.Generator().manual_seed(n + m + k)
    x = torch.randn(n, k, generator=g)
    w = torch.randn(m, k, generator=g) / k ** 0.5
    b = torch.randn(m, generator=g)
    brs = torch.rand(n, generator=g)
    dy = torch.randn(n, m, generator=g)
    for act in (None, "tanh"):
        for use_brs in (False, True):
            xr, wr, br = (t.clone().double().requires_grad_() for t in (x, w, b))
            pre = xr @ wr.T + (brs.double()[:, None] if use_brs else 1.0) * br
            ref = torch.tanh(pre) if act else pre
            ref.backward(dy.double())
            xg, wg, bg = (t.clone().cuda().requires_grad_() for t in (x, w, b))
            y = ag.linear(xg, wg, bg, act=act, bias_rowscale=brs.cuda() if use_brs else None)
            y.backward(dy.cuda())
            if n:
                assert global_rel(y.detach().cpu(), ref.detach().float()) < TOL
                assert global_rel(xg.grad.cpu(), xr.grad.float()) < TOL
            assert wg.grad.shape == (m, k) and bg.grad.shape == (m,)
            if n:
                assert global_rel(wg.grad.cpu(), wr.grad.float()) < TOL
                assert global_rel(bg.grad.cpu(), br.grad.float()) < TOL
            else:
                assert float(wg.grad.abs().max()) == 0.0 and float(bg.grad.abs().max()) == 0.0


def test_linear_backward_is_deterministic_and_takes_strided_views(gpu_device):
    from nbd import autograd as ag
    g = torch.Generator().manual_seed(0)
    wide = torch.randn(9000, 200, generator=g).cuda()
    x = wide[:, 10:78]                       # a column slice: row stride 200
    w = (torch.randn(40, 68, generator=g) / 8).cuda().requires_grad_()
    outs = []
    for _ in range(2):
        w.grad = None
        ag.linear(x, w, None, act="tanh").square().sum().backward()
        outs.append(w.grad.clone())
    assert torch.equal(outs[0], outs[1])
    wr = w.detach().cpu().double().requires_grad_()
    torch.tanh(x.cpu().double() @ wr.T).square().sum().backward()
    assert global_rel(outs[0].cpu(), wr.grad.float()) < TOL


@pytest.mark.parametrize("n,c", [(1, 5), (100, 68), (4097, 128), (300, 200), (0, 7)])
def test_layernorm_backward_matches_torch(n, c, gpu_device):
    from nbd import autograd as ag
    g = torch.Generator().manual_seed(n + c)
    x = torch.randn(n, c, generator=g) * 3 + 1
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    dy = torch.randn(n, c, generator=g)
    xr, gr, br = (t.clone().double().requires_grad_() for t in (x, gamma, beta))
    torch.nn.functional.layer_norm(xr, (c,), gr, br, 1e-5).backward(dy.double())
    xg, gg, bg = (t.clone().cuda().requires_grad_() for t in (x, gamma, beta))
    ag.LayerNormFn.apply(xg, gg, bg, 1e-5).backward(dy.cuda())
    if n:
        assert global_rel(xg.grad.cpu(), xr.grad.float()) < TOL
        assert global_rel(gg.grad.cpu(), gr.grad.float()) < TOL
        assert global_rel(bg.grad.cpu(), br.grad.float()) < TOL
    else:
        assert float(gg.grad.abs().max()) == 0.0 and float(bg.grad.abs().max()) == 0.0


@pytest.mark.parametrize("aggr", ["sum", "mean"])
@pytest.mark.parametrize("regular", [True, False])
def test_edge_aggregate_backward_matches_autograd(aggr, regular, gpu_device):
    from nbd import autograd as ag
    from nbd import graphops
    from oracle import surrogate_oracle as so
    n, h, k = 600, 70, 9
    pos, _, _ = _plummer(n, 3)
    ei = so.knn_graph(pos, k)
    if not regular:
        keep = torch.rand(ei.shape[1], generator=torch.Generator().manual_seed(1)) < 0.6
        keep &= ei[1] != 5                                  # a target without edges
        ei = ei[:, keep]
    g = torch.Generator().manual_seed(2)
    pq = torch.randn(n, 2 * h, generator=g)
    ds = torch.randn(n, h, generator=g)
    pr = pq.clone().double().requires_grad_()
    msg = torch.tanh(pr[ei[1], :h] + pr[ei[0], h:])
    ref = so.scatter(msg, ei[1], n, aggr)
    ref.backward(ds.double())
    pg = pq.clone().cuda().requires_grad_()
    if regular:
        lists = ag.EdgeLists(n, None, ei[0].contiguous().cuda(), k)
    else:
        rowptr, src = graphops.csr_by_target(ei.cuda(), n)
        lists = ag.EdgeLists(n, rowptr, src, -1)
    s = ag.EdgeAggregateFn.apply(pg, lists, h, aggr)
    s.backward(ds.cuda())
    assert global_rel(s.detach().cpu(), ref.detach().float()) < TOL
    assert global_rel(pg.grad.cpu(), pr.grad.float()) < TOL
    pg2 = pq.clone().cuda().requires_grad_()
    ag.EdgeAggregateFn.apply(pg2, lists, h, aggr).backward(ds.cuda())
    assert torch.equal(pg.grad, pg2.grad)                    # fixed summation order


# ------------------------------------------------------------------ whole models
GNN_CFGS = [
    dict(input_dim=4, gnn_dim=64, message_passing_steps=2, aggr="mean", neighbors=10),            # published shape
    dict(input_dim=4, gnn_dim=32, message_passing_steps=3, aggr="sum", neighbors=5, output_hiddens=[16, 8],
         scale_factor=7.0),
    dict(input_dim=7, gnn_dim=48, message_passing_steps=1, aggr="mean", neighbors=8, node_encoder_dims=[20, 24]),
    dict(input_dim=4, gnn_dim=40, message_passing_steps=2, aggr="max", neighbors=9),
]


def _gnn_pair(cfg, seed):
    import gnn
    from oracle import surrogate_oracle as so
    torch.manual_seed(seed)
    ora = so.GraphModelOracle(**{k: v for k, v in cfg.items() if k != "scale_factor"})
    model = gnn.GraphModel(device="cuda", **cfg)
    model.load_state_dict(ora.state_dict(), strict=True)
    return model, ora


def _oracle_loss(ora, x7, ei, y, sf):
    pred = ora.forward_graph(x7, ei)
    return torch.sqrt(torch.nn.functional.mse_loss(pred * sf, y * sf)), torch.nn.functional.mse_loss(pred, y)


@pytest.mark.parametrize("cfg", GNN_CFGS)
def test_gnn_gradients_match_oracle_autograd(cfg, gpu_device):
    from nbd.data import Data
    from oracle import surrogate_oracle as so
    model, ora = _gnn_pair(cfg, 3)
    n = 500
    pos, vel, m = _plummer(n, 8)
    x7 = torch.cat([pos, vel, m[:, None] * n], 1)
    y = torch.randn(n, cfg.get("output_dim", 3), generator=torch.Generator().manual_seed(5)) * 0.3
    ei = so.knn_graph(pos, cfg["neighbors"])
    keep = torch.rand(ei.shape[1], generator=torch.Generator().manual_seed(3)) < 0.8
    for edges, regular in ((ei, True), (ei[:, keep], False)):
        ora.zero_grad(); model.zero_grad()
        ora.train(); model.train()
        lo, mo = _oracle_loss(ora, x7, edges, y, cfg.get("scale_factor", 1))
        lo.backward()
        d = Data(x=x7.cuda(), edge_index=edges.cuda(), y=y.cuda())
        if regular:
            d._regular_k = cfg["neighbors"]
        lg, mg = model.compute_loss(d)
        lg.backward()
        assert abs(lg.item() - lo.item()) < 1e-5 * abs(lo.item()) and abs(mg.item() - mo.item()) < 1e-5 * abs(mo.item())
        ref = dict(ora.named_parameters())
        for name, p in model.named_parameters():
            assert p.grad is not None, name
            r = ref[name].grad
            scale = float(r.norm())
            # a whole-tensor bound; gradients that are ~0 against the loss scale are compared absolutely
            assert float((p.grad.cpu() - r).norm()) <= TOL * max(scale, 1e-3 * lo.item()), (name, regular)


def test_gnn_training_follows_the_oracle(gpu_device):
    """Adam for 25 steps on both sides from the same state: the loss curves stay together and fall."""
    from nbd.data import Data
    from oracle import surrogate_oracle as so
    cfg = GNN_CFGS[0]
    model, ora = _gnn_pair(cfg, 11)
    n = 400
    pos, vel, m = _plummer(n, 9)
    x7 = torch.cat([pos, vel, m[:, None] * n], 1)
    from oracle import galaxify_oracle as go
    y = go.accelerations(pos, m, 1.0, 0.1)
    ei = so.knn_graph(pos, cfg["neighbors"])
    d = Data(x=x7.cuda(), edge_index=ei.cuda(), y=y.cuda())
    opt_g = torch.optim.Adam(model.parameters(), lr=2e-3)
    opt_o = torch.optim.Adam(ora.parameters(), lr=2e-3)
    lg, lo = [], []
    for _ in range(25):
        loss, mse = model.train_graph_batch(opt_g, d)
        lg.append(loss)
        ora.train(); opt_o.zero_grad()
        l, _ = _oracle_loss(ora, x7, ei, y, 1)
        l.backward(); opt_o.step()
        lo.append(l.item())
    assert lg[-1] < 0.7 * lg[0]
    assert max(abs(a - b) / b for a, b in zip(lg, lo)) < 2e-3
    assert model.training


def test_train_batch_builds_k50_graphs(gpu_device):
    model, _ = _gnn_pair(GNN_CFGS[0], 4)
    pos = torch.stack([_plummer(128, s)[0] for s in (1, 2, 3)]).cuda()
    vel = torch.stack([_plummer(128, s)[1] for s in (1, 2, 3)]).cuda()
    feat = torch.cat([vel, torch.full((3, 128, 1), 1.0, device="cuda")], -1)
    acc = torch.randn(3, 128, 3, generator=torch.Generator().manual_seed(0)).cuda()
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    before = [p.detach().clone() for p in model.parameters()]
    loss, mse = model.train_batch(opt, pos, feat, acc)
    assert np.isfinite(loss) and np.isfinite(mse) and abs(loss - mse ** 0.5) < 1e-5 * loss
    assert any(not torch.equal(a, b) for a, b in zip(before, model.parameters()))


def test_train_from_dir_learns_and_checkpoints(tmp_path, gpu_device):
    """trainer.py:20-92 end to end: dataset CSV written by the HIP integrator -> datautils graph batches ->
    train_graph_batch (HIP forward + backward) with Adam + ReduceLROnPlateau -> model_{epoch}.pt files that
    a second Trainer resumes from."""
    import importlib.util
    import os
    import gnn
    import trainer
    from conftest import PKG
    path = tmp_path / "data"
    path.mkdir()
    spec = importlib.util.spec_from_file_location("s01", f"{PKG}/s01-dataset-generation.py")
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    cli.main(["--integrator", "leapfrog", "--n-bodies", "40", "57", "--sim-type", "spiral", "--steps", "6",
              "--dt", "0.01", "--g", "1.0", "--softening", "0.1", "--seed", "3", "--output",
              str(path / "output_file_1.csv"), "--device", "cuda"])
    torch.manual_seed(0)
    model = gnn.GraphModel(input_dim=4, gnn_dim=32, message_passing_steps=2, aggr="mean", neighbors=6, device="cuda",
                           scale_factor=10.0)
    opt = torch.optim.Adam(model.parameters(), lr=3e-3)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=50)
    tr = trainer.Trainer(model, opt, scheduler=sched, device="cuda", dt=0.01)
    save = tmp_path / "models"
    save.mkdir()
    losses, mses = tr.train_from_dir(str(path), epochs=12, batch_size=4, save_every=5, save_path=str(save))
    assert len(losses) == 12 and len(mses) == 12 and np.isfinite(losses).all()
    assert losses[-1] < 0.8 * losses[0]
    assert sorted(os.listdir(save)) == ["model_10.pt", "model_5.pt"]
    state = torch.load(save / "model_10.pt", map_location="cpu")
    assert set(state) == set(model.state_dict())
    # a fresh model resumes from the newest checkpoint (model_10) before training on
    model2 = gnn.GraphModel(input_dim=4, gnn_dim=32, message_passing_steps=2, aggr="mean", neighbors=6, device="cuda")
    tr2 = trainer.Trainer(model2, torch.optim.SGD(model2.parameters(), lr=0.0), device="cuda")
    tr2.train_from_dir(str(path), epochs=1, batch_size=4, save_every=0, save_path=str(save))
    for k, v in model2.state_dict().items():
        assert torch.equal(v.cpu(), state[k]), k
