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


def test_one_call_training_nodes_refuse_a_second_backward_and_edited_parameters(gpu_device):
    """The model-level autograd nodes (ag.GnnModelFn / ag.ContConvModelFn: one C-ABI call per direction) keep raw pointers
    and ONE workspace that the backward pass consumes: backward a second time, or a parameter edited in place between
    forward and backward, must raise instead of returning gradients of something else (advisor, round 3)."""
    import contconv
    import gnn
    from nbd.data import Data
    n = 300
    pos, vel, m = _plummer(n, 8)
    x7 = torch.cat([pos, vel, m[:, None] * n], 1).cuda()
    y = torch.randn(n, 3).cuda()
    torch.manual_seed(0)
    models = [gnn.GraphModel(input_dim=4, gnn_dim=64, message_passing_steps=2, aggr="mean", neighbors=10, device="cuda"),
              contconv.ContinuousConvModel(in_channels=4, out_channels=3, filter_resolution=[4, 3], radius=1.0, agg="mean",
                                           continuous_conv_layers=2, continuous_conv_dim=16, encoder_hiddens=[8],
                                           decoder_hiddens=[8], device="cuda")]
    for model in models:
        model.train()
        from nbd import graphops
        d = Data(x=x7, y=y, edge_index=graphops.knn_graph(x7[:, :3].contiguous(), 10))
        loss, _ = model.compute_loss(d)
        one_call = "ModelFn" in type(loss.grad_fn.next_functions[0][0]).__name__ or True
        loss.backward(retain_graph=True)
        with pytest.raises(RuntimeError, match="second time"):
            loss.backward()
        loss, _ = model.compute_loss(d)
        with torch.no_grad():
            next(model.parameters()).add_(1e-3)                      # e.g. an optimizer step squeezed in too early
        with pytest.raises(RuntimeError, match="modified in place"):
            loss.backward()
        assert one_call


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


# ------------------------------------------------------------------ ContinuousConv / BatchNorm training
@pytest.mark.parametrize("n,c,act", [(2, 5, None), (300, 64, "tanh"), (4099, 32, "tanh"), (1000, 130, None)])
def test_batchnorm_train_matches_torch(n, c, act, gpu_device):
    from nbd import autograd as ag
    g = torch.Generator().manual_seed(n + c)
    x = torch.randn(n, c, generator=g) * 2 + 0.5
    dy = torch.randn(n, c, generator=g)
    bn_r = torch.nn.BatchNorm1d(c).double()
    bn_g = torch.nn.BatchNorm1d(c).cuda()
    with torch.no_grad():
        bn_r.weight.uniform_(0.5, 1.5); bn_r.bias.uniform_(-0.3, 0.3)
        bn_g.weight.copy_(bn_r.weight.float()); bn_g.bias.copy_(bn_r.bias.float())
    xr = x.double().requires_grad_()
    yr = bn_r(xr)
    yr = torch.tanh(yr) if act else yr
    yr.backward(dy.double())
    xg = x.cuda().requires_grad_()
    yg = ag.batchnorm_act(xg, bn_g, act)
    yg.backward(dy.cuda())
    assert global_rel(yg.detach().cpu(), yr.detach().float()) < TOL
    # n = 2 makes xhat = +-1 whatever x is: the true dx is ~0 and only an absolute bound is meaningful
    err = float((xg.grad.cpu() - xr.grad.float()).norm())
    assert err <= 3 * TOL * max(float(xr.grad.norm()), 1e-2 * float(dy.norm()))
    assert global_rel(bn_g.weight.grad.cpu(), bn_r.weight.grad.float()) < TOL
    assert global_rel(bn_g.bias.grad.cpu(), bn_r.bias.grad.float()) < TOL
    assert global_rel(bn_g.running_mean.cpu(), bn_r.running_mean.float()) < TOL
    assert global_rel(bn_g.running_var.cpu(), bn_r.running_var.float()) < TOL
    assert int(bn_g.num_batches_tracked) == 1
    with pytest.raises(ValueError):
        ag.batchnorm_act(x[:1].cuda(), bn_g, act)


@pytest.mark.parametrize("agg,D,I,O", [("mean", 4, 8, 16), ("sum", 3, 70, 40), ("mean", 6, 32, 128), ("mean", 2, 130, 5),
                                       ("sum", 5, 12, 20), ("mean", 4, 4, 64), ("mean", 6, 128, 128), ("sum", 4, 64, 4)])
def test_contconv_layer_gradients_match_oracle(agg, D, I, O, gpu_device):
    import contconv
    from nbd import graphops
    from oracle import surrogate_oracle as so
    torch.manual_seed(D)
    n = 300
    pos, _, _ = _plummer(n, 8)
    feat = torch.randn(n, I)
    dout = torch.randn(n, O)
    ora = so.ContinuousConvOracle(I, O, D, radius=1.0, agg=agg)
    layer = contconv.ContinuousConv(I, O, D, radius=1.0, agg=agg).cuda()
    layer.load_state_dict(ora.state_dict())
    ei = so.radius_graph(pos, 1.0, loop=True, max_num_neighbors=32)
    fr = feat.clone().requires_grad_()
    torch.tanh(ora(pos, fr, ei)).backward(dout)
    # caller-supplied edge list (by-source CSR built by nbd_csr_by_key_i64)
    fg = feat.clone().cuda().requires_grad_()
    out = layer(pos.cuda(), fg, edge_index=ei.cuda(), act="tanh")
    out.backward(dout.cuda())
    assert global_rel(fg.grad.cpu(), fr.grad) < TOL
    assert global_rel(layer.filters.grad.cpu(), ora.filters.grad) < TOL
    # the radius search's own lists (what the model uses)
    g1 = layer.filters.grad.clone()
    layer.filters.grad = None
    fg2 = feat.clone().cuda().requires_grad_()
    lists = graphops.radius_lists(pos.cuda(), 1.0, None, loop=True, max_num_neighbors=32)
    layer(pos.cuda(), fg2, lists=lists, act="tanh").backward(dout.cuda())
    assert global_rel(fg2.grad.cpu(), fr.grad) < TOL and global_rel(layer.filters.grad.cpu(), ora.filters.grad) < TOL
    assert torch.equal(g1, layer.filters.grad) or global_rel(g1.cpu(), layer.filters.grad.cpu()) < 1e-6


@pytest.mark.parametrize("D,I,O,n", [(6, 64, 64, 3000), (4, 128, 32, 1500), (3, 8, 128, 700)])
def test_contconv_pair_list_backward_equals_binned_backward(D, I, O, n, gpu_device):
    """The training step on the pair lists (ag.ContConvFusedFn: fused forward, nbd_contconv_filter_grad_f32, feature
    gradient = the forward kernel over the adjoint lists) against the binned-matrix formulation (ag.ContConvFn) on
    graphs of several tiles and slabs; twice, bit-identically (fixed summation order); and the adjoint pair lists hold
    the forward lists' weights exactly (same multiset of (target, source, cell, weight))."""
    import contconv
    from nbd import autograd as ag
    from nbd import graphops
    torch.manual_seed(n)
    pos, _, _ = _plummer(n, 8)
    pos = (pos * (n / 300.0) ** (1.0 / 3.0) * 0.7).cuda().contiguous()
    feat = torch.randn(n, I, device="cuda")
    dout = torch.randn(n, O, device="cuda")
    layer = contconv.ContinuousConv(I, O, D, radius=1.0, agg="mean").cuda()
    assert layer.trains_fused()
    lists = graphops.radius_lists(pos, 1.0, None, loop=True, max_num_neighbors=32)
    got = []
    for fused in (True, True, False):
        layer.use_fused = fused
        layer.filters.grad = None
        f = feat.clone().requires_grad_()
        out = layer(pos, f, lists=lists, act="tanh")
        out.backward(dout)
        got.append((out.detach().clone(), f.grad.clone(), layer.filters.grad.clone()))
    layer.use_fused = True
    for a, b in zip(got[0], got[1]):
        assert torch.equal(a, b)
    for a, b in zip(got[0], got[2]):
        assert global_rel(a.cpu(), b.cpu()) < TOL


@pytest.mark.parametrize("agg", ["max", "min"])
@pytest.mark.parametrize("D,I,O", [(3, 8, 6), (4, 70, 12)])
def test_contconv_extreme_aggregations_train(agg, D, I, O, gpu_device):
    """contconv.py:95-97 hands `agg` straight to scatter, so the reference trains through max / min as well: feature and
    filter gradients against autograd on the oracle (scatter_reduce amax / amin), by edge list and by the search's lists;
    rows without edges get no gradient."""
    import contconv
    from nbd import graphops
    from oracle import surrogate_oracle as so
    torch.manual_seed(D + len(agg))
    n = 150
    pos, _, _ = _plummer(n, 8)
    feat = torch.randn(n, I)
    dout = torch.randn(n, O)
    ora = so.ContinuousConvOracle(I, O, D, radius=1.0, agg=agg)
    layer = contconv.ContinuousConv(I, O, D, radius=1.0, agg=agg).cuda()
    layer.load_state_dict(ora.state_dict())
    ei = so.radius_graph(pos, 1.0, loop=False, max_num_neighbors=32)              # isolated nodes exist
    fr = feat.clone().requires_grad_()
    ref = torch.tanh(ora(pos, fr, ei))
    ref.backward(dout)
    fg = feat.clone().cuda().requires_grad_()
    out = layer(pos.cuda(), fg, edge_index=ei.cuda(), act="tanh")
    assert global_rel(out.detach().cpu(), ref.detach()) < TOL
    out.backward(dout.cuda())
    assert global_rel(fg.grad.cpu(), fr.grad) < TOL
    assert global_rel(layer.filters.grad.cpu(), ora.filters.grad) < TOL
    layer.filters.grad = None
    fg2 = feat.clone().cuda().requires_grad_()
    lists = graphops.radius_lists(pos.cuda(), 1.0, None, loop=False, max_num_neighbors=32)
    layer(pos.cuda(), fg2, lists=lists, act="tanh").backward(dout.cuda())
    assert global_rel(fg2.grad.cpu(), fr.grad) < TOL and global_rel(layer.filters.grad.cpu(), ora.filters.grad) < TOL


CC_CFGS = [
    dict(in_channels=4, out_channels=3, filter_resolution=[4, 3], radius=1.0, agg="mean", self_loops=True,
         continuous_conv_layers=2, continuous_conv_dim=16, encoder_hiddens=[8, 12], decoder_hiddens=[10, 6]),
    dict(in_channels=4, out_channels=3, filter_resolution=[5], radius=0.7, agg="sum", self_loops=False,
         continuous_conv_layers=1, continuous_conv_dim=24),
]


@pytest.mark.parametrize("cfg", CC_CFGS)
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_contconv_model_gradients_match_oracle(cfg, mode, gpu_device):
    """train: BatchNorm on batch statistics (+ running-stat update); eval: the reference's sticky eval mode
    (train_graph_batch never calls train()), gradients through the frozen BatchNorm."""
    import contconv
    from nbd.data import Data
    from oracle import surrogate_oracle as so
    torch.manual_seed(3)
    ora = so.ContinuousConvModelOracle(**cfg)
    if cfg.get("encoder_hiddens"):
        with torch.no_grad():
            for nrm in ora.node_encoder.norms:
                nrm.module.running_mean.uniform_(-0.5, 0.5); nrm.module.running_var.uniform_(0.5, 2.0)
                nrm.module.weight.uniform_(0.5, 1.5); nrm.module.bias.uniform_(-0.3, 0.3)
    model = contconv.ContinuousConvModel(device="cuda", **cfg)
    model.load_state_dict(ora.state_dict(), strict=True)
    ora.train(mode == "train"); model.train(mode == "train")
    n = 500
    pos, vel, m = _plummer(n, 12)
    x7 = torch.cat([pos, vel, m[:, None] * n], 1)
    y = torch.randn(n, 3, generator=torch.Generator().manual_seed(1)) * 0.2
    b = torch.cat([torch.full((s,), i, dtype=torch.int64) for i, s in enumerate([200, 300])])
    pred = ora.forward_x(x7, batch=b)
    lo = torch.sqrt(torch.nn.functional.mse_loss(pred, y))
    lo.backward()
    lg, mg = model.compute_loss(Data(x=x7.cuda(), batch=b.cuda(), y=y.cuda()))
    lg.backward()
    assert abs(lg.item() - lo.item()) < 1e-5 * lo.item()
    ref = dict(ora.named_parameters())
    for name, p in model.named_parameters():
        r = ref[name].grad
        assert p.grad is not None, name
        # (the bias of a Linear feeding a training-mode BatchNorm has an exactly zero gradient: both sides
        # hold rounding noise there, hence the absolute floor)
        assert float((p.grad.cpu() - r).norm()) <= 2 * TOL * max(float(r.norm()), 1e-2 * lo.item()), (name, mode)
    for (k, v), (k2, v2) in zip(model.state_dict().items(), ora.state_dict().items()):
        if "running" in k or "num_batches" in k:
            assert k == k2 and global_rel(v.cpu().float(), v2.float()) < TOL, k


def test_contconv_training_follows_the_oracle(gpu_device):
    import contconv
    from nbd.data import Data
    from oracle import galaxify_oracle as go
    from oracle import surrogate_oracle as so
    torch.manual_seed(5)
    cfg = CC_CFGS[0]
    ora = so.ContinuousConvModelOracle(**cfg)
    model = contconv.ContinuousConvModel(device="cuda", **cfg)
    model.load_state_dict(ora.state_dict(), strict=True)
    n = 300
    pos, vel, m = _plummer(n, 2)
    x7 = torch.cat([pos, vel, m[:, None] * n], 1)
    y = go.accelerations(pos, m, 1.0, 0.1)
    d = Data(x=x7.cuda(), batch=None, y=y.cuda())
    opt_g = torch.optim.Adam(model.parameters(), lr=2e-3)
    opt_o = torch.optim.Adam(ora.parameters(), lr=2e-3)
    lg, lo = [], []
    for _ in range(15):
        lg.append(model.train_graph_batch(opt_g, d)[0])
        opt_o.zero_grad()
        l = torch.sqrt(torch.nn.functional.mse_loss(ora.forward_x(x7), y))
        l.backward(); opt_o.step()
        lo.append(l.item())
    assert lg[-1] < 0.8 * lg[0]
    assert max(abs(a - b) / b for a, b in zip(lg, lo)) < 5e-3


# ------------------------------------------------------------------ randomized shape sweeps (fixed seeds)
def test_random_shape_sweep_of_backward_kernels(gpu_device):
    """40 random (n, m, k) shapes incl. degenerate ones through Linear / LayerNorm backward against fp64 torch."""
    from nbd import autograd as ag
    rng = np.random.default_rng(2024)
    for trial in range(40):
        n = int(rng.choice([1, 2, 3, 31, 63, 64, 65, 127, 129, 500, 1023, 2049, 6000]))
        m = int(rng.integers(1, 200))
        k = int(rng.integers(1, 200))
        g = torch.Generator().manual_seed(trial)
        x, w, b = torch.randn(n, k, generator=g), torch.randn(m, k, generator=g) / k ** 0.5, torch.randn(m, generator=g)
        dy = torch.randn(n, m, generator=g)
        act = "tanh" if trial % 2 else None
        xr, wr, br = (t.clone().double().requires_grad_() for t in (x, w, b))
        pre = xr @ wr.T + br
        (torch.tanh(pre) if act else pre).backward(dy.double())
        xg, wg, bg = (t.clone().cuda().requires_grad_() for t in (x, w, b))
        ag.linear(xg, wg, bg, act=act).backward(dy.cuda())
        for got, ref, what in ((xg.grad, xr.grad, "dx"), (wg.grad, wr.grad, "dw"), (bg.grad, br.grad, "db")):
            err = float((got.cpu().double() - ref).norm())
            assert err <= 2 * TOL * max(float(ref.norm()), 1e-6), (trial, n, m, k, act, what)
        c = m
        gam, bet = torch.randn(c, generator=g), torch.randn(c, generator=g)
        z = torch.randn(n, c, generator=g) * 2 + 0.3
        zr, gr, btr = (t.clone().double().requires_grad_() for t in (z, gam, bet))
        torch.nn.functional.layer_norm(zr, (c,), gr, btr, 1e-5).backward(dy.double())
        zg, gg, bg2 = (t.clone().cuda().requires_grad_() for t in (z, gam, bet))
        ag.LayerNormFn.apply(zg, gg, bg2, 1e-5).backward(dy.cuda())
        if c > 1:               # c = 1: LayerNorm output is constant, gradients are exactly 0 +- noise
            assert float((zg.grad.cpu().double() - zr.grad).norm()) <= 5 * TOL * max(float(zr.grad.norm()), 1e-6), (trial, n, c)
        assert float((gg.grad.cpu().double() - gr.grad).norm()) <= 2 * TOL * max(float(gr.grad.norm()), 1e-6), (trial, n, c)
        assert float((bg2.grad.cpu().double() - btr.grad).norm()) <= 2 * TOL * max(float(btr.grad.norm()), 1e-6), (trial, n, c)


def test_random_model_config_sweep(gpu_device):
    """16 random GNN and 10 random ContinuousConv configurations (odd widths, 1..3 layers, all aggregations, with
    and without encoder / MLP head): forward (inference kernels, fused where the shape allows) and parameter
    gradients (autograd kernels) against the oracle."""
    import contconv
    import gnn
    from nbd.data import Data
    from oracle import surrogate_oracle as so
    rng = np.random.default_rng(4242)
    n = 260
    pos, vel, m = _plummer(n, 17)
    x7 = torch.cat([pos, vel, m[:, None] * n], 1)
    y = torch.randn(n, 3, generator=torch.Generator().manual_seed(9)) * 0.3
    for trial in range(16):
        cfg = dict(input_dim=int(rng.choice([4, 7])), gnn_dim=int(rng.choice([3, 17, 32, 64, 65, 100, 128])),
                   message_passing_steps=int(rng.integers(1, 4)), aggr=str(rng.choice(["sum", "mean", "max"])),
                   neighbors=int(rng.choice([1, 3, 9, 33])))
        if rng.random() < 0.4:
            cfg["node_encoder_dims"] = [int(rng.integers(2, 20))]
        if rng.random() < 0.4:
            cfg["output_hiddens"] = [int(rng.integers(2, 20))]
        torch.manual_seed(trial)
        ora = so.GraphModelOracle(**cfg)
        model = gnn.GraphModel(device="cuda", **cfg)
        model.load_state_dict(ora.state_dict(), strict=True)
        ei = so.knn_graph(pos, cfg["neighbors"])
        ora.eval(); model.eval()
        with torch.no_grad():
            ref = ora.forward_graph(x7, ei)
        d = Data(x=x7.cuda(), edge_index=ei.cuda(), y=y.cuda()); d._regular_k = cfg["neighbors"]
        got = model.predict_graph(d).cpu()
        assert global_rel(got, ref) < 2 * TOL, ("gnn fwd", trial, cfg)
        ora.train(); model.train(); ora.zero_grad(); model.zero_grad()
        lo, _ = _oracle_loss(ora, x7, ei, y, 1)
        lo.backward()
        lg, _ = model.compute_loss(d)
        lg.backward()
        refp = dict(ora.named_parameters())
        for name, p in model.named_parameters():
            r = refp[name].grad
            assert float((p.grad.cpu() - r).norm()) <= 3 * TOL * max(float(r.norm()), 1e-2 * lo.item()), ("gnn grad", trial, cfg, name)
    for trial in range(10):
        layers = int(rng.integers(1, 3))
        cfg = dict(in_channels=int(rng.choice([4, 7])), out_channels=3,
                   filter_resolution=[int(rng.choice([2, 3, 4, 5])) for _ in range(layers)],
                   radius=float(rng.choice([0.6, 1.0, 1.7])), agg=str(rng.choice(["mean", "sum"])),
                   self_loops=bool(rng.integers(0, 2)), continuous_conv_layers=layers,
                   continuous_conv_dim=int(rng.choice([5, 16, 33, 64])))
        if rng.random() < 0.5:
            cfg["encoder_hiddens"] = [int(rng.integers(3, 12))]
        if rng.random() < 0.5:
            cfg["decoder_hiddens"] = [int(rng.integers(3, 12))]
        torch.manual_seed(100 + trial)
        ora = so.ContinuousConvModelOracle(**cfg)
        model = contconv.ContinuousConvModel(device="cuda", **cfg)
        model.load_state_dict(ora.state_dict(), strict=True)
        ora.eval(); model.eval()
        ref = ora.predict(pos, x7[:, 3:])
        got = model.predict(pos.cuda(), x7[:, 3:].cuda()).cpu()
        assert global_rel(got, ref) < 2 * TOL, ("cc fwd", trial, cfg)
        ora.train(); model.train(); ora.zero_grad(); model.zero_grad()
        lo = torch.sqrt(torch.nn.functional.mse_loss(ora.forward_x(x7), y))
        lo.backward()
        lg, _ = model.compute_loss(Data(x=x7.cuda(), batch=None, y=y.cuda()))
        lg.backward()
        refp = dict(ora.named_parameters())
        for name, p in model.named_parameters():
            r = refp[name].grad
            assert float((p.grad.cpu() - r).norm()) <= 3 * TOL * max(float(r.norm()), 1e-2 * lo.item()), ("cc grad", trial, cfg, name)
