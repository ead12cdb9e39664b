"""CPU checks of the surrogate oracle (oracle/surrogate_oracle.py) and of the host logic of the
drop-in modules. The reference's gnn.py / contconv.py cannot be imported here (torch_geometric,
torch_cluster, torch_scatter are absent), so the oracle is held to hand-computed known answers,
to torch's own operators where the reference calls them (F.grid_sample), and to independent
loop-level restatements of the published PyG semantics. No GPU needed."""
import math

import numpy as np
import pytest
import torch

from oracle import surrogate_oracle as so


def test_knn_graph_known_answer_and_orientation():
    x = torch.tensor([[0., 0, 0], [1, 0, 0], [3, 0, 0], [6, 0, 0], [10, 0, 0]])
    ei = so.knn_graph(x, 2)
    # row 0 = neighbour j, row 1 = centre i; grouped by centre, nearest first
    assert ei.tolist() == [[1, 2, 0, 2, 1, 0, 2, 4, 3, 2], [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]]
    assert so.knn_graph(x, 10).shape == (2, 5 * 4)            # k > n-1 -> n-1 neighbours each
    assert so.knn_graph(x[:1], 3).shape == (2, 0)
    # exact tie: the lower index wins
    y = torch.tensor([[0., 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0]])
    assert so.knn_graph(y, 1)[0].tolist()[0] == 1
    assert so.knn_graph(y, 1, loop=True)[0].tolist() == [0, 1, 2, 3]
    # batch segments do not see each other
    b = torch.tensor([0, 0, 1, 1, 1])
    eb = so.knn_graph(x, 2, batch=b)
    assert eb.tolist() == [[1, 0, 3, 4, 2, 4, 3, 2], [0, 1, 2, 2, 3, 3, 4, 4]]


def test_radius_graph_known_answer_cap_and_strictness():
    x = torch.tensor([[0., 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0]])
    assert so.radius_graph(x, 1.0).shape == (2, 0)                       # d2 < r2 is strict
    e = so.radius_graph(x, 1.5, loop=False)
    assert e.tolist() == [[1, 0, 2, 1, 3, 2], [0, 1, 1, 2, 2, 3]]
    e = so.radius_graph(x, 1.5, loop=True)
    assert e.tolist() == [[0, 1, 0, 1, 2, 1, 2, 3, 2, 3], [0, 0, 1, 1, 1, 2, 2, 2, 3, 3]]
    e = so.radius_graph(x, 10.0, loop=True, max_num_neighbors=2)          # first 2 by index
    assert e.tolist() == [[0, 1, 0, 1, 0, 1, 0, 1], [0, 0, 1, 1, 2, 2, 3, 3]]
    # torch_cluster 1.6.3: without self loops the search runs with cap + 1 slots and self as a candidate, THEN drops
    # row == col: centres 0-2 find themselves among their first 3 hits and keep 2 others; centre 3 has three
    # lower-indexed hits, never sees itself, and keeps all 3
    e = so.radius_graph(x, 10.0, loop=False, max_num_neighbors=2)
    assert e.tolist() == [[1, 2, 0, 2, 0, 1, 0, 1, 2], [0, 0, 1, 1, 2, 2, 3, 3, 3]]
    # r2 = float(double(r) * double(r)): for r = 0.7 that is 0.49000001, one ulp above fp32(0.7)^2 = 0.48999998 -- a
    # body at distance exactly fp32(0.7) (d2 = 0.48999998) is inside
    assert np.float32(0.7) * np.float32(0.7) < np.float32(0.7 * 0.7) == np.float32(so.radius_r2(0.7))
    y = torch.tensor([[0., 0, 0], [float(np.float32(0.7)), 0, 0]])
    assert so.radius_graph(y, 0.7, loop=False).tolist() == [[1, 0], [0, 1]]


def test_knn_graph_is_search_k_plus_one_then_drop_self():
    # three coincident bodies then two others; k = 1: bodies 0 and 1 find themselves among their 2 nearest (ties ->
    # lower index: 0, 1) and keep one neighbour; body 2 has two lower-indexed bodies at distance 0, its 2 nearest
    # are 0 and 1, row == col drops nothing: TWO neighbours (masking the diagonal would give one)
    x = torch.tensor([[1., 1, 1], [1, 1, 1], [1, 1, 1], [2, 1, 1], [5, 1, 1]])
    ei = so.knn_graph(x, 1)
    assert ei.tolist() == [[1, 0, 0, 1, 0, 3], [0, 1, 2, 2, 3, 4]]
    # with k = 2 every body sees itself within its 3 nearest: two neighbours each, as with a masked diagonal
    assert so.knn_graph(x, 2).tolist() == [[1, 2, 0, 2, 0, 1, 0, 1, 3, 0], [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]]


def test_edgeconv_matches_loop_restatement():
    torch.manual_seed(0)
    x = torch.randn(6, 3)
    ei = torch.tensor([[1, 2, 0, 3, 5, 5, 4], [0, 0, 1, 1, 1, 3, 3]])         # nodes 2,4,5 receive nothing
    nn_ = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 5))
    for aggr in ("sum", "mean", "max"):
        got = so.EdgeConv(nn_, aggr)(x, ei)
        ref = torch.zeros(6, 5)
        for i in range(6):
            msgs = [nn_(torch.cat([x[i], x[j] - x[i]])) for j, t in zip(ei[0].tolist(), ei[1].tolist()) if t == i]
            if msgs:
                m = torch.stack(msgs)
                ref[i] = m.sum(0) if aggr == "sum" else (m.mean(0) if aggr == "mean" else m.max(0).values)
        assert torch.allclose(got, ref, atol=1e-6)


def test_edgeconv_factoring_identity():
    """The algebra the HIP path relies on (gnn.py docstring): W1[x_i || x_j - x_i] + b1 = P_i + Q_j and
    mean_j(W2 h + b2) = W2 mean_j(h) + b2."""
    torch.manual_seed(1)
    n, f, h = 40, 4, 16
    x = torch.randn(n, f, dtype=torch.float64)
    ei = so.knn_graph(x[:, :3].float(), 5)
    lin1, lin2 = torch.nn.Linear(2 * f, h).double(), torch.nn.Linear(h, h).double()
    conv = so.EdgeConv(torch.nn.Sequential(lin1, torch.nn.Tanh(), lin2), "mean")
    ref = conv(x, ei)
    w1a, w1b = lin1.weight[:, :f], lin1.weight[:, f:]
    p = x @ (w1a - w1b).t() + lin1.bias
    q = x @ w1b.t()
    s = so.scatter(torch.tanh(p[ei[1]] + q[ei[0]]), ei[1], n, "mean")
    assert torch.allclose(s @ lin2.weight.t() + lin2.bias, ref, atol=1e-12)


def test_trilinear_axis_order_against_grid_sample():
    """contconv.py:62-75: coordinate component 0 indexes filter axis 2, component 2 axis 0
    (filters[z, y, x]); align_corners=True; out-of-range corner -> 0. Checked against F.grid_sample,
    the reference's own call, through the oracle."""
    torch.manual_seed(2)
    D, I, O = 5, 3, 2
    layer = so.ContinuousConvOracle(I, O, D, radius=1.0)
    coords = torch.rand(200, 3) * (D - 1)
    coords[0] = torch.tensor([D - 1.0, 0.0, 2.0])                       # exactly on the upper face
    got = layer.trilinear_interpolate(coords)
    f = layer.filters.detach()
    ref = torch.zeros(200, I, O)
    for e in range(200):
        gx, gy, gz = coords[e].tolist()
        ix, iy, iz = math.floor(gx), math.floor(gy), math.floor(gz)
        tx, ty, tz = gx - ix, gy - iy, gz - iz
        for az in (0, 1):
            for ay in (0, 1):
                for ax in (0, 1):
                    cx, cy, cz = ix + ax, iy + ay, iz + az
                    if cx >= D or cy >= D or cz >= D:
                        continue
                    w = (tx if ax else 1 - tx) * (ty if ay else 1 - ty) * (tz if az else 1 - tz)
                    ref[e] += w * f[cz, cy, cx]
    assert torch.allclose(got, ref, atol=2e-6)


def test_contconv_binning_identity():
    """The restructuring the HIP path uses: sum_e window_e * filt_e . feat = A . filters with
    A[n][cell] = sum_e window_e t_cell(e) feat[c_e] (dense algebra check in float64 on CPU)."""
    torch.manual_seed(3)
    n, D, I, O = 60, 4, 5, 3
    pos = torch.randn(n, 3) * 0.7
    feat = torch.randn(n, I)
    ei = so.radius_graph(pos, 1.0, loop=True, max_num_neighbors=8)
    layer = so.ContinuousConvOracle(I, O, D, radius=1.0, agg="mean")
    ref = layer(pos, feat, ei).detach()
    A = torch.zeros(n, D ** 3, I, dtype=torch.float64)
    row, col = ei
    for e in range(ei.shape[1]):
        r = (pos[col[e]] - pos[row[e]]).double()
        d2 = float((r ** 2).sum())
        if not d2 < 1.0:
            continue
        window = (1 - d2) ** 3
        nrm = math.sqrt(d2)
        g = (r / (nrm + 1e-8) * math.tanh(nrm) + 1) * ((D - 1) / 2)
        ix, iy, iz = (int(math.floor(v)) for v in g.tolist())
        tx, ty, tz = g[0] - ix, g[1] - iy, g[2] - iz
        for az in (0, 1):
            for ay in (0, 1):
                for ax in (0, 1):
                    cx, cy, cz = ix + ax, iy + ay, iz + az
                    if max(cx, cy, cz) >= D:
                        continue
                    w = (tx if ax else 1 - tx) * (ty if ay else 1 - ty) * (tz if az else 1 - tz)
                    A[row[e], (cz * D + cy) * D + cx] += window * w * feat[col[e]].double()
    out = A.reshape(n, -1) @ layer.filters.detach().double().reshape(D ** 3 * I, O)
    deg = torch.bincount(row, minlength=n).clamp(min=1).double()
    assert torch.allclose(out / deg[:, None], ref.double(), atol=2e-6)


def test_batchnorm_folding_identity():
    import gnn
    torch.manual_seed(4)
    mlp = gnn.MLP([4, 8, 6, 5])
    with torch.no_grad():
        for nrm in mlp.norms:
            nrm.module.running_mean.uniform_(-1, 1); nrm.module.running_var.uniform_(0.3, 3)
            nrm.module.weight.uniform_(0.5, 2); nrm.module.bias.uniform_(-1, 1)
    ora = so.PygMLP([4, 8, 6, 5]).eval()
    ora.load_state_dict(mlp.state_dict())
    x = torch.randn(30, 4)
    ref = ora(x)
    y = x
    for w, b, act in mlp.folded():
        y = y @ w.t() + b
        y = torch.tanh(y) if act == "tanh" else y
    assert torch.allclose(y, ref, atol=1e-5)


def test_state_dicts_are_interchangeable_with_reference_layout():
    import contconv
    import gnn
    g = gnn.GraphModel(input_dim=4, gnn_dim=64, message_passing_steps=2, aggr="mean", neighbors=10)
    assert list(g.state_dict()) == ["gnns.0.nn.0.weight", "gnns.0.nn.0.bias", "gnns.0.nn.2.weight", "gnns.0.nn.2.bias",
                                    "gnns.1.nn.0.weight", "gnns.1.nn.0.bias", "gnns.1.nn.2.weight", "gnns.1.nn.2.bias",
                                    "layer_norm.weight", "layer_norm.bias", "output.weight", "output.bias"]
    assert g.gnns[0].nn[0].weight.shape == (64, 8) and g.layer_norm.normalized_shape == (68,)    # gnn_experiment.py:61-72
    g.load_state_dict(so.GraphModelOracle(input_dim=4, gnn_dim=64, message_passing_steps=2, aggr="mean").state_dict())
    c = contconv.ContinuousConvModel(in_channels=4, out_channels=3, filter_resolution=[6, 4], radius=1.0,
                                     continuous_conv_layers=2, continuous_conv_dim=128, encoder_hiddens=[32, 64],
                                     decoder_hiddens=[64, 32], device="cpu")
    keys = list(c.state_dict())
    assert "contconv.0.filters" in keys and "node_encoder.norms.1.module.running_var" in keys and "output.4.bias" in keys
    assert c.contconv[0].filters.shape == (6, 6, 6, 128, 128) and c.contconv[1].filters.shape == (4, 4, 4, 128, 128)
    assert c.layer_norm.normalized_shape == (256,) and c.neighbors == 0
    with pytest.raises(NotImplementedError):
        gnn.GraphModel(aggr="min")
    # training runs on the HIP kernels too: CPU tensors are refused, never silently computed on the host
    from nbd._lib import NbdError
    from nbd.data import Data
    with pytest.raises(NbdError):
        g.compute_loss(Data(x=torch.zeros(5, 7), edge_index=torch.zeros((2, 0), dtype=torch.int64), y=torch.zeros(5, 3)))


def test_forward_on_cpu_tensors_fails_loudly():
    import gnn
    from nbd._lib import NbdError
    from nbd.data import Data
    g = gnn.GraphModel(input_dim=4, gnn_dim=8, message_passing_steps=1, aggr="mean")
    with pytest.raises(NbdError):
        g.predict_graph(Data(x=torch.zeros(5, 7), edge_index=torch.zeros((2, 0), dtype=torch.int64)))


def test_collate_offsets_edges_and_builds_batch():
    from nbd.data import Data, collate
    a = Data(x=torch.zeros(3, 7), y=torch.zeros(3, 3), edge_index=torch.tensor([[0, 1], [1, 2]]),
             step=torch.zeros(3, dtype=torch.int64))
    b = Data(x=torch.ones(2, 7), y=torch.ones(2, 3), edge_index=torch.tensor([[1], [0]]), step=torch.ones(2, dtype=torch.int64))
    c = collate([a, b])
    assert c.x.shape == (5, 7) and c.batch.tolist() == [0, 0, 0, 1, 1] and c.step.tolist() == [0, 0, 0, 1, 1]
    assert c.edge_index.tolist() == [[0, 1, 4], [1, 2, 3]]


def test_trainer_step_restatement_is_leapfrog():
    pos, vel = torch.randn(7, 3), torch.randn(7, 3)
    m, acc = torch.rand(7, 1), torch.randn(7, 3)
    p2, v2, a2 = so.trainer_step(lambda p, f: -p, pos, vel, m, acc, 0.1)
    v_half = vel + 0.05 * acc
    assert torch.allclose(p2, pos + 0.1 * v_half) and torch.allclose(a2, -p2) and torch.allclose(v2, v_half + 0.05 * a2)


def test_contconv_oracle_product_aggregation_known_answer():
    """scatter(reduce="mul") in the oracle: a row's output is the product of its messages, 1 without any -- checked
    against the sum-aggregation oracle's single-edge messages multiplied by hand."""
    import torch
    from oracle import surrogate_oracle as so
    torch.manual_seed(3)
    pos = torch.tensor([[0.0, 0, 0], [0.3, 0, 0], [0, 0.4, 0], [5.0, 5, 5]])
    feat = torch.randn(4, 3)
    mul = so.ContinuousConvOracle(3, 2, 3, radius=1.0, agg="mul")
    add = so.ContinuousConvOracle(3, 2, 3, radius=1.0, agg="sum")
    add.load_state_dict(mul.state_dict())
    ei = torch.tensor([[0, 0, 1], [1, 2, 0]])
    with torch.no_grad():
        out = mul(pos, feat, ei)
        m01 = add(pos, feat, ei[:, 0:1])[0]
        m02 = add(pos, feat, ei[:, 1:2])[0]
        m10 = add(pos, feat, ei[:, 2:3])[1]
    assert torch.allclose(out[0], m01 * m02, rtol=1e-6, atol=0) and torch.allclose(out[1], m10, rtol=1e-6, atol=0)
    assert torch.equal(out[2], torch.ones(2)) and torch.equal(out[3], torch.ones(2))
