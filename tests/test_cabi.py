"""The C-ABI shared library loads on a machine WITHOUT a GPU and exports every symbol that
include/*.h declares; the ctypes signature table covers the same set. No compute calls here."""
import ctypes
import glob
import os
import re
import sys

import pytest

from conftest import ROOT
from nbd import _lib


def declared_symbols():
    names = []
    for hdr in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(hdr).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(nbd_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_something():
    assert len(declared_symbols()) >= 10


def test_library_builds_and_loads():
    _lib.build()
    assert os.path.exists(_lib.LIB_PATH)
    assert _lib.lib().nbd_abi_version() == _lib.ABI_VERSION == 2


@pytest.mark.parametrize("sym", declared_symbols())
def test_symbol_exported(sym):
    handle = ctypes.CDLL(_lib.LIB_PATH)
    assert hasattr(handle, sym), f"{sym} declared in include/ but not exported"


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_ctypes_struct_mirrors_have_the_c_sizes():
    """Every argument struct of include/nbd.h has a ctypes mirror (nbd/_lib.py) of exactly the compiled size -- a field
    added on one side only would shift every later field silently."""
    L = _lib.lib()
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "nbd.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"typedef struct (nbd_[a-z0-9_]+)", text)))
    assert declared == sorted(_lib.STRUCT_MIRRORS)
    for name, mirror in _lib.STRUCT_MIRRORS.items():
        assert L.nbd_struct_size(name.encode()) == ctypes.sizeof(mirror) > 0, name
    assert L.nbd_struct_size(b"no_such_struct") == 0 and L.nbd_struct_size(None) == 0


def test_host_only_queries():
    L = _lib.lib()
    assert L.nbd_posm_padded_len(0) == 0
    assert L.nbd_posm_padded_len(1) == 64
    assert L.nbd_posm_padded_len(64) == 64
    assert L.nbd_posm_padded_len(65) == 128
    assert L.nbd_strerror(0) == b"ok"
    assert b"workspace" in L.nbd_strerror(-2)
    g, s, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert L.nbd_accel_plan(65536, 65536, g, s, c) == 0
    # 512 target groups x 16 slabs = 8192 workgroups (6.4 residency rounds of 5 per CU), 16 chunks per wave
    assert (g.value, s.value, c.value) == (512, 16, 16)
    assert L.nbd_accel_plan(0, 5, None, None, None) == -1
    assert L.nbd_step_workspace_bytes(65536) == 16 * 65536 * 12
    for n_src, n_tgt in [(3, 3), (1000, 1000), (65536, 8192), (524288, 65536), (100, 7), (65536, 16384), (8192, 8192)]:
        assert L.nbd_accel_plan(n_src, n_tgt, g, s, c) == 0
        chunks = (n_src + 63) // 64
        assert s.value * 4 * c.value >= chunks          # every chunk is covered
        assert 1 <= s.value <= 64 and (c.value - 1) * s.value * 4 < chunks      # ... and no wave is given more than its share
        assert g.value * 128 >= n_tgt
    # the range-sharded split: one of 8 ranks at N = 65 536 (own block + remote block), ragged shards, a lone rank
    a, b, cc, d = (ctypes.c_int() for _ in range(4))
    assert L.nbd_shard_plan(65536, 3 * 8192, 8192, a, b, cc, d) == 0
    assert a.value * 4 * b.value >= 128 and cc.value * 4 * d.value >= 896 and 16 <= cc.value <= 64
    assert L.nbd_shard_workspace_bytes(65536, 3 * 8192, 8192) == (a.value + cc.value) * 8192 * 12
    assert L.nbd_shard_plan(1000, 0, 1000, a, b, cc, d) == 0 and cc.value == 0          # nothing remote
    assert L.nbd_shard_plan(1000, 990, 20, a, b, cc, d) == -1                           # range outside the system
    assert L.nbd_contconv_fused_supported(128, 128, 160) == 1 and L.nbd_contconv_fused_supported(70, 40, 27) == 0
    assert L.nbd_contconv_fused_supported(128, 128, 216) == 0                            # pair-list LDS tables: <= 160 cells
    # round 3: the training-side queries
    assert L.nbd_contconv_filter_grad_workspace_bytes(2034, 160, 128, 128) == 4096 + (512 + 160) * 128 * 128 * 4
    assert L.nbd_contconv_filter_grad_workspace_bytes(0, 160, 128, 128) == 0
    dims = (ctypes.c_int * 4)(256, 64, 32, 3)
    assert L.nbd_ln_mlp_head_lds_bytes(256, 3, dims) > 0                                 # the published ContinuousConv decoder
    assert L.nbd_ln_mlp_head_lds_bytes(300, 3, dims) == 0                                # dims[0] must be c, c <= 256
    wide = (ctypes.c_int * 3)(64, 100, 3)
    assert L.nbd_ln_mlp_head_lds_bytes(64, 2, wide) == 0                                 # hidden width <= 64
    ga = _lib.GnnTrainArgs()
    assert L.nbd_gnn_train_workspace_bytes(ctypes.byref(ga)) == 0                        # an all-zero configuration is rejected
    ga.n, ga.f, ga.h, ga.n_layers, ga.n_head, ga.fixed_k = 100, 4, 64, 2, 1, 10
    ga.head_dim[0], ga.head_dim[1] = 68, 3
    assert L.nbd_gnn_train_workspace_bytes(ctypes.byref(ga)) > 100 * (68 + 2 * 128) * 4
    ca = _lib.CcTrainArgs()
    assert L.nbd_cc_train_workspace_bytes(ctypes.byref(ca)) == 0
    ca.n, ca.in_ch, ca.cdim, ca.n_layers, ca.n_head = 500, 4, 64, 1, 1
    ca.n_cells[0], ca.cells_total[0] = 64, 64
    ca.head_dim[0], ca.head_dim[1] = 68, 3
    assert L.nbd_cc_train_workspace_bytes(ctypes.byref(ca)) > 0
    ca.cdim = 66                                                                        # out_channels % 4: the adjoint direction
    ca.head_dim[0] = 70
    assert L.nbd_cc_train_workspace_bytes(ctypes.byref(ca)) == 0
    # the exponential tables of a one-call GNN forward: one [n][2h] block per layer, only for h <= 64 with the first
    # layer formed from x and every layer feeding the next one's [P|Q]
    fa = _lib.GnnForwardArgs()
    assert L.nbd_gnn_forward_workspace_bytes(None) == 0 and L.nbd_gnn_forward_workspace_bytes(ctypes.byref(fa)) == 0
    fa.n, fa.k, fa.n_layers = 4096, 50, 2
    for l in range(2):
        fa.layers[l].h = 64
    fa.layers[0].x, fa.layers[0].f, fa.layers[0].epilogue, fa.layers[0].ep_out = 0x1000, 4, 4, 128
    fa.layers[0].out, fa.layers[0].ldout = 0x2000, 128
    fa.layers[1].pq, fa.layers[1].ldpq = 0x2000, 128
    assert L.nbd_gnn_forward_workspace_bytes(ctypes.byref(fa)) == 2 * 4096 * 128 * 4
    fa.layers[1].pq = 0x3000                                                            # not the first layer's output
    assert L.nbd_gnn_forward_workspace_bytes(ctypes.byref(fa)) == 0
    fa.layers[1].pq, fa.n = 0x2000, 10000                                               # beyond the staged search
    assert L.nbd_gnn_forward_workspace_bytes(ctypes.byref(fa)) == 0


def test_bad_arguments_are_rejected_without_touching_the_gpu():
    L = _lib.lib()
    assert L.nbd_pack_posm_f32(None, None, -1, None, None) == -1
    assert L.nbd_pack_posm_f32(None, None, 5, None, None) == -1
    assert L.nbd_accel_f32(None, -1, None, 1, 0, 0.01, 1.0, None, None, 0, None) == -1
    assert L.nbd_accel_f32(None, 4, None, 4, 0, 0.01, 1.0, None, None, 0, None) == -1
    assert L.nbd_kick_f32(None, None, 3, 0.1, None) == -1
    assert L.nbd_leapfrog_step_f32(None, None, None, None, None, 8, 0.1, 0.1, 0.01, 1.0, None, None, 0, None) == -1
    assert L.nbd_energy_f32(None, None, 4, 0.1, 1.0, None, None, 0, None) == -1
    assert L.nbd_contconv_pairs_jobs_f32(None, 5, 1.0, 0, None, None) == -1
    assert L.nbd_contconv_filter_grad_f32(None, 128, 128, None, 128, 128, None, 5, 100, None, 160, None, None, 0, None) == -1
    assert L.nbd_contconv_filter_grad_f32(None, 127, 127, None, 128, 128, None, 5, 100, None, 160, None, None, 0, None) == -1
    assert L.nbd_contconv_shuffle_filters_f32(None, None, 4, 8, 8, 0, None, None) == -1
    assert L.nbd_linear_wgrad_bias_f32(None, 4, None, 4, None, 10, 4, 4, None, 4, None, None, 0, None) == -1
    assert L.nbd_rowptr_sorted_i64(None, 5, 3, None, None) == -1
    assert L.nbd_ball_to_cube_f32(None, 5, None, None) == -1 and L.nbd_ball_to_cube_f32(None, 0, None, None) == 0
    assert L.nbd_trilinear_interpolate_f32(None, 4, 3, 5, None, 7, None, None) == -1          # NULL buffers
    assert L.nbd_trilinear_interpolate_f32(None, 1, 3, 5, None, 0, None, None) == -1          # a grid needs two points per axis
    assert L.nbd_trilinear_interpolate_f32(None, 4, 3, 5, None, 0, None, None) == 0
    assert L.nbd_gnn_train_forward_f32(None, None) == -1 and L.nbd_cc_train_forward_f32(None, None) == -1
    assert L.nbd_knn_graph_hint_pq_f32(None, 5, 3, 0, 15, None, None, None, None) == -1
    pq = _lib.KnnPqArgs()
    assert L.nbd_knn_graph_hint_pq_f32(None, 5, 3, 0, 15, None, None, ctypes.byref(pq), None) == -1      # no tables to write
    dims = (ctypes.c_int * 2)(300, 3)
    assert L.nbd_ln_mlp_head_f32(None, 300, 300, None, None, 1e-5, 1, None, None, dims, None, 3, None, 0.0, 4, None) == -1
    # n == 0 is a no-op, not an error
    assert L.nbd_pack_posm_f32(None, None, 0, None, None) == 0
    assert L.nbd_kick_f32(None, None, 0, 0.1, None) == 0


def test_no_cpu_path():
    from galaxify import simulation
    import numpy as np
    import torch
    z = np.zeros((4, 3))
    with pytest.raises(ValueError):
        simulation.LeapFrogSimulator(positions=z, velocities=z, masses=np.ones(4), device="tpu")
    with pytest.raises(RuntimeError):
        simulation.LeapFrogSimulator(positions=z, velocities=z, masses=np.ones(4), device="cpu")
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            simulation.LeapFrogSimulator(positions=z, velocities=z, masses=np.ones(4))


def test_fused_contconv_prefetch_survives_the_compiler():
    """The fused ContinuousConv kernel keeps the next cell's filter fragment in flight behind a hand-counted
    `s_waitcnt vmcnt(8)` issued from inline asm, which hipcc's own wait insertion does not track: nothing may touch
    the fragment registers between a load and its wait, and the kernel must stay within 128 VGPRs / its two known
    spills (tools/check_contconv_isa.py disassembles the gfx950 code and checks exactly that; no GPU needed)."""
    import shutil
    if not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_contconv_isa
    ok, report = check_contconv_isa.main()
    assert ok, report
