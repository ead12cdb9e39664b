"""ctypes loader for libnbd_hip.so (C-ABI: include/nbd.h).

There is NO CPU fallback: if the shared library is missing or a call fails, the product path
raises. `build()` compiles it in-tree with hipcc (gfx950); the built .so travels with the
repo snapshot to the GPU box.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p, POINTER

# torch must be imported BEFORE libnbd_hip.so is dlopen'ed: torch ships its own libamdhip64 /
# libhsa-runtime64 under torch/lib with the same sonames as /opt/rocm/lib. Whichever is loaded
# first satisfies both; if ours pulled in /opt/rocm's copy first, torch would bring up a second
# HSA runtime in the process and every HIP call from here fails with hipErrorNoDevice (100).
import torch  # noqa: F401  (load order matters, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
# NBD_LIB_OVERRIDE: a probe / ablation build of the SAME library (tools/build_contconv_trace.sh -> tools/_trace/*.so),
# for the measurement tools only; there is still no other path than a HIP build of csrc/.
LIB_PATH = os.environ.get("NBD_LIB_OVERRIDE") or os.path.join(CSRC_DIR, "libnbd_hip.so")


class NbdError(RuntimeError):
    pass


class NbdUnsupported(NbdError):
    """A configuration a fast path does not cover (NBD_E_UNSUPPORTED, or the Python-side check in front of it): callers that
    probe a path catch THIS and fall back; any other NbdError is a real failure and propagates."""


def build(verbose: bool = False) -> str:
    """Compile every .hip under csrc/ into libnbd_hip.so (make is incremental)."""
    res = subprocess.run(["make", "-C", CSRC_DIR, "-j4"], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise NbdError("building libnbd_hip.so failed (hipcc/make):\n" + (res.stderr or res.stdout)[-4000:])
    return LIB_PATH


class GnnLayerArgs(ctypes.Structure):
    """Mirror of `nbd_gnn_layer_args` (include/nbd.h), field for field."""
    _fields_ = [("rowptr", c_void_p), ("src", c_void_p), ("fixed_k", c_int), ("n", c_int),
                ("pq", c_void_p), ("ldpq", c_int),
                ("x", c_void_p), ("ldx", c_int), ("f", c_int), ("wpq", c_void_p), ("bpq", c_void_p),
                ("h", c_int), ("aggr", c_int), ("w2t", c_void_p), ("b2", c_void_p),
                ("epilogue", c_int), ("w_ep", c_void_p), ("b_ep", c_void_p), ("ep_out", c_int),
                ("enc", c_void_p), ("ldenc", c_int), ("e", c_int), ("ln_g", c_void_p), ("ln_b", c_void_p),
                ("ln_eps", c_float), ("out", c_void_p), ("ldout", c_int), ("kick_vel", c_void_p), ("kick_c", c_float),
                ("epq", c_void_p), ("ldepq", c_int), ("out_epq", c_void_p), ("ldout_epq", c_int),
                ("adv_vel_half", c_void_p), ("adv_pos", c_void_p), ("adv_posm", c_void_p), ("adv_pos_out", c_void_p),
                ("adv_dt", c_float)]


GNN_MAX_LAYERS = 8


class GnnForwardArgs(ctypes.Structure):
    """Mirror of `nbd_gnn_forward_args` (include/nbd.h), field for field."""
    _fields_ = [("pos", c_void_p), ("n", c_int), ("k", c_int), ("loop", c_int), ("use_hint", c_int),
                ("edge_index", c_void_p), ("n_layers", c_int), ("layers", GnnLayerArgs * GNN_MAX_LAYERS),
                ("workspace", c_void_p), ("workspace_bytes", c_size_t)]


class KnnPqArgs(ctypes.Structure):
    """Mirror of `nbd_knn_pq_args` (include/nbd.h), field for field."""
    _fields_ = [("x", c_void_p), ("ldx", c_int), ("f", c_int), ("h", c_int), ("wpq", c_void_p), ("bpq", c_void_p),
                ("epq", c_void_p), ("ldepq", c_int)]


TRAIN_MAX_MLP = 8
_PF = c_void_p * TRAIN_MAX_MLP
_PL = c_void_p * GNN_MAX_LAYERS


class GnnTrainArgs(ctypes.Structure):
    """Mirror of `nbd_gnn_train_args` (include/nbd.h), field for field."""
    _fields_ = [("n", c_int), ("rowptr", c_void_p), ("src", c_void_p), ("fixed_k", c_int), ("rowptr_t", c_void_p),
                ("tgt_t", c_void_p), ("aggr", c_int),
                ("x", c_void_p), ("ldx", c_int), ("f", c_int),
                ("n_enc", c_int), ("enc_w", _PF), ("enc_b", _PF), ("enc_dim", c_int * (TRAIN_MAX_MLP + 1)),
                ("n_layers", c_int), ("h", c_int), ("w1", _PL), ("b1", _PL), ("w2", _PL), ("b2", _PL),
                ("ln_g", c_void_p), ("ln_b", c_void_p), ("ln_eps", c_float),
                ("n_head", c_int), ("head_w", _PF), ("head_b", _PF), ("head_dim", c_int * (TRAIN_MAX_MLP + 1)),
                ("out", c_void_p), ("ldout", c_int), ("workspace", c_void_p), ("workspace_bytes", c_size_t)]


class GnnTrainGrads(ctypes.Structure):
    """Mirror of `nbd_gnn_train_grads` (include/nbd.h), field for field."""
    _fields_ = [("enc_w", _PF), ("enc_b", _PF), ("w1", _PL), ("b1", _PL), ("w2", _PL), ("b2", _PL),
                ("ln_g", c_void_p), ("ln_b", c_void_p), ("head_w", _PF), ("head_b", _PF)]


class CcTrainArgs(ctypes.Structure):
    """Mirror of `nbd_cc_train_args` (include/nbd.h), field for field."""
    _fields_ = [("n", c_int), ("x", c_void_p), ("ldx", c_int), ("in_ch", c_int),
                ("n_enc", c_int), ("enc_w", _PF), ("enc_b", _PF), ("enc_dim", c_int * (TRAIN_MAX_MLP + 1)),
                ("enc_bn", c_int), ("bn_g", _PF), ("bn_b", _PF), ("bn_eps", c_float * TRAIN_MAX_MLP),
                ("bn_rmean", _PF), ("bn_rvar", _PF), ("bn_momentum", c_float * TRAIN_MAX_MLP),
                ("n_layers", c_int), ("cdim", c_int),
                ("filt", _PL), ("kept", _PL), ("cell_map", _PL), ("n_cells", c_int * GNN_MAX_LAYERS),
                ("cells_total", c_int * GNN_MAX_LAYERS), ("pairs_fwd", _PL), ("pairs_adj", _PL),
                ("rowptr_fwd", c_void_p), ("cap_fwd", c_int64), ("rowptr_adj", c_void_p), ("cap_adj", c_int64),
                ("scale", c_void_p), ("ln_g", c_void_p), ("ln_b", c_void_p), ("ln_eps", c_float),
                ("n_head", c_int), ("head_w", _PF), ("head_b", _PF), ("head_dim", c_int * (TRAIN_MAX_MLP + 1)),
                ("out", c_void_p), ("ldout", c_int), ("workspace", c_void_p), ("workspace_bytes", c_size_t)]


class CcTrainGrads(ctypes.Structure):
    """Mirror of `nbd_cc_train_grads` (include/nbd.h), field for field."""
    _fields_ = [("enc_w", _PF), ("enc_b", _PF), ("bn_g", _PF), ("bn_b", _PF), ("filt", _PL),
                ("ln_g", c_void_p), ("ln_b", c_void_p), ("head_w", _PF), ("head_b", _PF)]


class CcPairsJob(ctypes.Structure):
    """Mirror of `nbd_cc_pairs_job` (include/nbd.h), field for field."""
    _fields_ = [("rowptr", c_void_p), ("centres", c_void_p), ("deg", c_void_p), ("edge_capacity", c_int64),
                ("filter_resolution", c_int), ("cell_map", c_void_p), ("n_cells", c_int), ("adjoint", c_int),
                ("pair_lists", c_void_p), ("pair_lists_bytes", c_size_t)]


CC_MAX_RES = 4
ABI_VERSION = 2       # NBD_ABI_VERSION of include/nbd.h (2: the fused ContinuousConv filter operand is bf16 x 3)

# name -> (restype, argtypes); mirrors include/nbd.h one to one (tests check the two agree)
_F = POINTER(c_float)
SIGNATURES = {
    "nbd_abi_version": (c_int, []),
    "nbd_strerror": (c_char_p, [c_int]),
    "nbd_posm_padded_len": (c_int, [c_int]),
    "nbd_pack_posm_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "nbd_accel_workspace_bytes": (c_size_t, [c_int, c_int]),
    "nbd_accel_plan": (c_int, [c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "nbd_accel_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_float, c_void_p,
                              c_void_p, c_size_t, c_void_p]),
    "nbd_accel_tuned_workspace_bytes": (c_size_t, [c_int, c_int]),
    "nbd_accel_tuned_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_float, c_float,
                                    c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "nbd_shard_plan": (c_int, [c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int),
                               POINTER(c_int)]),
    "nbd_shard_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "nbd_shard_force_local_f32": (c_int, [c_void_p, c_int, c_float, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "nbd_shard_force_remote_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_float, c_void_p,
                                           c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    "nbd_shard_force_local_uniform_f32": (c_int, [c_void_p, c_int, c_float, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "nbd_shard_force_remote_uniform_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_float, c_float,
                                                   c_void_p, c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    "nbd_kick_drift_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float,
                                   c_void_p, c_void_p]),
    "nbd_kick_f32": (c_int, [c_void_p, c_void_p, c_int, c_float, c_void_p]),
    "nbd_drift_f32": (c_int, [c_void_p, c_void_p, c_int, c_float, c_void_p]),
    "nbd_snapshot_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "nbd_step_workspace_bytes": (c_size_t, [c_int]),
    "nbd_leapfrog_step_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                      c_float, c_float, c_float, c_float, c_void_p, c_void_p,
                                      c_size_t, c_void_p]),
    "nbd_leapfrog_step_ev_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                         c_float, c_float, c_float, c_float, c_void_p, c_void_p,
                                         c_size_t, c_void_p, c_void_p, c_void_p]),
    "nbd_leapfrog_step_uniform_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                              c_float, c_float, c_float, c_float, c_void_p, c_void_p,
                                              c_size_t, c_void_p, c_void_p, c_void_p]),
    "nbd_euler_step_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float,
                                   c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nbd_energy_workspace_bytes": (c_size_t, [c_int]),
    "nbd_energy_f32": (c_int, [c_void_p, c_void_p, c_int, c_float, c_float, c_void_p, c_void_p,
                               c_size_t, c_void_p]),
    # --- generators on the device (csrc/generators.hip)
    "nbd_disk_workspace_bytes": (c_size_t, [c_int]),
    "nbd_disk_from_draws_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_double, c_double, c_double, c_double,
                                        c_double, c_int, c_void_p, POINTER(c_double), POINTER(c_double), c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nbd_spiral_from_draws_f64": (c_int, [c_void_p, c_int, c_double, c_double, c_double, c_double, c_double, c_int,
                                          c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    # --- dataset CSV rows, host code (csrc/csv_format.hip)
    "nbd_format_f32": (c_int, [c_float, c_void_p]),
    "nbd_format_f32_array": (c_int, [c_void_p, c_int64, c_void_p, c_int]),
    "nbd_csv_state_bound": (c_size_t, [c_int, c_size_t, c_size_t, c_size_t]),
    "nbd_csv_format_state": (c_int64, [c_void_p, c_size_t, c_char_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int, c_char_p, c_size_t]),
    # --- surrogate models: graph build (csrc/graph.hip)
    "nbd_knn_graph_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64,
                                  c_void_p, c_void_p]),
    "nbd_knn_graph_hint_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64,
                                       c_void_p, c_void_p, c_void_p]),
    "nbd_radius_cached_state_bytes": (c_size_t, [c_int, c_int]),
    "nbd_radius_cached_workspace_bytes": (c_size_t, [c_int, c_int]),
    "nbd_radius_cached_search_f32": (c_int, [c_void_p, c_int, c_float, c_float, c_float, c_int, c_int, c_int, c_void_p,
                                             c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nbd_radius_cached_transpose_f32": (c_int, [c_void_p, c_int, c_float, c_int, c_int, c_void_p, c_size_t, c_void_p,
                                                c_void_p, c_void_p, c_void_p]),
    "nbd_radius_search_f32": (c_int, [c_void_p, c_int, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "nbd_radius_search_workspace_bytes": (c_size_t, [c_int, c_int]),
    "nbd_radius_search_ws_f32": (c_int, [c_void_p, c_int, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nbd_radius_drop_self_i32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "nbd_radius_transpose_lists": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p]),
    "nbd_radius_transpose_count_f32": (c_int, [c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p]),
    "nbd_radius_transpose_fill_f32": (c_int, [c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p]),
    "nbd_exclusive_scan_i32": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "nbd_rowptr_sorted_i64": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "nbd_ell_to_edge_index": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p]),
    # --- surrogate models: dense blocks (csrc/nn.hip)
    "nbd_linear_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                               c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "nbd_linear_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "nbd_edgeconv_aggregate_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                           c_void_p, c_int, c_void_p]),
    "nbd_edge_messages_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p]),
    "nbd_segment_reduce_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "nbd_layernorm_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int,
                                  c_void_p]),
    "nbd_ln_mlp_head_lds_bytes": (c_size_t, [c_int, c_int, POINTER(c_int)]),
    "nbd_ln_mlp_head_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_int, POINTER(c_void_p),
                                    POINTER(c_void_p), POINTER(c_int), c_void_p, c_int, c_void_p, c_float, c_int, c_void_p]),
    "nbd_contconv_bin_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                     c_float, c_void_p, c_int, c_void_p, c_void_p]),
    "nbd_contconv_fused_supported": (c_int, [c_int, c_int, c_int]),
    "nbd_ball_to_cube_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "nbd_trilinear_interpolate_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "nbd_contconv_pairs_layout": (c_int, [c_int, c_int64, c_int, POINTER(c_size_t)]),
    "nbd_contconv_pairs_bytes": (c_size_t, [c_int, c_int64, c_int]),
    "nbd_contconv_pairs_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_float, c_void_p, c_int,
                                       c_void_p, c_size_t, c_void_p]),
    "nbd_contconv_pairs_batch_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_float, c_int, POINTER(c_int),
                                             POINTER(c_void_p), POINTER(c_int), POINTER(c_void_p), POINTER(c_size_t),
                                             c_void_p]),
    "nbd_contconv_pairs_jobs_f32": (c_int, [c_void_p, c_int, c_float, c_int, POINTER(CcPairsJob), c_void_p]),
    "nbd_contconv_filter_grad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "nbd_contconv_filter_grad_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int64,
                                             c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nbd_contconv_filter_grad_full_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int64,
                                                  c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nbd_contconv_shuffle_filters_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "nbd_contconv_fused_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "nbd_contconv_filter_floats": (c_size_t, [c_int, c_int, c_int]),
    "nbd_contconv_fused_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_int, c_int,
                                       c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "nbd_degree_scale_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "nbd_gnn_forward_f32": (c_int, [c_void_p, c_void_p]),
    "nbd_cc_train_workspace_bytes": (c_size_t, [POINTER(CcTrainArgs)]),
    "nbd_cc_train_forward_f32": (c_int, [POINTER(CcTrainArgs), c_void_p]),
    "nbd_cc_train_backward_f32": (c_int, [POINTER(CcTrainArgs), c_void_p, c_int, POINTER(CcTrainGrads), c_void_p]),
    "nbd_gnn_train_workspace_bytes": (c_size_t, [POINTER(GnnTrainArgs)]),
    "nbd_gnn_train_forward_f32": (c_int, [POINTER(GnnTrainArgs), c_void_p]),
    "nbd_gnn_train_backward_f32": (c_int, [POINTER(GnnTrainArgs), c_void_p, c_int, POINTER(GnnTrainGrads), c_void_p]),
    "nbd_struct_size": (c_size_t, [ctypes.c_char_p]),
    "nbd_gnn_layer_f32": (c_int, [POINTER(GnnLayerArgs), c_void_p]),
    "nbd_gnn_forward_workspace_bytes": (c_size_t, [POINTER(GnnForwardArgs)]),
    "nbd_knn_graph_hint_pq_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p, POINTER(KnnPqArgs),
                                          c_void_p]),
    # --- backward kernels (csrc/train.hip) and the transposed adjacency they gather over
    "nbd_csr_by_key_i64": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p]),
    "nbd_act_bwd_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                c_void_p]),
    "nbd_batchnorm_train_workspace_bytes": (c_size_t, [c_int, c_int]),
    "nbd_batchnorm_train_fwd_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_int,
                                            c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nbd_batchnorm_train_bwd_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                            c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                            c_void_p, c_size_t, c_void_p]),
    "nbd_contconv_bin_bwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                         c_float, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "nbd_colsum_workspace_bytes": (c_size_t, [c_int, c_int]),
    "nbd_colsum_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nbd_linear_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "nbd_linear_wgrad_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                     c_void_p, c_size_t, c_void_p]),
    "nbd_linear_wgrad_bias_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "nbd_linear_wgrad_bias_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                                          c_void_p, c_void_p, c_size_t, c_void_p]),
    "nbd_edgeconv_aggregate_bwd_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                               c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "nbd_segment_max_bwd_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                        c_void_p, c_int, c_void_p]),
    "nbd_segment_mul_bwd_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "nbd_layernorm_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "nbd_layernorm_bwd_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_float, c_void_p, c_int, c_void_p, c_int,
                                      c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
}

_lib = None


# C struct name -> its ctypes mirror (include/nbd.h; sizes checked against nbd_struct_size() when the library is loaded)
STRUCT_MIRRORS = {"nbd_gnn_layer_args": GnnLayerArgs, "nbd_gnn_forward_args": GnnForwardArgs, "nbd_knn_pq_args": KnnPqArgs,
                  "nbd_gnn_train_args": GnnTrainArgs, "nbd_gnn_train_grads": GnnTrainGrads, "nbd_cc_train_args": CcTrainArgs,
                  "nbd_cc_train_grads": CcTrainGrads, "nbd_cc_pairs_job": CcPairsJob}


def lib() -> ctypes.CDLL:
    """The loaded library; raises NbdError (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            why = ""
            try:                      # a fresh checkout has sources only: compile now if hipcc is here
                build()
            except (NbdError, OSError) as exc:      # keep hipcc's / make's own message for the caller
                why = f"\nThe automatic build failed: {exc}"
            if not os.path.exists(LIB_PATH):
                raise NbdError(
                    f"{LIB_PATH} not found: the HIP extension is not built. Run "
                    "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                    "There is no CPU fallback for this path." + why)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.nbd_abi_version() != ABI_VERSION:
            raise NbdError("libnbd_hip.so ABI version mismatch; rebuild")
        for cname, mirror in STRUCT_MIRRORS.items():       # a stale .so beside newer Python (or the reverse) must not run
            if handle.nbd_struct_size(cname.encode()) != ctypes.sizeof(mirror):
                raise NbdError(f"libnbd_hip.so: sizeof({cname}) = {handle.nbd_struct_size(cname.encode())}, the ctypes mirror "
                               f"has {ctypes.sizeof(mirror)} bytes; rebuild the library (python -c 'import __graft_entry__ as g; g.build()')")
        _lib = handle
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().nbd_strerror(code)
        raise (NbdUnsupported if code == -3 else NbdError)(f"{what} failed with code {code}: {msg.decode() if msg else '?'}")


def ptr(t) -> int | None:
    """Device pointer of a torch tensor (None passes NULL)."""
    return None if t is None else t.data_ptr()


def current_stream(device=None) -> int:
    """Raw hipStream_t of torch's current stream on `device` (the fast accessor: this is called once per
    kernel launch, and torch.cuda.current_stream() costs ~5 us of Python per call)."""
    idx = getattr(device, "index", device)
    if idx is None:
        idx = torch.cuda.current_device()
    return _raw_stream(idx)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (
    lambda idx: torch.cuda.current_stream(idx).cuda_stream)


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(device):
    """Context that makes `device` current for a launch; a no-op object when it already is."""
    idx = getattr(device, "index", None)
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)
