// train_model.hip -- a whole model's training pass behind ONE C-ABI call per direction (gfx950).
//
// The reference's training step (gnn.py:150-191, trainer.py:60-72) is `loss.backward()` over a chain of PyG ops; round 2
// ran it as ~20 torch.autograd Functions whose forward and backward were single kernels of this library: 0.52 ms of
// kernels inside 1.7 ms of Python / autograd-engine time per step (tools/train_phases.py). Here the chain itself is
// native: nbd_gnn_train_forward_f32 enqueues every kernel of GraphModel.forward (gnn.py:130-148) and keeps what the
// backward pass needs in a caller-owned workspace, nbd_gnn_train_backward_f32 enqueues the whole adjoint and writes every
// parameter gradient. No allocation, no synchronisation, no global state; torch sees ONE autograd node and keeps the
// optimiser (a torch.optim object, as in the reference). The kernels are the ones of nn.hip / train.hip -- the same
// fixed summation orders, so gradients stay bit-reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nbd.h"

namespace {

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int status() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : (int)e; }

// wt[c][r] = w[r][c]: the operand of dX = g W as nbd_linear_f32 wants it (weights are at most a few hundred square)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ w, int rows, int cols, float* __restrict__ wt) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * cols) return;
  const int c = idx / rows, r = idx - c * rows;
  wt[idx] = w[(size_t)r * cols + c];
}
// EdgeConv's first Linear W1 = [W1a | W1b] (H x 2F) in the per-node form: rows 0..H-1 = W1a - W1b (P), rows H..2H-1 = W1b (Q);
// bias [b1 | 0]
__global__ __launch_bounds__(256) void pq_weight_kernel(const float* __restrict__ w1, const float* __restrict__ b1, int H, int F,
                                                        float* __restrict__ wpq, float* __restrict__ bpq) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < 2 * H) bpq[idx] = idx < H ? b1[idx] : 0.f;
  if (idx >= 2 * H * F) return;
  const int row = idx / F, f = idx - row * F;
  const int h = row < H ? row : row - H;
  const float a = w1[(size_t)h * 2 * F + f], b = w1[(size_t)h * 2 * F + F + f];
  wpq[idx] = row < H ? a - b : b;
}
// ... and its adjoint: dW1a = dWP, dW1b = dWQ - dWP, db1 = dbpq[:H]
__global__ __launch_bounds__(256) void pq_weight_bwd_kernel(const float* __restrict__ dwpq, const float* __restrict__ dbpq, int H,
                                                            int F, float* __restrict__ dw1, float* __restrict__ db1) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < H) db1[idx] = dbpq[idx];
  if (idx >= H * F) return;
  const int h = idx / F, f = idx - h * F;
  const float dp = dwpq[(size_t)h * F + f], dq = dwpq[(size_t)(H + h) * F + f];
  dw1[(size_t)h * 2 * F + f] = dp;
  dw1[(size_t)h * 2 * F + F + f] = dq - dp;
}
__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int n,
                                                     int c) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n * c) return;
  const int r = (int)(idx / c), col = (int)(idx - (size_t)r * c);
  dst[(size_t)r * ldd + col] = src[(size_t)r * lds + col];
}
__global__ __launch_bounds__(256) void add2d_kernel(float* __restrict__ dst, int ldd, const float* __restrict__ src, int lds, int n,
                                                    int c) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n * c) return;
  const int r = (int)(idx / c), col = (int)(idx - (size_t)r * c);
  dst[(size_t)r * ldd + col] += src[(size_t)r * lds + col];
}
__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, int n, float v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

struct Arena {
  char* base; size_t off;
  float* take(size_t floats) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += (floats * sizeof(float) + 255) & ~(size_t)255;
    return p;
  }
};
struct Scratch { void* p; size_t bytes; };

struct Lin { const float* w; const float* b; int in, out; };

// y = act(x W^T + brs * b)
int lin_fwd(const float* x, int ldx, const Lin& L, int act, const float* brs, float* y, int ldy, int n, const Scratch& sc,
            nbd_stream_t st) {
  return nbd_linear_f32(x, ldx, L.w, L.in, L.b, nullptr, brs, act, y, ldy, n, L.out, L.in, sc.p, sc.bytes, st);
}
// Adjoint of lin_fwd: g = dy * act'(y) (into gbuf when act = tanh), dW = g^T x, db = sum_n brs_n g_n, dx = g W (optional).
int lin_bwd(const float* dy, int lddy, const float* y, int ldy, int act, const float* x, int ldx, const Lin& L, const float* brs,
            float* gbuf, float* dW, float* db, float* dx, int lddx, float* wt, int n, const Scratch& sc, nbd_stream_t st) {
  const float* g = dy;
  int ldg = lddy;
  int rc = 0;
  if (act == 1) {
    rc = nbd_act_bwd_f32(dy, lddy, y, ldy, 1, nullptr, gbuf, L.out, n, L.out, st);
    if (rc) return rc;
    g = gbuf; ldg = L.out;
  }
  if (dW && db) {           // the bias gradient as one more column of the weight gradient's product
    rc = nbd_linear_wgrad_bias_f32(g, ldg, x, ldx, brs, n, L.out, L.in, dW, L.in, db, sc.p, sc.bytes, st);
    if (rc) return rc;
  } else {
    if (dW) { rc = nbd_linear_wgrad_f32(g, ldg, x, ldx, n, L.out, L.in, dW, L.in, sc.p, sc.bytes, st); if (rc) return rc; }
    if (db) { rc = nbd_colsum_f32(g, ldg, brs, n, L.out, db, sc.p, sc.bytes, st); if (rc) return rc; }
  }
  if (dx) {
    transpose_kernel<<<ceil_div(L.in * L.out, 256), 256, 0, (hipStream_t)st>>>(L.w, L.out, L.in, wt);
    rc = nbd_linear_f32(g, ldg, wt, L.out, nullptr, nullptr, nullptr, 0, dx, lddx, n, L.in, L.out, sc.p, sc.bytes, st);
  }
  return rc;
}

struct GnnWs {
  float* enc_act[NBD_TRAIN_MAX_MLP];       // encoder layer outputs but the last (that one lands in zcat[:, :E])
  float* brs;                              // bias row scale of the W2 Linear: deg (sum) / [deg > 0] (mean)
  float *wpq[NBD_GNN_MAX_LAYERS], *bpq[NBD_GNN_MAX_LAYERS], *pq[NBD_GNN_MAX_LAYERS], *s[NBD_GNN_MAX_LAYERS], *xl[NBD_GNN_MAX_LAYERS];
  float *zcat, *ln, *head_act[NBD_TRAIN_MAX_MLP];
  float *g0, *g1, *dz, *dpq, *ds, *dwpq, *dbpq, *wt, *dxa, *dxb;
  Scratch sc;
  int E, C, M;
};

bool gnn_args_ok(const nbd_gnn_train_args& a) {
  if (a.n < 0 || a.f <= 0 || a.h <= 0 || a.n_layers < 1 || a.n_layers > NBD_GNN_MAX_LAYERS) return false;
  if (a.n_enc < 0 || a.n_enc > NBD_TRAIN_MAX_MLP || a.n_head < 1 || a.n_head > NBD_TRAIN_MAX_MLP) return false;
  if (a.aggr < 0 || a.aggr > 1 || (!a.rowptr && a.fixed_k < 0)) return false;
  if (a.n_enc && a.enc_dim[0] != a.f) return false;
  const int E = a.n_enc ? a.enc_dim[a.n_enc] : a.f;
  if (a.head_dim[0] != E + a.h) return false;
  for (int i = 0; i <= a.n_enc && a.n_enc; ++i) if (a.enc_dim[i] <= 0) return false;
  for (int i = 0; i <= a.n_head; ++i) if (a.head_dim[i] <= 0) return false;
  return true;
}

GnnWs gnn_layout(const nbd_gnn_train_args& a, void* base, size_t* total) {
  Arena ar{static_cast<char*>(base), 0};
  GnnWs w{};
  const size_t n = (size_t)(a.n > 0 ? a.n : 1);
  const int H = a.h;
  w.E = a.n_enc ? a.enc_dim[a.n_enc] : a.f;
  w.C = w.E + H;
  int M = w.C > 2 * H ? w.C : 2 * H;
  for (int i = 0; i <= a.n_enc && a.n_enc; ++i) M = a.enc_dim[i] > M ? a.enc_dim[i] : M;
  for (int i = 0; i <= a.n_head; ++i) M = a.head_dim[i] > M ? a.head_dim[i] : M;
  if (a.f > M) M = a.f;
  w.M = M;
  for (int i = 0; i + 1 < a.n_enc; ++i) w.enc_act[i] = ar.take(n * a.enc_dim[i + 1]);
  w.brs = ar.take(n);
  for (int l = 0; l < a.n_layers; ++l) {
    const int F = l == 0 ? w.E : H;
    w.wpq[l] = ar.take((size_t)2 * H * F); w.bpq[l] = ar.take(2 * H);
    w.pq[l] = ar.take(n * 2 * H); w.s[l] = ar.take(n * H);
    w.xl[l] = l + 1 < a.n_layers ? ar.take(n * H) : nullptr;
  }
  w.zcat = ar.take(n * w.C); w.ln = ar.take(n * w.C);
  for (int i = 0; i + 1 < a.n_head; ++i) w.head_act[i] = ar.take(n * a.head_dim[i + 1]);
  w.g0 = ar.take(n * M); w.g1 = ar.take(n * M); w.dz = ar.take(n * w.C);
  w.dpq = ar.take(n * 2 * H); w.ds = ar.take(n * H);
  w.dwpq = ar.take((size_t)2 * H * M); w.dbpq = ar.take(2 * H); w.wt = ar.take((size_t)M * M);
  w.dxa = ar.take(n * M); w.dxb = ar.take(n * M);
  // scratch of the library calls: split-K partials of a Linear, slabs of a weight gradient (<= 32 of m x k), column-sum /
  // LayerNorm partials
  size_t sb = (size_t)32 * M * (M + 1) * sizeof(float);
  const size_t cs = nbd_colsum_workspace_bytes(a.n, 2 * M), ls = nbd_layernorm_bwd_workspace_bytes(a.n, w.C);
  const size_t lw = nbd_linear_workspace_bytes(a.n, M, M);
  sb = cs > sb ? cs : sb; sb = ls > sb ? ls : sb; sb = lw > sb ? lw : sb;
  w.sc.bytes = sb;
  w.sc.p = ar.take(sb / sizeof(float) + 64);
  if (total) *total = ar.off + 256;
  return w;
}

}  // namespace

extern "C" {

size_t nbd_gnn_train_workspace_bytes(const nbd_gnn_train_args* args) {
  if (!args || !gnn_args_ok(*args)) return 0;
  size_t total = 0;
  gnn_layout(*args, nullptr, &total);
  return total;
}

int nbd_gnn_train_forward_f32(const nbd_gnn_train_args* args, nbd_stream_t stream) {
  if (!args || !gnn_args_ok(*args)) return NBD_E_BADARG;
  const nbd_gnn_train_args& a = *args;
  if (a.n == 0) return NBD_E_UNSUPPORTED;      // an empty batch has no workspace to hand to the backward pass: the caller's own zero path
  if (!a.x || a.ldx < a.f || !a.out || !a.workspace || (reinterpret_cast<uintptr_t>(a.workspace) & 255)) return NBD_E_BADARG;
  if (!a.src && (a.rowptr || a.fixed_k > 0)) return NBD_E_BADARG;
  size_t need = 0;
  const GnnWs w = gnn_layout(a, a.workspace, &need);
  if (a.workspace_bytes < need) return NBD_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int n = a.n, H = a.h, E = w.E, C = w.C;
  int rc = 0;
  // ---- node encoder (gnn.py:57-63,134-137): Linear + tanh ... plain last Linear, into the left columns of zcat
  if (a.n_enc == 0) {
    copy2d_kernel<<<(unsigned)(((size_t)n * E + 255) / 256), 256, 0, st>>>(a.x, a.ldx, w.zcat, C, n, E);
  } else {
    const float* cur = a.x;
    int ld = a.ldx;
    for (int i = 0; i < a.n_enc; ++i) {
      const bool last = i == a.n_enc - 1;
      const Lin L{a.enc_w[i], a.enc_b[i], a.enc_dim[i], a.enc_dim[i + 1]};
      float* y = last ? w.zcat : w.enc_act[i];
      const int ldy = last ? C : L.out;
      rc = lin_fwd(cur, ld, L, last ? 0 : 1, nullptr, y, ldy, n, w.sc, stream);
      if (rc) return rc;
      cur = y; ld = ldy;
    }
  }
  // ---- bias row scale of the hoisted second Linear: sum_j (W2 t_ij + b2) = W2 S_i + deg_i b2; mean: [deg_i > 0] b2
  if (a.rowptr) {
    rc = nbd_degree_scale_f32(a.rowptr, n, a.aggr == 1 ? 2 : 1, w.brs, stream);
    if (rc) return rc;
  } else {
    fill_kernel<<<ceil_div(n, 256), 256, 0, st>>>(w.brs, n, a.aggr == 1 ? (a.fixed_k > 0 ? 1.f : 0.f) : (float)a.fixed_k);
  }
  // ---- EdgeConv layers (gnn.py:75-93,140-141) in the per-node factored form
  const float* xin = w.zcat;
  int ldx = C;
  for (int l = 0; l < a.n_layers; ++l) {
    const int F = l == 0 ? E : H;
    pq_weight_kernel<<<ceil_div(2 * H * F, 256), 256, 0, st>>>(a.w1[l], a.b1[l], H, F, w.wpq[l], w.bpq[l]);
    rc = lin_fwd(xin, ldx, Lin{w.wpq[l], w.bpq[l], F, 2 * H}, 0, nullptr, w.pq[l], 2 * H, n, w.sc, stream);
    if (rc) return rc;
    rc = nbd_edgeconv_aggregate_f32(w.pq[l], 2 * H, H, a.rowptr, a.src, a.fixed_k, n, a.aggr, w.s[l], H, stream);
    if (rc) return rc;
    const bool last = l == a.n_layers - 1;
    float* xo = last ? w.zcat + E : w.xl[l];
    const int ldo = last ? C : H;
    rc = lin_fwd(w.s[l], H, Lin{a.w2[l], a.b2[l], H, H}, 0, w.brs, xo, ldo, n, w.sc, stream);
    if (rc) return rc;
    xin = xo; ldx = ldo;
  }
  // ---- LayerNorm over [enc | x] and the decoder (gnn.py:144-148)
  rc = nbd_layernorm_f32(w.zcat, C, C, a.ln_g, a.ln_b, a.ln_eps, w.ln, C, n, stream);
  if (rc) return rc;
  const float* cur = w.ln;
  int ld = C;
  for (int i = 0; i < a.n_head; ++i) {
    const bool last = i == a.n_head - 1;
    const Lin L{a.head_w[i], a.head_b[i], a.head_dim[i], a.head_dim[i + 1]};
    float* y = last ? a.out : w.head_act[i];
    const int ldy = last ? a.ldout : L.out;
    if (last && a.ldout < L.out) return NBD_E_BADARG;
    rc = lin_fwd(cur, ld, L, last ? 0 : 1, nullptr, y, ldy, n, w.sc, stream);
    if (rc) return rc;
    cur = y; ld = ldy;
  }
  return status();
}

int nbd_gnn_train_backward_f32(const nbd_gnn_train_args* args, const float* dout, int lddout, const nbd_gnn_train_grads* grads,
                               nbd_stream_t stream) {
  if (!args || !grads || !gnn_args_ok(*args)) return NBD_E_BADARG;
  const nbd_gnn_train_args& a = *args;
  const nbd_gnn_train_grads& gr = *grads;
  size_t need = 0;
  if (a.n == 0) return NBD_E_UNSUPPORTED;
  if (!a.workspace || (reinterpret_cast<uintptr_t>(a.workspace) & 255)) return NBD_E_BADARG;
  const GnnWs w = gnn_layout(a, a.workspace, &need);
  if (a.workspace_bytes < need) return NBD_E_WORKSPACE;
  if (!dout || lddout < a.head_dim[a.n_head] || !a.rowptr_t || !a.tgt_t) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int n = a.n, H = a.h, E = w.E, C = w.C;
  int rc = 0;
  // ---- decoder, last Linear first
  const float* dy = dout;
  int lddy = lddout;
  for (int i = a.n_head - 1; i >= 0; --i) {
    const bool last = i == a.n_head - 1;
    const Lin L{a.head_w[i], a.head_b[i], a.head_dim[i], a.head_dim[i + 1]};
    const float* x = i == 0 ? w.ln : w.head_act[i - 1];
    const int ldx = i == 0 ? C : a.head_dim[i];
    const float* y = last ? a.out : w.head_act[i];
    const int ldy = last ? a.ldout : L.out;
    float* dx = (i & 1) ? w.dxa : w.dxb;
    rc = lin_bwd(dy, lddy, y, ldy, last ? 0 : 1, x, ldx, L, nullptr, w.g0, gr.head_w[i], gr.head_b[i], dx, L.in, w.wt, n, w.sc, stream);
    if (rc) return rc;
    dy = dx; lddy = L.in;
  }
  // ---- LayerNorm
  rc = nbd_layernorm_bwd_f32(w.zcat, C, C, a.ln_g, a.ln_eps, dy, lddy, w.dz, C, gr.ln_g, gr.ln_b, n, w.sc.p, w.sc.bytes, stream);
  if (rc) return rc;
  // ---- EdgeConv layers, last first. dz = [d enc | d x_L]
  const float* dxl = w.dz + E;
  int lddxl = C;
  for (int l = a.n_layers - 1; l >= 0; --l) {
    const int F = l == 0 ? E : H;
    const float* xin = l == 0 ? w.zcat : (l - 1 == a.n_layers - 1 ? w.zcat + E : w.xl[l - 1]);
    const int ldxin = l == 0 ? C : H;
    // x_l = s_l W2^T + brs b2
    rc = lin_bwd(dxl, lddxl, nullptr, 0, 0, w.s[l], H, Lin{a.w2[l], a.b2[l], H, H}, w.brs, w.g0, gr.w2[l], gr.b2[l], w.ds, H, w.wt,
                 n, w.sc, stream);
    if (rc) return rc;
    // s_l = aggr_j tanh(P_i + Q_j)
    rc = nbd_edgeconv_aggregate_bwd_f32(w.pq[l], 2 * H, H, w.ds, H, a.rowptr, a.src, a.fixed_k, a.rowptr_t, a.tgt_t, n, a.aggr,
                                        w.dpq, 2 * H, stream);
    if (rc) return rc;
    // pq_l = xin [W1a - W1b ; W1b]^T + [b1 | 0]; the input's gradient is needed unless the input is the model input itself
    const bool need_dx = l > 0 || a.n_enc > 0;
    float* dxin = need_dx ? ((l & 1) ? w.dxa : w.dxb) : nullptr;
    rc = lin_bwd(w.dpq, 2 * H, nullptr, 0, 0, xin, ldxin, Lin{w.wpq[l], w.bpq[l], F, 2 * H}, nullptr, w.g0, w.dwpq, w.dbpq, dxin, F,
                 w.wt, n, w.sc, stream);
    if (rc) return rc;
    pq_weight_bwd_kernel<<<ceil_div(H * F > H ? H * F : H, 256), 256, 0, st>>>(w.dwpq, w.dbpq, H, F, gr.w1[l], gr.b1[l]);
    if (l > 0) { dxl = dxin; lddxl = H; }
    else if (need_dx)                // the encoder's output feeds both layer 0 and the concatenation
      add2d_kernel<<<(unsigned)(((size_t)n * E + 255) / 256), 256, 0, st>>>(w.dz, C, dxin, E, n, E);
  }
  // ---- node encoder
  dy = w.dz; lddy = C;
  for (int i = a.n_enc - 1; i >= 0; --i) {
    const bool last = i == a.n_enc - 1;
    const Lin L{a.enc_w[i], a.enc_b[i], a.enc_dim[i], a.enc_dim[i + 1]};
    const float* x = i == 0 ? a.x : w.enc_act[i - 1];
    const int ldx = i == 0 ? a.ldx : a.enc_dim[i];
    const float* y = last ? w.zcat : w.enc_act[i];
    const int ldy = last ? C : L.out;
    float* dx = i > 0 ? ((i & 1) ? w.dxa : w.dxb) : nullptr;
    rc = lin_bwd(dy, lddy, y, ldy, last ? 0 : 1, x, ldx, L, nullptr, w.g1, gr.enc_w[i], gr.enc_b[i], dx, L.in, w.wt, n, w.sc, stream);
    if (rc) return rc;
    dy = dx; lddy = L.in;
  }
  return status();
}


// ============================================================================================ ContinuousConvModel
// contconv.py:218-247 the same way: node encoder (PyG MLP: Linear -> BatchNorm1d on BATCH statistics -> tanh per hidden
// layer, plain last Linear; or no encoder), the ContinuousConv layers on the pair lists of the caller's graph (forward:
// fused kernel; filters.grad: nbd_contconv_filter_grad_full_f32; feature gradient: the fused kernel over the adjoint
// lists with every cell's filter transposed), LayerNorm over [enc | h], decoder. The pair lists are the caller's
// (nbd_contconv_pairs_jobs_f32, once per forward pass for all layers); everything else is enqueued here.
namespace {

__global__ __launch_bounds__(256) void bn_running_kernel(const float* __restrict__ mean, const float* __restrict__ var, int c,
                                                         float momentum, float unbias, float* __restrict__ rmean,
                                                         float* __restrict__ rvar) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= c) return;
  // torch: running = (1 - momentum) * running + momentum * batch statistic (the variance unbiased, n / (n - 1))
  rmean[i] = rmean[i] * (1.0f - momentum) + mean[i] * momentum;
  rvar[i] = rvar[i] * (1.0f - momentum) + var[i] * (momentum * unbias);
}

struct CcWs {
  float* enc_lin[NBD_TRAIN_MAX_MLP];     // hidden encoder layers: the Linear's output (BatchNorm's input) ...
  float* enc_act[NBD_TRAIN_MAX_MLP];     // ... and tanh(BatchNorm(.)) (or tanh(.) without norm)
  float *bn_mean[NBD_TRAIN_MAX_MLP], *bn_var[NBD_TRAIN_MAX_MLP], *bn_rstd[NBD_TRAIN_MAX_MLP];
  float* h[NBD_GNN_MAX_LAYERS];          // conv layer outputs but the last (that one lands in zcat[:, E:])
  float *zcat, *ln, *head_act[NBD_TRAIN_MAX_MLP];
  float *wf, *g, *g0, *g1, *dz, *wt, *dxa, *dxb;
  void *fws, *gws; size_t fws_bytes, gws_bytes;
  Scratch sc;
  int E, C, M;
};

bool cc_args_ok(const nbd_cc_train_args& a) {
  if (a.n <= 0 || a.in_ch <= 0 || a.cdim <= 0 || a.cdim > 128 || a.n_layers < 1 || a.n_layers > NBD_GNN_MAX_LAYERS) return false;
  if (a.n_enc < 0 || a.n_enc > NBD_TRAIN_MAX_MLP || a.n_head < 1 || a.n_head > NBD_TRAIN_MAX_MLP) return false;
  if (a.n_enc && a.enc_dim[0] != a.in_ch) return false;
  const int E = a.n_enc ? a.enc_dim[a.n_enc] : a.in_ch;
  if (a.head_dim[0] != E + a.cdim || ((E + a.cdim) & 1)) return false;
  for (int i = 0; i <= a.n_enc && a.n_enc; ++i) if (a.enc_dim[i] <= 0) return false;
  for (int i = 0; i <= a.n_head; ++i) if (a.head_dim[i] <= 0) return false;
  for (int l = 0; l < a.n_layers; ++l) {
    const int I = l == 0 ? E : a.cdim;
    if (!nbd_contconv_fused_supported(I, a.cdim, a.n_cells[l]) || !nbd_contconv_fused_supported(a.cdim, I, a.n_cells[l])) return false;
    if (a.cells_total[l] < a.n_cells[l]) return false;
  }
  return true;
}

CcWs cc_layout(const nbd_cc_train_args& a, void* base, size_t* total) {
  Arena ar{static_cast<char*>(base), 0};
  CcWs w{};
  const size_t n = (size_t)a.n;
  const int O = a.cdim;
  w.E = a.n_enc ? a.enc_dim[a.n_enc] : a.in_ch;
  w.C = w.E + O;
  int M = w.C > O ? w.C : O;
  for (int i = 0; i <= a.n_enc && a.n_enc; ++i) M = a.enc_dim[i] > M ? a.enc_dim[i] : M;
  for (int i = 0; i <= a.n_head; ++i) M = a.head_dim[i] > M ? a.head_dim[i] : M;
  if (a.in_ch > M) M = a.in_ch;
  w.M = M;
  for (int i = 0; i + 1 < a.n_enc; ++i) {
    const int d = a.enc_dim[i + 1];
    w.enc_lin[i] = a.enc_bn ? ar.take(n * d) : nullptr;
    w.enc_act[i] = ar.take(n * d);
    w.bn_mean[i] = ar.take(d); w.bn_var[i] = ar.take(d); w.bn_rstd[i] = ar.take(d);
  }
  for (int l = 0; l + 1 < a.n_layers; ++l) w.h[l] = ar.take(n * O);
  w.zcat = ar.take(n * w.C); w.ln = ar.take(n * w.C);
  for (int i = 0; i + 1 < a.n_head; ++i) w.head_act[i] = ar.take(n * a.head_dim[i + 1]);
  size_t wf_floats = 0, fws = 0, gws = 0;
  for (int l = 0; l < a.n_layers; ++l) {
    const int I = l == 0 ? w.E : O;
    const size_t f1 = nbd_contconv_filter_floats(I, O, a.n_cells[l]), f2 = nbd_contconv_filter_floats(O, I, a.n_cells[l]);
    wf_floats = f1 > wf_floats ? f1 : wf_floats; wf_floats = f2 > wf_floats ? f2 : wf_floats;
    const size_t b1 = nbd_contconv_fused_workspace_bytes(a.n, a.n_cells[l], O), b2 = nbd_contconv_fused_workspace_bytes(a.n, a.n_cells[l], I);
    fws = b1 > fws ? b1 : fws; fws = b2 > fws ? b2 : fws;
    const size_t g1 = (size_t)a.n_cells[l] * I * O * sizeof(float) + nbd_contconv_filter_grad_workspace_bytes(a.n, a.n_cells[l], I, O);
    gws = g1 > gws ? g1 : gws;
  }
  w.wf = ar.take(wf_floats + 16);
  w.g = ar.take(n * M); w.g0 = ar.take(n * M); w.g1 = ar.take(n * M); w.dz = ar.take(n * w.C);
  w.wt = ar.take((size_t)M * M); w.dxa = ar.take(n * M); w.dxb = ar.take(n * M);
  w.fws_bytes = fws; w.fws = ar.take(fws / sizeof(float) + 64);
  w.gws_bytes = gws; w.gws = ar.take(gws / sizeof(float) + 64);
  size_t sb = (size_t)32 * M * (M + 1) * sizeof(float);
  const size_t cs = nbd_colsum_workspace_bytes(a.n, 2 * M), ls = nbd_layernorm_bwd_workspace_bytes(a.n, w.C);
  const size_t lw = nbd_linear_workspace_bytes(a.n, M, M), bw = nbd_batchnorm_train_workspace_bytes(a.n, M);
  sb = cs > sb ? cs : sb; sb = ls > sb ? ls : sb; sb = lw > sb ? lw : sb; sb = bw > sb ? bw : sb;
  w.sc.bytes = sb;
  w.sc.p = ar.take(sb / sizeof(float) + 64);
  if (total) *total = ar.off + 256;
  return w;
}

}  // namespace

size_t nbd_cc_train_workspace_bytes(const nbd_cc_train_args* args) {
  if (!args || !cc_args_ok(*args)) return 0;
  size_t total = 0;
  cc_layout(*args, nullptr, &total);
  return total;
}

int nbd_cc_train_forward_f32(const nbd_cc_train_args* args, nbd_stream_t stream) {
  if (!args || !cc_args_ok(*args)) return NBD_E_BADARG;
  const nbd_cc_train_args& a = *args;
  if (!a.x || a.ldx < a.in_ch || !a.out || !a.rowptr_fwd || !a.workspace || (reinterpret_cast<uintptr_t>(a.workspace) & 255))
    return NBD_E_BADARG;
  size_t need = 0;
  const CcWs w = cc_layout(a, a.workspace, &need);
  if (a.workspace_bytes < need) return NBD_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int n = a.n, O = a.cdim, E = w.E, C = w.C;
  int rc = 0;
  // ---- node encoder (contconv.py:135-143,222): Linear -> BatchNorm (batch statistics) -> tanh ... plain last Linear
  if (a.n_enc == 0) {
    copy2d_kernel<<<(unsigned)(((size_t)n * E + 255) / 256), 256, 0, st>>>(a.x, a.ldx, w.zcat, C, n, E);
  } else {
    const float* cur = a.x;
    int ld = a.ldx;
    for (int i = 0; i < a.n_enc; ++i) {
      const bool last = i == a.n_enc - 1;
      const Lin L{a.enc_w[i], a.enc_b[i], a.enc_dim[i], a.enc_dim[i + 1]};
      if (last) {
        rc = lin_fwd(cur, ld, L, 0, nullptr, w.zcat, C, n, w.sc, stream);
        if (rc) return rc;
      } else if (a.enc_bn) {
        rc = lin_fwd(cur, ld, L, 0, nullptr, w.enc_lin[i], L.out, n, w.sc, stream);
        if (rc) return rc;
        rc = nbd_batchnorm_train_fwd_f32(w.enc_lin[i], L.out, n, L.out, a.bn_g[i], a.bn_b[i], a.bn_eps[i], 1, w.enc_act[i], L.out,
                                         w.bn_mean[i], w.bn_var[i], w.bn_rstd[i], w.sc.p, w.sc.bytes, stream);
        if (rc) return rc;
        if (a.bn_rmean[i] && a.bn_rvar[i])     // running statistics, as torch updates them in training mode
          bn_running_kernel<<<ceil_div(L.out, 256), 256, 0, st>>>(w.bn_mean[i], w.bn_var[i], L.out, a.bn_momentum[i],
                                                                  n > 1 ? (float)n / (float)(n - 1) : 1.0f, a.bn_rmean[i],
                                                                  a.bn_rvar[i]);
        cur = w.enc_act[i]; ld = L.out;
        continue;
      } else {
        rc = lin_fwd(cur, ld, L, 1, nullptr, w.enc_act[i], L.out, n, w.sc, stream);
        if (rc) return rc;
      }
      cur = last ? w.zcat : w.enc_act[i]; ld = last ? C : L.out;
    }
  }
  // ---- ContinuousConv layers (contconv.py:80-98,225-231): h = tanh(scale * sum_cells A . F)
  const float* feat = w.zcat;
  int ldf = C;
  for (int l = 0; l < a.n_layers; ++l) {
    const int I = l == 0 ? E : O;
    const bool last = l == a.n_layers - 1;
    rc = nbd_contconv_shuffle_filters_f32(a.filt[l], a.kept[l], a.n_cells[l], I, O, 0, w.wf, stream);
    if (rc) return rc;
    float* ho = last ? w.zcat + E : w.h[l];
    const int ldo = last ? C : O;
    rc = nbd_contconv_fused_f32(feat, ldf, I, a.rowptr_fwd, n, a.cap_fwd, a.pairs_fwd[l], w.wf, a.n_cells[l], O, a.scale, 1, ho, ldo,
                                w.fws, w.fws_bytes, stream);
    if (rc) return rc;
    feat = ho; ldf = ldo;
  }
  // ---- LayerNorm over [enc | h] and the decoder (contconv.py:233-234)
  rc = nbd_layernorm_f32(w.zcat, C, C, a.ln_g, a.ln_b, a.ln_eps, w.ln, C, n, stream);
  if (rc) return rc;
  const float* cur = w.ln;
  int ld = C;
  for (int i = 0; i < a.n_head; ++i) {
    const bool last = i == a.n_head - 1;
    const Lin L{a.head_w[i], a.head_b[i], a.head_dim[i], a.head_dim[i + 1]};
    float* y = last ? a.out : w.head_act[i];
    const int ldy = last ? a.ldout : L.out;
    if (last && a.ldout < L.out) return NBD_E_BADARG;
    rc = lin_fwd(cur, ld, L, last ? 0 : 1, nullptr, y, ldy, n, w.sc, stream);
    if (rc) return rc;
    cur = y; ld = ldy;
  }
  return status();
}

int nbd_cc_train_backward_f32(const nbd_cc_train_args* args, const float* dout, int lddout, const nbd_cc_train_grads* grads,
                              nbd_stream_t stream) {
  if (!args || !grads || !cc_args_ok(*args)) return NBD_E_BADARG;
  const nbd_cc_train_args& a = *args;
  const nbd_cc_train_grads& gr = *grads;
  if (!a.workspace || (reinterpret_cast<uintptr_t>(a.workspace) & 255)) return NBD_E_BADARG;
  size_t need = 0;
  const CcWs w = cc_layout(a, a.workspace, &need);
  if (a.workspace_bytes < need) return NBD_E_WORKSPACE;
  if (!dout || lddout < a.head_dim[a.n_head]) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int n = a.n, O = a.cdim, E = w.E, C = w.C;
  int rc = 0;
  // ---- decoder, last Linear first
  const float* dy = dout;
  int lddy = lddout;
  for (int i = a.n_head - 1; i >= 0; --i) {
    const bool last = i == a.n_head - 1;
    const Lin L{a.head_w[i], a.head_b[i], a.head_dim[i], a.head_dim[i + 1]};
    const float* x = i == 0 ? w.ln : w.head_act[i - 1];
    const int ldx = i == 0 ? C : a.head_dim[i];
    const float* y = last ? a.out : w.head_act[i];
    const int ldy = last ? a.ldout : L.out;
    float* dx = (i & 1) ? w.dxa : w.dxb;
    rc = lin_bwd(dy, lddy, y, ldy, last ? 0 : 1, x, ldx, L, nullptr, w.g0, gr.head_w[i], gr.head_b[i], dx, L.in, w.wt, n, w.sc, stream);
    if (rc) return rc;
    dy = dx; lddy = L.in;
  }
  rc = nbd_layernorm_bwd_f32(w.zcat, C, C, a.ln_g, a.ln_eps, dy, lddy, w.dz, C, gr.ln_g, gr.ln_b, n, w.sc.p, w.sc.bytes, stream);
  if (rc) return rc;
  // ---- ContinuousConv layers, last first. dz = [d enc | d h_L]
  const float* dh = w.dz + E;
  int lddh = C;
  for (int l = a.n_layers - 1; l >= 0; --l) {
    const int I = l == 0 ? E : O;
    const bool last = l == a.n_layers - 1;
    const float* hl = last ? w.zcat + E : w.h[l];
    const int ldh = last ? C : O;
    const float* fin = l == 0 ? w.zcat : (l - 1 == a.n_layers - 1 ? w.zcat + E : w.h[l - 1]);
    const int ldfin = l == 0 ? C : O;
    // g = scale * (1 - h^2) * dh
    rc = nbd_act_bwd_f32(dh, lddh, hl, ldh, 1, a.scale, w.g, O, n, O, stream);
    if (rc) return rc;
    rc = nbd_contconv_filter_grad_full_f32(fin, ldfin, I, w.g, O, O, a.rowptr_fwd, n, a.cap_fwd, a.pairs_fwd[l], a.n_cells[l],
                                           a.cell_map[l], a.cells_total[l], gr.filt[l], w.gws, w.gws_bytes, stream);
    if (rc) return rc;
    const bool need_dx = l > 0 || a.n_enc > 0;
    if (need_dx) {
      if (!a.pairs_adj[l] || !a.rowptr_adj) return NBD_E_BADARG;
      rc = nbd_contconv_shuffle_filters_f32(a.filt[l], a.kept[l], a.n_cells[l], I, O, 1, w.wf, stream);
      if (rc) return rc;
      float* dfin = (l & 1) ? w.dxa : w.dxb;
      rc = nbd_contconv_fused_f32(w.g, O, O, a.rowptr_adj, n, a.cap_adj, a.pairs_adj[l], w.wf, a.n_cells[l], I, nullptr, 0, dfin, I,
                                  w.fws, w.fws_bytes, stream);
      if (rc) return rc;
      if (l > 0) { dh = dfin; lddh = O; }
      else add2d_kernel<<<(unsigned)(((size_t)n * E + 255) / 256), 256, 0, st>>>(w.dz, C, dfin, E, n, E);   // enc feeds layer 0 and the concatenation
    }
  }
  // ---- node encoder
  dy = w.dz; lddy = C;
  for (int i = a.n_enc - 1; i >= 0; --i) {
    const bool last = i == a.n_enc - 1;
    const Lin L{a.enc_w[i], a.enc_b[i], a.enc_dim[i], a.enc_dim[i + 1]};
    const float* x = i == 0 ? a.x : w.enc_act[i - 1];
    const int ldx = i == 0 ? a.ldx : a.enc_dim[i];
    float* dx = i > 0 ? ((i & 1) ? w.dxa : w.dxb) : nullptr;
    if (last) {
      rc = lin_bwd(dy, lddy, nullptr, 0, 0, x, ldx, L, nullptr, w.g1, gr.enc_w[i], gr.enc_b[i], dx, L.in, w.wt, n, w.sc, stream);
    } else if (a.enc_bn) {
      // y = tanh(BatchNorm(t)), t = Linear(x): dt, dgamma, dbeta from dy; then the Linear
      rc = nbd_batchnorm_train_bwd_f32(w.enc_lin[i], L.out, n, L.out, a.bn_g[i], w.bn_mean[i], w.bn_rstd[i], 1, w.enc_act[i], L.out,
                                       dy, lddy, w.g1, L.out, gr.bn_g[i], gr.bn_b[i], w.sc.p, w.sc.bytes, stream);
      if (rc) return rc;
      rc = lin_bwd(w.g1, L.out, nullptr, 0, 0, x, ldx, L, nullptr, w.g0, gr.enc_w[i], gr.enc_b[i], dx, L.in, w.wt, n, w.sc, stream);
    } else {
      rc = lin_bwd(dy, lddy, w.enc_act[i], L.out, 1, x, ldx, L, nullptr, w.g1, gr.enc_w[i], gr.enc_b[i], dx, L.in, w.wt, n, w.sc, stream);
    }
    if (rc) return rc;
    dy = dx; lddy = L.in;
  }
  return status();
}

}  // extern "C"
