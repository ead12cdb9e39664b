// train.hip -- backward kernels of the surrogate models for gfx950 (MI355X), fp32 throughout: what
// `loss.backward()` needs in the reference's training step (gnn.py:150-191, contconv.py:236-247,
// trainer.py:20-92) for the layers whose forward lives in nn.hip.
//
//   nbd_act_bwd_f32              g = dy * act'(y)           (tanh: 1 - y^2), the Tanh between Linears
//   nbd_colsum_f32               out[c] = sum_n w_n x[n][c] (bias gradients; fixed two-stage order)
//   nbd_linear_wgrad_f32         dW[m][k] = sum_n g[n][m] x[n][k]  -- torch.nn.Linear's weight gradient,
//                                fp32 MFMA with the reduction over ROWS, split over row slabs and
//                                summed in fixed order (deterministic, no float atomics)
//   nbd_edgeconv_aggregate_bwd_f32   d[P|Q] from dS for S_i = scale_i sum_j tanh(P_i + Q_j): the P half
//                                gathers over each target's edge list, the Q half over each SOURCE's
//                                list of targets (the transposed adjacency, nbd_csr_by_key_i64), so both
//                                are gathers with a fixed summation order
//   nbd_layernorm_bwd_f32        dx, and per-block partial dgamma/dbeta reduced by nbd_colsum_f32
//
// (the data gradient of a Linear, dX = g W, is nbd_linear_f32 itself with the transposed weight.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nbd.h"

typedef float f16v __attribute__((ext_vector_type(16)));

namespace {

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int status() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : (int)e; }

// same tanh as the forward aggregation (nn.hip), so that the recomputed messages are the forward's
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// ------------------------------------------------------------------ activation backward
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y,
                                                      int ldy, int act, const float* __restrict__ rowscale,
                                                      float* __restrict__ g, int ldg, int n, int c) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n * c) return;
  const int r = (int)(idx / c), col = (int)(idx - (size_t)r * c);
  const float d = dy[(size_t)r * lddy + col];
  float v = d;
  if (act == 1) { const float t = y[(size_t)r * ldy + col]; v = d * (1.0f - t * t); }
  if (rowscale) v *= rowscale[r];
  g[(size_t)r * ldg + col] = v;
}

// ------------------------------------------------------------------ column sums (two stages)
constexpr int kColRows = 128;     // rows per stage-1 block (512 at first: 128 dependent loads per thread, 26 us for 2 000 rows)
// stage 1: grid = (row blocks, 64-column groups); the block's 256 threads are 4 row phases x 64 columns
// (coalesced 256-B reads), each thread four independent partial sums (loads in flight together) combined in fixed
// order; the 4 phase sums meet in LDS in fixed order.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int ldx,
                                                             const float* __restrict__ w, int n, int c,
                                                             float* __restrict__ part) {
  __shared__ float red[256];
  const int r0 = blockIdx.x * kColRows, r1 = min(r0 + kColRows, n);
  const int col = blockIdx.y * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < c) {
    int r = r0 + ph;
    if (w) {
      for (; r + 12 < r1; r += 16) {
        s0 = __builtin_fmaf(w[r], x[(size_t)r * ldx + col], s0);
        s1 = __builtin_fmaf(w[r + 4], x[(size_t)(r + 4) * ldx + col], s1);
        s2 = __builtin_fmaf(w[r + 8], x[(size_t)(r + 8) * ldx + col], s2);
        s3 = __builtin_fmaf(w[r + 12], x[(size_t)(r + 12) * ldx + col], s3);
      }
      for (; r < r1; r += 4) s0 = __builtin_fmaf(w[r], x[(size_t)r * ldx + col], s0);
    } else {
      for (; r + 12 < r1; r += 16) {
        s0 += x[(size_t)r * ldx + col];
        s1 += x[(size_t)(r + 4) * ldx + col];
        s2 += x[(size_t)(r + 8) * ldx + col];
        s3 += x[(size_t)(r + 12) * ldx + col];
      }
      for (; r < r1; r += 4) s0 += x[(size_t)r * ldx + col];
    }
  }
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ph == 0 && col < c)
    part[(size_t)blockIdx.x * c + col] = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}
// stage 2: grid = 64-column groups; the block's 256 threads are 4 phases x 64 columns, phase p adds the partial rows
// p, p + 4, ... (two accumulators each: loads in flight together), the four phase sums meet in LDS -- one fixed
// summation order. (Thread = column walking all partial rows: 10 us for the 74 partial rows of a 9 458-row batch.)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int ldpart, int blocks,
                                                           int c, float* __restrict__ out) {
  __shared__ float red[256];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f;
  if (col < c) {
    int b = ph;
    for (; b + 4 < blocks; b += 8) {
      s0 += part[(size_t)b * ldpart + col];
      s1 += part[(size_t)(b + 4) * ldpart + col];
    }
    if (b < blocks) s0 += part[(size_t)b * ldpart + col];
  }
  red[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (ph == 0 && col < c)
    out[col] = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}

// ------------------------------------------------------------------ weight gradient: dW = G^T X
// Block = 4 waves, output tile 64 (m) x 64 (k), wave (wm, wk) owns a 32 x 32 quarter. The reduction
// runs over rows: per step 32 rows of G[:, m-tile] and X[:, k-tile] are staged in LDS row-major with
// a stride of 96 floats (== 32 mod 64 banks), so that the MFMA operand reads -- lanes 0-31: 32
// consecutive columns of row r, lanes 32-63: the same columns of row r+1 -- touch 64 distinct banks.
constexpr int WG_T = 64, WG_R = 32, WG_LD = 96;
// With `db` the bias gradient rides along as one more column of X: column k holds the row weight (1 without), so
// output column k is db[m] = sum_n w_n g[n][m] -- the two launches of a separate column sum are gone (a training step
// has eleven Linears; its backward pass is bounded by launches, tools/train_phases.py).
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ X,
                                                    int ldx, int n, int m, int k, int rows_per_slab,
                                                    float* __restrict__ out, int ldo, size_t slab_stride,
                                                    const float* __restrict__ roww = nullptr, float* __restrict__ db = nullptr,
                                                    int with_bias = 0) {
  __shared__ float gs[WG_R * WG_LD], xs[WG_R * WG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wk = wave >> 1;
  const int m0 = blockIdx.x * WG_T, k0 = blockIdx.y * WG_T;
  const int r_begin = blockIdx.z * rows_per_slab, r_end = min(r_begin + rows_per_slab, n);
  f16v acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int lc = tid & 63, lr = tid >> 6;            // loader: 64 columns x 4 rows per pass
  // The next stage's 32 rows travel to registers while this stage multiplies (one global-load latency per stage was
  // the whole cost of a 2 000-row gradient: 26 us for 64 MFMAs).
  float gr[WG_R / 4], xr[WG_R / 4];
  const bool gcol = m0 + lc < m, xcol = k0 + lc < k, bcol = with_bias && k0 + lc == k;
  auto fetch = [&](int r0) {
#pragma unroll
    for (int u = 0; u < WG_R / 4; ++u) {
      const int row = r0 + lr + 4 * u;
      const bool ok = row < r_end;
      gr[u] = (ok && gcol) ? G[(size_t)row * ldg + m0 + lc] : 0.f;
      xr[u] = (ok && xcol) ? X[(size_t)row * ldx + k0 + lc] : (ok && bcol) ? (roww ? roww[row] : 1.0f) : 0.f;
    }
  };
  if (r_begin < r_end) fetch(r_begin);
  for (int r0 = r_begin; r0 < r_end; r0 += WG_R) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < WG_R / 4; ++u) {
      gs[(lr + 4 * u) * WG_LD + lc] = gr[u];
      xs[(lr + 4 * u) * WG_LD + lc] = xr[u];
    }
    __syncthreads();
    if (r0 + WG_R < r_end) fetch(r0 + WG_R);
    const float* a_ptr = gs + (lane >> 5) * WG_LD + wm * 32 + (lane & 31);
    const float* b_ptr = xs + (lane >> 5) * WG_LD + wk * 32 + (lane & 31);
#pragma unroll
    for (int rr = 0; rr < WG_R; rr += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ptr[rr * WG_LD], b_ptr[rr * WG_LD], acc, 0, 0, 0);
  }
  // C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float* o = out + blockIdx.z * slab_stride;
  const int col = k0 + wk * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < m && col < k) o[(size_t)row * ldo + col] = acc[r];
    if (row < m && with_bias && col == k) {               // slabs: the extra column of the slab; one slab: db itself
      if (db) db[row] = acc[r]; else o[(size_t)row * ldo + k] = acc[r];
    }
  }
}
// kk = k (+ 1 with the bias column): slabs are m x kk; column k goes to db
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ slabs, int n_slabs, int m, int k,
                                                           float* __restrict__ dw, int lddw, int kk = 0,
                                                           float* __restrict__ db = nullptr) {
  if (kk == 0) kk = k;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= m * kk) return;
  float s0 = 0.f, s1 = 0.f;
  int b = 0;
  for (; b + 2 <= n_slabs; b += 2) {                      // two chains: the loads of a pair of slabs travel together
    s0 += slabs[(size_t)b * m * kk + idx];
    s1 += slabs[(size_t)(b + 1) * m * kk + idx];
  }
  if (b < n_slabs) s0 += slabs[(size_t)b * m * kk + idx];
  const float s = s0 + s1;
  const int row = idx / kk, col = idx - row * kk;
  if (col < k) dw[(size_t)row * lddw + col] = s; else db[row] = s;
}
struct WgradPlan { int slabs, rows_per_slab; size_t ws; };
WgradPlan plan_wgrad(int n, int m, int k) {
  const int tiles = ceil_div(m, WG_T) * ceil_div(k, WG_T);
  int s = ceil_div(512, tiles);
  if (s > 32) s = 32;                                 // the fixed-order finish walks the slabs serially
  const int max_s = ceil_div(n, 2 * WG_R);            // >= 64 rows per slab
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  WgradPlan p;
  p.rows_per_slab = ceil_div(ceil_div(n, s), WG_R) * WG_R;
  p.slabs = ceil_div(n, p.rows_per_slab);
  p.ws = p.slabs > 1 ? (size_t)p.slabs * m * k * sizeof(float) : 0;
  return p;
}

// ------------------------------------------------------------------ EdgeConv aggregation backward
// One wave per node, lanes own channels (as the forward). by_source = false: node is a target i, its
// list holds sources j:  dP_i = scale_i dS_i (.) sum_j (1 - t_ij^2).  by_source = true: node is a
// source j, its list holds targets i:  dQ_j = sum_i scale_i dS_i (.) (1 - t_ij^2).  t_ij = tanh(P_i + Q_j).
// scale_i = 1 (sum) or 1 / max(deg_i, 1) (mean), deg_i from the by-target lists.
__device__ __forceinline__ float tgt_scale(const int* rowptr, int fixed_k, int i, int mean) {
  if (!mean) return 1.0f;
  const int d = rowptr ? rowptr[i + 1] - rowptr[i] : fixed_k;
  return 1.0f / (float)max(d, 1);
}
template <bool BY_SOURCE>
__global__ __launch_bounds__(256) void edgeconv_bwd_kernel(
    const float* __restrict__ PQ, int ldpq, int H, const float* __restrict__ dS, int ldds,
    const int* __restrict__ rowptr, const int64_t* __restrict__ src, int fixed_k,      // by-target lists
    const int* __restrict__ rowptr_t, const int* __restrict__ tgt_t,                   // by-source lists
    int n, int mean, float* __restrict__ dPQ, int lddpq) {
  const int v = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (v >= n) return;
  const int lane = threadIdx.x & 63;
  if (!BY_SOURCE) {
    const int e0 = rowptr ? rowptr[v] : v * fixed_k, e1 = rowptr ? rowptr[v + 1] : (v + 1) * fixed_k;
    const float sc = tgt_scale(rowptr, fixed_k, v, mean);
    for (int h = lane; h < H; h += 64) {
      const float p = PQ[(size_t)v * ldpq + h];
      float acc = 0.f;
      for (int e = e0; e < e1; ++e) {
        const int j = (int)src[e];
        const float t = fast_tanh(__fadd_rn(p, PQ[(size_t)j * ldpq + H + h]));
        acc += 1.0f - t * t;
      }
      dPQ[(size_t)v * lddpq + h] = (sc * dS[(size_t)v * ldds + h]) * acc;
    }
  } else {
    const int e0 = rowptr_t[v], e1 = rowptr_t[v + 1];
    for (int h = lane; h < H; h += 64) {
      const float q = PQ[(size_t)v * ldpq + H + h];
      float acc = 0.f;
      for (int e = e0; e < e1; ++e) {
        const int i = tgt_t[e];
        const float t = fast_tanh(__fadd_rn(PQ[(size_t)i * ldpq + h], q));
        const float d = tgt_scale(rowptr, fixed_k, i, mean) * dS[(size_t)i * ldds + h];
        acc = __builtin_fmaf(d, 1.0f - t * t, acc);
      }
      dPQ[(size_t)v * lddpq + H + h] = acc;
    }
  }
}

// ------------------------------------------------------------------ LayerNorm backward
// Block = 4 waves x kLnRows rows each. Per row (one wave): mean / rstd recomputed as the forward does,
// a = mean_c(dy g), b = mean_c(dy g xhat), dx = rstd (dy g - a - xhat b). The block then sums
// dy xhat and dy over its rows per channel into part[block][0:c | c:2c] for nbd_colsum's second stage.
constexpr int kLnRows = 4;        // (16 at first: 64 serial rows per block, 56 us for 2 000 rows)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int ldx, int c,
                                                            const float* __restrict__ gamma, float eps,
                                                            const float* __restrict__ dy, int lddy,
                                                            float* __restrict__ dx, int lddx, int n,
                                                            float* __restrict__ part) {
  __shared__ float stat[4 * kLnRows * 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * 4 * kLnRows;
  const float inv_c = 1.0f / (float)c;
  for (int rr = 0; rr < kLnRows; ++rr) {
    const int lr = wave * kLnRows + rr, row = row0 + lr;
    float mean = 0.f, rstd = 0.f;
    if (row < n) {
      const float* xr = x + (size_t)row * ldx;
      const float* dr = dy + (size_t)row * lddy;
      float s = 0.f;
      for (int h = lane; h < c; h += 64) s += xr[h];
      mean = wave_sum(s) * inv_c;
      float v = 0.f;
      for (int h = lane; h < c; h += 64) { const float d = xr[h] - mean; v += d * d; }
      rstd = 1.0f / sqrtf(wave_sum(v) * inv_c + eps);
      float a = 0.f, b = 0.f;
      for (int h = lane; h < c; h += 64) {
        const float dg = dr[h] * (gamma ? gamma[h] : 1.0f), xh = (xr[h] - mean) * rstd;
        a += dg; b += dg * xh;
      }
      a = wave_sum(a) * inv_c; b = wave_sum(b) * inv_c;
      float* o = dx + (size_t)row * lddx;
      for (int h = lane; h < c; h += 64) {
        const float dg = dr[h] * (gamma ? gamma[h] : 1.0f), xh = (xr[h] - mean) * rstd;
        o[h] = rstd * (dg - a - xh * b);
      }
    }
    if (lane == 0) { stat[lr * 2] = mean; stat[lr * 2 + 1] = rstd; }
  }
  __syncthreads();
  const int rows_here = min(4 * kLnRows, n - row0);
  for (int h = threadIdx.x; h < c; h += 256) {
    float dg = 0.f, db = 0.f;
    for (int lr = 0; lr < rows_here; ++lr) {
      const float d = dy[(size_t)(row0 + lr) * lddy + h];
      const float xh = (x[(size_t)(row0 + lr) * ldx + h] - stat[lr * 2]) * stat[lr * 2 + 1];
      dg = __builtin_fmaf(d, xh, dg); db += d;
    }
    part[(size_t)blockIdx.x * 2 * c + h] = dg;
    part[(size_t)blockIdx.x * 2 * c + c + h] = db;
  }
}

// ------------------------------------------------------------------ segment max backward (aggr = "max")
// dm[e][c] = dx[i][c] for the FIRST row e of target i whose message equals the maximum, else 0.
__global__ __launch_bounds__(256) void segment_max_bwd_kernel(const float* __restrict__ m, int ldm, int H,
                                                              const float* __restrict__ x, int ldx,
                                                              const int* __restrict__ rowptr, int n,
                                                              const float* __restrict__ dx, int lddx,
                                                              float* __restrict__ dm, int lddm) {
  const int i = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const int e0 = rowptr[i], e1 = rowptr[i + 1];
  for (int h = lane; h < H; h += 64) {
    const float top = x[(size_t)i * ldx + h], d = dx[(size_t)i * lddx + h];
    bool given = false;
    for (int e = e0; e < e1; ++e) {
      const bool hit = !given && m[(size_t)e * ldm + h] == top;
      dm[(size_t)e * lddm + h] = hit ? d : 0.f;
      given = given || hit;
    }
  }
}

// ------------------------------------------------------------------ segment product backward (agg = "mul")
// dm[e][c] = dx[i][c] * prod_{e' != e} m[e'][c]: exclusive prefix times exclusive suffix, no division (exact with zeros).
// The suffix products are parked in dm itself on a first pass from the back, then the forward pass multiplies in the prefix.
__global__ __launch_bounds__(256) void segment_mul_bwd_kernel(const float* __restrict__ m, int ldm, int H,
                                                              const int* __restrict__ rowptr, int n,
                                                              const float* __restrict__ dx, int lddx,
                                                              float* __restrict__ dm, int lddm) {
  const int i = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const int e0 = rowptr[i], e1 = rowptr[i + 1];
  for (int h = lane; h < H; h += 64) {
    float suf = dx[(size_t)i * lddx + h];
    for (int e = e1 - 1; e >= e0; --e) { dm[(size_t)e * lddm + h] = suf; suf = __fmul_rn(suf, m[(size_t)e * ldm + h]); }
    float pre = 1.f;
    for (int e = e0; e < e1; ++e) {
      dm[(size_t)e * lddm + h] = __fmul_rn(dm[(size_t)e * lddm + h], pre);
      pre = __fmul_rn(pre, m[(size_t)e * ldm + h]);
    }
  }
}

// ------------------------------------------------------------------ BatchNorm1d, training mode
// (the PyG MLP of contconv.py:136-141 carries BatchNorm; a module that was never put in eval() normalises
// with the batch statistics.) Column statistics in two fixed-order stages like nbd_colsum_f32.
// stage 1, MODE 0: part[b][c] = sum_r x            MODE 1: part[b][c] = sum_r (x - mean_c)^2
//          MODE 2: part[b][0:c] = sum_r g xhat, part[b][c:2c] = sum_r g   with g = dy * act'(y)
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int ldx, int n, int c,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ dy, int lddy,
                                                         const float* __restrict__ y, int ldy, int act,
                                                         float* __restrict__ part) {
  __shared__ float red[2][256];
  const int r0 = blockIdx.x * kColRows, r1 = min(r0 + kColRows, n);
  const int col = blockIdx.y * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  float s = 0.f, s2 = 0.f;
  if (col < c) {
    const float mu = MODE ? mean[col] : 0.f, rs = MODE == 2 ? rstd[col] : 0.f;
    for (int r = r0 + ph; r < r1; r += 4) {
      const float v = x[(size_t)r * ldx + col];
      if (MODE == 0) s += v;
      if (MODE == 1) { const float d = v - mu; s = __builtin_fmaf(d, d, s); }
      if (MODE == 2) {
        float g = dy[(size_t)r * lddy + col];
        if (act == 1) { const float t = y[(size_t)r * ldy + col]; g *= 1.0f - t * t; }
        s = __builtin_fmaf(g, (v - mu) * rs, s);
        s2 += g;
      }
    }
  }
  red[0][threadIdx.x] = s; red[1][threadIdx.x] = s2;
  __syncthreads();
  if (ph == 0 && col < c) {
    const int t = threadIdx.x;
    const float a = (red[0][t] + red[0][t + 64]) + (red[0][t + 128] + red[0][t + 192]);
    if (MODE == 2) {
      part[(size_t)blockIdx.x * 2 * c + col] = a;
      part[(size_t)blockIdx.x * 2 * c + c + col] = (red[1][t] + red[1][t + 64]) + (red[1][t + 128] + red[1][t + 192]);
    } else part[(size_t)blockIdx.x * c + col] = a;
  }
}
// out[col] = (sum_b part[b][col]) * scale; optionally also rstd = 1/sqrt(out + eps)
__global__ __launch_bounds__(256) void bn_final_kernel(const float* __restrict__ part, int ldpart, int blocks, int c,
                                                       float scale, float* __restrict__ out, float eps,
                                                       float* __restrict__ rstd) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= c) return;
  float s = 0.f;
  for (int b = 0; b < blocks; ++b) s += part[(size_t)b * ldpart + col];
  s *= scale;
  out[col] = s;
  if (rstd) rstd[col] = 1.0f / sqrtf(s + eps);
}
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int ldx, int n, int c,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int act, float* __restrict__ y, int ldy) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n * c) return;
  const int r = (int)(idx / c), col = (int)(idx - (size_t)r * c);
  float v = (x[(size_t)r * ldx + col] - mean[col]) * rstd[col];
  if (gamma) v *= gamma[col];
  if (beta) v += beta[col];
  y[(size_t)r * ldy + col] = act == 1 ? tanhf(v) : v;
}
// dx = gamma rstd (g - dbeta/n - xhat dgamma/n)
__global__ __launch_bounds__(256) void bn_dx_kernel(const float* __restrict__ x, int ldx, int n, int c,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma, const float* __restrict__ dy, int lddy,
                                                    const float* __restrict__ y, int ldy, int act,
                                                    const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                    float* __restrict__ dx, int lddx) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n * c) return;
  const int r = (int)(idx / c), col = (int)(idx - (size_t)r * c);
  float g = dy[(size_t)r * lddy + col];
  if (act == 1) { const float t = y[(size_t)r * ldy + col]; g *= 1.0f - t * t; }
  const float xh = (x[(size_t)r * ldx + col] - mean[col]) * rstd[col];
  const float inv_n = 1.0f / (float)n;
  dx[(size_t)r * lddx + col] = (gamma ? gamma[col] : 1.0f) * rstd[col] * (g - dbeta[col] * inv_n - xh * dgamma[col] * inv_n);
}

// ------------------------------------------------------------------ ContinuousConv binning, backward
// dfeat[c][i] = sum over the edges (n <- c) of window_e * sum_{8 corners} t_corner(e) * dA[n][cell][i]:
// the adjoint of contconv_bin_kernel, as a gather per SOURCE c (one wave per source and 64-channel
// group, lane = channel) over the source's list of aggregation targets n -- the radius search's own
// per-centre lists (ELL: nbr[c][0..deg[c]), or CSR rowptr_s/tgt_s for a caller-supplied edge list).
struct Geo { int ix, iy, iz; float tx, ty, tz, window; };
__device__ __forceinline__ Geo edge_geo(const float* __restrict__ pos, float xc, float yc, float zc, int n_node,
                                        float r2max, float half) {
  Geo g;
  // r = positions[col] - positions[row] (contconv.py:84): source c minus target n -- same ops as the forward
  const float rx = xc - pos[3 * n_node], ry = yc - pos[3 * n_node + 1], rz = zc - pos[3 * n_node + 2];
  const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(rx, rx), __fmul_rn(ry, ry)), __fmul_rn(rz, rz));
  const float qq = 1.0f - d2 / r2max;
  g.window = (d2 < r2max) ? qq * qq * qq : 0.f;
  const float nrm = sqrtf(d2);
  const float sc = tanhf(nrm) / (nrm + 1e-8f);
  const float gx = (rx * sc + 1.0f) * half, gy = (ry * sc + 1.0f) * half, gz = (rz * sc + 1.0f) * half;
  const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
  g.ix = (int)fx; g.iy = (int)fy; g.iz = (int)fz;
  g.tx = gx - fx; g.ty = gy - fy; g.tz = gz - fz;
  return g;
}
__global__ __launch_bounds__(64) void contconv_bin_bwd_kernel(
    const float* __restrict__ pos, const float* __restrict__ dA, int I, const int* __restrict__ rowptr_s,
    const int* __restrict__ tgt_s, const int* __restrict__ deg, int cap, int D, float r2max,
    const int* __restrict__ cell_map, int cells_out, float* __restrict__ dfeat, int lddf) {
  __shared__ Geo geo[64];
  __shared__ int tgt[64];
  const int c = blockIdx.x, lane = threadIdx.x, ch = blockIdx.y * 64 + lane;
  const int e0 = rowptr_s ? rowptr_s[c] : c * cap, e1 = rowptr_s ? rowptr_s[c + 1] : e0 + deg[c];
  const float xc = pos[3 * c], yc = pos[3 * c + 1], zc = pos[3 * c + 2];
  const float half = (float)(D - 1) / 2.0f;
  const size_t row_len = (size_t)cells_out * I;
  float acc = 0.f;
  for (int eb = e0; eb < e1; eb += 64) {
    const int cnt = min(64, e1 - eb);
    __builtin_amdgcn_wave_barrier();
    if (lane < cnt) {
      const int n_node = tgt_s[eb + lane];
      tgt[lane] = n_node;
      geo[lane] = edge_geo(pos, xc, yc, zc, n_node, r2max, half);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (ch < I) {
      for (int e = 0; e < cnt; ++e) {
        const Geo g = geo[e];
        if (g.window == 0.f) continue;
        const float* row = dA + (size_t)tgt[e] * row_len + ch;
        float sum = 0.f;
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
          const int ax = corner & 1, ay = (corner >> 1) & 1, az = corner >> 2;
          const int cx = g.ix + ax, cy = g.iy + ay, cz = g.iz + az;
          if (cx < 0 || cx >= D || cy < 0 || cy >= D || cz < 0 || cz >= D) continue;
          const float t = ((ax ? g.tx : 1.0f - g.tx) * (ay ? g.ty : 1.0f - g.ty)) * (az ? g.tz : 1.0f - g.tz);
          const int cell = (cz * D + cy) * D + cx;
          const int mc = cell_map ? cell_map[cell] : cell;
          if (mc >= 0) sum = __builtin_fmaf(t, row[(size_t)mc * I], sum);
        }
        acc = __builtin_fmaf(g.window, sum, acc);
      }
    }
  }
  if (ch < I) dfeat[(size_t)c * lddf + ch] = acc;
}

}  // namespace

extern "C" {

int nbd_act_bwd_f32(const float* dy, int lddy, const float* y, int ldy, int act, const float* rowscale, float* g,
                    int ldg, int n, int c, nbd_stream_t stream) {
  if (n < 0 || c < 0 || act < 0 || act > 1) return NBD_E_BADARG;
  if (n == 0 || c == 0) return 0;
  if (!dy || !g || (act == 1 && !y) || lddy < c || ldg < c || (act == 1 && ldy < c)) return NBD_E_BADARG;
  const size_t total = (size_t)n * c;
  act_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(dy, lddy, y, ldy, act, rowscale, g, ldg, n, c);
  return status();
}

size_t nbd_colsum_workspace_bytes(int n, int c) {
  if (n <= 0 || c <= 0) return 0;
  return (size_t)ceil_div(n, kColRows) * c * sizeof(float);
}

int nbd_colsum_f32(const float* x, int ldx, const float* rowweight, int n, int c, float* out, void* workspace,
                   size_t workspace_bytes, nbd_stream_t stream) {
  if (n < 0 || c < 0) return NBD_E_BADARG;
  if (c == 0) return 0;
  if (!out || (n > 0 && (!x || ldx < c))) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = ceil_div(n, kColRows);
  if (n > 0 && (!workspace || workspace_bytes < nbd_colsum_workspace_bytes(n, c))) return NBD_E_BADARG;
  float* part = static_cast<float*>(workspace);
  if (n > 0) colsum_partial_kernel<<<dim3(blocks, ceil_div(c, 64)), 256, 0, st>>>(x, ldx, rowweight, n, c, part);
  colsum_final_kernel<<<ceil_div(c, 64), 256, 0, st>>>(part, c, blocks, c, out);
  return status();
}

size_t nbd_linear_wgrad_workspace_bytes(int n, int m, int k) {
  if (n <= 0 || m <= 0 || k <= 0) return 0;
  return plan_wgrad(n, m, k).ws;
}

size_t nbd_linear_wgrad_bias_workspace_bytes(int n, int m, int k) {
  if (n <= 0 || m <= 0 || k <= 0) return 0;
  const WgradPlan p = plan_wgrad(n, m, k + 1);
  return p.slabs > 1 ? (size_t)p.slabs * m * (k + 1) * sizeof(float) : 0;
}

int nbd_linear_wgrad_bias_f32(const float* g, int ldg, const float* x, int ldx, const float* rowweight, int n, int m, int k,
                              float* dw, int lddw, float* db, void* workspace, size_t workspace_bytes, nbd_stream_t stream) {
  if (n < 0 || m < 0 || k < 0) return NBD_E_BADARG;
  if (m == 0) return 0;
  if (!db || (k > 0 && (!dw || lddw < k)) || (n > 0 && (!g || ldg < m || (k > 0 && (!x || ldx < k))))) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ceil_div(m, WG_T), ceil_div(k + 1, WG_T), 1);
  if (n == 0) {
    wgrad_kernel<<<grid, 256, 0, st>>>(g, ldg, x, ldx, 0, m, k, WG_R, dw, lddw, 0, nullptr, db, 1);
    return status();
  }
  const WgradPlan p = plan_wgrad(n, m, k + 1);
  if (p.slabs == 1) {
    wgrad_kernel<<<grid, 256, 0, st>>>(g, ldg, x, ldx, n, m, k, p.rows_per_slab, dw, lddw, 0, rowweight, db, 1);
    return status();
  }
  if (!workspace || workspace_bytes < (size_t)p.slabs * m * (k + 1) * sizeof(float)) return NBD_E_BADARG;
  grid.z = p.slabs;
  float* slabs = static_cast<float*>(workspace);
  wgrad_kernel<<<grid, 256, 0, st>>>(g, ldg, x, ldx, n, m, k, p.rows_per_slab, slabs, k + 1, (size_t)m * (k + 1), rowweight, nullptr, 1);
  int rc = status();
  if (rc) return rc;
  wgrad_finish_kernel<<<ceil_div(m * (k + 1), 256), 256, 0, st>>>(slabs, p.slabs, m, k, dw, lddw, k + 1, db);
  return status();
}

int nbd_linear_wgrad_f32(const float* g, int ldg, const float* x, int ldx, int n, int m, int k, float* dw, int lddw,
                         void* workspace, size_t workspace_bytes, nbd_stream_t stream) {
  if (n < 0 || m < 0 || k < 0) return NBD_E_BADARG;
  if (m == 0 || k == 0) return 0;
  if (!dw || lddw < k || (n > 0 && (!g || !x || ldg < m || ldx < k))) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ceil_div(m, WG_T), ceil_div(k, WG_T), 1);
  if (n == 0) {           // empty batch: dW = 0 (one slab over zero rows writes zeros)
    wgrad_kernel<<<grid, 256, 0, st>>>(g, ldg, x, ldx, 0, m, k, WG_R, dw, lddw, 0);
    return status();
  }
  const WgradPlan p = plan_wgrad(n, m, k);
  if (p.slabs == 1) {
    wgrad_kernel<<<grid, 256, 0, st>>>(g, ldg, x, ldx, n, m, k, p.rows_per_slab, dw, lddw, 0);
    return status();
  }
  if (!workspace || workspace_bytes < p.ws) return NBD_E_BADARG;
  grid.z = p.slabs;
  float* slabs = static_cast<float*>(workspace);
  wgrad_kernel<<<grid, 256, 0, st>>>(g, ldg, x, ldx, n, m, k, p.rows_per_slab, slabs, k, (size_t)m * k);
  int rc = status();
  if (rc) return rc;
  wgrad_finish_kernel<<<ceil_div(m * k, 256), 256, 0, st>>>(slabs, p.slabs, m, k, dw, lddw);
  return status();
}

int nbd_edgeconv_aggregate_bwd_f32(const float* pq, int ldpq, int h, const float* ds, int ldds, const int* rowptr,
                                   const int64_t* src, int fixed_k, const int* rowptr_t, const int* tgt_t, int n,
                                   int aggr, float* dpq, int lddpq, nbd_stream_t stream) {
  if (n < 0 || h < 0 || aggr < 0 || aggr > 1 || (!rowptr && fixed_k < 0)) return NBD_E_BADARG;
  if (n == 0 || h == 0) return 0;
  if (!pq || !ds || !dpq || !rowptr_t || ldpq < 2 * h || lddpq < 2 * h || ldds < h) return NBD_E_BADARG;
  if (!src && (rowptr || fixed_k > 0)) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  edgeconv_bwd_kernel<false><<<ceil_div(n, 4), 256, 0, st>>>(pq, ldpq, h, ds, ldds, rowptr, src, fixed_k, rowptr_t,
                                                             tgt_t, n, aggr, dpq, lddpq);
  edgeconv_bwd_kernel<true><<<ceil_div(n, 4), 256, 0, st>>>(pq, ldpq, h, ds, ldds, rowptr, src, fixed_k, rowptr_t,
                                                            tgt_t, n, aggr, dpq, lddpq);
  return status();
}

size_t nbd_batchnorm_train_workspace_bytes(int n, int c) {
  if (n <= 0 || c <= 0) return 0;
  return (size_t)ceil_div(n, kColRows) * 2 * c * sizeof(float);
}

int nbd_batchnorm_train_fwd_f32(const float* x, int ldx, int n, int c, const float* gamma, const float* beta, float eps,
                                int act, float* y, int ldy, float* mean, float* var, float* rstd, void* workspace,
                                size_t workspace_bytes, nbd_stream_t stream) {
  if (n < 2 || c <= 0 || act < 0 || act > 1) return NBD_E_BADARG;      // torch: more than one value per channel
  if (!x || !y || !mean || !var || !rstd || ldx < c || ldy < c) return NBD_E_BADARG;
  if (!workspace || workspace_bytes < nbd_batchnorm_train_workspace_bytes(n, c)) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  float* part = static_cast<float*>(workspace);
  const int blocks = ceil_div(n, kColRows), cb = ceil_div(c, 256);
  bn_partial_kernel<0><<<dim3(blocks, ceil_div(c, 64)), 256, 0, st>>>(x, ldx, n, c, nullptr, nullptr, nullptr, 0, nullptr, 0, 0, part);
  bn_final_kernel<<<cb, 256, 0, st>>>(part, c, blocks, c, 1.0f / (float)n, mean, 0.f, nullptr);
  bn_partial_kernel<1><<<dim3(blocks, ceil_div(c, 64)), 256, 0, st>>>(x, ldx, n, c, mean, nullptr, nullptr, 0, nullptr, 0, 0, part);
  bn_final_kernel<<<cb, 256, 0, st>>>(part, c, blocks, c, 1.0f / (float)n, var, eps, rstd);
  const size_t total = (size_t)n * c;
  bn_apply_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(x, ldx, n, c, mean, rstd, gamma, beta, act, y, ldy);
  return status();
}

int nbd_batchnorm_train_bwd_f32(const float* x, int ldx, int n, int c, const float* gamma, const float* mean,
                                const float* rstd, int act, const float* y, int ldy, const float* dy, int lddy,
                                float* dx, int lddx, float* dgamma, float* dbeta, void* workspace,
                                size_t workspace_bytes, nbd_stream_t stream) {
  if (n < 2 || c <= 0 || act < 0 || act > 1) return NBD_E_BADARG;
  if (!x || !mean || !rstd || !dy || !dx || !dgamma || !dbeta || (act == 1 && !y)) return NBD_E_BADARG;
  if (ldx < c || lddy < c || lddx < c || (act == 1 && ldy < c)) return NBD_E_BADARG;
  if (!workspace || workspace_bytes < nbd_batchnorm_train_workspace_bytes(n, c)) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  float* part = static_cast<float*>(workspace);
  const int blocks = ceil_div(n, kColRows), cb = ceil_div(c, 256);
  bn_partial_kernel<2><<<dim3(blocks, ceil_div(c, 64)), 256, 0, st>>>(x, ldx, n, c, mean, rstd, dy, lddy, y, ldy, act, part);
  bn_final_kernel<<<cb, 256, 0, st>>>(part, 2 * c, blocks, c, 1.0f, dgamma, 0.f, nullptr);
  bn_final_kernel<<<cb, 256, 0, st>>>(part + c, 2 * c, blocks, c, 1.0f, dbeta, 0.f, nullptr);
  const size_t total = (size_t)n * c;
  bn_dx_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(x, ldx, n, c, mean, rstd, gamma, dy, lddy, y, ldy, act,
                                                                 dgamma, dbeta, dx, lddx);
  return status();
}

int nbd_contconv_bin_bwd_f32(const float* pos, const float* da, int in_channels, const int* rowptr_s, const int* tgt_s,
                             const int* deg, int cap, int n, int filter_resolution, float radius_sq,
                             const int* cell_map, int cells_out, float* dfeat, int lddf, nbd_stream_t stream) {
  if (n < 0 || in_channels <= 0 || filter_resolution < 1 || !(radius_sq > 0.f)) return NBD_E_BADARG;
  const int cells = filter_resolution * filter_resolution * filter_resolution;
  if (!cell_map) cells_out = cells;
  if (cells_out <= 0 || cells_out > cells) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!pos || !da || !tgt_s || !dfeat || lddf < in_channels) return NBD_E_BADARG;
  if (!rowptr_s && (!deg || cap < 0)) return NBD_E_BADARG;
  dim3 grid(n, ceil_div(in_channels, 64));
  contconv_bin_bwd_kernel<<<grid, 64, 0, (hipStream_t)stream>>>(pos, da, in_channels, rowptr_s, tgt_s, deg, cap,
                                                               filter_resolution, radius_sq, cell_map, cells_out, dfeat,
                                                               lddf);
  return status();
}

int nbd_segment_mul_bwd_f32(const float* m, int ldm, int h, const int* rowptr, int n, const float* dx, int lddx,
                            float* dm, int lddm, nbd_stream_t stream) {
  if (n < 0 || h < 0) return NBD_E_BADARG;
  if (n == 0 || h == 0) return 0;
  if (!m || !rowptr || !dx || !dm || ldm < h || lddx < h || lddm < h) return NBD_E_BADARG;
  segment_mul_bwd_kernel<<<ceil_div(n, 4), 256, 0, (hipStream_t)stream>>>(m, ldm, h, rowptr, n, dx, lddx, dm, lddm);
  return status();
}

int nbd_segment_max_bwd_f32(const float* m, int ldm, int h, const float* x, int ldx, const int* rowptr, int n,
                            const float* dx, int lddx, float* dm, int lddm, nbd_stream_t stream) {
  if (n < 0 || h < 0) return NBD_E_BADARG;
  if (n == 0 || h == 0) return 0;
  if (!m || !x || !rowptr || !dx || !dm || ldm < h || ldx < h || lddx < h || lddm < h) return NBD_E_BADARG;
  segment_max_bwd_kernel<<<ceil_div(n, 4), 256, 0, (hipStream_t)stream>>>(m, ldm, h, x, ldx, rowptr, n, dx, lddx, dm, lddm);
  return status();
}

size_t nbd_layernorm_bwd_workspace_bytes(int n, int c) {
  if (n <= 0 || c <= 0) return 0;
  return (size_t)ceil_div(n, 4 * kLnRows) * 2 * c * sizeof(float);
}

int nbd_layernorm_bwd_f32(const float* x, int ldx, int c, const float* gamma, float eps, const float* dy, int lddy,
                          float* dx, int lddx, float* dgamma, float* dbeta, int n, void* workspace,
                          size_t workspace_bytes, nbd_stream_t stream) {
  if (n < 0 || c <= 0) return NBD_E_BADARG;
  if (!dgamma || !dbeta) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = ceil_div(n, 4 * kLnRows);
  if (n > 0) {
    if (!x || !dy || !dx || ldx < c || lddy < c || lddx < c) return NBD_E_BADARG;
    if (!workspace || workspace_bytes < nbd_layernorm_bwd_workspace_bytes(n, c)) return NBD_E_BADARG;
    layernorm_bwd_kernel<<<blocks, 256, 0, st>>>(x, ldx, c, gamma, eps, dy, lddy, dx, lddx, n,
                                                 static_cast<float*>(workspace));
  }
  float* part = static_cast<float*>(workspace);            // [blocks][dgamma partials (c) | dbeta partials (c)]
  colsum_final_kernel<<<ceil_div(c, 64), 256, 0, st>>>(part, 2 * c, blocks, c, dgamma);
  colsum_final_kernel<<<ceil_div(c, 64), 256, 0, st>>>(part ? part + c : part, 2 * c, blocks, c, dbeta);
  return status();
}

}  // extern "C"
