// nn.hip -- dense blocks of the surrogate forward passes for gfx950 (MI355X), fp32 throughout.
//
//   nbd_linear_f32          Y = act(rowscale * (X W^T) + b): torch.nn.Linear (+tanh, + PyG-MLP/BatchNorm
//                           folded on the host) -- gnn.py:57-63,75-93,105-114; contconv.py:136-141,206-216;
//                           also the ContinuousConv contraction (contconv.py:92) after cell binning.
//                           fp32-input MFMA (v_mfma_f32_32x32x2_f32): exact fp32 products, k-ordered fmaf chain.
//   nbd_edgeconv_aggregate_f32   S_i = aggr_j tanh(P_i + Q_j) over the edges grouped by target i:
//                           EdgeConv (gnn.py:75-93) after factoring its first Linear per node:
//                           W1 [x_i || x_j - x_i] + b1 = (W1a - W1b) x_i + b1  +  W1b x_j  =  P_i + Q_j.
//   nbd_layernorm_f32       torch.nn.LayerNorm over the last dim (gnn.py:102,146; contconv.py:204,233).
//   nbd_contconv_bin_f32    A[n][cell][i] = sum_{edges e -> n} window_e * t_cell(e) * feat[c_e][i]:
//                           the trilinear filter lookup of contconv.py:53-98 moved from the filter side
//                           to the feature side, so that out = A (N x D^3 I) . filters (D^3 I x O) is one
//                           dense MFMA GEMM and the (E, I, O) interpolated-filter tensor never exists.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/nbd.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

namespace {

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int status() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : (int)e; }

__device__ __forceinline__ float act_apply(float v, int act) { return act == 1 ? tanhf(v) : v; }

// tanh(x) = 1 - 2 / (2^(2x log2 e) + 1): two transcendental + three VALU instructions instead of libm's
// ~40; abs error <= ~2e-7. Used where tanh is evaluated per EDGE and channel (the aggregation's bound).
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// ------------------------------------------------------------------ linear: NT GEMM on fp32 MFMA
// Block = 4 waves. Block tile (32*WM) rows x (32*NT*WN) cols with WM*WN = 4; each wave owns a
// 32 x (32*NT) strip: one A fragment feeds NT MFMAs. K is walked in steps of 32 through LDS tiles
// stored [row][k] with a +1 pad (odd stride => the lane->row fragment read is bank-conflict free:
// lanes 0-31 read 32 different rows at k, lanes 32-63 the same rows at k+1).
constexpr int BK = 32;
constexpr int LD = BK + 1;

template <int WM, int WN, int NT, bool VEC>
__global__ __launch_bounds__(256) void linear_kernel(
    const float* __restrict__ X, int ldx, const float* __restrict__ W, int ldw, const float* __restrict__ bias,
    const float* __restrict__ rowscale, const float* __restrict__ bias_rowscale, int act, float* __restrict__ Y,
    int ldy, int n_rows, int n_cols, int K) {
  constexpr int BM = 32 * WM, BN = 32 * NT * WN;
  __shared__ float lds[(BM + BN) * LD];
  float* As = lds;
  float* Bs = lds + BM * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int row0 = blockIdx.x * BM, col0 = blockIdx.y * BN;

  f16v acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // VEC path: the K-slab of step k + 1 is fetched into registers before the MFMAs of step k (a product with K = 256
  // otherwise exposes eight global-load latencies back to back: 20 us for 16 384 x 256 x 64 against 4 us of traffic)
  constexpr int NR = (BM + BN) / 32;            // K-slab rows per thread (8 lanes x float4 cover one 128-B row)
  f4 pre[NR];
  auto prefetch = [&](int k0) {
    const int q = tid & 7, r8 = tid >> 3;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int r = r8 + 32 * i;
      const bool isA = r < BM;
      const int rr = isA ? r : r - BM;
      const int g = (isA ? row0 : col0) + rr;
      const int lim = isA ? n_rows : n_cols;
      const int k = k0 + 4 * q;
      const bool ok = g < lim && k < K;                                   // K % 4 == 0 on this path
      const float* src = (isA ? X + (size_t)min(g, lim - 1) * ldx : W + (size_t)min(g, lim - 1) * ldw) + min(k, K - 4);
      const f4 v = *reinterpret_cast<const f4*>(src);                       // unpredicated: clamped address, masked value
      pre[i] = ok ? v : f4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if (VEC && K > 0) prefetch(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();   // previous step's fragment reads are done
    if (VEC) {
      const int q = tid & 7, r8 = tid >> 3;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int r = r8 + 32 * i;
        const bool isA = r < BM;
        const int rr = isA ? r : r - BM;
        float* dst = (isA ? As : Bs) + rr * LD + 4 * q;
        dst[0] = pre[i].x; dst[1] = pre[i].y; dst[2] = pre[i].z; dst[3] = pre[i].w;
      }
    } else {           // 32 lanes x 4 B cover one row of the K-slab
      const int kk = tid & 31, r8 = tid >> 5;
      for (int r = r8; r < BM + BN; r += 8) {
        const bool isA = r < BM;
        const int rr = isA ? r : r - BM;
        const int g = (isA ? row0 : col0) + rr;
        const int lim = isA ? n_rows : n_cols;
        const float* src = isA ? X + (size_t)g * ldx : W + (size_t)g * ldw;
        float v = 0.f;
        if (g < lim && k0 + kk < K) v = src[k0 + kk];
        ((isA ? As : Bs) + rr * LD)[kk] = v;
      }
    }
    __syncthreads();
    if (VEC && k0 + BK < K) prefetch(k0 + BK);
    const float* a_ptr = As + (wm * 32 + (lane & 31)) * LD + (lane >> 5);
    const float* b_ptr = Bs + (wn * 32 * NT + (lane & 31)) * LD + (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a = a_ptr[kk];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float b = b_ptr[t * 32 * LD + kk];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
      }
    }
  }
  // epilogue: C/D map of the 32x32 tile -- col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = col0 + wn * 32 * NT + t * 32 + (lane & 31);
    const float b = (bias && col < n_cols) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < n_rows && col < n_cols) {
        float v = acc[t][r];
        if (rowscale) v = __fmul_rn(v, rowscale[row]);
        const float bb = bias_rowscale ? __fmul_rn(b, bias_rowscale[row]) : b;
        Y[(size_t)row * ldy + col] = act_apply(__fadd_rn(v, bb), act);
      }
    }
  }
}


// ------------------------------------------------------------------ big linear: 128x128x32 tiles, split-K
// For the ContinuousConv contraction (rows = nodes, K = D^3*I up to 27 648, 128 output channels).
// Block 256 threads = 2x2 waves, each wave a 64x64 strip = 2x2 MFMA 32x32 tiles (64 accumulator regs).
// LDS tiles are [row][k] with a 36-float row stride: 16-B aligned for ds_write_b128 staging and
// conflict-free for the ds_read_b128 fragment reads (36*i mod 64 hits every 4-bank group once for
// i = 0..15). A lane's b128 gives 4 k-values for its row; lanes 0-31 take k0..k0+3, lanes 32-63
// k0+4..k0+7, so MFMA c pairs (k0+c, k0+4+c) -- any pairing is valid as long as A and B agree.
// Global->register prefetch of tile t+1 is issued before the 64 MFMAs of tile t (one barrier per step).
// gridDim.z = split-K slices; with more than one slice raw partials go to slabs[z] and
// linear_finish_kernel applies scale/bias/activation in fixed slice order (deterministic).
constexpr int G2_BM = 128, G2_BN = 128, G2_LD = 36;

__global__ __launch_bounds__(256) void linear_big_kernel(
    const float* __restrict__ X, int ldx, const float* __restrict__ W, int ldw, const float* __restrict__ bias,
    const float* __restrict__ rowscale, const float* __restrict__ bias_rowscale, int act, float* __restrict__ Y,
    int ldy, int n_rows, int n_cols, int K, int k_per_slice, float* __restrict__ slabs) {
  extern __shared__ float smem[];                 // 2 x (128 + 128) x 36 floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int row0 = blockIdx.x * G2_BM, col0 = blockIdx.y * G2_BN;
  const int k_begin = blockIdx.z * k_per_slice, k_end = min(K, k_begin + k_per_slice);
  const int q = tid & 7, r0 = tid >> 3;           // 8 lanes x float4 = one 128-B row piece; 32 rows per pass

  f16v acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  f4 pa[4], pb[4];
  auto load_tile = [&](int k0) {
    const int k = k0 + 4 * q;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ra = row0 + r0 + 32 * p, rb = col0 + r0 + 32 * p;
      pa[p] = (ra < n_rows && k < k_end) ? *reinterpret_cast<const f4*>(X + (size_t)ra * ldx + k) : f4{0.f, 0.f, 0.f, 0.f};
      pb[p] = (rb < n_cols && k < k_end) ? *reinterpret_cast<const f4*>(W + (size_t)rb * ldw + k) : f4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_tile = [&](int buf) {
    float* As = smem + buf * (G2_BM + G2_BN) * G2_LD;
    float* Bs = As + G2_BM * G2_LD;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *reinterpret_cast<f4*>(As + (r0 + 32 * p) * G2_LD + 4 * q) = pa[p];
      *reinterpret_cast<f4*>(Bs + (r0 + 32 * p) * G2_LD + 4 * q) = pb[p];
    }
  };

  load_tile(k_begin);
  store_tile(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += 32) {
    const bool more = k0 + 32 < k_end;
    if (more) load_tile(k0 + 32);
    const float* As = smem + buf * (G2_BM + G2_BN) * G2_LD;
    const float* Bs = As + G2_BM * G2_LD;
    const float* a_base = As + (wm * 64 + (lane & 31)) * G2_LD + (lane >> 5) * 4;
    const float* b_base = Bs + (wn * 64 + (lane & 31)) * G2_LD + (lane >> 5) * 4;
#pragma unroll
    for (int k8 = 0; k8 < 4; ++k8) {
      const f4 a0 = *reinterpret_cast<const f4*>(a_base + k8 * 8);
      const f4 a1 = *reinterpret_cast<const f4*>(a_base + 32 * G2_LD + k8 * 8);
      const f4 b0 = *reinterpret_cast<const f4*>(b_base + k8 * 8);
      const f4 b1 = *reinterpret_cast<const f4*>(b_base + 32 * G2_LD + k8 * 8);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b0[c], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b1[c], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b0[c], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b1[c], acc[1][1], 0, 0, 0);
      }
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  const bool direct = gridDim.z == 1;
  float* dst = direct ? Y : slabs + (size_t)blockIdx.z * n_rows * n_cols;
  const int ldd = direct ? ldy : n_cols;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = col0 + wn * 64 + b * 32 + (lane & 31);
      const float bv = (direct && bias && col < n_cols) ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < n_rows && col < n_cols) {
          float v = acc[a][b][r];
          if (direct) {
            if (rowscale) v = __fmul_rn(v, rowscale[row]);
            const float bb = bias_rowscale ? __fmul_rn(bv, bias_rowscale[row]) : bv;
            v = act_apply(__fadd_rn(v, bb), act);
          }
          dst[(size_t)row * ldd + col] = v;
        }
      }
    }
}

__global__ __launch_bounds__(256) void linear_finish_kernel(const float* __restrict__ slabs, int n_slabs,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ rowscale,
                                                            const float* __restrict__ bias_rowscale, int act,
                                                            float* __restrict__ Y, int ldy, int n_rows, int n_cols) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)n_rows * n_cols;
  if (i >= total) return;
  const int row = (int)(i / n_cols), col = (int)(i - (size_t)row * n_cols);
  float v = slabs[i];
  for (int s = 1; s < n_slabs; ++s) v += slabs[(size_t)s * total + i];
  if (rowscale) v = __fmul_rn(v, rowscale[row]);
  float b = bias ? bias[col] : 0.f;
  if (bias_rowscale) b = __fmul_rn(b, bias_rowscale[row]);
  Y[(size_t)row * ldy + col] = act_apply(__fadd_rn(v, b), act);
}

struct BigPlan { bool use; int slices, k_per_slice; size_t ws; };
BigPlan plan_big(int n_rows, int n_cols, int k) {
  BigPlan p{false, 1, k, 0};
  if (n_rows < 512 || k < 512 || n_cols < 96) return p;
  p.use = true;
  const int blocks = ceil_div(n_rows, G2_BM) * ceil_div(n_cols, G2_BN);
  int s = ceil_div(512, blocks);
  const int max_s = k / 256;                        // >= 8 k-steps per slice
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  p.k_per_slice = ceil_div(ceil_div(k, s), 32) * 32;
  p.slices = ceil_div(k, p.k_per_slice);
  p.ws = p.slices > 1 ? (size_t)p.slices * n_rows * n_cols * sizeof(float) : 0;
  return p;
}

// ------------------------------------------------------------------ EdgeConv aggregation
// One wave per target node; lanes own channels h, h+64, ... (Q rows are read coalesced).
// mode: 0 = sum, 1 = mean (sum / max(count,1)), 2 = max (empty -> 0).
__global__ __launch_bounds__(256) void edgeconv_aggregate_kernel(
    const float* __restrict__ PQ, int ldpq, int H, const int* __restrict__ rowptr, const int64_t* __restrict__ src,
    int fixed_k, int n, int mode, float* __restrict__ S, int lds_) {
  const int i = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const int e0 = rowptr ? rowptr[i] : i * fixed_k, e1 = rowptr ? rowptr[i + 1] : (i + 1) * fixed_k;
  for (int h = lane; h < H; h += 64) {
    const float p = PQ[(size_t)i * ldpq + h];
    float acc = mode == 2 ? -__builtin_inff() : 0.f;
    for (int e = e0; e < e1; ++e) {
      const int j = (int)src[e];
      const float v = fast_tanh(__fadd_rn(p, PQ[(size_t)j * ldpq + H + h]));
      acc = mode == 2 ? fmaxf(acc, v) : acc + v;
    }
    if (mode == 1) acc = acc / (float)max(e1 - e0, 1);
    if (mode == 2 && e1 == e0) acc = 0.f;
    S[(size_t)i * lds_ + h] = acc;
  }
}

// EdgeConv with a NON-linear aggregation (max) cannot hoist the second Linear out of the edge sum, so
// the messages are materialised: m_e = tanh(P_i + Q_j) per edge (rows grouped by target), the second
// Linear runs over the E rows (nbd_linear_f32), and the rows of each target are reduced.
__global__ __launch_bounds__(256) void edge_messages_kernel(const float* __restrict__ PQ, int ldpq, int H,
                                                            const int64_t* __restrict__ src,
                                                            const int64_t* __restrict__ tgt, int64_t n_edges,
                                                            float* __restrict__ M, int ldm) {
  const int64_t e = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (e >= n_edges) return;
  const int lane = threadIdx.x & 63;
  const int64_t i = tgt[e], j = src[e];
  for (int h = lane; h < H; h += 64)
    M[(size_t)e * ldm + h] = tanhf(__fadd_rn(PQ[(size_t)i * ldpq + h], PQ[(size_t)j * ldpq + H + h]));
}

// out[i] = reduce over rows rowptr[i] .. rowptr[i+1] of M (0 = sum, 1 = mean, 2 = max; empty -> 0; 3 = product in row
// order, empty -> 1: torch_scatter's scatter_mul starts from ones)
__global__ __launch_bounds__(256) void segment_reduce_kernel(const float* __restrict__ M, int ldm, int H,
                                                             const int* __restrict__ rowptr, int n, int mode,
                                                             float* __restrict__ out, int ldo) {
  const int i = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const int e0 = rowptr[i], e1 = rowptr[i + 1];
  for (int h = lane; h < H; h += 64) {
    float acc = mode == 2 ? -__builtin_inff() : mode == 3 ? 1.f : 0.f;
    for (int e = e0; e < e1; ++e) {
      const float v = M[(size_t)e * ldm + h];
      acc = mode == 2 ? fmaxf(acc, v) : mode == 3 ? __fmul_rn(acc, v) : acc + v;
    }
    if (mode == 1) acc = acc / (float)max(e1 - e0, 1);
    if (e1 == e0) acc = mode == 3 ? 1.f : 0.f;
    out[(size_t)i * ldo + h] = acc;
  }
}

// ------------------------------------------------------------------ LayerNorm + decoder MLP in one launch
// out = MLP(LayerNorm(x)) for the decoders of gnn.py:105-114,146-148 / contconv.py:206-216,233-234: LayerNorm over
// C <= 256 channels, up to two hidden Linear + tanh layers of <= 64 outputs, a last Linear of <= 8 outputs; optionally
// the caller's half-kick v += c * out in the epilogue (Trainer.step, trainer.py:225-226). At the published ContinuousConv
// shape (N = 16 384, 256 -> 64 -> 32 -> 3) the four separate launches (LayerNorm 10.5 us, Linear 19 + 9 + 9 us) sat at
// the END of the rollout step's dependency chain with the chip idle but for them; here a wave takes a row through the
// whole decoder: the row in registers (lane = channel, 4 per lane), the hidden layers' matrices in LDS as [in][out]
// (staged once per persistent block, lane = output: conflict-free reads), each input broadcast out of its lane with
// v_readlane, the last layer's few dot products as wave reductions. w[i] for i < n_layers - 1 is the TRANSPOSED weight
// ([in][out] contiguous: the caller keeps it with its other derived weights); w[n_layers - 1] is out x in as torch holds it.
constexpr int kHeadMaxOut = 8;
struct LnHeadArgs {
  const float* x; int ldx, c; const float* gamma; const float* beta; float eps;
  int n_layers; const float* w[3]; const float* b[3]; int dims[4];
  float* out; int ldout; float* kick_vel; float kick_c; int n;
};
__global__ __launch_bounds__(256) void ln_mlp_head_kernel(const LnHeadArgs a) {
  extern __shared__ float lds_w[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int C = a.c, NL = a.n_layers;
  const int d1 = a.dims[1], d2 = a.dims[2];
  // the hidden layers' matrices as [in (padded to a multiple of 8 with zero rows)][out]: the mat-vec loops run in
  // unrolled batches of 8 inputs (eight LDS reads in flight per wait)
  const int kp1 = (C + 7) & ~7, kp2 = (d1 + 7) & ~7;
  float* w1t = lds_w;                                      // [kp1][d1]     (NL >= 2)
  float* w2t = lds_w + (NL >= 2 ? kp1 * d1 : 0);            // [kp2][d2]     (NL == 3)
  if (NL >= 2) for (int i = threadIdx.x; i < kp1 * d1; i += 256) w1t[i] = i < C * d1 ? a.w[0][i] : 0.f;
  if (NL == 3) for (int i = threadIdx.x; i < kp2 * d2; i += 256) w2t[i] = i < d1 * d2 ? a.w[1][i] : 0.f;
  __syncthreads();
  const int dl = a.dims[NL - 1], dout = a.dims[NL];         // the last Linear: dl -> dout
  const float* wl = a.w[NL - 1];
  const float* bl = a.b[NL - 1];
  for (int row = blockIdx.x * 4 + wave; row < a.n; row += gridDim.x * 4) {
    const float* x = a.x + (size_t)row * a.ldx;
    // ---- LayerNorm, as layernorm_kernel computes it (same operations, same order)
    float xv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int c = r * 64 + lane; const float t = x[min(c, C - 1)]; xv[r] = c < C ? t : 0.f; }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) s += xv[r];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)C;
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float d = xv[r] - mean; v += (r * 64 + lane < C) ? d * d : 0.f; }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const float rstd = 1.0f / sqrtf(v / (float)C + a.eps);
    float z[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = r * 64 + lane;
      float y = (xv[r] - mean) * rstd;
      if (a.gamma) y = y * a.gamma[min(c, C - 1)];
      if (a.beta) y = y + a.beta[min(c, C - 1)];
      z[r] = c < C ? y : 0.f;
    }
    // ---- hidden layers: lane = output channel
    float h = 0.f;                                         // the last hidden activation (lane < dl), when NL >= 2
    if (NL >= 2) {
      float acc = 0.f;
      const int lo = min(lane, d1 - 1);
      for (int k0 = 0; k0 < kp1; k0 += 8) {                // uniform
        const int r = k0 >> 6, l0 = k0 & 63;
        const float zr = r == 0 ? z[0] : r == 1 ? z[1] : r == 2 ? z[2] : z[3];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float zk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zr), l0 + u));
          acc = __builtin_fmaf(zk, w1t[(k0 + u) * d1 + lo], acc);
        }
      }
      h = lane < d1 ? tanhf(acc + (a.b[0] ? a.b[0][lo] : 0.f)) : 0.f;
      if (NL == 3) {
        float acc2 = 0.f;
        const int lo2 = min(lane, d2 - 1);
        for (int k0 = 0; k0 < kp2; k0 += 8) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float hk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, h), k0 + u));
            acc2 = __builtin_fmaf(hk, w2t[(k0 + u) * d2 + lo2], acc2);
          }
        }
        h = lane < d2 ? tanhf(acc2 + (a.b[1] ? a.b[1][lo2] : 0.f)) : 0.f;
      }
    }
    // ---- last Linear: dout <= 8 dot products over dl inputs, reduced across the wave together
    float part[kHeadMaxOut];
#pragma unroll
    for (int j = 0; j < kHeadMaxOut; ++j) {
      part[j] = 0.f;
      if (j < dout) {                                      // uniform
        if (NL >= 2) part[j] = lane < dl ? h * wl[(size_t)j * dl + lane] : 0.f;
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) { const int c = r * 64 + lane; if (c < C) part[j] = __builtin_fmaf(z[r], wl[(size_t)j * C + c], part[j]); }
        }
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int j = 0; j < kHeadMaxOut; ++j)
        if (j < dout) part[j] += __shfl_xor(part[j], off);
    }
#pragma unroll
    for (int j = 0; j < kHeadMaxOut; ++j) {
      if (j < dout && lane == 0) {
        const float o = part[j] + (bl ? bl[j] : 0.f);
        a.out[(size_t)row * a.ldout + j] = o;
        if (a.kick_vel) {                                  // v += c * a, rounded as the separate kick kernel rounds it
          float* vv = a.kick_vel + (size_t)row * dout + j;
          *vv = __fadd_rn(*vv, __fmul_rn(a.kick_c, o));
        }
      }
    }
  }
}

// The same decoder with hidden layers on the matrix pipe: a wave takes 16 rows at a time. The wave-per-row form above
// (kept for a decoder that is a single small Linear) ran the published 256 -> 64 -> 32 -> 3 decoder in ~110 us at
// N = 16 384 -- 256 dependent LDS-latency-bound mat-vec steps per row at 8 waves per CU -- against 48 us for the four
// launches it replaced. Here the normalised rows never leave registers: lane (m = lane & 15, q = lane >> 4) loads, as 16
// float4, the channels 16 i + 4 q + e of row m -- which IS an A-operand layout of v_mfma_f32_16x16x4_f32 when k-step
// t = 4 i + e is declared to contract channel 16 i + 4 q + e (the B operand reads the matching row of W^T from LDS) --
// so LayerNorm is 64 values per lane plus two cross-lane adds, and the first Linear is 64 k-steps x <= 4 column blocks.
// gamma / beta are folded into the first Linear by the caller (W1 . diag(gamma), b1 + W1 beta: exact algebra), the
// hidden activations go through a wave-private LDS tile to change from the C/D layout to the A layout, the last
// (<= 8 outputs) Linear is a dot product reduced over the 16 lanes that share a row. tanh is the exp2 / rcp form the
// EdgeConv kernels use (abs error <= ~2e-7): libm's tanhf, inlined 24 times per lane with its magnitude branches, was
// two thirds of the instruction stream (24 000 lines of ISA, 330 branches).
constexpr int kHeadHP = 68;          // row stride of the wave-private activation tile (floats)
struct LnHead2Args {
  const float* x; int ldx, c; float eps;
  int n_layers;                      // 2 or 3
  const float* w1t; const float* b1; // [c][d1] (gamma folded in), [d1] (beta folded in)
  const float* w2t; const float* b2; // [d1][d2], [d2]            (n_layers == 3)
  const float* wl; const float* bl;  // [dout][dl] as torch holds it, [dout]
  int d1, d2, dout;
  float* out; int ldout; float* kick_vel; float kick_c; int n;
};
// CB2 = 16-column blocks of the second hidden layer (0: no second hidden layer). The first hidden layer always runs its
// four column blocks (its matrix is zero-padded to 64 columns in LDS): with run-time block counts every MFMA sat behind
// its own uniform branch (330 of them), which also fenced the LDS reads -- one exposed LDS round trip per MFMA, 55 us.
template <int CB2>
__global__ __launch_bounds__(256) void ln_mlp_head_mfma_kernel(const LnHead2Args a) {
  extern __shared__ float lds_w[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int C = a.c, d1 = a.d1, d2 = a.d2, dout = a.dout;
  constexpr int d1p = 64, d2p = CB2 * 16;
  const int kp1 = (C + 15) & ~15;
  // LDS row strides: a B-operand read touches, per k-step, the rows r, r + 4, r + 8, r + 12 (first Linear: channel
  // 16 i + 4 q + e) or r .. r + 3 (second: 4 t + q) at 16 consecutive columns -- s1 = 4 (mod 16) and s2 = 16 (mod 64)
  // spread those four rows over the 64 banks (with the natural strides 64 / 32 every read was a 4- / 2-way conflict)
  constexpr int s1 = d1p + 4, s2 = d2p + 16;
  float* w1t = lds_w;                                       // [kp1][s1], zero padded
  float* w2t = w1t + kp1 * s1;                              // [d1p][s2]
  float* hb = w2t + d1p * s2 + wave * 16 * kHeadHP;         // this wave's 16 x 64 activation tile
  // staging: many loads in flight per thread (a plain one-load-per-iteration loop made the 80 KB of the published decoder
  // 80 dependent L2 round trips per thread: 60 of the first version's 74 us)
  auto stage = [&](float* dst, int lds_stride, const float* src, int rows, int cols, int rowsp, int colsp) {
    if ((cols & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
      const int qpr = colsp / 4, quadsp = rowsp * qpr;                      // float4 per padded row
      for (int i0 = threadIdx.x; i0 < quadsp; i0 += 256 * 8) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + 256 * u, k = i / qpr, o = (i - k * qpr) * 4;
          const f4 t = *reinterpret_cast<const f4*>(src + (size_t)min(k, rows - 1) * cols + min(o, cols - 4));
          v[u] = (i < quadsp && k < rows && o < cols) ? t : f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + 256 * u, k = i / qpr, o = (i - k * qpr) * 4;
          if (i < quadsp) *reinterpret_cast<f4*>(dst + k * lds_stride + o) = v[u];
        }
      }
    } else {
      for (int i0 = threadIdx.x; i0 < rowsp * colsp; i0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + 256 * u, k = i / colsp, o = i - k * colsp;
          const float t = src[(size_t)min(k, rows - 1) * cols + min(o, cols - 1)];
          v[u] = (i < rowsp * colsp && k < rows && o < cols) ? t : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + 256 * u, k = i / colsp, o = i - k * colsp;
          if (i < rowsp * colsp) dst[k * lds_stride + o] = v[u];
        }
      }
    }
  };
  stage(w1t, s1, a.w1t, C, d1, kp1, d1p);
  if (CB2 > 0) stage(w2t, s2, a.w2t, d1, d2, d1p, d2p);
  __syncthreads();
  const int m = lane & 15, q = lane >> 4;
  const int dl = CB2 ? d2 : d1;                                     // the last Linear's input
  constexpr int dlp = CB2 ? d2p : d1p;
  // this lane's entries of the last Linear: column cb * 16 + m of every output row
  float wl[kHeadMaxOut][4], bl[kHeadMaxOut];
#pragma unroll
  for (int j = 0; j < kHeadMaxOut; ++j) {
    bl[j] = (j < dout && a.bl) ? a.bl[j] : 0.f;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int col = cb * 16 + m;
      wl[j][cb] = (j < dout && col < dl) ? a.wl[(size_t)j * dl + col] : 0.f;
    }
  }
  float b1v[4], b2v[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int col = cb * 16 + m;
    b1v[cb] = (col < d1 && a.b1) ? a.b1[col] : 0.f;
    b2v[cb] = (CB2 > 0 && col < d2 && a.b2) ? a.b2[col] : 0.f;
  }
  const bool vec4 = (C & 3) == 0 && (a.ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
  const int groups = (a.n + 15) >> 4;
  for (int g = blockIdx.x * 4 + wave; g < groups; g += gridDim.x * 4) {
    const int row0 = g * 16;
    const float* xr = a.x + (size_t)min(row0 + m, a.n - 1) * a.ldx;
    // ---- the row's channels 16 i + 4 q + e, LayerNorm without the affine part
    f4 zq[16];
    if (vec4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {       // unconditional loads at a clamped channel, masked afterwards: a predicate around
        const int k = 16 * i + 4 * q;        // the load makes hipcc branch and wait per load (16 dependent round trips)
        const f4 t = *reinterpret_cast<const f4*>(xr + min(k, C - 4));
        zq[i] = k < C ? t : f4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int k = 16 * i + 4 * q + e; const float t = xr[min(k, C - 1)]; zq[i][e] = k < C ? t : 0.f; }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += (zq[i][0] + zq[i][1]) + (zq[i][2] + zq[i][3]);
    s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
    const float mean = s / (float)C;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = zq[i][e] - mean; v += (16 * i + 4 * q + e < C) ? d * d : 0.f; }
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    const float rstd = 1.0f / sqrtf(v / (float)C + a.eps);
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) zq[i][e] = (16 * i + 4 * q + e < C) ? (zq[i][e] - mean) * rstd : 0.f;
    // ---- first Linear: k-step t = 4 i + e contracts channel 16 i + 4 q + e
    f4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (16 * i < kp1) {                                  // uniform: one branch per 16 MFMAs
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* wrow = w1t + (16 * i + 4 * q + e) * s1 + m;
#pragma unroll
          for (int cb = 0; cb < 4; ++cb)
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(zq[i][e], wrow[cb * 16], acc[cb], 0, 0, 0);
        }
      }
    }
    // C/D layout: lane holds rows 4 q + v of column cb * 16 + m
    float hl[4][4];                                        // [cb][v]: the last hidden activation, in the C/D layout
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int vv = 0; vv < 4; ++vv) hl[cb][vv] = (cb * 16 + m < d1) ? fast_tanh(acc[cb][vv] + b1v[cb]) : 0.f;
    if (CB2 > 0) {
      // ---- second Linear: the activations change layout through the wave's LDS tile
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) hb[(4 * q + vv) * kHeadHP + cb * 16 + m] = hl[cb][vv];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      f4 acc2[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc2[cb] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < d1p / 4; ++t) {
        const float av = hb[m * kHeadHP + 4 * t + q];
        const float* wrow = w2t + (4 * t + q) * s2 + m;
#pragma unroll
        for (int cb = 0; cb < (CB2 ? CB2 : 1); ++cb)
          acc2[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wrow[cb * 16], acc2[cb], 0, 0, 0);
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) hl[cb][vv] = (cb * 16 + m < d2) ? fast_tanh(acc2[cb][vv] + b2v[cb]) : 0.f;
      __builtin_amdgcn_wave_barrier();                     // the tile is rewritten by the next group
    }
    // ---- last Linear: out[row 4 q + v][j] = sum over the row's dl activations, spread over the 16 lanes of its q
#pragma unroll
    for (int j = 0; j < kHeadMaxOut; ++j) {
      if (j >= dout) break;                                // uniform
#pragma unroll
      for (int vv = 0; vv < 4; ++vv) {
        float p = 0.f;
#pragma unroll
        for (int cb = 0; cb < dlp / 16; ++cb) p = __builtin_fmaf(hl[cb][vv], wl[j][cb], p);
        p += __shfl_xor(p, 1); p += __shfl_xor(p, 2); p += __shfl_xor(p, 4); p += __shfl_xor(p, 8);
        const int row = row0 + 4 * q + vv;
        if (m == 0 && row < a.n) {
          const float o = p + bl[j];
          a.out[(size_t)row * a.ldout + j] = o;
          if (a.kick_vel) {                                // v += c * a, rounded as the separate kick kernel rounds it
            float* kv = a.kick_vel + (size_t)row * dout + j;
            *kv = __fadd_rn(*kv, __fmul_rn(a.kick_c, o));
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------ LayerNorm (one wave per row)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ X, int ldx, int C,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, float* __restrict__ Y, int ldy, int n) {
  const int i = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const float* x = X + (size_t)i * ldx;
  if (C <= 256) {
    // the row in registers: four loads issued together (clamped index, masked value) instead of three passes of
    // dependent loads over the same row (13.6 -> see tools/ubench_mlp.py); same sums in the same order
    float xv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int c = r * 64 + lane; const float t = x[min(c, C - 1)]; xv[r] = c < C ? t : 0.f; }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) s += xv[r];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)C;
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float d = xv[r] - mean; v += (r * 64 + lane < C) ? d * d : 0.f; }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const float rstd = 1.0f / sqrtf(v / (float)C + eps);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = r * 64 + lane;
      if (c < C) {
        float y = (xv[r] - mean) * rstd;
        if (gamma) y = y * gamma[c];
        if (beta) y = y + beta[c];
        Y[(size_t)i * ldy + c] = y;
      }
    }
    return;
  }
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += x[c];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = x[c] - mean; v += d * d; }
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const float rstd = 1.0f / sqrtf(v / (float)C + eps);
  for (int c = lane; c < C; c += 64) {
    float y = (x[c] - mean) * rstd;
    if (gamma) y = y * gamma[c];
    if (beta) y = y + beta[c];
    Y[(size_t)i * ldy + c] = y;
  }
}

// ------------------------------------------------------------------ ContinuousConv cell binning
// grid = (nodes, channel groups of 64); block = one wave, lane = channel. The A row of a node is
// built one z-slab (D*D cells x 64 channels, <= 16 KiB of LDS) at a time so that many waves fit a
// CU: for each slab the wave walks the node's incoming edges (CSR by aggregation target =
// edge_index[0], contconv.py:82,95) in chunks of 64 -- lane e evaluates the geometry of edge e
// (window, ball_to_cube, trilinear weights) and parks it in LDS -- then every lane (= channel) adds
// w * feat[c][lane] into the <= 4 touched cells of the slab, and the slab is streamed out (256-B
// coalesced stores) and re-zeroed.
// Trilinear weights follow F.grid_sample(align_corners=True) with coordinate component 0 indexing
// filter axis 2 (x fastest) and component 2 indexing axis 0: cell = (z*D + y)*D + x  (contconv.py:62-75).
struct EdgeGeo { int c, ix, iy, iz; float tx, ty, tz, window; };
// edge geometries kept in LDS per node (in-degree above this: recomputed per slab). 64 = one chunk: a cache
// of 256 entries (8 KiB more LDS per wave, 6 instead of 8 waves per CU) cost 5 % of the rollout step.
constexpr int kGeoCache = 64;

__device__ __forceinline__ EdgeGeo edge_geometry(const float* __restrict__ pos, int c, float xn, float yn, float zn,
                                                 float r2max, float half) {
  EdgeGeo g;
  g.c = c;
  // r = positions[col] - positions[row] (contconv.py:84): centre minus this node
  const float rx = pos[3 * c] - xn, ry = pos[3 * c + 1] - yn, rz = pos[3 * c + 2] - zn;
  const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(rx, rx), __fmul_rn(ry, ry)), __fmul_rn(rz, rz));
  const float qq = 1.0f - d2 / r2max;
  g.window = (d2 < r2max) ? qq * qq * qq : 0.f;                  // contconv.py:85-87
  const float nrm = sqrtf(d2);
  const float sc = tanhf(nrm) / (nrm + 1e-8f);                   // ball_to_cube (contconv.py:30-33)
  const float gx = (rx * sc + 1.0f) * half, gy = (ry * sc + 1.0f) * half, gz = (rz * sc + 1.0f) * half;
  const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
  g.ix = (int)fx; g.iy = (int)fy; g.iz = (int)fz;
  g.tx = gx - fx; g.ty = gy - fy; g.tz = gz - fz;
  return g;
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Variants measured at N = 16 384, D = 6 / 4 (tools/bench_surrogates.py + rocprofv3): whole 216-cell image
// per wave (55 KB LDS, 2 waves/CU) 2390 / 490 us; z-slab images with the geometry recomputed per slab
// 586 / 368 us; geometry cached once per node, only the edges that touch the slab visited, two feature
// rows in flight 552 / 310 us; staging the chunk's feature rows in LDS (27 KB per wave) 850 / 420 us --
// the kernel lives on the number of waves in flight and on instructions per byte, hence CPL channels
// per lane (one wave covers 64*CPL channels with 8-byte LDS/global accesses).
template <int CPL>
__global__ __launch_bounds__(64) void contconv_bin_kernel(
    const float* __restrict__ pos, const float* __restrict__ feat, int ldf, int I, const int* __restrict__ rowptr,
    const int* __restrict__ centres, int node_begin, int D, float r2max, const int* __restrict__ cell_map,
    int cells_out, float* __restrict__ A) {
  typedef float vec __attribute__((ext_vector_type(CPL)));
  extern __shared__ float img_raw[];        // [D*D][64] vec slab image, then kGeoCache EdgeGeo records
  vec* img = reinterpret_cast<vec*>(img_raw);
  const int node = node_begin + blockIdx.x, cg = blockIdx.y, lane = threadIdx.x;
  const int ch = (cg * 64 + lane) * CPL;                 // first of this lane's CPL consecutive channels
  const int slab_cells = D * D;
  EdgeGeo* geo = reinterpret_cast<EdgeGeo*>(img + slab_cells * 64);
  const float xn = pos[3 * node], yn = pos[3 * node + 1], zn = pos[3 * node + 2];
  const float half = (float)(D - 1) / 2.0f;
  const int e0 = rowptr[node], e1 = rowptr[node + 1];
  const bool cached = (e1 - e0) <= kGeoCache;
  const bool live = ch + CPL <= I;                       // host guarantees I % CPL == 0
  const vec zero = {};
  for (int c = 0; c < slab_cells; ++c) img[c * 64 + lane] = zero;
  if (cached) {       // every edge's geometry once (lane = edge), reused by all D slabs
    for (int e = e0 + lane; e < e1; e += 64) geo[e - e0] = edge_geometry(pos, centres[e], xn, yn, zn, r2max, half);
    wave_lds_sync();
  }
  for (int z = 0; z < D; ++z) {
    for (int eb = e0; eb < e1; eb += 64) {
      const int cnt = min(64, e1 - eb);
      const int gbase = cached ? eb - e0 : 0;
      if (!cached) {
        if (lane < cnt) geo[lane] = edge_geometry(pos, centres[eb + lane], xn, yn, zn, r2max, half);
        wave_lds_sync();
      }
      // edges of this chunk that touch slab z (iz == z or iz + 1 == z) and carry weight
      bool act = false;
      if (lane < cnt) {
        const int az = z - geo[gbase + lane].iz;
        act = (az == 0 || az == 1) && geo[gbase + lane].window != 0.f;
      }
      unsigned long long m = __ballot(act);
      while (m) {                                   // two active edges per trip: their feature loads overlap
        const int ea = __builtin_ctzll(m); m &= m - 1;
        const bool two = m != 0;
        const int eb2 = two ? __builtin_ctzll(m) : ea;
        if (two) m &= m - 1;
        const EdgeGeo g0 = geo[gbase + ea], g1 = geo[gbase + eb2];     // wave-uniform LDS broadcasts
        vec f0 = zero, f1 = zero;
        if (live) {
          f0 = *reinterpret_cast<const vec*>(feat + (size_t)g0.c * ldf + ch);
          f1 = *reinterpret_cast<const vec*>(feat + (size_t)g1.c * ldf + ch);
        }
        if (!two) f1 = zero;
        f0 *= ((z - g0.iz) ? g0.tz : 1.0f - g0.tz) * g0.window;
        f1 *= ((z - g1.iz) ? g1.tz : 1.0f - g1.tz) * g1.window;
#pragma unroll
        for (int corner = 0; corner < 4; ++corner) {
          const int ax = corner & 1, ay = corner >> 1;
          const int cx = g0.ix + ax, cy = g0.iy + ay;
          if (cx < 0 || cx >= D || cy < 0 || cy >= D) continue;        // zero padding of grid_sample
          img[(cy * D + cx) * 64 + lane] += ((ax ? g0.tx : 1.0f - g0.tx) * (ay ? g0.ty : 1.0f - g0.ty)) * f0;
        }
        if (two) {
#pragma unroll
          for (int corner = 0; corner < 4; ++corner) {
            const int ax = corner & 1, ay = corner >> 1;
            const int cx = g1.ix + ax, cy = g1.iy + ay;
            if (cx < 0 || cx >= D || cy < 0 || cy >= D) continue;
            img[(cy * D + cx) * 64 + lane] += ((ax ? g1.tx : 1.0f - g1.tx) * (ay ? g1.ty : 1.0f - g1.ty)) * f1;
          }
        }
      }
      if (!cached) __builtin_amdgcn_wave_barrier();
    }
    if (live) {
      // cell_map compacts the row to the grid points a sample can reach at all (ball_to_cube keeps every
      // sample inside |mapped| < tanh(R)): unreachable cells are structurally zero and are not stored
      float* dst = A + (size_t)blockIdx.x * cells_out * I + ch;
      for (int c = 0; c < slab_cells; ++c) {
        const int m = cell_map ? cell_map[z * slab_cells + c] : z * slab_cells + c;
        if (m >= 0) *reinterpret_cast<vec*>(dst + (size_t)m * I) = img[c * 64 + lane];
        img[c * 64 + lane] = zero;
      }
    } else {
      for (int c = 0; c < slab_cells; ++c) img[c * 64 + lane] = zero;
    }
  }
}

// rowscale[n] = 1 / max(indeg, 1) for mean aggregation, 1 for sum (scatter, contconv.py:95-97)
__global__ __launch_bounds__(256) void degree_scale_kernel(const int* __restrict__ rowptr, int n, int mode,
                                                           float* __restrict__ scale) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int d = rowptr[i + 1] - rowptr[i];
  // mode 0: 1/max(d,1) (mean);  1: d (bias multiplier of a sum aggregation);  2: d > 0 ? 1 : 0
  scale[i] = mode == 0 ? 1.0f / (float)max(d, 1) : (mode == 1 ? (float)d : (d > 0 ? 1.0f : 0.0f));
}

template <bool VEC>
int launch_linear(const float* X, int ldx, const float* W, int ldw, const float* b, const float* rs,
                  const float* brs, int act, float* Y, int ldy, int n, int m, int K, hipStream_t st) {
  if (m > 64 && ceil_div(n, 64) * ceil_div(m, 128) < 512) {   // 32 x 128 (waves 1 x 4): twice the workgroups when rows are few
    dim3 grid(ceil_div(n, 32), ceil_div(m, 128));
    linear_kernel<1, 4, 1, VEC><<<grid, 256, 0, st>>>(X, ldx, W, ldw, b, rs, brs, act, Y, ldy, n, m, K);
  } else if (m > 64) {   // 64 x 128 block tile: waves 2 x 2, strips of 32 x 64
    dim3 grid(ceil_div(n, 64), ceil_div(m, 128));
    linear_kernel<2, 2, 2, VEC><<<grid, 256, 0, st>>>(X, ldx, W, ldw, b, rs, brs, act, Y, ldy, n, m, K);
  } else if (ceil_div(n, 128) < 512) {   // 64 x 64 (waves 2 x 2, strips of 32 x 32): a skinny product on few
    dim3 grid(ceil_div(n, 64), ceil_div(m, 64));    // rows leaves half the CUs idle with 128-row tiles (n = 16 384: 128 workgroups)
    linear_kernel<2, 2, 1, VEC><<<grid, 256, 0, st>>>(X, ldx, W, ldw, b, rs, brs, act, Y, ldy, n, m, K);
  } else if (m > 32) {   // 128 x 64: waves 4 x 1, strips of 32 x 64
    dim3 grid(ceil_div(n, 128), ceil_div(m, 64));
    linear_kernel<4, 1, 2, VEC><<<grid, 256, 0, st>>>(X, ldx, W, ldw, b, rs, brs, act, Y, ldy, n, m, K);
  } else {               // 128 x 32
    dim3 grid(ceil_div(n, 128), ceil_div(m, 32));
    linear_kernel<4, 1, 1, VEC><<<grid, 256, 0, st>>>(X, ldx, W, ldw, b, rs, brs, act, Y, ldy, n, m, K);
  }
  return status();
}

}  // namespace

// ContinuousConv's two public helpers (contconv.py:30-33, 53-78), callable on the drop-in layer as in the reference.
// ball_to_cube: r / (|r| + 1e-8) * tanh |r| with torch's own order of operations (norm -> divide -> multiply).
__global__ __launch_bounds__(256) void ball_to_cube_kernel(const float* __restrict__ r, int n, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = r[3 * i], y = r[3 * i + 1], z = r[3 * i + 2];
  const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
  const float den = nrm + 1e-8f, th = tanhf(nrm);
  out[3 * i] = __fmul_rn(x / den, th); out[3 * i + 1] = __fmul_rn(y / den, th); out[3 * i + 2] = __fmul_rn(z / den, th);
}

// trilinear_interpolate: F.grid_sample(filters as (1, I O, D, D, D), mode = "bilinear", align_corners = True, zero
// padding) at coords in grid units -- coordinate component 0 indexes the LAST filter axis (filters[z][y][x]).
// One thread per (sample, 4 consecutive (i, o) entries): the eight corners' weights from grid_sample's own arithmetic
// (normalise to [-1, 1], un-normalise, floor, products of the three distances).
__global__ __launch_bounds__(256) void trilinear_interpolate_kernel(const float* __restrict__ filt, int D, int io,
                                                                    const float* __restrict__ coords, int n, float* __restrict__ out) {
  const int q4 = (io + 3) >> 2;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n * q4) return;
  const int s = (int)(idx / q4), c0 = (int)(idx - (size_t)s * q4) * 4;
  float g[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float nc = (coords[3 * s + a] / (float)(D - 1)) * 2.0f - 1.0f;     // contconv.py:62
    g[a] = ((nc + 1.0f) / 2.0f) * (float)(D - 1);                            // grid_sampler_unnormalize, align_corners
  }
  const float fx = floorf(g[0]), fy = floorf(g[1]), fz = floorf(g[2]);
  const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
  const float tx = g[0] - fx, ty = g[1] - fy, tz = g[2] - fz;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int corner = 0; corner < 8; ++corner) {
    const int ax = corner & 1, ay = (corner >> 1) & 1, az = corner >> 2;
    const int cx = ix + ax, cy = iy + ay, cz = iz + az;
    if ((unsigned)cx >= (unsigned)D || (unsigned)cy >= (unsigned)D || (unsigned)cz >= (unsigned)D) continue;   // zero padding
    const float w = (ax ? tx : 1.0f - tx) * (ay ? ty : 1.0f - ty) * (az ? tz : 1.0f - tz);
    const float* f = filt + ((size_t)(cz * D + cy) * D + cx) * io + c0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + j < io) acc[j] = __builtin_fmaf(f[j], w, acc[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (c0 + j < io) out[(size_t)s * io + c0 + j] = acc[j];
}

extern "C" {

size_t nbd_linear_workspace_bytes(int n_rows, int n_cols, int k) {
  if (n_rows <= 0 || n_cols <= 0 || k <= 0) return 0;
  return plan_big(n_rows, n_cols, k).ws;
}

int nbd_linear_f32(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* rowscale,
                   const float* bias_rowscale, int act, float* y, int ldy, int n_rows, int n_cols, int k,
                   void* workspace, size_t workspace_bytes, nbd_stream_t stream) {
  if (n_rows < 0 || n_cols < 0 || k < 0 || act < 0 || act > 1) return NBD_E_BADARG;
  if (n_rows == 0 || n_cols == 0) return 0;
  if (!x || !w || !y || ldx < k || ldw < k || ldy < n_cols) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (k % 4 == 0) && (ldx % 4 == 0) && (ldw % 4 == 0) &&
                   ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) == 0;
  const BigPlan bp = plan_big(n_rows, n_cols, k);
  if (vec && bp.use && (bp.slices == 1 || (workspace && workspace_bytes >= bp.ws))) {
    const size_t shmem = 2 * (G2_BM + G2_BN) * G2_LD * sizeof(float);     // 73 728 B
    // > 64 KiB of dynamic LDS needs the attribute; it is a per-function constant, set once (idempotent,
    // so the unsynchronised flag is benign) and kept out of later calls so they can be graph-captured
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_big_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    dim3 grid(ceil_div(n_rows, G2_BM), ceil_div(n_cols, G2_BN), bp.slices);
    float* slabs = static_cast<float*>(workspace);
    linear_big_kernel<<<grid, 256, shmem, st>>>(x, ldx, w, ldw, bias, rowscale, bias_rowscale, act, y, ldy, n_rows,
                                                n_cols, k, bp.k_per_slice, slabs);
    int rc = status();
    if (rc || bp.slices == 1) return rc;
    const size_t total = (size_t)n_rows * n_cols;
    linear_finish_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(slabs, bp.slices, bias, rowscale, bias_rowscale,
                                                                          act, y, ldy, n_rows, n_cols);
    return status();
  }
  return vec ? launch_linear<true>(x, ldx, w, ldw, bias, rowscale, bias_rowscale, act, y, ldy, n_rows, n_cols, k, st)
             : launch_linear<false>(x, ldx, w, ldw, bias, rowscale, bias_rowscale, act, y, ldy, n_rows, n_cols, k, st);
}

int nbd_edgeconv_aggregate_f32(const float* pq, int ldpq, int h, const int* rowptr, const int64_t* src,
                               int fixed_k, int n, int aggr, float* s, int lds, nbd_stream_t stream) {
  if (n < 0 || h < 0 || aggr < 0 || aggr > 2 || (!rowptr && fixed_k < 0)) return NBD_E_BADARG;
  if (n == 0 || h == 0) return 0;
  if (!pq || !s || ldpq < 2 * h || lds < h) return NBD_E_BADARG;
  if (!src && (rowptr || fixed_k > 0)) return NBD_E_BADARG;
  edgeconv_aggregate_kernel<<<ceil_div(n, 4), 256, 0, (hipStream_t)stream>>>(pq, ldpq, h, rowptr, src, fixed_k, n,
                                                                            aggr, s, lds);
  return status();
}

int nbd_edge_messages_f32(const float* pq, int ldpq, int h, const int64_t* src, const int64_t* tgt, int64_t n_edges,
                          float* m, int ldm, nbd_stream_t stream) {
  if (n_edges < 0 || h <= 0) return NBD_E_BADARG;
  if (n_edges == 0) return 0;
  if (!pq || !src || !tgt || !m || ldpq < 2 * h || ldm < h) return NBD_E_BADARG;
  if (n_edges > 0x7fffffffLL * 4) return NBD_E_UNSUPPORTED;
  edge_messages_kernel<<<(unsigned)((n_edges + 3) / 4), 256, 0, (hipStream_t)stream>>>(pq, ldpq, h, src, tgt, n_edges, m, ldm);
  return status();
}

int nbd_segment_reduce_f32(const float* m, int ldm, int h, const int* rowptr, int n, int mode, float* out, int ldo,
                           nbd_stream_t stream) {
  if (n < 0 || h <= 0 || mode < 0 || mode > 3) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!rowptr || !out || ldo < h || (!m && ldm != 0) || ldm < h) return NBD_E_BADARG;
  segment_reduce_kernel<<<ceil_div(n, 4), 256, 0, (hipStream_t)stream>>>(m, ldm, h, rowptr, n, mode, out, ldo);
  return status();
}

int nbd_layernorm_f32(const float* x, int ldx, int c, const float* gamma, const float* beta, float eps, float* y,
                      int ldy, int n, nbd_stream_t stream) {
  if (n < 0 || c <= 0) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!x || !y || ldx < c || ldy < c) return NBD_E_BADARG;
  layernorm_kernel<<<ceil_div(n, 4), 256, 0, (hipStream_t)stream>>>(x, ldx, c, gamma, beta, eps, y, ldy, n);
  return status();
}

size_t nbd_ln_mlp_head_lds_bytes(int c, int n_layers, const int* dims) {
  if (!dims || n_layers < 1 || n_layers > 3 || c <= 0 || c > 256 || dims[0] != c) return 0;
  for (int i = 1; i < n_layers; ++i) if (dims[i] <= 0 || dims[i] > 64) return 0;
  if (dims[n_layers] <= 0 || dims[n_layers] > kHeadMaxOut) return 0;
  if (n_layers == 1) return 16;
  // the first hidden layer padded to 64 columns, the second to 32 or 64 (ln_mlp_head_mfma_kernel<CB2>)
  const int kp1 = (c + 15) & ~15, d2p = n_layers == 3 ? (dims[2] <= 32 ? 32 : 64) : 0;
  return ((size_t)kp1 * (64 + 4) + (size_t)64 * (d2p + 16) + 4 * 16 * kHeadHP) * sizeof(float);
}

int nbd_ln_mlp_head_f32(const float* x, int ldx, int c, const float* gamma, const float* beta, float eps, int n_layers,
                        const float* const* w, const float* const* b, const int* dims, float* out, int ldout,
                        float* kick_vel, float kick_c, int n, nbd_stream_t stream) {
  if (n < 0 || !dims || !w || !b) return NBD_E_BADARG;
  const size_t lds = nbd_ln_mlp_head_lds_bytes(c, n_layers, dims);
  if (lds == 0) return NBD_E_UNSUPPORTED;
  if (n == 0) return 0;
  if (!x || !out || ldx < c || ldout < dims[n_layers]) return NBD_E_BADARG;
  for (int i = 0; i < n_layers; ++i) if (!w[i]) return NBD_E_BADARG;
  if (n_layers == 1) {                 // LayerNorm + one small Linear: a wave per row, dot products as wave reductions
    LnHeadArgs a;
    a.x = x; a.ldx = ldx; a.c = c; a.gamma = gamma; a.beta = beta; a.eps = eps; a.n_layers = 1;
    for (int i = 0; i < 3; ++i) { a.w[i] = i < 1 ? w[i] : nullptr; a.b[i] = i < 1 ? b[i] : nullptr; }
    for (int i = 0; i < 4; ++i) a.dims[i] = i <= 1 ? dims[i] : 0;
    a.out = out; a.ldout = ldout; a.kick_vel = kick_vel; a.kick_c = kick_c; a.n = n;
    int blocks = ceil_div(n, 4);
    if (blocks > 2048) blocks = 2048;
    ln_mlp_head_kernel<<<blocks, 256, 16, (hipStream_t)stream>>>(a);
    return status();
  }
  // hidden layers on the matrix pipe. gamma / beta must have been folded into the first Linear by the caller
  // (w[0] = (W1 diag(gamma))^T, b[0] = b1 + W1 beta): passing them here as well would apply them twice.
  if (gamma || beta) return NBD_E_BADARG;
  LnHead2Args a;
  a.x = x; a.ldx = ldx; a.c = c; a.eps = eps; a.n_layers = n_layers;
  a.w1t = w[0]; a.b1 = b[0];
  a.w2t = n_layers == 3 ? w[1] : nullptr; a.b2 = n_layers == 3 ? b[1] : nullptr;
  a.wl = w[n_layers - 1]; a.bl = b[n_layers - 1];
  a.d1 = dims[1]; a.d2 = n_layers == 3 ? dims[2] : 0; a.dout = dims[n_layers];
  a.out = out; a.ldout = ldout; a.kick_vel = kick_vel; a.kick_c = kick_c; a.n = n;
  // persistent blocks of 4 waves, 16 rows per wave at a time; every block stages the hidden layers' matrices once
  int blocks = ceil_div(ceil_div(n, 16), 4);
  if (blocks > 256) blocks = 256;
#define NBD_HEAD_LAUNCH(CB2)                                                                                        \
  do {                                                                                                              \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ln_mlp_head_mfma_kernel<CB2>),                 \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);                     \
    if (e != hipSuccess) return (int)e;                                                                             \
    ln_mlp_head_mfma_kernel<CB2><<<blocks, 256, lds, (hipStream_t)stream>>>(a);                                     \
  } while (0)
  if (n_layers == 2) NBD_HEAD_LAUNCH(0);
  else if (dims[2] <= 32) NBD_HEAD_LAUNCH(2);
  else NBD_HEAD_LAUNCH(4);
#undef NBD_HEAD_LAUNCH
  return status();
}

int nbd_contconv_bin_f32(const float* pos, const float* feat, int ldf, int in_channels, const int* rowptr,
                         const int* centres, int node_begin, int n, int filter_resolution, float radius_sq,
                         const int* cell_map, int cells_out, float* a_out, nbd_stream_t stream) {
  if (n < 0 || node_begin < 0 || in_channels <= 0 || filter_resolution < 2) return NBD_E_BADARG;
  const int cells = filter_resolution * filter_resolution * filter_resolution;
  if (!cell_map) cells_out = cells;
  if (cells_out <= 0 || cells_out > cells) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!pos || !feat || !rowptr || !centres || !a_out || ldf < in_channels) return NBD_E_BADARG;
  // two channels per lane (8-byte accesses) when the layout allows it: even I, even ldf, 8-byte aligned bases
  const bool cpl2 = in_channels % 2 == 0 && ldf % 2 == 0 &&
                    ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(a_out)) & 7) == 0;
  const int cpl = cpl2 ? 2 : 1;
  const size_t shmem = (size_t)filter_resolution * filter_resolution * 64 * cpl * sizeof(float) + kGeoCache * sizeof(EdgeGeo);
  if (shmem > 64 * 1024) return NBD_E_UNSUPPORTED;       // D <= 10
  dim3 grid(n, ceil_div(in_channels, 64 * cpl));
  hipStream_t st = (hipStream_t)stream;
  if (cpl2)
    contconv_bin_kernel<2><<<grid, 64, shmem, st>>>(pos, feat, ldf, in_channels, rowptr, centres, node_begin,
                                                    filter_resolution, radius_sq, cell_map, cells_out, a_out);
  else
    contconv_bin_kernel<1><<<grid, 64, shmem, st>>>(pos, feat, ldf, in_channels, rowptr, centres, node_begin,
                                                    filter_resolution, radius_sq, cell_map, cells_out, a_out);
  return status();
}

int nbd_degree_scale_f32(const int* rowptr, int n, int mode, float* scale, nbd_stream_t stream) {
  if (n < 0 || mode < 0 || mode > 2) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!rowptr || !scale) return NBD_E_BADARG;
  degree_scale_kernel<<<ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(rowptr, n, mode, scale);
  return status();
}

int nbd_ball_to_cube_f32(const float* r, int n, float* out, nbd_stream_t stream) {
  if (n < 0) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!r || !out) return NBD_E_BADARG;
  ball_to_cube_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(r, n, out);
  return status();
}

int nbd_trilinear_interpolate_f32(const float* filters, int filter_resolution, int in_channels, int out_channels,
                                  const float* coords, int n, float* out, nbd_stream_t stream) {
  if (n < 0 || filter_resolution < 2 || in_channels <= 0 || out_channels <= 0) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!filters || !coords || !out) return NBD_E_BADARG;
  const int io = in_channels * out_channels;
  const size_t threads = (size_t)n * ((io + 3) / 4);
  trilinear_interpolate_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, (hipStream_t)stream>>>(filters, filter_resolution, io, coords, n, out);
  return status();
}

}  // extern "C"
