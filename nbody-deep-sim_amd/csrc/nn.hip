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

#include "../../include/nbd.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

namespace {

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int status() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : (int)e; }

__device__ __forceinline__ float act_apply(float v, int act) { return act == 1 ? tanhf(v) : v; }

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

  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();   // previous step's fragment reads are done
    if (VEC) {         // 8 lanes x float4 cover one 128-B row of the K-slab
      const int q = tid & 7, r8 = tid >> 3;
      for (int r = r8; r < BM + BN; r += 32) {
        const bool isA = r < BM;
        const int rr = isA ? r : r - BM;
        const int g = (isA ? row0 : col0) + rr;
        const int lim = isA ? n_rows : n_cols;
        const float* src = isA ? X + (size_t)g * ldx : W + (size_t)g * ldw;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        const int k = k0 + 4 * q;
        if (g < lim && k < K) v = *reinterpret_cast<const f4*>(src + k);   // K % 4 == 0 on this path
        float* dst = (isA ? As : Bs) + rr * LD + 4 * q;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
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
      const float v = tanhf(__fadd_rn(p, PQ[(size_t)j * ldpq + H + h]));
      acc = mode == 2 ? fmaxf(acc, v) : acc + v;
    }
    if (mode == 1) acc = acc / (float)max(e1 - e0, 1);
    if (mode == 2 && e1 == e0) acc = 0.f;
    S[(size_t)i * lds_ + h] = acc;
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
// grid = (nodes, channel groups of 64); block = one wave, lane = channel. The wave walks the node's
// incoming edges (CSR by aggregation target = edge_index[0], contconv.py:82,95), evaluates the edge
// geometry once (wave-uniform) and adds w * feat[c][lane] into the 8 touched cells of an LDS image of
// this node's A row (cells x 64 channels), then streams the image out and re-zeroes it.
// Trilinear weights follow F.grid_sample(align_corners=True) with coordinate component 0 indexing
// filter axis 2 (x fastest) and component 2 indexing axis 0: cell = (z*D + y)*D + x  (contconv.py:62-75).
__global__ __launch_bounds__(64) void contconv_bin_kernel(
    const float* __restrict__ pos, const float* __restrict__ feat, int ldf, int I, const int* __restrict__ rowptr,
    const int* __restrict__ centres, int n, int D, float r2max, float* __restrict__ A, int* __restrict__ indeg_out) {
  extern __shared__ float img[];            // [cells][64]
  const int node = blockIdx.x, cg = blockIdx.y, lane = threadIdx.x;
  const int ch = cg * 64 + lane;
  const int cells = D * D * D;
  for (int c = 0; c < cells; ++c) img[c * 64 + lane] = 0.f;
  const float xn = pos[3 * node], yn = pos[3 * node + 1], zn = pos[3 * node + 2];
  const float half = (float)(D - 1) / 2.0f;
  const int e0 = rowptr[node], e1 = rowptr[node + 1];
  for (int e = e0; e < e1; ++e) {
    const int c = centres[e];
    // r = positions[col] - positions[row] (contconv.py:84): centre minus this node
    const float rx = pos[3 * c] - xn, ry = pos[3 * c + 1] - yn, rz = pos[3 * c + 2] - zn;
    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(rx, rx), __fmul_rn(ry, ry)), __fmul_rn(rz, rz));
    if (!(d2 < r2max)) continue;                                   // window = 0 (contconv.py:86-87)
    const float q = 1.0f - d2 / r2max;
    const float window = q * q * q;
    const float nrm = sqrtf(d2);
    const float sc = tanhf(nrm) / (nrm + 1e-8f);                   // ball_to_cube (contconv.py:30-33)
    const float gx = (rx * sc + 1.0f) * half, gy = (ry * sc + 1.0f) * half, gz = (rz * sc + 1.0f) * half;
    const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
    const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    const float tx = gx - fx, ty = gy - fy, tz = gz - fz;
    const float f = (ch < I) ? feat[(size_t)c * ldf + ch] * window : 0.f;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      const int ax = corner & 1, ay = (corner >> 1) & 1, az = corner >> 2;
      const int cx = ix + ax, cy = iy + ay, cz = iz + az;
      if (cx < 0 || cx >= D || cy < 0 || cy >= D || cz < 0 || cz >= D) continue;   // zero padding
      const float w = (ax ? tx : 1.0f - tx) * (ay ? ty : 1.0f - ty) * (az ? tz : 1.0f - tz);
      img[((cz * D + cy) * D + cx) * 64 + lane] += w * f;
    }
  }
  if (ch < I) {
    float* dst = A + (size_t)node * cells * I + ch;
    for (int c = 0; c < cells; ++c) dst[(size_t)c * I] = img[c * 64 + lane];
  }
  if (indeg_out && cg == 0 && lane == 0) indeg_out[node] = e1 - e0;
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
  if (m > 64) {          // 64 x 128 block tile: waves 2 x 2, strips of 32 x 64
    dim3 grid(ceil_div(n, 64), ceil_div(m, 128));
    linear_kernel<2, 2, 2, VEC><<<grid, 256, 0, st>>>(X, ldx, W, ldw, b, rs, brs, act, Y, ldy, n, m, K);
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

extern "C" {

int nbd_linear_f32(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* rowscale,
                   const float* bias_rowscale, int act, float* y, int ldy, int n_rows, int n_cols, int k,
                   nbd_stream_t stream) {
  if (n_rows < 0 || n_cols < 0 || k < 0 || act < 0 || act > 1) return NBD_E_BADARG;
  if (n_rows == 0 || n_cols == 0) return 0;
  if (!x || !w || !y || ldx < k || ldw < k || ldy < n_cols) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (k % 4 == 0) && (ldx % 4 == 0) && (ldw % 4 == 0) &&
                   ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) == 0;
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

int nbd_layernorm_f32(const float* x, int ldx, int c, const float* gamma, const float* beta, float eps, float* y,
                      int ldy, int n, nbd_stream_t stream) {
  if (n < 0 || c <= 0) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!x || !y || ldx < c || ldy < c) return NBD_E_BADARG;
  layernorm_kernel<<<ceil_div(n, 4), 256, 0, (hipStream_t)stream>>>(x, ldx, c, gamma, beta, eps, y, ldy, n);
  return status();
}

int nbd_contconv_bin_f32(const float* pos, const float* feat, int ldf, int in_channels, const int* rowptr,
                         const int* centres, int n, int filter_resolution, float radius_sq, float* a_out,
                         nbd_stream_t stream) {
  if (n < 0 || in_channels <= 0 || filter_resolution < 2) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!pos || !feat || !rowptr || !centres || !a_out || ldf < in_channels) return NBD_E_BADARG;
  const int cells = filter_resolution * filter_resolution * filter_resolution;
  const size_t shmem = (size_t)cells * 64 * sizeof(float);
  if (shmem > 160 * 1024) return NBD_E_UNSUPPORTED;      // D <= 8
  if (shmem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(contconv_bin_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return (int)e;
  }
  dim3 grid(n, ceil_div(in_channels, 64));
  contconv_bin_kernel<<<grid, 64, shmem, (hipStream_t)stream>>>(pos, feat, ldf, in_channels, rowptr, centres, n,
                                                                filter_resolution, radius_sq, a_out, nullptr);
  return status();
}

int nbd_degree_scale_f32(const int* rowptr, int n, int mode, float* scale, nbd_stream_t stream) {
  if (n < 0 || mode < 0 || mode > 2) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!rowptr || !scale) return NBD_E_BADARG;
  degree_scale_kernel<<<ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(rowptr, n, mode, scale);
  return status();
}

}  // extern "C"
