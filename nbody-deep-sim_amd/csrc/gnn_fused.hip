// gnn_fused.hip -- one launch per EdgeConv layer of the GNN surrogate (gnn.py:75-93,130-148), gfx950.
//
// At the reference's sizes (N = 4096, H = 64) every dense block of the GNN is a few microseconds of
// work, so the forward pass is bounded by launches and dependent kernel latencies, not by FLOPs
// (tools/ubench_linear.py: 8-10 us per tiny Linear, 11-16 us per aggregation). This kernel therefore
// keeps one node per wave (lane = channel) from the edge aggregation to the layer's output:
//
//   S_i      = aggr_j tanh(P_i + Q_j)                       (EdgeConv after the per-node factoring)
//   y_i      = W2 S_i + beta_i b2                           (beta = 1 / deg / [deg>0], see nbd.h)
//   then one of
//     NEXT_PQ     pq'_i = Wpq' y_i + bpq'                   (the next layer's [P|Q], gnn.py:140-141)
//     WRITE_X     x'_i  = y_i                               (into a column slice: replaces torch.cat)
//     FINAL_HEAD  out_i = Wh LayerNorm([enc_i || y_i]) + bh (gnn.py:144-148, out_dim <= 8)
//     FINAL_LN    z_i   = LayerNorm([enc_i || y_i])         (an MLP head then runs on z)
//
// P/Q come either from a [N][2H] buffer or, for a first layer with F <= 8 input features, are formed
// on the fly from x (8 FMAs per edge instead of a 256-B row gather). The small matrices (W2^T and the
// epilogue's) are staged once per workgroup in LDS as [k][out] so that lane = out reads are
// conflict-free; the per-node mat-vecs broadcast the k-th input with v_readlane (no LDS, no shuffles).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nbd.h"

namespace {

constexpr int kFMax = 8;        // on-the-fly P/Q: input features, zero padded
constexpr int kMaxR = 2;        // channels per lane: H <= 128
constexpr int kMaxZR = 4;       // concat width per lane: E + H <= 256
constexpr int kMaxOut = 8;
constexpr int kEF = 16;         // edges (neighbour rows) in flight per wave

__device__ __forceinline__ float lane_bcast(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// tanh(x) = 1 - 2 / (2^(2x log2 e) + 1): v_exp_f32 + v_rcp_f32 + 3 VALU instead of libm's ~40-instruction
// tanhf. The E x H tanh evaluations are what bounds the EdgeConv aggregation (8.4 M at N=4096, k=32,
// H=64), so this is the kernel's roofline lever. Absolute error <= ~2e-7 (1-ulp exp2/rcp and one
// cancellation at ulp(1)); saturates to +-1 exactly for large |x|, NaN stays NaN.
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
// the exponential-table form (nbd.h, nbd_gnn_layer_args.epq): 2^(c v), c = 2 log2 e; NaN marks an entry outside the range
// in which EP * EQ can neither be inf * 0 nor lose a factor to a denormal
constexpr float kExpScale = 2.8853900817779268f;
__device__ __forceinline__ float exp_entry(float v) {
  const float t = v * kExpScale;
  return fabsf(t) <= 100.f ? __builtin_amdgcn_exp2f(t) : __builtin_nanf("");
}
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float wave_sum(float v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// y[out] += sum_k in_k * Wt[k][out]  (in: lane = k over RIN registers, zero beyond K; Wt rows are
// zero-padded to 64*RIN in LDS, so the trip count is a compile-time constant and the loop unrolls).
// The LDS reads carry no per-lane predicate: with `if (o < n_out)` around them hipcc emitted, per k, a branch, the
// ds_read, an s_waitcnt lgkmcnt(0) and the fma -- 64 serialised LDS latencies, 4.0 us of an 18.7 us layer kernel
// (in-kernel stamps); unpredicated, eight reads are in flight per wait.
template <int RIN, int ROUT>
__device__ __forceinline__ void matvec(const float (&in)[RIN], const float* __restrict__ wt, int n_out, int lane,
                                       float (&out)[ROUT]) {
#pragma unroll
  for (int r = 0; r < RIN; ++r) {
#pragma unroll 8
    for (int l = 0; l < 64; ++l) {
      const float v = lane_bcast(in[r], l);
      const float* row = wt + (r * 64 + l) * n_out;
#pragma unroll
      for (int ro = 0; ro < ROUT; ++ro)       // lanes past n_out re-read the row's last entry (their result is never stored):
        out[ro] = __builtin_fmaf(v, row[min(ro * 64 + lane, n_out - 1)], out[ro]);   // a guard here cost one LDS latency per k
    }
  }
}

}  // namespace

namespace {

#ifdef NBD_GNN_TRACE
__device__ long long* g_gnn_trace = nullptr;
#define GT(i) if (lane == 0 && g_gnn_trace) g_gnn_trace[((size_t)blockIdx.x * WPB + wave) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define GT(i)
#endif
template <int R, int FM, int WPB>      // FM: on-the-fly feature count the loops are unrolled for (4 or kFMax); WPB: waves (= nodes in flight) per workgroup
__global__ __launch_bounds__(64 * WPB) void gnn_layer_kernel(const nbd_gnn_layer_args a) {
  extern __shared__ float smem[];
  const int H = a.h;
  constexpr int KP = 64 * R;               // mat-vec depth, zero padded
  const bool folded = a.epilogue == NBD_GNN_NEXT_PQ_FOLDED;
  float* w2t = smem;                       // [KP][H]   (a.w2t is already W2 transposed: [k][out]); absent when folded
  float* ept = smem + (folded ? 0 : KP * H);   // [KP][n_ep] epilogue matrix (NEXT_PQ / NEXT_PQ_FOLDED)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_ep = (a.epilogue == NBD_GNN_NEXT_PQ || folded) ? a.ep_out : 0;
  GT(0)
  for (int idx = threadIdx.x; !folded && idx < KP * H; idx += 64 * WPB) w2t[idx] = idx < H * H ? a.w2t[idx] : 0.f;
  for (int idx = threadIdx.x; idx < KP * n_ep; idx += 64 * WPB) ept[idx] = idx < H * n_ep ? a.w_ep[idx] : 0.f;
  __syncthreads();
  GT(1)

  // on-the-fly P/Q weights of this lane's channels (first layer, F <= 8)
  float wp[R][FM], wq[R][FM], bp[R];
  if (a.pq == nullptr) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int h = r * 64 + lane;
      bp[r] = h < H ? a.bpq[h] : 0.f;
#pragma unroll
      for (int f = 0; f < FM; ++f) {
        const bool ok = h < H && f < a.f;
        wp[r][f] = ok ? a.wpq[(size_t)h * a.f + f] : 0.f;
        wq[r][f] = ok ? a.wpq[(size_t)(H + h) * a.f + f] : 0.f;
      }
    }
  }

  for (int node = blockIdx.x * WPB + wave; node < a.n; node += gridDim.x * WPB) {
    const int e0 = a.rowptr ? a.rowptr[node] : node * a.fixed_k;
    const int e1 = a.rowptr ? a.rowptr[node + 1] : (node + 1) * a.fixed_k;
    const int deg = e1 - e0;
    float p[R], s[R];
    if (a.pq) {
#pragma unroll
      for (int r = 0; r < R; ++r) { const int h = r * 64 + lane; p[r] = h < H ? a.pq[(size_t)node * a.ldpq + h] : 0.f; }
    } else {
      float xi[FM];
#pragma unroll
      for (int f = 0; f < FM; ++f) xi[f] = f < a.f ? a.x[(size_t)node * a.ldx + f] : 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float acc = bp[r];
#pragma unroll
        for (int f = 0; f < FM; ++f) acc = __builtin_fmaf(wp[r][f], xi[f], acc);
        p[r] = acc;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] = 0.f;
    // With tables: tanh(P_i + Q_j) = 1 - 2 / (EP_i EQ_j + 1), summed as deg - 2 sum 1 / (EP_i EQ_j + 1). Per edge and channel
    // ONE quarter-rate instruction (v_rcp_f32) and 1.5 packed ones (two edges per v_pk_mul / v_pk_add) where the form below
    // spends two quarter-rate and five full-rate ones -- 22 issue cycles per edge against 52; the rows gathered are EQ_j.
    bool exact = true;
    if constexpr (R == 1) {
      if (a.epq) {
        const int hl = min(lane, H - 1);                // lanes past H redo channel H-1 (never stored)
        const float ep = a.epq[(size_t)node * a.ldepq + hl];
        const int eqo = H + hl;                         // row base uniform (SGPR), lane offset in a VGPR: saddr loads
        const f2 ep2 = {ep, ep}, one2 = {1.f, 1.f};
        f2 acc = {0.f, 0.f};
        for (int eb = e0; eb < e1; eb += 64) {
          const int cnt = min(64, e1 - eb);
          const int jv = lane < cnt ? (int)a.src[eb + lane] : 0;
          int t = 0;
          for (; t + kEF <= cnt; t += kEF) {            // whole groups: kEF rows in flight, no masks
            float q[kEF];
#pragma unroll
            for (int u = 0; u < kEF; ++u) q[u] = (a.epq + (size_t)__builtin_amdgcn_readlane(jv, t + u) * a.ldepq)[eqo];
#pragma unroll
            for (int u = 0; u < kEF; u += 2) {
              const f2 d = ep2 * f2{q[u], q[u + 1]} + one2;
              acc += f2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
            }
          }
          for (; t < cnt; t += 4) {                     // the rest, four at a time: a missing edge is EQ = +inf, 1 / inf = 0
            float q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const float v = (a.epq + (size_t)__builtin_amdgcn_readlane(jv, min(t + u, cnt - 1)) * a.ldepq)[eqo];
              q[u] = t + u < cnt ? v : __builtin_inff();
            }
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
              const f2 d = ep2 * f2{q[u], q[u + 1]} + one2;
              acc += f2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
            }
          }
        }
        s[0] = (float)deg - 2.0f * (acc.x + acc.y);
        exact = __ballot(s[0] != s[0]) != 0;            // a marked table entry (or inf * 0) somewhere in this node's edges
        if (exact) s[0] = 0.f;
      }
    }
    // edges in chunks of 64: one coalesced index load, then wave-uniform j's, kEF neighbour rows in flight
    // (the loop is a chain of L2 round trips at 2 waves/SIMD: depth is what hides them)
    for (int eb = e0; exact && eb < e1; eb += 64) {
      const int cnt = min(64, e1 - eb);
      const int jv = lane < cnt ? (int)a.src[eb + lane] : 0;
      for (int t = 0; a.pq && t < cnt; t += kEF) {
        int j[kEF];
#pragma unroll
        for (int u = 0; u < kEF; ++u) j[u] = __builtin_amdgcn_readlane(jv, min(t + u, cnt - 1));
        {
          float q[kEF][R];
#pragma unroll
          for (int u = 0; u < kEF; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const int h = r * 64 + lane;
              q[u][r] = h < H ? a.pq[(size_t)j[u] * a.ldpq + H + h] : 0.f;
            }
#pragma unroll
          for (int u = 0; u < kEF; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const float v = fast_tanh(__fadd_rn(p[r], q[u][r]));
              s[r] += (t + u < cnt) ? v : 0.f;
            }
        }
      }
      if (!a.pq) {
        // first layer: every lane fetches ONE neighbour's features (a single gather latency for the whole
        // chunk), then neighbour t's row is broadcast out of lane t; Q is formed on the fly
        float xv[FM];
#pragma unroll
        for (int f = 0; f < FM; ++f) xv[f] = (lane < cnt && f < a.f) ? a.x[(size_t)jv * a.ldx + f] : 0.f;
        for (int t = 0; t < cnt; ++t) {
          float xj[FM];
#pragma unroll
          for (int f = 0; f < FM; ++f)
            xj[f] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xv[f]), t));
#pragma unroll
          for (int r = 0; r < R; ++r) {
            float q = 0.f;
#pragma unroll
            for (int f = 0; f < FM; ++f) q = __builtin_fmaf(wq[r][f], xj[f], q);
            s[r] += fast_tanh(__fadd_rn(p[r], q));
          }
        }
      }
    }
    GT(2)
    if (a.aggr == 1) {
      const float inv = 1.0f / (float)max(deg, 1);
#pragma unroll
      for (int r = 0; r < R; ++r) s[r] *= inv;
    }
    // y = W2 S + beta b2
    const float beta = a.aggr == 1 ? (deg > 0 ? 1.f : 0.f) : (float)deg;
    if (folded) {
      // next [P|Q] = (Wpq W2) S + beta (Wpq b2) + bpq: the two mat-vecs folded into one on the host
      // (w_ep = (Wpq W2)^T, b2 = Wpq b2 here), so W2^T is neither staged nor applied
      float o[2 * R];
#pragma unroll
      for (int r = 0; r < 2 * R; ++r) { const int c = r * 64 + lane; o[r] = c < n_ep ? __builtin_fmaf(beta, a.b2[c], a.b_ep[c]) : 0.f; }
      matvec<R, 2 * R>(s, ept, n_ep, lane, o);
#pragma unroll
      for (int r = 0; r < 2 * R; ++r) {
        const int c = r * 64 + lane;
        if (c < n_ep) {
          a.out[(size_t)node * a.ldout + c] = o[r];
          if (a.out_epq) a.out_epq[(size_t)node * a.ldout_epq + c] = exp_entry(o[r]);
        }
      }
      GT(3)
      continue;
    }
    float y[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { const int h = r * 64 + lane; y[r] = h < H ? beta * a.b2[h] : 0.f; }
    matvec<R, R>(s, w2t, H, lane, y);
    GT(4)

    if (a.epilogue == NBD_GNN_WRITE_X) {
#pragma unroll
      for (int r = 0; r < R; ++r) { const int h = r * 64 + lane; if (h < H) a.out[(size_t)node * a.ldout + h] = y[r]; }
    } else if (a.epilogue == NBD_GNN_NEXT_PQ) {
      float o[2 * R];
#pragma unroll
      for (int r = 0; r < 2 * R; ++r) { const int c = r * 64 + lane; o[r] = c < n_ep ? a.b_ep[c] : 0.f; }
      matvec<R, 2 * R>(y, ept, n_ep, lane, o);
#pragma unroll
      for (int r = 0; r < 2 * R; ++r) {
        const int c = r * 64 + lane;
        if (c < n_ep) {
          a.out[(size_t)node * a.ldout + c] = o[r];
          if (a.out_epq) a.out_epq[(size_t)node * a.ldout_epq + c] = exp_entry(o[r]);
        }
      }
    } else {
      // LayerNorm over [enc (E) || y (H)] without materialising the concatenation
      const int E = a.e, C = E + H;
      float enc[kMaxZR];
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < kMaxZR; ++r) {
        const int c = r * 64 + lane;
        enc[r] = c < E ? a.enc[(size_t)node * a.ldenc + c] : 0.f;
        sum += enc[r];
      }
#pragma unroll
      for (int r = 0; r < R; ++r) sum += (r * 64 + lane < H) ? y[r] : 0.f;
      const float mean = wave_sum(sum) / (float)C;
      float var = 0.f;
#pragma unroll
      for (int r = 0; r < kMaxZR; ++r) { const float d = enc[r] - mean; var += (r * 64 + lane < E) ? d * d : 0.f; }
#pragma unroll
      for (int r = 0; r < R; ++r) { const float d = y[r] - mean; var += (r * 64 + lane < H) ? d * d : 0.f; }
      const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)C + a.ln_eps);
      float ze[kMaxZR], zy[R];
#pragma unroll
      for (int r = 0; r < kMaxZR; ++r) {
        const int c = r * 64 + lane;
        ze[r] = c < E ? (enc[r] - mean) * rstd * a.ln_g[c] + a.ln_b[c] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int h = r * 64 + lane;
        zy[r] = h < H ? (y[r] - mean) * rstd * a.ln_g[E + h] + a.ln_b[E + h] : 0.f;
      }
      GT(5)
      if (a.epilogue == NBD_GNN_FINAL_LN) {
#pragma unroll
        for (int r = 0; r < kMaxZR; ++r) { const int c = r * 64 + lane; if (c < E) a.out[(size_t)node * a.ldout + c] = ze[r]; }
#pragma unroll
        for (int r = 0; r < R; ++r) { const int h = r * 64 + lane; if (h < H) a.out[(size_t)node * a.ldout + E + h] = zy[r]; }
      } else {  // FINAL_HEAD: out[d] = sum_c z[c] Wh[d][c] + bh[d]
        // all heads at once: their weight rows are fetched together and their wave sums advance in lock step (same
        // arithmetic, same order per head; measured: no change on the captured step). Un-predicating the LayerNorm /
        // head loads as well (clamped indices) was WORSE: hipcc hoisted them all, 128 VGPRs + scratch, step 66 -> 82 us.
        float part[kMaxOut];
#pragma unroll
        for (int d = 0; d < kMaxOut; ++d) {
          part[d] = 0.f;
          if (d < a.ep_out) {                                 // wave-uniform
            const float* wrow = a.w_ep + (size_t)d * C;
#pragma unroll
            for (int r = 0; r < kMaxZR; ++r) { const int c = r * 64 + lane; if (c < E) part[d] = __builtin_fmaf(ze[r], wrow[c], part[d]); }
#pragma unroll
            for (int r = 0; r < R; ++r) { const int h = r * 64 + lane; if (h < H) part[d] = __builtin_fmaf(zy[r], wrow[E + h], part[d]); }
          }
        }
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
          for (int d = 0; d < kMaxOut; ++d)
            if (d < a.ep_out) part[d] += __shfl_xor(part[d], off);
        }
#pragma unroll
        for (int d = 0; d < kMaxOut; ++d) {
          if (d < a.ep_out && lane == 0) {
            const float o_d = part[d] + a.b_ep[d];
            a.out[(size_t)node * a.ldout + d] = o_d;
            if (a.kick_vel) {                              // v += c * a, rounded as the separate kick kernel rounds it
              float* v = a.kick_vel + (size_t)node * a.ep_out + d;
              *v = __fadd_rn(*v, __fmul_rn(a.kick_c, o_d));
            }
          }
        }
      }
      GT(6)
    }
  }
}

inline int status() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : (int)e; }

}  // namespace

extern "C" {

#ifdef NBD_GNN_TRACE
int nbd_debug_gnn_trace(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gnn_trace), &buf, sizeof(buf)); }
#endif

int nbd_gnn_layer_f32(const nbd_gnn_layer_args* args, nbd_stream_t stream) {
  if (!args) return NBD_E_BADARG;
  const nbd_gnn_layer_args a = *args;
  if (a.n < 0 || a.h <= 0 || a.aggr < 0 || a.aggr > 1) return NBD_E_BADARG;
  if (a.n == 0) return 0;
  if (a.h > 64 * kMaxR) return NBD_E_UNSUPPORTED;
  if ((!a.w2t && a.epilogue != NBD_GNN_NEXT_PQ_FOLDED) || !a.b2 || !a.out || (!a.rowptr && a.fixed_k < 0)) return NBD_E_BADARG;
  if (!a.src && (a.rowptr || a.fixed_k > 0)) return NBD_E_BADARG;
  if (a.pq) { if (a.ldpq < 2 * a.h) return NBD_E_BADARG; }
  else { if (!a.x || !a.wpq || !a.bpq || a.f <= 0 || a.ldx < a.f) return NBD_E_BADARG; if (a.f > kFMax) return NBD_E_UNSUPPORTED; }
  if (a.kick_vel && a.epilogue != NBD_GNN_FINAL_HEAD) return NBD_E_BADARG;
  if (a.epq && (a.h > 64 || a.ldepq < 2 * a.h)) return NBD_E_BADARG;
  if (a.out_epq && ((a.epilogue != NBD_GNN_NEXT_PQ && a.epilogue != NBD_GNN_NEXT_PQ_FOLDED) || a.ldout_epq < a.ep_out)) return NBD_E_BADARG;
  int n_ep = 0;
  switch (a.epilogue) {
    case NBD_GNN_WRITE_X: if (a.ldout < a.h) return NBD_E_BADARG; break;
    case NBD_GNN_NEXT_PQ:
    case NBD_GNN_NEXT_PQ_FOLDED:
      if (!a.w_ep || !a.b_ep || a.ep_out <= 0 || a.ldout < a.ep_out) return NBD_E_BADARG;
      if (a.ep_out > 2 * 64 * kMaxR || a.ep_out > 2 * 64 * ((a.h + 63) / 64)) return NBD_E_UNSUPPORTED;
      n_ep = a.ep_out; break;
    case NBD_GNN_FINAL_HEAD:
    case NBD_GNN_FINAL_LN:
      if (a.e < 0 || (a.e > 0 && (!a.enc || a.ldenc < a.e)) || !a.ln_g || !a.ln_b) return NBD_E_BADARG;
      if (a.e > 64 * kMaxZR) return NBD_E_UNSUPPORTED;
      if (a.epilogue == NBD_GNN_FINAL_HEAD) {
        if (!a.w_ep || !a.b_ep || a.ep_out <= 0 || a.ldout < a.ep_out) return NBD_E_BADARG;
        if (a.ep_out > kMaxOut) return NBD_E_UNSUPPORTED;
      } else if (a.ldout < a.e + a.h) return NBD_E_BADARG;
      break;
    default: return NBD_E_BADARG;
  }
  const int kp = 64 * ((a.h + 63) / 64);
  const size_t shmem = ((a.epilogue == NBD_GNN_NEXT_PQ_FOLDED ? 0 : (size_t)kp * a.h) + (size_t)kp * n_ep) * sizeof(float);
  if (shmem > 64 * 1024) return NBD_E_UNSUPPORTED;   // H = 64: 48 KiB; H = 128 fits only without NEXT_PQ
  // One node per wave; a workgroup stages the layer's matrices into LDS once (48 KiB at H = 64 with NEXT_PQ) for all
  // its waves. 16 waves per workgroup: 4096 nodes are 256 workgroups, one per CU, every node in flight at once and
  // 256 stagings per launch. Measured on the captured GNN step (N = 4096, k = 50): 4 / 8 / 16 waves per workgroup
  // 76.1 / 72.7 / 71.5 us (round 1's shape: 4 waves, 512 workgroups looping over two nodes each).
  hipStream_t st = (hipStream_t)stream;
  const bool f4 = a.pq != nullptr || a.f <= 4;
  constexpr int W = 16;
  int blocks = (a.n + W - 1) / W;
  if (blocks > 512) blocks = 512;                        // two resident workgroups per CU at most (LDS, 32 waves); more nodes loop
  if (a.h <= 64) {
    if (f4) gnn_layer_kernel<1, 4, W><<<blocks, 64 * W, shmem, st>>>(a);
    else gnn_layer_kernel<1, kFMax, W><<<blocks, 64 * W, shmem, st>>>(a);
  } else {
    if (f4) gnn_layer_kernel<2, 4, W><<<blocks, 64 * W, shmem, st>>>(a);
    else gnn_layer_kernel<2, kFMax, W><<<blocks, 64 * W, shmem, st>>>(a);
  }
  return status();
}

// GraphModel.predict's device work in ONE call (gnn.py:205-215 -> transform_to_graph :11-22 -> forward :130-148): the
// kNN search (hinted by the buffer's previous content when asked) and the fused layers on its result. The host side of
// an eager rollout step was five ctypes calls, each marshalling a 30-field struct, around 66 us of kernels; now it fills
// this struct once per (model, n, k) and patches a handful of pointers per call.
// the exponential tables of a forward pass: one [n][2h] block per layer, in args->workspace
static bool forward_uses_tables(const nbd_gnn_forward_args& a) {
  if (a.n <= 0 || a.n > 8192 || a.k > 200 || a.n_layers < 1 || a.n_layers > NBD_GNN_MAX_LAYERS) return false;
  const nbd_gnn_layer_args& l0 = a.layers[0];
  if (l0.pq || !l0.x || l0.f <= 0 || l0.f > kFMax || l0.h <= 0 || l0.h > 64) return false;
  for (int l = 0; l < a.n_layers; ++l) {
    const nbd_gnn_layer_args& la = a.layers[l];
    if (la.h != l0.h) return false;
    if (l + 1 < a.n_layers && !((la.epilogue == NBD_GNN_NEXT_PQ || la.epilogue == NBD_GNN_NEXT_PQ_FOLDED) && la.ep_out == 2 * la.h &&
                                a.layers[l + 1].pq == la.out && a.layers[l + 1].ldpq == la.ldout)) return false;
  }
  return true;
}
size_t nbd_gnn_forward_workspace_bytes(const nbd_gnn_forward_args* args) {
  if (!args || !forward_uses_tables(*args)) return 0;
  return (size_t)args->n_layers * args->n * 2 * args->layers[0].h * sizeof(float);
}

int nbd_gnn_forward_f32(const nbd_gnn_forward_args* args, nbd_stream_t stream) {
  if (!args) return NBD_E_BADARG;
  const nbd_gnn_forward_args& a = *args;
  if (a.n < 0 || a.k < 0 || a.n_layers < 1 || a.n_layers > NBD_GNN_MAX_LAYERS) return NBD_E_BADARG;
  if (a.n == 0) return 0;
  if (!a.pos || !a.edge_index) return NBD_E_BADARG;
  const int avail = a.n - (a.loop ? 0 : 1);
  const int kk = a.k < avail ? a.k : (avail > 0 ? avail : 0);
  const int64_t e = (int64_t)a.n * kk;
  const size_t need = nbd_gnn_forward_workspace_bytes(args);
  bool tables = need > 0 && a.workspace && a.workspace_bytes >= need && kk > 0;
  float* tab = static_cast<float*>(a.workspace);
  const int h = a.layers[0].h;
  int rc = NBD_E_UNSUPPORTED;
  if (tables) {
    const nbd_gnn_layer_args& l0 = a.layers[0];
    const nbd_knn_pq_args pq = {l0.x, l0.ldx, l0.f, h, l0.wpq, l0.bpq, tab, 2 * h};
    rc = nbd_knn_graph_hint_pq_f32(a.pos, a.n, a.k, a.loop, e, a.edge_index, a.use_hint ? a.edge_index : nullptr, &pq, stream);
    if (rc == NBD_E_UNSUPPORTED) tables = false;
    else if (rc) return rc;
  }
  if (!tables) {
    rc = nbd_knn_graph_hint_f32(a.pos, a.n, a.k, a.loop, nullptr, nullptr, nullptr, e, a.edge_index,
                                a.use_hint ? a.edge_index : nullptr, stream);
    if (rc) return rc;
  }
  for (int l = 0; l < a.n_layers; ++l) {
    nbd_gnn_layer_args la = a.layers[l];
    la.rowptr = nullptr; la.src = a.edge_index; la.fixed_k = kk; la.n = a.n;      // edge_index[0]: the sources, kk per node
    if (tables) {
      la.epq = tab + (size_t)l * a.n * 2 * h; la.ldepq = 2 * h;
      if (l + 1 < a.n_layers) { la.out_epq = tab + (size_t)(l + 1) * a.n * 2 * h; la.ldout_epq = 2 * h; }
    }
    rc = nbd_gnn_layer_f32(&la, stream);
    if (rc) return rc;
  }
  return 0;
}

}  // extern "C"
