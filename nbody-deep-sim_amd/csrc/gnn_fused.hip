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
#include <stdlib.h>

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
// the exponential-table form (nbd.h, nbd_gnn_layer_args.epq): 2^(c v), c = 2 log2 e; NaN marks an entry outside |c v| <= 100,
// the range in which EP * EQ can neither be inf * 0 nor lose a factor to a denormal; nodes with a marked entry are recomputed
// exactly. Accuracy inside the range: the rounding of t = c v puts a relative error of 0.69 |t| 2^-24 on an entry, which the
// quotient (EP EQ - 1) / (EP EQ + 1) turns into an ABSOLUTE error of at most half the sum of the two where P + Q is near 0 --
// 4e-6 at |t_P| = |t_Q| = 100, on one edge of one channel of a mean over k edges (tests/test_surrogate_gpu.py holds a model
// with pre-activations all over the range to the oracle at 1e-5). Tighter limits were measured (round 4, advisor's note): 16,
// 32 and 64 mark the outlying bodies of a Plummer sphere (|x| up to 50 enters P and Q) and cost 10 / 10 / 8 % of the captured
// rollout step (0.0403 / 0.0403 / 0.0394 against 0.0364 ms, same box) for an error already under the bar.
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

// S_i = sum_j tanh(P_i + Q_j) over the edges [e0, e1) of `node`, lane = channel (R channels per lane): the edge loop of
// both layer kernels below.
template <int R, int FM>
__device__ __forceinline__ void edge_aggregate(const nbd_gnn_layer_args& a, int node, int lane, int H, int e0, int e1,
                                               const float (&wp)[R][FM], const float (&wq)[R][FM], const float (&bp)[R],
                                               float (&s)[R]) {
  const int deg = e1 - e0;
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
  if (!exact) return;
  float p[R];
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
  // edges in chunks of 64: one coalesced index load, then wave-uniform j's, kEF neighbour rows in flight
  // (the loop is a chain of L2 round trips at 2 waves/SIMD: depth is what hides them)
  for (int eb = e0; eb < e1; eb += 64) {
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
}

#ifdef NBD_GNN_TRACE
__device__ long long* g_gnn_trace = nullptr;
__device__ int g_gnn_trace_epi = -1;        // trace only launches with this epilogue (-1: all; the last writer wins)
#define GT(i) if (lane == 0 && g_gnn_trace && (g_gnn_trace_epi < 0 || g_gnn_trace_epi == a.epilogue)) g_gnn_trace[((size_t)blockIdx.x * WPB + wave) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();
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
    float s[R];
    edge_aggregate<R, FM>(a, node, lane, H, e0, e1, wp, wq, bp, s);
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

// ---- H = 64: the layer with its dense tail on the matrix pipe.
// In-kernel stamps of the kernel above at N = 4096, k = 50 (tools/gnn_layer_trace.py): of 13.9 us, 1.3-1.9 go to staging
// W^T into LDS in front of the first edge, 2.5-3.4 to the mat-vec (LDS-bandwidth bound: each of a workgroup's 16 waves
// reads the whole 16-32 KB matrix for ITS node), 4.3 to LayerNorm + head on the last layer (cold global loads of their
// constants and of the velocity, LDS-routed wave reductions). Here the 16 nodes of a workgroup are ONE 16-row operand:
//   - nothing is staged: every wave asks for its registers' worth of what the tail needs -- its 16 x 64 x 16 block's B
//     fragments (v_mfma_f32_16x16x4_f32: lane (k = l >> 4, n = l & 15)), biases, LayerNorm / head rows, its velocity --
//     together with its first edge indices, so one cold round trip covers them all;
//   - the edge loop (edge_aggregate) leaves S_i in lane = channel; the 16 rows meet in LDS, waves 0 .. n_ep/16 - 1
//     multiply (16 MFMAs each, fp32 in, fp32 accumulate) and hand the 16 x n_ep result back through LDS;
//   - wave = node again: bias, stores (+ the next layer's exponential table), or LayerNorm + head with DPP reductions.
typedef float f4m __attribute__((ext_vector_type(4)));
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_add(float x) {      // x + (x moved by one DPP step; lanes the move does not reach add 0)
  return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROWS, 0xf, false));
}
// the sum over the wave, in every lane: row_shr 1/2/4/8 inside the rows of 16, row_bcast 15 and 31 across them, all VALU
// (__shfl_xor is a ds_bpermute -- an LDS round trip -- per step)
__device__ __forceinline__ float wave_sum_dpp(float x) {
  x = dpp_add<0x111, 0xf>(x); x = dpp_add<0x112, 0xf>(x); x = dpp_add<0x114, 0xf>(x); x = dpp_add<0x118, 0xf>(x);
  x = dpp_add<0x142, 0xa>(x); x = dpp_add<0x143, 0xc>(x);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

constexpr int kT = 16;               // nodes (= waves) per workgroup = rows of the MFMA operand
constexpr int kSS = 66;              // row stride of S in LDS: A-fragment reads (m, 4 ks + q) fall on banks 2 m + q
constexpr int kOS = 132;             // row stride of the product

template <int FM>
__global__ __launch_bounds__(64 * kT) void gnn_layer64_kernel(const nbd_gnn_layer_args a) {
  __shared__ float S[kT * kSS];
  __shared__ float O[kT * kOS];
  constexpr int H = 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  [[maybe_unused]] constexpr int WPB = kT;
  GT(0)
  const bool folded = a.epilogue == NBD_GNN_NEXT_PQ_FOLDED;
  const int n_out = folded ? a.ep_out : H;              // columns of the product: 128 or 64
  const int node = blockIdx.x * kT + wave;
  const bool live = node < a.n;
  const int nc = live ? node : a.n - 1;                 // waves past the end shadow the last node (they only keep the barriers)
  const int e0 = a.rowptr ? a.rowptr[nc] : nc * a.fixed_k;
  const int e1 = a.rowptr ? a.rowptr[nc + 1] : (nc + 1) * a.fixed_k;
  const int deg = e1 - e0;

  // ---- requests for everything the tail needs
  float bfrag[16];
  const float* wmat = folded ? a.w_ep : a.w2t;          // [64][n_out], k-major
  const bool mult = wave * 16 < n_out;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) bfrag[ks] = mult ? wmat[(size_t)(4 * ks + (lane >> 4)) * n_out + 16 * wave + (lane & 15)] : 0.f;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;          // folded: b2', b_ep of columns lane and 64 + lane; head: b2
  float lg_e = 0.f, lb_e = 0.f, lg_y = 0.f, lb_y = 0.f, encv = 0.f, velv = 0.f, posv = 0.f, bh = 0.f;
  float wh_e[kMaxOut], wh_y[kMaxOut];
  const int E = a.e;
  if (folded) {
    c0 = a.b2[lane]; c1 = a.b_ep[lane];
    if (64 + lane < n_out) { c2 = a.b2[64 + lane]; c3 = a.b_ep[64 + lane]; }
  } else {
    c0 = a.b2[lane];
    lg_y = a.ln_g[E + lane]; lb_y = a.ln_b[E + lane];
    if (lane < E) { lg_e = a.ln_g[lane]; lb_e = a.ln_b[lane]; encv = a.enc[(size_t)nc * a.ldenc + lane]; }
#pragma unroll
    for (int d = 0; d < kMaxOut; ++d) {
      wh_e[d] = 0.f; wh_y[d] = 0.f;
      if (d < a.ep_out) {                               // wave-uniform
        wh_y[d] = a.w_ep[(size_t)d * (E + H) + E + lane];
        if (lane < E) wh_e[d] = a.w_ep[(size_t)d * (E + H) + lane];
      }
    }
    if (lane < a.ep_out) {
      bh = a.b_ep[lane];
      if (a.adv_pos) { velv = a.adv_vel_half[(size_t)nc * 3 + lane]; posv = a.adv_pos[(size_t)nc * 3 + lane]; }
      else if (a.kick_vel) velv = a.kick_vel[(size_t)nc * a.ep_out + lane];
    }
  }
  float wp[1][FM], wq[1][FM], bp[1];
  if (a.pq == nullptr) {
    bp[0] = a.bpq[lane];
#pragma unroll
    for (int f = 0; f < FM; ++f) {
      const bool ok = f < a.f;
      wp[0][f] = ok ? a.wpq[(size_t)lane * a.f + f] : 0.f;
      wq[0][f] = ok ? a.wpq[(size_t)(H + lane) * a.f + f] : 0.f;
    }
  }
  GT(1)

  // ---- edges
  float s[1];
  edge_aggregate<1, FM>(a, nc, lane, H, e0, e1, wp, wq, bp, s);
  GT(2)
  if (a.aggr == 1) s[0] *= 1.0f / (float)max(deg, 1);
  const float beta = a.aggr == 1 ? (deg > 0 ? 1.f : 0.f) : (float)deg;

  // ---- the 16 rows times W^T
  S[wave * kSS + lane] = s[0];
  __syncthreads();
  if (mult) {
    f4m acc = {0.f, 0.f, 0.f, 0.f};
    const float* arow = S + (lane & 15) * kSS + (lane >> 4);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * ks], bfrag[ks], acc, 0, 0, 0);
    float* ocol = O + (4 * (lane >> 4)) * kOS + 16 * wave + (lane & 15);     // D: rows 4 (l >> 4) + v, column l & 15
#pragma unroll
    for (int v = 0; v < 4; ++v) ocol[v * kOS] = acc[v];
  }
  __syncthreads();
  GT(4)
  if (!live) return;

  if (folded) {
    // next [P|Q] = (Wpq W2) S + beta (Wpq b2) + bpq
    const float o0 = O[wave * kOS + lane] + __builtin_fmaf(beta, c0, c1);
    a.out[(size_t)node * a.ldout + lane] = o0;
    if (a.out_epq) a.out_epq[(size_t)node * a.ldout_epq + lane] = exp_entry(o0);
    if (64 + lane < n_out) {
      const float o1 = O[wave * kOS + 64 + lane] + __builtin_fmaf(beta, c2, c3);
      a.out[(size_t)node * a.ldout + 64 + lane] = o1;
      if (a.out_epq) a.out_epq[(size_t)node * a.ldout_epq + 64 + lane] = exp_entry(o1);
    }
    GT(3)
    return;
  }
  // y = W2 S + beta b2, then LayerNorm over [enc (E <= 64) || y (64)] and the head
  const float y = O[wave * kOS + lane] + beta * c0;
  const float C = (float)(E + H);
  const float mean = wave_sum_dpp(y + (lane < E ? encv : 0.f)) / C;
  const float dy = y - mean, de = lane < E ? encv - mean : 0.f;
  const float rstd = 1.0f / sqrtf(wave_sum_dpp(dy * dy + de * de) / C + a.ln_eps);
  const float zy = dy * rstd * lg_y + lb_y;
  const float ze = lane < E ? de * rstd * lg_e + lb_e : 0.f;
  GT(5)
  if (a.epilogue == NBD_GNN_FINAL_LN) {
    if (lane < E) a.out[(size_t)node * a.ldout + lane] = ze;
    a.out[(size_t)node * a.ldout + E + lane] = zy;
    return;
  }
  float mine = 0.f;                                      // lane d keeps output d
#pragma unroll
  for (int d = 0; d < kMaxOut; ++d) {
    if (d < a.ep_out) {
      const float t = wave_sum_dpp(__builtin_fmaf(zy, wh_y[d], ze * wh_e[d]));
      mine = lane == d ? t : mine;
    }
  }
  if (lane < a.ep_out) {
    const float o_d = mine + bh;
    a.out[(size_t)node * a.ldout + lane] = o_d;
    if (a.adv_pos) {                                     // nbd.h: this step's (x, v), then the next step's half-kick + drift
      const float vfull = __fadd_rn(velv, __fmul_rn(a.kick_c, o_d));
      a.kick_vel[(size_t)node * 3 + lane] = vfull;
      a.adv_pos_out[(size_t)node * 3 + lane] = posv;
      const float vnext = __fadd_rn(vfull, __fmul_rn(a.kick_c, o_d));
      a.adv_vel_half[(size_t)node * 3 + lane] = vnext;
      const float xn = __fadd_rn(posv, __fmul_rn(a.adv_dt, vnext));
      a.adv_pos[(size_t)node * 3 + lane] = xn;
      a.adv_posm[(size_t)node * 4 + lane] = xn;
    } else if (a.kick_vel) {
      a.kick_vel[(size_t)node * a.ep_out + lane] = __fadd_rn(velv, __fmul_rn(a.kick_c, o_d));
    }
  }
  GT(6)
}

inline int status() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : (int)e; }

}  // namespace

extern "C" {

#ifdef NBD_GNN_TRACE
int nbd_debug_gnn_trace(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gnn_trace), &buf, sizeof(buf)); }
int nbd_debug_gnn_trace_epilogue(int epi) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gnn_trace_epi), &epi, sizeof(epi)); }
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
  const bool adv = a.adv_vel_half || a.adv_pos || a.adv_posm || a.adv_pos_out;
  if (adv && (!a.adv_vel_half || !a.adv_pos || !a.adv_posm || !a.adv_pos_out || !a.kick_vel)) return NBD_E_BADARG;
  if (adv && (a.epilogue != NBD_GNN_FINAL_HEAD || a.ep_out != 3 || a.h != 64 || a.e > 64)) return NBD_E_UNSUPPORTED;
  // The pre-advance epilogue overwrites adv_pos / adv_posm rows of the nodes a workgroup owns while other workgroups of the
  // same launch still gather their neighbours' rows. That is safe only when the edge loop reads the PREVIOUS layer's
  // tables (pq / epq): a first-layer launch (pq == NULL) gathers from x -- the very position rows a one-layer model's
  // caller hands in as adv_posm -- and would mix step t and step t + 1 positions. Refused, whatever x aliases.
  if (adv && !a.pq) return NBD_E_UNSUPPORTED;
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
  // H = 64 with a 64 x 128 folded next-[P|Q] or a LayerNorm (+ head) over at most 64 encoder columns: the MFMA-tail kernel
  // (NBD_GNN_MFMA_TAIL=0: the kernel below, for comparison)
  static const bool mfma_tail = [] { const char* e = getenv("NBD_GNN_MFMA_TAIL"); return !(e && e[0] == '0'); }();
  if (mfma_tail && a.h == 64 &&
      ((a.epilogue == NBD_GNN_NEXT_PQ_FOLDED && a.ep_out % 16 == 0 && a.ep_out <= 128) ||
       ((a.epilogue == NBD_GNN_FINAL_HEAD || a.epilogue == NBD_GNN_FINAL_LN) && a.e <= 64))) {
    const int blocks64 = (a.n + kT - 1) / kT;
    if (f4) gnn_layer64_kernel<4><<<blocks64, 64 * kT, 0, st>>>(a);
    else gnn_layer64_kernel<kFMax><<<blocks64, 64 * kT, 0, st>>>(a);
    return status();
  }
  if (adv) return NBD_E_UNSUPPORTED;                     // the pre-advance epilogue lives in the kernel above only
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
