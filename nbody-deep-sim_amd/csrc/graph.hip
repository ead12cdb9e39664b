// graph.hip -- neighbour search for the surrogate models, gfx950 (MI355X).
//
// Replaces the third-party kernels the reference reaches through PyG: torch_cluster.knn behind
// knn_graph (gnn.py:13, datautils.py:36) and torch_cluster.radius behind radius_graph
// (contconv.py:225). Index-exact specification (restated in oracle/surrogate_oracle.py):
//   d2 = (dx*dx + dy*dy) + dz*dz in fp32 (no FMA);
//   kNN:    k smallest (d2, j) per centre, ties -> lower j, listed ascending; self excluded unless loop
//   radius: the first `cap` indices j (ascending) with d2 < r2 strictly; self included iff loop
//   both restricted to the centre's batch segment [seg_lo, seg_hi).
// Structure: ONE WAVE PER CENTRE. The 64 lanes test 64 candidates per step (coalesced reads of the
// L2-resident position array); decisions are made with wave-wide ballots, so there is no divergence
// and no atomics, and results are bit-reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/nbd.h"

namespace {

constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ float dist2(const float* __restrict__ pos, int j, float xi, float yi, float zi) {
  const float dx = pos[3 * j] - xi, dy = pos[3 * j + 1] - yi, dz = pos[3 * j + 2] - zi;
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// lane l <- lane l-1 across the whole wave (lane 0 <- `lane0`): one DPP move (wave_shr:1, GFX9) instead of
// the LDS round trip __shfl_up compiles to -- the sorted-insert chain below is latency-bound on it.
__device__ __forceinline__ int shift_up1(int v, int lane0) {
  return __builtin_amdgcn_update_dpp(lane0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ float shift_up1(float v, float lane0) {
  return __builtin_bit_cast(float, shift_up1(__builtin_bit_cast(int, v), __builtin_bit_cast(int, lane0)));
}

// ---- kNN: the wave keeps the current best 64*R (d2, j) pairs sorted across lanes (rank = r*64+lane).
template <int R>
__device__ __forceinline__ void knn_insert_centre(const float* __restrict__ pos, int i, int lo, int hi, int kk, int loop,
                                                  int64_t base, int64_t e_total, int64_t* __restrict__ edge_index) {
  const int lane = threadIdx.x & 63;
  const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
  float bd[R];
  int bj[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { bd[r] = __builtin_inff(); bj[r] = -1; }
  float thr = __builtin_inff();                // d2 of the current kk-th best
  const int thr_r = (kk - 1) >> 6, thr_lane = (kk - 1) & 63;

  // candidate positions are prefetched one chunk ahead: the ballot/insert chain of chunk c hides the
  // L2 latency of chunk c+1 (the kernel is latency-bound: one wave per centre, 64 chunks at N = 4096)
  // (loads are unconditional with a clamped index: a predicated load makes hipcc wait for it on the spot)
  const int j_first = min(lo + lane, hi - 1);
  float nx = pos[3 * j_first], ny = pos[3 * j_first + 1], nz = pos[3 * j_first + 2];
  for (int c0 = lo; c0 < hi; c0 += 64) {
    const int j = c0 + lane;
    const float cx = nx, cy = ny, cz = nz;
    const int jn_ = min(j + 64, hi - 1);
    nx = pos[3 * jn_]; ny = pos[3 * jn_ + 1]; nz = pos[3 * jn_ + 2];
    float d = __builtin_inff();
    if (j < hi && (loop || j != i)) {
      const float dx = cx - xi, dy = cy - yi, dz = cz - zi;
      d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    }
    unsigned long long m = __ballot(d < thr);
    while (m) {
      const int b = __builtin_ctzll(m);        // lowest candidate index first
      m &= m - 1;
      const float dn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), b));
      if (!(dn < thr)) continue;               // thr may have dropped inside this chunk
      const int jn = c0 + b;
      int p = 0;                               // insertion rank: after every entry with d2 <= dn
#pragma unroll
      for (int r = 0; r < R; ++r) p += __builtin_popcountll(__ballot(bd[r] <= dn));
#pragma unroll
      for (int r = R - 1; r >= 0; --r) {
        float cd = 0.f;
        int cj = 0;
        if (r > 0) {                                   // carry: last lane of the previous register
          cd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bd[r - 1]), 63));
          cj = __builtin_amdgcn_readlane(bj[r - 1], 63);
        }
        const float ud = shift_up1(bd[r], cd);
        const int uj = shift_up1(bj[r], cj);
        const int rank = r * 64 + lane;
        if (rank > p) { bd[r] = ud; bj[r] = uj; }
        else if (rank == p) { bd[r] = dn; bj[r] = jn; }
      }
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (r == thr_r) t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bd[r]), thr_lane));
      thr = t;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int rank = r * 64 + lane;
    if (rank < kk) {
      edge_index[base + rank] = bj[r];             // row 0: neighbour j (source)
      edge_index[e_total + base + rank] = i;       // row 1: centre i (target)
    }
  }
}

template <int R>
__global__ __launch_bounds__(64 * kWavesPerBlock) void knn_kernel(
    const float* __restrict__ pos, int n, int k, int loop, const int* __restrict__ seg_lo,
    const int* __restrict__ seg_hi, const int64_t* __restrict__ out_off, int64_t e_total,
    int64_t* __restrict__ edge_index) {
  const int i = blockIdx.x * kWavesPerBlock + wave_id();
  if (i >= n) return;
  const int lo = seg_lo ? seg_lo[i] : 0, hi = seg_hi ? seg_hi[i] : n;
  const int kk = min(k, (hi - lo) - (loop ? 0 : 1));
  if (kk <= 0) return;
  knn_insert_centre<R>(pos, i, lo, hi, kk, loop, out_off ? out_off[i] : (int64_t)i * kk, e_total, edge_index);
}

// ---- kNN, selection form (the default for k <= 200): the insertion kernel above is a chain of ~k ln(N/k)
// dependent wave-wide inserts; this one has no serial chain.
//   A  every lane keeps the R smallest d2 of ITS candidates (lane-strided scan, loads pipelined);
//      the kk-th smallest of those 64*R values bounds the kk-th neighbour distance from above (they are
//      64*R distinct candidates), found by a rank count over readlane broadcasts;
//   B  second scan: candidates with d2 <= bound are compacted (ballot prefix) into a per-wave LDS list --
//      a few tens of entries;
//   C  every listed candidate counts how many listed ones precede it in (d2, j) order and, if that rank
//      is < kk, writes its edge directly to slot `rank`.
// Same ordering rule, so the output is identical to the insertion kernel's (tested); if the list
// overflows (only possible with hundreds of exactly tied distances) the wave falls back to the
// insertion form for that centre.
constexpr int kSelCap = 512;     // LDS list entries per wave
constexpr int kBU2 = 2;          // candidate double-chunks (of 128) whose position loads are in flight together in scan B
typedef float f2v __attribute__((ext_vector_type(2)));

template <int R, int RI>
__global__ __launch_bounds__(64 * kWavesPerBlock) void knn_select_kernel(
    const float* __restrict__ pos, int n, int k, int loop, const int* __restrict__ seg_lo,
    const int* __restrict__ seg_hi, const int64_t* __restrict__ out_off, int64_t e_total,
    int64_t* edge_index, const int64_t* hint) {      // hint may alias edge_index: no __restrict__ on either
  __shared__ float ld[kWavesPerBlock][kSelCap];
  __shared__ int lj[kWavesPerBlock][kSelCap];
  const int w = wave_id();
  const int i = blockIdx.x * kWavesPerBlock + w;
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const int lo = seg_lo ? seg_lo[i] : 0, hi = seg_hi ? seg_hi[i] : n;
  const int kk = min(k, (hi - lo) - (loop ? 0 : 1));
  if (kk <= 0) return;
  const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
  const float inf = __builtin_inff();

  // A: per-lane R smallest (sorted ascending in m[0..R-1])
  auto phase_a = [&]() -> float {
  float m[R];
#pragma unroll
  for (int r = 0; r < R; ++r) m[r] = inf;
  for (int c0 = lo; c0 < hi; c0 += 256) {     // four chunks per trip: 12 position loads in flight per lane
    float px[4], py[4], pz[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int jc = min(c0 + 64 * u + lane, hi - 1);
      px[u] = pos[3 * jc]; py[u] = pos[3 * jc + 1]; pz[u] = pos[3 * jc + 2];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = c0 + 64 * u + lane;
      const float dx = px[u] - xi, dy = py[u] - yi, dz = pz[u] - zi;
      float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      if (j >= hi || (!loop && j == i)) d = inf;
#pragma unroll
      for (int r = 0; r < R; ++r) { const float lo_v = fminf(m[r], d); d = fmaxf(m[r], d); m[r] = lo_v; }
    }
  }
  // bound = kk-th smallest of the 64*R kept values (ties: any consistent order gives the same VALUE)
  float bound = inf;
  if (kk <= 64 * R) {
    int rank[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rank[r] = 0;
    for (int l = 0; l < 64; ++l) {
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m[q]), l));
#pragma unroll
        for (int r = 0; r < R; ++r)
          rank[r] += (v < m[r] || (v == m[r] && (q * 64 + l) < (r * 64 + lane))) ? 1 : 0;
      }
    }
    float cand = -inf;     // exactly one (lane, r) has rank == kk-1
#pragma unroll
    for (int r = 0; r < R; ++r) cand = rank[r] == kk - 1 ? m[r] : cand;
    const unsigned long long who = __ballot(cand != -inf);
    bound = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cand), __builtin_ctzll(who)));
  }
  return bound;
  };
  // A': with a hint -- kk distinct neighbours of this centre from an earlier, similar configuration (the
  // previous rollout step's own list) -- the largest of THEIR current distances already bounds the kk-th
  // neighbour distance from above, and the whole first scan is skipped. Invalid entries (out of range,
  // the centre itself) void the hint; duplicates can make the bound too small, which phase B detects.
  float bound = inf;
  bool hinted = false;
  if (hint) {
    float hm = -inf;
    for (int t = lane; t < kk; t += 64) {
      const int64_t j = hint[(int64_t)i * kk + t];
      const bool ok = j >= lo && j < hi && (loop || j != i);
      hm = fmaxf(hm, ok ? dist2(pos, (int)j, xi, yi, zi) : inf);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) hm = fmaxf(hm, __shfl_xor(hm, off));
    bound = hm;
    hinted = bound < inf;
  }
  if (!hinted) bound = phase_a();
  // B: compact everything within the bound. The kernel is bound by VALU issue (r02 counters: 0.5 VALU instructions per
  // candidate, 0.125 of them the distance), so the scan is written for instruction count: a lane tests TWO candidates
  // (j and j + 64) with packed fp32 operations -- v_pk_add / v_pk_mul are IEEE per element and un-fused, so d2 is the
  // specification's (dx*dx + dy*dy) + dz*dz bit for bit -- whole 128-candidate chunks need no bound checks, the
  // self-exclusion test exists only in the loop = 0 instance, and the compaction runs only behind a non-empty mask. Chunk j..j+63 is compacted before j+64..j+127, so the list stays in ascending j (phase C's tie rule).
  auto scan_b = [&](float bnd, auto kLoop) -> int {
  int count = 0;
  // `hit` is the lane's own bit of `mask`; a list that would overflow is not written at all (the count still is, and
  // count > kSelCap sends the centre to the insertion form below)
  auto compact = [&](unsigned long long mask, bool hit, float d, int j) {
    if (mask) {
      const int pop = __builtin_popcountll(mask);
      if (count + pop <= kSelCap) {
        const int slot = count + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
        if (hit) { ld[w][slot] = d; lj[w][slot] = j; }
      }
      count += pop;
    }
  };
  f2v xi2 = {xi, xi}, yi2 = {yi, yi}, zi2 = {zi, zi};
  asm volatile("" : "+v"(xi2), "+v"(yi2), "+v"(zi2));        // VGPR operands: an SGPR source halves v_pk issue
  int c0 = lo;
  for (; c0 + 128 * kBU2 <= hi; c0 += 128 * kBU2) {
    f2v px[kBU2], py[kBU2], pz[kBU2];
#pragma unroll
    for (int u = 0; u < kBU2; ++u) {
      const float* q = pos + 3 * (size_t)(c0 + 128 * u + lane);
      px[u] = f2v{q[0], q[192]}; py[u] = f2v{q[1], q[193]}; pz[u] = f2v{q[2], q[194]};
    }
#pragma unroll
    for (int u = 0; u < kBU2; ++u) {
      f2v d;
      {
#pragma clang fp contract(off)
        const f2v dx = px[u] - xi2, dy = py[u] - yi2, dz = pz[u] - zi2;
        d = (dx * dx + dy * dy) + dz * dz;
      }
      const int j = c0 + 128 * u + lane;
      const bool h0 = d.x <= bnd && (kLoop || j != i), h1 = d.y <= bnd && (kLoop || j + 64 != i);
      compact(__ballot(h0), h0, d.x, j);
      compact(__ballot(h1), h1, d.y, j + 64);
    }
  }
  for (; c0 < hi; c0 += 64) {                 // ragged end of the segment: one chunk at a time, loads clamped
    const int j = c0 + lane;
    const int jc = min(j, hi - 1);
    const float dx = pos[3 * jc] - xi, dy = pos[3 * jc + 1] - yi, dz = pos[3 * jc + 2] - zi;
    const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    const bool h = j < hi && d <= bnd && (kLoop || j != i);
    compact(__ballot(h), h, d, j);
  }
  return count;
  };
  auto phase_b = [&](float bnd) -> int {
    return loop ? scan_b(bnd, std::true_type{}) : scan_b(bnd, std::false_type{});
  };
  int count = phase_b(bound);
  if (hinted && count < kk) {              // the hint was not kk distinct neighbours: do the full first scan
    __builtin_amdgcn_wave_barrier();
    bound = phase_a();
    count = phase_b(bound);
  }
  const int64_t base = out_off ? out_off[i] : (int64_t)i * kk;
  if (count > kSelCap) {                     // cannot happen with distinct distances; massive ties can
    knn_insert_centre<RI>(pos, i, lo, hi, kk, loop, base, e_total, edge_index);
    return;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // C: rank inside the list; the list is in ascending j, so (d2, j) order = (d2, list position)
  for (int h0 = 0; h0 < count; h0 += 64) {
    const int h = h0 + lane;
    const float dh = h < count ? ld[w][h] : inf;
    int rank = 0;
    for (int t = 0; t < count; ++t) {
      const float dt = ld[w][t];                                   // wave-uniform LDS broadcast
      rank += (dt < dh || (dt == dh && t < h)) ? 1 : 0;
    }
    if (h < count && rank < kk) {
      edge_index[base + rank] = lj[w][h];
      edge_index[e_total + base + rank] = i;
    }
  }
}

// ---- kNN, selection form with the positions resident in LDS (one un-segmented system of n <= kStagedMaxN bodies: the
// rollout's search, gnn.py:205-215). The form above fetches every candidate chunk from L2 inside its scan and decides
// chunk by chunk with scalar branches; in-kernel stamps at N = 4096, k = 50 (tools/knn_trace.py) put 7 us of its 15 into
// the scan and 3.7 into the ranking, both chains of VALU -> SGPR -> branch round trips rather than arithmetic. Here a
// 16-wave workgroup copies the position array into LDS once and every later phase is straight-line vector code:
//   stage  4 bodies per thread: three 128-bit loads, three 128-bit LDS writes (layout below); the tail is padded with
//          NaN, so the scan needs no bound checks (NaN <= bound is false)
//   bound  from the hint (the previous step's own list) or, without one, phase A of the form above
//   scan   256 candidates per trip, two per lane in packed fp32 (un-fused: d2 is the specification's bit for bit); the
//          four ballot masks of a trip are parked with v_writelane in the lane whose number is the chunk's -- no
//          branch, no compaction inside the loop
//   expand lane c owns chunk c's mask: an exclusive prefix over the lanes' population counts gives its first list
//          slot; it walks its set bits (a handful), recomputes d2 and writes (key = d2 bits : slot, j) -- ascending j
//   rank   every listed entry counts the smaller keys, read back as 128-bit LDS broadcasts (two keys per read, one
//          64-bit compare + one add-with-carry per key); d2 >= +0, so its bit pattern orders like the value and the
//          slot in the low word is the tie rule (lower j first)
// Same ordering rule and the same output as the form above (tested against it and the insertion form).
#ifdef NBD_KNN_TRACE
__device__ long long* g_knn_trace = nullptr;      // probe build (tools/build_probe.sh): 8 stamps per wave
#define KT(s) if (lane == 0 && g_knn_trace) g_knn_trace[(size_t)(blockIdx.x * kLW + w) * 8 + (s)] = __builtin_amdgcn_s_memrealtime();
#else
#define KT(s)
#endif
constexpr int kLW = 16;              // waves (= centres) per workgroup
// the exponential tables of nbd_gnn_layer_args.epq: 2^(c v), c = 2 log2 e; NaN beyond |c v| = 100 (csrc/gnn_fused.hip)
__device__ __forceinline__ float exp_entry(float v) {
  const float t = v * 2.8853900817779268f;
  return fabsf(t) <= 100.f ? __builtin_amdgcn_exp2f(t) : __builtin_nanf("");
}
constexpr int kLCap = 256;           // list entries per wave; beyond: the insertion form on global memory
constexpr int kStagedMaxN = 8192;    // 12 B * n + 12 B * kLW * kLCap <= 144 KiB of the CU's 160
// LDS layout: per 128 bodies (two chunks c, c+1 of 64) x_c[64] x_c+1[64] y_c[64] y_c+1[64] z_c[64] z_c+1[64], 384 floats
// -- one lane address serves a whole trip of the scan and ds_read2st64_b32 with immediate offsets (0,1) (2,3) (4,5)
// delivers x, y, z of candidates j and j + 64 straight into the register pairs the packed arithmetic wants (with
// x[64] y[64] z[64] per chunk hipcc paired the reads the other way and spent ten v_mov per trip re-pairing them).
constexpr int kSC = 128;             // float stride between the components of a body in that layout
__device__ __forceinline__ int staged_at(int j) { return (j >> 7) * 384 + (j & 127); }
__device__ __forceinline__ int jl_of(const unsigned long long* key, int h) { return reinterpret_cast<const int*>(key)[2 * h]; }

// Inclusive scans across the wave on DPP moves (row_shr 1/2/4/8 inside a row of 16, then row_bcast 15 and 31): six VALU
// steps, no LDS round trips (__shfl_up is a ds_bpermute per step). Lanes a move does not reach keep the identity.
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_move(int identity, int v) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROWS, 0xf, false);
}
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += dpp_move<0x111, 0xf>(0, v); v += dpp_move<0x112, 0xf>(0, v); v += dpp_move<0x114, 0xf>(0, v); v += dpp_move<0x118, 0xf>(0, v);
  v += dpp_move<0x142, 0xa>(0, v); v += dpp_move<0x143, 0xc>(0, v);
  return v;
}
__device__ __forceinline__ float wave_max(float x) {       // the maximum over the wave, in every lane
  const int ninf = __builtin_bit_cast(int, -__builtin_inff());
  auto step = [&](int moved) { x = fmaxf(x, __builtin_bit_cast(float, moved)); };
  step(dpp_move<0x111, 0xf>(ninf, __builtin_bit_cast(int, x))); step(dpp_move<0x112, 0xf>(ninf, __builtin_bit_cast(int, x)));
  step(dpp_move<0x114, 0xf>(ninf, __builtin_bit_cast(int, x))); step(dpp_move<0x118, 0xf>(ninf, __builtin_bit_cast(int, x)));
  step(dpp_move<0x142, 0xa>(ninf, __builtin_bit_cast(int, x))); step(dpp_move<0x143, 0xc>(ninf, __builtin_bit_cast(int, x)));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

template <int R, int RI>
__global__ __launch_bounds__(64 * kLW) void knn_select_staged_kernel(
    const float* __restrict__ pos, int n, int k, int loop, int64_t e_total, int64_t* edge_index, const int64_t* hint,
    int np, const nbd_knn_pq_args pq) {                // np: floats of the position block (384 per 128 bodies)
  extern __shared__ float sm[];
  const int w = wave_id();
  const int lane = threadIdx.x & 63;
  unsigned long long* key = reinterpret_cast<unsigned long long*>(sm + np) + w * kLCap;      // (d2 bits : j), one list per wave
  unsigned* dd = reinterpret_cast<unsigned*>(sm + np + 2 * kLW * kLCap) + w * kLCap;        // the d2 bits alone, for the ranking
  const int i = blockIdx.x * kLW + w;
  const int kk = min(k, n - (loop ? 0 : 1));           // > 0: checked by the launcher
  const float inf = __builtin_inff();
  KT(0)
  // the hint's entries are requested first: their round trip hides behind the staging
  int hj[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int t = 64 * q + lane;
    hj[q] = (hint && i < n && t < kk) ? (int)hint[(int64_t)i * kk + t] : -1;
  }
  // ... and so are the operands of the centre's row of the first EdgeConv layer's tables (nbd_knn_graph_hint_pq_f32)
  constexpr int kPF = 8;
  float pw[kPF], qw[kPF], px[kPF], pb = 0.f;
#pragma unroll
  for (int f = 0; f < kPF; ++f) pw[f] = qw[f] = px[f] = 0.f;
  if (pq.x) {
    const int hl = min(lane, pq.h - 1), ic = min(i, n - 1);
    pb = pq.bpq[hl];
    if (pq.f == 4 && (pq.ldx & 3) == 0 && ((reinterpret_cast<uintptr_t>(pq.x) | reinterpret_cast<uintptr_t>(pq.wpq)) & 15) == 0) {
      // the published shape (velocity + mass): three 16-byte loads per lane instead of 24 scalar ones in front of the staging
      const float4 a = reinterpret_cast<const float4*>(pq.wpq)[hl], b = reinterpret_cast<const float4*>(pq.wpq)[pq.h + hl];
      const float4 c = *reinterpret_cast<const float4*>(pq.x + (size_t)ic * pq.ldx);
      pw[0] = a.x; pw[1] = a.y; pw[2] = a.z; pw[3] = a.w;
      qw[0] = b.x; qw[1] = b.y; qw[2] = b.z; qw[3] = b.w;
      px[0] = c.x; px[1] = c.y; px[2] = c.z; px[3] = c.w;
    } else {
#pragma unroll
      for (int f = 0; f < kPF; ++f) {
        if (f < pq.f) {                                  // wave-uniform
          pw[f] = pq.wpq[(size_t)hl * pq.f + f];
          qw[f] = pq.wpq[(size_t)(pq.h + hl) * pq.f + f];
          px[f] = pq.x[(size_t)ic * pq.ldx + f];
        }
      }
    }
  }
  {
    const float4* p4 = reinterpret_cast<const float4*>(pos);
    const int ngrp = n >> 2;                           // groups of 4 bodies = 48 B = three aligned float4
    constexpr int kRounds = kStagedMaxN / 4 / (64 * kLW);
    float4 a[kRounds][3];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {                // every load is issued before the first LDS write
      const int g = min((int)threadIdx.x + 64 * kLW * r, max(ngrp - 1, 0));
#pragma unroll
      for (int c = 0; c < 3; ++c) a[r][c] = p4[3 * g + c];
    }
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int g = threadIdx.x + 64 * kLW * r;
      if (g < ngrp) {
        float* d = sm + staged_at(4 * g);              // bodies 4g .. 4g+3 sit in one chunk: three 128-bit writes
        *reinterpret_cast<float4*>(d) = make_float4(a[r][0].x, a[r][0].w, a[r][1].z, a[r][2].y);
        *reinterpret_cast<float4*>(d + kSC) = make_float4(a[r][0].y, a[r][1].x, a[r][1].w, a[r][2].z);
        *reinterpret_cast<float4*>(d + 2 * kSC) = make_float4(a[r][0].z, a[r][1].y, a[r][2].x, a[r][2].w);
      }
    }
    const int n_pad = (n + 127) & ~127;
    for (int j = 4 * ngrp + threadIdx.x; j < n_pad; j += 64 * kLW) {      // the last n % 4 bodies, then the padding (x = +inf)
      float* d = sm + staged_at(j);
      d[0] = j < n ? pos[3 * j] : inf; d[kSC] = j < n ? pos[3 * j + 1] : 0.f; d[2 * kSC] = j < n ? pos[3 * j + 2] : 0.f;
    }
  }
  KT(1)
  __syncthreads();
  KT(2)
  if (i >= n) return;                                  // no barrier below
  if (pq.x) {                                          // P_i = b + Wp x_i, Q_i = Wq x_i in the layer kernel's own fma order
    float pv = pb, qv = 0.f;
#pragma unroll
    for (int f = 0; f < kPF; ++f) { pv = __builtin_fmaf(pw[f], px[f], pv); qv = __builtin_fmaf(qw[f], px[f], qv); }
    if (lane < pq.h) {
      pq.epq[(size_t)i * pq.ldepq + lane] = exp_entry(pv);
      pq.epq[(size_t)i * pq.ldepq + pq.h + lane] = exp_entry(qv);
    }
  }
  const float xi = sm[staged_at(i)], yi = sm[staged_at(i) + kSC], zi = sm[staged_at(i) + 2 * kSC];

  auto phase_a = [&]() -> float {
    float m[R];
#pragma unroll
    for (int r = 0; r < R; ++r) m[r] = inf;
    for (int c0 = 0; c0 < n; c0 += 256) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = c0 + 64 * u + lane;
        const float* q = sm + staged_at(min(j, n - 1));
        const float dx = q[0] - xi, dy = q[kSC] - yi, dz = q[2 * kSC] - zi;
        float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        if (j >= n || (!loop && j == i)) d = inf;
#pragma unroll
        for (int r = 0; r < R; ++r) { const float lo_v = fminf(m[r], d); d = fmaxf(m[r], d); m[r] = lo_v; }
      }
    }
    float bound = inf;
    if (kk <= 64 * R) {
      int rank[R];
#pragma unroll
      for (int r = 0; r < R; ++r) rank[r] = 0;
      for (int l = 0; l < 64; ++l) {
#pragma unroll
        for (int q = 0; q < R; ++q) {
          const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m[q]), l));
#pragma unroll
          for (int r = 0; r < R; ++r)
            rank[r] += (v < m[r] || (v == m[r] && (q * 64 + l) < (r * 64 + lane))) ? 1 : 0;
        }
      }
      float cand = -inf;
#pragma unroll
      for (int r = 0; r < R; ++r) cand = rank[r] == kk - 1 ? m[r] : cand;
      const unsigned long long who = __ballot(cand != -inf);
      bound = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cand), __builtin_ctzll(who)));
    }
    return bound;
  };

  float bound = inf;
  bool hinted = false;
  if (hint) {
    float hm = -inf;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (64 * q < kk) {                               // wave-uniform
        const int j = hj[q];
        const bool ok = j >= 0 && j < n && (loop || j != i);
        const float* g = sm + staged_at(ok ? j : 0);
        const float dx = g[0] - xi, dy = g[kSC] - yi, dz = g[2 * kSC] - zi;
        const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        hm = fmaxf(hm, 64 * q + lane < kk ? (ok ? d : inf) : -inf);
      }
    }
    bound = wave_max(hm);
    hinted = bound < inf;
  }
  if (!hinted) bound = phase_a();
  KT(3)

  // scan + expand; returns the list length (entries beyond kLCap are counted, not written). bnd is finite.
  auto phase_b = [&](float bnd) -> int {
    f2v xi2 = {xi, xi}, yi2 = {yi, yi}, zi2 = {zi, zi}, bnd2 = {bnd, bnd};
    asm volatile("" : "+v"(xi2), "+v"(yi2), "+v"(zi2), "+v"(bnd2));      // VGPR operands: an SGPR source halves v_pk issue
    // candidates 128 b + 2 lane and + 1 of block b: one ds_read_b64 per component (256 B/clk; ds_read2_b32 is 128)
    auto far_pair = [&](const float* blk) -> f2v {      // bnd - d2 per candidate: sign bit set <=> d2 > bnd (exact)
#pragma clang fp contract(off)
      const f2v px = *reinterpret_cast<const f2v*>(blk + 2 * lane), py = *reinterpret_cast<const f2v*>(blk + kSC + 2 * lane),
                pz = *reinterpret_cast<const f2v*>(blk + 2 * kSC + 2 * lane);
      const f2v dx = px - xi2, dy = py - yi2, dz = pz - zi2;
      return bnd2 - ((dx * dx + dy * dy) + dz * dz);
    };
    // The decisions stay in the lane: the sign bits of its candidates are shifted into hw[] (v_alignbit), 32 candidates =
    // 16 blocks per word. No compare, no SGPR, no LDS store in the loop (parking ballot masks in LDS made the LDS pipe the
    // bound: a ds_write2_b64 costs 13 LDS-path cycles, four per trip and wave). Padding bodies sit at x = +inf: d2 = inf,
    // bnd - inf = -inf, "far".
    // (plain shifts: hipcc folds them into one v_alignbit_b32; __builtin_amdgcn_alignbit on the elements of the packed
    // result read the .x register for both elements in this toolchain)
    auto push_sign = [](unsigned a, float v) -> unsigned { return (a << 1) | (__builtin_bit_cast(unsigned, v) >> 31); };
    const int n_blocks = (n + 127) >> 7;
    constexpr int kHW = kStagedMaxN / 128 / 16;         // words of 16 blocks
    unsigned hw[kHW];
#pragma unroll
    for (int q = 0; q < kHW; ++q) {
      unsigned acc = ~0u;                               // blocks past the end read as "far"
      const int b_end = min(n_blocks, 16 * (q + 1));
      int blk = 16 * q;
      for (; blk + 2 <= b_end; blk += 2) {              // 256 candidates per trip: six 64-bit LDS reads in flight
        const f2v f0 = far_pair(sm + blk * 384), f1 = far_pair(sm + blk * 384 + 384);
        acc = push_sign(acc, f0.x);
        acc = push_sign(acc, f0.y);
        acc = push_sign(acc, f1.x);
        acc = push_sign(acc, f1.y);
      }
      if (blk < b_end) {
        const f2v f0 = far_pair(sm + blk * 384);
        acc = push_sign(acc, f0.x);
        acc = push_sign(acc, f0.y);
        ++blk;
      }
      // bit p of ~acc <-> the (2 * done - 1 - p)-th candidate of this word, done = blocks scanned into it
      const int done = max(blk - 16 * q, 0);
      hw[q] = done > 0 ? (~acc) & (done >= 16 ? ~0u : ((1u << (2 * done)) - 1u)) : 0u;
      if (!loop && done > 0 && (i >> 11) == q && lane == ((i & 127) >> 1))         // the centre itself (d2 = 0) is no candidate
        hw[q] &= ~(1u << (2 * done - 1 - (2 * ((i >> 7) - 16 * q) + (i & 1))));
    }
    KT(7)
    // expand: a lane first only NAMES its hits -- it walks its set bits (highest = lowest candidate first) and writes j to
    // its run of list slots, found by a prefix sum over the lanes' population counts; no gather in that divergent loop.
    // Then list entry h is finished by lane h, all lanes at once: one gather of the position, d2 again, key = (d2 : j).
    int count = 0;
    int* jl = reinterpret_cast<int*>(key);               // the names live in the low words of the keys-to-be
#pragma unroll
    for (int q = 0; q < kHW; ++q) {
      if (16 * q >= n_blocks) break;                    // wave-uniform
      unsigned m = hw[q];
      const int done = min(n_blocks - 16 * q, 16);
      const int pop = __builtin_popcount(m);
      const int incl = wave_incl_scan(pop);
      int off = count + incl - pop;
      count += __builtin_amdgcn_readlane(incl, 63);
      while (__ballot(m != 0)) {
        if (m) {
          const int p = 31 - __builtin_clz(m);
          m &= ~(1u << p);
          const int cand = 2 * done - 1 - p;            // candidate number inside the word: 2 * block + parity
          if (off < kLCap) jl[2 * off] = 128 * (16 * q + (cand >> 1)) + 2 * lane + (cand & 1);
          ++off;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int h = lane; h < min(count, kLCap); h += 64) {
      const int j = jl[2 * h];
      const float* g = sm + staged_at(j);
      const float dx = g[0] - xi, dy = g[kSC] - yi, dz = g[2 * kSC] - zi;
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      jl[2 * h + 1] = __builtin_bit_cast(int, d);       // high word: d2 >= +0 orders like its bit pattern
      dd[h] = __builtin_bit_cast(unsigned, d);
    }
    return count;
  };
  const int64_t base = (int64_t)i * kk;
  if (!(bound < inf)) {                                // positions that overflow fp32: not this kernel's business
    knn_insert_centre<RI>(pos, i, 0, n, kk, loop, base, e_total, edge_index);
    return;
  }
  int count = phase_b(bound);
  KT(4)
  if (hinted && count < kk) {                          // the hint was not kk distinct neighbours
    __builtin_amdgcn_wave_barrier();
    bound = phase_a();
    count = bound < inf ? phase_b(bound) : kLCap + 1;
  }
  if (count > kLCap) {                                 // cannot happen with distinct distances; massive ties can
    knn_insert_centre<RI>(pos, i, 0, n, kk, loop, base, e_total, edge_index);
    return;
  }
  // rank: entry h counts the entries with a smaller d2 -- 16 of them per trip, read as four 128-bit LDS broadcasts, one
  // 32-bit compare + one add-with-carry each. Without equal distances that is the rank; equal distances make the ranks'
  // sum fall short of count (count - 1) / 2, and only then the (d2 : j) keys are compared (ties -> lower j).
  if (lane < 16 && count + lane < ((count + 15) & ~15)) { dd[count + lane] = ~0u; key[count + lane] = ~0ull; }   // kLCap % 16 == 0
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  int rank[kLCap / 64];
  int rank_sum = 0;
#pragma unroll
  for (int r = 0; r < kLCap / 64; ++r) {
    rank[r] = 0;
    if (64 * r >= count) break;                        // wave-uniform
    const int h = 64 * r + lane;
    const unsigned dh = h < count ? dd[h] : 0u;
    int rk = 0;
    for (int t0 = 0; t0 < count; t0 += 16) {
      const uint4* dp = reinterpret_cast<const uint4*>(dd + t0);      // wave-uniform address
      const uint4 a0 = dp[0], a1 = dp[1], a2 = dp[2], a3 = dp[3];
      rk += (a0.x < dh) + (a0.y < dh) + (a0.z < dh) + (a0.w < dh) + (a1.x < dh) + (a1.y < dh) + (a1.z < dh) + (a1.w < dh) +
            (a2.x < dh) + (a2.y < dh) + (a2.z < dh) + (a2.w < dh) + (a3.x < dh) + (a3.y < dh) + (a3.z < dh) + (a3.w < dh);
    }
    rank[r] = rk;
    rank_sum += rk;
  }
  rank_sum = __builtin_amdgcn_readlane(wave_incl_scan(rank_sum), 63);
  if (rank_sum != count * (count - 1) / 2) {           // equal distances in the list: the exact order
#pragma unroll
    for (int r = 0; r < kLCap / 64; ++r) {
      if (64 * r >= count) break;
      const int h = 64 * r + lane;
      const unsigned long long kh = h < count ? key[h] : 0ull;
      int rk = 0;
      for (int t0 = 0; t0 < count; t0 += 8) {
        const ulonglong2* kp = reinterpret_cast<const ulonglong2*>(key + t0);
        const ulonglong2 k0 = kp[0], k1 = kp[1], k2 = kp[2], k3 = kp[3];
        rk += (k0.x < kh) + (k0.y < kh) + (k1.x < kh) + (k1.y < kh) + (k2.x < kh) + (k2.y < kh) + (k3.x < kh) + (k3.y < kh);
      }
      rank[r] = rk;
    }
  }
  KT(5)
#pragma unroll
  for (int r = 0; r < kLCap / 64; ++r) {
    if (64 * r >= count) break;
    const int h = 64 * r + lane;
    if (h < count && rank[r] < kk) {
      edge_index[base + rank[r]] = (int64_t)jl_of(key, h);
      edge_index[e_total + base + rank[r]] = i;
    }
  }
  KT(6)
}

// ---- radius: first `cap` hits in index order -> ELL lists nbr[n][cap], deg[n], last[n]
__global__ __launch_bounds__(64 * kWavesPerBlock) void radius_kernel(
    const float* __restrict__ pos, int n, float r2, int loop, int cap, const int* __restrict__ seg_lo,
    const int* __restrict__ seg_hi, int* __restrict__ nbr, int* __restrict__ deg, int* __restrict__ last,
    int* __restrict__ indeg) {
  const int i = blockIdx.x * kWavesPerBlock + wave_id();
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const int lo = seg_lo ? seg_lo[i] : 0, hi = seg_hi ? seg_hi[i] : n;
  const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
  int count = 0, last_j = -1;
  for (int c0 = lo; c0 < hi && count < cap; c0 += 64) {
    const int j = c0 + lane;
    bool hit = false;
    if (j < hi && (loop || j != i)) hit = dist2(pos, j, xi, yi, zi) < r2;
    const unsigned long long m = __ballot(hit);
    const int slot = count + __builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (hit && slot < cap) {
      nbr[(size_t)i * cap + slot] = j;
      if (indeg) atomicAdd(&indeg[j], 1);          // integer counts: order-independent, deterministic
    }
    const int taken = min(__builtin_popcountll(m), cap - count);
    if (taken > 0) {                                // index of the taken-th set bit = last listed j
      unsigned long long mm = m;
      for (int t = 1; t < taken; ++t) mm &= mm - 1;
      last_j = c0 + __builtin_ctzll(mm);
    }
    count += taken;
  }
  if (lane == 0) { deg[i] = count; last[i] = last_j; }
}

// ---- radius search, streaming form: lane = centre, sources broadcast from LDS (the structure of the
// all-pairs force kernel). A wave owns 64 centres and ONE slice of the source range, walks it in
// ascending index and appends each hit to its centre's slice-local list (at most `cap` kept: the final
// list is the first `cap` hits overall, so no slice ever contributes more). radius_merge_kernel then
// concatenates the slices in order. The distance test is the same rounded expression as dist2().
// bits (31 - u) for the sources j = base + u, u in [0, 32), that satisfy lo <= j < hi and j != self
__device__ __forceinline__ unsigned valid_bits(int base, int lo, int hi, int self) {
  const int a = max(lo - base, 0), b = min(hi - base, 32);           // u in [a, b)
  if (a >= b) return 0u;
  unsigned m = (0xffffffffu >> a) & (b >= 32 ? 0xffffffffu : ~(0xffffffffu >> b));
  const int su = self - base;
  if (su >= 0 && su < 32) m &= ~(0x80000000u >> su);
  return m;
}

template <bool BATCH>
__global__ __launch_bounds__(256) void radius_stream_kernel(
    const float* __restrict__ pos, int n, float r2, int loop, int cap, const int* __restrict__ seg_lo,
    const int* __restrict__ seg_hi, int n_slices, int slice_len, int* __restrict__ tmp_list,
    int* __restrict__ tmp_cnt, const int* __restrict__ run_flag) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  if (run_flag && *run_flag == 0) return;                  // cached search (below): this launch is only needed on a rebuild
  typedef float f2 __attribute__((ext_vector_type(2)));
  __shared__ f4 stage[kWavesPerBlock][64];
  const int w = wave_id(), lane = threadIdx.x & 63;
  const int gw = blockIdx.x * kWavesPerBlock + w;
  const int group = gw / n_slices, slice = gw - group * n_slices;
  if (group * 128 >= n) return;
  // two centres per lane (ia, ib = ia + 64): one LDS broadcast of a source feeds both
  const int ia = group * 128 + lane, ib = ia + 64;
  const bool va = ia < n, vb = ib < n;
  float xa = 0.f, ya = 0.f, za = 0.f, xb = 0.f, yb = 0.f, zb = 0.f;
  int loa = 0, hia = 0, lob = 0, hib = 0;
  if (va) { xa = pos[3 * ia]; ya = pos[3 * ia + 1]; za = pos[3 * ia + 2]; loa = BATCH ? seg_lo[ia] : 0; hia = BATCH ? seg_hi[ia] : n; }
  if (vb) { xb = pos[3 * ib]; yb = pos[3 * ib + 1]; zb = pos[3 * ib + 2]; lob = BATCH ? seg_lo[ib] : 0; hib = BATCH ? seg_hi[ib] : n; }
  int s0 = slice * slice_len, s1 = min(s0 + slice_len, n);
  if (BATCH) {   // the wave scans the part of its slice that some lane's batch segment reaches
    int wlo = min(va ? loa : 0x7fffffff, vb ? lob : 0x7fffffff), whi = max(va ? hia : 0, vb ? hib : 0);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { wlo = min(wlo, __shfl_xor(wlo, off)); whi = max(whi, __shfl_xor(whi, off)); }
    s0 = max(s0, __builtin_amdgcn_readfirstlane(wlo) & ~63);
    s1 = min(s1, __builtin_amdgcn_readfirstlane(whi));
  }
  f4* st = stage[w];
  int ca = 0, cb = 0;
  int* la = tmp_list + ((size_t)slice * n + (va ? ia : 0)) * cap;
  int* lb = tmp_list + ((size_t)slice * n + (vb ? ib : 0)) * cap;
  const int self_a = loop ? -1 : ia, self_b = loop ? -1 : ib;
  const f2 xab = {xa, xb}, yab = {ya, yb}, zab = {za, zb};
  // the chunk after the one being scanned is already on its way from HBM/L2 into registers
  auto fetch = [&](int c0) {
    const int j = c0 + lane;
    f4 v = {__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f};
    if (j < n) { v.x = pos[3 * j]; v.y = pos[3 * j + 1]; v.z = pos[3 * j + 2]; }
    return v;
  };
  f4 next = {};
  if (s0 < s1) next = fetch(s0);
  for (int c0 = s0; c0 < s1; c0 += 64) {
    __builtin_amdgcn_wave_barrier();
    st[lane] = next;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (c0 + 64 < s1) next = fetch(c0 + 64);
    // 32 sources at a time, branch-free: bit (31 - u) of a per-lane mask records "source u is inside the
    // radius" (mask = 2 mask + hit, one shift-or per test); validity (batch segment, slice end, self) is
    // applied to the whole mask afterwards, and only then do the lanes that hit anything append -- one
    // wave-level branch per 32 sources instead of two per source.
#pragma unroll 1
    for (int jb = 0; jb < 64 && c0 + jb < s1; jb += 32) {
      unsigned ma = 0u, mb = 0u;
#pragma unroll 8
      for (int u = 0; u < 32; ++u) {
        const f4 sp = st[jb + u];                                // wave-uniform address: LDS broadcast
        const f2 dx = f2{sp.x, sp.x} - xab, dy = f2{sp.y, sp.y} - yab, dz = f2{sp.z, sp.z} - zab;
        const f2 d2 = (dx * dx + dy * dy) + dz * dz;             // -ffp-contract=off: rounded as dist2()
        ma = (ma << 1) | (d2.x < r2 ? 1u : 0u);
        mb = (mb << 1) | (d2.y < r2 ? 1u : 0u);
      }
      const int base = c0 + jb;
      ma &= valid_bits(base, max(loa, 0), min(hia, s1), self_a);
      mb &= valid_bits(base, max(lob, 0), min(hib, s1), self_b);
      if (__builtin_amdgcn_ballot_w64((ma | mb) != 0u)) {
        while (ma) {
          const int u = __builtin_clz(ma);
          ma &= ~(0x80000000u >> u);
          if (ca < cap) la[ca] = base + u;
          ++ca;
        }
        while (mb) {
          const int u = __builtin_clz(mb);
          mb &= ~(0x80000000u >> u);
          if (cb < cap) lb[cb] = base + u;
          ++cb;
        }
      }
    }
  }
  if (va) tmp_cnt[(size_t)slice * n + ia] = min(ca, cap);
  if (vb) tmp_cnt[(size_t)slice * n + ib] = min(cb, cap);
}

// one wave per centre: exclusive scan of its slice counts (lane = slice), then slot t of the final list
// comes from the slice whose range holds t.
__global__ __launch_bounds__(64 * kWavesPerBlock) void radius_merge_kernel(
    const int* __restrict__ tmp_list, const int* __restrict__ tmp_cnt, int n, int n_slices, int cap,
    int* __restrict__ nbr, int* __restrict__ deg, int* __restrict__ last, int* __restrict__ indeg,
    const int* __restrict__ run_flag, const float* __restrict__ snap_pos = nullptr, float* __restrict__ snap_ref = nullptr,
    int* __restrict__ snap_flags = nullptr) {
  if (run_flag && *run_flag == 0) return;
  const int i = blockIdx.x * kWavesPerBlock + wave_id();
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  // cached search, rebuild step: the reference positions of this build and its bookkeeping ride along (a launch of its
  // own -- radius_snapshot_kernel -- cost ~6 us on every step of a captured rollout, rebuild or not)
  if (snap_ref && lane < 3) snap_ref[3 * i + lane] = snap_pos[3 * i + lane];
  if (snap_flags && i == 0 && lane == 0) { snap_flags[1] = 1; snap_flags[2] = n; snap_flags[3] += 1; }
  const int c = lane < n_slices ? tmp_cnt[(size_t)lane * n + i] : 0;
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  const int excl = incl - c;
  const int total = __shfl(incl, 63);
  const int d = min(total, cap);
  // lane = slice: each lane moves its own (few) entries to their final slots excl .. excl + c
  const int* mine = tmp_list + ((size_t)min(lane, n_slices - 1) * n + i) * cap;
  for (int e = 0; e < c; ++e) {
    const int t = excl + e;
    if (t >= d) break;
    const int j = mine[e];
    nbr[(size_t)i * cap + t] = j;
    if (indeg) atomicAdd(&indeg[j], 1);
    if (t == d - 1) last[i] = j;
  }
  if (lane == 0) { deg[i] = d; if (d == 0) last[i] = -1; }
}

// ---- radius search for a ROLLOUT: consecutive calls see almost the same configuration, so the O(N^2) scan is
// replaced by a re-test of a cached candidate list. The cache is the same search run once with a larger radius
// r + skin and a larger cap W ("wide" lists: the first W indices within r + skin of each centre at the reference
// positions `ref`, self included). As long as no body has moved more than 0.45 skin from `ref`, a body outside a
// centre's wide list is still farther than r, so the first `cap` hits in ascending index are found inside the
// list -- unless the list was truncated at W and holds fewer than `cap` current hits, in which case the wave
// continues with a plain scan behind the list's last index. The result is exactly that of nbd_radius_search_f32.
//   radius_disp_kernel      flags[0] = rebuild needed (never built, or some |pos - ref| too large)
//   stream + merge (above)  the wide lists; they return at once unless flags[0]
//   (merge, on a rebuild)   ref = pos, flags[1] = built, flags[3] += 1
//   radius_refresh_kernel   one wave per centre: the wide list re-tested against r
__global__ __launch_bounds__(256) void radius_disp_kernel(const float* __restrict__ pos, const float* __restrict__ ref,
                                                          int n, float thr2, int* __restrict__ flags,
                                                          int* __restrict__ indeg) {
  {   // the in-degree counters the refresh kernel adds into: zeroed here, one launch less
    const int z = blockIdx.x * 256 + threadIdx.x;
    if (indeg && z < n) indeg[z] = 0;
  }
  // flags[0] was cleared by the previous call's refresh kernel (or is zero in a fresh state): every thread that finds
  // a reason to rebuild stores 1 (benign race). A state never built / built for another n has no valid `ref`.
  if (flags[1] == 0 || flags[2] != n) { if (threadIdx.x == 0) flags[0] = 1; return; }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float dx = pos[3 * i] - ref[3 * i], dy = pos[3 * i + 1] - ref[3 * i + 1], dz = pos[3 * i + 2] - ref[3 * i + 2];
  const float d2 = (dx * dx + dy * dy) + dz * dz;
  if (!(d2 <= thr2)) flags[0] = 1;                                 // NaN counts as moved
}

__global__ __launch_bounds__(64 * kWavesPerBlock) void radius_refresh_kernel(
    const float* __restrict__ pos, int n, float r2, int loop, int cap, const int* __restrict__ wnbr,
    const int* __restrict__ wdeg, int wcap, int* __restrict__ nbr, int* __restrict__ deg, int* __restrict__ last,
    int* __restrict__ indeg, int* __restrict__ flags) {
  if (blockIdx.x == 0 && threadIdx.x == 0) flags[0] = 0;          // consumed by the launches before this one
  const int i = blockIdx.x * kWavesPerBlock + wave_id();
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
  const int wd = wdeg[i];
  int hits = 0, last_j = -1;
  auto take = [&](int j, bool ok) {                               // append the wave's hits in lane (= index) order
    const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
    const int slot = hits + __popcll(m & ((1ull << lane) - 1ull));
    if (ok && slot < cap) {
      nbr[(size_t)i * cap + slot] = j;
      if (indeg) atomicAdd(&indeg[j], 1);
    }
    const int total = hits + __popcll(m);
    if (total > hits) {                                            // the last LISTED hit so far
      const int want = min(total, cap) - 1;                        // its slot
      const bool mine = ok && slot == want;
      const unsigned long long mm = __builtin_amdgcn_ballot_w64(mine);
      if (mm) last_j = __shfl(j, (int)__builtin_ctzll(mm));
    }
    hits = total;
  };
  // 256 candidates per trip, index loads together, then positions together (see the transposing kernel below)
  for (int base = 0; base < wd && hits < cap; base += 256) {
    int j[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = base + 64 * q + lane;
      j[q] = t < wd ? wnbr[(size_t)i * wcap + t] : -1;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int jj = max(j[q], 0);
      ok[q] = j[q] >= 0 && dist2(pos, jj, xi, yi, zi) < r2 && (loop || jj != i);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (hits < cap) take(j[q], ok[q]);                     // wave-uniform: the list is full, later batches are not listed
  }
  if (hits < cap && wd >= wcap) {                                  // truncated list, not enough hits inside: scan on
    const int from = wnbr[(size_t)i * wcap + wcap - 1] + 1;
    for (int base = from; base < n && hits < cap; base += 64) {
      const int j = base + lane;
      const bool ok = j < n && dist2(pos, j, xi, yi, zi) < r2 && (loop || j != i);
      take(j, ok);
    }
  }
  if (lane == 0) { deg[i] = min(hits, cap); last[i] = last_j; }
}

// The transposed lists from the same cache: "c lists j" <=> d2(c, j) < r2, (loop or c != j) and j <= last[c] (a list
// is the first `cap` hits in ascending index), and the distance is symmetric, so every centre that can list j is in
// j's own candidate list -- or, if that list is truncated, behind its last index. One wave per node walks its
// candidates in ascending index and writes the row in order: no scatter, no atomics, no per-row sort.
__global__ __launch_bounds__(64 * kWavesPerBlock) void radius_transpose_cached_kernel(
    const float* __restrict__ pos, int n, float r2, int loop, const int* __restrict__ wnbr, const int* __restrict__ wdeg,
    int wcap, const int* __restrict__ last, const int* __restrict__ rowptr, int* __restrict__ centres) {
  const int j = blockIdx.x * kWavesPerBlock + wave_id();
  if (j >= n) return;
  const int lane = threadIdx.x & 63;
  const float xj = pos[3 * j], yj = pos[3 * j + 1], zj = pos[3 * j + 2];
  const int wd = wdeg[j];
  int* row = centres + rowptr[j];
  int cnt = 0;
  auto take = [&](int c, bool ok) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
    if (ok) row[cnt + __popcll(m & ((1ull << lane) - 1ull))] = c;
    cnt += __popcll(m);
  };
  // the candidate list, 256 entries per trip: the four index loads together, then their positions and `last` together
  // (two memory round trips per trip; one 64-entry batch at a time made it six for a 192-entry list, and with one wave
  // per node the kernel is nothing but those round trips: 28 us at N = 16 384)
  for (int base = 0; base < wd; base += 256) {
    int c[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = base + 64 * q + lane;
      c[q] = t < wd ? wnbr[(size_t)j * wcap + t] : -1;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = max(c[q], 0);
      ok[q] = c[q] >= 0 && dist2(pos, cc, xj, yj, zj) < r2 && (loop || cc != j) && j <= last[cc];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) take(c[q], ok[q]);
  }
  if (wd >= wcap) {
    // a truncated candidate list (a body of a dense clump): the row is completed by a scan behind the list's last index --
    // 256 indices per trip (their positions and `last` in flight together), and only until the row is FULL: its length is
    // known (the search counted the in-degrees that rowptr was scanned from), and rows are in ascending index, so nothing
    // can follow the last entry. (64 per trip to the end of the index range, the first form, was the kernel's whole tail.)
    const int want = rowptr[j + 1] - rowptr[j];
    for (int base = wnbr[(size_t)j * wcap + wcap - 1] + 1; base < n && cnt < want; base += 256) {
      bool ok[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = base + 64 * q + lane, cc = min(c, n - 1);
        ok[q] = c < n && dist2(pos, cc, xj, yj, zj) < r2 && (loop || cc != j) && j <= last[cc];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) take(base + 64 * q + lane, ok[q]);
    }
  }
}

// ---- transpose of the capped lists: for node j, the centres c (ascending) whose list contains j.
// "c lists j"  <=>  d2(c,j) < r2, (loop or c != j), same segment, and j <= last[c]  (lists are the
// first `cap` hits in ascending index, so membership is a comparison, not a search).
template <bool FILL>
__global__ __launch_bounds__(64 * kWavesPerBlock) void radius_transpose_kernel(
    const float* __restrict__ pos, int n, float r2, int loop, const int* __restrict__ seg_lo,
    const int* __restrict__ seg_hi, const int* __restrict__ last, const int* __restrict__ rowptr,
    int* __restrict__ indeg, int* __restrict__ centres) {
  const int j = blockIdx.x * kWavesPerBlock + wave_id();
  if (j >= n) return;
  const int lane = threadIdx.x & 63;
  const int lo = seg_lo ? seg_lo[j] : 0, hi = seg_hi ? seg_hi[j] : n;
  const float xj = pos[3 * j], yj = pos[3 * j + 1], zj = pos[3 * j + 2];
  int count = 0;
  const int base = FILL ? rowptr[j] : 0;
  for (int c0 = lo; c0 < hi; c0 += 64) {
    const int c = c0 + lane;
    bool hit = false;
    // same operand order as the forward search: d = pos[j] - pos[c] component-wise
    if (c < hi && (loop || c != j)) {
      const float dx = xj - pos[3 * c], dy = yj - pos[3 * c + 1], dz = zj - pos[3 * c + 2];
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      hit = d < r2 && j <= last[c];
    }
    const unsigned long long m = __ballot(hit);
    if (FILL && hit) centres[base + count + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = c;
    count += __builtin_popcountll(m);
  }
  if (!FILL && lane == 0) indeg[j] = count;
}

// ---- O(E) transpose of the capped lists (replaces the two O(N^2) radius_transpose passes on the hot
// path): scatter every listed (c -> j) into row j through an atomic cursor, then sort each row's
// centres ascending so the result is deterministic and identical to the scanning version.
__global__ __launch_bounds__(256) void transpose_scatter_kernel(const int* __restrict__ nbr, const int* __restrict__ deg,
                                                               int n, int cap, const int* __restrict__ rowptr,
                                                               int* __restrict__ cursor, int* __restrict__ centres) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int c = t / cap, s = t - c * cap;
  if (c >= n || s >= deg[c]) return;
  const int j = nbr[(size_t)c * cap + s];
  centres[rowptr[j] + atomicAdd(&cursor[j], 1)] = c;
}

// one wave per row, out of place: rank of x = number of row entries below it (entries are distinct
// centre indices); lane k handles entries k, k+64, ...; the comparison operand is a wave-uniform load
// (a register-resident form comparing through v_readlane measured slower: 34 us against 23 us at N = 16 384)
__global__ __launch_bounds__(64 * kWavesPerBlock) void sort_rows_kernel(const int* __restrict__ rowptr, int n,
                                                                        const int* __restrict__ unsorted,
                                                                        int* __restrict__ centres) {
  const int j = blockIdx.x * kWavesPerBlock + wave_id();
  if (j >= n) return;
  const int lane = threadIdx.x & 63;
  const int b = rowptr[j], d = rowptr[j + 1] - b;
  for (int k0 = 0; k0 < d; k0 += 64) {
    const int k = k0 + lane;
    const int x = k < d ? unsorted[b + k] : 0x7fffffff;
    int rank = 0;
    for (int k2 = 0; k2 < d; ++k2) rank += (unsorted[b + k2] < x) ? 1 : 0;
    if (k < d) centres[b + rank] = x;
  }
}

// exclusive scan of int32 counts into ptr[0..n] (single workgroup; n up to a few million). Four consecutive
// counts per thread and iteration: 4096 per trip, one wave scan and two barriers per trip (first form: 1024 per
// trip and three barriers, 19 us for n = 16 384; this one 10 us).
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(const int* __restrict__ cnt, int n,
                                                              int* __restrict__ ptr) {
  // 16 counts per thread and trip (four int4 loads in flight): n = 16 384 -- the transposed radius lists of a
  // ContinuousConv rollout step -- is ONE trip, one memory round trip, one wave scan, two barriers (round 2: 4096 per
  // trip, four dependent trips, 10-11 us in the rollout's kernel trace).
  __shared__ int wave_sum[16];
  __shared__ int carry_s;
  constexpr int PER = 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const bool vec = (reinterpret_cast<uintptr_t>(cnt) & 15) == 0 && (reinterpret_cast<uintptr_t>(ptr) & 15) == 0;
  for (int base = 0; base < n; base += 1024 * PER) {
    const int i = base + PER * threadIdx.x;
    int v[PER];
    if (vec && i + PER <= n) {
#pragma unroll
      for (int q = 0; q < PER; q += 4) {
        const int4 t = *reinterpret_cast<const int4*>(cnt + i + q);
        v[q] = t.x; v[q + 1] = t.y; v[q + 2] = t.z; v[q + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < PER; ++q) v[q] = i + q < n ? cnt[i + q] : 0;
    }
    int mine = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) mine += v[q];
    int s = mine;
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(s, off);
      if (lane >= off) s += t;
    }
    if (lane == 63) wave_sum[wave] = s;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += wave_sum[w];
    const int carry = carry_s;
    int run = carry + wave_off + s - mine;
    if (vec && i + PER <= n) {
#pragma unroll
      for (int q = 0; q < PER; q += 4) {
        int4 o;
        o.x = run; run += v[q]; o.y = run; run += v[q + 1]; o.z = run; run += v[q + 2]; o.w = run; run += v[q + 3];
        *reinterpret_cast<int4*>(ptr + i + q) = o;
      }
    } else {
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        if (i + q < n) ptr[i + q] = run;
        run += v[q];
      }
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = run;
  }
  __syncthreads();
  if (threadIdx.x == 0) ptr[n] = carry_s;
}

// ELL (nbr/deg) -> compact edge_index[2][E] grouped by centre (radius_graph's return value)
__global__ __launch_bounds__(256) void ell_to_edges_kernel(const int* __restrict__ nbr, const int* __restrict__ deg,
                                                          const int* __restrict__ ptr, int n, int cap,
                                                          int64_t e_total, int64_t* __restrict__ edge_index) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int i = t / cap, s = t - i * cap;
  if (i >= n || s >= deg[i]) return;
  const int64_t e = ptr[i] + s;
  edge_index[e] = nbr[(size_t)i * cap + s];
  edge_index[e_total + e] = i;
}

// rowptr[i] = first edge whose (ascending) target is >= i: the CSR of an edge list already grouped by ascending target
// (what knn_graph and the collation of such graphs produce) in one launch -- the torch route (bincount, cumsum, cast,
// assignment) was five launches and 0.1 ms of host time in front of every training step
__global__ __launch_bounds__(256) void rowptr_sorted_kernel(const int64_t* __restrict__ tgt, int64_t e, int n,
                                                            int* __restrict__ rowptr) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i > n) return;
  int64_t lo = 0, hi = e;                                  // first index with tgt[index] >= i
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (tgt[mid] < i) lo = mid + 1; else hi = mid; }
  rowptr[i] = (int)lo;
}

// Zero fill as a kernel of our own: hipMemsetAsync nodes captured into a hipGraph did not re-execute on
// replay on this stack (ROCm 7.2 + torch 2.10 stream capture) -- the counters below then kept their
// previous values, the second replay doubled every in-degree and the scatter ran off its buffer.
__global__ __launch_bounds__(256) void zero_i32_kernel(int* __restrict__ p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int status() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : (int)e; }

// ---- edge list -> CSR grouped by an int64 key row (the transposed adjacency the backward pass needs)
__global__ __launch_bounds__(256) void key_count_kernel(const int64_t* __restrict__ key, int64_t e, int n,
                                                        int* __restrict__ counts, int* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= e) return;
  const int64_t k = key[i];
  if (k < 0 || k >= n) { *bad = 1; return; }
  atomicAdd(&counts[k], 1);
}
// slot order inside a key's list is arbitrary here (atomic cursor); key_sort_kernel fixes it
__global__ __launch_bounds__(256) void key_scatter_kernel(const int64_t* __restrict__ key, const int64_t* __restrict__ val,
                                                          int64_t e, int n, const int* __restrict__ rowptr,
                                                          int* __restrict__ cursor, int* __restrict__ scratch) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= e) return;
  const int64_t k = key[i];
  if (k < 0 || k >= n) return;
  scratch[rowptr[k] + atomicAdd(&cursor[k], 1)] = (int)val[i];
}
// one wave per key: stable rank sort of its values (duplicates allowed: ties broken by slot), so the
// result does not depend on the order the atomics of key_scatter_kernel happened to run in
__global__ __launch_bounds__(64 * kWavesPerBlock) void key_sort_kernel(const int* __restrict__ rowptr, int n,
                                                                       const int* __restrict__ unsorted,
                                                                       int* __restrict__ sorted) {
  const int j = blockIdx.x * kWavesPerBlock + wave_id();
  if (j >= n) return;
  const int lane = threadIdx.x & 63;
  const int b = rowptr[j], d = rowptr[j + 1] - b;
  for (int k0 = 0; k0 < d; k0 += 64) {
    const int k = k0 + lane;
    const int x = k < d ? unsorted[b + k] : 0x7fffffff;
    int rank = 0;
    for (int k2 = 0; k2 < d; ++k2) {
      const int y = unsorted[b + k2];
      rank += (y < x) ? 1 : 0;
    }
    // equal values are interchangeable (same int), so place duplicates at consecutive ranks
    int dup = 0;
    for (int k2 = 0; k2 < k && k2 < d; ++k2) dup += (unsorted[b + k2] == x) ? 1 : 0;
    if (k < d) sorted[b + rank + dup] = x;
  }
}

}  // namespace

// the staged search (knn_select_staged_kernel); NBD_E_UNSUPPORTED, nothing launched, when the system is not its kind
static int launch_staged(const float* pos, int n, int k, int loop, int64_t num_edges, int64_t* edge_index,
                         const int64_t* hint, const nbd_knn_pq_args* pq, hipStream_t st) {
  static const bool staged_ok = [] { const char* e = getenv("NBD_KNN_STAGED"); return !(e && e[0] == '0'); }();
  const int kk_all = k < n - (loop ? 0 : 1) ? k : n - (loop ? 0 : 1);
  if (!staged_ok || n > kStagedMaxN || k > 200 || kk_all <= 0 || num_edges != (int64_t)n * kk_all ||
      (reinterpret_cast<uintptr_t>(pos) & 15) != 0)
    return NBD_E_UNSUPPORTED;
  const int np = (n + 127) / 128 * 384;
  const size_t lds = ((size_t)np + 3 * kLW * kLCap) * sizeof(float);      // positions + the waves' 64-bit keys + d2 words
  dim3 g2(ceil_div(n, kLW)), b2(64 * kLW);
  using Kern = void (*)(const float*, int, int, int, int64_t, int64_t*, const int64_t*, int, const nbd_knn_pq_args);
  static const Kern kerns[4] = {knn_select_staged_kernel<1, 1>, knn_select_staged_kernel<2, 1>,
                                knn_select_staged_kernel<2, 2>, knn_select_staged_kernel<4, 4>};
  static bool raised[4] = {false, false, false, false};      // > 64 KiB of dynamic LDS needs the attribute, once per kernel
  const int v = k <= 40 ? 0 : k <= 64 ? 1 : k <= 100 ? 2 : 3;
  if (!raised[v]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[v]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return status();
    raised[v] = true;
  }
  nbd_knn_pq_args none = {};
  kerns[v]<<<g2, b2, lds, st>>>(pos, n, k, loop, num_edges, edge_index, hint, np, pq ? *pq : none);
  return status();
}

extern "C" {

#ifdef NBD_KNN_TRACE
int nbd_debug_knn_trace(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_knn_trace), &buf, sizeof(buf)); }
#endif

int nbd_knn_graph_f32(const float* pos, int n, int k, int loop, const int* seg_lo, const int* seg_hi,
                      const int64_t* out_off, int64_t num_edges, int64_t* edge_index, nbd_stream_t stream) {
  return nbd_knn_graph_hint_f32(pos, n, k, loop, seg_lo, seg_hi, out_off, num_edges, edge_index, nullptr, stream);
}

int nbd_knn_graph_hint_f32(const float* pos, int n, int k, int loop, const int* seg_lo, const int* seg_hi,
                           const int64_t* out_off, int64_t num_edges, int64_t* edge_index, const int64_t* hint,
                           nbd_stream_t stream) {
  if (n < 0 || k < 0 || num_edges < 0 || (seg_lo == nullptr) != (seg_hi == nullptr)) return NBD_E_BADARG;
  // a hint is a regular list (min(k, n - 1 + loop) sources per centre, no batch segments)
  if (hint && (seg_lo || out_off || num_edges != (int64_t)n * (k < n - (loop ? 0 : 1) ? k : n - (loop ? 0 : 1)))) return NBD_E_BADARG;
  if (n == 0 || k == 0 || num_edges == 0) return 0;
  if (!pos || !edge_index) return NBD_E_BADARG;
  if (k > 256) return NBD_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ceil_div(n, kWavesPerBlock)), block(64 * kWavesPerBlock);
  // selection form by default; R = kept minima per lane (bound tightness), RI = insertion fallback width.
  // NBD_KNN_INSERTION=1 forces the insertion form (cross-check / comparison).
  static const bool force_insert = [] { const char* e = getenv("NBD_KNN_INSERTION"); return e && e[0] == '1'; }();
  // one un-segmented system that fits LDS: the staged form (NBD_KNN_STAGED=0: the form that scans L2, for comparison)
  if (!force_insert && !seg_lo && !out_off) {
    const int rc = launch_staged(pos, n, k, loop, num_edges, edge_index, hint, nullptr, st);
    if (rc != NBD_E_UNSUPPORTED) return rc;
  }
  if (!force_insert && k <= 40)
    knn_select_kernel<1, 1><<<grid, block, 0, st>>>(pos, n, k, loop, seg_lo, seg_hi, out_off, num_edges, edge_index, hint);
  else if (!force_insert && k <= 64)
    knn_select_kernel<2, 1><<<grid, block, 0, st>>>(pos, n, k, loop, seg_lo, seg_hi, out_off, num_edges, edge_index, hint);
  else if (!force_insert && k <= 100)
    knn_select_kernel<2, 2><<<grid, block, 0, st>>>(pos, n, k, loop, seg_lo, seg_hi, out_off, num_edges, edge_index, hint);
  else if (!force_insert && k <= 200)
    knn_select_kernel<4, 4><<<grid, block, 0, st>>>(pos, n, k, loop, seg_lo, seg_hi, out_off, num_edges, edge_index, hint);
  else if (k <= 64)
    knn_kernel<1><<<grid, block, 0, st>>>(pos, n, k, loop, seg_lo, seg_hi, out_off, num_edges, edge_index);
  else if (k <= 128)
    knn_kernel<2><<<grid, block, 0, st>>>(pos, n, k, loop, seg_lo, seg_hi, out_off, num_edges, edge_index);
  else
    knn_kernel<4><<<grid, block, 0, st>>>(pos, n, k, loop, seg_lo, seg_hi, out_off, num_edges, edge_index);
  return status();
}

int nbd_knn_graph_hint_pq_f32(const float* pos, int n, int k, int loop, int64_t num_edges, int64_t* edge_index,
                              const int64_t* hint, const nbd_knn_pq_args* pq, nbd_stream_t stream) {
  if (n < 0 || k < 0 || num_edges < 0 || !pq) return NBD_E_BADARG;
  if (!pq->x || !pq->wpq || !pq->bpq || !pq->epq || pq->f <= 0 || pq->h <= 0 || pq->ldx < pq->f || pq->ldepq < 2 * pq->h) return NBD_E_BADARG;
  if (pq->f > 8 || pq->h > 64) return NBD_E_UNSUPPORTED;
  if (n == 0) return 0;
  if (!pos || !edge_index) return NBD_E_BADARG;
  return launch_staged(pos, n, k, loop, num_edges, edge_index, hint, pq, (hipStream_t)stream);
}

int nbd_radius_search_f32(const float* pos, int n, float radius_sq, int loop, int max_num_neighbors,
                          const int* seg_lo, const int* seg_hi, int* nbr, int* deg, int* last, int* indeg,
                          nbd_stream_t stream) {
  if (n < 0 || max_num_neighbors < 0 || (seg_lo == nullptr) != (seg_hi == nullptr)) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!pos || !deg || !last || (max_num_neighbors > 0 && !nbr)) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (indeg) zero_i32_kernel<<<ceil_div(n, 256), 256, 0, st>>>(indeg, n);
  radius_kernel<<<ceil_div(n, kWavesPerBlock), 64 * kWavesPerBlock, 0, st>>>(
      pos, n, radius_sq, loop, max_num_neighbors, seg_lo, seg_hi, nbr, deg, last, indeg);
  return status();
}

struct RadiusPlan { int slices, slice_len; size_t ws; };
static RadiusPlan plan_radius(int n, int cap) {
  const int groups = ceil_div(n, 128);
  int s = ceil_div(8192, groups);                  // ~8 waves per SIMD chip-wide
  s = s > 64 ? 64 : s;
  const int max_s = ceil_div(n, 256);              // >= 256 sources per slice
  s = s > max_s ? max_s : s;
  s = s < 1 ? 1 : s;
  RadiusPlan p;
  p.slice_len = ceil_div(ceil_div(n, s), 64) * 64;
  p.slices = ceil_div(n, p.slice_len);
  p.ws = ((size_t)p.slices * n * cap + (size_t)p.slices * n) * sizeof(int);
  return p;
}

size_t nbd_radius_search_workspace_bytes(int n, int max_num_neighbors) {
  if (n <= 0 || max_num_neighbors <= 0) return 0;
  return plan_radius(n, max_num_neighbors).ws;
}

int nbd_radius_search_ws_f32(const float* pos, int n, float radius_sq, int loop, int max_num_neighbors,
                             const int* seg_lo, const int* seg_hi, int* nbr, int* deg, int* last, int* indeg,
                             void* workspace, size_t workspace_bytes, nbd_stream_t stream) {
  if (n < 0 || max_num_neighbors < 0 || (seg_lo == nullptr) != (seg_hi == nullptr)) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!pos || !deg || !last || (max_num_neighbors > 0 && !nbr)) return NBD_E_BADARG;
  if (max_num_neighbors == 0)
    return nbd_radius_search_f32(pos, n, radius_sq, loop, 0, seg_lo, seg_hi, nbr, deg, last, indeg, stream);
  const RadiusPlan p = plan_radius(n, max_num_neighbors);
  if (!workspace || workspace_bytes < p.ws) return NBD_E_BADARG;
  if ((long long)p.slices * n * max_num_neighbors > 0x7fffffffLL * 2) return NBD_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  int* tmp_list = static_cast<int*>(workspace);
  int* tmp_cnt = tmp_list + (size_t)p.slices * n * max_num_neighbors;
  if (indeg) zero_i32_kernel<<<ceil_div(n, 256), 256, 0, st>>>(indeg, n);
  const int waves = ceil_div(n, 128) * p.slices;
  if (seg_lo)
    radius_stream_kernel<true><<<ceil_div(waves, kWavesPerBlock), 64 * kWavesPerBlock, 0, st>>>(
        pos, n, radius_sq, loop, max_num_neighbors, seg_lo, seg_hi, p.slices, p.slice_len, tmp_list, tmp_cnt, nullptr);
  else
    radius_stream_kernel<false><<<ceil_div(waves, kWavesPerBlock), 64 * kWavesPerBlock, 0, st>>>(
        pos, n, radius_sq, loop, max_num_neighbors, seg_lo, seg_hi, p.slices, p.slice_len, tmp_list, tmp_cnt, nullptr);
  int rc = status();
  if (rc) return rc;
  radius_merge_kernel<<<ceil_div(n, kWavesPerBlock), 64 * kWavesPerBlock, 0, st>>>(
      tmp_list, tmp_cnt, n, p.slices, max_num_neighbors, nbr, deg, last, indeg, nullptr);
  return status();
}

size_t nbd_radius_cached_state_bytes(int n, int wide_cap) {
  if (n <= 0 || wide_cap <= 0) return 0;
  return 64 + ((size_t)3 * n * sizeof(float) + 63) / 64 * 64 + (size_t)2 * n * sizeof(int) + (size_t)n * wide_cap * sizeof(int);
}

size_t nbd_radius_cached_workspace_bytes(int n, int wide_cap) { return nbd_radius_search_workspace_bytes(n, wide_cap); }

int nbd_radius_cached_search_f32(const float* pos, int n, float radius_sq, float wide_radius_sq, float moved_sq, int loop,
                                 int max_num_neighbors, int wide_cap, void* state, size_t state_bytes, int* nbr, int* deg,
                                 int* last, int* indeg, void* workspace, size_t workspace_bytes, nbd_stream_t stream) {
  if (n < 0 || max_num_neighbors <= 0 || wide_cap < max_num_neighbors || !(wide_radius_sq >= radius_sq) || !(moved_sq >= 0.f))
    return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!pos || !state || !nbr || !deg || !last || (reinterpret_cast<uintptr_t>(state) & 63) != 0) return NBD_E_BADARG;
  if (state_bytes < nbd_radius_cached_state_bytes(n, wide_cap)) return NBD_E_WORKSPACE;
  const RadiusPlan p = plan_radius(n, wide_cap);
  if (!workspace || workspace_bytes < p.ws) return NBD_E_WORKSPACE;
  if ((long long)p.slices * n * wide_cap > 0x7fffffffLL * 2) return NBD_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  char* sp = static_cast<char*>(state);
  int* flags = reinterpret_cast<int*>(sp);                         // [0] rebuild now, [1] built, [2] n of the build, [3] builds so far
  float* ref = reinterpret_cast<float*>(sp + 64);
  int* wdeg = reinterpret_cast<int*>(sp + 64 + ((size_t)3 * n * sizeof(float) + 63) / 64 * 64);
  int* wlast = wdeg + n;
  int* wnbr = wlast + n;
  int* tmp_list = static_cast<int*>(workspace);
  int* tmp_cnt = tmp_list + (size_t)p.slices * n * wide_cap;
  radius_disp_kernel<<<ceil_div(n, 256), 256, 0, st>>>(pos, ref, n, moved_sq, flags, indeg);
  const int waves = ceil_div(n, 128) * p.slices;
  radius_stream_kernel<false><<<ceil_div(waves, kWavesPerBlock), 64 * kWavesPerBlock, 0, st>>>(
      pos, n, wide_radius_sq, 1, wide_cap, nullptr, nullptr, p.slices, p.slice_len, tmp_list, tmp_cnt, flags);
  radius_merge_kernel<<<ceil_div(n, kWavesPerBlock), 64 * kWavesPerBlock, 0, st>>>(
      tmp_list, tmp_cnt, n, p.slices, wide_cap, wnbr, wdeg, wlast, nullptr, flags, pos, ref, flags);
  int rc = status();
  if (rc) return rc;
  radius_refresh_kernel<<<ceil_div(n, kWavesPerBlock), 64 * kWavesPerBlock, 0, st>>>(
      pos, n, radius_sq, loop, max_num_neighbors, wnbr, wdeg, wide_cap, nbr, deg, last, indeg, flags);
  return status();
}

int nbd_radius_cached_transpose_f32(const float* pos, int n, float radius_sq, int loop, int wide_cap, const void* state,
                                    size_t state_bytes, const int* last, const int* rowptr, int* centres,
                                    nbd_stream_t stream) {
  if (n < 0 || wide_cap <= 0) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!pos || !state || !last || !rowptr || !centres || (reinterpret_cast<uintptr_t>(state) & 63) != 0) return NBD_E_BADARG;
  if (state_bytes < nbd_radius_cached_state_bytes(n, wide_cap)) return NBD_E_WORKSPACE;
  const char* sp = static_cast<const char*>(state);
  const int* wdeg = reinterpret_cast<const int*>(sp + 64 + ((size_t)3 * n * sizeof(float) + 63) / 64 * 64);
  const int* wnbr = wdeg + 2 * (size_t)n;
  radius_transpose_cached_kernel<<<ceil_div(n, kWavesPerBlock), 64 * kWavesPerBlock, 0, (hipStream_t)stream>>>(
      pos, n, radius_sq, loop, wnbr, wdeg, wide_cap, last, rowptr, centres);
  return status();
}

}  // extern "C"
namespace {
// radius_graph(loop = False) as torch_cluster 1.6.3 computes it: the search ran WITH self as a candidate and a cap of
// max_num_neighbors + 1 ("first cap hits in index order, self included"); now row == col is dropped. One thread per
// centre (lists hold <= 33 entries here); the in-degree the search counted for the self entry is taken back.
__global__ __launch_bounds__(256) void radius_drop_self_kernel(int* __restrict__ nbr, int* __restrict__ deg,
                                                               int* __restrict__ last, int* __restrict__ indeg, int n,
                                                               int cap) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int* row = nbr + (size_t)i * cap;
  const int d = deg[i];
  int at = -1;
  for (int t = 0; t < d; ++t)
    if (row[t] == i) { at = t; break; }
  if (at < 0) return;                       // >= cap lower-indexed hits: self never made the list, all cap entries stay
  for (int t = at; t + 1 < d; ++t) row[t] = row[t + 1];
  deg[i] = d - 1;
  last[i] = d > 1 ? row[d - 2] : -1;
  if (indeg) indeg[i] -= 1;                 // only this thread touches indeg[i] here (the search's atomics are done)
}
}  // namespace
extern "C" {

int nbd_radius_drop_self_i32(int* nbr, int* deg, int* last, int* indeg, int n, int cap, nbd_stream_t stream) {
  if (n < 0 || cap <= 0) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!nbr || !deg || !last) return NBD_E_BADARG;
  radius_drop_self_kernel<<<ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(nbr, deg, last, indeg, n, cap);
  return status();
}

int nbd_radius_transpose_lists(const int* nbr, const int* deg, int n, int cap, const int* rowptr, int* cursor,
                               int* scratch, int* centres, nbd_stream_t stream) {
  if (n < 0 || cap < 0) return NBD_E_BADARG;
  if (n == 0 || cap == 0) return 0;
  if (!nbr || !deg || !rowptr || !cursor || !scratch || !centres) return NBD_E_BADARG;
  const long long total = (long long)n * cap;
  if (total > 0x7fffffffLL) return NBD_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  zero_i32_kernel<<<ceil_div(n, 256), 256, 0, st>>>(cursor, n);
  transpose_scatter_kernel<<<ceil_div((int)total, 256), 256, 0, st>>>(nbr, deg, n, cap, rowptr, cursor, scratch);
  int rc = status();
  if (rc) return rc;
  sort_rows_kernel<<<ceil_div(n, kWavesPerBlock), 64 * kWavesPerBlock, 0, st>>>(rowptr, n, scratch, centres);
  return status();
}

int nbd_radius_transpose_count_f32(const float* pos, int n, float radius_sq, int loop, const int* seg_lo,
                                   const int* seg_hi, const int* last, int* indeg, nbd_stream_t stream) {
  if (n < 0 || (seg_lo == nullptr) != (seg_hi == nullptr)) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!pos || !last || !indeg) return NBD_E_BADARG;
  radius_transpose_kernel<false><<<ceil_div(n, kWavesPerBlock), 64 * kWavesPerBlock, 0, (hipStream_t)stream>>>(
      pos, n, radius_sq, loop, seg_lo, seg_hi, last, nullptr, indeg, nullptr);
  return status();
}

int nbd_radius_transpose_fill_f32(const float* pos, int n, float radius_sq, int loop, const int* seg_lo,
                                  const int* seg_hi, const int* last, const int* rowptr, int* centres,
                                  nbd_stream_t stream) {
  if (n < 0 || (seg_lo == nullptr) != (seg_hi == nullptr)) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!pos || !last || !rowptr || !centres) return NBD_E_BADARG;
  radius_transpose_kernel<true><<<ceil_div(n, kWavesPerBlock), 64 * kWavesPerBlock, 0, (hipStream_t)stream>>>(
      pos, n, radius_sq, loop, seg_lo, seg_hi, last, rowptr, nullptr, centres);
  return status();
}

int nbd_csr_by_key_i64(const int64_t* key, const int64_t* val, int64_t n_edges, int n, int* rowptr, int* cursor,
                       int* scratch, int* out, int* bad_flag, nbd_stream_t stream) {
  if (n < 0 || n_edges < 0) return NBD_E_BADARG;
  if (!rowptr || !bad_flag) return NBD_E_BADARG;
  if (n_edges > 0x7fffffffLL) return NBD_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (n_edges > 0 && (!key || !val || !cursor || !scratch || !out)) return NBD_E_BADARG;
  zero_i32_kernel<<<1, 256, 0, st>>>(bad_flag, 1);
  if (n > 0) {
    if (!cursor) return NBD_E_BADARG;
    zero_i32_kernel<<<ceil_div(n, 256), 256, 0, st>>>(cursor, n);
  }
  const unsigned eb = (unsigned)((n_edges + 255) / 256);
  if (n_edges > 0) key_count_kernel<<<eb, 256, 0, st>>>(key, n_edges, n, cursor, bad_flag);
  exclusive_scan_kernel<<<1, 1024, 0, st>>>(cursor, n, rowptr);
  if (n_edges == 0 || n == 0) return status();
  zero_i32_kernel<<<ceil_div(n, 256), 256, 0, st>>>(cursor, n);
  key_scatter_kernel<<<eb, 256, 0, st>>>(key, val, n_edges, n, rowptr, cursor, scratch);
  key_sort_kernel<<<ceil_div(n, kWavesPerBlock), 64 * kWavesPerBlock, 0, st>>>(rowptr, n, scratch, out);
  return status();
}

int nbd_exclusive_scan_i32(const int* counts, int n, int* ptr, nbd_stream_t stream) {
  if (n < 0 || !ptr || (n > 0 && !counts)) return NBD_E_BADARG;
  exclusive_scan_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(counts, n, ptr);
  return status();
}

int nbd_rowptr_sorted_i64(const int64_t* tgt, int64_t n_edges, int n, int* rowptr, nbd_stream_t stream) {
  if (n < 0 || n_edges < 0 || n_edges > 0x7fffffffLL || !rowptr || (n_edges > 0 && !tgt)) return NBD_E_BADARG;
  rowptr_sorted_kernel<<<ceil_div(n + 1, 256), 256, 0, (hipStream_t)stream>>>(tgt, n_edges, n, rowptr);
  return status();
}

int nbd_ell_to_edge_index(const int* nbr, const int* deg, const int* ptr, int n, int cap, int64_t num_edges,
                          int64_t* edge_index, nbd_stream_t stream) {
  if (n < 0 || cap < 0 || num_edges < 0) return NBD_E_BADARG;
  if (n == 0 || cap == 0 || num_edges == 0) return 0;
  if (!nbr || !deg || !ptr || !edge_index) return NBD_E_BADARG;
  const long long total = (long long)n * cap;
  if (total > 0x7fffffffLL) return NBD_E_UNSUPPORTED;
  ell_to_edges_kernel<<<ceil_div((int)total, 256), 256, 0, (hipStream_t)stream>>>(nbr, deg, ptr, n, cap, num_edges,
                                                                                 edge_index);
  return status();
}

}  // extern "C"
