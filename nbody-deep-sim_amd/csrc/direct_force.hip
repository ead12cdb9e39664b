// direct_force.hip -- all-pairs softened gravity + leapfrog/Euler updates for gfx950 (MI355X).
//
// Replaces the arithmetic of the reference's BaseSimulator.compute_accelerations
// (src/galaxify/simulation.py:71-89), LeapFrogSimulator.step (:153-170),
// EulerSimulator.step (:173-187) and compute_energies (:91-115). C-ABI: include/nbd.h.
//
// K1 design:
//   * the kernel is VALU-issue bound (16 FMA-slot equivalents per pair, v_rsq_f32 = 4 of them),
//     so everything is arranged to keep the four SIMDs of a CU issuing packed fp32 math:
//     each lane owns TWO targets held as float2 register pairs, so one broadcast source feeds
//     v_pk_add/v_pk_fma/v_pk_mul on both; two register shapes of the same code are built: eight
//     sources in flight per wave (90 VGPRs, 5 waves/SIMD: best issue rate, the default) and four
//     (<=64 VGPRs, 8 waves/SIMD: more, smaller workgroup slots for launches with few targets);
//   * every wave is autonomous: it streams its own slice of the source array in 64-body
//     (1 KiB) chunks HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: coalesced float4
//     loads, no VGPR staging), double-buffered behind a counted vmcnt, and reads the chunk
//     back with wave-uniform ds_read_b128 (LDS broadcast). No workgroup barrier in the loop;
//   * the 4 waves of a workgroup share the same 128 targets and split the sources (J-split);
//     their partial forces are reduced through LDS (wavefront-level partials -> one coalesced
//     store per workgroup). A second J-split across workgroups (gridDim.y slabs) fills the
//     256 CUs when there are few targets; slabs are summed in fixed order by the finishing
//     kernel, so results are bit-reproducible (no float atomics).
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdint.h>

#include "../../include/nbd.h"

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

namespace {

constexpr int kWaves = 4;                  // waves per workgroup (J-split inside the workgroup)
constexpr int kTgtPerLane = 2;             // packed pair of targets per lane
constexpr int kTgtPerWG = 64 * kTgtPerLane;  // 128 targets per workgroup
constexpr int kChunk = NBD_SRC_PAD;        // 64 sources = one 1-KiB LDS-DMA piece
constexpr int kMaxSlabs = 64;
// below this softening^2 the cube of rsq overflows fp32 for coincident bodies (and the i==j
// term), so the index-masked kernel is used (fill_diagonal_ semantics, simulation.py:85)
constexpr float kEps2Masked = 1e-24f;

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Which sources a launch walks: the 64-source chunks of `src` minus a run of skipped physical chunks,
// with an element-wise exclusion in the (at most two) chunks that hold a partial piece of the excluded
// index range. The un-sharded force uses the trivial view; the range-sharded step (nbd_shard_*) walks
// "all bodies except my own [lo, hi)" with it while its own block runs from a separate launch.
struct SrcView {
  int n_src;           // real entries; the padding behind them is zero-mass
  int n_chunks;        // logical chunks walked (physical chunks minus the skipped run)
  int cpw_q, cpw_r;    // balanced split: every wave walks cpw_q chunks, the first cpw_r waves one more
  int skip_c0, skip_cn;  // physical chunks [skip_c0, skip_c0 + skip_cn) are not visited
  int ex_lo, ex_hi;    // source indices [ex_lo, ex_hi) contribute nothing (checked only where needed)
  int edge0, edge1;    // physical chunks that straddle ex_lo / ex_hi (-1: none): these take the masked path
  int tail;            // uniform-mass kernels only: the physical chunk that holds padding behind n_src (-1: none); it takes
                       // the masked path too (without the per-source mass factor a padding entry is not a zero any more)
};

// One source against the lane's two targets. 12 packed ops + 2 v_rsq_f32 (UNI: 11, see accel_kernel).
template <bool MASKED, bool UNI = false>
__device__ __forceinline__ void interact(const f4 p, const f2 xi, const f2 yi, const f2 zi,
                                         const f2 e2, f2& ax, f2& ay, f2& az, int j, int i0,
                                         int i1, const SrcView& sv) {
  const f2 dx = f2{p.x, p.x} - xi, dy = f2{p.y, p.y} - yi, dz = f2{p.z, p.z} - zi;  // r_j - r_i
  f2 r2 = __builtin_elementwise_fma(dx, dx, e2);
  r2 = __builtin_elementwise_fma(dy, dy, r2);
  r2 = __builtin_elementwise_fma(dz, dz, r2);
  f2 s = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
  if (MASKED) {  // exact fill_diagonal_(0): only j == i is dropped; padding and the excluded range too
    const bool live = j < sv.n_src && (unsigned)(j - sv.ex_lo) >= (unsigned)(sv.ex_hi - sv.ex_lo);
    s.x = (live && j != i0) ? s.x : 0.0f;
    s.y = (live && j != i1) ? s.y : 0.0f;
  }
  // w = m_j * s^3 with m_j broadcast from the HIGH half of the {z,m} register pair; hipcc does not
  // fold that splat into op_sel by itself (it inserts a v_mov), hence the one asm line. The asm
  // multiply takes s^3 (an ordinary VALU result), never s itself: gfx950 needs a wait state between
  // a transcendental result and its VALU consumer, and hipcc pads that only for instructions it
  // can see (an asm consumer right behind v_rsq_f32/v_rcp_f32 reads a stale register).
  const f2 zm = {p.z, p.w};
  const f2 s3 = (s * s) * s;
  f2 w;  // m_j (r^2 + eps^2)^(-3/2)
  if (UNI) w = s3;               // equal masses: the common factor is applied once, to the finished sum
  else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(w) : "v"(zm), "v"(s3));
  ax = __builtin_elementwise_fma(w, dx, ax);
  ay = __builtin_elementwise_fma(w, dy, ay);
  az = __builtin_elementwise_fma(w, dz, az);
}

// KU sources at once for the un-masked path: same arithmetic as interact(), with the 2*KU v_rsq_f32
// issued back to back (__builtin_amdgcn_sched_group_barrier on the TRANS class). Switching between the
// quarter-rate transcendental unit and the packed-math stream costs issue cycles on gfx950 (3 fma : 1
// rsq mixes run ~10 % under the sum of their parts, tools/ubench_valu.hip), so the switches are
// batched; the rsq stays a compiler builtin so that hipcc fills the transcendental -> VALU wait state
// with independent work instead of the s_nop it must put behind an opaque asm block. Measured
// (tools/k1_variants.hip, N = 65 536): KU = 8 at 90 VGPRs / 5 waves per SIMD beats KU = 4 at 58 VGPRs /
// 8 waves (1.004 vs 1.010 ms) and an inline-asm rsq block (1.021 ms).
template <int KU, bool UNI = false>
__device__ __forceinline__ void interact_block(const f4* __restrict__ buf, const f2 xi, const f2 yi, const f2 zi,
                                               const f2 e2, f2& ax, f2& ay, f2& az) {
  f4 p[KU];
  f2 dx[KU], dy[KU], dz[KU], s[KU];
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    p[u] = buf[u];
    if (UNI) asm("" : "+v"(p[u]));      // keep the source a whole 4-register tuple: with the mass unused hipcc loads 96 bits
                                        // and then copies z out of its odd register to splat it (a v_mov per source)
    dx[u] = f2{p[u].x, p[u].x} - xi; dy[u] = f2{p[u].y, p[u].y} - yi; dz[u] = f2{p[u].z, p[u].z} - zi;
    f2 r2 = __builtin_elementwise_fma(dx[u], dx[u], e2);
    r2 = __builtin_elementwise_fma(dy[u], dy[u], r2);
    s[u] = __builtin_elementwise_fma(dz[u], dz[u], r2);
  }
#pragma unroll
  for (int u = 0; u < KU; ++u) s[u] = f2{__builtin_amdgcn_rsqf(s[u].x), __builtin_amdgcn_rsqf(s[u].y)};
  __builtin_amdgcn_sched_group_barrier(0x400, 2 * KU, 0);      // 0x400 = TRANS: keep the rsq's together
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    const f2 zm = {p[u].z, p[u].w};
    const f2 s3 = (s[u] * s[u]) * s[u];          // compiler-visible consumers of the rsq results (hazard-padded)
    f2 w;
    if (UNI) w = s3;
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(w) : "v"(zm), "v"(s3));
    ax = __builtin_elementwise_fma(w, dx[u], ax);
    ay = __builtin_elementwise_fma(w, dy[u], ay);
    az = __builtin_elementwise_fma(w, dz[u], az);
  }
}

// grid = (target groups of 128, slabs); block = 256.
// Wave jw = blockIdx.y*4 + w handles the logical source chunks [jw*q + min(jw, r), ... + q (+1 if jw < r)):
// all chunks are spread over all waves to within one chunk (no idle tail waves).
// KU = 8: 90 VGPRs, 5 waves/SIMD. KU = 4: capped at 64 VGPRs, 8 waves/SIMD.
// UNI (round 3): every body has the SAME mass -- the published configurations (Plummer, m = 1 / N) among them. The mass
// then factors out of the whole sum, a = (G m) sum_j d_ij s_ij^3: the per-pair multiply by m_j goes (11 packed ops +
// 2 v_rsq_f32 per source and pair of targets instead of 12 + 2: 60 issue cycles per 128 pairs instead of 64), the
// differences stay the exact fp32 subtractions of the reference, and the only change in rounding is ONE multiplication
// of the finished sum instead of one per term. (Folding unequal masses into the coordinates -- c_j = m_j^(-1/2) scales
// source j so that rsq^3 carries m_j -- also reaches 11 ops and was built first; it gives up the exact difference
// c_j r_j - c_j r_i rounds the product c_j r_j -- and a close pair amplifies that half-ulp shift of the source:
// 1.8e-4 on one row of the Plummer N = 1000 golden, against the 1e-5 bar. Rejected; measured +2.9 %.) Padding entries
// are no zeros without the mass factor: the chunk that holds them (sv.tail) takes the masked path.
template <bool MASKED, int KU, bool UNI = false>
__global__ __launch_bounds__(64 * kWaves, KU == 4 ? 8 : 5) void accel_kernel(
    const f4* __restrict__ src, const SrcView sv, const f4* __restrict__ tgt,
    int n_tgt, int tgt_off, float eps2, float scale, float* __restrict__ out) {
  // [wave][buffer][64] staging + [wave][6][64] partials, ONE object (keeps hipcc's waits sane)
  __shared__ f4 lds[kWaves * 2 * kChunk + kWaves * 6 * 64 / 4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t_base = blockIdx.x * kTgtPerWG;
  const int i0 = t_base + lane, i1 = t_base + 64 + lane;
  const f4 t0 = tgt[min(i0, n_tgt - 1)], t1 = tgt[min(i1, n_tgt - 1)];
  const f2 xi = {t0.x, t1.x}, yi = {t0.y, t1.y}, zi = {t0.z, t1.z};
  f2 ax = {0.f, 0.f}, ay = {0.f, 0.f}, az = {0.f, 0.f};
  f2 e2 = {eps2, eps2};
  asm volatile("" : "+v"(e2));  // keep eps^2 in VGPRs: an SGPR operand halves v_pk_fma issue

  const int jw = blockIdx.y * kWaves + wave;
  const int c_begin = jw * sv.cpw_q + min(jw, sv.cpw_r), c_end = c_begin + sv.cpw_q + (jw < sv.cpw_r ? 1 : 0);
  f4* stage = &lds[wave * 2 * kChunk];
  const f4* s_lane = src + lane;
  // logical -> physical chunk: hop over the skipped run
  auto phys = [&](int c) { return c + (c >= sv.skip_c0 ? sv.skip_cn : 0); };
  if (c_begin < c_end)
    __builtin_amdgcn_global_load_lds(GPTR(s_lane + (size_t)phys(c_begin) * kChunk), LPTR(stage), 16, 0, 0);
  for (int c = c_begin; c < c_end; ++c) {
    const int b = (c - c_begin) & 1;
    if (c + 1 < c_end) {
      __builtin_amdgcn_global_load_lds(GPTR(s_lane + (size_t)phys(c + 1) * kChunk),
                                       LPTR(stage + (b ^ 1) * kChunk), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");  // chunk c has landed, c+1 in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const f4* buf = stage + b * kChunk;
    const int pc = phys(c);
    const int j0 = pc * kChunk;
    if (MASKED || pc == sv.edge0 || pc == sv.edge1 || (UNI && pc == sv.tail)) {
#pragma unroll 4
      for (int j = 0; j < kChunk; ++j)
        interact<true, UNI>(buf[j], xi, yi, zi, e2, ax, ay, az, j0 + j, tgt_off + i0, tgt_off + i1, sv);
    } else {
#pragma unroll 1
      for (int j = 0; j < kChunk; j += KU) interact_block<KU, UNI>(buf + j, xi, yi, zi, e2, ax, ay, az);
    }
  }

  // wavefront partials -> LDS -> one coalesced (128 x 3) store per workgroup
  float* red = reinterpret_cast<float*>(&lds[kWaves * 2 * kChunk]);  // [wave][comp*2+half][64]
  float* mine = red + wave * 6 * 64;
  mine[0 * 64 + lane] = ax.x; mine[1 * 64 + lane] = ax.y;
  mine[2 * 64 + lane] = ay.x; mine[3 * 64 + lane] = ay.y;
  mine[4 * 64 + lane] = az.x; mine[5 * 64 + lane] = az.y;
  __syncthreads();
  float* dst = out + ((size_t)blockIdx.y * n_tgt + t_base) * 3;
  const int n_valid = min(kTgtPerWG, n_tgt - t_base) * 3;
  for (int o = threadIdx.x; o < n_valid; o += 64 * kWaves) {
    const int lt = o / 3, comp = o - lt * 3;
    const int idx = (comp * 2 + (lt >> 6)) * 64 + (lt & 63);
    float sum = red[idx];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) sum += red[w * 6 * 64 + idx];
    dst[o] = __fmul_rn(scale, sum);
  }
}

// acc = g * (slab_0 + slab_1 + ...), optional fused kick v += c * acc (simulation.py:88,170).
// Block = 4 waves on 64 consecutive outputs: wave w sums slabs w, w+4, w+8, ... (coalesced 256-B loads, all
// in flight), the four partial sums are combined as (p0 + p1) + (p2 + p3) through LDS -- a fixed association,
// so the result is bit-reproducible. One thread per output summing every slab serially took 9 us for the
// 28 slabs x 24 576 outputs of a sharded rank (96 workgroups, one dependent chain each); this form 3 us.
__global__ __launch_bounds__(256) void finish_kernel(const float* __restrict__ slabs, int n_slabs,
                                                     size_t slab_stride, float g, float* __restrict__ acc,
                                                     float* __restrict__ vel, float c_kick, int n3) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  float sum = 0.f;
  if (i < n3)
    for (int s = w; s < n_slabs; s += 4) sum += slabs[s * slab_stride + i];
  part[w][lane] = sum;
  __syncthreads();
  if (w != 0 || i >= n3) return;
  const float a = __fmul_rn(g, (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
  acc[i] = a;
  if (vel) vel[i] = __fadd_rn(vel[i], __fmul_rn(c_kick, a));
}

__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ pos,
                                                   const float* __restrict__ mass, int n, int n_pad,
                                                   f4* __restrict__ posm) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pad) return;
  f4 v = {0.f, 0.f, 0.f, 0.f};
  if (i < n) v = f4{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], mass[i]};
  posm[i] = v;
}

// v += ck*a ; x += cd*v ; posm = {x, m}. mul and add round separately (torch eager order).
__global__ __launch_bounds__(256) void kick_drift_kernel(float* __restrict__ pos, float* __restrict__ vel,
                                                         const float* __restrict__ acc,
                                                         const float* __restrict__ mass, int n, int n_pad,
                                                         float ck, float cd, f4* __restrict__ posm) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pad) return;
  f4 pm = {0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    float x[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float v = vel[3 * i + k];
      if (acc) { v = __fadd_rn(v, __fmul_rn(ck, acc[3 * i + k])); vel[3 * i + k] = v; }
      x[k] = __fadd_rn(pos[3 * i + k], __fmul_rn(cd, v));
      pos[3 * i + k] = x[k];
    }
    pm = f4{x[0], x[1], x[2], mass ? mass[i] : 0.f};
  }
  if (posm) posm[i] = pm;
}

// zero fill by kernel, not hipMemsetAsync: memset nodes captured into a hipGraph were observed not to
// re-execute on replay on this stack (see csrc/graph.hip), and every entry point here must be capturable
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

// state snapshot of one step into a device ring slot: out = [pos | vel | acc], each (n,3) -- BaseSimulator.run's
// per-step clones (simulation.py:135-139) as ONE launch that a captured chunk of steps can contain
__global__ __launch_bounds__(256) void snapshot_kernel(const float* __restrict__ pos, const float* __restrict__ vel,
                                                       const float* __restrict__ acc, int n3, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n3) return;
  out[i] = pos[i]; out[n3 + i] = vel[i]; out[2 * n3 + i] = acc[i];
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                   float c, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = __fadd_rn(y[i], __fmul_rn(c, x[i]));
}

// ---- energies (simulation.py:91-115). U = sum_{i<j} -G m_i m_j / (|r_ij| + eps), K = sum 0.5 m v^2.
// Same streaming structure as K1 (two targets per lane in packed registers, wave-private LDS-DMA
// chunks, J-split over waves and slabs) restricted to the upper triangle: a target group only
// walks the source chunks at or above its own first index; the (at most three) chunks that
// straddle the diagonal take the masked path (j > i), the rest run mask-free. Per pair
// 8 packed ops + 1 v_mov + 2 v_sqrt_f32 + 2 v_rcp_f32. fp32 per-lane partial sums, fp64 across lanes/blocks.
template <bool MASKED>
__device__ __forceinline__ void energy_pair(const f4 p, const f2 xi, const f2 yi, const f2 zi, const f2 soft,
                                            f2& u, int j, int i0, int i1, int n) {
  const f2 dx = f2{p.x, p.x} - xi, dy = f2{p.y, p.y} - yi, dz = f2{p.z, p.z} - zi;
  f2 d2 = dx * dx;
  d2 = __builtin_elementwise_fma(dy, dy, d2);
  d2 = __builtin_elementwise_fma(dz, dz, d2);
  const f2 den = f2{__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)} + soft;   // |r| + eps (:105)
  const f2 inv = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  f2 t = f2{p.w, p.w} * inv;                  // m_j / den (plain C: the consumer of v_rcp_f32 must be
                                              // visible to hipcc's hazard padding -- see interact())
  if (MASKED) {                               // triu(1): strictly above the diagonal (:113)
    t.x = (j > i0 && j < n) ? t.x : 0.f;
    t.y = (j > i1 && j < n) ? t.y : 0.f;
  }
  u += t;
}

__global__ __launch_bounds__(64 * kWaves) void energy_kernel(const f4* __restrict__ posm, int n, int n_chunks,
                                                             float soft_, int all_masked,
                                                             double* __restrict__ partial_u) {
  __shared__ f4 lds[kWaves * 2 * kChunk + 8];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t_base = blockIdx.x * kTgtPerWG;
  const int i0 = t_base + lane, i1 = t_base + 64 + lane;
  const f4 t0 = posm[min(i0, n - 1)], t1 = posm[min(i1, n - 1)];
  const f2 xi = {t0.x, t1.x}, yi = {t0.y, t1.y}, zi = {t0.z, t1.z};
  f2 u = {0.f, 0.f};
  f2 soft = {soft_, soft_};
  asm volatile("" : "+v"(soft));
  // this block's share of the chunks [first chunk of the group, n_chunks), split over slabs x waves
  const int c_lo = t_base / kChunk;
  const int span = n_chunks - c_lo;
  const int parts = gridDim.y * kWaves;
  const int cpw = (span + parts - 1) / parts;
  const int jw = blockIdx.y * kWaves + wave;
  const int c_begin = min(c_lo + jw * cpw, n_chunks), c_end = min(c_begin + cpw, n_chunks);
  const int c_diag_end = (t_base + kTgtPerWG + kChunk - 1) / kChunk;      // chunks below this touch j <= i
  f4* stage = &lds[wave * 2 * kChunk];
  const f4* s_lane = posm + lane;
  if (c_begin < c_end)
    __builtin_amdgcn_global_load_lds(GPTR(s_lane + (size_t)c_begin * kChunk), LPTR(stage), 16, 0, 0);
  for (int c = c_begin; c < c_end; ++c) {
    const int b = (c - c_begin) & 1;
    if (c + 1 < c_end) {
      __builtin_amdgcn_global_load_lds(GPTR(s_lane + (size_t)(c + 1) * kChunk), LPTR(stage + (b ^ 1) * kChunk), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const f4* buf = stage + b * kChunk;
    const int j0 = c * kChunk;
    if (all_masked || c < c_diag_end || c == n_chunks - 1) {     // diagonal chunks and the padded tail
#pragma unroll 4
      for (int j = 0; j < kChunk; ++j) energy_pair<true>(buf[j], xi, yi, zi, soft, u, j0 + j, i0, i1, n);
    } else {
#pragma unroll 4
      for (int j = 0; j < kChunk; ++j) energy_pair<false>(buf[j], xi, yi, zi, soft, u, j0 + j, i0, i1, n);
    }
  }
  // U contribution of this wave: sum_i m_i u_i  (the -G factor is applied by the final kernel)
  double acc = 0.0;
  if (i0 < n) acc += (double)t0.w * (double)u.x;
  if (i1 < n) acc += (double)t1.w * (double)u.y;
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  double* red = reinterpret_cast<double*>(&lds[kWaves * 2 * kChunk]);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    partial_u[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void kinetic_kernel(const f4* __restrict__ posm, const float* __restrict__ vel,
                                                      int n, double* __restrict__ partial_k) {
  __shared__ double red[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double k = 0.0;
  if (i < n) {
    const float vx = vel[3 * i], vy = vel[3 * i + 1], vz = vel[3 * i + 2];
    const float v2 = __fadd_rn(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)), __fmul_rn(vz, vz));
    k = (double)__fmul_rn(__fmul_rn(0.5f, posm[i].w), v2);                // 0.5 * m * |v|^2 (:100)
  }
  for (int off = 32; off > 0; off >>= 1) k += __shfl_down(k, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = k;
  __syncthreads();
  if (threadIdx.x == 0) partial_k[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void energy_final_kernel(const double* __restrict__ pu, int nu,
                                                           const double* __restrict__ pk, int nk, float g,
                                                           double* __restrict__ out) {
  __shared__ double ru[4], rk[4];
  double u = 0.0, k = 0.0;
  for (int b = threadIdx.x; b < nu; b += 256) u += pu[b];
  for (int b = threadIdx.x; b < nk; b += 256) k += pk[b];
  for (int off = 32; off > 0; off >>= 1) { u += __shfl_down(u, off); k += __shfl_down(k, off); }
  if ((threadIdx.x & 63) == 0) { ru[threadIdx.x >> 6] = u; rk[threadIdx.x >> 6] = k; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = -(double)g * ((ru[0] + ru[1]) + (ru[2] + ru[3]));
    out[1] = (rk[0] + rk[1]) + (rk[2] + rk[3]);
  }
}

inline int energy_slabs(int groups) {
  int s = (4096 + groups - 1) / groups;        // ~2 residency rounds; the triangle is balanced dynamically
  return s < 1 ? 1 : (s > 32 ? 32 : s);
}

struct AccelPlan { int groups, slabs, n_chunks, cpw, variant; };   // cpw = the LARGEST chunk count of a wave

// Launch geometry for `n_chunks` logical source chunks against n_tgt targets.
//
// A workgroup is 4 waves (one per SIMD of its CU) on 128 targets; the kernel is VALU-issue bound, so a
// launch costs (to first order) the largest number of chunk-times any SIMD is handed:
//     cost(slabs) = [workgroups per CU] x [chunks per wave]       (workgroups go to the CUs round-robin)
// divided by the issue efficiency at that many waves per SIMD and plus a per-workgroup prologue/epilogue
// term. Fitted to the hardware sweep of every slab count (nbd_accel_tuned_f32, tools/sweep_accel_plan.py,
// profiles/r02_plan_sweep*.jsonl): e.g. 8192 targets x 57 344 sources (the remote block of one of 8 ranks):
// 2/3/4/>=5 workgroups per CU at equal chunk totals ran at 0.935/0.975/0.99/1.0 of the best rate, a slab
// count that leaves the work uneven across CUs (9 slabs: 576 workgroups) 30 % slower. Large launches keep
// ~32 workgroups per CU with >= 16 chunks (1024 sources) per wave: the tail then balances dynamically.
constexpr int kCUs = 256;
double plan_cost(int groups, int n_chunks, int slabs) {
  static const double eff[6] = {1.0, 0.70, 0.935, 0.975, 0.99, 1.0};
  const int waves = slabs * kWaves, q = n_chunks / waves, r = n_chunks % waves;
  const int wgs = groups * slabs, per_cu = ceil_div(wgs, kCUs);
  const int heavy_per_cu = ceil_div(groups * ceil_div(r, kWaves), kCUs);   // workgroups holding a (q+1)-chunk wave
  const double chunks = (double)per_cu * q + (heavy_per_cu < per_cu ? heavy_per_cu : per_cu);
  // exactly one residency round (<= 5 workgroups per CU) has no slack for uneven placement: +3 % measured
  // (128 groups x 10 slabs: 268 us, x 20 slabs: 261 us); 0.25 chunk-times of prologue/epilogue per workgroup
  return chunks / eff[per_cu < 5 ? per_cu : 5] * (per_cu <= 5 ? 1.03 : 1.0) + 0.25 * per_cu;
}

AccelPlan plan_chunks(int n_chunks, int n_tgt) {
  AccelPlan p;
  p.variant = 0;
  p.groups = ceil_div(n_tgt, kTgtPerWG);
  p.n_chunks = n_chunks;
  int cap = n_chunks / kWaves;                     // at least one chunk per wave
  cap = cap < 1 ? 1 : (cap > kMaxSlabs ? kMaxSlabs : cap);
  int slabs;
  const int pref = ceil_div(8192, p.groups), cap_pref = n_chunks / (16 * kWaves);
  if ((pref < cap_pref ? pref : cap_pref) * p.groups >= 10 * kCUs) {
    slabs = pref < cap_pref ? pref : cap_pref;
    // accuracy: a wave adds its sources in one sequential fp32 chain, so very large systems get enough slabs to keep
    // a chain <= 64 chunks (4096 sources): at N = 524 288 two slabs (65 536-source chains) measured 3e-6 per-row
    // against fp64, 32 slabs 1e-6; the extra slab traffic is < 0.1 % of such a step
    const int min_slabs = ceil_div(n_chunks, kWaves * 64);
    if (slabs < min_slabs) slabs = min_slabs;
  } else {
    slabs = 1;
    double best = plan_cost(p.groups, n_chunks, 1);
    for (int s = 2; s <= cap; ++s) {
      const double c = plan_cost(p.groups, n_chunks, s);
      if (c < best * 0.999) { best = c; slabs = s; }       // ties: fewer slabs to sum
    }
  }
  if (slabs > kMaxSlabs) slabs = kMaxSlabs;
  if (slabs < 1) slabs = 1;
  p.slabs = slabs;
  p.cpw = ceil_div(n_chunks, p.slabs * kWaves);
  return p;
}

AccelPlan plan_accel(int n_src, int n_tgt) { return plan_chunks(ceil_div(n_src, kChunk), n_tgt); }

inline int check(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }
inline int launch_status() { return check(hipGetLastError()); }

// all sources of an n_src array
SrcView full_view(int n_src, const AccelPlan& p) {
  SrcView v;
  v.n_src = n_src; v.n_chunks = p.n_chunks; v.cpw_q = 0; v.cpw_r = 0;
  v.skip_c0 = p.n_chunks; v.skip_cn = 0; v.ex_lo = 0; v.ex_hi = 0; v.edge0 = -1; v.edge1 = -1;
  v.tail = (n_src % kChunk) ? n_src / kChunk : -1;
  return v;
}

// logical chunk count of "n_src sources without the indices [ex_lo, ex_hi)": whole chunks inside the
// excluded range are hopped over, chunks that straddle one of its ends are walked with the element mask
int excluded_view(int n_src, int ex_lo, int ex_hi, SrcView* v) {
  const int phys = ceil_div(n_src, kChunk);
  int c0 = ceil_div(ex_lo, kChunk), c1 = ex_hi / kChunk;      // whole chunks [c0, c1) lie inside
  if (ex_hi >= n_src) c1 = phys;                               // the tail chunk holds padding only beyond ex_hi
  if (c1 < c0) c1 = c0;
  if (v) {
    v->n_src = n_src; v->skip_c0 = c0; v->skip_cn = c1 - c0; v->ex_lo = ex_lo; v->ex_hi = ex_hi;
    v->edge0 = (ex_lo % kChunk) ? ex_lo / kChunk : -1;
    v->edge1 = (ex_hi % kChunk && ex_hi < n_src) ? ex_hi / kChunk : -1;
    if (ex_hi <= ex_lo) { v->skip_c0 = phys; v->skip_cn = 0; v->edge0 = v->edge1 = -1; }
    v->tail = (n_src % kChunk) ? n_src / kChunk : -1;
  }
  return ex_hi <= ex_lo ? phys : phys - (c1 - c0);
}

// force into slabs (or straight into acc_out when one slab), no finishing pass
int launch_accel(const float* posm_src, SrcView sv, const float* posm_tgt, int n_tgt, int off,
                 float eps2, float direct_scale, float* slabs_or_acc, const AccelPlan& p,
                 hipStream_t st, bool uniform = false) {
  dim3 grid(p.groups, p.slabs), block(64 * kWaves);
  const f4* s = reinterpret_cast<const f4*>(posm_src);
  const f4* t = reinterpret_cast<const f4*>(posm_tgt);
  sv.n_chunks = p.n_chunks;
  sv.cpw_q = p.n_chunks / (p.slabs * kWaves); sv.cpw_r = p.n_chunks % (p.slabs * kWaves);
  const bool masked = eps2 < kEps2Masked;
#define NBD_LAUNCH(M, K, U) accel_kernel<M, K, U><<<grid, block, 0, st>>>(s, sv, t, n_tgt, off, eps2, direct_scale, slabs_or_acc)
  if (uniform) {
    if (p.variant == 1) { if (masked) NBD_LAUNCH(true, 4, true); else NBD_LAUNCH(false, 4, true); }
    else                { if (masked) NBD_LAUNCH(true, 8, true); else NBD_LAUNCH(false, 8, true); }
  } else {
    if (p.variant == 1) { if (masked) NBD_LAUNCH(true, 4, false); else NBD_LAUNCH(false, 4, false); }
    else                { if (masked) NBD_LAUNCH(true, 8, false); else NBD_LAUNCH(false, 8, false); }
  }
#undef NBD_LAUNCH
  return launch_status();
}

bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

}  // namespace

extern "C" {

int nbd_abi_version(void) { return NBD_ABI_VERSION; }

size_t nbd_struct_size(const char* name) {
  if (!name) return 0;
#define NBD_SZ(T) if (!strcmp(name, #T)) return sizeof(T);
  NBD_SZ(nbd_gnn_layer_args) NBD_SZ(nbd_gnn_forward_args) NBD_SZ(nbd_knn_pq_args) NBD_SZ(nbd_gnn_train_args)
  NBD_SZ(nbd_gnn_train_grads) NBD_SZ(nbd_cc_train_args) NBD_SZ(nbd_cc_train_grads) NBD_SZ(nbd_cc_pairs_job)
#undef NBD_SZ
  return 0;
}

const char* nbd_strerror(int code) {
  if (code == 0) return "ok";
  if (code == NBD_E_BADARG) return "nbd: bad argument (null/negative/misaligned)";
  if (code == NBD_E_WORKSPACE) return "nbd: workspace too small";
  if (code == NBD_E_UNSUPPORTED) return "nbd: unsupported configuration";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "nbd: unknown error";
}

int nbd_posm_padded_len(int n) { return n <= 0 ? 0 : ceil_div(n, kChunk) * kChunk; }

int nbd_pack_posm_f32(const float* pos, const float* mass, int n, float* posm, nbd_stream_t stream) {
  if (n < 0 || (n > 0 && (!pos || !mass || !posm)) || misaligned16(posm)) return NBD_E_BADARG;
  if (n == 0) return 0;
  const int n_pad = nbd_posm_padded_len(n);
  pack_kernel<<<ceil_div(n_pad, 256), 256, 0, (hipStream_t)stream>>>(pos, mass, n, n_pad,
                                                                    reinterpret_cast<f4*>(posm));
  return launch_status();
}

size_t nbd_accel_workspace_bytes(int n_src, int n_tgt) {
  if (n_src <= 0 || n_tgt <= 0) return 0;
  const AccelPlan p = plan_accel(n_src, n_tgt);
  return p.slabs > 1 ? (size_t)p.slabs * n_tgt * 3 * sizeof(float) : 0;
}

int nbd_accel_plan(int n_src, int n_tgt, int* groups, int* slabs, int* chunks_per_wave) {
  if (n_src <= 0 || n_tgt <= 0) return NBD_E_BADARG;
  const AccelPlan p = plan_accel(n_src, n_tgt);
  if (groups) *groups = p.groups;
  if (slabs) *slabs = p.slabs;
  if (chunks_per_wave) *chunks_per_wave = p.cpw;
  return 0;
}

size_t nbd_step_workspace_bytes(int n) {
  if (n <= 0) return 0;
  return (size_t)plan_accel(n, n).slabs * n * 3 * sizeof(float);
}

int nbd_accel_f32(const float* posm_src, int n_src, const float* posm_tgt, int n_tgt,
                  int tgt_global_offset, float softening_sq, float g_const, float* acc_out,
                  void* workspace, size_t workspace_bytes, nbd_stream_t stream) {
  if (n_src < 0 || n_tgt < 0) return NBD_E_BADARG;
  if (n_tgt == 0) return 0;
  if (!acc_out || !posm_tgt || misaligned16(posm_tgt)) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (n_src == 0) {
    zero_f32_kernel<<<ceil_div(n_tgt * 3, 256), 256, 0, st>>>(acc_out, (size_t)n_tgt * 3);
    return launch_status();
  }
  if (!posm_src || misaligned16(posm_src)) return NBD_E_BADARG;
  const AccelPlan p = plan_accel(n_src, n_tgt);
  if (p.slabs == 1)
    return launch_accel(posm_src, full_view(n_src, p), posm_tgt, n_tgt, tgt_global_offset, softening_sq,
                        g_const, acc_out, p, st);
  const size_t need = (size_t)p.slabs * n_tgt * 3 * sizeof(float);
  if (!workspace || workspace_bytes < need) return NBD_E_WORKSPACE;
  float* slabs = static_cast<float*>(workspace);
  int rc = launch_accel(posm_src, full_view(n_src, p), posm_tgt, n_tgt, tgt_global_offset, softening_sq,
                        1.0f, slabs, p, st);
  if (rc) return rc;
  const int n3 = n_tgt * 3;
  finish_kernel<<<ceil_div(n3, 64), 256, 0, st>>>(slabs, p.slabs, (size_t)n3, g_const, acc_out,
                                                   nullptr, 0.f, n3);
  return launch_status();
}

// ---- tuning hook: the force with an explicit launch geometry (tools/sweep_accel_plan.py, tests)
size_t nbd_accel_tuned_workspace_bytes(int n_tgt, int slabs) {
  if (n_tgt <= 0 || slabs <= 0) return 0;
  return (size_t)slabs * n_tgt * 3 * sizeof(float);
}

int nbd_accel_tuned_f32(const float* posm_src, int n_src, int exclude_lo, int exclude_hi, const float* posm_tgt,
                        int n_tgt, int tgt_global_offset, float softening_sq, float g_const, float* acc_out,
                        void* workspace, size_t workspace_bytes, int slabs, int variant, nbd_stream_t stream) {
  if (n_src <= 0 || n_tgt <= 0 || slabs < 1 || slabs > kMaxSlabs || variant < 0 || variant > 1) return NBD_E_BADARG;
  if (exclude_lo < 0 || exclude_hi < exclude_lo || exclude_hi > n_src) return NBD_E_BADARG;
  if (!acc_out || !posm_tgt || !posm_src || misaligned16(posm_tgt) || misaligned16(posm_src)) return NBD_E_BADARG;
  if (!workspace || workspace_bytes < nbd_accel_tuned_workspace_bytes(n_tgt, slabs)) return NBD_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  SrcView sv;
  AccelPlan p;
  p.n_chunks = excluded_view(n_src, exclude_lo, exclude_hi, &sv);
  p.groups = ceil_div(n_tgt, kTgtPerWG);
  p.slabs = slabs;
  p.variant = variant;
  p.cpw = ceil_div(p.n_chunks > 0 ? p.n_chunks : 1, slabs * kWaves);
  float* sl = static_cast<float*>(workspace);
  const int n3 = n_tgt * 3;
  if (p.n_chunks == 0) {
    zero_f32_kernel<<<ceil_div(n3, 256), 256, 0, st>>>(acc_out, (size_t)n3);
    return launch_status();
  }
  int rc = launch_accel(posm_src, sv, posm_tgt, n_tgt, tgt_global_offset, softening_sq, 1.0f, sl, p, st);
  if (rc) return rc;
  finish_kernel<<<ceil_div(n3, 64), 256, 0, st>>>(sl, p.slabs, (size_t)n3, g_const, acc_out, nullptr, 0.f, n3);
  return launch_status();
}

// ---- range-sharded step (one rank of a torch.distributed group). The rank's targets are its
// own bodies [lo, lo + n_local). The force is issued in two launches so that the all-gather of the other
// ranks' bodies can be in flight during the first:
//   local  : sources = the rank's own packed bodies (posm_local, just written by nbd_kick_drift_f32)
//   remote : sources = the gathered array without [lo, lo + n_local), then the fixed-order slab sum,
//            acc = G * sum, and the second kick fused (finish_kernel), as in nbd_leapfrog_step_f32.
struct ShardPlan { AccelPlan local, remote; };

ShardPlan plan_shard_uncached(int n_total, int lo, int n_local) {
  ShardPlan sp;
  sp.local = plan_chunks(ceil_div(n_local, kChunk), n_local);
  const int rc = excluded_view(n_total, lo, lo + n_local, nullptr);
  sp.remote = plan_chunks(rc > 0 ? rc : 1, n_local);
  if (rc == 0) { sp.remote.slabs = 0; sp.remote.n_chunks = 0; }
  return sp;
}

// A sharded step asks for its plan four times (two launches, each sizing its workspace first) and every plan is two
// searches of up to 64 cost evaluations -- host time on the critical path of a ~140 us rank step. The plan is a pure
// function of (n_total, lo, n_local): memoised per calling thread (a rank steps ONE partition; no locks, no shared state).
ShardPlan plan_shard(int n_total, int lo, int n_local) {
  struct Memo { int n_total, lo, n_local; ShardPlan sp; };
  thread_local Memo memo[4] = {{-1, 0, 0, {}}, {-1, 0, 0, {}}, {-1, 0, 0, {}}, {-1, 0, 0, {}}};
  thread_local int next = 0;
  for (const Memo& m : memo)
    if (m.n_total == n_total && m.lo == lo && m.n_local == n_local) return m.sp;
  Memo& m = memo[next];
  next = (next + 1) & 3;
  m.n_total = n_total; m.lo = lo; m.n_local = n_local;
  m.sp = plan_shard_uncached(n_total, lo, n_local);
  return m.sp;
}

int nbd_shard_plan(int n_total, int lo, int n_local, int* slabs_local, int* cpw_local, int* slabs_remote,
                   int* cpw_remote) {
  if (n_total <= 0 || lo < 0 || n_local <= 0 || lo + n_local > n_total) return NBD_E_BADARG;
  const ShardPlan sp = plan_shard(n_total, lo, n_local);
  if (slabs_local) *slabs_local = sp.local.slabs;
  if (cpw_local) *cpw_local = sp.local.cpw;
  if (slabs_remote) *slabs_remote = sp.remote.slabs;
  if (cpw_remote) *cpw_remote = sp.remote.cpw;
  return 0;
}

size_t nbd_shard_workspace_bytes(int n_total, int lo, int n_local) {
  if (n_total <= 0 || lo < 0 || n_local <= 0 || lo + n_local > n_total) return 0;
  const ShardPlan sp = plan_shard(n_total, lo, n_local);
  return (size_t)(sp.local.slabs + sp.remote.slabs) * n_local * 3 * sizeof(float);
}

int nbd_shard_force_local_f32(const float* posm_local, int n_local, float softening_sq, void* workspace,
                              size_t workspace_bytes, int n_total, int lo, nbd_stream_t stream) {
  if (n_local < 0 || n_total < 0 || lo < 0 || lo + n_local > n_total) return NBD_E_BADARG;
  if (n_local == 0) return 0;
  if (!posm_local || misaligned16(posm_local)) return NBD_E_BADARG;
  if (!workspace || workspace_bytes < nbd_shard_workspace_bytes(n_total, lo, n_local)) return NBD_E_WORKSPACE;
  const ShardPlan sp = plan_shard(n_total, lo, n_local);
  // targets and sources are the same array: the diagonal is at j == i (offset 0)
  return launch_accel(posm_local, full_view(n_local, sp.local), posm_local, n_local, 0, softening_sq, 1.0f,
                      static_cast<float*>(workspace), sp.local, (hipStream_t)stream);
}

int nbd_shard_force_remote_f32(const float* posm_all, int n_total, const float* posm_local, int n_local, int lo,
                               float softening_sq, float g_const, float* acc_out, float* vel, float c_kick,
                               void* workspace, size_t workspace_bytes, nbd_stream_t stream) {
  if (n_local < 0 || n_total < 0 || lo < 0 || lo + n_local > n_total) return NBD_E_BADARG;
  if (n_local == 0) return 0;
  if (!posm_all || !posm_local || !acc_out || misaligned16(posm_all) || misaligned16(posm_local)) return NBD_E_BADARG;
  if (!workspace || workspace_bytes < nbd_shard_workspace_bytes(n_total, lo, n_local)) return NBD_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const ShardPlan sp = plan_shard(n_total, lo, n_local);
  float* slabs = static_cast<float*>(workspace);
  const int n3 = 3 * n_local;
  if (sp.remote.slabs > 0) {
    SrcView sv;
    excluded_view(n_total, lo, lo + n_local, &sv);
    // the diagonal never occurs here (every j in [lo, lo + n_local) is excluded); lo keeps the index meaning
    int rc = launch_accel(posm_all, sv, posm_local, n_local, lo, softening_sq, 1.0f,
                          slabs + (size_t)sp.local.slabs * n3, sp.remote, st);
    if (rc) return rc;
  }
  finish_kernel<<<ceil_div(n3, 64), 256, 0, st>>>(slabs, sp.local.slabs + sp.remote.slabs, (size_t)n3, g_const,
                                                   acc_out, vel, c_kick, n3);
  return launch_status();
}

// The two force launches of the range-sharded step for a system of EQUAL masses (see nbd_leapfrog_step_uniform_f32): the
// kernels without their per-pair mass multiply, g_const * mass_value applied once by the finishing kernel.
int nbd_shard_force_local_uniform_f32(const float* posm_local, int n_local, float softening_sq, void* workspace,
                                      size_t workspace_bytes, int n_total, int lo, nbd_stream_t stream) {
  if (n_local < 0 || n_total < 0 || lo < 0 || lo + n_local > n_total) return NBD_E_BADARG;
  if (n_local == 0) return 0;
  if (!posm_local || misaligned16(posm_local)) return NBD_E_BADARG;
  if (!workspace || workspace_bytes < nbd_shard_workspace_bytes(n_total, lo, n_local)) return NBD_E_WORKSPACE;
  const ShardPlan sp = plan_shard(n_total, lo, n_local);
  return launch_accel(posm_local, full_view(n_local, sp.local), posm_local, n_local, 0, softening_sq, 1.0f,
                      static_cast<float*>(workspace), sp.local, (hipStream_t)stream, true);
}

int nbd_shard_force_remote_uniform_f32(const float* posm_all, int n_total, const float* posm_local, int n_local, int lo,
                                       float softening_sq, float g_const, float mass_value, float* acc_out, float* vel,
                                       float c_kick, void* workspace, size_t workspace_bytes, nbd_stream_t stream) {
  if (n_local < 0 || n_total < 0 || lo < 0 || lo + n_local > n_total) return NBD_E_BADARG;
  if (n_local == 0) return 0;
  if (!posm_all || !posm_local || !acc_out || misaligned16(posm_all) || misaligned16(posm_local)) return NBD_E_BADARG;
  if (!workspace || workspace_bytes < nbd_shard_workspace_bytes(n_total, lo, n_local)) return NBD_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const ShardPlan sp = plan_shard(n_total, lo, n_local);
  float* slabs = static_cast<float*>(workspace);
  const int n3 = 3 * n_local;
  if (sp.remote.slabs > 0) {
    SrcView sv;
    excluded_view(n_total, lo, lo + n_local, &sv);
    int rc = launch_accel(posm_all, sv, posm_local, n_local, lo, softening_sq, 1.0f,
                          slabs + (size_t)sp.local.slabs * n3, sp.remote, st, true);
    if (rc) return rc;
  }
  finish_kernel<<<ceil_div(n3, 64), 256, 0, st>>>(slabs, sp.local.slabs + sp.remote.slabs, (size_t)n3, g_const * mass_value,
                                                   acc_out, vel, c_kick, n3);
  return launch_status();
}

int nbd_kick_drift_f32(float* pos, float* vel, const float* acc, const float* mass, int n,
                       float c_kick, float c_drift, float* posm, nbd_stream_t stream) {
  if (n < 0 || (n > 0 && (!pos || !vel)) || (posm && (!mass || misaligned16(posm)))) return NBD_E_BADARG;
  if (n == 0) return 0;
  const int n_pad = posm ? nbd_posm_padded_len(n) : n;
  kick_drift_kernel<<<ceil_div(n_pad, 256), 256, 0, (hipStream_t)stream>>>(
      pos, vel, acc, mass, n, n_pad, c_kick, c_drift, reinterpret_cast<f4*>(posm));
  return launch_status();
}

int nbd_kick_f32(float* vel, const float* acc, int n, float c, nbd_stream_t stream) {
  if (n < 0 || (n > 0 && (!vel || !acc))) return NBD_E_BADARG;
  if (n == 0) return 0;
  axpy_kernel<<<ceil_div(3 * n, 256), 256, 0, (hipStream_t)stream>>>(vel, acc, c, 3 * n);
  return launch_status();
}

int nbd_drift_f32(float* pos, const float* vel, int n, float c, nbd_stream_t stream) {
  if (n < 0 || (n > 0 && (!pos || !vel))) return NBD_E_BADARG;
  if (n == 0) return 0;
  axpy_kernel<<<ceil_div(3 * n, 256), 256, 0, (hipStream_t)stream>>>(pos, vel, c, 3 * n);
  return launch_status();
}

int nbd_snapshot_f32(const float* pos, const float* vel, const float* acc, int n, float* out, nbd_stream_t stream) {
  if (n < 0 || (n > 0 && (!pos || !vel || !acc || !out))) return NBD_E_BADARG;
  if (n == 0) return 0;
  snapshot_kernel<<<ceil_div(3 * n, 256), 256, 0, (hipStream_t)stream>>>(pos, vel, acc, 3 * n, out);
  return launch_status();
}

int nbd_leapfrog_step_ev_f32(float* pos, float* vel, const float* acc_in, float* acc_out,
                          const float* mass, int n, float dt_half, float dt, float softening_sq,
                          float g_const, float* posm, void* workspace, size_t workspace_bytes,
                          nbd_stream_t stream, void* ev_force_begin, void* ev_force_end) {
  if (n < 0) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!pos || !vel || !acc_in || !acc_out || !mass || !posm || misaligned16(posm)) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const AccelPlan p = plan_accel(n, n);
  const size_t need = (size_t)p.slabs * n * 3 * sizeof(float);  // one slab also goes to scratch
  if (!workspace || workspace_bytes < need) return NBD_E_WORKSPACE;
  int rc = nbd_kick_drift_f32(pos, vel, acc_in, mass, n, dt_half, dt, posm, stream);
  if (rc) return rc;
  float* slabs = static_cast<float*>(workspace);
  if (ev_force_begin && (rc = check(hipEventRecord((hipEvent_t)ev_force_begin, st)))) return rc;
  rc = launch_accel(posm, full_view(n, p), posm, n, 0, softening_sq, 1.0f, slabs, p, st);
  if (rc) return rc;
  if (ev_force_end && (rc = check(hipEventRecord((hipEvent_t)ev_force_end, st)))) return rc;
  const int n3 = 3 * n;
  finish_kernel<<<ceil_div(n3, 64), 256, 0, st>>>(slabs, p.slabs, (size_t)n3, g_const, acc_out, vel,
                                                   dt_half, n3);
  return launch_status();
}

// LeapFrogSimulator.step (simulation.py:153-170) for a system whose bodies all have the SAME mass (the published
// configurations: Plummer, m = 1 / N): the mass factors out of the force sum, a = (G m) sum_j d_ij s_ij^3, so the kernel
// drops its per-pair multiply by m_j (accel_kernel<.., UNI = true>: 11 packed fp32 ops + 2 v_rsq_f32 per source and pair
// of targets instead of 12 + 2) and the finishing kernel applies g_const * mass_value once. Same differences, same slab
// sums; one multiplication rounds differently (per finished sum instead of per term). The CALLER vouches that every
// entry of `mass` equals mass_value (the Python simulator checks once, at construction); `mass` is still read to write
// the packed bodies {x, y, z, m} that the energy kernel and the surrogates consume.
int nbd_leapfrog_step_uniform_f32(float* pos, float* vel, const float* acc_in, float* acc_out, const float* mass,
                                  float mass_value, int n, float dt_half, float dt, float softening_sq, float g_const,
                                  float* posm, void* workspace, size_t workspace_bytes, nbd_stream_t stream,
                                  void* ev_force_begin, void* ev_force_end) {
  if (n < 0) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!pos || !vel || !acc_in || !acc_out || !mass || !posm || misaligned16(posm)) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const AccelPlan p = plan_accel(n, n);
  const size_t need = (size_t)p.slabs * n * 3 * sizeof(float);
  if (!workspace || workspace_bytes < need) return NBD_E_WORKSPACE;
  int rc = nbd_kick_drift_f32(pos, vel, acc_in, mass, n, dt_half, dt, posm, stream);
  if (rc) return rc;
  float* slabs = static_cast<float*>(workspace);
  if (ev_force_begin && (rc = check(hipEventRecord((hipEvent_t)ev_force_begin, st)))) return rc;
  rc = launch_accel(posm, full_view(n, p), posm, n, 0, softening_sq, 1.0f, slabs, p, st, true);
  if (rc) return rc;
  if (ev_force_end && (rc = check(hipEventRecord((hipEvent_t)ev_force_end, st)))) return rc;
  const int n3 = 3 * n;
  finish_kernel<<<ceil_div(n3, 64), 256, 0, st>>>(slabs, p.slabs, (size_t)n3, g_const * mass_value, acc_out, vel,
                                                   dt_half, n3);
  return launch_status();
}

int nbd_leapfrog_step_f32(float* pos, float* vel, const float* acc_in, float* acc_out,
                          const float* mass, int n, float dt_half, float dt, float softening_sq,
                          float g_const, float* posm, void* workspace, size_t workspace_bytes,
                          nbd_stream_t stream) {
  return nbd_leapfrog_step_ev_f32(pos, vel, acc_in, acc_out, mass, n, dt_half, dt, softening_sq, g_const,
                                  posm, workspace, workspace_bytes, stream, nullptr, nullptr);
}

int nbd_euler_step_f32(float* pos, float* vel, float* acc_out, const float* mass, int n, float dt,
                       float softening_sq, float g_const, float* posm, void* workspace,
                       size_t workspace_bytes, nbd_stream_t stream) {
  if (n < 0) return NBD_E_BADARG;
  if (n == 0) return 0;
  if (!pos || !vel || !acc_out || !mass || !posm || misaligned16(posm)) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const AccelPlan p = plan_accel(n, n);
  const size_t need = (size_t)p.slabs * n * 3 * sizeof(float);
  if (!workspace || workspace_bytes < need) return NBD_E_WORKSPACE;
  int rc = nbd_pack_posm_f32(pos, mass, n, posm, stream);
  if (rc) return rc;
  float* slabs = static_cast<float*>(workspace);
  rc = launch_accel(posm, full_view(n, p), posm, n, 0, softening_sq, 1.0f, slabs, p, st);
  if (rc) return rc;
  const int n3 = 3 * n;
  finish_kernel<<<ceil_div(n3, 64), 256, 0, st>>>(slabs, p.slabs, (size_t)n3, g_const, acc_out, vel, dt, n3);
  rc = launch_status();
  if (rc) return rc;
  return nbd_drift_f32(pos, vel, n, dt, stream);
}

size_t nbd_energy_workspace_bytes(int n) {
  if (n <= 0) return 0;
  const int groups = ceil_div(n, kTgtPerWG);
  return ((size_t)groups * energy_slabs(groups) + (size_t)ceil_div(n, 256)) * sizeof(double);
}

int nbd_energy_f32(const float* posm, const float* vel, int n, float softening, float g_const,
                   double* out_uk, void* workspace, size_t workspace_bytes, nbd_stream_t stream) {
  if (n < 0 || !out_uk) return NBD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    zero_f32_kernel<<<1, 256, 0, st>>>(reinterpret_cast<float*>(out_uk), 4);   // two doubles
    return launch_status();
  }
  if (!posm || !vel || misaligned16(posm)) return NBD_E_BADARG;
  if (!workspace || workspace_bytes < nbd_energy_workspace_bytes(n)) return NBD_E_WORKSPACE;
  const int groups = ceil_div(n, kTgtPerWG), slabs = energy_slabs(groups), nk = ceil_div(n, 256);
  double* pu = static_cast<double*>(workspace);
  double* pk = pu + (size_t)groups * slabs;
  const f4* pm = reinterpret_cast<const f4*>(posm);
  energy_kernel<<<dim3(groups, slabs), 64 * kWaves, 0, st>>>(pm, n, ceil_div(n, kChunk), softening,
                                                           softening > 0.f ? 0 : 1, pu);
  int rc = launch_status();
  if (rc) return rc;
  kinetic_kernel<<<nk, 256, 0, st>>>(pm, vel, n, pk);
  rc = launch_status();
  if (rc) return rc;
  energy_final_kernel<<<1, 256, 0, st>>>(pu, groups * slabs, pk, nk, g_const, out_uk);
  return launch_status();
}

}  // extern "C"
