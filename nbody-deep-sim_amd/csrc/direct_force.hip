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
//     v_pk_add/v_pk_fma/v_pk_mul on both; 8 waves/SIMD (<=64 VGPRs) hide LDS/rsq latency;
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

// One source against the lane's two targets. 12 packed ops + 2 v_rsq_f32.
template <bool MASKED>
__device__ __forceinline__ void interact(const f4 p, const f2 xi, const f2 yi, const f2 zi,
                                         const f2 e2, f2& ax, f2& ay, f2& az, int j, int i0,
                                         int i1, int n_src) {
  const f2 dx = f2{p.x, p.x} - xi, dy = f2{p.y, p.y} - yi, dz = f2{p.z, p.z} - zi;  // r_j - r_i
  f2 r2 = __builtin_elementwise_fma(dx, dx, e2);
  r2 = __builtin_elementwise_fma(dy, dy, r2);
  r2 = __builtin_elementwise_fma(dz, dz, r2);
  f2 s = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
  if (MASKED) {  // exact fill_diagonal_(0): only j == i is dropped; padding is dropped too
    const bool live = j < n_src;
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
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(w) : "v"(zm), "v"(s3));
  ax = __builtin_elementwise_fma(w, dx, ax);
  ay = __builtin_elementwise_fma(w, dy, ay);
  az = __builtin_elementwise_fma(w, dz, az);
}

// kU sources at once for the un-masked kernel: same arithmetic as interact(), with the 2*kU v_rsq_f32
// issued back to back (__builtin_amdgcn_sched_group_barrier on the TRANS class). Switching between the
// quarter-rate transcendental unit and the packed-math stream costs issue cycles on gfx950 (3 fma : 1
// rsq mixes run ~10 % under the sum of their parts, tools/ubench_valu.hip), so the switches are
// batched; the rsq stays a compiler builtin so that hipcc fills the transcendental -> VALU wait state
// with independent work instead of the s_nop it must put behind an opaque asm block. Measured
// (tools/k1_variants.hip, N = 65 536): kU = 8 at 90 VGPRs / 5 waves per SIMD beats kU = 4 at 58 VGPRs /
// 8 waves (1.004 vs 1.010 ms) and an inline-asm rsq block (1.021 ms).
constexpr int kU = 8;
__device__ __forceinline__ void interact_block(const f4* __restrict__ buf, const f2 xi, const f2 yi, const f2 zi,
                                               const f2 e2, f2& ax, f2& ay, f2& az) {
  f4 p[kU];
  f2 dx[kU], dy[kU], dz[kU], s[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    p[u] = buf[u];
    dx[u] = f2{p[u].x, p[u].x} - xi; dy[u] = f2{p[u].y, p[u].y} - yi; dz[u] = f2{p[u].z, p[u].z} - zi;
    f2 r2 = __builtin_elementwise_fma(dx[u], dx[u], e2);
    r2 = __builtin_elementwise_fma(dy[u], dy[u], r2);
    s[u] = __builtin_elementwise_fma(dz[u], dz[u], r2);
  }
#pragma unroll
  for (int u = 0; u < kU; ++u) s[u] = f2{__builtin_amdgcn_rsqf(s[u].x), __builtin_amdgcn_rsqf(s[u].y)};
  __builtin_amdgcn_sched_group_barrier(0x400, 2 * kU, 0);      // 0x400 = TRANS: keep the rsq's together
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const f2 zm = {p[u].z, p[u].w};
    const f2 s3 = (s[u] * s[u]) * s[u];          // compiler-visible consumers of the rsq results (hazard-padded)
    f2 w;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(w) : "v"(zm), "v"(s3));
    ax = __builtin_elementwise_fma(w, dx[u], ax);
    ay = __builtin_elementwise_fma(w, dy[u], ay);
    az = __builtin_elementwise_fma(w, dz[u], az);
  }
}

// grid = (target groups of 128, slabs); block = 256.
// Wave (blockIdx.y, w) handles source chunks [jw*cpw, (jw+1)*cpw) with jw = blockIdx.y*4 + w.
template <bool MASKED>
__global__ __launch_bounds__(64 * kWaves) void accel_kernel(
    const f4* __restrict__ src, int n_src, int n_chunks, int cpw, const f4* __restrict__ tgt,
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
  const int c_begin = min(jw * cpw, n_chunks), c_end = min(c_begin + cpw, n_chunks);
  f4* stage = &lds[wave * 2 * kChunk];
  const f4* s_lane = src + lane;
  if (c_begin < c_end)
    __builtin_amdgcn_global_load_lds(GPTR(s_lane + (size_t)c_begin * kChunk), LPTR(stage), 16, 0, 0);
  for (int c = c_begin; c < c_end; ++c) {
    const int b = (c - c_begin) & 1;
    if (c + 1 < c_end) {
      __builtin_amdgcn_global_load_lds(GPTR(s_lane + (size_t)(c + 1) * kChunk),
                                       LPTR(stage + (b ^ 1) * kChunk), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");  // chunk c has landed, c+1 in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const f4* buf = stage + b * kChunk;
    const int j0 = c * kChunk;
    if (MASKED) {
#pragma unroll 4
      for (int j = 0; j < kChunk; ++j)
        interact<true>(buf[j], xi, yi, zi, e2, ax, ay, az, j0 + j, tgt_off + i0, tgt_off + i1, n_src);
    } else {
#pragma unroll 1
      for (int j = 0; j < kChunk; j += kU) interact_block(buf + j, xi, yi, zi, e2, ax, ay, az);
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

// acc = g * (slab_0 + slab_1 + ...), optional fused kick v += c * acc (simulation.py:88,170)
__global__ __launch_bounds__(256) void finish_kernel(const float* __restrict__ slabs, int n_slabs,
                                                     size_t slab_stride, float g, float* __restrict__ acc,
                                                     float* __restrict__ vel, float c_kick, int n3) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n3) return;
  float sum = slabs[i];
  for (int s = 1; s < n_slabs; ++s) sum += slabs[s * slab_stride + i];
  const float a = __fmul_rn(g, sum);
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

struct AccelPlan { int groups, slabs, n_chunks, cpw; };

AccelPlan plan_accel(int n_src, int n_tgt) {
  AccelPlan p;
  p.groups = ceil_div(n_tgt, kTgtPerWG);
  p.n_chunks = ceil_div(n_src, kChunk);
  // Source split across workgroups. Preferred: ~4 residency rounds (8192 workgroups; measured 2-3 %
  // faster than exactly one round at N = 65 536, the tail is balanced dynamically) with >= 16 chunks
  // (1024 sources) per wave. If that cannot even fill the chip once (few targets), go down to one
  // chunk per wave to get as close to 2048 workgroups = 8 per CU as the problem allows.
  const int want_fill = ceil_div(2048, p.groups), want_pref = ceil_div(8192, p.groups);
  const int cap_pref = p.n_chunks / (16 * kWaves), cap_fill = p.n_chunks / kWaves;
  int slabs = want_pref < cap_pref ? want_pref : cap_pref;
  if (slabs < want_fill) slabs = want_fill < cap_fill ? want_fill : cap_fill;
  if (slabs > kMaxSlabs) slabs = kMaxSlabs;
  if (slabs < 1) slabs = 1;
  p.slabs = slabs;
  p.cpw = ceil_div(p.n_chunks, p.slabs * kWaves);
  return p;
}

inline int check(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }
inline int launch_status() { return check(hipGetLastError()); }

// force into slabs (or straight into acc_out when one slab), no finishing pass
int launch_accel(const float* posm_src, int n_src, const float* posm_tgt, int n_tgt, int off,
                 float eps2, float direct_scale, float* slabs_or_acc, const AccelPlan& p,
                 hipStream_t st) {
  dim3 grid(p.groups, p.slabs), block(64 * kWaves);
  const f4* s = reinterpret_cast<const f4*>(posm_src);
  const f4* t = reinterpret_cast<const f4*>(posm_tgt);
  if (eps2 < kEps2Masked)
    accel_kernel<true><<<grid, block, 0, st>>>(s, n_src, p.n_chunks, p.cpw, t, n_tgt, off, eps2,
                                               direct_scale, slabs_or_acc);
  else
    accel_kernel<false><<<grid, block, 0, st>>>(s, n_src, p.n_chunks, p.cpw, t, n_tgt, off, eps2,
                                                direct_scale, slabs_or_acc);
  return launch_status();
}

bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

}  // namespace

extern "C" {

int nbd_abi_version(void) { return NBD_ABI_VERSION; }

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
    return launch_accel(posm_src, n_src, posm_tgt, n_tgt, tgt_global_offset, softening_sq, g_const,
                        acc_out, p, st);
  const size_t need = (size_t)p.slabs * n_tgt * 3 * sizeof(float);
  if (!workspace || workspace_bytes < need) return NBD_E_WORKSPACE;
  float* slabs = static_cast<float*>(workspace);
  int rc = launch_accel(posm_src, n_src, posm_tgt, n_tgt, tgt_global_offset, softening_sq, 1.0f,
                        slabs, p, st);
  if (rc) return rc;
  const int n3 = n_tgt * 3;
  finish_kernel<<<ceil_div(n3, 256), 256, 0, st>>>(slabs, p.slabs, (size_t)n3, g_const, acc_out,
                                                   nullptr, 0.f, n3);
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
  rc = launch_accel(posm, n, posm, n, 0, softening_sq, 1.0f, slabs, p, st);
  if (rc) return rc;
  if (ev_force_end && (rc = check(hipEventRecord((hipEvent_t)ev_force_end, st)))) return rc;
  const int n3 = 3 * n;
  finish_kernel<<<ceil_div(n3, 256), 256, 0, st>>>(slabs, p.slabs, (size_t)n3, g_const, acc_out, vel,
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
  rc = launch_accel(posm, n, posm, n, 0, softening_sq, 1.0f, slabs, p, st);
  if (rc) return rc;
  const int n3 = 3 * n;
  finish_kernel<<<ceil_div(n3, 256), 256, 0, st>>>(slabs, p.slabs, (size_t)n3, g_const, acc_out, vel, dt, n3);
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
