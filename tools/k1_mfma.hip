// K1 prototype (diagnostic, round 3): can the idle matrix pipe take the accumulation off the vector ALU?
//
//   a_i = sum_j w_ij (x_j - x_i) = [sum_j w_ij x_j] - x_i [sum_j w_ij],     w_ij = m_j (|x_j - x_i|^2 + eps^2)^-3/2
//
// Variant B keeps w_ij on the vector ALU (3 sub, 3 fma, rsq, 3 mul: 9 packed ops + 2 quarter-rate v_rsq per source and
// lane pair) and hands the three accumulating FMAs to the matrix pipe: W (targets x sources) times S = [x_j, y_j, z_j, 1]
// with v_mfma_f32_4x4x1_16b_f32 -- 16 independent 4x4 blocks per instruction, block b = the four targets of lanes
// 4b .. 4b + 3, ALL 16 blocks fed the same source: A operand = w of the lane's own target (exactly what the lane holds),
// B operand = component (lane & 3) of the source's {x, y, z, 1} (one 4-byte LDS read, no vector ALU), 4 x 4 = 16 useful
// outputs per block (the 16x16x4 shape would use 4 of its 16 columns and keep the pipe busy for as long as the whole
// VALU stream). Per source and lane pair the issue stream goes from 12 packed + 2 rsq = 64 cycles to 9 packed + 2 rsq +
// 2 MFMA issues = 60 cycles at best: a 6.7 % ceiling. The accumulation runs against the GLOBAL centre (the bodies are in
// arbitrary order: a tile of targets spans the system), so sum w x - x_i sum w cancels; the error against an fp64
// evaluation is measured below.
//
// Variant A is the product's inner loop (csrc/direct_force.hip, KU = 8, 4 waves) in the same harness.
// Prints one JSON line.   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/k1_mfma.hip -o tools/k1_mfma
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ f2 mulm(const f2 zm, const f2 s) {
  f2 u;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(u) : "v"(zm), "v"(s));
  return u;
}

constexpr int U = 8, WAVES = 4;

// ---- variant A: all on the vector ALU
__device__ __forceinline__ void block_valu(const f4* buf, const f2 xi, const f2 yi, const f2 zi, const f2 e2, f2& ax, f2& ay, f2& az) {
  f2 dx[U], dy[U], dz[U], r2[U], s[U];
  f4 p[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    p[u] = buf[u];
    dx[u] = f2{p[u].x, p[u].x} - xi; dy[u] = f2{p[u].y, p[u].y} - yi; dz[u] = f2{p[u].z, p[u].z} - zi;
    r2[u] = __builtin_elementwise_fma(dx[u], dx[u], e2);
    r2[u] = __builtin_elementwise_fma(dy[u], dy[u], r2[u]);
    r2[u] = __builtin_elementwise_fma(dz[u], dz[u], r2[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) s[u] = f2{__builtin_amdgcn_rsqf(r2[u].x), __builtin_amdgcn_rsqf(r2[u].y)};
  __builtin_amdgcn_sched_group_barrier(0x400, 2 * U, 0);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const f2 w = mulm(f2{p[u].z, p[u].w}, (s[u] * s[u]) * s[u]);
    ax = __builtin_elementwise_fma(w, dx[u], ax);
    ay = __builtin_elementwise_fma(w, dy[u], ay);
    az = __builtin_elementwise_fma(w, dz[u], az);
  }
}

// ---- variant B: w on the vector ALU, sum_j w [x y z 1] on the matrix pipe
__device__ __forceinline__ void block_mfma(const f4* buf, const float* comp, const f2 xi, const f2 yi, const f2 zi, const f2 e2,
                                           f4& dlo, f4& dhi) {
  f2 dx[U], dy[U], dz[U], r2[U], s[U];
  f4 p[U];
  float b[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    p[u] = buf[u];
    b[u] = comp[4 * u];                                   // component (lane & 3) of {x, y, z, 1} of source u
    dx[u] = f2{p[u].x, p[u].x} - xi; dy[u] = f2{p[u].y, p[u].y} - yi; dz[u] = f2{p[u].z, p[u].z} - zi;
    r2[u] = __builtin_elementwise_fma(dx[u], dx[u], e2);
    r2[u] = __builtin_elementwise_fma(dy[u], dy[u], r2[u]);
    r2[u] = __builtin_elementwise_fma(dz[u], dz[u], r2[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) s[u] = f2{__builtin_amdgcn_rsqf(r2[u].x), __builtin_amdgcn_rsqf(r2[u].y)};
  __builtin_amdgcn_sched_group_barrier(0x400, 2 * U, 0);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    // (plain C++ here, not the inline-asm mass splat of variant A: hipcc pads the VALU-write -> MFMA-read hazard only for
    //  instructions it can see, and with the asm product feeding the MFMA directly the matrix pipe read stale weights)
    const f2 w = f2{p[u].w, p[u].w} * ((s[u] * s[u]) * s[u]);
    dlo = __builtin_amdgcn_mfma_f32_4x4x1f32(w.x, b[u], dlo, 0, 0, 0);
    dhi = __builtin_amdgcn_mfma_f32_4x4x1f32(w.y, b[u], dhi, 0, 0, 0);
  }
}

template <int VAR>
__global__ __launch_bounds__(64 * WAVES) void k(const f4* __restrict__ src, const f4* __restrict__ src1, int n_chunks, int cpw,
                                                const f4* __restrict__ tgt, float eps2, float* __restrict__ out, int n_tgt) {
  // per wave: two chunks of {x,y,z,m}, (variant B) two chunks of {x,y,z,1}; then 6 x 64 floats per wave for the reduction
  __shared__ f4 lds[WAVES * 4 * 64 + WAVES * 8 * 64 / 4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t_base = blockIdx.x * 128;
  const f4 t0 = tgt[t_base + lane], t1 = tgt[t_base + 64 + lane];
  const f2 xi = {t0.x, t1.x}, yi = {t0.y, t1.y}, zi = {t0.z, t1.z};
  f2 ax = {0, 0}, ay = {0, 0}, az = {0, 0};
  f4 dlo = {0, 0, 0, 0}, dhi = {0, 0, 0, 0};
  f2 e2 = {eps2, eps2};
  asm volatile("" : "+v"(e2));
  const int jw = blockIdx.y * WAVES + wave;
  const int c_begin = min(jw * cpw, n_chunks), c_end = min(c_begin + cpw, n_chunks);
  f4* stage = &lds[wave * 256];
  f4* stage1 = stage + 128;
  const f4* s_lane = src + lane;
  const f4* s1_lane = src1 + lane;
  auto fetch = [&](int c, int b) {
    __builtin_amdgcn_global_load_lds(GPTR(s_lane + (size_t)c * 64), LPTR(stage + b * 64), 16, 0, 0);
    if (VAR == 1) __builtin_amdgcn_global_load_lds(GPTR(s1_lane + (size_t)c * 64), LPTR(stage1 + b * 64), 16, 0, 0);
  };
  if (c_begin < c_end) fetch(c_begin, 0);
  for (int c = c_begin; c < c_end; ++c) {
    const int b = (c - c_begin) & 1;
    if (c + 1 < c_end) {
      fetch(c + 1, b ^ 1);
      if (VAR == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const f4* buf = stage + b * 64;
    const float* comp = reinterpret_cast<const float*>(stage1 + b * 64) + (lane & 3);
#pragma unroll 1
    for (int j = 0; j < 64; j += U) {
      if (VAR == 0) block_valu(buf + j, xi, yi, zi, e2, ax, ay, az);
      else block_mfma(buf + j, comp + 4 * j, xi, yi, zi, e2, dlo, dhi);
    }
  }
  float* red = reinterpret_cast<float*>(&lds[WAVES * 256]);
  float* mine = red + wave * 512;
  if (VAR == 1) {
    // lane 4b + j, register i holds component j of target 4b + i: through LDS to "lane = target"
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mine[((lane & ~3) + i) * 4 + (lane & 3)] = dlo[i];
      mine[256 + ((lane & ~3) + i) * 4 + (lane & 3)] = dhi[i];
    }
    __builtin_amdgcn_wave_barrier();
    const f4 a = *reinterpret_cast<const f4*>(mine + lane * 4), bq = *reinterpret_cast<const f4*>(mine + 256 + lane * 4);
    __builtin_amdgcn_wave_barrier();
    ax = f2{a[0] - xi.x * a[3], bq[0] - xi.y * bq[3]};            // sum w x_j - x_i sum w
    ay = f2{a[1] - yi.x * a[3], bq[1] - yi.y * bq[3]};
    az = f2{a[2] - zi.x * a[3], bq[2] - zi.y * bq[3]};
  }
  mine[lane] = ax.x; mine[64 + lane] = ax.y; mine[128 + lane] = ay.x; mine[192 + lane] = ay.y; mine[256 + lane] = az.x; mine[320 + lane] = az.y;
  __syncthreads();
  float* dst = out + ((size_t)blockIdx.y * n_tgt + t_base) * 3;
  for (int o = threadIdx.x; o < 384; o += 64 * WAVES) {
    const int lt = o / 3, cmp = o - lt * 3, idx = (cmp * 2 + (lt >> 6)) * 64 + (lt & 63);
    float sum = 0;
    for (int w = 0; w < WAVES; ++w) sum += red[w * 512 + idx];
    dst[o] = sum;
  }
}

template <int VAR>
double run(const f4* src, const f4* src1, float* out, int n, int slabs, std::vector<float>& acc) {
  const int n_chunks = n / 64, cpw = (n_chunks + slabs * WAVES - 1) / (slabs * WAVES);
  dim3 grid(n / 128, slabs), block(64 * WAVES);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 300; ++i) k<VAR><<<grid, block>>>(src, src1, n_chunks, cpw, src, 0.01f, out, n);     // clocks ramp
  CK(hipDeviceSynchronize());
  std::vector<float> times;
  for (int r = 0; r < 9; ++r) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) k<VAR><<<grid, block>>>(src, src1, n_chunks, cpw, src, 0.01f, out, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    times.push_back(ms / 20);
  }
  std::sort(times.begin(), times.end());
  std::vector<float> h((size_t)slabs * n * 3);
  CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
  acc.assign((size_t)n * 3, 0.f);
  for (int s = 0; s < slabs; ++s)
    for (size_t i = 0; i < acc.size(); ++i) acc[i] += h[(size_t)s * n * 3 + i];
  return times[4];
}

int main(int argc, char** argv) {
  const int n = 65536, slabs = 16;
  const bool only_mfma = argc > 1 && atoi(argv[1]) == 1, only_valu = argc > 1 && atoi(argv[1]) == 0;
  // Plummer-like cloud (a = 1, radii clipped at 20), centre of mass removed: the shape the headline bench runs on
  std::vector<f4> h(n), h1(n);
  srand(1);
  double cx = 0, cy = 0, cz = 0;
  for (auto& p : h) {
    const double u = (rand() + 1.0) / (RAND_MAX + 2.0);
    double r = 1.0 / std::sqrt(std::pow(u, -2.0 / 3.0) - 1.0);
    if (r > 20) r = 20;
    const double ct = 2.0 * rand() / RAND_MAX - 1.0, ph = 6.283185307179586 * rand() / RAND_MAX, st = std::sqrt(1 - ct * ct);
    p.x = (float)(r * st * std::cos(ph)); p.y = (float)(r * st * std::sin(ph)); p.z = (float)(r * ct); p.w = 1.0f / n;
    cx += p.x; cy += p.y; cz += p.z;
  }
  for (size_t i = 0; i < h.size(); ++i) {
    h[i].x -= (float)(cx / n); h[i].y -= (float)(cy / n); h[i].z -= (float)(cz / n);
    h1[i] = f4{h[i].x, h[i].y, h[i].z, 1.0f};
  }
  f4 *d, *d1;
  float* out;
  CK(hipMalloc(&d, n * sizeof(f4))); CK(hipMalloc(&d1, n * sizeof(f4))); CK(hipMalloc(&out, (size_t)slabs * n * 3 * 4));
  CK(hipMemcpy(d, h.data(), n * sizeof(f4), hipMemcpyHostToDevice));
  CK(hipMemcpy(d1, h1.data(), n * sizeof(f4), hipMemcpyHostToDevice));
  std::vector<float> a_valu, a_mfma;
  double t_valu = 0, t_mfma = 0;
  if (!only_mfma) t_valu = run<0>(d, d1, out, n, slabs, a_valu);
  if (!only_valu) t_mfma = run<1>(d, d1, out, n, slabs, a_mfma);
  // fp64 reference on a sample of rows (every 257th body: all radii)
  double worst_valu = 0, worst_mfma = 0, sum_valu = 0, sum_mfma = 0;
  int rows = 0;
  for (int i = 0; i < n; i += 257, ++rows) {
    double ax = 0, ay = 0, az = 0;
    for (int j = 0; j < n; ++j) {
      const double dx = (double)h[j].x - h[i].x, dy = (double)h[j].y - h[i].y, dz = (double)h[j].z - h[i].z;
      const double q = dx * dx + dy * dy + dz * dz + 0.01, w = h[j].w / (q * std::sqrt(q));
      ax += w * dx; ay += w * dy; az += w * dz;
    }
    const double nrm = std::sqrt(ax * ax + ay * ay + az * az);
    auto rel = [&](const std::vector<float>& a) {
      if (a.empty()) return 0.0;
      const double ex = a[3 * i] - ax, ey = a[3 * i + 1] - ay, ez = a[3 * i + 2] - az;
      return std::sqrt(ex * ex + ey * ey + ez * ez) / nrm;
    };
    worst_valu = std::max(worst_valu, rel(a_valu)); worst_mfma = std::max(worst_mfma, rel(a_mfma));
    sum_valu += rel(a_valu); sum_mfma += rel(a_mfma);
  }
  const double pairs = (double)n * n;
  printf("{\"n\": %d, \"slabs\": %d, \"rows_checked_fp64\": %d, "
         "\"valu\": {\"ms\": %.4f, \"pairs_per_s\": %.4e, \"frac_fp32_peak_20flop\": %.4f, \"row_rel_err_max\": %.3e, \"row_rel_err_mean\": %.3e}, "
         "\"mfma_4x4x1\": {\"ms\": %.4f, \"pairs_per_s\": %.4e, \"frac_fp32_peak_20flop\": %.4f, \"row_rel_err_max\": %.3e, \"row_rel_err_mean\": %.3e}}\n",
         n, slabs, rows, t_valu, t_valu > 0 ? pairs / (t_valu * 1e-3) : 0.0, t_valu > 0 ? pairs * 20 / (t_valu * 1e-3) / 157.3e12 : 0.0,
         worst_valu, sum_valu / rows, t_mfma, t_mfma > 0 ? pairs / (t_mfma * 1e-3) : 0.0,
         t_mfma > 0 ? pairs * 20 / (t_mfma * 1e-3) / 157.3e12 : 0.0, worst_mfma, sum_mfma / rows);
  return 0;
}
