// K1 tuning harness (diagnostic): times inner-loop variants of the all-pairs kernel at N = 65536.
// Not product code; the winner is folded back into csrc/direct_force.hip by hand.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);}}while(0)

__device__ __forceinline__ f2 mulm(const f2 zm, const f2 s) {
  f2 u; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(u) : "v"(zm), "v"(s)); return u;
}

// VAR 0: baseline (compiler schedule), VAR 1: rsq clustered per U sources in one asm block
template <int VAR, int U>
__device__ __forceinline__ void block_u(const f4* buf, const f2 xi, const f2 yi, const f2 zi, const f2 e2, f2& ax, f2& ay, f2& az) {
  f2 dx[U], dy[U], dz[U], r2[U], s[U]; f4 p[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    p[u] = buf[u];
    dx[u] = f2{p[u].x, p[u].x} - xi; dy[u] = f2{p[u].y, p[u].y} - yi; dz[u] = f2{p[u].z, p[u].z} - zi;
    r2[u] = __builtin_elementwise_fma(dx[u], dx[u], e2);
    r2[u] = __builtin_elementwise_fma(dy[u], dy[u], r2[u]);
    r2[u] = __builtin_elementwise_fma(dz[u], dz[u], r2[u]);
  }
  if (VAR == 2) {
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] = f2{__builtin_amdgcn_rsqf(r2[u].x), __builtin_amdgcn_rsqf(r2[u].y)};
    __builtin_amdgcn_sched_group_barrier(0x400, 2 * U, 0);   // the 2U transcendentals back to back
  } else if (VAR == 1 && U == 4) {
    asm volatile("v_rsq_f32 %0, %0\n\tv_rsq_f32 %1, %1\n\tv_rsq_f32 %2, %2\n\tv_rsq_f32 %3, %3\n\t"
                 "v_rsq_f32 %4, %4\n\tv_rsq_f32 %5, %5\n\tv_rsq_f32 %6, %6\n\tv_rsq_f32 %7, %7"
                 : "+v"(r2[0].x), "+v"(r2[0].y), "+v"(r2[1].x), "+v"(r2[1].y), "+v"(r2[2].x), "+v"(r2[2].y), "+v"(r2[3].x), "+v"(r2[3].y));
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] = r2[u];
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] = f2{__builtin_amdgcn_rsqf(r2[u].x), __builtin_amdgcn_rsqf(r2[u].y)};
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const f2 w = mulm(f2{p[u].z, p[u].w}, (s[u] * s[u]) * s[u]);
    ax = __builtin_elementwise_fma(w, dx[u], ax);
    ay = __builtin_elementwise_fma(w, dy[u], ay);
    az = __builtin_elementwise_fma(w, dz[u], az);
  }
}

template <int VAR, int U, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(const f4* __restrict__ src, int n_chunks, int cpw, const f4* __restrict__ tgt, float eps2, float* __restrict__ out, int n_tgt) {
  __shared__ f4 lds[WAVES * 2 * 64 + WAVES * 6 * 64 / 4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t_base = blockIdx.x * 128;
  const f4 t0 = tgt[t_base + lane], t1 = tgt[t_base + 64 + lane];
  const f2 xi = {t0.x, t1.x}, yi = {t0.y, t1.y}, zi = {t0.z, t1.z};
  f2 ax = {0, 0}, ay = {0, 0}, az = {0, 0};
  f2 e2 = {eps2, eps2}; asm volatile("" : "+v"(e2));
  const int jw = blockIdx.y * WAVES + wave;
  const int c_begin = min(jw * cpw, n_chunks), c_end = min(c_begin + cpw, n_chunks);
  f4* stage = &lds[wave * 128];
  const f4* s_lane = src + lane;
  if (c_begin < c_end) __builtin_amdgcn_global_load_lds(GPTR(s_lane + (size_t)c_begin * 64), LPTR(stage), 16, 0, 0);
  for (int c = c_begin; c < c_end; ++c) {
    const int b = (c - c_begin) & 1;
    if (c + 1 < c_end) {
      __builtin_amdgcn_global_load_lds(GPTR(s_lane + (size_t)(c + 1) * 64), LPTR(stage + (b ^ 1) * 64), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const f4* buf = stage + b * 64;
#pragma unroll 1
    for (int j = 0; j < 64; j += U) block_u<VAR, U>(buf + j, xi, yi, zi, e2, ax, ay, az);
  }
  float* red = reinterpret_cast<float*>(&lds[WAVES * 128]);
  float* mine = red + wave * 384;
  mine[lane] = ax.x; mine[64 + lane] = ax.y; mine[128 + lane] = ay.x; mine[192 + lane] = ay.y; mine[256 + lane] = az.x; mine[320 + lane] = az.y;
  __syncthreads();
  float* dst = out + ((size_t)blockIdx.y * n_tgt + t_base) * 3;
  for (int o = threadIdx.x; o < 384; o += 64 * WAVES) {
    const int lt = o / 3, comp = o - lt * 3, idx = (comp * 2 + (lt >> 6)) * 64 + (lt & 63);
    float sum = 0; for (int w = 0; w < WAVES; ++w) sum += red[w * 384 + idx];
    dst[o] = sum;
  }
}

template <int VAR, int U, int WAVES>
float run(const char* name, const f4* src, const f4* tgt, float* out, int n, int slabs, std::vector<float>* ref) {
  const int n_chunks = n / 64, cpw = (n_chunks + slabs * WAVES - 1) / (slabs * WAVES);
  dim3 grid(n / 128, slabs), block(64 * WAVES);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) k<VAR, U, WAVES><<<grid, block>>>(src, n_chunks, cpw, tgt, 0.01f, out, n);
  CK(hipDeviceSynchronize());
  std::vector<float> times;
  for (int r = 0; r < 7; ++r) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) k<VAR, U, WAVES><<<grid, block>>>(src, n_chunks, cpw, tgt, 0.01f, out, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); times.push_back(ms / 20);
  }
  std::sort(times.begin(), times.end());
  std::vector<float> h((size_t)slabs * n * 3);
  CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
  std::vector<float> acc((size_t)n * 3, 0.f);
  for (int s = 0; s < slabs; ++s) for (size_t i = 0; i < acc.size(); ++i) acc[i] += h[(size_t)s * n * 3 + i];
  double err = 0, nrm = 0;
  if (ref->empty()) *ref = acc;
  for (size_t i = 0; i < acc.size(); ++i) { err += (acc[i] - (*ref)[i]) * (double)(acc[i] - (*ref)[i]); nrm += (*ref)[i] * (double)(*ref)[i]; }
  printf("%-34s slabs=%2d  median %.4f ms  min %.4f  -> %.3fe12 pairs/s   rel diff vs first %.2e\n", name, slabs, times[3], times[0], (double)n * n / times[3] * 1e-9, std::sqrt(err / nrm));
  return times[3];
}

int main() {
  const int n = 65536;
  std::vector<f4> h(n);
  srand(1);
  for (auto& p : h) { p.x = rand() / (float)RAND_MAX * 2 - 1; p.y = rand() / (float)RAND_MAX * 2 - 1; p.z = rand() / (float)RAND_MAX * 2 - 1; p.w = (0.5f + rand() / (float)RAND_MAX) / n; }
  f4* d; float* out; CK(hipMalloc(&d, n * sizeof(f4))); CK(hipMalloc(&out, (size_t)32 * n * 3 * 4));
  CK(hipMemcpy(d, h.data(), n * sizeof(f4), hipMemcpyHostToDevice));
  std::vector<float> ref;
  for (int rep = 0; rep < 2; ++rep) {
    run<2, 4, 4>("builtin rsq + sched_group U4", d, d, out, n, 16, &ref);
    for (int slabs : {4, 8, 16, 32}) run<2, 8, 4>("builtin rsq + sched_group U8 W4", d, d, out, n, slabs, &ref);
    for (int slabs : {8, 16, 32}) run<2, 8, 2>("builtin rsq + sched_group U8 W2", d, d, out, n, slabs, &ref);
    for (int slabs : {4, 8, 16}) run<2, 8, 8>("builtin rsq + sched_group U8 W8", d, d, out, n, slabs, &ref);
    for (int slabs : {8, 16, 32}) run<2, 16, 4>("builtin rsq + sched_group U16 W4", d, d, out, n, slabs, &ref);
  }
  return 0;
}
