// Proof-of-concept (diagnostic, not product code): all-pairs forces with Newton's third law.
// A wave owns one 64-body tile I (one body per lane) and walks 64-body tiles J >= I. Inside a tile pair
// the sources stay in their lanes; lane l of a 16-lane row sees source (l - t) & 15 of the row through a
// DPP row_ror:t operand (folded into the consuming VOP2, no data movement instruction), and the reaction
// -m_i s^3 d travels back through the inverse rotation. Four row offsets (re-loaded with a rotated lane
// index) cover the 64 x 64 tile. i-side sums live in registers, j-side sums in wave-private LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);}}while(0)

template <int T>
__device__ __forceinline__ float ror(float v) {
  if (T == 0) return v;
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (T & 15), 0xf, 0xf, true));
}

template <int T, bool SYM>
__device__ __forceinline__ void step(const f4 sj, const float xi, const float yi, const float zi, const float mi, const float e2,
                                     float& ax, float& ay, float& az, float& rx, float& ry, float& rz) {
  const float dx = ror<T>(sj.x) - xi, dy = ror<T>(sj.y) - yi, dz = ror<T>(sj.z) - zi;
  float r2 = __builtin_fmaf(dx, dx, e2);
  r2 = __builtin_fmaf(dy, dy, r2);
  r2 = __builtin_fmaf(dz, dz, r2);
  const float s = __builtin_amdgcn_rsqf(r2);
  const float s3 = (s * s) * s;
  const float wj = ror<T>(sj.w) * s3;
  ax = __builtin_fmaf(wj, dx, ax); ay = __builtin_fmaf(wj, dy, ay); az = __builtin_fmaf(wj, dz, az);
  if (SYM) {
    const float wi = mi * s3;
    const float cx = wi * dx, cy = wi * dy, cz = wi * dz;
    rx -= ror<(16 - T) & 15>(cx); ry -= ror<(16 - T) & 15>(cy); rz -= ror<(16 - T) & 15>(cz);
  }
}

template <bool SYM>
__device__ __forceinline__ void row16(const f4 sj, const float xi, const float yi, const float zi, const float mi, const float e2,
                                      float& ax, float& ay, float& az, float& rx, float& ry, float& rz) {
  step<0, SYM>(sj, xi, yi, zi, mi, e2, ax, ay, az, rx, ry, rz);   step<1, SYM>(sj, xi, yi, zi, mi, e2, ax, ay, az, rx, ry, rz);
  step<2, SYM>(sj, xi, yi, zi, mi, e2, ax, ay, az, rx, ry, rz);   step<3, SYM>(sj, xi, yi, zi, mi, e2, ax, ay, az, rx, ry, rz);
  step<4, SYM>(sj, xi, yi, zi, mi, e2, ax, ay, az, rx, ry, rz);   step<5, SYM>(sj, xi, yi, zi, mi, e2, ax, ay, az, rx, ry, rz);
  step<6, SYM>(sj, xi, yi, zi, mi, e2, ax, ay, az, rx, ry, rz);   step<7, SYM>(sj, xi, yi, zi, mi, e2, ax, ay, az, rx, ry, rz);
  step<8, SYM>(sj, xi, yi, zi, mi, e2, ax, ay, az, rx, ry, rz);   step<9, SYM>(sj, xi, yi, zi, mi, e2, ax, ay, az, rx, ry, rz);
  step<10, SYM>(sj, xi, yi, zi, mi, e2, ax, ay, az, rx, ry, rz);  step<11, SYM>(sj, xi, yi, zi, mi, e2, ax, ay, az, rx, ry, rz);
  step<12, SYM>(sj, xi, yi, zi, mi, e2, ax, ay, az, rx, ry, rz);  step<13, SYM>(sj, xi, yi, zi, mi, e2, ax, ay, az, rx, ry, rz);
  step<14, SYM>(sj, xi, yi, zi, mi, e2, ax, ay, az, rx, ry, rz);  step<15, SYM>(sj, xi, yi, zi, mi, e2, ax, ay, az, rx, ry, rz);
}

// grid = (tiles / W, slabs of NJ tiles); block = 64 W.  ipart[slab][n][3], rpart[blockIdx.x][n][3]
template <int W, int NJ>
__global__ __launch_bounds__(64 * W) void sym_kernel(const f4* __restrict__ posm, int n, float eps2,
                                                     float* __restrict__ ipart, float* __restrict__ rpart) {
  __shared__ float jacc[W][NJ * 64 * 3];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int I = blockIdx.x * W + wave, J0 = blockIdx.y * NJ;
  float* mine = jacc[wave];
  for (int o = lane; o < NJ * 192; o += 64) mine[o] = 0.f;
  const f4 ti = posm[I * 64 + lane];
  float e2 = eps2; asm volatile("" : "+v"(e2));
  float ax = 0, ay = 0, az = 0;
  for (int J = max(J0, I); J < J0 + NJ; ++J) {
    if (J == I) {
      float d0 = 0, d1 = 0, d2 = 0;
#pragma unroll 1
      for (int k = 0; k < 4; ++k) {
        const f4 sj = posm[J * 64 + ((lane + 16 * k) & 63)];
        row16<false>(sj, ti.x, ti.y, ti.z, ti.w, e2, ax, ay, az, d0, d1, d2);
      }
    } else {
#pragma unroll 1
      for (int k = 0; k < 4; ++k) {
        const int sl = (lane + 16 * k) & 63;
        const f4 sj = posm[J * 64 + sl];
        float rx = 0, ry = 0, rz = 0;
        row16<true>(sj, ti.x, ti.y, ti.z, ti.w, e2, ax, ay, az, rx, ry, rz);
        float* a = mine + ((J - J0) * 64 + sl) * 3;
        a[0] += rx; a[1] += ry; a[2] += rz;
      }
    }
  }
  float* ip = ipart + ((size_t)blockIdx.y * n + I * 64 + lane) * 3;
  ip[0] = ax; ip[1] = ay; ip[2] = az;
  __syncthreads();
  float* rp = rpart + ((size_t)blockIdx.x * n + J0 * 64) * 3;
  for (int o = threadIdx.x; o < NJ * 192; o += 64 * W) {
    float s = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) s += jacc[w][o];
    rp[o] = s;
  }
}

int main(int argc, char** argv) {
  constexpr int W = 4, NJ = 8;
  for (int n : {4096, 65536}) {
    std::vector<f4> h(n);
    srand(1);
    for (auto& p : h) { p.x = rand() / (float)RAND_MAX * 2 - 1; p.y = rand() / (float)RAND_MAX * 2 - 1; p.z = rand() / (float)RAND_MAX * 2 - 1; p.w = (0.5f + rand() / (float)RAND_MAX) / n; }
    const int tiles = n / 64, gx = tiles / W, slabs = tiles / NJ;
    f4* d; float *ip, *rp;
    CK(hipMalloc(&d, n * sizeof(f4)));
    CK(hipMalloc(&ip, (size_t)slabs * n * 12)); CK(hipMalloc(&rp, (size_t)gx * n * 12));
    CK(hipMemcpy(d, h.data(), n * sizeof(f4), hipMemcpyHostToDevice));
    CK(hipMemset(ip, 0, (size_t)slabs * n * 12)); CK(hipMemset(rp, 0, (size_t)gx * n * 12));
    dim3 grid(gx, slabs), block(64 * W);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
      for (int i = 0; i < 5; ++i) sym_kernel<W, NJ><<<grid, block>>>(d, n, 0.01f, ip, rp);
      CK(hipEventRecord(e0));
      for (int i = 0; i < 10; ++i) sym_kernel<W, NJ><<<grid, block>>>(d, n, 0.01f, ip, rp);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
      printf("n=%d sym kernel %.4f ms -> %.3fe12 pair-interactions/s\n", n, ms, (double)n * n / ms * 1e-9);
    }
    if (n == 4096) {
      std::vector<float> hi((size_t)slabs * n * 3), hr((size_t)gx * n * 3);
      CK(hipMemcpy(hi.data(), ip, hi.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), rp, hr.size() * 4, hipMemcpyDeviceToHost));
      double err = 0, nrm = 0;
      for (int i = 0; i < n; ++i) {
        double a[3] = {0, 0, 0};
        for (int j = 0; j < n; ++j) {
          const double dx = (double)h[j].x - h[i].x, dy = (double)h[j].y - h[i].y, dz = (double)h[j].z - h[i].z;
          const double r2 = dx * dx + dy * dy + dz * dz + 0.01, s = 1 / std::sqrt(r2), w = h[j].w * s * s * s;
          a[0] += w * dx; a[1] += w * dy; a[2] += w * dz;
        }
        for (int c = 0; c < 3; ++c) {
          double g = 0;
          for (int s = 0; s < slabs; ++s) g += hi[((size_t)s * n + i) * 3 + c];
          for (int b = 0; b < gx; ++b) g += hr[((size_t)b * n + i) * 3 + c];
          err += (g - a[c]) * (g - a[c]); nrm += a[c] * a[c];
        }
      }
      printf("n=%d rel L2 error vs fp64 direct sum: %.3e\n", n, std::sqrt(err / nrm));
    }
    CK(hipFree(d)); CK(hipFree(ip)); CK(hipFree(rp));
  }
  return 0;
}
