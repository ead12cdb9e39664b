// ContinuousConv consumer step on the bf16 matrix pipe with fp32 accuracy (diagnostic, round 3; DESIGN.md "the lever that is
// left"): one step of the fused kernel's consumers is C[16 rows][128 columns] += A[16][128] . B[128][128], eight waves of 16
// columns each. The product kernel runs it as 32 v_mfma_f32_16x16x4_f32 per wave (32 matrix-pipe cycles each: the fp32 pipe
// is the kernel's bound). Here the same step with both operands split into three bf16 terms,
//     a = a_hi + a_mid + a_lo  (each the bf16 rounding of what the previous terms left: 24 mantissa bits in all),
// and the six products hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid accumulated in fp32 by v_mfma_f32_16x16x32_bf16: 24
// instructions per wave and step. Measured: time per step in both forms with the A fragments re-read from LDS every step
// (as the consumers do) and the B fragments resident in registers, and the error of both results against an fp64 product.
// Prints one JSON line.   hipcc -O3 --offload-arch=gfx950 tools/cc_bf16x3.hip -o tools/cc_bf16x3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ROWS = 16, K = 128, COLS = 128, WAVES = 8;
constexpr int LDA = K + 4;            // fp32 A rows in LDS (floats): (m, 4 ks + q) on banks 4 m + q + ...
constexpr int LDH = K + 16;           // bf16 A rows in LDS (elements): 72 dwords apart = 8 mod 64, so the 16-lane groups of a
                                      // ds_read_b128 (bank 4 (2 m + q) with m = l & 15, q = l >> 4) touch 16 different quads (K + 8: 3x slower reads)

// ---- fp32: D = A (16 x 4) . B (4 x 16), lane l: A[m = l & 15][k = l >> 4], B[k = l >> 4][n = l & 15], D[4 (l >> 4) + v][l & 15]
__global__ __launch_bounds__(64 * WAVES) void step_f32(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int iters) {
  __shared__ float a_s[ROWS * LDA];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < ROWS * K; i += 64 * WAVES) a_s[(i / K) * LDA + (i % K)] = A[i];
  float bf[32];
#pragma unroll
  for (int ks = 0; ks < 32; ++ks) bf[ks] = B[(size_t)(4 * ks + (lane >> 4)) * COLS + 16 * w + (lane & 15)];
  __syncthreads();
  f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;           // two accumulators, as the product's consumers
  const float* arow = a_s + (lane & 15) * LDA + (lane >> 4);
  for (int it = 0; it < iters; ++it) {
    float av[32];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) av[ks] = arow[4 * ks];
#pragma unroll
    for (int ks = 0; ks < 32; ks += 2) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bf[ks], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks + 1], bf[ks + 1], acc1, 0, 0, 0);
    }
    asm volatile("" ::: "memory");                       // the next step reads its A tile again
  }
  if (blockIdx.x == 0)
    for (int v = 0; v < 4; ++v) C[(4 * (lane >> 4) + v) * COLS + 16 * w + (lane & 15)] = (acc0[v] + acc1[v]) / (float)iters;
}

// ---- bf16 x 3: v_mfma_f32_16x16x32_bf16, lane l: A[m = l & 15][k = 8 (l >> 4) .. + 7], B[k = 8 (l >> 4) .. + 7][n = l & 15]
__global__ __launch_bounds__(64 * WAVES) void step_bf16x3(const uint16_t* __restrict__ A3, const uint16_t* __restrict__ B3, float* __restrict__ C, int iters) {
  __shared__ __attribute__((aligned(16))) uint16_t a_s[3 * ROWS * LDH];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 3 * ROWS * K; i += 64 * WAVES) {
    const int t = i / (ROWS * K), r = (i / K) % ROWS, k = i % K;
    a_s[(t * ROWS + r) * LDH + k] = A3[i];
  }
  b8 bfr[3][4];                                          // [term][K slab of 32]: this lane's 8 consecutive k of column n
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint16_t tmp[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) tmp[e] = B3[((size_t)t * K + 32 * s + 8 * (lane >> 4) + e) * COLS + 16 * w + (lane & 15)];
      memcpy(&bfr[t][s], tmp, 16);
    }
  __syncthreads();
  f4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // one per K slab: independent chains
  const uint16_t* arow = a_s + (lane & 15) * LDH + 8 * (lane >> 4);
  for (int it = 0; it < iters; ++it) {
    b8 av[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) av[t][s] = *reinterpret_cast<const b8*>(arow + t * ROWS * LDH + 32 * s);
    // smallest products first; the four K slabs keep four independent accumulator chains interleaved (six dependent
    // MFMAs back to back on one accumulator ran at 62 % of the bf16 rate)
#define NBD_TERM(TA, TB)                                                                                         \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[TA][s], bfr[TB][s], acc[s], 0, 0, 0);
    NBD_TERM(1, 1) NBD_TERM(2, 0) NBD_TERM(0, 2) NBD_TERM(1, 0) NBD_TERM(0, 1) NBD_TERM(0, 0)
#undef NBD_TERM
    asm volatile("" ::: "memory");
  }
  if (blockIdx.x == 0)
    for (int v = 0; v < 4; ++v)
      C[(4 * (lane >> 4) + v) * COLS + 16 * w + (lane & 15)] = ((acc[0][v] + acc[1][v]) + (acc[2][v] + acc[3][v])) / (float)iters;
}

static uint16_t bf16_rn(float x) {                       // round to nearest even
  uint32_t u; memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf16_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float x; memcpy(&x, &u, 4); return x; }
static void split3(const std::vector<float>& src, std::vector<uint16_t>& dst) {       // dst[t][i]
  const size_t n = src.size();
  dst.resize(3 * n);
  for (size_t i = 0; i < n; ++i) {
    float r = src[i];
    for (int t = 0; t < 3; ++t) { const uint16_t h = bf16_rn(r); dst[t * n + i] = h; r -= bf16_f(h); }
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  std::vector<float> A(ROWS * K), B(K * COLS);
  srand(7);
  auto rnd = [] { float s = 0; for (int i = 0; i < 12; ++i) s += (float)rand() / (float)RAND_MAX; return s - 6.f; };     // ~N(0, 1)
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd();
  std::vector<double> C64(ROWS * COLS, 0.0);
  for (int m = 0; m < ROWS; ++m) for (int k = 0; k < K; ++k) for (int n = 0; n < COLS; ++n) C64[m * COLS + n] += (double)A[m * K + k] * (double)B[k * COLS + n];
  std::vector<uint16_t> A3, B3;
  split3(A, A3); split3(B, B3);
  float *dA, *dB, *dC; uint16_t *dA3, *dB3;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, ROWS * COLS * 4));
  CK(hipMalloc(&dA3, A3.size() * 2)); CK(hipMalloc(&dB3, B3.size() * 2));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dA3, A3.data(), A3.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB3, B3.data(), B3.size() * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256;                                  // one workgroup of 8 waves per CU: two waves per SIMD, as the consumers
  std::vector<float> C(ROWS * COLS);
  double ms[2], err[2];
  for (int v = 0; v < 2; ++v) {
    for (int rep = 0; rep < 2; ++rep) {                   // warm, then timed
      CK(hipEventRecord(e0));
      if (v == 0) step_f32<<<grid, 64 * WAVES>>>(dA, dB, dC, iters); else step_bf16x3<<<grid, 64 * WAVES>>>(dA3, dB3, dC, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); ms[v] = t;
    }
    // the error from ONE step (the timed launches sum the same tile `iters` times: their fp32 running sum is not the subject)
    if (v == 0) step_f32<<<1, 64 * WAVES>>>(dA, dB, dC, 1); else step_bf16x3<<<1, 64 * WAVES>>>(dA3, dB3, dC, 1);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int m = 0; m < ROWS; ++m) {
      double rmax = 0, dmax = 0;
      for (int n = 0; n < COLS; ++n) { rmax = std::max(rmax, std::fabs(C64[m * COLS + n])); dmax = std::max(dmax, std::fabs((double)C[m * COLS + n] - C64[m * COLS + n])); }
      worst = std::max(worst, dmax / rmax);
    }
    err[v] = worst;
  }
  const double flop = 2.0 * ROWS * K * COLS;              // per workgroup-step
  printf("{\"iters\": %d, \"workgroups\": %d, \"step\": \"C[16][128] += A[16][128] . B[128][128], 8 waves x 16 columns\", "
         "\"fp32_mfma_16x16x4\": {\"ns_per_step\": %.2f, \"TFLOPs_chip\": %.1f, \"row_rel_err_vs_fp64\": %.3g}, "
         "\"bf16x3_mfma_16x16x32\": {\"ns_per_step\": %.2f, \"TFLOPs_fp32_equivalent_chip\": %.1f, \"row_rel_err_vs_fp64\": %.3g}, "
         "\"speedup\": %.2f}\n",
         iters, grid, ms[0] * 1e6 / iters, flop * grid * iters / (ms[0] * 1e-3) / 1e12, err[0],
         ms[1] * 1e6 / iters, flop * grid * iters / (ms[1] * 1e-3) / 1e12, err[1], ms[0] / ms[1]);
  return 0;
}
