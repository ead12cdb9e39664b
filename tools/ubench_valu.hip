// VALU issue-rate microbenchmark for gfx950: settles whether f32 FMA is full rate
// un-packed (SIMD-32) or only packed, and what v_rsq_f32 costs. Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
  float b = 1.0f + threadIdx.x * 1e-9f, c = 1e-9f;
  float a[16];
  f2 p[16];
  f2 pb = {b, b}, pc = {c, c};
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = 1.0f + i; p[i] = f2{1.0f + i, 2.0f + i}; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
      if (MODE == 2) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
      if (MODE == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
      if (MODE == 5) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (MODE == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
      if (MODE == 7) {  // 3 fma : 1 rsq mix (12 fma + 4 rsq per 16)
        if ((i & 3) == 3) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      }
      if (MODE == 9) asm volatile("v_sub_f32_dpp %0, %1, %0 row_ror:3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b));
      if (MODE == 10) asm volatile("v_mov_b32_dpp %0, %1 row_ror:3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b));
      if (MODE == 11) {  // dpp source freshly written by the previous VALU op of the same wave (hazard: 2 wait states)
        asm volatile("v_sub_f32_dpp %0, %1, %0 row_ror:3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(a[(i + 15) & 15]));
      }
      if (MODE == 8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(iters));  // sgpr operand
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
int run(const char* name, int lanes_per_instr_flop) {
  const int iters = 20000;
  for (int bpc : {1, 2, 4, 8}) {
    int grid = 256 * bpc;
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, grid * 256 * sizeof(float)));
    CK(hipMalloc(&cyc, grid * sizeof(unsigned long long)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) k<MODE><<<grid, 256>>>(out, iters, cyc);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) k<MODE><<<grid, 256>>>(out, iters, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    std::vector<unsigned long long> h(grid);
    CK(hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += v; avg /= grid;
    double winstr = (double)grid * 4 * iters * 16;          // wave-instructions issued chip-wide
    double per_simd_cyc = avg / ((double)iters * 16 * bpc);  // cycles per wave-instr per SIMD (bpc waves share a SIMD)
    printf("%-12s waves/SIMD=%d  %.3f ms  %.2f Gwave-instr/s  %.2f Tlane-op/s  cyc/instr/SIMD=%.2f  clk=%.2f GHz (memtime)\n", name, bpc, ms,
           winstr / ms * 1e-6, winstr * 64 * lanes_per_instr_flop / ms * 1e-9, per_simd_cyc, avg / (ms * 1e6));
    CK(hipFree(out)); CK(hipFree(cyc));
  }
  return 0;
}
int main() {
  run<0>("v_fma_f32", 1); run<1>("v_pk_fma_f32", 2); run<2>("v_rsq_f32", 1); run<3>("v_mul_f32", 1);
  run<4>("v_pk_mul_f32", 2); run<5>("v_sub_f32", 1); run<6>("v_pk_add_f32", 2); run<7>("3fma:1rsq", 1); run<8>("fma_sgpr", 1);
  run<9>("sub_dpp_ror", 1); run<10>("mov_dpp_ror", 1); run<11>("sub_dpp_dep", 1);
  return 0;
}
