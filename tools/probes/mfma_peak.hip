// What the f16 matrix pipe sustains on this chip, by wave count and accumulator file (developer probe):
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool ACC_A, int THREADS, int WPE>
__global__ __launch_bounds__(THREADS, WPE) void spin(float *out, int iters, float seed) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x4 a, b;
  // seed = 0: zero operands (no toggling); else pseudo-random float16 values in [-2, 2)
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  unsigned h = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
  for (int e = 0; e < 4; ++e) {
    h2 x, y;
    for (int q = 0; q < 2; ++q) {
      h = h * 1664525u + 1013904223u;
      x[q] = (_Float16)(seed * ((float)(h >> 8) * (1.f / 4194304.f) - 2.f));
      h = h * 1664525u + 1013904223u;
      y[q] = (_Float16)(seed * ((float)(h >> 8) * (1.f / 4194304.f) - 2.f));
    }
    a[e] = __builtin_bit_cast(float, x);
    b[e] = __builtin_bit_cast(float, y);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (ACC_A)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      else
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) out[0] = s;
}

template <int NACC, bool ACC_A, int THREADS, int WPE>
void run(const char *name, float *d, float seed) {
  const int iters = 40000, grid = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((spin<NACC, ACC_A, THREADS, WPE>), dim3(grid), dim3(THREADS), 0, 0, d, iters, seed);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * (THREADS / 64) * iters * NACC * 32768.0;
    if (rep == 2) printf("%-44s %8.2f ms  %7.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
  }
}

int main() {
  float *d;
  hipMalloc(&d, 64);
  for (float seed : {0.f, 1.f}) {
    printf("operands: %s\n", seed == 0.f ? "zeros" : "pseudo-random float16 in [-2, 2)");
    run<8, false, 256, 1>("1 wave/SIMD,  8 acc in VGPRs", d, seed);
    run<8, true, 256, 1>("1 wave/SIMD,  8 acc in AGPRs", d, seed);
    run<16, true, 256, 1>("1 wave/SIMD, 16 acc in AGPRs", d, seed);
    run<8, false, 512, 2>("2 waves/SIMD, 8 acc in VGPRs", d, seed);
    run<8, true, 512, 2>("2 waves/SIMD, 8 acc in AGPRs (128 + 128)", d, seed);
  }
  return 0;
}
