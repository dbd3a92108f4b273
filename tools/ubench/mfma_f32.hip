// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate on MI355X as a function of
// waves per SIMD and operand data (developer tool; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32 mfma_f32.hip && ./mfma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(const float *in, float *out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = in[(threadIdx.x + 64 * i) & 4095];
    b[i] = in[(threadIdx.x * 7 + 64 * i + 1) & 4095];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float *in, *out;
  hipMalloc(&in, 4096 * 4);
  hipMalloc(&out, 4096 * 256 * 4);
  std::vector<float> h(4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int data = 0; data < 3; ++data) {
    for (auto &v : h) v = data == 0 ? 0.f : data == 1 ? 0.01f : (float)rand() / RAND_MAX * 2 - 1;
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
      const int grid = 256 * blocks_per_cu, iters = 20000;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)grid * 4 * iters * 8 * 4 * (2.0 * 32 * 32 * 2);
        if (rep == 2)
          printf("data=%s waves/SIMD=%d: %.2f ms  %.1f TF/s (%.1f%% of 157.3)\n",
                 data == 0 ? "zero" : data == 1 ? "const" : "random", blocks_per_cu, ms, flops / ms / 1e9,
                 100 * flops / ms / 1e9 / 157.3);
      }
    }
  }
  return 0;
}
