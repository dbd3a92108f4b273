// Micro-benchmark: does a co-resident LOADER wave (LDS-DMA only, no MFMA) slow down the
// MFMA wave of the same SIMD?  512-thread workgroups: waves 0-3 issue back-to-back
// v_mfma_f32_32x32x2_f32 (as mfma_f32.hip), waves 4-7 stream data into LDS with
// global_load_lds_dwordx4 at the match kernel's rate (48 KB per 128 MFMAs per CU).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_loader mfma_loader.hip && ./mfma_loader
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int READS>  // MODE 0: loaders idle, 1: loaders stream, 2: compute waves issue the loads themselves; READS: 24 ds_read_b128 per step in the compute waves
__global__ __launch_bounds__(512, 2) void k(const float *in, const char *big, float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wv < 4) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) {
      a[i] = in[(threadIdx.x + 64 * i) & 4095];
      b[i] = in[(threadIdx.x * 7 + 64 * i + 1) & 4095];
    }
    const char *src = big + (size_t)blockIdx.x * (1 << 20) + wv * 12288 + lane * 16;
    for (int it = 0; it < iters; ++it) {  // one "step": 128 MFMAs
#pragma unroll
      for (int u = 0; u < 16; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[u & 7]), "v"(b[(u + i) & 7]));
          if (READS == 2 && (u & 1) == 0 && (i == 1 || i == 4 || i == 7)) {  // spread: one read per ~3 MFMAs
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const f32x4 t = *(const f32x4 *)(smem + ((u * 3 + i) & 31) * 4096 + lane * 16 + (it & 1) * 1024);
            a[(u + i) & 7] += t[0] * 1e-30f;
          }
        }
        if (READS == 1 && (u & 1) == 0) {  // clustered: 3 fragment reads after every 16 MFMAs
          typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const f32x4 t = *(const f32x4 *)(smem + ((u * 3 + q) & 31) * 4096 + lane * 16 + (it & 1) * 1024);
            a[(u + q) & 7] += t[0] * 1e-30f;
          }
        }
        if (MODE == 2 && u < 12) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + ((it * 12 + u) & 63) * 1024 * 16),
                                           (__attribute__((address_space(3))) void *)(smem + (wv * 12 + u) * 1024), 16, 0, 0);
        }
      }
      if (MODE == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < 8; ++i)
      for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + (threadIdx.x & 255)] = s;
    if (MODE == 1) ((volatile int *)(smem + 144 * 1024 - 16))[0] = 1;  // tell the loaders to stop
  } else if (MODE == 1) {
    const int lw = wv - 4;
    const char *src = big + (size_t)blockIdx.x * (1 << 20) + lw * 12288 + lane * 16;
    // the compute waves need iters * 8192 cycles; issue 12 pieces per 8192 cycles
    volatile int *done = (volatile int *)(smem + 144 * 1024 - 16);
    if (threadIdx.x == 256) *done = 0;
    for (int it = 0; *done == 0; ++it) {  // stream as long as the compute waves run (>= the real rate)
#pragma unroll
      for (int u = 0; u < 11; ++u)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + ((it * 12 + u) & 63) * 1024 * 16),
                                         (__attribute__((address_space(3))) void *)(smem + (lw * 12 + u) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_sleep(40);
    }
  }
}

template <int MODE, int READS>
void run(const char *name, const float *in, const char *big, float *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256, iters = 2000;
  hipFuncSetAttribute((const void *)k<MODE, READS>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, READS>), dim3(grid), dim3(512), 144 * 1024, 0, in, big, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 128 * (2.0 * 32 * 32 * 2);
    if (rep == 2) printf("%-34s %.2f ms  %.1f TF/s (%.1f%% of 157.3)\n", name, ms, flops / ms / 1e9, 100 * flops / ms / 1e9 / 157.3);
  }
}

int main() {
  float *in, *out;
  char *big;
  hipMalloc(&in, 4096 * 4);
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&big, (size_t)260 << 20);
  hipMemset(big, 0, (size_t)260 << 20);
  std::vector<float> h(4096);
  for (auto &v : h) v = (float)rand() / RAND_MAX * 2 - 1;
  hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  run<0, 0>("MFMA waves + idle loader waves", in, big, out);
  run<1, 0>("MFMA waves + streaming loader waves", in, big, out);
  run<2, 0>("MFMA waves issue the loads themselves", in, big, out);
  run<0, 1>("MFMA+ds_read waves, idle loaders", in, big, out);
  run<1, 1>("MFMA+ds_read waves, streaming loaders", in, big, out);
  run<2, 1>("MFMA+ds_read waves issue the loads", in, big, out);
  run<0, 2>("MFMA+spread ds_read, idle loaders", in, big, out);
  run<1, 2>("MFMA+spread ds_read, streaming loaders", in, big, out);
  run<0, 0>("MFMA waves + idle loader waves (again)", in, big, out);
  return 0;
}
