// Does v_mfma_f64_16x16x4_f64 run beside the vector ALU work of ANOTHER wave of the same SIMD on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f64_overlap.hip -o /tmp/overlap && /tmp/overlap
// One 512-thread workgroup per CU (100 KB of LDS requested), i.e. two waves per SIMD: waves 0-3 run chain A, waves 4-7 chain B
// (or idle).  If the pair takes max(A, B) the two pipes overlap; if it takes A + B they are one resource.
// Written to decide whether preproc.hip's correlations belong on the float64 matrix pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
enum { IDLE = 0, MFMA64 = 1, VALU32 = 2, VALU64 = 3, MFMA32 = 4, MFMA64_NOP = 5, MFMA64_SLEEP = 6 };

__device__ __forceinline__ void chain(int kind, int iters, double *sink) {
  const int lane = threadIdx.x & 63;
  if (kind == MFMA64) {
    f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    *sink = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (kind == MFMA64_NOP || kind == MFMA64_SLEEP) {
    // the same chain, but the wave steps aside (scalar no-ops / a sleep) while each MFMA occupies the pipe: does the vector
    // ALU port then go to the other wave?
    f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
#define STEP_ASIDE()                                    \
  if (kind == MFMA64_NOP) {                             \
    asm volatile("s_nop 15");                         \
    asm volatile("s_nop 15");                         \
    asm volatile("s_nop 15");                         \
  } else {                                              \
    __builtin_amdgcn_s_sleep(1);                        \
  }
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      STEP_ASIDE();
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      STEP_ASIDE();
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      STEP_ASIDE();
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
      STEP_ASIDE();
    }
    *sink = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (kind == MFMA32) {
    f32x16 c0 = {0}, c1 = c0;
    const float a = 1.0f + lane * 1e-6f, b = 1.0f - lane * 1e-6f;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    }
    *sink = c0[0] + c1[1];
  } else if (kind == VALU32) {
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = lane + j;
    const float m = 1.0000001f, q = 1e-7f;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)  // 64 FMAs per iteration = 256 issue cycles, like 4 float64 MFMAs
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = __builtin_fmaf(x[j], m, q);
    float t = 0;
    for (int j = 0; j < 16; ++j) t += x[j];
    *sink = t;
  } else if (kind == VALU64) {
    double x[16];
    for (int j = 0; j < 16; ++j) x[j] = lane + j;
    const double m = 1.0000001, q = 1e-7;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = __builtin_fma(x[j], m, q);
    double t = 0;
    for (int j = 0; j < 16; ++j) t += x[j];
    *sink = t;
  }
}

__global__ __launch_bounds__(512) void probe(int kind_a, int kind_b, int iters, double *out) {
  extern __shared__ char lds[];
  const int w = threadIdx.x >> 6;
  double s = 0;
  chain(w < 4 ? kind_a : kind_b, iters, &s);
  if (s == 12345.678) out[blockIdx.x * 512 + threadIdx.x] = s + lds[threadIdx.x];
}

int main() {
  double *out;
  hipMalloc(&out, 256 * 512 * sizeof(double));
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char *name[] = {"idle", "mfma f64 16x16x4", "valu f32 fma", "valu f64 fma", "mfma f32 32x32x2", "mfma f64 + 3 s_nop 15", "mfma f64 + s_sleep 1"};
  const int pairs[][2] = {{MFMA64, IDLE}, {IDLE, VALU32}, {MFMA64, VALU32}, {IDLE, VALU64}, {MFMA64, VALU64}, {MFMA64, MFMA64},
                          {MFMA32, IDLE}, {MFMA32, VALU32}, {VALU32, VALU32},
                          {MFMA64_NOP, IDLE}, {MFMA64_NOP, VALU32}, {MFMA64_SLEEP, IDLE}, {MFMA64_SLEEP, VALU32}, {MFMA64_SLEEP, VALU64}};
  const int iters = 20000;
  for (auto &p : pairs) {
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe, dim3(256), dim3(512), 100 * 1024, 0, p[0], p[1], iters, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("waves 0-3: %-22s waves 4-7: %-18s %8.3f ms  (%.1f cycles per iteration at 2.4 GHz)\n", name[p[0]], name[p[1]], best,
           best * 1e-3 * 2.4e9 / iters);
  }
  return 0;
}
