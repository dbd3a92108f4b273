// UniformDivisor::divide (csrc/preproc.hip) against the compiler's IEEE float32 division, bit for bit:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probes/div_check.hip -o /tmp/div_check && /tmp/div_check
// numerators: what rescale() sees (differences of pattern values / filtered values, |n| <= 2^17, incl. 0 and values with
// all 24 significant bits); divisors: ranges of such values (> 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

struct UniformDivisor {
  float d, rcp;
  bool fast;
  __device__ explicit UniformDivisor(float divisor) : d(divisor) {
    const float a = fabsf(divisor);
    fast = a > 0x1p-40f && a < 0x1p40f;
    float r = __builtin_amdgcn_rcpf(divisor);
    const float e = __builtin_fmaf(-divisor, r, 1.f);
    rcp = __builtin_fmaf(e, r, r);
  }
  __device__ float divide(float n) const {
    if (!fast) return n / d;
    float q = n * rcp;
    float r = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(r, rcp, q);
    r = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(r, rcp, q);
  }
};

__device__ uint32_t rng(uint64_t &s) {
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  return (uint32_t)(s >> 33) ^ (uint32_t)s;
}

__global__ void check(unsigned long long *bad, unsigned long long *total, float *worst) {
  uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
  unsigned long long nbad = 0;
  for (int it = 0; it < 4096; ++it) {
    // divisor: integer-valued (u8 / u16 ranges), or a float with random mantissa in [2^-10, 2^17)
    const uint32_t a = rng(s), b = rng(s);
    float d;
    if (a & 1) d = (float)(1 + (b % 65535));
    else d = __uint_as_float(((117u + (b % 27u)) << 23) | (a >> 9));
    const UniformDivisor u(d);
    for (int k = 0; k < 64; ++k) {
      const uint32_t c = rng(s), e = rng(s);
      float n;
      switch (c & 3) {
        case 0: n = (float)(e % 65536); break;
        case 1: n = __uint_as_float(((100u + (e % 44u)) << 23) | (c >> 9)); break;
        case 2: n = d * (float)(e % 256) / 255.f; break;  // near the top of the range
        default: n = (float)(e % 65536) - __uint_as_float(((110u + (c % 20u)) << 23) | (e >> 9)); break;
      }
      if (c & 4) n = -n;
      const float want = n / d, got = u.divide(n);
      // (+0 vs -0 for a zero numerator is not counted: rescale() casts the quotient to an integer)
      if (__float_as_uint(want) != __float_as_uint(got) && !(want == 0.f && got == 0.f)) {
        ++nbad;
        worst[0] = n;
        worst[1] = d;
      }
    }
  }
  atomicAdd(bad, nbad);
  atomicAdd(total, 4096ull * 64ull);
}

int main() {
  unsigned long long *bad, *total, h[2];
  float *worst, hw[2] = {0, 0};
  (void)hipMalloc(&bad, 8); (void)hipMalloc(&total, 8); (void)hipMalloc(&worst, 8);
  (void)hipMemset(bad, 0, 8); (void)hipMemset(total, 0, 8); (void)hipMemset(worst, 0, 8);
  hipLaunchKernelGGL(check, dim3(1024), dim3(256), 0, 0, bad, total, worst);
  (void)hipMemcpy(&h[0], bad, 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(&h[1], total, 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hw, worst, 8, hipMemcpyDeviceToHost);
  printf("UniformDivisor vs '/': %llu mismatches in %llu divisions", h[0], h[1]);
  if (h[0]) printf(" (e.g. %a / %a)", hw[0], hw[1]);
  printf("\n");
  return h[0] != 0;
}
