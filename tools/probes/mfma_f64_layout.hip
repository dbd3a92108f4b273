// Operand / result layout of v_mfma_f64_16x16x4_f64 (what csrc/preproc.hip's banded-matrix correlation relies on):
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f64_layout.hip -o /tmp/mfma_f64_layout && /tmp/mfma_f64_layout
// Hypothesis: lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16]; it receives D[4 (l / 16) + v][l % 16], v = 0 .. 3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
// the kernel hands lane l's operands from plain per-lane arrays; the host tries the candidate layouts
__global__ void k(const double *a_lane, const double *b_lane, double *D) {
  const int l = threadIdx.x;
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a_lane[l], b_lane[l], c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) D[l * 4 + v] = c[v];
}
int main() {
  double A[16][4], B[4][16], ha[64], hb[64], hD[256], *da, *db, *dD;
  for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 4; ++kk) A[i][kk] = sin(i * 1.3 + kk * 7.1) + 2;
  for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < 16; ++j) B[kk][j] = cos(j * 0.7 + kk * 3.3) - 3;
  (void)hipMalloc(&da, sizeof ha); (void)hipMalloc(&db, sizeof hb); (void)hipMalloc(&dD, sizeof hD);
  int found = 0;
  for (int la = 0; la < 2; ++la)      // A: lane -> (i, k) = (l % 16, l / 16) | (l / 4, l % 4)
    for (int lb = 0; lb < 2; ++lb) {  // B: lane -> (k, j) = (l / 16, l % 16) | (l % 4, l / 4)
      for (int l = 0; l < 64; ++l) {
        ha[l] = la == 0 ? A[l % 16][l / 16] : A[l / 4][l % 4];
        hb[l] = lb == 0 ? B[l / 16][l % 16] : B[l % 4][l / 4];
      }
      (void)hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dD);
      (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
      for (int ld = 0; ld < 4; ++ld) {  // D: (lane, v) -> (i, j)
        double worst = 0;
        for (int l = 0; l < 64; ++l)
          for (int v = 0; v < 4; ++v) {
            int i, j;
            switch (ld) {
              case 0: i = 4 * (l / 16) + v; j = l % 16; break;
              case 1: i = l % 16; j = 4 * (l / 16) + v; break;
              case 2: i = (l / 16) + 4 * v; j = l % 16; break;
              default: i = l % 16; j = (l / 16) + 4 * v; break;
            }
            double want = 0;
            for (int kk = 0; kk < 4; ++kk) want += A[i][kk] * B[kk][j];
            worst = fmax(worst, fabs(want - hD[l * 4 + v]));
          }
        if (worst < 1e-12) {
          printf("MATCH: A layout %d, B layout %d, D layout %d\n", la, lb, ld);
          ++found;
        }
      }
    }
  printf("A: 0 = (i, k) = (l %% 16, l / 16), 1 = (l / 4, l %% 4); B: 0 = (k, j) = (l / 16, l %% 16), 1 = (l %% 4, l / 4);\n"
         "D (lane, v) -> (i, j): 0 = (4 (l / 16) + v, l %% 16), 1 = (l %% 16, 4 (l / 16) + v), 2 = (l / 16 + 4 v, l %% 16), 3 = (l %% 16, l / 16 + 4 v)\n");
  return found ? 0 : 1;
}
