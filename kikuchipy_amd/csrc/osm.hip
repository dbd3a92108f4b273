// osm.hip - orientation similarity map of an indexing result
// (indexing/_orientation_similarity_map.py:30-152): per map point the mean, over its
// neighbours in a footprint, of the number of dictionary indices shared by the n best
// matches of the point and of the neighbour (|unique(a) & unique(b)|, np.intersect1d).
//
// One wave per map point and layer.  Lane l owns the l-th (l + 64-th, ...) index of the
// centre point's list; the neighbour's list is read by every lane (same address: one
// broadcast load from L2 per element).  Integer work on M x k x 4 bytes: latency-bound,
// microseconds for a 200 x 200 map; it can run on the best-k lists still resident from
// the sweep, so a map comes out of `finalize` without another upload.
#include "kernels.h"

namespace kpdi {

constexpr int OSM_THREADS = 256;
constexpr int OSM_MAX_FP = 64;

struct OsmFootprint {
  int n;                     // footprint points, row-major order of the non-zero elements
  int center_index;          // which of them is "the centre" (reference: v[center_index])
  short dy[OSM_MAX_FP], dx[OSM_MAX_FP];
};

__global__ __launch_bounds__(OSM_THREADS) void osm_kernel(const int *idx, int ny, int nx, int keep_n, int n_best,
                                                          int n_layers, OsmFootprint fp, int normalize, float *out) {
  const int lane = threadIdx.x & 63;
  const int64_t job = (int64_t)blockIdx.x * (OSM_THREADS / 64) + (threadIdx.x >> 6);
  const int64_t n_points = (int64_t)ny * nx;
  if (job >= n_points * n_layers) return;
  const int p = (int)(job / n_layers), layer = (int)(job % n_layers);
  const int n = n_best - layer;
  const int y = p / nx, x = p % nx;
  // the footprint values v: flat index of every footprint point, -1 outside the map
  auto at = [&](int f) {
    const int yy = y + fp.dy[f], xx = x + fp.dx[f];
    return (yy >= 0 && yy < ny && xx >= 0 && xx < nx) ? yy * nx + xx : -1;
  };
  const int centre = at(fp.center_index);
  float result = __builtin_nanf("");
  if (centre >= 0) {
    const int *a = idx + (int64_t)centre * keep_n;
    long long total = 0;
    int neighbours = 0;
    for (int f = 0; f < fp.n; ++f) {
      const int q = at(f);
      if (q < 0 || q == centre) continue;
      const int *b = idx + (int64_t)q * keep_n;
      int mine = 0;
      for (int i = lane; i < n; i += 64) {
        const int v = a[i];
        bool first = true;  // count every distinct value of `a` once
        for (int j = 0; j < i; ++j) first &= a[j] != v;
        bool found = false;
        for (int j = 0; j < n; ++j) found |= b[j] == v;
        mine += (first && found) ? 1 : 0;
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
      total += mine;
      ++neighbours;
    }
    if (neighbours > 0) {
      double v = (double)total / (double)neighbours;  // np.nanmean of the integer counts
      if (normalize) v /= (double)n;
      result = (float)v;
    }
  }
  if (lane == 0) out[(int64_t)p * n_layers + layer] = result;
}

hipError_t launch_osm(const int *idx, int ny, int nx, int keep_n, int n_best, int from_n_best, const int *offsets,
                      int n_fp, int center_index, int normalize, float *out, hipStream_t s) {
  if (n_fp < 1 || n_fp > OSM_MAX_FP || center_index < 0 || center_index >= n_fp) return hipErrorInvalidValue;
  OsmFootprint fp;
  fp.n = n_fp;
  fp.center_index = center_index;
  for (int f = 0; f < n_fp; ++f) {
    fp.dy[f] = (short)offsets[2 * f];
    fp.dx[f] = (short)offsets[2 * f + 1];
  }
  const int n_layers = n_best - from_n_best + 1;
  const int64_t jobs = (int64_t)ny * nx * n_layers;
  const int64_t blocks = (jobs + OSM_THREADS / 64 - 1) / (OSM_THREADS / 64);
  hipLaunchKernelGGL(osm_kernel, dim3((unsigned)blocks), dim3(OSM_THREADS), 0, s, idx, ny, nx, keep_n, n_best, n_layers,
                     fp, normalize, out);
  return hipGetLastError();
}

}  // namespace kpdi
