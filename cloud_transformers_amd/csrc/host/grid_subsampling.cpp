// Voxel-grid subsampling on the host (include/cloudct_host.h).  Sort-based: every point gets its cell key, an index
// permutation is sorted by (key, input position) — stable, so the float sums run in input order like the reference's
// running sums — and each run of equal keys is reduced to one output row.  No hash map, no per-cell allocations.
#include "../../../include/cloudct_host.h"

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <numeric>
#include <vector>

extern "C" int64_t ct_grid_subsample(const float* points, const float* features, const int32_t* classes, int64_t N, int fdim,
                                     int ldim, float dl, float* out_points, float* out_features, int32_t* out_classes) {
  if (!points || !out_points || N < 0 || fdim < 0 || ldim < 0 || !(dl > 0.0f)) return -1;
  if ((fdim > 0 && (!features || !out_features)) || (ldim > 0 && (!classes || !out_classes))) return -1;
  if (N == 0) return 0;
  float lo[3], hi[3];
  for (int a = 0; a < 3; ++a) lo[a] = hi[a] = points[a];
  for (int64_t i = 1; i < N; ++i)
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::min(lo[a], points[3 * i + a]);
      hi[a] = std::max(hi[a], points[3 * i + a]);
    }
  const float inv = 1 / dl;
  float org[3];
  for (int a = 0; a < 3; ++a) org[a] = std::floor(lo[a] * inv) * dl;
  const size_t nx = (size_t)std::floor((hi[0] - org[0]) / dl) + 1;
  const size_t ny = (size_t)std::floor((hi[1] - org[1]) / dl) + 1;
  std::vector<size_t> key((size_t)N);
  for (int64_t i = 0; i < N; ++i) {
    const size_t ix = (size_t)std::floor((points[3 * i + 0] - org[0]) / dl);
    const size_t iy = (size_t)std::floor((points[3 * i + 1] - org[1]) / dl);
    const size_t iz = (size_t)std::floor((points[3 * i + 2] - org[2]) / dl);
    key[(size_t)i] = ix + nx * iy + nx * ny * iz;
  }
  std::vector<int64_t> order((size_t)N);
  std::iota(order.begin(), order.end(), (int64_t)0);
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return key[(size_t)a] < key[(size_t)b]; });
  int64_t M = 0;
  std::vector<float> fsum((size_t)fdim);
  for (int64_t s = 0; s < N;) {
    int64_t e = s;
    float px = 0, py = 0, pz = 0;
    std::fill(fsum.begin(), fsum.end(), 0.0f);
    for (int l = 0; l < ldim; ++l) out_classes[M * ldim + l] = classes[order[(size_t)s] * ldim + l];
    const size_t k = key[(size_t)order[(size_t)s]];
    while (e < N && key[(size_t)order[(size_t)e]] == k) {
      const int64_t i = order[(size_t)e];
      px += points[3 * i + 0];
      py += points[3 * i + 1];
      pz += points[3 * i + 2];
      for (int f = 0; f < fdim; ++f) fsum[(size_t)f] += features[i * fdim + f];
      for (int l = 0; l < ldim; ++l) out_classes[M * ldim + l] = std::max(out_classes[M * ldim + l], classes[i * ldim + l]);
      ++e;
    }
    const float w = (float)(1.0 / (double)(e - s));       // the reference multiplies by (1.0 / count) narrowed to float
    out_points[3 * M + 0] = px * w;
    out_points[3 * M + 1] = py * w;
    out_points[3 * M + 2] = pz * w;
    const float cnt = (float)(e - s);
    for (int f = 0; f < fdim; ++f) out_features[M * fdim + f] = fsum[(size_t)f] / cnt;
    ++M;
    s = e;
  }
  return M;
}
