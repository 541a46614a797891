// Driver that exposes the REFERENCE's own grid_subsampling (compiled from /root/reference where it lies, see oracle/Makefile
// target _ref) behind a C function, for tests/test_grid_subsampling.py — TEST INFRASTRUCTURE ONLY.  Nothing of the
// reference is copied: this file only includes its header and marshals arrays into its std::vector interface, as its
// Python wrapper does (cpp_wrappers/cpp_subsampling/wrapper.cpp).
#include "grid_subsampling/grid_subsampling.h"

#include <cstdint>

extern "C" int64_t ref_grid_subsample(const float* points, const float* features, const int32_t* classes, int64_t N, int fdim,
                                      int ldim, float dl, float* out_points, float* out_features, int32_t* out_classes) {
  std::vector<PointXYZ> op((size_t)N), sp;
  for (int64_t i = 0; i < N; ++i) op[(size_t)i] = PointXYZ(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
  std::vector<float> of, sf;
  std::vector<int> oc, sc;
  if (fdim > 0) of.assign(features, features + N * fdim);
  if (ldim > 0) oc.assign(classes, classes + N * ldim);
  grid_subsampling(op, sp, of, sf, oc, sc, dl, 0);
  const int64_t M = (int64_t)sp.size();
  for (int64_t i = 0; i < M; ++i) {
    out_points[3 * i] = sp[(size_t)i].x;
    out_points[3 * i + 1] = sp[(size_t)i].y;
    out_points[3 * i + 2] = sp[(size_t)i].z;
  }
  for (size_t i = 0; i < sf.size(); ++i) out_features[i] = sf[i];
  for (size_t i = 0; i < sc.size(); ++i) out_classes[i] = sc[i];
  return M;
}
