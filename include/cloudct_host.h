/*
 * cloudct_host.h — C ABI of libcloudct_host.so: the CPU-side data preparation of the MHCT pipelines (plain C++17, no GPU
 * runtime).  Same conventions as cloudct.h: extern "C", plain pointers and sizes, caller-owned buffers, no global state.
 *
 * ct_grid_subsample — barycentre voxel-grid subsampling of a labelled, coloured cloud: what the reference's S3DIS
 * "closer look" loader calls through its C++ extension (datasets/s3dis_closer.py:10-31,192 ->
 * cpp_wrappers/cpp_subsampling/wrapper.cpp `compute` -> grid_subsampling/grid_subsampling.cpp:4-104):
 *   origin  = floor(min corner * (1 / dl)) * dl                  (per axis, float arithmetic)
 *   cell    = floor((p - origin) / dl) per axis; cells are keyed ix + nx * iy + nx * ny * iz
 *   point   = (sum of the cell's points, accumulated in float in input order) * (1 / count)
 *   feature = (sum of the cell's feature rows) / count
 *   class   = per label column, the LARGEST label value present in the cell — the reference takes max_element over a
 *             (label -> count) map, which compares keys before counts (grid_subsampling.cpp:100); kept as is
 * points f32[N,3]; features f32[N,fdim] | NULL (fdim = 0); classes i32[N,ldim] | NULL (ldim = 0); dl > 0.
 * Outputs must hold N rows (the worst case); returns the number of occupied cells M, rows 0..M-1 are written in
 * ascending cell-key order (the reference's order is its hash map's: compare as sets), or -1 on a bad argument.
 */
#ifndef CLOUDCT_HOST_H
#define CLOUDCT_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int64_t ct_grid_subsample(const float* points, const float* features, const int32_t* classes, int64_t N, int fdim, int ldim,
                          float dl, float* out_points, float* out_features, int32_t* out_classes);
#ifdef __cplusplus
}
#endif
#endif
