// Chamfer distance for gfx950: brute-force bidirectional nearest neighbour and
// its gradient (replaces chamfer_extension/chamfer.cu:12-195 of the reference).
//
// Forward is O(n*m) fp32 VALU work.  One 1024-thread workgroup owns 64*Q query
// points (Q per lane, in registers); each of its 4 waves scans one quarter of
// the target cloud.  Target coordinates are wave-uniform, so they are fetched
// with scalar loads (s_load_dwordx*) straight into SGPRs — no LDS staging and
// no barriers in the scan loop.  The four partial (min, argmin) pairs are
// merged through LDS in ascending target order, which preserves the
// reference's tie rule (strict '<' while scanning ascending: lowest index
// wins, chamfer.cu:36,46,126).
#include "ct_common.h"

namespace {

constexpr int kQ = 4;          // queries per lane
constexpr int kUnroll = 8;     // targets per scalar-load batch
constexpr int kWaves = 16;     // waves per workgroup = target slices (B*n/256 query groups alone would
                               // leave most of the 256 CUs idle: 16 slices give ~2k waves at n = 16k)

__global__ void __launch_bounds__(64 * kWaves)
nn_kernel(const float* __restrict__ q, const float* __restrict__ t, float* __restrict__ dist,
          int* __restrict__ idx, int n, int m) {
  __shared__ float s_best[kWaves][64 * kQ];
  __shared__ int s_idx[kWaves][64 * kQ];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q0 = blockIdx.x * (64 * kQ);
  q += (size_t)b * n * 3;
  t += (size_t)b * m * 3;

  float qx[kQ], qy[kQ], qz[kQ], best[kQ];
  int bi[kQ];
#pragma unroll
  for (int k = 0; k < kQ; ++k) {
    int i = min(q0 + k * 64 + lane, n - 1);
    qx[k] = q[i * 3 + 0];
    qy[k] = q[i * 3 + 1];
    qz[k] = q[i * 3 + 2];
    best[k] = __builtin_inff();
    bi[k] = 0;
  }
  const int per = (m + kWaves - 1) / kWaves;
  const int j_beg = wave * per;
  const int j_end = min(m, j_beg + per);
  auto visit = [&](int j, float tx, float ty, float tz) {
#pragma unroll
    for (int k = 0; k < kQ; ++k) {
      float dx = tx - qx[k], dy = ty - qy[k], dz = tz - qz[k];
      float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
      if (d < best[k]) {
        best[k] = d;
        bi[k] = j;
      }
    }
  };
  int j = j_beg;
  const float* tp = t + (size_t)j_beg * 3;   // wave-uniform address -> scalar loads
  for (; j + kUnroll <= j_end; j += kUnroll, tp += 3 * kUnroll) {
    float buf[3 * kUnroll];
#pragma unroll
    for (int u = 0; u < 3 * kUnroll; ++u) buf[u] = tp[u];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) visit(j + u, buf[3 * u], buf[3 * u + 1], buf[3 * u + 2]);
  }
  for (; j < j_end; ++j, tp += 3) visit(j, tp[0], tp[1], tp[2]);
#pragma unroll
  for (int k = 0; k < kQ; ++k) {
    s_best[wave][k * 64 + lane] = best[k];
    s_idx[wave][k * 64 + lane] = bi[k];
  }
  __syncthreads();
  if (threadIdx.x < 64 * kQ) {
    float bb = s_best[0][threadIdx.x];
    int ii = s_idx[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) {
      float o = s_best[w][threadIdx.x];
      if (o < bb) {   // strict: the earlier (lower-index) range wins ties
        bb = o;
        ii = s_idx[w][threadIdx.x];
      }
    }
    int i = q0 + threadIdx.x;
    if (i < n) {
      dist[(size_t)b * n + i] = bb;
      idx[(size_t)b * n + i] = ii;
    }
  }
}

// chamfer.cu:155-174: g = 2*grad_dist[i]; grad_a[i] += g (a_i - b_idx);  grad_b[idx] -= g (a_i - b_idx)
__global__ void nn_grad_kernel(const float* __restrict__ a, const float* __restrict__ bpts,
                               const float* __restrict__ g_dist, const int* __restrict__ idx,
                               float* g_a, float* g_b, int n, int m) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t ia = ((size_t)b * n + i) * 3;
  const int j = idx[(size_t)b * n + i];
  const size_t ib = ((size_t)b * m + j) * 3;
  const float g = g_dist[(size_t)b * n + i] * 2.0f;
  const float dx = g * (a[ia + 0] - bpts[ib + 0]);
  const float dy = g * (a[ia + 1] - bpts[ib + 1]);
  const float dz = g * (a[ia + 2] - bpts[ib + 2]);
  atomicAdd(&g_a[ia + 0], dx);
  atomicAdd(&g_a[ia + 1], dy);
  atomicAdd(&g_a[ia + 2], dz);
  atomicAdd(&g_b[ib + 0], -dx);
  atomicAdd(&g_b[ib + 1], -dy);
  atomicAdd(&g_b[ib + 2], -dz);
}

}  // namespace

extern "C" {

int ct_chamfer_fwd(const float* xyz1, const float* xyz2, float* dist1, float* dist2, int32_t* idx1, int32_t* idx2,
                   int B, int n, int m, ct_stream_t s) {
  if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2 || B <= 0 || n <= 0 || m <= 0 || B > 65535) return CT_EINVAL;
  hipStream_t st = (hipStream_t)s;
  const int per = 64 * kQ;
  CT_CLEAR_ERROR();
  hipLaunchKernelGGL(nn_kernel, dim3((n + per - 1) / per, B), dim3(64 * kWaves), 0, st, xyz1, xyz2, dist1, idx1, n, m);
  hipLaunchKernelGGL(nn_kernel, dim3((m + per - 1) / per, B), dim3(64 * kWaves), 0, st, xyz2, xyz1, dist2, idx2, m, n);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

int ct_chamfer_bwd(const float* xyz1, const float* xyz2, const float* g_dist1, const float* g_dist2,
                   const int32_t* idx1, const int32_t* idx2, float* g_xyz1, float* g_xyz2,
                   int B, int n, int m, ct_stream_t s) {
  if (!xyz1 || !xyz2 || !g_dist1 || !g_dist2 || !idx1 || !idx2 || !g_xyz1 || !g_xyz2 || B <= 0 || n <= 0 || m <= 0 || B > 65535)
    return CT_EINVAL;
  hipStream_t st = (hipStream_t)s;
  if (hipMemsetAsync(g_xyz1, 0, (size_t)B * n * 3 * 4, st) != hipSuccess) return CT_ELAUNCH;
  if (hipMemsetAsync(g_xyz2, 0, (size_t)B * m * 3 * 4, st) != hipSuccess) return CT_ELAUNCH;
  CT_CLEAR_ERROR();
  hipLaunchKernelGGL(nn_grad_kernel, dim3((n + 255) / 256, B), dim3(256), 0, st, xyz1, xyz2, g_dist1, idx1, g_xyz1, g_xyz2, n, m);
  hipLaunchKernelGGL(nn_grad_kernel, dim3((m + 255) / 256, B), dim3(256), 0, st, xyz2, xyz1, g_dist2, idx2, g_xyz2, g_xyz1, m, n);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

}  // extern "C"
