// Chamfer distance for gfx950: brute-force bidirectional nearest neighbour and
// its gradient (replaces chamfer_extension/chamfer.cu:12-195 of the reference).
//
// Forward is O(n*m) fp32 VALU work.  One 1024-thread workgroup owns 64*Q query
// points (Q per lane, in registers); each of its 16 waves scans one sixteenth
// of the target cloud.  Target coordinates are wave-uniform, so they are fetched
// with scalar loads (s_load_dwordx*) straight into SGPRs — no LDS staging and
// no barriers in the scan loop.  The 16 partial (min, argmin) pairs are
// merged through LDS in ascending target order, which preserves the
// reference's tie rule (strict '<' while scanning ascending: lowest index
// wins, chamfer.cu:36,46,126).
#include "ct_common.h"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kUnroll = 8;     // targets per scalar-load batch
constexpr int kWaves = 16;     // waves per workgroup = target slices (B*n/256 query groups alone would
                               // leave most of the 256 CUs idle: 16 slices give ~2k waves at n = 16k)

// kQ = queries per lane (4, or 2 when 256-query workgroups would leave CUs without one).
template <int kQ>
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

  // Queries sit in registers as float pairs so the distance arithmetic runs on the packed fp32 pipe
  // (v_pk_add/mul/fma_f32: two queries per issue slot).  The scan only tracks, per query, the running minimum and
  // the 8-target block it came from (one compare per block instead of one per target); the index inside the block
  // is recovered afterwards by recomputing those 8 distances with the same expression.
  f2 qx[kQ / 2], qy[kQ / 2], qz[kQ / 2];
  float best[kQ];
  int blk[kQ];
#pragma unroll
  for (int k = 0; k < kQ; ++k) {
    int i = min(q0 + k * 64 + lane, n - 1);
    qx[k >> 1][k & 1] = q[i * 3 + 0];
    qy[k >> 1][k & 1] = q[i * 3 + 1];
    qz[k >> 1][k & 1] = q[i * 3 + 2];
    best[k] = __builtin_inff();
    blk[k] = 0;
  }
  const int per = (m + kWaves - 1) / kWaves;
  const int j_beg = min(wave * per, m);
  const int j_end = min(m, j_beg + per);
  auto dist2 = [&](int p, float tx, float ty, float tz) -> f2 {
    const f2 dx = f2{tx, tx} - qx[p], dy = f2{ty, ty} - qy[p], dz = f2{tz, tz} - qz[p];
    return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
  };
  auto scan_block = [&](int j, const float* buf) {
#pragma unroll
    for (int p = 0; p < kQ / 2; ++p) {
      f2 mn = dist2(p, buf[0], buf[1], buf[2]);
#pragma unroll
      for (int u = 1; u < kUnroll; ++u) {
        const f2 d = dist2(p, buf[3 * u], buf[3 * u + 1], buf[3 * u + 2]);
        mn.x = fminf(mn.x, d.x), mn.y = fminf(mn.y, d.y);
      }
      if (mn.x < best[2 * p]) best[2 * p] = mn.x, blk[2 * p] = j;          // strict: the earlier block keeps a tie
      if (mn.y < best[2 * p + 1]) best[2 * p + 1] = mn.y, blk[2 * p + 1] = j;
    }
  };
  int j = j_beg;
  const float* tp = t + (size_t)j_beg * 3;   // wave-uniform address -> scalar loads
  for (; j + kUnroll <= j_end; j += kUnroll, tp += 3 * kUnroll) {
    float buf[3 * kUnroll];
#pragma unroll
    for (int u = 0; u < 3 * kUnroll; ++u) buf[u] = tp[u];
    scan_block(j, buf);
  }
  if (j < j_end) {                           // ragged last block: repeat the slice's last target
    float buf[3 * kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int o = 3 * (min(j + u, j_end - 1) - j);
      buf[3 * u] = tp[o], buf[3 * u + 1] = tp[o + 1], buf[3 * u + 2] = tp[o + 2];
    }
    scan_block(j, buf);
  }
  int bi[kQ];
#pragma unroll
  for (int k = 0; k < kQ; ++k) {
    bi[k] = 0;
    if (j_beg < j_end) {
#pragma unroll
      for (int u = kUnroll - 1; u >= 0; --u) {   // descending, so the lowest index among equal distances wins
        const int jj = min(blk[k] + u, j_end - 1);
        const float dx = t[jj * 3 + 0] - qx[k >> 1][k & 1], dy = t[jj * 3 + 1] - qy[k >> 1][k & 1],
                    dz = t[jj * 3 + 2] - qz[k >> 1][k & 1];
        if (fmaf(dz, dz, fmaf(dy, dy, dx * dx)) == best[k]) bi[k] = jj;
      }
    }
  }
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
  CT_CLEAR_ERROR();
  auto launch = [&](const float* q, const float* t, float* d, int32_t* i, int nq, int nt) {
    if ((size_t)B * ((nq + 255) / 256) >= 256 /* CUs on MI355X */)
      hipLaunchKernelGGL(nn_kernel<4>, dim3((nq + 255) / 256, B), dim3(64 * kWaves), 0, st, q, t, d, i, nq, nt);
    else
      hipLaunchKernelGGL(nn_kernel<2>, dim3((nq + 127) / 128, B), dim3(64 * kWaves), 0, st, q, t, d, i, nq, nt);
  };
  launch(xyz1, xyz2, dist1, idx1, n, m);
  launch(xyz2, xyz1, dist2, idx2, m, n);
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
