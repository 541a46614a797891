// Per-head rigid transform + tanh ("lattice") for the MHCT blocks, forward and backward, fused:
//   p      = xyz + kscale * residual + shift_h                       (3-vector per point and head)
//   keys_n = (sum_c p_c * R_h[c][n]) * scales_h[n]    n < dim        (row vector times R; planes keep n < 2)
//   lattice = tanh(keys)
// replaces the reference's chain  add -> einsum('bhcp,hcn->bhnp') -> slice -> mul -> reshape -> tanh
// (layers/utils.py:25-34,53-61, layers/multihead_ct.py:93-97) and its autograd: ~15 elementwise /
// reduction launches per block become two.  R_h = so3_exponential_map(log_R_h) stays a tiny host-side
// torch op (H 3x3 matrices) so that its own autograd produces g_log_R from g_R.
//
// Layouts: xyz (B,3,N); residual (B,H*3,N); keys / lattice (B,H*dim,N); R (H,3,3); shift (H,3);
// scales (H,dim) or null; kscale: device scalar or null (= 1).
#include "ct_common.h"

namespace {

struct LatticeArgs {
  const float* xyz;
  const float* res;
  const float* R;
  const float* shift;
  const float* scales;   // nullable
  const float* kscale;   // nullable
  int B, H, N, dim;
  // ct_lattice_so3_fwd: the rotations come from their so3 parameters inside the launch (log_R non-null: R is an OUTPUT then,
  // written by the head's first workgroup for the backward), and the key statistics are finished by the launch's last
  // workgroup (ticket: a zeroed word the kernel leaves zeroed)
  const float* log_R;
  float* R_out;
  float so3_eps;
  unsigned* ticket;
  float* key_stats;
  double nkeys;
  int ppw;               // points of a row per workgroup of lattice_fwd_kernel (a multiple of 256)
};

// Rodrigues' formula of one head in double (so3_exp_fwd_kernel's arithmetic): R[9] row-major
__device__ __forceinline__ void so3_exp_one(const float* log_R, int h, float eps, float (&R)[9]) {
  const double x = log_R[h * 3 + 0], y = log_R[h * 3 + 1], z = log_R[h * 3 + 2];
  const double s = x * x + y * y + z * z;
  const double th = sqrt(s > (double)eps ? s : (double)eps);
  const double a = sin(th) / th, b = (1.0 - cos(th)) / (th * th);
  const double v[3] = {x, y, z};
  const double K[9] = {0, -z, y, z, 0, -x, -y, x, 0};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      R[i * 3 + j] = (float)((i == j ? 1.0 : 0.0) + a * K[i * 3 + j] + b * (v[i] * v[j] - (i == j ? s : 0.0)));
}

// wave64 sum over DPP (row_shr 1/2/4/8, row_bcast 15/31: no LDS traffic; ct_raster_hot.h has the integer forms), returned
// wave-uniform.  The 16 parameter partials of the backward took 96 ds_bpermute per thread with __shfl_xor.
__device__ __forceinline__ float wave_sum_f32(float v) {
#define CT_FSTEP(CTRL, ROWMASK) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWMASK, 0xf, false))
  CT_FSTEP(0x111, 0xf);
  CT_FSTEP(0x112, 0xf);
  CT_FSTEP(0x114, 0xf);
  CT_FSTEP(0x118, 0xf);
  CT_FSTEP(0x142, 0xa);
  CT_FSTEP(0x143, 0xc);
#undef CT_FSTEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// stat_parts (nullable): per-workgroup (sum, sum of squares) of the keys it wrote, [workgroup][2], for the lattice
// statistics the blocks report (mean / variance of the keys, layers/multihead_ct.py:109-112) — reduced by
// lattice_stats_kernel instead of two more passes over the keys.
__global__ void __launch_bounds__(256) lattice_fwd_kernel(LatticeArgs a, float* keys, float* lattice, float* stat_parts) {
  __shared__ float red[4][2];
  __shared__ float Rs[9];
  __shared__ unsigned s_last;
  // a workgroup walks `a.ppw` points of its (cloud, head) row, 256 at a time: the launch is a few hundred workgroups, not N / 256
  // per row — every workgroup ends with ONE arrival on the launch's ticket, and 2 048 same-address atomics (B8 H16 N4096) took
  // longer than the 17 MB the kernel moves (45 us; profiles/r4_model_breakdown.txt)
  const int n_beg = blockIdx.x * a.ppw, n_end = min(a.N, n_beg + a.ppw);
  const int h = blockIdx.y, b = blockIdx.z;
  if (a.log_R != nullptr) {            // kernel-uniform: this head's rotation from its so3 parameters (one lane, double)
    if (threadIdx.x == 0) {
      float Rl[9];
      so3_exp_one(a.log_R, h, a.so3_eps, Rl);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        Rs[i] = Rl[i];
        if (blockIdx.x == 0 && b == 0) a.R_out[h * 9 + i] = Rl[i];
      }
    }
    __syncthreads();
  }
  float s1 = 0.0f, s2 = 0.0f;
  for (int n = n_beg + (int)threadIdx.x; n < n_end; n += 256) {
    const float ks = a.kscale ? a.kscale[0] : 1.0f;
    float p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      p[c] = a.xyz[((size_t)b * 3 + c) * a.N + n] + ks * a.res[((size_t)(b * a.H + h) * 3 + c) * a.N + n] + a.shift[h * 3 + c];
    const float* R = a.log_R != nullptr ? Rs : a.R + h * 9;
    for (int j = 0; j < a.dim; ++j) {
      float k = p[0] * R[0 * 3 + j] + p[1] * R[1 * 3 + j] + p[2] * R[2 * 3 + j];
      if (a.scales) k *= a.scales[h * a.dim + j];
      const size_t o = ((size_t)(b * a.H + h) * a.dim + j) * a.N + n;
      keys[o] = k;
      lattice[o] = tanhf(k);
      s1 += k;
      s2 += k * k;
    }
  }
  if (stat_parts) {
    for (int o = 32; o > 0; o >>= 1) {
      s1 += __shfl_xor(s1, o, 64);
      s2 += __shfl_xor(s2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s1; red[threadIdx.x >> 6][1] = s2; }
    __syncthreads();
    if (threadIdx.x < 2) {
      const size_t wg = ((size_t)b * gridDim.y + h) * gridDim.x + blockIdx.x;
      const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
      if (a.ticket != nullptr) __hip_atomic_store(stat_parts + wg * 2 + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else stat_parts[wg * 2 + threadIdx.x] = v;
    }
    if (a.ticket != nullptr) {         // kernel-uniform: the launch's last workgroup reduces the partials (lattice_stats_kernel's sum)
      const unsigned total = gridDim.x * gridDim.y * gridDim.z;
      if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's two write-through stores have left
        const unsigned old = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == total - 1u;
        if (old == total - 1u) {
          __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      __syncthreads();
      if (s_last) {
        __shared__ double dred[2][4];
        double t1 = 0.0, t2 = 0.0;
        for (unsigned i = threadIdx.x; i < total; i += blockDim.x) {
          t1 += (double)stat_parts[(size_t)i * 2 + 0];
          t2 += (double)stat_parts[(size_t)i * 2 + 1];
        }
        for (int o = 32; o > 0; o >>= 1) {
          t1 += __shfl_xor(t1, o, 64);
          t2 += __shfl_xor(t2, o, 64);
        }
        if ((threadIdx.x & 63) == 0) { dred[0][threadIdx.x >> 6] = t1; dred[1][threadIdx.x >> 6] = t2; }
        __syncthreads();
        if (threadIdx.x == 0) {
          const double u1 = dred[0][0] + dred[0][1] + dred[0][2] + dred[0][3], u2 = dred[1][0] + dred[1][1] + dred[1][2] + dred[1][3];
          const double mean = u1 / a.nkeys;
          a.key_stats[0] = (float)mean;
          a.key_stats[1] = (float)(a.nkeys > 1.0 ? (u2 - a.nkeys * mean * mean) / (a.nkeys - 1.0) : 0.0);
        }
      }
    }
  }
}

// key_stats[0] = mean, key_stats[1] = unbiased variance of all n keys, from the per-workgroup partials (double accumulation)
__global__ void __launch_bounds__(256) lattice_stats_kernel(const float* stat_parts, size_t nwg, double n, float* key_stats) {
  __shared__ double red[2][4];
  double s1 = 0.0, s2 = 0.0;
  for (size_t i = threadIdx.x; i < nwg; i += blockDim.x) {
    s1 += (double)stat_parts[i * 2 + 0];
    s2 += (double)stat_parts[i * 2 + 1];
  }
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t1 = red[0][0] + red[0][1] + red[0][2] + red[0][3], t2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double mean = t1 / n;
    key_stats[0] = (float)mean;
    key_stats[1] = (float)(n > 1.0 ? (t2 - n * mean * mean) / (n - 1.0) : 0.0);
  }
}

// grid = (ceil(N/256), H, B).  Point-wise output: the cotangent gp of the moved point, stored per head in g_res;
// the finish half of lattice_bwd_tail_kernel then sums it over the heads into g_xyz and scales g_res by kscale in place (g_xyz used
// to be accumulated with B*3*N*H device-scope float atomics, most of this pass's time).  Parameter cotangents (g_R 9, g_shift 3, g_scales dim, g_kscale 1 per head) are
// reduced per workgroup and added with one atomic each.
// kBwdPts points per thread.  With __shfl_xor reductions (96 ds_bpermute per thread for the 16 parameter partials) four points
// per thread amortised them: 19.3 us at B8 H16 N4096; with the DPP sum above the reduction is 96 vector instructions and one
// point per thread — four times the workgroups — is fastest: 12.3 us at two, 10.6 us at one.
#ifndef CT_LATTICE_BWD_PTS
#define CT_LATTICE_BWD_PTS 1
#endif
constexpr int kBwdPts = CT_LATTICE_BWD_PTS;

__global__ void __launch_bounds__(256) lattice_bwd_kernel(LatticeArgs a, const float* lattice, const float* g_lattice,
                                                          const float* g_keys, float* g_xyz, float* g_res, float* g_R, float* g_shift,
                                                          float* g_scales, float* g_kscale, float* parts) {
  __shared__ float red[4][16];
  const int h = blockIdx.y, b = blockIdx.z;
  const float ks = a.kscale ? a.kscale[0] : 1.0f;
  const float* R = a.R + h * 9;
  // 16 per-head parameter partials: g_R[c][j] = p_c * gq_j (9), g_shift[c] = gp_c (3), g_scales[j] (3), g_kscale (1)
  float part[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) part[i] = 0.0f;
#pragma unroll
  for (int k = 0; k < kBwdPts; ++k) {
    const int n = (blockIdx.x * kBwdPts + k) * blockDim.x + threadIdx.x;
    if (n < a.N) {
      float p[3], r[3], gq[3] = {0, 0, 0}, gsc[3] = {0, 0, 0};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        r[c] = a.res[((size_t)(b * a.H + h) * 3 + c) * a.N + n];
        p[c] = a.xyz[((size_t)b * 3 + c) * a.N + n] + ks * r[c] + a.shift[h * 3 + c];
      }
      for (int j = 0; j < a.dim; ++j) {
        const size_t o = ((size_t)(b * a.H + h) * a.dim + j) * a.N + n;
        const float t = lattice[o];
        float gk = g_lattice ? g_lattice[o] * (1.0f - t * t) : 0.0f;     // d tanh
        if (g_keys) gk += g_keys[o];                                    // direct cotangent of the pre-tanh keys
        const float rot = p[0] * R[0 * 3 + j] + p[1] * R[1 * 3 + j] + p[2] * R[2 * 3 + j];
        const float sc = a.scales ? a.scales[h * a.dim + j] : 1.0f;
        gsc[j] = gk * rot;                                              // d / d scales_j
        gq[j] = gk * sc;                                                // cotangent of the rotated coordinate
      }
      float gp[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gp[c] = R[c * 3 + 0] * gq[0] + R[c * 3 + 1] * gq[1] + R[c * 3 + 2] * gq[2];
        g_res[((size_t)(b * a.H + h) * 3 + c) * a.N + n] = gp[c];      // unscaled: lattice_bwd_tail_kernel (finish)
      }
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 3; ++j) part[c * 3 + j] += p[c] * gq[j];
#pragma unroll
      for (int c = 0; c < 3; ++c) part[9 + c] += gp[c];
#pragma unroll
      for (int j = 0; j < 3; ++j) part[12 + j] += gsc[j];
      part[15] += gp[0] * r[0] + gp[1] * r[1] + gp[2] * r[2];
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float v = wave_sum_f32(part[i]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int i = threadIdx.x;
    const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    if (parts) {        // workspace: this workgroup's 16 partials, summed in a fixed order by lattice_bwd_tail_kernel (parameter sums)
      parts[(((size_t)b * gridDim.x + blockIdx.x) * a.H + h) * 16 + i] = v;
    } else if (i < 9) atomicAdd(&g_R[h * 9 + i], v);
    else if (i < 12) atomicAdd(&g_shift[h * 3 + (i - 9)], v);
    else if (i < 15) { if (g_scales && (i - 12) < a.dim) atomicAdd(&g_scales[h * a.dim + (i - 12)], v); }
    else if (g_kscale) atomicAdd(g_kscale, v);
  }
}

// zero the per-head parameter cotangents (atomic targets of lattice_bwd_kernel) in ONE launch
__global__ void __launch_bounds__(256) lattice_zero_kernel(float* g_R, float* g_shift, float* g_scales, float* g_kscale, int H, int dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < H * 9) g_R[i] = 0.0f;
  if (i < H * 3) g_shift[i] = 0.0f;
  if (g_scales && i < H * dim) g_scales[i] = 0.0f;
  if (g_kscale && i == 0) g_kscale[0] = 0.0f;
}

// parameter cotangents from the per-workgroup partials [B * nbx][H][16].  One workgroup per head: its 256 threads are
// 16 partial indices x 16 lanes over the workgroups (coalesced 64-byte rows), combined by a fixed-order tree in LDS.
// The residual scale is one scalar for all heads: workgroup 0 also sums partial 15 of every (workgroup, head).
// g_log_R of one head from its g_R (so3_exp_bwd_kernel's arithmetic, double)
__device__ __forceinline__ void so3_exp_bwd_one(const float* log_R, const float* gR, float* g_log_R, int h, float eps) {
  const double x = log_R[h * 3 + 0], y = log_R[h * 3 + 1], z = log_R[h * 3 + 2];
  const double v[3] = {x, y, z};
  double G[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) G[i] = gR[i];
  const double s = x * x + y * y + z * z;
  const bool live = s >= (double)eps;
  const double th = sqrt(live ? s : (double)eps);
  const double sn = sin(th), cs = cos(th);
  const double a = sn / th, b = (1.0 - cs) / (th * th);
  const double trG = G[0] + G[4] + G[8];
  double Gv[3], GTv[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    Gv[i] = G[i * 3 + 0] * x + G[i * 3 + 1] * y + G[i * 3 + 2] * z;
    GTv[i] = G[0 * 3 + i] * x + G[1 * 3 + i] * y + G[2 * 3 + i] * z;
  }
  const double gK[3] = {G[7] - G[5], G[2] - G[6], G[3] - G[1]};
  const double dLda = -z * G[1] + y * G[2] + z * G[3] - x * G[5] - y * G[6] + x * G[7];
  const double dLdb = x * Gv[0] + y * Gv[1] + z * Gv[2] - s * trG;
  double radial = 0.0;
  if (live) {
    const double da = (th * cs - sn) / (th * th);
    const double db = (th * sn - 2.0 * (1.0 - cs)) / (th * th * th);
    radial = (dLda * da + dLdb * db) / th;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) g_log_R[h * 3 + i] = (float)(a * gK[i] + b * (Gv[i] + GTv[i] - 2.0 * trG * v[i]) + radial * v[i]);
}

__device__ __forceinline__ void lattice_param_sum_body(const float* parts, int nwg, int H, int dim, float* g_R, float* g_shift,
                                                       float* g_scales, float* g_kscale, const int h, const float* log_R = nullptr,
                                                       float* g_log_R = nullptr, float so3_eps = 0.0f) {
  __shared__ float red[16][17];
  __shared__ float kred[256];
  __shared__ float gRs[9];
  const int i = threadIdx.x & 15, wl = threadIdx.x >> 4;
  float s = 0.0f;
  for (int w = wl; w < nwg; w += 16) s += parts[((size_t)w * H + h) * 16 + i];
  red[wl][i] = s;
  __syncthreads();
  if (wl == 0 && i < 15) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][i];
    if (i < 9) { g_R[h * 9 + i] = t; gRs[i] = t; }
    else if (i < 12) g_shift[h * 3 + (i - 9)] = t;
    else if (g_scales && (i - 12) < dim) g_scales[h * dim + (i - 12)] = t;
  }
  if (g_log_R != nullptr) {            // kernel-uniform: the head's so3 gradient right where its g_R was summed
    __syncthreads();
    if (threadIdx.x == 0) so3_exp_bwd_one(log_R, gRs, g_log_R, h, so3_eps);
  }
  if (g_kscale && h == 0) {
    float k = 0.0f;
    for (int j = threadIdx.x; j < nwg * H; j += blockDim.x) k += parts[(size_t)j * 16 + 15];
    kred[threadIdx.x] = k;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) kred[threadIdx.x] += kred[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) g_kscale[0] = kred[0];
  }
}

// g_xyz[b,c,n] = sum_h gp[b,h,c,n];  g_res[b,h,c,n] = kscale * gp[b,h,c,n]  (in place).  One thread per (b, c, n).
__device__ __forceinline__ void lattice_bwd_finish_body(float* g_res, float* g_xyz, const float* kscale, int B, int H, int N, const int bx,
                                                        const int b) {
  const size_t i = (size_t)bx * blockDim.x + threadIdx.x;                 // over (c, n)
  if (i >= (size_t)3 * N) return;
  const float ks = kscale ? kscale[0] : 1.0f;
  float* p = g_res + (size_t)b * H * 3 * N + i;
  float s = 0.0f;
  for (int h = 0; h < H; ++h) {
    const float v = p[(size_t)h * 3 * N];
    s += v;
    if (kscale) p[(size_t)h * 3 * N] = ks * v;
  }
  g_xyz[(size_t)b * 3 * N + i] = s;
}

// The two tails of the backward in ONE launch (they do not depend on each other; each is a 5-6 us launch of a few dozen
// workgroups): workgroups [0, nsum) are lattice_param_sum's (one per head; nsum = 0 without a workspace), the rest are
// lattice_bwd_finish's (nfx per cloud).
__global__ void __launch_bounds__(256) lattice_bwd_tail_kernel(const float* parts, int nwg, int nsum, int H, int dim, float* g_R, float* g_shift,
                                                               float* g_scales, float* g_kscale, float* g_res, float* g_xyz,
                                                               const float* kscale, int B, int N, int nfx, const float* log_R,
                                                               float* g_log_R, float so3_eps) {
  const int blk = blockIdx.x;
  if (blk < nsum) {
    lattice_param_sum_body(parts, nwg, H, dim, g_R, g_shift, g_scales, g_kscale, blk, log_R, g_log_R, so3_eps);
  } else {
    const int f = blk - nsum;
    lattice_bwd_finish_body(g_res, g_xyz, kscale, B, H, N, f % nfx, f / nfx);
  }
}

// so3 exponential map of the per-head rotation parameters (Rodrigues; the map the reference imports from
// pytorch3d, layers/utils.py:6,29,56):  theta = sqrt(max(|v|^2, eps)),  R = I + (sin theta / theta) K +
// ((1 - cos theta) / theta^2) K^2,  K = hat(v), K^2 = v v^T - |v|^2 I.  One thread per head, evaluated in
// double (H is 4..64: the cost is the launch) — in eager torch the map and its autograd are ~40 tiny launches per block.
__global__ void so3_exp_fwd_kernel(const float* log_R, float* R, int H, float eps) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  const double x = log_R[h * 3 + 0], y = log_R[h * 3 + 1], z = log_R[h * 3 + 2];
  const double s = x * x + y * y + z * z;
  const double th = sqrt(s > (double)eps ? s : (double)eps);
  const double a = sin(th) / th, b = (1.0 - cos(th)) / (th * th);
  const double v[3] = {x, y, z};
  const double K[9] = {0, -z, y, z, 0, -x, -y, x, 0};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      R[h * 9 + i * 3 + j] = (float)((i == j ? 1.0 : 0.0) + a * K[i * 3 + j] + b * (v[i] * v[j] - (i == j ? s : 0.0)));
}

// g_log_R from g_R:  d<G,R> = da <G,K> + a <G,dK> + db (v^T G v - s tr G) + b d(v^T G v - s tr G),
// a and b depend on v through theta only where |v|^2 >= eps (torch.clamp passes the gradient at equality).
__global__ void so3_exp_bwd_kernel(const float* log_R, const float* g_R, float* g_log_R, int H, float eps) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  const double x = log_R[h * 3 + 0], y = log_R[h * 3 + 1], z = log_R[h * 3 + 2];
  const double v[3] = {x, y, z};
  double G[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) G[i] = g_R[h * 9 + i];
  const double s = x * x + y * y + z * z;
  const bool live = s >= (double)eps;
  const double th = sqrt(live ? s : (double)eps);
  const double sn = sin(th), cs = cos(th);
  const double a = sn / th, b = (1.0 - cs) / (th * th);
  const double trG = G[0] + G[4] + G[8];
  double Gv[3], GTv[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    Gv[i] = G[i * 3 + 0] * x + G[i * 3 + 1] * y + G[i * 3 + 2] * z;
    GTv[i] = G[0 * 3 + i] * x + G[1 * 3 + i] * y + G[2 * 3 + i] * z;
  }
  const double gK[3] = {G[7] - G[5], G[2] - G[6], G[3] - G[1]};          // d<G,K>/dv
  const double dLda = -z * G[1] + y * G[2] + z * G[3] - x * G[5] - y * G[6] + x * G[7];
  const double dLdb = x * Gv[0] + y * Gv[1] + z * Gv[2] - s * trG;
  double radial = 0.0;
  if (live) {
    const double da = (th * cs - sn) / (th * th);
    const double db = (th * sn - 2.0 * (1.0 - cs)) / (th * th * th);
    radial = (dLda * da + dLdb * db) / th;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) g_log_R[h * 3 + i] = (float)(a * gK[i] + b * (Gv[i] + GTv[i] - 2.0 * trG * v[i]) + radial * v[i]);
}

bool valid(const LatticeArgs& a) {
  return a.xyz && a.res && (a.R || a.log_R) && a.shift && a.B > 0 && a.H > 0 && a.N > 0 && (a.dim == 2 || a.dim == 3) &&
         a.B <= 65535 && a.H <= 65535;
}

}  // namespace

extern "C" {

// workgroups per (cloud, head) row of lattice_fwd_kernel: about two rounds of the chip over the whole launch
static int lattice_fwd_nbx(int B, int H, int N, int& ppw) {
  const int rows = B * H, max_nbx = (N + 255) / 256;
  int nbx = (512 + rows - 1) / rows;
  if (nbx > max_nbx) nbx = max_nbx;
  if (nbx < 1) nbx = 1;
  ppw = ((N + nbx - 1) / nbx + 255) / 256 * 256;
  return (N + ppw - 1) / ppw;
}

size_t ct_lattice_fwd_workspace_bytes(int B, int H, int N) {
  if (B <= 0 || H <= 0 || N <= 0) return 0;
  return (size_t)B * H * ((N + 255) / 256) * 2 * sizeof(float);
}

// lattice + so3 exponential map + key statistics in ONE launch (see LatticeArgs): R f32[H,3,3] is written for the backward;
// `ticket`: one zeroed 32-bit word the launch leaves zeroed (with key_stats; e.g. a word of a ct_tickets_init buffer)
int ct_lattice_so3_fwd(const float* xyz, const float* residual, const float* log_R, float so3_eps, const float* shift,
                       const float* scales, const float* kscale, float* R, float* keys, float* lattice, float* key_stats,
                       void* workspace, size_t workspace_bytes, void* ticket, int B, int H, int N, int dim, ct_stream_t s) {
  LatticeArgs a = {xyz, residual, nullptr, shift, scales, kscale, B, H, N, dim, log_R, R, so3_eps, (unsigned*)ticket, key_stats,
                   (double)B * H * dim * N};
  if (!log_R || !R || !(so3_eps > 0.0f) || !valid(a) || !keys || !lattice) return CT_EINVAL;
  if (key_stats && (!ticket || !workspace || workspace_bytes < ct_lattice_fwd_workspace_bytes(B, H, N))) return CT_EWORKSPACE;
  if (!key_stats) a.ticket = nullptr;
  const int nbx = lattice_fwd_nbx(B, H, N, a.ppw);
  CT_CLEAR_ERROR();
  hipLaunchKernelGGL(lattice_fwd_kernel, dim3(nbx, H, B), dim3(256), 0, (hipStream_t)s, a, keys, lattice,
                     key_stats ? (float*)workspace : nullptr);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

int ct_lattice_fwd(const float* xyz, const float* residual, const float* R, const float* shift, const float* scales,
                   const float* kscale, float* keys, float* lattice, float* key_stats, void* workspace, size_t workspace_bytes,
                   int B, int H, int N, int dim, ct_stream_t s) {
  LatticeArgs a = {xyz, residual, R, shift, scales, kscale, B, H, N, dim, nullptr, nullptr, 0.0f, nullptr, nullptr, 0.0};
  if (!valid(a) || !keys || !lattice) return CT_EINVAL;
  if (key_stats && (!workspace || workspace_bytes < ct_lattice_fwd_workspace_bytes(B, H, N))) return CT_EWORKSPACE;
  const int nbx = lattice_fwd_nbx(B, H, N, a.ppw);
  CT_CLEAR_ERROR();
  hipLaunchKernelGGL(lattice_fwd_kernel, dim3(nbx, H, B), dim3(256), 0, (hipStream_t)s, a, keys, lattice,
                     key_stats ? (float*)workspace : nullptr);
  if (key_stats)
    hipLaunchKernelGGL(lattice_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, (const float*)workspace, (size_t)B * H * nbx,
                       (double)B * H * dim * N, key_stats);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

size_t ct_lattice_bwd_workspace_bytes(int B, int H, int N) {
  if (B <= 0 || H <= 0 || N <= 0) return 0;
  return (size_t)B * ((N + 256 * kBwdPts - 1) / (256 * kBwdPts)) * H * 16 * sizeof(float);
}

static int lattice_bwd_impl(const float* xyz, const float* residual, const float* R, const float* shift, const float* scales,
                            const float* kscale, const float* lattice, const float* g_lattice, const float* g_keys, float* g_xyz,
                            float* g_residual, float* g_R, float* g_shift, float* g_scales, float* g_kscale, void* workspace,
                            size_t workspace_bytes, int B, int H, int N, int dim, ct_stream_t s, const float* log_R,
                            float* g_log_R, float so3_eps);

int ct_lattice_bwd(const float* xyz, const float* residual, const float* R, const float* shift, const float* scales,
                   const float* kscale, const float* lattice, const float* g_lattice, const float* g_keys, float* g_xyz,
                   float* g_residual, float* g_R, float* g_shift, float* g_scales, float* g_kscale, void* workspace,
                   size_t workspace_bytes, int B, int H, int N, int dim, ct_stream_t s) {
  return lattice_bwd_impl(xyz, residual, R, shift, scales, kscale, lattice, g_lattice, g_keys, g_xyz, g_residual, g_R, g_shift,
                          g_scales, g_kscale, workspace, workspace_bytes, B, H, N, dim, s, nullptr, nullptr, 0.0f);
}

// ct_lattice_bwd + the so3 map's backward (g_log_R f32[H,3] from g_R, inside the tail launch: the workgroup that sums a
// head's g_R finishes with its g_log_R); needs the workspace (the ordered parameter sums), R from ct_lattice_so3_fwd
int ct_lattice_so3_bwd(const float* xyz, const float* residual, const float* log_R, float so3_eps, const float* R,
                       const float* shift, const float* scales, const float* kscale, const float* lattice, const float* g_lattice,
                       const float* g_keys, float* g_xyz, float* g_residual, float* g_log_R, float* g_R, float* g_shift,
                       float* g_scales, float* g_kscale, void* workspace, size_t workspace_bytes, int B, int H, int N, int dim,
                       ct_stream_t s) {
  if (!log_R || !g_log_R || !(so3_eps > 0.0f) || !workspace) return CT_EINVAL;
  return lattice_bwd_impl(xyz, residual, R, shift, scales, kscale, lattice, g_lattice, g_keys, g_xyz, g_residual, g_R, g_shift,
                          g_scales, g_kscale, workspace, workspace_bytes, B, H, N, dim, s, log_R, g_log_R, so3_eps);
}

static int lattice_bwd_impl(const float* xyz, const float* residual, const float* R, const float* shift, const float* scales,
                            const float* kscale, const float* lattice, const float* g_lattice, const float* g_keys, float* g_xyz,
                            float* g_residual, float* g_R, float* g_shift, float* g_scales, float* g_kscale, void* workspace,
                            size_t workspace_bytes, int B, int H, int N, int dim, ct_stream_t s, const float* log_R,
                            float* g_log_R, float so3_eps) {
  LatticeArgs a = {xyz, residual, R, shift, scales, kscale, B, H, N, dim, nullptr, nullptr, 0.0f, nullptr, nullptr, 0.0};
  if (!valid(a) || !lattice || (!g_lattice && !g_keys) || !g_xyz || !g_residual || !g_R || !g_shift) return CT_EINVAL;
  if ((scales != nullptr) != (g_scales != nullptr) || (kscale != nullptr) != (g_kscale != nullptr)) return CT_EINVAL;
  hipStream_t st = (hipStream_t)s;
  if (workspace && workspace_bytes < ct_lattice_bwd_workspace_bytes(B, H, N)) return CT_EWORKSPACE;
  float* parts = (float*)workspace;
  const int nbx = (N + 256 * kBwdPts - 1) / (256 * kBwdPts);
  CT_CLEAR_ERROR();
  if (!parts) hipLaunchKernelGGL(lattice_zero_kernel, dim3((H * 9 + 255) / 256), dim3(256), 0, st, g_R, g_shift, g_scales, g_kscale, H, dim);
  hipLaunchKernelGGL(lattice_bwd_kernel, dim3(nbx, H, B), dim3(256), 0, st, a, lattice, g_lattice, g_keys, g_xyz,
                     g_residual, g_R, g_shift, g_scales, g_kscale, parts);
  const int nfx = (int)(((size_t)3 * N + 255) / 256), nsum = parts ? H : 0;
  hipLaunchKernelGGL(lattice_bwd_tail_kernel, dim3((unsigned)(nsum + nfx * B)), dim3(256), 0, st, parts, B * nbx, nsum, H, dim, g_R, g_shift,
                     g_scales, g_kscale, g_residual, g_xyz, kscale, B, N, nfx, log_R, g_log_R, so3_eps);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

int ct_so3_exp_fwd(const float* log_R, float* R, int H, float eps, ct_stream_t s) {
  if (!log_R || !R || H < 0 || !(eps > 0.0f)) return CT_EINVAL;
  if (H == 0) return CT_OK;
  CT_CLEAR_ERROR();
  hipLaunchKernelGGL(so3_exp_fwd_kernel, dim3((H + 63) / 64), dim3(64), 0, (hipStream_t)s, log_R, R, H, eps);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

int ct_so3_exp_bwd(const float* log_R, const float* g_R, float* g_log_R, int H, float eps, ct_stream_t s) {
  if (!log_R || !g_R || !g_log_R || H < 0 || !(eps > 0.0f)) return CT_EINVAL;
  if (H == 0) return CT_OK;
  CT_CLEAR_ERROR();
  hipLaunchKernelGGL(so3_exp_bwd_kernel, dim3((H + 63) / 64), dim3(64), 0, (hipStream_t)s, log_R, g_R, g_log_R, H, eps);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

}  // extern "C"
