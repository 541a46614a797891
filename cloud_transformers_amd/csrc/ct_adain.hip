// Adaptive instance normalisation for the AdaIN MHCT blocks, forward and backward, one launch each:
//   xhat = (x - mean_n x) * rsqrt(var_n x + eps)           per (b, c) row of N points, biased variance
//   y    = xhat * (gamma[b,c] + 1) + beta[b,c]             (optionally followed by ReLU)
// replaces the reference's  InstanceNorm1d(affine=False) -> mul -> add (-> ReLU)  chain
// (layers/utils.py:82-97, used at layers/multihead_ct_adain.py:57-66,176-187) and its autograd.
// The reference's chain moves the row 4x forward (norm read+write, mul, add as separate kernels, more
// with ReLU) and again in backward; here a row is read once and written once (8 B/element forward,
// 12 B/element backward), so the kernel is HBM-bound by construction.
//
// Layouts: x, y, gy, gx (B, C, N) contiguous; gamma_beta (B, 2, C) = the Linear(style) output viewed
// as the reference views it (utils.py:94-96): [:,0] scale, [:,1] bias; mean, rstd (B*C).
//
// One workgroup of 256 threads owns one row.  Rows with N % 4 == 0 and N <= 16384 are held in
// registers as float4 (NV per thread) between the statistics and the normalisation, so nothing is
// re-read; other rows take the strided kernel, which re-reads the row (from L2) for each pass.
#include "ct_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / CT_WAVE;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = CT_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, CT_WAVE);
  return v;
}

// sum of up to two values over the workgroup; every thread gets the result
template <int K>
__device__ __forceinline__ void block_sum(float (&v)[K], float (*red)[kWaves]) {
  const int lane = threadIdx.x & (CT_WAVE - 1), wave = threadIdx.x / CT_WAVE;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    v[k] = wave_sum(v[k]);
    if (lane == 0) red[k][wave] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) s += red[k][w];
    v[k] = s;
  }
  __syncthreads();
}

// max over the workgroup of a non-negative value (thread 0 gets the result)
__device__ __forceinline__ float block_max(float v, float (*red)[kWaves]) {
#pragma unroll
  for (int o = CT_WAVE / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, CT_WAVE));
  if ((threadIdx.x & (CT_WAVE - 1)) == 0) red[0][threadIdx.x / CT_WAVE] = v;
  __syncthreads();
  float m = 0.f;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) m = fmaxf(m, red[0][w]);
  return m;
}

struct AdainArgs {
  const float* x;
  const float* gamma_beta;
  float* mean;
  float* rstd;
  int B, C, N;
  float eps;
  int relu;
  long long xbs, ybs;        // batch strides in floats (C*N when contiguous; larger for a channel slice of a wider tensor)
  const float* residual;     // nullable: added after the ReLU (the union block's skip connection)
  long long rbs;
  // nullable: amax_out[b * amax_bs + c] = max |y| of row (b, c) as written — the operand maxima of the pointwise GEMM that reads
  // y next (ct_pw_gemm folds them like ct_amax_f32's partials), at no extra pass over y
  float* amax_out;
  long long amax_bs;
  long long gbbs;            // batch stride of gamma_beta in floats (2*C when contiguous; larger for a slice of a stacked projection)
};

template <int NV>
__device__ __forceinline__ void adain_fwd_reg_body(const AdainArgs& a, float* __restrict__ y, const int row) {
  __shared__ float red[2][kWaves];
  const int b = row / a.C, c = row - b * a.C;
  const int nq = a.N >> 2;
  const float4* xr = reinterpret_cast<const float4*>(a.x + (size_t)b * a.xbs + (size_t)c * a.N);
  float4 v[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int q = threadIdx.x + k * kThreads;
    v[k] = q < nq ? xr[q] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float s[1] = {0.f};
#pragma unroll
  for (int k = 0; k < NV; ++k) s[0] += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  block_sum<1>(s, red);
  const float inv_n = 1.0f / (float)a.N;
  const float mu = s[0] * inv_n;
  float ss[1] = {0.f};
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int q = threadIdx.x + k * kThreads;
    if (q < nq) {
      const float dx = v[k].x - mu, dy = v[k].y - mu, dz = v[k].z - mu, dw = v[k].w - mu;
      ss[0] += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  block_sum<1>(ss, red);
  const float rs = rsqrtf(ss[0] * inv_n + a.eps);
  if (threadIdx.x == 0) {
    a.mean[row] = mu;
    a.rstd[row] = rs;
  }
  const float g = (a.gamma_beta[(size_t)b * a.gbbs + c] + 1.0f) * rs;
  const float be = a.gamma_beta[(size_t)b * a.gbbs + a.C + c];
  const float lo = a.relu ? 0.0f : -INFINITY;
  float4* yr = reinterpret_cast<float4*>(y + (size_t)b * a.ybs + (size_t)c * a.N);
  const float4* rr = a.residual ? reinterpret_cast<const float4*>(a.residual + (size_t)b * a.rbs + (size_t)c * a.N) : nullptr;
  float am = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int q = threadIdx.x + k * kThreads;
    if (q < nq) {
      float4 o;
      o.x = fmaxf((v[k].x - mu) * g + be, lo);
      o.y = fmaxf((v[k].y - mu) * g + be, lo);
      o.z = fmaxf((v[k].z - mu) * g + be, lo);
      o.w = fmaxf((v[k].w - mu) * g + be, lo);
      if (rr) {
        const float4 r = rr[q];
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      yr[q] = o;
      am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
  }
  if (a.amax_out) {
    am = block_max(am, red);
    if (threadIdx.x == 0) a.amax_out[(size_t)b * a.amax_bs + c] = am;
  }
}

__device__ __forceinline__ void adain_fwd_strided_body(const AdainArgs& a, float* __restrict__ y, const int row) {
  __shared__ float red[2][kWaves];
  const int b = row / a.C, c = row - b * a.C;
  const float* xr = a.x + (size_t)b * a.xbs + (size_t)c * a.N;
  float s[1] = {0.f};
  for (int n = threadIdx.x; n < a.N; n += kThreads) s[0] += xr[n];
  block_sum<1>(s, red);
  const float inv_n = 1.0f / (float)a.N;
  const float mu = s[0] * inv_n;
  float ss[1] = {0.f};
  for (int n = threadIdx.x; n < a.N; n += kThreads) {
    const float d = xr[n] - mu;
    ss[0] += d * d;
  }
  block_sum<1>(ss, red);
  const float rs = rsqrtf(ss[0] * inv_n + a.eps);
  if (threadIdx.x == 0) {
    a.mean[row] = mu;
    a.rstd[row] = rs;
  }
  const float g = (a.gamma_beta[(size_t)b * a.gbbs + c] + 1.0f) * rs;
  const float be = a.gamma_beta[(size_t)b * a.gbbs + a.C + c];
  const float lo = a.relu ? 0.0f : -INFINITY;
  float* yr = y + (size_t)b * a.ybs + (size_t)c * a.N;
  const float* rr = a.residual ? a.residual + (size_t)b * a.rbs + (size_t)c * a.N : nullptr;
  float am = 0.f;
  for (int n = threadIdx.x; n < a.N; n += kThreads) {
    const float o = fmaxf((xr[n] - mu) * g + be, lo) + (rr ? rr[n] : 0.0f);
    yr[n] = o;
    am = fmaxf(am, fabsf(o));
  }
  if (a.amax_out) {
    am = block_max(am, red);
    if (threadIdx.x == 0) a.amax_out[(size_t)b * a.amax_bs + c] = am;
  }
}

// Backward of y = relu?(xhat * (gamma + 1) + beta) wrt x, gamma, beta.  With g' = gy masked by the ReLU:
//   g_beta = sum g',  g_gamma = sum g' * xhat,
//   gx = rstd * (gamma + 1) * (g' - mean(g') - xhat * mean(g' * xhat))
struct AdainBwdArgs {
  const float* x;
  const float* gamma_beta;
  const float* mean;
  const float* rstd;
  const float* gy;
  float* gx;
  float* g_gamma_beta;
  int B, C, N;
  int relu;
  long long xbs, gybs, gxbs; // batch strides in floats
  float* amax_out;           // nullable: amax_out[b * amax_bs + c] = max |gx| of row (b, c) (see AdainArgs::amax_out)
  long long amax_bs;
  long long gbbs;            // batch stride of gamma_beta (read) in floats; g_gamma_beta is written contiguous
};

// the ReLU mask is recomputed with EXACTLY the forward's expression ((x - mean) * ((gamma + 1) * rstd) + beta, same
// operation order, contraction off), so an element is masked in backward iff the forward wrote a zero for it
__device__ __forceinline__ float masked(float gy, float xc, float gfw, float be, int relu) {
  return (relu && !(xc * gfw + be > 0.0f)) ? 0.0f : gy;
}

template <int NV>
__device__ __forceinline__ void adain_bwd_reg_body(const AdainBwdArgs& a, const int row) {
  __shared__ float red[2][kWaves];
  const int b = row / a.C, c = row - b * a.C;
  const int nq = a.N >> 2;
  const float mu = a.mean[row], rs = a.rstd[row];
  const float g1 = a.gamma_beta[(size_t)b * a.gbbs + c] + 1.0f;
  const float gfw = g1 * rs;                            // the forward's scale
  const float be = a.gamma_beta[(size_t)b * a.gbbs + a.C + c];
  const float4* xr = reinterpret_cast<const float4*>(a.x + (size_t)b * a.xbs + (size_t)c * a.N);
  const float4* gr = reinterpret_cast<const float4*>(a.gy + (size_t)b * a.gybs + (size_t)c * a.N);
  float4 xh[NV], g[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int q = threadIdx.x + k * kThreads;
    const bool ok = q < nq;
    const float4 xv = ok ? xr[q] : make_float4(mu, mu, mu, mu);
    const float4 gv = ok ? gr[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    xh[k] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
    g[k] = make_float4(masked(gv.x, xv.x - mu, gfw, be, a.relu), masked(gv.y, xv.y - mu, gfw, be, a.relu),
                       masked(gv.z, xv.z - mu, gfw, be, a.relu), masked(gv.w, xv.w - mu, gfw, be, a.relu));
  }
  float s[2] = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    s[0] += (g[k].x + g[k].y) + (g[k].z + g[k].w);
    s[1] += (g[k].x * xh[k].x + g[k].y * xh[k].y) + (g[k].z * xh[k].z + g[k].w * xh[k].w);
  }
  block_sum<2>(s, red);
  if (threadIdx.x == 0) {
    a.g_gamma_beta[((size_t)b * 2 + 0) * a.C + c] = s[1];
    a.g_gamma_beta[((size_t)b * 2 + 1) * a.C + c] = s[0];
  }
  const float inv_n = 1.0f / (float)a.N;
  const float m0 = s[0] * inv_n, m1 = s[1] * inv_n, sc = rs * g1;
  float4* or_ = reinterpret_cast<float4*>(a.gx + (size_t)b * a.gxbs + (size_t)c * a.N);
  float am = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int q = threadIdx.x + k * kThreads;
    if (q < nq) {
      const float4 o = make_float4(sc * (g[k].x - m0 - xh[k].x * m1), sc * (g[k].y - m0 - xh[k].y * m1),
                                   sc * (g[k].z - m0 - xh[k].z * m1), sc * (g[k].w - m0 - xh[k].w * m1));
      or_[q] = o;
      am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
  }
  if (a.amax_out) {
    am = block_max(am, red);
    if (threadIdx.x == 0) a.amax_out[(size_t)b * a.amax_bs + c] = am;
  }
}

__device__ __forceinline__ void adain_bwd_strided_body(const AdainBwdArgs& a, const int row) {
  __shared__ float red[2][kWaves];
  const int b = row / a.C, c = row - b * a.C;
  const float mu = a.mean[row], rs = a.rstd[row];
  const float g1 = a.gamma_beta[(size_t)b * a.gbbs + c] + 1.0f;
  const float gfw = g1 * rs;                            // the forward's scale
  const float be = a.gamma_beta[(size_t)b * a.gbbs + a.C + c];
  const float* xr = a.x + (size_t)b * a.xbs + (size_t)c * a.N;
  const float* gr = a.gy + (size_t)b * a.gybs + (size_t)c * a.N;
  float s[2] = {0.f, 0.f};
  for (int n = threadIdx.x; n < a.N; n += kThreads) {
    const float xh = (xr[n] - mu) * rs;
    const float g = masked(gr[n], xr[n] - mu, gfw, be, a.relu);
    s[0] += g;
    s[1] += g * xh;
  }
  block_sum<2>(s, red);
  if (threadIdx.x == 0) {
    a.g_gamma_beta[((size_t)b * 2 + 0) * a.C + c] = s[1];
    a.g_gamma_beta[((size_t)b * 2 + 1) * a.C + c] = s[0];
  }
  const float inv_n = 1.0f / (float)a.N;
  const float m0 = s[0] * inv_n, m1 = s[1] * inv_n, sc = rs * g1;
  float* or_ = a.gx + (size_t)b * a.gxbs + (size_t)c * a.N;
  float am = 0.f;
  for (int n = threadIdx.x; n < a.N; n += kThreads) {
    const float xh = (xr[n] - mu) * rs;
    const float g = masked(gr[n], xr[n] - mu, gfw, be, a.relu);
    const float o = sc * (g - m0 - xh * m1);
    or_[n] = o;
    am = fmaxf(am, fabsf(o));
  }
  if (a.amax_out) {
    am = block_max(am, red);
    if (threadIdx.x == 0) a.amax_out[(size_t)b * a.amax_bs + c] = am;
  }
}

// Several norms in ONE launch (the keys / values norms of the heads on channel ranges of the stacked projection, the heads' `after`
// norms on ranges of the concatenation): a workgroup per (cloud, channel) row of every norm, its norm found by the row prefix
// sums — as ct_bnorm.hip's BnTable.
constexpr int kAdainMaxItems = 8;
struct AdainTable {
  int n;
  int rstart[kAdainMaxItems + 1];
  AdainArgs item[kAdainMaxItems];
  float* y[kAdainMaxItems];
};
struct AdainBwdTable {
  int n;
  int rstart[kAdainMaxItems + 1];
  AdainBwdArgs item[kAdainMaxItems];
};

template <typename T>
__device__ __forceinline__ int adain_item_of(const T& t, int& row) {
  int i = 0;
  while (i + 1 < t.n && row >= t.rstart[i + 1]) ++i;
  row -= t.rstart[i];
  return i;
}

template <int NV>
__global__ void __launch_bounds__(kThreads) adain_fwd_reg_kernel(AdainTable t) {
  int row = blockIdx.x;
  const int i = adain_item_of(t, row);
  adain_fwd_reg_body<NV>(t.item[i], t.y[i], row);
}
__global__ void __launch_bounds__(kThreads) adain_fwd_strided_kernel(AdainTable t) {
  int row = blockIdx.x;
  const int i = adain_item_of(t, row);
  adain_fwd_strided_body(t.item[i], t.y[i], row);
}
template <int NV>
__global__ void __launch_bounds__(kThreads) adain_bwd_reg_kernel(AdainBwdTable t) {
  int row = blockIdx.x;
  const int i = adain_item_of(t, row);
  adain_bwd_reg_body<NV>(t.item[i], row);
}
__global__ void __launch_bounds__(kThreads) adain_bwd_strided_kernel(AdainBwdTable t) {
  int row = blockIdx.x;
  const int i = adain_item_of(t, row);
  adain_bwd_strided_body(t.item[i], row);
}

bool vec_ok(int N, const void* p0, const void* p1, const void* p2) {
  const uintptr_t bits = (uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2;
  return (N & 3) == 0 && N <= 4 * kThreads * 16 && (bits & 15) == 0;
}

int nv_for(int N) {
  const int per = ((N >> 2) + kThreads - 1) / kThreads;
  int nv = 1;
  while (nv < per) nv <<= 1;
  return nv;
}

}  // namespace

#define CT_ADAIN_DISPATCH(NVV, KERNEL, ...)                                                       \
  switch (NVV) {                                                                                  \
    case 1: hipLaunchKernelGGL((KERNEL<1>), dim3(rows), dim3(kThreads), 0, stream, __VA_ARGS__); break;   \
    case 2: hipLaunchKernelGGL((KERNEL<2>), dim3(rows), dim3(kThreads), 0, stream, __VA_ARGS__); break;   \
    case 4: hipLaunchKernelGGL((KERNEL<4>), dim3(rows), dim3(kThreads), 0, stream, __VA_ARGS__); break;   \
    case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(rows), dim3(kThreads), 0, stream, __VA_ARGS__); break;   \
    default: hipLaunchKernelGGL((KERNEL<16>), dim3(rows), dim3(kThreads), 0, stream, __VA_ARGS__); break; \
  }

// a batch stride: 0 = contiguous (C*N), else >= C*N floats (a multiple of 4 for the float4 kernels: checked by vec_ok's callers)
static bool adain_stride(long long bs, int C, int N, long long& out) {
  const long long dense = (long long)C * N;
  if (bs == 0) { out = dense; return true; }
  if (bs < dense) return false;
  out = bs;
  return true;
}

static int adain_fwd_prepare(AdainArgs& a, float* y, long long x_batch_stride, long long y_batch_stride,
                             long long residual_batch_stride, bool& vec) {
  if (!a.x || !a.gamma_beta || !y || !a.mean || !a.rstd) return CT_EINVAL;
  if (!adain_stride(x_batch_stride, a.C, a.N, a.xbs) || !adain_stride(y_batch_stride, a.C, a.N, a.ybs) ||
      !adain_stride(residual_batch_stride, a.C, a.N, a.rbs))
    return CT_EINVAL;
  vec = vec_ok(a.N, a.x, y, a.residual) && ((a.xbs | a.ybs | a.rbs) & 3) == 0;
  return CT_OK;
}

static int adain_fwd_launch_table(AdainTable& t, bool vec, hipStream_t stream) {
  const int rows = t.rstart[t.n];
  CT_CLEAR_ERROR();
  if (vec) {
    CT_ADAIN_DISPATCH(nv_for(t.item[0].N), adain_fwd_reg_kernel, t)
  } else {
    hipLaunchKernelGGL(adain_fwd_strided_kernel, dim3(rows), dim3(kThreads), 0, stream, t);
  }
  CT_CHECK_LAUNCH();
  return CT_OK;
}

static int adain_fwd_impl(const float* x, long long x_batch_stride, const float* gamma_beta, const float* residual,
                          long long residual_batch_stride, float* y, long long y_batch_stride, float* mean, float* rstd,
                          float* amax_out, long long amax_batch_stride, int B, int C, int N, float eps, int relu, ct_stream_t s) {
  if (B < 0 || C < 0 || N < 0 || !(eps >= 0.0f)) return CT_EINVAL;
  if ((size_t)B * C == 0 || N == 0) return CT_OK;
  if ((size_t)B * C > 0x7fffffffull) return CT_EINVAL;
  AdainArgs a{x, gamma_beta, mean, rstd, B, C, N, eps, relu, 0, 0, residual, 0, amax_out, amax_batch_stride ? amax_batch_stride : C,
              2LL * C};
  bool vec;
  const int rc = adain_fwd_prepare(a, y, x_batch_stride, y_batch_stride, residual_batch_stride, vec);
  if (rc != CT_OK) return rc;
  AdainTable t{};
  t.n = 1; t.rstart[0] = 0; t.rstart[1] = B * C; t.item[0] = a; t.y[0] = y;
  return adain_fwd_launch_table(t, vec, (hipStream_t)s);
}

// Up to kAdainMaxItems norms over the same (B, N) in one launch: ct_adain_fwd_amax of every item (non-empty shapes only).
extern "C" int ct_adain_group_fwd(const ct_adain_fwd_item* items, int n, int B, int N, ct_stream_t s) {
  if (!items || n < 1 || n > kAdainMaxItems || B < 1 || N < 1) return CT_EINVAL;
  AdainTable t{};
  t.n = n;
  bool vec_all = true;
  long long r0 = 0;
  for (int i = 0; i < n; ++i) {
    const ct_adain_fwd_item& it = items[i];
    if (it.C < 1 || !(it.eps >= 0.0f) || (it.amax_batch_stride != 0 && it.amax_batch_stride < it.C)) return CT_EINVAL;
    if (it.gamma_beta_batch_stride != 0 && it.gamma_beta_batch_stride < 2LL * it.C) return CT_EINVAL;
    AdainArgs a{it.x, it.gamma_beta, it.mean, it.rstd, B, it.C, N, it.eps, it.relu, 0, 0, it.residual, 0, it.amax_out,
                it.amax_batch_stride ? it.amax_batch_stride : it.C,
                it.gamma_beta_batch_stride ? it.gamma_beta_batch_stride : 2LL * it.C};
    bool vec;
    const int rc = adain_fwd_prepare(a, it.y, it.x_batch_stride, it.y_batch_stride, it.residual_batch_stride, vec);
    if (rc != CT_OK) return rc;
    vec_all = vec_all && vec;
    t.rstart[i] = (int)r0;
    t.item[i] = a;
    t.y[i] = it.y;
    r0 += (long long)B * it.C;
    if (r0 > 0x7fffffffLL) return CT_EINVAL;
  }
  t.rstart[n] = (int)r0;
  return adain_fwd_launch_table(t, vec_all, (hipStream_t)s);
}

extern "C" int ct_adain_fwd(const float* x, long long x_batch_stride, const float* gamma_beta, const float* residual,
                            long long residual_batch_stride, float* y, long long y_batch_stride, float* mean, float* rstd,
                            int B, int C, int N, float eps, int relu, ct_stream_t s) {
  return adain_fwd_impl(x, x_batch_stride, gamma_beta, residual, residual_batch_stride, y, y_batch_stride, mean, rstd, nullptr, 0,
                        B, C, N, eps, relu, s);
}

// with amax_out f32 (nullable): amax_out[b * amax_batch_stride + c] = max |y| of row (b, c); stride 0 = C
extern "C" int ct_adain_fwd_amax(const float* x, long long x_batch_stride, const float* gamma_beta, const float* residual,
                                 long long residual_batch_stride, float* y, long long y_batch_stride, float* mean, float* rstd,
                                 float* amax_out, long long amax_batch_stride, int B, int C, int N, float eps, int relu,
                                 ct_stream_t s) {
  if (amax_batch_stride != 0 && amax_batch_stride < C) return CT_EINVAL;
  return adain_fwd_impl(x, x_batch_stride, gamma_beta, residual, residual_batch_stride, y, y_batch_stride, mean, rstd, amax_out,
                        amax_batch_stride, B, C, N, eps, relu, s);
}

static int adain_bwd_prepare(AdainBwdArgs& a, long long x_batch_stride, long long gy_batch_stride, long long gx_batch_stride,
                             bool& vec) {
  if (!a.x || !a.gy || !a.gx || !a.gamma_beta || !a.mean || !a.rstd || !a.g_gamma_beta) return CT_EINVAL;
  if (!adain_stride(x_batch_stride, a.C, a.N, a.xbs) || !adain_stride(gy_batch_stride, a.C, a.N, a.gybs) ||
      !adain_stride(gx_batch_stride, a.C, a.N, a.gxbs))
    return CT_EINVAL;
  vec = vec_ok(a.N, a.x, a.gy, a.gx) && ((a.xbs | a.gybs | a.gxbs) & 3) == 0;
  return CT_OK;
}

static int adain_bwd_launch_table(AdainBwdTable& t, bool vec, hipStream_t stream) {
  const int rows = t.rstart[t.n];
  CT_CLEAR_ERROR();
  if (vec) {
    CT_ADAIN_DISPATCH(nv_for(t.item[0].N), adain_bwd_reg_kernel, t)
  } else {
    hipLaunchKernelGGL(adain_bwd_strided_kernel, dim3(rows), dim3(kThreads), 0, stream, t);
  }
  CT_CHECK_LAUNCH();
  return CT_OK;
}

static int adain_bwd_impl(const float* x, long long x_batch_stride, const float* gamma_beta, const float* mean,
                          const float* rstd, const float* gy, long long gy_batch_stride, float* gx,
                          long long gx_batch_stride, float* g_gamma_beta, float* amax_out, long long amax_batch_stride, int B, int C,
                          int N, int relu, ct_stream_t s) {
  hipStream_t stream = (hipStream_t)s;
  if (B < 0 || C < 0 || N < 0) return CT_EINVAL;
  if ((size_t)B * C == 0) return CT_OK;
  if (!gamma_beta || !mean || !rstd || !g_gamma_beta) return CT_EINVAL;
  if ((size_t)B * C > 0x7fffffffull) return CT_EINVAL;
  if (N == 0) return hipMemsetAsync(g_gamma_beta, 0, (size_t)B * 2 * C * sizeof(float), stream) == hipSuccess ? CT_OK : CT_ELAUNCH;
  AdainBwdArgs a{x, gamma_beta, mean, rstd, gy, gx, g_gamma_beta, B, C, N, relu, 0, 0, 0, amax_out,
                 amax_batch_stride ? amax_batch_stride : C, 2LL * C};
  bool vec;
  const int rc = adain_bwd_prepare(a, x_batch_stride, gy_batch_stride, gx_batch_stride, vec);
  if (rc != CT_OK) return rc;
  AdainBwdTable t{};
  t.n = 1; t.rstart[0] = 0; t.rstart[1] = B * C; t.item[0] = a;
  return adain_bwd_launch_table(t, vec, stream);
}

extern "C" int ct_adain_group_bwd(const ct_adain_bwd_item* items, int n, int B, int N, ct_stream_t s) {
  if (!items || n < 1 || n > kAdainMaxItems || B < 1 || N < 1) return CT_EINVAL;
  AdainBwdTable t{};
  t.n = n;
  bool vec_all = true;
  long long r0 = 0;
  for (int i = 0; i < n; ++i) {
    const ct_adain_bwd_item& it = items[i];
    if (it.C < 1 || (it.amax_batch_stride != 0 && it.amax_batch_stride < it.C)) return CT_EINVAL;
    if (it.gamma_beta_batch_stride != 0 && it.gamma_beta_batch_stride < 2LL * it.C) return CT_EINVAL;
    AdainBwdArgs a{it.x, it.gamma_beta, it.mean, it.rstd, it.gy, it.gx, it.g_gamma_beta, B, it.C, N, it.relu, 0, 0, 0, it.amax_out,
                   it.amax_batch_stride ? it.amax_batch_stride : it.C,
                   it.gamma_beta_batch_stride ? it.gamma_beta_batch_stride : 2LL * it.C};
    bool vec;
    const int rc = adain_bwd_prepare(a, it.x_batch_stride, it.gy_batch_stride, it.gx_batch_stride, vec);
    if (rc != CT_OK) return rc;
    vec_all = vec_all && vec;
    t.rstart[i] = (int)r0;
    t.item[i] = a;
    r0 += (long long)B * it.C;
    if (r0 > 0x7fffffffLL) return CT_EINVAL;
  }
  t.rstart[n] = (int)r0;
  return adain_bwd_launch_table(t, vec_all, (hipStream_t)s);
}

extern "C" int ct_adain_bwd(const float* x, long long x_batch_stride, const float* gamma_beta, const float* mean,
                            const float* rstd, const float* gy, long long gy_batch_stride, float* gx,
                            long long gx_batch_stride, float* g_gamma_beta, int B, int C, int N, int relu, ct_stream_t s) {
  return adain_bwd_impl(x, x_batch_stride, gamma_beta, mean, rstd, gy, gy_batch_stride, gx, gx_batch_stride, g_gamma_beta, nullptr, 0,
                        B, C, N, relu, s);
}

// with amax_out f32 (nullable): amax_out[b * amax_batch_stride + c] = max |gx| of row (b, c); stride 0 = C
extern "C" int ct_adain_bwd_amax(const float* x, long long x_batch_stride, const float* gamma_beta, const float* mean,
                                 const float* rstd, const float* gy, long long gy_batch_stride, float* gx,
                                 long long gx_batch_stride, float* g_gamma_beta, float* amax_out, long long amax_batch_stride,
                                 int B, int C, int N, int relu, ct_stream_t s) {
  if (amax_batch_stride != 0 && amax_batch_stride < C) return CT_EINVAL;
  return adain_bwd_impl(x, x_batch_stride, gamma_beta, mean, rstd, gy, gy_batch_stride, gx, gx_batch_stride, g_gamma_beta, amax_out,
                        amax_batch_stride, B, C, N, relu, s);
}
