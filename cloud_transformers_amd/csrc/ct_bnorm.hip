// Training-mode BatchNorm1d fused with the ReLU that follows it in the MHCT blocks' `after` stacks
// (layers/multihead_ct.py:67-68,149-153: nn.Sequential(BatchNorm1d, ReLU(inplace=True))), forward and backward, one
// launch each:
//   mean_c = mean over (b, n) of x[b,c,n],  var_c = biased variance,  xhat = (x - mean_c) * rsqrt(var_c + eps)
//   y = relu?(xhat * weight_c + bias_c);   running_mean / running_var updated like torch (momentum, unbiased variance)
// The library pair moves the activation 2x (norm) + 2x (ReLU) forward and again in backward; here a channel's B*N values
// are read once, held in registers between the statistics and the normalisation, and written once.
//
// Layout: x, y, gy, gx (B, C, N) contiguous.  One workgroup of 1024 threads owns one channel: its B rows of N floats
// (N % 4 == 0) as float4, NV <= 8 per thread (the backward holds x and gy), so B*N <= 32768 — the training shapes of
// the segmenter / classifier (B8 N4096, B8 N2048).  Longer channels and rows that are not float4-addressable take the
// loop kernels below, which re-read the channel (from L2, mostly) for each pass.
#include "ct_common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / CT_WAVE;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = CT_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, CT_WAVE);
  return v;
}

// sums of K values over the workgroup; every thread gets the results (fixed order: deterministic)
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

struct BnArgs {
  const float* x;
  const float* weight;
  const float* bias;
  float* running_mean;   // nullable
  float* running_var;
  float* save_mean;
  float* save_rstd;
  int B, C, N;
  float eps, momentum;
  int relu;
  long long xbs, ybs;    // batch strides of x and y in floats (C*N when contiguous; larger for a channel slice)
  const float* residual; // nullable: added after the ReLU (the union block's skip connection)
  long long rbs;
  long long* num_batches_tracked;   // nullable: incremented once per launch, as nn.BatchNorm1d.forward does
  // Synchronised statistics (SyncBatchNorm under data parallelism) split the pass around the exchange between ranks:
  //   mode 1: local statistics only — save_mean[c] = mean, save_rstd[c] = sum (x - mean)^2 — nothing else is written;
  //   mode 2: normalise with the statistics of the whole job: every rank's (mean, sum of squares, count) as gathered
  //           by ONE all_gather — g_mean[r * g_stride + c], g_m2[r * g_stride + c], g_count[r * g_stride], r < world —
  //           merged here, per channel, by the parallel-variance rule (no host arithmetic between the collective and
  //           this launch); save_mean / save_rstd receive the merged mean and rsqrt(var + eps), count_out[0] the
  //           job's number of values per channel;
  //   mode 0: both in one pass over this rank's batch (the channel stays in registers in between).
  int mode;
  float* count_out;      // mode 1: this rank's B*N (written by channel 0); mode 2: the job's total (nullable)
  const float* g_mean;
  const float* g_m2;
  const float* g_count;
  int world;
  long long g_stride;
  // nullable: amax_out[c] = max |y| of channel c as written (after ReLU and skip) — the per-tensor scale of the pointwise GEMM
  // that reads y next (ct_pw_gemm folds per-channel maxima like ct_amax_f32's partials), at no extra pass over y
  float* amax_out;
};

// mode 2: merge the ranks' statistics of channel c (Chan et al.: M2 = sum M2_r + sum n_r (mean_r - mean)^2)
__device__ __forceinline__ void merge_rank_stats(const BnArgs& a, int c, float& mu, float& var, float& M) {
  float total = 0.f, acc = 0.f;
  for (int r = 0; r < a.world; ++r) {
    const float n = a.g_count[r * a.g_stride];
    total += n;
    acc += a.g_mean[r * a.g_stride + c] * n;
  }
  mu = acc / total;
  float m2 = 0.f;
  for (int r = 0; r < a.world; ++r) {
    const float d = a.g_mean[r * a.g_stride + c] - mu;
    m2 += a.g_m2[r * a.g_stride + c] + a.g_count[r * a.g_stride] * d * d;
  }
  var = m2 / total;
  M = total;
}

// float offset of quad q (over the B rows of channel c, N/4 quads each) in a tensor whose batch stride is bs floats
__device__ __forceinline__ size_t quad_offset(int q, int nq, int c, long long bs, int N) {
  const int b = q / nq, n4 = q - b * nq;
  return (size_t)b * (size_t)bs + (size_t)c * N + ((size_t)n4 << 2);
}

template <int NV>
__device__ __forceinline__ void bn_fwd_reg_body(const BnArgs& a, float* __restrict__ y, const int c) {
  __shared__ float red[1][kWaves];
  const int nq = a.N >> 2, total = a.B * nq;
  float4 v[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int q = threadIdx.x + k * kThreads;
    v[k] = q < total ? *reinterpret_cast<const float4*>(a.x + quad_offset(q, nq, c, a.xbs, a.N)) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float M = (float)a.B * (float)a.N;
  float mu, var;
  if (a.mode != 2) {
    float s[1] = {0.f};
#pragma unroll
    for (int k = 0; k < NV; ++k) s[0] += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    block_sum<1>(s, red);
    mu = s[0] / M;
    float ss[1] = {0.f};
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int q = threadIdx.x + k * kThreads;
      if (q < total) {
        const float dx = v[k].x - mu, dy = v[k].y - mu, dz = v[k].z - mu, dw = v[k].w - mu;
        ss[0] += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
    }
    block_sum<1>(ss, red);
    if (a.mode == 1) {          // local statistics for the exchange
      if (threadIdx.x == 0) {
        a.save_mean[c] = mu;
        a.save_rstd[c] = ss[0];
        if (c == 0 && a.count_out) a.count_out[0] = M;
      }
      return;
    }
    var = ss[0] / M;
  } else {
    merge_rank_stats(a, c, mu, var, M);
    if (threadIdx.x == 0 && c == 0 && a.count_out) a.count_out[0] = M;
  }
  const float rs = rsqrtf(var + a.eps);
  if (threadIdx.x == 0) {
    a.save_mean[c] = mu;
    a.save_rstd[c] = rs;
    if (a.running_mean) {
      a.running_mean[c] = (1.0f - a.momentum) * a.running_mean[c] + a.momentum * mu;
      a.running_var[c] = (1.0f - a.momentum) * a.running_var[c] + a.momentum * (var * (M / (M - 1.0f)));
    }
    if (c == 0 && a.num_batches_tracked) a.num_batches_tracked[0] += 1;
  }
  const float g = a.weight[c] * rs;
  const float be = a.bias[c];
  const float lo = a.relu ? 0.0f : -INFINITY;
  float am = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int q = threadIdx.x + k * kThreads;
    if (q < total) {
      float4 o;
      o.x = fmaxf((v[k].x - mu) * g + be, lo);
      o.y = fmaxf((v[k].y - mu) * g + be, lo);
      o.z = fmaxf((v[k].z - mu) * g + be, lo);
      o.w = fmaxf((v[k].w - mu) * g + be, lo);
      if (a.residual) {
        const float4 r = *reinterpret_cast<const float4*>(a.residual + quad_offset(q, nq, c, a.rbs, a.N));
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      *reinterpret_cast<float4*>(y + quad_offset(q, nq, c, a.ybs, a.N)) = o;
      am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
  }
  if (a.amax_out) {
    am = block_max(am, red);
    if (threadIdx.x == 0) a.amax_out[c] = am;
  }
}

struct BnBwdArgs {
  const float* x;
  const float* weight;
  const float* bias;
  const float* save_mean;
  const float* save_rstd;
  const float* gy;
  float* gx;
  float* g_weight;
  float* g_bias;
  int B, C, N;
  int relu;
  long long xbs, gybs, gxbs;   // batch strides in floats
  // mode 1: the two per-channel sums only (g_bias = sum g', g_weight = sum g' * xhat over THIS rank's batch);
  // mode 2: gx from the job-wide sums sum0[c] / sum1[c] (this rank's sums all-reduced in ONE collective) over count[0]
  //         values per channel (g_weight / g_bias not written);
  // mode 0: both in one pass.
  int mode;
  const float* sum0;
  const float* sum1;
  const float* count;
  float* amax_out;   // nullable: amax_out[c] = max |gx| of channel c (see BnArgs::amax_out)
  // mode 1, nullable: a second copy of the two sums (the collective overwrites the first in place; this one stays this rank's
  // g_bias / g_weight — without it the caller clones the buffer: one more launch in every norm group's backward chain)
  float* g_bias2;
  float* g_weight2;
};

// the ReLU mask is recomputed with EXACTLY the forward's expression ((x - mean) * (weight * rstd) + bias, same operation
// order, contraction off), so an element is masked in backward iff the forward wrote a zero for it
__device__ __forceinline__ float masked(float gy, float xc, float gfw, float be, int relu) {
  return (relu && !(xc * gfw + be > 0.0f)) ? 0.0f : gy;
}

// With g' = gy masked by the ReLU:  g_bias = sum g',  g_weight = sum g' * xhat,
//   gx = weight * rstd * (g' - mean(g') - xhat * mean(g' * xhat))
template <int NV>
__device__ __forceinline__ void bn_bwd_reg_body(const BnBwdArgs& a, const int c) {
  __shared__ float red[2][kWaves];
  const int nq = a.N >> 2, total = a.B * nq;
  const float mu = a.save_mean[c], rs = a.save_rstd[c];
  const float w = a.weight[c];
  const float gfw = w * rs;                            // the forward's scale
  const float be = a.bias[c];
  float4 xh[NV], g[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int q = threadIdx.x + k * kThreads;
    const bool ok = q < total;
    const float4 xv = ok ? *reinterpret_cast<const float4*>(a.x + quad_offset(q, nq, c, a.xbs, a.N)) : make_float4(mu, mu, mu, mu);
    const float4 gv = ok ? *reinterpret_cast<const float4*>(a.gy + quad_offset(q, nq, c, a.gybs, a.N)) : make_float4(0.f, 0.f, 0.f, 0.f);
    xh[k] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
    g[k] = make_float4(masked(gv.x, xv.x - mu, gfw, be, a.relu), masked(gv.y, xv.y - mu, gfw, be, a.relu),
                       masked(gv.z, xv.z - mu, gfw, be, a.relu), masked(gv.w, xv.w - mu, gfw, be, a.relu));
  }
  float s[2] = {0.f, 0.f};
  float M = (float)a.B * (float)a.N;
  if (a.mode != 2) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      s[0] += (g[k].x + g[k].y) + (g[k].z + g[k].w);
      s[1] += (g[k].x * xh[k].x + g[k].y * xh[k].y) + (g[k].z * xh[k].z + g[k].w * xh[k].w);
    }
    block_sum<2>(s, red);
    if (threadIdx.x == 0) {
      a.g_bias[c] = s[0];
      a.g_weight[c] = s[1];
      if (a.g_bias2 != nullptr) {
        a.g_bias2[c] = s[0];
        a.g_weight2[c] = s[1];
      }
    }
    if (a.mode == 1) return;      // this rank's sums for the exchange
  } else {
    s[0] = a.sum0[c];
    s[1] = a.sum1[c];
    M = a.count[0];
  }
  const float m0 = s[0] / M, m1 = s[1] / M;
  float am = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int q = threadIdx.x + k * kThreads;
    if (q < total) {
      const float4 o = make_float4(gfw * (g[k].x - m0 - xh[k].x * m1), gfw * (g[k].y - m0 - xh[k].y * m1),
                                   gfw * (g[k].z - m0 - xh[k].z * m1), gfw * (g[k].w - m0 - xh[k].w * m1));
      *reinterpret_cast<float4*>(a.gx + quad_offset(q, nq, c, a.gxbs, a.N)) = o;
      am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
  }
  if (a.amax_out) {
    am = block_max(am, red);
    if (threadIdx.x == 0) a.amax_out[c] = am;
  }
}

// Channels too long for the registers (B*N > 32768) or rows that are not float4-addressable: the same arithmetic with the
// channel re-read from memory for each pass (a 256 KiB..1 MiB channel mostly stays in L2 between the passes).
// VEC: N % 4 == 0, strides % 4 == 0, 16-byte aligned bases -> float4 accesses.
template <bool VEC>
__device__ __forceinline__ int chan_items(int B, int N) { return VEC ? B * (N >> 2) : B * N; }

template <bool VEC>
__device__ __forceinline__ size_t item_offset(int i, int N, int c, long long bs) {
  if (VEC) return quad_offset(i, N >> 2, c, bs, N);
  const int b = i / N, n = i - b * N;
  return (size_t)b * (size_t)bs + (size_t)c * N + n;
}

template <bool VEC>
__device__ __forceinline__ void bn_fwd_loop_body(const BnArgs& a, float* __restrict__ y, const int c) {
  __shared__ float red[1][kWaves];
  const int total = chan_items<VEC>(a.B, a.N);
  float M = (float)a.B * (float)a.N;
  float mu, var;
  if (a.mode != 2) {
    float s[1] = {0.f};
    for (int i = threadIdx.x; i < total; i += kThreads) {
      const float* p = a.x + item_offset<VEC>(i, a.N, c, a.xbs);
      if (VEC) { const float4 v = *reinterpret_cast<const float4*>(p); s[0] += (v.x + v.y) + (v.z + v.w); }
      else s[0] += p[0];
    }
    block_sum<1>(s, red);
    mu = s[0] / M;
    float ss[1] = {0.f};
    for (int i = threadIdx.x; i < total; i += kThreads) {
      const float* p = a.x + item_offset<VEC>(i, a.N, c, a.xbs);
      if (VEC) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
        ss[0] += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      } else {
        const float d = p[0] - mu;
        ss[0] += d * d;
      }
    }
    block_sum<1>(ss, red);
    if (a.mode == 1) {          // local statistics for the exchange
      if (threadIdx.x == 0) {
        a.save_mean[c] = mu;
        a.save_rstd[c] = ss[0];
        if (c == 0 && a.count_out) a.count_out[0] = M;
      }
      return;
    }
    var = ss[0] / M;
  } else {
    merge_rank_stats(a, c, mu, var, M);
    if (threadIdx.x == 0 && c == 0 && a.count_out) a.count_out[0] = M;
  }
  const float rs = rsqrtf(var + a.eps);
  if (threadIdx.x == 0) {
    a.save_mean[c] = mu;
    a.save_rstd[c] = rs;
    if (a.running_mean) {
      a.running_mean[c] = (1.0f - a.momentum) * a.running_mean[c] + a.momentum * mu;
      a.running_var[c] = (1.0f - a.momentum) * a.running_var[c] + a.momentum * (var * (M / (M - 1.0f)));
    }
    if (c == 0 && a.num_batches_tracked) a.num_batches_tracked[0] += 1;
  }
  const float g = a.weight[c] * rs;
  const float be = a.bias[c];
  const float lo = a.relu ? 0.0f : -INFINITY;
  float am = 0.f;
  for (int i = threadIdx.x; i < total; i += kThreads) {
    const float* p = a.x + item_offset<VEC>(i, a.N, c, a.xbs);
    float* o = y + item_offset<VEC>(i, a.N, c, a.ybs);
    const float* r = a.residual ? a.residual + item_offset<VEC>(i, a.N, c, a.rbs) : nullptr;
    if (VEC) {
      const float4 v = *reinterpret_cast<const float4*>(p);
      float4 q;
      q.x = fmaxf((v.x - mu) * g + be, lo);
      q.y = fmaxf((v.y - mu) * g + be, lo);
      q.z = fmaxf((v.z - mu) * g + be, lo);
      q.w = fmaxf((v.w - mu) * g + be, lo);
      if (r) { const float4 rv = *reinterpret_cast<const float4*>(r); q.x += rv.x; q.y += rv.y; q.z += rv.z; q.w += rv.w; }
      *reinterpret_cast<float4*>(o) = q;
      am = fmaxf(fmaxf(am, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
    } else {
      o[0] = fmaxf((p[0] - mu) * g + be, lo) + (r ? r[0] : 0.0f);
      am = fmaxf(am, fabsf(o[0]));
    }
  }
  if (a.amax_out) {
    am = block_max(am, red);
    if (threadIdx.x == 0) a.amax_out[c] = am;
  }
}

template <bool VEC>
__device__ __forceinline__ void bn_bwd_loop_body(const BnBwdArgs& a, const int c) {
  __shared__ float red[2][kWaves];
  const int total = chan_items<VEC>(a.B, a.N);
  const float mu = a.save_mean[c], rs = a.save_rstd[c];
  const float gfw = a.weight[c] * rs;
  const float be = a.bias[c];
  float s[2] = {0.f, 0.f};
  float M = (float)a.B * (float)a.N;
  if (a.mode != 2) {
    for (int i = threadIdx.x; i < total; i += kThreads) {
      const float* px = a.x + item_offset<VEC>(i, a.N, c, a.xbs);
      const float* pg = a.gy + item_offset<VEC>(i, a.N, c, a.gybs);
      if (VEC) {
        const float4 xv = *reinterpret_cast<const float4*>(px), gv = *reinterpret_cast<const float4*>(pg);
        const float g0 = masked(gv.x, xv.x - mu, gfw, be, a.relu), g1 = masked(gv.y, xv.y - mu, gfw, be, a.relu);
        const float g2 = masked(gv.z, xv.z - mu, gfw, be, a.relu), g3 = masked(gv.w, xv.w - mu, gfw, be, a.relu);
        s[0] += (g0 + g1) + (g2 + g3);
        s[1] += (g0 * ((xv.x - mu) * rs) + g1 * ((xv.y - mu) * rs)) + (g2 * ((xv.z - mu) * rs) + g3 * ((xv.w - mu) * rs));
      } else {
        const float g0 = masked(pg[0], px[0] - mu, gfw, be, a.relu);
        s[0] += g0;
        s[1] += g0 * ((px[0] - mu) * rs);
      }
    }
    block_sum<2>(s, red);
    if (threadIdx.x == 0) {
      a.g_bias[c] = s[0];
      a.g_weight[c] = s[1];
      if (a.g_bias2 != nullptr) {
        a.g_bias2[c] = s[0];
        a.g_weight2[c] = s[1];
      }
    }
    if (a.mode == 1) return;      // this rank's sums for the exchange
  } else {
    s[0] = a.sum0[c];
    s[1] = a.sum1[c];
    M = a.count[0];
  }
  const float m0 = s[0] / M, m1 = s[1] / M;
  float am = 0.f;
  for (int i = threadIdx.x; i < total; i += kThreads) {
    const float* px = a.x + item_offset<VEC>(i, a.N, c, a.xbs);
    const float* pg = a.gy + item_offset<VEC>(i, a.N, c, a.gybs);
    float* po = a.gx + item_offset<VEC>(i, a.N, c, a.gxbs);
    if (VEC) {
      const float4 xv = *reinterpret_cast<const float4*>(px), gv = *reinterpret_cast<const float4*>(pg);
      float4 o;
      o.x = gfw * (masked(gv.x, xv.x - mu, gfw, be, a.relu) - m0 - ((xv.x - mu) * rs) * m1);
      o.y = gfw * (masked(gv.y, xv.y - mu, gfw, be, a.relu) - m0 - ((xv.y - mu) * rs) * m1);
      o.z = gfw * (masked(gv.z, xv.z - mu, gfw, be, a.relu) - m0 - ((xv.z - mu) * rs) * m1);
      o.w = gfw * (masked(gv.w, xv.w - mu, gfw, be, a.relu) - m0 - ((xv.w - mu) * rs) * m1);
      *reinterpret_cast<float4*>(po) = o;
      am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    } else {
      po[0] = gfw * (masked(pg[0], px[0] - mu, gfw, be, a.relu) - m0 - ((px[0] - mu) * rs) * m1);
      am = fmaxf(am, fabsf(po[0]));
    }
  }
  if (a.amax_out) {
    am = block_max(am, red);
    if (threadIdx.x == 0) a.amax_out[c] = am;
  }
}

// Several norms in ONE launch (the key / values norms of the heads on channel ranges of the stacked projection, the heads' `after`
// norms on ranges of the concatenation): a workgroup per channel of every norm, its norm found by the channel prefix sums.
// The norms of a block's group are 32-512 channels each: one by one they are 13-28 us launches of 32-512 workgroups on 256
// CUs; together they are one launch that fills the chip.
constexpr int kBnMaxItems = 8;
struct BnTable {
  int n;
  int cstart[kBnMaxItems + 1];
  BnArgs item[kBnMaxItems];
  float* y[kBnMaxItems];
};
struct BnBwdTable {
  int n;
  int cstart[kBnMaxItems + 1];
  BnBwdArgs item[kBnMaxItems];
};

template <typename T>
__device__ __forceinline__ int bn_item_of(const T& t, int& c) {
  int i = 0;
  while (i + 1 < t.n && c >= t.cstart[i + 1]) ++i;
  c -= t.cstart[i];
  return i;
}

template <int NV>
__global__ void __launch_bounds__(kThreads) bn_fwd_reg_kernel(BnTable t) {
  int c = blockIdx.x;
  const int i = bn_item_of(t, c);
  bn_fwd_reg_body<NV>(t.item[i], t.y[i], c);
}
template <int NV>
__global__ void __launch_bounds__(kThreads) bn_bwd_reg_kernel(BnBwdTable t) {
  int c = blockIdx.x;
  const int i = bn_item_of(t, c);
  bn_bwd_reg_body<NV>(t.item[i], c);
}
template <bool VEC>
__global__ void __launch_bounds__(kThreads) bn_fwd_loop_kernel(BnTable t) {
  int c = blockIdx.x;
  const int i = bn_item_of(t, c);
  bn_fwd_loop_body<VEC>(t.item[i], t.y[i], c);
}
template <bool VEC>
__global__ void __launch_bounds__(kThreads) bn_bwd_loop_kernel(BnBwdTable t) {
  int c = blockIdx.x;
  const int i = bn_item_of(t, c);
  bn_bwd_loop_body<VEC>(t.item[i], c);
}

int nv_for(long long quads) {
  const long long per = (quads + kThreads - 1) / kThreads;
  int nv = 1;
  while (nv < per) nv <<= 1;
  return nv;
}

// any channel with at least two values (one value has no variance: torch raises) that a workgroup can index with ints
bool shape_ok(int B, int C, int N) {
  if (B <= 0 || C <= 0 || N <= 0) return false;
  const long long M = (long long)B * N;
  return M >= 2 && M <= 0x7fffffffLL;
}

// the channel fits the registers (float4 rows, NV <= 8)
bool reg_ok(int B, int N) { return (N & 3) == 0 && (long long)B * (N >> 2) <= (long long)kThreads * 8; }

}  // namespace

#define CT_BN_DISPATCH(NVV, KERNEL, ...)                                                                 \
  switch (NVV) {                                                                                         \
    case 1: hipLaunchKernelGGL((KERNEL<1>), dim3(C), dim3(kThreads), 0, stream, __VA_ARGS__); break;     \
    case 2: hipLaunchKernelGGL((KERNEL<2>), dim3(C), dim3(kThreads), 0, stream, __VA_ARGS__); break;     \
    case 4: hipLaunchKernelGGL((KERNEL<4>), dim3(C), dim3(kThreads), 0, stream, __VA_ARGS__); break;     \
    default: hipLaunchKernelGGL((KERNEL<8>), dim3(C), dim3(kThreads), 0, stream, __VA_ARGS__); break;    \
  }

// 1 when ct_bn_relu_fwd / _bwd take (B, C, N): every shape with 2 <= B*N < 2^31.  Channels with N % 4 == 0 and
// B*N <= 32768 are held in registers (one read, one write); the others are re-read per pass.
extern "C" int ct_bn_relu_supported(int B, int C, int N) { return shape_ok(B, C, N) ? 1 : 0; }

// a batch stride is valid when it is 0 (= contiguous, C*N) or at least C*N
static bool stride_ok(long long bs, int C, int N, long long& out) {
  const long long dense = (long long)C * N;
  if (bs == 0) { out = dense; return true; }
  if (bs < dense) return false;
  out = bs;
  return true;
}

// strides and the float4 condition of one norm of a launch
static int bn_fwd_prepare(BnArgs& a, float* y, long long x_batch_stride, long long y_batch_stride, long long residual_batch_stride,
                          bool& vec) {
  if (!shape_ok(a.B, a.C, a.N)) return CT_EINVAL;
  if (!stride_ok(x_batch_stride, a.C, a.N, a.xbs) || !stride_ok(y_batch_stride, a.C, a.N, a.ybs) ||
      !stride_ok(residual_batch_stride, a.C, a.N, a.rbs))
    return CT_EINVAL;
  vec = (a.N & 3) == 0 && ((a.xbs | a.ybs | a.rbs) & 3) == 0 &&
        ((((uintptr_t)a.x) | ((uintptr_t)y) | ((uintptr_t)a.residual)) & 15) == 0;
  return CT_OK;
}

static int bn_fwd_launch_table(BnTable& t, bool vec, hipStream_t stream) {
  const int B = t.item[0].B, N = t.item[0].N, C = t.cstart[t.n];
  CT_CLEAR_ERROR();
  if (vec && reg_ok(B, N)) {
    CT_BN_DISPATCH(nv_for((long long)B * (N >> 2)), bn_fwd_reg_kernel, t)
  } else if (vec) {
    hipLaunchKernelGGL(bn_fwd_loop_kernel<true>, dim3(C), dim3(kThreads), 0, stream, t);
  } else {
    hipLaunchKernelGGL(bn_fwd_loop_kernel<false>, dim3(C), dim3(kThreads), 0, stream, t);
  }
  CT_CHECK_LAUNCH();
  return CT_OK;
}

static int bn_fwd_launch(BnArgs a, float* y, long long x_batch_stride, long long y_batch_stride, long long residual_batch_stride,
                         hipStream_t stream) {
  bool vec;
  const int rc = bn_fwd_prepare(a, y, x_batch_stride, y_batch_stride, residual_batch_stride, vec);
  if (rc != CT_OK) return rc;
  BnTable t{};
  t.n = 1; t.cstart[0] = 0; t.cstart[1] = a.C; t.item[0] = a; t.y[0] = y;
  return bn_fwd_launch_table(t, vec, stream);
}

static int bn_bwd_prepare(BnBwdArgs& a, long long x_batch_stride, long long gy_batch_stride, long long gx_batch_stride, bool& vec) {
  if (!shape_ok(a.B, a.C, a.N)) return CT_EINVAL;
  if (!stride_ok(x_batch_stride, a.C, a.N, a.xbs) || !stride_ok(gy_batch_stride, a.C, a.N, a.gybs) ||
      !stride_ok(gx_batch_stride, a.C, a.N, a.gxbs))
    return CT_EINVAL;
  vec = (a.N & 3) == 0 && ((a.xbs | a.gybs | a.gxbs) & 3) == 0 &&
        ((((uintptr_t)a.x) | ((uintptr_t)a.gy) | ((uintptr_t)a.gx)) & 15) == 0;
  return CT_OK;
}

static int bn_bwd_launch_table(BnBwdTable& t, bool vec, hipStream_t stream) {
  const int B = t.item[0].B, N = t.item[0].N, C = t.cstart[t.n];
  CT_CLEAR_ERROR();
  if (vec && reg_ok(B, N)) {
    CT_BN_DISPATCH(nv_for((long long)B * (N >> 2)), bn_bwd_reg_kernel, t)
  } else if (vec) {
    hipLaunchKernelGGL(bn_bwd_loop_kernel<true>, dim3(C), dim3(kThreads), 0, stream, t);
  } else {
    hipLaunchKernelGGL(bn_bwd_loop_kernel<false>, dim3(C), dim3(kThreads), 0, stream, t);
  }
  CT_CHECK_LAUNCH();
  return CT_OK;
}

static int bn_bwd_launch(BnBwdArgs a, long long x_batch_stride, long long gy_batch_stride, long long gx_batch_stride,
                         hipStream_t stream) {
  bool vec;
  const int rc = bn_bwd_prepare(a, x_batch_stride, gy_batch_stride, gx_batch_stride, vec);
  if (rc != CT_OK) return rc;
  BnBwdTable t{};
  t.n = 1; t.cstart[0] = 0; t.cstart[1] = a.C; t.item[0] = a;
  return bn_bwd_launch_table(t, vec, stream);
}

// Up to kBnMaxItems norms over the same (B, N) in one launch: ct_bn_relu_fwd_amax / _bwd_amax of every item.
extern "C" int ct_bn_group_fwd(const ct_bn_fwd_item* items, int n, int B, int N, ct_stream_t s) {
  if (!items || n < 1 || n > kBnMaxItems) return CT_EINVAL;
  BnTable t{};
  t.n = n;
  bool vec_all = true;
  int c0 = 0;
  for (int i = 0; i < n; ++i) {
    const ct_bn_fwd_item& it = items[i];
    if (!it.x || !it.weight || !it.bias || !it.y || !it.save_mean || !it.save_rstd || !(it.eps >= 0.0f)) return CT_EINVAL;
    if ((it.running_mean == nullptr) != (it.running_var == nullptr)) return CT_EINVAL;
    BnArgs a{it.x, it.weight, it.bias, it.running_mean, it.running_var, it.save_mean, it.save_rstd, B, it.C, N, it.eps, it.momentum,
             it.relu, 0, 0, it.residual, 0, it.num_batches_tracked, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, it.amax_out};
    bool vec;
    const int rc = bn_fwd_prepare(a, it.y, it.x_batch_stride, it.y_batch_stride, it.residual_batch_stride, vec);
    if (rc != CT_OK) return rc;
    vec_all = vec_all && vec;
    t.cstart[i] = c0;
    t.item[i] = a;
    t.y[i] = it.y;
    c0 += it.C;
  }
  t.cstart[n] = c0;
  return bn_fwd_launch_table(t, vec_all, (hipStream_t)s);
}

extern "C" int ct_bn_group_bwd(const ct_bn_bwd_item* items, int n, int B, int N, ct_stream_t s) {
  if (!items || n < 1 || n > kBnMaxItems) return CT_EINVAL;
  BnBwdTable t{};
  t.n = n;
  bool vec_all = true;
  int c0 = 0;
  for (int i = 0; i < n; ++i) {
    const ct_bn_bwd_item& it = items[i];
    if (!it.x || !it.weight || !it.bias || !it.save_mean || !it.save_rstd || !it.gy || !it.gx || !it.g_weight || !it.g_bias)
      return CT_EINVAL;
    BnBwdArgs a{it.x, it.weight, it.bias, it.save_mean, it.save_rstd, it.gy, it.gx, it.g_weight, it.g_bias, B, it.C, N, it.relu,
                0, 0, 0, 0, nullptr, nullptr, nullptr, it.amax_out};
    bool vec;
    const int rc = bn_bwd_prepare(a, it.x_batch_stride, it.gy_batch_stride, it.gx_batch_stride, vec);
    if (rc != CT_OK) return rc;
    vec_all = vec_all && vec;
    t.cstart[i] = c0;
    t.item[i] = a;
    c0 += it.C;
  }
  t.cstart[n] = c0;
  return bn_bwd_launch_table(t, vec_all, (hipStream_t)s);
}

// ---- the norms of a group around ONE statistics exchange (SyncBatchNorm): each phase of all items in ONE launch ----
// Buffer layouts (Ct = sum of the items' C, item i's channels start at c0_i):
//   local / every rank's block of `gathered` (stride 2 Ct + 1): [mean: Ct | sum (x - mean)^2: Ct | count: 1]
//   sums: [sum g': Ct | sum g' xhat: Ct]
static int bn_group_fwd_table(const ct_bn_fwd_item* items, int n, int B, int N, int mode, float* local, const float* gathered,
                              int world, float* count_out, BnTable& t, bool& vec_all) {
  if (!items || n < 1 || n > kBnMaxItems) return CT_EINVAL;
  int Ct = 0;
  for (int i = 0; i < n; ++i) Ct += items[i].C;
  const long long stride = 2ll * Ct + 1;
  t.n = n;
  vec_all = true;
  int c0 = 0;
  for (int i = 0; i < n; ++i) {
    const ct_bn_fwd_item& it = items[i];
    if (!it.x || it.C < 1) return CT_EINVAL;
    BnArgs a{};
    bool vec;
    int rc;
    if (mode == 1) {          // local statistics only: nothing but `local` is written
      a = BnArgs{it.x, nullptr, nullptr, nullptr, nullptr, local + c0, local + Ct + c0, B, it.C, N, 0.0f, 0.0f, 0, 0, 0, nullptr, 0,
                 nullptr, 1, i == 0 ? local + 2 * Ct : nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr};
      rc = bn_fwd_prepare(a, const_cast<float*>(it.x), it.x_batch_stride, it.x_batch_stride, 0, vec);
      t.y[i] = const_cast<float*>(it.x);
    } else {
      if (!it.weight || !it.bias || !it.y || !it.save_mean || !it.save_rstd || !(it.eps >= 0.0f)) return CT_EINVAL;
      if ((it.running_mean == nullptr) != (it.running_var == nullptr)) return CT_EINVAL;
      a = BnArgs{it.x, it.weight, it.bias, it.running_mean, it.running_var, it.save_mean, it.save_rstd, B, it.C, N, it.eps,
                 it.momentum, it.relu, 0, 0, it.residual, 0, it.num_batches_tracked, 2, i == 0 ? count_out : nullptr,
                 gathered + c0, gathered + Ct + c0, gathered + 2 * Ct, world, stride, it.amax_out};
      rc = bn_fwd_prepare(a, it.y, it.x_batch_stride, it.y_batch_stride, it.residual_batch_stride, vec);
      t.y[i] = it.y;
    }
    if (rc != CT_OK) return rc;
    vec_all = vec_all && vec;
    t.cstart[i] = c0;
    t.item[i] = a;
    c0 += it.C;
  }
  t.cstart[n] = c0;
  return CT_OK;
}

extern "C" int ct_bn_group_stats_fwd(const ct_bn_fwd_item* items, int n, int B, int N, float* local, ct_stream_t s) {
  if (!local) return CT_EINVAL;
  BnTable t{};
  bool vec;
  const int rc = bn_group_fwd_table(items, n, B, N, 1, local, nullptr, 0, nullptr, t, vec);
  return rc != CT_OK ? rc : bn_fwd_launch_table(t, vec, (hipStream_t)s);
}

extern "C" int ct_bn_group_apply_fwd(const ct_bn_fwd_item* items, int n, int B, int N, const float* gathered, int world,
                                     float* count_total, ct_stream_t s) {
  if (!gathered || world < 1) return CT_EINVAL;
  BnTable t{};
  bool vec;
  const int rc = bn_group_fwd_table(items, n, B, N, 2, nullptr, gathered, world, count_total, t, vec);
  return rc != CT_OK ? rc : bn_fwd_launch_table(t, vec, (hipStream_t)s);
}

static int bn_group_bwd_table(const ct_bn_bwd_item* items, int n, int B, int N, int mode, float* sums, const float* count,
                              BnBwdTable& t, bool& vec_all, float* sums_copy = nullptr) {
  if (!items || n < 1 || n > kBnMaxItems || !sums) return CT_EINVAL;
  int Ct = 0;
  for (int i = 0; i < n; ++i) Ct += items[i].C;
  t.n = n;
  vec_all = true;
  int c0 = 0;
  for (int i = 0; i < n; ++i) {
    const ct_bn_bwd_item& it = items[i];
    if (!it.x || !it.weight || !it.bias || !it.save_mean || !it.save_rstd || !it.gy || it.C < 1) return CT_EINVAL;
    BnBwdArgs a{};
    bool vec;
    int rc;
    if (mode == 1) {          // this rank's two sums per channel (g_bias = sum g' -> sums[c], g_weight = sum g' xhat -> sums[Ct + c])
      a = BnBwdArgs{it.x, it.weight, it.bias, it.save_mean, it.save_rstd, it.gy, const_cast<float*>(it.x), sums + Ct + c0, sums + c0,
                    B, it.C, N, it.relu, 0, 0, 0, 1, nullptr, nullptr, nullptr, nullptr,
                    sums_copy ? sums_copy + c0 : nullptr, sums_copy ? sums_copy + Ct + c0 : nullptr};
      rc = bn_bwd_prepare(a, it.x_batch_stride, it.gy_batch_stride, it.x_batch_stride, vec);
    } else {
      if (!it.gx || !count) return CT_EINVAL;
      a = BnBwdArgs{it.x, it.weight, it.bias, it.save_mean, it.save_rstd, it.gy, it.gx, nullptr, nullptr, B, it.C, N, it.relu, 0, 0, 0,
                    2, sums + c0, sums + Ct + c0, count, it.amax_out};
      rc = bn_bwd_prepare(a, it.x_batch_stride, it.gy_batch_stride, it.gx_batch_stride, vec);
    }
    if (rc != CT_OK) return rc;
    vec_all = vec_all && vec;
    t.cstart[i] = c0;
    t.item[i] = a;
    c0 += it.C;
  }
  t.cstart[n] = c0;
  return CT_OK;
}

extern "C" int ct_bn_group_reduce_bwd(const ct_bn_bwd_item* items, int n, int B, int N, float* sums, ct_stream_t s) {
  BnBwdTable t{};
  bool vec;
  const int rc = bn_group_bwd_table(items, n, B, N, 1, sums, nullptr, t, vec);
  return rc != CT_OK ? rc : bn_bwd_launch_table(t, vec, (hipStream_t)s);
}

extern "C" int ct_bn_group_reduce_bwd_copy(const ct_bn_bwd_item* items, int n, int B, int N, float* sums, float* sums_copy,
                                           ct_stream_t s) {
  if (!sums_copy) return CT_EINVAL;
  BnBwdTable t{};
  bool vec;
  const int rc = bn_group_bwd_table(items, n, B, N, 1, sums, nullptr, t, vec, sums_copy);
  return rc != CT_OK ? rc : bn_bwd_launch_table(t, vec, (hipStream_t)s);
}

extern "C" int ct_bn_group_apply_bwd(const ct_bn_bwd_item* items, int n, int B, int N, const float* sums, const float* count,
                                     ct_stream_t s) {
  BnBwdTable t{};
  bool vec;
  const int rc = bn_group_bwd_table(items, n, B, N, 2, const_cast<float*>(sums), count, t, vec);
  return rc != CT_OK ? rc : bn_bwd_launch_table(t, vec, (hipStream_t)s);
}

extern "C" int ct_bn_relu_fwd(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                              float* running_mean, float* running_var, long long* num_batches_tracked,
                              const float* residual, long long residual_batch_stride, float* y, long long y_batch_stride,
                              float* save_mean, float* save_rstd, int B, int C, int N, float eps, float momentum,
                              int relu, ct_stream_t s) {
  if (!x || !weight || !bias || !y || !save_mean || !save_rstd || !(eps >= 0.0f)) return CT_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return CT_EINVAL;
  BnArgs a{x, weight, bias, running_mean, running_var, save_mean, save_rstd, B, C, N, eps, momentum, relu, 0, 0,
           residual, 0, num_batches_tracked, 0, nullptr, nullptr, nullptr, nullptr, 0, 0};
  return bn_fwd_launch(a, y, x_batch_stride, y_batch_stride, residual_batch_stride, (hipStream_t)s);
}

extern "C" int ct_bn_relu_bwd(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                              const float* save_mean, const float* save_rstd, const float* gy, long long gy_batch_stride,
                              float* gx, long long gx_batch_stride, float* g_weight, float* g_bias, int B, int C, int N,
                              int relu, ct_stream_t s) {
  if (!x || !weight || !bias || !save_mean || !save_rstd || !gy || !gx || !g_weight || !g_bias) return CT_EINVAL;
  BnBwdArgs a{x, weight, bias, save_mean, save_rstd, gy, gx, g_weight, g_bias, B, C, N, relu, 0, 0, 0, 0, nullptr, nullptr, nullptr};
  return bn_bwd_launch(a, x_batch_stride, gy_batch_stride, gx_batch_stride, (hipStream_t)s);
}

// The same two with amax_out f32[C] (nullable): max |y| / max |gx| per channel, for ct_pw_gemm's operand scale.
extern "C" int ct_bn_relu_fwd_amax(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                                   float* running_mean, float* running_var, long long* num_batches_tracked,
                                   const float* residual, long long residual_batch_stride, float* y, long long y_batch_stride,
                                   float* save_mean, float* save_rstd, float* amax_out, int B, int C, int N, float eps,
                                   float momentum, int relu, ct_stream_t s) {
  if (!x || !weight || !bias || !y || !save_mean || !save_rstd || !(eps >= 0.0f)) return CT_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return CT_EINVAL;
  BnArgs a{x, weight, bias, running_mean, running_var, save_mean, save_rstd, B, C, N, eps, momentum, relu, 0, 0,
           residual, 0, num_batches_tracked, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, amax_out};
  return bn_fwd_launch(a, y, x_batch_stride, y_batch_stride, residual_batch_stride, (hipStream_t)s);
}

extern "C" int ct_bn_relu_bwd_amax(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                                   const float* save_mean, const float* save_rstd, const float* gy, long long gy_batch_stride,
                                   float* gx, long long gx_batch_stride, float* g_weight, float* g_bias, float* amax_out,
                                   int B, int C, int N, int relu, ct_stream_t s) {
  if (!x || !weight || !bias || !save_mean || !save_rstd || !gy || !gx || !g_weight || !g_bias) return CT_EINVAL;
  BnBwdArgs a{x, weight, bias, save_mean, save_rstd, gy, gx, g_weight, g_bias, B, C, N, relu, 0, 0, 0, 0, nullptr, nullptr, nullptr,
              amax_out};
  return bn_bwd_launch(a, x_batch_stride, gy_batch_stride, gx_batch_stride, (hipStream_t)s);
}

// ---- the same norm split around a statistics exchange between ranks (SyncBatchNorm under data parallelism) ----

extern "C" int ct_bn_stats_fwd(const float* x, long long x_batch_stride, float* mean, float* m2, float* count, int B, int C,
                               int N, ct_stream_t s) {
  if (!x || !mean || !m2) return CT_EINVAL;
  BnArgs a{x, nullptr, nullptr, nullptr, nullptr, mean, m2, B, C, N, 0.0f, 0.0f, 0, 0, 0, nullptr, 0, nullptr,
           1, count, nullptr, nullptr, nullptr, 0, 0};
  return bn_fwd_launch(a, const_cast<float*>(x), x_batch_stride, x_batch_stride, 0, (hipStream_t)s);
}

static int bn_apply_fwd_impl(float* amax_out, const float* x, long long x_batch_stride, const float* weight, const float* bias,
                               const float* g_mean, const float* g_m2, const float* g_count, int world, long long g_stride,
                               float* running_mean, float* running_var, long long* num_batches_tracked,
                               const float* residual, long long residual_batch_stride, float* y, long long y_batch_stride,
                               float* save_mean, float* save_rstd, float* count_total, int B, int C, int N, float eps,
                               float momentum, int relu, ct_stream_t s) {
  if (!x || !weight || !bias || !y || !g_mean || !g_m2 || !g_count || !save_mean || !save_rstd || !(eps >= 0.0f) ||
      world < 1 || g_stride < 1)
    return CT_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return CT_EINVAL;
  BnArgs a{x, weight, bias, running_mean, running_var, save_mean, save_rstd, B, C, N, eps, momentum, relu, 0, 0,
           residual, 0, num_batches_tracked, 2, count_total, g_mean, g_m2, g_count, world, g_stride, amax_out};
  return bn_fwd_launch(a, y, x_batch_stride, y_batch_stride, residual_batch_stride, (hipStream_t)s);
}

extern "C" int ct_bn_apply_fwd(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                               const float* g_mean, const float* g_m2, const float* g_count, int world, long long g_stride,
                               float* running_mean, float* running_var, long long* num_batches_tracked,
                               const float* residual, long long residual_batch_stride, float* y, long long y_batch_stride,
                               float* save_mean, float* save_rstd, float* count_total, int B, int C, int N, float eps,
                               float momentum, int relu, ct_stream_t s) {
  return bn_apply_fwd_impl(nullptr, x, x_batch_stride, weight, bias, g_mean, g_m2, g_count, world, g_stride, running_mean, running_var,
                           num_batches_tracked, residual, residual_batch_stride, y, y_batch_stride, save_mean, save_rstd, count_total,
                           B, C, N, eps, momentum, relu, s);
}

// with amax_out f32[C] (nullable), as ct_bn_relu_fwd_amax
extern "C" int ct_bn_apply_fwd_amax(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                                    const float* g_mean, const float* g_m2, const float* g_count, int world, long long g_stride,
                                    float* running_mean, float* running_var, long long* num_batches_tracked,
                                    const float* residual, long long residual_batch_stride, float* y, long long y_batch_stride,
                                    float* save_mean, float* save_rstd, float* count_total, float* amax_out, int B, int C, int N,
                                    float eps, float momentum, int relu, ct_stream_t s) {
  return bn_apply_fwd_impl(amax_out, x, x_batch_stride, weight, bias, g_mean, g_m2, g_count, world, g_stride, running_mean,
                           running_var, num_batches_tracked, residual, residual_batch_stride, y, y_batch_stride, save_mean,
                           save_rstd, count_total, B, C, N, eps, momentum, relu, s);
}

extern "C" int ct_bn_reduce_bwd(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                                const float* mean, const float* rstd, const float* gy, long long gy_batch_stride,
                                float* sum_g, float* sum_gxhat, int B, int C, int N, int relu, ct_stream_t s) {
  if (!x || !weight || !bias || !mean || !rstd || !gy || !sum_g || !sum_gxhat) return CT_EINVAL;
  BnBwdArgs a{x, weight, bias, mean, rstd, gy, const_cast<float*>(x), sum_gxhat, sum_g, B, C, N, relu, 0, 0, 0,
              1, nullptr, nullptr, nullptr};
  return bn_bwd_launch(a, x_batch_stride, gy_batch_stride, x_batch_stride, (hipStream_t)s);
}

static int bn_apply_bwd_impl(float* amax_out, const float* x, long long x_batch_stride, const float* weight, const float* bias,
                               const float* mean, const float* rstd, const float* gy, long long gy_batch_stride,
                               const float* sum_g, const float* sum_gxhat, const float* count, float* gx,
                               long long gx_batch_stride, int B, int C, int N, int relu, ct_stream_t s) {
  if (!x || !weight || !bias || !mean || !rstd || !gy || !gx || !sum_g || !sum_gxhat || !count) return CT_EINVAL;
  BnBwdArgs a{x, weight, bias, mean, rstd, gy, gx, nullptr, nullptr, B, C, N, relu, 0, 0, 0, 2, sum_g, sum_gxhat, count, amax_out};
  return bn_bwd_launch(a, x_batch_stride, gy_batch_stride, gx_batch_stride, (hipStream_t)s);
}

extern "C" int ct_bn_apply_bwd(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                               const float* mean, const float* rstd, const float* gy, long long gy_batch_stride,
                               const float* sum_g, const float* sum_gxhat, const float* count, float* gx,
                               long long gx_batch_stride, int B, int C, int N, int relu, ct_stream_t s) {
  return bn_apply_bwd_impl(nullptr, x, x_batch_stride, weight, bias, mean, rstd, gy, gy_batch_stride, sum_g, sum_gxhat, count, gx,
                           gx_batch_stride, B, C, N, relu, s);
}

// with amax_out f32[C] (nullable), as ct_bn_relu_bwd_amax
extern "C" int ct_bn_apply_bwd_amax(const float* x, long long x_batch_stride, const float* weight, const float* bias,
                                    const float* mean, const float* rstd, const float* gy, long long gy_batch_stride,
                                    const float* sum_g, const float* sum_gxhat, const float* count, float* gx,
                                    long long gx_batch_stride, float* amax_out, int B, int C, int N, int relu, ct_stream_t s) {
  return bn_apply_bwd_impl(amax_out, x, x_batch_stride, weight, bias, mean, rstd, gy, gy_batch_stride, sum_g, sum_gxhat, count, gx,
                           gx_batch_stride, B, C, N, relu, s);
}
