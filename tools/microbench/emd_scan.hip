// Micro-benchmark: one auction bidder scanning n targets with one 256-thread workgroup (the late iterations of
// emd_bid_kernel: a handful of bidders, one workgroup each) — what does the scan cost, and which part of it?
//   variant 0: the bid value as the kernel computes it: (float)(3.0 - (double)sqrtf(d2) - (double)price), top-2 push
//   variant 1: all in float (3.0f - sqrtf(d2) - price)
//   variant 2: no sqrt (3.0f - d2 - price)
//   variant 3: loads only (sum)
// Prints us per launch for U bidders (workgroups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Top2 { float best, better; int idx; };
__device__ __forceinline__ void push(Top2& t, float d, int idx) {
  t.idx = d > t.best ? idx : t.idx;
  t.better = __builtin_amdgcn_fmed3f(t.best, t.better, d);
  t.best = fmaxf(t.best, d);
}

template <int VAR, int KB>
__global__ void __launch_bounds__(256) scan(const float* __restrict__ pts, const float* __restrict__ prc, const float* __restrict__ q,
                                            float* out, int n) {
  const float x1 = q[blockIdx.x * 3], y1 = q[blockIdx.x * 3 + 1], z1 = q[blockIdx.x * 3 + 2];
  const int sub = threadIdx.x, T = 256;
  Top2 t2 = {-1e9f, -1e9f, 0x7fffffff};
  float cur[KB][4], nxt[KB][4];
  auto loadb = [&](int k, float (&v)[KB][4]) {
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int kk = k + u * T;
      v[u][0] = pts[kk * 3]; v[u][1] = pts[kk * 3 + 1]; v[u][2] = pts[kk * 3 + 2]; v[u][3] = prc[kk];
    }
  };
  loadb(sub, cur);
  for (int k = sub; k < n; k += KB * T) {
    const bool more = k + KB * T < n;
    if (more) loadb(k + KB * T, nxt);
    float d[KB];
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const float x2 = cur[u][0] - x1, y2 = cur[u][1] - y1, z2 = cur[u][2] - z1;
      const float d2 = fmaf(z2, z2, fmaf(y2, y2, x2 * x2));
      if (VAR == 0) d[u] = (float)(3.0 - (double)sqrtf(d2) - (double)cur[u][3]);
      else if (VAR == 1) d[u] = 3.0f - sqrtf(d2) - cur[u][3];
      else if (VAR == 2) d[u] = 3.0f - d2 - cur[u][3];
      else d[u] = cur[u][0] + cur[u][1] + cur[u][2] + cur[u][3];
    }
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      if (VAR == 3) t2.best += d[u];
      else push(t2, d[u], k + u * T);
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < KB; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) cur[u][c] = nxt[u][c];
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = t2.best + t2.better + (float)t2.idx;
}

__global__ void touch(float* prc, int n) {      // another kernel writes the prices between two scans, as emd_update_kernel does
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) prc[i] += 1e-9f;
}

// scan launches alternating with a writer kernel, all queued back to back: (time of 50 pairs) - (time of 50 writers)
template <int VAR, int KB>
float run_after_writer(const float* pts, float* prc, const float* q, float* out, int n, int U) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 50;
  float both, alone;
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(touch, dim3(n / 256), dim3(256), 0, 0, prc, n);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) {
      hipLaunchKernelGGL(touch, dim3(n / 256), dim3(256), 0, 0, prc, n);
      if (pass == 0) hipLaunchKernelGGL((scan<VAR, KB>), dim3(U), dim3(256), 0, 0, pts, prc, q, out, n);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    (pass == 0 ? both : alone) = ms;
  }
  return (both - alone) / reps * 1e3f;
}

template <int VAR, int KB>
float run(const float* pts, const float* prc, const float* q, float* out, int n, int U) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((scan<VAR, KB>), dim3(U), dim3(256), 0, 0, pts, prc, q, out, n);
  hipEventRecord(e0);
  const int reps = 50;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((scan<VAR, KB>), dim3(U), dim3(256), 0, 0, pts, prc, q, out, n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

int main() {
  const int n = 16384;
  std::vector<float> h(n * 3), p(n), q(1024 * 3);
  unsigned s = 12345;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return (s >> 8) * (1.0f / 16777216.0f); };
  for (auto& v : h) v = rnd();
  for (auto& v : p) v = rnd() * 0.1f;
  for (auto& v : q) v = rnd();
  float *dp, *dprc, *dq, *dout;
  hipMalloc(&dp, h.size() * 4); hipMalloc(&dprc, p.size() * 4); hipMalloc(&dq, q.size() * 4); hipMalloc(&dout, 1024 * 256 * 4);
  hipMemcpy(dp, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dprc, p.data(), p.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice);
  const int Us[] = {1, 32, 256, 1024};
  for (int U : Us) {
    printf("U=%4d bidders, n=%d: double %.1f us | float %.1f | no sqrt %.1f | loads only %.1f   (8 per batch)\n", U, n,
           run<0, 8>(dp, dprc, dq, dout, n, U), run<1, 8>(dp, dprc, dq, dout, n, U), run<2, 8>(dp, dprc, dq, dout, n, U),
           run<3, 8>(dp, dprc, dq, dout, n, U));
    printf("                             alternating with a writer kernel (pairs minus writers): double %.1f us | loads only %.1f\n",
           run_after_writer<0, 8>(dp, dprc, dq, dout, n, U), run_after_writer<3, 8>(dp, dprc, dq, dout, n, U));
    printf("                             double %.1f us | float %.1f | no sqrt %.1f | loads only %.1f   (4 per batch)\n",
           run<0, 4>(dp, dprc, dq, dout, n, U), run<1, 4>(dp, dprc, dq, dout, n, U), run<2, 4>(dp, dprc, dq, dout, n, U),
           run<3, 4>(dp, dprc, dq, dout, n, U));
  }
  return 0;
}
