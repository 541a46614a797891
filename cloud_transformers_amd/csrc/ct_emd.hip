// Approximate Earth Mover's Distance by parallel auction, for gfx950
// (replaces emd_linear/emd_cuda.cu:23-316 of the reference).
//
// Same algorithm and arithmetic as the reference, restructured for the machine:
//   * batches are independent, so everything that only needs a per-batch barrier
//     (GetMax -> Assign -> recount -> compaction of the unassigned list) is ONE
//     kernel with one 1024-thread workgroup per batch: an iteration is 2 launches
//     (update, bid) instead of the reference's 7;
//   * Bid is the O(U*n) part.  A 256-thread workgroup takes ceil(U / blocks)
//     bidders; T = 256 / bidders lanes (a power of two <= 64, inside one wave)
//     share a bidder and scan interleaved targets of an LDS tile
//     {x,y,z,price} (conflict-free ds_read_b128), then merge their (best,
//     second best) pairs with wave shuffles — no LDS reduction, no barrier;
//   * the per-target maximum increment is an integer atomicMax on the float's bit
//     pattern (increments are > 0), not a compare-and-swap loop;
//   * GetMax ties (several bidders within 1e-6 of the maximum) resolve to the
//     HIGHEST bidder index (the reference: last writer, unspecified), so the
//     result is deterministic and equal to oracle/emd_ref.c.
#include "ct_common.h"

namespace {

constexpr int kBidThreads = 256;
constexpr int kTile = 1024;   // targets per LDS tile (16 KiB as float4)

struct EmdWs {
  float* price;      // [B,n]
  float* bid_inc;    // [B,n]
  float* max_inc;    // [B,n]
  int* ass_inv;      // [B,n]
  int* bid;          // [B,n]
  int* max_idx;      // [B,n]
  int* unass_idx;    // [B,n]
  int* unass_cnt;    // [B]
};

__device__ __forceinline__ int ld_coherent(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_coherent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void emd_init_kernel(EmdWs w, int* assignment, size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  assignment[i] = -1;
  w.ass_inv[i] = -1;
  w.price[i] = 0.0f;
  w.max_inc[i] = 0.0f;       // the reference's caller passes zeros (emd_module.py:48)
  w.max_idx[i] = -1;
  w.bid[i] = 0;
  w.bid_inc[i] = 0.0f;
}

// One workgroup per batch.  do_assign: GetMax + Assign for the bids of the iteration
// that just ran (emd_cuda.cu:181-215).  do_compact: rebuild the unassigned list
// (emd_cuda.cu:30-93).  do_dist: CalcDist (emd_cuda.cu:217-226).
__global__ void __launch_bounds__(1024)
emd_update_kernel(EmdWs w, int* assignment, const float* xyz1, const float* xyz2, float* dist,
                  int n, int do_assign, int last, int do_compact, int do_dist) {
  __shared__ int s_scan[1024];
  __shared__ int s_base;
  const int b = blockIdx.x;
  const size_t off = (size_t)b * n;
  int* ass = assignment + off;
  int* ass_inv = w.ass_inv + off;
  int* bid = w.bid + off;
  int* max_idx = w.max_idx + off;
  float* bid_inc = w.bid_inc + off;
  float* max_inc = w.max_inc + off;
  float* price = w.price + off;

  if (do_assign) {
    // GetMax
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      if (ass[j] == -1) {
        const int t = bid[j];
        const float bi = bid_inc[j];
        const float mi = ld_coherent(&max_inc[t]);
        if ((double)bi - 1e-6 <= (double)mi && (double)mi <= (double)bi + 1e-6) atomicMax(&max_idx[t], j);
      }
    }
    __threadfence_block();
    __syncthreads();
    // Assign
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      if (ld_coherent(&ass[j]) == -1) {
        const int t = bid[j];
        if (last || ld_coherent(&max_idx[t]) == j) {
          const float bi = bid_inc[j];
          if (!last) {
            const int prev = ass_inv[t];
            if (prev != -1) __hip_atomic_store(&ass[prev], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ass_inv[t] = j;
            price[t] += bi;
            max_inc[t] = -1e9f;
            max_idx[t] = -1;
            __hip_atomic_store(&ass[j], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            // forced assignment of every remaining bidder: several bidders may share a
            // target, so the price update must be atomic to stay well defined
            ass_inv[t] = j;
            atomicAdd(&price[t], bi);
            max_inc[t] = -1e9f;
            __hip_atomic_store(&ass[j], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    __threadfence_block();
    __syncthreads();
  }

  if (do_compact) {
    // ascending list of unassigned points (block-wide exclusive scan per 1024-point chunk)
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += blockDim.x) {
      const int j = c0 + threadIdx.x;
      const int flag = (j < n && ld_coherent(&ass[j]) == -1) ? 1 : 0;
      // inclusive scan: wave shuffle + per-wave totals
      int v = flag;
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
      }
      if (lane == 63) s_scan[wave] = v;
      __syncthreads();
      int wave_off = 0;
      for (int q = 0; q < wave; ++q) wave_off += s_scan[q];
      const int base = s_base;
      if (flag) w.unass_idx[off + base + wave_off + v - 1] = j;
      __syncthreads();
      if (threadIdx.x == blockDim.x - 1) s_base = base + wave_off + v;
      __syncthreads();
    }
    if (threadIdx.x == 0) w.unass_cnt[b] = s_base;
  }

  if (do_dist) {
    const float* p1 = xyz1 + off * 3;
    const float* p2 = xyz2 + off * 3;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      const int k = ld_coherent(&ass[j]);
      const float dx = p1[j * 3 + 0] - p2[k * 3 + 0];
      const float dy = p1[j * 3 + 1] - p2[k * 3 + 1];
      const float dz = p1[j * 3 + 2] - p2[k * 3 + 2];
      dist[off + j] = dx * dx + dy * dy + dz * dz;
    }
  }
}

struct Top2 {
  float best, better;
  int idx;
};

// merge two partial scans; ties on the value -> the lower target index, exactly what a
// single ascending scan with strict '>' produces (emd_cuda.cu:150-157)
__device__ __forceinline__ Top2 merge_top2(const Top2& a, const Top2& o) {
  Top2 r;
  const bool take_o = o.best > a.best || (o.best == a.best && (unsigned)o.idx < (unsigned)a.idx);
  if (take_o) {
    r.best = o.best; r.idx = o.idx; r.better = fmaxf(o.better, a.best);
  } else {
    r.best = a.best; r.idx = a.idx; r.better = fmaxf(a.better, o.best);
  }
  return r;
}

// grid = (blocks per batch, B)
__global__ void __launch_bounds__(kBidThreads)
emd_bid_kernel(EmdWs w, const float* __restrict__ xyz1, const float* __restrict__ xyz2, int n, float eps) {
  __shared__ float4 tile[kTile];
  const int b = blockIdx.y;
  const size_t off = (size_t)b * n;
  const int U = w.unass_cnt[b];
  if (U == 0) return;
  const int nblk = gridDim.x;
  const int per_blk = (U + nblk - 1) / nblk;                    // bidders of this workgroup
  const int first = blockIdx.x * per_blk;
  const int mine = max(0, min(per_blk, U - first));
  if (mine == 0) return;                                        // block-uniform
  int T = kBidThreads / per_blk;                                // lanes per bidder
  T = T < 1 ? 1 : (T > 64 ? 64 : T);
  T = 1 << (31 - __clz(T));                                     // power of two: bidders never straddle a wave
  // per_blk > 256 cannot happen: nblk = n/64 >= U/64  =>  per_blk <= 64
  const int slot = threadIdx.x / T, sub = threadIdx.x % T;
  const bool active = slot < mine;
  int j = -1;
  float x1 = 0, y1 = 0, z1 = 0;
  if (active) {
    j = w.unass_idx[off + first + slot];
    x1 = xyz1[(off + j) * 3 + 0];
    y1 = xyz1[(off + j) * 3 + 1];
    z1 = xyz1[(off + j) * 3 + 2];
  }
  Top2 t2 = {-1e9f, -1e9f, 0x7fffffff};
  for (int k0 = 0; k0 < n; k0 += kTile) {
    const int cnt = min(kTile, n - k0);
    __syncthreads();
    for (int k = threadIdx.x; k < cnt; k += blockDim.x) {
      const float* p = xyz2 + (off + k0 + k) * 3;
      tile[k] = make_float4(p[0], p[1], p[2], w.price[off + k0 + k]);
    }
    __syncthreads();
    if (active) {
      for (int k = sub; k < cnt; k += T) {
        const float4 q = tile[k];
        const float x2 = q.x - x1, y2 = q.y - y1, z2 = q.z - z1;
        const float d2 = fmaf(z2, z2, fmaf(y2, y2, x2 * x2));
        // evaluated in double like the reference (its literal 3.0 is a double), rounded once
        const float d = (float)(3.0 - (double)sqrtf(d2) - (double)q.w);
        if (d > t2.best) {
          t2.better = t2.best; t2.best = d; t2.idx = k0 + k;
        } else if (d > t2.better) {
          t2.better = d;
        }
      }
    }
  }
  // butterfly over the T lanes of a bidder (all inside one wave; inactive lanes hold the identity)
  for (int m = 1; m < T; m <<= 1) {
    Top2 o;
    o.best = __shfl_xor(t2.best, m, 64);
    o.better = __shfl_xor(t2.better, m, 64);
    o.idx = __shfl_xor(t2.idx, m, 64);
    t2 = merge_top2(t2, o);
  }
  if (active && sub == 0) {
    const float inc = t2.best - t2.better + eps;
    w.bid[off + j] = t2.idx;
    w.bid_inc[off + j] = inc;
    // increments are positive: float order == signed-int order of the bit patterns,
    // also against the -1e9 / 0 the slot holds between rounds
    atomicMax((int*)&w.max_inc[off + t2.idx], __float_as_int(inc));
  }
}

__global__ void emd_grad_kernel(const float* xyz1, const float* xyz2, const float* g_dist, const int* assignment,
                                float* g_xyz1, int n, size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t b = i / n;
  const int k = assignment[i];
  const float g = g_dist[i] * 2.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) g_xyz1[i * 3 + c] = g * (xyz1[i * 3 + c] - xyz2[(b * n + k) * 3 + c]);
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

size_t ct_emd_workspace_bytes(int B, int n) {
  if (B <= 0 || n <= 0) return 0;
  return 7 * align256((size_t)B * n * 4) + align256((size_t)B * 4);
}

int ct_emd_fwd(const float* xyz1, const float* xyz2, float* dist, int32_t* assignment, void* workspace,
               size_t workspace_bytes, int B, int n, float eps, int iters, ct_stream_t s) {
  if (!xyz1 || !xyz2 || !dist || !assignment || !workspace || B <= 0 || n <= 0 || iters < 1) return CT_EINVAL;
  // emd_cuda.cu:236-249
  if (n % 1024 != 0 || B > 512) return CT_EPRECOND;
  if (workspace_bytes < ct_emd_workspace_bytes(B, n)) return CT_EWORKSPACE;
  hipStream_t st = (hipStream_t)s;
  char* p = (char*)workspace;
  const size_t seg = align256((size_t)B * n * 4);
  EmdWs w;
  w.price = (float*)p; p += seg;
  w.bid_inc = (float*)p; p += seg;
  w.max_inc = (float*)p; p += seg;
  w.ass_inv = (int*)p; p += seg;
  w.bid = (int*)p; p += seg;
  w.max_idx = (int*)p; p += seg;
  w.unass_idx = (int*)p; p += seg;
  w.unass_cnt = (int*)p;
  const size_t total = (size_t)B * n;
  CT_CLEAR_ERROR();
  hipLaunchKernelGGL(emd_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, (int*)assignment, total);
  hipLaunchKernelGGL(emd_update_kernel, dim3(B), dim3(1024), 0, st, w, (int*)assignment, xyz1, xyz2, dist, n, 0, 0, 1, 0);
  const dim3 bid_grid(n / 64, B);
  for (int it = 0; it < iters; ++it) {
    const int last = it == iters - 1;
    hipLaunchKernelGGL(emd_bid_kernel, bid_grid, dim3(kBidThreads), 0, st, w, xyz1, xyz2, n, eps);
    hipLaunchKernelGGL(emd_update_kernel, dim3(B), dim3(1024), 0, st, w, (int*)assignment, xyz1, xyz2, dist, n,
                       1, last, last ? 0 : 1, last);
  }
  CT_CHECK_LAUNCH();
  return CT_OK;
}

int ct_emd_bwd(const float* xyz1, const float* xyz2, const float* g_dist, const int32_t* assignment,
               float* g_xyz1, int B, int n, ct_stream_t s) {
  if (!xyz1 || !xyz2 || !g_dist || !assignment || !g_xyz1 || B <= 0 || n <= 0) return CT_EINVAL;
  const size_t total = (size_t)B * n;
  CT_CLEAR_ERROR();
  hipLaunchKernelGGL(emd_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s,
                     xyz1, xyz2, g_dist, (const int*)assignment, g_xyz1, n, total);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

}  // extern "C"
