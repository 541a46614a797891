// Approximate Earth Mover's Distance by parallel auction, for gfx950
// (replaces emd_linear/emd_cuda.cu:23-316 of the reference).
//
// Same algorithm and arithmetic as the reference, restructured for the machine:
//   * batches are independent, so everything that only needs a per-batch barrier
//     (GetMax -> Assign -> recount -> compaction of the unassigned list) is ONE
//     kernel with one 1024-thread workgroup per batch: an iteration is 2 launches
//     (update, bid) instead of the reference's 7;
//   * the per-batch kernel is latency-bound (dependent L2 accesses from one workgroup), so it walks the
//     unassigned LIST rather than all n points and batches the loads of each dependency level;
//   * Bid is the O(U*n) part.  A 512-thread workgroup takes ceil(U / blocks)
//     bidders; T = 512 / bidders lanes (a power of two, 8..256) share a bidder and scan interleaved
//     targets of an LDS tile {x,y,z,price} (conflict-free ds_read_b128, four in flight), then merge
//     their (best, second best) pairs with wave shuffles (and through LDS when a bidder spans whole
//     waves); the next tiles' global loads are in flight while a tile is scanned;
//   * the per-target maximum increment is an integer atomicMax on the float's bit
//     pattern (increments are > 0), not a compare-and-swap loop;
//   * GetMax ties (several bidders within 1e-6 of the maximum) resolve to the
//     HIGHEST bidder index (the reference: last writer, unspecified), so the
//     result is deterministic and equal to oracle/emd_ref.c.
#include "ct_common.h"
#include <cstdlib>

namespace {

constexpr int kBidThreads = 512;   // (256: 8 % slower over 50 iterations at n = 16384, 1024: 25 % slower)
constexpr int kTile = 1024;   // targets per LDS tile (16 KiB as float4)
#ifndef CT_EMD_BIG_FROM
#define CT_EMD_BIG_FROM 5
#endif
constexpr int kEmdBigFrom = CT_EMD_BIG_FROM;
constexpr int kEmdBigMaxU = 2048;   // ... for batches with at most this many bidders   // first iteration (0-based) on the 4096-target tiles

struct EmdWs {
  float* price;      // [B,n]
  float* bid_inc;    // [B,n]
  float* max_inc;    // [B,n]
  int* ass_inv;      // [B,n]
  int* bid;          // [B,n]
  int* max_idx;      // [B,n]
  int* unass_idx;    // [B,n]
  int* unass_cnt;    // [B]
  int* sync;         // [B][3]: arrivals before Assign | arrivals at the end | entries of the list being written (emd_update_multi_kernel)
};

__device__ __forceinline__ int ld_coherent(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_coherent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void emd_init_kernel(EmdWs w, int* assignment, size_t total, int nsync) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  assignment[i] = -1;
  w.ass_inv[i] = -1;
  w.price[i] = 0.0f;
  w.max_inc[i] = 0.0f;       // the reference's caller passes zeros (emd_module.py:48)
  w.max_idx[i] = -1;
  w.bid[i] = 0;
  w.bid_inc[i] = 0.0f;
  if (i < (size_t)nsync) w.sync[i] = 0;
}

// One workgroup per batch.  do_assign: GetMax + Assign for the bids of the iteration
// that just ran (emd_cuda.cu:181-215).  do_compact: rebuild the unassigned list
// (emd_cuda.cu:30-93).  do_dist: CalcDist (emd_cuda.cu:217-226).
__global__ void __launch_bounds__(1024)
emd_update_kernel(EmdWs w, int* assignment, const float* xyz1, const float* xyz2, float* dist,
                  int n, int do_assign, int last, int do_compact, int do_dist) {
  __shared__ int s_scan[16];
  const int b = blockIdx.x;
  const size_t off = (size_t)b * n;
  int* ass = assignment + off;
  int* ass_inv = w.ass_inv + off;
  int* bid = w.bid + off;
  int* max_idx = w.max_idx + off;
  float* bid_inc = w.bid_inc + off;
  float* max_inc = w.max_inc + off;
  float* price = w.price + off;

  // Every loop below is latency-bound (dependent L2 accesses, one workgroup per batch): each thread takes kUpR
  // entries per pass and issues the loads of one dependency level for all of them before using any, so a pass
  // costs a few memory latencies instead of a few per entry.  GetMax / Assign walk the unassigned list of the
  // previous compaction — exactly the points with assignment == -1, since only the bid kernel ran in between —
  // instead of testing all n points.
  constexpr int kUpR = 8;
  // Fast path (at most 8192 list entries, every iteration but the first few): a thread's entries stay in registers from GetMax
  // to Assign (two dependent load levels less), and the next list is written by the same threads — an entry that did not
  // win stays, an evicted owner joins — with wave-aggregated appends: no pass over all n assignments, no block scan.  The list
  // is then no longer ascending; its order decides which workgroup scans which bidder in Bid and nothing else (every
  // bid depends on the prices alone, the per-target maxima are atomic maxima, GetMax ties go to the highest index):
  // the assignment stays bit for bit the oracle's (tests/test_emd_gpu.py).
  if (do_assign && w.unass_cnt[b] <= 1024 * kUpR && (do_compact || last)) {
    __shared__ int s_new;
    const int U = w.unass_cnt[b];
    int* list = w.unass_idx + off;
    if (threadIdx.x == 0) s_new = 0;
    int j[kUpR], t[kUpR], prev[kUpR];
    float bi[kUpR], mi[kUpR];
    bool win[kUpR];
#pragma unroll
    for (int u = 0; u < kUpR; ++u) {
      const int i = u * 1024 + (int)threadIdx.x;
      j[u] = i < U ? list[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < kUpR; ++u) {
      t[u] = j[u] >= 0 ? bid[j[u]] : 0;
      bi[u] = j[u] >= 0 ? bid_inc[j[u]] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < kUpR; ++u) mi[u] = j[u] >= 0 ? max_inc[t[u]] : 0.0f;       // written by the bid kernel
#pragma unroll
    for (int u = 0; u < kUpR; ++u) {
      if (j[u] >= 0 && (double)bi[u] - 1e-6 <= (double)mi[u] && (double)mi[u] <= (double)bi[u] + 1e-6)
        atomicMax(&max_idx[t[u]], j[u]);
    }
    __threadfence_block();
    __syncthreads();                                  // every entry is in registers: the list may be rewritten from here on
#pragma unroll
    for (int u = 0; u < kUpR; ++u) win[u] = j[u] >= 0 && (last || ld_coherent(&max_idx[t[u]]) == j[u]);
#pragma unroll
    for (int u = 0; u < kUpR; ++u) prev[u] = (win[u] && !last) ? ass_inv[t[u]] : -1;
#pragma unroll
    for (int u = 0; u < kUpR; ++u) {
      if (win[u]) {
        if (!last) {
          if (prev[u] != -1) __hip_atomic_store(&ass[prev[u]], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ass_inv[t[u]] = j[u];
          price[t[u]] += bi[u];
          max_inc[t[u]] = -1e9f;
          max_idx[t[u]] = -1;
        } else {
          ass_inv[t[u]] = j[u];
          atomicAdd(&price[t[u]], bi[u]);
          max_inc[t[u]] = -1e9f;
        }
        __hip_atomic_store(&ass[j[u]], t[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (do_compact) {
      const int lane = threadIdx.x & 63;
      const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
      for (int u = 0; u < kUpR; ++u) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {                  // k = 0: the entry stays unassigned; k = 1: its target's previous owner is evicted
          const bool add = k == 0 ? (j[u] >= 0 && !win[u]) : (win[u] && prev[u] != -1);
          const unsigned long long m = __builtin_amdgcn_ballot_w64(add);
          if (m == 0ull) continue;                     // wave-uniform
          int base = 0;
          if (lane == 0) base = atomicAdd(&s_new, __popcll(m));
          base = __builtin_amdgcn_readfirstlane(base);
          if (add) list[base + __popcll(m & lt)] = k == 0 ? j[u] : prev[u];
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) w.unass_cnt[b] = s_new;
    }
    __threadfence_block();
    __syncthreads();
  } else {
  if (do_assign) {
    const int U = w.unass_cnt[b];
    const int* list = w.unass_idx + off;
    // GetMax
    for (int i0 = 0; i0 < U; i0 += 1024 * kUpR) {
      int j[kUpR], t[kUpR];
      float bi[kUpR], mi[kUpR];
#pragma unroll
      for (int u = 0; u < kUpR; ++u) {
        const int i = i0 + u * 1024 + (int)threadIdx.x;
        j[u] = i < U ? list[i] : -1;
      }
#pragma unroll
      for (int u = 0; u < kUpR; ++u) {
        t[u] = j[u] >= 0 ? bid[j[u]] : 0;
        bi[u] = j[u] >= 0 ? bid_inc[j[u]] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < kUpR; ++u) mi[u] = j[u] >= 0 ? max_inc[t[u]] : 0.0f;     // written by the bid kernel
#pragma unroll
      for (int u = 0; u < kUpR; ++u) {
        if (j[u] >= 0 && (double)bi[u] - 1e-6 <= (double)mi[u] && (double)mi[u] <= (double)bi[u] + 1e-6)
          atomicMax(&max_idx[t[u]], j[u]);
      }
    }
    __threadfence_block();
    __syncthreads();
    // Assign
    for (int i0 = 0; i0 < U; i0 += 1024 * kUpR) {
      int j[kUpR], t[kUpR], prev[kUpR];
      float bi[kUpR];
      bool win[kUpR];
#pragma unroll
      for (int u = 0; u < kUpR; ++u) {
        const int i = i0 + u * 1024 + (int)threadIdx.x;
        j[u] = i < U ? list[i] : -1;
      }
#pragma unroll
      for (int u = 0; u < kUpR; ++u) {
        t[u] = j[u] >= 0 ? bid[j[u]] : 0;
        bi[u] = j[u] >= 0 ? bid_inc[j[u]] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < kUpR; ++u) win[u] = j[u] >= 0 && (last || ld_coherent(&max_idx[t[u]]) == j[u]);
      // a target has one winner, so the winners of a pass touch disjoint targets and disjoint previous owners
#pragma unroll
      for (int u = 0; u < kUpR; ++u) prev[u] = (win[u] && !last) ? ass_inv[t[u]] : -1;
#pragma unroll
      for (int u = 0; u < kUpR; ++u) {
        if (win[u]) {
          if (!last) {
            if (prev[u] != -1) __hip_atomic_store(&ass[prev[u]], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ass_inv[t[u]] = j[u];
            price[t[u]] += bi[u];
            max_inc[t[u]] = -1e9f;
            max_idx[t[u]] = -1;
          } else {
            // forced assignment of every remaining bidder: several bidders may share a
            // target, so the price update must be atomic to stay well defined
            ass_inv[t[u]] = j[u];
            atomicAdd(&price[t[u]], bi[u]);
            max_inc[t[u]] = -1e9f;
          }
          __hip_atomic_store(&ass[j[u]], t[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    __threadfence_block();
    __syncthreads();
  }

  if (do_compact) {
    // ascending list of unassigned points: a thread owns n/1024 (<= 64 per segment) consecutive points, keeps their
    // flags as a bit mask, and one block-wide exclusive scan of the per-thread counts places its entries
    int base = 0;
    for (int seg = 0; seg < n; seg += 65536) {
      const int per = min(n - seg, 65536) >> 10;        // n % 1024 == 0
      const int j0 = seg + (int)threadIdx.x * per;
      unsigned long long mask = 0ull;
      for (int u0 = 0; u0 < per; u0 += kUpR) {
        int a[kUpR];
#pragma unroll
        for (int u = 0; u < kUpR; ++u) a[u] = (u0 + u < per) ? ld_coherent(&ass[j0 + u0 + u]) : 0;
#pragma unroll
        for (int u = 0; u < kUpR; ++u)
          if (a[u] == -1) mask |= 1ull << (u0 + u);
      }
      const int cnt = __popcll(mask);
      int v = cnt;                                      // inclusive scan: wave shuffle + per-wave totals
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      for (int d = 1; d < 64; d <<= 1) {
        const int tt = __shfl_up(v, d, 64);
        if (lane >= d) v += tt;
      }
      __syncthreads();                                  // s_scan of the previous segment has been consumed
      if (lane == 63) s_scan[wave] = v;
      __syncthreads();
      int wave_off = 0, total = 0;
      for (int q = 0; q < (int)(blockDim.x >> 6); ++q) {
        const int c = s_scan[q];
        if (q < wave) wave_off += c;
        total += c;
      }
      int pos = base + wave_off + v - cnt;
      while (mask) {
        const int u = __builtin_ctzll(mask);
        mask &= mask - 1;
        w.unass_idx[off + pos++] = j0 + u;
      }
      base += total;
    }
    if (threadIdx.x == 0) w.unass_cnt[b] = base;
  }
  }   // (slow path)

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

// GetMax + Assign + the next unassigned list of one auction iteration on G workgroups per batch (grid = (G, B)): the
// single-workgroup kernel above is pure latency at a few hundred entries, but a collapsed cloud (the completion network's
// output early in training) keeps 6 000-8 000 bidders to the last iteration and the first iterations of any cloud have as many:
// 44-104 us per launch on one workgroup.  Here a thread owns at most kMultiR list entries, the G workgroups of a batch meet
// once — after GetMax's atomic maxima, before Assign reads them (arrival counter in the workspace, one spinning lane per
// workgroup; every workgroup of the grid is resident: B * G <= the CU count, checked by the host) — and append to the new list
// with one global atomic per workgroup.  The list's order is again free (see the fast path above).  A batch with at most
// kMultiMin entries is handled by its first workgroup alone, without the meeting.
#ifndef CT_EMD_MULTI_MIN
#define CT_EMD_MULTI_MIN 4096
#endif
constexpr int kMultiR = 2, kSoloR = 4, kMultiMin = CT_EMD_MULTI_MIN;      // entries per thread: shared / alone (kSoloR * 1024 >= kMultiMin)
static_assert(kSoloR * 1024 >= kMultiMin && kSoloR >= kMultiR, "entries per thread");
__global__ void __launch_bounds__(1024)
emd_update_multi_kernel(EmdWs w, int* assignment, int n, int gen) {
  __shared__ int s_new, s_base;
  const int b = blockIdx.y, g = blockIdx.x, G = gridDim.x;
  const size_t off = (size_t)b * n;
  int* ass = assignment + off;
  int* ass_inv = w.ass_inv + off;
  int* bid = w.bid + off;
  int* max_idx = w.max_idx + off;
  float* bid_inc = w.bid_inc + off;
  float* max_inc = w.max_inc + off;
  float* price = w.price + off;
  int* list = w.unass_idx + off;
  int* arrive = w.sync + 3 * b;
  int* done = arrive + 1;
  int* newc = arrive + 2;
  const int U = w.unass_cnt[b];
  const bool solo = U <= kMultiMin;                  // the same for every workgroup of the batch
  if (solo && g != 0) return;
  if (threadIdx.x == 0) s_new = 0;
  int j[kSoloR], t[kSoloR], prev[kSoloR], pos[kSoloR][2];
  float bi[kSoloR], mi[kSoloR];
  bool win[kSoloR];
#pragma unroll
  for (int u = 0; u < kSoloR; ++u) {
    const int i = solo ? u * 1024 + (int)threadIdx.x : (g * kMultiR + u) * 1024 + (int)threadIdx.x;
    j[u] = (i < U && (solo || u < kMultiR)) ? list[i] : -1;
  }
#pragma unroll
  for (int u = 0; u < kSoloR; ++u) {
    t[u] = j[u] >= 0 ? bid[j[u]] : 0;
    bi[u] = j[u] >= 0 ? bid_inc[j[u]] : 0.0f;
  }
#pragma unroll
  for (int u = 0; u < kSoloR; ++u) mi[u] = j[u] >= 0 ? max_inc[t[u]] : 0.0f;       // written by the bid kernel
#pragma unroll
  for (int u = 0; u < kSoloR; ++u) {
    if (j[u] >= 0 && (double)bi[u] - 1e-6 <= (double)mi[u] && (double)mi[u] <= (double)bi[u] + 1e-6)
      atomicMax(&max_idx[t[u]], j[u]);
  }
  // this thread's maxima are performed (device-scope atomics: counted in vmcnt until acknowledged) — no agent-scope fence: on
  // eight XCDs that is an L2 write-back, ~3 us per launch, and nothing but atomics has been written so far
  __builtin_amdgcn_s_waitcnt(0x0070);
  __threadfence_block();
  __syncthreads();                                   // ... and everybody's entries are in registers: the list may be rewritten
  if (!solo) {
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const int want = G * (gen + 1);
      // (bounded: the partners are resident by construction — B * G <= CU count, checked by the host — so this ends within
      // microseconds; should it ever not, the launch traps after ~1e8 polls instead of hanging the stream)
      unsigned polls = 0;
      while (__hip_atomic_load(arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (++polls > 100000000u) __builtin_trap();
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < kSoloR; ++u) win[u] = j[u] >= 0 && ld_coherent(&max_idx[t[u]]) == j[u];
  // a target has one winner, so the winners touch disjoint targets and disjoint previous owners
#pragma unroll
  for (int u = 0; u < kSoloR; ++u) prev[u] = win[u] ? ass_inv[t[u]] : -1;
#pragma unroll
  for (int u = 0; u < kSoloR; ++u) {
    if (win[u]) {
      if (prev[u] != -1) __hip_atomic_store(&ass[prev[u]], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ass_inv[t[u]] = j[u];
      price[t[u]] += bi[u];
      max_inc[t[u]] = -1e9f;
      max_idx[t[u]] = -1;
      __hip_atomic_store(&ass[j[u]], t[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // the next list: an entry that did not win stays, an evicted owner joins; positions inside the workgroup by wave-aggregated
  // LDS appends, the workgroup's run by one global atomic
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
  for (int u = 0; u < kSoloR; ++u) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const bool add = k == 0 ? (j[u] >= 0 && !win[u]) : (win[u] && prev[u] != -1);
      const unsigned long long m = __builtin_amdgcn_ballot_w64(add);
      pos[u][k] = -1;
      if (m == 0ull) continue;                       // wave-uniform
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_new, __popcll(m));
      base = __builtin_amdgcn_readfirstlane(base);
      if (add) pos[u][k] = base + __popcll(m & lt);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) s_base = solo ? 0 : __hip_atomic_fetch_add(newc, s_new, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int base = s_base;
#pragma unroll
  for (int u = 0; u < kSoloR; ++u) {
    if (pos[u][0] >= 0) list[base + pos[u][0]] = j[u];
    if (pos[u][1] >= 0) list[base + pos[u][1]] = prev[u];
  }
  if (solo) {                                        // (the counters advance as if all G workgroups had met)
    if (threadIdx.x == 0) {
      w.unass_cnt[b] = s_new;
      __hip_atomic_fetch_add(arrive, G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(done, G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (threadIdx.x == 0) {
    // the last workgroup to get here publishes the list's length (a workgroup's run was added to newc — a returning atomic,
    // s_base — before it arrives here, so every run is counted when the last arrival reads the sum)
    const int d = __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d == G * (gen + 1) - 1) {
      w.unass_cnt[b] = __hip_atomic_load(newc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(newc, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// a lane's value seen through a DPP permutation of its 16-lane row (every lane has a source: no bound control needed)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// one candidate target against a lane's running (best, second best); strict '>' so that an ascending scan keeps
// the lowest index on ties (emd_cuda.cu:150-157) — branch-free
__device__ __forceinline__ void top2_push(Top2& t, float d, int idx) {
  // better <= best always, so the new second best is the median of (best, better, d): one v_med3_f32
  t.idx = d > t.best ? idx : t.idx;
  t.better = __builtin_amdgcn_fmed3f(t.best, t.better, d);
  t.best = fmaxf(t.best, d);
}

// grid = (blocks per batch, B).  n % 1024 == 0 (checked by ct_emd_fwd), so every tile is full.
// Late iterations have a handful of bidders and are pure latency: the next tile's global loads are in flight
// while the current one is scanned (registers -> the other LDS buffer, one barrier per tile), a bidder gets up
// to all 256 lanes, and the scan is unrolled by 4 so that four ds_read_b128 and four sqrt chains overlap.
//
// KTILE / KDEPTH: targets per LDS tile and tiles in flight to registers.  The first iterations (thousands of bidders: compute
// bound) run 1024-target tiles four deep at two workgroups per CU; from the fourth iteration on (a few hundred to ~2000 bidders:
// every tile step is a barrier + an LDS round trip + a filter refresh, 16 of them per launch at n = 16384) the launch takes
// 4096-target tiles two deep on half as many workgroups: 4 steps instead of 16 (50 iterations at B2 n = 16384: 3.48 -> 3.34 ms
// on uniform clouds, 2.69 -> 2.49 ms on a blob against a sphere shell).  Measured and dropped: the targets packed with their
// prices as one float4 array (one 16-byte load per target instead of four dwords: +10 us per launch), rotated tile orders.
template <int KTILE, int KDEPTH>
__global__ void __launch_bounds__(kBidThreads)
emd_bid_kernel(EmdWs w, const float* __restrict__ xyz1, const float* __restrict__ xyz2, int n, float eps, int u_min, int u_max) {
  constexpr int kTile = KTILE, kDepth = KDEPTH;
  extern __shared__ __align__(16) float4 tile_lds[];            // [2][kTile]
  float4 (*tile)[kTile] = (float4 (*)[kTile])tile_lds;
  __shared__ float s_best[kBidThreads / 64], s_better[kBidThreads / 64];
  __shared__ int s_idx[kBidThreads / 64];
  constexpr int kPer = kTile / kBidThreads;                     // tile elements staged per thread
  const int b = blockIdx.y;
  const size_t off = (size_t)b * n;
  const int U = w.unass_cnt[b];
  if (U == 0 || U < u_min || U > u_max) return;                 // (the other tile size takes this batch: see ct_emd_fwd)
  const int nblk = gridDim.x;
  const int per_blk = (U + nblk - 1) / nblk;                    // bidders of this workgroup
  const int first = blockIdx.x * per_blk;
  const int mine = max(0, min(per_blk, U - first));
  if (mine == 0) return;                                        // block-uniform
  // Many bidders per workgroup (>= kPairFrom: the first iterations, and every iteration of a collapsed cloud — 6 000-8 000
  // bidders to the end): a lane group takes TWO bidders, reads each target once for both and does the arithmetic of the pair
  // test on packed fp32 (v_pk_add / v_pk_mul / v_pk_fma: the same IEEE operations, two lanes wide) — the dense regime is bound
  // by one ds_read_b128 and ~9 vector instructions per (bidder, target) pair; this halves the first and nearly the second.
#ifndef CT_EMD_PAIR_FROM
#define CT_EMD_PAIR_FROM 16
#endif
  constexpr int kPairFrom = CT_EMD_PAIR_FROM;
  const bool two = KTILE == 1024 && per_blk >= kPairFrom;       // block-uniform
  const int groups = two ? (per_blk + 1) / 2 : per_blk;         // lane groups of this workgroup
  int T = kBidThreads / groups;                                 // lanes per bidder (pair)
  T = T < 1 ? 1 : (T > 256 ? 256 : T);                         // (a tile holds 1024 targets: 4 per lane and step at most)
  T = 1 << (31 - __clz(T));                                     // power of two: a bidder is a lane group of a wave, or whole waves
  // nblk >= n/128 >= U/128  =>  per_blk <= 128  =>  T >= 4
  const int slot = threadIdx.x / T, sub = threadIdx.x % T;
  const bool active = two ? 2 * slot < mine : slot < mine;
  int j = -1, jb = -1;                                          // (jb: the pair's second bidder, -1 when the count is odd)
  float x1 = 0, y1 = 0, z1 = 0, xb = 0, yb = 0, zb = 0;
  if (active) {
    j = w.unass_idx[off + first + (two ? 2 * slot : slot)];
    x1 = xyz1[(off + j) * 3 + 0];
    y1 = xyz1[(off + j) * 3 + 1];
    z1 = xyz1[(off + j) * 3 + 2];
    xb = x1; yb = y1; zb = z1;                                  // (an odd pair scans its first bidder twice; nothing is written for it)
    if (two && 2 * slot + 1 < mine) {
      jb = w.unass_idx[off + first + 2 * slot + 1];
      xb = xyz1[(off + jb) * 3 + 0];
      yb = xyz1[(off + jb) * 3 + 1];
      zb = xyz1[(off + jb) * 3 + 2];
    }
  }
  // software pipeline: tile t is scanned from LDS while tiles t+2 .. t+1+kDepth are in flight to registers
  float4 stage[kDepth][kPer];
  auto fetch = [&](int t, float4 (&st)[kPer]) {
#pragma unroll
    for (int e = 0; e < kPer; ++e) {
      const size_t k = off + (size_t)t * kTile + e * kBidThreads + threadIdx.x;
      const float* p = xyz2 + k * 3;
      st[e] = make_float4(p[0], p[1], p[2], w.price[k]);
    }
  };
  auto commit = [&](int buf, const float4 (&st)[kPer]) {
#pragma unroll
    for (int e = 0; e < kPer; ++e) tile[buf][e * kBidThreads + threadIdx.x] = st[e];
  };
  const int ntiles = n / kTile;
#pragma unroll
  for (int r = 0; r < kDepth; ++r)
    if (r < ntiles) fetch(r, stage[r]);
  commit(0, stage[0]);
  if (kDepth < ntiles) fetch(kDepth, stage[0]);
  __syncthreads();
  Top2 t2 = {-1e9f, -1e9f, 0x7fffffff}, tb = {-1e9f, -1e9f, 0x7fffffff};
  // Candidate filter.  Only the bidder's two largest values matter, so a candidate below the second-best value the
  // bidder's lanes (of this wave) have seen so far can be dropped before its sqrt and its double-precision tail (2/3 of the
  // work): with thr that value, value = 3 - sqrt(d2) - price < thr  <=>  sqrt(d2) > 3 - thr - price, tested on the squared
  // distance as d2 > tt*|tt|, tt = (3 - thr + 1e-5) - price (the margin covers the roundings of the test; tt <= 0 means the
  // price alone rules the target out).  thr only uses earlier, lower-indexed targets, so a tie with it can never win the
  // lowest-index rule either: the result is bit-identical.  The slow path runs when ANY lane of the wave has a candidate
  // (wave-uniform branch); thr is refreshed from the lanes' running pairs after such a batch.
  float cthr = 2e9f, cthb = 2e9f;    // 3 - thr + margin; thr = -1e9 at the start: nothing is dropped
  const int Tw = T < 64 ? T : 64;    // lanes of this wave that work for the same bidder
  typedef float f2 __attribute__((ext_vector_type(2)));
  // second-best value over the bidder's lanes in this wave -> the filter's threshold (see the single-bidder scan below)
  auto refresh = [&](const Top2& tt2) -> float {
    float gb = tt2.best, g2 = tt2.better;
    auto fold = [&](float ob, float o2) {
      g2 = fmaxf(fminf(gb, ob), fmaxf(g2, o2));
      gb = fmaxf(gb, ob);
    };
    if (Tw >= 2) fold(dpp_f32<0xB1>(gb), dpp_f32<0xB1>(g2));
    if (Tw >= 4) fold(dpp_f32<0x4E>(gb), dpp_f32<0x4E>(g2));
    if (Tw >= 8) fold(dpp_f32<0x141>(gb), dpp_f32<0x141>(g2));
    if (Tw >= 16) fold(dpp_f32<0x140>(gb), dpp_f32<0x140>(g2));
    if (Tw >= 32) fold(__shfl_xor(gb, 16, 64), __shfl_xor(g2, 16, 64));
    if (Tw >= 64) fold(__shfl_xor(gb, 32, 64), __shfl_xor(g2, 32, 64));
    return fabsf(g2) <= 32.0f ? 3.0f - g2 + 4e-5f : __builtin_inff();
  };
  for (int t0 = 0; t0 < ntiles; t0 += kDepth) {
#pragma unroll
    for (int r = 0; r < kDepth; ++r) {
      const int t = t0 + r;
      if (t < ntiles) {                                        // block-uniform
        if (active && two) {
          // two bidders per lane group: one LDS read per target, the pair test on packed fp32 (the operations and their
          // order per bidder are those of the single-bidder scan: bit-identical values), ONE slow-path decision for both
          const float4* tl = tile[t & 1];
          const int k0 = t * kTile;
          const f2 X = {x1, xb}, Y = {y1, yb}, Z = {z1, zb};
          for (int k = sub; k < kTile; k += 4 * T) {
            float4 q[4];
            f2 d2[4];
            unsigned long long keepa = 0ull, keepb = 0ull;
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = tl[k + u * T];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const f2 qx = {q[u].x, q[u].x}, qy = {q[u].y, q[u].y}, qz = {q[u].z, q[u].z}, qw = {q[u].w, q[u].w};
              const f2 x2 = qx - X, y2 = qy - Y, z2 = qz - Z;
              d2[u] = __builtin_elementwise_fma(z2, z2, __builtin_elementwise_fma(y2, y2, x2 * x2));
              const f2 cth = {cthr, cthb};
              const f2 tt = cth - qw;
              keepa |= __builtin_amdgcn_ballot_w64(!(d2[u].x > tt.x * fabsf(tt.x)));
              keepb |= __builtin_amdgcn_ballot_w64(!(d2[u].y > tt.y * fabsf(tt.y)));
            }
            // the slow path per bidder of the pair (wave-uniform each): a candidate of one does not cost the other's chains
            if (keepa != 0ull) {
#pragma unroll
              for (int u = 0; u < 4; ++u) top2_push(t2, (float)(3.0 - (double)sqrtf(d2[u].x) - (double)q[u].w), k0 + k + u * T);
              cthr = refresh(t2);
            }
            if (keepb != 0ull) {
#pragma unroll
              for (int u = 0; u < 4; ++u) top2_push(tb, (float)(3.0 - (double)sqrtf(d2[u].y) - (double)q[u].w), k0 + k + u * T);
              cthb = refresh(tb);
            }
          }
        } else if (active) {
          const float4* tl = tile[t & 1];
          const int k0 = t * kTile;
          for (int k = sub; k < kTile; k += 4 * T) {           // kTile / T is a multiple of 4 (T <= 256)
            float4 q[4];
            float d2[4];
            unsigned long long keep = 0ull;                    // lanes with a candidate, as compare masks (one v_cmp + s_or each:
                                                               // per-lane booleans made hipcc build them with 16-bit shifts and ORs)
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = tl[k + u * T];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const float x2 = q[u].x - x1, y2 = q[u].y - y1, z2 = q[u].z - z1;
              d2[u] = fmaf(z2, z2, fmaf(y2, y2, x2 * x2));
              const float tt = cthr - q[u].w;
              keep |= __builtin_amdgcn_ballot_w64(!(d2[u] > tt * fabsf(tt)));
            }
            if (keep != 0ull) {                                // wave-uniform
              float d[4];
#pragma unroll
              for (int u = 0; u < 4; ++u)      // evaluated in double like the reference (its literal 3.0 is a double), rounded once
                d[u] = (float)(3.0 - (double)sqrtf(d2[u]) - (double)q[u].w);
#pragma unroll
              for (int u = 0; u < 4; ++u) top2_push(t2, d[u], k0 + k + u * T);
              // second-best value over the bidder's lanes in this wave (values only; the lanes keep their own pairs)
              // (the first four steps — 16 lanes — over DPP: quad permutes, then the 8- and 16-lane mirrors; a step over LDS
              //  (__shfl_xor = ds_bpermute) is a dependent ~100-cycle round trip, and this refresh runs in every slow-path step)
              float gb = t2.best, g2 = t2.better;
              auto fold = [&](float ob, float o2) {
                g2 = fmaxf(fminf(gb, ob), fmaxf(g2, o2));
                gb = fmaxf(gb, ob);
              };
              if (Tw >= 2) fold(dpp_f32<0xB1>(gb), dpp_f32<0xB1>(g2));          // quad_perm [1,0,3,2]
              if (Tw >= 4) fold(dpp_f32<0x4E>(gb), dpp_f32<0x4E>(g2));          // quad_perm [2,3,0,1]
              if (Tw >= 8) fold(dpp_f32<0x141>(gb), dpp_f32<0x141>(g2));        // row_half_mirror
              if (Tw >= 16) fold(dpp_f32<0x140>(gb), dpp_f32<0x140>(g2));       // row_mirror
              if (Tw >= 32) fold(__shfl_xor(gb, 16, 64), __shfl_xor(g2, 16, 64));
              if (Tw >= 64) fold(__shfl_xor(gb, 32, 64), __shfl_xor(g2, 32, 64));
              // safety margin of the filter: a candidate whose exact value reaches g2 must survive the float test.  The
              // roundings of 3 - g2, of + margin, of - price and of tt^2 add up to < 1.6e-5 while |g2| <= 32 (values of
              // interest then have |price| <= 37: ulp(64) = 7.6e-6); beyond that (large eps, late iterations: prices in
              // the hundreds) an absolute margin sinks below one ulp, so the filter is switched off (cthr = inf keeps all)
              cthr = fabsf(g2) <= 32.0f ? 3.0f - g2 + 4e-5f : __builtin_inff();
            }
          }
        }
        if (t + 1 < ntiles) {
          // the other buffer was last read before the previous barrier; its register slot is then free again
          commit((t + 1) & 1, stage[(r + 1) % kDepth]);
          if (t + 1 + kDepth < ntiles) fetch(t + 1 + kDepth, stage[(r + 1) % kDepth]);
        }
        __syncthreads();
      }
    }
  }
  // butterfly over the lanes of a bidder inside its wave (inactive lanes hold the identity)
  for (int m = 1; m < Tw; m <<= 1) {
    Top2 o;
    o.best = __shfl_xor(t2.best, m, 64);
    o.better = __shfl_xor(t2.better, m, 64);
    o.idx = __shfl_xor(t2.idx, m, 64);
    t2 = merge_top2(t2, o);
  }
  if (two) {                         // block-uniform; a pair's lanes are part of one wave (groups >= 8: T <= 64)
    for (int m = 1; m < Tw; m <<= 1) {
      Top2 o;
      o.best = __shfl_xor(tb.best, m, 64);
      o.better = __shfl_xor(tb.better, m, 64);
      o.idx = __shfl_xor(tb.idx, m, 64);
      tb = merge_top2(tb, o);
    }
    if (active && sub == 0 && jb >= 0) {
      const float inc = tb.best - tb.better + eps;
      w.bid[off + jb] = tb.idx;
      w.bid_inc[off + jb] = inc;
      atomicMax((int*)&w.max_inc[off + tb.idx], __float_as_int(inc));
    }
  }
  if (T > 64) {                      // block-uniform: a bidder spans T/64 whole waves
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_best[wave] = t2.best; s_better[wave] = t2.better; s_idx[wave] = t2.idx; }
    __syncthreads();
    if (sub == 0) {
      for (int q = 1; q < T / 64; ++q) {
        Top2 o = {s_best[wave + q], s_better[wave + q], s_idx[wave + q]};
        t2 = merge_top2(t2, o);
      }
    }
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

int emd_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            ? prop.multiProcessorCount : 256;
  }
  return n;
}
int g_emd_single_update = -1;         // test hook (ct_debug_set_emd; env CLOUDCT_EMD_SINGLE_UPDATE read once): the update on one workgroup per batch

}  // namespace

extern "C" {

size_t ct_emd_workspace_bytes(int B, int n) {
  if (B <= 0 || n <= 0) return 0;
  return 7 * align256((size_t)B * n * 4) + align256((size_t)B * 4) + align256((size_t)B * 12);
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
  w.unass_cnt = (int*)p; p += align256((size_t)B * 4);
  w.sync = (int*)p;
  const size_t total = (size_t)B * n;
  CT_CLEAR_ERROR();
  hipLaunchKernelGGL(emd_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, (int*)assignment, total, 3 * B);
  hipLaunchKernelGGL(emd_update_kernel, dim3(B), dim3(1024), 0, st, w, (int*)assignment, xyz1, xyz2, dist, n, 0, 0, 1, 0);
  const dim3 bid_grid(n / 64, B);
  constexpr int kBigTile = 4096;
  const bool big_ok = n % kBigTile == 0 && n >= 2 * kBigTile &&
                      hipFuncSetAttribute((const void*)emd_bid_kernel<kBigTile, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          2 * kBigTile * (int)sizeof(float4)) == hipSuccess;
  // the per-batch update on several workgroups (emd_update_multi_kernel): every list entry needs a thread's register slot and
  // every workgroup of the grid must be resident for their meeting
  int multi_g = (n + 1024 * kMultiR - 1) / (1024 * kMultiR), multi_gen = 0;
  if (g_emd_single_update < 0) {
    const char* e = getenv("CLOUDCT_EMD_SINGLE_UPDATE");
    g_emd_single_update = (e && atoi(e) != 0) ? 1 : 0;
  }
  if (multi_g > 16 || (long long)B * multi_g > emd_cu_count() || g_emd_single_update) multi_g = 1;
  for (int it = 0; it < iters; ++it) {
    const int last = it == iters - 1;
    // The host never learns the number of bidders, so from the sixth iteration on BOTH variants are launched and each batch
    // picks its own on the device: few bidders -> the 4096-target tiles, many -> the 1024-target ones (the other launch returns
    // at once, ~2 us).  Uniform clouds are down to ~2000 bidders by then (3.48 -> 3.3x ms per 50 iterations with the big
    // tiles); a collapsed cloud — the completion network's output early in training — keeps 6000-8000 bidding to the end,
    // where the big tiles on half the workgroups cost 9.87 vs 8.78 ms.
    if (big_ok && it >= kEmdBigFrom) {
      hipLaunchKernelGGL((emd_bid_kernel<kBigTile, 2>), dim3(n / 128, B), dim3(kBidThreads), 2 * kBigTile * sizeof(float4), st, w, xyz1, xyz2, n, eps,
                         0, kEmdBigMaxU);
      hipLaunchKernelGGL((emd_bid_kernel<kTile, 4>), bid_grid, dim3(kBidThreads), 2 * kTile * sizeof(float4), st, w, xyz1, xyz2, n, eps,
                         kEmdBigMaxU + 1, 0x7fffffff);
    } else {
      hipLaunchKernelGGL((emd_bid_kernel<kTile, 4>), bid_grid, dim3(kBidThreads), 2 * kTile * sizeof(float4), st, w, xyz1, xyz2, n, eps,
                         0, 0x7fffffff);
    }
    if (multi_g > 1 && !last)
      hipLaunchKernelGGL(emd_update_multi_kernel, dim3(multi_g, B), dim3(1024), 0, st, w, (int*)assignment, n, multi_gen++);
    else
      hipLaunchKernelGGL(emd_update_kernel, dim3(B), dim3(1024), 0, st, w, (int*)assignment, xyz1, xyz2, dist, n,
                         1, last, last ? 0 : 1, last);
  }
  CT_CHECK_LAUNCH();
  return CT_OK;
}

void ct_debug_set_emd(unsigned flags) { g_emd_single_update = (flags & 1u) ? 1 : 0; }

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
