// Raster backward passes on a plane's points SORTED BY BASE CELL (2D grids, corners from keys, N <= 4096, N % 4 == 0).
// Included by ct_raster.hip inside its anonymous namespace, after ct_raster_hot.h (uses its helpers).
//
// The scatter form of Slice backward (ct_raster_hot.h: slice_bwd_fused_kernel) issues, per point and four channels, 8 random
// 64-bit LDS atomics and 4 random 16-byte LDS reads: 0.65 of its LDS cycles are bank conflicts, and ~100 vector instructions per
// point and group go into rounding, pair packing and a corner set-up recomputed per group (0.46 of the roofline).
// Here the plane's points are counting-sorted by base cell ONCE per workgroup (deterministic: per-wave histograms, returning
// LDS atomics — a function of the keys alone), the sorted list is cut into ITEMS — runs of at most four entries of ONE base cell — and a thread owns at most
// two items for the whole kernel:
//   * per-entry state (the two fractional weights; w0 = 1 - w1 holds bit for bit) and the g_keys sums live in registers;
//   * g_out of a four-channel group is read coalesced in point order and staged into LDS in SORTED order as one 16-byte word
//     per entry, so an item reads its entries' channels with one ds_read_b128 each;
//   * the conv tile is read once per item (lanes = consecutive cells: conflict-free), not once per point;
//   * an item sums its <= 4 products per (corner, channel) in float registers and rounds the SUM once to the channel's
//     fixed-point quantum: 16 integer LDS adds per item — four times fewer than one set per point, issued by lanes on
//     consecutive cells (no bank conflicts but for a cell's own extra items) — integer adds commute and the items are a
//     deterministic function of the keys, so g_grid stays bitwise reproducible;
//   * the next group's g_out rows and conv cells are requested before the current group is processed (registers), so HBM
//     keeps streaming through the LDS phase.
// One 1024-thread workgroup per (b, h) plane and CU (LDS: 16 G + 16 G + 16 (N + 1) + 8 N + 4 G + 8 KiB ~ 140 KiB at 32^2, N 4096).
#pragma once

constexpr int kSortThreads = 1024;
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kItemLen = 4;
constexpr int kMaxItems = 2 * kSortThreads;

// CT_SORT_STAMPS (experiments): workgroup (0,0,0)'s thread 0 leaves s_memtime at the phase boundaries in g_sorted_stamps
// (read by ct_debug_sorted_stamps; tools/dev/sorted_check.py --stamps)
#ifdef CT_SORT_STAMPS
__device__ unsigned long long g_sorted_stamps[64];
#define CT_WSTAMP(i)                                                                                               \
  do {                                                                                                              \
    if ((threadIdx.x & 63) == 0 && blockIdx.x + blockIdx.y + blockIdx.z == 0) g_sorted_stamps[i] = clock64();        \
  } while (0)
#define CT_STAMP(i)                                                                                      \
  do {                                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x + blockIdx.y + blockIdx.z == 0) g_sorted_stamps[i] = clock64();   \
  } while (0)
// every workgroup's {entry, exit, hardware id, XCC id} (ct_debug_wg_stamps): who ran where, when — concurrency and tails of a launch
__device__ unsigned long long g_wg_stamps[4096][4];
#define CT_WG_STAMP(slot)                                                                                            \
  do {                                                                                                                \
    if (threadIdx.x == 0) {                                                                                           \
      const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                            \
      if (wg_ < 4096u) {                                                                                              \
        g_wg_stamps[wg_][slot] = clock64();                                                                           \
        if ((slot) == 0) {                                                                                            \
          unsigned hw_, xcc_;                                                                                         \
          asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                           \
          asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                         \
          g_wg_stamps[wg_][2] = hw_;                                                                                  \
          g_wg_stamps[wg_][3] = xcc_;                                                                                 \
        }                                                                                                             \
      }                                                                                                               \
    }                                                                                                                 \
  } while (0)
#else
#define CT_STAMP(i) ((void)0)
#define CT_WSTAMP(i) ((void)0)
#define CT_WG_STAMP(slot) ((void)0)
#endif

// inclusive wave64 scans over DPP (the reduction sequence of wave_sum_i32 IS a scan: every lane ends with its prefix)
__device__ __forceinline__ unsigned wave_scan_add_u32(unsigned v) {
#define CT_UADD(a, b) ((a) + (b))
  CT_DPP_STEP(v, CT_UADD, 0x111, 0xf);
  CT_DPP_STEP(v, CT_UADD, 0x112, 0xf);
  CT_DPP_STEP(v, CT_UADD, 0x114, 0xf);
  CT_DPP_STEP(v, CT_UADD, 0x118, 0xf);
  CT_DPP_STEP(v, CT_UADD, 0x142, 0xa);
  CT_DPP_STEP(v, CT_UADD, 0x143, 0xc);
#undef CT_UADD
  return v;
}
__device__ __forceinline__ unsigned wave_scan_max_u32(unsigned v) {
#define CT_UMAX(a, b) max((a), (b))
  CT_DPP_STEP(v, CT_UMAX, 0x111, 0xf);
  CT_DPP_STEP(v, CT_UMAX, 0x112, 0xf);
  CT_DPP_STEP(v, CT_UMAX, 0x114, 0xf);
  CT_DPP_STEP(v, CT_UMAX, 0x118, 0xf);
  CT_DPP_STEP(v, CT_UMAX, 0x142, 0xa);
  CT_DPP_STEP(v, CT_UMAX, 0x143, 0xc);
#undef CT_UMAX
  return v;
}

// LDS carve-up shared by the sorted kernels (bytes from the start of dynamic LDS)
struct SortLds {
  size_t tile;     // float4[G]: the group's conv cells, channel-interleaved (Slice bwd) / first staged tile
  size_t acc;      // int[4][G]: fixed-point accumulators of the group (Slice bwd) / second staged tile
  size_t stage;    // float4[N + 1]: the group's point values in sorted order (+ one zero entry); set-up: per-wave histograms
  size_t ab;       // float2[N + 1]: the entries' fractional weights (w1x, w1y) in sorted order (+ one zero entry)
  size_t tot;      // unsigned[G]: points per base cell
  size_t stp;      // unsigned[G]: {first sorted position | first item << 16} of the cell
  size_t mark;     // unsigned[kMaxItems]: first-item marks of the non-empty cells
  size_t misc;     // unsigned[C] channel maxima | K | scan scratch [64]
  size_t total;
};
__host__ __device__ inline SortLds sort_lds(int G, int N, int C) {
  SortLds L;
  size_t o = 0;
  L.tile = o;  o += (size_t)16 * G;
  L.acc = o;   o += (size_t)16 * G;
  L.stage = o;
  {
    const size_t st = (size_t)16 * (N + 1), hist = (size_t)kSortWaves * G * 4;
    o += ((st > hist ? st : hist) + 15) & ~(size_t)15;
  }
  L.ab = o;    o += (size_t)8 * (N + 1);
  L.tot = o;   o += (size_t)4 * G;
  L.stp = o;   o += (size_t)4 * G;
  L.mark = o;  o += (size_t)4 * kMaxItems;
  L.misc = o;  o += (size_t)4 * (C + 1 + 64 + 3);
  L.total = (o + 15) & ~(size_t)15;
  return L;
}

// What a thread keeps of the sorted plane: the sorted positions of the four points it LOADS (packed 16 bits each), and its
// (at most) two ITEMS: base cell and, per slot, the entry's sorted position (N — the zero entries of the stage and weight
// areas — for a slot beyond the item's entries).  The entries' weights stay in LDS (SortLds::ab): 16 registers less.
constexpr unsigned kRankMask = 0x1fffu, kInsideX = 0x4000u, kInsideY = 0x8000u;
struct SortedPlane {
  unsigned rk01, rk23;            // sorted positions of points 4 tid .. 4 tid + 3, 16 bits each: position | kInsideX | kInsideY
  int cell[2];                    // -1: no item
  unsigned ent8[2][kItemLen / 2]; // 8 x the sorted position of two entries, 16 bits each: the byte offset of the entry's weights
                                  // (its stage word: twice that)
};

// item u of the thread: `n` entries of cell Y from sorted position `first` (valid = false: none)
__device__ __forceinline__ void set_item(SortedPlane& S, int u, bool valid, int Y, int first, int n, int N) {
  S.cell[u] = valid ? Y : -1;
#pragma unroll
  for (int j = 0; j < kItemLen; j += 2) {
    const unsigned e0 = (valid && j < n) ? first + j : N, e1 = (valid && j + 1 < n) ? first + j + 1 : N;
    S.ent8[u][j >> 1] = (e0 << 3) | (e1 << 19);
  }
}

// ---------------------------------------------------------------------------
// The sorted plane as a RECORD in global memory (ct_plane_sort): sorted once per key tensor, read by every raster pass that
// takes those keys — and by SEVERAL workgroups of a plane where there are fewer planes than CUs.  Per (b, h) plane:
//   +0            unsigned nitems, K (max contributions to a cell), 2 x reserved
//   +16           float2 ab[N]            fractional weights (w1x, w1y) in sorted order
//   +16 + 8 N     uint16 rk[N]            sorted position of point p | kInsideX | kInsideY
//   +16 + 10 N    unsigned item[kMaxItems]  first | (n - 1) << 13 | cell << 16
// ---------------------------------------------------------------------------
__host__ __device__ inline size_t sort_record_bytes(int N) { return ((size_t)16 + (size_t)10 * N + (size_t)4 * kMaxItems + 255) & ~(size_t)255; }

// Sorts the plane's points by base cell and deals the items.  All kSortThreads threads call it; the block's LDS must hold the
// SortLds carve-up; K (max contributions to a cell) is left in misc[C].  On return the stage area is free (hist is dead).
struct PlaneKeys {
  int base[4];
  float fa[4], fb[4];      // w1x, w1y
  unsigned inside;         // bits 2i, 2i + 1: key x / y of point i lies inside the clamp range (its cotangent passes)
};
// the thread's four points: base cells and fractional weights (issued FIRST: the sort waits for nothing else)
__device__ __forceinline__ void load_plane_keys(const RasterArgs& a, const GridW<2>& g, int W1, size_t bh, PlaneKeys& K) {
  const int tid = threadIdx.x, N = a.N;
  const bool has = (tid << 2) < N;
  const int n0 = has ? (tid << 2) : 0;
  const float4 tx = *(const float4*)(a.pos.keys + (bh * 2 + 0) * N + n0);
  const float4 ty = *(const float4*)(a.pos.keys + (bh * 2 + 1) * N + n0);
  const float kx[4] = {tx.x, tx.y, tx.z, tx.w}, ky[4] = {ty.x, ty.y, ty.z, ty.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    Pt2 p;
    pt2_from_keys(kx[i], ky[i], g, W1, p);
    K.base[i] = p.base; K.fa[i] = p.w1x; K.fb[i] = p.w1y;
  }
  K.inside = 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    K.inside |= (ct_key_mask(kx[i]) != 0.0f ? 1u : 0u) << (2 * i) | (ct_key_mask(ky[i]) != 0.0f ? 2u : 0u) << (2 * i);
}

// LDS of the sort itself
struct SortPtrs {
  unsigned* hist;     // [kSortWaves][G / 2]: per-wave counts, two cells (16 bits each) per word
  float2* ab;         // [N + 1] weights in sorted order, or null (AB_GLOBAL: they go to the record)
  unsigned* cnt;      // [G]
  unsigned* stp;      // [G]
  unsigned* mark;     // [kMaxItems]
  unsigned* s_k;      // K; channel maxima in the nC words before it
  unsigned* scr;      // [64]
};
__device__ __forceinline__ SortPtrs sort_ptrs(unsigned char* lds, const SortLds& L, int C) {
  SortPtrs P;
  P.hist = (unsigned*)(lds + L.stage);
  P.ab = (float2*)(lds + L.ab);
  P.cnt = (unsigned*)(lds + L.tot);
  P.stp = (unsigned*)(lds + L.stp);
  P.mark = (unsigned*)(lds + L.mark);
  P.s_k = (unsigned*)(lds + L.misc) + C;
  P.scr = P.s_k + 1;
  return P;
}

// record != null: the items, ranks and header are also written there (the weights: by the caller).  P.ab null: the weights are
// not placed (the caller does it, e.g. into the histograms' space once they are dead: on return).  nC: channel-maximum words in front of K that are cleared here.
struct SortNoHook {
  __device__ __forceinline__ void operator()() const {}
};
// after_count: called once the keys have been used (the counting adds are issued) — where a caller puts its first bulk loads,
// so that they do not compete with the keys the whole sort waits for
template <typename F = SortNoHook>
__device__ __forceinline__ void sort_plane(const RasterArgs& a, const PlaneKeys& PK, int G, int W1, const SortPtrs& P, int nC,
                                           SortedPlane& S, unsigned char* record = nullptr, F after_count = F()) {
  const int tid = threadIdx.x, N = a.N, wave = tid >> 6;
  unsigned* const hist = P.hist;
  float2* const AB = P.ab;
  unsigned* const cnt = P.cnt;
  unsigned* const stp = P.stp;
  unsigned* const mark = P.mark;
  unsigned* const s_k = P.s_k;
  unsigned* const scr = P.scr;            // [0..15] wave sums, [16..47] wave maxima of the item marks, [48] items in all
  const int G2 = G >> 1;
  const bool has = (tid << 2) < N;
  CT_STAMP(0);
  // every wave clears ITS OWN histogram (a wave's LDS operations complete in order: no barrier before it counts into it); the
  // marks, maxima and K are first touched behind later barriers
  if ((G2 & 3) == 0) {
    for (int i = (tid & 63); i < (G2 >> 2); i += 64) ((uint4*)(hist + wave * G2))[i] = make_uint4(0u, 0u, 0u, 0u);
  } else {
    for (int i = (tid & 63); i < G2; i += 64) hist[wave * G2 + i] = 0u;
  }
  for (int i = tid; i < kMaxItems; i += kSortThreads) mark[i] = 0u;
  if (tid < nC + 1) (s_k - nC)[tid] = 0u;          // channel maxima, K
  // rank inside (wave, cell): the value the returning add hands back — the points of a thread in index order, the lanes of one
  // instruction in the order the LDS serves them (a fixed function of the addresses: the same on every run)
  CT_STAMP(1);
  unsigned r[4] = {0u, 0u, 0u, 0u};
  if (has) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned sh = (unsigned)(PK.base[i] & 1) << 4;        // a wave holds 256 points: its counts fit 16 bits
      r[i] = (atomicAdd(&hist[wave * G2 + (PK.base[i] >> 1)], 1u << sh) >> sh) & 0xffffu;
    }
  }
  asm volatile("" ::: "memory");
  after_count();
  __syncthreads();
  CT_STAMP(2);
  // per PAIR of cells (one histogram word): the waves' counts -> their exclusive prefix (in place) and the cells' totals; then the
  // exclusive scan over the cells of {points, items} packed into one word (points <= 4096, items <= 2048: no carry between the
  // halves; a pair's two words are scanned as their sum)
  unsigned carry = 0u;
  for (int Y0 = 0; Y0 < G2; Y0 += kSortThreads) {
    const int Wd = Y0 + tid;
    unsigned t = 0u;
    if (Wd < G2) {
#pragma unroll 8
      for (int w = 0; w < kSortWaves; ++w) {
        const unsigned c = hist[w * G2 + Wd];
        hist[w * G2 + Wd] = t;
        t += c;                                        // (both halves at once: a cell holds at most 4096 points)
      }
      *(uint2*)(cnt + 2 * Wd) = make_uint2(t & 0xffffu, t >> 16);
    }
    const unsigned t0 = t & 0xffffu, t1 = t >> 16;
    const unsigned v0 = t0 | (((t0 + kItemLen - 1) / kItemLen) << 16), v1 = t1 | (((t1 + kItemLen - 1) / kItemLen) << 16);
    const unsigned inc = wave_scan_add_u32(v0 + v1);
    if ((tid & 63) == 63) scr[wave] = inc;
    __syncthreads();
    CT_STAMP(3);
    unsigned pre = carry, all = 0u;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) {
      const unsigned sw = scr[w];
      pre += w < wave ? sw : 0u;
      all += sw;
    }
    const unsigned ex0 = pre + inc - (v0 + v1), ex1 = ex0 + v0;
    if (Wd < G2) {
      *(uint2*)(stp + 2 * Wd) = make_uint2(ex0, ex1);
      if (t0) mark[ex0 >> 16] = (unsigned)(2 * Wd) + 1u;      // the cell's first item
      if (t1) mark[ex1 >> 16] = (unsigned)(2 * Wd) + 2u;
    }
    carry += all;
    if (Y0 + kSortThreads >= G2) {      // last round: every cell's count is in place
      // K: contributions per cell = points based at the cell and at its three lower neighbours (cells of the last row /
      // column are never a base, so the wrapped neighbours of column 0 read zeros)
      unsigned kloc = 0u;
      for (int X = tid; X < G; X += kSortThreads) {
        unsigned c = cnt[X];
        if (X >= 1) c += cnt[X - 1];
        if (X >= W1) c += cnt[X - W1];
        if (X >= W1 + 1) c += cnt[X - W1 - 1];
        kloc = max(kloc, c);
      }
      kloc = wave_max_u32(kloc);
      if ((tid & 63) == 0) atomicMax(s_k, kloc);
      if (tid == 0) scr[48] = carry >> 16;
    }
    __syncthreads();                                  // (scr is reused by the next round)
  }
  CT_STAMP(4);
  // points: sorted position = cell start + the earlier waves' points of the cell + rank inside the wave
  unsigned rk[4] = {0u, 0u, 0u, 0u};
  if (has) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned sh = (unsigned)(PK.base[i] & 1) << 4;
      rk[i] = (stp[PK.base[i]] & 0xffffu) + ((hist[wave * G2 + (PK.base[i] >> 1)] >> sh) & 0xffffu) + r[i];
      if (AB != nullptr) AB[rk[i]] = make_float2(PK.fa[i], PK.fb[i]);
    }
  }
  if (AB != nullptr && tid == 0) AB[N] = make_float2(0.0f, 0.0f);
#pragma unroll
  for (int i = 0; i < 4; ++i) rk[i] |= ((PK.inside >> (2 * i)) & 1u ? kInsideX : 0u) | ((PK.inside >> (2 * i)) & 2u ? kInsideY : 0u);
  S.rk01 = rk[0] | (rk[1] << 16);
  S.rk23 = rk[2] | (rk[3] << 16);
  // items: item k belongs to the last cell marked at or before k (inclusive max-scan of the marks)
  unsigned mk[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    mk[u] = wave_scan_max_u32(mark[tid + u * kSortThreads]);
    if ((tid & 63) == 63) scr[16 + u * kSortWaves + wave] = mk[u];
  }
  __syncthreads();          // hist (the stage area) is dead from here on
  CT_STAMP(5);
  const int nitems = (int)scr[48];
  // the waves' maxima ahead of this wave (and, for the second slot, the whole first slot): lanes 0..31 hold one each
  const unsigned wm = (tid & 63) < 2 * kSortWaves ? scr[16 + (tid & 63)] : 0u;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const unsigned m = max(mk[u], wave_max_u32((int)(tid & 63) < u * kSortWaves + wave ? wm : 0u));
    const int k = tid + u * kSortThreads;
    const bool valid = k < nitems;
    const int Y = valid ? (int)m - 1 : 0;
    const unsigned sp = stp[Y];
    const int first = (int)(sp & 0xffffu) + (k - (int)(sp >> 16)) * kItemLen;
    const int end = (int)(sp & 0xffffu) + (int)cnt[Y];
    const int n = min(kItemLen, end - first);
    set_item(S, u, valid, Y, first, n, N);
    if (record != nullptr && valid)
      ((unsigned*)(record + 16 + (size_t)10 * N))[k] = (unsigned)first | (unsigned)(n - 1) << 13 | (unsigned)Y << 16;
  }
  if (record != nullptr) {
    if (tid == 0) {
      ((unsigned*)record)[0] = (unsigned)nitems;
      ((unsigned*)record)[1] = *s_k;
    }
    if (has) *(uint2*)(record + 16 + (size_t)8 * N + (size_t)8 * tid) = make_uint2(S.rk01, S.rk23);
  }
  // (no barrier: the weights written above are read behind the first group's staging barrier)
  CT_STAMP(6);
}

// The record of a plane -> what sort_plane leaves behind: the weights into LDS (visible after the caller's next barrier), ranks
// and items into registers, K into misc[C].  One memory round trip instead of the sort.
__device__ __forceinline__ void load_sorted_plane(const RasterArgs& a, const unsigned char* record, unsigned char* lds,
                                                  const SortLds& L, SortedPlane& S) {
  const int tid = threadIdx.x, N = a.N;
  float4* AB4 = (float4*)(lds + L.ab);
  unsigned* misc = (unsigned*)(lds + L.misc);
  const bool has = (tid << 2) < N;
  const unsigned nitems = ((const unsigned*)record)[0];
  const uint2 rk = has ? *(const uint2*)(record + 16 + (size_t)8 * N + (size_t)8 * tid) : make_uint2(0u, 0u);
  unsigned desc[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const unsigned k = (unsigned)tid + (unsigned)u * kSortThreads;
    desc[u] = k < nitems ? ((const unsigned*)(record + 16 + (size_t)10 * N))[k] : 0xffffffffu;
  }
  for (int i = tid; i < (N >> 1); i += kSortThreads) AB4[i] = ((const float4*)(record + 16))[i];
  if (tid < a.C) misc[tid] = 0u;                                  // channel maxima
  if (tid == 0) {
    misc[a.C] = ((const unsigned*)record)[1];                     // K
    ((float2*)(lds + L.ab))[N] = make_float2(0.0f, 0.0f);
  }
  S.rk01 = rk.x; S.rk23 = rk.y;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const bool valid = desc[u] != 0xffffffffu;
    set_item(S, u, valid, (int)(desc[u] >> 16), (int)(desc[u] & 0x1fffu), (int)((desc[u] >> 13) & 3u) + 1, N);
  }
}

// ---------------------------------------------------------------------------
// KP: the plane sort alone (ct_plane_sort): one workgroup per (b, h) plane writes the plane's record.  grid = (1, H, B)
// ---------------------------------------------------------------------------
// LDS of the sort alone: packed histograms 32 G | cnt 4 G | stp 4 G | marks | K + scratch  (48.5 KiB at 32^2: two workgroups per
// CU, whose latency-bound phases overlap)
__host__ __device__ inline size_t plane_sort_hist_bytes(int G, int N) {
  const size_t h = (size_t)kSortWaves * (G >> 1) * 4, w = (size_t)8 * N;       // (the weights pass through the histograms' space)
  return ((h > w ? h : w) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t plane_sort_lds(int G, int N) { return plane_sort_hist_bytes(G, N) + (size_t)8 * G + (size_t)4 * kMaxItems + 4 * 68; }

template <int WT>
__global__ void __launch_bounds__(kSortThreads, 8) plane_sort_kernel(RasterArgs a, GridW<2> g, unsigned char* records, size_t stride) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int G = WT ? WT * WT : g.G, W1 = WT ? WT : g.W[1];
  const size_t bh = (size_t)blockIdx.z * a.H + blockIdx.y;
  SortPtrs P;
  P.hist = (unsigned*)lds_raw;
  P.ab = nullptr;
  P.cnt = (unsigned*)(lds_raw + plane_sort_hist_bytes(G, a.N));
  P.stp = P.cnt + G;
  P.mark = P.stp + G;
  P.s_k = P.mark + kMaxItems;
  P.scr = P.s_k + 1;
  PlaneKeys PK;
  load_plane_keys(a, g, W1, bh, PK);
  SortedPlane S;
  unsigned char* rec = records + bh * stride;
  sort_plane(a, PK, G, W1, P, 0, S, rec);
  // the weights: into sorted order through the (now dead) histograms, out in 16-byte rows (8-byte stores scattered over the
  // record were measured: the whole kernel 14 -> 24 us)
  float2* ABl = (float2*)P.hist;
  if ((threadIdx.x << 2) < a.N) {
    const unsigned rk[4] = {S.rk01 & kRankMask, (S.rk01 >> 16) & kRankMask, S.rk23 & kRankMask, (S.rk23 >> 16) & kRankMask};
#pragma unroll
    for (int i = 0; i < 4; ++i) ABl[rk[i]] = make_float2(PK.fa[i], PK.fb[i]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (a.N >> 1); i += kSortThreads) ((float4*)(rec + 16))[i] = ((const float4*)ABl)[i];
}

// ---------------------------------------------------------------------------
// KF': Slice backward on the sorted plane.  grid = (1, H, B), kSortThreads threads.
// CT_SORT_ABL (experiments, tools/dev/build_raster_exp.sh): 1 = the sort and the epilogue only, 2 = + staging, barriers and
// write-out (no items).  Results are wrong then: timing only.
// ---------------------------------------------------------------------------
#ifndef CT_SORT_ABL
#define CT_SORT_ABL 0
#endif
// byte offset of slot j's weights (8 x its sorted position); its stage word is at twice that
#define CT_E8(S, u, j) (((j) & 1) ? ((S).ent8[u][(j) >> 1] >> 16) : ((S).ent8[u][(j) >> 1] & 0xffffu))
#ifndef CT_SB
#define CT_SB __builtin_amdgcn_sched_barrier(0)
#endif
// PRESORTED: the plane's record (a.sorted, ct_plane_sort) is loaded instead of sorting here.
// GATHER = false: the scatter-add alone (Splat(sum) forward, ct_slice_bwd_grid): no conv tile, no g_keys.
// NT, CT > 0: the point and channel counts as template constants too (the headline's 4096 x 16 on one workgroup per plane: a.ncg = 1)
template <bool HAS_PAD, int WT, bool PRESORTED, bool GATHER = true, int NT = 0, int CT = 0>
__global__ void __launch_bounds__(kSortThreads) slice_bwd_sorted_kernel(RasterArgs a_arg, GridW<2> g_arg) {
  const GridW<2> g = grid2_of<WT>(g_arg);
  RasterArgs a = a_arg;
  if constexpr (NT > 0) { a.N = NT; a.ncg = 1; }
  if constexpr (CT > 0) a.C = CT;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int G = WT ? WT * WT : g.G, W1 = WT ? WT : g.W[1], N = a.N, C = a.C;
  const SortLds L = sort_lds(G, N, C);
  float4* T4 = (float4*)(lds_raw + L.tile);
  int* acc = (int*)(lds_raw + L.acc);
  float4* Sg = (float4*)(lds_raw + L.stage);
  const float2* AB = (const float2*)(lds_raw + L.ab);
  unsigned* s_max = (unsigned*)(lds_raw + L.misc);
  unsigned* s_k = s_max + C;
  // Few planes (the H16 blocks: B8 x H16 = 128): the plane's channel groups are dealt to a.ncg workgroups (grid.x), every one
  // sorting the plane for itself (or reading its record) and writing its partial g_keys to its slice of the workspace; the
  // plane's last workgroup adds them (arrival tickets) or a sum_parts launch does — as in slice_bwd_fused_kernel
  const WgCoord wg = wg_coord(a.ncg, 1, a.H, a.B);
  const int b = wg.b, cgi = wg.cgi, ncg = a.ncg;
  const size_t bh = (size_t)b * a.H + wg.h;
  const int tid = threadIdx.x;
  const int off[4] = {0, W1, 1, W1 + 1};
  const bool has = (tid << 2) < N;
  const int n0 = has ? (tid << 2) : 0;
  const int ngroups = C >> 2;
  const bool fold_keys = GATHER && a.tickets != nullptr && ncg > 1;

  // the first group's rows are requested before the sort: they land while it runs
  float gq[4][4];           // [channel][point] of the thread's quad
  float cvq[4];             // conv cell `tid` of the group's four channels (cells beyond blockDim: loaded in the loop)
  const float* const src0 = a.src + bh * C * (size_t)N;               // wave-uniform bases, 32-bit lane offsets
  const float* const cnv0 = a.tile_in + bh * C * (size_t)G;
  const unsigned ln0 = (unsigned)n0, lc0 = (unsigned)(tid < G ? tid : 0);
  auto request1 = [&](int grp, int cj) {          // one channel of group grp: the quad's g_out and the thread's conv cell
    const float4 t = ld_stream4(src0 + (size_t)(grp * 4 + cj) * N + ln0);
    gq[cj][0] = t.x; gq[cj][1] = t.y; gq[cj][2] = t.z; gq[cj][3] = t.w;
#if CT_SORT_ABL == 5        // experiment: no conv loads (wrong g_keys): what do the 64 one-dword load instructions per group cost?
    cvq[cj] = 1.0f;
#else
    if constexpr (GATHER) cvq[cj] = ld_stream(cnv0 + (size_t)(grp * 4 + cj) * G + lc0);
#endif
  };
  auto request = [&](int grp) {
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) request1(grp, cj);
  };
#ifdef CT_SORT_STAGGER
  if ((blockIdx.y + blockIdx.z) & 1) {      // experiment: odd planes start late, so that sort and streaming phases of different CUs interleave
    const long long t0 = clock64();
    while (clock64() - t0 < (long long)a.cnt_mask) __builtin_amdgcn_s_sleep(8);
  }
#endif
  PlaneKeys PK;
  if constexpr (!PRESORTED) load_plane_keys(a, g, W1, bh, PK);
  SortedPlane S;
  if constexpr (PRESORTED) load_sorted_plane(a, a.sorted + bh * a.sorted_stride, lds_raw, L, S);
#ifndef CT_SORT_LATE_REQUEST
#define CT_SORT_LATE_REQUEST 1
#endif
  if (PRESORTED || !CT_SORT_LATE_REQUEST) request(cgi);
  float pv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pv[i] = (HAS_PAD && has) ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * N + n0 + i) : 1.0f;

  // (the first group's rows are requested behind the keys' use: 28 MB asked for at once by all CUs made every plane wait ~2 k
  //  cycles longer for its 32 KiB of keys)
  if constexpr (!PRESORTED) {
    if (CT_SORT_LATE_REQUEST) sort_plane(a, PK, G, W1, sort_ptrs(lds_raw, L, C), C, S, nullptr, [&]() { request(cgi); });
    else sort_plane(a, PK, G, W1, sort_ptrs(lds_raw, L, C), C, S);
  }
  else __syncthreads();            // K (and the cleared channel maxima) for everybody
  for (int i = tid; i < G; i += kSortThreads) ((int4*)acc)[i] = make_int4(0, 0, 0, 0);
  if (tid == 0) Sg[N] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);       // what the slots beyond an item's entries read
  const float Kf = (float)(*s_k);

  float gsx[2][kItemLen], gsy[2][kItemLen];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int j = 0; j < kItemLen; ++j) gsx[u][j] = gsy[u][j] = 0.0f;

#if CT_SORT_ABL != 1
  // one four-channel group; `more`: another group follows — its rows are requested as soon as this one's are staged (a
  // compile-time flag: a conditional request would make the compiler copy, and so wait for, the loaded registers at once)
  auto group = [&](const int grp, auto more) {
    const int ch0 = grp * 4;
    if (grp == 1) CT_STAMP(16);
    // per-channel max |g_out * pad| of the plane (the fixed-point quantum), and the group into LDS in sorted order
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) {
      float m = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = HAS_PAD ? gq[cj][i] * pv[i] : gq[cj][i];
        x = has ? x : 0.0f;
        gq[cj][i] = x;
        x = fabsf(x);
        m = fmaxf(m, (x < __builtin_inff()) ? x : __builtin_inff());     // inf / NaN -> inf
      }
      const unsigned mb = wave_max_u32(__float_as_uint(m));
      if ((tid & 63) == 0) atomicMax(&s_max[ch0 + cj], mb);
    }
    if (grp == 1) CT_STAMP(17);
    if (has) {
      const unsigned rk[4] = {S.rk01 & kRankMask, (S.rk01 >> 16) & kRankMask, S.rk23 & kRankMask, (S.rk23 >> 16) & kRankMask};
#pragma unroll
      for (int i = 0; i < 4; ++i) Sg[rk[i]] = make_float4(gq[0][i], gq[1][i], gq[2][i], gq[3][i]);
    }
    if (GATHER && tid < G) T4[tid] = make_float4(cvq[0], cvq[1], cvq[2], cvq[3]);
    for (int cell = tid + kSortThreads; GATHER && cell < G; cell += kSortThreads) {       // grids of more than blockDim cells
      const float* p = a.tile_in + (bh * C + ch0) * (size_t)G + cell;
      T4[cell] = make_float4(ld_stream(p), ld_stream(p + G), ld_stream(p + 2 * (size_t)G), ld_stream(p + 3 * (size_t)G));
    }
    if (grp == 1) CT_STAMP(18);
    __syncthreads();
    if (grp == 1) CT_STAMP(19);
    if (grp == 1) CT_STAMP(20);
    if (grp == 1) CT_WSTAMP(24 + (threadIdx.x >> 6));

    float iq[4];
    bool any_float = false;
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) {
      float q;
      bool fixed;
      fx_quantum(__uint_as_float(s_max[ch0 + cj]) * Kf, q, iq[cj], fixed);
      if (!fixed) {
        iq[cj] = 0.0f;         // the channel's sums are rounded to 0 here (0 * x is 0 or NaN) and added by the float pass below
        any_float = true;
      }
      iq[cj] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(iq[cj])));      // wave-uniform: a scalar register
    }
#if CT_SORT_ABL != 2
    // One loop per item: an entry's channels and weights are read once and feed both sides; the next entry's reads are issued
    // before this entry's arithmetic, the conv corners before anything else.  The NEXT group's eight global loads are issued
    // one channel per entry of the first item: a CU's vector-memory path takes ~3 k cycles to issue the 80 KiB of a group, and
    // issued as one block behind the barrier they held the last waves' items back by that long (profiles/r5_sorted_stamps.txt).
    auto item = [&](auto U) {
      constexpr int u = decltype(U)::value;
      constexpr bool spread = u == 0 && decltype(more)::value;
      const int Y = S.cell[u] < 0 ? 0 : S.cell[u];       // (a lane without an item reads the zero entries and adds nothing)
      // (opaque per group: the compiler would otherwise unpack the eight offsets once, outside the group loop — eight more
      //  live registers, spilled)
      asm volatile("" : "+v"(S.ent8[u][0]), "+v"(S.ent8[u][1]));
      float4 cv[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) cv[v] = GATHER ? T4[Y + off[v]] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      float4 xn = *(const float4*)((const unsigned char*)Sg + 2u * CT_E8(S, u, 0));
      float2 wn = *(const float2*)((const unsigned char*)AB + CT_E8(S, u, 0));
      ct_f2 s01[4], s23[4];      // [corner] x channels (0,1) / (2,3)
#pragma unroll
      for (int v = 0; v < 4; ++v) s01[v] = s23[v] = ct_f2{0.0f, 0.0f};
#pragma unroll
      for (int j = 0; j < kItemLen; ++j) {
#ifndef CT_SORT_SPREAD
#define CT_SORT_SPREAD 1      // channels of the next group requested per entry of the first item (2: two per entry, over the first two)
#endif
        if constexpr (spread) {
          if (CT_SORT_SPREAD == 1) request1(grp + ncg, j);
          else if (j < 2) { request1(grp + ncg, 2 * j); request1(grp + ncg, 2 * j + 1); }
        }
        const float4 x = xn;
        const float2 wf = wn;
        if (j + 1 < kItemLen) {
          xn = *(const float4*)((const unsigned char*)Sg + 2u * CT_E8(S, u, j + 1));
          wn = *(const float2*)((const unsigned char*)AB + CT_E8(S, u, j + 1));
        }
        const ct_f2 x01 = {x.x, x.y}, x23 = {x.z, x.w};
        const float w1x = wf.x, w1y = wf.y, w0x = 1.0f - w1x, w0y = 1.0f - w1y;
        const float cw[4] = {w0x * w0y, w1x * w0y, w0x * w1y, w1x * w1y};
        float gw[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const ct_f2 cwv = {cw[v], cw[v]};
          s01[v] = __builtin_elementwise_fma(x01, cwv, s01[v]);
          s23[v] = __builtin_elementwise_fma(x23, cwv, s23[v]);
          if constexpr (GATHER) {
            // (c0 x0 + c2 x2) + (c1 x1 + c3 x3): two packed instructions and one add per corner
            const ct_f2 c01 = {cv[v].x, cv[v].y}, c23 = {cv[v].z, cv[v].w};
            const ct_f2 pr = __builtin_elementwise_fma(c23, x23, c01 * x01);
            gw[v] = pr.x + pr.y;
          }
        }
        if constexpr (GATHER) {
          gsx[u][j] = __builtin_fmaf(gw[3] - gw[2], w1y, __builtin_fmaf(gw[1] - gw[0], w0y, gsx[u][j]));
          gsy[u][j] = __builtin_fmaf(gw[3] - gw[1], w1x, __builtin_fmaf(gw[2] - gw[0], w0x, gsy[u][j]));
          asm volatile("" : "+v"(gsx[u][j]), "+v"(gsy[u][j]));
        }
        CT_SB;
      }
      if (S.cell[u] >= 0) {
        const ct_f2 iq01 = {iq[0], iq[1]}, iq23 = {iq[2], iq[3]};
        int* Tc = acc + Y;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const ct_f2 a01 = s01[v] * iq01, a23 = s23[v] * iq23;
          atomicAdd(Tc + off[v], cvt_rpi(a01.x));
          atomicAdd(Tc + G + off[v], cvt_rpi(a01.y));
          atomicAdd(Tc + 2 * G + off[v], cvt_rpi(a23.x));
          atomicAdd(Tc + 3 * G + off[v], cvt_rpi(a23.y));
        }
      }
      CT_SB;
    };
    item(std::integral_constant<int, 0>{});              // always: it carries the next group's loads
    if (S.cell[1] >= 0) item(std::integral_constant<int, 1>{});
#else
    if constexpr (decltype(more)::value) request(grp + ncg);
#endif
    if (grp == 1) CT_STAMP(21);
    if (grp == 1) CT_WSTAMP(40 + (threadIdx.x >> 6));
    if (any_float) {           // block-uniform, rare: IEEE float atomics for a channel with inf / NaN (or beyond the fixed-point bound)
#pragma unroll 1
      for (int cj = 0; cj < 4; ++cj)
        if (iq[cj] == 0.0f) scatter_float_channel<HAS_PAD>(a, g, bh, b, ch0 + cj, (float*)(acc + cj * G), 1, 0);
    }
    __syncthreads();
    if (grp == 1) CT_STAMP(22);
    // the group's g_grid rows out, the accumulators cleared for the next group (the next barrier orders both)
    float* gout = a.tile_out + (bh * C + ch0) * (size_t)G;
    for (int t = tid; t < G; t += kSortThreads) {          // G int4 = 4 channels x G cells
      const int ch = (t << 2) / G;
      float q, iqd;
      bool fixed;
      fx_quantum(__uint_as_float(s_max[ch0 + ch]) * Kf, q, iqd, fixed);
      const int4 rr = ((const int4*)acc)[t];
      float4 o;
      if (fixed) o = make_float4((float)rr.x * q, (float)rr.y * q, (float)rr.z * q, (float)rr.w * q);
      else o = make_float4(__int_as_float(rr.x), __int_as_float(rr.y), __int_as_float(rr.z), __int_as_float(rr.w));
      st_stream4(gout + ((size_t)t << 2), o);
      ((int4*)acc)[t] = make_int4(0, 0, 0, 0);
    }
    if (grp == 1) CT_STAMP(23);
  };
  CT_STAMP(8);
  int grp = cgi;
  for (; grp + ncg < ngroups; grp += ncg) group(grp, std::true_type{});
  group(grp, std::false_type{});
  CT_STAMP(9);
#endif
  if constexpr (!GATHER) return;
  // g_keys: from the item owners (sorted order) back to the point owners through LDS (the stage area is free: the last
  // group's readers are behind the barrier above)
  float2* Gs = (float2*)Sg;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int j = 0; j < kItemLen; ++j)
      *(float2*)((unsigned char*)Gs + CT_E8(S, u, j)) = make_float2(gsx[u][j], gsy[u][j]);       // (slots beyond an item's entries: word N, nobody's)
  __syncthreads();
  if (has) {
    const unsigned rw[4] = {S.rk01 & 0xffffu, S.rk01 >> 16, S.rk23 & 0xffffu, S.rk23 >> 16};
    float2 gk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gk[i] = Gs[rw[i] & kRankMask];
      gk[i].x *= (rw[i] & kInsideX) ? 1.0f : 0.0f;       // torch.clamp passes the cotangent only inside [lo, hi]
      gk[i].y *= (rw[i] & kInsideY) ? 1.0f : 0.0f;
    }
    float* gp = a.g_pos + (size_t)cgi * a.gpos_stride;
    st_part4(gp + (bh * 2 + 0) * N + n0, make_float4(gk[0].x, gk[1].x, gk[2].x, gk[3].x), fold_keys);
    st_part4(gp + (bh * 2 + 1) * N + n0, make_float4(gk[0].y, gk[1].y, gk[2].y, gk[3].y), fold_keys);
  }
  if (fold_keys) {       // kernel-uniform
    if (arrive_last(a.tickets + bh, (unsigned)ncg, nullptr, 0u, s_k + 1) & 1u) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fold_rows(a.g_pos + (bh * 2 + j) * N, a.gpos_stride, ncg, a.fold_gpos + (bh * 2 + j) * N, N >> 2, nullptr);
    }
  }
  CT_STAMP(10);
}
