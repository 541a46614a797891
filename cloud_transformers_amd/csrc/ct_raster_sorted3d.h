// Slice backward (and the scatter-add alone) on a plane's points SORTED BY BASE CELL, 3D grids of up to 1024 cells — the zoo's
// 8^3 C32 volume heads (model_zoo/s3dis/segmenter.py:40-45; semantics: layers/cloud_transform.py:216-221 through
// layers/utils.py:100-155).  Included by ct_raster.hip inside its anonymous namespace, after ct_raster_sorted.h (whose DPP scans
// and rank-word flags it shares) and ct_raster_hot3d.h.
//
// The scatter form (ct_raster_hot3d.h: slice_bwd_fused3_kernel) issues per point and four channels 16 random 64-bit LDS atomics
// and 8 random 16-byte reads and recomputes the point's corner set-up per group: 0.69 of its LDS cycles are bank conflicts and
// 1.9e7 vector instructions per launch keep it at 0.19 of the HBM roofline (profiles/r5_zoo_counters.txt).  A
// small dense 3D grid is where sorting pays most: 4096 points on 8^3 = 512 cells are ~8 points per base cell, so an ITEM — a run
// of at most four sorted entries of ONE base cell — reads its eight conv corners once and adds its eight corner sums once.
//
// Layout of the work (differs from the 2D kernel where 3D forced it):
//   * 512-thread workgroups of at most 2048 points (a point SEGMENT of the plane) and a share of the plane's four-channel groups:
//     two workgroups per CU (<= 80 KiB of LDS each) whose barriers and HBM waits overlap each other — the 2D kernel's one
//     1024-thread workgroup per CU has nothing to run during its own barriers (24 % of its time, r5_sorted_stamps.txt);
//   * a 3D item is TWO 2D items: the z = 0 face (corners 0..3, cells Y + {0, sx, sy, sx + sy}) and the z = 1 face (the cells
//     right behind them: z is the fastest axis).  A face is the 2D item body — four conv corners and 4 x 4 sums in registers —
//     with the entry's values scaled by the face's z weight, so the register footprint is the 2D kernel's, not twice it
//     (HISTORY.md §7's objection to a 3D item form); the entries are read twice (sequential 16-byte LDS reads);
//   * the key cotangent of a face: d/dx and d/dy are the 2D expressions times the face's z weight, d/dz is -+ the face's
//     bilinear value (ct_corner_grad<3> regrouped by face: same terms, summed face by face).
// Everything else is the 2D design: deterministic counting sort (per-wave histograms, returning LDS adds), weights w1 per
// entry in LDS (w0 = 1 - w1 bit for bit), g_out staged in sorted order as one 16-byte word per entry, item sums rounded ONCE to
// the channel's fixed-point quantum (integer adds commute: g_grid is bitwise reproducible), partial g_keys of the channel
// groups and partial g_grid tiles of the segments folded by arrival tickets or by sum_parts launches.
#pragma once

constexpr int kS3Threads = 512;
constexpr int kS3Waves = kS3Threads / 64;
constexpr int kS3MaxItems = 2 * kS3Threads;      // a thread owns at most two items: n / 4 + 3 G / 4 <= kS3MaxItems
constexpr int kS3MaxPoints = 4 * kS3Threads;     // points of a workgroup's segment
constexpr unsigned kInsideZ = 0x2000u;           // beside kInsideX / kInsideY in a rank word (ranks here are <= 2048)
constexpr unsigned kS3NoItem = 0xffffffffu;

// LDS carve-up (bytes).  The sort's tables (points per cell, cell starts, item marks) live where the conv tile and the
// accumulators go afterwards, its histograms where the staged group goes.
struct Sort3Lds {
  size_t tile;     // float4[G]: the group's conv cells, channel-interleaved        | sort: cnt[G], stp[G], mark[kS3MaxItems]
  size_t acc;      // int[4][G]: the group's fixed-point accumulators
  size_t stage;    // float4[n + 1]: the group's point values in sorted order       | sort: per-wave histograms
  size_t ab;       // float2[n + 1]: (w1x, w1y) in sorted order
  size_t cz;       // float[n + 1]:  w1z in sorted order
  size_t misc;     // unsigned[C] channel maxima | K | flag | scan scratch [64]
  size_t total;
};
__host__ __device__ inline Sort3Lds sort3_lds(int G, int n, int C, int dim = 3) {
  Sort3Lds L;
  size_t o = 0;
  const size_t tiles = (size_t)32 * G, tables = (size_t)8 * G + (size_t)4 * kS3MaxItems;
  L.tile = o;
  L.acc = o + (size_t)16 * G;
  o += ((tiles > tables ? tiles : tables) + 15) & ~(size_t)15;
  L.stage = o;
  {
    const size_t st = (size_t)16 * (n + 1), hist = (size_t)kS3Waves * ((G + 1) >> 1) * 4;
    o += ((st > hist ? st : hist) + 15) & ~(size_t)15;
  }
  L.ab = o;  o += (size_t)8 * (n + 2);
  L.cz = o;  o += dim == 3 ? (((size_t)4 * (n + 1) + 15) & ~(size_t)15) : 0;      // (2D: no third weight)
  L.misc = o; o += (size_t)4 * (C + 2 + 64 + 2);
  L.total = (o + 15) & ~(size_t)15;
  return L;
}

// the thread's four points: base cells, fractional weights, clamp masks (bits 3 i .. 3 i + 2: key x / y / z of point i inside)
struct Plane3Keys {
  int base[4];
  float fa[4], fb[4], fc[4];
  unsigned inside;
};
template <int DIM>
__device__ __forceinline__ void load_plane3_keys(const float* keys, const GridW<DIM>& g, size_t bh, int Nr, int so, int n, Plane3Keys& K) {
  const int tid = threadIdx.x;
  const bool has = (tid << 2) < n;
  const int n0 = has ? (tid << 2) : 0;
  float k[DIM][4];
#pragma unroll
  for (int j = 0; j < DIM; ++j) {
    const float4 t = *(const float4*)(keys + (bh * DIM + j) * Nr + so + n0);
    k[j][0] = t.x; k[j][1] = t.y; k[j][2] = t.z; k[j][3] = t.w;
  }
  K.inside = 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float w0, w1[3] = {0.0f, 0.0f, 0.0f};
    int base = 0;
    unsigned in = 0u;
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
      int f;
      ct_axis(k[j][i], g.hw[j], g.W[j], w0, w1[j], f);
      base = base * g.W[j] + f;                          // (axis 0 slowest: ct_corners)
      in |= (ct_key_mask(k[j][i]) != 0.0f ? 1u : 0u) << j;
    }
    K.base[i] = base;
    K.fa[i] = w1[0]; K.fb[i] = w1[1]; K.fc[i] = w1[2];
    K.inside |= in << (3 * i);
  }
}

// What a thread keeps of the sorted segment: the sorted positions of the four points it loads (16 bits each, with the clamp
// masks), and its (at most) two items, one word each: first sorted position | (entries - 1) << 12 | base cell << 14.
struct Sorted3Plane {
  unsigned rk01, rk23;
  unsigned item[2];          // kS3NoItem: none
};

// Counting sort of the workgroup's points by base cell; the items dealt to the threads.  All kS3Threads threads call it.  K (max
// contributions to a cell) is left in *s_k; the nC words in front of it (channel maxima) are cleared.  The tables alias the tile
// and accumulator areas and the histograms the stage area: the caller puts a barrier between this and its first write to them.
template <int DIM>
__device__ __forceinline__ void sort3_plane(const Plane3Keys& PK, int n, int G, int sx, int sy, unsigned char* lds, const Sort3Lds& L,
                                            int nC, Sorted3Plane& S) {
  const int tid = threadIdx.x, wave = tid >> 6;
  unsigned* const hist = (unsigned*)(lds + L.stage);
  unsigned* const cnt = (unsigned*)(lds + L.tile);
  unsigned* const stp = cnt + G;
  unsigned* const mark = stp + G;
  unsigned* const s_k = (unsigned*)(lds + L.misc) + nC;
  unsigned* const scr = s_k + 2;           // [0..7] wave sums, [16..31] wave maxima of the item marks, [48] items in all
  const int G2 = (G + 1) >> 1;
  const bool has = (tid << 2) < n;
  // every wave clears ITS OWN histogram (a wave's LDS operations complete in order: no barrier before it counts into it)
  for (int i = (tid & 63); i < G2; i += 64) hist[wave * G2 + i] = 0u;
  for (int i = tid; i < kS3MaxItems; i += kS3Threads) mark[i] = 0u;
  for (int i = tid; i < nC + 1; i += kS3Threads) (s_k - nC)[i] = 0u;
  // rank inside (wave, cell): what the returning add hands back — a fixed function of the addresses (ct_raster_sorted.h)
  unsigned r[4] = {0u, 0u, 0u, 0u};
  if (has) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned sh = (unsigned)(PK.base[i] & 1) << 4;        // a wave holds 256 points: its counts fit 16 bits
      r[i] = (atomicAdd(&hist[wave * G2 + (PK.base[i] >> 1)], 1u << sh) >> sh) & 0xffffu;
    }
  }
  __syncthreads();
  // per PAIR of cells: the waves' counts -> their exclusive prefix (in place) and the cells' totals; then the exclusive scan over the
  // cells of {points, items} packed into one word (points <= 2048, items <= 1024: no carry between the halves)
  unsigned carry = 0u;
  for (int Y0 = 0; Y0 < G2; Y0 += kS3Threads) {
    const int Wd = Y0 + tid;
    unsigned t = 0u;
    if (Wd < G2) {
#pragma unroll
      for (int w = 0; w < kS3Waves; ++w) {
        const unsigned c = hist[w * G2 + Wd];
        hist[w * G2 + Wd] = t;
        t += c;
      }
      cnt[2 * Wd] = t & 0xffffu;
      if (2 * Wd + 1 < G) cnt[2 * Wd + 1] = t >> 16;
    }
    const unsigned t0 = t & 0xffffu, t1 = t >> 16;
    const unsigned v0 = t0 | (((t0 + kItemLen - 1) / kItemLen) << 16), v1 = t1 | (((t1 + kItemLen - 1) / kItemLen) << 16);
    const unsigned inc = wave_scan_add_u32(v0 + v1);
    if ((tid & 63) == 63) scr[wave] = inc;
    __syncthreads();
    unsigned pre = carry, all = 0u;
#pragma unroll
    for (int w = 0; w < kS3Waves; ++w) {
      const unsigned sw = scr[w];
      pre += w < wave ? sw : 0u;
      all += sw;
    }
    const unsigned ex0 = pre + inc - (v0 + v1), ex1 = ex0 + v0;
    if (Wd < G2) {
      stp[2 * Wd] = ex0;
      if (2 * Wd + 1 < G) stp[2 * Wd + 1] = ex1;
      if (t0) mark[ex0 >> 16] = (unsigned)(2 * Wd) + 1u;      // the cell's first item
      if (t1) mark[ex1 >> 16] = (unsigned)(2 * Wd) + 2u;
    }
    carry += all;
    if (Y0 + kS3Threads >= G2) {      // last round: every cell's count is in place
      // K: contributions per cell = points based at the cell and at its 2^DIM - 1 lower neighbours (a wrapped neighbour index lands
      // on a cell of the last row / column / slice, which is never a base: it reads zero)
      const int off[8] = {0, sx, sy, sx + sy, 1, sx + 1, sy + 1, sx + sy + 1};      // (2D: sy = 1, the first four)
      unsigned kloc = 0u;
      for (int X = tid; X < G; X += kS3Threads) {
        unsigned c = 0u;
#pragma unroll
        for (int v = 0; v < (1 << DIM); ++v)
          if (X >= off[v]) c += cnt[X - off[v]];
        kloc = max(kloc, c);
      }
      kloc = wave_max_u32(kloc);
      if ((tid & 63) == 0) atomicMax(s_k, kloc);
      if (tid == 0) scr[48] = carry >> 16;
    }
    __syncthreads();                                  // (scr is reused by the next round)
  }
  // points: sorted position = cell start + the earlier waves' points of the cell + rank inside the wave
  unsigned rk[4] = {0u, 0u, 0u, 0u};
  if (has) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned sh = (unsigned)(PK.base[i] & 1) << 4;
      rk[i] = (stp[PK.base[i]] & 0xffffu) + ((hist[wave * G2 + (PK.base[i] >> 1)] >> sh) & 0xffffu) + r[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned in = (PK.inside >> (3 * i)) & 7u;
    rk[i] |= (in & 1u ? kInsideX : 0u) | (in & 2u ? kInsideY : 0u) | (in & 4u ? kInsideZ : 0u);
  }
  S.rk01 = rk[0] | (rk[1] << 16);
  S.rk23 = rk[2] | (rk[3] << 16);
  // items: item k belongs to the last cell marked at or before k (inclusive max-scan of the marks)
  unsigned mk[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    mk[u] = wave_scan_max_u32(mark[tid + u * kS3Threads]);
    if ((tid & 63) == 63) scr[16 + u * kS3Waves + wave] = mk[u];
  }
  __syncthreads();
  const int nitems = (int)scr[48];
  const unsigned wm = (tid & 63) < 2 * kS3Waves ? scr[16 + (tid & 63)] : 0u;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const unsigned m = max(mk[u], wave_max_u32((int)(tid & 63) < u * kS3Waves + wave ? wm : 0u));
    const int k = tid + u * kS3Threads;
    const bool valid = k < nitems;
    const int Y = valid ? (int)m - 1 : 0;
    const unsigned sp = stp[Y];
    const int first = (int)(sp & 0xffffu) + (k - (int)(sp >> 16)) * kItemLen;
    const int end = (int)(sp & 0xffffu) + (int)cnt[Y];
    const int ne = min(kItemLen, end - first);
    S.item[u] = valid ? ((unsigned)first | (unsigned)(ne - 1) << 12 | (unsigned)Y << 14) : kS3NoItem;
  }
}

// IEEE float scatter-add of one channel of the segment into its LDS accumulator row (the channel holds inf / NaN or would
// overflow the fixed-point bound): re-reads the channel's src row; rare.
template <bool HAS_PAD>
__device__ __forceinline__ void scatter_float_channel3(const RasterArgs& a, const GridW<3>& g, size_t bh, int b, int ch, float* row_acc,
                                                       int n, int Nr, int so) {
  int off[8];
  corner_offsets3(g, off);
  const float* src = a.src + (bh * a.C + ch) * (size_t)Nr + so;
  for (int qd = threadIdx.x; qd < (n >> 2); qd += blockDim.x) {
    const int nn = qd << 2;
    float k[3][4];
    load_keys3(a.pos.keys, bh, Nr, so + nn, k);
    const float4 tf = *(const float4*)(src + nn);
    const float f[4] = {tf.x, tf.y, tf.z, tf.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Pt3 p;
      pt3_from_keys(k[0][i], k[1][i], k[2][i], g, p);
      const float x = HAS_PAD ? f[i] * ct_load_pad(a.pad, a.pad_dtype, (size_t)b * Nr + so + nn + i) : f[i];
#pragma unroll
      for (int v = 0; v < 8; ++v) atomicAdd(row_acc + p.base + off[v], x * p.cw[v]);
    }
  }
}

#ifndef CT_S2_PREFETCH
#define CT_S2_PREFETCH 1
#endif
#ifndef CT_S3_PREFETCH
#define CT_S3_PREFETCH 0      // 1: the next group's rows requested during this group's first item (20 more live registers: spills)
#endif

// ---------------------------------------------------------------------------
// KF3': Slice backward on the sorted segment.  grid = (ncg, H, B * nseg), kS3Threads threads.
//   a.N: points of a segment (<= kS3MaxPoints), a.Nrow: row length, a.nseg / a.ncg: segments / channel-group workgroups per plane
//   GATHER = false: the scatter-add alone (Splat(sum) forward, ct_slice_bwd_grid): no conv tile, no g_keys.
// ---------------------------------------------------------------------------
// DIM = 2: the same on a small 2D grid (the zoo's 16^2 C16 head at few planes): ONE face, no z weight — the 2D item body itself.
template <int DIM, int WK>
__device__ __forceinline__ GridW<DIM> seg_grid_of(const GridW<DIM>& g) {
  if constexpr (DIM == 3) return grid3_of<WK>(g);
  else return grid2_of<WK>(g);
}
template <bool HAS_PAD, int DIM, int WK, bool GATHER>
__global__ void __launch_bounds__(kS3Threads, 4) slice_bwd_sorted_seg_kernel(RasterArgs a, GridW<DIM> g_arg) {
  const GridW<DIM> g = seg_grid_of<DIM, WK>(g_arg);
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int G = g.G, n = a.N, C = a.C;
  constexpr int kFaces = DIM == 3 ? 2 : 1;
  // the next group's rows requested during this group's first item (20 more live registers): the 2D form with its grid known at
  // compile time has them to spare (126 registers), the 3D form spills 27 with it (54 vs 49 us)
  constexpr bool kPrefetch = CT_S3_PREFETCH != 0 || (DIM == 2 && WK > 0 && CT_S2_PREFETCH != 0);
  // cell offsets of the in-face corners {0, sx, sy, sx + sy}; 3D: the z = 1 face is +1 (z fastest); 2D: x slowest, y fastest
  const int sy = DIM == 3 ? g.W[DIM - 1] : 1, sx = DIM == 3 ? g.W[1] * g.W[DIM - 1] : g.W[1];
  const Sort3Lds L = sort3_lds(G, n, C, DIM);
  float4* T4 = (float4*)(lds_raw + L.tile);
  int* acc = (int*)(lds_raw + L.acc);
  float4* Sg = (float4*)(lds_raw + L.stage);
  float2* AB = (float2*)(lds_raw + L.ab);
  float* CZ = (float*)(lds_raw + L.cz);
  unsigned* s_max = (unsigned*)(lds_raw + L.misc);
  unsigned* s_k = s_max + C;
  const int nsg = a.nseg > 0 ? a.nseg : 1, ncg = a.ncg;
  const WgCoord wg = wg_coord(ncg, nsg, a.H, a.B);
  const int b = wg.b, cgi = wg.cgi, seg = wg.seg;
  const size_t bh = (size_t)b * a.H + wg.h;
  const int Nr = a.Nrow > 0 ? a.Nrow : a.N;
  const int so = seg * n;
  const int tid = threadIdx.x;
  const bool has = (tid << 2) < n;
  const int n0 = has ? (tid << 2) : 0;
  const int ngroups = C >> 2;
  const bool fold_keys = GATHER && a.tickets != nullptr && ncg > 1, fold_grid = a.tickets != nullptr && nsg > 1;
  const int off2[4] = {0, sx, sy, sx + sy};

  float gq[4][4];           // [channel][point] of the thread's quad
  float cvq[4];             // conv cell `tid` of the group's four channels (cells beyond blockDim: loaded at staging)
  const float* const src0 = a.src + bh * C * (size_t)Nr + so;
  const float* const cnv0 = a.tile_in + bh * C * (size_t)G;
  const unsigned lc0 = (unsigned)(tid < G ? tid : 0);
  auto request1 = [&](int grp, int cj) {
    const float4 t = ld_stream4(src0 + (size_t)(grp * 4 + cj) * Nr + n0);
    gq[cj][0] = t.x; gq[cj][1] = t.y; gq[cj][2] = t.z; gq[cj][3] = t.w;
    if constexpr (GATHER) cvq[cj] = ld_stream(cnv0 + (size_t)(grp * 4 + cj) * G + lc0);
  };
  auto request = [&](int grp) {
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) request1(grp, cj);
  };

  CT_STAMP(0);
  CT_WG_STAMP(0);
  Plane3Keys PK;
  load_plane3_keys<DIM>(a.pos.keys, g, bh, Nr, so, n, PK);
  float pv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pv[i] = (HAS_PAD && has) ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * Nr + so + n0 + i) : 1.0f;
  Sorted3Plane S;
  sort3_plane<DIM>(PK, n, G, sx, sy, lds_raw, L, C, S);
  CT_STAMP(1);
  request(cgi);                                   // (behind the keys' use: see slice_bwd_sorted_kernel)
  const float Kf = (float)(*s_k);
  __syncthreads();                                // the sort's tables and histograms are dead: their space is the tiles' and the stage's
  if (has) {
    const unsigned rk[4] = {S.rk01 & kRankMask, (S.rk01 >> 16) & kRankMask, S.rk23 & kRankMask, (S.rk23 >> 16) & kRankMask};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      AB[rk[i]] = make_float2(PK.fa[i], PK.fb[i]);
      if constexpr (DIM == 3) CZ[rk[i]] = PK.fc[i];
    }
  }
  if (tid == 0) {                                 // what the slots beyond an item's entries read
    AB[n] = make_float2(0.0f, 0.0f);
    if constexpr (DIM == 3) CZ[n] = 0.0f;
    Sg[n] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  for (int i = tid; i < G; i += kS3Threads) ((int4*)acc)[i] = make_int4(0, 0, 0, 0);

  float gsx[2][kItemLen], gsy[2][kItemLen], gsz[2][kItemLen];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int j = 0; j < kItemLen; ++j) gsx[u][j] = gsy[u][j] = gsz[u][j] = 0.0f;

  CT_STAMP(2);
  auto group = [&](const int grp, auto more) {
    const int ch0 = grp * 4;
    const bool stamp = grp == cgi + ncg;
    if (stamp) CT_STAMP(16);
    // per-channel max |g_out * pad| of the segment (the fixed-point quantum), and the group into LDS in sorted order
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
    if (stamp) CT_STAMP(17);
    if (has) {
      const unsigned rk[4] = {S.rk01 & kRankMask, (S.rk01 >> 16) & kRankMask, S.rk23 & kRankMask, (S.rk23 >> 16) & kRankMask};
#pragma unroll
      for (int i = 0; i < 4; ++i) Sg[rk[i]] = make_float4(gq[0][i], gq[1][i], gq[2][i], gq[3][i]);
    }
    if (GATHER && tid < G) T4[tid] = make_float4(cvq[0], cvq[1], cvq[2], cvq[3]);
    for (int cell = tid + kS3Threads; GATHER && cell < G; cell += kS3Threads) {       // grids of more than blockDim cells
      const float* p = cnv0 + (size_t)ch0 * G + cell;
      T4[cell] = make_float4(ld_stream(p), ld_stream(p + G), ld_stream(p + 2 * (size_t)G), ld_stream(p + 3 * (size_t)G));
    }
    if (stamp) CT_STAMP(18);
    __syncthreads();
    if (stamp) CT_STAMP(19);
    if (stamp) CT_WSTAMP(24 + (threadIdx.x >> 6));

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
    // One item: its z = 0 face, then its z = 1 face; a face is the 2D item body on the face's four cells with the entries' values
    // scaled by the face's z weight.  The next group's loads are issued one channel per entry of the first item's first face.
    auto item = [&](auto U) {
      constexpr int u = decltype(U)::value;
      constexpr bool spread = kPrefetch && u == 0 && decltype(more)::value;
      unsigned it = S.item[u];
      asm volatile("" : "+v"(it));                  // (opaque per group: the unpacked fields are not kept across groups)
      const bool live = it != kS3NoItem;
      const int first = live ? (int)(it & 0xfffu) : n, ne = live ? (int)((it >> 12) & 3u) + 1 : 0, Y = live ? (int)(it >> 14) : 0;
#pragma unroll
      for (int f = 0; f < kFaces; ++f) {
        float4 cv[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) cv[v] = GATHER ? T4[Y + off2[v] + f] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        int e = ne > 0 ? first : n;
        float4 xn = Sg[e];
        float2 wn = AB[e];
        float zn = DIM == 3 ? CZ[e] : 0.0f;
        ct_f2 s01[4], s23[4];      // [corner of the face] x channels (0,1) / (2,3)
#pragma unroll
        for (int v = 0; v < 4; ++v) s01[v] = s23[v] = ct_f2{0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < kItemLen; ++j) {
          if constexpr (spread) {
            if (f == 0) request1(grp + ncg, j);
          }
          const float4 x = xn;
          const float2 wf = wn;
          const float w1z = zn;
          if (j + 1 < kItemLen) {
            e = j + 1 < ne ? first + j + 1 : n;
            xn = Sg[e];
            wn = AB[e];
            if constexpr (DIM == 3) zn = CZ[e];
          }
          const float w1x = wf.x, w1y = wf.y, w0x = 1.0f - w1x, w0y = 1.0f - w1y;
          const float wz = DIM == 2 ? 1.0f : (f ? w1z : 1.0f - w1z);
          const float cw2[4] = {w0x * w0y, w1x * w0y, w0x * w1y, w1x * w1y};      // (wx * wy) * wz: ct_corners<3>'s order
          const ct_f2 x01 = {x.x, x.y}, x23 = {x.z, x.w};
          float gw[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const float cw = DIM == 2 ? cw2[v] : cw2[v] * wz;
            const ct_f2 cwv = {cw, cw};
            s01[v] = __builtin_elementwise_fma(x01, cwv, s01[v]);
            s23[v] = __builtin_elementwise_fma(x23, cwv, s23[v]);
            if constexpr (GATHER) {
              const ct_f2 c01 = {cv[v].x, cv[v].y}, c23 = {cv[v].z, cv[v].w};
              const ct_f2 pr = __builtin_elementwise_fma(c23, x23, c01 * x01);
              gw[v] = pr.x + pr.y;
            }
          }
          if constexpr (GATHER) {
            // ct_corner_grad<3> face by face: d/dx, d/dy = the face's 2D expressions x its z weight, d/dz = -+ its bilinear value
            const float gxf = __builtin_fmaf(gw[3] - gw[2], w1y, (gw[1] - gw[0]) * w0y);
            const float gyf = __builtin_fmaf(gw[3] - gw[1], w1x, (gw[2] - gw[0]) * w0x);
            if constexpr (DIM == 3) {
              const float val = __builtin_fmaf(gw[3], cw2[3], __builtin_fmaf(gw[2], cw2[2], __builtin_fmaf(gw[1], cw2[1], gw[0] * cw2[0])));
              gsx[u][j] = __builtin_fmaf(gxf, wz, gsx[u][j]);
              gsy[u][j] = __builtin_fmaf(gyf, wz, gsy[u][j]);
              gsz[u][j] = f ? gsz[u][j] + val : gsz[u][j] - val;
              asm volatile("" : "+v"(gsx[u][j]), "+v"(gsy[u][j]), "+v"(gsz[u][j]));
            } else {
              gsx[u][j] += gxf;
              gsy[u][j] += gyf;
              asm volatile("" : "+v"(gsx[u][j]), "+v"(gsy[u][j]));
            }
          }
          CT_SB;
        }
        if (live) {
          const ct_f2 iq01 = {iq[0], iq[1]}, iq23 = {iq[2], iq[3]};
          int* Tc = acc + Y + f;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const ct_f2 a01 = s01[v] * iq01, a23 = s23[v] * iq23;
            atomicAdd(Tc + off2[v], cvt_rpi(a01.x));
            atomicAdd(Tc + G + off2[v], cvt_rpi(a01.y));
            atomicAdd(Tc + 2 * G + off2[v], cvt_rpi(a23.x));
            atomicAdd(Tc + 3 * G + off2[v], cvt_rpi(a23.y));
          }
        }
        CT_SB;
      }
    };
    item(std::integral_constant<int, 0>{});              // always: it carries the next group's loads
    if (stamp) CT_STAMP(20);
    if (S.item[1] != kS3NoItem) item(std::integral_constant<int, 1>{});
    if (stamp) CT_STAMP(21);
    if (stamp) CT_WSTAMP(40 + (threadIdx.x >> 6));
    if constexpr (decltype(more)::value && !kPrefetch) request(grp + ncg);
    if (any_float) {           // block-uniform, rare: IEEE float atomics for a channel with inf / NaN (or beyond the fixed-point bound)
#pragma unroll 1
      for (int cj = 0; cj < 4; ++cj)
        if (iq[cj] == 0.0f) {
          if constexpr (DIM == 3) scatter_float_channel3<HAS_PAD>(a, g, bh, b, ch0 + cj, (float*)(acc + cj * G), n, Nr, so);
          else scatter_float_channel<HAS_PAD>(a, g, bh, b, ch0 + cj, (float*)(acc + cj * G), 1, (size_t)so);
        }
    }
    __syncthreads();
    if (stamp) CT_STAMP(22);
    // the group's g_grid rows out (a segment's: its partial tile), the accumulators cleared for the next group
    float* gout = a.tile_out + (((size_t)seg * a.B * a.H + bh) * C + ch0) * (size_t)G;
    for (int t = tid; t < G; t += kS3Threads) {          // G int4 = 4 channels x G cells
      const int ch = (t << 2) / G;
      float q, iqd;
      bool fixed;
      fx_quantum(__uint_as_float(s_max[ch0 + ch]) * Kf, q, iqd, fixed);
      const int4 rr = ((const int4*)acc)[t];
      float4 o;
      if (fixed) o = make_float4((float)rr.x * q, (float)rr.y * q, (float)rr.z * q, (float)rr.w * q);
      else o = make_float4(__int_as_float(rr.x), __int_as_float(rr.y), __int_as_float(rr.z), __int_as_float(rr.w));
      st_part4(gout + ((size_t)t << 2), o, fold_grid);
      ((int4*)acc)[t] = make_int4(0, 0, 0, 0);
    }
    if (stamp) CT_STAMP(23);
  };
  int grp = cgi;
  for (; grp + ncg < ngroups; grp += ncg) group(grp, std::true_type{});
  group(grp, std::false_type{});
  CT_STAMP(3);

  if constexpr (GATHER) {
    // g_keys: from the item owners (sorted order) back to the point owners through LDS (the stage area is free: the last
    // group's readers are behind the barrier above)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned it = S.item[u];
      if (it != kS3NoItem) {
        const int first = (int)(it & 0xfffu), ne = (int)((it >> 12) & 3u) + 1;
#pragma unroll
        for (int j = 0; j < kItemLen; ++j)
          if (j < ne) Sg[first + j] = make_float4(gsx[u][j], gsy[u][j], gsz[u][j], 0.0f);
      }
    }
    __syncthreads();
    float* gp = a.g_pos + (size_t)cgi * a.gpos_stride;
    if (has) {
      const unsigned rw[4] = {S.rk01 & 0xffffu, S.rk01 >> 16, S.rk23 & 0xffffu, S.rk23 >> 16};
      float4 gk[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        gk[i] = Sg[rw[i] & kRankMask];
        gk[i].x *= (rw[i] & kInsideX) ? 1.0f : 0.0f;       // torch.clamp passes the cotangent only inside [lo, hi]
        gk[i].y *= (rw[i] & kInsideY) ? 1.0f : 0.0f;
        gk[i].z *= (rw[i] & kInsideZ) ? 1.0f : 0.0f;
      }
      st_part4(gp + (bh * DIM + 0) * Nr + so + n0, make_float4(gk[0].x, gk[1].x, gk[2].x, gk[3].x), fold_keys);
      st_part4(gp + (bh * DIM + 1) * Nr + so + n0, make_float4(gk[0].y, gk[1].y, gk[2].y, gk[3].y), fold_keys);
      if constexpr (DIM == 3) st_part4(gp + (bh * 3 + 2) * Nr + so + n0, make_float4(gk[0].z, gk[1].z, gk[2].z, gk[3].z), fold_keys);
    }
  }
  CT_STAMP(4);
  if (fold_keys || fold_grid) {       // kernel-uniform (see slice_bwd_fused_kernel)
    unsigned* s_flag = s_k + 1;
    const unsigned f = arrive_last(fold_keys ? a.tickets + (bh * nsg + seg) : nullptr, (unsigned)ncg,
                                   fold_grid ? a.tickets + kTicketHalf + (bh * ncg + cgi) : nullptr, (unsigned)nsg, s_flag);
    if (f & 1u) {      // this segment's g_keys: the channel groups' partials, ascending
#pragma unroll
      for (int j = 0; j < DIM; ++j)
        fold_rows(a.g_pos + (bh * DIM + j) * Nr + so, a.gpos_stride, ncg, a.fold_gpos + (bh * DIM + j) * Nr + so, n >> 2, nullptr);
    }
    if (f & 2u) {      // this workgroup's channel groups: the segments' partial tiles, ascending
      const size_t grid_n = (size_t)a.B * a.H * C * G;
      for (int gg = cgi; gg < ngroups; gg += ncg) {
        const size_t o = (bh * C + (size_t)gg * 4) * (size_t)G;
        fold_rows(a.tile_out + o, grid_n, nsg, a.fold_grid + o, G, nullptr);
      }
    }
  }
  CT_STAMP(5);
  CT_WG_STAMP(1);
}
