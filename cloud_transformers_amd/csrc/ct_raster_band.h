// Banded backward kernels for four-channel heads on grids whose tiles do not fit a CU (the zoo's 128^2 C4 planes and 32^3 C4
// volumes, model_zoo/s3dis/segmenter.py:28-33).  Included by ct_raster.hip after ct_raster_hot3d.h (uses their helpers).
//
// The kernels these replace keep ONE channel of the whole grid in LDS: every channel's workgroup re-reads the keys, the
// partial g_keys of the channels meet through memory (sum_parts), Slice backward is three launches (zero the statistics
// slots, gather + statistics, scatter) — 1.6-3.1x the algorithmic HBM traffic and 0.14-0.36 of the roofline
// (profiles/r3_zoo_counters.txt).  Here a workgroup owns a BAND of the grid — R consecutive x-rows (2D) / x-slabs (3D) —
// with all four channels of a cell in one 16-byte LDS word, exactly as the hot kernels of ct_raster_hot.h do for grids
// that fit whole:
//   * a band's workgroup scans the plane's keys (coalesced, L2-resident: the bands of a plane share an XCD) and compacts
//     the points whose base row lies in [x0 - 1, x0 + R) into an LDS ring; dense lanes then take points from the ring,
//     fetch their four channel values with scattered dword loads (4 per point: affordable at C = 4, the reason the
//     scheme stops there), and do the corner work on the band's tile;
//   * a cell is OWNED by the band its row belongs to: scatter contributions are applied by the owner only (points of row
//     x0 - 1 are processed a second time by the next band, for their dx = 1 corners: (R + 1) / R of the point work), so
//     nothing is merged between workgroups and the fixed-point sums stay bitwise reproducible;
//   * a point is owned by the band of its base row: its g_keys (all four channels, all corners: the tile holds the halo
//     row x0 + R read-only) and its g_feat are complete in one workgroup — no partial sums, no second launch.
// One launch per pass; keys read nb times from L2 and once from HBM.
#pragma once

#ifndef CT_BAND_THREADS
#define CT_BAND_THREADS 256
#endif
constexpr int kBandThreads = CT_BAND_THREADS;
constexpr int kBandScan = kBandThreads * 4;             // points scanned per round (a quad per thread)
constexpr int kBandCap = kBandScan + kBandThreads;      // ring capacity: a round's worth on top of an unprocessed remainder
constexpr int kBandCnt = 1024;                          // hashed per-cell contribution counters (an upper bound is enough)

template <int DIM>
struct BandPt {
  float w0[DIM], w1[DIM];
  float cw[1 << DIM];
  int lb;            // base cell relative to the tile's first row (x0 - 1)
  int fx;
};

template <int DIM>
__device__ __forceinline__ void band_point(const float (&k)[DIM], const GridW<DIM>& g, int S, int x0, BandPt<DIM>& p) {
  if constexpr (DIM == 2) {
    Pt2 q;
    pt2_from_keys(k[0], k[1], g, S, q);
    p.w0[0] = q.w0x; p.w1[0] = q.w1x; p.w0[1] = q.w0y; p.w1[1] = q.w1y;
#pragma unroll
    for (int v = 0; v < 4; ++v) p.cw[v] = q.cw[v];
    p.fx = q.base / S;
    p.lb = q.base - (x0 - 1) * S;
  } else {
    Pt3 q;
    pt3_from_keys(k[0], k[1], k[2], g, q);
#pragma unroll
    for (int j = 0; j < 3; ++j) { p.w0[j] = q.w0[j]; p.w1[j] = q.w1[j]; }
#pragma unroll
    for (int v = 0; v < 8; ++v) p.cw[v] = q.cw[v];
    p.fx = q.base / S;
    p.lb = q.base - (x0 - 1) * S;
  }
}

// base row of a key (the x axis term of ct_axis): which band(s) a point belongs to
template <int DIM>
__device__ __forceinline__ int band_fx(float kx, const GridW<DIM>& g) {
  float w0, w1;
  int f;
  ct_axis(kx, g.hw[0], g.W[0], w0, w1, f);
  return f;
}

template <int DIM>
__device__ __forceinline__ void band_offsets(const GridW<DIM>& g, int S, int (&off)[1 << DIM]) {
  if constexpr (DIM == 2) {
    off[0] = 0; off[1] = S; off[2] = 1; off[3] = S + 1;       // corner order of ct_corners<2>: (0,0), (1,0), (0,1), (1,1)
  } else {
    corner_offsets3(g, off);
  }
}

// The ring of selected points: n, keys.  Every thread scans one quad per round; selected points are appended with one LDS
// atomic each (the order of the ring is the order of arrival — it decides only which lane works on which point).
template <int DIM>
struct BandRing {
  int* n;
  float* k[DIM];
  int* count;        // entries in the ring
};

template <int DIM>
__device__ __forceinline__ void band_ring_init(BandRing<DIM>& r, float* mem) {
  r.n = (int*)mem;
#pragma unroll
  for (int j = 0; j < DIM; ++j) r.k[j] = mem + (size_t)(1 + j) * kBandCap;
  r.count = (int*)(mem + (size_t)(1 + DIM) * kBandCap);
}
template <int DIM>
constexpr size_t band_ring_bytes() { return ((size_t)(1 + DIM) * kBandCap + 4) * 4; }

// scan one round of the plane's points [n_beg, n_beg + kBandScan): append those with fx in [lo, hi] to the ring
template <int DIM>
__device__ __forceinline__ void band_scan_round(const float* keys, size_t bh, int N, int n_beg, const GridW<DIM>& g, int lo, int hi,
                                                BandRing<DIM>& r) {
  const int n0 = n_beg + ((int)threadIdx.x << 2);
  if (n0 < N) {
    const float4 tx = *(const float4*)(keys + (bh * DIM + 0) * N + n0);
    const float kx[4] = {tx.x, tx.y, tx.z, tx.w};
    int sel = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int fx = band_fx<DIM>(kx[i], g);
      sel |= (fx >= lo && fx <= hi) ? (1 << i) : 0;
    }
    if (sel) {
      float ko[DIM - 1][4];
#pragma unroll
      for (int j = 1; j < DIM; ++j) {
        const float4 t = *(const float4*)(keys + (bh * DIM + j) * N + n0);
        ko[j - 1][0] = t.x; ko[j - 1][1] = t.y; ko[j - 1][2] = t.z; ko[j - 1][3] = t.w;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (sel & (1 << i)) {
          const int slot = atomicAdd(r.count, 1);
          r.n[slot] = n0 + i;
          r.k[0][slot] = kx[i];
#pragma unroll
          for (int j = 1; j < DIM; ++j) r.k[j][slot] = ko[j - 1][i];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// BF: Slice backward of a band, fused: g_grid rows [x0, x0 + R) (fixed-point scatter-add, per-channel quantum from the
//   PLANE's max |g_out * pad| and the band's max contributions per cell) and g_keys of the band's points.
//   LDS: conv tile float4[(R+2) S] | accumulators {lo, hi} u64 x 2 pairs [(R+2) S] | counters [kBandCnt] | maxima [4] | K | ring
//   grid = (nb, H, B) through wg_coord (the bands of a plane on one XCD)
// ---------------------------------------------------------------------------
template <int DIM, bool HAS_PAD>
__global__ void __launch_bounds__(kBandThreads, 2) band_slice_bwd_kernel(RasterArgs a, GridW<DIM> g) {
  constexpr int V = 1 << DIM;
  extern __shared__ __align__(16) float lds[];
  const int R = a.nsplit, nb = a.ncg, N = a.N;
  const int S = g.G / g.W[0];                               // cells per row / slab
  const int cells = (R + 2) * S;
  float4* T4 = (float4*)lds;
  unsigned long long* acc = (unsigned long long*)(lds + (size_t)cells * 4);      // [2 pairs][cells]
  int* cnt = (int*)(lds + (size_t)cells * 8);
  unsigned* s_max = (unsigned*)(cnt + kBandCnt);            // [4]
  unsigned* s_k = s_max + 4;
  BandRing<DIM> ring;
  band_ring_init<DIM>(ring, (float*)(s_k + 4));
  const WgCoord wg = wg_coord(nb, 1, a.H, a.B);
  const int band = wg.cgi, h = wg.h, b = wg.b;
  const size_t bh = (size_t)b * a.H + h;
  const int tid = threadIdx.x;
  const int x0 = band * R;
  const int rows_own = min(R, g.W[0] - x0);                 // the last band may be short
  int off[V];
  band_offsets<DIM>(g, S, off);

  // ---- stage the conv rows [x0 - 1, x0 + R] (clamped to the grid; rows outside it are never addressed), zero the rest
  {
    const int r_lo = max(x0 - 1, 0), r_hi = min(x0 + R, g.W[0] - 1);
    const float* gin = a.tile_in + bh * 4 * (size_t)g.G;
    for (int t = tid; t < cells; t += blockDim.x) {
      const int row = x0 - 1 + t / S;
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (row >= r_lo && row <= r_hi) {
        const float* p = gin + (size_t)row * S + (t % S);
        v = make_float4(ld_stream(p), ld_stream(p + g.G), ld_stream(p + 2 * (size_t)g.G), ld_stream(p + 3 * (size_t)g.G));
      }
      T4[t] = v;
      acc[t] = 0ull;
      acc[t + cells] = 0ull;
    }
    for (int t = tid; t < kBandCnt + 8; t += blockDim.x) cnt[t] = 0;       // cnt, s_max, s_k are contiguous
    if (tid == 0) *ring.count = 0;
  }
  __syncthreads();

  // ---- pass A: per-channel max |g_out * pad| over the PLANE (coalesced rows), contributions per cell of the band's points
  {
    float mx[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int n0 = tid << 2; n0 < N; n0 += blockDim.x << 2) {
      float pv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pv[i] = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * N + n0 + i) : 1.0f;
#pragma unroll
      for (int cj = 0; cj < 4; ++cj) {
        const float4 t = *(const float4*)(a.src + (bh * 4 + cj) * (size_t)N + n0);
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x = fabsf(HAS_PAD ? tv[i] * pv[i] : tv[i]);
          mx[cj] = fmaxf(mx[cj], (x < __builtin_inff()) ? x : __builtin_inff());     // inf / NaN -> inf
        }
      }
      const float4 tx = *(const float4*)(a.pos.keys + (bh * DIM + 0) * N + n0);
      const float kx[4] = {tx.x, tx.y, tx.z, tx.w};
      int sel = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int fx = band_fx<DIM>(kx[i], g);
        sel |= (fx >= x0 - 1 && fx < x0 + R) ? (1 << i) : 0;
      }
      if (sel) {
        float ko[DIM][4];
        ko[0][0] = kx[0]; ko[0][1] = kx[1]; ko[0][2] = kx[2]; ko[0][3] = kx[3];
#pragma unroll
        for (int j = 1; j < DIM; ++j) {
          const float4 t = *(const float4*)(a.pos.keys + (bh * DIM + j) * N + n0);
          ko[j][0] = t.x; ko[j][1] = t.y; ko[j][2] = t.z; ko[j][3] = t.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (sel & (1 << i)) {
            float k[DIM];
#pragma unroll
            for (int j = 0; j < DIM; ++j) k[j] = ko[j][i];
            BandPt<DIM> p;
            band_point<DIM>(k, g, S, x0, p);
#pragma unroll
            for (int v = 0; v < V; ++v) atomicAdd(&cnt[(p.lb + off[v]) & (kBandCnt - 1)], 1);
          }
        }
      }
    }
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) {
      const unsigned mb = wave_max_u32(__float_as_uint(mx[cj]));
      if ((tid & 63) == 0) atomicMax(&s_max[cj], mb);
    }
  }
  __syncthreads();
  {
    unsigned kloc = 0;
    for (int t = tid; t < kBandCnt; t += blockDim.x) kloc = max(kloc, (unsigned)cnt[t]);
    kloc = wave_max_u32(kloc);
    if ((tid & 63) == 0) atomicMax(s_k, kloc);
  }
  __syncthreads();
  const float Kf = (float)max(*s_k, 1u);
  float iq[4], qv[4];
  bool fixedc[4];
#pragma unroll
  for (int cj = 0; cj < 4; ++cj) {
    fx_quantum(__uint_as_float(s_max[cj]) * Kf, qv[cj], iq[cj], fixedc[cj]);
    if (!fixedc[cj]) iq[cj] = 0.0f;
  }
  // channel pairs share a 64-bit accumulator word (slice_bwd_fused_kernel): a pair with a float-path channel is float whole
  const bool pair_fixed[2] = {fixedc[0] && fixedc[1], fixedc[2] && fixedc[3]};
  if (!pair_fixed[0]) iq[0] = iq[1] = 0.0f;
  if (!pair_fixed[1]) iq[2] = iq[3] = 0.0f;

  // ---- pass B: scan -> ring -> dense lanes
  auto process = [&](int slot) {
    const int n = ring.n[slot];
    float k[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) k[j] = ring.k[j][slot];
    BandPt<DIM> p;
    band_point<DIM>(k, g, S, x0, p);
    const float pad = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * N + n) : 1.0f;
    float f[4];
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) {
      const float x = a.src[(bh * 4 + cj) * (size_t)N + n];
      f[cj] = HAS_PAD ? x * pad : x;
    }
    const bool owned = p.fx >= x0;            // (fx < x0 + R by selection)
    if (owned) {
      float gw[V];
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const float4 cv = T4[p.lb + off[v]];
        float s = cv.x * f[0];
        s = __builtin_fmaf(cv.y, f[1], s);
        s = __builtin_fmaf(cv.z, f[2], s);
        s = __builtin_fmaf(cv.w, f[3], s);
        gw[v] = s;
      }
      float gd[DIM];
      ct_corner_grad<DIM>(p.w0, p.w1, gw, gd);
#pragma unroll
      for (int j = 0; j < DIM; ++j) a.g_pos[(bh * DIM + j) * (size_t)N + n] = gd[j] * ct_key_mask(k[j]);
    }
    // the scatter: every corner of every selected point (rows outside [x0, x0 + R) land in the tile's two halo rows and
    // are dropped at write-out: no branch here)
#pragma unroll
    for (int pj = 0; pj < 2; ++pj) {
      unsigned long long* Tc = acc + (size_t)pj * cells + p.lb;
      if (pair_fixed[pj]) {
        const float fa = f[2 * pj] * iq[2 * pj], fb = f[2 * pj + 1] * iq[2 * pj + 1];
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const int lo = cvt_rpi(fa * p.cw[v]), hi = cvt_rpi(fb * p.cw[v]);
          atomicAdd(Tc + off[v], ((unsigned long long)(unsigned)(hi + (lo >> 31)) << 32) | (unsigned)lo);
        }
      } else {            // inf / NaN / overflow of the bound in this pair: IEEE float atomics on the two halves (rare)
#pragma unroll
        for (int v = 0; v < V; ++v) {
          float* w = (float*)(Tc + off[v]);
          atomicAdd(w, f[2 * pj] * p.cw[v]);
          atomicAdd(w + 1, f[2 * pj + 1] * p.cw[v]);
        }
      }
    }
  };
  for (int n_beg = 0; n_beg < N; n_beg += kBandScan) {
    band_scan_round<DIM>(a.pos.keys, bh, N, n_beg, g, x0 - 1, x0 + R - 1, ring);
    __syncthreads();
    int have = *ring.count;
    while (have >= (int)blockDim.x) {            // block-uniform
      process(have - (int)blockDim.x + tid);
      have -= (int)blockDim.x;
    }
    __syncthreads();
    if (tid == 0) *ring.count = have;
    __syncthreads();
  }
  {
    const int have = *ring.count;
    if (tid < have) process(tid);
  }
  __syncthreads();

  // ---- write the band's own rows [x0, x0 + rows_own): local rows 1 .. rows_own
  {
    float* gout = a.tile_out + bh * 4 * (size_t)g.G + (size_t)x0 * S;
    const int n4 = (rows_own * S) >> 2;
    for (int t = tid; t < 2 * n4; t += blockDim.x) {
      const int pr = t / n4, cell = (t - pr * n4) << 2;
      const int4* w = (const int4*)(acc + (size_t)pr * cells + S + cell);
      const int4 r0 = w[0], r1 = w[1];
      const int lo[4] = {r0.x, r0.z, r1.x, r1.z}, hw[4] = {r0.y, r0.w, r1.y, r1.w};
      float4 oa, ob;
      if (pair_fixed[pr]) {
        const float qa = qv[2 * pr], qb = qv[2 * pr + 1];
        oa = make_float4((float)lo[0] * qa, (float)lo[1] * qa, (float)lo[2] * qa, (float)lo[3] * qa);
        ob = make_float4((float)(hw[0] - (lo[0] >> 31)) * qb, (float)(hw[1] - (lo[1] >> 31)) * qb,
                         (float)(hw[2] - (lo[2] >> 31)) * qb, (float)(hw[3] - (lo[3] >> 31)) * qb);
      } else {
        oa = make_float4(__int_as_float(lo[0]), __int_as_float(lo[1]), __int_as_float(lo[2]), __int_as_float(lo[3]));
        ob = make_float4(__int_as_float(hw[0]), __int_as_float(hw[1]), __int_as_float(hw[2]), __int_as_float(hw[3]));
      }
      st_stream4(gout + (size_t)(2 * pr) * g.G + cell, oa);
      st_stream4(gout + (size_t)(2 * pr + 1) * g.G + cell, ob);
    }
  }
}

// ---------------------------------------------------------------------------
// BB: Splat(max0) backward of a band: g_feat and g_keys (= gpos_add + result) of the band's points.
//   LDS: {z(c), z(c+1), g_z(c), g_z(c+1)} float4 x 2 pairs [(R+2) S] | counts | ring
//   The single-winner rule: a contribution whose product is bit-equal to a non-zero z wins its (cell, channel); the band
//   counts the matches that land in the rows its points touch (its own and the next band's first) against the non-zero
//   cells of those rows — equal when there is no tie there.  Otherwise the band is redone with claims whose winner is the contribution with the LOWEST (point, corner) index:
//   a rule every band evaluates alike for the rows it shares with its neighbour (both see all contributions to such a
//   row), so a cell still has one winner although two workgroups look at it.  The redo stages one channel pair at a time
//   (the claim words take the room of the other pair).
//   grid = (nb, H, B) through wg_coord
// ---------------------------------------------------------------------------
template <int DIM, bool HAS_PAD>
__global__ void __launch_bounds__(kBandThreads, 2) band_splat_bwd_kernel(RasterArgs a, GridW<DIM> g) {
  constexpr int V = 1 << DIM;
  extern __shared__ __align__(16) float lds[];
  const int R = a.nsplit, nb = a.ncg, N = a.N;
  const int S = g.G / g.W[0];
  const int cells = (R + 2) * S;
  float4* ZG = (float4*)lds;                                  // [2 pairs][cells]
  int* s_cnt = (int*)(lds + (size_t)cells * 8);               // nz, nm
  BandRing<DIM> ring;
  band_ring_init<DIM>(ring, (float*)(s_cnt + 4));
  const WgCoord wg = wg_coord(nb, 1, a.H, a.B);
  const int band = wg.cgi, h = wg.h, b = wg.b;
  const size_t bh = (size_t)b * a.H + h;
  const int tid = threadIdx.x;
  const int x0 = band * R;
  const int rows_own = min(R, g.W[0] - x0);
  const int r_lo = max(x0 - 1, 0), r_hi = min(x0 + R, g.W[0] - 1);
  int off[V];
  band_offsets<DIM>(g, S, off);
  // The tie test covers every row an owned point touches — the band's own rows and the next band's first (local rows
  // 1 .. R + 1, clipped to the grid): two bands that share a row must agree on whether it needs claims.
  const int own_lo = S, own_hi = (r_hi - (x0 - 1) + 1) * S;
  (void)rows_own;

  // stage one channel pair's {z, z, g, g} words of rows [x0 - 1, x0 + R]; returns this thread's count of non-zero z in own rows
  auto stage_pair = [&](int pr, float4* dst) {
    const float* zin = a.tile_in + (bh * 4 + 2 * pr) * (size_t)g.G;
    const float* gin = a.tile_in2 + (bh * 4 + 2 * pr) * (size_t)g.G;
    int nz = 0;
    for (int t = tid; t < cells; t += blockDim.x) {
      const int row = x0 - 1 + t / S;
      float4 v = make_float4(__uint_as_float(kNoMatch), __uint_as_float(kNoMatch), 0.0f, 0.0f);
      if (row >= r_lo && row <= r_hi) {
        const size_t o = (size_t)row * S + (t % S);
        const unsigned z0 = __float_as_uint(ld_stream(zin + o)), z1 = __float_as_uint(ld_stream(zin + o + g.G));
        v = make_float4(__uint_as_float(z0 ? z0 : kNoMatch), __uint_as_float(z1 ? z1 : kNoMatch), ld_stream(gin + o), ld_stream(gin + o + g.G));
        if (t >= own_lo && t < own_hi) nz += (z0 != 0u) + (z1 != 0u);
      }
      dst[t] = v;
    }
    return nz;
  };

  if (tid < 4) s_cnt[tid] = 0;
  if (tid == 0) *ring.count = 0;
  int nz = stage_pair(0, ZG) + stage_pair(1, ZG + cells);
  __syncthreads();

  int nm = 0;
  auto load_point = [&](int slot, int& n, float (&k)[DIM], BandPt<DIM>& p, float (&f)[4], float& pad) {
    n = ring.n[slot];
#pragma unroll
    for (int j = 0; j < DIM; ++j) k[j] = ring.k[j][slot];
    band_point<DIM>(k, g, S, x0, p);
    pad = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * N + n) : 1.0f;
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) {
      const float x = a.src[(bh * 4 + cj) * (size_t)N + n];
      f[cj] = HAS_PAD ? x * pad : x;
    }
  };
  auto store_point = [&](int n, const float (&k)[DIM], const BandPt<DIM>& p, const float (&gf)[4], const float (&gw)[V], float pad,
                         bool first) {
    // first: g_feat = gf, g_keys = add + contribution; else (the second channel pair of the claims redo): += on both
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) {
      float* q = a.dst + (bh * 4 + cj) * (size_t)N + n;
      const float v = HAS_PAD ? gf[cj] * pad : gf[cj];
      *q = first ? v : *q + v;
    }
    float gd[DIM];
    ct_corner_grad<DIM>(p.w0, p.w1, gw, gd);
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
      const size_t o = (bh * DIM + j) * (size_t)N + n;
      float v = gd[j] * ct_key_mask(k[j]);
      if (first) v = a.gpos_add != nullptr ? a.gpos_add[o] + v : v;
      else v = a.g_pos[o] + v;
      a.g_pos[o] = v;
    }
  };

  // ---- the claim-free pass
  auto process = [&](int slot) {
    int n;
    float k[DIM], f[4], pad;
    BandPt<DIM> p;
    load_point(slot, n, k, p, f, pad);
    const bool owned = p.fx >= x0 && p.fx < x0 + R;
    float gf[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gw[V];
#pragma unroll
    for (int v = 0; v < V; ++v) gw[v] = 0.0f;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const float xa = f[2 * pr], xb = f[2 * pr + 1];
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int cell = min(p.lb + off[v], cells - 1);       // (a row-(x0 + R) point's dx = 1 corners lie beyond the tile: never counted, never stored)
        const float4 zg = ZG[(size_t)pr * cells + cell];
        const unsigned ba = __float_as_uint(xa * p.cw[v]), bb = __float_as_uint(xb * p.cw[v]);
        const bool ma = ba == __float_as_uint(zg.x), mb = bb == __float_as_uint(zg.y);
        const bool own_cell = p.lb + off[v] >= own_lo && p.lb + off[v] < own_hi;
        nm += (int)(ma & own_cell) + (int)(mb & own_cell);
        const float ga = ma ? zg.z : 0.0f, gb = mb ? zg.w : 0.0f;
        gf[2 * pr] = __builtin_fmaf(ga, p.cw[v], gf[2 * pr]);
        gf[2 * pr + 1] = __builtin_fmaf(gb, p.cw[v], gf[2 * pr + 1]);
        gw[v] = __builtin_fmaf(gb, xb, __builtin_fmaf(ga, xa, gw[v]));
      }
    }
    if (owned) store_point(n, k, p, gf, gw, pad, true);
  };
  for (int n_beg = 0; n_beg < N; n_beg += kBandScan) {
    band_scan_round<DIM>(a.pos.keys, bh, N, n_beg, g, x0 - 1, x0 + R, ring);
    __syncthreads();
    int have = *ring.count;
    while (have >= (int)blockDim.x) {
      process(have - (int)blockDim.x + tid);
      have -= (int)blockDim.x;
    }
    __syncthreads();
    if (tid == 0) *ring.count = have;
    __syncthreads();
  }
  {
    const int have = *ring.count;
    if (tid < have) process(tid);
  }
  nz = wave_sum_i32(nz);
  nm = wave_sum_i32(nm);
  if ((tid & 63) == 0) {
    atomicAdd(&s_cnt[0], nz);
    atomicAdd(&s_cnt[1], nm);
  }
  __syncthreads();
  if (s_cnt[0] == s_cnt[1]) return;            // block-uniform: no exact ties among the contributions to this band's rows

  // ---- exact ties (duplicated points): redo the band with lowest-index claims, one channel pair at a time.
  //   Claims are needed for every row an owned point touches: [x0, x0 + R]; candidates: points of rows [x0 - 1, x0 + R].
  unsigned* claim = (unsigned*)(ZG + cells);                 // [cells][2]: the second pair's room
  for (int pr = 0; pr < 2; ++pr) {
    __syncthreads();
    (void)stage_pair(pr, ZG);
    for (int t = tid; t < 2 * cells; t += blockDim.x) claim[t] = 0xFFFFFFFFu;
    if (tid == 0) *ring.count = 0;
    __syncthreads();
    for (int phase = 0; phase < 2; ++phase) {
      auto work = [&](int slot) {
        int n;
        float k[DIM], f[4], pad;
        BandPt<DIM> p;
        load_point(slot, n, k, p, f, pad);
        const float xa = f[2 * pr], xb = f[2 * pr + 1];
        const bool owned = p.fx >= x0 && p.fx < x0 + R;
        float gf[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gw[V];
#pragma unroll
        for (int v = 0; v < V; ++v) gw[v] = 0.0f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const int cell = p.lb + off[v];
          if (cell < 0 || cell >= cells) continue;           // a corner of a row-(x0 + R) point beyond the tile: not arbitrated here
          const float4 zg = ZG[cell];
          const unsigned ba = __float_as_uint(xa * p.cw[v]), bb = __float_as_uint(xb * p.cw[v]);
          const bool ma = ba == __float_as_uint(zg.x), mb = bb == __float_as_uint(zg.y);
          const unsigned id = (unsigned)n * (unsigned)V + (unsigned)v;
          if (phase == 0) {
            if (ma) atomicMin(&claim[2 * cell], id);
            if (mb) atomicMin(&claim[2 * cell + 1], id);
          } else if (owned) {
            const float ga = (ma && claim[2 * cell] == id) ? zg.z : 0.0f;
            const float gb = (mb && claim[2 * cell + 1] == id) ? zg.w : 0.0f;
            gf[2 * pr] = __builtin_fmaf(ga, p.cw[v], gf[2 * pr]);
            gf[2 * pr + 1] = __builtin_fmaf(gb, p.cw[v], gf[2 * pr + 1]);
            gw[v] = __builtin_fmaf(gb, xb, __builtin_fmaf(ga, xa, gw[v]));
          }
        }
        if (phase == 1 && owned) {
          if (pr == 0) {
            store_point(n, k, p, gf, gw, pad, true);
          } else {               // the first pair's share is in memory: add this pair's (the other channels' g_feat are untouched)
            float* q0 = a.dst + (bh * 4 + 2) * (size_t)N + n;
            float* q1 = a.dst + (bh * 4 + 3) * (size_t)N + n;
            *q0 = HAS_PAD ? gf[2] * pad : gf[2];
            *q1 = HAS_PAD ? gf[3] * pad : gf[3];
            float gd[DIM];
            ct_corner_grad<DIM>(p.w0, p.w1, gw, gd);
#pragma unroll
            for (int j = 0; j < DIM; ++j) {
              const size_t o = (bh * DIM + j) * (size_t)N + n;
              a.g_pos[o] = a.g_pos[o] + gd[j] * ct_key_mask(k[j]);
            }
          }
        }
      };
      for (int n_beg = 0; n_beg < N; n_beg += kBandScan) {
        band_scan_round<DIM>(a.pos.keys, bh, N, n_beg, g, x0 - 1, x0 + R, ring);
        __syncthreads();
        const int have = *ring.count;
        for (int s0 = 0; s0 < have; s0 += (int)blockDim.x)
          if (s0 + tid < have) work(s0 + tid);
        __syncthreads();
        if (tid == 0) *ring.count = 0;
        __syncthreads();
      }
    }
  }
}
