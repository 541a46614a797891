// 3D (trilinear, 8 corners) forms of the hot-shape kernels of ct_raster_hot.h — the zoo's volume heads (16^3 C16,
// 8^3 C32; model_zoo/s3dis/segmenter.py:28-45).  Same design: channel-interleaved LDS tiles read with ds_read_b128,
// the fused Slice backward with a per-channel fixed-point quantum, the branch-free Splat(max) backward with its
// match-count tie detection.  A point's 8 corner reads are issued in two halves of 4 to stay inside 128 registers.
// Included by ct_raster.hip after ct_raster_hot.h (uses its helpers).
#pragma once

struct Pt3 {
  float w0[3], w1[3];
  float cw[8];          // corner v = dx + 2 dy + 4 dz, weight (wx * wy) * wz  (ct_corners<3>)
  int base;
};

__device__ __forceinline__ void pt3_from_keys(float kx, float ky, float kz, const GridW<3>& g, Pt3& p) {
  int f[3];
  ct_axis(kx, g.hw[0], g.W[0], p.w0[0], p.w1[0], f[0]);
  ct_axis(ky, g.hw[1], g.W[1], p.w0[1], p.w1[1], f[1]);
  ct_axis(kz, g.hw[2], g.W[2], p.w0[2], p.w1[2], f[2]);
  p.base = (f[0] * g.W[1] + f[1]) * g.W[2] + f[2];
  const float xy00 = p.w0[0] * p.w0[1], xy10 = p.w1[0] * p.w0[1], xy01 = p.w0[0] * p.w1[1], xy11 = p.w1[0] * p.w1[1];
  p.cw[0] = xy00 * p.w0[2]; p.cw[1] = xy10 * p.w0[2]; p.cw[2] = xy01 * p.w0[2]; p.cw[3] = xy11 * p.w0[2];
  p.cw[4] = xy00 * p.w1[2]; p.cw[5] = xy10 * p.w1[2]; p.cw[6] = xy01 * p.w1[2]; p.cw[7] = xy11 * p.w1[2];
}

// cell offsets of the 8 corners relative to the base cell
__device__ __forceinline__ void corner_offsets3(const GridW<3>& g, int (&off)[8]) {
  const int sx = g.W[1] * g.W[2], sy = g.W[2];
  off[0] = 0; off[1] = sx; off[2] = sy; off[3] = sx + sy;
  off[4] = 1; off[5] = sx + 1; off[6] = sy + 1; off[7] = sx + sy + 1;
}

__device__ __forceinline__ void load_keys3(const float* keys, size_t bh, int N, int n0, float (&k)[3][4]) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float4 t = *(const float4*)(keys + (bh * 3 + j) * N + n0);
    k[j][0] = t.x; k[j][1] = t.y; k[j][2] = t.z; k[j][3] = t.w;
  }
}

__device__ __forceinline__ void store_gkeys3(float* gpos, size_t bh, int N, int n0, const float (&gs)[4][3], const float (&k)[3][4],
                                             bool accumulate, bool handoff = false /* write-through: read by another workgroup */,
                                             const float* add = nullptr /* accumulate from these rows instead of gpos's own */) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float4 o = make_float4(gs[0][j] * ct_key_mask(k[j][0]), gs[1][j] * ct_key_mask(k[j][1]),
                           gs[2][j] * ct_key_mask(k[j][2]), gs[3][j] * ct_key_mask(k[j][3]));
    float* p = gpos + (bh * 3 + j) * N + n0;
    if (accumulate) {
      const float4 q = *(const float4*)(add != nullptr ? add + (bh * 3 + j) * N + n0 : p);
      o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
    }
    if (handoff) st_sc1_4(p, o);
    else *(float4*)p = o;
  }
}

// ---------------------------------------------------------------------------
// KF3: Slice backward, fused (see slice_bwd_fused_kernel).  grid = (ncg, H, B)
// ---------------------------------------------------------------------------
template <bool HAS_PAD, int QPT, int W3 = 0, int NTB = kHotThreads>
__global__ void __launch_bounds__(NTB, 4) slice_bwd_fused3_kernel(RasterArgs a, GridW<3> g_arg) {
  const GridW<3> g = grid3_of<W3>(g_arg);
  extern __shared__ __align__(16) float lds[];
  const int G = g.G, CC = a.CC, N = a.N;
  float4* T4 = (float4*)lds;
  int* acc = (int*)(lds + (size_t)CC * G);
  int* cnt = acc + (size_t)CC * G;
  unsigned* s_max = (unsigned*)(cnt + G);
  unsigned* s_k = s_max + a.C;
  const int nsg = a.nseg > 0 ? a.nseg : 1;                      // point segments, placement, folds: see slice_bwd_fused_kernel
  const WgCoord wg = wg_coord(a.ncg, nsg, a.H, a.B);
  const int h = wg.h, b = wg.b, seg = wg.seg;
  const bool fold_keys = a.tickets != nullptr && a.ncg > 1, fold_grid = a.tickets != nullptr && nsg > 1;
  const size_t bh = (size_t)b * a.H + h;
  const int Nr = a.Nrow > 0 ? a.Nrow : a.N;          // (0: a caller that knows no segments)
  const int so = seg * a.N;
  const int tid = threadIdx.x;
  int off[8];
  corner_offsets3(g, off);
  int n0[QPT], n0c[QPT];
  bool active[QPT];
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    n0[u] = (tid + u * (int)blockDim.x) << 2;
    active[u] = n0[u] < N;
    n0c[u] = active[u] ? n0[u] : 0;
  }
  float pv[QPT][4];
#pragma unroll
  for (int u = 0; u < QPT; ++u)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      pv[u][i] = (HAS_PAD && active[u]) ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * Nr + so + n0[u] + i) : 1.0f;
  for (int i = tid; i < G + a.C + 1; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    if (active[u]) {
      float k[3][4];
      load_keys3(a.pos.keys, bh, Nr, so + n0[u], k);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Pt3 p;
        pt3_from_keys(k[0][i], k[1][i], k[2][i], g, p);
        atomicAdd(&cnt[p.base], 1);
      }
    }
  }
  __syncthreads();
  {
    // contributions per cell = points based at the cell and at its 7 lower neighbours (a wrapped neighbour index lands
    // on a cell of the last row / column / slice, which is never a base: it reads zero)
    unsigned kloc = 0;
    for (int X = tid; X < G; X += blockDim.x) {
      unsigned c = 0;
#pragma unroll
      for (int v = 0; v < 8; ++v)
        if (X >= off[v]) c += (unsigned)cnt[X - off[v]];
      kloc = max(kloc, c);
    }
    kloc = wave_max_u32(kloc);
    if ((tid & 63) == 0) atomicMax(s_k, kloc);
  }
  // one quad per thread: its 12 g_keys values stay in registers over all channels; more quads: each (channel group, quad)
  // adds its share to the workgroup's own g_keys rows (read-modify-write, L2-resident) — 24 live values would spill
  constexpr bool kKeepGs = QPT == 1;
  float gs[1][4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i) gs[0][i][0] = gs[0][i][1] = gs[0][i][2] = 0.0f;

  const int cgi = wg.cgi;
  float* const gpos = a.g_pos + (size_t)cgi * a.gpos_stride;
  for (int chunk = cgi; chunk < a.nchunks; chunk += a.ncg) {
    const int c0 = chunk * CC;
    const int cc = min(CC, a.C - c0);
    const float* gin = a.tile_in + (bh * a.C + c0) * (size_t)G;
    float* gout = a.tile_out + (((size_t)seg * a.B * a.H + bh) * a.C + c0) * (size_t)G;
    for (int t = tid; t < (cc >> 2) * G; t += blockDim.x) {
      const int cq = t / G, cell = t - cq * G;
      const float* p = gin + (size_t)(cq * 4) * G + cell;
      T4[t] = make_float4(ld_stream(p), ld_stream(p + G), ld_stream(p + 2 * (size_t)G), ld_stream(p + 3 * (size_t)G));
    }
    if (chunk == cgi)
      for (int t = tid; t < (cc * G) >> 2; t += blockDim.x) ((int4*)acc)[t] = make_int4(0, 0, 0, 0);
    __syncthreads();
    const float Kf = (float)(*s_k);
    for (int cq = 0; cq < (cc >> 2); ++cq) {
      const int ch0 = c0 + cq * 4;
      float fv[4][4];
      float mx[4];
#pragma unroll
      for (int cj = 0; cj < 4; ++cj) {
        const float* row = a.src + (bh * a.C + ch0 + cj) * (size_t)Nr + so;
        const float4 t = ld_stream4(row + n0c[0]);
        fv[cj][0] = t.x; fv[cj][1] = t.y; fv[cj][2] = t.z; fv[cj][3] = t.w;
      }
#pragma unroll
      for (int cj = 0; cj < 4; ++cj) {
        float m = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float x = HAS_PAD ? fv[cj][i] * pv[0][i] : fv[cj][i];
          x = active[0] ? x : 0.0f;
          fv[cj][i] = x;
          x = fabsf(x);
          m = fmaxf(m, (x < __builtin_inff()) ? x : __builtin_inff());
        }
        mx[cj] = m;
      }
#pragma unroll
      for (int u = 1; u < QPT; ++u) {
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) {
          const float* row = a.src + (bh * a.C + ch0 + cj) * (size_t)Nr + so;
          const float4 t = *(const float4*)(row + n0c[u]);
          const float tv[4] = {t.x, t.y, t.z, t.w};
          float m = mx[cj];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float x = fabsf(HAS_PAD ? tv[i] * pv[u][i] : tv[i]);
            x = active[u] ? x : 0.0f;
            m = fmaxf(m, (x < __builtin_inff()) ? x : __builtin_inff());
          }
          mx[cj] = m;
        }
      }
#pragma unroll
      for (int cj = 0; cj < 4; ++cj) {
        const unsigned mb = wave_max_u32(__float_as_uint(mx[cj]));
        if ((tid & 63) == 0) atomicMax(&s_max[ch0 + cj], mb);
      }
      __syncthreads();
      float iq[4];
      bool any_float = false;
#pragma unroll
      for (int cj = 0; cj < 4; ++cj) {
        float q;
        bool fixed;
        fx_quantum(__uint_as_float(s_max[ch0 + cj]) * Kf, q, iq[cj], fixed);
        if (!fixed) {
          iq[cj] = 0.0f;
          any_float = true;
        }
      }
#if CT_FUSED_PAIR
      if (iq[0] == 0.0f || iq[1] == 0.0f) iq[0] = iq[1] = 0.0f;      // 64-bit {lo, hi} words: see slice_bwd_fused_kernel
      if (iq[2] == 0.0f || iq[3] == 0.0f) iq[2] = iq[3] = 0.0f;
#endif
      const float4* Tq = T4 + (size_t)cq * G;
      int* accq = acc + (size_t)(cq * 4) * G;
#pragma unroll
      for (int u = 0; u < QPT; ++u) {
        if (u > 0) {
          int n0r = n0c[u];
          asm volatile("" : "+v"(n0r));
#pragma unroll
          for (int cj = 0; cj < 4; ++cj) {
            const float* row = a.src + (bh * a.C + ch0 + cj) * (size_t)Nr + so;
            const float4 t = ld_stream4(row + n0r);
            fv[cj][0] = t.x; fv[cj][1] = t.y; fv[cj][2] = t.z; fv[cj][3] = t.w;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float x = HAS_PAD ? fv[cj][i] * pv[u][i] : fv[cj][i];
              fv[cj][i] = active[u] ? x : 0.0f;
            }
          }
        }
        float k[3][4];
        load_keys3(a.pos.keys, bh, Nr, so + n0c[u], k);
        if (!kKeepGs) {
#pragma unroll
          for (int i = 0; i < 4; ++i) gs[0][i][0] = gs[0][i][1] = gs[0][i][2] = 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float kx = k[0][i], ky = k[1][i], kz = k[2][i];
          asm volatile("" : "+v"(kx), "+v"(ky), "+v"(kz));        // one point's corner setup live at a time
          Pt3 p;
          pt3_from_keys(kx, ky, kz, g, p);
          float gw[8];
#pragma unroll
          for (int hv = 0; hv < 2; ++hv) {
            float4 cv[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) cv[v] = Tq[p.base + off[hv * 4 + v]];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              float s = cv[v].x * fv[0][i];
              s = __builtin_fmaf(cv[v].y, fv[1][i], s);
              s = __builtin_fmaf(cv[v].z, fv[2][i], s);
              s = __builtin_fmaf(cv[v].w, fv[3][i], s);
              gw[hv * 4 + v] = s;
            }
          }
          float gd[3];
          ct_corner_grad<3>(p.w0, p.w1, gw, gd);
          gs[0][i][0] += gd[0];
          gs[0][i][1] += gd[1];
          gs[0][i][2] += gd[2];
          asm volatile("" : "+v"(gs[0][i][0]), "+v"(gs[0][i][1]), "+v"(gs[0][i][2]));
#if CT_FUSED_PAIR
#pragma unroll
          for (int pj = 0; pj < 2; ++pj) {
            unsigned long long* Tc = (unsigned long long*)accq + (size_t)pj * G + p.base;
            const float fa = fv[2 * pj][i] * iq[2 * pj], fb = fv[2 * pj + 1][i] * iq[2 * pj + 1];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
              const int lo = cvt_rpi(fa * p.cw[v]), hi = cvt_rpi(fb * p.cw[v]);
              atomicAdd(Tc + off[v], ((unsigned long long)(unsigned)(hi + (lo >> 31)) << 32) | (unsigned)lo);
            }
          }
#else
#pragma unroll
          for (int cj = 0; cj < 4; ++cj) {
            int* Tc = accq + cj * G + p.base;
            const float fq = fv[cj][i] * iq[cj];
#pragma unroll
            for (int v = 0; v < 8; ++v) atomicAdd(Tc + off[v], cvt_rpi(fq * p.cw[v]));
          }
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
        if (!kKeepGs && active[u])      // (the workgroup's last share of a quad is the finished partial: handed off when folded)
          store_gkeys3(gpos, bh, Nr, so + n0c[u], gs[0], k, chunk > cgi || cq > 0,
                       fold_keys && chunk + a.ncg >= a.nchunks && cq == (cc >> 2) - 1);
      }
      if (any_float) {
#pragma unroll 1
        for (int cj = 0; cj < 4; ++cj) {
          float q, iqd;
          bool fixed;
          fx_quantum(__uint_as_float(s_max[ch0 + cj]) * Kf, q, iqd, fixed);
#if CT_FUSED_PAIR
          constexpr int es = 2;
          if (iq[cj] == 0.0f) {
            float* row_acc = (float*)(accq + (size_t)(cj >> 1) * 2 * G) + (cj & 1);
#else
          constexpr int es = 1;
          if (!fixed) {
            float* row_acc = (float*)(accq + cj * G);
#endif
            const float* src = a.src + (bh * a.C + ch0 + cj) * (size_t)Nr + so;
            for (int qd = tid; qd < (N >> 2); qd += blockDim.x) {
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
                for (int v = 0; v < 8; ++v) atomicAdd(row_acc + (p.base + off[v]) * es, x * p.cw[v]);
              }
            }
          }
        }
      }
    }
    __syncthreads();
    const bool more = chunk + a.ncg < a.nchunks;
#if CT_FUSED_PAIR
    for (int t = tid; t < (cc >> 1) * (G >> 2); t += blockDim.x) {
      const int pr = t / (G >> 2), cell = (t - pr * (G >> 2)) << 2;
      float qa, qb, iqd;
      bool fa, fb;
      fx_quantum(__uint_as_float(s_max[c0 + 2 * pr]) * Kf, qa, iqd, fa);
      fx_quantum(__uint_as_float(s_max[c0 + 2 * pr + 1]) * Kf, qb, iqd, fb);
      int4* w = (int4*)(acc + ((size_t)pr * G + cell) * 2);
      const int4 r0 = w[0], r1 = w[1];
      const int lo[4] = {r0.x, r0.z, r1.x, r1.z}, hw[4] = {r0.y, r0.w, r1.y, r1.w};
      float4 oa, ob;
      if (fa && fb) {
        oa = make_float4((float)lo[0] * qa, (float)lo[1] * qa, (float)lo[2] * qa, (float)lo[3] * qa);
        ob = make_float4((float)(hw[0] - (lo[0] >> 31)) * qb, (float)(hw[1] - (lo[1] >> 31)) * qb,
                         (float)(hw[2] - (lo[2] >> 31)) * qb, (float)(hw[3] - (lo[3] >> 31)) * qb);
      } else {
        oa = make_float4(__int_as_float(lo[0]), __int_as_float(lo[1]), __int_as_float(lo[2]), __int_as_float(lo[3]));
        ob = make_float4(__int_as_float(hw[0]), __int_as_float(hw[1]), __int_as_float(hw[2]), __int_as_float(hw[3]));
      }
      st_part4(gout + (size_t)(2 * pr) * G + cell, oa, fold_grid);
      st_part4(gout + (size_t)(2 * pr + 1) * G + cell, ob, fold_grid);
      if (more) w[0] = w[1] = make_int4(0, 0, 0, 0);
    }
#else
    for (int t = tid; t < (cc * G) >> 2; t += blockDim.x) {
      const int ch = (t << 2) / G;
      float q, iqd;
      bool fixed;
      fx_quantum(__uint_as_float(s_max[c0 + ch]) * Kf, q, iqd, fixed);
      const int4 r = ((const int4*)acc)[t];
      float4 o;
      if (fixed) o = make_float4((float)r.x * q, (float)r.y * q, (float)r.z * q, (float)r.w * q);
      else o = make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
      st_part4(gout + ((size_t)t << 2), o, fold_grid);
      if (more) ((int4*)acc)[t] = make_int4(0, 0, 0, 0);
    }
#endif
  }
  if (kKeepGs) {
    if (active[0]) {
      float k[3][4];
      load_keys3(a.pos.keys, bh, Nr, so + n0[0], k);
      store_gkeys3(gpos, bh, Nr, so + n0[0], gs[0], k, false, fold_keys);
    }
  }
  if (fold_keys || fold_grid) {       // kernel-uniform (see slice_bwd_fused_kernel)
    unsigned* s_flag = s_k + 1;
    const unsigned f = arrive_last(fold_keys ? a.tickets + (bh * nsg + seg) : nullptr, (unsigned)a.ncg,
                                   fold_grid ? a.tickets + kTicketHalf + (bh * a.ncg + cgi) : nullptr, (unsigned)nsg, s_flag);
    if (f & 1u) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        fold_rows(a.g_pos + (bh * 3 + j) * Nr + so, a.gpos_stride, a.ncg, a.fold_gpos + (bh * 3 + j) * Nr + so, N >> 2, nullptr);
    }
    if (f & 2u) {
      const size_t grid_n = (size_t)a.B * a.H * a.C * G;
      for (int chunk = cgi; chunk < a.nchunks; chunk += a.ncg) {
        const int c0 = chunk * CC, cc = min(CC, a.C - c0);
        const size_t o = (bh * a.C + c0) * (size_t)G;
        fold_rows(a.tile_out + o, grid_n, nsg, a.fold_grid + o, (cc * G) >> 2, nullptr);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// The repair of ONE exact tie through memory, for the forms whose key cotangents are already in their rows when the tie is known
// (3D; point segments; N beyond the register forms) — see splat_bwd_fix_one_tie / plane_sum_bits in ct_raster_hot.h.  `gbits`:
// the bit pattern of g_z at the tied (cell, channel), from the pass's bit sums.  The (cell, channel) pairs of the caller's chunks
// whose g_z has these bits are tried in turn: every point with a corner in the cell tests its product against the cell's
// maximum; exactly two must match (else the pair is not the tied one and nothing is written), the lower point index keeps the
// award, the other gives it back (one lane does that) — its g_feat element recomputed without that corner, its g_keys elements corrected in place
// by the negated award.  Any thread handles any point: the caller guarantees that every workgroup that wrote these rows is done
// (its own pass, or the plane's segments behind their last ticket).  False (block-uniform): redo.
//   rows: Nr floats per row (the whole plane); gpos: the g_keys rows to correct; wt: written through (another workgroup reads them)
// ---------------------------------------------------------------------------
template <int DIM>
__device__ __forceinline__ int tie_corner(int d, const int (&off)[1 << DIM]) {
  int v = -1;
#pragma unroll
  for (int c = 0; c < (1 << DIM); ++c) v = d == off[c] ? c : v;
  return v;
}

// one point's corner weights, base cell and per-axis terms from its keys (both dimensions behind one interface)
template <int DIM>
struct TiePoint {
  float w0[DIM], w1[DIM], cw[1 << DIM];
  int base;
};
template <int DIM>
__device__ __forceinline__ void tie_point(const float (&kp)[DIM], const GridW<DIM>& g, TiePoint<DIM>& tp) {
  if constexpr (DIM == 2) {
    Pt2 p;
    pt2_from_keys(kp[0], kp[1], g, g.W[1], p);
    tp.w0[0] = p.w0x; tp.w1[0] = p.w1x; tp.w0[1] = p.w0y; tp.w1[1] = p.w1y;
#pragma unroll
    for (int v = 0; v < 4; ++v) tp.cw[v] = p.cw[v];
    tp.base = p.base;
  } else {
    Pt3 p;
    pt3_from_keys(kp[0], kp[1], kp[2], g, p);
#pragma unroll
    for (int j = 0; j < 3; ++j) { tp.w0[j] = p.w0[j]; tp.w1[j] = p.w1[j]; }
#pragma unroll
    for (int v = 0; v < 8; ++v) tp.cw[v] = p.cw[v];
    tp.base = p.base;
  }
}

template <int DIM, bool HAS_PAD>
__device__ __forceinline__ bool splat_bwd_fix_mem_cell(const RasterArgs& a, const GridW<DIM>& g, size_t bh, int b, int ch, int t, int Nr,
                                                       float* gpos, bool wt, int* s_fix) {
  constexpr int V = 1 << DIM;
  const int G = g.G, tid = threadIdx.x, nq = Nr >> 2;
  int off[V];
  if constexpr (DIM == 2) {
    off[0] = 0; off[1] = g.W[1]; off[2] = 1; off[3] = g.W[1] + 1;
  } else {
    corner_offsets3(g, off);
  }
  const float* zrow = a.tile_in + (bh * a.C + ch) * (size_t)G;
  const float* grow = a.tile_in2 + (bh * a.C + ch) * (size_t)G;
  const float* srow = a.src + (bh * a.C + ch) * (size_t)Nr;
  const unsigned zt = __float_as_uint(zrow[t]);
  __syncthreads();
  if (tid < 4) s_fix[tid] = tid == 0 ? 0x7fffffff : 0;      // [0] lowest matching point, [1] matches, [2], [3] the first two of them
  __syncthreads();
  // every point of the plane: a corner in `t`?  (nothing is written in this loop: the key loads of several quads travel together)
#pragma unroll 1
  for (int q = tid; q < nq; q += blockDim.x) {
    float k[DIM][4];
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
      const float4 kv = *(const float4*)(a.pos.keys + (bh * DIM + j) * Nr + (q << 2));
      k[j][0] = kv.x; k[j][1] = kv.y; k[j][2] = kv.z; k[j][3] = kv.w;
    }
    unsigned cand = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int base = 0;
#pragma unroll
      for (int j = 0; j < DIM; ++j) {
        float w0, w1;
        int f;
        ct_axis(k[j][i], g.hw[j], g.W[j], w0, w1, f);
        base = base * g.W[j] + f;
      }
      if (tie_corner<DIM>(t - base, off) >= 0) cand |= 1u << i;
    }
#pragma unroll 1
    for (unsigned m = cand; m != 0u; m &= m - 1u) {          // rare
      const int i = __builtin_ctz(m), n = (q << 2) + i;
      float kp[DIM];
#pragma unroll
      for (int j = 0; j < DIM; ++j) kp[j] = i == 0 ? k[j][0] : i == 1 ? k[j][1] : i == 2 ? k[j][2] : k[j][3];
      TiePoint<DIM> tp;
      tie_point<DIM>(kp, g, tp);
      const int vt = tie_corner<DIM>(t - tp.base, off);
      float wc = 0.0f;
#pragma unroll
      for (int v = 0; v < V; ++v) wc = v == vt ? tp.cw[v] : wc;
      const float pv = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * Nr + n) : 1.0f;
      const float x = HAS_PAD ? srow[n] * pv : srow[n];
      if (__float_as_uint(x * wc) == zt) {
        const int c = atomicAdd(&s_fix[1], 1);
        if (c < 2) s_fix[2 + c] = n;
        atomicMin(&s_fix[0], n);
      }
    }
  }
  __syncthreads();
  if (s_fix[1] != 2) return false;          // block-uniform: not the tied pair, nothing written
  if (tid == 0) {
    // the loser (the higher of the two point indices): the award it was given, taken back through the corner-weight gradient,
    // and its g_feat element without that corner — the other corners keep what they matched, in the pass's order
    const int n = s_fix[2] == s_fix[0] ? s_fix[3] : s_fix[2];
    float kp[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) kp[j] = a.pos.keys[(bh * DIM + j) * Nr + n];
    TiePoint<DIM> tp;
    tie_point<DIM>(kp, g, tp);
    const int vt = tie_corner<DIM>(t - tp.base, off);
    const float pv = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * Nr + n) : 1.0f;
    const float x = HAS_PAD ? srow[n] * pv : srow[n];
    const float gt = grow[t];
    float gw[V], gd[DIM];
#pragma unroll
    for (int v = 0; v < V; ++v) gw[v] = v == vt ? -(gt * x) : 0.0f;
    ct_corner_grad<DIM>(tp.w0, tp.w1, gw, gd);
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
      float* pk = gpos + (bh * DIM + j) * Nr + n;
      const float nv = *pk + gd[j] * ct_key_mask(kp[j]);
      if (wt) __hip_atomic_store(pk, nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else *pk = nv;
    }
    float gf = 0.0f;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const unsigned zc = __float_as_uint(zrow[tp.base + off[v]]);
      const float gc = grow[tp.base + off[v]];
      const bool won = v != vt && zc != 0u && __float_as_uint(x * tp.cw[v]) == zc;
      gf = __builtin_fmaf(won ? gc : 0.0f, tp.cw[v], gf);
    }
    a.dst[(bh * a.C + ch) * (size_t)Nr + n] = HAS_PAD ? gf * pv : gf;
  }
  __syncthreads();
  return true;
}

template <int DIM, bool HAS_PAD>
__device__ __forceinline__ bool splat_bwd_fix_mem(const RasterArgs& a, const GridW<DIM>& g, size_t bh, int b, int cgi, unsigned gbits,
                                                  int Nr, float* gpos, bool wt, int* s_cnt) {
  const int G = g.G, tid = threadIdx.x;
  int* s_fix = s_cnt + kTieMemFix;
  int* s_list = s_cnt + kTieMemList;
  if (G > 0xffff) return false;
  __syncthreads();
  if (tid == 0) s_list[0] = 0;
  __syncthreads();
  for (int chunk = cgi; chunk < a.nchunks; chunk += a.ncg) {
    const int c0 = chunk * a.CC, cc = min(a.CC, a.C - c0);
    const float* zrow = a.tile_in + (bh * a.C + c0) * (size_t)G;
    const float* grow = a.tile_in2 + (bh * a.C + c0) * (size_t)G;
    // 2D: four loads in flight, the tests behind them (64^2 C16 B2 N16384: a workgroup walks 16 k words).  3D: one at a time — ANY
    // batching here puts 10+ scratch accesses into the 3D kernel's LOOP (the register allocator's doing, measured: +8 us per launch)
    const int nn = cc * G, step = (int)blockDim.x;
    constexpr int UB = DIM == 2 ? 4 : 1;
    for (int i0 = tid; i0 < nn; i0 += UB * step) {
      unsigned gv[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) gv[u] = __float_as_uint(grow[min(i0 + u * step, nn - 1)]);
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int i = i0 + u * step;
        if (i < nn && gv[u] == gbits && __float_as_uint(zrow[i]) != 0u) {
          const int k = atomicAdd(&s_list[0], 1);
          if (k < kTieTry) s_list[1 + k] = ((c0 + i / G) << 16) | (i % G);
        }
      }
    }
  }
  __syncthreads();
  const int n = s_list[0];
  if (n == 0 || n > kTieTry) return false;
#pragma unroll 1
  for (int k = 0; k < n; ++k) {
    const int e = s_list[1 + k];
    if (splat_bwd_fix_mem_cell<DIM, HAS_PAD>(a, g, bh, b, e >> 16, e & 0xffff, Nr, gpos, wt, s_fix)) { CT_TIE_COUNT(9, 1); return true; }
  }
  CT_TIE_COUNT(10, 1);
  return false;
}

// The caller's arguments, read again from the kernel's argument segment: the repair is a cold path at the end of kernels whose
// loops have no register to spare, and holding the arguments for it across those loops costs them dearly (3D: +8..14 us per
// launch when the repair used the kernel's own copies).  Valid in kernels whose parameters are (RasterArgs, GridW<DIM>).
// ENFORCED: the kernels that call this (splat_max_bwd_hot_kernel, splat_max_bwd_hot3_kernel) carry a static_assert on
// their signature right behind their definitions (CT_KERNARG_IS_ARGS_AND_GRID); the argument types must be trivially copyable
// standard-layout structs (the segment holds their bytes as the host passed them); and those kernels must NOT patch their
// arguments in place (the sorted kernels do — a.N, a.C — and therefore do not use this): the repair would see the caller's values.
#define CT_KERNARG_IS_ARGS_AND_GRID(KERNEL_INSTANCE, DIMV)                                                                   \
  static_assert(std::is_same<decltype(&KERNEL_INSTANCE), void (*)(RasterArgs, GridW<DIMV>)>::value,                          \
                "splat_bwd_fix_mem_cold reads (RasterArgs, GridW<DIM>) back from the kernarg segment: the kernel's parameter list must be exactly that")
template <int DIM, bool HAS_PAD>
__device__ __forceinline__ bool splat_bwd_fix_mem_cold(size_t bh, int b, int cgi, unsigned gbits, int Nr, size_t gpos_off, bool wt,
                                                       int* s_cnt) {
  static_assert(std::is_trivially_copyable<RasterArgs>::value && std::is_standard_layout<RasterArgs>::value &&
                    std::is_trivially_copyable<GridW<DIM>>::value && std::is_standard_layout<GridW<DIM>>::value,
                "kernel arguments are read back as raw bytes");
  static_assert(alignof(RasterArgs) <= 8 && alignof(GridW<DIM>) <= 8, "the kernarg segment is 8-byte aligned at least: nothing here may want more");
  const char* kp = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  const RasterArgs& a = *(const RasterArgs*)kp;
  constexpr size_t goff = (sizeof(RasterArgs) + alignof(GridW<DIM>) - 1) / alignof(GridW<DIM>) * alignof(GridW<DIM>);
  const GridW<DIM>& g = *(const GridW<DIM>*)(kp + goff);
  return splat_bwd_fix_mem<DIM, HAS_PAD>(a, g, bh, b, cgi, gbits, Nr, a.g_pos + gpos_off, wt, s_cnt);
}

// ---------------------------------------------------------------------------
// KB3: Splat(max0) backward (see splat_max_bwd_hot_kernel).  grid = (ncg, H, B)
// ---------------------------------------------------------------------------
template <bool HAS_PAD, bool CLAIMS>
__device__ __forceinline__ void splat_bwd_quad3(const RasterArgs& a, const GridW<3>& g, float4* ZG, size_t bh, int b, int c0,
                                                int cc, int n0, const PtRows& R, const float (&k)[3][4], const int (&off)[8],
                                                float (&gs)[4][3], int& nm, unsigned& xs) {
  const int G = g.G;
  float pv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pv[i] = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * R.Nr + R.so + n0 + i) : 1.0f;
  for (int cg0 = 0; cg0 < cc; cg0 += 4) {
    float fv[4][4];
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) {
      const float* row = a.src + (bh * a.C + c0 + cg0 + cj) * (size_t)R.Nr + R.so;
      const float4 t = ld_stream4(row + n0);
      fv[cj][0] = t.x; fv[cj][1] = t.y; fv[cj][2] = t.z; fv[cj][3] = t.w;
    }
    float4* Zc = ZG + (size_t)(cg0 >> 1) * G;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // the corner cells / weights are recomputed per channel group: hoisted out of the channel loop they would keep
      // 4 x 15 registers alive (the opaque copies stop the hoisting)
      float kx = k[0][i], ky = k[1][i], kz = k[2][i];
      asm volatile("" : "+v"(kx), "+v"(ky), "+v"(kz));
      Pt3 p;
      pt3_from_keys(kx, ky, kz, g, p);
      float gw[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        float4* Zp = Zc + (size_t)pr * G + p.base;
        const float xa = HAS_PAD ? fv[2 * pr][i] * pv[i] : fv[2 * pr][i];
        const float xb = HAS_PAD ? fv[2 * pr + 1][i] * pv[i] : fv[2 * pr + 1][i];
        float gfa = 0.0f, gfb = 0.0f;
#pragma unroll
        for (int hv = 0; hv < 2; ++hv) {
          ct_f4 zg[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) zg[v] = *(const ct_f4*)(Zp + off[hv * 4 + v]);
          // (whole reads, pinned per component: see splat_bwd_quad)
#pragma unroll
          for (int v = 0; v < 4; ++v) asm volatile("" : "+v"(zg[v].x), "+v"(zg[v].y), "+v"(zg[v].z), "+v"(zg[v].w));
          if (!CLAIMS) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const float w = p.cw[hv * 4 + v];
              const unsigned ba = __float_as_uint(xa * w), bb = __float_as_uint(xb * w);
              const bool ma = ba == __float_as_uint(zg[v].x), mb = bb == __float_as_uint(zg[v].y);
              nm += (int)ma;
              nm += (int)mb;
              asm volatile("" : "+v"(nm));
              const float ga = ma ? zg[v].z : 0.0f, gb = mb ? zg[v].w : 0.0f;
              if (CT_TIE_FIX) {      // the bit patterns of the awarded cotangents, summed (ct_raster_hot.h: plane_sum_bits)
                xs += __float_as_uint(ga) + __float_as_uint(gb);
                asm volatile("" : "+v"(xs));
              }
              gfa = __builtin_fmaf(ga, w, gfa);
              gfb = __builtin_fmaf(gb, w, gfb);
              gw[hv * 4 + v] = __builtin_fmaf(gb, xb, __builtin_fmaf(ga, xa, gw[hv * 4 + v]));
            }
          } else {
            unsigned ba[4], bb[4];
            bool ma[4], mb[4];
            bool any = false;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const float w = p.cw[hv * 4 + v];
              ba[v] = __float_as_uint(xa * w);
              bb[v] = __float_as_uint(xb * w);
              ma[v] = ba[v] == __float_as_uint(zg[v].x);
              mb[v] = bb[v] == __float_as_uint(zg[v].y);
              any = any | ma[v] | mb[v];
            }
            if (any) {
#pragma unroll
              for (int v = 0; v < 4; ++v) {
                const float w = p.cw[hv * 4 + v];
                unsigned* zw = (unsigned*)(Zp + off[hv * 4 + v]);
                const unsigned oa = atomicCAS(zw, ma[v] ? ba[v] : kNoMatch, kNoMatch);
                const unsigned ob = atomicCAS(zw + 1, mb[v] ? bb[v] : kNoMatch, kNoMatch);
                const float ga = (ma[v] & (oa == ba[v])) ? zg[v].z : 0.0f;
                const float gb = (mb[v] & (ob == bb[v])) ? zg[v].w : 0.0f;
                gfa = __builtin_fmaf(ga, w, gfa);
                gfb = __builtin_fmaf(gb, w, gfb);
                gw[hv * 4 + v] = __builtin_fmaf(gb, xb, __builtin_fmaf(ga, xa, gw[hv * 4 + v]));
              }
            }
          }
        }
        fv[2 * pr][i] = HAS_PAD ? gfa * pv[i] : gfa;
        fv[2 * pr + 1][i] = HAS_PAD ? gfb * pv[i] : gfb;
      }
      float gd[3];
      ct_corner_grad<3>(p.w0, p.w1, gw, gd);
      gs[i][0] += gd[0];
      gs[i][1] += gd[1];
      gs[i][2] += gd[2];
      asm volatile("" : "+v"(gs[i][0]), "+v"(gs[i][1]), "+v"(gs[i][2]), "+v"(fv[0][i]), "+v"(fv[1][i]), "+v"(fv[2][i]), "+v"(fv[3][i]));
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int cj = 0; cj < 4; ++cj)
      st_part4(a.dst + (bh * a.C + c0 + cg0 + cj) * (size_t)R.Nr + R.so + n0, make_float4(fv[cj][0], fv[cj][1], fv[cj][2], fv[cj][3]), R.wt);
  }
}

template <bool HAS_PAD, bool CLAIMS, int QPT>
__device__ __forceinline__ void splat_bwd_plane_pass3(const RasterArgs& a, const GridW<3>& g, float4* ZG, int* s_cnt, size_t bh,
                                                     int b, int cgi, int N, const PtRows& R, const int (&off)[8],
                                                     float (&gs_reg)[QPT ? QPT : 1][4][3], bool& tie) {
  const int G = g.G, CC = a.CC;
  const int tid = threadIdx.x;
  const int nq = N >> 2;
  int nz = 0, nm = 0;
  unsigned xm = 0u, xz = 0u;      // (CT_TIE_FIX) bit-pattern sums of the awarded cotangents / of those of the non-zero cells
  float* gpos = a.g_pos + (size_t)cgi * a.gpos_stride;
  for (int chunk = cgi; chunk < a.nchunks; chunk += a.ncg) {
    const int c0 = chunk * CC;
    const int cc = min(CC, a.C - c0);
    const float* zin = a.tile_in + (bh * a.C + c0) * (size_t)G;
    const float* gin = a.tile_in2 + (bh * a.C + c0) * (size_t)G;
    __syncthreads();
    for (int t = tid; t < (cc >> 1) * G; t += blockDim.x) {
      const int cp = t / G, cell = t - cp * G;
      const size_t o = (size_t)(cp * 2) * G + cell;
      const unsigned z0 = __float_as_uint(ld_stream(zin + o)), z1 = __float_as_uint(ld_stream(zin + o + G));
      const float g0 = ld_stream(gin + o), g1 = ld_stream(gin + o + G);
      ZG[t] = make_float4(__uint_as_float(z0 ? z0 : kNoMatch), __uint_as_float(z1 ? z1 : kNoMatch), g0, g1);
      if (!CLAIMS) {
        nz += (z0 != 0u) + (z1 != 0u);
        if (CT_TIE_FIX) xz += (z0 ? __float_as_uint(g0) : 0u) + (z1 ? __float_as_uint(g1) : 0u);
      }
    }
    __syncthreads();
    if constexpr (QPT > 0) {
#pragma unroll
      for (int u = 0; u < QPT; ++u) {
        const int q = tid + u * (int)blockDim.x;
        if (q < nq) {
          float k[3][4];
          load_keys3(a.pos.keys, bh, R.Nr, (int)R.so + (q << 2), k);
          splat_bwd_quad3<HAS_PAD, CLAIMS>(a, g, ZG, bh, b, c0, cc, q << 2, R, k, off, gs_reg[u], nm, xm);
        }
      }
    } else {
      for (int q = tid; q < nq; q += blockDim.x) {
        float k[3][4];
        load_keys3(a.pos.keys, bh, R.Nr, (int)R.so + (q << 2), k);
        float gs[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) gs[i][0] = gs[i][1] = gs[i][2] = 0.0f;
        splat_bwd_quad3<HAS_PAD, CLAIMS>(a, g, ZG, bh, b, c0, cc, q << 2, R, k, off, gs, nm, xm);
        // a chunk adds to the workgroup's own rows; the first starts from the incoming cotangent where there is one
        // (a.gpos_add); the finished rows go out write-through where another workgroup reads or overwrites them
        store_gkeys3(gpos, bh, R.Nr, (int)R.so + (q << 2), gs, k, chunk > cgi || a.gpos_add != nullptr,
                     chunk + a.ncg >= a.nchunks && (R.wt || (a.tickets != nullptr && a.ncg > 1)),
                     chunk > cgi ? nullptr : a.gpos_add);
      }
    }
  }
  if (!CLAIMS) {
    nz = wave_sum_i32(nz);
    nm = wave_sum_i32(nm);
    if ((tid & 63) == 0) {
      atomicAdd(&s_cnt[0], nz);
      atomicAdd(&s_cnt[1], nm);
    }
    if (CT_TIE_FIX) {
      plane_sum_bits(s_cnt + kTieSumPos, xm, 1);
      plane_sum_bits(s_cnt + kTieSumNeg, xz, 1);
    }
    __syncthreads();
    tie = s_cnt[0] != s_cnt[1];
  }
}

template <bool HAS_PAD, int QPT, int W3 = 0, int NTB = kHotThreads>
__global__ void __launch_bounds__(NTB, 4) splat_max_bwd_hot3_kernel(RasterArgs a, GridW<3> g_arg) {
  const GridW<3> g = grid3_of<W3>(g_arg);
  extern __shared__ __align__(16) float lds[];
  float4* ZG = (float4*)lds;
  int* s_cnt = (int*)(lds + (size_t)a.CC * g.G * 2);
  const int nsg = a.nseg > 0 ? a.nseg : 1;               // point segments and their tie test: see splat_max_bwd_hot_kernel
  const WgCoord wg = wg_coord(a.ncg, nsg, a.H, a.B);
  const int h = wg.h, b = wg.b;
  const size_t bh = (size_t)b * a.H + h;
  const int N = a.N;
  PtRows R;
  R.Nr = a.Nrow > 0 ? a.Nrow : a.N;
  R.so = (size_t)wg.seg * a.N;
  R.wt = nsg > 1;
  const bool fold_keys = a.tickets != nullptr && a.ncg > 1;
  int off[8];
  corner_offsets3(g, off);
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  if (threadIdx.x < 4 + kTieFixWords + kTieFixWords) s_cnt[kTieSumPos + threadIdx.x] = 0;
  float gs[QPT ? QPT : 1][4][3];
#pragma unroll
  for (int u = 0; u < (QPT ? QPT : 1); ++u)
#pragma unroll
    for (int i = 0; i < 4; ++i) gs[u][i][0] = gs[u][i][1] = gs[u][i][2] = 0.0f;
  bool tie = false;
  splat_bwd_plane_pass3<HAS_PAD, false, QPT>(a, g, ZG, s_cnt, bh, b, wg.cgi, N, R, off, gs, tie);
  if (CT_TIE_FIX && QPT == 0 && tie && nsg == 1 && s_cnt[1] - s_cnt[0] == 1) {      // block-uniform: ONE surplus match in this pass
    const unsigned gbits = (unsigned)(s_cnt[kTieSumPos] - s_cnt[kTieSumNeg]);
    // (+0: the surplus award added nothing anywhere)
    if (gbits == 0u || splat_bwd_fix_mem_cold<3, HAS_PAD>(bh, b, wg.cgi, gbits, R.Nr, (size_t)wg.cgi * a.gpos_stride, fold_keys, s_cnt))
      tie = false;
  }
  if (tie && nsg == 1) {
#pragma unroll
    for (int u = 0; u < (QPT ? QPT : 1); ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) gs[u][i][0] = gs[u][i][1] = gs[u][i][2] = 0.0f;
    splat_bwd_plane_pass3<HAS_PAD, true, QPT>(a, g, ZG, s_cnt, bh, b, wg.cgi, N, R, off, gs, tie);
  }
  if constexpr (QPT > 0) {
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
      const int n0 = ((int)threadIdx.x + u * (int)blockDim.x) << 2;
      if (n0 < N) {
        float k[3][4];
        load_keys3(a.pos.keys, bh, R.Nr, (int)R.so + n0, k);
        store_gkeys3(a.g_pos + (size_t)wg.cgi * a.gpos_stride, bh, R.Nr, (int)R.so + n0, gs[u], k, a.gpos_add != nullptr,
                     fold_keys || R.wt, a.gpos_add);
      }
    }
  }
  if (fold_keys) {       // kernel-uniform (see splat_max_bwd_hot_kernel)
    if (arrive_last(a.tickets + bh, (unsigned)a.ncg, nullptr, 0u, (unsigned*)(s_cnt + 2)) & 1u) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        fold_rows(a.g_pos + (bh * 3 + j) * N, a.gpos_stride, a.ncg, a.fold_gpos + (bh * 3 + j) * N, N >> 2,
                  a.fold_add != nullptr ? a.fold_add + (bh * 3 + j) * N : nullptr);
    }
  }
  if (nsg > 1) {         // kernel-uniform: the plane's tie test across its segments
    // the plane's matches (low word) and the bit sum of their cotangents (high word: plane_sum_bits) in ONE 64-bit word of the
    // ticket buffer's second half (splat_bwd_hot_plan: planes <= kTicketHalf / 2)
    unsigned long long* matches = (unsigned long long*)(a.tickets + kTicketHalf) + bh;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(matches, ((unsigned long long)(unsigned)s_cnt[kTieSumPos] << 32) | (unsigned)s_cnt[1], __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // counted before the ticket is taken
    }
    const bool last = (arrive_last(a.tickets + bh, (unsigned)nsg, nullptr, 0u, (unsigned*)(s_cnt + 2)) & 1u) != 0;
    if (last) {          // block-uniform
      // (read and reset in ONE atomic: a load could be served from a line an earlier launch left in this XCD's L2)
      if (threadIdx.x == 0) {
        const unsigned long long m = __hip_atomic_exchange(matches, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_cnt[3] = (int)(unsigned)m;
        s_cnt[2] = (int)(unsigned)(m >> 32);      // (the low word never carries into it: the matches of a plane are < 2^32)
      }
      __syncthreads();
      bool redo = s_cnt[3] != s_cnt[0];
      if (CT_TIE_FIX && s_cnt[3] - s_cnt[0] == 1) {      // ONE surplus match in the plane: s_cnt[2] = the segments' bit sums
        const unsigned gbits = (unsigned)(s_cnt[2] - s_cnt[kTieSumNeg]);
        if (gbits == 0u || splat_bwd_fix_mem_cold<3, HAS_PAD>(bh, b, 0, gbits, R.Nr, 0, false, s_cnt)) redo = false;
      }
      if (redo) {       // the plane has exact ties: all of it again, with claims, by this workgroup
        PtRows Rall;
        Rall.Nr = R.Nr; Rall.so = 0; Rall.wt = false;
        float gs0[1][4][3];
        splat_bwd_plane_pass3<HAS_PAD, true, 0>(a, g, ZG, s_cnt, bh, b, 0, R.Nr, Rall, off, gs0, tie);
      }
    }
  }
}

CT_KERNARG_IS_ARGS_AND_GRID((splat_max_bwd_hot3_kernel<false, 0, 0>), 3);
CT_KERNARG_IS_ARGS_AND_GRID((splat_max_bwd_hot3_kernel<true, 0, 16, kHotWideThreads>), 3);

// ---------------------------------------------------------------------------
// KG3: Slice forward / gather with the channel-interleaved tile.  grid = (nchunks * nsplit, H, B)
// ---------------------------------------------------------------------------
template <bool HAS_PAD, int W3 = 0>
__global__ void __launch_bounds__(kHotThreads, 4) gather_ci3_kernel(RasterArgs a, GridW<3> g_arg) {
  const GridW<3> g = grid3_of<W3>(g_arg);
  extern __shared__ __align__(16) float lds[];
  const int G = g.G, N = a.N;
  float4* T4 = (float4*)lds;
  const BlockXHB blk = block_xhb();          // (one XCD per plane: see gather_ci_kernel)
  const int chunk = blk.x / a.nsplit, sp = blk.x % a.nsplit;
  const int h = blk.h, b = blk.b;
  const size_t bh = (size_t)b * a.H + h;
  const int c0 = chunk * a.CC;
  const int cc = min(a.CC, a.C - c0);
  const int tid = threadIdx.x;
  int off[8];
  corner_offsets3(g, off);
  const float* gin = a.tile_in + (bh * a.C + c0) * (size_t)G;
  for (int t = tid; t < (cc >> 2) * G; t += blockDim.x) {
    const int cq = t / G, cell = t - cq * G;
    const float* p = gin + (size_t)(cq * 4) * G + cell;
    T4[t] = make_float4(ld_stream(p), ld_stream(p + G), ld_stream(p + 2 * (size_t)G), ld_stream(p + 3 * (size_t)G));
  }
  __syncthreads();
  const int nq = N >> 2;
  const int per = (nq + a.nsplit - 1) / a.nsplit;
  const int q_beg = sp * per, q_end = min(nq, q_beg + per);
  float* dst = a.dst + (bh * a.C + c0) * (size_t)N;
  for (int q = q_beg + tid; q < q_end; q += blockDim.x) {
    const int n0 = q << 2;
    float k[3][4], pv[4];
    load_keys3(a.pos.keys, bh, N, n0, k);
#pragma unroll
    for (int i = 0; i < 4; ++i) pv[i] = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * N + n0 + i) : 1.0f;
    for (int cq = 0; cq < (cc >> 2); ++cq) {
      const float4* Tq = T4 + (size_t)cq * G;
      float o[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Pt3 p;
        pt3_from_keys(k[0][i], k[1][i], k[2][i], g, p);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int hv = 0; hv < 2; ++hv) {
          float4 cv[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) cv[v] = Tq[p.base + off[hv * 4 + v]];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const float w = p.cw[hv * 4 + v];
            // the reference's sum over corners in corner order (the first product initialises the sum)
            if (hv == 0 && v == 0) { s0 = cv[0].x * w; s1 = cv[0].y * w; s2 = cv[0].z * w; s3 = cv[0].w * w; }
            else { s0 += cv[v].x * w; s1 += cv[v].y * w; s2 += cv[v].z * w; s3 += cv[v].w * w; }
          }
        }
        o[0][i] = HAS_PAD ? s0 * pv[i] : s0;
        o[1][i] = HAS_PAD ? s1 * pv[i] : s1;
        o[2][i] = HAS_PAD ? s2 * pv[i] : s2;
        o[3][i] = HAS_PAD ? s3 * pv[i] : s3;
        asm volatile("" : "+v"(o[0][i]), "+v"(o[1][i]), "+v"(o[2][i]), "+v"(o[3][i]));
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int cj = 0; cj < 4; ++cj)
        st_stream4(dst + (size_t)(cq * 4 + cj) * N + n0, make_float4(o[cj][0], o[cj][1], o[cj][2], o[cj][3]));
    }
  }
}
