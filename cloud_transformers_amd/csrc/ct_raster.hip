// Differentiable rasterize (Splat) / de-rasterize (Slice) for gfx950.
//
// Generic kernels: any dim in {2,3}, any extents, any C, optional padding mask,
// corners either recomputed from keys (fused hot path) or read from explicit
// (local_coord, flat_idx) tensors (API-compatible path).
//
// Work decomposition: one workgroup owns one (b, h, channel-chunk) tile of the
// grid and keeps it in LDS, so that the random-access part of the op (atomic
// max / atomic add for the scatter passes, corner reads for the gather passes)
// never leaves the CU; HBM only sees coalesced streams of keys / features along
// N and one linear copy of the tile.  Grids whose single-channel tile exceeds
// the LDS budget fall back to global atomics / global gathers.
//
// Math spec: SURVEY.md Appendix A (derived from layers/cloud_transform.py:72-227
// and layers/utils.py:100-186 of the reference).
#include "ct_common.h"
#include <string.h>
#include <type_traits>
#include <atomic>
#include <mutex>

#ifndef CT_QUAD_THREADS
#define CT_QUAD_THREADS 512
#endif
#ifndef CT_QUAD_WAVES
#define CT_QUAD_WAVES 4
#endif
#ifndef CT_QUAD_CG
#define CT_QUAD_CG 4
#endif

namespace {

#include "ct_raster_args.h"

// (M, K) of channel `ch` of plane bh as the scatter kernels need them: from the partial statistics (a.stats) or from the
// two slot words at the head of the channel's output tile
__device__ __forceinline__ float stats_MK(const RasterArgs& a, size_t bh, int ch, const float* tile, int G) {
  if (a.stats != nullptr) {
    const size_t per = (size_t)a.B * a.H * a.C;
    unsigned m = 0u, k = 0u;
    for (int sp = 0; sp < a.nsplit; ++sp) {
      const unsigned* w = a.stats + ((size_t)sp * per + bh * a.C + ch) * 2;
      m = max(m, w[0]);      // non-negative floats order like their bit patterns
      k += w[1];
    }
    return __uint_as_float(m) * (float)k;
  }
  const unsigned* slot = (const unsigned*)(tile + (size_t)ch * G);
  return __uint_as_float(slot[0]) * (float)slot[1];
}

template <int DIM, bool FROM_KEYS>
struct PointPos {
  float w0[DIM], w1[DIM];
  float mask[DIM];
};

// Loads one point's corners. bh = b*H + h.
template <int DIM, bool FROM_KEYS>
__device__ __forceinline__ void load_point(const PosSrc& P, const GridW<DIM>& g, size_t bh, int N, int n,
                                           Corners<DIM>& c, PointPos<DIM, FROM_KEYS>& pp) {
  constexpr int V = 1 << DIM;
  if constexpr (FROM_KEYS) {
    int f[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
      float k = P.keys[(bh * DIM + j) * N + n];
      ct_axis(k, g.hw[j], g.W[j], pp.w0[j], pp.w1[j], f[j]);
      pp.mask[j] = ct_key_mask(k);
    }
    ct_corners<DIM>(pp.w0, pp.w1, f, g, c);
  } else {
#pragma unroll
    for (int v = 0; v < V; ++v) {
      c.w[v] = P.lc[(bh * V + v) * N + n];
      c.cell[v] = (int)P.idx[(bh * V + v) * N + n];
    }
  }
}

// Writes / accumulates one point's position cotangent.
template <int DIM, bool FROM_KEYS>
__device__ __forceinline__ void store_gpos(float* g_pos, size_t bh, int N, int n,
                                           const PointPos<DIM, FROM_KEYS>& pp, const float (&gw)[1 << DIM],
                                           bool first, bool atomic) {
  constexpr int V = 1 << DIM;
  if constexpr (FROM_KEYS) {
    float gs[DIM];
    ct_corner_grad<DIM>(pp.w0, pp.w1, gw, gs);
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
      float val = gs[j] * pp.mask[j];
      float* p = g_pos + (bh * DIM + j) * N + n;
      if (atomic) atomicAdd(p, val);
      else if (first) *p = val;
      else *p += val;
    }
  } else {
#pragma unroll
    for (int v = 0; v < V; ++v) {
      float* p = g_pos + (bh * V + v) * N + n;
      if (atomic) atomicAdd(p, gw[v]);
      else if (first) *p = gw[v];
      else *p += gw[v];
    }
  }
}

__device__ __forceinline__ void lds_fill_zero(float* tile, int count) {
  for (int i = threadIdx.x; i < count; i += blockDim.x) tile[i] = 0.0f;
}

__device__ __forceinline__ void copy_linear(float* __restrict__ dst, const float* __restrict__ src, int count) {
  // both sides are 16-byte aligned whenever count % 4 == 0 (tiles start at multiples of G*CC)
  if ((count & 3) == 0 && ((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0) {
    const float4* s4 = (const float4*)src;
    float4* d4 = (float4*)dst;
    for (int i = threadIdx.x; i < (count >> 2); i += blockDim.x) d4[i] = s4[i];
  } else {
    for (int i = threadIdx.x; i < count; i += blockDim.x) dst[i] = src[i];
  }
}

// Tile staging global -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR round trip,
// every 1-KiB piece of the tile is in flight at once and the following __syncthreads()
// (which waits vmcnt(0)) retires them all — one memory latency per tile instead of one
// per dependent load->ds_write pair.  The LDS destination of a piece is wave-uniform
// base + lane*16, which is exactly a linear copy.
__device__ __forceinline__ void stage_tile(float* __restrict__ lds_dst, const float* __restrict__ gsrc, int count) {
  if ((count & 3) == 0 && (((uintptr_t)gsrc) & 15) == 0) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int pieces = count >> 8;                     // 256 floats = 64 lanes x 16 B
    for (int c = wave; c < pieces; c += nw) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (size_t)c * 256 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(lds_dst + c * 256), 16, 0, 0);
    }
    for (int i = (pieces << 8) + threadIdx.x; i < count; i += blockDim.x) lds_dst[i] = gsrc[i];
  } else {
    for (int i = threadIdx.x; i < count; i += blockDim.x) lds_dst[i] = gsrc[i];
  }
}

// ---------------------------------------------------------------------------
// K1: scatter pass.  tile_out[(b,h,c), cell_v(n)] (max0|+)= (src[(b,h,c), n] * pad[b,n]) * w_v(n)
//   Splat forward (both reduce modes) and the g_grid half of Slice backward
//   (a scatter-add of g_out, layers/cloud_transform.py:216-221 autograd).
//   grid = (nchunks, H, B).  LDS_TILE=false: tile_out pre-zeroed, global atomics.
// ---------------------------------------------------------------------------
template <int DIM, bool FROM_KEYS, bool SUM, bool LDS_TILE>
__global__ void __launch_bounds__(1024, DIM == 2 ? 8 : 4) scatter_kernel(RasterArgs a, GridW<DIM> g) {
  constexpr int V = 1 << DIM;
  extern __shared__ __align__(16) float lds[];
  const BlockXHB blk = block_xhb();
  const int chunk = blk.x, h = blk.h, b = blk.b;
  const size_t bh = (size_t)b * a.H + h;
  const int c0 = chunk * a.CC;
  const int cc = min(a.CC, a.C - c0);
  float* gout = a.tile_out + (bh * a.C + c0) * (size_t)g.G;
  float* T = LDS_TILE ? lds : gout;
  if (LDS_TILE) {
    lds_fill_zero(lds, cc * g.G);
    __syncthreads();
  }
  const float* src = a.src + (bh * a.C + c0) * (size_t)a.N;
  for (int n = threadIdx.x; n < a.N; n += blockDim.x) {
    Corners<DIM> c;
    PointPos<DIM, FROM_KEYS> pp;
    load_point<DIM, FROM_KEYS>(a.pos, g, bh, a.N, n, c, pp);
    const bool has_pad = a.pad_dtype != CT_PAD_NONE;
    const float p = ct_load_pad(a.pad, a.pad_dtype, (size_t)b * a.N + n);
    for (int ch = 0; ch < cc; ++ch) {
      float f = src[(size_t)ch * a.N + n];
      if (has_pad) f = f * p;
      float* Tc = T + (size_t)ch * g.G;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        float prod = f * c.w[v];
        if (SUM) {
          atomicAdd(&Tc[c.cell[v]], prod);
        } else {
          // zero floor: only positive products can win, and positive IEEE-754
          // floats order like their bit patterns
          if (prod > 0.0f) atomicMax((unsigned*)&Tc[c.cell[v]], __float_as_uint(prod));
        }
      }
    }
  }
  if (LDS_TILE) {
    __syncthreads();
    copy_linear(gout, lds, cc * g.G);
  }
}

// ---------------------------------------------------------------------------
// K1s: scatter-ADD pass in 32-bit fixed point (LDS tile only).
//   gfx950's LDS float atomic add (ds_add_f32) retires ~1 lane per 3 clocks
//   (81 ns per wave-instruction, measured: tools/microbench/lds_atomics.hip)
//   while the integer LDS atomics run at the ds_write rate (2-3 ns).  So the
//   tile is accumulated as int32 multiples of a per-tile power-of-two quantum
//        q = 2^ceil(log2(M*K)) / 2^30,   M = max |src*pad| of the slab,
//                                       K = max contributions per cell,
//   which cannot overflow (|sum| <= K*M) and resolves M*K*2^-30 (<= fp32 eps of
//   the largest possible sum).  Every product is formed in fp32 exactly as the
//   reference does ((src*pad)*w), rounded once to the quantum, and integer adds
//   commute: the result is BITWISE REPRODUCIBLE, unlike a float scatter-add.
//   Slabs with non-finite values fall back to float atomics.
//   grid = (nchunks, H, B).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float block_max(float v, float* red) {
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, red[w]);
  return r;
}

template <int DIM, bool FROM_KEYS>
__global__ void __launch_bounds__(1024, DIM == 2 ? 8 : 4) scatter_add_fx_kernel(RasterArgs a, GridW<DIM> g) {
  constexpr int V = 1 << DIM;
  extern __shared__ __align__(16) float lds[];
  __shared__ float red[16];
  int* acc = (int*)lds;
  float* s_q = lds + (size_t)a.CC * g.G;      // [CC] per-channel max, then quantum (< 0: float atomics)
  float* s_iq = s_q + a.CC;                   // [CC] 1 / quantum
  const BlockXHB blk = block_xhb();
  const int chunk = blk.x, h = blk.h, b = blk.b;
  const size_t bh = (size_t)b * a.H + h;
  const int c0 = chunk * a.CC;
  const int cc = min(a.CC, a.C - c0);
  float* gout = a.tile_out + (bh * a.C + c0) * (size_t)g.G;
  const float* src = a.src + (bh * a.C + c0) * (size_t)a.N;
  const bool has_pad = a.pad_dtype != CT_PAD_NONE;
  lds_fill_zero(lds, cc * g.G);
  __syncthreads();
  // phase 1: contributions per cell (counted in the first channel's tile) ...
  for (int n = threadIdx.x; n < a.N; n += blockDim.x) {
    Corners<DIM> c;
    PointPos<DIM, FROM_KEYS> pp;
    load_point<DIM, FROM_KEYS>(a.pos, g, bh, a.N, n, c, pp);
#pragma unroll
    for (int v = 0; v < V; ++v) atomicAdd(&acc[c.cell[v]], 1);
  }
  __syncthreads();
  float k = 0.0f;
  for (int i = threadIdx.x; i < g.G; i += blockDim.x) k = fmaxf(k, (float)acc[i]);
  const float K = block_max(k, red);
  for (int i = threadIdx.x; i < g.G; i += blockDim.x) acc[i] = 0;
  // ... and max |src*pad| PER CHANNEL: the quantum of a channel follows that channel's own magnitude (channels of one
  // head may differ by orders of magnitude; a shared quantum would cost the quiet ones their precision)
  for (int ch = 0; ch < cc; ++ch) {
    float m = 0.0f;
    for (int n = threadIdx.x; n < a.N; n += blockDim.x) {
      float f = src[(size_t)ch * a.N + n];
      if (has_pad) f = f * ct_load_pad(a.pad, a.pad_dtype, (size_t)b * a.N + n);
      const float af = fabsf(f);
      m = fmaxf(m, (af < __builtin_inff()) ? af : __builtin_inff());   // inf / NaN -> inf
    }
    const float M = block_max(m, red);
    if (threadIdx.x == 0) s_q[ch] = M;
  }
  __syncthreads();
  // s_q[ch] := quantum (negative: float atomics for that channel)
  for (int ch = threadIdx.x; ch < cc; ch += blockDim.x) {
    const float MK = s_q[ch] * K;
    const bool fixed = MK < 1e37f;
    int ex = 0;
    if (fixed && MK > 0.0f) (void)frexpf(MK, &ex);   // MK <= 2^ex
    ex = max(ex, -90);
    s_q[ch] = fixed ? ldexpf(1.0f, ex - 30) : -1.0f;
    s_iq[ch] = ldexpf(1.0f, 30 - ex);
  }
  __syncthreads();
  for (int n = threadIdx.x; n < a.N; n += blockDim.x) {
    Corners<DIM> c;
    PointPos<DIM, FROM_KEYS> pp;
    load_point<DIM, FROM_KEYS>(a.pos, g, bh, a.N, n, c, pp);
    const float p = ct_load_pad(a.pad, a.pad_dtype, (size_t)b * a.N + n);
    for (int ch = 0; ch < cc; ++ch) {
      // (LDS returns in order: the quantum reads wait for the previous channel's atomics — issued before the HBM load,
      //  that wait hides behind it)
      const float q = s_q[ch], iqv = s_iq[ch];
      float f = src[(size_t)ch * a.N + n];
      if (has_pad) f = f * p;
      int* Tc = acc + (size_t)ch * g.G;
      if (q >= 0.0f) {          // block-uniform
        const float fq = f * iqv;             // a power of two: exact
#pragma unroll
        for (int v = 0; v < V; ++v) atomicAdd(&Tc[c.cell[v]], __float2int_rn(fq * c.w[v]));
      } else {                  // non-finite or astronomically large channel: plain float atomics keep IEEE semantics
#pragma unroll
        for (int v = 0; v < V; ++v) atomicAdd((float*)&Tc[c.cell[v]], f * c.w[v]);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < cc * g.G; i += blockDim.x) {
    const float q = s_q[i / g.G];
    gout[i] = q >= 0.0f ? (float)acc[i] * q : lds[i];
  }
}

// ---------------------------------------------------------------------------
// K1t: streaming form of K1s for Slice backward: the per-channel max |src*pad| and the
//   per-plane max contributions per cell were left in the first two words of each channel's
//   output tile by quad_kernel<..., QM_GATHER_GW, STATS> (which reads the same src anyway),
//   so the quantum is known up front and the pass is a plain one-point-per-thread stream
//   with integer LDS atomics — the structure of the max scatter, which runs at the HBM rate.
//   grid = (nchunks, H, B)
// ---------------------------------------------------------------------------
template <int DIM, bool FROM_KEYS>
__global__ void __launch_bounds__(1024, DIM == 2 ? 8 : 4) scatter_add_fx_stream_kernel(RasterArgs a, GridW<DIM> g) {
  constexpr int V = 1 << DIM;
  extern __shared__ __align__(16) float lds[];
  int* acc = (int*)lds;
  float* s_q = lds + (size_t)a.CC * g.G;      // [CC] quantum (< 0: float fallback)
  float* s_iq = s_q + a.CC;                   // [CC] 1 / quantum
  const BlockXHB blk = block_xhb();
  const int chunk = blk.x, h = blk.h, b = blk.b;
  const size_t bh = (size_t)b * a.H + h;
  const int c0 = chunk * a.CC;
  const int cc = min(a.CC, a.C - c0);
  float* gout = a.tile_out + (bh * a.C + c0) * (size_t)g.G;
  const float* src = a.src + (bh * a.C + c0) * (size_t)a.N;
  const bool has_pad = a.pad_dtype != CT_PAD_NONE;
  // One quantum PER CHANNEL (its own max |src*pad| from its slot, K of the plane): channels of a head may differ by
  // orders of magnitude, and a shared quantum would cost the quiet ones their precision.
  for (int ch = threadIdx.x; ch < cc; ch += blockDim.x) {
    const float MK = stats_MK(a, bh, c0 + ch, a.tile_out + (bh * a.C) * (size_t)g.G, g.G);
    const bool fixed = MK < 1e37f;
    int ex = 0;
    if (fixed && MK > 0.0f) (void)frexpf(MK, &ex);
    ex = max(ex, -90);
    s_q[ch] = fixed ? ldexpf(1.0f, ex - 30) : -1.0f;
    s_iq[ch] = ldexpf(1.0f, 30 - ex);
  }
  __syncthreads();      // slots are read before the tile is touched
  for (int i = threadIdx.x; i < cc * g.G; i += blockDim.x) acc[i] = 0;
  __syncthreads();
  for (int n = threadIdx.x; n < a.N; n += blockDim.x) {
    Corners<DIM> c;
    PointPos<DIM, FROM_KEYS> pp;
    load_point<DIM, FROM_KEYS>(a.pos, g, bh, a.N, n, c, pp);
    const float p = ct_load_pad(a.pad, a.pad_dtype, (size_t)b * a.N + n);
    // channels in groups of 4: the group's loads are issued together, then its 16/32 atomics
    for (int c4 = 0; c4 < cc; c4 += 4) {
      float f[4], qg[4], iqg[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {          // quanta first: see scatter_quad_kernel
        qg[u] = s_q[min(c4 + u, cc - 1)];
        iqg[u] = s_iq[min(c4 + u, cc - 1)];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) f[u] = (c4 + u < cc) ? src[(size_t)(c4 + u) * a.N + n] : 0.0f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ch = c4 + u;
        if (ch < cc) {
          const float fu = has_pad ? f[u] * p : f[u];
          int* Tc = acc + (size_t)ch * g.G;
          if (qg[u] >= 0.0f) {              // block-uniform
            const float fq = fu * iqg[u];    // a power of two: exact
#pragma unroll
            for (int v = 0; v < V; ++v) atomicAdd(&Tc[c.cell[v]], __float2int_rn(fq * c.w[v]));
          } else {
#pragma unroll
            for (int v = 0; v < V; ++v) atomicAdd((float*)&Tc[c.cell[v]], fu * c.w[v]);
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < cc * g.G; i += blockDim.x) {
    const float q = s_q[i / g.G];
    gout[i] = q >= 0.0f ? (float)acc[i] * q : lds[i];
  }
}

// ---------------------------------------------------------------------------
// K1q: quad form of the two streaming scatters of the hot path (corners from keys, N % 4 == 0,
//   16-byte aligned rows, G % 4 == 0): Splat(max0) forward and the fixed-point scatter-add of
//   Slice backward (quantum from the slots, as K1t).  Each thread owns 4 consecutive points; keys
//   and src move as dwordx4 along N and the loads of a group of CG channels are issued before the
//   group's LDS atomics, so a wave keeps CG KiB of HBM reads in flight instead of CG*256 B.
//   grid = (nchunks, H, B)
// ---------------------------------------------------------------------------
#ifndef CT_SCATTER3_CUBES
#define CT_SCATTER3_CUBES 1      // the 3D Splat forward instantiated for the zoo's 8^3 / 16^3 grids
#endif
template <int DIM, bool ADD, bool HAS_PAD, int WT = 0>
__global__ void __launch_bounds__(DIM == 2 ? 1024 : 512, DIM == 2 ? 8 : 4) scatter_quad_kernel(RasterArgs a, GridW<DIM> g_arg) {
  GridW<DIM> g = g_arg;
  if constexpr (DIM == 2 && WT > 0) g = grid2_of<WT>(g_arg);      // (the headline's 32 x 32: the grid's constants folded)
  if constexpr (DIM == 3 && WT > 0) g = grid3_of<WT>(g_arg);      // (the zoo's 8^3 and 16^3)
  constexpr int V = 1 << DIM;
  constexpr int CG = DIM == 2 ? 4 : 2;
  extern __shared__ __align__(16) float lds[];
  int* acc = (int*)lds;
  const BlockXHB blk = block_xhb();
  const int chunk = blk.x, h = blk.h, b = blk.b;
  const size_t bh = (size_t)b * a.H + h;
  const int c0 = chunk * a.CC;
  const int cc = min(a.CC, a.C - c0);
  float* gout = a.tile_out + (bh * a.C + c0) * (size_t)g.G;
  const float* src = a.src + (bh * a.C + c0) * (size_t)a.N;
  float* s_q = lds + (size_t)a.CC * g.G;      // ADD: [CC] quantum per channel (< 0: float atomics), [CC] its inverse
  float* s_iq = s_q + a.CC;
  if (ADD) {
    // per channel: its own max |src*pad| (slot 0) and the plane's max contributions per cell (slot 1)
    for (int ch = threadIdx.x; ch < cc; ch += blockDim.x) {
      const float MK = stats_MK(a, bh, c0 + ch, a.tile_out + (bh * a.C) * (size_t)g.G, g.G);
      const bool fx = MK < 1e37f;
      int ex = 0;
      if (fx && MK > 0.0f) (void)frexpf(MK, &ex);
      ex = max(ex, -90);
      s_q[ch] = fx ? ldexpf(1.0f, ex - 30) : -1.0f;
      s_iq[ch] = fx ? ldexpf(1.0f, 30 - ex) : 1.0f;      // float fallback: values stay unscaled
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < (cc * g.G) >> 2; i += blockDim.x) ((int4*)acc)[i] = make_int4(0, 0, 0, 0);
  __syncthreads();
  // cell offsets of the V corners relative to the base cell
  Corners<DIM> off;
  {
    const float one[DIM] = {};
    int f0[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) f0[j] = 0;
    ct_corners<DIM>(one, one, f0, g, off);
  }
  const int nq = a.N >> 2;
  for (int qd = threadIdx.x; qd < nq; qd += blockDim.x) {
    const int n0 = qd << 2;
    float cw[4][V];
    int base[4];
    {
      float kk[DIM][4];
#pragma unroll
      for (int j = 0; j < DIM; ++j) {
        const float4 t = *(const float4*)(a.pos.keys + (bh * DIM + j) * a.N + n0);
        kk[j][0] = t.x; kk[j][1] = t.y; kk[j][2] = t.z; kk[j][3] = t.w;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float w0[DIM], w1[DIM];
        int f[DIM];
#pragma unroll
        for (int j = 0; j < DIM; ++j) ct_axis(kk[j][i], g.hw[j], g.W[j], w0[j], w1[j], f[j]);
        Corners<DIM> c;
        ct_corners<DIM>(w0, w1, f, g, c);
        base[i] = c.cell[0];
#pragma unroll
        for (int v = 0; v < V; ++v) cw[i][v] = c.w[v];
      }
    }
    float pv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pv[i] = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * a.N + n0 + i) : 1.0f;
    for (int cg0 = 0; cg0 < cc; cg0 += CG) {
      // the group's quanta are read from LDS HERE, together with the group's HBM loads: LDS returns in order, so a read
      // placed between the channels' atomics would wait for every atomic issued before it (measured: the scatter of the
      // 64^2 C16 B2 head 17 -> 32 us)
      float iqg[CG];
      bool fxg[CG];
#pragma unroll
      for (int cj = 0; cj < CG; ++cj) {
        const int ch = min(cg0 + cj, cc - 1);
        iqg[cj] = ADD ? s_iq[ch] : 1.0f;
        fxg[cj] = !ADD || s_q[ch] >= 0.0f;                // block-uniform
      }
      float fv[CG][4];
#pragma unroll
      for (int cj = 0; cj < CG; ++cj) {
        const float4 t = (cg0 + cj < cc) ? *(const float4*)(src + (size_t)(cg0 + cj) * a.N + n0) : make_float4(0, 0, 0, 0);
        fv[cj][0] = t.x; fv[cj][1] = t.y; fv[cj][2] = t.z; fv[cj][3] = t.w;
      }
#pragma unroll
      for (int cj = 0; cj < CG; ++cj) {
        const int ch = cg0 + cj;
        if (ch < cc) {
          int* Tc = acc + (size_t)ch * g.G;
          const float iqc = iqg[cj];
          const bool fixed = fxg[cj];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float f = HAS_PAD ? fv[cj][i] * pv[i] : fv[cj][i];
            if (ADD) f = f * iqc;                          // power-of-two scale: exact
#pragma unroll
            for (int v = 0; v < V; ++v) {
              const float prod = f * cw[i][v];
              int* cell = Tc + base[i] + off.cell[v];
              if (!ADD) {
                // zero floor: only positive products can win, and positive IEEE-754 floats order like their bit patterns
                if (prod > 0.0f) atomicMax((unsigned*)cell, __float_as_uint(prod));
              } else if (fixed) {     // block-uniform
                atomicAdd(cell, __float2int_rn(prod));
              } else {
                atomicAdd((float*)cell, prod);
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (cc * g.G) >> 2; i += blockDim.x) {
    const float q = ADD ? s_q[(i << 2) / g.G] : -1.0f;     // G % 4 == 0: a float4 never straddles channels
    if (ADD && q >= 0.0f) {
      const int4 t = ((const int4*)acc)[i];
      ((float4*)gout)[i] = make_float4((float)t.x * q, (float)t.y * q, (float)t.z * q, (float)t.w * q);
    } else {
      ((float4*)gout)[i] = ((const float4*)lds)[i];
    }
  }
}

// ---------------------------------------------------------------------------
// K1r: register-resident form of K1s for the hot path (corners from keys,
//   N <= blockDim*PPT, N % 4 == 0): each thread owns PPT consecutive points and
//   keeps their src values for the whole channel chunk (<= CCR channels) in
//   registers, so the slab is read from HBM exactly once (dwordx4 along N).
//   The chunk is processed in groups of CG channels, each with its own quantum:
//   the atomics of group k run while the loads of groups k+1.. are still in
//   flight (vmcnt is in order), instead of after the whole slab has landed.
// ---------------------------------------------------------------------------
#ifndef CT_FXREG_WAVES
#define CT_FXREG_WAVES 8
#endif
#ifndef CT_FXREG_CG
#define CT_FXREG_CG 4
#endif
template <int DIM, int PPT, int CCR, int CG>
__global__ void __launch_bounds__(1024, CT_FXREG_WAVES) scatter_add_fx_reg_kernel(RasterArgs a, GridW<DIM> g) {
  static_assert(PPT == 4, "one float4 per channel row");
  static_assert(CCR % CG == 0, "groups tile the chunk");
  constexpr int V = 1 << DIM;
  constexpr int NG = CCR / CG;
  extern __shared__ __align__(16) float lds[];
  __shared__ float red[16];
  __shared__ float redg[16 * CG];   // per-wave maxima of the CG channels of a group
  __shared__ float qs[CCR];     // quantum per channel; < 0 marks a float (non-finite) channel
  int* acc = (int*)lds;
  const BlockXHB blk = block_xhb();
  const int chunk = blk.x, h = blk.h, b = blk.b;
  const size_t bh = (size_t)b * a.H + h;
  const int c0 = chunk * a.CC;
  const int cc = min(a.CC, a.C - c0);
  float* gout = a.tile_out + (bh * a.C + c0) * (size_t)g.G;
  const float* src = a.src + (bh * a.C + c0) * (size_t)a.N;
  const bool has_pad = a.pad_dtype != CT_PAD_NONE;
  const int n0 = threadIdx.x * PPT;
  const bool active = n0 < a.N;

  // issue every global load first: keys, pad, then the src slab in channel order
  float kv[DIM][PPT];
  float pv[PPT];
  float sv[CCR][PPT];
#pragma unroll
  for (int j = 0; j < DIM; ++j) {
    float4 t = active ? *(const float4*)(a.pos.keys + (bh * DIM + j) * a.N + n0) : make_float4(0, 0, 0, 0);
    kv[j][0] = t.x; kv[j][1] = t.y; kv[j][2] = t.z; kv[j][3] = t.w;
  }
#pragma unroll
  for (int i = 0; i < PPT; ++i) pv[i] = (active && has_pad) ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * a.N + n0 + i) : 1.0f;
#pragma unroll
  for (int ch = 0; ch < CCR; ++ch) {
    float4 t = (active && ch < cc) ? *(const float4*)(src + (size_t)ch * a.N + n0) : make_float4(0, 0, 0, 0);
    sv[ch][0] = t.x; sv[ch][1] = t.y; sv[ch][2] = t.z; sv[ch][3] = t.w;
  }
  for (int i = threadIdx.x; i < (cc * g.G) >> 2; i += blockDim.x) ((int4*)acc)[i] = make_int4(0, 0, 0, 0);
  for (int i = ((cc * g.G) & ~3) + threadIdx.x; i < cc * g.G; i += blockDim.x) acc[i] = 0;
  __syncthreads();

  // contributions per cell (counted in channel 0's tile) — needs the keys only
  if (active) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      float w0[DIM], w1[DIM];
      int f[DIM];
#pragma unroll
      for (int j = 0; j < DIM; ++j) ct_axis(kv[j][i], g.hw[j], g.W[j], w0[j], w1[j], f[j]);
      Corners<DIM> c;
      ct_corners<DIM>(w0, w1, f, g, c);
#pragma unroll
      for (int v = 0; v < V; ++v) atomicAdd(&acc[c.cell[v]], 1);
    }
  }
  __syncthreads();
  float k = 0.0f;
  for (int i = threadIdx.x; i < g.G; i += blockDim.x) k = fmaxf(k, (float)acc[i]);
  const float K = block_max(k, red);
  for (int i = threadIdx.x; i < g.G; i += blockDim.x) acc[i] = 0;
  // (the barrier inside the first block_max below orders this re-zeroing before any atomic)

#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    if (gi * CG < cc) {     // block-uniform
      // max |src*pad| PER CHANNEL of the group (inf / NaN -> inf): wave reduction, then across the waves through LDS
      float m[CG];
#pragma unroll
      for (int cj = 0; cj < CG; ++cj) {
        m[cj] = 0.0f;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
          float x = sv[gi * CG + cj][i];
          if (has_pad) x = x * pv[i];
          sv[gi * CG + cj][i] = x;
          const float af = fabsf(x);
          m[cj] = fmaxf(m[cj], (af < __builtin_inff()) ? af : __builtin_inff());
        }
        for (int off = 32; off > 0; off >>= 1) m[cj] = fmaxf(m[cj], __shfl_xor(m[cj], off, 64));
      }
      __syncthreads();
      if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int cj = 0; cj < CG; ++cj) redg[(threadIdx.x >> 6) * CG + cj] = m[cj];
      }
      __syncthreads();
      float inv_q[CG];
      bool fixed[CG];
#pragma unroll
      for (int cj = 0; cj < CG; ++cj) {
        float M = redg[cj];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) M = fmaxf(M, redg[w * CG + cj]);
        const float MK = M * K;
        fixed[cj] = MK < 1e37f;
        int ex = 0;
        if (fixed[cj] && MK > 0.0f) (void)frexpf(MK, &ex);   // MK <= 2^ex
        ex = max(ex, -90);
        inv_q[cj] = ldexpf(1.0f, 30 - ex);
        if (threadIdx.x == 0) qs[gi * CG + cj] = fixed[cj] ? ldexpf(1.0f, ex - 30) : -1.0f;
      }
      if (active) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
          float w0[DIM], w1[DIM];
          int f[DIM];
#pragma unroll
          for (int j = 0; j < DIM; ++j) ct_axis(kv[j][i], g.hw[j], g.W[j], w0[j], w1[j], f[j]);
          Corners<DIM> c;
          ct_corners<DIM>(w0, w1, f, g, c);
#pragma unroll
          for (int cj = 0; cj < CG; ++cj) {
            const int ch = gi * CG + cj;
            if (ch < cc) {
              if (fixed[cj]) {      // block-uniform
                const float fq = sv[ch][i] * inv_q[cj];      // a power of two: exact
#pragma unroll
                for (int v = 0; v < V; ++v) atomicAdd(&acc[ch * g.G + c.cell[v]], __float2int_rn(fq * c.w[v]));
              } else {              // non-finite channel: IEEE semantics
#pragma unroll
                for (int v = 0; v < V; ++v) atomicAdd(&lds[ch * g.G + c.cell[v]], sv[ch][i] * c.w[v]);
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (cc * g.G) >> 2; i += blockDim.x) {
    const float q = qs[(i << 2) / g.G];   // G % 4 == 0: a float4 never straddles channels
    if (q < 0.0f) {
      ((float4*)gout)[i] = ((const float4*)lds)[i];
    } else {
      int4 t = ((const int4*)acc)[i];
      ((float4*)gout)[i] = make_float4((float)t.x * q, (float)t.y * q, (float)t.z * q, (float)t.w * q);
    }
  }
}

// ---------------------------------------------------------------------------
// K2: gather pass.  dst[(b,h,c), n] = (sum_v tile_in[(b,h,c), cell_v(n)] * w_v(n)) * pad[b,n]
//   Slice forward; also the g_feat half of Splat(sum) backward.
//   grid = (nchunks * nsplit, H, B)
// ---------------------------------------------------------------------------
template <int DIM, bool FROM_KEYS, bool LDS_TILE>
__global__ void __launch_bounds__(1024, DIM == 2 ? 8 : 4) gather_kernel(RasterArgs a, GridW<DIM> g) {
  constexpr int V = 1 << DIM;
  extern __shared__ __align__(16) float lds[];
  const BlockXHB blk = block_xhb();
  const int chunk = blk.x / a.nsplit, sp = blk.x % a.nsplit;
  const int h = blk.h, b = blk.b;
  const size_t bh = (size_t)b * a.H + h;
  const int c0 = chunk * a.CC;
  const int cc = min(a.CC, a.C - c0);
  const float* gin = a.tile_in + (bh * a.C + c0) * (size_t)g.G;
  const float* T = LDS_TILE ? lds : gin;
  if (LDS_TILE) {
    stage_tile(lds, gin, cc * g.G);
    __syncthreads();
  }
  const int per = (a.N + a.nsplit - 1) / a.nsplit;
  const int n_beg = sp * per, n_end = min(a.N, n_beg + per);
  float* dst = a.dst + (bh * a.C + c0) * (size_t)a.N;
  for (int n = n_beg + threadIdx.x; n < n_end; n += blockDim.x) {
    Corners<DIM> c;
    PointPos<DIM, FROM_KEYS> pp;
    load_point<DIM, FROM_KEYS>(a.pos, g, bh, a.N, n, c, pp);
    const bool has_pad = a.pad_dtype != CT_PAD_NONE;
    const float p = ct_load_pad(a.pad, a.pad_dtype, (size_t)b * a.N + n);
    for (int ch = 0; ch < cc; ++ch) {
      const float* Tc = T + (size_t)ch * g.G;
      float acc = Tc[c.cell[0]] * c.w[0];
#pragma unroll
      for (int v = 1; v < V; ++v) acc += Tc[c.cell[v]] * c.w[v];
      if (has_pad) acc = acc * p;
      dst[(size_t)ch * a.N + n] = acc;
    }
  }
}

// ---------------------------------------------------------------------------
// K4: corner-cotangent pass of a gather:  gw[v,n] = sum_c tile_in[(b,h,c), cell_v(n)] * (src[(b,h,c), n] * pad)
//   Slice backward wrt the weights (tile_in = conv output, src = g_out);
//   Splat(sum) backward wrt the weights (tile_in = g_grid, src = feat).
//   Result goes through the positions backward (FROM_KEYS) or to g_local_coord.
//   grid = (ncg * nsplit, H, B); each workgroup loops over its share of chunks.
// ---------------------------------------------------------------------------
template <int DIM, bool FROM_KEYS, bool LDS_TILE>
__global__ void __launch_bounds__(1024, DIM == 2 ? 8 : 4) gather_gw_kernel(RasterArgs a, GridW<DIM> g) {
  constexpr int V = 1 << DIM;
  extern __shared__ __align__(16) float lds[];
  const BlockXHB blk = block_xhb();
  const int cg = blk.x / a.nsplit, sp = blk.x % a.nsplit;
  const int h = blk.h, b = blk.b;
  const size_t bh = (size_t)b * a.H + h;
  const int per = (a.N + a.nsplit - 1) / a.nsplit;
  const int n_beg = sp * per, n_end = min(a.N, n_beg + per);
  const bool atomic = a.atomic_gpos != 0;
  bool first = true;
  for (int chunk = cg; chunk < a.nchunks; chunk += a.ncg) {
    const int c0 = chunk * a.CC;
    const int cc = min(a.CC, a.C - c0);
    const float* gin = a.tile_in + (bh * a.C + c0) * (size_t)g.G;
    const float* T = LDS_TILE ? lds : gin;
    if (LDS_TILE) {
      __syncthreads();
      stage_tile(lds, gin, cc * g.G);
      __syncthreads();
    }
    const float* src = a.src + (bh * a.C + c0) * (size_t)a.N;
    for (int n = n_beg + threadIdx.x; n < n_end; n += blockDim.x) {
      Corners<DIM> c;
      PointPos<DIM, FROM_KEYS> pp;
      load_point<DIM, FROM_KEYS>(a.pos, g, bh, a.N, n, c, pp);
      const bool has_pad = a.pad_dtype != CT_PAD_NONE;
      const float p = ct_load_pad(a.pad, a.pad_dtype, (size_t)b * a.N + n);
      float gw[V];
#pragma unroll
      for (int v = 0; v < V; ++v) gw[v] = 0.0f;
      for (int ch = 0; ch < cc; ++ch) {
        float s = src[(size_t)ch * a.N + n];
        if (has_pad) s = s * p;
        const float* Tc = T + (size_t)ch * g.G;
#pragma unroll
        for (int v = 0; v < V; ++v) gw[v] += Tc[c.cell[v]] * s;
      }
      store_gpos<DIM, FROM_KEYS>(a.g_pos + (size_t)cg * a.gpos_stride, bh, a.N, n, pp, gw, first && !a.accumulate, atomic);
    }
    first = false;
  }
}

// ---------------------------------------------------------------------------
// K5: Splat(max0) backward.  tile_in = z (forward output), tile_in2 = g_z.
//   A contribution (c, v, n) is the winner of its cell iff its product is
//   positive and bit-equal to z[c, cell]; the first such contribution to CLAIM
//   the cell (atomic compare-and-swap of the tile copy to 0) receives g_z, so
//   that exactly one contribution wins even on exact ties (torch_scatter's
//   backward routes the cotangent to a single arg-max element).
//     g_feat[c,n] = pad * sum_v win * g_z[c,cell_v] * w_v
//     gw[v,n]     = sum_c win * g_z[c,cell_v] * feat[c,n]*pad
//   grid = (ncg, H, B); the whole N range stays in one workgroup per chunk so
//   that claims are unique.
// ---------------------------------------------------------------------------
template <int DIM, bool FROM_KEYS, bool LDS_TILE, bool GZ_LDS>
__global__ void __launch_bounds__(1024, DIM == 2 ? 8 : 4) splat_max_bwd_kernel(RasterArgs a, GridW<DIM> g) {
  constexpr int V = 1 << DIM;
  extern __shared__ __align__(16) float lds[];
  const BlockXHB blk = block_xhb();
  const int cg = blk.x;
  const int h = blk.h, b = blk.b;
  const size_t bh = (size_t)b * a.H + h;
  const bool atomic = a.atomic_gpos != 0;
  bool first = true;
  for (int chunk = cg; chunk < a.nchunks; chunk += a.ncg) {
    const int c0 = chunk * a.CC;
    const int cc = min(a.CC, a.C - c0);
    const size_t toff = (bh * a.C + c0) * (size_t)g.G;
    unsigned* T = LDS_TILE ? (unsigned*)lds : (a.claim + toff);
    const float* gz = GZ_LDS ? (lds + (size_t)a.CC * g.G) : (a.tile_in2 + toff);
    if (LDS_TILE) {
      __syncthreads();
      stage_tile(lds, a.tile_in + toff, cc * g.G);
      if (GZ_LDS) stage_tile(lds + (size_t)a.CC * g.G, a.tile_in2 + toff, cc * g.G);
      __syncthreads();
    }
    const float* src = a.src + (bh * a.C + c0) * (size_t)a.N;
    float* dst = a.dst + (bh * a.C + c0) * (size_t)a.N;
    for (int n = threadIdx.x; n < a.N; n += blockDim.x) {
      Corners<DIM> c;
      PointPos<DIM, FROM_KEYS> pp;
      load_point<DIM, FROM_KEYS>(a.pos, g, bh, a.N, n, c, pp);
      const bool has_pad = a.pad_dtype != CT_PAD_NONE;
      const float p = ct_load_pad(a.pad, a.pad_dtype, (size_t)b * a.N + n);
      float gw[V];
#pragma unroll
      for (int v = 0; v < V; ++v) gw[v] = 0.0f;
#pragma unroll 2
      for (int ch = 0; ch < cc; ++ch) {
        float f = src[(size_t)ch * a.N + n];
        if (has_pad) f = f * p;
        unsigned* Tc = T + (size_t)ch * g.G;
        const float* Gc = gz + (size_t)ch * g.G;
        // Branch-free matching; one divergent region per (point, channel) for the rare winners
        // (see quad_kernel<QM_SPLAT_MAX_BWD>): a non-positive product can never be bit-equal to a
        // positive tile value, non-matching corners compare-and-swap against all ones.
        unsigned zb[V], bits[V];
        bool m[V];
        bool any = false;
#pragma unroll
        for (int v = 0; v < V; ++v) zb[v] = Tc[c.cell[v]];
#pragma unroll
        for (int v = 0; v < V; ++v) {
          bits[v] = __float_as_uint(f * c.w[v]);
          m[v] = (zb[v] == bits[v]) & (zb[v] != 0u);
          any = any | m[v];
        }
        float gf = 0.0f;
        if (any) {
          unsigned old[V];
#pragma unroll
          for (int v = 0; v < V; ++v) old[v] = atomicCAS(&Tc[c.cell[v]], m[v] ? bits[v] : 0xFFFFFFFFu, 0u);
#pragma unroll
          for (int v = 0; v < V; ++v) {
            const bool win = m[v] & (old[v] == bits[v]);
            float gzw = 0.0f;
            if (GZ_LDS) gzw = win ? Gc[c.cell[v]] : 0.0f;
            else if (win) gzw = Gc[c.cell[v]];
            gf += gzw * c.w[v];
            gw[v] += gzw * f;
          }
        }
        if (has_pad) gf = gf * p;
        dst[(size_t)ch * a.N + n] = gf;
      }
      store_gpos<DIM, FROM_KEYS>(a.g_pos + (size_t)cg * a.gpos_stride, bh, a.N, n, pp, gw, first && !a.accumulate, atomic);
    }
    first = false;
  }
}

// ---------------------------------------------------------------------------
// KQ: "quad" form of the gather-type passes for the hot path (corners from keys,
//   N % 4 == 0, 16-byte aligned rows).  Each thread owns 4 consecutive points:
//   keys, src and dst move as dwordx4 along N, and the src values of a group of
//   CG channels are fetched into registers BEFORE the LDS work of that group, so
//   a wave keeps CG*1 KiB of HBM reads in flight instead of one dependent dword
//   load per (point, channel).
//     QM_GATHER        : Slice forward / Splat(sum) backward wrt features
//     QM_GATHER_GW     : corner cotangents -> g_keys (Slice backward, Splat(sum) backward)
//     QM_SPLAT_MAX_BWD : Splat(max0) backward (z tile + g_z tile in LDS, single-winner claims)
//   grid = (ncg * nsplit, H, B); each workgroup loops over its chunks.
// ---------------------------------------------------------------------------
enum { QM_GATHER = 0, QM_GATHER_GW = 1, QM_SPLAT_MAX_BWD = 2 };

template <int DIM, int MODE, int CG, int THREADS, bool STATS, bool HAS_PAD>
__global__ void __launch_bounds__(THREADS, THREADS == 1024 ? 4 : CT_QUAD_WAVES) quad_kernel(RasterArgs a, GridW<DIM> g) {
  constexpr int V = 1 << DIM;
  constexpr bool kSrc = MODE != QM_GATHER;          // reads a point-sized input
  constexpr bool kDst = MODE != QM_GATHER_GW;       // writes a point-sized output
  constexpr bool kGw = MODE != QM_GATHER;           // produces g_keys
  extern __shared__ __align__(16) float lds[];
  const BlockXHB blk = block_xhb();
  const int cgi = blk.x / a.nsplit, sp = blk.x % a.nsplit;
  const int h = blk.h, b = blk.b;
  const size_t bh = (size_t)b * a.H + h;
  constexpr bool has_pad = HAS_PAD;            // compile-time: no per-element select on the no-padding path
  const bool atomic = a.atomic_gpos != 0;
  const int nq = a.N >> 2;
  const int per = (nq + a.nsplit - 1) / a.nsplit;
  const int q_beg = sp * per, q_end = min(nq, q_beg + per);
  bool first = true;
  for (int chunk = cgi; chunk < a.nchunks; chunk += a.ncg) {
    const int c0 = chunk * a.CC;
    const int cc = min(a.CC, a.C - c0);
    const size_t toff = (bh * a.C + c0) * (size_t)g.G;
    float* T = lds;                                   // tile_in chunk (z for SPLAT_MAX_BWD)
    float* T2 = lds + (size_t)a.CC * g.G;             // g_z chunk (SPLAT_MAX_BWD only)
    // STATS (Slice backward): by-products for the fixed-point scatter that follows —
    // per-channel max |src*pad| and the max number of contributions per cell
    // the maxima are kept per channel in 32 lane-indexed slots each (one conflict-free ds_max per thread,
    // quad and channel) and folded at the end
    unsigned* s_max = (unsigned*)(lds + (size_t)a.CC * g.G);         // [CC][32]
    int* s_cnt = (int*)(lds + (size_t)a.CC * g.G + a.CC * 32);       // [cnt_mask + 1]
    __syncthreads();
    if (STATS) {
      for (int i = threadIdx.x; i < a.CC * 32 + a.cnt_mask + 1; i += blockDim.x) s_max[i] = 0u;
    }
    stage_tile(T, a.tile_in + toff, cc * g.G);
    if (MODE == QM_SPLAT_MAX_BWD) stage_tile(T2, a.tile_in2 + toff, cc * g.G);
    __syncthreads();
    const float* src = kSrc ? a.src + (bh * a.C + c0) * (size_t)a.N : nullptr;
    float* dst = kDst ? a.dst + (bh * a.C + c0) * (size_t)a.N : nullptr;
    for (int q = q_beg + (int)threadIdx.x; q < q_end; q += blockDim.x) {
      const int n0 = q << 2;
      // corner weights and base cell of the 4 points (per-axis terms are recomputed
      // from the keys for the positions backward, to keep the register footprint low)
      float cw[4][V], pv[4];
      int base[4];
      {
        float kk[DIM][4];
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
          const float4 t = *(const float4*)(a.pos.keys + (bh * DIM + j) * a.N + n0);
          kk[j][0] = t.x; kk[j][1] = t.y; kk[j][2] = t.z; kk[j][3] = t.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float w0[DIM], w1[DIM];
          int f[DIM];
#pragma unroll
          for (int j = 0; j < DIM; ++j) ct_axis(kk[j][i], g.hw[j], g.W[j], w0[j], w1[j], f[j]);
          Corners<DIM> c;
          ct_corners<DIM>(w0, w1, f, g, c);
          base[i] = c.cell[0];
#pragma unroll
          for (int v = 0; v < V; ++v) cw[i][v] = c.w[v];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) pv[i] = has_pad ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * a.N + n0 + i) : 1.0f;
      if (STATS && first) {
        const float one[DIM] = {};
        int f0[DIM];
#pragma unroll
        for (int j = 0; j < DIM; ++j) f0[j] = 0;
        Corners<DIM> off;
        ct_corners<DIM>(one, one, f0, g, off);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int v = 0; v < V; ++v) atomicAdd(&s_cnt[(base[i] + off.cell[v]) & a.cnt_mask], 1);
      }
      float gw[4][V];
      if (kGw) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int v = 0; v < V; ++v) gw[i][v] = 0.0f;
      }
      for (int cg0 = 0; cg0 < cc; cg0 += CG) {
        float fv[CG][4];
        if (kSrc) {
#pragma unroll
          for (int cj = 0; cj < CG; ++cj) {
            const float4 t = (cg0 + cj < cc) ? *(const float4*)(src + (size_t)(cg0 + cj) * a.N + n0) : make_float4(0, 0, 0, 0);
            fv[cj][0] = t.x; fv[cj][1] = t.y; fv[cj][2] = t.z; fv[cj][3] = t.w;
          }
        }
        if (STATS) {
          // max |src*pad| per channel (inf/NaN -> inf)
#pragma unroll
          for (int cj = 0; cj < CG; ++cj) {
            float m = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float x = fabsf(has_pad ? fv[cj][i] * pv[i] : fv[cj][i]);
              m = fmaxf(m, (x < __builtin_inff()) ? x : __builtin_inff());
            }
            if (cg0 + cj < cc) atomicMax(&s_max[(cg0 + cj) * 32 + (threadIdx.x & 31)], __float_as_uint(m));
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          Corners<DIM> c;
          {
            // cell offsets of the V corners relative to the base cell
            const float one[DIM] = {};
            int f0[DIM];
#pragma unroll
            for (int j = 0; j < DIM; ++j) f0[j] = 0;
            Corners<DIM> off;
            ct_corners<DIM>(one, one, f0, g, off);
#pragma unroll
            for (int v = 0; v < V; ++v) {
              c.cell[v] = base[i] + off.cell[v];
              c.w[v] = cw[i][v];
            }
          }
#pragma unroll
          for (int cj = 0; cj < CG; ++cj) {
            const int ch = cg0 + cj;
            if (ch < cc) {
              const float* Tc = T + (size_t)ch * g.G;
              if (MODE == QM_GATHER) {
                float acc = Tc[c.cell[0]] * c.w[0];
#pragma unroll
                for (int v = 1; v < V; ++v) acc += Tc[c.cell[v]] * c.w[v];
                fv[cj][i] = has_pad ? acc * pv[i] : acc;
              } else if (MODE == QM_GATHER_GW) {
                const float sgn = has_pad ? fv[cj][i] * pv[i] : fv[cj][i];
#pragma unroll
                for (int v = 0; v < V; ++v) gw[i][v] += Tc[c.cell[v]] * sgn;
              } else {
                const float f = has_pad ? fv[cj][i] * pv[i] : fv[cj][i];
                unsigned* Zc = (unsigned*)T + (size_t)ch * g.G;
                const float* Gc = T2 + (size_t)ch * g.G;
                // Matching is branch-free; everything a winner needs (claim, g_z, accumulation)
                // sits in ONE divergent region per (point, channel) entered only by lanes with a
                // match on any corner (~22% of them).  Non-matching corners of those lanes issue
                // their compare-and-swap with a pattern no tile value can equal (all ones).
                // (a non-positive product has its sign bit set or is zero, so it can never be
                // bit-equal to a positive tile value: no separate f > 0 test is needed)
                unsigned zb[V], bits[V];
#pragma unroll
                for (int v = 0; v < V; ++v) zb[v] = Zc[c.cell[v]];
                bool m[V];
                bool any = false;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                  bits[v] = __float_as_uint(f * c.w[v]);
                  m[v] = (zb[v] == bits[v]) & (zb[v] != 0u);   // zb != 0 <=> product > 0 here
                  any = any | m[v];
                }
                float gf = 0.0f;
                if (any) {
                  unsigned old[V];
                  float gzv[V];
#pragma unroll
                  for (int v = 0; v < V; ++v) old[v] = atomicCAS(&Zc[c.cell[v]], m[v] ? bits[v] : 0xFFFFFFFFu, 0u);
#pragma unroll
                  for (int v = 0; v < V; ++v) gzv[v] = Gc[c.cell[v]];
#pragma unroll
                  for (int v = 0; v < V; ++v) {
                    const float gzw = (m[v] & (old[v] == bits[v])) ? gzv[v] : 0.0f;
                    gf += gzw * c.w[v];
                    gw[i][v] += gzw * f;
                  }
                }
                fv[cj][i] = has_pad ? gf * pv[i] : gf;
              }
            }
          }
        }
        if (kDst) {
#pragma unroll
          for (int cj = 0; cj < CG; ++cj)
            if (cg0 + cj < cc)
              *(float4*)(dst + (size_t)(cg0 + cj) * a.N + n0) = make_float4(fv[cj][0], fv[cj][1], fv[cj][2], fv[cj][3]);
        }
      }
      if (kGw) {
        float gs[4][DIM];
        {
          float kk[DIM][4];
#pragma unroll
          for (int j = 0; j < DIM; ++j) {
            const float4 t = *(const float4*)(a.pos.keys + (bh * DIM + j) * a.N + n0);
            kk[j][0] = t.x; kk[j][1] = t.y; kk[j][2] = t.z; kk[j][3] = t.w;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float w0[DIM], w1[DIM];
            int f[DIM];
#pragma unroll
            for (int j = 0; j < DIM; ++j) ct_axis(kk[j][i], g.hw[j], g.W[j], w0[j], w1[j], f[j]);
            ct_corner_grad<DIM>(w0, w1, gw[i], gs[i]);
#pragma unroll
            for (int j = 0; j < DIM; ++j) gs[i][j] = gs[i][j] * ct_key_mask(kk[j][i]);
          }
        }
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
          float* pk = a.g_pos + (size_t)cgi * a.gpos_stride + (bh * DIM + j) * a.N + n0;
          if (atomic) {
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(pk + i, gs[i][j]);
          } else {
            float4 o = make_float4(gs[0][j], gs[1][j], gs[2][j], gs[3][j]);
            if (!first || a.accumulate) {
              const float4 prev = *(const float4*)pk;
              o.x += prev.x; o.y += prev.y; o.z += prev.z; o.w += prev.w;
            }
            *(float4*)pk = o;
          }
        }
      }
    }
    if (STATS) {
      // publish into the first two words of each channel's g_grid tile (tile_out), which the
      // scatter kernel reads before it overwrites the tile
      __syncthreads();
      for (int ch = threadIdx.x; ch < cc; ch += blockDim.x) {
        unsigned m = 0u;
        for (int j = 0; j < 32; ++j) m = max(m, s_max[ch * 32 + j]);
        s_max[ch * 32] = m;
      }
      __syncthreads();
      unsigned kmax = 0;
      if (first) {
        for (int i = threadIdx.x; i <= a.cnt_mask; i += blockDim.x) kmax = max(kmax, (unsigned)s_cnt[i]);
        for (int o = 32; o > 0; o >>= 1) kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, o, 64));
      }
      unsigned* slots = (unsigned*)(a.tile_out + toff);
      // K belongs to the (b,h) plane: stored with every channel so that each scatter workgroup finds
      // it inside its own tile.  With one workgroup per plane plain stores do; when N is split over
      // workgroups the slots were zeroed and are combined atomically: max for M, and for K the SUM of
      // the per-split maxima (an upper bound of the true per-cell maximum, which is all K has to be).
      const bool split = a.nsplit > 1;
      // a.stats: this split's own pair per channel (no atomics, nothing to clear beforehand: see RasterArgs::stats)
      unsigned* mine = a.stats != nullptr ? a.stats + ((size_t)sp * a.B * a.H * a.C + bh * a.C) * 2 : nullptr;
      if (first) {
        if ((threadIdx.x & 63) == 0) atomicMax((unsigned*)&s_cnt[0], kmax);   // s_cnt[0] now holds max over waves
        __syncthreads();
        const unsigned kall = (unsigned)s_cnt[0];
        for (int ch = threadIdx.x; ch < a.C; ch += blockDim.x) {
          unsigned* kslot = (unsigned*)(a.tile_out + (bh * a.C + ch) * (size_t)g.G) + 1;
          if (mine != nullptr) mine[2 * ch + 1] = kall;
          else if (split) atomicAdd(kslot, kall);
          else *kslot = kall;
        }
      }
      for (int ch = threadIdx.x; ch < cc; ch += blockDim.x) {
        if (mine != nullptr) mine[2 * (c0 + ch)] = s_max[ch * 32];
        else if (split) atomicMax(slots + (size_t)ch * g.G, s_max[ch * 32]);
        else slots[(size_t)ch * g.G] = s_max[ch * 32];
      }
    }
    first = false;
  }
}

#include "ct_raster_hot.h"
#include "ct_raster_hot3d.h"
CT_KERNARG_IS_ARGS_AND_GRID((splat_max_bwd_hot_kernel<false, 32, 2>), 2);      // (defined in ct_raster_hot.h, ahead of the macro)
CT_KERNARG_IS_ARGS_AND_GRID((splat_max_bwd_hot_kernel<true, 0, 0, kHotWideThreads>), 2);
#include "ct_raster_band.h"
#include "ct_raster_sorted.h"
#include "ct_raster_sorted3d.h"

// ---------------------------------------------------------------------------
// K0: DifferentiablePositions forward / backward (API path only)
// ---------------------------------------------------------------------------
template <int DIM>
__global__ void positions_fwd_kernel(const float* keys, float* lc, long long* idx, int BH, int N, GridW<DIM> g) {
  constexpr int V = 1 << DIM;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)BH * N) return;
  size_t bh = i / N;
  int n = (int)(i % N);
  float w0[DIM], w1[DIM];
  int f[DIM];
#pragma unroll
  for (int j = 0; j < DIM; ++j) ct_axis(keys[(bh * DIM + j) * N + n], g.hw[j], g.W[j], w0[j], w1[j], f[j]);
  Corners<DIM> c;
  ct_corners<DIM>(w0, w1, f, g, c);
#pragma unroll
  for (int v = 0; v < V; ++v) {
    lc[(bh * V + v) * N + n] = c.w[v];
    idx[(bh * V + v) * N + n] = (long long)c.cell[v];
  }
}

template <int DIM>
__global__ void positions_bwd_kernel(const float* keys, const float* g_lc, float* g_keys, int BH, int N, GridW<DIM> g) {
  constexpr int V = 1 << DIM;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)BH * N) return;
  size_t bh = i / N;
  int n = (int)(i % N);
  float w0[DIM], w1[DIM], mask[DIM], gw[V], gs[DIM];
  int f[DIM];
#pragma unroll
  for (int j = 0; j < DIM; ++j) {
    float k = keys[(bh * DIM + j) * N + n];
    ct_axis(k, g.hw[j], g.W[j], w0[j], w1[j], f[j]);
    mask[j] = ct_key_mask(k);
  }
#pragma unroll
  for (int v = 0; v < V; ++v) gw[v] = g_lc[(bh * V + v) * N + n];
  ct_corner_grad<DIM>(w0, w1, gw, gs);
#pragma unroll
  for (int j = 0; j < DIM; ++j) g_keys[(bh * DIM + j) * N + n] = gs[j] * mask[j];
}

__global__ void __launch_bounds__(1024) occupancy_kernel(const float* grid, long long n, unsigned long long* count) {
  // dwordx4 stream, wave + block reduction, ONE atomic per workgroup (thousands of atomics on a
  // single address serialise at ~12 ns each and used to dominate this kernel)
  __shared__ unsigned s_part[16];
  unsigned local = 0;
  const long long n4 = n >> 2;
  const float4* g4 = (const float4*)grid;
  const bool aligned = (((uintptr_t)grid) & 15) == 0;
  if (aligned) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
      const float4 v = g4[i];
      local += (fabsf(v.x) > 1e-9f) + (fabsf(v.y) > 1e-9f) + (fabsf(v.z) > 1e-9f) + (fabsf(v.w) > 1e-9f);
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      local += fabsf(grid[i]) > 1e-9f;
  } else {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      local += fabsf(grid[i]) > 1e-9f;
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_part[w];
    if (t) atomicAdd(count, t);
  }
}

// The occupancy statistic as the blocks report it — count / (B * C * H) as a float — in ONE launch: per-workgroup counts into the
// caller's workspace, the last workgroup to arrive (ticket in word 0, handed back as zero) adds them and writes
// float(count) * inv.  ct_grid_occupancy + `.float()` + `/ K` were a memset node, a kernel and two 5 us elementwise launches per
// head and forward: 96 graph nodes per segmenter step for 24 numbers.
constexpr int kOccBlocks = 256;
__global__ void __launch_bounds__(1024) occupancy_ratio_kernel(const float* grid, long long n, float inv, float* out,
                                                               unsigned long long* ws) {
  __shared__ unsigned s_part[16];
  __shared__ unsigned s_last;
  unsigned local = 0;
  const long long n4 = n >> 2;
  const float4* g4 = (const float4*)grid;
  if ((((uintptr_t)grid) & 15) == 0) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
      const float4 v = g4[i];
      local += (fabsf(v.x) > 1e-9f) + (fabsf(v.y) > 1e-9f) + (fabsf(v.z) > 1e-9f) + (fabsf(v.w) > 1e-9f);
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      local += fabsf(grid[i]) > 1e-9f;
  } else {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      local += fabsf(grid[i]) > 1e-9f;
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_part[w];
    __hip_atomic_store(ws + 1 + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the partial has left before the ticket is taken
    const unsigned old = __hip_atomic_fetch_add((unsigned*)ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = old == gridDim.x - 1u;
    if (old == gridDim.x - 1u) {
      __hip_atomic_store((unsigned*)ws, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (s_last) {           // block-uniform: this workgroup arrived last
    unsigned long long t = 0;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += blockDim.x)
      t += __hip_atomic_load(ws + 1 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned lo = (unsigned)t;              // (a launch counts < 2^32 elements per workgroup slot: n <= 2^31 here)
    for (int off = 32; off > 0; off >>= 1) lo += __shfl_down(lo, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = lo;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long tot = 0;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += s_part[w];
      out[0] = (float)(long long)tot * inv;      // torch: count.float() * (1 / K)  (a float tensor divided by a Python number)
    }
  }
}

// zero the two statistics words at the head of every channel tile
__global__ void zero_slots_kernel(float* tiles, size_t stride, size_t rows) {
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) {
    tiles[r * stride] = 0.0f;
    tiles[r * stride + 1] = 0.0f;
  }
}

// ---------------------------------------------------------------------------
// test hooks: which kernel family an entry point picked, and a switch that keeps the hot-shape kernels off
// (host state, never read by the kernels).  Both are process-wide — a test sets the flags on the main thread and reads the
// tags there, while the backward passes run on autograd's thread — so the flags are atomic and the tag buffer is guarded by
// a mutex (writers: every Splat / Slice entry point; the reader gets a copy in a buffer of its own thread).
// ---------------------------------------------------------------------------
std::atomic<unsigned> t_dbg_flags{0};
std::atomic<int> t_dbg_nseg{0};          // ct_debug_set_nseg: point segments of the hot Splat(max) backward (0: automatic)
std::mutex t_last_mu;
char t_last[256] = "";

void note_reset() {
  std::lock_guard<std::mutex> lk(t_last_mu);
  t_last[0] = 0;
}
void note(const char* tag) {
  std::lock_guard<std::mutex> lk(t_last_mu);
  size_t n = strlen(t_last), m = strlen(tag);
  if (n + m + 2 >= sizeof(t_last)) return;
  if (n) t_last[n++] = '+';
  memcpy(t_last + n, tag, m + 1);
}
inline bool hot_enabled() { return (t_dbg_flags.load(std::memory_order_relaxed) & CT_DEBUG_NO_HOT) == 0; }

// ---------------------------------------------------------------------------
// host-side planning
// ---------------------------------------------------------------------------
struct Plan {
  int CC, nchunks, threads;
  size_t lds_bytes;
  bool lds_tile;
};

bool valid_common(int B, int H, int C, int N, int dim, const int* W) {
  if (B <= 0 || H <= 0 || C <= 0 || N <= 0 || (dim != 2 && dim != 3) || !W) return false;
  long long G = 1;
  for (int j = 0; j < dim; ++j) {
    if (W[j] < 2) return false;
    G *= W[j];
    if (G > (1ll << 30)) return false;
  }
  if (H > 65535 || B > 65535) return false;
  return true;
}

template <int DIM>
GridW<DIM> make_grid(const int* W) {
  GridW<DIM> g;
  g.G = 1;
  for (int j = 0; j < DIM; ++j) {
    g.W[j] = W[j];
    g.hw[j] = (float)(W[j] - 1) * 0.5f;
    g.G *= W[j];
  }
  return g;
}

int round_threads(int n) {
  int t = ((n + 63) / 64) * 64;
  return t < 64 ? 64 : (t > 1024 ? 1024 : t);
}

// Choose channels-per-tile: the largest chunk that fits the LDS budget while
// leaving at least ~2 workgroups per CU chip-wide when C allows it.
Plan make_plan(int B, int H, int C, int N, int G, int tiles_per_wg /*1 or 2 tiles resident*/,
               long long want = 256 /*workgroups to aim for*/, int min_cc = 4 /*do not shrink chunks below this*/) {
  // few (b,h) planes (the decoders: B2 x H16): parallelism matters more than the corners recomputed once per chunk
  // (measured: B2 N16384 zoo heads 10-25 % faster with single-channel chunks, the B8 shapes slower)
  if (min_cc == 4 && (long long)B * H <= 64) min_cc = 1;
  Plan p;
  size_t per_ch = (size_t)G * 4 * tiles_per_wg;
  p.lds_tile = true;
  if (per_ch > (size_t)kBigLdsBytes) {  // not even one channel fits: global path
    p.lds_tile = false;
    p.CC = C;
    p.nchunks = 1;
    p.lds_bytes = 0;
    p.threads = round_threads(N);
    return p;
  }
  int cc_fit = (int)(kMaxLdsBytes / per_ch);
  // tiles that hold fewer than min_cc channels in 64 KiB take the whole CU instead (one workgroup
  // per CU, up to 160 KiB): thin chunks recompute every point's corners once per chunk
  if (cc_fit < min_cc) cc_fit = (int)((size_t)kBigLdsBytes / per_ch) < min_cc ? (int)((size_t)kBigLdsBytes / per_ch) : min_cc;
  if (cc_fit < 1) cc_fit = 1;
  int cc = cc_fit < C ? cc_fit : C;
  // enough workgroups to fill the 256 CUs — but chunks thinner than min_cc channels recompute the
  // corners of every point too often to be worth the extra parallelism
  if (min_cc > cc) min_cc = cc;
  while (cc > min_cc && (long long)B * H * ((C + cc - 1) / cc) < want) cc = (cc + 1) / 2 < min_cc ? min_cc : (cc + 1) / 2;
  p.CC = cc;
  p.nchunks = (C + cc - 1) / cc;
  p.lds_bytes = (size_t)cc * per_ch;
  p.threads = round_threads(N);
  return p;
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (bytes > 48 * 1024) {
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
      return CT_ELAUNCH;
  }
  return CT_OK;
}

#define CT_LAUNCH(KERNEL, GRID, THREADS, LDS, STREAM, ...)                   \
  do {                                                                      \
    if (set_lds(KERNEL, LDS) != CT_OK) return CT_ELAUNCH;                   \
    CT_CLEAR_ERROR();                                                       \
    hipLaunchKernelGGL(KERNEL, GRID, dim3(THREADS), LDS, STREAM, __VA_ARGS__); \
    CT_CHECK_LAUNCH();                                                      \
  } while (0)


// quad kernels are specialised on "has a padding mask" (compile-time: no per-element select
// on the common no-padding path); TARGS = (DIM, MODE, CG, THREADS, STATS)
#define CT_QUAD_T(DIMV, MODE, CG, THREADS, STATS, PAD) quad_kernel<DIMV, MODE, CG, THREADS, STATS, PAD>
#define CT_STRIP(...) __VA_ARGS__
#define CT_LAUNCH_QUAD(TARGS, GRID, NT, LDS, STREAM, ARGS, GW)                                        \
  do {                                                                                                \
    if ((ARGS).pad_dtype != CT_PAD_NONE)                                                              \
      CT_LAUNCH((CT_QUAD_T_APPLY_(CT_STRIP TARGS, true)), GRID, NT, LDS, STREAM, ARGS, GW);           \
    else                                                                                              \
      CT_LAUNCH((CT_QUAD_T_APPLY_(CT_STRIP TARGS, false)), GRID, NT, LDS, STREAM, ARGS, GW);          \
  } while (0)
#define CT_QUAD_T_APPLY_(...) CT_QUAD_T(__VA_ARGS__)

// the quad scatters (4 points per thread, dwordx4) need corners from keys and 16-byte aligned rows
bool scatter_quad_ok(const RasterArgs& a, bool from_keys, int G) {
  if (!from_keys || (a.N & 3) != 0 || (G & 3) != 0) return false;
  return ((((uintptr_t)a.src) | ((uintptr_t)a.pos.keys) | ((uintptr_t)a.tile_out)) & 15) == 0;
}

int scatter_quad_threads(int dim, int N) {
  const int cap = dim == 2 ? 1024 : 512;
  const int t = round_threads(N >> 2);
  return t > cap ? cap : t;
}

// workgroups the forward scatter's channel chunks aim for: one per CU in 2D, two in 3D (eight corners of atomics per point); with two
// or more planes per CU (the headline: 512) a plane's channels still go to TWO workgroups — 1024 thinner workgroups balance the
// tail better than 512 (measured: profiles/r5_nsplit_want.txt)
#ifndef CT_SCATTER_WANT
#define CT_SCATTER_WANT (DIM == 3 ? 512 : ((long long)a.B * a.H >= 512 ? 2ll * a.B * a.H : 256))
#endif
// scatter: Splat fwd (max/sum) and Slice bwd g_grid
template <int DIM, bool FROM_KEYS>
int run_scatter(RasterArgs a, const int* W, bool sum, hipStream_t st) {
  GridW<DIM> g = make_grid<DIM>(W);
  Plan p = make_plan(a.B, a.H, a.C, a.N, g.G, 1, CT_SCATTER_WANT);
  a.CC = p.CC;
  a.nchunks = p.nchunks;
  dim3 grid(p.nchunks, a.H, a.B);
  if (p.lds_tile) {
    if (sum) {
      if constexpr (DIM == 2 && FROM_KEYS) {
        const int r = run_scatter_add_hot(a, g, st);
        if (r != CT_EINVAL) return r;
      }
      if constexpr (DIM == 3 && FROM_KEYS) {
        const int r = run_scatter_add_sorted3(a, g, st);      // small 3D grids, one sorted segment per plane
        if (r != CT_EINVAL) return r;
      }
      constexpr int kRegCh = 8;
      const bool aligned = ((((uintptr_t)a.src) | ((uintptr_t)a.pos.keys) | ((uintptr_t)a.tile_out)) & 15) == 0;
      if (FROM_KEYS && aligned && (a.N & 3) == 0 && a.N <= 4096 && (g.G & 3) == 0) {
        if (a.CC > kRegCh) {
          a.CC = kRegCh;
          a.nchunks = (a.C + kRegCh - 1) / kRegCh;
        }
        dim3 rgrid(a.nchunks, a.H, a.B);
        CT_LAUNCH((scatter_add_fx_reg_kernel<DIM, 4, kRegCh, CT_FXREG_CG>), rgrid, round_threads(a.N / 4),
                  (size_t)a.CC * g.G * 4, st, a, g);
        note("scatter_add_fx_reg");
      } else {
        CT_LAUNCH((scatter_add_fx_kernel<DIM, FROM_KEYS>), grid, p.threads, p.lds_bytes + (size_t)a.CC * 8, st, a, g);
        note("scatter_add_fx");
      }
    } else {
      if (scatter_quad_ok(a, FROM_KEYS, g.G)) {
        const int qt = scatter_quad_threads(DIM, a.N);
        if (a.pad_dtype != CT_PAD_NONE) CT_LAUNCH((scatter_quad_kernel<DIM, false, true>), grid, qt, p.lds_bytes, st, a, g);
        else if (DIM == 2 && g.W[0] == 32 && g.W[DIM - 1] == 32) CT_LAUNCH((scatter_quad_kernel<DIM, false, false, DIM == 2 ? 32 : 0>), grid, qt, p.lds_bytes, st, a, g);
        else if (DIM == 3 && CT_SCATTER3_CUBES && g.W[0] == 8 && g.W[1] == 8 && g.W[DIM - 1] == 8) CT_LAUNCH((scatter_quad_kernel<DIM, false, false, DIM == 3 ? 8 : 0>), grid, qt, p.lds_bytes, st, a, g);
        else if (DIM == 3 && CT_SCATTER3_CUBES && g.W[0] == 16 && g.W[1] == 16 && g.W[DIM - 1] == 16) CT_LAUNCH((scatter_quad_kernel<DIM, false, false, DIM == 3 ? 16 : 0>), grid, qt, p.lds_bytes, st, a, g);
        else CT_LAUNCH((scatter_quad_kernel<DIM, false, false>), grid, qt, p.lds_bytes, st, a, g);
        note("scatter_quad_max");
      } else {
        CT_LAUNCH((scatter_kernel<DIM, FROM_KEYS, false, true>), grid, p.threads, p.lds_bytes, st, a, g);
        note("scatter_generic_max");
      }
    }
  } else {
    if (hipMemsetAsync(a.tile_out, 0, (size_t)a.B * a.H * a.C * g.G * 4, st) != hipSuccess) return CT_ELAUNCH;
    if (sum) CT_LAUNCH((scatter_kernel<DIM, FROM_KEYS, true, false>), grid, p.threads, 0, st, a, g);
    else CT_LAUNCH((scatter_kernel<DIM, FROM_KEYS, false, false>), grid, p.threads, 0, st, a, g);
    note("scatter_global_atomics");
  }
  return CT_OK;
}


// quad (4 points per thread, dwordx4) kernels need 16-byte aligned rows
bool quad_ok(const RasterArgs& a, bool from_keys, bool lds_tile) {
  if (!from_keys || !lds_tile || (a.N & 3) != 0) return false;
  uintptr_t bits = (uintptr_t)a.pos.keys | (uintptr_t)a.src | (uintptr_t)a.dst | (uintptr_t)a.g_pos |
                   (uintptr_t)a.tile_in | (uintptr_t)a.tile_in2;
  return (bits & 15) == 0;
}

int quad_threads(int N, int nsplit) {
  int nq = ((N >> 2) + nsplit - 1) / nsplit;
  int t = round_threads(nq);
  return t > CT_QUAD_THREADS ? CT_QUAD_THREADS : t;
}

// dim: 2 / 3 for the Slice-forward gathers, 0 for the other callers.  Measured over the zoo rows (profiles/r5_nsplit_want.txt): the 2D
// gathers and the short 3D clouds are fastest at ONE workgroup per CU (every split stages the tile again and a 128-thread workgroup
// is mostly latency), the long 3D clouds (N = 16384, eight corners of work per point) at two.
int pick_nsplit(int B, int H, int nchunks, int N, int dim = 0) {
  // pure gather kernels may split N freely (the tile is re-staged per split)
  const long long want = (dim == 2 || (dim == 3 && N <= 2048)) ? 256 : 512;
  int ns = 1;
  while ((long long)B * H * nchunks * ns < want && N / (ns * 2) >= 256) ns *= 2;
  return ns;
}


// out[i] (+)= sum_k parts[k*stride + i] (ascending k): the partial g_keys of the channel-chunk groups
template <typename V>
__global__ void __launch_bounds__(256) sum_parts_kernel(const V* parts, V* out, size_t n, size_t stride, int k, const V* add) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  V s = parts[i];
  for (int j = 1; j < k; ++j) {
    const V t = parts[(size_t)j * stride + i];
    if constexpr (sizeof(V) == 16) { s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; } else { s += t; }
  }
  if (add) {
    const V o = add[i];
    if constexpr (sizeof(V) == 16) { s.x = o.x + s.x; s.y = o.y + s.y; s.z = o.z + s.z; s.w = o.w + s.w; } else { s = o + s; }
  }
  out[i] = s;
}

// out = (add +) parts[0] + parts[1] + ... ; add may be out itself (accumulate in place) or null
int launch_sum_parts(const float* parts, float* out, size_t n, size_t stride, int k, const float* add, hipStream_t st) {
  CT_CLEAR_ERROR();
  if (((n | stride) & 3) == 0 && ((((uintptr_t)parts) | ((uintptr_t)out) | ((uintptr_t)add)) & 15) == 0)
    hipLaunchKernelGGL(sum_parts_kernel<float4>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, (const float4*)parts,
                       (float4*)out, n / 4, stride / 4, k, (const float4*)add);
  else
    hipLaunchKernelGGL(sum_parts_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, parts, out, n, stride, k, add);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

// ---------------------------------------------------------------------------
// hot-shape dispatch (ct_raster_hot.h): 2D, corners from keys, N % 4 == 0, C % 4 == 0, G % 4 == 0, 16-byte
// aligned rows, enough (b,h) planes to fill the chip with one workgroup per plane
// ---------------------------------------------------------------------------
constexpr int kHalfCuLdsBytes = 80 * 1024;     // two workgroups per CU
#ifndef CT_FUSED_LDS_BUDGET
#define CT_FUSED_LDS_BUDGET kHalfCuLdsBytes
#endif

struct HotPlan {
  int CC, nchunks;
  size_t lds;
};

// channels per chunk: a multiple of 4 such that `fixed + CC * per_ch` bytes fit half a CU's LDS (else a whole CU's)
// LDS of the Splat(max) backward kernels behind their tiles: four counters and, per four-channel group of the plane (<= 64), its
// non-zero cells and its matches, per chunk the sum of the awarded cotangents' bit patterns (ct_raster_hot.h: kTieGroups), and the
// words of the single-tie repair and of its search
constexpr size_t kSplatBwdFixed = 16 + 3 * 64 * 4 + 64;

bool hot_chunks(int C, size_t per_ch, size_t fixed, HotPlan& hp, long long budget = kHalfCuLdsBytes) {
  if ((C & 3) != 0) return false;
  long long cc = (budget - (long long)fixed) / (long long)per_ch;
  cc &= ~3ll;
  if (cc < 4) {
    cc = ((long long)kBigLdsBytes - (long long)fixed) / (long long)per_ch;
    cc &= ~3ll;
    if (cc < 4) return false;
  }
  if (cc > C) cc = C;
  hp.nchunks = (int)((C + cc - 1) / cc);
  hp.CC = (((C + hp.nchunks - 1) / hp.nchunks) + 3) & ~3;      // balanced chunks
  hp.nchunks = (C + hp.CC - 1) / hp.CC;
  hp.lds = fixed + (size_t)hp.CC * per_ch;
  return true;
}

// Backward hot kernels: chunks of a plane may be dealt to `ncg` workgroups when there are too few (b,h) planes to fill
// the chip with one workgroup per plane (the zoo's B8 x H16 blocks: 128 planes) — thinner chunks, more workgroups;
// each group's partial g_keys goes to the workspace.  Pure shape arithmetic: the workspace queries use it too.
bool hot_bwd_plan(int B, int H, int C, int G, size_t per_ch, size_t fixed, long long budget, HotPlan& hp, int& ncg) {
  if (!hot_chunks(C, per_ch, fixed, hp, budget)) return false;
  ncg = 1;
  const long long planes = (long long)B * H;
  if (planes >= 256 || C <= 4) return true;
  // chunk groups that would give ~512 workgroups (two per CU) — or ~256 when even a four-channel chunk takes more than
  // half a CU's LDS: a second round of workgroups would repeat the per-plane setup and double the partial g_keys
  const long long wgs = (fixed + 4 * per_ch > (size_t)kHalfCuLdsBytes) ? 256 : 512;
  const int want = (int)((wgs + planes - 1) / planes);
  int cc = (((C + want - 1) / want) + 3) & ~3;
  if (cc < 4) cc = 4;
  if (cc < hp.CC) {
    hp.CC = cc;
    hp.nchunks = (C + cc - 1) / cc;
    hp.lds = fixed + (size_t)cc * per_ch;
  }
  ncg = hp.nchunks < want ? hp.nchunks : want;
  return true;
}

// arrival tickets cover the launch: one word per (plane, segment) for the g_keys folds, one per (plane, chunk group) for the
// g_grid folds, each kind in its half of the CT_TICKETS_BYTES buffer (ct_raster_hot.h: kTicketHalf)
bool tickets_cover(const void* tickets, long long planes, int ncg, int nseg) {
  const long long half = CT_TICKETS_BYTES / 8;
  return tickets != nullptr && ((uintptr_t)tickets & 3) == 0 && planes * nseg <= half && planes * ncg <= half;
}

// Is the in-kernel fold worth it?  The plane's LAST workgroup adds all of the plane's partials alone (~60-100 GB/s) while
// a sum_parts launch spreads them over the chip: measured equal at ~128-192 KiB per fold (B8 N4096 heads: +-1 us), 1-5 us
// better below (N2048), and 10-40 us WORSE at the decoders' 0.5-1.5 MiB (B2 N16384: 39 -> 49, 67 -> 106 us).
bool fold_pays(size_t bytes_per_fold) { return bytes_per_fold <= 192 * 1024; }

bool hot_shape_ok(const RasterArgs& a, int G, uintptr_t ptr_bits) {
  return hot_enabled() && (a.N & 3) == 0 && (G & 3) == 0 && (a.C & 3) == 0 && (ptr_bits & 15) == 0 &&
         ((long long)a.B * a.H >= 32 || (t_dbg_flags & CT_DEBUG_FORCE_HOT));
}

// kernels are instantiated for the 32 x 32 grid of the headline shape, the zoo's 64 x 64 and 16 x 16 (extent a template constant:
// immediate corner offsets, half-widths, cell count) and for any grid
#ifndef CT_HOT2_SQUARES
#define CT_HOT2_SQUARES 1
#endif
#ifndef CT_SORTED_HEADLINE
#define CT_SORTED_HEADLINE 1      // the sorted Slice backward with the headline's point and channel counts as constants too
#endif
#define CT_HOT_KERNEL0(KERNEL, PADV, WTV) KERNEL<PADV, WTV>
#define CT_HOT_KERNEL1(KERNEL, PADV, WTV, QPTV) KERNEL<PADV, WTV, QPTV>
#define CT_LAUNCH_HOT_(MK, GRID, NT, LDS, STREAM, ARGS, GW, ...)                          \
  do {                                                                                   \
    if ((ARGS).pad_dtype != CT_PAD_NONE)                                                  \
      CT_LAUNCH((MK(true, 0, ##__VA_ARGS__)), GRID, NT, LDS, STREAM, ARGS, GW);           \
    else if ((GW).W[0] == 32 && (GW).W[1] == 32)                                          \
      CT_LAUNCH((MK(false, 32, ##__VA_ARGS__)), GRID, NT, LDS, STREAM, ARGS, GW);         \
    else if (CT_HOT2_SQUARES && (GW).W[0] == 64 && (GW).W[1] == 64)                       \
      CT_LAUNCH((MK(false, 64, ##__VA_ARGS__)), GRID, NT, LDS, STREAM, ARGS, GW);         \
    else if (CT_HOT2_SQUARES && (GW).W[0] == 16 && (GW).W[1] == 16)                       \
      CT_LAUNCH((MK(false, 16, ##__VA_ARGS__)), GRID, NT, LDS, STREAM, ARGS, GW);         \
    else                                                                                  \
      CT_LAUNCH((MK(false, 0, ##__VA_ARGS__)), GRID, NT, LDS, STREAM, ARGS, GW);          \
  } while (0)

// the sorted-plane kernels (ct_raster_sorted.h) also know the 16 x 16 grid of the H16 blocks at compile time
#define CT_LAUNCH_SORTED_(MK, GRID, NT, LDS, STREAM, ARGS, GW, ...)                       \
  do {                                                                                   \
    if ((ARGS).pad_dtype != CT_PAD_NONE)                                                  \
      CT_LAUNCH((MK(true, 0, ##__VA_ARGS__)), GRID, NT, LDS, STREAM, ARGS, GW);           \
    else if ((GW).W[0] == 32 && (GW).W[1] == 32)                                          \
      CT_LAUNCH((MK(false, 32, ##__VA_ARGS__)), GRID, NT, LDS, STREAM, ARGS, GW);         \
    else if ((GW).W[0] == 16 && (GW).W[1] == 16)                                          \
      CT_LAUNCH((MK(false, 16, ##__VA_ARGS__)), GRID, NT, LDS, STREAM, ARGS, GW);         \
    else                                                                                  \
      CT_LAUNCH((MK(false, 0, ##__VA_ARGS__)), GRID, NT, LDS, STREAM, ARGS, GW);          \
  } while (0)

int hot_threads(int nq) {
  const int t = round_threads(nq);
  return t > kHotThreads ? kHotThreads : t;
}

// WIDE launches of the backward hot kernels: where a workgroup's tiles take more than half a CU's LDS only ONE 512-thread
// workgroup runs per CU — 8 waves, which neither hide the kernel's HBM waits nor fill its LDS and vector pipes (64^2 C16 / 16^3 C16
// at B8 N4096: LDS 28-39 % busy, vector ALU 30-39 %, profiles/r5_zoo_counters.txt).  One 1024-thread workgroup with one quad per
// thread doubles the waves: 64^2 C16 Slice backward 40.0 -> 34.6 us, Splat(max) backward 45.4 -> 34.7; 16^3 C16 54.0 -> 45.7, 58.5 ->
// 47.9 (profiles/r6_wide.txt).  Where two workgroups fit a CU (8^3 C32) the same launch LOSES (63 -> 70 us): not taken there.
// CLOUDCT_WIDE=0 turns it off (A/B runs).
bool wide_enabled() {
  static const int env = [] {
    const char* e = getenv("CLOUDCT_WIDE");
    return e ? atoi(e) : 1;
  }();
  return env != 0 && kHotThreads < kHotWideThreads;
}
bool hot_wide(size_t lds, int nq) { return wide_enabled() && lds > (size_t)kHalfCuLdsBytes && nq > kHotThreads; }
// does even a four-channel chunk of the fused Slice backward (conv tile + accumulators + counters) take more than half a CU's LDS?
// (then its launches are WIDE and a point segment may hold 8192 points: two quads per thread of a 1024-thread workgroup —
//  64^2 C16 B2 N16384: 2 segments instead of 4, 47.0 -> 37.7 us; 16^3 C16: 60.7 -> 48.1, profiles/r6_wide.txt)
bool slice_bwd_wide_shape(int G, int C) {
  return wide_enabled() && (size_t)(G + C + 2) * 4 + (size_t)4 * G * 8 > (size_t)kHalfCuLdsBytes;
}
#define CT_HOT_KERNEL1W(KERNEL, PADV, WTV, QPTV) KERNEL<PADV, WTV, QPTV, kHotWideThreads>

// Slice forward / gather with the channel-interleaved tile.  CT_EINVAL: not eligible.
int run_gather_hot(RasterArgs a, const GridW<2>& g, hipStream_t st) {
  const uintptr_t bits = (uintptr_t)a.pos.keys | (uintptr_t)a.dst | (uintptr_t)a.tile_in;
  if (!hot_shape_ok(a, g.G, bits)) return CT_EINVAL;
  HotPlan hp;
  if (!hot_chunks(a.C, (size_t)g.G * 4, 0, hp)) return CT_EINVAL;
  a.CC = hp.CC; a.nchunks = hp.nchunks; a.ncg = hp.nchunks;
  a.nsplit = pick_nsplit(a.B, a.H, hp.nchunks, a.N, 2);
  const int nq = ((a.N >> 2) + a.nsplit - 1) / a.nsplit;
  dim3 grid(hp.nchunks * a.nsplit, a.H, a.B);
#define CT_MK_GATHER(PADV, WTV) CT_HOT_KERNEL0(gather_ci_kernel, PADV, WTV)
  CT_LAUNCH_HOT_(CT_MK_GATHER, grid, hot_threads(nq), hp.lds, st, a, g);
  note("gather_ci");
  return CT_OK;
}

// Point segments of the fused Slice backward.  A workgroup keeps its points' g_keys sums in registers: at most 4096 points
// (two quads per thread).  Longer clouds (the decoders: N = 16384 at B2) are cut into nseg equal segments, one workgroup
// (per chunk group) each, scattering into its own partial g_grid tile in the workspace; the partial tiles are added in a
// fixed order afterwards (sum_parts_kernel) — N-split + second-stage merge, still bitwise reproducible.  The partial tiles
// are small where this matters (grids of 16^2 .. 16^3 cells), and the 32 planes of such a batch become 128 workgroups.
// Returns 0 when N cannot be cut into equal float4-addressable segments.
int slice_bwd_segments(int B, int H, int C, int N, int G, int dim) {
  const int cap = 8 * (slice_bwd_wide_shape(G, C) ? kHotWideThreads : kHotThreads);
  if (N <= cap) return 1;
  int nseg = (N + cap - 1) / cap;
  while (nseg <= 64 && (N % nseg != 0 || ((N / nseg) & 3) != 0)) ++nseg;
  if (nseg > 64) return 0;
  // Workgroups beyond one per (plane, segment) come from chunk groups, whose partial g_keys (dim * N floats per plane and
  // group) go through memory; where a plane's grid is smaller than its keys (16^2, 8^3: the decoders' small heads) more
  // segments are the cheaper way to fill the chip: their partial tiles are C * G floats.
  while ((long long)B * H * nseg < 256 && (long long)C * G <= (long long)dim * N && N % (2 * nseg) == 0 &&
         ((N / (2 * nseg)) & 3) == 0 && N / (2 * nseg) >= 1024)
    nseg *= 2;
  return nseg;
}

// Slice backward, fused.  CT_EINVAL: not eligible.  ws: scratch for the chunk groups' partial g_keys (may be null: then
// only shapes that need a single group per plane qualify).
// channel-group workgroups per plane of the sorted kernels: enough workgroups for the CUs, at least two groups each
int sorted_groups(long long planes, int C) {
  int ncg = 1;
  while (planes * ncg < 256 && (C >> 2) / (2 * ncg) >= 2) ncg *= 2;
  return ncg;
}

// sorted segments on small grids (ct_raster_sorted3d.h; defined with the 3D forms below)
template <int DIM>
size_t sorted3_workspace(int B, int H, int C, int N, const GridW<DIM>& g);
template <int DIM>
int run_slice_bwd_sorted3(RasterArgs a, const float* grid, float* g_pos, const GridW<DIM>& g, void* ws, size_t ws_bytes, hipStream_t st);

size_t slice_bwd_hot_workspace(int B, int H, int C, int N, const GridW<2>& g) {
  HotPlan hp;
  int ncg = 1;
  const int nseg = slice_bwd_segments(B, H, C, N, g.G, 2);
  if ((C & 3) || (N & 3) || (g.G & 3) || nseg == 0) return 0;
  size_t sorted_need = sorted3_workspace<2>(B, H, C, N, g);       // sorted segments (taken at few planes, or forced: 0 where not legal)
  // the sorted kernel's channel-group workgroups (few planes): partial g_keys
  if (nseg == 1 && N <= 4096 && sorted_groups((long long)B * H, C) > 1) {
    const size_t need1 = (size_t)sorted_groups((long long)B * H, C) * B * H * 2 * N * 4;
    sorted_need = need1 > sorted_need ? need1 : sorted_need;
  }
  if (!hot_bwd_plan(B * nseg, H, C, g.G, (size_t)g.G * 8, (size_t)(g.G + C + 2) * 4, CT_FUSED_LDS_BUDGET, hp, ncg)) return sorted_need;
  const size_t fused_need = (ncg > 1 ? (size_t)ncg * B * H * 2 * N * 4 : 0) + (nseg > 1 ? (size_t)nseg * B * H * C * g.G * 4 : 0);
  return fused_need > sorted_need ? fused_need : sorted_need;
}

// Sorted-plane kernels (ct_raster_sorted.h): one 1024-thread workgroup per (b, h) plane; every thread owns at most two items
// (N / 4 + 3 G / 4 <= 2048), positions and counts fit 16 bits, the carve-up fits a CU's LDS.  Worth it only where there is a
// plane per CU at least (the kernel does not split a plane) and three channel groups to spread the sort over.  CLOUDCT_SORTED=0
// turns the form off (A/B runs).
bool sorted_plane_ok(const RasterArgs& a, int G) {
  static const int env = [] {
    const char* e = getenv("CLOUDCT_SORTED");
    return e ? atoi(e) : -1;
  }();
  const unsigned f = t_dbg_flags.load(std::memory_order_relaxed);
  if ((f & CT_DEBUG_NO_SORTED) || (env == 0 && !(f & CT_DEBUG_FORCE_SORTED))) return false;
  if (a.N > 4096 || (a.N & 3) || (a.C & 3) || (G & 3) || a.N / 4 + (3 * G) / 4 > kMaxItems) return false;
  if (sort_lds(G, a.N, a.C).total > (size_t)kBigLdsBytes) return false;
  if (f & CT_DEBUG_FORCE_SORTED) return true;
  // the sort is paid once per plane and workgroup, the gain per four-channel group: B8 H64 N4096 32^2, sorted | scatter form
  // C4 27.8 | 23.1 us, C8 38.7 | 38.3, C12 51.3 | 57.6, C16 60.3 | 69.5, C32 99.9 | 138.9 (profiles/r5_sorted_channels.txt);
  // with fewer planes than CUs the scatter form is further from its best and two groups per workgroup already pay
  // (profiles/r5_sorted_h16.txt)
  const long long planes = (long long)a.B * a.H;
  const int ncg = sorted_groups(planes, a.C);
  if (planes * ncg < 256) return false;
  return ncg > 1 ? true : a.C >= 12;
}

// can ct_plane_sort take this layout?  (the consumers add their own conditions: channels, planes)
bool plane_sort_ok(int N, int dim, int G) {
  return dim == 2 && N <= 4096 && (N & 3) == 0 && (G & 3) == 0 && N / 4 + (3 * G) / 4 <= kMaxItems && G <= 65535 &&
         sort_lds(G, N, 64).total <= (size_t)kBigLdsBytes;
}

int run_slice_bwd_hot(RasterArgs a, const float* grid, float* g_pos, const GridW<2>& g, void* ws, size_t ws_bytes, hipStream_t st) {
  const uintptr_t bits = (uintptr_t)a.pos.keys | (uintptr_t)a.src | (uintptr_t)a.tile_out | (uintptr_t)grid | (uintptr_t)g_pos |
                         (uintptr_t)ws;
  const int nseg = slice_bwd_segments(a.B, a.H, a.C, a.N, g.G, 2);
  if (!hot_shape_ok(a, g.G, bits) || nseg == 0) return CT_EINVAL;
  {
    const int r = run_slice_bwd_sorted3<2>(a, grid, g_pos, g, ws, ws_bytes, st);      // small grids, few planes: sorted segments
    if (r != CT_EINVAL) return r;
  }
  // few workgroups (32 planes of 4096 points or less): one or two workgroups per plane lose to the split-N scatter +
  // gather pair (measured 67 vs 49 us on 64^2 C16, 49 vs 33 on 16^2 C16); long clouds are cut into segments (above)
  if ((long long)a.B * a.H * nseg < 64 && !(t_dbg_flags & CT_DEBUG_FORCE_HOT)) return CT_EINVAL;
  if (nseg == 1 && sorted_plane_ok(a, g.G)) {
    // the plane's points sorted by base cell, items of <= 4 entries per thread (ct_raster_sorted.h)
    int sg = (t_dbg_flags & CT_DEBUG_FORCE_SORTED) && (long long)a.B * a.H * sorted_groups((long long)a.B * a.H, a.C) < 256
                 ? ((a.C >> 2) >= 2 ? 2 : 1)          // tests: the group split on small shapes too
                 : sorted_groups((long long)a.B * a.H, a.C);
    const size_t gpos_n = (size_t)a.B * a.H * 2 * a.N;
    // no scratch for the partial key cotangents: one workgroup per plane — which only fills the chip where there is a plane
    // per CU (else the scatter form below, unless a test forces the sorted one)
    bool sorted_here = true;
    if (sg > 1 && (!ws || ws_bytes < (size_t)sg * gpos_n * 4)) {
      sg = 1;
      sorted_here = (long long)a.B * a.H >= 256 || (t_dbg_flags & CT_DEBUG_FORCE_SORTED);
    }
    if (sorted_here) {
    a.tile_in = grid;
    a.CC = 4; a.nchunks = a.C >> 2; a.ncg = sg; a.nseg = 1; a.Nrow = 0;
    a.g_pos = sg > 1 ? (float*)ws : g_pos;
    a.gpos_stride = sg > 1 ? gpos_n : 0;
    a.fold_gpos = g_pos;
    const bool fold = sg > 1 && tickets_cover(a.tickets, (long long)a.B * a.H, sg, 1) && fold_pays((size_t)sg * 2 * a.N * 4);
    if (!fold) a.tickets = nullptr;
    const SortLds L = sort_lds(g.G, a.N, a.C);
#ifdef CT_SORT_STAGGER
    { const char* e = getenv("CLOUDCT_SORT_STAGGER"); a.cnt_mask = e ? atoi(e) : 0; }
#endif
    dim3 wgrid(sg, a.H, a.B);
#define CT_MK_SLICE_BWD_SORTED(PADV, WTV, PSV) slice_bwd_sorted_kernel<PADV, WTV, PSV>
    if (a.sorted != nullptr) {
      a.sorted_stride = sort_record_bytes(a.N);
      CT_LAUNCH_SORTED_(CT_MK_SLICE_BWD_SORTED, wgrid, kSortThreads, L.total, st, a, g, true);
      note(sg > 1 ? "slice_bwd_presorted_groups" : "slice_bwd_presorted");
    } else {
      if (CT_SORTED_HEADLINE && a.pad_dtype == CT_PAD_NONE && g.W[0] == 32 && g.W[1] == 32 && a.N == 4096 && a.C == 16 && a.ncg == 1)
        CT_LAUNCH((slice_bwd_sorted_kernel<false, 32, false, true, 4096, 16>), wgrid, kSortThreads, L.total, st, a, g);
      else
        CT_LAUNCH_SORTED_(CT_MK_SLICE_BWD_SORTED, wgrid, kSortThreads, L.total, st, a, g, false);
      note(sg > 1 ? "slice_bwd_sorted_groups" : "slice_bwd_sorted");
    }
    if (fold) {
      note("folded");
    } else if (sg > 1) {
      CT_CLEAR_ERROR();
      if (launch_sum_parts((const float*)ws, g_pos, gpos_n, gpos_n, sg, nullptr, st) != CT_OK) return CT_ELAUNCH;
      CT_CHECK_LAUNCH();
    }
    return CT_OK;
    }
  }
  HotPlan hp;
  int ncg = 1;
  if (!hot_bwd_plan(a.B * nseg, a.H, a.C, g.G, (size_t)g.G * 8, (size_t)(g.G + a.C + 2) * 4, CT_FUSED_LDS_BUDGET, hp, ncg)) return CT_EINVAL;
  const size_t gpos_n = (size_t)a.B * a.H * 2 * a.N;
  const size_t grid_n = (size_t)a.B * a.H * a.C * g.G;
  if (nseg > 1 && (!ws || ws_bytes < (ncg > 1 ? (size_t)ncg * gpos_n * 4 : 0) + (size_t)nseg * grid_n * 4)) return CT_EINVAL;
  if (ncg > 1 && (!ws || ws_bytes < (size_t)ncg * gpos_n * 4)) {
    // no scratch for the partial sums: one group per plane if there are planes enough to be worth it
    if ((long long)a.B * a.H < 128 && !(t_dbg_flags & CT_DEBUG_FORCE_HOT)) return CT_EINVAL;
    if (!hot_chunks(a.C, (size_t)g.G * 8, (size_t)(g.G + a.C + 2) * 4, hp, CT_FUSED_LDS_BUDGET)) return CT_EINVAL;
    ncg = 1;
  }
  a.tile_in = grid;
  a.CC = hp.CC; a.nchunks = hp.nchunks; a.ncg = ncg;
  a.g_pos = ncg > 1 ? (float*)ws : g_pos;
  a.gpos_stride = ncg > 1 ? gpos_n : 0;
  float* const g_grid = a.tile_out;
  float* const grid_parts = nseg > 1 ? (float*)((char*)ws + (ncg > 1 ? (size_t)ncg * gpos_n * 4 : 0)) : nullptr;
  if (nseg > 1) a.tile_out = grid_parts;
  a.nseg = nseg; a.Nrow = a.N; a.N = a.N / nseg;
  // with arrival tickets the plane's last workgroup adds the partials itself: no sum_parts launches behind the kernel
  const bool fold = (ncg > 1 || nseg > 1) && tickets_cover(a.tickets, (long long)a.B * a.H, ncg, nseg) &&
                    fold_pays(ncg > 1 ? (size_t)ncg * 2 * a.N * 4 : 0) &&
                    fold_pays(nseg > 1 ? (size_t)nseg * ((a.C + ncg - 1) / ncg) * g.G * 4 : 0);
  if (!fold) a.tickets = nullptr;
  a.fold_gpos = g_pos; a.fold_grid = g_grid;
  dim3 wgrid(ncg, a.H, a.B * nseg);
  const int nq = a.N >> 2;
#define CT_MK_SLICE_BWD(PADV, WTV, QPTV) slice_bwd_fused_kernel<PADV, WTV, QPTV, true>
#define CT_MK_SLICE_BWD_WIDE(PADV, WTV, QPTV) slice_bwd_fused_kernel<PADV, WTV, QPTV, true, kHotWideThreads>
  if (hot_wide(hp.lds, nq) && nq <= kHotWideThreads) CT_LAUNCH_HOT_(CT_MK_SLICE_BWD_WIDE, wgrid, round_threads(nq), hp.lds, st, a, g, 1);
  else if (hot_wide(hp.lds, nq)) CT_LAUNCH_HOT_(CT_MK_SLICE_BWD_WIDE, wgrid, round_threads((nq + 1) >> 1), hp.lds, st, a, g, 2);
  else if (nq <= kHotThreads) CT_LAUNCH_HOT_(CT_MK_SLICE_BWD, wgrid, hot_threads(nq), hp.lds, st, a, g, 1);
  else CT_LAUNCH_HOT_(CT_MK_SLICE_BWD, wgrid, hot_threads((nq + 1) >> 1), hp.lds, st, a, g, 2);
  note(nseg > 1 ? "slice_bwd_fused_segments" : ncg > 1 ? "slice_bwd_fused_groups" : "slice_bwd_fused");
  if (hot_wide(hp.lds, nq)) note("wide");
  if (fold) {
    note("folded");
    return CT_OK;
  }
  if (ncg > 1) {
    CT_CLEAR_ERROR();
    if (launch_sum_parts((const float*)ws, g_pos, gpos_n, gpos_n, ncg, nullptr, st) != CT_OK) return CT_ELAUNCH;
    CT_CHECK_LAUNCH();
  }
  if (nseg > 1) {
    CT_CLEAR_ERROR();
    if (launch_sum_parts(grid_parts, g_grid, grid_n, grid_n, nseg, nullptr, st) != CT_OK) return CT_ELAUNCH;
    CT_CHECK_LAUNCH();
  }
  return CT_OK;
}

// The scatter-add alone (Splat(sum) forward, ct_slice_bwd_grid) on the fused kernel without its gather side.
int run_scatter_add_hot(RasterArgs a, const GridW<2>& g, hipStream_t st) {
  const uintptr_t bits = (uintptr_t)a.pos.keys | (uintptr_t)a.src | (uintptr_t)a.tile_out;
  if (!hot_shape_ok(a, g.G, bits) || a.N > 8 * kHotThreads) return CT_EINVAL;
  if (sorted_plane_ok(a, g.G)) {       // the sorted-plane kernel without its gather side (ct_raster_sorted.h)
    const int sg = sorted_groups((long long)a.B * a.H, a.C);      // (no g_keys here: the groups share nothing)
    a.CC = 4; a.nchunks = a.C >> 2; a.ncg = sg; a.nseg = 1; a.Nrow = 0; a.tickets = nullptr; a.tile_in = nullptr; a.sorted = nullptr;
    const SortLds L = sort_lds(g.G, a.N, a.C);
    dim3 wgrid(sg, a.H, a.B);
#define CT_MK_SCATTER_ADD_SORTED(PADV, WTV) slice_bwd_sorted_kernel<PADV, WTV, false, false>
    CT_LAUNCH_SORTED_(CT_MK_SCATTER_ADD_SORTED, wgrid, kSortThreads, L.total, st, a, g);
    note("scatter_add_sorted");
    return CT_OK;
  }
  HotPlan hp;
  if (!hot_chunks(a.C, (size_t)g.G * 4, (size_t)(g.G + a.C + 2) * 4, hp)) return CT_EINVAL;
  int ncg = 1;
  hot_bwd_plan(a.B, a.H, a.C, g.G, (size_t)g.G * 4, (size_t)(g.G + a.C + 2) * 4, kHalfCuLdsBytes, hp, ncg);   // no g_keys: groups are free
  a.CC = hp.CC; a.nchunks = hp.nchunks; a.ncg = ncg;
  a.tickets = nullptr;
  dim3 wgrid(ncg, a.H, a.B);
  const int nq = a.N >> 2;
#define CT_MK_SCATTER_ADD(PADV, WTV, QPTV) slice_bwd_fused_kernel<PADV, WTV, QPTV, false>
  if (nq <= kHotThreads) CT_LAUNCH_HOT_(CT_MK_SCATTER_ADD, wgrid, hot_threads(nq), hp.lds, st, a, g, 1);
  else CT_LAUNCH_HOT_(CT_MK_SCATTER_ADD, wgrid, hot_threads((nq + 1) >> 1), hp.lds, st, a, g, 2);
  note("scatter_add_fused");
  return CT_OK;
}

// Splat(sum) backward in one pass.  CT_EINVAL: not eligible.
int run_splat_sum_bwd_hot(RasterArgs a, const GridW<2>& g, hipStream_t st) {
  const uintptr_t bits = (uintptr_t)a.pos.keys | (uintptr_t)a.src | (uintptr_t)a.dst | (uintptr_t)a.g_pos | (uintptr_t)a.tile_in;
  if (!hot_shape_ok(a, g.G, bits)) return CT_EINVAL;
  const size_t lds = (size_t)a.C * g.G * 4;
  if (lds > (size_t)kBigLdsBytes) return CT_EINVAL;
  a.nsplit = pick_nsplit(a.B, a.H, 1, a.N);
  const int nq = ((a.N >> 2) + a.nsplit - 1) / a.nsplit;
  dim3 grid(a.nsplit, a.H, a.B);
#define CT_MK_SPLAT_SUM_BWD(PADV, WTV) CT_HOT_KERNEL0(splat_sum_bwd_kernel, PADV, WTV)
  CT_LAUNCH_HOT_(CT_MK_SPLAT_SUM_BWD, grid, hot_threads(nq), lds, st, a, g);
  note("splat_sum_bwd_hot");
  return CT_OK;
}

// Point segments of the hot Splat(max) backward (ct_raster_hot.h: splat_max_bwd_hot_kernel): nseg workgroups per plane, each
// walking all chunks for its own points.  1 = none.  Needs arrival tickets (the plane's tie test) and an incoming key
// cotangent that does not alias the output.  Chosen where one workgroup per plane leaves the chip empty (few planes) or the
// cloud outgrows a workgroup's registers (2D: N > 4096), as long as re-staging the plane's tiles per segment stays within
// ~4x the segment's own point traffic (nseg * G <= 4 N: fine for 8^3 .. 64^2 / 16^3 at the zoo's sizes, not for 128^2).
// CLOUDCT_SPLAT_BWD_NSEG / ct_debug_set_nseg = 1: off, n > 1: that many where legal (A/B runs, tests of the 3D form).
int splat_bwd_segments(int B, int H, int C, int N, int G, int dim, size_t lds_per_wg) {
  static const int env_forced = [] {
    const char* e = getenv("CLOUDCT_SPLAT_BWD_NSEG");
    return e ? atoi(e) : 0;
  }();
  const int dbg = t_dbg_nseg.load(std::memory_order_relaxed);
  const int forced = dbg ? dbg : env_forced;
  if (forced == 1) return 1;
  auto legal = [&](int ns) { return ns >= 1 && ns <= 64 && N % (4 * ns) == 0 && N / ns >= 256 && (long long)ns * G <= 4ll * N; };
  if (forced > 1) return legal(forced) ? forced : 1;
  // Measured (tools/zoo_sweep.py, B8 N4096 / B2 N16384, H16): 16^2 C16 29.2 -> 23.5 / 39.1 -> 24.0 us; but 8^3 C32 56.6 -> 72.6,
  // 16^3 C16 59 -> 75, 64^2 C16 47 -> 58: a segment's threads walk ALL channels, so there are nchunks-times fewer waves in
  // flight than with chunk groups, and every segment re-stages every chunk.  Automatic only where the plane's tiles are one
  // small chunk (all channels' {z, g_z} within 32 KiB) and the cloud is long enough for >= 1024 points per segment.
  if ((size_t)C * G * 8 > 32 * 1024 || N < 4096) return 1;
  // ~256 workgroups: B8 N4096 nseg 2 / 4 / 8 = 20.3 / 22.1 / 27.1 us, B2 N16384 nseg 4 / 8 / 16 = 31.3 / 20.1 / 23.1 (tools/dev/nseg_sweep.py)
  const long long planes = (long long)B * H;
  const long long target = 256;
  (void)lds_per_wg;
  int nseg = 1;
  while (planes * nseg < target && legal(2 * nseg) && N / (2 * nseg) >= 1024) nseg *= 2;
  while (dim == 2 && N / nseg > 8 * kHotThreads && legal(2 * nseg)) nseg *= 2;      // the register form: <= 4096 points
  return nseg;
}

// can the hot Splat(max) backward take this call?  ncg: chunk groups (partial g_keys through the workspace); nseg: point
// segments (preferred where legal: nothing to add up); single: one group whose threads own their quads
template <int DIM>
bool splat_bwd_hot_plan(const RasterArgs& a, const GridW<DIM>& g, HotPlan& hp, int& ncg, int& nseg, bool& single) {
  const uintptr_t bits = (uintptr_t)a.pos.keys | (uintptr_t)a.src | (uintptr_t)a.dst | (uintptr_t)a.g_pos |
                         (uintptr_t)a.tile_in | (uintptr_t)a.tile_in2 | (uintptr_t)a.gpos_add;
  nseg = 1;
  if (!hot_shape_ok(a, g.G, bits)) return false;
  // segments first: every workgroup stages every chunk, so the chunks are as fat as LDS allows (no chunk groups)
  if (a.tickets != nullptr && tickets_cover(a.tickets, (long long)a.B * a.H, 1, 1) && (long long)a.B * a.H <= kTicketHalf / 2 &&
      ((uintptr_t)a.tickets & 7) == 0 && (a.gpos_add == nullptr || a.gpos_add != a.g_pos)) {
    HotPlan sp;
    if (hot_chunks(a.C, (size_t)g.G * 8, kSplatBwdFixed, sp)) {
      const int ns = splat_bwd_segments(a.B, a.H, a.C, a.N, g.G, DIM, sp.lds);
      if (ns > 1) {
        hp = sp; ncg = 1; nseg = ns; single = true;
        return true;
      }
    }
  }
  if (!hot_bwd_plan(a.B, a.H, a.C, g.G, (size_t)g.G * 8, kSplatBwdFixed, kHalfCuLdsBytes, hp, ncg)) return false;
  if constexpr (DIM == 2) {
    single = (a.N >> 2) <= 2 * kHotThreads;
  } else {
    // a workgroup walks all N points of its plane: with fewer workgroups than CUs (16^3 at B2 H16: 128) the generic
    // kernel's thinner single-channel chunks win (72 vs 147 us)
    if ((long long)a.B * a.H * ncg < 256 && !(t_dbg_flags & CT_DEBUG_FORCE_HOT)) return false;
    single = true;       // (the loop form: every thread owns its quads' rows)
  }
  return true;
}

size_t splat_bwd_hot_workspace(int B, int H, int C, int N, const GridW<2>& g) {
  HotPlan hp;
  int ncg = 1;
  if ((C & 3) || (N & 3) || (g.G & 3)) return 0;
  if (!hot_bwd_plan(B, H, C, g.G, (size_t)g.G * 8, kSplatBwdFixed, kHalfCuLdsBytes, hp, ncg)) return 0;
  return ncg > 1 ? (size_t)ncg * B * H * 2 * N * 4 : 0;
}

int run_splat_max_bwd_hot(RasterArgs a, const GridW<2>& g, const HotPlan& hp, int ncg, int nseg, void* ws, hipStream_t st) {
  a.CC = hp.CC; a.nchunks = hp.nchunks; a.ncg = ncg;
  float* const out = a.g_pos;
  const float* const add = a.gpos_add;
  const size_t gpos_n = (size_t)a.B * a.H * 2 * a.N;
#define CT_MK_SPLAT_BWD(PADV, WTV, QPTV) CT_HOT_KERNEL1(splat_max_bwd_hot_kernel, PADV, WTV, QPTV)
#define CT_MK_SPLAT_BWD_WIDE(PADV, WTV, QPTV) CT_HOT_KERNEL1W(splat_max_bwd_hot_kernel, PADV, WTV, QPTV)
  if (nseg > 1) {       // point segments: ncg == 1, tickets present (splat_bwd_hot_plan)
    a.nseg = nseg; a.Nrow = a.N; a.N = a.N / nseg;
    dim3 wgrid(1, a.H, a.B * nseg);
    const int nq = a.N >> 2;
    if (hot_wide(hp.lds, nq) && nq <= kHotWideThreads) CT_LAUNCH_HOT_(CT_MK_SPLAT_BWD_WIDE, wgrid, round_threads(nq), hp.lds, st, a, g, 1);
    else if (hot_wide(hp.lds, nq)) CT_LAUNCH_HOT_(CT_MK_SPLAT_BWD_WIDE, wgrid, kHotWideThreads, hp.lds, st, a, g, 0);
    else if (nq <= kHotThreads) CT_LAUNCH_HOT_(CT_MK_SPLAT_BWD, wgrid, hot_threads(nq), hp.lds, st, a, g, 1);
    else if (nq <= 2 * kHotThreads) CT_LAUNCH_HOT_(CT_MK_SPLAT_BWD, wgrid, hot_threads((nq + 1) >> 1), hp.lds, st, a, g, 2);
    else CT_LAUNCH_HOT_(CT_MK_SPLAT_BWD, wgrid, kHotThreads, hp.lds, st, a, g, 0);
    note("splat_max_bwd_hot_segments");
    return CT_OK;
  }
  const bool fold = ncg > 1 && tickets_cover(a.tickets, (long long)a.B * a.H, ncg, 1) && fold_pays((size_t)ncg * 2 * a.N * 4);
  if (!fold) a.tickets = nullptr;
  if (ncg > 1) {
    a.g_pos = (float*)ws;
    a.gpos_stride = gpos_n;
    a.gpos_add = nullptr;
    a.fold_gpos = out; a.fold_add = add;
  }
  dim3 wgrid(ncg, a.H, a.B);
  const int nq = a.N >> 2;
  if (hot_wide(hp.lds, nq) && nq <= kHotWideThreads) CT_LAUNCH_HOT_(CT_MK_SPLAT_BWD_WIDE, wgrid, round_threads(nq), hp.lds, st, a, g, 1);
  else if (hot_wide(hp.lds, nq)) CT_LAUNCH_HOT_(CT_MK_SPLAT_BWD_WIDE, wgrid, kHotWideThreads, hp.lds, st, a, g, 0);
  else if (nq <= kHotThreads) CT_LAUNCH_HOT_(CT_MK_SPLAT_BWD, wgrid, hot_threads(nq), hp.lds, st, a, g, 1);
  else if (nq <= 2 * kHotThreads) CT_LAUNCH_HOT_(CT_MK_SPLAT_BWD, wgrid, hot_threads((nq + 1) >> 1), hp.lds, st, a, g, 2);
  else CT_LAUNCH_HOT_(CT_MK_SPLAT_BWD, wgrid, kHotThreads, hp.lds, st, a, g, 0);
  note(ncg > 1 ? "splat_max_bwd_hot_groups" : "splat_max_bwd_hot");
  if (hot_wide(hp.lds, nq)) note("wide");
  if (fold) {
    note("folded");
    return CT_OK;
  }
  if (ncg > 1) {
    CT_CLEAR_ERROR();
    if (launch_sum_parts((const float*)ws, out, gpos_n, gpos_n, ncg, add, st) != CT_OK) return CT_ELAUNCH;
    CT_CHECK_LAUNCH();
  }
  return CT_OK;
}


// ---- 3D forms (ct_raster_hot3d.h) ----
#define CT_LAUNCH_HOT3_(KPAD, KNOPAD, GRID, NT, LDS, STREAM, ARGS, GW)                    \
  do {                                                                                   \
    if ((ARGS).pad_dtype != CT_PAD_NONE) CT_LAUNCH((KPAD), GRID, NT, LDS, STREAM, ARGS, GW); \
    else CT_LAUNCH((KNOPAD), GRID, NT, LDS, STREAM, ARGS, GW);                             \
  } while (0)
// the same for kernels instantiated for the zoo's cubic grids (8^3, 16^3: the extent a template constant) beside the general form
#ifndef CT_HOT3_CUBES
#define CT_HOT3_CUBES 1
#endif
inline int cube_of(const GridW<3>& g) {
  if (!CT_HOT3_CUBES || g.W[0] != g.W[1] || g.W[1] != g.W[2]) return 0;
  return g.W[0] == 8 ? 8 : g.W[0] == 16 ? 16 : 0;
}
#define CT_LAUNCH_HOT3W_(KERNEL, QPTV, GRID, NT, LDS, STREAM, ARGS, GW)                                              \
  do {                                                                                                               \
    const int cube_ = cube_of(GW);                                                                                   \
    if (cube_ == 8) CT_LAUNCH_HOT3_((KERNEL<true, QPTV, 8>), (KERNEL<false, QPTV, 8>), GRID, NT, LDS, STREAM, ARGS, GW);       \
    else if (cube_ == 16) CT_LAUNCH_HOT3_((KERNEL<true, QPTV, 16>), (KERNEL<false, QPTV, 16>), GRID, NT, LDS, STREAM, ARGS, GW); \
    else CT_LAUNCH_HOT3_((KERNEL<true, QPTV, 0>), (KERNEL<false, QPTV, 0>), GRID, NT, LDS, STREAM, ARGS, GW);                    \
  } while (0)
// the WIDE launches (hot_wide): 1024-thread workgroups, for the 16^3 cube and any grid
#define CT_LAUNCH_HOT3WIDE_(KERNEL, QPTV, GRID, NT, LDS, STREAM, ARGS, GW)                                           \
  do {                                                                                                               \
    if (cube_of(GW) == 16)                                                                                           \
      CT_LAUNCH_HOT3_((KERNEL<true, QPTV, 16, kHotWideThreads>), (KERNEL<false, QPTV, 16, kHotWideThreads>), GRID, NT, LDS, STREAM, ARGS, GW); \
    else                                                                                                             \
      CT_LAUNCH_HOT3_((KERNEL<true, QPTV, 0, kHotWideThreads>), (KERNEL<false, QPTV, 0, kHotWideThreads>), GRID, NT, LDS, STREAM, ARGS, GW);   \
  } while (0)

int run_gather_hot(RasterArgs a, const GridW<3>& g, hipStream_t st) {
  const uintptr_t bits = (uintptr_t)a.pos.keys | (uintptr_t)a.dst | (uintptr_t)a.tile_in;
  if (!hot_shape_ok(a, g.G, bits)) return CT_EINVAL;
  HotPlan hp;
  if (!hot_chunks(a.C, (size_t)g.G * 4, 0, hp)) return CT_EINVAL;
  // few planes: thinner chunks give more workgroups (the tile is staged once per chunk either way)
  while ((long long)a.B * a.H * hp.nchunks < 256 && hp.CC > 4) {
    hp.CC = ((hp.CC / 2) + 3) & ~3;
    hp.nchunks = (a.C + hp.CC - 1) / hp.CC;
    hp.lds = (size_t)hp.CC * g.G * 4;
  }
  a.CC = hp.CC; a.nchunks = hp.nchunks; a.ncg = hp.nchunks;
  a.nsplit = pick_nsplit(a.B, a.H, hp.nchunks, a.N, 3);
  const int nq = ((a.N >> 2) + a.nsplit - 1) / a.nsplit;
  dim3 grid(hp.nchunks * a.nsplit, a.H, a.B);
  {
    const int cube = cube_of(g);
    if (cube == 8) CT_LAUNCH_HOT3_((gather_ci3_kernel<true, 8>), (gather_ci3_kernel<false, 8>), grid, hot_threads(nq), hp.lds, st, a, g);
    else if (cube == 16) CT_LAUNCH_HOT3_((gather_ci3_kernel<true, 16>), (gather_ci3_kernel<false, 16>), grid, hot_threads(nq), hp.lds, st, a, g);
    else CT_LAUNCH_HOT3_((gather_ci3_kernel<true, 0>), (gather_ci3_kernel<false, 0>), grid, hot_threads(nq), hp.lds, st, a, g);
  }
  note("gather_ci3");
  return CT_OK;
}

// Sorted-segment kernels for small 3D grids (ct_raster_sorted3d.h): 512-thread workgroups of at most 2048 points (nseg point
// segments per plane, each with its own partial g_grid tile) and a share of the plane's four-channel groups (ncg workgroups per
// segment, each with its own partial g_keys); every thread owns at most two items (n / 4 + 3 G / 4 <= 1024).  CLOUDCT_SORTED=0 /
// CT_DEBUG_NO_SORTED turn the form off, CT_DEBUG_FORCE_SORTED takes it wherever it is legal (tests).
struct Sorted3Plan {
  int nseg, ncg, n;
  size_t lds;
};
bool sorted3_plan(int B, int H, int C, int N, int G, int dim, bool gather, Sorted3Plan& p) {
  static const int env = [] {
    const char* e = getenv("CLOUDCT_SORTED");
    return e ? atoi(e) : -1;
  }();
  const unsigned f = t_dbg_flags.load(std::memory_order_relaxed);
  const bool forced = (f & (dim == 3 ? CT_DEBUG_FORCE_SORTED : CT_DEBUG_FORCE_SORTED_SEG)) != 0;
  if ((f & CT_DEBUG_NO_SORTED) || (env == 0 && !forced)) return false;
  if (dim == 2 && !forced && (f & CT_DEBUG_FORCE_SORTED)) return false;      // (tests of the one-workgroup-per-plane 2D kernel)
  if ((N & 3) || (C & 3) || (G & 3) || G > 1024) return false;
  int nseg = (N + kS3MaxPoints - 1) / kS3MaxPoints;
  while (nseg <= 64 && (N % nseg != 0 || ((N / nseg) & 3) != 0)) ++nseg;
  if (nseg > 64) return false;
  const int n = N / nseg;
  if (n / 4 + (3 * G) / 4 > kS3MaxItems) return false;
  const size_t lds = sort3_lds(G, n, C, dim).total;
  if (lds > (size_t)kBigLdsBytes) return false;
  const long long planes = (long long)B * H;
  const int ngroups = C >> 2;
  int ncg = 1;
  // two workgroups per CU, at least two groups per workgroup to spread its sort over (the scatter-add alone has no partial
  // g_keys: its groups share nothing, one group per workgroup is fine)
  while (planes * nseg * ncg < 512 && ngroups / (2 * ncg) >= (gather ? 2 : 1)) ncg *= 2;
  if (forced && ncg == 1 && ngroups >= 2 && planes * nseg < 256) ncg = 2;      // tests: the group split on small shapes too
  {
    static const int env_ncg = [] {      // experiments: CLOUDCT_S3_NCG = channel-group workgroups per segment
      const char* e = getenv("CLOUDCT_S3_NCG");
      return e ? atoi(e) : 0;
    }();
    if (env_ncg > 0 && env_ncg <= ngroups) ncg = env_ncg;
  }
  p.nseg = nseg; p.ncg = ncg; p.n = n; p.lds = lds;
  if (forced) return true;
  // the sort is paid per workgroup, the gain per group: worth it where the chip fills and there are groups to spread it over
  return planes * nseg * ncg >= 256 && ngroups / ncg >= (gather ? 2 : 1);
}

template <int DIM>
size_t sorted3_workspace(int B, int H, int C, int N, const GridW<DIM>& g) {
  Sorted3Plan p;
  if (!sorted3_plan(B, H, C, N, g.G, DIM, true, p)) return 0;
  return (p.ncg > 1 ? (size_t)p.ncg * B * H * DIM * N * 4 : 0) + (p.nseg > 1 ? (size_t)p.nseg * B * H * C * g.G * 4 : 0);
}

size_t slice_bwd_hot_workspace(int B, int H, int C, int N, const GridW<3>& g) {
  HotPlan hp;
  int ncg = 1;
  const size_t sorted_need = sorted3_workspace(B, H, C, N, g);
  const int nseg = slice_bwd_segments(B, H, C, N, g.G, 3);
  if ((C & 3) || (N & 3) || (g.G & 3) || nseg == 0) return sorted_need;
  if (!hot_bwd_plan(B * nseg, H, C, g.G, (size_t)g.G * 8, (size_t)(g.G + C + 2) * 4, CT_FUSED_LDS_BUDGET, hp, ncg)) return sorted_need;
  const size_t fused_need = (ncg > 1 ? (size_t)ncg * B * H * 3 * N * 4 : 0) + (nseg > 1 ? (size_t)nseg * B * H * C * g.G * 4 : 0);
  return fused_need > sorted_need ? fused_need : sorted_need;
}

#define CT_LAUNCH_SORTED3_(GATHERV, GRID, LDS, STREAM, ARGS, GW)                                                            \
  do {                                                                                                                     \
    if ((ARGS).pad_dtype != CT_PAD_NONE) CT_LAUNCH((slice_bwd_sorted_seg_kernel<true, 3, 0, GATHERV>), GRID, kS3Threads, LDS, STREAM, ARGS, GW); \
    else if (cube_of(GW) == 8) CT_LAUNCH((slice_bwd_sorted_seg_kernel<false, 3, 8, GATHERV>), GRID, kS3Threads, LDS, STREAM, ARGS, GW);         \
    else CT_LAUNCH((slice_bwd_sorted_seg_kernel<false, 3, 0, GATHERV>), GRID, kS3Threads, LDS, STREAM, ARGS, GW);           \
  } while (0)
// the same kernel on small 2D grids (DIM = 2: one face); the zoo's 16 x 16 known at compile time
#define CT_LAUNCH_SORTED2S_(GATHERV, GRID, LDS, STREAM, ARGS, GW)                                                           \
  do {                                                                                                                     \
    if ((ARGS).pad_dtype != CT_PAD_NONE) CT_LAUNCH((slice_bwd_sorted_seg_kernel<true, 2, 0, GATHERV>), GRID, kS3Threads, LDS, STREAM, ARGS, GW); \
    else if ((GW).W[0] == 16 && (GW).W[1] == 16) CT_LAUNCH((slice_bwd_sorted_seg_kernel<false, 2, 16, GATHERV>), GRID, kS3Threads, LDS, STREAM, ARGS, GW); \
    else CT_LAUNCH((slice_bwd_sorted_seg_kernel<false, 2, 0, GATHERV>), GRID, kS3Threads, LDS, STREAM, ARGS, GW);           \
  } while (0)

// The scatter-add alone (Splat(sum) forward, ct_slice_bwd_grid) on ONE sorted segment per plane (there is no workspace for partial
// tiles behind these entry points): the sorted kernel without its gather side; the channel groups share nothing.  Measured against
// scatter_add_fx_reg (8^3 C32 B8 N2048, reduce=sum, tools/dev/zoo_shape.py): see profiles/r6_zoo_shape_checks.txt.
int run_scatter_add_sorted3(RasterArgs a, const GridW<3>& g, hipStream_t st) {
  const uintptr_t bits = (uintptr_t)a.pos.keys | (uintptr_t)a.src | (uintptr_t)a.tile_out;
  Sorted3Plan p;
  if (!hot_shape_ok(a, g.G, bits) || !sorted3_plan(a.B, a.H, a.C, a.N, g.G, 3, false, p) || p.nseg != 1) return CT_EINVAL;
  a.CC = 4; a.nchunks = a.C >> 2; a.ncg = p.ncg; a.nseg = 1; a.Nrow = a.N; a.tickets = nullptr; a.tile_in = nullptr;
  a.g_pos = nullptr; a.gpos_stride = 0; a.sorted = nullptr;
  dim3 wgrid(p.ncg, a.H, a.B);
  CT_LAUNCH_SORTED3_(false, wgrid, p.lds, st, a, g);
  note("scatter_add_sorted3");
  return CT_OK;
}

// Slice backward on sorted segments.  CT_EINVAL: not eligible (the scatter form below takes the call).
template <int DIM>
int run_slice_bwd_sorted3(RasterArgs a, const float* grid, float* g_pos, const GridW<DIM>& g, void* ws, size_t ws_bytes, hipStream_t st) {
  Sorted3Plan p;
  if (!sorted3_plan(a.B, a.H, a.C, a.N, g.G, DIM, true, p)) return CT_EINVAL;
  if constexpr (DIM == 2) {
    // 2D: only where the one-workgroup-per-plane sorted kernel (ct_raster_sorted.h) would leave the chip part empty or split a
    // plane's channel groups over workgroups that each sort the whole plane — few planes (the H16 blocks' 16 x 16 head): two
    // 512-thread workgroups per CU on 2048-point segments instead (profiles/r6_zoo_shape_checks.txt).  CLOUDCT_SORTED2S=0: off.
    static const int env2 = [] {
      const char* e = getenv("CLOUDCT_SORTED2S");
      return e ? atoi(e) : 1;
    }();
    if (!(t_dbg_flags & CT_DEBUG_FORCE_SORTED_SEG) && (env2 == 0 || (long long)a.B * a.H >= 256)) return CT_EINVAL;
  }
  const size_t gpos_n = (size_t)a.B * a.H * DIM * a.N;
  const size_t grid_n = (size_t)a.B * a.H * a.C * g.G;
  const size_t keys_need = p.ncg > 1 ? (size_t)p.ncg * gpos_n * 4 : 0, need = keys_need + (p.nseg > 1 ? (size_t)p.nseg * grid_n * 4 : 0);
  if (need > 0 && (!ws || ws_bytes < need)) return CT_EINVAL;
  a.tile_in = grid;
  a.CC = 4; a.nchunks = a.C >> 2; a.ncg = p.ncg;
  a.g_pos = p.ncg > 1 ? (float*)ws : g_pos;
  a.gpos_stride = p.ncg > 1 ? gpos_n : 0;
  float* const g_grid = a.tile_out;
  float* const grid_parts = p.nseg > 1 ? (float*)((char*)ws + keys_need) : nullptr;
  if (p.nseg > 1) a.tile_out = grid_parts;
  a.nseg = p.nseg; a.Nrow = a.N; a.N = p.n;
  const int per_wg = ((a.C >> 2) + p.ncg - 1) / p.ncg;
  const bool fold = (p.ncg > 1 || p.nseg > 1) && tickets_cover(a.tickets, (long long)a.B * a.H, p.ncg, p.nseg) &&
                    fold_pays(p.ncg > 1 ? (size_t)p.ncg * DIM * a.N * 4 : 0) &&
                    fold_pays(p.nseg > 1 ? (size_t)p.nseg * per_wg * 4 * g.G * 4 : 0);
  if (!fold) a.tickets = nullptr;
  a.fold_gpos = g_pos; a.fold_grid = g_grid;
  dim3 wgrid(p.ncg, a.H, a.B * p.nseg);
  if constexpr (DIM == 3) {
    CT_LAUNCH_SORTED3_(true, wgrid, p.lds, st, a, g);
    note(p.nseg > 1 ? "slice_bwd_sorted3_segments" : p.ncg > 1 ? "slice_bwd_sorted3_groups" : "slice_bwd_sorted3");
  } else {
    CT_LAUNCH_SORTED2S_(true, wgrid, p.lds, st, a, g);
    note(p.nseg > 1 ? "slice_bwd_sorted2s_segments" : p.ncg > 1 ? "slice_bwd_sorted2s_groups" : "slice_bwd_sorted2s");
  }
  if (fold) {
    note("folded");
    return CT_OK;
  }
  if (p.ncg > 1) {
    CT_CLEAR_ERROR();
    if (launch_sum_parts((const float*)ws, g_pos, gpos_n, gpos_n, p.ncg, nullptr, st) != CT_OK) return CT_ELAUNCH;
    CT_CHECK_LAUNCH();
  }
  if (p.nseg > 1) {
    CT_CLEAR_ERROR();
    if (launch_sum_parts(grid_parts, g_grid, grid_n, grid_n, p.nseg, nullptr, st) != CT_OK) return CT_ELAUNCH;
    CT_CHECK_LAUNCH();
  }
  return CT_OK;
}

int run_slice_bwd_hot(RasterArgs a, const float* grid, float* g_pos, const GridW<3>& g, void* ws, size_t ws_bytes, hipStream_t st) {
  const uintptr_t bits = (uintptr_t)a.pos.keys | (uintptr_t)a.src | (uintptr_t)a.tile_out | (uintptr_t)grid | (uintptr_t)g_pos |
                         (uintptr_t)ws;
  const int nseg = slice_bwd_segments(a.B, a.H, a.C, a.N, g.G, 3);
  if (!hot_shape_ok(a, g.G, bits) || nseg == 0) return CT_EINVAL;
  {
    const int r = run_slice_bwd_sorted3(a, grid, g_pos, g, ws, ws_bytes, st);      // small grids: sorted segments
    if (r != CT_EINVAL) return r;
  }
  // few workgroups (32 planes of 4096 points or less): one or two workgroups per plane lose to the split-N scatter +
  // gather pair (measured 67 vs 49 us on 64^2 C16, 49 vs 33 on 16^2 C16); long clouds are cut into segments (above)
  if ((long long)a.B * a.H * nseg < 64 && !(t_dbg_flags & CT_DEBUG_FORCE_HOT)) return CT_EINVAL;
  HotPlan hp;
  int ncg = 1;
  if (!hot_bwd_plan(a.B * nseg, a.H, a.C, g.G, (size_t)g.G * 8, (size_t)(g.G + a.C + 2) * 4, CT_FUSED_LDS_BUDGET, hp, ncg)) return CT_EINVAL;
  const size_t gpos_n = (size_t)a.B * a.H * 3 * a.N;
  const size_t grid_n = (size_t)a.B * a.H * a.C * g.G;
  if (nseg > 1 && (!ws || ws_bytes < (ncg > 1 ? (size_t)ncg * gpos_n * 4 : 0) + (size_t)nseg * grid_n * 4)) return CT_EINVAL;
  if (ncg > 1 && (!ws || ws_bytes < (size_t)ncg * gpos_n * 4)) {
    if ((long long)a.B * a.H < 128 && !(t_dbg_flags & CT_DEBUG_FORCE_HOT)) return CT_EINVAL;
    if (!hot_chunks(a.C, (size_t)g.G * 8, (size_t)(g.G + a.C + 2) * 4, hp, CT_FUSED_LDS_BUDGET)) return CT_EINVAL;
    ncg = 1;
  }
  a.tile_in = grid;
  a.CC = hp.CC; a.nchunks = hp.nchunks; a.ncg = ncg;
  a.g_pos = ncg > 1 ? (float*)ws : g_pos;
  a.gpos_stride = ncg > 1 ? gpos_n : 0;
  float* const g_grid = a.tile_out;
  float* const grid_parts = nseg > 1 ? (float*)((char*)ws + (ncg > 1 ? (size_t)ncg * gpos_n * 4 : 0)) : nullptr;
  if (nseg > 1) a.tile_out = grid_parts;
  a.nseg = nseg; a.Nrow = a.N; a.N = a.N / nseg;
  const bool fold = (ncg > 1 || nseg > 1) && tickets_cover(a.tickets, (long long)a.B * a.H, ncg, nseg) &&     // see the 2D form
                    fold_pays(ncg > 1 ? (size_t)ncg * 3 * a.N * 4 : 0) &&
                    fold_pays(nseg > 1 ? (size_t)nseg * ((a.C + ncg - 1) / ncg) * g.G * 4 : 0);
  if (!fold) a.tickets = nullptr;
  a.fold_gpos = g_pos; a.fold_grid = g_grid;
  dim3 wgrid(ncg, a.H, a.B * nseg);
  const int nq = a.N >> 2;
  if (hot_wide(hp.lds, nq) && nq <= kHotWideThreads)
    CT_LAUNCH_HOT3WIDE_(slice_bwd_fused3_kernel, 1, wgrid, round_threads(nq), hp.lds, st, a, g);
  else if (hot_wide(hp.lds, nq))
    CT_LAUNCH_HOT3WIDE_(slice_bwd_fused3_kernel, 2, wgrid, round_threads((nq + 1) >> 1), hp.lds, st, a, g);
  else if (nq <= kHotThreads)
    CT_LAUNCH_HOT3W_(slice_bwd_fused3_kernel, 1, wgrid, hot_threads(nq), hp.lds, st, a, g);
  else
    CT_LAUNCH_HOT3W_(slice_bwd_fused3_kernel, 2, wgrid, hot_threads((nq + 1) >> 1), hp.lds, st, a, g);
  note(nseg > 1 ? "slice_bwd_fused3_segments" : ncg > 1 ? "slice_bwd_fused3_groups" : "slice_bwd_fused3");
  if (hot_wide(hp.lds, nq)) note("wide");
  if (fold) {
    note("folded");
    return CT_OK;
  }
  if (ncg > 1) {
    CT_CLEAR_ERROR();
    if (launch_sum_parts((const float*)ws, g_pos, gpos_n, gpos_n, ncg, nullptr, st) != CT_OK) return CT_ELAUNCH;
    CT_CHECK_LAUNCH();
  }
  if (nseg > 1) {
    CT_CLEAR_ERROR();
    if (launch_sum_parts(grid_parts, g_grid, grid_n, grid_n, nseg, nullptr, st) != CT_OK) return CT_ELAUNCH;
    CT_CHECK_LAUNCH();
  }
  return CT_OK;
}

// The 3D hot Splat(max) backward runs in its loop form only (a quad's g_keys share is added to the workgroup's rows per
// chunk): with the 12 values of a quad kept in registers across chunks — and the corner setup of all four points hoisted
// out of the channel loop by the compiler — it spilled 170-450 registers and lost to the generic kernel (118 vs 72 us on
// 16^3 C16 B8 N4096); now 53 us.
size_t splat_bwd_hot_workspace(int B, int H, int C, int N, const GridW<3>& g) {
  HotPlan hp;
  int ncg = 1;
  if ((C & 3) || (N & 3) || (g.G & 3)) return 0;
  if (!hot_bwd_plan(B, H, C, g.G, (size_t)g.G * 8, kSplatBwdFixed, kHalfCuLdsBytes, hp, ncg)) return 0;
  return ncg > 1 ? (size_t)ncg * B * H * 3 * N * 4 : 0;
}

int run_splat_max_bwd_hot(RasterArgs a, const GridW<3>& g, const HotPlan& hp, int ncg, int nseg, void* ws, hipStream_t st) {
  a.CC = hp.CC; a.nchunks = hp.nchunks; a.ncg = ncg;
  float* const out = a.g_pos;
  const float* const add = a.gpos_add;
  const size_t gpos_n = (size_t)a.B * a.H * 3 * a.N;
  if (nseg > 1) {       // point segments (see the 2D form)
    a.nseg = nseg; a.Nrow = a.N; a.N = a.N / nseg;
    dim3 wgrid(1, a.H, a.B * nseg);
    if (hot_wide(hp.lds, a.N >> 2)) CT_LAUNCH_HOT3WIDE_(splat_max_bwd_hot3_kernel, 0, wgrid, round_threads(a.N >> 2), hp.lds, st, a, g);
    else CT_LAUNCH_HOT3W_(splat_max_bwd_hot3_kernel, 0, wgrid, hot_threads(a.N >> 2), hp.lds, st, a, g);
    note("splat_max_bwd_hot3_segments");
    return CT_OK;
  }
  const bool fold = ncg > 1 && tickets_cover(a.tickets, (long long)a.B * a.H, ncg, 1) && fold_pays((size_t)ncg * 3 * a.N * 4);
  if (!fold) a.tickets = nullptr;
  if (ncg > 1) {
    a.g_pos = (float*)ws;
    a.gpos_stride = gpos_n;
    a.gpos_add = nullptr;
    a.fold_gpos = out; a.fold_add = add;
  }
  dim3 wgrid(ncg, a.H, a.B);
  const int nq = a.N >> 2;
  // always the loop form (g_keys partials stored per chunk): keeping a quad's 12 g_keys values in registers across the
  // chunks makes the 3D kernel spill
  if (hot_wide(hp.lds, nq)) CT_LAUNCH_HOT3WIDE_(splat_max_bwd_hot3_kernel, 0, wgrid, round_threads(nq), hp.lds, st, a, g);
  else CT_LAUNCH_HOT3W_(splat_max_bwd_hot3_kernel, 0, wgrid, hot_threads(nq), hp.lds, st, a, g);
  note(ncg > 1 ? "splat_max_bwd_hot3_groups" : "splat_max_bwd_hot3");
  if (hot_wide(hp.lds, nq)) note("wide");
  if (fold) {
    note("folded");
    return CT_OK;
  }
  if (ncg > 1) {
    CT_CLEAR_ERROR();
    if (launch_sum_parts((const float*)ws, out, gpos_n, gpos_n, ncg, add, st) != CT_OK) return CT_ELAUNCH;
    CT_CHECK_LAUNCH();
  }
  return CT_OK;
}

// ---- banded kernels (ct_raster_band.h): four-channel heads on grids too large for the hot kernels ----
// R rows per band: as many as keep two workgroups per CU (else one), but bands enough to fill the chip's 512 slots.
// Measured on the zoo's C4 heads (tools/zoo_sweep.py, B8 N4096 / B2 N16384): 128^2 Slice backward 34 -> 52 / 37 -> 41 us, Splat
// backward 29 -> 57 / 36 -> 36; 32^3 73 -> 179 and 57 -> 227: every band streams the WHOLE plane's keys (and g_out, for the
// maxima) through L2 and the scattered 4-byte fetches move 64-byte sectors, nb x the point traffic where the single-channel
// kernels re-read only the keys (C x).  So the bands serve what nothing else serves well: grids whose single-channel tile
// exceeds a CU's LDS (> 40K cells: 256^2, 64^3 ...), which otherwise run on global atomics, nondeterministically.
template <int DIM>
bool band_plan(const RasterArgs& a, const GridW<DIM>& g, uintptr_t ptr_bits, int& R, int& nb, size_t& lds) {
  const int S = g.G / g.W[0];
  if (!hot_enabled() || a.C != 4 || (a.N & 3) != 0 || (S & 3) != 0 || (ptr_bits & 15) != 0 || g.W[0] < 4) return false;
  if ((size_t)g.G * 4 <= (size_t)kBigLdsBytes && !(t_dbg_flags & CT_DEBUG_FORCE_BAND)) return false;
  const size_t fixed = (size_t)(kBandCnt + 8) * 4 + band_ring_bytes<DIM>();
  auto bytes = [&](int r) { return (size_t)(r + 2) * S * 32 + fixed; };
  int r = (int)(((size_t)kHalfCuLdsBytes - fixed) / ((size_t)S * 32)) - 2;
  if (r < 2) r = (int)(((size_t)kBigLdsBytes - fixed) / ((size_t)S * 32)) - 2;
  if (r < 1) return false;
  const long long planes = (long long)a.B * a.H;
  const int want_nb = (int)((512 + planes - 1) / planes);
  const int r_fill = (g.W[0] + want_nb - 1) / want_nb;
  if (r > r_fill) r = r_fill < 2 ? 2 : r_fill;
  if (r > g.W[0]) r = g.W[0];
  R = r;
  nb = (g.W[0] + r - 1) / r;
  lds = bytes(r);
  return lds <= (size_t)kBigLdsBytes;
}

template <int DIM>
int run_band_slice_bwd(RasterArgs a, const float* grid, float* g_pos, const GridW<DIM>& g, hipStream_t st) {
  const uintptr_t bits = (uintptr_t)a.pos.keys | (uintptr_t)a.src | (uintptr_t)a.tile_out | (uintptr_t)grid | (uintptr_t)g_pos;
  int R, nb;
  size_t lds;
  if (!band_plan<DIM>(a, g, bits, R, nb, lds)) return CT_EINVAL;
  a.tile_in = grid; a.g_pos = g_pos;
  a.nsplit = R; a.ncg = nb; a.tickets = nullptr;
  dim3 wgrid(nb, a.H, a.B);
  if (a.pad_dtype != CT_PAD_NONE) CT_LAUNCH((band_slice_bwd_kernel<DIM, true>), wgrid, kBandThreads, lds, st, a, g);
  else CT_LAUNCH((band_slice_bwd_kernel<DIM, false>), wgrid, kBandThreads, lds, st, a, g);
  note(DIM == 2 ? "band_slice_bwd" : "band_slice_bwd3");
  return CT_OK;
}

// (needs an incoming key cotangent that is not the output: a band with exact ties is redone from it)
template <int DIM>
int run_band_splat_bwd(RasterArgs a, const GridW<DIM>& g, hipStream_t st) {
  const uintptr_t bits = (uintptr_t)a.pos.keys | (uintptr_t)a.src | (uintptr_t)a.dst | (uintptr_t)a.g_pos | (uintptr_t)a.tile_in |
                         (uintptr_t)a.tile_in2 | (uintptr_t)a.gpos_add;
  int R, nb;
  size_t lds;
  if ((a.gpos_add != nullptr && a.gpos_add == a.g_pos) || !band_plan<DIM>(a, g, bits, R, nb, lds)) return CT_EINVAL;
  a.nsplit = R; a.ncg = nb; a.tickets = nullptr;
  dim3 wgrid(nb, a.H, a.B);
  if (a.pad_dtype != CT_PAD_NONE) CT_LAUNCH((band_splat_bwd_kernel<DIM, true>), wgrid, kBandThreads, lds, st, a, g);
  else CT_LAUNCH((band_splat_bwd_kernel<DIM, false>), wgrid, kBandThreads, lds, st, a, g);
  note(DIM == 2 ? "band_splat_bwd" : "band_splat_bwd3");
  return CT_OK;
}

// y += x
__global__ void __launch_bounds__(256) add_inplace_kernel(float* y, const float* x, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += x[i];
}

template <int DIM, bool FROM_KEYS>
int run_gather(RasterArgs a, const int* W, hipStream_t st) {
  GridW<DIM> g = make_grid<DIM>(W);
  Plan p = make_plan(a.B, a.H, a.C, a.N, g.G, 1);
  a.CC = p.CC;
  a.nchunks = p.nchunks;
  a.nsplit = pick_nsplit(a.B, a.H, p.nchunks, a.N, DIM);
  int threads = round_threads((a.N + a.nsplit - 1) / a.nsplit);
  dim3 grid(p.nchunks * a.nsplit, a.H, a.B);
  if constexpr (FROM_KEYS) {
    const int r = run_gather_hot(a, g, st);
    if (r != CT_EINVAL) return r;
  }
  if constexpr (DIM == 2) {   // 3D exceeds the register budget of the quad form: generic kernel
    if (quad_ok(a, FROM_KEYS, p.lds_tile) && (g.G & 3) == 0) {
      note("gather_quad");
      a.ncg = p.nchunks;   // one chunk per workgroup
      CT_LAUNCH_QUAD((2, QM_GATHER, CT_QUAD_CG, CT_QUAD_THREADS, false), grid, quad_threads(a.N, a.nsplit), p.lds_bytes, st, a, g);
      return CT_OK;
    }
  }
  if (p.lds_tile) CT_LAUNCH((gather_kernel<DIM, FROM_KEYS, true>), grid, threads, p.lds_bytes, st, a, g);
  else CT_LAUNCH((gather_kernel<DIM, FROM_KEYS, false>), grid, threads, 0, st, a, g);
  note("gather_generic");
  return CT_OK;
}

template <int DIM, bool FROM_KEYS>
size_t gpos_bytes(const RasterArgs& a) {
  return (size_t)a.B * a.H * (FROM_KEYS ? DIM : (1 << DIM)) * a.N * 4;
}

template <int DIM, bool FROM_KEYS>
int run_gather_gw(RasterArgs a, const int* W, hipStream_t st) {
  GridW<DIM> g = make_grid<DIM>(W);
  Plan p = make_plan(a.B, a.H, a.C, a.N, g.G, 1);
  a.CC = p.CC;
  a.nchunks = p.nchunks;
  a.nsplit = pick_nsplit(a.B, a.H, 1, a.N);
  a.ncg = 1;
  while ((long long)a.B * a.H * a.nsplit * a.ncg < 512 && a.ncg * 2 <= p.nchunks) a.ncg *= 2;
  a.atomic_gpos = a.ncg > 1;
  if (a.atomic_gpos && hipMemsetAsync(a.g_pos, 0, gpos_bytes<DIM, FROM_KEYS>(a), st) != hipSuccess) return CT_ELAUNCH;
  int threads = round_threads((a.N + a.nsplit - 1) / a.nsplit);
  dim3 grid(a.ncg * a.nsplit, a.H, a.B);
  if (quad_ok(a, FROM_KEYS, p.lds_tile) && (g.G & 3) == 0) {
    CT_LAUNCH_QUAD((DIM, QM_GATHER_GW, (DIM == 2 ? CT_QUAD_CG : 2), CT_QUAD_THREADS, false), grid, quad_threads(a.N, a.nsplit), p.lds_bytes, st, a, g);
    note("gather_gw_quad");
    return CT_OK;
  }
  if (p.lds_tile) CT_LAUNCH((gather_gw_kernel<DIM, FROM_KEYS, true>), grid, threads, p.lds_bytes, st, a, g);
  else CT_LAUNCH((gather_gw_kernel<DIM, FROM_KEYS, false>), grid, threads, 0, st, a, g);
  note("gather_gw_generic");
  return CT_OK;
}

// channel-chunk groups a Splat(max) backward launch is split into (more workgroups for few (b,h) planes)
inline int splat_bwd_ncg(int B, int H, int nchunks) {
  int ncg = 1;
  while ((long long)B * H * ncg < 512 && ncg * 2 <= nchunks) ncg *= 2;
  return ncg;
}

// the launches of Splat(max) backward; `parts` (set by the caller) tells whether the kernels wrote per-group partial
// g_keys that still have to be summed — the whole-head form clears it
template <int DIM, bool FROM_KEYS>
int launch_splat_max_bwd(RasterArgs a, const GridW<DIM>& g, const Plan& p, bool two, float* g_pos_out, bool& parts, void* ws,
                         size_t ws_bytes, int accumulate, hipStream_t st) {
  dim3 grid(a.ncg, a.H, a.B);
  if (!p.lds_tile) {
    size_t need = (size_t)a.B * a.H * a.C * g.G * 4;
    if (!ws || ws_bytes < need) return CT_EWORKSPACE;
    if (hipMemcpyAsync(ws, a.tile_in, need, hipMemcpyDeviceToDevice, st) != hipSuccess) return CT_ELAUNCH;
    a.claim = (unsigned*)ws;
    CT_LAUNCH((splat_max_bwd_kernel<DIM, FROM_KEYS, false, false>), grid, p.threads, 0, st, a, g);
    return CT_OK;
  }
  if constexpr (DIM == 2) {
#ifndef CT_NO_WHOLE_HEAD
    // whole-head form: z and g_z tiles of ALL channels of a (b,h) plane resident (<= 160 KiB,
    // one 1024-thread workgroup per CU): one staging pass, keys read once, g_keys written once
    const size_t wh_bytes = (size_t)2 * a.C * g.G * 4;
    if (two && wh_bytes <= (size_t)kBigLdsBytes && wh_bytes > (size_t)kMaxLdsBytes && (long long)a.B * a.H >= 256 &&
        quad_ok(a, FROM_KEYS, true) && (g.G & 3) == 0) {
      a.CC = a.C; a.nchunks = 1; a.ncg = 1; a.atomic_gpos = 0;
      a.g_pos = g_pos_out; a.gpos_stride = 0; parts = false;
      a.accumulate = accumulate;
      int t = round_threads(a.N >> 2);
      dim3 wgrid(1, a.H, a.B);
      if (t > 512) CT_LAUNCH_QUAD((2, QM_SPLAT_MAX_BWD, CT_QUAD_CG, 1024, false), wgrid, 1024, wh_bytes, st, a, g);
      else CT_LAUNCH_QUAD((2, QM_SPLAT_MAX_BWD, CT_QUAD_CG, 512, false), wgrid, t, wh_bytes, st, a, g);
      note("splat_max_bwd_whole_head");
      return CT_OK;
    }
#endif
    if (two && quad_ok(a, FROM_KEYS, true) && (g.G & 3) == 0) {
      CT_LAUNCH_QUAD((2, QM_SPLAT_MAX_BWD, CT_QUAD_CG, CT_QUAD_THREADS, false), grid, quad_threads(a.N, 1), p.lds_bytes, st, a, g);
      note("splat_max_bwd_quad");
      return CT_OK;
    }
  }
  if (two) CT_LAUNCH((splat_max_bwd_kernel<DIM, FROM_KEYS, true, true>), grid, p.threads, p.lds_bytes, st, a, g);
  else CT_LAUNCH((splat_max_bwd_kernel<DIM, FROM_KEYS, true, false>), grid, p.threads, p.lds_bytes, st, a, g);
  note("splat_max_bwd_generic");
  return CT_OK;
}

template <int DIM, bool FROM_KEYS>
int run_splat_max_bwd(RasterArgs a, const int* W, void* ws, size_t ws_bytes, hipStream_t st) {
  GridW<DIM> g = make_grid<DIM>(W);
  if constexpr (FROM_KEYS) {
    if (t_dbg_flags & CT_DEBUG_FORCE_BAND) {
      const int r = run_band_splat_bwd<DIM>(a, g, st);
      if (r != CT_EINVAL) return r;
    }
    HotPlan hp;
    bool single = false;
    int ncg = 1, nseg = 1;
    if (splat_bwd_hot_plan<DIM>(a, g, hp, ncg, nseg, single)) {
      // In-place accumulation (gpos_add == g_pos) by ONE group whose g_keys sums go through memory (2D beyond the register
      // form, every 3D call): the optimistic pass leaves incoming + its result in g_pos chunk by chunk, and a plane with exact
      // ties would start its single-winner redo from THAT (incoming + optimistic + claims).  Not eligible: the
      // caller computes the plain cotangent into its scratch (this very kernel, not accumulating) and adds it.
      const bool reg_form = DIM == 2 && (a.N >> 2) <= 2 * kHotThreads;
      if (a.gpos_add != nullptr && a.gpos_add == a.g_pos && ncg == 1 && nseg == 1 && !reg_form) return CT_EINVAL;
      const size_t need = ncg > 1 ? (size_t)ncg * a.B * a.H * DIM * a.N * 4 : 0;
      // several groups: the partial sums go through the workspace and the final sum adds the incoming cotangent; one
      // group (or point segments): the kernel adds it in its own store — every thread owns its rows
      if (need <= ws_bytes && (ws || !need) && ((uintptr_t)ws & 15) == 0)
        return run_splat_max_bwd_hot(a, g, hp, ncg, nseg, ws, st);
    }
    a.tickets = nullptr;
    if (!(t_dbg_flags & CT_DEBUG_NO_BAND)) {
      const int r = run_band_splat_bwd<DIM>(a, g, st);
      if (r != CT_EINVAL) return r;
    }
  }
  if (a.gpos_add != nullptr && a.gpos_add != a.g_pos) return CT_EINVAL;     // only the hot / banded kernels add from another tensor
  // z and g_z tiles both in LDS when two single-channel tiles fit the 64 KiB budget
  const bool two = (size_t)g.G * 8 <= (size_t)kMaxLdsBytes;
  Plan p = make_plan(a.B, a.H, a.C, a.N, g.G, two ? 2 : 1);
  a.CC = p.CC;
  a.nchunks = p.nchunks;
  a.nsplit = 1;
  a.ncg = splat_bwd_ncg(a.B, a.H, p.nchunks);
  // g_keys sums over the channel chunks.  Across chunk GROUPS (different workgroups) that sum went through
  // device-scope float atomics — a third of this kernel's time on the 64^2 C16 zoo head.  With the caller's
  // workspace each group stores its partial and sum_parts_kernel adds them in a fixed order.
  // a.accumulate (g_keys += result): a workgroup that owns its points adds in its own store, the atomic form skips
  // the zero fill, the partial-sum form accumulates in the final sum.
  float* const g_pos_out = a.g_pos;
  const int accumulate = a.accumulate;
  const size_t gpos_n = gpos_bytes<DIM, FROM_KEYS>(a) / 4;
  bool parts = a.ncg > 1 && p.lds_tile && ws && ws_bytes >= (size_t)a.ncg * gpos_n * 4;
  if (parts) { a.g_pos = (float*)ws; a.gpos_stride = gpos_n; a.accumulate = 0; }
  a.atomic_gpos = a.ncg > 1 && !parts;
  if (a.atomic_gpos && !accumulate && hipMemsetAsync(a.g_pos, 0, gpos_bytes<DIM, FROM_KEYS>(a), st) != hipSuccess) return CT_ELAUNCH;
  const int ncg = a.ncg;
  const int r = launch_splat_max_bwd<DIM, FROM_KEYS>(a, g, p, two, g_pos_out, parts, ws, ws_bytes, accumulate, st);
  if (r != CT_OK) return r;
  if (parts) {
    CT_CLEAR_ERROR();
    if (launch_sum_parts((const float*)ws, g_pos_out, gpos_n, gpos_n, ncg, accumulate ? g_pos_out : nullptr, st) != CT_OK) return CT_ELAUNCH;
    CT_CHECK_LAUNCH();
  }
  return CT_OK;
}


// splits of N of the gather + statistics kernel of the two-kernel Slice backward (few planes: more workgroups)
int slice_bwd_stats_nsplit(int B, int H, int N) {
  int ns = 1;
  while ((long long)B * H * ns < 256 && (N >> 2) / (ns * 2) >= 128) ns *= 2;
  return ns;
}

// Slice backward, hot path: returns CT_EINVAL when the shape is not eligible.
template <int DIM>
int run_slice_bwd_fast(RasterArgs a, const float* grid, float* g_pos, const int* W, void* ws, size_t ws_bytes, hipStream_t st) {
  GridW<DIM> g = make_grid<DIM>(W);
  if ((g.G & 3) != 0) return CT_EINVAL;
  if (t_dbg_flags & CT_DEBUG_FORCE_BAND) {       // tests: the banded kernel on grids the hot kernels would take
    const int r = run_band_slice_bwd<DIM>(a, grid, g_pos, g, st);
    if (r != CT_EINVAL) return r;
  }
  {
    const int r = run_slice_bwd_hot(a, grid, g_pos, g, ws, ws_bytes, st);
    if (r != CT_EINVAL) return r;
  }
  if (!(t_dbg_flags & CT_DEBUG_NO_BAND)) {
    const int r = run_band_slice_bwd<DIM>(a, grid, g_pos, g, st);
    if (r != CT_EINVAL) return r;
  }
  // gather side: tile + [CC] maxima + [G] counters in LDS
  RasterArgs ga = a;
  ga.tile_in = grid; ga.g_pos = g_pos;     // ga.tile_out = g_grid receives the statistics
  if (!quad_ok(ga, true, true)) return CT_EINVAL;
  // gather side: the largest channel chunk that fits (parallelism comes from splitting N, which a
  // gather may do freely); scatter side: its own chunking (it owns whole tiles)
  Plan pg = make_plan(a.B, a.H, a.C, a.N, g.G, 1, /*want=*/1);
  Plan ps = make_plan(a.B, a.H, a.C, a.N, g.G, 1);
  if (!pg.lds_tile || !ps.lds_tile) return CT_EINVAL;
  // contributions per cell are counted in a power-of-two table indexed by cell & mask: one counter per cell when
  // that fits beside the tile, otherwise several cells share a counter — the maximum over the table is then still
  // an upper bound of the per-cell maximum, which is all K has to be (a looser bound costs log2 of the slack in
  // the 30-bit fixed-point resolution)
  // The table is also kept small enough for TWO workgroups per CU: a full-resolution table for 128^2 cells (64 KiB beside
  // the 64 KiB tile) left one workgroup per CU that spent its time zeroing and scanning counters (a 16x looser K costs 4 of
  // the 30 bits).
  int cnt = 1024;
  while (cnt < g.G) cnt *= 2;
  while (cnt > 1024 && pg.lds_bytes + (size_t)(pg.CC * 32 + cnt) * 4 > (size_t)kHalfCuLdsBytes) cnt /= 2;
  const size_t extra = (size_t)(pg.CC * 32 + cnt) * 4;
  if (pg.lds_bytes + extra > (size_t)kBigLdsBytes) return CT_EINVAL;
  ga.cnt_mask = cnt - 1;
  ga.CC = pg.CC; ga.nchunks = pg.nchunks; ga.ncg = 1; ga.atomic_gpos = 0;
  ga.nsplit = slice_bwd_stats_nsplit(a.B, a.H, a.N);
  const size_t stats_bytes = (size_t)ga.nsplit * a.B * a.H * a.C * 2 * 4;
  unsigned* stats = nullptr;
  if (ga.nsplit > 1) {
    if (ws != nullptr && ws_bytes >= stats_bytes && ((uintptr_t)ws & 3) == 0) {
      stats = (unsigned*)ws;             // every split writes its own statistics: nothing to clear, no atomics
    } else {
      // no scratch: the statistics are combined across the splits with atomics on the slots (first two words of every
      // channel tile of g_grid), which a launch has to zero first
      // (a kernel rather than hipMemset2DAsync: the 2-D memset node crashed HIP-graph capture on ROCm 7.0/7.2)
      const size_t rows = (size_t)a.B * a.H * a.C;
      CT_CLEAR_ERROR();
      hipLaunchKernelGGL(zero_slots_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, a.tile_out, (size_t)g.G, rows);
      CT_CHECK_LAUNCH();
    }
  }
  ga.stats = stats;
  dim3 ggrid(ga.nsplit, a.H, a.B);
  if constexpr (DIM == 2)
    CT_LAUNCH_QUAD((2, QM_GATHER_GW, CT_QUAD_CG, CT_QUAD_THREADS, true), ggrid, quad_threads(a.N, ga.nsplit), pg.lds_bytes + extra, st, ga, g);
  else
    CT_LAUNCH_QUAD((3, QM_GATHER_GW, 2, CT_QUAD_THREADS, true), ggrid, quad_threads(a.N, ga.nsplit), pg.lds_bytes + extra, st, ga, g);
  // scatter side
  RasterArgs sa = a;
  sa.CC = ps.CC; sa.nchunks = ps.nchunks;
  sa.stats = stats; sa.nsplit = ga.nsplit;
  const int sthreads = ps.threads;   // (finer chunks / smaller workgroups were measured: no gain)
  dim3 sgrid(sa.nchunks, a.H, a.B);
  note(ga.nsplit > 1 ? (stats != nullptr ? "slice_bwd_gw_stats_parts" : "slice_bwd_gw_stats_nsplit") : "slice_bwd_gw_stats");
  if (scatter_quad_ok(sa, true, g.G)) {
    const int qt = scatter_quad_threads(DIM, a.N);
    if (sa.pad_dtype != CT_PAD_NONE) CT_LAUNCH((scatter_quad_kernel<DIM, true, true>), sgrid, qt, (size_t)sa.CC * g.G * 4 + (size_t)sa.CC * 8, st, sa, g);
    else CT_LAUNCH((scatter_quad_kernel<DIM, true, false>), sgrid, qt, (size_t)sa.CC * g.G * 4 + (size_t)sa.CC * 8, st, sa, g);
    note("scatter_quad_add");
    return CT_OK;
  }
  CT_LAUNCH((scatter_add_fx_stream_kernel<DIM, true>), sgrid, sthreads, (size_t)sa.CC * g.G * 4 + (size_t)2 * sa.CC * 4, st, sa, g);
  note("scatter_add_fx_stream");
  return CT_OK;
}

bool valid_pad(const void* pad, int pad_dtype) {
  if (pad_dtype == CT_PAD_NONE) return true;
  return (pad_dtype == CT_PAD_F32 || pad_dtype == CT_PAD_I32) && pad != nullptr;
}

RasterArgs base_args(int B, int H, int C, int N, const void* pad, int pad_dtype) {
  RasterArgs a = {};
  a.B = B; a.H = H; a.C = C; a.N = N;
  a.pad = pad_dtype == CT_PAD_NONE ? nullptr : pad;
  a.pad_dtype = pad_dtype;
  a.nsplit = 1; a.ncg = 1; a.nchunks = 1; a.CC = C;
  return a;
}

template <bool FROM_KEYS>
int splat_fwd_impl(PosSrc pos, const float* feat, const void* pad, int pad_dtype, float* grid,
                   int B, int H, int C, int N, int dim, const int* W, int reduce, hipStream_t st) {
  if (!valid_common(B, H, C, N, dim, W) || !feat || !grid || !valid_pad(pad, pad_dtype)) return CT_EINVAL;
  if (reduce != CT_REDUCE_MAX0 && reduce != CT_REDUCE_SUM) return CT_EINVAL;
  note_reset();
  RasterArgs a = base_args(B, H, C, N, pad, pad_dtype);
  a.pos = pos; a.src = feat; a.tile_out = grid;
  return dim == 2 ? run_scatter<2, FROM_KEYS>(a, W, reduce == CT_REDUCE_SUM, st)
                  : run_scatter<3, FROM_KEYS>(a, W, reduce == CT_REDUCE_SUM, st);
}

// g_pos_add: null (g_pos = result), g_pos itself (g_pos += result: CT_BWD_ACCUMULATE_KEYS) or another tensor of the same
// shape (g_pos = g_pos_add + result: ct_splat_bwd_tk — the form the point segments of the hot kernel need)
template <bool FROM_KEYS>
int splat_bwd_impl(PosSrc pos, const float* feat, const void* pad, int pad_dtype, const float* grid,
                   const float* g_grid, float* g_feat, float* g_pos, void* ws, size_t ws_bytes,
                   int B, int H, int C, int N, int dim, const int* W, int reduce, hipStream_t st,
                   const float* g_pos_add = nullptr, void* tickets = nullptr) {
  if (!valid_common(B, H, C, N, dim, W) || !feat || !g_grid || !g_feat || !g_pos || !valid_pad(pad, pad_dtype)) return CT_EINVAL;
  note_reset();
  if (g_pos_add != nullptr) {
    // The hot Splat(max) backward adds in its own store; every other path computes the plain result — into the tail of the
    // workspace when the sum is in place — and adds.
    const bool in_place = g_pos_add == g_pos;
    const size_t gpos_n = (size_t)B * H * (FROM_KEYS ? dim : (1 << dim)) * N;
    if (in_place && (!ws || ws_bytes < gpos_n * 4)) return CT_EWORKSPACE;
    const size_t head = in_place ? ws_bytes - gpos_n * 4 : ws_bytes;
    if (reduce == CT_REDUCE_MAX0 && grid) {
      RasterArgs a = base_args(B, H, C, N, pad, pad_dtype);
      a.pos = pos; a.src = feat; a.dst = g_feat; a.g_pos = g_pos; a.tile_in = grid; a.tile_in2 = g_grid;
      a.accumulate = in_place ? 1 : 0;
      a.gpos_add = g_pos_add;
      a.tickets = (unsigned*)tickets;
      const int r = dim == 2 ? run_splat_max_bwd<2, FROM_KEYS>(a, W, ws, head, st) : run_splat_max_bwd<3, FROM_KEYS>(a, W, ws, head, st);
      if (r != CT_EINVAL) return r;
    }
    if constexpr (FROM_KEYS) {
      if (reduce == CT_REDUCE_SUM && dim == 2) {
        RasterArgs a = base_args(B, H, C, N, pad, pad_dtype);
        a.pos = pos; a.src = feat; a.dst = g_feat; a.g_pos = g_pos; a.tile_in = g_grid;
        a.gpos_add = g_pos_add;
        const int r = run_splat_sum_bwd_hot(a, make_grid<2>(W), st);
        if (r != CT_EINVAL) return r;
      }
    }
    float* tmp = in_place ? (float*)((char*)ws + head) : g_pos;
    const int r = splat_bwd_impl<FROM_KEYS>(pos, feat, pad, pad_dtype, grid, g_grid, g_feat, tmp, ws, head, B, H, C, N, dim, W,
                                            reduce, st, nullptr, tickets);
    if (r != CT_OK) return r;
    CT_CLEAR_ERROR();
    hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((gpos_n + 255) / 256)), dim3(256), 0, st, g_pos,
                       in_place ? (const float*)tmp : g_pos_add, gpos_n);
    CT_CHECK_LAUNCH();
    note("add_inplace");
    return CT_OK;
  }
  RasterArgs a = base_args(B, H, C, N, pad, pad_dtype);
  a.pos = pos; a.src = feat; a.dst = g_feat; a.g_pos = g_pos;
  if (reduce == CT_REDUCE_MAX0) {
    if (!grid) return CT_EINVAL;
    a.tile_in = grid; a.tile_in2 = g_grid;
    a.tickets = (unsigned*)tickets;
    return dim == 2 ? run_splat_max_bwd<2, FROM_KEYS>(a, W, ws, ws_bytes, st)
                    : run_splat_max_bwd<3, FROM_KEYS>(a, W, ws, ws_bytes, st);
  } else if (reduce == CT_REDUCE_SUM) {
    // linear op: g_feat = Slice(g_grid), gw = sum_c g_grid * feat
    a.tile_in = g_grid;
    if constexpr (FROM_KEYS) {
      if (dim == 2) {
        const int r = run_splat_sum_bwd_hot(a, make_grid<2>(W), st);
        if (r != CT_EINVAL) return r;
      }
    }
    int r = dim == 2 ? run_gather<2, FROM_KEYS>(a, W, st) : run_gather<3, FROM_KEYS>(a, W, st);
    if (r != CT_OK) return r;
    return dim == 2 ? run_gather_gw<2, FROM_KEYS>(a, W, st) : run_gather_gw<3, FROM_KEYS>(a, W, st);
  }
  return CT_EINVAL;
}

template <bool FROM_KEYS>
int slice_fwd_impl(PosSrc pos, const float* grid, const void* pad, int pad_dtype, float* out,
                   int B, int H, int C, int N, int dim, const int* W, hipStream_t st) {
  if (!valid_common(B, H, C, N, dim, W) || !grid || !out || !valid_pad(pad, pad_dtype)) return CT_EINVAL;
  note_reset();
  RasterArgs a = base_args(B, H, C, N, pad, pad_dtype);
  a.pos = pos; a.tile_in = grid; a.dst = out;
  return dim == 2 ? run_gather<2, FROM_KEYS>(a, W, st) : run_gather<3, FROM_KEYS>(a, W, st);
}

template <bool FROM_KEYS>
int slice_bwd_impl(PosSrc pos, const float* grid, const void* pad, int pad_dtype, const float* g_out,
                   float* g_grid, float* g_pos, int B, int H, int C, int N, int dim, const int* W, hipStream_t st,
                   void* ws = nullptr, size_t ws_bytes = 0, void* tickets = nullptr, const void* sorted = nullptr) {
  if (!valid_common(B, H, C, N, dim, W) || !grid || !g_out || !g_grid || !g_pos || !valid_pad(pad, pad_dtype)) return CT_EINVAL;
  note_reset();
  RasterArgs a = base_args(B, H, C, N, pad, pad_dtype);
  a.pos = pos; a.src = g_out; a.tile_out = g_grid;
  a.tickets = (unsigned*)tickets;          // only the hot kernels (run_slice_bwd_hot) fold; every other path ignores them
  a.sorted = (const unsigned char*)sorted; // only the sorted-plane kernels read it
  // (A single fused kernel — grid tile + accumulator tile of a whole (b,h) plane in LDS, one
  //  1024-thread workgroup per CU — was built twice and measured SLOWER than the pair below
  //  (105-124 us vs 92-99 us on the headline shape): the pass is co-bound by LDS atomics/reads
  //  and HBM, and one workgroup per CU overlaps the two worse than two streaming kernels do.)
  if (FROM_KEYS && (N & 3) == 0) {
    // fast path: g_keys first (its kernel also leaves the scatter's quantum statistics in g_grid),
    // then the streaming fixed-point scatter-add
    int r = dim == 2 ? run_slice_bwd_fast<2>(a, grid, g_pos, W, ws, ws_bytes, st) : run_slice_bwd_fast<3>(a, grid, g_pos, W, ws, ws_bytes, st);
    if (r != CT_EINVAL) return r;       // CT_EINVAL: shape not eligible, use the generic pair below
  }
  int r = dim == 2 ? run_scatter<2, FROM_KEYS>(a, W, true, st) : run_scatter<3, FROM_KEYS>(a, W, true, st);
  if (r != CT_OK) return r;
  a.tile_out = nullptr; a.tile_in = grid; a.g_pos = g_pos;
  return dim == 2 ? run_gather_gw<2, FROM_KEYS>(a, W, st) : run_gather_gw<3, FROM_KEYS>(a, W, st);
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int ct_abi_version(void) { return CT_ABI_VERSION; }
#ifdef CT_TIE_DEBUG
// experiments only (-DCT_TIE_DEBUG): what the Splat(max) backward's tie paths did since the last call (read and reset)
int ct_debug_tie_counters(unsigned* dst) {
  static const unsigned zero[16] = {};
  if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tie_dbg), sizeof(zero)) != hipSuccess) return CT_ELAUNCH;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_tie_dbg), zero, sizeof(zero)) == hipSuccess ? CT_OK : CT_ELAUNCH;
}
#endif
#ifdef CT_SORT_STAMPS
// experiments only (tools/dev/build_raster_exp.sh ... -DCT_SORT_STAMPS): the phase stamps of the sorted kernels' first workgroup
int ct_debug_sorted_stamps(unsigned long long* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_sorted_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? CT_OK : CT_ELAUNCH;
}
// {entry clock, exit clock, HW_ID, XCC_ID} of the first 4096 workgroups of the last stamped launch
int ct_debug_wg_stamps(unsigned long long* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wg_stamps), sizeof(unsigned long long) * 4096 * 4) == hipSuccess ? CT_OK : CT_ELAUNCH;
}
#endif

const char* ct_strerror(int status) {
  switch (status) {
    case CT_OK: return "ok";
    case CT_EINVAL: return "invalid argument";
    case CT_ELAUNCH: return "HIP launch error";
    case CT_EWORKSPACE: return "workspace too small";
    case CT_EPRECOND: return "reference precondition violated";
    default: return "unknown status";
  }
}

int ct_positions_fwd(const float* keys, float* lc, int64_t* idx, int B, int H, int N, int dim, const int* W, ct_stream_t s) {
  if (!valid_common(B, H, 1, N, dim, W) || !keys || !lc || !idx) return CT_EINVAL;
  hipStream_t st = (hipStream_t)s;
  CT_CLEAR_ERROR();
  size_t total = (size_t)B * H * N;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dim == 2) hipLaunchKernelGGL(positions_fwd_kernel<2>, grid, dim3(256), 0, st, keys, lc, (long long*)idx, B * H, N, make_grid<2>(W));
  else hipLaunchKernelGGL(positions_fwd_kernel<3>, grid, dim3(256), 0, st, keys, lc, (long long*)idx, B * H, N, make_grid<3>(W));
  CT_CHECK_LAUNCH();
  return CT_OK;
}

int ct_positions_bwd(const float* keys, const float* g_lc, float* g_keys, int B, int H, int N, int dim, const int* W, ct_stream_t s) {
  if (!valid_common(B, H, 1, N, dim, W) || !keys || !g_lc || !g_keys) return CT_EINVAL;
  hipStream_t st = (hipStream_t)s;
  CT_CLEAR_ERROR();
  size_t total = (size_t)B * H * N;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dim == 2) hipLaunchKernelGGL(positions_bwd_kernel<2>, grid, dim3(256), 0, st, keys, g_lc, g_keys, B * H, N, make_grid<2>(W));
  else hipLaunchKernelGGL(positions_bwd_kernel<3>, grid, dim3(256), 0, st, keys, g_lc, g_keys, B * H, N, make_grid<3>(W));
  CT_CHECK_LAUNCH();
  return CT_OK;
}

int ct_splat_fwd(const float* keys, const float* feat, const void* pad, int pad_dtype, float* grid,
                 int B, int H, int C, int N, int dim, const int* W, int reduce, ct_stream_t s) {
  if (!keys) return CT_EINVAL;
  PosSrc pos = {keys, nullptr, nullptr};
  return splat_fwd_impl<true>(pos, feat, pad, pad_dtype, grid, B, H, C, N, dim, W, reduce, (hipStream_t)s);
}

static size_t splat_bwd_workspace_generic(int B, int H, int C, int N, int dim, const int* W) {
  size_t G = 1;
  for (int j = 0; j < dim; ++j) G *= W[j];
  if (G * 4 > (size_t)kBigLdsBytes) return (size_t)B * H * C * G * 4;   // claim copy of z (tile does not fit LDS)
  // tile lives in LDS: partial g_keys / g_lc of the channel-chunk groups, if the launch is split into any
  const bool two = G * 8 <= (size_t)kMaxLdsBytes;
  const Plan p = make_plan(B, H, C, N, (int)G, two ? 2 : 1);
  const int ncg = splat_bwd_ncg(B, H, p.nchunks);
  return ncg > 1 ? (size_t)ncg * B * H * ((size_t)1 << dim) * N * 4 : 0;
}

size_t ct_splat_bwd_workspace_bytes(int B, int H, int C, int N, int dim, const int* W, int reduce) {
  if (!valid_common(B, H, C, N, dim, W) || reduce != CT_REDUCE_MAX0) return 0;
  size_t n = splat_bwd_workspace_generic(B, H, C, N, dim, W);
  {                     // the hot kernels' chunk groups (whichever family a call ends up on, the scratch is enough)
    const size_t hot = dim == 2 ? splat_bwd_hot_workspace(B, H, C, N, make_grid<2>(W)) : splat_bwd_hot_workspace(B, H, C, N, make_grid<3>(W));
    if (hot > n) n = hot;
  }
  return n;
}

int ct_splat_bwd(const float* keys, const float* feat, const void* pad, int pad_dtype, const float* grid,
                 const float* g_grid, float* g_feat, float* g_keys, void* ws, size_t ws_bytes,
                 int B, int H, int C, int N, int dim, const int* W, int reduce, ct_stream_t s) {
  if (!keys) return CT_EINVAL;
  PosSrc pos = {keys, nullptr, nullptr};
  return splat_bwd_impl<true>(pos, feat, pad, pad_dtype, grid, g_grid, g_feat, g_keys, ws, ws_bytes,
                              B, H, C, N, dim, W, reduce, (hipStream_t)s);
}

size_t ct_splat_bwd_ex_workspace_bytes(int B, int H, int C, int N, int dim, const int* W, int reduce, int flags) {
  size_t n = ct_splat_bwd_workspace_bytes(B, H, C, N, dim, W, reduce);
  if ((flags & CT_BWD_ACCUMULATE_KEYS) && valid_common(B, H, C, N, dim, W)) n += (size_t)B * H * dim * N * 4;
  return n;
}

int ct_splat_bwd_ex(const float* keys, const float* feat, const void* pad, int pad_dtype, const float* grid,
                    const float* g_grid, float* g_feat, float* g_keys, void* ws, size_t ws_bytes,
                    int B, int H, int C, int N, int dim, const int* W, int reduce, int flags, ct_stream_t s) {
  if (!keys || (flags & ~CT_BWD_ACCUMULATE_KEYS)) return CT_EINVAL;
  PosSrc pos = {keys, nullptr, nullptr};
  return splat_bwd_impl<true>(pos, feat, pad, pad_dtype, grid, g_grid, g_feat, g_keys, ws, ws_bytes,
                              B, H, C, N, dim, W, reduce, (hipStream_t)s, (flags & CT_BWD_ACCUMULATE_KEYS) ? g_keys : nullptr);
}

void ct_debug_set_flags(unsigned flags) { t_dbg_flags = flags; }

const char* ct_debug_last_launch(void) {
  static thread_local char copy[256];
  std::lock_guard<std::mutex> lk(t_last_mu);
  memcpy(copy, t_last, sizeof(copy));
  return copy;
}

int ct_slice_fwd(const float* keys, const float* grid, const void* pad, int pad_dtype, float* out,
                 int B, int H, int C, int N, int dim, const int* W, ct_stream_t s) {
  if (!keys) return CT_EINVAL;
  PosSrc pos = {keys, nullptr, nullptr};
  return slice_fwd_impl<true>(pos, grid, pad, pad_dtype, out, B, H, C, N, dim, W, (hipStream_t)s);
}

int ct_slice_bwd(const float* keys, const float* grid, const void* pad, int pad_dtype, const float* g_out,
                 float* g_grid, float* g_keys, int B, int H, int C, int N, int dim, const int* W, ct_stream_t s) {
  if (!keys) return CT_EINVAL;
  PosSrc pos = {keys, nullptr, nullptr};
  return slice_bwd_impl<true>(pos, grid, pad, pad_dtype, g_out, g_grid, g_keys, B, H, C, N, dim, W, (hipStream_t)s);
}

size_t ct_slice_bwd_workspace_bytes(int B, int H, int C, int N, int dim, const int* W) {
  if (!valid_common(B, H, C, N, dim, W)) return 0;
  const size_t hot = dim == 2 ? slice_bwd_hot_workspace(B, H, C, N, make_grid<2>(W)) : slice_bwd_hot_workspace(B, H, C, N, make_grid<3>(W));
  // the two-kernel form's per-split statistics (run_slice_bwd_fast): [nsplit][B*H*C][2] words
  const int ns = slice_bwd_stats_nsplit(B, H, N);
  const size_t stats = ns > 1 ? (size_t)ns * B * H * C * 8 : 0;
  return hot > stats ? hot : stats;
}

int ct_slice_bwd_ws(const float* keys, const float* grid, const void* pad, int pad_dtype, const float* g_out,
                    float* g_grid, float* g_keys, void* ws, size_t ws_bytes, int B, int H, int C, int N, int dim,
                    const int* W, ct_stream_t s) {
  if (!keys) return CT_EINVAL;
  PosSrc pos = {keys, nullptr, nullptr};
  return slice_bwd_impl<true>(pos, grid, pad, pad_dtype, g_out, g_grid, g_keys, B, H, C, N, dim, W, (hipStream_t)s, ws, ws_bytes);
}

void ct_debug_set_nseg(int nseg) { t_dbg_nseg.store(nseg, std::memory_order_relaxed); }

int ct_splat_bwd_tk_segments(int B, int H, int C, int N, int dim, const int* W) {
  if (!valid_common(B, H, C, N, dim, W) || !hot_enabled() || (N & 3) || (C & 3)) return 1;
  long long G = 1;
  for (int j = 0; j < dim; ++j) G *= W[j];
  if ((G & 3) || ((long long)B * H < 32 && !(t_dbg_flags & CT_DEBUG_FORCE_HOT)) || (long long)B * H > CT_TICKETS_BYTES / 8) return 1;
  HotPlan sp;
  if ((t_dbg_flags & CT_DEBUG_FORCE_BAND) && C == 4 && ((G / W[0]) & 3) == 0 && W[0] >= 4) return 2;
  if (!hot_chunks(C, (size_t)G * 8, kSplatBwdFixed, sp)) {
    // the banded kernels (ct_raster_band.h: grids beyond a CU's LDS) redo a band with exact ties from the incoming
    // cotangent: a second tensor too
    return (C == 4 && ((G / W[0]) & 3) == 0 && W[0] >= 4 && (size_t)G * 4 > (size_t)kBigLdsBytes && !(t_dbg_flags & CT_DEBUG_NO_BAND)) ? 2 : 1;
  }
  return splat_bwd_segments(B, H, C, N, (int)G, dim, sp.lds);
}

int ct_tickets_init(void* tickets, ct_stream_t s) {
  if (!tickets || ((uintptr_t)tickets & 3) != 0) return CT_EINVAL;
  return hipMemsetAsync(tickets, 0, CT_TICKETS_BYTES, (hipStream_t)s) == hipSuccess ? CT_OK : CT_ELAUNCH;
}

int ct_slice_bwd_tk(const float* keys, const float* grid, const void* pad, int pad_dtype, const float* g_out,
                    float* g_grid, float* g_keys, void* ws, size_t ws_bytes, void* tickets, int B, int H, int C, int N, int dim,
                    const int* W, ct_stream_t s) {
  if (!keys) return CT_EINVAL;
  PosSrc pos = {keys, nullptr, nullptr};
  return slice_bwd_impl<true>(pos, grid, pad, pad_dtype, g_out, g_grid, g_keys, B, H, C, N, dim, W, (hipStream_t)s, ws, ws_bytes,
                              tickets);
}

size_t ct_plane_sort_bytes(int B, int H, int N, int dim, const int* W) {
  if (B <= 0 || H <= 0 || N <= 0 || !W || dim != 2 || W[0] < 2 || W[1] < 2) return 0;
  const long long G = (long long)W[0] * W[1];
  if (G > 65535 || !plane_sort_ok(N, dim, (int)G)) return 0;
  return (size_t)B * H * sort_record_bytes(N);
}

int ct_plane_sort(const float* keys, void* sorted, size_t sorted_bytes, int B, int H, int N, int dim, const int* W, ct_stream_t s) {
  const size_t need = ct_plane_sort_bytes(B, H, N, dim, W);
  if (!keys || !sorted || need == 0 || sorted_bytes < need || (((uintptr_t)keys | (uintptr_t)sorted) & 15)) return CT_EINVAL;
  note_reset();
  const GridW<2> g = make_grid<2>(W);
  RasterArgs a = base_args(B, H, 4, N, nullptr, CT_PAD_NONE);
  a.pos = {keys, nullptr, nullptr};
  const size_t lds = plane_sort_lds(g.G, N);
  dim3 wgrid(1, H, B);
  if (g.W[0] == 32 && g.W[1] == 32)
    CT_LAUNCH((plane_sort_kernel<32>), wgrid, kSortThreads, lds, (hipStream_t)s, a, g, (unsigned char*)sorted, sort_record_bytes(N));
  else
    CT_LAUNCH((plane_sort_kernel<0>), wgrid, kSortThreads, lds, (hipStream_t)s, a, g, (unsigned char*)sorted, sort_record_bytes(N));
  note("plane_sort");
  return CT_OK;
}

int ct_slice_bwd_ps(const float* keys, const float* grid, const void* pad, int pad_dtype, const float* g_out,
                    float* g_grid, float* g_keys, void* ws, size_t ws_bytes, void* tickets, const void* sorted, int B, int H, int C,
                    int N, int dim, const int* W, ct_stream_t s) {
  if (!keys) return CT_EINVAL;
  if (sorted != nullptr && (ct_plane_sort_bytes(B, H, N, dim, W) == 0 || ((uintptr_t)sorted & 15))) return CT_EINVAL;
  PosSrc pos = {keys, nullptr, nullptr};
  return slice_bwd_impl<true>(pos, grid, pad, pad_dtype, g_out, g_grid, g_keys, B, H, C, N, dim, W, (hipStream_t)s, ws, ws_bytes,
                              tickets, sorted);
}

int ct_splat_bwd_tk(const float* keys, const float* feat, const void* pad, int pad_dtype, const float* grid,
                    const float* g_grid, float* g_feat, const float* g_keys_add, float* g_keys, void* ws, size_t ws_bytes,
                    void* tickets, int B, int H, int C, int N, int dim, const int* W, int reduce, ct_stream_t s) {
  if (!keys) return CT_EINVAL;
  PosSrc pos = {keys, nullptr, nullptr};
  return splat_bwd_impl<true>(pos, feat, pad, pad_dtype, grid, g_grid, g_feat, g_keys, ws, ws_bytes,
                              B, H, C, N, dim, W, reduce, (hipStream_t)s, g_keys_add, tickets);
}

int ct_slice_bwd_grid(const float* keys, const void* pad, int pad_dtype, const float* g_out, float* g_grid,
                      int B, int H, int C, int N, int dim, const int* W, ct_stream_t s) {
  // a scatter-add of g_out: the same pass as Splat(sum) forward
  return ct_splat_fwd(keys, g_out, pad, pad_dtype, g_grid, B, H, C, N, dim, W, CT_REDUCE_SUM, s);
}

int ct_slice_bwd_keys(const float* keys, const float* grid, const void* pad, int pad_dtype, const float* g_out,
                      float* g_keys, int B, int H, int C, int N, int dim, const int* W, ct_stream_t s) {
  if (!keys || !valid_common(B, H, C, N, dim, W) || !grid || !g_out || !g_keys || !valid_pad(pad, pad_dtype)) return CT_EINVAL;
  RasterArgs a = base_args(B, H, C, N, pad, pad_dtype);
  a.pos = PosSrc{keys, nullptr, nullptr};
  a.src = g_out; a.tile_in = grid; a.g_pos = g_keys;
  return dim == 2 ? run_gather_gw<2, true>(a, W, (hipStream_t)s) : run_gather_gw<3, true>(a, W, (hipStream_t)s);
}

int ct_splat_lc_fwd(const float* lc, const int64_t* idx, const float* feat, const void* pad, int pad_dtype,
                    float* grid, int B, int H, int C, int N, int dim, const int* W, int reduce, ct_stream_t s) {
  if (!lc || !idx) return CT_EINVAL;
  PosSrc pos = {nullptr, lc, (const long long*)idx};
  return splat_fwd_impl<false>(pos, feat, pad, pad_dtype, grid, B, H, C, N, dim, W, reduce, (hipStream_t)s);
}

int ct_splat_lc_bwd(const float* lc, const int64_t* idx, const float* feat, const void* pad, int pad_dtype,
                    const float* grid, const float* g_grid, float* g_feat, float* g_lc, void* ws, size_t ws_bytes,
                    int B, int H, int C, int N, int dim, const int* W, int reduce, ct_stream_t s) {
  if (!lc || !idx) return CT_EINVAL;
  PosSrc pos = {nullptr, lc, (const long long*)idx};
  return splat_bwd_impl<false>(pos, feat, pad, pad_dtype, grid, g_grid, g_feat, g_lc, ws, ws_bytes,
                               B, H, C, N, dim, W, reduce, (hipStream_t)s);
}

int ct_slice_lc_fwd(const float* lc, const int64_t* idx, const float* grid, const void* pad, int pad_dtype,
                    float* out, int B, int H, int C, int N, int dim, const int* W, ct_stream_t s) {
  if (!lc || !idx) return CT_EINVAL;
  PosSrc pos = {nullptr, lc, (const long long*)idx};
  return slice_fwd_impl<false>(pos, grid, pad, pad_dtype, out, B, H, C, N, dim, W, (hipStream_t)s);
}

int ct_slice_lc_bwd(const float* lc, const int64_t* idx, const float* grid, const void* pad, int pad_dtype,
                    const float* g_out, float* g_grid, float* g_lc, int B, int H, int C, int N, int dim,
                    const int* W, ct_stream_t s) {
  if (!lc || !idx) return CT_EINVAL;
  PosSrc pos = {nullptr, lc, (const long long*)idx};
  return slice_bwd_impl<false>(pos, grid, pad, pad_dtype, g_out, g_grid, g_lc, B, H, C, N, dim, W, (hipStream_t)s);
}

int ct_grid_occupancy(const float* grid, int64_t n, int64_t* count, ct_stream_t s) {
  if (!grid || !count || n < 0) return CT_EINVAL;
  hipStream_t st = (hipStream_t)s;
  if (hipMemsetAsync(count, 0, sizeof(int64_t), st) != hipSuccess) return CT_ELAUNCH;
  if (n == 0) return CT_OK;
  CT_CLEAR_ERROR();
  // one 1024-thread workgroup per CU at most: the per-workgroup atomics on the single counter serialise (~12 ns each)
  int blocks = (int)((n + 16383) / 16384);
  if (blocks > 256) blocks = 256;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(occupancy_kernel, dim3(blocks), dim3(1024), 0, st, grid, (long long)n, (unsigned long long*)count);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

int ct_grid_occupancy_ratio(const float* grid, int64_t n, float inv_denominator, float* out, void* workspace, ct_stream_t s) {
  if (!grid || !out || !workspace || n < 0 || n > (int64_t)0x7fffffff || ((uintptr_t)workspace & 7) != 0) return CT_EINVAL;
  hipStream_t st = (hipStream_t)s;
  CT_CLEAR_ERROR();
  int blocks = (int)((n + 16383) / 16384);
  if (blocks > kOccBlocks) blocks = kOccBlocks;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(occupancy_ratio_kernel, dim3(blocks), dim3(1024), 0, st, grid, (long long)n, inv_denominator, out,
                     (unsigned long long*)workspace);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

}  // extern "C"
