// Hot-shape kernels of the rasterize / de-rasterize backward passes (2D grids, corners from keys,
// N % 4 == 0, 16-byte aligned rows, C % 4 == 0).  Included by ct_raster.hip inside its anonymous
// namespace (uses RasterArgs, GridW, ct_axis, ct_corners ... from there).
//
// What bounds these passes is the LDS pipeline, not HBM (profiles/r1_bench_sq_counters.txt: 47-71 % of the
// LDS cycles of the round-1 kernels were bank conflicts of one-dword random reads).  Two changes of data
// layout cut the LDS cycles per point:
//   * gather side: the grid tile is kept CHANNEL-INTERLEAVED in LDS — 4 channels of a cell are one 16-byte
//     word — so one ds_read_b128 per corner fetches 4 channels (4 LDS cycles per conflict-free wave access for
//     1 KiB, against 2 cycles per 256 B for ds_read_b32: MI355X_MICROARCH.md §LDS), 16 reads per point
//     instead of 64;
//   * Splat(max) backward: z and g_z of a channel pair share one 16-byte word {z0, z1, g0, g1}; the match
//     test and the cotangent of BOTH channels come from one read, and the single-winner rule costs nothing
//     on tie-free planes: a plane without exact ties has exactly one bit-equal contribution per non-zero
//     cell, so the kernel counts matches against non-zero cells and only a plane where the counts differ
//     (duplicated points) is redone with the compare-and-swap claims.
// Slice backward is ONE kernel: g_out is read once into registers, feeds the g_keys gather and the
// fixed-point scatter-add, whose quantum is PER CHANNEL (max |g_out| of that channel in the plane, found by
// a block reduction of the register-resident group before its atomics are issued).
#pragma once

// exact power-of-two quantum of a slab whose sums are bounded by MK = max|src| * max contributions per cell
__device__ __forceinline__ void fx_quantum(float MK, float& q, float& iq, bool& fixed) {
  fixed = MK < 1e37f;                              // false for inf / NaN as well
  int ex = 0;
  if (fixed && MK > 0.0f) (void)frexpf(MK, &ex);   // MK <= 2^ex
  ex = max(ex, -90);
  q = ldexpf(1.0f, ex - 30);
  iq = ldexpf(1.0f, 30 - ex);
}

// wave64 reductions over DPP (no LDS traffic, no index registers): row_shr 1/2/4/8 leave each 16-lane row's result in
// its last lane, row_bcast:15 / row_bcast:31 carry it across the rows, lane 63 holds the result (returned as a
// wave-uniform value).  Lanes without a source read `old` = 0, the identity of both reductions.
#define CT_DPP_STEP(V, OP, CTRL, ROWMASK) \
  V = OP(V, (decltype(V))__builtin_amdgcn_update_dpp(0, (int)(V), CTRL, ROWMASK, 0xf, false))
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define CT_UMAX(a, b) max((a), (b))
  CT_DPP_STEP(v, CT_UMAX, 0x111, 0xf);
  CT_DPP_STEP(v, CT_UMAX, 0x112, 0xf);
  CT_DPP_STEP(v, CT_UMAX, 0x114, 0xf);
  CT_DPP_STEP(v, CT_UMAX, 0x118, 0xf);
  CT_DPP_STEP(v, CT_UMAX, 0x142, 0xa);
  CT_DPP_STEP(v, CT_UMAX, 0x143, 0xc);
#undef CT_UMAX
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ int wave_sum_i32(int v) {
#define CT_IADD(a, b) ((a) + (b))
  CT_DPP_STEP(v, CT_IADD, 0x111, 0xf);
  CT_DPP_STEP(v, CT_IADD, 0x112, 0xf);
  CT_DPP_STEP(v, CT_IADD, 0x114, 0xf);
  CT_DPP_STEP(v, CT_IADD, 0x118, 0xf);
  CT_DPP_STEP(v, CT_IADD, 0x142, 0xa);
  CT_DPP_STEP(v, CT_IADD, 0x143, 0xc);
#undef CT_IADD
  return __builtin_amdgcn_readlane(v, 63);
}

// float -> int, rounded to nearest (floor(x + 0.5): ties up) in ONE instruction; __float2int_rn is v_rndne_f32 +
// v_cvt_i32_f32.  The fixed-point scatter rounds every product once to the quantum; which way an exact tie goes
// does not matter.
__device__ __forceinline__ int cvt_rpi(float x) {
  int r;
  asm("v_cvt_rpi_i32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

typedef float ct_f2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
// Partial results of a (b, h) plane's workgroups, folded INSIDE the producing kernel (no sum_parts launch behind it).
// Few-plane launches deal a plane's channel chunks (ncg groups) or its points (nseg segments) to several workgroups; every
// group leaves a partial g_keys (every segment a partial g_grid tile) in the caller's workspace.  With arrival tickets
// (RasterArgs::tickets: zeroed once by ct_tickets_init, self-resetting) the workgroup whose ticket is the plane's last
// adds the partials in ascending order — the order, and so the bits, of sum_parts_kernel.  Hand-off as in ct_mhct.hip /
// MI355X_MICROARCH.md "Valid forms": write-through (sc1) stores of the partials -> every storing wave's vmcnt(0) ->
// workgroup barrier -> one relaxed agent-scope add -> the workgroup whose add returned the last value takes an agent
// acquire, waits for it, joins the barrier -> plain loads.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void st_sc1_4(float* p, float4 v) {
  const ct_f4 t = {v.x, v.y, v.z, v.w};
  // (the two wait states are the hazard the compiler cannot see behind inline asm: a vector write to the data registers of
  //  a store of more than 8 bytes right behind it — it corrupted single lanes of g_keys until the s_nop went in)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
// a partial that another workgroup will read goes out write-through, everything else as a streaming store
__device__ __forceinline__ void st_part4(float* p, float4 v, bool handoff) {
  if (handoff) st_sc1_4(p, v);
  else st_stream4(p, v);
}

constexpr int kTicketWords = CT_TICKETS_BYTES / 4;
constexpr int kTicketHalf = kTicketWords / 2;       // [0, half): g_keys folds, [half, 2 half): g_grid folds

// All threads of the workgroup call this after their partial stores.  Returns (block-uniform) bit 0: this workgroup is
// the last of `nk` to arrive at ticket `tk` (null: not taking part), bit 1: the same for `tg` / `ng`.
__device__ __forceinline__ unsigned arrive_last(unsigned* tk, unsigned nk, unsigned* tg, unsigned ng, unsigned* s_flag,
                                                bool plain_stores = false) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    if (plain_stores) {       // partials written with ordinary stores (kept in this XCD's L2): one agent-scope release
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    unsigned f = 0;
    if (tk && __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nk - 1u) {
      f |= 1u;
      __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // everybody has arrived: ready for the next launch
    }
    if (tg && __hip_atomic_fetch_add(tg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1u) {
      f |= 2u;
      __hip_atomic_store(tg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (f) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *s_flag = f;
  }
  __syncthreads();
  return *s_flag;
}

// out[i] = (add[i] +) parts[i] + parts[stride + i] + ... (k copies, ascending), n4 float4 — sum_parts_kernel's arithmetic
__device__ __forceinline__ void fold_rows(const float* parts, size_t stride, int k, float* out, int n4, const float* add) {
  for (int i = threadIdx.x; i < n4; i += blockDim.x) {
    float4 s = ((const float4*)parts)[i];
    for (int j = 1; j < k; ++j) {
      const float4 t = ((const float4*)(parts + (size_t)j * stride))[i];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    if (add != nullptr) {
      const float4 o = ((const float4*)add)[i];
      s.x = o.x + s.x; s.y = o.y + s.y; s.z = o.z + s.z; s.w = o.w + s.w;
    }
    ((float4*)out)[i] = s;
  }
}

// Workgroup -> (chunk group, point segment, head, cloud).  The launch order is x-fastest and consecutive workgroups go to
// the 8 XCDs in turn (observed; speed only), so the `per` = ncg * nseg workgroups of a plane — which share its keys, its
// conv tile and the partials the last of them folds — are given linear ids congruent mod 8: one XCD, one L2.
struct WgCoord {
  int cgi, seg, h, b;
};
__device__ __forceinline__ WgCoord wg_coord(int ncg, int nseg, int H, int B) {
  WgCoord w;
  const unsigned per = (unsigned)(ncg * nseg), planes = (unsigned)(H * B);
  if (per > 1 && (planes & 7u) == 0) {
    const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned grp = L / (8u * per), r = L - grp * 8u * per;
    const unsigned plane = grp * 8u + (r & 7u), k = r >> 3;
    w.cgi = (int)(k % (unsigned)ncg);
    w.seg = (int)(k / (unsigned)ncg);
    w.h = (int)(plane % (unsigned)H);
    w.b = (int)(plane / (unsigned)H);
  } else {
    w.cgi = blockIdx.x;
    w.h = blockIdx.y;
    w.b = blockIdx.z / nseg;
    w.seg = blockIdx.z - w.b * nseg;
  }
  return w;
}


// per-axis terms and corner weights of one 2D point
struct Pt2 {
  float w0x, w1x, w0y, w1y;
  float cw[4];
  int base;
};

__device__ __forceinline__ void pt2_from_keys(float kx, float ky, const GridW<2>& g, int W1, Pt2& p) {
  int fx, fy;
  ct_axis(kx, g.hw[0], g.W[0], p.w0x, p.w1x, fx);
  ct_axis(ky, g.hw[1], g.W[1], p.w0y, p.w1y, fy);
  p.base = fx * W1 + fy;
  p.cw[0] = p.w0x * p.w0y;      // corner order of ct_corners<2>: (0,0), (1,0), (0,1), (1,1)
  p.cw[1] = p.w1x * p.w0y;
  p.cw[2] = p.w0x * p.w1y;
  p.cw[3] = p.w1x * p.w1y;
}

// ---------------------------------------------------------------------------
// KF: Slice backward, fused.  One 512-thread workgroup per (b, h) plane (two per CU), thread = QPT quads of 4
//   consecutive points (N <= 4 * QPT * blockDim).
//   LDS: conv chunk [CC/4][G] x float4 (channel-interleaved) | int accumulators [CC][G] | base-cell counts [G]
//        | per-channel max |g_out*pad| [C] | K
//   per chunk of CC channels: stage + zero, then per group of 4 channels:
//     g_out group -> registers, per-channel block max (one barrier), then per point: 4 ds_read_b128 (conv at the
//     corners, 4 channels each) -> corner cotangents, and 16 ds_add_u32 of the rounded products.
//   grid = (1, H, B)
// ---------------------------------------------------------------------------
#ifndef CT_HOT_THREADS
#define CT_HOT_THREADS 512
#endif
constexpr int kHotThreads = CT_HOT_THREADS;
constexpr int kHotWideThreads = 1024;
#ifndef CT_FUSED_WAVES
#define CT_FUSED_WAVES 4
#endif

// IEEE float scatter-add of one channel of a plane into its LDS accumulator row (the channel holds inf / NaN or
// would overflow the fixed-point bound): re-reads the channel's src row; rare.
template <bool HAS_PAD>
__device__ __forceinline__ void scatter_float_channel(const RasterArgs& a, const GridW<2>& g, size_t bh, int b, int ch, float* row_acc,
                                                      int es = 1 /* element stride of the channel's cells in row_acc */,
                                                      size_t so = 0 /* start of the workgroup's point segment in its rows */) {
  const int N = a.N, W1 = g.W[1], Nr = a.Nrow > 0 ? a.Nrow : a.N;
  const float* src = a.src + (bh * a.C + ch) * (size_t)Nr + so;
  for (int q = threadIdx.x; q < (N >> 2); q += blockDim.x) {
    const int n0 = q << 2;
    const float4 tx = *(const float4*)(a.pos.keys + (bh * 2 + 0) * Nr + so + n0);
    const float4 ty = *(const float4*)(a.pos.keys + (bh * 2 + 1) * Nr + so + n0);
    const float4 tf = *(const float4*)(src + n0);
    const float kx[4] = {tx.x, tx.y, tx.z, tx.w}, ky[4] = {ty.x, ty.y, ty.z, ty.w}, f[4] = {tf.x, tf.y, tf.z, tf.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Pt2 p;
      pt2_from_keys(kx[i], ky[i], g, W1, p);
      const float x = HAS_PAD ? f[i] * ct_load_pad(a.pad, a.pad_dtype, (size_t)b * Nr + so + n0 + i) : f[i];
      float* T = row_acc + p.base * es;
      atomicAdd(T, x * p.cw[0]);
      atomicAdd(T + W1 * es, x * p.cw[1]);
      atomicAdd(T + es, x * p.cw[2]);
      atomicAdd(T + (W1 + 1) * es, x * p.cw[3]);
    }
  }
}

#ifndef CT_FUSED_PAIR
#define CT_FUSED_PAIR 1
#endif

// GATHER = false: the scatter-add alone (Splat(sum) forward, ct_slice_bwd_grid): no conv tile, no g_keys.
// NTB: the launch's thread bound — kHotWideThreads for the WIDE launches (one 1024-thread workgroup per CU where the tiles leave no
// room for a second 512-thread one: 16 waves per CU instead of 8, see hot_wide in ct_raster.hip)
template <bool HAS_PAD, int WT, int QPT, bool GATHER, int NTB = kHotThreads>
__global__ void __launch_bounds__(NTB, CT_FUSED_WAVES) slice_bwd_fused_kernel(RasterArgs a, GridW<2> g_arg) {
  const GridW<2> g = grid2_of<WT>(g_arg);
  extern __shared__ __align__(16) float lds[];
  // WT > 0: square WT x WT grid known at compile time (corner offsets become instruction immediates)
  const int G = WT ? WT * WT : g.G, W1 = WT ? WT : g.W[1], CC = a.CC, N = a.N;
  float4* T4 = (float4*)lds;
  int* acc = (int*)(lds + (GATHER ? (size_t)CC * G : 0));
  int* cnt = acc + (size_t)CC * G;
  unsigned* s_max = (unsigned*)(cnt + G);
  unsigned* s_k = s_max + a.C;
  // blockIdx.z = (cloud, point segment): with more points per plane than a workgroup's registers hold (N > 4096) or too
  // few planes, a plane's points are dealt to a.nseg workgroups, each scattering into its own partial tile (summed
  // afterwards in a fixed order: still bitwise reproducible); rows are Nr floats long, this segment starts at `so`
  const int nsg = a.nseg > 0 ? a.nseg : 1;
  const WgCoord wg = wg_coord(a.ncg, nsg, a.H, a.B);
  const int h = wg.h, b = wg.b, seg = wg.seg;
  const size_t bh = (size_t)b * a.H + h;
  const int Nr = a.Nrow > 0 ? a.Nrow : a.N;          // (0: a caller that knows no segments)
  const size_t so = (size_t)seg * a.N;
  const int tid = threadIdx.x;
  const int off[4] = {0, W1, 1, W1 + 1};
  // partials another workgroup folds (tickets) are stored write-through
  const bool fold_keys = GATHER && a.tickets != nullptr && a.ncg > 1, fold_grid = a.tickets != nullptr && nsg > 1;
  int n0[QPT], n0c[QPT];
  bool active[QPT];
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    n0[u] = (tid + u * (int)blockDim.x) << 2;
    active[u] = n0[u] < N;
    n0c[u] = active[u] ? n0[u] : 0;      // threads past the end load the first quad and ignore it
  }

  const float* keyx = a.pos.keys + (bh * 2 + 0) * Nr + so;      // wave-uniform row pointers
  const float* keyy = a.pos.keys + (bh * 2 + 1) * Nr + so;
  float pv[QPT][4];
#pragma unroll
  for (int u = 0; u < QPT; ++u)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      pv[u][i] = (HAS_PAD && active[u]) ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * Nr + so + n0[u] + i) : 1.0f;
  for (int i = tid; i < G + a.C + 1; i += blockDim.x) cnt[i] = 0;     // cnt, s_max, s_k are contiguous
  __syncthreads();
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    if (active[u]) {
      const float4 tx = *(const float4*)(keyx + n0[u]), ty = *(const float4*)(keyy + n0[u]);
      const float kx[4] = {tx.x, tx.y, tx.z, tx.w}, ky[4] = {ty.x, ty.y, ty.z, ty.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Pt2 p;
        pt2_from_keys(kx[i], ky[i], g, W1, p);
        atomicAdd(&cnt[p.base], 1);
      }
    }
  }
  __syncthreads();
  {
    // contributions per cell = points based at the cell and at its three lower neighbours; cells of the last
    // row / column are never a base, so the wrapped neighbours of column 0 read zeros
    unsigned kloc = 0;
    for (int X = tid; X < G; X += blockDim.x) {
      unsigned c = (unsigned)cnt[X];
      if (X >= 1) c += (unsigned)cnt[X - 1];
      if (X >= W1) c += (unsigned)cnt[X - W1];
      if (X >= W1 + 1) c += (unsigned)cnt[X - W1 - 1];
      kloc = max(kloc, c);
    }
    kloc = wave_max_u32(kloc);
    if ((tid & 63) == 0) atomicMax(s_k, kloc);
  }

  float gs[QPT][4][2];
#pragma unroll
  for (int u = 0; u < QPT; ++u)
#pragma unroll
    for (int i = 0; i < 4; ++i) gs[u][i][0] = gs[u][i][1] = 0.0f;

  // Few (b,h) planes (the zoo's H16 blocks): the chunks of a plane are dealt to a.ncg workgroups (blockIdx.x), each
  // writing its partial g_keys to its own slice of the workspace (summed afterwards in a fixed order).
  const int cgi = wg.cgi;
  for (int chunk = cgi; chunk < a.nchunks; chunk += a.ncg) {
    const int c0 = chunk * CC;
    const int cc = min(CC, a.C - c0);            // multiple of 4
    const float* gin = a.tile_in + (bh * a.C + c0) * (size_t)G;
    float* gout = a.tile_out + (((size_t)seg * a.B * a.H + bh) * a.C + c0) * (size_t)G;
    // stage the conv chunk channel-interleaved (4 coalesced dword loads -> one conflict-free ds_write_b128)
    if (GATHER) {
      for (int t = tid; t < (cc >> 2) * G; t += blockDim.x) {
        const int cq = t / G, cell = t - cq * G;
        const float* p = gin + (size_t)(cq * 4) * G + cell;
        T4[t] = make_float4(ld_stream(p), ld_stream(p + G), ld_stream(p + 2 * (size_t)G), ld_stream(p + 3 * (size_t)G));
      }
    }
    if (chunk == cgi)
      for (int t = tid; t < (cc * G) >> 2; t += blockDim.x) ((int4*)acc)[t] = make_int4(0, 0, 0, 0);
    __syncthreads();
    const float Kf = (float)(*s_k);
    for (int cq = 0; cq < (cc >> 2); ++cq) {
      const int ch0 = c0 + cq * 4;
      // Quad 0's values stay in registers across the barrier; a second quad is read twice — now for the maxima,
      // and again when it is processed (an L2 hit: the plane's group was fetched microseconds ago) — so that the
      // live set stays inside 128 registers.
      float fv[4][4];            // [channel][point] of the quad being processed
      float mx[4];
#pragma unroll
      for (int cj = 0; cj < 4; ++cj) {
        const float* row = a.src + (bh * a.C + ch0 + cj) * (size_t)Nr + so;      // wave-uniform
        const float4 t = ld_stream4(row + n0c[0]);                         // read once
        fv[cj][0] = t.x; fv[cj][1] = t.y; fv[cj][2] = t.z; fv[cj][3] = t.w;
      }
#pragma unroll
      for (int cj = 0; cj < 4; ++cj) {
        float m = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float x = HAS_PAD ? fv[cj][i] * pv[0][i] : fv[cj][i];
          x = active[0] ? x : 0.0f;               // threads past the end loaded the cloud's first quad
          fv[cj][i] = x;
          x = fabsf(x);
          m = fmaxf(m, (x < __builtin_inff()) ? x : __builtin_inff());     // inf / NaN -> inf
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
        const unsigned mb = wave_max_u32(__float_as_uint(mx[cj]));           // non-negative floats order like uints
        if ((tid & 63) == 0) atomicMax(&s_max[ch0 + cj], mb);
      }
      __syncthreads();
      // quantum per channel; a channel that holds inf / NaN (or overflows the bound) adds zeros here (iq = 0:
      // x * 0 is 0 or NaN, both convert to 0) and is accumulated with IEEE float atomics in the rare block below
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
      // channel pairs share a 64-bit accumulator word: a pair with a float-path channel takes the float path whole
      if (iq[0] == 0.0f || iq[1] == 0.0f) iq[0] = iq[1] = 0.0f;
      if (iq[2] == 0.0f || iq[3] == 0.0f) iq[2] = iq[3] = 0.0f;
#endif
      const float4* Tq = T4 + (size_t)cq * G;
      int* accq = acc + (size_t)(cq * 4) * G;
#pragma unroll
      for (int u = 0; u < QPT; ++u) {
        if (u > 0) {
          int n0r = n0c[u];
          asm volatile("" : "+v"(n0r));      // a real second load: keeps the compiler from carrying the first one's 16 values
#pragma unroll
          for (int cj = 0; cj < 4; ++cj) {
            const float* row = a.src + (bh * a.C + ch0 + cj) * (size_t)Nr + so;
            const float4 t = ld_stream4(row + n0r);                        // second and last read
            fv[cj][0] = t.x; fv[cj][1] = t.y; fv[cj][2] = t.z; fv[cj][3] = t.w;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float x = HAS_PAD ? fv[cj][i] * pv[u][i] : fv[cj][i];
              fv[cj][i] = active[u] ? x : 0.0f;
            }
          }
        }
        // keys are re-read per (group, quad) — L2 hits — rather than held in 8 registers per quad
        const float4 tx = *(const float4*)(keyx + n0c[u]), ty = *(const float4*)(keyy + n0c[u]);
        const float kx[4] = {tx.x, tx.y, tx.z, tx.w}, ky[4] = {ty.x, ty.y, ty.z, ty.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          Pt2 p;
          pt2_from_keys(kx[i], ky[i], g, W1, p);
          if (GATHER) {
            float4 cv[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) cv[v] = Tq[p.base + off[v]];
            float gw[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              float s = cv[v].x * fv[0][i];
              s = __builtin_fmaf(cv[v].y, fv[1][i], s);
              s = __builtin_fmaf(cv[v].z, fv[2][i], s);
              s = __builtin_fmaf(cv[v].w, fv[3][i], s);
              gw[v] = s;
            }
            gs[u][i][0] = __builtin_fmaf(gw[3] - gw[2], p.w1y, __builtin_fmaf(gw[1] - gw[0], p.w0y, gs[u][i][0]));
            gs[u][i][1] = __builtin_fmaf(gw[3] - gw[1], p.w1x, __builtin_fmaf(gw[2] - gw[0], p.w0x, gs[u][i][1]));
            // pin the two sums here: without it the compiler sinks the whole gather -> g_keys chain of a point past
            // the following points' work and spills the 16 gathered values meanwhile
            asm volatile("" : "+v"(gs[u][i][0]), "+v"(gs[u][i][1]));
          }
          // (threads past the end of the cloud hold zeros: they add 0 to the cells of the cloud's first quad — no
          //  branch here, so that the point's work stays one basic block in source order.  Issuing the next point's
          //  gathers ahead of these atomics was measured: no gain.)
          const ct_f2 cw01 = {p.cw[0], p.cw[1]}, cw23 = {p.cw[2], p.cw[3]};
#if CT_FUSED_PAIR
          // Two channels per LDS atomic: the pair's fixed-point values go into one 64-bit word {lo = channel 2k, hi =
          // channel 2k+1} as hi * 2^32 + lo in two's complement (high word = hi + (lo >> 31)), so the word holds
          // sum(hi) * 2^32 + sum(lo) exactly and the halves are recovered at write-out.  ds_add_u64 5.5 ns against
          // 2 x 3.3 ns for two ds_add_u32 (tools/microbench/lds_atomics.hip).
#pragma unroll
          for (int pj = 0; pj < 2; ++pj) {
            unsigned long long* Tc = (unsigned long long*)accq + (size_t)pj * G + p.base;
            const float fa = fv[2 * pj][i] * iq[2 * pj], fb = fv[2 * pj + 1][i] * iq[2 * pj + 1];
            const ct_f2 a01 = cw01 * fa, a23 = cw23 * fa, b01 = cw01 * fb, b23 = cw23 * fb;
            const int lo[4] = {cvt_rpi(a01.x), cvt_rpi(a01.y), cvt_rpi(a23.x), cvt_rpi(a23.y)};
            const int hi[4] = {cvt_rpi(b01.x), cvt_rpi(b01.y), cvt_rpi(b23.x), cvt_rpi(b23.y)};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const unsigned long long val = ((unsigned long long)(unsigned)(hi[v] + (lo[v] >> 31)) << 32) | (unsigned)lo[v];
              atomicAdd(Tc + off[v], val);
            }
          }
#else
#pragma unroll
          for (int cj = 0; cj < 4; ++cj) {
            int* Tc = accq + cj * G + p.base;
            const float fq = fv[cj][i] * iq[cj];          // power-of-two scale: exact
            const ct_f2 p01 = cw01 * fq, p23 = cw23 * fq; // v_pk_mul_f32
            atomicAdd(Tc + off[0], cvt_rpi(p01.x));
            atomicAdd(Tc + off[1], cvt_rpi(p01.y));
            atomicAdd(Tc + off[2], cvt_rpi(p23.x));
            atomicAdd(Tc + off[3], cvt_rpi(p23.y));
          }
#endif
          __builtin_amdgcn_sched_barrier(0);     // one point at a time: keeps the live set inside the register budget
        }
      }
      if (any_float) {         // block-uniform, rare
#pragma unroll 1
        for (int cj = 0; cj < 4; ++cj) {
          float q, iqd;
          bool fixed;
          fx_quantum(__uint_as_float(s_max[ch0 + cj]) * Kf, q, iqd, fixed);
#if CT_FUSED_PAIR
          if (iq[cj] == 0.0f)      // this channel's pair is on the float path: its half of the pair's words holds a float
            scatter_float_channel<HAS_PAD>(a, g, bh, b, ch0 + cj, (float*)(accq + (size_t)(cj >> 1) * 2 * G) + (cj & 1), 2, so);
#else
          if (!fixed) scatter_float_channel<HAS_PAD>(a, g, bh, b, ch0 + cj, (float*)(accq + cj * G), 1, so);
#endif
        }
      }
    }
    __syncthreads();
    // write the chunk out (and clear the accumulators for the next chunk in the same sweep)
    const bool more = chunk + a.ncg < a.nchunks;
#if CT_FUSED_PAIR
    // a thread takes 4 cells of a channel pair: two 16-byte reads of {lo, hi} words -> one float4 per channel
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
        // word = hi * 2^32 + lo (two's complement): lo is the low half as it stands, hi = high half + (lo < 0)
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
      const int ch = (t << 2) / G;                 // G % 4 == 0: a float4 never straddles channels
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
    // (the next chunk's staging overwrites T4 only: every gather of this chunk is behind the barrier above)
  }
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    if (GATHER && active[u]) {
      const float4 tx = *(const float4*)(keyx + n0[u]), ty = *(const float4*)(keyy + n0[u]);
      float4 ox, oy;
      ox.x = gs[u][0][0] * ct_key_mask(tx.x); ox.y = gs[u][1][0] * ct_key_mask(tx.y);
      ox.z = gs[u][2][0] * ct_key_mask(tx.z); ox.w = gs[u][3][0] * ct_key_mask(tx.w);
      oy.x = gs[u][0][1] * ct_key_mask(ty.x); oy.y = gs[u][1][1] * ct_key_mask(ty.y);
      oy.z = gs[u][2][1] * ct_key_mask(ty.z); oy.w = gs[u][3][1] * ct_key_mask(ty.w);
      float* gp = a.g_pos + (size_t)cgi * a.gpos_stride;
      st_part4(gp + (bh * 2 + 0) * Nr + so + n0[u], ox, fold_keys);
      st_part4(gp + (bh * 2 + 1) * Nr + so + n0[u], oy, fold_keys);
    }
  }
  if (fold_keys || fold_grid) {       // kernel-uniform
    unsigned* s_flag = s_k + 1;
    const unsigned f = arrive_last(fold_keys ? a.tickets + (bh * nsg + seg) : nullptr, (unsigned)a.ncg,
                                   fold_grid ? a.tickets + kTicketHalf + (bh * a.ncg + cgi) : nullptr, (unsigned)nsg, s_flag);
    if (f & 1u) {      // this segment's g_keys: the chunk groups' partials, ascending
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fold_rows(a.g_pos + (bh * 2 + j) * Nr + so, a.gpos_stride, a.ncg, a.fold_gpos + (bh * 2 + j) * Nr + so, N >> 2, nullptr);
    }
    if (f & 2u) {      // this chunk group's g_grid tiles: the segments' partials, ascending
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
// KB: Splat(max0) backward, hot form.  One 512-thread workgroup per (b, h) plane; chunks of CC channels:
//   LDS: [CC/2][G] x float4 {z(c), z(c+1), g_z(c), g_z(c+1)}.
//   CLAIMS = false: a contribution whose product is bit-equal to a non-zero z receives g_z (no claim); the
//     kernel counts such matches and the non-zero cells of z.  On a tie-free plane the counts agree and every
//     cell had exactly one match.  Otherwise (duplicated points ...) the plane is redone with CLAIMS = true:
//     the first tied contribution to compare-and-swap the cell's z word to 0 wins (single winner, as
//     torch_scatter's backward; which of the tied contributions is unspecified).
//   QPT > 0: every thread owns at most QPT quads (N <= 4*QPT*blockDim): g_keys stays in registers across the
//     chunks and is added to an incoming cotangent in the same store (a.gpos_add) — a block's keys feed Splat and Slice.
//   QPT == 0: any N; the partial g_keys of the chunks go through memory.
//   grid = (1, H, B)
// ---------------------------------------------------------------------------
// bit pattern of an empty / already claimed cell in the staged z tile: a NaN with an all-ones payload.  z itself is
// never NaN (a NaN product does not beat the zero floor in the forward), and a product equals this pattern only if
// a feature is that very NaN.
constexpr unsigned kNoMatch = 0x7FFFFFFFu;
#ifndef CT_TIE_FIX
// 1: the optimistic pass also sums the BIT PATTERNS of the cotangents it awards (one three-operand add per corner and channel
// pair), the staging pass those of the non-zero cells; with ONE surplus match in a chunk the difference is the tied cell's
// cotangent, the cell is found by value, and that cell alone is repaired (splat_bwd_fix_one_tie) instead of its group being
// redone.  0: every tied group is redone.  profiles/r5_splat_bwd_loop.txt has both, and the cell-sum form that lost.
#define CT_TIE_FIX 1
#endif

// rows of the point-sized tensors as one workgroup sees them: Nr floats long, the workgroup's points start at `so`
// (point segments: RasterArgs::nseg; Nr = N, so = 0 without them)
struct PtRows {
  int Nr;
  size_t so;
  bool wt;      // results another workgroup may have to overwrite (a segmented plane with ties) go out write-through
};

// groups of four channels a plane's tie test distinguishes (s_cnt[4 + i]: non-zero cells of group i, s_cnt[4 + kTieGroups + i]:
// its matches): a plane with C > 4 * kTieGroups channels only has the plane-wide test
constexpr int kTieGroups = 64;
// forms without per-group counters (3D, point segments, N beyond the register forms): the whole pass's two bit-pattern sums
// (plane_sum_bits) sit in the first words of the group area, the repair's words behind them
constexpr int kTieSumPos = 4, kTieSumNeg = 5, kTieMemFix = 8, kTieMemList = 16;      // indices into s_cnt
constexpr int kTieFixWords = 8;      // behind the 3 * kTieGroups words: splat_bwd_fix_one_tie's winners and counts

// s_cnt[4 + 2 * kTieGroups + i]: the sum of `cell` over the group's matches MINUS the sum over its non-zero (cell, channel) pairs,
// modulo 2^13.  Every non-zero pair has at least one match, so with exactly one surplus match the word IS the tied cell, and
// the plane's workgroup repairs that cell alone (splat_bwd_fix_one_tie).  Per thread the sums travel as four 16-bit fields
// (the groups of a chunk), each term reduced to 13 bits before it is added: no carry crosses a field.
// s_cnt[4 + 2 * kTieGroups + chunk]: the sum (mod 2^32) of the bit patterns of g_z over the chunk's matches MINUS the sum over its
// non-zero (cell, channel) pairs.  Every non-zero pair has at least one match, so with exactly one surplus match in the chunk the
// word IS the bit pattern of g_z at the tied pair (0: the surplus award was +0, nothing to repair).
__device__ __forceinline__ void plane_sum_bits(int* word, unsigned v, int sign) {
  const int t = wave_sum_i32((int)v);
  if ((threadIdx.x & 63) == 0 && t) atomicAdd(word, sign * t);
}

// DELTA (with CLAIMS): the redo of ONE four-channel group after an optimistic pass — its g_feat rows are rewritten with the
// single-winner award, and gs receives only the difference to what the optimistic pass had added for these channels (the
// awards of matches that lose their claim, negated).  nmp: the optimistic pass's per-group match counter of this chunk.
template <bool HAS_PAD, bool CLAIMS, int WT, bool DELTA = false>
__device__ __forceinline__ void splat_bwd_quad(const RasterArgs& a, const GridW<2>& g, float4* ZG, size_t bh, int b,
                                               int c0, int cc, int n0, const PtRows& R, const float (&kx)[4], const float (&ky)[4],
                                               float (&gs)[4][2], int& nm, unsigned& nmp, unsigned& xs, bool per_group) {
  // nmp: the per-group match counts of this chunk (per_group), xs: the running sum of the bit patterns of the awarded cotangents
  // (CT_TIE_FIX; plane_sum_bits).  REFERENCES to the caller's registers: through a pointer that may be null (rounds 4 and 5 until
  // here) the compiler kept both in scratch, and the read-modify-write behind every group waited for the group's stores.
  const int G = WT ? WT * WT : g.G, W1 = WT ? WT : g.W[1];
  const int off[4] = {0, W1, 1, W1 + 1};
  float pv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pv[i] = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * R.Nr + R.so + n0 + i) : 1.0f;
  // 4 channels (two {z,z,g,g} pairs) per step: the corner weights of a point are computed once per 4 channels.
  // (Requesting the next step's rows before this step is processed was measured: the 16 extra registers spill, 73 -> 102 us.)
  for (int cg0 = 0; cg0 < cc; cg0 += 4) {
    const int nm_before = nm;
    float fv[4][4];          // [channel][point]
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) {
      const float* row = a.src + (bh * a.C + c0 + cg0 + cj) * (size_t)R.Nr + R.so;      // wave-uniform
      const float4 t = ld_stream4(row + n0);
      fv[cj][0] = t.x; fv[cj][1] = t.y; fv[cj][2] = t.z; fv[cj][3] = t.w;
    }
    float4* Zc = ZG + (size_t)(cg0 >> 1) * G;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Pt2 p;
      pt2_from_keys(kx[i], ky[i], g, W1, p);
      float gw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        float4* Zp = Zc + (size_t)pr * G + p.base;
        const float xa = HAS_PAD ? fv[2 * pr][i] * pv[i] : fv[2 * pr][i];
        const float xb = HAS_PAD ? fv[2 * pr + 1][i] * pv[i] : fv[2 * pr + 1][i];
        ct_f4 zg[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) zg[v] = *(const ct_f4*)(Zp + off[v]);
        // Whole 16-byte reads, pinned per component: left alone the compiler splits them into a narrow read plus conditional ones.
        // (Pinned as register tuples — together or one at a time — the loop has 15 % fewer instructions and is SLOWER, in 3D by
        // 15 %: profiles/r5_splat_bwd_loop.txt.)
#pragma unroll
        for (int v = 0; v < 4; ++v) asm volatile("" : "+v"(zg[v].x), "+v"(zg[v].y), "+v"(zg[v].z), "+v"(zg[v].w));
        float gfa = 0.0f, gfb = 0.0f;
        if (!CLAIMS) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            // the products are formed exactly as the forward formed them (one rounding: no fma).  Empty cells were
            // staged as kNoMatch, so bit-equality alone is the winner test.
            const unsigned ba = __float_as_uint(xa * p.cw[v]), bb = __float_as_uint(xb * p.cw[v]);
            const bool ma = ba == __float_as_uint(zg[v].x), mb = bb == __float_as_uint(zg[v].y);
            nm += (int)ma;       // (add-with-carry of the compare mask: one instruction each ...
            nm += (int)mb;
            asm volatile("" : "+v"(nm));      //  ... issued here: otherwise all 32 masks of a point are kept for a final sum)
            const float ga = ma ? zg[v].z : 0.0f, gb = mb ? zg[v].w : 0.0f;
            if (CT_TIE_FIX) {
              xs += __float_as_uint(ga) + __float_as_uint(gb);
              asm volatile("" : "+v"(xs));      // (added here and now: see nm)
            }
            gfa = __builtin_fmaf(ga, p.cw[v], gfa);
            gfb = __builtin_fmaf(gb, p.cw[v], gfb);
            gw[v] = __builtin_fmaf(gb, xb, __builtin_fmaf(ga, xa, gw[v]));
          }
        } else {
          // claims: a cell's z word is won by the first matching contribution that flips its sign bit (z > 0 in every non-empty
          // cell: the zero floor of the forward), so a later contribution still sees WHAT the maximum was and knows that it
          // matched and lost — which the redo of a single group needs (DELTA: the award the optimistic pass gave it is taken back)
          constexpr unsigned kSign = 0x80000000u;
          unsigned ba[4], bb[4];
          bool ma[4], mb[4], la[4], lb[4];
          bool any = false;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            ba[v] = __float_as_uint(xa * p.cw[v]);
            bb[v] = __float_as_uint(xb * p.cw[v]);
            const unsigned za = __float_as_uint(zg[v].x), zb = __float_as_uint(zg[v].y);
            ma[v] = ba[v] == za && !(za & kSign);
            mb[v] = bb[v] == zb && !(zb & kSign);
            la[v] = (za & kSign) != 0u && ba[v] == (za ^ kSign);          // matched, already claimed by another contribution
            lb[v] = (zb & kSign) != 0u && bb[v] == (zb ^ kSign);
            any = any | ma[v] | mb[v] | (DELTA & (la[v] | lb[v]));
          }
          if (any) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              unsigned* zw = (unsigned*)(Zp + off[v]);
              const unsigned oa = atomicCAS(zw, ma[v] ? ba[v] : kNoMatch, ma[v] ? (ba[v] | kSign) : kNoMatch);
              const unsigned ob = atomicCAS(zw + 1, mb[v] ? bb[v] : kNoMatch, mb[v] ? (bb[v] | kSign) : kNoMatch);
              const float ga = (ma[v] & (oa == ba[v])) ? zg[v].z : 0.0f;
              const float gb = (mb[v] & (ob == bb[v])) ? zg[v].w : 0.0f;
              gfa = __builtin_fmaf(ga, p.cw[v], gfa);
              gfb = __builtin_fmaf(gb, p.cw[v], gfb);
              if (DELTA) {      // what the optimistic pass added for a match that does not hold the claim, taken back
                const float da = (la[v] | (ma[v] & (oa != ba[v]))) ? -zg[v].z : 0.0f;
                const float db = (lb[v] | (mb[v] & (ob != bb[v]))) ? -zg[v].w : 0.0f;
                gw[v] = __builtin_fmaf(db, xb, __builtin_fmaf(da, xa, gw[v]));
              } else {
                gw[v] = __builtin_fmaf(gb, xb, __builtin_fmaf(ga, xa, gw[v]));
              }
            }
          }
        }
        fv[2 * pr][i] = HAS_PAD ? gfa * pv[i] : gfa;
        fv[2 * pr + 1][i] = HAS_PAD ? gfb * pv[i] : gfb;
      }
      gs[i][0] = __builtin_fmaf(gw[3] - gw[2], p.w1y, __builtin_fmaf(gw[1] - gw[0], p.w0y, gs[i][0]));
      gs[i][1] = __builtin_fmaf(gw[3] - gw[1], p.w1x, __builtin_fmaf(gw[2] - gw[0], p.w0x, gs[i][1]));
      // pin the point's results here (see slice_bwd_fused_kernel): one point at a time keeps the live set small
      asm volatile("" : "+v"(gs[i][0]), "+v"(gs[i][1]), "+v"(fv[0][i]), "+v"(fv[1][i]), "+v"(fv[2][i]), "+v"(fv[3][i]));
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int cj = 0; cj < 4; ++cj)
      st_part4(a.dst + (bh * a.C + c0 + cg0 + cj) * (size_t)R.Nr + R.so + n0, make_float4(fv[cj][0], fv[cj][1], fv[cj][2], fv[cj][3]), R.wt);
    // this group's matches into the thread's packed per-chunk counter (8 bits per four-channel group of the chunk: <= 4 groups,
    // <= 2 quads x 64 products per thread and group); LDS adds per (quad, group) — even one per wave — cost 3.5 us of 68 on the headline
    if (!CLAIMS && per_group) nmp += (unsigned)(nm - nm_before) << (8 * ((cg0 >> 2) & 3));
  }
}

// The repair of ONE exact tie (one surplus match in the four-channel group at `cabs`, all of it in cell `t`)
// behind an optimistic pass, by the plane's workgroup with its points in registers (QPT > 0).  Only the points with a corner
// in `t` do anything: they test their four products against the cell's maxima; per channel the lowest point index keeps the
// award, every other match gives it back — its g_feat element recomputed without that corner, its key cotangent corrected by
// the negated award exactly as splat_bwd_quad<.., DELTA> would.  Two dependent trips to memory for a handful of lanes, where the
// redo of the group walks every point of the plane again (headline: +10..14 us -> see DESIGN 4.1).  Returns false (block-uniform,
// nothing written) if the cell does not hold what the counters promised; the caller then redoes the group.
template <bool HAS_PAD, int WT, int QPT>
__device__ __forceinline__ bool splat_bwd_fix_one_tie(const RasterArgs& a, const GridW<2>& g, size_t bh, int b, int cabs, int t,
                                                      const PtRows& R, float (&gs)[QPT ? QPT : 1][4][2], int* s_fix,
                                                      const float4 (&keysx)[QPT ? QPT : 1], const float4 (&keysy)[QPT ? QPT : 1]) {
  const int G = WT ? WT * WT : g.G, W1 = WT ? WT : g.W[1];
  const int off[4] = {0, W1, 1, W1 + 1};
  const int tid = threadIdx.x, nq = a.N >> 2;
  __syncthreads();
  if (tid < kTieFixWords) s_fix[tid] = tid < 4 ? 0x7fffffff : 0;
  __syncthreads();
  const float* zrow = a.tile_in + (bh * a.C + cabs) * (size_t)G;
  const float* grow = a.tile_in2 + (bh * a.C + cabs) * (size_t)G;
  const float* keyx = a.pos.keys + (bh * 2 + 0) * R.Nr + R.so;
  const float* keyy = a.pos.keys + (bh * 2 + 1) * R.Nr + R.so;
  unsigned zt[4];
  int expect = 1;
#pragma unroll
  for (int cj = 0; cj < 4; ++cj) {
    zt[cj] = __float_as_uint(zrow[(size_t)cj * G + t]);
    expect += zt[cj] != 0u;
  }
  // which of the thread's points have a corner in `t` (no memory behind this: base cells from the keys in registers); the few
  // that do are then walked one at a time, so that the code behind exists once and not per point slot
  unsigned cand = 0u;
#pragma unroll
  for (int u = 0; u < (QPT ? QPT : 1); ++u) {
    const float kx[4] = {keysx[u].x, keysx[u].y, keysx[u].z, keysx[u].w}, ky[4] = {keysy[u].x, keysy[u].y, keysy[u].z, keysy[u].w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float w0, w1;
      int fx, fy;
      ct_axis(kx[i], g.hw[0], g.W[0], w0, w1, fx);
      ct_axis(ky[i], g.hw[1], g.W[1], w0, w1, fy);
      const int d = t - (fx * W1 + fy);
      if ((d == 0 || d == W1 || d == 1 || d == W1 + 1) && tid + u * (int)blockDim.x < nq) cand |= 1u << (4 * u + i);
    }
  }
  unsigned hit = 0u;          // per point slot (u, i): 4 bits, the channels of the group whose maximum this point's product equals
#pragma unroll 1
  for (unsigned m = cand; m != 0u; m &= m - 1u) {
    const int slot = __builtin_ctz(m);
    const int n = ((tid + (slot >> 2) * (int)blockDim.x) << 2) + (slot & 3);
    Pt2 p;
    pt2_from_keys(keyx[n], keyy[n], g, W1, p);
    const int d = t - p.base;
    const float w = d == 0 ? p.cw[0] : d == W1 ? p.cw[1] : d == 1 ? p.cw[2] : p.cw[3];
    const float pv = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * R.Nr + R.so + n) : 1.0f;
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) {
      const float f = a.src[(bh * a.C + cabs + cj) * (size_t)R.Nr + R.so + n];
      const float x = HAS_PAD ? f * pv : f;
      if (zt[cj] != 0u && __float_as_uint(x * w) == zt[cj]) {
        hit |= 1u << (4 * slot + cj);
        atomicMin(&s_fix[cj], n);
        atomicAdd(&s_fix[4], 1);
      }
    }
  }
  __syncthreads();
  if (s_fix[4] != expect) return false;          // block-uniform
#pragma unroll 1
  for (unsigned m = cand; m != 0u; m &= m - 1u) {
    const int slot = __builtin_ctz(m);
    const unsigned mine = (hit >> (4 * slot)) & 0xfu;
    const int n = ((tid + (slot >> 2) * (int)blockDim.x) << 2) + (slot & 3);
    unsigned lost = 0u;
#pragma unroll
    for (int cj = 0; cj < 4; ++cj)
      if (((mine >> cj) & 1u) && s_fix[cj] != n) lost |= 1u << cj;
    if (lost == 0u) continue;
    Pt2 p;
    pt2_from_keys(keyx[n], keyy[n], g, W1, p);
    const int vt = t - p.base == 0 ? 0 : t - p.base == W1 ? 1 : t - p.base == 1 ? 2 : 3;
    const float pv = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * R.Nr + R.so + n) : 1.0f;
    float x[4];
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) {
      const float f = a.src[(bh * a.C + cabs + cj) * (size_t)R.Nr + R.so + n];
      x[cj] = HAS_PAD ? f * pv : f;
    }
    // the key cotangent: the lost awards, negated, through the corner-weight gradient (the arithmetic of DELTA)
    float gwt = 0.0f;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const float da = ((lost >> (2 * pr)) & 1u) ? -grow[(size_t)(2 * pr) * G + t] : 0.0f;
      const float db = ((lost >> (2 * pr + 1)) & 1u) ? -grow[(size_t)(2 * pr + 1) * G + t] : 0.0f;
      gwt = __builtin_fmaf(db, x[2 * pr + 1], __builtin_fmaf(da, x[2 * pr], gwt));
    }
    float gw[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) gw[v] = v == vt ? gwt : 0.0f;
    const float d0 = __builtin_fmaf(gw[3] - gw[2], p.w1y, (gw[1] - gw[0]) * p.w0y);
    const float d1 = __builtin_fmaf(gw[3] - gw[1], p.w1x, (gw[2] - gw[0]) * p.w0x);
    // (the slot is a run-time value here: the sums are addressed by selects, not by index)
#pragma unroll
    for (int u = 0; u < (QPT ? QPT : 1); ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (slot == 4 * u + i) {
          gs[u][i][0] += d0;
          gs[u][i][1] += d1;
        }
    // g_feat of the lost channels: the point's other corners keep what they matched
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) {
      if (!((lost >> cj) & 1u)) continue;
      float gf = 0.0f;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const unsigned zc = __float_as_uint(zrow[(size_t)cj * G + p.base + off[v]]);
        const float gc = grow[(size_t)cj * G + p.base + off[v]];
        const bool won = v != vt && zc != 0u && __float_as_uint(x[cj] * p.cw[v]) == zc;
        gf = __builtin_fmaf(won ? gc : 0.0f, p.cw[v], gf);
      }
      a.dst[(bh * a.C + cabs + cj) * (size_t)R.Nr + R.so + n] = HAS_PAD ? gf * pv : gf;
    }
  }
  return true;
}

// The tied cell found by the VALUE of its cotangent (plane_sum_bits): the cells of the group's four rows whose g_z has these bits
// (and whose z is not empty) are tried one after the other — a cell that is not the tied one fails the repair's own count and is
// left untouched.  False (block-uniform) if none fits or there are more than kTieTry of them (a constant g_z: redo the group).
constexpr int kTieTry = 7;
#ifdef CT_TIE_DEBUG
__device__ unsigned g_tie_dbg[16];
#define CT_TIE_COUNT(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_tie_dbg[i], (unsigned)(v)); } while (0)
#else
#define CT_TIE_COUNT(i, v) do { } while (0)
#endif
template <bool HAS_PAD, int WT, int QPT>
__device__ __forceinline__ bool splat_bwd_fix_by_value(const RasterArgs& a, const GridW<2>& g, size_t bh, int b, int cabs, unsigned gbits,
                                                       const PtRows& R, float (&gs)[QPT ? QPT : 1][4][2], int* s_fix, int* s_list,
                                                       const float4* resident) {
  const int G = WT ? WT * WT : g.G;
  const int tid = threadIdx.x, nq = a.N >> 2;
  const float* zrow = a.tile_in + (bh * a.C + cabs) * (size_t)G;
  const float* grow = a.tile_in2 + (bh * a.C + cabs) * (size_t)G;
  // the thread's keys travel with the search's loads (one trip to memory less in front of the repair)
  float4 keysx[QPT ? QPT : 1], keysy[QPT ? QPT : 1];
#pragma unroll
  for (int u = 0; u < (QPT ? QPT : 1); ++u) {
    const int q = min(tid + u * (int)blockDim.x, nq - 1);
    keysx[u] = *(const float4*)(a.pos.keys + (bh * 2 + 0) * R.Nr + R.so + (q << 2));
    keysy[u] = *(const float4*)(a.pos.keys + (bh * 2 + 1) * R.Nr + R.so + (q << 2));
  }
  __syncthreads();
  if (tid == 0) s_list[0] = 0;
  __syncthreads();
  if (resident != nullptr) {      // the group's pairs are still staged: {z(c), z(c+1), g_z(c), g_z(c+1)} per cell, empty cells as kNoMatch
    for (int i = tid; i < 2 * G; i += blockDim.x) {
      const float4 e = resident[i];
      const bool h0 = __float_as_uint(e.z) == gbits && __float_as_uint(e.x) != kNoMatch;
      const bool h1 = __float_as_uint(e.w) == gbits && __float_as_uint(e.y) != kNoMatch;
      if (h0 | h1) {
        const int k = atomicAdd(&s_list[0], 1);
        if (k < kTieTry) s_list[1 + k] = i % G;
      }
    }
  } else {
    // (eight loads per thread in flight, the tests behind them: one trip to memory, not one per element)
    const int n = 4 * G, step = (int)blockDim.x;
    for (int i0 = tid; i0 < n; i0 += 8 * step) {
      unsigned gv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) gv[u] = __float_as_uint(grow[min(i0 + u * step, n - 1)]);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * step;
        if (i < n && gv[u] == gbits && __float_as_uint(zrow[i]) != 0u) {
          const int k = atomicAdd(&s_list[0], 1);
          if (k < kTieTry) s_list[1 + k] = i % G;
        }
      }
    }
  }
  __syncthreads();
  const int n = s_list[0];
  CT_TIE_COUNT(0, 1); CT_TIE_COUNT(1, n); CT_TIE_COUNT(2, resident != nullptr);
  if (n > kTieTry) return false;
  int cells[kTieTry];
#pragma unroll
  for (int k = 0; k < kTieTry; ++k) cells[k] = k < n ? s_list[1 + k] : -1;
#pragma unroll 1
  for (int k = 0; k < n; ++k) {
    int t = cells[0];
#pragma unroll
    for (int j = 1; j < kTieTry; ++j) t = k == j ? cells[j] : t;
    if (splat_bwd_fix_one_tie<HAS_PAD, WT, QPT>(a, g, bh, b, cabs, t, R, gs, s_fix, keysx, keysy)) { CT_TIE_COUNT(3, 1); return true; }
    CT_TIE_COUNT(4, 1);
  }
  return false;
}

// one pass over the workgroup's points (N of them, rows R) and its chunks; cgi: chunk group (see slice_bwd_fused_kernel)
template <bool HAS_PAD, bool CLAIMS, int WT, int QPT>
__device__ __forceinline__ void splat_bwd_plane_pass(const RasterArgs& a, const GridW<2>& g, float4* ZG, int* s_cnt,
                                                    size_t bh, int b, int cgi, int N, const PtRows& R,
                                                    float (&gs_reg)[QPT ? QPT : 1][4][2], bool& tie, int* grp = nullptr) {
  const int G = WT ? WT * WT : g.G, CC = a.CC;
  const int tid = threadIdx.x;
  const int nq = N >> 2;
  int nz = 0, nm = 0;
  float* gpos = a.g_pos + (size_t)cgi * a.gpos_stride;
  const float* keyx = a.pos.keys + (bh * 2 + 0) * R.Nr + R.so;
  const float* keyy = a.pos.keys + (bh * 2 + 1) * R.Nr + R.so;
  for (int chunk = cgi; chunk < a.nchunks; chunk += a.ncg) {
    const int c0 = chunk * CC;
    const int cc = min(CC, a.C - c0);            // multiple of 4
    const float* zin = a.tile_in + (bh * a.C + c0) * (size_t)G;
    const float* gin = a.tile_in2 + (bh * a.C + c0) * (size_t)G;
    __syncthreads();                              // readers of the previous chunk (or pass) are done
    unsigned long long nzp = 0ull;
    unsigned xzp = 0u;
    for (int t = tid; t < (cc >> 1) * G; t += blockDim.x) {
      const int cp = t / G, cell = t - cp * G;
      const size_t o = (size_t)(cp * 2) * G + cell;
      // an empty cell (z = 0: nothing beat the zero floor) is staged as kNoMatch, a bit pattern no product of finite
      // inputs has, so that the winner test in the loop is one compare
      const unsigned z0 = __float_as_uint(ld_stream(zin + o)), z1 = __float_as_uint(ld_stream(zin + o + G));
      const float g0 = ld_stream(gin + o), g1 = ld_stream(gin + o + G);
      ZG[t] = make_float4(__uint_as_float(z0 ? z0 : kNoMatch), __uint_as_float(z1 ? z1 : kNoMatch), g0, g1);
      if (!CLAIMS) {
        const int nzt = (z0 != 0u) + (z1 != 0u);
        nz += nzt;
        nzp += (unsigned long long)(unsigned)nzt << (16 * ((cp >> 1) & 3));      // per four-channel group of the chunk (packed: see nmp)
        if (CT_TIE_FIX) xzp += (z0 ? __float_as_uint(g0) : 0u) + (z1 ? __float_as_uint(g1) : 0u);
      }
    }
    if (!CLAIMS && grp != nullptr) {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int v = wave_sum_i32((int)((nzp >> (16 * f)) & 0xffffu));
        if ((tid & 63) == 0 && v) atomicAdd(grp + (c0 >> 2) + f, v);
      }
    }
    // (forms without the per-group words — point segments, N beyond the register forms — keep the pass's two sums apart)
    if (CT_TIE_FIX && !CLAIMS) plane_sum_bits(grp != nullptr ? grp + 2 * kTieGroups + chunk : s_cnt + kTieSumNeg, xzp, grp != nullptr ? -1 : 1);
    __syncthreads();
    unsigned nmp = 0u;
    unsigned xmp = 0u;
    const bool per_group = !CLAIMS && grp != nullptr;
    if constexpr (QPT > 0) {
#pragma unroll
      for (int u = 0; u < QPT; ++u) {
        const int q = tid + u * (int)blockDim.x;
        if (q < nq) {
          const int n0 = q << 2;
          const float4 tx = *(const float4*)(keyx + n0);
          const float4 ty = *(const float4*)(keyy + n0);
          const float kx[4] = {tx.x, tx.y, tx.z, tx.w}, ky[4] = {ty.x, ty.y, ty.z, ty.w};
          splat_bwd_quad<HAS_PAD, CLAIMS, WT>(a, g, ZG, bh, b, c0, cc, n0, R, kx, ky, gs_reg[u], nm, nmp, xmp, per_group);
        }
      }
    } else {
      for (int q = tid; q < nq; q += blockDim.x) {
        const int n0 = q << 2;
        const float4 tx = *(const float4*)(keyx + n0);
        const float4 ty = *(const float4*)(keyy + n0);
        const float kx[4] = {tx.x, tx.y, tx.z, tx.w}, ky[4] = {ty.x, ty.y, ty.z, ty.w};
        float gs[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) gs[i][0] = gs[i][1] = 0.0f;
        splat_bwd_quad<HAS_PAD, CLAIMS, WT>(a, g, ZG, bh, b, c0, cc, n0, R, kx, ky, gs, nm, nmp, xmp, per_group);
        // the partial g_keys sums of the chunks go through memory (plain read-modify-write: the thread owns these
        // addresses); the first chunk starts from the incoming cotangent where there is one (a.gpos_add)
        float4 ox = make_float4(gs[0][0] * ct_key_mask(kx[0]), gs[1][0] * ct_key_mask(kx[1]),
                                gs[2][0] * ct_key_mask(kx[2]), gs[3][0] * ct_key_mask(kx[3]));
        float4 oy = make_float4(gs[0][1] * ct_key_mask(ky[0]), gs[1][1] * ct_key_mask(ky[1]),
                                gs[2][1] * ct_key_mask(ky[2]), gs[3][1] * ct_key_mask(ky[3]));
        float* px = gpos + (bh * 2 + 0) * R.Nr + R.so + n0;
        float* py = gpos + (bh * 2 + 1) * R.Nr + R.so + n0;
        if (chunk > cgi || a.gpos_add != nullptr) {
          const float* ax = chunk > cgi ? px : a.gpos_add + (bh * 2 + 0) * R.Nr + R.so + n0;
          const float* ay = chunk > cgi ? py : a.gpos_add + (bh * 2 + 1) * R.Nr + R.so + n0;
          const float4 qx = *(const float4*)ax, qy = *(const float4*)ay;
          ox.x += qx.x; ox.y += qx.y; ox.z += qx.z; ox.w += qx.w;
          oy.x += qy.x; oy.y += qy.y; oy.z += qy.z; oy.w += qy.w;
        }
        // handed to another workgroup — the finished partial of a chunk group that is folded (arrive_last), the results of a
        // segment that a tie may force the plane's last workgroup to overwrite: write-through
        if (chunk + a.ncg >= a.nchunks && (R.wt || (a.tickets != nullptr && a.ncg > 1))) {
          st_sc1_4(px, ox);
          st_sc1_4(py, oy);
        } else {
          *(float4*)px = ox;
          *(float4*)py = oy;
        }
      }
    }
    if (per_group) {                              // (all lanes are back here: threads without a quad add zeros)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int v = wave_sum_i32((int)((nmp >> (8 * f)) & 0xffu));
        if ((tid & 63) == 0 && v) atomicAdd(grp + kTieGroups + (c0 >> 2) + f, v);
      }
    }
    if (CT_TIE_FIX && !CLAIMS) plane_sum_bits(grp != nullptr ? grp + 2 * kTieGroups + chunk : s_cnt + kTieSumPos, xmp, 1);
  }
  if (!CLAIMS) {
    nz = wave_sum_i32(nz);
    nm = wave_sum_i32(nm);
    if ((tid & 63) == 0) {
      atomicAdd(&s_cnt[0], nz);
      atomicAdd(&s_cnt[1], nm);
    }
    __syncthreads();
    tie = s_cnt[0] != s_cnt[1];
  }
}

// (ct_raster_hot3d.h: the repair of one tie through memory, 2D and 3D)
template <int DIM, bool HAS_PAD>
__device__ __forceinline__ bool splat_bwd_fix_mem_cold(size_t bh, int b, int cgi, unsigned gbits, int Nr, size_t gpos_off, bool wt,
                                                       int* s_cnt);

// Point segments (a.nseg > 1): the plane's points are dealt to nseg workgroups, each walking ALL chunks for its own points —
// no partial g_keys, nothing to fold, nseg times the workgroups (the decoders' 32 planes fill the chip; the zoo's 128 planes
// need no chunk groups).  What a segment cannot know alone is whether the PLANE has exact ties (matches are counted per
// segment, non-zero cells per plane): every segment adds its matches to the plane's word in the ticket buffer, takes a
// ticket, and the holder of the last ticket compares; on a tie (duplicated points: rare) it redoes the whole plane with
// single-winner claims, overwriting what the segments wrote — which is why their results went out write-through and
// why the incoming key cotangent (a.gpos_add) must not alias the output then.
template <bool HAS_PAD, int WT, int QPT, int NTB = kHotThreads>
__global__ void __launch_bounds__(NTB, 4) splat_max_bwd_hot_kernel(RasterArgs a, GridW<2> g_arg) {
  const GridW<2> g = grid2_of<WT>(g_arg);
  extern __shared__ __align__(16) float lds[];
  float4* ZG = (float4*)lds;
  int* s_cnt = (int*)(lds + (size_t)a.CC * g.G * 2);
  const int nsg = a.nseg > 0 ? a.nseg : 1;
  const WgCoord wg = wg_coord(a.ncg, nsg, a.H, a.B);
  const int h = wg.h, b = wg.b;
  const size_t bh = (size_t)b * a.H + h;
  const int N = a.N;
  PtRows R;
  R.Nr = a.Nrow > 0 ? a.Nrow : a.N;
  R.so = (size_t)wg.seg * a.N;
  R.wt = nsg > 1;
  const bool fold_keys = a.tickets != nullptr && a.ncg > 1;
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;       // ordered before use by the barriers of the pass
  float gs[QPT ? QPT : 1][4][2];
#pragma unroll
  for (int u = 0; u < (QPT ? QPT : 1); ++u)
#pragma unroll
    for (int i = 0; i < 4; ++i) gs[u][i][0] = gs[u][i][1] = 0.0f;
  bool tie = false;
  // Exact ties are rare but not exotic: two different points whose products equal a cell's maximum bit for bit turn up in
  // about one random B8 H64 C16 32^2 workload out of three (8.4 M (cell, channel) pairs, ~2e-7 each), duplicated points make
  // them the rule.  Redoing the whole plane doubled its workgroup's time and, as the launch's tail, cost 68 -> 117 us on the
  // headline.  With the plane's points in registers (QPT > 0) the test is kept per four-channel group — non-zero cells counted
  // while the tile is staged, matches per group by one LDS add per quad and group — and only a tied group is redone
  // (splat_bwd_quad<.., DELTA>): its tile pairs staged again, its g_feat rows rewritten with single-winner claims, g_keys
  // corrected by the difference.
  int* const grp = (QPT > 0 && QPT <= 2 && nsg == 1 && a.C <= 4 * kTieGroups && a.CC <= 16) ? s_cnt + 4 : nullptr;
  if (grp != nullptr) {
    for (int i = threadIdx.x; i < 3 * kTieGroups; i += blockDim.x) grp[i] = 0;
  } else if (threadIdx.x < 20) {
    s_cnt[kTieSumPos + threadIdx.x] = 0;       // the pass's bit sums, the words of splat_bwd_fix_mem
  }
#ifdef CT_EXP_CLAIMS_ONLY       // experiment: no optimistic pass, every plane with single-winner claims (the cost of a tie-proof single pass)
  tie = nsg == 1;
  if (nsg > 1)
#endif
  splat_bwd_plane_pass<HAS_PAD, false, WT, QPT>(a, g, ZG, s_cnt, bh, b, wg.cgi, N, R, gs, tie, grp);
#ifndef CT_EXP_CLAIMS_ONLY
  if constexpr (QPT > 0) {
    if (tie && grp != nullptr) {    // block-uniform
      const int G = WT ? WT * WT : g.G;
      const int tid = threadIdx.x, nq = N >> 2;
      bool staged = true;
      for (int gi = 0; gi < (a.C >> 2); ++gi) {
        if (grp[gi] == grp[kTieGroups + gi]) continue;           // block-uniform (LDS words, written before the pass's last barrier)
        const int cabs = gi << 2;                                 // the group's first channel
        if ((cabs / a.CC) % a.ncg != wg.cgi) continue;            // (another chunk group's channels)
        if (CT_TIE_FIX) {
          // one surplus match in the whole chunk: its cotangent's bit pattern is what the chunk's word holds
          const int chunk = cabs / a.CC;
          int extra = 0;
          for (int g2 = (chunk * a.CC) >> 2; g2 < (min(chunk * a.CC + a.CC, a.C) >> 2); ++g2) extra += grp[kTieGroups + g2] - grp[g2];
          const unsigned gbits = (unsigned)grp[2 * kTieGroups + chunk];
          CT_TIE_COUNT(7, 1); CT_TIE_COUNT(8, extra);
          if (extra == 1) {
            if (gbits == 0u) continue;      // the surplus award was +0: nothing went anywhere
            // (the pass's last chunk is still staged unless a redo below has overwritten it)
            const int last_chunk = wg.cgi + ((a.nchunks - 1 - wg.cgi) / a.ncg) * a.ncg;
            const float4* res = (staged && chunk == last_chunk) ? ZG + (size_t)((cabs - chunk * a.CC) >> 1) * G : nullptr;
            if (splat_bwd_fix_by_value<HAS_PAD, WT, QPT>(a, g, bh, b, cabs, gbits, R, gs, grp + 3 * kTieGroups,
                                                         grp + 3 * kTieGroups + kTieFixWords, res))
              continue;
          }
        }
        const float* zin = a.tile_in + (bh * a.C + cabs) * (size_t)G;
        const float* gin = a.tile_in2 + (bh * a.C + cabs) * (size_t)G;
        staged = false;
        CT_TIE_COUNT(5, 1);
        __syncthreads();
        for (int t = tid; t < 2 * G; t += blockDim.x) {
          const int cp = t / G, cell = t - cp * G;
          const size_t o = (size_t)(cp * 2) * G + cell;
          const unsigned z0 = __float_as_uint(ld_stream(zin + o)), z1 = __float_as_uint(ld_stream(zin + o + G));
          ZG[t] = make_float4(__uint_as_float(z0 ? z0 : kNoMatch), __uint_as_float(z1 ? z1 : kNoMatch), ld_stream(gin + o),
                              ld_stream(gin + o + G));
        }
        __syncthreads();
        const float* keyx = a.pos.keys + (bh * 2 + 0) * R.Nr + R.so;
        const float* keyy = a.pos.keys + (bh * 2 + 1) * R.Nr + R.so;
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
          const int q = tid + u * (int)blockDim.x;
          if (q < nq) {
            const int n0 = q << 2;
            const float4 tx = *(const float4*)(keyx + n0);
            const float4 ty = *(const float4*)(keyy + n0);
            const float kx[4] = {tx.x, tx.y, tx.z, tx.w}, ky[4] = {ty.x, ty.y, ty.z, ty.w};
            int nm_unused = 0;
            unsigned u0 = 0u, u1 = 0u;
            splat_bwd_quad<HAS_PAD, true, WT, true>(a, g, ZG, bh, b, cabs, 4, n0, R, kx, ky, gs[u], nm_unused, u0, u1, false);
          }
        }
      }
      tie = false;
    }
  }
#endif
  if constexpr (QPT == 0) {       // the through-memory form: the key cotangents are in their rows, one surplus match is repaired there
    if (CT_TIE_FIX && tie && nsg == 1 && s_cnt[1] - s_cnt[0] == 1) {
      const unsigned gbits = (unsigned)(s_cnt[kTieSumPos] - s_cnt[kTieSumNeg]);
      if (gbits == 0u || splat_bwd_fix_mem_cold<2, HAS_PAD>(bh, b, wg.cgi, gbits, R.Nr, (size_t)wg.cgi * a.gpos_stride, fold_keys, s_cnt))
        tie = false;
    }
  }
  if (tie && nsg == 1) {          // block-uniform: exact ties in this plane — redo it with single-winner claims
    CT_TIE_COUNT(6, 1);
#pragma unroll
    for (int u = 0; u < (QPT ? QPT : 1); ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) gs[u][i][0] = gs[u][i][1] = 0.0f;
    splat_bwd_plane_pass<HAS_PAD, true, WT, QPT>(a, g, ZG, s_cnt, bh, b, wg.cgi, N, R, gs, tie);
  }
  if constexpr (QPT > 0) {
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
      const int n0 = ((int)threadIdx.x + u * (int)blockDim.x) << 2;
      if (n0 < N) {
        const float4 tx = *(const float4*)(a.pos.keys + (bh * 2 + 0) * R.Nr + R.so + n0);
        const float4 ty = *(const float4*)(a.pos.keys + (bh * 2 + 1) * R.Nr + R.so + n0);
        float4 ox = make_float4(gs[u][0][0] * ct_key_mask(tx.x), gs[u][1][0] * ct_key_mask(tx.y),
                                gs[u][2][0] * ct_key_mask(tx.z), gs[u][3][0] * ct_key_mask(tx.w));
        float4 oy = make_float4(gs[u][0][1] * ct_key_mask(ty.x), gs[u][1][1] * ct_key_mask(ty.y),
                                gs[u][2][1] * ct_key_mask(ty.z), gs[u][3][1] * ct_key_mask(ty.w));
        const size_t ox_off = (bh * 2 + 0) * R.Nr + R.so + n0, oy_off = (bh * 2 + 1) * R.Nr + R.so + n0;
        float* px = a.g_pos + (size_t)wg.cgi * a.gpos_stride + ox_off;
        float* py = a.g_pos + (size_t)wg.cgi * a.gpos_stride + oy_off;
        if (a.gpos_add != nullptr) {
          const float4 qx = *(const float4*)(a.gpos_add + ox_off), qy = *(const float4*)(a.gpos_add + oy_off);
          ox.x += qx.x; ox.y += qx.y; ox.z += qx.z; ox.w += qx.w;
          oy.x += qy.x; oy.y += qy.y; oy.z += qy.z; oy.w += qy.w;
        }
        if (fold_keys || R.wt) {
          st_sc1_4(px, ox);
          st_sc1_4(py, oy);
        } else {
          *(float4*)px = ox;
          *(float4*)py = oy;
        }
      }
    }
  }
  if (fold_keys) {       // kernel-uniform: the chunk groups' partial g_keys, added by the plane's last workgroup
    if (arrive_last(a.tickets + bh, (unsigned)a.ncg, nullptr, 0u, (unsigned*)(s_cnt + 2)) & 1u) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fold_rows(a.g_pos + (bh * 2 + j) * N, a.gpos_stride, a.ncg, a.fold_gpos + (bh * 2 + j) * N, N >> 2,
                  a.fold_add != nullptr ? a.fold_add + (bh * 2 + j) * N : nullptr);
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
      if (CT_TIE_FIX && s_cnt[3] - s_cnt[0] == 1) {      // ONE surplus match in the plane: repaired in the segments' rows
        const unsigned gbits = (unsigned)(s_cnt[2] - s_cnt[kTieSumNeg]);
        if (gbits == 0u || splat_bwd_fix_mem_cold<2, HAS_PAD>(bh, b, 0, gbits, R.Nr, 0, false, s_cnt)) redo = false;
      }
      if (redo) {       // the plane has exact ties: all of it again, with claims, by this workgroup
        PtRows Rall;
        Rall.Nr = R.Nr; Rall.so = 0; Rall.wt = false;
        float gs0[1][4][2];
        splat_bwd_plane_pass<HAS_PAD, true, WT, 0>(a, g, ZG, s_cnt, bh, b, 0, R.Nr, Rall, gs0, tie);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// KG: Slice forward (and the g_feat half of Splat(sum) backward) with the channel-interleaved tile:
//   one ds_read_b128 per (point, corner, 4 channels).   grid = (nchunks * nsplit, H, B)
// ---------------------------------------------------------------------------
template <bool HAS_PAD, int WT>
__global__ void __launch_bounds__(kHotThreads, 4) gather_ci_kernel(RasterArgs a, GridW<2> g_arg) {
  const GridW<2> g = grid2_of<WT>(g_arg);
  extern __shared__ __align__(16) float lds[];
  const int G = WT ? WT * WT : g.G, W1 = WT ? WT : g.W[1], N = a.N;
  float4* T4 = (float4*)lds;
  // (a plane's chunk x split workgroups on ONE XCD — block_xhb — re-read its keys, and its tile per split, from that L2)
  const BlockXHB blk = block_xhb();
  const int chunk = blk.x / a.nsplit, sp = blk.x % a.nsplit;
  const int h = blk.h, b = blk.b;
  const size_t bh = (size_t)b * a.H + h;
  const int c0 = chunk * a.CC;
  const int cc = min(a.CC, a.C - c0);            // multiple of 4
  const int tid = threadIdx.x;
  const int off[4] = {0, W1, 1, W1 + 1};
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
    float cw[4][4], pv[4];
    int base[4];
    {
      const float4 tx = *(const float4*)(a.pos.keys + (bh * 2 + 0) * N + n0);
      const float4 ty = *(const float4*)(a.pos.keys + (bh * 2 + 1) * N + n0);
      const float kx[4] = {tx.x, tx.y, tx.z, tx.w}, ky[4] = {ty.x, ty.y, ty.z, ty.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Pt2 p;
        pt2_from_keys(kx[i], ky[i], g, W1, p);
        base[i] = p.base;
#pragma unroll
        for (int v = 0; v < 4; ++v) cw[i][v] = p.cw[v];
        pv[i] = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * N + n0 + i) : 1.0f;
      }
    }
    for (int cq = 0; cq < (cc >> 2); ++cq) {
      const float4* Tq = T4 + (size_t)cq * G;
      float o[4][4];       // [channel][point]
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 cv[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) cv[v] = Tq[base[i] + off[v]];
        // same order as the reference's sum over corners: ((v0 + v1) + v2) + v3
        float s0 = cv[0].x * cw[i][0], s1 = cv[0].y * cw[i][0], s2 = cv[0].z * cw[i][0], s3 = cv[0].w * cw[i][0];
#pragma unroll
        for (int v = 1; v < 4; ++v) {
          s0 += cv[v].x * cw[i][v];
          s1 += cv[v].y * cw[i][v];
          s2 += cv[v].z * cw[i][v];
          s3 += cv[v].w * cw[i][v];
        }
        o[0][i] = HAS_PAD ? s0 * pv[i] : s0;
        o[1][i] = HAS_PAD ? s1 * pv[i] : s1;
        o[2][i] = HAS_PAD ? s2 * pv[i] : s2;
        o[3][i] = HAS_PAD ? s3 * pv[i] : s3;
      }
#pragma unroll
      for (int cj = 0; cj < 4; ++cj)
        st_stream4(dst + (size_t)(cq * 4 + cj) * N + n0, make_float4(o[cj][0], o[cj][1], o[cj][2], o[cj][3]));
    }
  }
}

// ---------------------------------------------------------------------------
// KS: Splat(sum) backward in one pass (linear op: g_feat = Slice(g_grid), corner cotangents gw = sum_c g_grid * feat):
//   the g_grid tile of ALL channels of the plane channel-interleaved in LDS, one ds_read_b128 per (point, corner,
//   4 channels) feeds both results; g_keys is written once (added to an incoming cotangent: a.gpos_add).
//   grid = (nsplit, H, B)
// ---------------------------------------------------------------------------
template <bool HAS_PAD, int WT>
__global__ void __launch_bounds__(kHotThreads, 4) splat_sum_bwd_kernel(RasterArgs a, GridW<2> g_arg) {
  const GridW<2> g = grid2_of<WT>(g_arg);
  extern __shared__ __align__(16) float lds[];
  const int G = WT ? WT * WT : g.G, W1 = WT ? WT : g.W[1], N = a.N, C = a.C;
  float4* T4 = (float4*)lds;
  const int sp = blockIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  const size_t bh = (size_t)b * a.H + h;
  const int tid = threadIdx.x;
  const int off[4] = {0, W1, 1, W1 + 1};
  const float* gin = a.tile_in + bh * C * (size_t)G;
  for (int t = tid; t < (C >> 2) * G; t += blockDim.x) {
    const int cq = t / G, cell = t - cq * G;
    const float* p = gin + (size_t)(cq * 4) * G + cell;
    T4[t] = make_float4(ld_stream(p), ld_stream(p + G), ld_stream(p + 2 * (size_t)G), ld_stream(p + 3 * (size_t)G));
  }
  __syncthreads();
  const int nq = N >> 2;
  const int per = (nq + a.nsplit - 1) / a.nsplit;
  const int q_beg = sp * per, q_end = min(nq, q_beg + per);
  for (int q = q_beg + tid; q < q_end; q += blockDim.x) {
    const int n0 = q << 2;
    const float4 tx = *(const float4*)(a.pos.keys + (bh * 2 + 0) * N + n0);
    const float4 ty = *(const float4*)(a.pos.keys + (bh * 2 + 1) * N + n0);
    const float kx[4] = {tx.x, tx.y, tx.z, tx.w}, ky[4] = {ty.x, ty.y, ty.z, ty.w};
    float pv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pv[i] = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * N + n0 + i) : 1.0f;
    float gs[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) gs[i][0] = gs[i][1] = 0.0f;
    for (int cq = 0; cq < (C >> 2); ++cq) {
      const float4* Tq = T4 + (size_t)cq * G;
      float fv[4][4];       // [channel][point]: feat in, g_feat out
#pragma unroll
      for (int cj = 0; cj < 4; ++cj) {
        const float* row = a.src + (bh * C + cq * 4 + cj) * (size_t)N;
        const float4 t = ld_stream4(row + n0);
        fv[cj][0] = t.x; fv[cj][1] = t.y; fv[cj][2] = t.z; fv[cj][3] = t.w;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Pt2 p;
        pt2_from_keys(kx[i], ky[i], g, W1, p);
        float4 cv[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) cv[v] = Tq[p.base + off[v]];
        float f[4];
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) f[cj] = HAS_PAD ? fv[cj][i] * pv[i] : fv[cj][i];
        float gw[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float s = cv[v].x * f[0];
          s = __builtin_fmaf(cv[v].y, f[1], s);
          s = __builtin_fmaf(cv[v].z, f[2], s);
          s = __builtin_fmaf(cv[v].w, f[3], s);
          gw[v] = s;
        }
        gs[i][0] = __builtin_fmaf(gw[3] - gw[2], p.w1y, __builtin_fmaf(gw[1] - gw[0], p.w0y, gs[i][0]));
        gs[i][1] = __builtin_fmaf(gw[3] - gw[1], p.w1x, __builtin_fmaf(gw[2] - gw[0], p.w0x, gs[i][1]));
        // g_feat: the reference's sum over corners, ((v0 + v1) + v2) + v3
        float s0 = cv[0].x * p.cw[0], s1 = cv[0].y * p.cw[0], s2 = cv[0].z * p.cw[0], s3 = cv[0].w * p.cw[0];
#pragma unroll
        for (int v = 1; v < 4; ++v) {
          s0 += cv[v].x * p.cw[v];
          s1 += cv[v].y * p.cw[v];
          s2 += cv[v].z * p.cw[v];
          s3 += cv[v].w * p.cw[v];
        }
        fv[0][i] = HAS_PAD ? s0 * pv[i] : s0;
        fv[1][i] = HAS_PAD ? s1 * pv[i] : s1;
        fv[2][i] = HAS_PAD ? s2 * pv[i] : s2;
        fv[3][i] = HAS_PAD ? s3 * pv[i] : s3;
        asm volatile("" : "+v"(gs[i][0]), "+v"(gs[i][1]), "+v"(fv[0][i]), "+v"(fv[1][i]), "+v"(fv[2][i]), "+v"(fv[3][i]));
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int cj = 0; cj < 4; ++cj)
        st_stream4(a.dst + (bh * C + cq * 4 + cj) * (size_t)N + n0, make_float4(fv[cj][0], fv[cj][1], fv[cj][2], fv[cj][3]));
    }
    float4 ox = make_float4(gs[0][0] * ct_key_mask(kx[0]), gs[1][0] * ct_key_mask(kx[1]),
                            gs[2][0] * ct_key_mask(kx[2]), gs[3][0] * ct_key_mask(kx[3]));
    float4 oy = make_float4(gs[0][1] * ct_key_mask(ky[0]), gs[1][1] * ct_key_mask(ky[1]),
                            gs[2][1] * ct_key_mask(ky[2]), gs[3][1] * ct_key_mask(ky[3]));
    float* px = a.g_pos + (bh * 2 + 0) * N + n0;
    float* py = a.g_pos + (bh * 2 + 1) * N + n0;
    if (a.gpos_add != nullptr) {       // the incoming key cotangent (may be these very rows: in place)
      const float4 qx = *(const float4*)(a.gpos_add + (bh * 2 + 0) * N + n0), qy = *(const float4*)(a.gpos_add + (bh * 2 + 1) * N + n0);
      ox.x += qx.x; ox.y += qx.y; ox.z += qx.z; ox.w += qx.w;
      oy.x += qy.x; oy.y += qy.y; oy.z += qy.z; oy.w += qy.w;
    }
    *(float4*)px = ox;
    *(float4*)py = oy;
  }
}
