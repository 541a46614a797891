// Plane-resident MHCT core: positions -> Splat(max, zero floor) -> grouped 3^d convolution (+bias)
// -> Slice of one (batch, head) plane in ONE kernel; the rasterised grid z and the convolved grid y live in LDS only
// (reference op sequence: layers/multihead_ct.py:99-107 = DifferentiablePositions -> Splat -> conv -> Slice,
//  layers/cloud_transform.py:72-227).  Built for the three grids of the zoo / headline whose two tiles fit one CU's
// 160 KiB beside the filter bank: 2D 32^2 C16 (147 KiB), 2D 16^2 C16 (48 KiB), 3D 8^3 C32 (150 KiB).
//
// Phases of a workgroup (1024 threads):
//   A  scatter-max of its share of the plane's points into the z tile (ds_max_u32 on the bit patterns of the
//      positive products: positive IEEE floats order like unsigned integers, and only positive products beat the
//      zero floor).  The tile is laid out as the convolution's zero-padded input: [C][halo rows][W], channel stride
//      == 16 mod 32 floats (conflict-free B-operand reads of the matrix instruction).
//   X  (clusters only) a plane handled by S workgroups: every workgroup publishes its partial tile through the
//      workspace, arrives on the plane's counter and merges the partners' tiles (element-wise max) — agent-scope
//      release / acquire as MI355X_MICROARCH.md "Valid forms" prescribes; partners sit on one XCD (block index
//      congruent mod 8), which is a speed matter only.  Few planes (the zoo's B8 x H16 = 128, the decoders' 32) would
//      otherwise leave half or seven eighths of the chip idle.
//   B  per block of 16 output channels: implicit-GEMM convolution on the matrix cores (v_mfma_f32_16x16x4_f32, exact
//      fp32: the quad formulation of gconv_fwd4_kernel, ct_gconv.hip), filter bank streamed through LDS one window
//      row ahead; the result is written to the channel-interleaved gather tile [4][cell] x float4;
//   C  gather (ds_read_b128 per point, corner and 4 channels) for the workgroup's share of the points -> out.
// Optional side outputs: z and y as tensors (training: the backward passes read them), the occupancy count
// |z| > 1e-9 (layers/multihead_ct.py:104-105).
#include "ct_common.h"
#include <atomic>

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

#ifndef CT_CORE_STAMP
#define CT_CORE_STAMP 0      // diagnostic builds: wave 0 of every workgroup records s_memrealtime (100 MHz) at the phase boundaries
#endif
#if CT_CORE_STAMP
#define CT_STAMP(K) do { if (threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + (K)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CT_STAMP(K) do { } while (0)
#endif
#ifndef CT_CORE_ABL
#define CT_CORE_ABL 0        // ablation builds (tools/dev/build_core_abl.sh): 1 no scatter, 2 no MFMA loop, 4 no gather, 8 no exchange
#endif
constexpr int kCoreThreads = 1024;
constexpr int kCoreSlack = 4;                  // floats in front of the z tile: the left neighbour of its first element
constexpr int kSpinLimit = 1 << 22;            // polls of ~0.5 us before a workgroup gives up on its partners

struct CoreArgs {
  const float* keys;     // (B, H*DIM, N)   the lattice (post-tanh keys)
  const float* feat;     // (B, H*C, N)
  const void* pad;       // (B, N) | null
  int pad_dtype;
  const float* w;        // (H*C, C, 3^DIM)
  const float* bias;     // (H*C) | null
  float* out;            // (B, H*C, N)
  float* z_save;         // (B, H*C, G) | null
  float* y_save;         // (B, H*C, G) | null
  long long* occ;        // [1] | null
  float* xch;            // [planes][S][C*G]   partial tiles of a cluster
  unsigned* flags;       // [planes] arrival counters, [planes] done counters, {occupancy sum (2 words), done}: zero between launches
  int* status;           // [1] set to 1 when a cluster gave up waiting (never in a correct run)
  unsigned long long* stamps;   // diagnostic builds only
  int B, H, N;
  int S, SC, SN;         // workgroups per plane = SC (blocks of output channels) x SN (point ranges)
  int planes, xcd_map;
  int spin_limit;        // polls before a workgroup gives up on its partners (kSpinLimit; tests shorten it)
  int fault;             // test hook: the last workgroup of plane 0 arrives late (its partners time out)
};

template <int DIM, int WT, int C>
struct CoreGeom {
  static constexpr int V = 1 << DIM;
  static constexpr int G = DIM == 2 ? WT * WT : WT * WT * WT;
  static constexpr int HS = WT + 2;                                    // halo'd extent of every axis but the fastest
  static constexpr int PLANE0 = DIM == 2 ? HS * WT : HS * HS * WT;
  static constexpr int PLANE = PLANE0 + ((48 - (PLANE0 % 32)) % 32);   // == 16 (mod 32)
  static constexpr int NR = DIM == 2 ? 3 : 9;                          // window rows (every tap axis but the fastest)
  static constexpr int TAPS = NR * 3;
  static constexpr int KB = C / 4;
  static constexpr int NRB = DIM == 2 ? 3 : 1;                         // window rows per staged slice of the bank
  static constexpr int NST = NR / NRB;
  static constexpr int SLICE = NRB * KB * 256;                         // floats: [NRB][KB][4 k][16 co][4 dx]
  static constexpr int NBUF = NST > 1 ? 2 : 1;
  static constexpr int NLD = (SLICE + kCoreThreads - 1) / kCoreThreads;
  static constexpr int Z_FLOATS = kCoreSlack + C * PLANE + 4;
  static constexpr size_t LDS_BYTES = (size_t)(Z_FLOATS + 16 * G + NBUF * SLICE) * 4;
  static_assert(PLANE % 32 == 16 && PLANE >= PLANE0, "channel stride");
  static_assert(C % 16 == 0 && WT % 4 == 0, "16-row blocks, quads");
};

// halo-layout address of compact cell index `cell` (x slowest)
template <int DIM, int WT>
__device__ __forceinline__ int halo_of(int cell) {
  if constexpr (DIM == 2) {
    return cell + WT;                                      // one halo row in front, rows keep their width
  } else {
    const int z = cell % WT, y = (cell / WT) % WT, x = cell / (WT * WT);
    return ((x + 1) * (WT + 2) + (y + 1)) * WT + z;
  }
}

template <int DIM>
struct CorePt {
  float cw[1 << DIM];
  int hb, cb;             // base cell in the halo layout / compact
};

// corner weights and base cell of one point: the op sequence of ct_axis / ct_corners (ct_common.h), which follow
// layers/cloud_transform.py:91-99 and layers/utils.py:100-186 rounding for rounding
template <int DIM, int WT>
__device__ __forceinline__ void core_pt(const float (&k)[DIM], CorePt<DIM>& p) {
  constexpr float hw = (float)(WT - 1) * 0.5f;
  float w0[DIM], w1[DIM];
  int f[DIM];
#pragma unroll
  for (int j = 0; j < DIM; ++j) ct_axis(k[j], hw, WT, w0[j], w1[j], f[j]);
  if constexpr (DIM == 2) {
    p.cb = f[0] * WT + f[1];
    p.hb = p.cb + WT;
    p.cw[0] = w0[0] * w0[1];
    p.cw[1] = w1[0] * w0[1];
    p.cw[2] = w0[0] * w1[1];
    p.cw[3] = w1[0] * w1[1];
  } else {
    p.cb = (f[0] * WT + f[1]) * WT + f[2];
    p.hb = ((f[0] + 1) * (WT + 2) + (f[1] + 1)) * WT + f[2];
    const float xy00 = w0[0] * w0[1], xy10 = w1[0] * w0[1], xy01 = w0[0] * w1[1], xy11 = w1[0] * w1[1];
    p.cw[0] = xy00 * w0[2]; p.cw[1] = xy10 * w0[2]; p.cw[2] = xy01 * w0[2]; p.cw[3] = xy11 * w0[2];
    p.cw[4] = xy00 * w1[2]; p.cw[5] = xy10 * w1[2]; p.cw[6] = xy01 * w1[2]; p.cw[7] = xy11 * w1[2];
  }
}

// corner v = dx + 2 dy (+ 4 dz): offsets relative to the base cell, halo layout and compact
template <int DIM, int WT>
__device__ __forceinline__ constexpr int off_halo(int v) {
  return DIM == 2 ? (v & 1) * WT + (v >> 1) : (v & 1) * (WT + 2) * WT + ((v >> 1) & 1) * WT + (v >> 2);
}
template <int DIM, int WT>
__device__ __forceinline__ constexpr int off_compact(int v) {
  return DIM == 2 ? (v & 1) * WT + (v >> 1) : (v & 1) * WT * WT + ((v >> 1) & 1) * WT + (v >> 2);
}

// write-through store (agent scope): the bytes leave the XCD's L2 when the store completes, so a partner on another XCD
// needs no write-back of this L2 before it may read them (MI355X_MICROARCH.md, publish-large: 3.0 vs 8.2 us per 64 KB)
__device__ __forceinline__ void st_sc1_f4(float* p, float4 v) {
  const ct_f4 t = {v.x, v.y, v.z, v.w};
  // s_nop 1: the wait states of "vector write to the data registers of a >8-byte store" (invisible to the compiler in asm)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}

__device__ __forceinline__ int wave_sum_int(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// slice `st` of the filter bank of output channels co0..co0+15 of head h:
//   ws[((rl*KB + kb)*4 + k)*64 + m*4 + dx] = W[co0 + m][ci = kb*4 + k][tap = (st*NRB + rl)*3 + dx]     (dx == 3: 0)
// TR (backward-data): the bank transposed and flipped, = W[kb*4 + k][co0 + m][taps - 1 - tap]
template <int DIM, int WT, int C, bool TR, int THREADS>
__device__ __forceinline__ void load_wslice(const float* wh, int co0, int st,
                                            float (&r)[(CoreGeom<DIM, WT, C>::SLICE + THREADS - 1) / THREADS]) {
  using Gm = CoreGeom<DIM, WT, C>;
#pragma unroll
  for (int u = 0; u < (Gm::SLICE + THREADS - 1) / THREADS; ++u) {
    const int i = (int)threadIdx.x + u * THREADS;
    const int dx = i & 3, m = (i >> 2) & 15, k = (i >> 6) & 3, rk = i >> 8;
    const int kb = rk % Gm::KB, rl = rk / Gm::KB;
    const int tap = (st * Gm::NRB + rl) * 3 + dx;
    if (!TR) r[u] = (i < Gm::SLICE && dx < 3) ? wh[((size_t)(co0 + m) * C + kb * 4 + k) * Gm::TAPS + tap] : 0.0f;
    else r[u] = (i < Gm::SLICE && dx < 3) ? wh[((size_t)(kb * 4 + k) * C + co0 + m) * Gm::TAPS + (Gm::TAPS - 1 - tap)] : 0.0f;
  }
}
template <int DIM, int WT, int C, int THREADS>
__device__ __forceinline__ void store_wslice(float* ws, const float (&r)[(CoreGeom<DIM, WT, C>::SLICE + THREADS - 1) / THREADS]) {
  using Gm = CoreGeom<DIM, WT, C>;
#pragma unroll
  for (int u = 0; u < (Gm::SLICE + THREADS - 1) / THREADS; ++u) {
    const int i = (int)threadIdx.x + u * THREADS;
    if (i < Gm::SLICE) ws[i] = r[u];
  }
}

// One block of 16 output channels (co0 .. co0 + 15) of a plane's grouped 3^DIM convolution on the matrix cores.  `Zin` is the
// zero-padded input tile [C][PLANE] (kCoreSlack floats of slack in front), `wh` the head's filter bank (C x C x taps); TR reads it
// transposed and flipped (backward-data: the input is then g_y and the result g_z of channels co0 ..).  The result lands in the
// channel-interleaved tile Y4 ([4][G] x float4) and, optionally, in `y_save` (rows of G floats, already offset to co0).  Every
// thread of the workgroup calls it (barriers inside; the first one also fences earlier readers of Y4 / WS, the last one
// publishes Y4).
template <int DIM, int WT, int C, bool TR, int THREADS = kCoreThreads>
__device__ __forceinline__ void core_conv_block(const float* Zin, const float* wh, int co0, const float* bias, float* WS, float4* Y4,
                                                float* y_save) {
  using Gm = CoreGeom<DIM, WT, C>;
  constexpr int G = Gm::G, PLANE = Gm::PLANE, HS = Gm::HS, KB = Gm::KB;
  constexpr int wq = WT >> 2;
  constexpr int nspans = G >> 6;                 // spans of 16 quads = 64 cells
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 15, kq = lane >> 4;
  float wr[(Gm::SLICE + THREADS - 1) / THREADS];
  load_wslice<DIM, WT, C, TR, THREADS>(wh, co0, 0, wr);
  __syncthreads();                             // previous block: gathers done with Y4, bank consumed
  store_wslice<DIM, WT, C, THREADS>(WS, wr);
  __syncthreads();
  // span of this wave.  With fewer spans than waves (16^2: 4, 8^3: 8 of 16) the contraction is split: wave w takes the
  // k-blocks [kpart * KB / NK, ...) of span w % nspans, the partial accumulators of the waves with kpart > 0 meet in LDS
  // (the gather tile is free until the block's result is written there) — four waves per SIMD instead of two hide the
  // operand reads behind the matrix instructions (8^3 C32: 34.6 -> see DESIGN)
  constexpr int NK = (THREADS / 64) / nspans >= 2 && 4 * G >= nspans * 256 ? 2 : 1;
  static_assert(KB % NK == 0, "k-blocks per part");
  const bool has_item = wave < nspans * NK;
  const int kpart = has_item ? wave / nspans : 0;
  const int q = (has_item ? wave % nspans : 0) * 16 + col;
  int off;
  bool bl, br;
  {
    const int xq = q % wq;
    const int x0 = xq * 4;
    bl = x0 == 0;
    br = x0 + 4 == WT;
    if constexpr (DIM == 2) {
      off = (q / wq) * WT + x0;
    } else {
      const int y = (q / wq) % WT, z = q / (wq * WT);
      off = (z * HS + y) * WT + x0;
    }
  }
  floatx4 acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float bv = (bias != nullptr && kpart == 0) ? bias[kq * 4 + r] : 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j][r] = bv;
  }
  for (int st = 0; st < Gm::NST; ++st) {
    if (st + 1 < Gm::NST) load_wslice<DIM, WT, C, TR, THREADS>(wh, co0, st + 1, wr);
    const float* ws = WS + (size_t)(st & (Gm::NBUF - 1)) * Gm::SLICE;
    if (has_item && !(CT_CORE_ABL & 2)) {
#pragma unroll
      for (int rl = 0; rl < Gm::NRB; ++rl) {
        const int r = st * Gm::NRB + rl;
        const int roff = DIM == 2 ? r * WT : ((r / 3) * HS + (r % 3)) * WT;
#pragma unroll 2
        for (int kb = kpart * (KB / NK); kb < (kpart + 1) * (KB / NK); ++kb) {
          const float4 a4 = *(const float4*)__builtin_assume_aligned(ws + ((size_t)(rl * KB + kb) * 4 + kq) * 64 + col * 4, 16);
          const float* rp = Zin + (size_t)(kb * 4 + kq) * PLANE + off + roff;
          const float4 q4 = *(const float4*)__builtin_assume_aligned(rp, 16);
          const float lf = bl ? 0.0f : rp[-1], rt = br ? 0.0f : rp[4];
          const float v[6] = {lf, q4.x, q4.y, q4.z, q4.w, rt};
          const float av[3] = {a4.x, a4.y, a4.z};
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[dx], v[j + dx], acc[j], 0, 0, 0);
        }
      }
    }
    if (st + 1 < Gm::NST) {
      store_wslice<DIM, WT, C, THREADS>(WS + (size_t)((st + 1) & (Gm::NBUF - 1)) * Gm::SLICE, wr);
      __syncthreads();
    }
  }
  if constexpr (NK > 1) {
    float4* scratch = Y4 + (size_t)((has_item ? wave : 0) - nspans) * 256 + lane * 4;      // [wave - nspans][lane][4]
    if (has_item && kpart > 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) scratch[j] = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
    }
    __syncthreads();
    if (has_item && kpart == 0) {
#pragma unroll
      for (int kp = 1; kp < NK; ++kp) {
        const float4* part = Y4 + (size_t)((kp - 1) * nspans + wave) * 256 + lane * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 t = part[j];
          acc[j][0] += t.x; acc[j][1] += t.y; acc[j][2] += t.z; acc[j][3] += t.w;
        }
      }
    }
    __syncthreads();                           // the partials are read: the tile may be overwritten
  }
  if (has_item && kpart == 0) {
    // D_j: row (output channel) = kq*4 + r, column = quad col, element j  ->  Y4[kq][cell] = 4 channels of a cell
#pragma unroll
    for (int j = 0; j < 4; ++j) Y4[(size_t)kq * G + q * 4 + j] = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
    if (y_save != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *(float4*)(y_save + (size_t)(kq * 4 + r) * G + q * 4) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
    }
  }
  __syncthreads();

}

// grid = (planes * S)
template <int DIM, int WT, int C, bool HAS_PAD>
__global__ void __launch_bounds__(kCoreThreads) mhct_core_fwd_kernel(CoreArgs a) {
  using Gm = CoreGeom<DIM, WT, C>;
  constexpr int V = Gm::V, G = Gm::G, PLANE = Gm::PLANE, HS = Gm::HS, KB = Gm::KB;
  constexpr int CG = DIM == 2 ? 4 : 2;           // channel rows requested together in the scatter
  extern __shared__ __align__(16) float lds[];
  __shared__ int s_flag[2];                      // [0] abort, [1] occupancy partial
  float* const Zf = lds + kCoreSlack;
  unsigned* const Zu = (unsigned*)Zf;
  float4* const Y4 = (float4*)(lds + Gm::Z_FLOATS);
  float* const WS = lds + Gm::Z_FLOATS + 16 * G;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = a.S, N = a.N;
  int plane, s;
  {
    const int i = blockIdx.x;
    if (a.xcd_map) {                             // the S workgroups of a plane share i % 8: one XCD (speed only)
      const int x = i & 7, j = i >> 3;
      plane = (j / S) * 8 + x;
      s = j % S;
    } else {
      plane = i / S;
      s = i % S;
    }
  }
  const int b = plane / a.H, h = plane - b * a.H;
  const size_t bh = (size_t)plane;
  const int sc = s / a.SN, sn = s - sc * a.SN;

  CT_STAMP(0);
  // ---- zero the z tile (halo included)
  for (int t = tid; t < (Gm::Z_FLOATS >> 2); t += kCoreThreads) ((float4*)lds)[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 2) s_flag[tid] = 0;
  __syncthreads();

  CT_STAMP(1);
  // ---- A: scatter-max of this workgroup's point range
  const int nq = N >> 2;
  {
    const int per = (nq + S - 1) / S;
    const int q0 = s * per, q1 = min(nq, q0 + per);
    const float* src = a.feat + bh * C * (size_t)N;
    for (int q = q0 + tid; q < ((CT_CORE_ABL & 1) ? 0 : q1); q += kCoreThreads) {
      const int n0 = q << 2;
      float cw[4][V];
      int hb[4];
      {
        float kk[DIM][4];
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
          const float4 t = *(const float4*)(a.keys + (bh * DIM + j) * N + n0);
          kk[j][0] = t.x; kk[j][1] = t.y; kk[j][2] = t.z; kk[j][3] = t.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float k1[DIM];
#pragma unroll
          for (int j = 0; j < DIM; ++j) k1[j] = kk[j][i];
          CorePt<DIM> p;
          core_pt<DIM, WT>(k1, p);
          hb[i] = p.hb;
#pragma unroll
          for (int v = 0; v < V; ++v) cw[i][v] = p.cw[v];
        }
      }
      float pv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pv[i] = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * N + n0 + i) : 1.0f;
      // the next group's rows are requested before this group's atomics are issued (a thread owns one or two quads: without
      // the prefetch every group would expose a full HBM round trip)
      float nx[CG][4];
#pragma unroll
      for (int cj = 0; cj < CG; ++cj) {
        const float4 t = *(const float4*)(src + (size_t)cj * N + n0);
        nx[cj][0] = t.x; nx[cj][1] = t.y; nx[cj][2] = t.z; nx[cj][3] = t.w;
      }
      for (int cg0 = 0; cg0 < C; cg0 += CG) {
        float fv[CG][4];
#pragma unroll
        for (int cj = 0; cj < CG; ++cj)
#pragma unroll
          for (int i = 0; i < 4; ++i) fv[cj][i] = nx[cj][i];
        if (cg0 + CG < C) {
#pragma unroll
          for (int cj = 0; cj < CG; ++cj) {
            const float4 t = *(const float4*)(src + (size_t)(cg0 + CG + cj) * N + n0);
            nx[cj][0] = t.x; nx[cj][1] = t.y; nx[cj][2] = t.z; nx[cj][3] = t.w;
          }
        }
#pragma unroll
        for (int cj = 0; cj < CG; ++cj) {
          unsigned* Tc = Zu + (size_t)(cg0 + cj) * PLANE;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float f = HAS_PAD ? fv[cj][i] * pv[i] : fv[cj][i];
#pragma unroll
            for (int v = 0; v < V; ++v) {
              const float prod = f * cw[i][v];
              if (prod > 0.0f) atomicMax(Tc + hb[i] + off_halo<DIM, WT>(v), __float_as_uint(prod));
            }
          }
        }
      }
    }
  }
  __syncthreads();

  CT_STAMP(2);
  bool aborted = false;
  // ---- X: merge the partial tiles of the plane's workgroups
  if (S > 1 && !(CT_CORE_ABL & 8)) {
    float* mine = a.xch + ((size_t)plane * S + s) * (size_t)(C * G);
    for (int t = tid; t < (C * G) >> 2; t += kCoreThreads) {
      const int c = t / (G >> 2), cell = (t - c * (G >> 2)) << 2;
      st_sc1_f4(mine + (size_t)c * G + cell, *(const float4*)(Zf + (size_t)c * PLANE + halo_of<DIM, WT>(cell)));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: its write-through stores are done
    __syncthreads();
    if (tid == 0) {
      if (a.fault && plane == 0 && s == S - 1)         // (test hook: a partner that arrives after the others gave up)
        for (int i = 0; i < 64 * a.spin_limit; ++i) __builtin_amdgcn_s_sleep(8);
      __hip_atomic_fetch_add(a.flags + plane, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int it = 0;
      bool ok = true;
      while (__hip_atomic_load(a.flags + plane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)S) {
        __builtin_amdgcn_s_sleep(4);
        if (++it > a.spin_limit) {
          ok = false;
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // this CU's L1 (and stale L2 lines) dropped before the reads
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!ok) {
        s_flag[0] = 1;
        atomicExch(a.status, 1);
      }
    }
    __syncthreads();
    aborted = s_flag[0] != 0;                    // block-uniform: this workgroup gave up waiting for its partners
    // (an abort must not end the workgroup here: the plane's counters below, and the launch's occupancy counter at the end
    //  of the kernel, count EVERY workgroup — one that left early would leave them non-zero for every later launch on this
    //  workspace, which would then pass the barrier before its partners' tiles exist.  It skips the work, poisons its
    //  share of the outputs with NaN, and still takes part in the counting.)
    for (int o = 1; o < (aborted ? 0 : S); ++o) {
      const int so = (s + o) % S;
      const float* other = a.xch + ((size_t)plane * S + so) * (size_t)(C * G);
      for (int t = tid; t < (C * G) >> 2; t += kCoreThreads) {
        const int c = t / (G >> 2), cell = (t - c * (G >> 2)) << 2;
        const float4 v = *(const float4*)(other + (size_t)c * G + cell);
        float4* zp = (float4*)(Zf + (size_t)c * PLANE + halo_of<DIM, WT>(cell));
        const float4 m = *zp;                    // all values are >= 0: float max == the scatter's unsigned max
        *zp = make_float4(fmaxf(m.x, v.x), fmaxf(m.y, v.y), fmaxf(m.z, v.z), fmaxf(m.w, v.w));
      }
    }
    __syncthreads();
    // the counters clean up after themselves (no memset per launch): the last workgroup of the plane to have read its
    // partners' tiles — nobody polls the plane's counter any more — zeroes both words for the next launch
    if (tid == 0) {
      unsigned* done = a.flags + a.planes + plane;
      if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)S - 1u) {
        __hip_atomic_store(a.flags + plane, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }

  CT_STAMP(3);
  // ---- side outputs of the merged tile: z (training) and the occupancy count; workgroup s takes C/S channels
  if (aborted) {          // a timed-out cluster: NaN where this workgroup's results would have gone (status word set above)
    const float4 bad = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
    const int cper = C / S, c_lo = s * cper;
    if (a.z_save != nullptr)
      for (int t = tid; t < (cper * G) >> 2; t += kCoreThreads) *(float4*)(a.z_save + (bh * C + c_lo) * (size_t)G + ((size_t)t << 2)) = bad;
    const int nmt_a = (C / 16) / a.SC, per_a = (nq + a.SN - 1) / a.SN, qa0 = sn * per_a, qa1 = min(nq, qa0 + per_a);
    for (int mi = 0; mi < nmt_a; ++mi) {
      const int co0 = (sc * nmt_a + mi) * 16;
      if (a.y_save != nullptr && sn == 0)
        for (int t = tid; t < (16 * G) >> 2; t += kCoreThreads) *(float4*)(a.y_save + (bh * C + co0) * (size_t)G + ((size_t)t << 2)) = bad;
      for (int c = 0; c < 16; ++c)
        for (int qd = qa0 + tid; qd < qa1; qd += kCoreThreads) *(float4*)(a.out + (bh * C + co0 + c) * (size_t)N + ((size_t)qd << 2)) = bad;
    }
  }
  if (!aborted && (a.z_save != nullptr || a.occ != nullptr)) {
    const int cper = C / S, c_lo = s * cper;
    int cnt = 0;
    for (int t = tid; t < (cper * G) >> 2; t += kCoreThreads) {
      const int cl = t / (G >> 2), cell = (t - cl * (G >> 2)) << 2;
      const int c = c_lo + cl;
      const float4 v = *(const float4*)(Zf + (size_t)c * PLANE + halo_of<DIM, WT>(cell));
      cnt += (int)(v.x > 1e-9f) + (int)(v.y > 1e-9f) + (int)(v.z > 1e-9f) + (int)(v.w > 1e-9f);   // z >= 0
      if (a.z_save != nullptr) *(float4*)(a.z_save + (bh * C + c) * (size_t)G + cell) = v;
    }
    if (a.occ != nullptr) {
      cnt = wave_sum_int(cnt);
      if (lane == 0 && cnt) atomicAdd(&s_flag[1], cnt);         // published to the other workgroups at the END of the kernel
    }
  }

  CT_STAMP(4);
  // ---- B + C per block of 16 output channels
  const int nmt = aborted ? 0 : (C / 16) / a.SC;
  const float* wh = a.w + (size_t)h * C * C * Gm::TAPS;
  for (int mi = 0; mi < nmt; ++mi) {
    const int co0 = (sc * nmt + mi) * 16;
    core_conv_block<DIM, WT, C, false>(Zf, wh, co0, a.bias != nullptr ? a.bias + (size_t)h * C + co0 : nullptr, WS, Y4,
                                       (a.y_save != nullptr && sn == 0) ? a.y_save + (bh * C + co0) * (size_t)G : nullptr);

    CT_STAMP(5 + 2 * mi);
    // ---- C: gather this block's 16 channels for the workgroup's point range
    {
      const int per = (nq + a.SN - 1) / a.SN;
      const int q0 = sn * per, q1 = min(nq, q0 + per);
      float* dst = a.out + (bh * C + co0) * (size_t)N;
      for (int qd = q0 + tid; qd < ((CT_CORE_ABL & 4) ? 0 : q1); qd += kCoreThreads) {
        const int n0 = qd << 2;
        float kk[DIM][4];
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
          const float4 t = *(const float4*)(a.keys + (bh * DIM + j) * N + n0);
          kk[j][0] = t.x; kk[j][1] = t.y; kk[j][2] = t.z; kk[j][3] = t.w;
        }
        float pv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pv[i] = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * N + n0 + i) : 1.0f;
#pragma unroll 1
        for (int cq = 0; cq < 4; ++cq) {
          const float4* Tq = Y4 + (size_t)cq * G;
          float o[4][4];      // [channel][point]
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float k1[DIM];
#pragma unroll
            for (int j = 0; j < DIM; ++j) k1[j] = kk[j][i];
            CorePt<DIM> p;
            core_pt<DIM, WT>(k1, p);
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int hv = 0; hv < V / 4; ++hv) {
              float4 cv[4];
#pragma unroll
              for (int v = 0; v < 4; ++v) cv[v] = Tq[p.cb + off_compact<DIM, WT>(hv * 4 + v)];
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
          }
#pragma unroll
          for (int cj = 0; cj < 4; ++cj)
            st_stream4(dst + (size_t)(cq * 4 + cj) * N + n0, make_float4(o[cj][0], o[cj][1], o[cj][2], o[cj][3]));
        }
      }
    }
    CT_STAMP(6 + 2 * mi);
  }
  // ---- occupancy: sum over all workgroups, written (and the words zeroed for the next launch) by the last one to arrive.
  //      Relaxed device-scope atomics — they execute at the memory side, in this lane's program order: the sum's add has
  //      returned before the arrival is counted — placed here, off the critical path (an acquire / release pair in the
  //      middle of the kernel wrote back this XCD's L2 behind the z stores: 6-10 us per launch).
  if (a.occ != nullptr) {
    __syncthreads();
    if (tid == 0) {
      unsigned long long* acc = (unsigned long long*)(a.flags + 2 * a.planes);
      unsigned* done = (unsigned*)(acc + 1);
      (void)__hip_atomic_fetch_add(acc, (unsigned long long)s_flag[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned old = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)(a.planes * S) - 1u) {
        *a.occ = (long long)__hip_atomic_exchange(acc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}


// ---------------------------------------------------------------------------
// LDS-resident BACKWARD of the core for the one grid whose five tiles fit a CU: 2D 16^2 with 16 features per head (z, y, the
// integer accumulators of g_y, g_y as the transposed convolution's input, g_z: 85 KiB; 32^2 C16 and 8^3 C32 would need
// > 190 KiB and run ct_mhct_core_bwd on saved grids).  One workgroup per (batch, head) plane RECOMPUTES z and y from the
// points (nothing of the forward is kept in HBM), then walks the chain backwards:
//   P1 scatter-max z (+ points per base cell)      P2 K = most contributions to a cell      P3 y = conv(z) + bias (matrix cores)
//   P4 per-channel max |g_out|                     P5 Slice backward: g_keys (gather of y at the corners) and the fixed-point
//   scatter-add of g_out into the accumulators (quantum per channel, as slice_bwd_fused_kernel)
//   P6 g_y -> float, padded layout; bias cotangent P7 g_z = transposed conv of g_y (matrix cores)
//   P8 filter cotangent of the plane: a tap per wave, K = the 256 positions (matrix cores), to the workspace
//   P9 Splat(max) backward: a contribution bit-equal to its cell's z claims the cell (ds_cmpst: single winner on exact ties,
//   as torch_scatter's backward) and receives g_z; g_feat, g_keys += .
// The per-plane filter / bias cotangents are added over the batch in a fixed order by core_wgrad_reduce_kernel.
// ---------------------------------------------------------------------------
struct CoreBwdArgs {
  const float* keys;     // (B, H*2, N)
  const float* feat;     // (B, H*C, N)
  const void* pad;
  int pad_dtype;
  const float* w;        // (H*C, C, 9)
  const float* bias;     // (H*C) | null
  const float* g_out;    // (B, H*C, N)
  float* g_feat;         // (B, H*C, N)
  float* g_keys;         // (B, H*2, N)
  float* gw_part;        // [planes][C*C*9]
  float* gb_part;        // [planes][C]
  float* gk_scr;         // [planes][2][N]   Splat's share of g_keys (added to Slice's at the end of the kernel)
  int B, H, N;
  unsigned long long* stamps;   // diagnostic builds only
};

constexpr unsigned kCoreClaimed = 0x7FFFFFFFu;     // a NaN pattern no product of finite inputs has: a claimed cell

__device__ __forceinline__ unsigned core_wave_max_u32(unsigned v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
  return v;
}

// power-of-two quantum of a channel whose sums are bounded by MK = max |src| * max contributions per cell
__device__ __forceinline__ void core_quantum(float MK, float& q, float& iq, bool& fixed) {
  fixed = MK < 1e37f;                              // false for inf / NaN as well
  int ex = 0;
  if (fixed && MK > 0.0f) (void)frexpf(MK, &ex);
  ex = max(ex, -90);
  q = ldexpf(1.0f, ex - 30);
  iq = ldexpf(1.0f, 30 - ex);
}

__device__ __forceinline__ int core_cvt_rpi(float x) {
  int r;
  asm("v_cvt_rpi_i32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

struct CorePt2 {
  float w0x, w1x, w0y, w1y, cw[4];
  int cb;        // compact base cell
};
template <int WT>
__device__ __forceinline__ void core_pt2(float kx, float ky, CorePt2& p) {
  constexpr float hw = (float)(WT - 1) * 0.5f;
  int fx, fy;
  ct_axis(kx, hw, WT, p.w0x, p.w1x, fx);
  ct_axis(ky, hw, WT, p.w0y, p.w1y, fy);
  p.cb = fx * WT + fy;
  p.cw[0] = p.w0x * p.w0y;
  p.cw[1] = p.w1x * p.w0y;
  p.cw[2] = p.w0x * p.w1y;
  p.cw[3] = p.w1x * p.w1y;
}

constexpr int kBwdThreads = 512;     // 8 waves: the register budget of 256 per lane keeps a point's 36 LDS values live without spills

template <int WT, int C, bool HAS_PAD>
__global__ void __launch_bounds__(kBwdThreads) mhct_core_bwd_kernel(CoreBwdArgs a) {
  using Gm = CoreGeom<2, WT, C>;
  constexpr int G = Gm::G, PLANE = Gm::PLANE, TAPS = Gm::TAPS;
  static_assert(C == 16 && Gm::NST == 1 && Gm::G == 256, "one 16-channel block, whole filter bank staged, one wave per 256-cell channel");
  extern __shared__ __align__(16) float lds[];
  float* const Zf = lds + kCoreSlack;
  unsigned* const Zu = (unsigned*)Zf;
  float4* const Y4 = (float4*)(lds + Gm::Z_FLOATS);
  float* const WS = lds + Gm::Z_FLOATS + 16 * G;
  int* const ACC = (int*)(WS + Gm::NBUF * Gm::SLICE);            // [C][G]
  float* const GYf = (float*)(ACC + C * G) + kCoreSlack;          // padded layout, as Zf
  int* const CNT = (int*)(GYf - kCoreSlack + Gm::Z_FLOATS);       // [G]
  unsigned* const SMAX = (unsigned*)(CNT + G);                    // [C]
  unsigned* const SK = SMAX + C;                                  // [1]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int plane = blockIdx.x, N = a.N;
  const int b = plane / a.H, h = plane - b * a.H;
  const size_t bh = (size_t)plane;
  const int nq = N >> 2;
  const float* keyx = a.keys + (bh * 2 + 0) * N;
  const float* keyy = a.keys + (bh * 2 + 1) * N;
  const float* wh = a.w + (size_t)h * C * C * TAPS;
  constexpr int offc[4] = {0, WT, 1, WT + 1};                     // corner v = dx + 2 dy: compact offsets (= padded-layout offsets)

  CT_STAMP(0);
  // ---- P0: zero everything that is accumulated into
  {
    constexpr int n4 = (Gm::Z_FLOATS + 16 * G + Gm::NBUF * Gm::SLICE + C * G + Gm::Z_FLOATS + G + C + 4) >> 2;
    for (int t = tid; t < n4; t += kBwdThreads) ((float4*)lds)[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();

  CT_STAMP(1);
  // ---- P1: z = scatter-max of the plane's points; points per base cell
  for (int q = tid; q < nq; q += kBwdThreads) {
    const int n0 = q << 2;
    const float4 tx = *(const float4*)(keyx + n0), ty = *(const float4*)(keyy + n0);
    const float kx[4] = {tx.x, tx.y, tx.z, tx.w}, ky[4] = {ty.x, ty.y, ty.z, ty.w};
    int cb[4];
    float cw[4][4], pv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      CorePt2 p;
      core_pt2<WT>(kx[i], ky[i], p);
      cb[i] = p.cb;
#pragma unroll
      for (int v = 0; v < 4; ++v) cw[i][v] = p.cw[v];
      atomicAdd(&CNT[p.cb], 1);
      pv[i] = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * N + n0 + i) : 1.0f;
    }
    float nx[4][4];                                   // next channel group's rows, requested before this group's atomics
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) {
      const float4 t = *(const float4*)(a.feat + (bh * C + cj) * (size_t)N + n0);
      nx[cj][0] = t.x; nx[cj][1] = t.y; nx[cj][2] = t.z; nx[cj][3] = t.w;
    }
    for (int c0 = 0; c0 < C; c0 += 4) {
      float fv[4][4];
#pragma unroll
      for (int cj = 0; cj < 4; ++cj)
#pragma unroll
        for (int i = 0; i < 4; ++i) fv[cj][i] = nx[cj][i];
      if (c0 + 4 < C) {
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) {
          const float4 t = *(const float4*)(a.feat + (bh * C + c0 + 4 + cj) * (size_t)N + n0);
          nx[cj][0] = t.x; nx[cj][1] = t.y; nx[cj][2] = t.z; nx[cj][3] = t.w;
        }
      }
#pragma unroll
      for (int cj = 0; cj < 4; ++cj)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float f = HAS_PAD ? fv[cj][i] * pv[i] : fv[cj][i];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const float prod = f * cw[i][v];
            if (prod > 0.0f) atomicMax(Zu + (size_t)(c0 + cj) * PLANE + cb[i] + WT + offc[v], __float_as_uint(prod));
          }
        }
    }
  }
  __syncthreads();

  CT_STAMP(2);
  // ---- P2: K = max over cells of the contributions a cell can receive (points based at it and at its three lower neighbours)
  {
    unsigned kloc = 0;
    for (int X = tid; X < G; X += kBwdThreads) {
      unsigned c = (unsigned)CNT[X];
      if (X >= 1) c += (unsigned)CNT[X - 1];
      if (X >= WT) c += (unsigned)CNT[X - WT];
      if (X >= WT + 1) c += (unsigned)CNT[X - WT - 1];
      kloc = max(kloc, c);
    }
    kloc = core_wave_max_u32(kloc);
    if (lane == 0) atomicMax(SK, kloc);
  }

  CT_STAMP(3);
  // ---- P3: y = conv(z) + bias -> Y4   (barriers inside)
  core_conv_block<2, WT, C, false, kBwdThreads>(Zf, wh, 0, a.bias != nullptr ? a.bias + (size_t)h * C : nullptr, WS, Y4, nullptr);

  CT_STAMP(4);
  // ---- P4: per-channel max |g_out * pad| of the plane
  {
    float m[C];
#pragma unroll
    for (int c = 0; c < C; ++c) m[c] = 0.0f;
    for (int q = tid; q < nq; q += kBwdThreads) {
      float pv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pv[i] = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * N + (q << 2) + i) : 1.0f;
      float4 t[C];
#pragma unroll
      for (int c = 0; c < C; ++c) t[c] = *(const float4*)(a.g_out + (bh * C + c) * (size_t)N + (q << 2));       // all rows in flight
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float tv[4] = {t[c].x, t[c].y, t[c].z, t[c].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x = fabsf(HAS_PAD ? tv[i] * pv[i] : tv[i]);
          m[c] = fmaxf(m[c], (x < __builtin_inff()) ? x : __builtin_inff());      // inf / NaN -> inf
        }
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const unsigned mb = core_wave_max_u32(__float_as_uint(m[c]));               // non-negative floats order like unsigned
      if (lane == 0) atomicMax(&SMAX[c], mb);
    }
  }
  __syncthreads();
  const float Kf = (float)(*SK);

  CT_STAMP(5);
  // ---- P5: Slice backward
  for (int q = tid; q < nq; q += kBwdThreads) {
    const int n0 = q << 2;
    const float4 tx = *(const float4*)(keyx + n0), ty = *(const float4*)(keyy + n0);
    const float kx[4] = {tx.x, tx.y, tx.z, tx.w}, ky[4] = {ty.x, ty.y, ty.z, ty.w};
    float pv[4], gs[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pv[i] = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * N + n0 + i) : 1.0f;
      gs[i][0] = gs[i][1] = 0.0f;
    }
    for (int cq = 0; cq < C / 4; ++cq) {
      float fv[4][4], iq[4];
#pragma unroll
      for (int cj = 0; cj < 4; ++cj) {
        const float4 t = *(const float4*)(a.g_out + (bh * C + cq * 4 + cj) * (size_t)N + n0);
        fv[cj][0] = t.x; fv[cj][1] = t.y; fv[cj][2] = t.z; fv[cj][3] = t.w;
        float qd;
        bool fixed;
        core_quantum(__uint_as_float(SMAX[cq * 4 + cj]) * Kf, qd, iq[cj], fixed);
        if (!fixed) iq[cj] = 0.0f;                    // this channel accumulates IEEE floats (inf / NaN inside): rare
      }
      const float4* Tq = Y4 + (size_t)cq * G;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        CorePt2 p;
        core_pt2<WT>(kx[i], ky[i], p);
        float g4[4];
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) g4[cj] = HAS_PAD ? fv[cj][i] * pv[i] : fv[cj][i];
        float gw[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float4 cv = Tq[p.cb + offc[v]];
          float sacc = cv.x * g4[0];
          sacc = __builtin_fmaf(cv.y, g4[1], sacc);
          sacc = __builtin_fmaf(cv.z, g4[2], sacc);
          sacc = __builtin_fmaf(cv.w, g4[3], sacc);
          gw[v] = sacc;
        }
        gs[i][0] = __builtin_fmaf(gw[3] - gw[2], p.w1y, __builtin_fmaf(gw[1] - gw[0], p.w0y, gs[i][0]));
        gs[i][1] = __builtin_fmaf(gw[3] - gw[1], p.w1x, __builtin_fmaf(gw[2] - gw[0], p.w0x, gs[i][1]));
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) {
          int* Tc = ACC + (size_t)(cq * 4 + cj) * G + p.cb;
          if (iq[cj] != 0.0f) {                       // workgroup-uniform
            const float fq = g4[cj] * iq[cj];         // power-of-two scale: exact
#pragma unroll
            for (int v = 0; v < 4; ++v) atomicAdd(Tc + offc[v], core_cvt_rpi(fq * p.cw[v]));
          } else {
#pragma unroll
            for (int v = 0; v < 4; ++v) atomicAdd((float*)(Tc + offc[v]), g4[cj] * p.cw[v]);
          }
        }
      }
    }
    float4 ox, oy;
    ox.x = gs[0][0] * ct_key_mask(kx[0]); ox.y = gs[1][0] * ct_key_mask(kx[1]); ox.z = gs[2][0] * ct_key_mask(kx[2]); ox.w = gs[3][0] * ct_key_mask(kx[3]);
    oy.x = gs[0][1] * ct_key_mask(ky[0]); oy.y = gs[1][1] * ct_key_mask(ky[1]); oy.z = gs[2][1] * ct_key_mask(ky[2]); oy.w = gs[3][1] * ct_key_mask(ky[3]);
    *(float4*)(a.g_keys + (bh * 2 + 0) * N + n0) = ox;          // Splat's share is added in P9 by the same thread
    *(float4*)(a.g_keys + (bh * 2 + 1) * N + n0) = oy;
  }
  __syncthreads();

  CT_STAMP(6);
  // ---- P6: g_y as floats in the padded layout; the plane's bias cotangent
  for (int t = tid; t < C * G; t += kBwdThreads) {
    const int c = t / G, cell = t - c * G;
    float qd, iqd;
    bool fixed;
    core_quantum(__uint_as_float(SMAX[c]) * Kf, qd, iqd, fixed);
    const int r = ACC[t];
    GYf[(size_t)c * PLANE + cell + WT] = fixed ? (float)r * qd : __int_as_float(r);
  }
  __syncthreads();
  for (int c = wave; c < C; c += kBwdThreads / 64) {  // a wave per channel: 4 cells per lane, then a fixed-order butterfly
    const float4 t = *(const float4*)(GYf + (size_t)c * PLANE + WT + lane * 4);
    float sb = (t.x + t.y) + (t.z + t.w);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sb += __shfl_xor(sb, o, 64);
    if (lane == 0) a.gb_part[bh * C + c] = sb;
  }

  CT_STAMP(7);
  // ---- P7: g_z = conv^T(g_y) -> Y4 (y is dead)   (barriers inside)
  core_conv_block<2, WT, C, true, kBwdThreads>(GYf, wh, 0, nullptr, WS, Y4, nullptr);

  CT_STAMP(8);
  // ---- P8: filter cotangent of this plane: wave w < 9 takes tap w; D[co][ci] += sum over positions g_y[co][p] * z[ci][p + tap]
  for (int tap = wave; tap < TAPS; tap += kBwdThreads / 64) {
    const int dy = tap / 3, dx = tap - dy * 3;
    const int m = lane & 15, k = lane >> 4;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int step = 0; step < G / 4; ++step) {
      const int row = step / (WT / 4), x = (step - row * (WT / 4)) * 4 + k;
      const float av = GYf[(size_t)m * PLANE + (row + 1) * WT + x];                  // A[co = m][pos]
      const int xi = x + dx - 1;
      const float bv = (xi >= 0 && xi < WT) ? Zf[(size_t)m * PLANE + (row + dy) * WT + xi] : 0.0f;    // B[pos][ci = m]
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
    float* gp = a.gw_part + bh * (size_t)(C * C * TAPS);
#pragma unroll
    for (int r = 0; r < 4; ++r) gp[((size_t)(k * 4 + r) * C + m) * TAPS + tap] = acc[r];    // D row = co, column = ci
  }
  if (tid < 2) CNT[tid] = 0;
  __syncthreads();                                    // z is intact until here: P9 may claim cells in place

  CT_STAMP(9);
  // ---- P9: Splat(max) backward.  Pass 0: a contribution bit-equal to its cell's z receives g_z, no claims; matches are
  //      counted against the non-zero cells — equal on a plane without exact ties (exactly one winner per cell).  Otherwise
  //      (duplicated points) pass 1 redoes the plane with compare-and-swap claims: the first tied contribution to swap the
  //      cell's word to the claimed pattern wins.  Splat's share of g_keys goes to a scratch row and is added afterwards.
  {
    int nz = 0;
    for (int t = tid; t < C * G; t += kBwdThreads) {
      const int c = t / G, cell = t - c * G;
      nz += Zu[(size_t)c * PLANE + cell + WT] != 0u;
    }
    nz = wave_sum_int(nz);
    if (lane == 0 && nz) atomicAdd(&CNT[0], nz);       // (CNT is free since P2; zeroed below)
  }
  float* const scr = a.gk_scr + bh * 2 * (size_t)N;
  for (int pass = 0; pass < 2; ++pass) {
    int nm = 0;
    for (int q = tid; q < nq; q += kBwdThreads) {
      const int n0 = q << 2;
      const float4 tx = *(const float4*)(keyx + n0), ty = *(const float4*)(keyy + n0);
      const float kx[4] = {tx.x, tx.y, tx.z, tx.w}, ky[4] = {ty.x, ty.y, ty.z, ty.w};
      float pv[4], gs[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        pv[i] = HAS_PAD ? ct_load_pad(a.pad, a.pad_dtype, (size_t)b * N + n0 + i) : 1.0f;
        gs[i][0] = gs[i][1] = 0.0f;
      }
      for (int cq = 0; cq < C / 4; ++cq) {
        float fv[4][4];
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) {
          const float4 t = *(const float4*)(a.feat + (bh * C + cq * 4 + cj) * (size_t)N + n0);
          fv[cj][0] = t.x; fv[cj][1] = t.y; fv[cj][2] = t.z; fv[cj][3] = t.w;
        }
        const float4* Gq = Y4 + (size_t)cq * G;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          CorePt2 p;
          core_pt2<WT>(kx[i], ky[i], p);
          // every LDS read of the point first: 4 x g_z (4 channels each) and 16 z words
          float4 gz[4];
          unsigned zc[4][4];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            gz[v] = Gq[p.cb + offc[v]];
#pragma unroll
            for (int cj = 0; cj < 4; ++cj) zc[v][cj] = Zu[(size_t)(cq * 4 + cj) * PLANE + p.cb + WT + offc[v]];
          }
          float gw[4] = {0.f, 0.f, 0.f, 0.f}, gf[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const float gzv[4] = {gz[v].x, gz[v].y, gz[v].z, gz[v].w};
#pragma unroll
            for (int cj = 0; cj < 4; ++cj) {
              const float xa = HAS_PAD ? fv[cj][i] * pv[i] : fv[cj][i];
              const float prod = xa * p.cw[v];                  // formed exactly as in P1 (one rounding)
              bool mt = prod > 0.0f && __float_as_uint(prod) == zc[v][cj];
              if (pass == 1 && mt)
                mt = atomicCAS(Zu + (size_t)(cq * 4 + cj) * PLANE + p.cb + WT + offc[v], __float_as_uint(prod), kCoreClaimed) ==
                     __float_as_uint(prod);
              nm += (int)mt;
              const float ga = mt ? gzv[cj] : 0.0f;
              gf[cj] = __builtin_fmaf(ga, p.cw[v], gf[cj]);
              gw[v] = __builtin_fmaf(ga, xa, gw[v]);
            }
          }
          gs[i][0] = __builtin_fmaf(gw[3] - gw[2], p.w1y, __builtin_fmaf(gw[1] - gw[0], p.w0y, gs[i][0]));
          gs[i][1] = __builtin_fmaf(gw[3] - gw[1], p.w1x, __builtin_fmaf(gw[2] - gw[0], p.w0x, gs[i][1]));
#pragma unroll
          for (int cj = 0; cj < 4; ++cj) fv[cj][i] = HAS_PAD ? gf[cj] * pv[i] : gf[cj];
        }
#pragma unroll
        for (int cj = 0; cj < 4; ++cj)
          *(float4*)(a.g_feat + (bh * C + cq * 4 + cj) * (size_t)N + n0) = make_float4(fv[cj][0], fv[cj][1], fv[cj][2], fv[cj][3]);
      }
      *(float4*)(scr + n0) = make_float4(gs[0][0] * ct_key_mask(kx[0]), gs[1][0] * ct_key_mask(kx[1]), gs[2][0] * ct_key_mask(kx[2]),
                                         gs[3][0] * ct_key_mask(kx[3]));
      *(float4*)(scr + N + n0) = make_float4(gs[0][1] * ct_key_mask(ky[0]), gs[1][1] * ct_key_mask(ky[1]), gs[2][1] * ct_key_mask(ky[2]),
                                             gs[3][1] * ct_key_mask(ky[3]));
    }
    if (pass == 1) break;
    nm = wave_sum_int(nm);
    if (lane == 0 && nm) atomicAdd(&CNT[1], nm);
    __syncthreads();
    if (CNT[0] == CNT[1]) break;                      // workgroup-uniform: no exact ties in this plane
  }
  // g_keys = Slice's share (stored in P5) + Splat's: every thread adds the quads it wrote itself
  for (int q = tid; q < nq; q += kBwdThreads) {
    const int n0 = q << 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float* pk = a.g_keys + (bh * 2 + j) * N + n0;
      const float4 u = *(const float4*)pk, v = *(const float4*)(scr + (size_t)j * N + n0);
      *(float4*)pk = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
  }
  CT_STAMP(10);
}

// g_w[h][e] = sum over the batch of the planes' partial filter cotangents, in batch order (bitwise reproducible); g_b likewise
__global__ void __launch_bounds__(256) core_wgrad_reduce_kernel(const float* gw_part, const float* gb_part, float* g_w, float* g_b,
                                                                int B, int H, int nw, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < H * nw) {
    const int h = i / nw, e = i - h * nw;
    float sacc = 0.0f;
    for (int b = 0; b < B; ++b) sacc += gw_part[((size_t)b * H + h) * nw + e];
    g_w[i] = sacc;
  }
  if (g_b != nullptr && i < H * C) {
    const int h = i / C, c = i - h * C;
    float sacc = 0.0f;
    for (int b = 0; b < B; ++b) sacc += gb_part[((size_t)b * H + h) * C + c];
    g_b[i] = sacc;
  }
}

template <int WT, int C>
constexpr size_t core_bwd_lds_bytes() {
  using Gm = CoreGeom<2, WT, C>;
  return (size_t)(Gm::Z_FLOATS + 16 * Gm::G + Gm::NBUF * Gm::SLICE + C * Gm::G + Gm::Z_FLOATS + Gm::G + C + 4) * 4;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct CoreShape {
  int dim, W, C;
  int cu_div;     // clusters grow while planes * S <= CUs / cu_div (measured: the light 16^2 plane is fastest at one
                  // workgroup per TWO CUs' worth of planes — its exchange costs what a halved scatter saves — the
                  // MFMA- and atomics-heavy 8^3 C32 volume at one workgroup per CU)
};
constexpr CoreShape kCoreShapes[] = {{2, 32, 16, 1}, {2, 16, 16, 2}, {3, 8, 32, 1}};

int core_shape_index(int C, int dim, const int* W) {
  if (!W) return -1;
  for (int j = 1; j < dim; ++j)
    if (W[j] != W[0]) return -1;
  for (int i = 0; i < (int)(sizeof(kCoreShapes) / sizeof(kCoreShapes[0])); ++i)
    if (kCoreShapes[i].dim == dim && kCoreShapes[i].W == W[0] && kCoreShapes[i].C == C) return i;
  return -1;
}

int device_cus() {
  static int cus = 0;          // immutable device property, cached
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    else
      cus = 256;
  }
  return cus;
}

std::atomic<unsigned> g_core_flags{0};     // test hook (ct_debug_set_core): bit 0 = no clusters, bit 1 = a late partner, bits 8.. = forced cluster size

// workgroups per plane: enough to cover the chip once, S | N/4 ranges, C % S == 0, at most 8
void core_split(int planes, int C, int N, int cu_div, int& S, int& SC, int& SN) {
  S = 1;
  const unsigned dbg = g_core_flags.load(std::memory_order_relaxed);
  if ((dbg & 1u) == 0) {
    const int cus = device_cus() / cu_div;
    while (S < 8 && planes * S * 2 <= cus && C % (S * 2) == 0 && (N >> 2) >= S * 2 * 64) S *= 2;
  }
  const unsigned forced = dbg >> 8;
  if (forced >= 1 && forced <= 8 && C % forced == 0 && (forced & (forced - 1)) == 0) S = (int)forced;
  SC = (C / 16) < S ? (C / 16) : S;
  SN = S / SC;
}

size_t core_ws_layout(int planes, int S, int C, int G, size_t& flags_off, size_t& status_off) {
  size_t xch = S > 1 ? (size_t)planes * S * C * G * 4 : 0;
  flags_off = (xch + 255) & ~(size_t)255;
  status_off = flags_off + ((size_t)2 * planes + 4) * 4;      // arrival + done counters per plane, {occupancy sum (2 words), done, spare}
#if CT_CORE_STAMP
  return status_off + 16 + (size_t)planes * S * 16 * 8;
#else
  return status_off + 16;
#endif
}

template <int DIM, int WT, int C>
int launch_core(CoreArgs a, hipStream_t st) {
  using Gm = CoreGeom<DIM, WT, C>;
  const dim3 grid(a.planes * a.S);
  if (a.pad_dtype != CT_PAD_NONE) {
    auto k = mhct_core_fwd_kernel<DIM, WT, C, true>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Gm::LDS_BYTES) != hipSuccess) return CT_ELAUNCH;
    CT_CLEAR_ERROR();
    hipLaunchKernelGGL(k, grid, dim3(kCoreThreads), Gm::LDS_BYTES, st, a);
  } else {
    auto k = mhct_core_fwd_kernel<DIM, WT, C, false>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Gm::LDS_BYTES) != hipSuccess) return CT_ELAUNCH;
    CT_CLEAR_ERROR();
    hipLaunchKernelGGL(k, grid, dim3(kCoreThreads), Gm::LDS_BYTES, st, a);
  }
  CT_CHECK_LAUNCH();
  return CT_OK;
}

}  // namespace

extern "C" {

void ct_debug_set_core(unsigned flags) { g_core_flags = flags; }

int ct_mhct_core_supported(int B, int H, int C, int N, int dim, const int* W) {
  if (B <= 0 || H <= 0 || N <= 0 || (N & 3) != 0 || (dim != 2 && dim != 3)) return 0;
  return core_shape_index(C, dim, W) >= 0 ? 1 : 0;
}

size_t ct_mhct_core_workspace_bytes(int B, int H, int C, int N, int dim, const int* W) {
  if (!ct_mhct_core_supported(B, H, C, N, dim, W)) return 0;
  int S, SC, SN;
  core_split(B * H, C, N, kCoreShapes[core_shape_index(C, dim, W)].cu_div, S, SC, SN);
  int G = 1;
  for (int j = 0; j < dim; ++j) G *= W[j];
  size_t fo, so;
  return core_ws_layout(B * H, S, C, G, fo, so);
}

int ct_mhct_core_fwd(const float* keys, const float* feat, const void* pad, int pad_dtype, const float* conv_w,
                     const float* conv_b, float* out, float* z_save, float* y_save, int64_t* occ_count, void* workspace,
                     size_t workspace_bytes, int B, int H, int C, int N, int dim, const int* W, ct_stream_t s) {
  if (!keys || !feat || !conv_w || !out || !ct_mhct_core_supported(B, H, C, N, dim, W)) return CT_EINVAL;
  if (pad_dtype != CT_PAD_NONE && !pad) return CT_EINVAL;
  if (((((uintptr_t)keys) | ((uintptr_t)feat) | ((uintptr_t)out) | ((uintptr_t)z_save) | ((uintptr_t)y_save)) & 15) != 0) return CT_EINVAL;
  hipStream_t st = (hipStream_t)s;
  CoreArgs a;
  a.keys = keys; a.feat = feat; a.pad = pad; a.pad_dtype = pad_dtype; a.w = conv_w; a.bias = conv_b;
  a.out = out; a.z_save = z_save; a.y_save = y_save; a.occ = (long long*)occ_count;
  a.B = B; a.H = H; a.N = N; a.planes = B * H;
  core_split(a.planes, C, N, kCoreShapes[core_shape_index(C, dim, W)].cu_div, a.S, a.SC, a.SN);
  a.xcd_map = (a.planes % 8 == 0) ? 1 : 0;
  {
    const unsigned dbg = g_core_flags.load(std::memory_order_relaxed);
    a.fault = (dbg & 2u) ? 1 : 0;                  // ct_debug_set_core bit 1: fault injection (tests of the abort path)
    a.spin_limit = a.fault ? 256 : kSpinLimit;
  }
  int G = 1;
  for (int j = 0; j < dim; ++j) G *= W[j];
  size_t fo, so;
  const size_t need = core_ws_layout(a.planes, a.S, C, G, fo, so);
  if (!workspace || workspace_bytes < need) return CT_EWORKSPACE;
  a.xch = (float*)workspace;
  a.flags = (unsigned*)((char*)workspace + fo);
  a.status = (int*)((char*)workspace + so);
  a.stamps = (unsigned long long*)((char*)workspace + so + 16);
  const int idx = core_shape_index(C, dim, W);
  if (idx == 0) return launch_core<2, 32, 16>(a, st);
  if (idx == 1) return launch_core<2, 16, 16>(a, st);
  return launch_core<3, 8, 32>(a, st);
}

size_t ct_mhct_core_bwd_workspace_bytes(int B, int H, int C, int N, int dim, const int* W) {
  if (!ct_mhct_core_supported(B, H, C, N, dim, W)) return 0;
  size_t G = 1;
  for (int j = 0; j < dim; ++j) G *= (size_t)W[j];
  const size_t grid_bytes = ((size_t)B * H * C * G * 4 + 255) & ~(size_t)255;
  size_t sub = ct_slice_bwd_workspace_bytes(B, H, C, N, dim, W);
  const size_t wg = ct_gconv_bwd_weight_workspace_bytes(B, H, C, C, dim, W);
  const size_t sp = ct_splat_bwd_ex_workspace_bytes(B, H, C, N, dim, W, CT_REDUCE_MAX0, CT_BWD_ACCUMULATE_KEYS);
  if (wg > sub) sub = wg;
  if (sp > sub) sub = sp;                    // the three passes run one after the other: they share the scratch
  // + Slice's key cotangent, which the Splat backward adds to its own (ct_splat_bwd_tk: a tensor of its own)
  const size_t keys_bytes = ((size_t)B * H * dim * N * 4 + 255) & ~(size_t)255;
  return 2 * grid_bytes + keys_bytes + sub;
}

int ct_mhct_core_bwd(const float* keys, const float* feat, const void* pad, int pad_dtype, const float* conv_w,
                     const float* z, const float* y, const float* g_out, float* g_feat, float* g_keys, float* g_w,
                     float* g_b, void* workspace, size_t workspace_bytes, int B, int H, int C, int N, int dim,
                     const int* W, ct_stream_t s) {
  return ct_mhct_core_bwd_tk(keys, feat, pad, pad_dtype, conv_w, z, y, g_out, g_feat, g_keys, g_w, g_b, workspace,
                             workspace_bytes, nullptr, B, H, C, N, dim, W, s);
}

int ct_mhct_core_bwd_tk(const float* keys, const float* feat, const void* pad, int pad_dtype, const float* conv_w,
                        const float* z, const float* y, const float* g_out, float* g_feat, float* g_keys, float* g_w,
                        float* g_b, void* workspace, size_t workspace_bytes, void* tickets, int B, int H, int C, int N, int dim,
                        const int* W, ct_stream_t s) {
  if (!keys || !feat || !conv_w || !z || !y || !g_out || !g_feat || !g_keys || !g_w) return CT_EINVAL;
  if (!ct_mhct_core_supported(B, H, C, N, dim, W)) return CT_EINVAL;
  const size_t need = ct_mhct_core_bwd_workspace_bytes(B, H, C, N, dim, W);
  if (!workspace || workspace_bytes < need) return CT_EWORKSPACE;
  size_t G = 1;
  for (int j = 0; j < dim; ++j) G *= (size_t)W[j];
  const size_t grid_bytes = ((size_t)B * H * C * G * 4 + 255) & ~(size_t)255;
  const size_t keys_bytes = ((size_t)B * H * dim * N * 4 + 255) & ~(size_t)255;
  float* g_y = (float*)workspace;
  float* g_z = (float*)((char*)workspace + grid_bytes);
  float* g_keys_slice = (float*)((char*)workspace + 2 * grid_bytes);
  void* sub = (char*)workspace + 2 * grid_bytes + keys_bytes;
  const size_t sub_bytes = workspace_bytes - 2 * grid_bytes - keys_bytes;
  // (a second key-cotangent tensor only where the Splat backward's point segments want one: else Slice writes g_keys, Splat adds in place)
  if (tickets == nullptr || ct_splat_bwd_tk_segments(B, H, C, N, dim, W) <= 1) g_keys_slice = g_keys;
  int rc = ct_slice_bwd_tk(keys, y, pad, pad_dtype, g_out, g_y, g_keys_slice, sub, sub_bytes, tickets, B, H, C, N, dim, W, s);
  if (rc != CT_OK) return rc;
  rc = ct_gconv_bwd_data(g_y, conv_w, g_z, B, H, C, C, dim, W, s);
  if (rc != CT_OK) return rc;
  rc = ct_gconv_bwd_weight(z, g_y, g_w, g_b, sub, sub_bytes, B, H, C, C, dim, W, s);
  if (rc != CT_OK) return rc;
  return ct_splat_bwd_tk(keys, feat, pad, pad_dtype, z, g_z, g_feat, g_keys_slice, g_keys, sub, sub_bytes, tickets, B, H, C, N,
                         dim, W, CT_REDUCE_MAX0, s);
}

int ct_mhct_core_bwd_fused_supported(int B, int H, int C, int N, int dim, const int* W) {
  return (B > 0 && H > 0 && N > 0 && (N & 3) == 0 && dim == 2 && W && W[0] == 16 && W[1] == 16 && C == 16) ? 1 : 0;
}

size_t ct_mhct_core_bwd_fused_workspace_bytes(int B, int H, int C, int N, int dim, const int* W) {
  if (!ct_mhct_core_bwd_fused_supported(B, H, C, N, dim, W)) return 0;
#if CT_CORE_STAMP
  return (size_t)B * H * (C * C * 9 + C + 2 * (size_t)N) * 4 + (size_t)B * H * 16 * 8;
#else
  return (size_t)B * H * (C * C * 9 + C + 2 * (size_t)N) * 4;
#endif
}

int ct_mhct_core_bwd_fused(const float* keys, const float* feat, const void* pad, int pad_dtype, const float* conv_w,
                           const float* conv_b, const float* g_out, float* g_feat, float* g_keys, float* g_w, float* g_b,
                           void* workspace, size_t workspace_bytes, int B, int H, int C, int N, int dim, const int* W, ct_stream_t s) {
  if (!keys || !feat || !conv_w || !g_out || !g_feat || !g_keys || !g_w) return CT_EINVAL;
  if (!ct_mhct_core_bwd_fused_supported(B, H, C, N, dim, W)) return CT_EINVAL;
  if (pad_dtype != CT_PAD_NONE && !pad) return CT_EINVAL;
  if (((((uintptr_t)keys) | ((uintptr_t)feat) | ((uintptr_t)g_out) | ((uintptr_t)g_feat) | ((uintptr_t)g_keys)) & 15) != 0) return CT_EINVAL;
  const size_t need = ct_mhct_core_bwd_fused_workspace_bytes(B, H, C, N, dim, W);
  if (!workspace || workspace_bytes < need) return CT_EWORKSPACE;
  hipStream_t st = (hipStream_t)s;
  CoreBwdArgs a;
  a.keys = keys; a.feat = feat; a.pad = pad; a.pad_dtype = pad_dtype; a.w = conv_w; a.bias = conv_b; a.g_out = g_out;
  a.g_feat = g_feat; a.g_keys = g_keys;
  a.gw_part = (float*)workspace;
  a.gb_part = a.gw_part + (size_t)B * H * C * C * 9;
  a.gk_scr = a.gb_part + (size_t)B * H * C;
  a.B = B; a.H = H; a.N = N;
  a.stamps = (unsigned long long*)((char*)workspace + (size_t)B * H * (C * C * 9 + C + 2 * (size_t)N) * 4);
  constexpr size_t lds = core_bwd_lds_bytes<16, 16>();
  const dim3 grid(B * H);
  if (pad_dtype != CT_PAD_NONE) {
    auto k = mhct_core_bwd_kernel<16, 16, true>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return CT_ELAUNCH;
    CT_CLEAR_ERROR();
    hipLaunchKernelGGL(k, grid, dim3(kBwdThreads), lds, st, a);
  } else {
    auto k = mhct_core_bwd_kernel<16, 16, false>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return CT_ELAUNCH;
    CT_CLEAR_ERROR();
    hipLaunchKernelGGL(k, grid, dim3(kBwdThreads), lds, st, a);
  }
  CT_CHECK_LAUNCH();
  const int nw = C * C * 9;
  hipLaunchKernelGGL(core_wgrad_reduce_kernel, dim3((H * nw + 255) / 256), dim3(256), 0, st, a.gw_part, a.gb_part, g_w, g_b, B, H, nw, C);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

int ct_mhct_core_workspace_init(void* workspace, size_t workspace_bytes, int B, int H, int C, int N, int dim, const int* W, ct_stream_t s) {
  if (!workspace || !ct_mhct_core_supported(B, H, C, N, dim, W)) return CT_EINVAL;
  int S, SC, SN;
  core_split(B * H, C, N, kCoreShapes[core_shape_index(C, dim, W)].cu_div, S, SC, SN);
  int G = 1;
  for (int j = 0; j < dim; ++j) G *= W[j];
  size_t fo, so;
  const size_t need = core_ws_layout(B * H, S, C, G, fo, so);
  if (workspace_bytes < need) return CT_EWORKSPACE;
  return hipMemsetAsync((char*)workspace + fo, 0, need - fo, (hipStream_t)s) == hipSuccess ? CT_OK : CT_ELAUNCH;
}

/* 0 = no cluster of the last launches on this workspace gave up waiting for its partners */
int ct_mhct_core_status(const void* workspace, size_t workspace_bytes, int B, int H, int C, int N, int dim, const int* W,
                        int* host_status, ct_stream_t s) {
  if (!workspace || !host_status || !ct_mhct_core_supported(B, H, C, N, dim, W)) return CT_EINVAL;
  int S, SC, SN;
  core_split(B * H, C, N, kCoreShapes[core_shape_index(C, dim, W)].cu_div, S, SC, SN);
  int G = 1;
  for (int j = 0; j < dim; ++j) G *= W[j];
  size_t fo, so;
  if (workspace_bytes < core_ws_layout(B * H, S, C, G, fo, so)) return CT_EWORKSPACE;
  if (hipMemcpyAsync(host_status, (const char*)workspace + so, 4, hipMemcpyDeviceToHost, (hipStream_t)s) != hipSuccess) return CT_ELAUNCH;
  if (hipStreamSynchronize((hipStream_t)s) != hipSuccess) return CT_ELAUNCH;
  return CT_OK;
}

}  // extern "C"
