// Grouped 3x3 / 3x3x3 convolution over the rasterised planes and volumes.
//
// The MHCT blocks convolve every head's grid with its own small filter bank
// (`groups = heads`, C_in = C_out in {4,16,32,64} per group; reference call sites
// layers/multihead_ct.py:50-65, unet2d/unet_parts.py:13-16, layers/v2v_groups.py:26-29).
// Per group this is an implicit GEMM  D[co, pos] = sum_{tap, ci} W[co, ci, tap] * X[ci, pos+tap]
// with a tiny N (=C_out) and K (=C_in*3^d): too small for a library GEMM to tile well.
//
// Kernels in this file:
//   gconv_fwd_kernel        forward and backward-data (bank read transposed + flipped) on the matrix cores:
//                           one exact-fp32 v_mfma_f32_16x16x4_f32 per (tap, 4 input channels, 16 positions);
//                             A (16x4)  = W[co 0..15][ci kb*4..+3][tap]        read from LDS
//                             B (4x16)  = X[ci kb*4..+3][16 positions + tap]   read from the LDS halo tile
//                             D (16x16) = 4 accumulator registers / lane       (row = co, col = position)
//                           a workgroup owns an output tile (TD x TH x W) of one (batch, group); each wave keeps
//                           kP position groups in flight so that one weight read feeds kP MFMAs
//   gconv_c4_kernel         the same two passes for four-channel groups on the vector ALU (a 16x16x4 MFMA would
//                           carry 4 useful rows / K slots of 16)
//   gconv_wrw_ring_kernel   backward-weight: implicit GEMM over K = (batch, positions), operands streamed through
//                           LDS rings by LDS-DMA one phase ahead; MFMA engine (16-channel blocks) or vector-ALU
//                           engine (four-channel groups); partial sums to a workspace, added by
//                           gconv_wrw_reduce_kernel / gconv_c4_wrw_reduce_kernel in a fixed order
//   gconv_bwd_weight_kernel backward-weight for rows that are not a multiple of 4 floats (tile form, float atomics)
#include "ct_common.h"
#include <atomic>
#include <type_traits>

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

// ct_debug_set_gconv: bit 0 = small-volume weight gradient on the vector ALU; bit 1 = four-channel 3D groups never / bit 2 = always
// on the matrix cores;
// bits 8..15 = input channels per group from which
// the forward / backward-data pass takes the K-split kernel (0 = default)
std::atomic<unsigned> t_gconv_debug{0};

constexpr int kThreads = 256;                // small tiles: several workgroups per CU
constexpr int kThreadsBig = 1024;            // tiles that leave room for one or two workgroups per CU
constexpr int kP = 4;                      // position groups per wave sharing one A read
#ifndef CT_GCONV_LDS
#define CT_GCONV_LDS (32 * 1024)
#endif
#ifndef CT_GCONV_LDS_WRW
#define CT_GCONV_LDS_WRW (152 * 1024)
#endif
constexpr int kLdsBudget = CT_GCONV_LDS;         // forward / backward-data tile budget: ~4 workgroups per CU (measured best)
constexpr int kLdsBudgetWrw = CT_GCONV_LDS_WRW;  // backward-weight tile budget
#ifndef CT_WRW_THREADS
#define CT_WRW_THREADS 512
#endif
#ifndef CT_WRW_WANT
#define CT_WRW_WANT 256
#endif
// MFMA engine of the ring kernel: one 512-thread workgroup per CU with the tallest row tile that fits, ~256 workgroups
// (measured against 256 threads / 80 KiB / 512 workgroups: 16^3 C16 128 -> 93 us, 64^2 C16 58 -> 47 us).  The vector-ALU
// engine is 4 waves (wave = input channel).
constexpr int kWrwThreads = CT_WRW_THREADS;
constexpr int kLdsBudgetMax = 152 * 1024;    // wide filter banks / large channel counts: one workgroup per CU

struct GconvArgs {
  const float* x;      // (B, groups*Cin, D, H, W)
  const float* w;      // forward: (groups*Cout, Cin, taps); transposed: (groups*Cin_of_y .. see stage_weights)
  const float* bias;   // (groups*Cout) or null
  float* y;            // (B, groups*Cout, D, H, W)
  int B, groups, Cin, Cout;
  int D, H, W;         // D == 1 for 2D
  int TD, TH;          // output tile (full W)
  int nD, nH;          // tiles along D and H
  int taps;            // 9 or 27
  int transposed;      // backward-data: w is indexed [ci_of_this_conv][co_of_this_conv] swapped + flipped taps
  int plane;           // LDS stride between input channels (== 16 mod 32: conflict-free B reads)
  int Hs, Ws;          // tile rows (TH+2) and row length (= W: no x halo); depth extent is TD+2 (3D) or 1 (2D)
  int KB;              // ceil(Cin / 4)
  int gstride;         // ring kernel: LDS stride between g_y channels
  int TZ;              // ring kernel: depth slices per unit (3D)
  int msplit;          // forward kernels: workgroups that share a tile, each taking every msplit-th 16-row block of output channels
};

// A-operand tile of one group and one 16-row block of output channels, laid out so that the 64
// lanes of a wave read 64 consecutive floats: ws[((tap*KB + kb)*4 + k)*16 + m]
__device__ __forceinline__ void stage_weights(float* ws, const GconvArgs& a, int grp, int mt) {
  const int total = a.taps * a.KB * 64;
  constexpr int kU = 4;                       // loads in flight per thread before the LDS writes
  for (int i0 = threadIdx.x; i0 < total; i0 += kU * blockDim.x) {
    float v[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = i0 + u * blockDim.x;
      const int m = i & 15, k = (i >> 4) & 3, r = i >> 6;
      const int kb = r % a.KB, tap = r / a.KB;
      const int co = mt * 16 + m, ci = kb * 4 + k;
      v[u] = 0.0f;
      if (i < total && co < a.Cout && ci < a.Cin) {
        if (!a.transposed) v[u] = a.w[((size_t)(grp * a.Cout + co) * a.Cin + ci) * a.taps + tap];
        else               v[u] = a.w[((size_t)(grp * a.Cin + ci) * a.Cout + co) * a.taps + (a.taps - 1 - tap)];
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = i0 + u * blockDim.x;
      if (i < total) ws[i] = v[u];
    }
  }
}

// Zero-padded halo tile of `planes` input channels (channels >= a.Cin stay zero).  The tile has a
// halo in depth and height only: rows keep the tensor's own width W, so the interior rows of a depth
// slice are CONTIGUOUS both in HBM and in LDS and move as whole 1-KiB LDS-DMA pieces
// (global_load_lds_dwordx4: LDS address = wave-uniform base + lane*16).  The left/right neighbours
// that fall outside a row are masked at the consumer (an out-of-row read lands on the neighbouring
// row — in bounds thanks to kSlack floats of slack in front of the tile — and is replaced by zero).
constexpr int kSlack = 4;

template <int DIM>
__device__ __forceinline__ void stage_halo_tile(float* xs, const float* xg, const GconvArgs& a, int planes,
                                                int td0, int th0, int lane, int wave, int nwaves, bool clear = true) {
  const int Dz = DIM == 3 ? a.TD + 2 : 1;
  const size_t vol = (size_t)a.D * a.H * a.W;
  const int total = planes * a.plane;
  if (clear) {                                                   // workgroup-uniform
    for (int i = threadIdx.x; i < (total >> 2); i += blockDim.x) ((float4*)xs)[i] = make_float4(0, 0, 0, 0);
    for (int i = ((total >> 2) << 2) + threadIdx.x; i < total; i += blockDim.x) xs[i] = 0.0f;
    __syncthreads();
  }
  // interior rows of this tile: tensor rows [gy_lo, gy_hi) -> tile rows [gy_lo - th0 + 1, ...)
  const int gy_lo = max(th0 - 1, 0), gy_hi = min(th0 + a.TH + 1, a.H);
  const int cnt = (gy_hi - gy_lo) * a.W;                       // contiguous floats per (channel, depth slice)
  const int nslabs = min(planes, a.Cin) * Dz;
  const bool vec = ((a.W & 3) == 0) && ((((uintptr_t)xg) & 15) == 0);
  for (int sl = wave; sl < nslabs; sl += nwaves) {
    const int c = sl / Dz, zz = sl % Dz;
    const int gz = DIM == 3 ? td0 + zz - 1 : 0;
    if (gz < 0 || gz >= a.D) continue;                          // wave-uniform
    const float* src = xg + (size_t)c * vol + ((size_t)gz * a.H + gy_lo) * a.W;
    float* dst = xs + (size_t)c * a.plane + (zz * a.Hs + (gy_lo - th0 + 1)) * a.W;
    if (vec) {
      const int n16 = cnt >> 2;                                 // 16-byte units
      for (int p0 = 0; p0 < n16; p0 += 64) {
        if (p0 + lane < n16)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(p0 + lane) * 4),
                                           (__attribute__((address_space(3))) void*)(dst + p0 * 4), 16, 0, 0);
      }
    } else {
      for (int p0 = 0; p0 < cnt; p0 += 64) {
        if (p0 + lane < cnt)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + p0 + lane),
                                           (__attribute__((address_space(3))) void*)(dst + p0), 4, 0, 0);
      }
    }
  }
}

// LDS-DMA issued through inline asm so that it stays OUT of the compiler's wait-count bookkeeping: with the
// builtin, hipcc puts `s_waitcnt vmcnt(0)` in front of the next LDS read (it cannot tell the ring slots
// apart), which drains the prefetch before the phase it was meant to overlap.  The issuing wave waits
// itself (dma_wait_all) before the barrier that publishes the slot.
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// wait until at most n of this wave's vector-memory operations are outstanding (they retire in issue order): the immediate of
// s_waitcnt must be a constant, hence the switch (n is wave-uniform)
__device__ __forceinline__ void dma_wait_upto(int n) {
#define CT_VMW(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n < 0 ? 0 : (n > 31 ? 31 : n)) {
    CT_VMW(0) CT_VMW(1) CT_VMW(2) CT_VMW(3) CT_VMW(4) CT_VMW(5) CT_VMW(6) CT_VMW(7) CT_VMW(8) CT_VMW(9) CT_VMW(10) CT_VMW(11)
    CT_VMW(12) CT_VMW(13) CT_VMW(14) CT_VMW(15) CT_VMW(16) CT_VMW(17) CT_VMW(18) CT_VMW(19) CT_VMW(20) CT_VMW(21) CT_VMW(22)
    CT_VMW(23) CT_VMW(24) CT_VMW(25) CT_VMW(26) CT_VMW(27) CT_VMW(28) CT_VMW(29) CT_VMW(30) CT_VMW(31)
  }
#undef CT_VMW
}

// grid = (nD*nH, groups, B)
template <int DIM>
__global__ void __launch_bounds__(kThreadsBig) gconv_fwd_kernel(GconvArgs a) {
  extern __shared__ __align__(16) float lds[];
  const int tile = blockIdx.x / a.msplit, ms = blockIdx.x % a.msplit, grp = blockIdx.y, b = blockIdx.z;
  const int td0 = (tile / a.nH) * a.TD, th0 = (tile % a.nH) * a.TH;
  const int td = min(a.TD, a.D - td0), th = min(a.TH, a.H - th0);
  const int Dz = DIM == 3 ? a.TD + 2 : 1;
  const int KB = a.KB;
  float* xs = lds + kSlack;                          // [KB*4][plane], kSlack floats of slack in front
  float* ws = lds + kSlack + (size_t)KB * 4 * a.plane + kSlack;   // [taps][KB][4][16]
  const size_t vol = (size_t)a.D * a.H * a.W;

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
  // ---- stage the input tile with its halo: zero-fill, then one LDS-DMA request per valid row
  //      (global_load_lds_dword: LDS address = wave-uniform row base + lane*4), all in flight at
  //      once — the zero padding is whatever the DMA does not overwrite
  stage_halo_tile<DIM>(xs, a.x + ((size_t)b * a.groups + grp) * a.Cin * vol, a, KB * 4, td0, th0, lane, wave, nwaves);
  const int col = lane & 15, kq = lane >> 4;
  const int npos = td * th * a.W;                   // outputs of this tile (row-major z, y, x)
  const int ngroups16 = (npos + 15) >> 4;
  const int MT = (a.Cout + 15) >> 4;

  for (int mt = ms; mt < MT; mt += a.msplit) {
    __syncthreads();                                 // xs ready / previous ws consumed
    stage_weights(ws, a, grp, mt);
    __syncthreads();
    float bias_r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = mt * 16 + kq * 4 + r;
      bias_r[r] = (a.bias != nullptr && co < a.Cout) ? a.bias[grp * a.Cout + co] : 0.0f;
    }
    for (int pg0 = wave * kP; pg0 < ngroups16; pg0 += nwaves * kP) {
      // window origin of this lane's position in each of the kP groups
      int off[kP];
      bool xl[kP], xr[kP];                                           // position sits on the left / right border
      floatx4 acc[kP];
#pragma unroll
      for (int p = 0; p < kP; ++p) {
        const int pos = min((pg0 + p) * 16 + col, npos - 1);      // clamped: inactive lanes read valid LDS
        const int x = pos % a.W, y = (pos / a.W) % th, z = pos / (a.W * th);
        off[p] = (z * a.Hs + y) * a.W + x;
        xl[p] = x == 0;
        xr[p] = x == a.W - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[p][r] = bias_r[r];
      }
#pragma unroll 9
      for (int tap = 0; tap < (DIM == 3 ? 27 : 9); ++tap) {
        const int dx = tap % 3, dy = (tap / 3) % 3, dz = tap / 9;
        const int toff = (dz * a.Hs + dy) * a.W + dx - 1;
        for (int kb = 0; kb < KB; ++kb) {
          const float av = ws[((tap * KB + kb) * 4 + kq) * 16 + col];
          const float* xb = xs + (size_t)(kb * 4 + kq) * a.plane + toff;
#pragma unroll
          for (int p = 0; p < kP; ++p) {
            float bv = xb[off[p]];
            if (dx == 0) bv = xl[p] ? 0.0f : bv;                   // left neighbour outside the row
            if (dx == 2) bv = xr[p] ? 0.0f : bv;                   // right neighbour outside the row
            acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[p], 0, 0, 0);
          }
        }
      }
      // D: row (output channel) = kq*4 + r, column (position) = col
      float* yg = a.y + ((size_t)b * a.groups + grp) * a.Cout * vol;
#pragma unroll
      for (int p = 0; p < kP; ++p) {
        const int pos = (pg0 + p) * 16 + col;
        if (pg0 + p < ngroups16 && pos < npos) {
          const int x = pos % a.W, y = (pos / a.W) % th, z = pos / (a.W * th);
          const size_t o = ((size_t)(td0 + z) * a.H + (th0 + y)) * a.W + x;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = mt * 16 + kq * 4 + r;
            if (co < a.Cout) yg[(size_t)co * vol + o] = acc[p][r];
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Forward / backward-data on the matrix cores, rows of W % 4 == 0 floats ("quad" form of gconv_fwd_kernel).
// A wave walks spans of 16 quads = 64 positions; lane (col, kq) owns the quad `col` of the span (4 consecutive x
// positions, one row) and input channel kb*4 + kq.  MFMA j of a span takes element j of every lane's quad as its
// column (any assignment of positions to MFMA columns is valid), so per (window row, 4 input channels) ONE aligned
// ds_read_b128 of the tile row plus its two neighbours feed the B operands of 12 MFMAs (3 horizontal taps x 4
// elements) and ONE ds_read_b128 of the filter bank ([row][kb][kq][co][dx 0..2 + pad]) feeds their A operands:
// 4 LDS reads per 12 MFMAs instead of 15.  The 4 accumulators of a lane's output channel are 4 consecutive
// positions: results leave as 16-byte stores (the one-position form stores 4 bytes per lane).
// grid = (nD*nH, groups, B)
// ---------------------------------------------------------------------------
template <int DIM>
__global__ void __launch_bounds__(kThreadsBig) gconv_fwd4_kernel(GconvArgs a) {
  constexpr int NR = DIM == 3 ? 9 : 3;               // window rows (dz, dy)
  extern __shared__ __align__(16) float lds[];
  const int tile = blockIdx.x / a.msplit, ms = blockIdx.x % a.msplit, grp = blockIdx.y, b = blockIdx.z;
  const int td0 = (tile / a.nH) * a.TD, th0 = (tile % a.nH) * a.TH;
  const int td = min(a.TD, a.D - td0), th = min(a.TH, a.H - th0);
  const int KB = a.KB;
  float* xs = lds + kSlack;                                          // [KB*4][plane]
  float* ws = lds + kSlack + (size_t)KB * 4 * a.plane + kSlack;       // [NR][KB][4 kq][16 co][4]
  const size_t vol = (size_t)a.D * a.H * a.W;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
  stage_halo_tile<DIM>(xs, a.x + ((size_t)b * a.groups + grp) * a.Cin * vol, a, KB * 4, td0, th0, lane, wave, nwaves);
  const int col = lane & 15, kq = lane >> 4;
  const int wq = a.W >> 2;
  const int nquads = td * th * wq;
  const int nspans = (nquads + 15) >> 4;
  const int MT = (a.Cout + 15) >> 4;
  float* yg = a.y + ((size_t)b * a.groups + grp) * a.Cout * vol;

  for (int mt = ms; mt < MT; mt += a.msplit) {
    __syncthreads();                                 // xs ready / previous bank consumed
    // filter bank of this 16-row block: ws[((r*KB + kb)*4 + k)*64 + m*4 + dx] = W[co = mt*16+m][ci = kb*4+k][tap = r*3+dx]
    for (int i = threadIdx.x; i < NR * KB * 256; i += blockDim.x) {
      const int dx = i & 3, m = (i >> 2) & 15, k = (i >> 6) & 3, rk = i >> 8;
      const int kb = rk % KB, r = rk / KB;
      const int co = mt * 16 + m, ci = kb * 4 + k, tap = r * 3 + dx;
      float v = 0.0f;
      if (dx < 3 && co < a.Cout && ci < a.Cin)
        v = a.transposed ? a.w[((size_t)(grp * a.Cin + ci) * a.Cout + co) * a.taps + (a.taps - 1 - tap)]
                         : a.w[((size_t)(grp * a.Cout + co) * a.Cin + ci) * a.taps + tap];
      ws[i] = v;
    }
    __syncthreads();
    float bias_r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = mt * 16 + kq * 4 + r;
      bias_r[r] = (a.bias != nullptr && co < a.Cout) ? a.bias[grp * a.Cout + co] : 0.0f;
    }
    for (int sp = wave; sp < nspans; sp += nwaves) {
      const int q = min(sp * 16 + col, nquads - 1);   // clamped: inactive lanes read valid LDS
      const int xq = q % wq, y = (q / wq) % th, z = q / (wq * th);
      const int x0 = xq * 4;
      const bool bl = x0 == 0, br = x0 + 4 == a.W;
      const int off = (z * a.Hs + y) * a.W + x0;
      floatx4 acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] = bias_r[r];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int roff = ((r / 3) * a.Hs + (r % 3)) * a.W;
        for (int kb = 0; kb < KB; ++kb) {
          const float4 a4 = *(const float4*)__builtin_assume_aligned(ws + ((size_t)(r * KB + kb) * 4 + kq) * 64 + col * 4, 16);
          const float* rp = xs + (size_t)(kb * 4 + kq) * a.plane + off + roff;
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
      // D_j: row (output channel) = kq*4 + r, column = quad col, element j
      if (sp * 16 + col < nquads) {
        const size_t o = ((size_t)(td0 + z) * a.H + (th0 + y)) * a.W + x0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = mt * 16 + kq * 4 + r;
          if (co < a.Cout) *(float4*)(yg + (size_t)co * vol + o) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------
// K-split form of the quad MFMA kernel for WIDE groups (more than 32 input channels per group: the grouped Res2D /
// Res3D stacks of the classifier / inpainter encoders, 64 channels per group on 8^3 .. 2^3 volumes;
// model_zoo/scanobject/classifier.py:74-92).  gconv_fwd4_kernel keeps a group's whole input tile and the whole filter
// bank of a 16-row block in LDS — 147 KiB of bank alone at 64 input channels in 3D.  Here the contraction runs over
// blocks of 16 input channels: per block the workgroup stages 16 input planes and the bank slice of ITS 16-row output
// blocks ([mt][row][4 kb][4 kq][16 co][4 dx]), and the accumulators of every (span of 16 quads, 16-row block) item a
// wave owns stay in registers across the blocks (MAXI items per wave).  Backward-data = the same kernel on the
// transposed + flipped bank.  grid = (tiles * msplit, groups, B)
// ---------------------------------------------------------------------------
constexpr int kKBlk = 16;       // input channels per contraction block (4 MFMA k-groups)

// The bank slice of a contraction block sits in LDS IN THE TENSOR'S OWN ORDER, so that staging is a handful of 1-KiB LDS-DMA
// pieces per wave (round 2 gathered it element by element into an MFMA-shaped layout: 18 dependent global loads per thread and
// block, ~40 % of the kernel at 8^3 64->64):
//   forward         rows = output channel:  ws[mtl][co 16][CS],  row = w[co][ci k0..k0+15][taps]   (16 * taps contiguous floats)
//   backward-data   rows = input channel of this pass (= output channel of the tensor):
//                                           ws[mtl][ci 16][CS],  row = w[ci][co mt*16..+15][taps]  (read flipped: taps-1-tap)
// The A operand of lane (co = l & 15, ci = l >> 4) is one ds_read_b32 per tap.  Row strides: forward 16*taps + 4 floats
// (== 4 mod 8: the 16 rows fall on the 8 multiples of 4 twice, the four ci offsets 0, taps, 2 taps, 3 taps on the four
// residues mod 4 — taps is odd), backward 16*taps (== 16 mod 32; co * taps covers 16 banks, + 16 the other 16): two lanes per
// bank, the minimum for 64 lanes.
template <int TAPS, bool TR> struct BankRow { static constexpr int RL = 16 * TAPS, CS = TR ? RL : RL + 4; };

template <int DIM, int MAXI, bool TR>
__global__ void __launch_bounds__(kThreadsBig) gconv_fwd4k_kernel(GconvArgs a, int dma_bank) {
  constexpr int NR = DIM == 3 ? 9 : 3, TAPS = NR * 3;
  constexpr int KBB = kKBlk / 4;
  constexpr int RL = BankRow<TAPS, TR>::RL, CS = BankRow<TAPS, TR>::CS;
  extern __shared__ __align__(16) float lds[];
  const int tile = blockIdx.x / a.msplit, ms = blockIdx.x % a.msplit, grp = blockIdx.y, b = blockIdx.z;
  const int td0 = (tile / a.nH) * a.TD, th0 = (tile % a.nH) * a.TH;
  const int td = min(a.TD, a.D - td0), th = min(a.TH, a.H - th0);
  float* xs = lds + kSlack;                                              // [16][plane]
  float* ws = lds + kSlack + (size_t)kKBlk * a.plane + kSlack;           // [nmt][16][CS]
  const size_t vol = (size_t)a.D * a.H * a.W;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int wq = a.W >> 2;
  const int nquads = td * th * wq;
  const int nspans = (nquads + 15) >> 4;
  const int MT = (a.Cout + 15) >> 4;
  const int nmt = (MT - ms + a.msplit - 1) / a.msplit;                    // 16-row blocks of this workgroup: ms, ms + msplit, ...
  const int nitems = nspans * nmt;
  float* yg = a.y + ((size_t)b * a.groups + grp) * a.Cout * vol;

  floatx4 acc[MAXI][4];
  int it_sp[MAXI], it_mt[MAXI];
#pragma unroll
  for (int u = 0; u < MAXI; ++u) {
    const int item = wave + u * nwaves;
    it_sp[u] = item < nitems ? item / nmt : -1;
    it_mt[u] = item < nitems ? item % nmt : 0;
    const int mt = ms + it_mt[u] * a.msplit;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = mt * 16 + kq * 4 + r;
      const float bv = (a.bias != nullptr && co < a.Cout && item < nitems) ? a.bias[grp * a.Cout + co] : 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[u][j][r] = bv;
    }
  }

  for (int k0 = 0; k0 < a.Cin; k0 += kKBlk) {
    __syncthreads();                                   // the previous block's operands are consumed
    GconvArgs ab = a;
    ab.Cin = min(kKBlk, a.Cin - k0);                   // channels of this block (the rest of the 16 planes stays zero)
    // the halo and the planes past Cin are zero from the first block on: later blocks only overwrite the interior
    stage_halo_tile<DIM>(xs, a.x + (((size_t)b * a.groups + grp) * a.Cin + k0) * vol, ab, kKBlk, td0, th0, lane, wave, nwaves,
                         /*clear=*/k0 == 0 || ab.Cin < kKBlk);
    if (dma_bank) {                                    // whole rows, 16-byte aligned: 2 LDS-DMA pieces per row
      for (int rw = wave; rw < nmt * 16; rw += nwaves) {
        const int mt = ms + (rw >> 4) * a.msplit, row = rw & 15;
        const float* src = TR ? a.w + ((size_t)(grp * a.Cin + k0 + row) * a.Cout + mt * 16) * TAPS
                              : a.w + ((size_t)(grp * a.Cout + mt * 16 + row) * a.Cin + k0) * TAPS;
        float* dst = ws + (size_t)rw * CS;
#pragma unroll
        for (int p0 = 0; p0 < RL / 4; p0 += 64) {
          if (p0 + lane < RL / 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(p0 + lane) * 4),
                                             (__attribute__((address_space(3))) void*)(dst + p0 * 4), 16, 0, 0);
        }
      }
    } else {                                           // ragged channel counts: element by element, zeros past the ends
      for (int i = threadIdx.x; i < nmt * 16 * RL; i += blockDim.x) {
        const int j = i % RL, rw = i / RL;
        const int mt = ms + (rw >> 4) * a.msplit, row = rw & 15;
        const int c16 = j / TAPS, tp = j - c16 * TAPS;
        const int co = TR ? mt * 16 + c16 : mt * 16 + row, ci = TR ? k0 + row : k0 + c16;
        float v = 0.0f;
        if (co < a.Cout && ci < a.Cin)
          v = TR ? a.w[((size_t)(grp * a.Cin + ci) * a.Cout + co) * TAPS + tp] : a.w[((size_t)(grp * a.Cout + co) * a.Cin + ci) * TAPS + tp];
        ws[(size_t)rw * CS + j] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < MAXI; ++u) {
      if (it_sp[u] < 0) continue;                      // wave-uniform
      const int q = min(it_sp[u] * 16 + col, nquads - 1);
      const int xq = q % wq, y = (q / wq) % th, z = q / (wq * th);
      const int x0 = xq * 4;
      const bool bl = x0 == 0, br = x0 + 4 == a.W;
      const int off = (z * a.Hs + y) * a.W + x0;
      // forward: ws[co][ci][tap]; backward-data: ws[ci][co][taps - 1 - tap]
      const float* wm = ws + (size_t)it_mt[u] * 16 * CS + (TR ? kq * CS + col * TAPS + (TAPS - 1) : col * CS + kq * TAPS);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int roff = ((r / 3) * a.Hs + (r % 3)) * a.W;
#pragma unroll
        for (int kb = 0; kb < KBB; ++kb) {
          const float* ap = wm + (TR ? kb * 4 * CS - r * 3 : kb * 4 * TAPS + r * 3);
          const float av[3] = {ap[0], TR ? ap[-1] : ap[1], TR ? ap[-2] : ap[2]};
          const float* rp = xs + (size_t)(kb * 4 + kq) * a.plane + off + roff;
          const float4 q4 = *(const float4*)__builtin_assume_aligned(rp, 16);
          const float lf = bl ? 0.0f : rp[-1], rt = br ? 0.0f : rp[4];
          const float v[6] = {lf, q4.x, q4.y, q4.z, q4.w, rt};
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[u][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[dx], v[j + dx], acc[u][j], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < MAXI; ++u) {
    if (it_sp[u] < 0) continue;
    const int qi = it_sp[u] * 16 + col;
    if (qi < nquads) {
      const int xq = qi % wq, y = (qi / wq) % th, z = qi / (wq * th);
      const size_t o = ((size_t)(td0 + z) * a.H + (th0 + y)) * a.W + xq * 4;
      const int mt = ms + it_mt[u] * a.msplit;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = mt * 16 + kq * 4 + r;
        if (co < a.Cout) *(float4*)(yg + (size_t)co * vol + o) = make_float4(acc[u][0][r], acc[u][1][r], acc[u][2][r], acc[u][3][r]);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Four-channel groups (the zoo's C=4 heads: 128^2 planes, 32^3 volumes) on the vector ALU.
// A 16x16x4 MFMA carries 4 useful rows and 4 useful K slots out of 16 for these filters; the op is
// HBM-bound (9-27 flop/B) once the padding work is gone.  Each lane computes 4 consecutive x positions x 4
// output channels (16 accumulators): per (input channel, dz, dy) one aligned ds_read_b128 of the tile row
// plus its two neighbours feed 48 FMAs, the 12 filter taps of that row are wave-uniform and come from SGPRs
// (scalar loads of the group's 4x4x3^d bank).  Same halo tile, staging and border masking as the MFMA kernel;
// backward-data reads the bank transposed + flipped.  grid = (nD*nH, groups, B)
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// 2^d volumes (the last stage of the Res3D / Res2D stacks: 1024 channels on 2x2x2): with padding 1 every output
// position sees every input position, so the layer is a small dense contraction whose cost is reading the filter bank
// once (7 MB at 1024 -> 1024, groups 16).  One 256-thread workgroup owns COB output channels of one group for the whole
// batch: their filters (a contiguous block, or COB x taps runs for the transposed pass) and the group's inputs sit in LDS;
// lane = (output channel, cloud) keeps the P outputs of its row in registers and reads a filter (7 x b128) and the input
// row (P floats) per input channel — P*P multiply-adds per 7 + P/4 LDS reads; the four waves split the input channels
// and their partial rows meet in LDS.   grid = (ceil(Cout / COB), groups)
// ---------------------------------------------------------------------------
constexpr int kTinyTapStride = 28;     // 27 taps padded to whole b128 reads (2D: 9 -> 12)

template <int DIM>
__global__ void __launch_bounds__(256) gconv_tiny_kernel(GconvArgs a, int COB) {
  constexpr int P = DIM == 3 ? 8 : 4;
  constexpr int TAPS = DIM == 3 ? 27 : 9;
  constexpr int TS = DIM == 3 ? kTinyTapStride : 12;
  extern __shared__ __align__(16) float lds[];
  const int Cin = a.Cin, B = a.B;
  const int wrow = Cin * TS + 4;                       // +4: the COB filter rows start in different bank quads
  const int xrow = Cin * P + 4;
  float* ws = lds;                                     // [COB][Cin][TS]
  float* xs = ws + COB * wrow;                         // [B][Cin][P]
  float* part = xs + B * xrow;                         // [4 waves][items][P]
  const int grp = blockIdx.y, co0 = blockIdx.x * COB;
  const int nco = min(COB, a.Cout - co0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // filters: ws[co][ci][tap] (taps flipped for the transposed pass)
  if (!a.transposed) {
    const float* wg = a.w + ((size_t)(grp * a.Cout + co0) * Cin) * TAPS;          // nco*Cin*TAPS contiguous floats
    for (int i = tid; i < nco * Cin * TAPS; i += 256) {
      const int tap = i % TAPS, r = i / TAPS, ci = r % Cin, co = r / Cin;
      ws[co * wrow + ci * TS + tap] = wg[i];
    }
  } else {
    // a.w is (groups * Cin_here, Cout_here, taps): runs of nco * TAPS floats per input channel of this pass
    for (int i = tid; i < Cin * nco * TAPS; i += 256) {
      const int tap = i % TAPS, r = i / TAPS, co = r % nco, ci = r / nco;
      ws[co * wrow + ci * TS + (TAPS - 1 - tap)] = a.w[((size_t)(grp * Cin + ci) * a.Cout + co0 + co) * TAPS + tap];
    }
  }
  for (int i = tid; i < B * Cin * P; i += 256) {
    const int b = i / (Cin * P), r = i - b * (Cin * P);
    xs[b * xrow + r] = a.x[((size_t)b * a.groups + grp) * Cin * P + r];
  }
  __syncthreads();
  const int items = nco * B;
  for (int it0 = 0; it0 < items; it0 += 64) {
    const int it = it0 + lane;
    const bool live = it < items;
    const int co = live ? it / B : 0, b = live ? it - (it / B) * B : 0;
    float acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p) acc[p] = 0.0f;
    const float* wr = ws + co * wrow;
    const float* xr = xs + b * xrow;
    for (int ci = wave; ci < Cin; ci += 4) {
      float w[TS], x[P];
#pragma unroll
      for (int t = 0; t < TS; t += 4) *(float4*)(w + t) = *(const float4*)(wr + ci * TS + t);
#pragma unroll
      for (int q = 0; q < P; q += 4) *(float4*)(x + q) = *(const float4*)(xr + ci * P + q);
#pragma unroll
      for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int q = 0; q < P; ++q) {
          // position index = (z*2 + y)*2 + x; tap = the offset q - p + 1 per axis
          const int dx = (q & 1) - (p & 1) + 1, dy = ((q >> 1) & 1) - ((p >> 1) & 1) + 1;
          const int dz = DIM == 3 ? ((q >> 2) & 1) - ((p >> 2) & 1) + 1 : 0;
          acc[p] = __builtin_fmaf(w[(dz * 3 + dy) * 3 + dx], x[q], acc[p]);
        }
      }
    }
    if (live) {
#pragma unroll
      for (int p = 0; p < P; p += 4) *(float4*)(part + ((size_t)wave * items + it) * P + p) = *(float4*)(acc + p);
    }
  }
  __syncthreads();
  for (int i = tid; i < items * (P / 4); i += 256) {
    const int it = i / (P / 4), h = i - it * (P / 4);
    const int co = it / B, b = it - co * B;
    float4 s = *(const float4*)(part + (size_t)it * P + h * 4);
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float4 o = *(const float4*)(part + ((size_t)k * items + it) * P + h * 4);
      s.x += o.x, s.y += o.y, s.z += o.z, s.w += o.w;
    }
    if (a.bias) {
      const float bv = a.bias[grp * a.Cout + co0 + co];
      s.x += bv, s.y += bv, s.z += bv, s.w += bv;
    }
    *(float4*)(a.y + (((size_t)b * a.groups + grp) * a.Cout + co0 + co) * P + h * 4) = s;
  }
}

template <int DIM, bool TRANSPOSED>
__global__ void __launch_bounds__(kThreads) gconv_c4_kernel(GconvArgs a) {
  constexpr int C = 4;
  constexpr int NZ = DIM == 3 ? 3 : 1;
  constexpr int TAPS = NZ * 9;
  extern __shared__ __align__(16) float lds[];
  const int tile = blockIdx.x, grp = blockIdx.y, b = blockIdx.z;
  const int td0 = (tile / a.nH) * a.TD, th0 = (tile % a.nH) * a.TH;
  const int td = min(a.TD, a.D - td0), th = min(a.TH, a.H - th0);
  float* xs = lds + kSlack;                          // [4][plane]
  const size_t vol = (size_t)a.D * a.H * a.W;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
  stage_halo_tile<DIM>(xs, a.x + ((size_t)b * a.groups + grp) * C * vol, a, C, td0, th0, lane, wave, nwaves);
  __syncthreads();
  const float* wg = a.w + (size_t)grp * C * C * TAPS;  // wave-uniform: scalar loads
  const int wq = a.W >> 2;                            // quads per row
  const int nquads = td * th * wq;
  float* yg = a.y + ((size_t)b * a.groups + grp) * C * vol;
  for (int q = threadIdx.x; q < nquads; q += blockDim.x) {
    const int xq = q % wq, y = (q / wq) % th, z = q / (wq * th);
    const int x0 = xq * 4;
    const bool bl = x0 == 0, br = x0 + 4 == a.W;
    float acc[C][4];
#pragma unroll
    for (int co = 0; co < C; ++co) {
      const float bv = a.bias ? a.bias[grp * C + co] : 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[co][j] = bv;
    }
#pragma unroll 1
    for (int ci = 0; ci < C; ++ci) {
#pragma unroll 1
      for (int r = 0; r < NZ * 3; ++r) {                // tile rows (dz, dy) of the window; 12 taps per row in SGPRs
        const int dz = r / 3, dy = r % 3;
        const float* rp = xs + (size_t)ci * a.plane + ((z + dz) * a.Hs + (y + dy)) * a.W + x0;
        const float4 q4 = *(const float4*)__builtin_assume_aligned(rp, 16);
        const float lf = bl ? 0.0f : rp[-1], rt = br ? 0.0f : rp[4];
        const float v[6] = {lf, q4.x, q4.y, q4.z, q4.w, rt};
#pragma unroll
        for (int co = 0; co < C; ++co) {
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int tap = r * 3 + dx;
            const float wv = TRANSPOSED ? wg[(ci * C + co) * TAPS + (TAPS - 1 - tap)] : wg[(co * C + ci) * TAPS + tap];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[co][j] = fmaf(wv, v[j + dx], acc[co][j]);
          }
        }
      }
    }
    const size_t o = ((size_t)(td0 + z) * a.H + (th0 + y)) * a.W + x0;
#pragma unroll
    for (int co = 0; co < C; ++co) *(float4*)(yg + (size_t)co * vol + o) = make_float4(acc[co][0], acc[co][1], acc[co][2], acc[co][3]);
  }
}

// ---------------------------------------------------------------------------
// Four-channel groups in 3D (the zoo's 32^3 C4 head) ON THE MATRIX CORES, forward and backward-data.
// With 4 output channels a 16x16x4 MFMA has 12 idle rows.  They are filled with the OTHER TWO DEPTH TAPS: for an input
// slice s, one MFMA per in-plane tap (dy, dx) computes
//     D[(g, co)][pos] += sum_ci W[co][ci][dz(g)][dy][dx] * X[ci][s][y + dy][pos + dx]        g = 0..2, 16 positions of a row
// where lane group g (D rows 4g..4g+3 live in lanes 16g..16g+15) accumulates the output slice z in {s-1, s, s+1} with
// z mod 3 == g, i.e. dz(g) = s - z: the B operand (the input slice) is shared by the three depth taps, 12 of 16 rows and all
// of K (= the 4 input channels) are useful.  A wave walks z for its columns (a row y x 16 positions): 9 MFMAs per slice; after
// slice s the output slice s-1 is complete — the lanes of group (s-1) mod 3 add the bias, store and clear their
// accumulators.  The A operands (3 rotations x 9 taps, one register each) are loaded once.  Input slices stream through a
// ring of three LDS buffers [ci][rows + 2][W + 2] (zero halo: no masks) by LDS-DMA two slices ahead, one barrier per slice.
// grid = (row tiles * depth segments, groups, B); needs W % 16 == 0.
// ---------------------------------------------------------------------------
#ifndef CT_C4M_THREADS
#define CT_C4M_THREADS 512
#endif
#ifndef CT_C4M_COLS
#define CT_C4M_COLS 4
#endif
#ifndef CT_C4M_WGS
#define CT_C4M_WGS 512
#endif
constexpr int kC4mThreads = CT_C4M_THREADS;
constexpr int kC4mCols = CT_C4M_COLS;                 // columns per wave

template <bool TR>
__global__ void __launch_bounds__(kC4mThreads, 4) gconv_c4_mfma3_kernel(GconvArgs a, int TH, int LZ, int nZ, int CSX) {
  extern __shared__ __align__(16) float lds[];
  const int W = a.W, H = a.H, D = a.D, XB = W >> 4;
  const int yt = blockIdx.x / nZ, zs = blockIdx.x % nZ, grp = blockIdx.y, b = blockIdx.z;
  const int th0 = yt * TH, th = min(TH, H - th0);
  const int l0 = zs * LZ, l1 = min(D, l0 + LZ);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int NW = kC4mThreads / 64;
  const int col = lane & 15, kq = lane >> 4;
  const size_t vol = (size_t)D * H * W;
  const float* xg = a.x + ((size_t)b * a.groups + grp) * 4 * vol;
  float* yg = a.y + ((size_t)b * a.groups + grp) * 4 * vol;
  const float* wg = a.w + (size_t)grp * 16 * 27;
  float* ring = lds + kSlack;                                        // [3][4][CSX]: slices, rows th0 - 1 .. th0 + th, no x halo
  float* outb = ring + (size_t)3 * 4 * CSX + kSlack;                 // [2][4][TH * W]: finished output slices

  // A operands: lane (row i = col -> group gi = col >> 2, output channel co = col & 3; k = kq = input channel)
  float A[3][9];
  {
    const int gi = col >> 2, co = col & 3, ci = kq;
#pragma unroll
    for (int rot = 0; rot < 3; ++rot) {
      // slice s (s mod 3 == rot) meets group gi's output slice z (z mod 3 == gi) at depth tap s - z + 1
      const int d = (rot - gi + 3) % 3;                              // (s - z) mod 3: 0 -> tap 1, 1 -> tap 2, 2 -> tap 0
      const int dzt = d == 0 ? 1 : (d == 1 ? 2 : 0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int tap = dzt * 9 + t;
        float v = 0.0f;
        if (gi < 3) v = TR ? wg[(ci * 4 + co) * 27 + (26 - tap)] : wg[(co * 4 + ci) * 27 + tap];
        A[rot][t] = v;
      }
    }
  }
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias[r] = a.bias != nullptr ? a.bias[grp * 4 + r] : 0.0f;

  // this wave's columns (a tile row x 16 positions)
  const int ncols = th * XB;
  int c_off[kC4mCols], c_out[kC4mCols];
  bool c_ok[kC4mCols], c_ml[kC4mCols], c_mr[kC4mCols];
  floatx4 acc[kC4mCols];
#pragma unroll
  for (int u = 0; u < kC4mCols; ++u) {
    const int c = wave + u * NW;
    c_ok[u] = c < ncols;
    const int yrow = c_ok[u] ? c / XB : 0, xb = c_ok[u] ? c % XB : 0;         // (a column past the tile reads column 0's operands)
    c_off[u] = kq * CSX + yrow * W + xb * 16 + col - 1;              // operand of (dy = 0, dx = 0): tile row yrow, x - 1
    c_out[u] = yrow * W + xb * 16 + col;
    c_ml[u] = xb == 0 && col == 0;                                   // the left / right neighbour is outside the row
    c_mr[u] = xb == XB - 1 && col == 15;
    acc[u] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
  }

  for (int i = threadIdx.x; i < 3 * 4 * CSX + 2 * kSlack; i += kC4mThreads) lds[i] = 0.0f;       // rows outside the image stay zero
  __syncthreads();
  // The rows of a (slice, channel) that exist in the image are contiguous in HBM and in LDS: 16-byte LDS-DMA pieces of 64
  // lanes, dealt to the waves; issued outside the compiler's wait-count bookkeeping (glds16).
  const int gy_lo = max(th0 - 1, 0), gy_hi = min(th0 + th + 1, H);
  const int n16 = (gy_hi - gy_lo) * W / 4;                            // 16-byte units per (slice, channel)
  const int ppc = (n16 + 63) / 64;                                    // pieces per channel
  constexpr int kMaxPc = 2;                                           // pieces per wave and slice (host: 4 * ppc <= kMaxPc * waves)
  int np = 0, pc_src[kMaxPc], pc_dst[kMaxPc];
  bool pc_on[kMaxPc];
#pragma unroll
  for (int k = 0; k < kMaxPc; ++k) {
    const int pc = wave + k * NW;
    const bool on = pc < 4 * ppc;
    const int ci = on ? pc / ppc : 0, p0 = on ? (pc - ci * ppc) * 64 : 0;
    pc_on[k] = on && p0 + lane < n16;
    np += on ? 1 : 0;
    pc_src[k] = ci * (int)vol + gy_lo * W + (p0 + lane) * 4;          // + s * H * W
    pc_dst[k] = (ci * CSX + (gy_lo - th0 + 1) * W + p0 * 4) * 4;       // bytes, + ring buffer
  }
  const unsigned ring_lds = lds_addr(ring);
  auto load_slice = [&](int s, int rb) {
    const float* src = xg + (size_t)s * H * W;
#pragma unroll
    for (int k = 0; k < kMaxPc; ++k) {
      if (wave + k * NW < 4 * ppc) {                                   // wave-uniform
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)(rb * 4 * CSX * 4 + pc_dst[k]));
        if (pc_on[k]) glds16(src + pc_src[k], dst);
      }
    }
  };
  // a finished slice goes to LDS (lanes of its group) and leaves as whole 16-byte rows one step later
  const int n4 = th * W;                                              // float4 units of a slice: [4][th * W / 4]
  constexpr int kMaxSt = 2;                                           // 16-byte stores per thread and slice (host: n4 <= kMaxSt * threads)
  int nst = 0, st_dst[kMaxSt];
  bool st_on[kMaxSt];
#pragma unroll
  for (int k = 0; k < kMaxSt; ++k) {
    const int t = threadIdx.x + k * kC4mThreads;
    st_on[k] = t < n4;
    nst += wave * 64 + k * kC4mThreads < n4 ? 1 : 0;                  // store instructions of this wave per slice
    const int per = th * W / 4, co = st_on[k] ? t / per : 0, i = t - co * per;
    st_dst[k] = co * (int)vol + th0 * W + i * 4;                      // + z * H * W
  }
  auto flush = [&](int z) {                                          // output slice z is complete (or out of range)
    const int gd = ((z % 3) + 3) % 3;
    if (kq == gd) {
      if (z >= l0 && z < l1) {
        float* ob = outb + (size_t)(z & 1) * 4 * TH * W;
#pragma unroll
        for (int u = 0; u < kC4mCols; ++u) {
          if (c_ok[u]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r * th * W + c_out[u]] = acc[u][r] + bias[r];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kC4mCols; ++u) acc[u] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  };
  auto store_slice = [&](int z) {                                    // -> number of store instructions issued
    if (z < l0 || z >= l1) return 0;
    const float4* ob = (const float4*)(outb + (size_t)(z & 1) * 4 * TH * W);
    float* dst = yg + (size_t)z * H * W;
#pragma unroll
    for (int k = 0; k < kMaxSt; ++k)
      if (st_on[k]) *(float4*)(dst + st_dst[k]) = ob[threadIdx.x + k * kC4mThreads];
    return nst;
  };
  unsigned m_l[kC4mCols], m_r[kC4mCols];                              // all ones, or zero where the neighbour is outside the row
#pragma unroll
  for (int u = 0; u < kC4mCols; ++u) { m_l[u] = c_ml[u] ? 0u : ~0u; m_r[u] = c_mr[u] ? 0u : ~0u; }
  auto step = [&](int rb, auto rotc) {                               // rb: ring buffer of the slice
    constexpr int rot = decltype(rotc)::value;
    const float* xs = ring + (size_t)rb * 4 * CSX;
    // The columns' MFMAs alternate (independent accumulators); a column past the tile multiplies column 0's operands and is
    // never stored.  Row dy + 1 is requested before row dy is multiplied; the out-of-row neighbours are masked by an AND (a
    // select would become a predicated load: two EXEC updates per operand).
    float v[2][kC4mCols][3];
    auto load_row = [&](int dy, float (&w)[kC4mCols][3]) {
#pragma unroll
      for (int u = 0; u < kC4mCols; ++u) {
        const float* bp = xs + c_off[u] + dy * W;
        w[u][0] = __uint_as_float(__float_as_uint(bp[0]) & m_l[u]);
        w[u][1] = bp[1];
        w[u][2] = __uint_as_float(__float_as_uint(bp[2]) & m_r[u]);
      }
    };
    load_row(0, v[0]);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      if (dy < 2) load_row(dy + 1, v[(dy + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int u = 0; u < kC4mCols; ++u)
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rot][dy * 3 + dx], v[dy & 1][u][dx], acc[u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Step s: wait for this wave's pieces of slice s only — the operations issued after them (the stores of the two previous
  // steps and the pieces of slice s + 1) stay in flight —, barrier, request slice s + 2, send slice s - 2 out, multiply,
  // park slice s - 1.
  const int s_beg = max(l0 - 1, 0), s_end = min(l1, D - 1);
  int rb = s_beg % 3, rot = rb;                       // ring buffer of slice s (= s mod 3, which is also the A rotation)
  load_slice(s_beg, rb);
  if (s_beg + 1 <= s_end) load_slice(s_beg + 1, (rb + 1) % 3);
  int st1 = 0, st2 = 0;                               // stores issued one / two steps ago
  for (int s = s_beg; s <= s_end; ++s) {
    const int later = s == s_beg ? (s + 1 <= s_end ? np : 0)
                                 : st2 + (s + 1 <= s_end ? np : 0) + st1;
    dma_wait_upto(later);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS writes of the previous step (flush) are done
    __builtin_amdgcn_s_barrier();                      // slice s has landed; slice s - 1 is consumed, slice s - 2 parked by every wave
    const int rb2 = rb == 0 ? 2 : rb - 1;              // (s + 2) mod 3
    if (s + 2 <= s_end) load_slice(s + 2, rb2);
    st2 = st1;
    st1 = store_slice(s - 2);
    if (rot == 0) step(rb, std::integral_constant<int, 0>());
    else if (rot == 1) step(rb, std::integral_constant<int, 1>());
    else step(rb, std::integral_constant<int, 2>());
    flush(s - 1);
    rb = rb == 2 ? 0 : rb + 1;
    rot = rb;
  }
  __syncthreads();
  store_slice(s_end - 1);
  if (l1 == D) {                                       // the last slice of the volume has no successor
    flush(D - 1);
    __syncthreads();
    store_slice(D - 1);
  }
}

// ---------------------------------------------------------------------------
// backward wrt the filter bank (and bias):
//   g_w[co, ci, tap] = sum_{b, pos} g_y[b, co, pos] * x[b, ci, pos + tap]
// implicit GEMM with K = (batch, positions): A (16x4) = g_y[co 0..15][4 positions], B (4x16) =
// x[ci 0..15][the same 4 positions + tap].  A workgroup owns one group and a CHUNK of the
// (batch, tile) units: it walks its units one after the other (tiles of x with halo and of g_y
// staged in LDS), keeping all 3^d accumulators of one 16x16 filter block in registers — so one
// A read feeds 9 MFMAs and the cross-wave reduction + the float atomics on g_w happen once per
// workgroup, not once per tile.  (g_w is zeroed first; the summation order across workgroups is
// not fixed: last bits may differ from run to run.)
// grid = (unit chunks, groups)
// ---------------------------------------------------------------------------
template <int DIM>
__global__ void __launch_bounds__(kThreads) gconv_bwd_weight_kernel(GconvArgs a, const float* gy, float* gw, int units_per_wg) {
  constexpr int NZ = DIM == 3 ? 3 : 1;                  // z-slabs of 9 taps
  extern __shared__ __align__(16) float lds[];
  const int grp = blockIdx.y;
  const int ntiles = a.nD * a.nH;
  const int U = a.B * ntiles;
  const int u_beg = blockIdx.x * units_per_wg, u_end = min(U, u_beg + units_per_wg);
  const int CiB = (a.Cin + 15) >> 4, CoB = (a.Cout + 15) >> 4;
  const size_t vol = (size_t)a.D * a.H * a.W;
  const int gstride_max = ((a.TD * a.TH * a.W + 3) & ~3) | 1;
  float* xs = lds + kSlack;                             // [16][plane]      input block (zero beyond Cin)
  float* gs = lds + kSlack + (size_t)16 * a.plane + kSlack;   // [16][gstride]    g_y block (zero beyond Cout / npos)
  float* red = gs + (size_t)16 * gstride_max;           // [nwaves][3][256] partials of 3 taps at a time
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int col = lane & 15, kq = lane >> 4;

  for (int cob = 0; cob < CoB; ++cob) {
    for (int cib = 0; cib < CiB; ++cib) {
      floatx4 acc[NZ * 9];
#pragma unroll
      for (int t = 0; t < NZ * 9; ++t) acc[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
      for (int u = u_beg; u < u_end; ++u) {
        const int b = u / ntiles, tile = u % ntiles;
        const int td0 = (tile / a.nH) * a.TD, th0 = (tile % a.nH) * a.TH;
        const int td = min(a.TD, a.D - td0), th = min(a.TH, a.H - th0);
        const int npos = td * th * a.W;
        const int nk = (npos + 3) & ~3;                   // positions walked, 4 per MFMA
        const int gstride = nk | 1;                       // odd: the 16 rows of an A read hit 16 banks
        __syncthreads();                                  // previous unit fully consumed
        // input block cib (16 channels) with halo
        {
          GconvArgs sub = a;
          sub.Cin = min(16, a.Cin - cib * 16);
          stage_halo_tile<DIM>(xs, a.x + (((size_t)b * a.groups + grp) * a.Cin + cib * 16) * vol, sub, 16, td0, th0, lane,
                               wave, nwaves);
        }
        // g_y block cob: the th rows of a depth slice are contiguous in memory (full-width tile)
        {
          const float* gg = gy + (((size_t)b * a.groups + grp) * a.Cout + cob * 16) * vol;
          const int slab = th * a.W;
          for (int row = wave; row < 16 * td; row += nwaves) {
            const int c = row / td, z = row % td;
            float* dst = gs + (size_t)c * gstride + z * slab;
            const float* src = gg + (size_t)c * vol + ((size_t)(td0 + z) * a.H + th0) * a.W;
            const bool ok = cob * 16 + c < a.Cout;
            for (int i = lane; i < slab; i += 64) dst[i] = ok ? src[i] : 0.0f;
          }
          for (int c = threadIdx.x; c < 16; c += blockDim.x)
            for (int i = npos; i < gstride; ++i) gs[(size_t)c * gstride + i] = 0.0f;
        }
        __syncthreads();
        // this wave's share of the unit's positions (multiple of 4)
        const int per = (((nk >> 2) + nwaves - 1) / nwaves) << 2;
        const int p_beg = min(nk, wave * per), p_end = min(nk, p_beg + per);
        const float* ga = gs + (size_t)col * gstride;                         // A: row = co, k = position
        const float* xb = xs + (size_t)col * a.plane;                          // B: col = ci
        int pos = min(p_beg + kq, npos - 1);
        int x = pos % a.W, y = (pos / a.W) % th, z = pos / (a.W * th);
        for (int p0 = p_beg; p0 < p_end; p0 += 4) {
          const float av = ga[p0 + kq];                    // positions >= npos carry g_y = 0
          const float* xp = xb + (z * a.Hs + y) * a.W + x - 1;
          const bool bl = x == 0, br = x == a.W - 1;
#pragma unroll
          for (int tz = 0; tz < NZ; ++tz)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
              float bv = xp[(tz * a.Hs + t / 3) * a.W + (t % 3)];
              if (t % 3 == 0) bv = bl ? 0.0f : bv;
              if (t % 3 == 2) bv = br ? 0.0f : bv;
              acc[tz * 9 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[tz * 9 + t], 0, 0, 0);
            }
          x += 4;
          while (x >= a.W) { x -= a.W; ++y; }
          while (y >= th) { y -= th; ++z; }
          if (z >= td) { z = td - 1; y = th - 1; x = a.W - 1; }     // padded tail: stay in bounds (g_y is 0 there)
        }
      }
      // sum the wave partials through LDS, three taps per round; D: row (co) = kq*4 + r, column (ci) = col
#pragma unroll
      for (int t3 = 0; t3 < NZ * 3; ++t3) {
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(wave * 3 + tt) * 256 + (kq * 4 + r) * 16 + col] = acc[t3 * 3 + tt][r];
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) {
          float sum = 0.0f;
          for (int w2 = 0; w2 < nwaves; ++w2) sum += red[w2 * 3 * 256 + i];
          const int t = t3 * 3 + (i >> 8), co = cob * 16 + ((i >> 4) & 15), ci = cib * 16 + (i & 15);
          if (co < a.Cout && ci < a.Cin)
            atomicAdd(&gw[((size_t)(grp * a.Cout + co) * a.Cin + ci) * a.taps + t], sum);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// backward wrt the filter bank, ring form.  Same implicit GEMM as above, but the operands stream
// through LDS rings filled by LDS-DMA one phase ahead of the MFMAs, so loads overlap the matrix work
// and nothing is staged twice along the walking axis:
//   3D: a unit is (batch, row tile, depth chunk); the workgroup walks the depth slices z of the chunk.
//       Phase z multiplies g_y slice z (ring of 2) with x slices z-1, z, z+1 (ring of 4: the slice for
//       phase z+1 lands while phase z computes).  Out-of-range slices are not loaded — their 9 taps are
//       skipped (wave-uniform) instead of multiplying zeros.
//   2D: a unit is (batch, row tile); the x / g_y tiles of unit u+1 land while unit u computes (rings of 2).
// Tile rows keep the tensor width (no x halo; left/right neighbours masked at the consumer), the row
// halo is one row above and below; rows outside the image are zero-filled by the loading wave.
// grid = (unit chunks, groups); the 3^d accumulators live in registers across all units of the
// workgroup; one cross-wave reduction + float atomics on g_w at the end (g_w zeroed first).
// ---------------------------------------------------------------------------
// `count` contiguous floats -> LDS at dst (wave-uniform); 16-byte pieces when src / dst / count allow
__device__ __forceinline__ void dma_run(float* dst, const float* src, int count, int lane, bool vec) {
  const unsigned base = __builtin_amdgcn_readfirstlane(lds_addr(dst));
  if (vec) {
    const int n16 = count >> 2;
    for (int p0 = 0; p0 < n16; p0 += 64)
      if (p0 + lane < n16) glds16(src + (size_t)(p0 + lane) * 4, base + p0 * 16);
  } else {
    for (int p0 = 0; p0 < count; p0 += 64)
      if (p0 + lane < count) glds4(src + p0 + lane, base + p0 * 4);
  }
}

// The MFMAs of one K group (16 positions) of the ring kernel for the slices in ZMASK.  Rows are
// fetched one ahead of the row being multiplied (two 6-float register sets).
template <int NZ, int ZMASK>
__device__ __forceinline__ void wrw_group(floatx4 (&acc)[NZ * 9], float& gsum, const float* const (&xb)[NZ], const float* ga, int o,
                                          int W, bool bl, bool br) {
  constexpr int NR = NZ * 3;
  const float4 a4 = *(const float4*)__builtin_assume_aligned(ga, 16);      // zero beyond the tile's positions
  const float av[4] = {a4.x, a4.y, a4.z, a4.w};
  gsum += (a4.x + a4.y) + (a4.z + a4.w);                 // bias gradient rides along: sum of g_y over positions
  float w[NR][6];
  auto fetch = [&](int r) {
    const float* rp = xb[r / 3] + o + (r % 3) * W;
    const float4 q4 = *(const float4*)__builtin_assume_aligned(rp, 16);     // one ds_read_b128
    const float l = rp[-1], rt = rp[4];
    w[r][0] = l; w[r][1] = q4.x; w[r][2] = q4.y; w[r][3] = q4.z; w[r][4] = q4.w; w[r][5] = rt;
  };
  constexpr int first = (ZMASK & 1) ? 0 : 3;
  constexpr int last = (NZ == 3 && (ZMASK & 4)) ? 8 : (NZ == 3 ? 5 : 2);
  fetch(first);
#pragma unroll
  for (int r = first; r <= last; ++r) {
    if (r < last) fetch(r + 1);
    __builtin_amdgcn_sched_barrier(0);
    const float lf = bl ? 0.0f : w[r][0], rt = br ? 0.0f : w[r][5];
    const float v[6] = {lf, w[r][1], w[r][2], w[r][3], w[r][4], rt};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
        acc[r * 3 + dx] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], v[j + dx], acc[r * 3 + dx], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int DIM, bool C4, int THREADS>
__global__ void __launch_bounds__(THREADS) gconv_wrw_ring_kernel(GconvArgs a, const float* gy, float* gw, float* ws, int units_per_wg) {
  constexpr int NZ = DIM == 3 ? 3 : 1;
  constexpr int R = DIM == 3 ? 4 : 2;                   // x ring slots
  constexpr int NCH = C4 ? 4 : 16;                      // channels per block: 16 (MFMA engine) or the 4 of a C=4 group (vector ALU)
  extern __shared__ __align__(16) float lds[];
  const int grp = blockIdx.y;
  const int U = a.B * a.nH * a.nD;
  const int u_beg = blockIdx.x * units_per_wg, u_end = min(U, u_beg + units_per_wg);
  const int CiB = (a.Cin + NCH - 1) / NCH, CoB = (a.Cout + NCH - 1) / NCH;
  const size_t vol = (size_t)a.D * a.H * a.W;
  const int cs = a.plane, gs = a.gstride;               // channel strides inside a slot
  const int xslot = NCH * cs, gslot = NCH * gs;
  float* xr = lds + kSlack;                              // [R][16][cs]
  float* gr = xr + (size_t)R * xslot;                    // [2][16][gs]
  float* red = lds;                                      // reduction buffer aliases the rings (after the last phase)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const bool vec = ((a.W & 3) == 0) && ((((uintptr_t)a.x | (uintptr_t)gy) & 15) == 0);
  const int lds_floats = kSlack + R * xslot + 2 * gslot;

  for (int cob = 0; cob < CoB; ++cob) {
    for (int cib = 0; cib < CiB; ++cib) {
      const int cin_here = min(NCH, a.Cin - cib * NCH), cout_here = min(NCH, a.Cout - cob * NCH);
      floatx4 acc[NZ * 9];                               // MFMA engine: one 16x16 block per tap
      float acc4[C4 ? 4 : 1][C4 ? NZ * 9 : 1];           // vector-ALU engine: g_w[co 0..3][ci = wave][tap] of this lane's positions
      float gsum = 0.0f;                                 // this lane's share of sum_pos g_y[co = col]
      float bsum4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int t = 0; t < NZ * 9; ++t) acc[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int co = 0; co < (C4 ? 4 : 1); ++co)
#pragma unroll
        for (int t = 0; t < (C4 ? NZ * 9 : 1); ++t) acc4[co][t] = 0.0f;
      __syncthreads();                                   // previous block's reduction has read `red`
      for (int i = threadIdx.x; i < lds_floats; i += blockDim.x) lds[i] = 0.0f;   // channels beyond Cin / Cout stay zero
      __syncthreads();

      // x slice z of one (batch, row tile) -> ring slot; rows outside the image are zero-filled
      auto load_x = [&](int slot, int b, int z, int th0, int th) {
        if (z < 0 || z >= a.D) {                         // depth padding: a slice of zeros
          for (int c = wave; c < cin_here; c += nwaves) {
            float* dst = xr + (size_t)slot * xslot + (size_t)c * cs;
            for (int i = lane; i < (th + 2) * a.W; i += 64) dst[i] = 0.0f;
          }
          return;
        }
        const int r_lo = th0 == 0 ? 1 : 0, r_hi = min(th + 2, a.H - th0 + 1);
        for (int c = wave; c < cin_here; c += nwaves) {
          float* dst = xr + (size_t)slot * xslot + (size_t)c * cs;
          const float* src = a.x + (((size_t)b * a.groups + grp) * a.Cin + cib * NCH + c) * vol +
                             ((size_t)z * a.H + (th0 - 1 + r_lo)) * a.W;
          dma_run(dst + r_lo * a.W, src, (r_hi - r_lo) * a.W, lane, vec);
          if (r_lo) for (int i = lane; i < a.W; i += 64) dst[i] = 0.0f;
          for (int i = r_hi * a.W + lane; i < (th + 2) * a.W; i += 64) dst[i] = 0.0f;
        }
      };
      auto load_g = [&](int slot, int b, int z, int th0, int th) {
        const int npos = th * a.W;
        for (int c = wave; c < cout_here; c += nwaves) {
          float* dst = gr + (size_t)slot * gslot + (size_t)c * gs;
          const float* src = gy + (((size_t)b * a.groups + grp) * a.Cout + cob * NCH + c) * vol + ((size_t)z * a.H + th0) * a.W;
          dma_run(dst, src, npos, lane, vec);
          if (lane < ((16 - (npos & 15)) & 15)) dst[npos + lane] = 0.0f;   // padded K tail (groups of 16 positions)
        }
      };
      // One phase: all positions of a (th x W) slice against the 9 taps of each present x slice.
      // K runs over positions, 16 per group: lane (col, kq) owns the 4 CONSECUTIVE positions
      // pq = p0 + 4*kq .. +3 (one row: W % 4 == 0) and MFMA j of a group takes element j of every lane —
      // any assignment of positions to K slots is valid as long as A and B agree.  So one aligned
      // ds_read_b128 of g_y feeds the A operand of all 108 (36 in 2D) MFMAs of the group, and per tile
      // row one ds_read_b128 plus the two neighbours (ds_read_b32) feed the B operands of 12 MFMAs
      // (3 horizontal taps x 4 elements): 28 LDS reads per 108 MFMAs instead of one per MFMA — on
      // gfx950 a ds_read_b32 per MFMA, not the matrix pipe, sets the pace (tools/microbench/mfma_f32.hip).
      auto compute = [&](const int (&xs_slot)[NZ], int zmask, int g_slot, int th) {
        if constexpr (C4) {
          // vector-ALU engine (four-channel groups: a 16x16x4 MFMA would carry 4 useful rows and columns of 16).
          // Wave w owns input channel w; a lane takes quads of 4 consecutive x positions: 4 ds_read_b128 of g_y and,
          // per window row, one ds_read_b128 + two neighbours of x feed 48 FMAs.
          const int wq = a.W >> 2, nquads = th * wq, ci = wave;
          for (int q = lane; q < nquads; q += 64) {
            const int y = q / wq, x0 = (q - y * wq) * 4;
            const bool bl = x0 == 0, br = x0 + 4 == a.W;
            float g[4][4];
#pragma unroll
            for (int co = 0; co < 4; ++co) {
              const float4 t = *(const float4*)__builtin_assume_aligned(gr + (size_t)g_slot * gslot + (size_t)co * gs + y * a.W + x0, 16);
              g[co][0] = t.x; g[co][1] = t.y; g[co][2] = t.z; g[co][3] = t.w;
              if (ci == 0) bsum4[co] += (t.x + t.y) + (t.z + t.w);
            }
#pragma unroll
            for (int r = 0; r < NZ * 3; ++r) {
              const float* rp = xr + (size_t)xs_slot[r / 3] * xslot + (size_t)ci * cs + (y + r % 3) * a.W + x0;
              const float4 q4 = *(const float4*)__builtin_assume_aligned(rp, 16);
              const float lf = bl ? 0.0f : rp[-1], rt = br ? 0.0f : rp[4];
              const float v[6] = {lf, q4.x, q4.y, q4.z, q4.w, rt};
#pragma unroll
              for (int co = 0; co < 4; ++co)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                  for (int j = 0; j < 4; ++j) acc4[co][r * 3 + dx] = fmaf(g[co][j], v[j + dx], acc4[co][r * 3 + dx]);
            }
          }
          return;
        }
        const int npos = th * a.W;                       // multiple of 4
        const int ng = (npos + 15) >> 4;
        const int per = (ng + nwaves - 1) / nwaves;
        const int g_beg = min(ng, wave * per), g_end = min(ng, g_beg + per);
        const float* ga = gr + (size_t)g_slot * gslot + (size_t)col * gs + 4 * kq;
        const float* xb[NZ];
#pragma unroll
        for (int tz = 0; tz < NZ; ++tz) xb[tz] = xr + (size_t)xs_slot[tz] * xslot + (size_t)col * cs;
        int pq = g_beg * 16 + 4 * kq;
        {
          const int pc = min(pq, npos - 4);
          pq = pc;                                        // lanes past the end re-read the last quad (their g_y is 0)
        }
        int x0 = pq % a.W, y = pq / a.W;
        for (int gi = g_beg; gi < g_end; ++gi) {
          const bool bl = x0 == 0, br = x0 + 4 == a.W;
          const int o = y * a.W + x0;
          wrw_group<NZ, NZ == 3 ? 7 : 1>(acc, gsum, xb, ga + gi * 16, o, a.W, bl, br);
          x0 += 16;
          while (x0 >= a.W) { x0 -= a.W; ++y; }
          if (y >= th) { y = th - 1; x0 = a.W - 4; }      // padded tail: stay in bounds (g_y is 0 there)
        }
      };
      auto unit_of = [&](int u, int& b, int& th0, int& th, int& z0, int& z1) {
        const int zc = u % a.nD, yt = (u / a.nD) % a.nH;
        b = u / (a.nD * a.nH);
        th0 = yt * a.TH;
        th = min(a.TH, a.H - th0);
        z0 = zc * a.TZ;
        z1 = min(a.D, z0 + a.TZ);
      };

      if constexpr (DIM == 3) {
        for (int u = u_beg; u < u_end; ++u) {
          int b, th0, th, z0, z1;
          unit_of(u, b, th0, th, z0, z1);
          // entries e = 0 .. (z1-z0)+1 of this unit are the slices z0-1+e; entry e lives in slot e % 4
          load_x(0, b, z0 - 1, th0, th);
          load_x(1, b, z0, th0, th);
          load_x(2, b, z0 + 1, th0, th);
          load_g(0, b, z0, th0, th);
          dma_wait_all();
          __syncthreads();
          for (int j = 0; j < z1 - z0; ++j) {
            const int z = z0 + j;
            if (j + 1 < z1 - z0) {
              load_x((j + 3) & 3, b, z + 2, th0, th);
              load_g((j + 1) & 1, b, z + 1, th0, th);
            }
            const int slots[NZ] = {j & 3, (j + 1) & 3, (j + 2) & 3};
            compute(slots, 7, j & 1, th);
            dma_wait_all();
            __syncthreads();                             // phase done: its slots may be refilled, the prefetch has landed
          }
        }
      } else {
        int b, th0, th, z0, z1;
        if (u_beg < u_end) {
          unit_of(u_beg, b, th0, th, z0, z1);
          load_x(0, b, 0, th0, th);
          load_g(0, b, 0, th0, th);
        }
        dma_wait_all();
        __syncthreads();
        for (int u = u_beg; u < u_end; ++u) {
          const int s = (u - u_beg) & 1;
          if (u + 1 < u_end) {
            unit_of(u + 1, b, th0, th, z0, z1);
            load_x(s ^ 1, b, 0, th0, th);
            load_g(s ^ 1, b, 0, th0, th);
          }
          unit_of(u, b, th0, th, z0, z1);
          const int slots[NZ] = {s};
          compute(slots, 1, s, th);
          dma_wait_all();
          __syncthreads();
        }
      }
      if constexpr (C4) {
        // lanes -> one value per (co, tap) of this wave's input channel; slot = (chunk, group)
        constexpr int T = NZ * 9;
        float* wsp4 = ws + ((size_t)blockIdx.x * a.groups + grp) * (16 * T);
#pragma unroll
        for (int co = 0; co < 4; ++co)
#pragma unroll
          for (int t = 0; t < T; ++t) {
            float v = acc4[co][t];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) wsp4[(wave * 4 + co) * T + t] = v;
          }
        if (wave == 0) {
          float* wb = ws + (size_t)gridDim.x * a.groups * (16 * T) + ((size_t)blockIdx.x * a.groups + grp) * 4;
#pragma unroll
          for (int co = 0; co < 4; ++co) {
            float v = bsum4[co];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) wb[co] = v;
          }
        }
        continue;
      }
      // sum the wave partials through LDS, three taps per round; D: row (co) = kq*4 + r, column (ci) = col.
      // With a workspace the workgroup's partial sums are STORED (slot = chunk, group, block) and a second small
      // kernel adds the chunks in a fixed order; without one they go to g_w by float atomics (zeroed first).
      // Thousands of device-scope atomics per workgroup on a few KiB of g_w were the kernel's whole run time.
      float* wsp = ws ? ws + ((((size_t)blockIdx.x * a.groups + grp) * CoB + cob) * CiB + cib) * (NZ * 9 * 256) : nullptr;
#pragma unroll
      for (int t3 = 0; t3 < NZ * 3; ++t3) {
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(wave * 3 + tt) * 256 + (kq * 4 + r) * 16 + col] = acc[t3 * 3 + tt][r];
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) {
          float sum = 0.0f;
          for (int w2 = 0; w2 < nwaves; ++w2) sum += red[w2 * 3 * 256 + i];
          if (wsp) {
            wsp[t3 * 3 * 256 + i] = sum;
          } else {
            const int t = t3 * 3 + (i >> 8), co = cob * 16 + ((i >> 4) & 15), ci = cib * 16 + (i & 15);
            if (co < a.Cout && ci < a.Cin)
              atomicAdd(&gw[((size_t)(grp * a.Cout + co) * a.Cin + ci) * a.taps + t], sum);
          }
        }
      }
      // bias partial of this chunk: lanes (col, kq = 0..3) and the waves hold disjoint positions of channel col
      if (ws && cib == 0) {
        gsum += __shfl_xor(gsum, 16, 64);
        gsum += __shfl_xor(gsum, 32, 64);
        __syncthreads();
        if (lane < 16) red[wave * 16 + lane] = gsum;
        __syncthreads();
        if (threadIdx.x < 16) {
          float sum = 0.0f;
          for (int w2 = 0; w2 < nwaves; ++w2) sum += red[w2 * 16 + threadIdx.x];
          float* wb = ws + (size_t)gridDim.x * a.groups * CoB * CiB * (NZ * 9 * 256);       // after the filter partials
          wb[(((size_t)blockIdx.x * a.groups + grp) * CoB + cob) * 16 + threadIdx.x] = sum;
        }
      }
    }
  }
}

// sum_c src[c * stride + idx] over `chunks` partial slots: blockDim = (64, 4), lane y takes chunks y, y+4, ...; the four
// lane sums are added in lane order (fixed).  More loads in flight than one thread walking every chunk.
__device__ __forceinline__ float sum_chunks(const float* src, size_t stride, size_t idx, bool valid, int chunks, float (*red)[64]) {
  float s = 0.0f;
  if (valid)
    for (int c = threadIdx.y; c < chunks; c += 4) s += src[(size_t)c * stride + idx];
  red[threadIdx.y][threadIdx.x] = s;
  __syncthreads();
  const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  __syncthreads();
  return t;
}

// second stage of the ring kernel's reduction: g_w[grp, co, ci, t] = sum over chunks (ascending) of the stored partials
// ---------------------------------------------------------------------------
// Backward-weight for SMALL volumes with many channels (the pooled 8^3 .. 2^3 volumes and 16^2 .. 4^2 planes of the
// Res3D / Res2D stacks: <= 512 positions, 32-64 channels per group).  There the ring kernel's per-unit pipeline never
// fills (its tiles are the whole volume) and its workspace reduction moves more bytes than the op: 300-800 us for
// 0.2-14 GFLOP.  This form is a plain register-tiled correlation on the vector ALU: one 256-thread workgroup per
// (group, 16 output channels, 16 input channels), thread = one (co, ci) pair holding its 3^d filter taps; per batch
// element the zero-padded input block and the g_y block sit in LDS, a thread walks the rows with g_y's row and the
// three shifted input rows in registers (W + 2 + W reads per 3 W multiply-adds per (dz, dy)).  Lanes of a wave differ
// in ci (16) and co (4): every LDS read is a 16- or 4-address broadcast.  No atomics, no workspace: bitwise
// reproducible.  The bias gradient rides along (ci block 0, lanes with ci == 0).   grid = (ci_blocks * co_blocks, groups)
// ---------------------------------------------------------------------------
// kWsSplit thread groups of 256 (co, ci) pairs walk disjoint row ranges of the volume (depth slices in 3D, rows in 2D) and
// their partial filters meet in LDS: one workgroup per CU then runs 4 waves per SIMD instead of 1 — the walk is a chain of
// LDS reads and dependent multiply-adds (8^3 B2 1024->1024: 147 -> see DESIGN).
constexpr int kWsSplit = 4;

template <int DIM, int WT>
__global__ void __launch_bounds__(256 * kWsSplit) gconv_wrw_small_kernel(GconvArgs a, const float* __restrict__ gy, float* __restrict__ gw,
                                                                        float* __restrict__ gbias) {
  constexpr int NR = DIM == 3 ? 9 : 3;
  constexpr int NT = 256 * kWsSplit;
  extern __shared__ __align__(16) float lds[];
  const int D = a.D, H = a.H;
  const int Hp = H + 2, Wp = WT + 2, Dp = DIM == 3 ? D + 2 : 1;
  int plane = Dp * Hp * Wp;
  plane |= 1;                                            // odd stride between input channels: conflict-free broadcasts
  const int P = D * H * WT;
  float* xs = lds;                                       // [16][plane], zero halo
  float* gs = lds + 16 * plane;                          // [16][P]
  const int cib = (a.Cin + 15) / 16;
  const int cb = blockIdx.x % cib, ob = blockIdx.x / cib, grp = blockIdx.y;
  const int pair = threadIdx.x & 255, slice = threadIdx.x >> 8;
  const int ci = pair & 15, co = pair >> 4;
  const int ci_g = cb * 16 + ci, co_g = ob * 16 + co;
  const size_t vol = (size_t)P;
  // this slice's rows: (z, y) pairs in row-major order, rows = D * H
  const int rows = D * H;
  const int per = (rows + kWsSplit - 1) / kWsSplit;
  const int r_beg = min(rows, slice * per), r_end = min(rows, r_beg + per);
  float acc[NR * 3];
#pragma unroll
  for (int t = 0; t < NR * 3; ++t) acc[t] = 0.0f;
  float bsum = 0.0f;
  for (int i = threadIdx.x; i < 16 * plane; i += NT) xs[i] = 0.0f;      // the halo stays zero for every batch element
  for (int b = 0; b < a.B; ++b) {
    __syncthreads();
    const float* xg = a.x + ((size_t)b * a.groups + grp) * a.Cin * vol;
    const float* gg = gy + ((size_t)b * a.groups + grp) * a.Cout * vol;
    for (int i = threadIdx.x; i < 16 * P; i += NT) {
      const int c = i / P, p = i - c * P;
      const int x = p % WT, y = (p / WT) % H, z = p / (WT * H);
      const int cin = cb * 16 + c, cout = ob * 16 + c;
      xs[c * plane + ((DIM == 3 ? z + 1 : 0) * Hp + y + 1) * Wp + x + 1] = cin < a.Cin ? xg[(size_t)cin * vol + p] : 0.0f;
      gs[i] = cout < a.Cout ? gg[(size_t)cout * vol + p] : 0.0f;
    }
    __syncthreads();
    const float* xc = xs + ci * plane;
    const float* gc = gs + co * P;
    for (int zy = r_beg; zy < r_end; ++zy) {
      const int z = zy / H, y = zy - z * H;
      float g[WT];
#pragma unroll
      for (int x = 0; x < WT; ++x) g[x] = gc[zy * WT + x];
      if (cb == 0) {
#pragma unroll
        for (int x = 0; x < WT; ++x) bsum += g[x];
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int dz = DIM == 3 ? r / 3 : 0, dy = r % 3;
        const float* row = xc + ((z + dz) * Hp + y + dy) * Wp;       // padded coordinates: (z + dz - 1) + 1, ...
        float xr[WT + 2];
#pragma unroll
        for (int x = 0; x < WT + 2; ++x) xr[x] = row[x];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          float s = acc[r * 3 + dx];
#pragma unroll
          for (int x = 0; x < WT; ++x) s = __builtin_fmaf(g[x], xr[x + dx], s);
          acc[r * 3 + dx] = s;
        }
      }
    }
  }
  // partial filters of slices 1.. -> LDS [slice - 1][tap (+ bias)][pair], summed by slice 0 in slice order
  __syncthreads();
  float* red = lds;
  constexpr int NV = NR * 3 + 1;
  if (slice > 0) {
#pragma unroll
    for (int t = 0; t < NR * 3; ++t) red[((slice - 1) * NV + t) * 256 + pair] = acc[t];
    red[((slice - 1) * NV + NR * 3) * 256 + pair] = bsum;
  }
  __syncthreads();
  if (slice == 0) {
    for (int sl = 0; sl < kWsSplit - 1; ++sl) {
#pragma unroll
      for (int t = 0; t < NR * 3; ++t) acc[t] += red[(sl * NV + t) * 256 + pair];
      bsum += red[(sl * NV + NR * 3) * 256 + pair];
    }
    if (co_g < a.Cout && ci_g < a.Cin) {
      float* o = gw + ((size_t)(grp * a.Cout + co_g) * a.Cin + ci_g) * a.taps;
#pragma unroll
      for (int t = 0; t < NR * 3; ++t) o[t] = acc[t];
    }
    if (gbias != nullptr && cb == 0 && ci == 0 && co_g < a.Cout) gbias[grp * a.Cout + co_g] = bsum;
  }
}


// ---------------------------------------------------------------------------
// The same small volumes ON THE MATRIX CORES.  Per (group, 16 output channels, 16 input channels) the weight gradient is
// 3^d small GEMMs  g_w[co, ci, tap] = sum_{b, p} g_y[b, co, p] * x[b, ci, p + tap]  with K = (batch, positions): one
// v_mfma_f32_16x16x4_f32 contracts 4 consecutive x positions of a row,
//     A (16x4)  = g_y[co 0..15][p .. p+3]            one ds_read_b32, shared by all 3^d taps
//     B (4x16)  = x[ci 0..15][p + tap .. p + tap+3]  one ds_read_b32 from the zero-haloed tile (x halo too: no masks)
//     D (16x16) = the tap's 16x16 block, 4 accumulator registers per lane, 3^d blocks per wave (108 registers in 3D).
// One 512-thread workgroup per block; its 8 waves split K (k-step = wave, wave + 8, ...) and add their partial blocks in a
// fixed tree through LDS at the end.  Operands are staged per (batch element, depth slab of TZ slices) through TWO LDS
// buffers: the next stage's global loads are issued before this stage's MFMA run and written to the other buffer after it
// (one barrier per stage).  Channel strides are == 2 (mod 4) floats with odd halves: the 16 channels x 2 positions a
// half-wave reads fall on 32 different banks.  When the blocks do not cover the chip (32 -> 64 channels: 128 of them) the
// batch is split over `ksplit` workgroups whose partials go to the workspace of the ring kernel's reduction
// (gconv_wrw_reduce_kernel, fixed order).   grid = (ci_blocks * co_blocks * ksplit, groups)
// ---------------------------------------------------------------------------
constexpr int kWmThreads = 512;
constexpr int kWmWaves = kWmThreads / 64;
constexpr int kWmUnits = 4;                      // 16-byte staging units per thread and operand, at most

template <int DIM, int WT>
__global__ void __launch_bounds__(kWmThreads) gconv_wrw_mfma_kernel(GconvArgs a, const float* __restrict__ gy, float* __restrict__ gw,
                                                                     float* __restrict__ gbias, float* __restrict__ ws,
                                                                     int TZ, int XS, int GS, int ksplit) {
  constexpr int NR = DIM == 3 ? 9 : 3, TAPS = NR * 3, WQ = WT / 4, Wp = WT + 2;
  extern __shared__ __align__(16) float lds[];
  const int D = a.D, H = a.H, Hp = H + 2;
  const int ZS = DIM == 3 ? TZ + 2 : 1;
  const int nslab = DIM == 3 ? D / TZ : 1;
  const int cib = (a.Cin + 15) >> 4, cob = (a.Cout + 15) >> 4;
  const int cb = blockIdx.x % cib, ob = (blockIdx.x / cib) % cob, ks = blockIdx.x / (cib * cob), grp = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 15, kq = lane >> 4;
  const size_t vol = (size_t)D * H * WT;
  const int STAGE = 16 * XS + 16 * GS;
  const int bper = (a.B + ksplit - 1) / ksplit;
  const int b_lo = min(a.B, ks * bper), b_hi = min(a.B, b_lo + bper);
  const int NS = (b_hi - b_lo) * nslab;
  const int rows = (DIM == 3 ? TZ : 1) * H;               // rows of a slab
  const int n_xu = 16 * ZS * H * WQ, n_gu = 16 * rows * WQ;
  const bool want_bias = gbias != nullptr && cb == 0;

  // stage-invariant part of this thread's staging units
  int x_src[kWmUnits], x_dst[kWmUnits], x_zr[kWmUnits], g_src[kWmUnits], g_dst[kWmUnits];
#pragma unroll
  for (int u = 0; u < kWmUnits; ++u) {
    const int i = threadIdx.x + u * kWmThreads;
    {
      const int xq = i % WQ, y = (i / WQ) % H, zs = (i / (WQ * H)) % ZS, c = i / (WQ * H * ZS);
      const bool ok = i < n_xu && cb * 16 + c < a.Cin;
      x_zr[u] = ok ? (DIM == 3 ? zs - 1 : 0) : -(1 << 20);
      x_src[u] = c * (int)vol + ((DIM == 3 ? zs - 1 : 0) * H + y) * WT + xq * 4;
      x_dst[u] = i < n_xu ? c * XS + (zs * Hp + y + 1) * Wp + xq * 4 + 1 : -1;
    }
    {
      const int p4 = i % (rows * WQ), c = i / (rows * WQ);
      const bool ok = i < n_gu && ob * 16 + c < a.Cout;
      g_src[u] = ok ? c * (int)vol + p4 * 4 : -1;
      g_dst[u] = i < n_gu ? c * GS + p4 * 4 : -1;
    }
  }
  float4 sx[kWmUnits], sg[kWmUnits];
  auto load_stage = [&](int s) {
    const int b = b_lo + s / nslab, z0 = (s % nslab) * TZ;
    const float* xg = a.x + (((size_t)b * a.groups + grp) * a.Cin + cb * 16) * vol + (size_t)z0 * H * WT;
    const float* gg = gy + (((size_t)b * a.groups + grp) * a.Cout + ob * 16) * vol + (size_t)z0 * H * WT;
#pragma unroll
    for (int u = 0; u < kWmUnits; ++u) {
      const int z = z0 + x_zr[u];
      sx[u] = (z >= 0 && z < D) ? *(const float4*)__builtin_assume_aligned(xg + x_src[u], 16) : make_float4(0, 0, 0, 0);
      sg[u] = g_src[u] >= 0 ? *(const float4*)__builtin_assume_aligned(gg + g_src[u], 16) : make_float4(0, 0, 0, 0);
    }
  };
  auto store_stage = [&](int buf) {
    float* xs = lds + buf * STAGE;
    float* gs = xs + 16 * XS;
#pragma unroll
    for (int u = 0; u < kWmUnits; ++u) {
      if (x_dst[u] >= 0) {
        float* p = xs + x_dst[u];
        p[0] = sx[u].x; p[1] = sx[u].y; p[2] = sx[u].z; p[3] = sx[u].w;
      }
      if (g_dst[u] >= 0) {
        float2* p = (float2*)__builtin_assume_aligned(gs + g_dst[u], 8);
        p[0] = make_float2(sg[u].x, sg[u].y); p[1] = make_float2(sg[u].z, sg[u].w);
      }
    }
  };

  floatx4 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) acc[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
  float bsum = 0.0f;
  if (NS > 0) load_stage(0);
  for (int i = threadIdx.x; i < 2 * STAGE; i += kWmThreads) lds[i] = 0.0f;          // halos (and unused channels) stay zero
  __syncthreads();
  if (NS > 0) store_stage(0);
  __syncthreads();
  const int nks = rows * WQ;
  for (int s = 0; s < NS; ++s) {
    if (s + 1 < NS) load_stage(s + 1);
    const float* xs = lds + (s & 1) * STAGE + col * XS + kq;
    const float* gs = lds + (s & 1) * STAGE + 16 * XS + col * GS + kq;
    // software pipeline over this wave's units = (k-step, dz): the 9 operands of the next unit are requested before this
    // unit's 9 MFMAs (the compiler, left alone, reads 7 values, waits, multiplies: every group's LDS latency is exposed; and
    // whole k-steps of 27 would put more than the 15 LDS requests in flight that s_waitcnt can count).  The request past the
    // wave's last unit re-reads a valid one: no branch, so no register merge that would wait.
    constexpr int NZ = DIM == 3 ? 3 : 1;
    auto load_u = [&](int it, int dz, float& av, float (&bv)[9]) {
      const int k = wave + it * kWmWaves;
      const int xq = k % WQ, zy = k / WQ;
      const int z = DIM == 3 ? zy / H : 0, y = DIM == 3 ? zy - z * H : zy;
      av = gs[k * 4];
      const float* bp = xs + ((z + dz) * Hp + y) * Wp + xq * 4;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) bv[dy * 3 + dx] = bp[dy * Wp + dx];
    };
    float aA, aB, bA[9], bB[9];
    const int nit = (nks - wave + kWmWaves - 1) / kWmWaves;          // k-steps of this wave
    const int nU = nit * NZ;
    if (nU > 0) load_u(0, 0, aA, bA);
    for (int u0 = 0; u0 < nU; u0 += 2 * NZ) {
#pragma unroll
      for (int j = 0; j < 2 * NZ; ++j) {
        if (u0 + j >= nU) break;
        const int itn = min((u0 + j + 1) / NZ, nit - 1);
        if ((j & 1) == 0) load_u(itn, (j + 1) % NZ, aB, bB); else load_u(itn, (j + 1) % NZ, aA, bA);
        __builtin_amdgcn_sched_barrier(0);
        const float av = (j & 1) == 0 ? aA : aB;
        if (j % NZ == 0) bsum += av;
#pragma unroll
        for (int t = 0; t < 9; ++t)
          acc[(j % NZ) * 9 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, (j & 1) == 0 ? bA[t] : bB[t], acc[(j % NZ) * 9 + t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (s + 1 < NS) store_stage((s + 1) & 1);
    __syncthreads();
  }

  // partial blocks of the 8 waves: a fixed tree through LDS (upper half writes, lower half adds)
  bsum += __shfl_xor(bsum, 16);
  bsum += __shfl_xor(bsum, 32);
  constexpr int NV = TAPS * 4 + 1;
  for (int half = kWmWaves / 2; half >= 1; half >>= 1) {
    if (wave >= half && wave < 2 * half) {
      float* red = lds + (size_t)(wave - half) * NV * 64 + lane;
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(t * 4 + r) * 64] = acc[t][r];
      red[TAPS * 4 * 64] = bsum;
    }
    __syncthreads();
    if (wave < half) {
      const float* red = lds + (size_t)wave * NV * 64 + lane;
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] += red[(t * 4 + r) * 64];
      bsum += red[TAPS * 4 * 64];
    }
    __syncthreads();
  }
  if (wave != 0) return;
  if (ws != nullptr) {
    // [chunk][grp][cob][cib][t][co16][ci16], bias partials behind: [chunk][grp][cob][16]
    const size_t per_chunk = (size_t)a.groups * cob * cib * TAPS * 256;
    float* o = ws + (size_t)ks * per_chunk + ((size_t)(grp * cob + ob) * cib + cb) * TAPS * 256 + (kq * 4) * 16 + col;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[t * 256 + r * 16] = acc[t][r];
    if (cb == 0 && lane < 16) ws[(size_t)ksplit * per_chunk + ((size_t)ks * a.groups + grp) * cob * 16 + ob * 16 + lane] = bsum;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = ob * 16 + kq * 4 + r, ci = cb * 16 + col;
      if (co < a.Cout && ci < a.Cin) {
        float* o = gw + ((size_t)(grp * a.Cout + co) * a.Cin + ci) * TAPS;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) o[t] = acc[t][r];
      }
    }
    if (want_bias && lane < 16 && ob * 16 + lane < a.Cout) gbias[grp * a.Cout + ob * 16 + lane] = bsum;
  }
}

__global__ void __launch_bounds__(256) gconv_wrw_reduce_kernel(const float* ws, float* gw, float* gbias, int chunks, int groups, int Cin,
                                                               int Cout, int taps) {
  __shared__ float red[4][64];
  const int CiB = (Cin + 15) >> 4, CoB = (Cout + 15) >> 4;
  const int per_blk = taps * 256;
  const size_t per_chunk = (size_t)groups * CoB * CiB * per_blk;
  const size_t idx = (size_t)blockIdx.x * 64 + threadIdx.x;             // (grp, cob, cib, t, co16, ci16)
  const float sum = sum_chunks(ws, per_chunk, idx, idx < per_chunk, chunks, red);
  if (idx < per_chunk && threadIdx.y == 0) {
    const int i = (int)(idx % 256), t = (int)((idx / 256) % taps);
    const int blk = (int)(idx / per_blk);
    const int cib = blk % CiB, cob = (blk / CiB) % CoB, grp = blk / (CiB * CoB);
    const int co = cob * 16 + (i >> 4), ci = cib * 16 + (i & 15);
    if (co < Cout && ci < Cin) gw[((size_t)(grp * Cout + co) * Cin + ci) * taps + t] = sum;
  }
  // bias partials follow the filter partials: [chunk][grp][cob][16]
  const size_t nb = (size_t)groups * CoB * 16;
  if (gbias && (size_t)blockIdx.x * 64 < nb) {                         // workgroup-uniform
    const float bsum = sum_chunks(ws + (size_t)chunks * per_chunk, nb, idx, idx < nb, chunks, red);
    if (idx < nb && threadIdx.y == 0) {
      const int co = (int)((idx / 16) % CoB) * 16 + (int)(idx % 16), grp = (int)(idx / (16 * CoB));
      if (co < Cout) gbias[grp * Cout + co] = bsum;
    }
  }
}

// ---------------------------------------------------------------------------
// Weight gradient of four-channel 3D groups ON THE MATRIX CORES.
//   g_w[co][ci][dz][dy][dx] = sum_p g_y[co][p] x[ci][p + (dz-1)HW + (dy-1)W + (dx-1)]
//                           = sum_q g_y[co][q - (dx-1)] x[ci][q + (dz-1)HW + (dy-1)W]           (q = p + dx - 1)
// so with the column shift on g_y and the depth / row shifts on x a 16x16x4 MFMA over 4 consecutive positions q computes
//     D[(dz, ci)][(dx, co)] += sum_k x[ci][slice z + dz-1][row + dy-1][q_k] * g_y[co][q_k - (dx-1)]
// for one row tap dy: nine taps x 16 channel pairs per MFMA (12 of 16 rows and columns in use), three MFMAs (dy) per four
// positions, the g_y operand shared by the three.  The lanes of an A operand differ in (dz, ci) — three ring slots with their
// own offsets — and kq, never in the row, so no two of them meet on a bank beyond the 2-per-bank minimum of 64 lanes (the first
// mapping had dy in the B operand's lanes: rows of exactly 32 floats put all of them on the same 8 banks, LDS 50 % busy at 63 %
// conflicts).  g_y shifted out of its row is masked (one AND per k-step); x rows / slices outside the volume are zeros in LDS.
// MFMA row 12 (dz group 3, ci 0) of the dy = 1 operand is the constant 1: D[12][(dx = 1, co)] = sum of g_y, the bias gradient.
// A WAVE owns R rows of one (batch, group) volume and walks a depth segment on its own: its x slices (rows +- 1) and g_y slices
// stream through wave-private LDS rings by LDS-DMA one step ahead — no workgroup barrier in the loop (a wave waits for its own
// pieces with s_waitcnt vmcnt(0): nothing else of it is in flight).  The waves of a workgroup are the row blocks of one (batch,
// group, depth segment); they add their partials through LDS in a fixed tree and write one slot of the ring kernel's
// workspace ([chunk][group][ci][co * 27 + tap], bias partials behind), summed by gconv_c4_wrw_reduce_kernel in a fixed order.
// grid = (depth segments, groups, B), block = ceil(H / R) waves
// ---------------------------------------------------------------------------
constexpr int kC4wRing = 4;                  // x slices in the ring (z-1, z, z+1 live + one in flight)

__global__ void __launch_bounds__(1024) gconv_c4_wrw_mfma3_kernel(GconvArgs a, const float* __restrict__ gy, float* __restrict__ ws,
                                                                  int R, int LZ, int SX, int SG, int chunks) {
  extern __shared__ __align__(16) float lds[];
  const int W = a.W, H = a.H, D = a.D, WQ = W >> 2;
  const int zs = blockIdx.x, grp = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
  const int col = lane & 15, kq = lane >> 4;
  const int y0 = wave * R, rows = min(R, H - y0);                    // this wave's rows
  const int l0 = zs * LZ, l1 = min(D, l0 + LZ);
  const size_t vol = (size_t)D * H * W;
  const float* xg = a.x + ((size_t)b * a.groups + grp) * 4 * vol;
  const float* gg = gy + ((size_t)b * a.groups + grp) * 4 * vol;
  // wave-private rings: x [kC4wRing][4 ci][SX] (rows y0 - 1 .. y0 + rows), g_y [2][4 co][SG] (rows y0 .. y0 + rows - 1)
  const int per_wave = kC4wRing * 4 * SX + 2 * 4 * SG + 2 * kSlack + 16;     // (+ 16: what a unit past the end of a slice reads)
  float* xr = lds + (size_t)wave * per_wave + kSlack;
  float* gr = xr + kC4wRing * 4 * SX + kSlack;
  for (int i = threadIdx.x; i < nwaves * per_wave; i += blockDim.x) lds[i] = 0.0f;      // x rows / slices outside the volume stay zero
  __syncthreads();

  floatx4 acc[3];                                                    // per row tap dy
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) acc[dy] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
  if (rows > 0) {
    // A operand: MFMA row i = col -> (dz = col >> 2, ci = col & 3), k = kq:  x[ci][slice z + dz - 1][row + dy - 1][x0 + kq]
    // B operand: column j = col -> (dx = col >> 2, co = col & 3), k = kq:    g_y[co][slice z][row][x0 + kq - (dx - 1)]
    const int grp4 = col >> 2, ch = col & 3;
    const unsigned m_first = (kq == 0 && grp4 == 2) ? 0u : ~0u;      // first quad of a row: q - 1 is outside (dx = 2)
    const unsigned m_last = (kq == 3 && grp4 == 0) ? 0u : ~0u;       // last quad: q + 1 is outside (dx = 0)
    const unsigned m_keep = col == 12 ? 0u : ~0u;                     // MFMA row 12 of the dy = 1 operand ...
    const unsigned m_one = col == 12 ? 0x3f800000u : 0u;             // ... is the constant 1
    const int b_off = ch * SG + kq - grp4 + 1;
    const int gy_lo = max(y0 - 1, 0), gy_hi = min(y0 + rows + 1, H);
    const unsigned xr_lds = lds_addr(xr), gr_lds = lds_addr(gr);
    const int nx16 = (gy_hi - gy_lo) * W / 4, ng16 = rows * W / 4;    // 16-byte units per channel
    auto load_x = [&](int z) {                                       // slice z, rows y0 - 1 .. y0 + rows -> ring slot z mod 4
      const int slot = z & (kC4wRing - 1);
      if (z < 0 || z >= D) {                                         // outside the volume: the slot must read as zeros
        for (int i = lane; i < 4 * SX; i += 64) xr[slot * 4 * SX + i] = 0.0f;
        return;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float* src = xg + (size_t)c * vol + ((size_t)z * H + gy_lo) * W;
        for (int p0 = 0; p0 < nx16; p0 += 64) {
          const unsigned dst = __builtin_amdgcn_readfirstlane(xr_lds + (unsigned)(((slot * 4 + c) * SX + (gy_lo - y0 + 1) * W + p0 * 4) * 4));
          if (p0 + lane < nx16) glds16(src + (size_t)(p0 + lane) * 4, dst);
        }
      }
    };
    auto load_g = [&](int z) {
      const int slot = z & 1;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float* src = gg + (size_t)c * vol + ((size_t)z * H + y0) * W;
        for (int p0 = 0; p0 < ng16; p0 += 64) {
          const unsigned dst = __builtin_amdgcn_readfirstlane(gr_lds + (unsigned)(((slot * 4 + c) * SG + p0 * 4) * 4));
          if (p0 + lane < ng16) glds16(src + (size_t)(p0 + lane) * 4, dst);
        }
      }
    };
    const int nunits = (rows * WQ) >> 1;                             // units of two k-steps = two neighbouring quads of a row
    auto step = [&](int z) {
      // this lane's slice: z + dz - 1 in slot (z + dz - 1) mod 4 (dz group 3 reads slot z + 2: any data, its rows are unused)
      const float* xs = xr + (((z + grp4 - 1) & (kC4wRing - 1)) * 4 + ch) * SX + kq;
      const float* gs = gr + (z & 1) * 4 * SG + b_off;
      // Software pipeline by hand: the next unit's 8 operands are requested before this unit's 6 MFMAs (left to itself hipcc
      // reads an operand pair, waits, multiplies), and stay raw until they are multiplied (an AND right behind a load would
      // wait for it on the spot).  Rows are contiguous in both rings: a unit's operands sit at 8 * unit floats, only the g_y
      // masks know about rows (W / 4 is even: a unit's first quad may start a row, its second may end one).  A unit past the
      // end reads the slack behind the slice with its g_y operands zeroed: no branch in the stream.
      int lu = 0, lxq = 0;
      auto load_unit = [&](float (&av)[2][3], float (&bv)[2], unsigned (&mk)[2]) {
        const int o = lu * 8;
        mk[0] = lxq == 0 ? m_first : ~0u;
        mk[1] = lxq + 2 == WQ ? m_last : ~0u;
        bv[0] = gs[o];
        bv[1] = gs[o + 4];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) { av[0][dy] = xs[o + dy * W]; av[1][dy] = xs[o + 4 + dy * W]; }
        ++lu;
        lxq = lxq + 2 == WQ ? 0 : lxq + 2;
      };
      auto mfma_unit = [&](const float (&av)[2][3], const float (&bv)[2], const unsigned (&mk)[2], unsigned live) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float bq = __uint_as_float(__float_as_uint(bv[u]) & (mk[u] & live));
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][0], bq, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float((__float_as_uint(av[u][1]) & m_keep) | m_one), bq, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][2], bq, acc[2], 0, 0, 0);
        }
      };
      float aA[2][3], aB[2][3], bA[2], bB[2];
      unsigned mA[2], mB[2];
      load_unit(aA, bA, mA);
      for (int u0 = 0; u0 < nunits; u0 += 2) {
        load_unit(aB, bB, mB);
        __builtin_amdgcn_sched_barrier(0);
        mfma_unit(aA, bA, mA, ~0u);
        __builtin_amdgcn_sched_barrier(0);
        load_unit(aA, bA, mA);
        __builtin_amdgcn_sched_barrier(0);
        mfma_unit(aB, bB, mB, u0 + 1 < nunits ? ~0u : 0u);           // (a unit past the end multiplies zeros)
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    load_x(l0 - 1); load_x(l0); load_x(l0 + 1); load_g(l0);
    for (int z = l0; z < l1; ++z) {
      dma_wait_all();                                                // everything this wave requested has landed (slices <= z + 1, g_y z)
      if (z + 2 <= l1) load_x(z + 2);                                // slot (z + 2) mod 4 held slice z - 2: lands while this step multiplies
      if (z + 1 < l1) load_g(z + 1);
      step(z);
    }
  }
  // partial filters of the waves: a fixed tree through LDS (upper half writes, lower half adds)
  __syncthreads();
  constexpr int NV = 12;
  int half = 1;
  while (half < nwaves) half <<= 1;
  for (half >>= 1; half >= 1; half >>= 1) {
    if (wave >= half && wave < 2 * half) {
      float* red = lds + (size_t)(wave - half) * NV * 64 + lane;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(dy * 4 + r) * 64] = acc[dy][r];
    }
    __syncthreads();
    if (wave < half && wave + half < nwaves) {
      const float* red = lds + (size_t)wave * NV * 64 + lane;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[dy][r] += red[(dy * 4 + r) * 64];
    }
    __syncthreads();
  }
  if (wave != 0) return;
  // lane (dz = kq, dx = col >> 2, co = col & 3), register r = ci: g_w[co][ci][dz][dy][dx]
  const int chunk = blockIdx.z * gridDim.x + blockIdx.x;
  float* o = ws + ((size_t)chunk * a.groups + grp) * 16 * 27;
  const int dx = col >> 2, co = col & 3, dz = kq;
  if (dz < 3 && dx < 3) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r * (4 * 27) + co * 27 + dz * 9 + dy * 3 + dx] = acc[dy][r];
  }
  // bias gradient: D row 12 = lanes 48..63, register 0, of the dy = 1 accumulator; columns (dx = 1, co)
  if (kq == 3 && dx == 1) ws[(size_t)chunks * a.groups * 16 * 27 + ((size_t)chunk * a.groups + grp) * 4 + co] = acc[1][0];
}

// second stage of the four-channel ring kernel's reduction.  Workspace: [chunk][group][ci][co*taps + t], then [chunk][group][co] (bias).
__global__ void __launch_bounds__(256) gconv_c4_wrw_reduce_kernel(const float* ws, float* gw, float* gbias, int chunks, int groups, int taps) {
  __shared__ float red[4][64];
  const int per = 16 * taps;                                  // per (chunk, group): [ci][co][t]
  const int idx = blockIdx.x * 64 + threadIdx.x;
  const float sum = sum_chunks(ws, (size_t)groups * per, idx, idx < groups * per, chunks, red);
  if (idx < groups * per && threadIdx.y == 0) {
    const int grp = idx / per, r = idx % per;
    const int ci = r / (4 * taps), co = (r / taps) % 4, t = r % taps;
    gw[((size_t)(grp * 4 + co) * 4 + ci) * taps + t] = sum;
  }
  if (gbias && blockIdx.x * 64 < groups * 4) {                // workgroup-uniform
    const float bsum = sum_chunks(ws + (size_t)chunks * groups * per, (size_t)groups * 4, idx, idx < groups * 4, chunks, red);
    if (idx < groups * 4 && threadIdx.y == 0) gbias[idx] = bsum;
  }
}

// bias gradient: g_bias[c] = sum over batch and positions of g_y (one wave per channel row chunk)
__global__ void gconv_bias_grad_kernel(const float* gy, float* gbias, int B, int C, size_t vol) {
  const int c = blockIdx.x;
  float s = 0.0f;
  for (int b = 0; b < B; ++b) {
    const float* p = gy + ((size_t)b * C + c) * vol;
    for (size_t i = threadIdx.x; i < vol; i += blockDim.x) s += p[i];
  }
  __shared__ float red[4];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) gbias[c] = red[0] + red[1] + red[2] + red[3];
}

bool plan_tiles_min_halo(GconvArgs& a, int dim, size_t extra_per_pos_bytes, size_t fixed_bytes, int cin_planes,
                         int plane_mod, size_t budget) {
  // LDS per workgroup = cin_planes*plane*4 + fixed + extra_per_pos*TD*TH*W  <= budget.
  // Among the (TD, TH) that fit, take the one that stages the fewest halo floats per output position
  // ((TD+2)(TH+2) / (TD*TH), counted on the tiles that actually cover the tensor); ties -> the larger tile.
  a.Ws = a.W;                                     // rows keep the tensor width (no x halo)
  const int dzh = dim == 3 ? 2 : 0;
  double best = 1e30;
  bool found = false;
  for (int TD = dim == 3 ? a.D : 1; TD >= 1; --TD) {
    for (int TH = a.H; TH >= 1; --TH) {
      const int Hs = TH + 2;
      int plane = (TD + dzh) * Hs * a.Ws;
      plane += (plane_mod - (plane & 31) + 32) & 31;   // plane == plane_mod (mod 32): conflict-free operand reads
      const size_t bytes = (size_t)cin_planes * plane * 4 + fixed_bytes + extra_per_pos_bytes * TD * TH * a.W + 64;
      if (bytes > budget) continue;
      const int nD = (a.D + TD - 1) / TD, nH = (a.H + TH - 1) / TH;
      const double staged = (double)nD * nH * (TD + dzh) * Hs;       // rows staged for the whole tensor
      const double cost = staged / ((double)a.D * a.H) - 1e-9 * TD * TH;
      if (cost < best) {
        best = cost;
        found = true;
        a.TD = TD; a.TH = TH; a.Hs = Hs; a.plane = plane; a.nD = nD; a.nH = nH;
      }
      break;                                       // smaller TH at this TD only stages more
    }
  }
  return found;
}

// depth-first policy: the full-height tile of the deepest depth that fits, else one slice with as many rows as fit
// (stages more halo than plan_tiles_min_halo, but measured faster for the MFMA forward kernel's position-group split)
bool plan_tiles_budget(GconvArgs& a, int dim, size_t extra_per_pos_bytes, size_t fixed_bytes, int cin_planes,
                       int plane_mod, size_t budget) {
  a.Ws = a.W;
  const int dzh = dim == 3 ? 2 : 0;
  for (int TD = dim == 3 ? a.D : 1; TD >= 1; --TD) {
    for (int TH = a.H; TH >= 1; --TH) {
      const int Hs = TH + 2;
      int plane = (TD + dzh) * Hs * a.Ws;
      plane += (plane_mod - (plane & 31) + 32) & 31;
      const size_t bytes = (size_t)cin_planes * plane * 4 + fixed_bytes + extra_per_pos_bytes * TD * TH * a.W + 64;
      if (bytes <= budget) {
        a.TD = TD; a.TH = TH; a.Hs = Hs; a.plane = plane;
        a.nD = (a.D + TD - 1) / TD; a.nH = (a.H + TH - 1) / TH;
        return true;
      }
      if (dim == 3 && TD > 1) break;
    }
  }
  return false;
}

bool plan_tiles(GconvArgs& a, int dim, size_t extra_per_pos_bytes, size_t fixed_bytes, int cin_planes, int plane_mod,
                size_t budget = kLdsBudget) {
  return plan_tiles_budget(a, dim, extra_per_pos_bytes, fixed_bytes, cin_planes, plane_mod, budget) ||
         plan_tiles_budget(a, dim, extra_per_pos_bytes, fixed_bytes, cin_planes, plane_mod, kLdsBudgetMax);
}

int gconv_common(GconvArgs& a, int B, int groups, int Cin, int Cout, int dim, const int* W) {
  if (B <= 0 || groups <= 0 || Cin <= 0 || Cout <= 0 || (dim != 2 && dim != 3) || !W) return CT_EINVAL;
  for (int j = 0; j < dim; ++j) if (W[j] < 1 || W[j] > 4096) return CT_EINVAL;
  if (B > 65535 || groups > 65535) return CT_EINVAL;
  a.B = B; a.groups = groups; a.Cin = Cin; a.Cout = Cout;
  a.D = dim == 3 ? W[0] : 1; a.H = dim == 3 ? W[1] : W[0]; a.W = dim == 3 ? W[2] : W[1];
  a.taps = dim == 3 ? 27 : 9;
  a.KB = (Cin + 3) / 4;
  a.msplit = 1;
  return CT_OK;
}

template <typename K>
int set_lds_attr(K kernel, size_t bytes) {
  if (bytes > 48 * 1024 &&
      hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return CT_ELAUNCH;
  return CT_OK;
}

// Few tiles x groups x batches (the 8^3 / 4^3 / 2^3 volumes of the Res3D stacks at batch 2: 32 workgroups for 256 CUs):
// several workgroups share a tile and split its 16-row blocks of output channels between them (each stages the small
// halo tile itself) until one workgroup per CU exists.
int pick_msplit(const GconvArgs& a) {
  const long long wgs = (long long)a.nD * a.nH * a.groups * a.B;
  const int MT = (a.Cout + 15) / 16;
  int m = 1;
  while (m < MT && wgs * m < 256) ++m;
  return m;
}

// four-channel groups in 3D with rows of 16-position blocks: the matrix-core kernel (see gconv_c4_mfma3_kernel)
int c4_mfma3_rows(const GconvArgs& a) {
  const int XB = a.W >> 4;
  int TH = (kC4mCols * (kC4mThreads / 64)) / XB;              // columns of a workgroup: rows x 16-position blocks
  return TH > a.H ? a.H : TH;
}

bool c4_mfma3_fits(const GconvArgs& a) {
  if ((a.W & 15) != 0 || a.W > 16 * kC4mCols * (kC4mThreads / 64)) return false;
  const int TH = c4_mfma3_rows(a);
  if (TH < 1) return false;
  // the kernel's static bounds: LDS-DMA pieces per wave (2) and 16-byte stores per thread (2) and slice
  const int ppc = ((TH + 2) * a.W / 4 + 63) / 64;
  return 4 * ppc <= 2 * (kC4mThreads / 64) && TH * a.W <= 2 * kC4mThreads;
}

int launch_c4_mfma3(GconvArgs a, hipStream_t st) {
  const int TH = c4_mfma3_rows(a);
  const int nH = (a.H + TH - 1) / TH;
  // depth segments until the chip is covered (each walks two extra slices)
  int nZ = 1;
  while ((long long)a.B * a.groups * nH * nZ < CT_C4M_WGS && a.D / (nZ * 2) >= 4) nZ *= 2;
  const int LZ = (a.D + nZ - 1) / nZ;
  nZ = (a.D + LZ - 1) / LZ;
  int CSX = (TH + 2) * a.W;
  CSX += (16 - (CSX & 31) + 32) & 31;                         // == 16 (mod 32): the four channels' 16-float runs fall on all 32 banks twice
  const size_t lds = ((size_t)3 * 4 * CSX + (size_t)2 * 4 * TH * a.W + 2 * kSlack) * 4;
  dim3 grid(nH * nZ, a.groups, a.B);
  CT_CLEAR_ERROR();
  if (a.transposed) {
    if (set_lds_attr(gconv_c4_mfma3_kernel<true>, lds) != CT_OK) return CT_ELAUNCH;
    hipLaunchKernelGGL(gconv_c4_mfma3_kernel<true>, grid, dim3(kC4mThreads), lds, st, a, TH, LZ, nZ, CSX);
  } else {
    if (set_lds_attr(gconv_c4_mfma3_kernel<false>, lds) != CT_OK) return CT_ELAUNCH;
    hipLaunchKernelGGL(gconv_c4_mfma3_kernel<false>, grid, dim3(kC4mThreads), lds, st, a, TH, LZ, nZ, CSX);
  }
  CT_CHECK_LAUNCH();
  return CT_OK;
}

// four-channel groups with 16-byte rows: the vector-ALU kernel
int launch_c4(GconvArgs a, int dim, hipStream_t st) {
  if (!plan_tiles_min_halo(a, dim, 0, 0, 4, 4, kLdsBudget) && !plan_tiles_min_halo(a, dim, 0, 0, 4, 4, kLdsBudgetMax)) return CT_EINVAL;
  const size_t lds = (size_t)4 * a.plane * 4 + 2 * kSlack * 4;
  dim3 grid(a.nD * a.nH, a.groups, a.B);
  CT_CLEAR_ERROR();
#define CT_C4_LAUNCH(DIMV, TR)                                                              \
  do {                                                                                      \
    if (set_lds_attr(gconv_c4_kernel<DIMV, TR>, lds) != CT_OK) return CT_ELAUNCH;           \
    hipLaunchKernelGGL((gconv_c4_kernel<DIMV, TR>), grid, dim3(kThreads), lds, st, a);      \
  } while (0)
  if (dim == 2) { if (a.transposed) CT_C4_LAUNCH(2, true); else CT_C4_LAUNCH(2, false); }
  else { if (a.transposed) CT_C4_LAUNCH(3, true); else CT_C4_LAUNCH(3, false); }
#undef CT_C4_LAUNCH
  CT_CHECK_LAUNCH();
  return CT_OK;
}

// quad form of the MFMA kernel (rows of W % 4 == 0 floats, 16-byte aligned tensors)
int launch_fwd4(GconvArgs a, int dim, hipStream_t st) {
  const int NR = dim == 3 ? 9 : 3;
  const size_t wbytes = (size_t)NR * a.KB * 256 * 4;
  // Tile budget: this kernel likes BIG tiles (one 1024-thread workgroup per CU: long MFMA runs per barrier, little
  // halo) as long as one workgroup per CU exists — take the largest budget that still yields 256 of them, else the most
  // workgroups (measured on the zoo shapes: 2D 32^2 H64 47 -> 37 us, 64^2 63 -> 43 us vs the 32 KiB tiles).
  const size_t budgets[] = {(size_t)kLdsBudgetMax, (size_t)96 * 1024, (size_t)64 * 1024, (size_t)kLdsBudget};
  bool ok = false;
  GconvArgs best = a;
  long long best_wgs = -1;
  for (size_t budget : budgets) {
    GconvArgs t = a;
    const bool fits = a.W < 16 ? plan_tiles_min_halo(t, dim, 0, wbytes, a.KB * 4, 16, budget)
                               : plan_tiles_budget(t, dim, 0, wbytes, a.KB * 4, 16, budget);
    if (!fits) continue;
    const long long wgs = (long long)t.nD * t.nH * a.groups * a.B;
    if (wgs >= 256) { best = t; ok = true; break; }
    if (wgs > best_wgs) { best = t; best_wgs = wgs; ok = true; }
  }
  if (!ok) return CT_EINVAL;
  a = best;
  const size_t lds = (size_t)a.KB * 4 * a.plane * 4 + wbytes + 2 * kSlack * 4;
  a.msplit = pick_msplit(a);
  dim3 grid(a.nD * a.nH * a.msplit, a.groups, a.B);
  const int threads = lds > 48 * 1024 ? kThreadsBig : kThreads;
  CT_CLEAR_ERROR();
  if (dim == 2) {
    if (set_lds_attr(gconv_fwd4_kernel<2>, lds) != CT_OK) return CT_ELAUNCH;
    hipLaunchKernelGGL(gconv_fwd4_kernel<2>, grid, dim3(threads), lds, st, a);
  } else {
    if (set_lds_attr(gconv_fwd4_kernel<3>, lds) != CT_OK) return CT_ELAUNCH;
    hipLaunchKernelGGL(gconv_fwd4_kernel<3>, grid, dim3(threads), lds, st, a);
  }
  CT_CHECK_LAUNCH();
  return CT_OK;
}


// K-split form for wide groups (> 32 input channels per group): see gconv_fwd4k_kernel
int launch_fwd4k(GconvArgs a, int dim, hipStream_t st) {
  const int NR = dim == 3 ? 9 : 3;
  const int MT = (a.Cout + 15) / 16;
  constexpr int kWaves = kThreadsBig / 64;
  // Pass 0: the smallest split of the 16-row output blocks over workgroups (msplit) for which ONE tile covers the whole
  // volume next to the bank slice (the pooled 8^3 .. 2^3 volumes: no halo re-staging at all); pass 1: any tiling, the
  // thinnest bank first.  Items = (span of 16 quads, 16-row block) pairs a wave keeps in registers: <= 4 per wave.
  for (int pass = 0; pass < 2; ++pass)
  for (int step = 0; step < MT; ++step) {
    const int msplit = pass == 0 ? step + 1 : MT - step;
    const int nmt = (MT + msplit - 1) / msplit;
    const int CS = dim == 3 ? (a.transposed ? BankRow<27, true>::CS : BankRow<27, false>::CS)
                            : (a.transposed ? BankRow<9, true>::CS : BankRow<9, false>::CS);
    const size_t wbytes = (size_t)nmt * 16 * CS * 4;
    GconvArgs t = a;
    if (!plan_tiles_min_halo(t, dim, 0, wbytes, kKBlk, 16, kLdsBudgetMax)) continue;
    if (pass == 0 && t.nD * t.nH != 1) continue;
    const int nquads = t.TD * t.TH * (a.W >> 2);
    const int items = ((nquads + 15) / 16) * nmt;
    const long long wgs = (long long)t.nD * t.nH * a.groups * a.B * msplit;
    if (items > 4 * kWaves) continue;                        // thinner blocks per workgroup
    if (pass == 0 && wgs < 256 && msplit < MT && items > kWaves / 2) continue;   // cover the chip while the waves still have work
    t.msplit = msplit;
    const size_t lds = (size_t)kKBlk * t.plane * 4 + wbytes + 2 * kSlack * 4;
    dim3 grid(t.nD * t.nH * msplit, a.groups, a.B);
    const int maxi = items <= kWaves ? 1 : (items <= 2 * kWaves ? 2 : 4);
    // whole 16-channel rows at 16-byte aligned addresses move by LDS-DMA
    const int dma = (a.Cin % 16 == 0 && a.Cout % 16 == 0 && (((uintptr_t)a.w) & 15) == 0) ? 1 : 0;
    CT_CLEAR_ERROR();
#define CT_F4K_LAUNCH(DIMV, MAXIV, TRV)                                                                \
    do {                                                                                               \
      if (set_lds_attr(gconv_fwd4k_kernel<DIMV, MAXIV, TRV>, lds) != CT_OK) return CT_ELAUNCH;         \
      hipLaunchKernelGGL((gconv_fwd4k_kernel<DIMV, MAXIV, TRV>), grid, dim3(kThreadsBig), lds, st, t, dma); \
    } while (0)
#define CT_F4K_MAXI(DIMV, TRV) \
    do { if (maxi == 1) CT_F4K_LAUNCH(DIMV, 1, TRV); else if (maxi == 2) CT_F4K_LAUNCH(DIMV, 2, TRV); else CT_F4K_LAUNCH(DIMV, 4, TRV); } while (0)
    if (dim == 2) { if (a.transposed) CT_F4K_MAXI(2, true); else CT_F4K_MAXI(2, false); }
    else { if (a.transposed) CT_F4K_MAXI(3, true); else CT_F4K_MAXI(3, false); }
#undef CT_F4K_MAXI
#undef CT_F4K_LAUNCH
    CT_CHECK_LAUNCH();
    return CT_OK;
  }
  return CT_EINVAL;
}

// 2^d volumes: the dense small-volume form
size_t tiny_lds(const GconvArgs& a, int dim, int cob) {
  const int P = dim == 3 ? 8 : 4, TS = dim == 3 ? kTinyTapStride : 12;
  return ((size_t)cob * (a.Cin * TS + 4) + (size_t)a.B * (a.Cin * P + 4) + (size_t)4 * cob * a.B * P) * 4;
}

int tiny_cob(const GconvArgs& a, int dim) {
  if (a.W != 2 || a.H != 2 || (dim == 3 && a.D != 2)) return 0;
  if ((((uintptr_t)a.x) | ((uintptr_t)a.y)) & 15) return 0;
  int cob = 64 / a.B;                                   // one lane per (output channel, cloud)
  if (cob < 4) cob = 4;
  if (cob > a.Cout) cob = a.Cout;
  while (cob > 1 && tiny_lds(a, dim, cob) > (size_t)84 * 1024) cob >>= 1;      // two workgroups per CU when it can
  return tiny_lds(a, dim, cob) <= (size_t)kLdsBudgetMax ? cob : 0;
}

int launch_tiny(GconvArgs a, int dim, int cob, hipStream_t st) {
  const size_t lds = tiny_lds(a, dim, cob);
  dim3 grid((a.Cout + cob - 1) / cob, a.groups);
  CT_CLEAR_ERROR();
  if (dim == 2) {
    if (set_lds_attr(gconv_tiny_kernel<2>, lds) != CT_OK) return CT_ELAUNCH;
    hipLaunchKernelGGL(gconv_tiny_kernel<2>, grid, dim3(256), lds, st, a, cob);
  } else {
    if (set_lds_attr(gconv_tiny_kernel<3>, lds) != CT_OK) return CT_ELAUNCH;
    hipLaunchKernelGGL(gconv_tiny_kernel<3>, grid, dim3(256), lds, st, a, cob);
  }
  CT_CHECK_LAUNCH();
  return CT_OK;
}

int launch_fwd(GconvArgs a, int dim, hipStream_t st) {
  if (const int cob = tiny_cob(a, dim)) return launch_tiny(a, dim, cob, st);
  const bool rows16 = (a.W & 3) == 0 && ((((uintptr_t)a.x) | ((uintptr_t)a.y)) & 15) == 0;
  // large four-channel volumes (the zoo's 32^3 C4 head at B8: 74 vs 93 us) take the matrix-core kernel; below ~2 M positions
  // (B2: 31 vs 30 us, 16^3: 29 vs 16) its per-slice barriers cost more than the vector ALU's extra multiply-adds
  // (debug bit 1 = never, bit 2 = always)
  if (dim == 3 && rows16 && a.Cin == 4 && a.Cout == 4 && c4_mfma3_fits(a)) {
    const unsigned dbg = t_gconv_debug.load(std::memory_order_relaxed);
    const long long positions = (long long)a.B * a.groups * a.D * a.H * a.W;
    if (!(dbg & 2) && ((dbg & 4) || positions >= (2ll << 20))) return launch_c4_mfma3(a, st);
  }
  if (rows16 && a.Cin == 4 && a.Cout == 4) return launch_c4(a, dim, st);
  const unsigned dbg_cin = (t_gconv_debug.load(std::memory_order_relaxed) >> 8) & 0xffu;
  if (rows16 && a.Cin >= (dbg_cin ? (int)dbg_cin : 32)) {      // wide groups (>= 32 input channels): contraction in blocks of 16
    const int r = launch_fwd4k(a, dim, st);
    if (r != CT_EINVAL) return r;
  }
  if (rows16) {
    // the quad form keeps the whole filter bank of a 16-row block in LDS ([rows][KB][4][16][4] floats): with 64 input
    // channels per group in 3D that alone is 147 KiB — such shapes take the K-split form above
    const int r = launch_fwd4(a, dim, st);
    if (r != CT_EINVAL) return r;
  }
  const size_t wbytes = (size_t)a.taps * a.KB * 64 * 4;
  // short rows (W < 16: the 8^3 volumes) do better on the minimum-halo tiles, the others on the depth-first ones (measured)
  const bool ok = a.W < 16 ? (plan_tiles_min_halo(a, dim, 0, wbytes, a.KB * 4, 16, kLdsBudget) ||
                              plan_tiles_min_halo(a, dim, 0, wbytes, a.KB * 4, 16, kLdsBudgetMax))
                           : plan_tiles(a, dim, 0, wbytes, a.KB * 4, 16);
  if (!ok) return CT_EINVAL;
  const size_t lds = (size_t)a.KB * 4 * a.plane * 4 + wbytes + 2 * kSlack * 4;
  a.msplit = pick_msplit(a);
  dim3 grid(a.nD * a.nH * a.msplit, a.groups, a.B);
  const int threads = lds > 48 * 1024 ? kThreadsBig : kThreads;
  CT_CLEAR_ERROR();
  if (dim == 2) {
    if (set_lds_attr(gconv_fwd_kernel<2>, lds) != CT_OK) return CT_ELAUNCH;
    hipLaunchKernelGGL(gconv_fwd_kernel<2>, grid, dim3(threads), lds, st, a);
  } else {
    if (set_lds_attr(gconv_fwd_kernel<3>, lds) != CT_OK) return CT_ELAUNCH;
    hipLaunchKernelGGL(gconv_fwd_kernel<3>, grid, dim3(threads), lds, st, a);
  }
  CT_CHECK_LAUNCH();
  return CT_OK;
}

// backward-weight, ring kernel (rows of W % 4 == 0 floats): the plan ...
struct WrwRingPlan { size_t lds; int chunks, units_per_wg; };

bool plan_wrw_ring(GconvArgs& a, int dim, WrwRingPlan& p, int nch = 16) {
  const int B = a.B, groups = a.groups;
  const int threads = nch == 16 ? kWrwThreads : kThreads;
  // the two engines were tuned separately: MFMA one big workgroup per CU, vector-ALU two of them and twice the workgroups
  const size_t budget0 = nch == 16 ? (size_t)kLdsBudgetWrw : (size_t)(80 * 1024 - 512);
  const int want_wgs = nch == 16 ? CT_WRW_WANT : 512;
  // the tallest row tile whose rings fit the budget (64 KiB: two workgroups per CU, then whatever one CU
  // holds), then depth chunks until ~512 workgroups exist
  const int R = dim == 3 ? 4 : 2;
  const size_t red_bytes = (size_t)(threads / 64) * 3 * 256 * 4;
  auto pad4 = [](int n) { return n + ((4 - (n & 63) + 64) & 63); };      // == 4 (mod 64): conflict-free ds_read_b128 operand reads
  auto ring_bytes = [&](int TH) {
    const size_t b = ((size_t)R * nch * pad4((TH + 2) * a.W) + (size_t)2 * nch * pad4((TH * a.W + 15) & ~15) + 2 * kSlack) * 4;
    return b > red_bytes ? b : red_bytes;
  };
  int TH = 0;
  for (size_t budget : {budget0, (size_t)kLdsBudgetMax}) {
    for (int t = a.H; t >= 1 && !TH; --t)
      if (ring_bytes(t) <= budget) TH = t;
    if (TH) break;
  }
  if (!TH) return false;
  a.TH = TH; a.nH = (a.H + TH - 1) / TH;
  a.plane = pad4((TH + 2) * a.W); a.gstride = pad4((TH * a.W + 15) & ~15);
  a.Hs = TH + 2; a.Ws = a.W; a.TD = 1;
  a.nD = 1; a.TZ = a.D;
  if (dim == 3) {
    const int want = (want_wgs + groups - 1) / groups;       // units wanted per group
    int nD = (want + B * a.nH - 1) / (B * a.nH);
    if (nD > a.D / 2) nD = a.D / 2 > 0 ? a.D / 2 : 1;    // at least two slices per chunk: each loads two extra x slices
    if (nD < 1) nD = 1;
    a.TZ = (a.D + nD - 1) / nD;
    a.nD = (a.D + a.TZ - 1) / a.TZ;
  }
  p.lds = ring_bytes(TH);
  const int U = B * a.nH * a.nD;
  int chunks = (want_wgs + groups - 1) / groups;
  if (chunks > U) chunks = U;
  if (chunks < 1) chunks = 1;
  p.units_per_wg = (U + chunks - 1) / chunks;
  p.chunks = (U + p.units_per_wg - 1) / p.units_per_wg;
  return true;
}

size_t wrw_ring_workspace(const GconvArgs& a, const WrwRingPlan& p) {
  const size_t CiB = (a.Cin + 15) >> 4, CoB = (a.Cout + 15) >> 4;
  return ((size_t)p.chunks * a.groups * CoB * CiB * a.taps * 256 + (size_t)p.chunks * a.groups * CoB * 16) * sizeof(float);
}

// ... and the launch.  `ws` (>= wrw_ring_workspace bytes) selects the two-stage reduction; NULL the atomics.
int launch_wrw_ring(GconvArgs a, int dim, const float* g_y, float* g_w, float* g_bias, float* ws, size_t ws_bytes, hipStream_t st) {
  WrwRingPlan p;
  if (!plan_wrw_ring(a, dim, p)) return CT_EINVAL;
  if (ws && ws_bytes < wrw_ring_workspace(a, p)) return CT_EWORKSPACE;
  if (!ws && hipMemsetAsync(g_w, 0, (size_t)a.groups * a.Cout * a.Cin * a.taps * 4, st) != hipSuccess) return CT_ELAUNCH;
  dim3 grid(p.chunks, a.groups);
  if (dim == 2) {
    if (set_lds_attr(gconv_wrw_ring_kernel<2, false, kWrwThreads>, p.lds) != CT_OK) return CT_ELAUNCH;
    hipLaunchKernelGGL((gconv_wrw_ring_kernel<2, false, kWrwThreads>), grid, dim3(kWrwThreads), p.lds, st, a, g_y, g_w, ws, p.units_per_wg);
  } else {
    if (set_lds_attr(gconv_wrw_ring_kernel<3, false, kWrwThreads>, p.lds) != CT_OK) return CT_ELAUNCH;
    hipLaunchKernelGGL((gconv_wrw_ring_kernel<3, false, kWrwThreads>), grid, dim3(kWrwThreads), p.lds, st, a, g_y, g_w, ws, p.units_per_wg);
  }
  if (ws) {
    const size_t CiB = (a.Cin + 15) >> 4, CoB = (a.Cout + 15) >> 4;
    const size_t n = (size_t)a.groups * CoB * CiB * a.taps * 256;
    hipLaunchKernelGGL(gconv_wrw_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64, 4), 0, st, ws, g_w, g_bias, p.chunks,
                       a.groups, a.Cin, a.Cout, a.taps);
  }
  return CT_OK;
}

// backward-weight of four-channel groups: the ring kernel with its vector-ALU engine
bool c4_wrw_eligible(const GconvArgs& a) { return a.Cin == 4 && a.Cout == 4 && (a.W & 3) == 0; }

size_t c4_wrw_workspace(const GconvArgs& a, const WrwRingPlan& p) {
  return ((size_t)p.chunks * a.groups * 16 * a.taps + (size_t)p.chunks * a.groups * 4) * sizeof(float);
}

int launch_c4_wrw(GconvArgs a, int dim, const float* g_y, float* g_w, float* g_bias, float* ws, size_t ws_bytes, hipStream_t st) {
  WrwRingPlan p;
  if (!plan_wrw_ring(a, dim, p, 4)) return CT_EINVAL;
  if (ws_bytes < c4_wrw_workspace(a, p)) return CT_EWORKSPACE;
  dim3 grid(p.chunks, a.groups);
  if (dim == 2) {
    if (set_lds_attr(gconv_wrw_ring_kernel<2, true, kThreads>, p.lds) != CT_OK) return CT_ELAUNCH;
    hipLaunchKernelGGL((gconv_wrw_ring_kernel<2, true, kThreads>), grid, dim3(kThreads), p.lds, st, a, g_y, g_w, ws, p.units_per_wg);
  } else {
    if (set_lds_attr(gconv_wrw_ring_kernel<3, true, kThreads>, p.lds) != CT_OK) return CT_ELAUNCH;
    hipLaunchKernelGGL((gconv_wrw_ring_kernel<3, true, kThreads>), grid, dim3(kThreads), p.lds, st, a, g_y, g_w, ws, p.units_per_wg);
  }
  const int n = a.groups * 16 * a.taps;
  hipLaunchKernelGGL(gconv_c4_wrw_reduce_kernel, dim3((n + 63) / 64), dim3(64, 4), 0, st, ws, g_w, g_bias, p.chunks, a.groups, a.taps);
  return CT_OK;
}

// four-channel 3D groups: the matrix-core weight gradient (see gconv_c4_wrw_mfma3_kernel)
struct C4WrwPlan { int R, nwaves, nZ, LZ, SX, SG, chunks; size_t lds; };

bool plan_c4_wrw_mfma3(const GconvArgs& a, int dim, C4WrwPlan& p) {
  if (dim != 3 || a.Cin != 4 || a.Cout != 4 || (a.W & 7) != 0) return false;      // (rows of an even number of quads)
  p.R = 4;                                                    // rows per wave (2: same time at 32^3, twice the LDS-DMA pieces)
  p.nwaves = (a.H + p.R - 1) / p.R;
  if (p.nwaves > 16) return false;
  p.nZ = 1;
  while ((long long)a.B * a.groups * p.nZ < 256 && a.D / (p.nZ * 2) >= 4) p.nZ *= 2;
  p.LZ = (a.D + p.nZ - 1) / p.nZ;
  p.nZ = (a.D + p.LZ - 1) / p.LZ;
  p.SX = (((p.R + 2) * a.W + 31) & ~31) + 4;                  // == 4 (mod 32): (ci, kq) on 16 banks, the ring slots (4 SX apart) on the other 16
  p.SG = ((p.R * a.W + 31) & ~31) + 8;                        // == 8 (mod 32): the four channels' 7-float g_y windows on 28 banks
  const size_t per_wave = (size_t)kC4wRing * 4 * p.SX + (size_t)2 * 4 * p.SG + 2 * kSlack + 16;
  p.lds = (size_t)p.nwaves * per_wave * 4;
  p.chunks = a.B * p.nZ;
  return p.lds <= (size_t)kLdsBudgetMax;
}

size_t c4_wrw_mfma3_workspace(const GconvArgs& a, const C4WrwPlan& p) {
  return ((size_t)p.chunks * a.groups * 16 * a.taps + (size_t)p.chunks * a.groups * 4) * sizeof(float);
}

int launch_c4_wrw_mfma3(GconvArgs a, const C4WrwPlan& p, const float* g_y, float* g_w, float* g_bias, float* ws, size_t ws_bytes, hipStream_t st) {
  if (ws_bytes < c4_wrw_mfma3_workspace(a, p)) return CT_EWORKSPACE;
  dim3 grid(p.nZ, a.groups, a.B);
  CT_CLEAR_ERROR();
  if (set_lds_attr(gconv_c4_wrw_mfma3_kernel, p.lds) != CT_OK) return CT_ELAUNCH;
  hipLaunchKernelGGL(gconv_c4_wrw_mfma3_kernel, grid, dim3(p.nwaves * 64), p.lds, st, a, g_y, ws, p.R, p.LZ, p.SX, p.SG, p.chunks);
  CT_CHECK_LAUNCH();
  const int n = a.groups * 16 * a.taps;
  hipLaunchKernelGGL(gconv_c4_wrw_reduce_kernel, dim3((n + 63) / 64), dim3(64, 4), 0, st, ws, g_w, g_bias, p.chunks, a.groups, a.taps);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

// backward-weight, tile kernel (any row length): plan + launch
int launch_wrw_tiles(GconvArgs a, int dim, const float* g_y, float* g_w, hipStream_t st) {
  const int B = a.B, groups = a.groups;
  if (hipMemsetAsync(g_w, 0, (size_t)groups * a.Cout * a.Cin * a.taps * 4, st) != hipSuccess) return CT_ELAUNCH;
  // LDS: 16 input planes with halo + 16 rows of g_y + the cross-wave reduction buffer
  const size_t red_bytes = (size_t)(kThreads / 64) * 3 * 256 * 4;
  if (!plan_tiles(a, dim, (size_t)16 * 4, 1024 + red_bytes, 16, /*plane == 4 (mod 32): 16-byte aligned for the DMA*/ 4,
                  kLdsBudgetWrw)) return CT_EINVAL;
  const int gstride_max = ((a.TD * a.TH * a.W + 3) & ~3) | 1;
  const size_t lds = ((size_t)16 * a.plane + (size_t)16 * gstride_max) * 4 + red_bytes + 2 * kSlack * 4;
  const int U = B * a.nD * a.nH;
  int chunks = (512 + groups - 1) / groups;           // ~512 workgroups in flight
  if (chunks > U) chunks = U;
  if (chunks < 1) chunks = 1;
  const int units_per_wg = (U + chunks - 1) / chunks;
  chunks = (U + units_per_wg - 1) / units_per_wg;
  dim3 grid(chunks, groups);
  if (dim == 2) {
    if (set_lds_attr(gconv_bwd_weight_kernel<2>, lds) != CT_OK) return CT_ELAUNCH;
    hipLaunchKernelGGL(gconv_bwd_weight_kernel<2>, grid, dim3(kThreads), lds, st, a, g_y, g_w, units_per_wg);
  } else {
    if (set_lds_attr(gconv_bwd_weight_kernel<3>, lds) != CT_OK) return CT_ELAUNCH;
    hipLaunchKernelGGL(gconv_bwd_weight_kernel<3>, grid, dim3(kThreads), lds, st, a, g_y, g_w, units_per_wg);
  }
  return CT_OK;
}

// plan-only checks behind ct_gconv_supported
// LDS of the small-volume weight gradient: the two operand tiles, or the partial filters of the row slices if larger
size_t wrw_small_lds(const GconvArgs& a, int dim) {
  const int P = a.D * a.H * a.W;
  const int plane = ((dim == 3 ? a.D + 2 : 1) * (a.H + 2) * (a.W + 2)) | 1;
  const size_t tiles = (size_t)(16 * plane + 16 * P) * 4;
  const size_t red = (size_t)(kWsSplit - 1) * (a.taps + 1) * 256 * 4;
  return tiles > red ? tiles : red;
}

// small volumes, many channels: the register-tiled vector-ALU form (see gconv_wrw_small_kernel)
bool wrw_small_eligible(const GconvArgs& a, int dim) {
  const int P = a.D * a.H * a.W;
  if (!(a.W == 2 || a.W == 4 || a.W == 8 || a.W == 16)) return false;
  // (the walk over the batch is serial per workgroup: beyond a few thousand positions x batch the MFMA ring kernel wins)
  if (P > 512 || (a.Cin < 32 && a.Cout < 32)) return false;
  const long long wgs = (long long)((a.Cin + 15) / 16) * ((a.Cout + 15) / 16) * a.groups;
  if ((long long)P * a.B > (wgs >= 256 ? 4096 : 2048)) return false;      // 8^3 B8 64->64 (256 workgroups): 327 vs 403 us ring;
                                                                          // 32->64 (128 workgroups): 326 vs 211 us
  return wrw_small_lds(a, dim) <= (size_t)kLdsBudgetMax;
}

int launch_wrw_small(GconvArgs a, int dim, const float* g_y, float* g_w, float* g_bias, hipStream_t st) {
  const size_t lds = wrw_small_lds(a, dim);
  dim3 grid(((a.Cin + 15) / 16) * ((a.Cout + 15) / 16), a.groups);
  CT_CLEAR_ERROR();
#define CT_WS_LAUNCH(DIMV, WTV)                                                                    \
  do {                                                                                             \
    if (set_lds_attr(gconv_wrw_small_kernel<DIMV, WTV>, lds) != CT_OK) return CT_ELAUNCH;          \
    hipLaunchKernelGGL((gconv_wrw_small_kernel<DIMV, WTV>), grid, dim3(256 * kWsSplit), lds, st, a, g_y, g_w, g_bias); \
  } while (0)
  if (dim == 2) {
    if (a.W == 2) CT_WS_LAUNCH(2, 2); else if (a.W == 4) CT_WS_LAUNCH(2, 4); else if (a.W == 8) CT_WS_LAUNCH(2, 8); else CT_WS_LAUNCH(2, 16);
  } else {
    if (a.W == 2) CT_WS_LAUNCH(3, 2); else if (a.W == 4) CT_WS_LAUNCH(3, 4); else if (a.W == 8) CT_WS_LAUNCH(3, 8); else CT_WS_LAUNCH(3, 16);
  }
#undef CT_WS_LAUNCH
  CT_CHECK_LAUNCH();
  return CT_OK;
}

// small volumes, many channels, on the matrix cores (see gconv_wrw_mfma_kernel)
struct WrwMfmaPlan { int TZ, XS, GS, ksplit; size_t lds; };

bool plan_wrw_mfma(const GconvArgs& a, int dim, WrwMfmaPlan& p) {
  if (!(a.W == 4 || a.W == 8 || a.W == 16)) return false;
  const int P = a.D * a.H * a.W;
  if (P > 512 || P < 16 || (a.Cin < 32 && a.Cout < 32)) return false;
  const int taps = dim == 3 ? 27 : 9;
  const size_t red = (size_t)(kWmWaves / 2) * (taps * 4 + 1) * 64 * 4;
  const int WQ = a.W / 4;
  auto pad2 = [](int n) { return ((n + 3) & ~3) + 2; };          // == 2 (mod 4): see the kernel's header
  p.TZ = 0;
  for (int tz = dim == 3 ? a.D : 1; tz >= 1; --tz) {
    if (dim == 3 && a.D % tz != 0) continue;
    const int ZS = dim == 3 ? tz + 2 : 1;
    const int XS = pad2(ZS * (a.H + 2) * (a.W + 2)), GS = pad2(tz * a.H * a.W);
    const size_t lds = (size_t)2 * 16 * (XS + GS) * 4;
    if (lds > (size_t)kLdsBudgetMax) continue;
    if (16 * ZS * a.H * WQ > kWmUnits * kWmThreads || 16 * tz * a.H * WQ > kWmUnits * kWmThreads) continue;
    p.TZ = tz; p.XS = XS; p.GS = GS; p.lds = lds > red ? lds : red;
    break;
  }
  if (!p.TZ) return false;
  const long long wgs = (long long)((a.Cin + 15) / 16) * ((a.Cout + 15) / 16) * a.groups;
  int ksplit = 1;
  while (wgs * ksplit < 256 && ksplit * 2 <= a.B) ksplit *= 2;
  p.ksplit = ksplit;
  return true;
}

size_t wrw_mfma_workspace(const GconvArgs& a, const WrwMfmaPlan& p) {
  if (p.ksplit <= 1) return 0;
  const size_t CiB = (a.Cin + 15) >> 4, CoB = (a.Cout + 15) >> 4;
  return ((size_t)p.ksplit * a.groups * CoB * CiB * a.taps * 256 + (size_t)p.ksplit * a.groups * CoB * 16) * sizeof(float);
}

int launch_wrw_mfma(GconvArgs a, int dim, WrwMfmaPlan p, const float* g_y, float* g_w, float* g_bias, float* ws, size_t ws_bytes,
                    hipStream_t st) {
  if (p.ksplit > 1 && (!ws || ws_bytes < wrw_mfma_workspace(a, p))) p.ksplit = 1;      // no workspace: one workgroup per block
  float* wsp = p.ksplit > 1 ? ws : nullptr;
  const int CiB = (a.Cin + 15) / 16, CoB = (a.Cout + 15) / 16;
  dim3 grid(CiB * CoB * p.ksplit, a.groups);
  CT_CLEAR_ERROR();
#define CT_WM_LAUNCH(DIMV, WTV)                                                                     \
  do {                                                                                              \
    if (set_lds_attr(gconv_wrw_mfma_kernel<DIMV, WTV>, p.lds) != CT_OK) return CT_ELAUNCH;          \
    hipLaunchKernelGGL((gconv_wrw_mfma_kernel<DIMV, WTV>), grid, dim3(kWmThreads), p.lds, st, a, g_y, g_w, g_bias, wsp, \
                       p.TZ, p.XS, p.GS, p.ksplit);                                                 \
  } while (0)
  if (dim == 2) { if (a.W == 4) CT_WM_LAUNCH(2, 4); else if (a.W == 8) CT_WM_LAUNCH(2, 8); else CT_WM_LAUNCH(2, 16); }
  else { if (a.W == 4) CT_WM_LAUNCH(3, 4); else if (a.W == 8) CT_WM_LAUNCH(3, 8); else CT_WM_LAUNCH(3, 16); }
#undef CT_WM_LAUNCH
  CT_CHECK_LAUNCH();
  if (wsp) {
    const size_t n = (size_t)a.groups * CoB * CiB * a.taps * 256;
    hipLaunchKernelGGL(gconv_wrw_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64, 4), 0, st, wsp, g_w, g_bias, p.ksplit,
                       a.groups, a.Cin, a.Cout, a.taps);
    CT_CHECK_LAUNCH();
  }
  return CT_OK;
}

bool fwd_plan_ok(GconvArgs a, int dim) {
  if (tiny_cob(a, dim)) return true;
  if (a.Cin == 4 && a.Cout == 4 && (a.W & 3) == 0 &&
      (plan_tiles_min_halo(a, dim, 0, 0, 4, 4, kLdsBudget) || plan_tiles_min_halo(a, dim, 0, 0, 4, 4, kLdsBudgetMax))) return true;
  const size_t wbytes = (size_t)a.taps * a.KB * 64 * 4;          // the one-position form: what the quad form falls back to
  return plan_tiles(a, dim, 0, wbytes, a.KB * 4, 16);
}

bool wrw_plan_ok(GconvArgs a, int dim) {
  WrwMfmaPlan pm;
  if (plan_wrw_mfma(a, dim, pm) || wrw_small_eligible(a, dim)) return true;
  WrwRingPlan p;
  if ((a.W & 3) == 0 && plan_wrw_ring(a, dim, p, c4_wrw_eligible(a) ? 4 : 16)) return true;
  const size_t red_bytes = (size_t)(kThreads / 64) * 3 * 256 * 4;
  return plan_tiles(a, dim, (size_t)16 * 4, 1024 + red_bytes, 16, 4, kLdsBudgetWrw);
}

}  // namespace

extern "C" {

void ct_debug_set_gconv(unsigned flags) { t_gconv_debug.store(flags, std::memory_order_relaxed); }

int ct_gconv_fwd(const float* x, const float* w, const float* bias, float* y,
                 int B, int groups, int Cin, int Cout, int dim, const int* W, ct_stream_t s) {
  if (!x || !w || !y) return CT_EINVAL;
  GconvArgs a = {};
  int r = gconv_common(a, B, groups, Cin, Cout, dim, W);
  if (r != CT_OK) return r;
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.transposed = 0;
  return launch_fwd(a, dim, (hipStream_t)s);
}

int ct_gconv_bwd_data(const float* g_y, const float* w, float* g_x,
                      int B, int groups, int Cin, int Cout, int dim, const int* W, ct_stream_t s) {
  if (!g_y || !w || !g_x) return CT_EINVAL;
  GconvArgs a = {};
  // a convolution of g_y (Cout channels) producing Cin channels, with the bank read transposed + flipped
  int r = gconv_common(a, B, groups, Cout, Cin, dim, W);
  if (r != CT_OK) return r;
  a.x = g_y; a.w = w; a.bias = nullptr; a.y = g_x; a.transposed = 1;
  return launch_fwd(a, dim, (hipStream_t)s);
}

int ct_gconv_supported(int B, int groups, int Cin, int Cout, int dim, const int* W) {
  GconvArgs a = {};
  if (gconv_common(a, B, groups, Cin, Cout, dim, W) != CT_OK) return 0;
  GconvArgs t = {};
  gconv_common(t, B, groups, Cout, Cin, dim, W);                  // backward-data: the same pass with the channel roles swapped
  return fwd_plan_ok(a, dim) && fwd_plan_ok(t, dim) && wrw_plan_ok(a, dim) ? 1 : 0;
}

size_t ct_gconv_bwd_weight_workspace_bytes(int B, int groups, int Cin, int Cout, int dim, const int* W) {
  GconvArgs a = {};
  if (gconv_common(a, B, groups, Cin, Cout, dim, W) != CT_OK || (a.W & 3) != 0) return 0;
  if (c4_wrw_eligible(a)) {
    WrwRingPlan p4;
    C4WrwPlan p4m;
    size_t need4 = plan_wrw_ring(a, dim, p4, 4) ? c4_wrw_workspace(a, p4) : 0;
    if (plan_c4_wrw_mfma3(a, dim, p4m) && c4_wrw_mfma3_workspace(a, p4m) > need4) need4 = c4_wrw_mfma3_workspace(a, p4m);
    if (need4) return need4;
  }
  WrwRingPlan p;
  size_t need = plan_wrw_ring(a, dim, p) ? wrw_ring_workspace(a, p) : 0;
  WrwMfmaPlan pm;
  if (plan_wrw_mfma(a, dim, pm) && wrw_mfma_workspace(a, pm) > need) need = wrw_mfma_workspace(a, pm);
  C4WrwPlan p4m;
  if (plan_c4_wrw_mfma3(a, dim, p4m) && c4_wrw_mfma3_workspace(a, p4m) > need) need = c4_wrw_mfma3_workspace(a, p4m);
  return need;
}

int ct_gconv_bwd_weight(const float* x, const float* g_y, float* g_w, float* g_bias, void* workspace, size_t workspace_bytes,
                        int B, int groups, int Cin, int Cout, int dim, const int* W, ct_stream_t s) {
  if (!x || !g_y || !g_w) return CT_EINVAL;
  GconvArgs a = {};
  int r = gconv_common(a, B, groups, Cin, Cout, dim, W);
  if (r != CT_OK) return r;
  a.x = x; a.transposed = 0;
  hipStream_t st = (hipStream_t)s;
  CT_CLEAR_ERROR();
  const bool aligned = ((((uintptr_t)x) | ((uintptr_t)g_y)) & 15) == 0;
  bool ring_bias = (a.W & 3) == 0 && workspace != nullptr;        // the ring kernels' workspace path produces g_bias itself
  WrwMfmaPlan pm;
  if (aligned && !(t_gconv_debug.load(std::memory_order_relaxed) & 1) && plan_wrw_mfma(a, dim, pm))
    return launch_wrw_mfma(a, dim, pm, g_y, g_w, g_bias, (float*)workspace, workspace_bytes, st);
  if (wrw_small_eligible(a, dim)) {
    r = launch_wrw_small(a, dim, g_y, g_w, g_bias, st);
    if (r != CT_OK) return r;
    CT_CHECK_LAUNCH();
    return CT_OK;
  }
  C4WrwPlan p4m;
  // four-channel 3D groups: the matrix-core kernel wherever its plan fits (161 -> 90 us at 32^3 B8, 67 -> 30 at B2, 34 -> 18 on a
  // 5 x 7 x 16 volume; debug bit 1 = never)
  if (workspace && aligned && plan_c4_wrw_mfma3(a, dim, p4m) && !(t_gconv_debug.load(std::memory_order_relaxed) & 2) &&
      workspace_bytes >= c4_wrw_mfma3_workspace(a, p4m))
    return launch_c4_wrw_mfma3(a, p4m, g_y, g_w, g_bias, (float*)workspace, workspace_bytes, st);
  if (workspace && aligned && c4_wrw_eligible(a)) {
    r = launch_c4_wrw(a, dim, g_y, g_w, g_bias, (float*)workspace, workspace_bytes, st);
  } else {
    r = (a.W & 3) == 0 ? launch_wrw_ring(a, dim, g_y, g_w, g_bias, (float*)workspace, workspace_bytes, st) : CT_EINVAL;
    if (r == CT_EINVAL) {                                          // rows off the 16-byte grid, or rings that do not fit LDS
      ring_bias = false;
      r = launch_wrw_tiles(a, dim, g_y, g_w, st);
    }
  }
  if (r != CT_OK) return r;
  if (g_bias && !ring_bias) {
    const size_t vol = (size_t)a.D * a.H * a.W;
    hipLaunchKernelGGL(gconv_bias_grad_kernel, dim3(groups * Cout), dim3(256), 0, st, g_y, g_bias, B, groups * Cout, vol);
  }
  CT_CHECK_LAUNCH();
  return CT_OK;
}

}  // extern "C"
