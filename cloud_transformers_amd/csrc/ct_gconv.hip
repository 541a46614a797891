// Grouped 3x3 / 3x3x3 convolution over the rasterised planes and volumes, on the
// gfx950 matrix cores (exact-fp32 MFMA, v_mfma_f32_16x16x4_f32).
//
// The MHCT blocks convolve every head's grid with its own small filter bank
// (`groups = heads`, C_in = C_out in {4,16,32,64} per group; reference call sites
// layers/multihead_ct.py:50-65, unet2d/unet_parts.py:13-16, layers/v2v_groups.py:26-29).
// Per group this is an implicit GEMM  D[co, pos] = sum_{tap, ci} W[co, ci, tap] * X[ci, pos+tap]
// with a tiny N (=C_out) and K (=C_in*3^d): too small for a library GEMM to tile well,
// a perfect fit for one 16x16x4 MFMA per (tap, 4 input channels, 16 output positions):
//     A (16x4)  = W[co 0..15][ci kb*4..+3][tap]        one f32 / lane, read from LDS
//     B (4x16)  = X[ci kb*4..+3][16 positions + tap]   one f32 / lane, read from LDS (halo tile)
//     D (16x16) = 4 accumulator registers / lane       (row = co, col = position)
// A workgroup owns an output tile (TD x TH x W positions) of one (batch, group): the input
// tile with its halo is staged once in LDS (zero-padded borders), each wave keeps P position
// groups in flight so that one weight read feeds P MFMAs.
// Backward-data is the same kernel with the filter bank read transposed and flipped.
// Backward-weight is a second implicit GEMM (K = positions) accumulated across workgroups.
#include "ct_common.h"

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;                // small tiles: several workgroups per CU
constexpr int kThreadsBig = 1024;            // tiles that leave room for one or two workgroups per CU
constexpr int kP = 4;                      // position groups per wave sharing one A read
#ifndef CT_GCONV_LDS
#define CT_GCONV_LDS (32 * 1024)
#endif
#ifndef CT_GCONV_LDS_WRW
#define CT_GCONV_LDS_WRW (64 * 1024)
#endif
constexpr int kLdsBudget = CT_GCONV_LDS;         // forward / backward-data tile budget: ~4 workgroups per CU (measured best)
constexpr int kLdsBudgetWrw = CT_GCONV_LDS_WRW;  // backward-weight tile budget
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
                                                int td0, int th0, int lane, int wave, int nwaves) {
  const int Dz = DIM == 3 ? a.TD + 2 : 1;
  const size_t vol = (size_t)a.D * a.H * a.W;
  const int total = planes * a.plane;
  for (int i = threadIdx.x; i < (total >> 2); i += blockDim.x) ((float4*)xs)[i] = make_float4(0, 0, 0, 0);
  for (int i = ((total >> 2) << 2) + threadIdx.x; i < total; i += blockDim.x) xs[i] = 0.0f;
  __syncthreads();
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

// grid = (nD*nH, groups, B)
template <int DIM>
__global__ void __launch_bounds__(kThreadsBig) gconv_fwd_kernel(GconvArgs a) {
  extern __shared__ __align__(16) float lds[];
  const int tile = blockIdx.x, grp = blockIdx.y, b = blockIdx.z;
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

  for (int mt = 0; mt < MT; ++mt) {
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

bool plan_tiles_budget(GconvArgs& a, int dim, size_t extra_per_pos_bytes, size_t fixed_bytes, int cin_planes,
                       int plane_mod, size_t budget) {
  // LDS per workgroup = cin_planes*plane*4 + fixed + extra_per_pos*TD*TH*W  <= budget
  a.Ws = a.W;                                     // rows keep the tensor width (no x halo)
  const int dzh = dim == 3 ? 2 : 0;
  for (int TD = dim == 3 ? a.D : 1; TD >= 1; --TD) {
    for (int TH = a.H; TH >= 1; --TH) {
      const int Hs = TH + 2;
      int plane = (TD + dzh) * Hs * a.Ws;
      plane += (plane_mod - (plane & 31) + 32) & 31;   // plane == plane_mod (mod 32): conflict-free operand reads
      const size_t bytes = (size_t)cin_planes * plane * 4 + fixed_bytes + extra_per_pos_bytes * TD * TH * a.W + 64;
      if (bytes <= budget) {
        a.TD = TD; a.TH = TH; a.Hs = Hs; a.plane = plane;
        a.nD = (a.D + TD - 1) / TD; a.nH = (a.H + TH - 1) / TH;
        return true;
      }
      if (dim == 3 && TD > 1) break;               // shrink depth first, then rows
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
  return CT_OK;
}

template <typename K>
int set_lds_attr(K kernel, size_t bytes) {
  if (bytes > 48 * 1024 &&
      hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return CT_ELAUNCH;
  return CT_OK;
}

int launch_fwd(GconvArgs a, int dim, hipStream_t st) {
  const size_t wbytes = (size_t)a.taps * a.KB * 64 * 4;
  if (!plan_tiles(a, dim, 0, wbytes, a.KB * 4, 16)) return CT_EINVAL;
  const size_t lds = (size_t)a.KB * 4 * a.plane * 4 + wbytes + 2 * kSlack * 4;
  dim3 grid(a.nD * a.nH, a.groups, a.B);
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

}  // namespace

extern "C" {

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

int ct_gconv_bwd_weight(const float* x, const float* g_y, float* g_w, float* g_bias,
                        int B, int groups, int Cin, int Cout, int dim, const int* W, ct_stream_t s) {
  if (!x || !g_y || !g_w) return CT_EINVAL;
  GconvArgs a = {};
  int r = gconv_common(a, B, groups, Cin, Cout, dim, W);
  if (r != CT_OK) return r;
  a.x = x; a.transposed = 0;
  hipStream_t st = (hipStream_t)s;
  // LDS: 16 input planes with halo + 16 rows of g_y + the cross-wave reduction buffer
  const size_t red_bytes = (size_t)(kThreads / 64) * 3 * 256 * 4;
  if (!plan_tiles(a, dim, (size_t)16 * 4, 1024 + red_bytes, 16, /*plane == 4 (mod 32): 16-byte aligned for the DMA*/ 4,
                  kLdsBudgetWrw)) return CT_EINVAL;
  const int gstride_max = ((a.TD * a.TH * a.W + 3) & ~3) | 1;
  const size_t lds = ((size_t)16 * a.plane + (size_t)16 * gstride_max) * 4 + red_bytes + 2 * kSlack * 4;
  if (hipMemsetAsync(g_w, 0, (size_t)groups * Cout * Cin * a.taps * 4, st) != hipSuccess) return CT_ELAUNCH;
  const int U = B * a.nD * a.nH;
  int chunks = (512 + groups - 1) / groups;           // ~512 workgroups in flight
  if (chunks > U) chunks = U;
  if (chunks < 1) chunks = 1;
  const int units_per_wg = (U + chunks - 1) / chunks;
  chunks = (U + units_per_wg - 1) / units_per_wg;
  dim3 grid(chunks, groups);
  CT_CLEAR_ERROR();
  if (dim == 2) {
    if (set_lds_attr(gconv_bwd_weight_kernel<2>, lds) != CT_OK) return CT_ELAUNCH;
    hipLaunchKernelGGL(gconv_bwd_weight_kernel<2>, grid, dim3(kThreads), lds, st, a, g_y, g_w, units_per_wg);
  } else {
    if (set_lds_attr(gconv_bwd_weight_kernel<3>, lds) != CT_OK) return CT_ELAUNCH;
    hipLaunchKernelGGL(gconv_bwd_weight_kernel<3>, grid, dim3(kThreads), lds, st, a, g_y, g_w, units_per_wg);
  }
  if (g_bias) {
    const size_t vol = (size_t)a.D * a.H * a.W;
    hipLaunchKernelGGL(gconv_bias_grad_kernel, dim3(groups * Cout), dim3(256), 0, st, g_y, g_bias, B, groups * Cout, vol);
  }
  CT_CHECK_LAUNCH();
  return CT_OK;
}

}  // extern "C"
