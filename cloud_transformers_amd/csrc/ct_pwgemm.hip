// Pointwise (kernel-size-1) convolutions of the MHCT blocks: the `keys_values_pred` / `after` / union projections
// (layers/multihead_ct.py:31-33,62-66,89-91; nn.Conv1d(k=1) = one [Co,Ci] x [Ci,N] product per cloud) and their two
// gradients, at fp32 accuracy on the f16 matrix pipes of gfx950.
//
// Every fp32 operand element x is scaled by a per-tensor power of two s (from the tensor's max |x|, so that s*max < 2^14:
// inside f16's range with headroom) and split into two f16 terms h = f16(s x), l = f16(s x - h): h + l carries 22 bits of
// s x.  The product is the sum of three v_mfma_f32_32x32x16_f16 terms h_a h_b + h_a l_b + l_a h_b accumulated in fp32 (f16 x
// f16 products are exact in fp32; the dropped l_a l_b term is 2^-22 of the product), rescaled by the exact 1/(s_a s_b) in
// the epilogue: relative error per product <= ~2^-21, the order of an fp32 GEMM's own summation error at these K, at 3/16 of
// the matrix-pipe time of the f32-input MFMA the library GEMMs use (cdna_hip_programming.md §3 "FP32-input MFMA").
//
// One kernel, two operand arrangements (forward and data gradient: A k-contiguous, B row-contiguous; weight gradient: both
// k-contiguous): a 128x128 output tile per 512-thread workgroup (eight waves as 2 x 4, 64x32 each = two 32x32 MFMA tiles, 32
// accumulator registers; <= 128 VGPRs: two workgroups = four waves per SIMD on a CU), K in steps of 32 through two LDS stages
// of four [128 rows][32 k] f16 images (A_h, A_l, B_h, B_l: 64 KiB per workgroup), one barrier per K-step.  The fp32 tiles are
// fetched into registers two K-steps ahead (two register sets) and split into the other stage BETWEEN the MFMAs of the
// current step.  An operand is staged from either orientation with 16-byte loads (see pw_load): k-contiguous rows into an
// image [row][k] whose 16-byte k-groups are XOR-swizzled with the row (one conflict-free ds_read_b128 per fragment,
// MI355X_MICROARCH.md §LDS lane groups); row-contiguous k rows into an image [k pair][row] of dwords (the transposition is
// the split's own pairing of two k rows; one ds_write_b128 per thread, four conflict-free dword reads per fragment).
// The data gradient runs the forward arrangement on a W^T copy.  The weight gradient sums over clouds and points: K = B*N is
// cut into chunks, each chunk's partial [Co,Ci] tile goes to a slab and a second kernel adds the slabs in a fixed order
// (deterministic).  The per-tensor maxima arrive as partial maxima (ct_amax_f32's per-block ones, or what the kernel that
// wrote the operand left per channel) and are folded when the kernel starts.  HISTORY.md §4.9 has the measurements.
#include "ct_common.h"
#include <cstdlib>
#include <type_traits>

typedef _Float16 pw_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 pw_h2 __attribute__((ext_vector_type(2)));
typedef float pw_f2 __attribute__((ext_vector_type(2)));
typedef float pw_acc __attribute__((ext_vector_type(16)));

constexpr int kPwTile = 128, kPwBK = 32, kPwThreads = 512, kPwR = 8;   // kPwR: operand elements per thread and K-step
constexpr int kPwImg = kPwTile * kPwBK;                 // halves per image
constexpr int kPwStage = 4 * kPwImg;                    // A_h, A_l, B_h, B_l
constexpr int kPwStageBytes = 2 * kPwStage * 2;         // two stages
// behind the stages: the scale exponents of the tile's 128 A rows and 128 B rows / columns (kept to the epilogue) and the
// scratch they are folded through
constexpr int kPwLdsBytes = kPwStageBytes + (256 + 1024) * 4;

struct PwArgs {
  const float* A; const float* B; float* C;
  long long a_bs, b_bs, c_zs;      // element strides: operand per cloud, output per z slice
  int lda, ldb, ldc;
  int M, N, K;                     // output rows / columns, summed extent per cloud
  int tilesM, tilesN, ksplit, Kc;  // z = cloud * ksplit + chunk; the chunk sums k in [chunk*Kc, min(K, (chunk+1)*Kc))
  int Z;                           // clouds * ksplit
  const float* amax_a; const float* amax_b;
  int n_amax_a, n_amax_b;          // partial maxima per operand (<= kPwAmaxMax)
  // > 0: the operand's maxima are PER ROW of its k-contiguous arrangement, laid out [n_amax / rows][rows] (a producer's
  // per-channel or per-(cloud, channel) maxima, ct_pw_prep_weight_rs's per-tile row / column maxima): every row gets its
  // own power-of-two scale, which factors out of the row's (column's) outputs exactly
  int rows_a, rows_b;
  // ADD instantiation only: out = product + D, D laid out as C (same ldc, same z stride) — the data gradient of a layer whose
  // input also feeds a skip connection takes the skip's cotangent here instead of leaving the sum to a separate pass
  const float* D;
#ifdef PW_STAMP
  unsigned long long* dbg;   // development builds: per-phase cycle sums of every wave (tools/dev/pw_stamp.py)
#endif
};

// power of two s with s * amax in [2^13, 2^14); 1 for an all-zero or non-finite tensor (inf / nan then flow through h)
__device__ __forceinline__ int pw_scale_exp(unsigned amax_bits) {
  const float m = __uint_as_float(amax_bits);
  if (!(m > 0.f) || !(m < __builtin_inff())) return 0;
  int e;
  (void)frexpf(m, &e);
  return min(14 - e, 126);   // 2^126 is finite; the epilogue undoes the two scales one after the other
}

// image element (row, k8 group g) -> half offset: 16-byte groups XOR-swizzled with the row
__device__ __forceinline__ int pw_slot(int row, int g) { return row * kPwBK + ((g ^ ((row >> 2) & 3)) << 3); }

// Registers of one operand tile [128 rows][32 k] for thread t of 512: eight elements, two 16-byte loads.
// KMAJOR (k contiguous in memory): row t>>2, k8 group t&3, r[i] = k-th element; image [row][k] halves, 16-byte k8 groups
// XOR-swizzled with the row (pw_slot), fragments by one ds_read_b128.
// Otherwise (row index contiguous): k pair t>>5 (k rows 2p, 2p+1), rows 4*(t&31) .. +3, r[4j + c] = (k row j, row c); a
// dword-wide load here costs 64 issue cycles against 90 for 16 bytes (phase stamps, tools/dev/pw_stamp.py), so the
// thread takes four rows of two k rows and the image is [k pair][row] dwords (the (k, k+1) halves of a row in one dword):
// the four rows' dwords are one contiguous ds_write_b128, and a fragment is four conflict-free dword reads 128 dwords apart.
// Loads are unconditional (a predicated load becomes a branch and a wait per load): rows past the operand's end read its
// last rows instead — they only reach output rows / columns that are never stored, so nothing has to be zeroed for them.
template <bool KMAJOR>
struct PwLane {
  const float* p;   // KMAJOR: src + min(row, R-1)*ld + 8*(t&3);  else: src + 2*(lane>>5)*ld + min(row, R-4)
  unsigned off;     // the same as a byte offset from src (an operand slice is < 4 GiB: pw_plan)
};

template <bool KMAJOR>
__device__ __forceinline__ PwLane<KMAJOR> pw_lane(const float* __restrict__ src, int ld, int row0, int R, int wave, int t) {
  PwLane<KMAJOR> L;
  if constexpr (KMAJOR) {
    const int row = row0 + (t >> 2);
    L.off = ((unsigned)min(row, R - 1) * (unsigned)ld + (t & 3) * 8) * 4u;
  } else {
    const int row = row0 + 4 * (t & 31);
    L.off = ((unsigned)(2 * ((t >> 5) & 1)) * (unsigned)ld + (unsigned)min(row, R - 4)) * 4u;
  }
  L.p = (const float*)((const char*)src + L.off);
  return L;
}

// A whole K-step: wave-uniform base (KMAJOR: src + k0 floats; else: src + (k0 + 4*wave) rows) + the lane's 32-bit offset.
template <bool KMAJOR>
__device__ __forceinline__ void pw_load_full(float (&r)[kPwR], const char* __restrict__ ubase, size_t ldbytes, unsigned off) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float4 v = *(const float4*)(ubase + (KMAJOR ? (size_t)16 * q : q * ldbytes) + off);
    r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
  }
}

// Any K-step (the first, and a partial last one): k indices clamped into [.., kend), the returned bits say which loads
// count: bit q per load.
template <bool KMAJOR>
__device__ __forceinline__ unsigned pw_load(float (&r)[kPwR], const PwLane<KMAJOR>& L, int ld, int k0, int kend, int wave, int t) {
  unsigned ok = 0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    float4 v;
    if constexpr (KMAJOR) {
      const int k = k0 + (t & 3) * 8 + 4 * q;
      v = *(const float4*)(L.p + (min(k, kend - 4) - (t & 3) * 8));
      ok |= (unsigned)(k < kend) << q;
    } else {
      const int lk = 2 * ((t >> 5) & 1), k = k0 + 4 * wave + lk + q;       // L.p already holds the lane's lk rows
      v = *(const float4*)(L.p + ((long long)min(k, kend - 1) - lk) * ld);
      ok |= (unsigned)(k < kend) << q;
    }
    r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
  }
  return ok;
}

// two scaled values -> packed f16 pairs h = f16(s a), l = f16(s a - h): v_fma_mix{lo,hi}_f16 take the fp32 value, the scale
// and the f16 h term in one instruction each (2 per element; the cvt / cvt-back / subtract chain is 3.5).  The s_nop covers
// the wait state a half-register write needs before a VALU reads it (nothing is padded inside an asm statement).
__device__ __forceinline__ void pw_split2(float a0, float a1, float s, unsigned& h, unsigned& l) {
  asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
      "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
      "s_nop 0\n\t"
      "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "s_nop 0"
      : "=&v"(h), "=&v"(l)
      : "v"(a0), "v"(a1), "v"(s));
}

// Two such units with their instructions interleaved: every half-register write is followed by an instruction of the other
// unit before it is read, so only the last one needs the wait state (s_nop issues in 4 cycles like a vector instruction:
// MI355X_MICROARCH.md 'vector-instruction ISSUE cost').
__device__ __forceinline__ void pw_split2x2(float a0, float a1, float sa, float b0, float b1, float sb, unsigned& ha, unsigned& la,
                                            unsigned& hb, unsigned& lb) {
  asm("v_fma_mixlo_f16 %0, %4, %6, 0\n\t"
      "v_fma_mixlo_f16 %2, %7, %9, 0\n\t"
      "v_fma_mixhi_f16 %0, %5, %6, 0\n\t"
      "v_fma_mixhi_f16 %2, %8, %9, 0\n\t"
      "v_fma_mixlo_f16 %1, %4, %6, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %3, %7, %9, -%2 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %1, %5, %6, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %3, %8, %9, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "s_nop 0"
      : "=&v"(ha), "=&v"(la), "=&v"(hb), "=&v"(lb)
      : "v"(a0), "v"(a1), "v"(sa), "v"(b0), "v"(b1), "v"(sb));
}

// eight values -> the h and l fragments' 16 bytes
__device__ __forceinline__ void pw_split8(const float (&v)[8], float s, uint4& h, uint4& l) {
  unsigned hh[4], ll[4];
#if defined(PW_ABL) && PW_ABL == 1
#pragma unroll
  for (int i = 0; i < 4; ++i) { hh[i] = __float_as_uint(v[2 * i]); ll[i] = __float_as_uint(v[2 * i + 1]); }
#else
#pragma unroll
  for (int i = 0; i < 4; ++i) pw_split2(v[2 * i], v[2 * i + 1], s, hh[i], ll[i]);
#endif
  h = make_uint4(hh[0], hh[1], hh[2], hh[3]);
  l = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}

template <bool KMAJOR>
__device__ __forceinline__ void pw_store(float (&r)[kPwR], unsigned ok, _Float16* imgH, _Float16* imgL, float s, int wave, int t) {
  if (ok != 0x3u) {                          // edge tiles and the last partial K-step only
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (!((ok >> (i >> 2)) & 1)) r[i] = 0.f;
  }
  uint4 h, l;
  if constexpr (KMAJOR) {
    pw_split8(r, s, h, l);
    const int o = pw_slot(t >> 2, t & 3);
    *(uint4*)(imgH + o) = h;
    *(uint4*)(imgL + o) = l;
  } else {
    pw_split2(r[0], r[4], s, h.x, l.x);
    pw_split2(r[1], r[5], s, h.y, l.y);
    pw_split2(r[2], r[6], s, h.z, l.z);
    pw_split2(r[3], r[7], s, h.w, l.w);
    const int o = ((2 * wave + ((t >> 5) & 1)) * kPwTile + 4 * (t & 31)) * 2;      // dword [k pair][row], in halves
    *(uint4*)(imgH + o) = h;
    *(uint4*)(imgL + o) = l;
  }
}

// The same in pieces, for interleaving with the MFMAs of the current K-step: the mask (edge tiles only), four units of one
// (k, k+1) pair each, the two stores.
__device__ __forceinline__ void pw_mask(float (&r)[kPwR], unsigned ok) {
  if (ok != 0x3u) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (!((ok >> (i >> 2)) & 1)) r[i] = 0.f;
  }
}
template <bool KMAJOR>
__device__ __forceinline__ void pw_unit(const float (&r)[kPwR], int i, float s, unsigned (&h)[4], unsigned (&l)[4]) {
  if constexpr (KMAJOR) pw_split2(r[2 * i], r[2 * i + 1], s, h[i], l[i]);
  else pw_split2(r[i], r[4 + i], s, h[i], l[i]);
}
template <bool KMAJOR>
__device__ __forceinline__ void pw_write(const unsigned (&h)[4], const unsigned (&l)[4], _Float16* imgH, _Float16* imgL, int wave, int t) {
  const int o = KMAJOR ? pw_slot(t >> 2, t & 3) : ((2 * wave + ((t >> 5) & 1)) * kPwTile + 4 * (t & 31)) * 2;
  *(uint4*)(imgH + o) = make_uint4(h[0], h[1], h[2], h[3]);
  *(uint4*)(imgL + o) = make_uint4(l[0], l[1], l[2], l[3]);
}

// fragment of the 32x32x16 MFMA for lane (row, k8 group g) of an image
template <bool KMAJOR>
__device__ __forceinline__ pw_h8 pw_frag(const _Float16* img, int row, int sw, int g) {
  if constexpr (KMAJOR) {
    return *(const pw_h8*)(img + row * kPwBK + ((g ^ sw) << 3));
  } else {
    const unsigned* d = (const unsigned*)img + (4 * g) * kPwTile + row;
    const uint4 v = make_uint4(d[0], d[kPwTile], d[2 * kPwTile], d[3 * kPwTile]);
    return __builtin_bit_cast(pw_h8, v);
  }
}

template <bool A_KMAJOR, bool B_KMAJOR, bool ADD = false>
__global__ void __launch_bounds__(kPwThreads, 4) pw_gemm_kernel(PwArgs a) {
  extern __shared__ __attribute__((aligned(16))) _Float16 pw_lds[];
  // blocks that share an XCD (id % 8) take consecutive tiles: the M tiles of one [K, 128] operand panel run side by side on
  // one L2 (bijective form of the remap, cdna_hip_programming.md §5)
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, rr = nwg & 7;
#ifdef PW_PHASE
  // experiment: the second workgroup of every CU (dispatch order: 32 workgroups per XCD fill its CUs once) starts half a
  // tile late, so that one's prologue / epilogue falls into the other's K loop
  if (blockIdx.x >= 256 && blockIdx.x < 512) {
    const int n = ((a.K < a.Kc ? a.K : a.Kc) / kPwBK) * PW_PHASE / 8128 + 1;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  const int id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (blockIdx.x >> 3);
  const int mt = id % a.tilesM, rest = id / a.tilesM, nt = rest % a.tilesN, z = rest / a.tilesN;
  const int cloud = z / a.ksplit, chunk = z - cloud * a.ksplit;
  const int kbeg = chunk * a.Kc, kend = min(a.K, kbeg + a.Kc);
  const float* A = a.A + cloud * a.a_bs;
  const float* B = a.B + cloud * a.b_bs;
  float* C = a.C + z * a.c_zs;
  const int m0 = mt * kPwTile, n0 = nt * kPwTile;
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wm = w >> 2, wn = w & 3, r = lane & 31, h = lane >> 5;
  // the operands' max |.|: the partial maxima (ct_amax_f32's kPwAmaxLen, or a producer's per-channel / per-row maxima) strided
  // over the threads,
  // folded through the (still unused) LDS
  unsigned ma = 0u, mb = 0u;
  // (an operand with per-row scales never uses its tensor-wide maximum: its up to 32768 partial maxima — 128 KiB per
  //  workgroup — are not folded)
  if (a.amax_a && a.rows_a <= 0)
    for (int i = t; i < a.n_amax_a; i += kPwThreads) ma = max(ma, __float_as_uint(a.amax_a[i]) & 0x7fffffffu);
  if (a.amax_b && !(B_KMAJOR && a.rows_b > 0))
    for (int i = t; i < a.n_amax_b; i += kPwThreads) mb = max(mb, __float_as_uint(a.amax_b[i]) & 0x7fffffffu);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    ma = max(ma, (unsigned)__shfl_xor((int)ma, o, 64));
    mb = max(mb, (unsigned)__shfl_xor((int)mb, o, 64));
  }
  unsigned* fold = (unsigned*)pw_lds;
  if (lane == 0) { fold[w] = ma; fold[8 + w] = mb; }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) { ma = max(ma, fold[i]); mb = max(mb, fold[8 + i]); }
  __syncthreads();
  // Scale exponents.  Per tensor: s * max in [2^13, 2^14).  Per row (rows_a / rows_b: the operand's maxima are per row of
  // its k-contiguous arrangement): the same with the row's own maximum — a channel 2^-20 below the tensor's maximum keeps
  // its 22 bits instead of sinking into f16's subnormals — and the factor comes out of the row's outputs exactly.  A
  // row-contiguous operand (x / g_y of forward and data gradient: k = channel) can only have ONE scale over k.
  int* etab = (int*)((char*)pw_lds + kPwStageBytes);      // [128] A rows of the tile | [128] B rows (= output columns)
  {
    unsigned* scr = (unsigned*)(etab + 256);              // [4][128] A | [4][128] B
    const int i = t & 127, part = t >> 7;
    if (a.rows_a > 0) {
      const int row = min(m0 + i, a.M - 1), nb = a.n_amax_a / a.rows_a;
      unsigned m = 0u;
      for (int j = part; j < nb; j += 4) m = max(m, __float_as_uint(a.amax_a[(size_t)j * a.rows_a + row]) & 0x7fffffffu);
      scr[part * 128 + i] = m;
    }
    if (B_KMAJOR && a.rows_b > 0) {
      const int row = min(n0 + i, a.N - 1), nb = a.n_amax_b / a.rows_b;
      unsigned m = 0u;
      for (int j = part; j < nb; j += 4) m = max(m, __float_as_uint(a.amax_b[(size_t)j * a.rows_b + row]) & 0x7fffffffu);
      scr[512 + part * 128 + i] = m;
    }
    __syncthreads();
    if (t < 128) {
      etab[t] = a.rows_a > 0 ? pw_scale_exp(max(max(scr[t], scr[128 + t]), max(scr[256 + t], scr[384 + t]))) : pw_scale_exp(ma);
    } else if (t < 256) {
      const int u = t - 128;
      etab[t] = (B_KMAJOR && a.rows_b > 0) ? pw_scale_exp(max(max(scr[512 + u], scr[640 + u]), max(scr[768 + u], scr[896 + u])))
                                           : pw_scale_exp(mb);
    }
    __syncthreads();
  }
  // the staging thread's scale: a k-contiguous operand's thread owns (part of) ONE row of the tile
  const float sa = ldexpf(1.f, etab[t >> 2]);
  const float sb = ldexpf(1.f, B_KMAJOR ? etab[128 + (t >> 2)] : etab[128]);

  pw_acc acc[2];                                     // wave tile 64 x 32: rows 64*wm + 32*i, columns 32*wn
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  // two register sets: the loads of K-step kt+2 go out at the top of step kt and are split into LDS at the end of step kt+1,
  // so an HBM / L2 round trip has two steps of MFMAs (and the CU's other workgroup) to hide behind
  float ra0[kPwR], rb0[kPwR], ra1[kPwR], rb1[kPwR];
  unsigned oka0, okb0, oka1 = 0, okb1 = 0;
  const PwLane<A_KMAJOR> la = pw_lane<A_KMAJOR>(A, a.lda, m0, a.M, w, t);
  const PwLane<B_KMAJOR> lb = pw_lane<B_KMAJOR>(B, a.ldb, n0, a.N, w, t);
  const int T = (kend - kbeg + kPwBK - 1) / kPwBK, F = (kend - kbeg) / kPwBK;   // K-steps, whole ones
  // wave-uniform bases of the next whole K-step to load and their advance per step
  const size_t lda4 = (size_t)a.lda * 4, ldb4 = (size_t)a.ldb * 4;
  const char* ua = (const char*)A + (A_KMAJOR ? (size_t)(kbeg + kPwBK) * 4 : (size_t)(kbeg + kPwBK + 4 * w) * lda4);
  const char* ub = (const char*)B + (B_KMAJOR ? (size_t)(kbeg + kPwBK) * 4 : (size_t)(kbeg + kPwBK + 4 * w) * ldb4);
  const size_t stepa = A_KMAJOR ? (size_t)kPwBK * 4 : kPwBK * lda4, stepb = B_KMAJOR ? (size_t)kPwBK * 4 : kPwBK * ldb4;

  // loads of K-step `step` into one register set.  WHOLE: the caller knows step < F — no branch, so that the number of loads in
  // flight is the same on every path into the split that follows (at a join the compiler waits for the shortest queue: a
  // conditional fetch drains the ring).  Otherwise any step (nothing past the last: its stores refill a stage nobody reads).
  auto fetch = [&](auto whole, int step, float (&xa)[kPwR], float (&xb)[kPwR], unsigned& xoka, unsigned& xokb) {
    if (decltype(whole)::value || step < F) {
#if !(defined(PW_ABL) && PW_ABL == 3)
      pw_load_full<A_KMAJOR>(xa, ua, lda4, la.off);
      pw_load_full<B_KMAJOR>(xb, ub, ldb4, lb.off);
#endif
      xoka = 0x3u; xokb = 0x3u;
      ua += stepa; ub += stepb;
    } else if (step < T) {                           // the partial last step
      xoka = pw_load<A_KMAJOR>(xa, la, a.lda, kbeg + step * kPwBK, kend, w, t);
      xokb = pw_load<B_KMAJOR>(xb, lb, a.ldb, kbeg + step * kPwBK, kend, w, t);
    }
  };

  oka0 = pw_load<A_KMAJOR>(ra0, la, a.lda, kbeg, kend, w, t);
  okb0 = pw_load<B_KMAJOR>(rb0, lb, a.ldb, kbeg, kend, w, t);
  fetch(std::false_type{}, 1, ra1, rb1, oka1, okb1);
  pw_store<A_KMAJOR>(ra0, oka0, pw_lds, pw_lds + kPwImg, sa, w, t);
  pw_store<B_KMAJOR>(rb0, okb0, pw_lds + 2 * kPwImg, pw_lds + 3 * kPwImg, sb, w, t);
  __syncthreads();

  // fragment rows: 64*wm + 32*i + r (A), 32*wn + r (B); k8 group 2*ks + h
  const int rowA = 64 * wm + r, rowB = 32 * wn + r, sw = (r >> 2) & 3;

#ifdef PW_STAMP
  unsigned long long tprev = __builtin_amdgcn_s_memtime(), tsum[6] = {0, 0, 0, 0, 0, 0};
#define PW_T(i)                                             \
  do {                                                      \
    const unsigned long long now__ = __builtin_amdgcn_s_memtime(); \
    tsum[i] += now__ - tprev;                               \
    tprev = now__;                                          \
  } while (0)
#else
#define PW_T(i) ((void)0)
#endif
  // K-step kt: fetch step kt+2 into (xa, xb), MFMAs on stage kt&1, split step kt+1 from (ya, yb) into the other stage
  auto kstep = [&](auto whole, int kt, float (&xa)[kPwR], float (&xb)[kPwR], unsigned& xoka, unsigned& xokb, float (&ya)[kPwR],
                   float (&yb)[kPwR], unsigned yoka, unsigned yokb) {
    const _Float16* st = pw_lds + (kt & 1) * kPwStage;
    _Float16* nx = pw_lds + ((kt + 1) & 1) * kPwStage;
    PW_T(0);
    fetch(whole, kt + 2, xa, xb, xoka, xokb);
    __builtin_amdgcn_sched_barrier(0);               // the loads go out before the MFMAs, not after them
    PW_T(1);
    // The split of step kt+1 (its loads landed a step ago) is issued in the shadow of this step's MFMAs: the wave's MFMAs queue
    // behind those of the SIMD's other three waves, and between two of them it has vector-ALU work that needs neither the
    // matrix pipe nor this step's LDS stage.  sched_barrier pins the order (the compiler keeps asm and MFMA blocks apart).
    if constexpr (!decltype(whole)::value) {          // only the last steps can hold a partial K-step
      pw_mask(ya, yoka);
      pw_mask(yb, yokb);
    }
    unsigned hA[4], lA[4], hB[4], lB[4];
    pw_h8 ah[2], al[2], bh, bl;
    auto frags = [&](int ks) {
      const int g = 2 * ks + h;
      bh = pw_frag<B_KMAJOR>(st + 2 * kPwImg, rowB, sw, g);
      bl = pw_frag<B_KMAJOR>(st + 3 * kPwImg, rowB, sw, g);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = pw_frag<A_KMAJOR>(st, rowA + 32 * i, sw, g);
        al[i] = pw_frag<A_KMAJOR>(st + kPwImg, rowA + 32 * i, sw, g);
      }
    };
#define PW_MF(x, y)                                                                   \
  do {                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[0], y, acc[0], 0, 0, 0);        \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[1], y, acc[1], 0, 0, 0);        \
    __builtin_amdgcn_sched_barrier(0);                                                \
  } while (0)
    frags(0);
    PW_MF(ah, bh);
    pw_unit<A_KMAJOR>(ya, 0, sa, hA, lA);
    pw_unit<A_KMAJOR>(ya, 1, sa, hA, lA);
    PW_MF(ah, bl);
    pw_unit<A_KMAJOR>(ya, 2, sa, hA, lA);
    pw_unit<A_KMAJOR>(ya, 3, sa, hA, lA);
    pw_write<A_KMAJOR>(hA, lA, nx, nx + kPwImg, w, t);
    PW_MF(al, bh);
    frags(1);
    pw_unit<B_KMAJOR>(yb, 0, sb, hB, lB);
    pw_unit<B_KMAJOR>(yb, 1, sb, hB, lB);
    PW_MF(ah, bh);
    pw_unit<B_KMAJOR>(yb, 2, sb, hB, lB);
    pw_unit<B_KMAJOR>(yb, 3, sb, hB, lB);
    pw_write<B_KMAJOR>(hB, lB, nx + 2 * kPwImg, nx + 3 * kPwImg, w, t);     // two MFMA groups ahead of the barrier
    PW_MF(ah, bl);
    PW_MF(al, bh);
#undef PW_MF
    PW_T(4);
    __syncthreads();
    PW_T(5);
  };
  int kt = 0;
  for (; kt + 3 < F; kt += 2) {                      // both fetches (steps kt+2, kt+3) are whole K-steps
    kstep(std::true_type{}, kt, ra0, rb0, oka0, okb0, ra1, rb1, oka1, okb1);
    kstep(std::true_type{}, kt + 1, ra1, rb1, oka1, okb1, ra0, rb0, oka0, okb0);
  }
  for (; kt < T; kt += 2) {                          // the last steps
    kstep(std::false_type{}, kt, ra0, rb0, oka0, okb0, ra1, rb1, oka1, okb1);
    if (kt + 1 >= T) break;
    kstep(std::false_type{}, kt + 1, ra1, rb1, oka1, okb1, ra0, rb0, oka0, okb0);
  }

#ifdef PW_STAMP
  if (a.dbg && lane == 0 && blockIdx.x < 4096)
    for (int i = 0; i < 6; ++i) a.dbg[((size_t)blockIdx.x * 8 + w) * 6 + i] = tsum[i];
#endif
  // D of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  const float ib = ldexpf(1.f, -etab[128 + 32 * wn + r]);
  const int col = n0 + 32 * wn + r;
  if constexpr (ADD) {
    // the addend's 32 values of this lane requested together (the K loop's operand registers are dead here), then the stores
    const float* D = a.D + z * a.c_zs;
    float dv[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + 64 * wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        dv[i][e] = (row < a.M && col < a.N) ? D[(size_t)row * a.ldc + col] : 0.f;
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int lr = 64 * wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h, row = m0 + lr;
        const float ia = ldexpf(1.f, -etab[lr]);
        if (row < a.M && col < a.N) C[(size_t)row * a.ldc + col] = acc[i][e] * ia * ib + dv[i][e];
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int lr = 64 * wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h, row = m0 + lr;
      const float ia = ldexpf(1.f, -etab[lr]);
#if defined(PW_ABL) && PW_ABL == 4
      if (row < a.M && col < a.N && acc[i][e] == 12345.678f) C[(size_t)row * a.ldc + col] = acc[i][e] * ia * ib;
#else
      if (row < a.M && col < a.N) C[(size_t)row * a.ldc + col] = acc[i][e] * ia * ib;
#endif
    }
}

// out[i] = sum_z slabs[z][i] in a fixed order: 32 float4 outputs x 8 z-groups per block, each group summed z ascending
// (four loads in flight), the groups added in order through LDS
__global__ void __launch_bounds__(256) pw_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, int n4, int Z) {
  __shared__ float4 part[8][32];
  const int li = threadIdx.x & 31, zg = threadIdx.x >> 5, i = blockIdx.x * 32 + li;
  const int per = (Z + 7) >> 3, z0 = zg * per, z1 = min(Z, z0 + per);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float4* p = (const float4*)slabs + i;
    int z = z0;
    for (; z + 4 <= z1; z += 4) {
      const float4 a = p[(size_t)z * n4], b = p[(size_t)(z + 1) * n4], c = p[(size_t)(z + 2) * n4], d = p[(size_t)(z + 3) * n4];
      s.x = (((s.x + a.x) + b.x) + c.x) + d.x; s.y = (((s.y + a.y) + b.y) + c.y) + d.y;
      s.z = (((s.z + a.z) + b.z) + c.z) + d.z; s.w = (((s.w + a.w) + b.w) + c.w) + d.w;
    }
    for (; z < z1; ++z) {
      const float4 a = p[(size_t)z * n4];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
  }
  part[zg][li] = s;
  __syncthreads();
  if (zg == 0 && i < n4) {
#pragma unroll
    for (int g = 1; g < 8; ++g) {
      const float4 a = part[g][li];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    ((float4*)out)[i] = s;
  }
}

// max |x| as the bit pattern's unsigned maximum, one partial maximum per block (kPwAmaxLen of them, zero where a block has no
// data): no same-address atomics (they serialise at ~40 ns each: 512 tickets cost as much as streaming 67 MB), no memset
// node, no second launch — the GEMM's 512 threads fold the 512 partials when it starts.  Eight 16-byte loads in flight per
// thread.
__global__ void __launch_bounds__(256) pw_amax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
  typedef unsigned pw_u4 __attribute__((ext_vector_type(4)));
  unsigned m = 0;
  const long long n4 = n >> 2;
  const pw_u4* p = (const pw_u4*)x;
  long long i = ((long long)blockIdx.x * 8) * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 8 * 256;
  for (; i + 7 * 256 < n4; i += stride) {
    pw_u4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * 256);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      m = max(max(m, v[u].x & 0x7fffffffu), max(v[u].y & 0x7fffffffu, max(v[u].z & 0x7fffffffu, v[u].w & 0x7fffffffu)));
  }
  for (int u = 0; u < 8; ++u)
    if (i + u * 256 < n4) {
      const pw_u4 v = p[i + u * 256];
      m = max(max(m, v.x & 0x7fffffffu), max(v.y & 0x7fffffffu, max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
    }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) m = max(m, __float_as_uint(x[n4 * 4 + threadIdx.x]) & 0x7fffffffu);
  __shared__ unsigned wave_max[4];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
}

// W [R][C] -> W^T [C][R] (the data gradient then runs the forward arrangement: its A operand k-contiguous)
__global__ void __launch_bounds__(256) pw_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int R, int C) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    if (r < R && c < C) tile[ty + 8 * i][tx] = w[(size_t)r * C + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < R && c < C) wt[(size_t)c * R + r] = tile[tx][ty + 8 * i];
  }
}

#ifndef CT_PW_ZTARGET
#define CT_PW_ZTARGET 512
#endif
constexpr int kPwZTarget = CT_PW_ZTARGET;
constexpr int kPwAmaxLen = 512;       // = kPwThreads: one partial maximum per GEMM thread
constexpr int kPwAmaxMax = 32768;     // what a producer may leave instead (per channel, per (cloud, channel), per (column tile, row))

// The same with the weight's partial maxima (one per 32x32 tile, <= kPwAmaxMax tiles): what a layer's forward needs of its
// weight for all three products, in one launch.
__global__ void __launch_bounds__(256) pw_prep_weight_kernel(const float* __restrict__ w, float* __restrict__ wt, unsigned* __restrict__ amax,
                                                             int R, int C) {
  __shared__ float tile[32][33];
  __shared__ unsigned wave_max[4];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    if (r < R && c < C) {
      const float v = w[(size_t)r * C + c];
      tile[ty + 8 * i][tx] = v;
      m = max(m, __float_as_uint(v) & 0x7fffffffu);
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) amax[blockIdx.y * gridDim.x + blockIdx.x] = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
  if (wt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 8 * i, r = r0 + tx;
      if (r < R && c < C) wt[(size_t)c * R + r] = tile[tx][ty + 8 * i];
    }
  }
}

// ct_pw_prep_weight_rs: W [R][C] -> W^T (optional), the per-row maxima of every 32-column tile rowmax[tile_c][R] and the
// per-column maxima of every 32-row tile colmax[tile_r][C] — the per-row maxima of W and of W^T in the [n / rows][rows]
// layout ct_pw_gemm_rs folds.  One launch, one read of W.
__global__ void __launch_bounds__(256) pw_prep_weight_rs_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                                unsigned* __restrict__ rowmax, unsigned* __restrict__ colmax, int R, int C) {
  __shared__ float tile[32][33];
  __shared__ unsigned cpart[8][32];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  unsigned cm = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    unsigned m = 0;
    if (r < R && c < C) {
      const float v = w[(size_t)r * C + c];
      tile[ty + 8 * i][tx] = v;
      m = __float_as_uint(v) & 0x7fffffffu;
    }
    cm = max(cm, m);
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));      // over the row's 32 columns (a half wave)
    if (tx == 0 && r < R) rowmax[(size_t)blockIdx.x * R + r] = m;
  }
  cpart[ty][tx] = cm;
  __syncthreads();
  if (ty == 0 && c0 + tx < C) {
    unsigned m = cpart[0][tx];
#pragma unroll
    for (int j = 1; j < 8; ++j) m = max(m, cpart[j][tx]);
    colmax[(size_t)blockIdx.y * C + c0 + tx] = m;
  }
  if (wt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 8 * i, r = r0 + tx;
      if (r < R && c < C) wt[(size_t)c * R + r] = tile[tx][ty + 8 * i];
    }
  }
}

// ct_amax_rows_f32: out[c] = max over (b, n) of |x[b][c][n]| for x [B][C][N] — the per-row maxima of an activation the
// weight gradient reads when no producer left them.  One workgroup per channel, 16-byte loads.
__global__ void __launch_bounds__(256) pw_amax_rows_kernel(const float* __restrict__ x, int B, int C, int N, unsigned* __restrict__ out) {
  typedef unsigned pw_u4 __attribute__((ext_vector_type(4)));
  const int c = blockIdx.x, n4 = N >> 2;
  unsigned m = 0;
  for (int b = 0; b < B; ++b) {
    const pw_u4* p = (const pw_u4*)(x + ((size_t)b * C + c) * N);
    for (int i = threadIdx.x; i < n4; i += 256) {
      const pw_u4 v = __builtin_nontemporal_load(p + i);
      m = max(max(m, v.x & 0x7fffffffu), max(v.y & 0x7fffffffu, max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
    }
  }
  __shared__ unsigned wave_max[4];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[c] = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
}

struct PwPlan {
  int M, N, K, Z, ksplit, Kc, tilesM, tilesN;
  size_t ws;
};

static bool pw_plan(int mode, int B, int Co, int Ci, int N, PwPlan& p) {
  if (B < 1 || Co < 1 || Ci < 1 || N < 1 || (Co & 3) || (Ci & 3) || (N & 3)) return false;
  if (mode == CT_PW_FWD) { p.M = Co; p.N = N; p.K = Ci; }
  else if (mode == CT_PW_DGRAD || mode == CT_PW_DGRAD_T) { p.M = Ci; p.N = N; p.K = Co; }
  else if (mode == CT_PW_WGRAD) { p.M = Co; p.N = Ci; p.K = N; }
  else return false;
  p.tilesM = (p.M + kPwTile - 1) / kPwTile;
  p.tilesN = (p.N + kPwTile - 1) / kPwTile;
  p.ksplit = 1;
  p.Kc = (p.K + kPwBK - 1) / kPwBK * kPwBK;
  p.ws = 0;
  if (mode == CT_PW_WGRAD) {
    // about one round of the chip's 512 workgroup slots (every slice costs a slab round trip); chunks of whole K-steps, at
    // least eight of them
    const int tiles = p.tilesM * p.tilesN;
    const int zwant = (kPwZTarget + tiles - 1) / tiles;
    int ks = (zwant + B - 1) / B;
    int kc = ((p.K + ks - 1) / ks + kPwBK - 1) / kPwBK * kPwBK;
    if (kc < 8 * kPwBK) kc = 8 * kPwBK;
    p.Kc = kc;
    p.ksplit = (p.K + kc - 1) / kc;
    if ((long long)B * p.ksplit > 1) p.ws = (size_t)B * p.ksplit * Co * Ci * sizeof(float);
  }
  if (mode == CT_PW_DGRAD) p.ws = (size_t)Co * Ci * sizeof(float);   // W^T
  p.Z = B * p.ksplit;
  if ((long long)p.tilesM * p.tilesN * p.Z > 0x7fffffffLL) return false;
  if ((long long)Co * N >= (1LL << 30) || (long long)Ci * N >= (1LL << 30) || (long long)Co * Ci >= (1LL << 30)) return false;   // 32-bit lane offsets
  return true;
}

// (A second, persistent 128 x 256 kernel lived here through round 4: faster stand-alone on wide weight gradients and at K >= 1024,
// nothing inside the training steps — 17.2 / 15.2 / 32.6 vs 17.2 / 15.1 / 31.6 ms, profiles/r4_model_steps.txt — and dispatched
// nowhere by default.  Out of the build since round 5: tools/dev/experiments/ct_pwgemm2.h.)
template <bool AK, bool BK, bool ADD = false>
static int pw_launch(const PwArgs& a, int blocks, hipStream_t st) {
  auto k = pw_gemm_kernel<AK, BK, ADD>;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kPwLdsBytes) != hipSuccess) return CT_ELAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3(blocks), dim3(kPwThreads), kPwLdsBytes, st, a);
  return CT_OK;
}

#ifdef PW_STAMP
static unsigned long long* g_pw_dbg = nullptr;
extern "C" void ct_debug_pw_stamp(unsigned long long* buf) { g_pw_dbg = buf; }
#endif

extern "C" {

int ct_amax_f32(const float* x, int64_t n, float* amax, ct_stream_t s) {
  if (!x || !amax || n < 1 || ((uintptr_t)x & 15)) return CT_EINVAL;
  CT_CLEAR_ERROR();
  hipLaunchKernelGGL(pw_amax_kernel, dim3(kPwAmaxLen), dim3(256), 0, (hipStream_t)s, x, (long long)n, (unsigned*)amax);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

int ct_amax_len(void) { return kPwAmaxLen; }

// partial maxima ct_pw_prep_weight writes for a [Co, Ci] weight (one per 32x32 tile), 0 when there would be more than ct_pw_gemm folds
int ct_pw_prep_weight_partials(int Co, int Ci) {
  if (Co < 1 || Ci < 1) return 0;
  const long long n = (long long)((Co + 31) / 32) * ((Ci + 31) / 32);
  return n <= kPwAmaxMax ? (int)n : 0;
}

// W f32[Co,Ci] -> its partial maxima amax f32[ct_pw_prep_weight_partials] and, when wt != NULL, W^T f32[Ci,Co] (the A operand
// of CT_PW_DGRAD_T), in one launch
int ct_pw_prep_weight(const float* w, float* wt, float* amax, int Co, int Ci, ct_stream_t s) {
  if (!w || !amax || ct_pw_prep_weight_partials(Co, Ci) == 0) return CT_EINVAL;
  CT_CLEAR_ERROR();
  hipLaunchKernelGGL(pw_prep_weight_kernel, dim3((Ci + 31) / 32, (Co + 31) / 32), dim3(256), 0, (hipStream_t)s, w, wt, (unsigned*)amax, Co,
                     Ci);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

size_t ct_pw_gemm_workspace_bytes(int mode, int B, int Co, int Ci, int N) {
  PwPlan p;
  return pw_plan(mode, B, Co, Ci, N, p) ? p.ws : 0;
}

// per-row / per-column maxima of W [Co][Ci] for ct_pw_gemm_rs: rowmax f32[ceil(Ci/32)][Co], colmax f32[ceil(Co/32)][Ci];
// wt f32[Ci][Co] = W^T or NULL
int ct_pw_prep_weight_rs(const float* w, float* wt, float* rowmax, float* colmax, int Co, int Ci, ct_stream_t s) {
  if (!w || !rowmax || !colmax || Co < 1 || Ci < 1) return CT_EINVAL;
  if ((long long)((Ci + 31) / 32) * Co > kPwAmaxMax || (long long)((Co + 31) / 32) * Ci > kPwAmaxMax) return CT_EINVAL;
  CT_CLEAR_ERROR();
  hipLaunchKernelGGL(pw_prep_weight_rs_kernel, dim3((Ci + 31) / 32, (Co + 31) / 32), dim3(256), 0, (hipStream_t)s, w, wt,
                     (unsigned*)rowmax, (unsigned*)colmax, Co, Ci);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

int ct_amax_rows_f32(const float* x, int B, int C, int N, float* amax, ct_stream_t s) {
  if (!x || !amax || B < 1 || C < 1 || N < 4 || (N & 3) || ((uintptr_t)x & 15)) return CT_EINVAL;
  CT_CLEAR_ERROR();
  hipLaunchKernelGGL(pw_amax_rows_kernel, dim3(C), dim3(256), 0, (hipStream_t)s, x, B, C, N, (unsigned*)amax);
  CT_CHECK_LAUNCH();
  return CT_OK;
}

int ct_pw_gemm(int mode, const float* a, const float* b, float* out, const float* amax_a, int n_amax_a, const float* amax_b,
               int n_amax_b, void* workspace, size_t workspace_bytes, int B, int Co, int Ci, int N, ct_stream_t s) {
  return ct_pw_gemm_rs(mode, a, b, out, amax_a, n_amax_a, 0, amax_b, n_amax_b, 0, workspace, workspace_bytes, B, Co, Ci, N, s);
}

int ct_pw_gemm_rs(int mode, const float* a, const float* b, float* out, const float* amax_a, int n_amax_a, int rows_a,
                  const float* amax_b, int n_amax_b, int rows_b, void* workspace, size_t workspace_bytes, int B, int Co, int Ci,
                  int N, ct_stream_t s) {
  return ct_pw_gemm_rs_add(mode, a, b, out, nullptr, amax_a, n_amax_a, rows_a, amax_b, n_amax_b, rows_b, workspace, workspace_bytes, B, Co, Ci,
                           N, s);
}

// ct_pw_gemm_rs with out = product + addend (addend f32 laid out as out, 16-byte aligned, may NOT alias out; NULL: ct_pw_gemm_rs).
// CT_PW_FWD / CT_PW_DGRAD / CT_PW_DGRAD_T only (the weight gradient's output is a fold of slabs): CT_EINVAL otherwise.
int ct_pw_gemm_rs_add(int mode, const float* a, const float* b, float* out, const float* addend, const float* amax_a, int n_amax_a,
                      int rows_a, const float* amax_b, int n_amax_b, int rows_b, void* workspace, size_t workspace_bytes, int B, int Co,
                      int Ci, int N, ct_stream_t s) {
  PwPlan p;
  if (!a || !b || !out) return CT_EINVAL;
  if (!pw_plan(mode, B, Co, Ci, N, p)) return CT_EINVAL;
  if ((amax_a && (n_amax_a < 1 || n_amax_a > kPwAmaxMax)) || (amax_b && (n_amax_b < 1 || n_amax_b > kPwAmaxMax))) return CT_EINVAL;
  // per-row maxima: of the A operand's rows in the arrangement the kernel reads it in (W: Co rows; W^T: Ci; g_y: Co) and, for
  // the weight gradient, of x's Ci rows; anything else about rows_* is an error, 0 = the maxima are partials of ONE maximum
  {
    const int want_a = (mode == CT_PW_FWD || mode == CT_PW_WGRAD) ? Co : Ci;
    if (rows_a != 0 && (!amax_a || rows_a != want_a || n_amax_a % rows_a != 0)) return CT_EINVAL;
    if (rows_b != 0 && (!amax_b || n_amax_b % rows_b != 0)) return CT_EINVAL;
    if (mode == CT_PW_WGRAD) { if (rows_b != 0 && rows_b != Ci) return CT_EINVAL; }
    else rows_b = 0;              // a row-contiguous operand has one scale: its maxima are folded into one
    if (mode == CT_PW_DGRAD) rows_a = 0;      // (the maxima belong to W, the kernel reads the W^T this call writes)
  }
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out | (uintptr_t)addend) & 15) return CT_EINVAL;
  if (addend && (mode == CT_PW_WGRAD || addend == out)) return CT_EINVAL;
  if (p.ws && (!workspace || workspace_bytes < p.ws || ((uintptr_t)workspace & 15))) return CT_EWORKSPACE;
  CT_CLEAR_ERROR();
  hipStream_t st = (hipStream_t)s;
  PwArgs g{};
  g.A = a; g.B = b; g.amax_a = amax_a; g.amax_b = amax_b; g.n_amax_a = n_amax_a; g.n_amax_b = n_amax_b;
  g.rows_a = rows_a; g.rows_b = rows_b;
  g.D = addend;
#ifdef PW_STAMP
  g.dbg = g_pw_dbg;
#endif
  g.M = p.M; g.N = p.N; g.K = p.K; g.tilesM = p.tilesM; g.tilesN = p.tilesN; g.ksplit = p.ksplit; g.Kc = p.Kc; g.Z = p.Z;
  const int blocks = p.tilesM * p.tilesN * p.Z;
  int rc;
  if (mode == CT_PW_FWD) {            // A = W [Co][Ci] (k contiguous), B = x[b] [Ci][N] (columns contiguous)
    g.lda = Ci; g.a_bs = 0; g.ldb = N; g.b_bs = (long long)Ci * N; g.C = out; g.ldc = N; g.c_zs = (long long)Co * N;
    rc = addend ? pw_launch<true, false, true>(g, blocks, st) : pw_launch<true, false>(g, blocks, st);
  } else if (mode == CT_PW_DGRAD) {   // A = W^T [Ci][Co] written to the workspace (k = co contiguous), B = g_y[b] [Co][N]
    hipLaunchKernelGGL(pw_transpose_kernel, dim3((Ci + 31) / 32, (Co + 31) / 32), dim3(256), 0, st, a, (float*)workspace, Co, Ci);
    CT_CHECK_LAUNCH();
    g.A = (const float*)workspace;
    g.lda = Co; g.a_bs = 0; g.ldb = N; g.b_bs = (long long)Co * N; g.C = out; g.ldc = N; g.c_zs = (long long)Ci * N;
    rc = addend ? pw_launch<true, false, true>(g, blocks, st) : pw_launch<true, false>(g, blocks, st);
  } else if (mode == CT_PW_DGRAD_T) {  // a IS W^T [Ci][Co] (ct_pw_prep_weight wrote it in the layer's forward)
    g.lda = Co; g.a_bs = 0; g.ldb = N; g.b_bs = (long long)Co * N; g.C = out; g.ldc = N; g.c_zs = (long long)Ci * N;
    rc = addend ? pw_launch<true, false, true>(g, blocks, st) : pw_launch<true, false>(g, blocks, st);
  } else {                            // A = g_y[b] [Co][N], B = x[b] [Ci][N]: both k (= point) contiguous
    g.lda = N; g.a_bs = (long long)Co * N; g.ldb = N; g.b_bs = (long long)Ci * N; g.ldc = Ci;
    g.C = p.ws ? (float*)workspace : out;
    g.c_zs = (long long)Co * Ci;
    rc = pw_launch<true, true>(g, blocks, st);
  }
  if (rc != CT_OK) return rc;
  CT_CHECK_LAUNCH();
  if (mode == CT_PW_WGRAD && p.ws) {
    const int n4 = Co * Ci / 4;
    hipLaunchKernelGGL(pw_reduce_kernel, dim3((n4 + 31) / 32), dim3(256), 0, st, (const float*)workspace, out, n4, p.Z);
    CT_CHECK_LAUNCH();
  }
  return CT_OK;
}

}  // extern "C"
