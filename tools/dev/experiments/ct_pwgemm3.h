// EXPERIMENT, not part of the build (measured and dropped in round 4; to try it again: copy next to ct_pwgemm2.h, include it after
// that header in ct_pwgemm.hip and launch pw3_gemm_kernel<BKM> on one 256-thread workgroup per CU with k2LdsBytes of LDS).
// Result, one box, B8 N4096, us forward / data gradient / weight gradient: 848x512 201.0 / 169.8 / 186.2 against 137.5 / 129.1 /
// 131.3 for the first kernel; 512x1024 172.9 / 177.5 / 162.1; 208x512 50.2 / 55.2 / 72.9.  Correct (errors equal to the other two
// kernels'), 1.4-1.5x slower: with ONE wave per SIMD every instruction of the step is issued by that wave — hipcc's stream for a
// step is 452 vector (221 of them v_accvgpr moves: the 128 accumulators live in AGPRs and the rest spills through them), 249
// scalar and 47 memory instructions for 48 MFMAs, 15 per gap where about 5 hide (MI355X_MICROARCH.md) — and every wait stalls
// the SIMD.  This form needs a hand-written instruction stream, not more C++.
// Pointwise-convolution GEMM, third kernel: pw2_gemm_kernel's tile, images and persistent pipeline (ct_pwgemm2.h) on FOUR waves
// — one per SIMD, up to 512 registers each — instead of eight.  What the second kernel's stamps showed (profiles/r4_pw_gemm_stamps.txt):
// two waves per SIMD leave the matrix pipes idle a third of the loop (their vector work collides), the barrier is a seventh,
// and a tile's store tail (all CUs at once: the HBM write rate) has nowhere to hide with 228 of 256 registers in use.  Here:
//   * a wave owns 64 x 128 of the 128 x 256 tile (eight 32x32 accumulators, 128 registers) and issues its MFMAs in pairs with
//     the step's other work — split pairs, permutes, LDS writes, one global load, one fragment read at a time — placed by hand in
//     the 24 gaps between them (`slot`): one instruction stream per SIMD, nothing to collide with;
//   * B fragments are fetched one 32-column block ahead, A fragments one k16 slice ahead (48 fragment registers, not 96);
//   * the finished tile's accumulators are copied aside and stored 16 bytes at a time in the gaps of the NEXT tile's steps.
#pragma once

constexpr int k3Threads = 256;

template <bool BKM>
__global__ void __launch_bounds__(k3Threads, 1) pw3_gemm_kernel(PwArgs a) {
  extern __shared__ __attribute__((aligned(16))) _Float16 pw_lds[];
  int* tab = (int*)((char*)pw_lds + k2Stages * k2Stage * 2);      // exponents: [M] A rows | [N] B rows (BKM) or [1]
  unsigned* scr = (unsigned*)(tab + k2TabMax);
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wm = w >> 1, wn = w & 1, r = lane & 31, hh = lane >> 5;

  const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot_wg = blockIdx.x >> 3;
  const int items = a.tilesM * a.tilesN * a.Z, ibase = items >> 3, irem = items & 7;
  const int ibeg = xcd * ibase + min(xcd, irem), iend = ibeg + ibase + (xcd < irem ? 1 : 0);
  const int istride = (nwg - xcd + 7) >> 3;
  if (ibeg + slot_wg >= iend) return;

  const size_t lda4 = (size_t)a.lda * 4, ldb4 = (size_t)a.ldb * 4;
  const int kh = t & 1;                                  // k16 half of a k-contiguous operand's staging thread (row t >> 1)

  // ---- load phase.  A: row t >> 1, k 16 kh + 4 q (q = 0..3).  B k-contiguous: rows t >> 1 and 128 + (t >> 1), q = 4 i + k quarter.
  // B row-contiguous: wave w = k octet, lane = (column quad lane & 31, k rows 4 (lane >> 5) + j), q = 4 nh + j (column half nh).
  Pw2Pos L;
  const char* ua = nullptr;
  const char* ub = nullptr;
  unsigned offa = 0, offb0 = 0, offb1 = 0;
  unsigned poffa[4], poffb[8];
  int lkk = 0, lkend = 0;
  auto set_load = [&](int id) {
    const Pw2Item it = pw2_item(a, id);
    L.id = id; L.kt = 0; L.T = it.T; L.alive = true;
    lkk = it.kbeg; lkend = it.kend;
    const int rem = it.kend - it.kbeg - (it.T - 1) * k2BK;
    ua = (const char*)(a.A + (size_t)it.cloud * a.a_bs + it.kbeg);
    const unsigned rowa = (unsigned)min(it.m0 + (t >> 1), a.M - 1) * (unsigned)lda4;
    offa = rowa + 64u * kh;
#pragma unroll
    for (int q = 0; q < 4; ++q) poffa[q] = rowa + 4u * (unsigned)min(16 * kh + 4 * q, rem - 4);
    if constexpr (BKM) {
      ub = (const char*)(a.B + (size_t)it.cloud * a.b_bs + it.kbeg);
      const unsigned rb0 = (unsigned)min(it.n0 + (t >> 1), a.N - 1) * (unsigned)ldb4;
      const unsigned rb1 = (unsigned)min(it.n0 + 128 + (t >> 1), a.N - 1) * (unsigned)ldb4;
      offb0 = rb0 + 64u * kh;
      offb1 = rb1 + 64u * kh;
#pragma unroll
      for (int q = 0; q < 8; ++q) poffb[q] = ((q >> 2) ? rb1 : rb0) + 4u * (unsigned)min(16 * kh + 4 * (q & 3), rem - 4);
    } else {
      ub = (const char*)(a.B + (size_t)it.cloud * a.b_bs) + (size_t)it.kbeg * ldb4;
      offb0 = (unsigned)min(it.n0 + 4 * (lane & 31), a.N - 4) * 4u;             // column half 0
      offb1 = (unsigned)min(it.n0 + 128 + 4 * (lane & 31), a.N - 4) * 4u;       // column half 1
#pragma unroll
      for (int q = 0; q < 8; ++q)
        poffb[q] = ((q >> 2) ? offb1 : offb0) + (unsigned)min(8 * w + 4 * (lane >> 5) + (q & 3), rem - 1) * (unsigned)ldb4;
    }
  };
  auto next_load = [&]() {
    if (!L.alive) return;
    if (L.kt + 1 == L.T) {
      const int id = L.id + istride;
      if (id < iend) set_load(id); else L.alive = false;
      return;
    }
    ++L.kt; lkk += k2BK;
    ua += k2BK * 4;
    ub += BKM ? (size_t)k2BK * 4 : (size_t)k2BK * ldb4;
  };
  auto issue_a1 = [&](float (&xa)[16], int q) {
    const bool part = lkk + k2BK > lkend;
    const float4 v = *(const float4*)(ua + (part ? poffa[q] : offa + 16u * q));
    xa[4 * q] = v.x; xa[4 * q + 1] = v.y; xa[4 * q + 2] = v.z; xa[4 * q + 3] = v.w;
  };
  auto issue_b1 = [&](float (&xb)[32], int q) {
    const bool part = lkk + k2BK > lkend;
    unsigned o;
    if constexpr (BKM) o = ((q >> 2) ? offb1 : offb0) + 16u * (q & 3);
    else o = ((q >> 2) ? offb1 : offb0) + (unsigned)(8 * w + 4 * (lane >> 5) + (q & 3)) * (unsigned)ldb4;
    const float4 v = *(const float4*)(ub + (part ? poffb[q] : o));
    xb[4 * q] = v.x; xb[4 * q + 1] = v.y; xb[4 * q + 2] = v.z; xb[4 * q + 3] = v.w;
  };
  auto issue_all = [&](float (&xa)[16], float (&xb)[32]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_a1(xa, q);
#pragma unroll
    for (int q = 0; q < 8; ++q) issue_b1(xb, q);
  };

  float ra0[16], rb0[32], ra1[16], rb1[32];
  const int first = ibeg + slot_wg;
  set_load(first);
  issue_all(ra0, rb0); next_load();
  __builtin_amdgcn_sched_barrier(0);
  issue_all(ra1, rb1); next_load();
  __builtin_amdgcn_sched_barrier(0);

  // ---- scale exponents (as pw2_gemm_kernel)
  {
    unsigned ma = 0u, mb = 0u;
    if (a.rows_a == 0 && a.amax_a)
      for (int i = t; i < a.n_amax_a; i += k3Threads) ma = max(ma, __float_as_uint(a.amax_a[i]) & 0x7fffffffu);
    if (!(BKM && a.rows_b > 0) && a.amax_b)
      for (int i = t; i < a.n_amax_b; i += k3Threads) mb = max(mb, __float_as_uint(a.amax_b[i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      ma = max(ma, (unsigned)__shfl_xor((int)ma, o, 64));
      mb = max(mb, (unsigned)__shfl_xor((int)mb, o, 64));
    }
    if (lane == 0) { scr[w] = ma; scr[8 + w] = mb; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) { ma = max(ma, scr[i]); mb = max(mb, scr[8 + i]); }
    const int ea = pw_scale_exp(ma), eb = pw_scale_exp(mb);
    if (a.rows_a > 0) {
      const int nb = a.n_amax_a / a.rows_a;
      for (int row = t; row < a.M; row += k3Threads) {
        unsigned m = 0u;
        for (int j = 0; j < nb; ++j) m = max(m, __float_as_uint(a.amax_a[(size_t)j * a.rows_a + row]) & 0x7fffffffu);
        tab[row] = pw_scale_exp(m);
      }
    } else {
      for (int row = t; row < a.M; row += k3Threads) tab[row] = ea;
    }
    if (BKM) {
      if (a.rows_b > 0) {
        const int nb = a.n_amax_b / a.rows_b;
        for (int row = t; row < a.N; row += k3Threads) {
          unsigned m = 0u;
          for (int j = 0; j < nb; ++j) m = max(m, __float_as_uint(a.amax_b[(size_t)j * a.rows_b + row]) & 0x7fffffffu);
          tab[a.M + row] = pw_scale_exp(m);
        }
      } else {
        for (int row = t; row < a.N; row += k3Threads) tab[a.M + row] = eb;
      }
    } else if (t == 0) {
      tab[a.M] = eb;
    }
    __syncthreads();
  }

  // ---- split phase
  Pw2Pos S;
  float sa = 1.f, sb0 = 1.f, sb1 = 1.f;
  int skk = 0, skend = 0;
  auto set_split = [&](int id) {
    const Pw2Item it = pw2_item(a, id);
    S.id = id; S.kt = 0; S.T = it.T; S.alive = true;
    skk = it.kbeg; skend = it.kend;
    sa = ldexpf(1.f, tab[min(it.m0 + (t >> 1), a.M - 1)]);
    if constexpr (BKM) {
      sb0 = ldexpf(1.f, tab[a.M + min(it.n0 + (t >> 1), a.N - 1)]);
      sb1 = ldexpf(1.f, tab[a.M + min(it.n0 + 128 + (t >> 1), a.N - 1)]);
    } else {
      sb0 = ldexpf(1.f, tab[a.M]);
    }
  };
  auto next_split = [&]() {
    if (!S.alive) return;
    if (S.kt + 1 == S.T) {
      const int id = S.id + istride;
      if (id < iend) set_split(id); else S.alive = false;
      return;
    }
    ++S.kt; skk += k2BK;
  };
  auto mask_partial = [&](float (&xa)[16], float (&xb)[32]) {
    const int rem = skend - skk;
    if (rem >= k2BK) return;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (16 * kh + 4 * q >= rem) { xa[4 * q] = 0.f; xa[4 * q + 1] = 0.f; xa[4 * q + 2] = 0.f; xa[4 * q + 3] = 0.f; }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool out = BKM ? (16 * kh + 4 * (q & 3) >= rem) : (8 * w + 4 * (lane >> 5) + (q & 3) >= rem);
      if (out) { xb[4 * q] = 0.f; xb[4 * q + 1] = 0.f; xb[4 * q + 2] = 0.f; xb[4 * q + 3] = 0.f; }
    }
  };
  // A: k8 group gi of the thread's two (xa[8 gi ..]): units (2i, 2i+1); p = pair 0 / 1 of the group
  auto pair_a = [&](const float (&xa)[16], int gi, int p, unsigned (&h)[4], unsigned (&l)[4]) {
    const float* v = xa + 8 * gi + 4 * p;
    pw_split2x2(v[0], v[1], sa, v[2], v[3], sa, h[2 * p], l[2 * p], h[2 * p + 1], l[2 * p + 1]);
  };
  auto write_a = [&](_Float16* st, int gi, const unsigned (&h)[4], const unsigned (&l)[4]) {
    const int o = pw_slot(t >> 1, 2 * kh + gi);
    *(uint4*)(st + o) = make_uint4(h[0], h[1], h[2], h[3]);
    *(uint4*)(st + k2ImgA + o) = make_uint4(l[0], l[1], l[2], l[3]);
  };
  // B k-contiguous: group gi = 2 i + half of the thread's four (xb[8 gi ..]), as A.  B row-contiguous: column half nh (xb[16 nh ..]:
  // xb[16 nh + 4 j + c] = (k row j, column c)), pair p = (columns 2p', k pair) as pw2_gemm_kernel: second = p >> 1, p' = p & 1.
  auto pair_b = [&](const float (&xb)[32], int part, int p, unsigned (&h)[8], unsigned (&l)[8]) {
    if constexpr (BKM) {
      const float* v = xb + 8 * part + 4 * p;
      const float sc = (part >> 1) ? sb1 : sb0;
      pw_split2x2(v[0], v[1], sc, v[2], v[3], sc, h[2 * p], l[2 * p], h[2 * p + 1], l[2 * p + 1]);
    } else {
      const float* v = xb + 16 * part;
      const int second = p >> 1, c = 2 * (p & 1), u = 2 * c + second;
      pw_split2x2(v[second * 8 + c], v[second * 8 + 4 + c], sb0, v[second * 8 + c + 1], v[second * 8 + 4 + c + 1], sb0,
                  h[u], l[u], h[u + 2], l[u + 2]);
    }
  };
  // B k-contiguous: write group `part` (h / l [0..3]).  B row-contiguous: the permutes and the four cells of column half `part`
  // (h / l [0..7]), in two pieces: which = 0 the h image, 1 the l image.
  auto write_b = [&](_Float16* st, int part, int which, const unsigned (&h)[8], const unsigned (&l)[8]) {
    _Float16* bh = st + 2 * k2ImgA;
    _Float16* bl = bh + k2ImgB;
    if constexpr (BKM) {
      if (which == 0) {
        const int o = pw_slot(128 * (part >> 1) + (t >> 1), 2 * kh + (part & 1));
        *(uint4*)(bh + o) = make_uint4(h[0], h[1], h[2], h[3]);
        *(uint4*)(bl + o) = make_uint4(l[0], l[1], l[2], l[3]);
      }
    } else {
      const unsigned (&src)[8] = which ? l : h;
      _Float16* img = which ? bl : bh;
      unsigned c8[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const auto sw = __builtin_amdgcn_permlane32_swap(src[i], src[4 + i], false, false);
        c8[i] = sw[0]; c8[4 + i] = sw[1];
      }
      const int n = 128 * part + 4 * (lane & 31) + 2 * (lane >> 5);
      const int o0 = pw2_cell(w, n) * 8, o1 = pw2_cell(w, n + 1) * 8;
      *(uint4*)(img + o0) = make_uint4(c8[0], c8[1], c8[4], c8[5]);
      *(uint4*)(img + o1) = make_uint4(c8[2], c8[3], c8[6], c8[7]);
    }
  };

  // ---- fragments: A rows 64 wm + 32 i + r; B rows / columns 128 wn + 32 j + r; k8 group 2 ks + hh
  const int swz = (r >> 2) & 3;
  const int fa = (64 * wm + r) * k2BK, fbk = (128 * wn + r) * k2BK;
  int fbn[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) fbn[j] = pw2_cell(hh, 128 * wn + 32 * j + r) * 8;
  struct FragA { pw_h8 h[2], l[2]; };
  struct FragB { pw_h8 h, l; };
  auto frag_a1 = [&](const _Float16* st, int ks, int idx, FragA& f) {     // idx: 0, 1 = h of rows i; 2, 3 = l
    const int i = idx & 1;
    const int o = fa + 32 * i * k2BK + (((2 * ks + hh) ^ swz) << 3);
    if (idx < 2) f.h[i] = *(const pw_h8*)(st + o);
    else f.l[i] = *(const pw_h8*)(st + k2ImgA + o);
  };
  auto frag_b1 = [&](const _Float16* st, int ks, int j, int which, FragB& f) {
    const _Float16* bh = st + 2 * k2ImgA;
    int o;
    if constexpr (BKM) o = fbk + 32 * j * k2BK + (((2 * ks + hh) ^ swz) << 3);
    else o = fbn[j] + ks * 4096;
    if (which == 0) f.h = *(const pw_h8*)(bh + o);
    else f.l = *(const pw_h8*)(bh + k2ImgB + o);
  };

  pw_acc acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

  Pw2Pos Cc;
  auto set_comp = [&](int id) {
    const Pw2Item it = pw2_item(a, id);
    Cc.id = id; Cc.kt = 0; Cc.T = it.T; Cc.alive = true;
  };
  // one item's output: acc * 2^-(ea + eb), 16 bytes per store (rows of the accumulator = B rows: four consecutive columns of C)
  auto epilogue = [&]() {
    const Pw2Item it = pw2_item(a, Cc.id);
    float* C = a.C + (size_t)it.z * a.c_zs;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = it.m0 + 64 * wm + 32 * i + r;
      const int ea = tab[min(m, a.M - 1)];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int n = it.n0 + 128 * wn + 32 * j + 8 * u + 4 * hh;
          int e0, e1, e2, e3;
          if constexpr (BKM) {
            const int nn = min(n, a.N - 4);
            e0 = tab[a.M + nn]; e1 = tab[a.M + nn + 1]; e2 = tab[a.M + nn + 2]; e3 = tab[a.M + nn + 3];
          } else {
            e0 = e1 = e2 = e3 = tab[a.M];
          }
          float4 v;
          v.x = ldexpf(acc[j][i][4 * u], -(ea + e0));
          v.y = ldexpf(acc[j][i][4 * u + 1], -(ea + e1));
          v.z = ldexpf(acc[j][i][4 * u + 2], -(ea + e2));
          v.w = ldexpf(acc[j][i][4 * u + 3], -(ea + e3));
          if (m < a.M && n < a.N) *(float4*)(C + (size_t)m * a.ldc + n) = v;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
  };

  // ---- prologue: steps 0 and 1 into stages 0 and 1, the loads of steps 2 and 3 in flight
  set_split(first); set_comp(first);
  auto split_all = [&](float (&xa)[16], float (&xb)[32], _Float16* st) {
    mask_partial(xa, xb);
    unsigned hA[4], lA[4], hB[8], lB[8];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      pair_a(xa, gi, 0, hA, lA); pair_a(xa, gi, 1, hA, lA);
      write_a(st, gi, hA, lA);
    }
    if constexpr (BKM) {
#pragma unroll
      for (int part = 0; part < 4; ++part) {
        pair_b(xb, part, 0, hB, lB); pair_b(xb, part, 1, hB, lB);
        write_b(st, part, 0, hB, lB);
      }
    } else {
#pragma unroll
      for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int p = 0; p < 4; ++p) pair_b(xb, part, p, hB, lB);
        write_b(st, part, 0, hB, lB);
        write_b(st, part, 1, hB, lB);
      }
    }
    next_split();
  };
  split_all(ra0, rb0, pw_lds);
  __builtin_amdgcn_sched_barrier(0);
  issue_all(ra0, rb0); next_load();
  __builtin_amdgcn_sched_barrier(0);
  split_all(ra1, rb1, pw_lds + k2Stage);
  __builtin_amdgcn_sched_barrier(0);
  issue_all(ra1, rb1); next_load();
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  FragA fa0, fa1;          // A fragments of the current / the next k16 slice
  FragB fb0, fb1;          // B fragments of the current / the next 32-column block
#pragma unroll
  for (int idx = 0; idx < 4; ++idx) frag_a1(pw_lds, 0, idx, fa0);
  frag_b1(pw_lds, 0, 0, 0, fb0);
  frag_b1(pw_lds, 0, 0, 1, fb0);
  __builtin_amdgcn_s_waitcnt(0xc07f);

  int cs = 0;
  // Two MFMAs (rows i = 0, 1 of the A tile against one B block), fenced so that the fillers stay in their gaps
#define PW3_MM(J, AH, BF)                                                                          \
  do {                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    acc[J][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(BF, AH[0], acc[J][0], 0, 0, 0);             \
    acc[J][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(BF, AH[1], acc[J][1], 0, 0, 0);             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)

  auto step = [&](float (&xa)[16], float (&xb)[32]) {
    const _Float16* st = pw_lds + cs * k2Stage;
    const int ws_i = cs >= 1 ? cs - 1 : 2, ns_i = cs == 2 ? 0 : cs + 1;
    _Float16* wst = pw_lds + ws_i * k2Stage;
    const _Float16* nst = pw_lds + ns_i * k2Stage;
    unsigned hA[4], lA[4], hB[8], lB[8];
    mask_partial(xa, xb);
    // the fillers of gap `s` (0 .. 23): split pairs, LDS writes, one global load at most; the fragment reads are placed by the
    // block loop below
    auto slot = [&](int s) {
      switch (s) {
        case 0: pair_a(xa, 0, 0, hA, lA); break;
        case 1: pair_a(xa, 0, 1, hA, lA); break;
        case 2: write_a(wst, 0, hA, lA); issue_a1(xa, 0); break;
        case 3: pair_a(xa, 1, 0, hA, lA); issue_a1(xa, 1); break;
        case 4: pair_a(xa, 1, 1, hA, lA); break;
        case 5: write_a(wst, 1, hA, lA); issue_a1(xa, 2); break;
        case 6: pair_b(xb, 0, 0, hB, lB); issue_a1(xa, 3); break;
        case 7: pair_b(xb, 0, 1, hB, lB); if (BKM) write_b(wst, 0, 0, hB, lB); break;
        case 8: if (BKM) pair_b(xb, 1, 0, hB, lB); else pair_b(xb, 0, 2, hB, lB); break;
        case 9: if (BKM) { pair_b(xb, 1, 1, hB, lB); write_b(wst, 1, 0, hB, lB); } else pair_b(xb, 0, 3, hB, lB); break;
        case 10: if (!BKM) write_b(wst, 0, 0, hB, lB); issue_b1(xb, 0); break;
        case 11: if (!BKM) write_b(wst, 0, 1, hB, lB); issue_b1(xb, 1); break;
        case 12: if (BKM) pair_b(xb, 2, 0, hB, lB); else pair_b(xb, 1, 0, hB, lB); issue_b1(xb, 2); break;
        case 13: if (BKM) { pair_b(xb, 2, 1, hB, lB); write_b(wst, 2, 0, hB, lB); } else pair_b(xb, 1, 1, hB, lB); issue_b1(xb, 3); break;
        case 14: if (BKM) pair_b(xb, 3, 0, hB, lB); else pair_b(xb, 1, 2, hB, lB); break;
        case 15: if (BKM) { pair_b(xb, 3, 1, hB, lB); write_b(wst, 3, 0, hB, lB); } else pair_b(xb, 1, 3, hB, lB); break;
        case 16: if (!BKM) write_b(wst, 1, 0, hB, lB); issue_b1(xb, 4); break;
        case 17: if (!BKM) write_b(wst, 1, 1, hB, lB); issue_b1(xb, 5); break;
        case 18: issue_b1(xb, 6); break;
        case 19: issue_b1(xb, 7); next_load(); next_split(); break;
        default: break;
      }
    };
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      FragA& fc = ks ? fa1 : fa0;          // this slice's A fragments
      FragA& fn = ks ? fa0 : fa1;          // the next slice's (ks = 1: the next step's first)
      const _Float16* nxt_st = ks ? nst : st;
      const int nxt_ks = ks ? 0 : 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        FragB& bc = (j & 1) ? fb1 : fb0;   // this block's B fragments
        FragB& bn = (j & 1) ? fb0 : fb1;   // the next block's
        const int s0 = (ks * 4 + j) * 3;
        // the next block's B fragments (the next slice's / step's first block after the last one) and one A fragment of the
        // next slice per block
        const _Float16* bst = j < 3 ? st : nxt_st;
        const int bks = j < 3 ? ks : nxt_ks, bj = j < 3 ? j + 1 : 0;
        frag_b1(bst, bks, bj, 0, bn);
        PW3_MM(j, fc.h, bc.h);
        slot(s0);
        frag_b1(bst, bks, bj, 1, bn);
        PW3_MM(j, fc.h, bc.l);
        slot(s0 + 1);
        frag_a1(nxt_st, nxt_ks, j, fn);
        PW3_MM(j, fc.l, bc.h);
        slot(s0 + 2);
      }
    }
    __syncthreads();
    cs = ns_i;
    if (Cc.alive && ++Cc.kt == Cc.T) {
      epilogue();
      const int id = Cc.id + istride;
      if (id < iend) set_comp(id); else Cc.alive = false;
    }
  };
  do {
    step(ra0, rb0);
    step(ra1, rb1);
  } while (Cc.alive);
#undef PW3_MM
}
