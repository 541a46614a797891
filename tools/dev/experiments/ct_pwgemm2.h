// Pointwise-convolution GEMM, second kernel: the same split-f16 arithmetic as pw_gemm_kernel (ct_pwgemm.hip: two f16 terms
// per fp32 operand element, three MFMA terms, per-row / per-tensor power-of-two scales), re-tiled for what that kernel's
// counters showed (profiles/r3_pw_gemm_counters.txt, r4_pw_gemm_ablations.txt): its LDS array was as busy as its matrix pipes
// (one fragment read per MFMA), every tile paid its own prologue (maxima fold, first HBM round trip) and epilogue (64 KiB of
// stores) with nothing to overlap them — all 512 resident workgroups are in the same phase — and the split cost 2.7 vector
// instructions per MFMA.
//
//   tile        128 (M) x 256 (N) per 512-thread workgroup, eight waves as 2 x 4, 64 x 64 each (four 32x32 accumulators):
//               8 fragment reads per 12 MFMAs (was 6 per 6), 2.0 split instructions per MFMA, 0.25 global loads per MFMA
//   persistent  one workgroup per CU walks its list of (tile, k chunk) items as ONE software pipeline: the loads of K-step
//               s+4 go out and K-step s+2 is split into LDS while the MFMAs of step s run, across item boundaries, so only
//               the accumulators' stores sit between two items' MFMAs; an XCD's workgroups take a contiguous range of items
//               (the M tiles of one operand panel side by side on one L2)
//   LDS         three stages of [A_h | A_l | B_h | B_l] (48 KiB each): stage s+2 is written while stage s is read, and the
//               first fragments of step s+1 are fetched BEFORE the barrier that ends step s (they were published by the one
//               before), so the matrix pipes do not wait out an LDS round trip after every barrier
//   B operand   row-contiguous x / g_y (forward, data gradient): a wave loads eight k rows x 128 columns (lanes 0-31 rows
//               0-3, lanes 32-63 rows 4-7, 512-byte runs); after the split one v_permlane32_swap per dword gives every lane
//               the whole k octet of two columns, i.e. two finished 16-byte fragment cells, so a fragment is ONE
//               ds_read_b128 (the first kernel's four dword reads move a quarter of the bytes per LDS cycle).  Cells are
//               placed so that the eight-lane groups of ds_write_b128 and the sixteen-lane groups of ds_read_b128 each
//               cover distinct banks (pw2_cell)
//   output      the MFMA takes the B tile as its row operand: a lane then holds four consecutive columns of one output row
//               and stores 16 bytes at a time (16 stores per lane and tile instead of 64)
//   scales      one table of the exponents of all M rows (and of all N rows for the weight gradient) in LDS, folded once per
//               workgroup, not per tile
#pragma once

constexpr int k2TM = 128, k2TN = 256, k2BK = 32, k2Threads = 512;
constexpr int k2ImgA = k2TM * k2BK, k2ImgB = k2TN * k2BK;          // halves per image
constexpr int k2Stage = 2 * k2ImgA + 2 * k2ImgB;                   // halves per stage: A_h, A_l, B_h, B_l
constexpr int k2Stages = 3;
constexpr int k2TabMax = 2816;                                     // exponent table entries: M + (weight gradient: N)
constexpr int k2LdsBytes = k2Stages * k2Stage * 2 + k2TabMax * 4 + 1024;
static_assert(k2LdsBytes <= 160 * 1024, "LDS");

struct Pw2Item { int m0, n0, z, cloud, kbeg, kend, T; };

__device__ __forceinline__ Pw2Item pw2_item(const PwArgs& a, int id) {
  Pw2Item it;
  const int mt = id % a.tilesM, rest = id / a.tilesM, nt = rest % a.tilesN;
  it.z = rest / a.tilesN;
  it.cloud = it.z / a.ksplit;
  const int chunk = it.z - it.cloud * a.ksplit;
  it.kbeg = chunk * a.Kc;
  it.kend = min(a.K, it.kbeg + a.Kc);
  it.T = (it.kend - it.kbeg + k2BK - 1) / k2BK;
  it.m0 = mt * k2TM;
  it.n0 = nt * k2TN;
  return it;
}

// position of a pipeline phase in the workgroup's item list
struct Pw2Pos {
  int id, kt, T;
  bool alive;
};

// 16-byte cell of column n (0..255), k octet g (0..3) in a row-contiguous operand's image, in cells: the four columns of a
// staging lane go to four 64-cell blocks (a ds_write_b128 group = eight lanes = eight consecutive cells of one block), and
// block c is rotated by 4 c cells so that the 16 lanes ds_read_b128 serves per LDS cycle (MI355X_MICROARCH.md LDS table:
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...) fall on 16 distinct cells modulo 16 (256 bytes = all 64 banks)
__device__ __forceinline__ int pw2_cell(int g, int n) {
  const int c = n & 3;
  return g * 256 + c * 64 + (((n >> 2) + 4 * c) & 63);
}

template <bool BKM>   // B operand k-contiguous (weight gradient) or row-contiguous (forward / data gradient)
__global__ void __launch_bounds__(k2Threads, 2) pw2_gemm_kernel(PwArgs a) {
  extern __shared__ __attribute__((aligned(16))) _Float16 pw_lds[];
  int* tab = (int*)((char*)pw_lds + k2Stages * k2Stage * 2);      // exponents: [M] A rows | [N] B rows (BKM) or [1]
  unsigned* scr = (unsigned*)(tab + k2TabMax);
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), wm = w >> 2, wn = w & 3, r = lane & 31, hh = lane >> 5;
#ifdef PW_STAMP
  const unsigned long long tkernel = __builtin_amdgcn_s_memtime();
#endif

  // ---- this workgroup's items: XCD x (blockIdx % 8) owns a contiguous range, its workgroups take it round robin
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int items = a.tilesM * a.tilesN * a.Z, ibase = items >> 3, irem = items & 7;
  const int ibeg = xcd * ibase + min(xcd, irem), iend = ibeg + ibase + (xcd < irem ? 1 : 0);
  const int istride = (nwg - xcd + 7) >> 3;
  if (ibeg + slot >= iend) return;

  const size_t lda4 = (size_t)a.lda * 4, ldb4 = (size_t)a.ldb * 4;
  const int g4 = t & 3;                                  // k8 group of a k-contiguous operand's staging thread

  // ---- load phase: position, wave-uniform bases of the current K-step, lane offsets (whole steps | the chunk's partial last)
  Pw2Pos L;
  const char* ua = nullptr;
  const char* ub = nullptr;
  unsigned offa = 0, offb0 = 0, offb1 = 0;
  unsigned poffa[2], poffb[4];
  int lkk = 0, lkend = 0;                                // first k of the load phase's step, end of its chunk
  auto set_load = [&](int id) {
    const Pw2Item it = pw2_item(a, id);
    L.id = id; L.kt = 0; L.T = it.T; L.alive = true;
    lkk = it.kbeg; lkend = it.kend;
    const int rem = it.kend - it.kbeg - (it.T - 1) * k2BK;          // k of the last step: 4 .. 32
    ua = (const char*)(a.A + (size_t)it.cloud * a.a_bs + it.kbeg);
    const unsigned rowa = (unsigned)min(it.m0 + (t >> 2), a.M - 1) * (unsigned)lda4;
    offa = rowa + 32u * g4;
#pragma unroll
    for (int q = 0; q < 2; ++q) poffa[q] = rowa + 4u * (unsigned)min(8 * g4 + 4 * q, rem - 4);
    if constexpr (BKM) {
      ub = (const char*)(a.B + (size_t)it.cloud * a.b_bs + it.kbeg);
      const unsigned rb0 = (unsigned)min(it.n0 + (t >> 2), a.N - 1) * (unsigned)ldb4;
      const unsigned rb1 = (unsigned)min(it.n0 + 128 + (t >> 2), a.N - 1) * (unsigned)ldb4;
      offb0 = rb0 + 32u * g4;
      offb1 = rb1 + 32u * g4;
#pragma unroll
      for (int q = 0; q < 4; ++q) poffb[q] = ((q >> 1) ? rb1 : rb0) + 4u * (unsigned)min(8 * g4 + 4 * (q & 1), rem - 4);
    } else {
      // wave w: k octet w >> 1, columns 128 (w & 1) + 4 (lane & 31) .. + 3, k rows 4 (lane >> 5) .. + 3 of the octet;
      // ub = the step's first row, the lane offset holds its rows
      ub = (const char*)(a.B + (size_t)it.cloud * a.b_bs) + (size_t)it.kbeg * ldb4;
      offb1 = (unsigned)min(it.n0 + 128 * (w & 1) + 4 * (lane & 31), a.N - 4) * 4u;
      offb0 = offb1 + (unsigned)(8 * (w >> 1) + 4 * (lane >> 5)) * (unsigned)ldb4;
#pragma unroll
      for (int q = 0; q < 4; ++q) poffb[q] = offb1 + (unsigned)min(8 * (w >> 1) + 4 * (lane >> 5) + q, rem - 1) * (unsigned)ldb4;
    }
  };
  // past the end of the list the phase stays on its last step: its loads go on (unused), so that the number of loads in
  // flight is the same on every path (at a join the compiler waits for the shortest queue)
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
  // The loads of the load phase's K-step into one register set, in three pieces (A | B rows 0, 1 | B rows 2, 3) that the step
  // places between its MFMA groups as the registers they overwrite are split: a burst of six fills the wave's queue and
  // stalls it (0.6k cycles per step, profiles/r4_pw_gemm_stamps.txt).  No branch — the partial last step of a chunk only
  // selects other lane offsets (k clamped into the chunk; what lies beyond is zeroed when the set is split).
  auto issue_a = [&](float (&xa)[8]) {
    const bool part = lkk + k2BK > lkend;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float4 v = *(const float4*)(ua + (part ? poffa[q] : offa + 16u * q));
      xa[4 * q] = v.x; xa[4 * q + 1] = v.y; xa[4 * q + 2] = v.z; xa[4 * q + 3] = v.w;
    }
  };
  auto issue_b = [&](float (&xb)[16], int q0) {
    const bool part = lkk + k2BK > lkend;
#pragma unroll
    for (int q = q0; q < q0 + 2; ++q) {
      unsigned o;
      if constexpr (BKM) o = ((q >> 1) ? offb1 : offb0) + 16u * (q & 1);
      else o = offb0 + (unsigned)q * (unsigned)ldb4;
      const float4 v = *(const float4*)(ub + (part ? poffb[q] : o));
      xb[4 * q] = v.x; xb[4 * q + 1] = v.y; xb[4 * q + 2] = v.z; xb[4 * q + 3] = v.w;
    }
  };
  auto issue = [&](float (&xa)[8], float (&xb)[16]) { issue_a(xa); issue_b(xb, 0); issue_b(xb, 2); };
  // one 16-byte load at a time (the step spreads the six over its MFMA half-groups: a wave that issues two back to back waits
  // for the CU's vector-memory path behind the other seven waves' pairs)
  auto issue_a1 = [&](float (&xa)[8], int q) {
    const bool part = lkk + k2BK > lkend;
    const float4 v = *(const float4*)(ua + (part ? poffa[q] : offa + 16u * q));
    xa[4 * q] = v.x; xa[4 * q + 1] = v.y; xa[4 * q + 2] = v.z; xa[4 * q + 3] = v.w;
  };
  auto issue_b1 = [&](float (&xb)[16], int q) {
    const bool part = lkk + k2BK > lkend;
    unsigned o;
    if constexpr (BKM) o = ((q >> 1) ? offb1 : offb0) + 16u * (q & 1);
    else o = offb0 + (unsigned)q * (unsigned)ldb4;
    const float4 v = *(const float4*)(ub + (part ? poffb[q] : o));
    xb[4 * q] = v.x; xb[4 * q + 1] = v.y; xb[4 * q + 2] = v.z; xb[4 * q + 3] = v.w;
  };

  // the first two K-steps' loads go out before anything else: the fold of the maxima below hides behind their round trip
  float ra0[8], rb0[16], ra1[8], rb1[16];
  const int first = ibeg + slot;
  set_load(first);
  issue(ra0, rb0); next_load();
  __builtin_amdgcn_sched_barrier(0);
  issue(ra1, rb1); next_load();
  __builtin_amdgcn_sched_barrier(0);

  // ---- scale exponents of every row this launch can touch (pw_gemm_kernel's rule: s * max in [2^13, 2^14))
  {
    unsigned ma = 0u, mb = 0u;
    if (a.rows_a == 0 && a.amax_a)
      for (int i = t; i < a.n_amax_a; i += k2Threads) ma = max(ma, __float_as_uint(a.amax_a[i]) & 0x7fffffffu);
    if (!(BKM && a.rows_b > 0) && a.amax_b)
      for (int i = t; i < a.n_amax_b; i += k2Threads) mb = max(mb, __float_as_uint(a.amax_b[i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      ma = max(ma, (unsigned)__shfl_xor((int)ma, o, 64));
      mb = max(mb, (unsigned)__shfl_xor((int)mb, o, 64));
    }
    if (lane == 0) { scr[w] = ma; scr[8 + w] = mb; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) { ma = max(ma, scr[i]); mb = max(mb, scr[8 + i]); }
    const int ea = pw_scale_exp(ma), eb = pw_scale_exp(mb);
    if (a.rows_a > 0) {
      const int nb = a.n_amax_a / a.rows_a;
      for (int row = t; row < a.M; row += k2Threads) {
        unsigned m = 0u;
        for (int j = 0; j < nb; ++j) m = max(m, __float_as_uint(a.amax_a[(size_t)j * a.rows_a + row]) & 0x7fffffffu);
        tab[row] = pw_scale_exp(m);
      }
    } else {
      for (int row = t; row < a.M; row += k2Threads) tab[row] = ea;
    }
    if (BKM) {
      if (a.rows_b > 0) {
        const int nb = a.n_amax_b / a.rows_b;
        for (int row = t; row < a.N; row += k2Threads) {
          unsigned m = 0u;
          for (int j = 0; j < nb; ++j) m = max(m, __float_as_uint(a.amax_b[(size_t)j * a.rows_b + row]) & 0x7fffffffu);
          tab[a.M + row] = pw_scale_exp(m);
        }
      } else {
        for (int row = t; row < a.N; row += k2Threads) tab[a.M + row] = eb;
      }
    } else if (t == 0) {
      tab[a.M] = eb;
    }
    __syncthreads();
  }

  // ---- split phase: position, the staging thread's scales, valid k of a partial step
  Pw2Pos S;
  float sa = 1.f, sb0 = 1.f, sb1 = 1.f;
  int skk = 0, skend = 0;
  auto set_split = [&](int id) {
    const Pw2Item it = pw2_item(a, id);
    S.id = id; S.kt = 0; S.T = it.T; S.alive = true;
    skk = it.kbeg; skend = it.kend;
    sa = ldexpf(1.f, tab[min(it.m0 + (t >> 2), a.M - 1)]);
    if constexpr (BKM) {
      sb0 = ldexpf(1.f, tab[a.M + min(it.n0 + (t >> 2), a.N - 1)]);
      sb1 = ldexpf(1.f, tab[a.M + min(it.n0 + 128 + (t >> 2), a.N - 1)]);
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
  // zero what a partial step's clamped loads fetched beyond the chunk
  auto mask_partial = [&](float (&xa)[8], float (&xb)[16]) {
    const int rem = skend - skk;
    if (rem >= k2BK) return;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (8 * g4 + 4 * q >= rem) { xa[4 * q] = 0.f; xa[4 * q + 1] = 0.f; xa[4 * q + 2] = 0.f; xa[4 * q + 3] = 0.f; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool out = BKM ? (8 * g4 + 4 * (q & 1) >= rem) : (8 * (w >> 1) + 4 * (lane >> 5) + q >= rem);
      if (out) { xb[4 * q] = 0.f; xb[4 * q + 1] = 0.f; xb[4 * q + 2] = 0.f; xb[4 * q + 3] = 0.f; }
    }
  };
  // Split units (one pw_split2 each) and the stores of a register set into stage `st`.
  //   A, and B when k-contiguous: unit i of a row = (k 2i, 2i+1) of the thread's eight; image [row][k] (pw_slot)
  //   B row-contiguous: xb[4 j + c] = (k row j, column c); unit u = (c = u >> 1, k pair u & 1)
  // two units at a time (pw_split2x2): A (i, i+1); B: the pair p of the half `second` — the first half reads only the
  // registers of loads 0, 1 (B k-contiguous: row 0 = units 0-3; row-contiguous: k rows 0, 1 = units 0, 2, 4, 6)
  auto pair_a = [&](const float (&xa)[8], int i, unsigned (&h)[4], unsigned (&l)[4]) {
    pw_split2x2(xa[2 * i], xa[2 * i + 1], sa, xa[2 * i + 2], xa[2 * i + 3], sa, h[i], l[i], h[i + 1], l[i + 1]);
  };
  auto pair_b = [&](const float (&xb)[16], int second, int p, unsigned (&h)[8], unsigned (&l)[8]) {
    if constexpr (BKM) {
      const int u = 4 * second + 2 * p;
      const float sc = second ? sb1 : sb0;
      pw_split2x2(xb[2 * u], xb[2 * u + 1], sc, xb[2 * u + 2], xb[2 * u + 3], sc, h[u], l[u], h[u + 1], l[u + 1]);
    } else {
      const int c = 2 * p, u = 2 * c + second;         // units (c, second), (c + 1, second)
      pw_split2x2(xb[second * 8 + c], xb[second * 8 + 4 + c], sb0, xb[second * 8 + c + 1], xb[second * 8 + 4 + c + 1], sb0,
                  h[u], l[u], h[u + 2], l[u + 2]);
    }
  };
  auto write_a = [&](_Float16* st, const unsigned (&h)[4], const unsigned (&l)[4]) {
    const int o = pw_slot(t >> 2, g4);
    *(uint4*)(st + o) = make_uint4(h[0], h[1], h[2], h[3]);
    *(uint4*)(st + k2ImgA + o) = make_uint4(l[0], l[1], l[2], l[3]);
  };
  auto write_b = [&](_Float16* st, const unsigned (&h)[8], const unsigned (&l)[8]) {
    _Float16* bh = st + 2 * k2ImgA;
    _Float16* bl = bh + k2ImgB;
    if constexpr (BKM) {
      const int o0 = pw_slot(t >> 2, g4), o1 = pw_slot(128 + (t >> 2), g4);
      *(uint4*)(bh + o0) = make_uint4(h[0], h[1], h[2], h[3]);
      *(uint4*)(bl + o0) = make_uint4(l[0], l[1], l[2], l[3]);
      *(uint4*)(bh + o1) = make_uint4(h[4], h[5], h[6], h[7]);
      *(uint4*)(bl + o1) = make_uint4(l[4], l[5], l[6], l[7]);
    } else {
      // unit 2 c + p = (column c, k pair p of the lane's four rows).  Swapping columns (0, 2) and (1, 3) between the wave's
      // halves: lanes 0-31 then hold rows 0-7 of columns 0, 1, lanes 32-63 rows 0-7 of columns 2, 3
      unsigned ch[8], cl[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const auto sh = __builtin_amdgcn_permlane32_swap(h[i], h[4 + i], false, false);
        const auto sl = __builtin_amdgcn_permlane32_swap(l[i], l[4 + i], false, false);
        ch[i] = sh[0]; ch[4 + i] = sh[1];
        cl[i] = sl[0]; cl[4 + i] = sl[1];
      }
      const int n = 128 * (w & 1) + 4 * (lane & 31) + 2 * (lane >> 5);
      const int o0 = pw2_cell(w >> 1, n) * 8, o1 = pw2_cell(w >> 1, n + 1) * 8;      // halves
      *(uint4*)(bh + o0) = make_uint4(ch[0], ch[1], ch[4], ch[5]);
      *(uint4*)(bh + o1) = make_uint4(ch[2], ch[3], ch[6], ch[7]);
      *(uint4*)(bl + o0) = make_uint4(cl[0], cl[1], cl[4], cl[5]);
      *(uint4*)(bl + o1) = make_uint4(cl[2], cl[3], cl[6], cl[7]);
    }
  };

  // ---- fragments of stage `st`, k16 slice ks: A rows 64 wm + 32 i + r, B rows 64 wn + 32 j + r, k8 group 2 ks + hh
  const int swz = (r >> 2) & 3;
  const int fa = (64 * wm + r) * k2BK, fbk = (64 * wn + r) * k2BK;                       // halves
  const int fbn0 = pw2_cell(hh, 64 * wn + r) * 8, fbn1 = pw2_cell(hh, 64 * wn + 32 + r) * 8;                   // halves
  struct Frags { pw_h8 ah[2], al[2], bh[2], bl[2]; };
  auto frags = [&](const _Float16* st, int ks, Frags& f) {
    const int g = 2 * ks + hh;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = fa + 32 * i * k2BK + ((g ^ swz) << 3);
      f.ah[i] = *(const pw_h8*)(st + o);
      f.al[i] = *(const pw_h8*)(st + k2ImgA + o);
    }
    const _Float16* bh = st + 2 * k2ImgA;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (BKM) {
        const int o = fbk + 32 * j * k2BK + ((g ^ swz) << 3);
        f.bh[j] = *(const pw_h8*)(bh + o);
        f.bl[j] = *(const pw_h8*)(bh + k2ImgB + o);
      } else {
        const int o = (j ? fbn1 : fbn0) + ks * 4096;
        f.bh[j] = *(const pw_h8*)(bh + o);
        f.bl[j] = *(const pw_h8*)(bh + k2ImgB + o);
      }
    }
  };

  // accumulators [j: B rows 32 j][i: A rows 32 i]: D row = B row (4 consecutive per register quad), D column = A row (lane)
  pw_acc acc[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

  Pw2Pos Cc;
  auto set_comp = [&](int id) {
    const Pw2Item it = pw2_item(a, id);
    Cc.id = id; Cc.kt = 0; Cc.T = it.T; Cc.alive = true;
  };

  // one item's output: acc * 2^-(ea + eb), 16 bytes per store
  auto epilogue = [&]() {
    const Pw2Item it = pw2_item(a, Cc.id);
    float* C = a.C + (size_t)it.z * a.c_zs;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = it.m0 + 64 * wm + 32 * i + r;
      const int ea = tab[min(m, a.M - 1)];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int n = it.n0 + 64 * wn + 32 * j + 8 * u + 4 * hh;
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
#if defined(PW_ABL) && PW_ABL == 4
          if (m < a.M && n < a.N && v.x == 12345.678f) *(float4*)(C + (size_t)m * a.ldc + n) = v;
#else
          if (m < a.M && n < a.N) *(float4*)(C + (size_t)m * a.ldc + n) = v;
#endif
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
  };

  // ---- prologue: steps 0 and 1 (loaded above) into stages 0 and 1, the loads of steps 2 and 3 in flight
  set_split(first); set_comp(first);
  auto split_all = [&](float (&xa)[8], float (&xb)[16], _Float16* st) {
    mask_partial(xa, xb);
    unsigned hA[4], lA[4], hB[8], lB[8];
    pair_a(xa, 0, hA, lA); pair_a(xa, 2, hA, lA);
    write_a(st, hA, lA);
    pair_b(xb, 0, 0, hB, lB); pair_b(xb, 0, 1, hB, lB); pair_b(xb, 1, 0, hB, lB); pair_b(xb, 1, 1, hB, lB);
    write_b(st, hB, lB);
    next_split();
  };
  // (the order of the two refills is what the loop's counted waits rest on: pinned)
  split_all(ra0, rb0, pw_lds);
  __builtin_amdgcn_sched_barrier(0);
  issue(ra0, rb0); next_load();
  __builtin_amdgcn_sched_barrier(0);
  split_all(ra1, rb1, pw_lds + k2Stage);
  __builtin_amdgcn_sched_barrier(0);
  issue(ra1, rb1); next_load();
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  Frags f0, f1;
  frags(pw_lds, 0, f0);
  __builtin_amdgcn_s_waitcnt(0xc07f);

#define PW2_MF(F, X, Y)                                                                          \
  do {                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.Y[0], F.X[0], acc[0][0], 0, 0, 0);      \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.Y[0], F.X[1], acc[0][1], 0, 0, 0);      \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.Y[1], F.X[0], acc[1][0], 0, 0, 0);      \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.Y[1], F.X[1], acc[1][1], 0, 0, 0);      \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  } while (0)

#ifdef PW_STAMP
  // development builds: s_memtime at five points of a step, summed per wave after the barrier (tools/dev/pw2_stamp.py):
  // [0] top -> ks0 MFMAs issued, [1] -> ks1 groups 1-2 + splits issued, [2] -> loads + next fragments + last group issued,
  // [3] -> barrier passed, [4] barrier -> next top (item switch, epilogue); [5] steps, [6] kernel start -> loop, [7] whole kernel
  unsigned long long tst[5], tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
  const unsigned long long tloop = __builtin_amdgcn_s_memtime();
#define PW2_STAMP(i)                                 \
  do {                                               \
    __builtin_amdgcn_sched_barrier(0);               \
    tst[i] = __builtin_amdgcn_s_memtime();           \
    __builtin_amdgcn_sched_barrier(0);               \
  } while (0)
#define PW2_STAMP_SUM()                                                   \
  do {                                                                    \
    if (tprev) tsum[4] += tst[0] - tprev;                                 \
    for (int i_ = 0; i_ < 4; ++i_) tsum[i_] += tst[i_ + 1] - tst[i_];     \
    tsum[5] += 1;                                                         \
    tprev = tst[4];                                                       \
  } while (0)
#else
#define PW2_STAMP(i) ((void)0)
#define PW2_STAMP_SUM() ((void)0)
#endif
#define PW2_MH(F, X, Y, J)                                                                        \
  do {                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    acc[J][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.Y[J], F.X[0], acc[J][0], 0, 0, 0);      \
    acc[J][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.Y[J], F.X[1], acc[J][1], 0, 0, 0);      \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  } while (0)
  int cs = 0;                                        // LDS stage of the step being multiplied
  // step s: MFMAs on stage cs; K-step s+2 (register set xa / xb) is split into stage cs+2 between them, then the set is
  // refilled with K-step s+4; the first fragments of step s+1 (stage cs+1) are fetched before the barrier
  auto step = [&](float (&xa)[8], float (&xb)[16]) {
    const _Float16* st = pw_lds + cs * k2Stage;
    const int ws_i = cs >= 1 ? cs - 1 : 2, ns_i = cs == 2 ? 0 : cs + 1;
    _Float16* wst = pw_lds + ws_i * k2Stage;
    const _Float16* nst = pw_lds + ns_i * k2Stage;
    unsigned hA[4], lA[4], hB[8], lB[8];
    PW2_STAMP(0);
    frags(st, 1, f1);
    mask_partial(xa, xb);
    PW2_MH(f0, ah, bh, 0);
    pair_a(xa, 0, hA, lA);
    PW2_MH(f0, ah, bh, 1);
    pair_a(xa, 2, hA, lA);
    PW2_MH(f0, ah, bl, 0);
    write_a(wst, hA, lA);
    issue_a1(xa, 0);
    PW2_MH(f0, ah, bl, 1);
    pair_b(xb, 0, 0, hB, lB);
    issue_a1(xa, 1);
    PW2_MH(f0, al, bh, 0);
    pair_b(xb, 0, 1, hB, lB);
    PW2_MH(f0, al, bh, 1);
    issue_b1(xb, 0);
    PW2_STAMP(1);
    PW2_MH(f1, ah, bh, 0);
    pair_b(xb, 1, 0, hB, lB);
    issue_b1(xb, 1);
    PW2_MH(f1, ah, bh, 1);
    pair_b(xb, 1, 1, hB, lB);
    PW2_MH(f1, ah, bl, 0);
    issue_b1(xb, 2);
    write_b(wst, hB, lB);
    PW2_MH(f1, ah, bl, 1);
    issue_b1(xb, 3);
    next_load();
    next_split();
    PW2_STAMP(2);
    PW2_MH(f1, al, bh, 0);
    frags(nst, 0, f0);
    PW2_MH(f1, al, bh, 1);
    PW2_STAMP(3);
    __syncthreads();
    PW2_STAMP(4);
    PW2_STAMP_SUM();
    cs = ns_i;
    if (Cc.alive && ++Cc.kt == Cc.T) {
      epilogue();
      const int id = Cc.id + istride;
      if (id < iend) set_comp(id); else Cc.alive = false;
    }
  };
  // steps in pairs (one per register set) and ONE exit: with a second exit between the two the compiler's counted waits for
  // the loads degrade to waits for everything in flight.  A workgroup with an odd number of steps runs one more on stale
  // stages, into accumulators nobody stores.
  do {
    step(ra0, rb0);
    step(ra1, rb1);
  } while (Cc.alive);
#undef PW2_MF
#undef PW2_MH
#ifdef PW_STAMP
  if (a.dbg && lane == 0) {
    tsum[6] = tloop - tkernel;
    tsum[7] = __builtin_amdgcn_s_memtime() - tkernel;
    for (int i = 0; i < 8; ++i) a.dbg[((size_t)blockIdx.x * 8 + w) * 8 + i] = tsum[i];
  }
#endif
}
