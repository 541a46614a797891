/*
 * Second, independent restatement of the reference's auction EMD — TEST INFRASTRUCTURE ONLY
 * (never linked into or called by the product).
 *
 * oracle/emd_ref.c is a SEQUENTIAL restatement of emd_linear/emd_cuda.cu.  The reference itself is a set of parallel
 * CUDA kernels whose source cannot be compiled here (no nvcc, no NVIDIA GPU), so nothing ties emd_ref.c's
 * tie-breaks and merge order to what the parallel code yields.  This file simulates the reference's THREAD
 * DECOMPOSITION on the CPU, thread by thread, so that the two can be compared:
 *
 *   Bid (emd_cuda.cu:95-179)   grid (b, n/1024) x 1024 threads.  Per batch: unass_per_block = ceil(U / block_cnt),
 *        thread_per_unass = 1024 / unass_per_block (:108-109); thread t of a bidder scans, in every 2048-target
 *        batch, the contiguous range [t*delta, min((t+1)*delta, end_k)), delta = ceil(end_k / thread_per_unass)
 *        (:138-141), keeping (best, better, best_i) with strict '>' (:150-157) ACROSS the batches; thread 0 of the
 *        bidder then merges threads 1.. in order (:165-176: strict '>' on best, better = max(...)), writes the bid,
 *        the increment best - better + eps, and raises max_increments[best_i] by a float atomicMax implemented as a
 *        compare-and-swap loop (:10-20) — a max, whatever the arrival order.
 *   GetMax (:181-194)  EVERY unassigned bidder whose increment is within 1e-6 of its target's maximum stores its
 *        index into max_idx[target]: plain stores, the last writer wins — any of them.  `getmax_order` picks the
 *        winner among the candidates: 0 lowest index, 1 highest index (emd_ref.c's choice), 2 pseudo-random.
 *   Assign (:196-215)  the winner takes the target and evicts its owner; on the last iteration every bidder is
 *        assigned to its bid (races on assignment_inv / price there do not reach `dist`).
 *   The unassigned list (:30-93) is built with atomicAdd: its ORDER is arbitrary; `list_order` = 0 ascending,
 *        1 descending, 2 pseudo-random permutes it (it decides which thread block scans which bidder, nothing else).
 *
 * The distance expression `x2*x2 + y2*y2 + z2*z2` (:148) is compiled by nvcc with floating-point contraction on
 * by default; which products are fused is the compiler's choice.  `contraction` selects the variant:
 *   0 none: (x*x + y*y) + z*z, every product rounded      1 fma(z,z, fma(y,y, x*x))  (emd_ref.c, ct_emd.hip)
 *   2 fma(z,z, x*x + y*y)                                  3 fma(z,z, fma(x,x, y*y))
 * Counters returned in stats[]: [0] bidders whose best VALUE was reached by more than one target (an exact tie:
 * which target gets the bid then depends on the decomposition), [1] GetMax events with more than one candidate.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static unsigned lcg(unsigned* s) {
  *s = *s * 1664525u + 1013904223u;
  return *s >> 8;
}

static float dist2(float x2, float y2, float z2, int contraction) {
  switch (contraction) {
    case 0: {
      volatile float a = x2 * x2, b = y2 * y2, c = z2 * z2;
      volatile float s = a + b;
      return s + c;
    }
    case 2: {
      volatile float a = x2 * x2, b = y2 * y2;
      volatile float s = a + b;
      return fmaf(z2, z2, s);
    }
    case 3: {
      volatile float b = y2 * y2;
      return fmaf(z2, z2, fmaf(x2, x2, b));
    }
    default: {
      volatile float a = x2 * x2;
      return fmaf(z2, z2, fmaf(y2, y2, a));
    }
  }
}

int emd_sim_forward(const float* xyz1, const float* xyz2, float* dist, int* assignment, int B, int n, float eps,
                    int iters, int contraction, int getmax_order, int list_order, unsigned seed, long long* stats) {
  if (n % 1024 != 0 || B > 512 || B <= 0 || n <= 0) return -1;
  const int batch = 2048, block_size = 1024, block_cnt = n / 1024;
  float* price = (float*)calloc((size_t)n, sizeof(float));
  float* bid_inc = (float*)calloc((size_t)n, sizeof(float));
  float* max_inc = (float*)calloc((size_t)n, sizeof(float));
  int* ass_inv = (int*)malloc((size_t)n * sizeof(int));
  int* bid = (int*)calloc((size_t)n, sizeof(int));
  int* max_idx = (int*)calloc((size_t)n, sizeof(int));
  int* unass = (int*)malloc((size_t)n * sizeof(int));
  int* ncand = (int*)calloc((size_t)n, sizeof(int));
  float* t_best = (float*)malloc(block_size * sizeof(float));
  float* t_better = (float*)malloc(block_size * sizeof(float));
  int* t_best_i = (int*)malloc(block_size * sizeof(int));
  stats[0] = stats[1] = 0;
  unsigned rng = seed ? seed : 1u;
  for (int b = 0; b < B; ++b) {
    const float* p1 = xyz1 + (size_t)b * n * 3;
    const float* p2 = xyz2 + (size_t)b * n * 3;
    int* ass = assignment + (size_t)b * n;
    for (int j = 0; j < n; ++j) { ass[j] = -1; ass_inv[j] = -1; price[j] = 0.f; max_inc[j] = 0.f; max_idx[j] = 0; }
    for (int it = 0; it < iters; ++it) {
      const int last = (it == iters - 1);
      /* unassigned list, in an arbitrary order (calc_unass_idx: atomicAdd slots) */
      int U = 0;
      for (int j = 0; j < n; ++j) if (ass[j] == -1) unass[U++] = j;
      if (list_order == 1) {
        for (int a = 0, z = U - 1; a < z; ++a, --z) { int t = unass[a]; unass[a] = unass[z]; unass[z] = t; }
      } else if (list_order == 2) {
        for (int a = U - 1; a > 0; --a) { int r = (int)(lcg(&rng) % (unsigned)(a + 1)); int t = unass[a]; unass[a] = unass[r]; unass[r] = t; }
      }
      if (U == 0) continue;
      /* Bid: thread blocks (blockIdx.y) x threads, as launched */
      const int upb = (U + block_cnt - 1) / block_cnt;
      const int tpu = block_size / upb;
      for (int by = 0; by < block_cnt; ++by) {
        int utb = U - by * upb;
        if (utb > upb) utb = upb;
        if (utb < 0) utb = 0;
        for (int ub = 0; ub < utb; ++ub) {               /* bidder `ub` of this block: threads ub*tpu .. ub*tpu+tpu-1 */
          const int j = unass[upb * by + ub];
          const float x1 = p1[j * 3 + 0], y1 = p1[j * 3 + 1], z1 = p1[j * 3 + 2];
          for (int t = 0; t < tpu; ++t) { t_best[t] = -1e9f; t_better[t] = -1e9f; t_best_i[t] = -1; }
          for (int k2 = 0; k2 < n; k2 += batch) {
            const int end_k = (n < k2 + batch ? n : k2 + batch) - k2;
            const int delta = (end_k + tpu - 1) / tpu;
            for (int t = 0; t < tpu; ++t) {
              const int l = t * delta;
              int r = (t + 1) * delta;
              if (r > end_k) r = end_k;
              float best = t_best[t], better = t_better[t];
              int best_i = t_best_i[t];
              for (int k = l; k < r; ++k) {
                const float x2 = p2[(k2 + k) * 3 + 0] - x1, y2 = p2[(k2 + k) * 3 + 1] - y1, z2 = p2[(k2 + k) * 3 + 2] - z1;
                const float d = (float)(3.0 - (double)sqrtf(dist2(x2, y2, z2, contraction)) - (double)price[k2 + k]);
                if (d > best) { better = best; best = d; best_i = k + k2; }
                else if (d > better) better = d;
              }
              t_best[t] = best; t_better[t] = better; t_best_i[t] = best_i;
            }
          }
          /* merge by the bidder's thread 0 (:165-176) */
          float best = t_best[0], better = t_better[0];
          int best_i = t_best_i[0];
          for (int t = 1; t < tpu; ++t) {
            if (t_best[t] > best) { better = best > t_better[t] ? best : t_better[t]; best = t_best[t]; best_i = t_best_i[t]; }
            else better = better > t_best[t] ? better : t_best[t];
          }
          if (best == better) stats[0] += 1;               /* the best value was reached twice: an exact tie */
          bid[j] = best_i;
          bid_inc[j] = best - better + eps;
          if (bid_inc[j] > max_inc[best_i]) max_inc[best_i] = bid_inc[j];     /* float atomicMax: a max */
        }
      }
      /* GetMax: the candidates of a target race with plain stores; the survivor is picked by `getmax_order` */
      for (int u = 0; u < U; ++u) ncand[bid[unass[u]]] = 0;
      {
        /* ascending bidder index regardless of the list order */
        for (int j = 0; j < n; ++j) {
          if (ass[j] != -1) continue;
          const int t = bid[j];
          const float bi = bid_inc[j], mi = max_inc[t];
          if (bi - 1e-6 <= mi && mi <= bi + 1e-6) {
            ncand[t] += 1;
            if (ncand[t] == 1) max_idx[t] = j;
            else {
              if (ncand[t] == 2) stats[1] += 1;
              if (getmax_order == 1) max_idx[t] = j;                                   /* highest index wins */
              else if (getmax_order == 2 && lcg(&rng) % (unsigned)ncand[t] == 0) max_idx[t] = j;   /* uniform among candidates */
              /* getmax_order == 0: the first (lowest) stays */
            }
          }
        }
      }
      /* Assign */
      for (int u = 0; u < U; ++u) {
        const int j = unass[u];
        const int t = bid[j];
        if (last || max_idx[t] == j) {
          const int prev = ass_inv[t];
          if (!last && prev != -1) ass[prev] = -1;
          ass_inv[t] = j;
          ass[j] = t;
          price[t] += bid_inc[j];
          max_inc[t] = -1e9f;
        }
      }
    }
    for (int j = 0; j < n; ++j) {
      const int k = ass[j];
      const float dx = p1[j * 3 + 0] - p2[k * 3 + 0], dy = p1[j * 3 + 1] - p2[k * 3 + 1], dz = p1[j * 3 + 2] - p2[k * 3 + 2];
      dist[(size_t)b * n + j] = dx * dx + dy * dy + dz * dz;
    }
  }
  free(price); free(bid_inc); free(max_inc); free(ass_inv); free(bid); free(max_idx); free(unass); free(ncand);
  free(t_best); free(t_better); free(t_best_i);
  return 1;
}
