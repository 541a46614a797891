// Argument block and LDS budgets of the raster kernels.  Included by ct_raster.hip inside its anonymous namespace, ahead of the
// kernel headers (ct_raster_hot.h ...), and by the one-kernel translation units of tools/dev (register / ISA studies).
#pragma once

constexpr int kMaxLdsBytes = 64 * 1024;        // tile budget per workgroup (2 WGs / CU)
constexpr int kBigLdsBytes = 160 * 1024 - 512; // whole-CU budget for huge single-channel tiles

struct PosSrc {
  const float* keys;        // (B, H*DIM, N)            when FROM_KEYS
  const float* lc;          // (B, H, V, N)             otherwise
  const long long* idx;     // (B, H, V, N) int64
};

struct RasterArgs {
  PosSrc pos;
  const float* src;     // point-sized input  (B, H*C, N): feat or g_out
  const void* pad;      // (B, N) or null
  int pad_dtype;
  float* tile_out;      // grid-sized output  (B, H*C, G)
  const float* tile_in; // grid-sized input   (B, H*C, G)
  const float* tile_in2;// second grid-sized input (g_grid for splat-max bwd)
  float* dst;           // point-sized output (B, H*C, N)
  float* g_pos;         // g_keys (B,H*DIM,N) or g_lc (B,H,V,N)
  unsigned* claim;      // global copy of z used for single-winner claims (no-LDS fallback)
  int B, H, C, N;
  int CC;               // channels per tile
  int nchunks;          // ceil(C / CC)
  int ncg;              // chunk groups (grid.x split of the chunk loop in gather-reduce kernels)
  int nsplit;           // splits of N for pure gather kernels
  int atomic_gpos;      // accumulate g_pos with global atomics (ncg > 1)
  int cnt_mask;         // STATS: contributions are counted in cnt_mask+1 (a power of two) counters indexed by cell & cnt_mask
  size_t gpos_stride;   // > 0: channel-chunk group cg writes its partial g_pos to g_pos + cg*gpos_stride floats (summed afterwards)
  int accumulate;       // hot Splat(max) backward: g_pos += result instead of g_pos = result
  int nseg;             // fused Slice backward: point segments per (b,h) plane (grid.z = B * nseg); a.N = points per segment,
  int Nrow;             //   Nrow = length of a row of the point-sized tensors (= N when nseg == 1)
  // in-kernel folds of the partials (ct_raster_hot.h, arrive_last): arrival tickets (null: the partials are added by
  // sum_parts launches), where the folded g_keys / g_grid go, and whether the g_keys fold adds to what is there
  unsigned* tickets;
  float* fold_gpos;
  float* fold_grid;
  const float* fold_add;  // the g_keys fold adds these rows (the incoming key cotangent) to the sum, or null
  const float* gpos_add;  // hot Splat(max) backward: g_pos = gpos_add + result (may alias g_pos: in place), or null
  // Slice backward's statistics (per-channel max |src*pad|, contributions per cell) when N is split over workgroups: each
  // split writes its own pair per channel here ([nsplit][B*H*C][2] words, plain stores) and the scatter kernel combines them
  // (max of the maxima, sum of the per-split K) — instead of atomics on slots that a zero_slots launch had to clear first
  unsigned* stats;
  // sorted planes (ct_raster_sorted.h): the records ct_plane_sort wrote for these keys (null: the kernels sort themselves)
  const unsigned char* sorted;
  size_t sorted_stride;
};

// Workgroup -> (x, head, cloud) of a grid (X, H, B) whose X workgroups per (b, h) plane share that plane's keys (and, for
// the scatter / gather pairs, its feature rows or its tile).  The launch order is x-fastest and consecutive workgroups go to
// the 8 XCDs in turn (observed; speed only), so a plane's workgroups are given linear ids congruent mod 8: one XCD, one
// L2 — what the plane's second and later workgroups re-read is then served there instead of from HBM (the single-channel
// kernels of the C4 heads moved 1.6-3.1x the algorithmic bytes, profiles/r3_zoo_counters.txt).
struct BlockXHB {
  int x, h, b;
};
__device__ __forceinline__ BlockXHB block_xhb() {
  BlockXHB r;
  const unsigned X = gridDim.x, planes = gridDim.y * gridDim.z;
  if (X > 1 && (planes & 7u) == 0) {
    const unsigned L = blockIdx.x + X * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned grp = L / (8u * X), rem = L - grp * 8u * X;
    const unsigned plane = grp * 8u + (rem & 7u);
    r.x = (int)(rem >> 3);
    r.h = (int)(plane % gridDim.y);
    r.b = (int)(plane / gridDim.y);
  } else {
    r.x = blockIdx.x; r.h = blockIdx.y; r.b = blockIdx.z;
  }
  return r;
}
