// Shared device helpers for libcloudct (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cloudct.h"

#define CT_WAVE 64

// float32(-1 + 1e-7) and float32(1 - 1e-7): the bounds torch uses when the
// reference clamps an fp32 tensor with python doubles
// (layers/cloud_transform.py:59,91).
#define CT_KEY_LO (-0.99999988f)
#define CT_KEY_HI (0.99999988f)

template <int DIM>
struct GridW {
  int W[DIM];      // extents, axis 0 slowest
  float hw[DIM];   // (W-1)*0.5f
  int G;           // cells
};

// A square grid whose extent is a template constant (WT > 0: the headline's 32 x 32, the zoo's 64 x 64 and 16 x 16): extents,
// half-widths and cell count as immediates (see grid3_of in ct_raster_hot3d.h); WT = 0: the caller's grid.
template <int WT>
__device__ __forceinline__ GridW<2> grid2_of(const GridW<2>& g) {
  if constexpr (WT == 0) {
    return g;
  } else {
    GridW<2> c;
    c.W[0] = c.W[1] = WT;
    c.hw[0] = c.hw[1] = (float)(WT - 1) * 0.5f;
    c.G = WT * WT;
    return c;
  }
}

// The zoo's 3D grids are cubes of 8 and 16 cells (model_zoo: 8^3 C32, 16^3 C16): with the extent a template constant the cell count,
// the corner offsets and the half-widths are immediates instead of registers in kernels that have none to spare (W3 = 0: any grid).
template <int W3>
__device__ __forceinline__ GridW<3> grid3_of(const GridW<3>& g) {
  if constexpr (W3 == 0) {
    return g;
  } else {
    GridW<3> c;
    c.W[0] = c.W[1] = c.W[2] = W3;
    c.hw[0] = c.hw[1] = c.hw[2] = (float)(W3 - 1) * 0.5f;
    c.G = W3 * W3 * W3;
    return c;
  }
}

// Per-axis terms of one key (layers/cloud_transform.py:91-94,
// layers/utils.py:122,168): clamp, (k+1)*((W-1)/2) in that rounding order,
// floor, low/high weights.
__device__ __forceinline__ void ct_axis(float key, float hw, int Wj, float& w0, float& w1, int& f) {
  float k = fminf(fmaxf(key, CT_KEY_LO), CT_KEY_HI);
  float s = (k + 1.0f) * hw;
  float fl = floorf(s);
  w0 = (fl + 1.0f) - s;
  w1 = s - fl;
  f = min((int)fl, Wj - 2);   // memory safety only: f <= W-2 already holds for W < 2^22
}

// clamp passes the gradient only inside [lo, hi] (torch.clamp backward)
__device__ __forceinline__ float ct_key_mask(float key) {
  return (key >= CT_KEY_LO && key <= CT_KEY_HI) ? 1.0f : 0.0f;
}

template <int DIM>
struct Corners {
  static constexpr int V = 1 << DIM;
  int cell[V];
  float w[V];
};

// corner v = dx + 2dy (+ 4dz); weight (wx*wy)*wz; cell = x*W1(*W2) + y(*W2) + z
template <int DIM>
__device__ __forceinline__ void ct_corners(const float (&w0)[DIM], const float (&w1)[DIM],
                                           const int (&f)[DIM], const GridW<DIM>& g, Corners<DIM>& c) {
  if constexpr (DIM == 2) {
    int base = f[0] * g.W[1] + f[1];
    c.cell[0] = base;
    c.cell[1] = base + g.W[1];
    c.cell[2] = base + 1;
    c.cell[3] = base + g.W[1] + 1;
    c.w[0] = w0[0] * w0[1];
    c.w[1] = w1[0] * w0[1];
    c.w[2] = w0[0] * w1[1];
    c.w[3] = w1[0] * w1[1];
  } else {
    int sx = g.W[1] * g.W[2], sy = g.W[2];
    int base = f[0] * sx + f[1] * sy + f[2];
    float xy00 = w0[0] * w0[1], xy10 = w1[0] * w0[1], xy01 = w0[0] * w1[1], xy11 = w1[0] * w1[1];
    c.cell[0] = base;
    c.cell[1] = base + sx;
    c.cell[2] = base + sy;
    c.cell[3] = base + sx + sy;
    c.cell[4] = base + 1;
    c.cell[5] = base + sx + 1;
    c.cell[6] = base + sy + 1;
    c.cell[7] = base + sx + sy + 1;
    c.w[0] = xy00 * w0[2];
    c.w[1] = xy10 * w0[2];
    c.w[2] = xy01 * w0[2];
    c.w[3] = xy11 * w0[2];
    c.w[4] = xy00 * w1[2];
    c.w[5] = xy10 * w1[2];
    c.w[6] = xy01 * w1[2];
    c.w[7] = xy11 * w1[2];
  }
}

// d(weights)/d(s_j) contracted with the corner cotangents gw[v]:
//   gs_j = sum_v gw[v] * sign_j(v) * prod_{i != j} w_i(d_i(v))
template <int DIM>
__device__ __forceinline__ void ct_corner_grad(const float (&w0)[DIM], const float (&w1)[DIM],
                                               const float (&gw)[1 << DIM], float (&gs)[DIM]) {
  if constexpr (DIM == 2) {
    gs[0] = (gw[1] - gw[0]) * w0[1] + (gw[3] - gw[2]) * w1[1];
    gs[1] = (gw[2] - gw[0]) * w0[0] + (gw[3] - gw[1]) * w1[0];
  } else {
    float yz00 = w0[1] * w0[2], yz10 = w1[1] * w0[2], yz01 = w0[1] * w1[2], yz11 = w1[1] * w1[2];
    float xz00 = w0[0] * w0[2], xz10 = w1[0] * w0[2], xz01 = w0[0] * w1[2], xz11 = w1[0] * w1[2];
    float xy00 = w0[0] * w0[1], xy10 = w1[0] * w0[1], xy01 = w0[0] * w1[1], xy11 = w1[0] * w1[1];
    gs[0] = (gw[1] - gw[0]) * yz00 + (gw[3] - gw[2]) * yz10 + (gw[5] - gw[4]) * yz01 + (gw[7] - gw[6]) * yz11;
    gs[1] = (gw[2] - gw[0]) * xz00 + (gw[3] - gw[1]) * xz10 + (gw[6] - gw[4]) * xz01 + (gw[7] - gw[5]) * xz11;
    gs[2] = (gw[4] - gw[0]) * xy00 + (gw[5] - gw[1]) * xy10 + (gw[6] - gw[2]) * xy01 + (gw[7] - gw[3]) * xy11;
  }
}

__device__ __forceinline__ float ct_load_pad(const void* pad, int pad_dtype, size_t i) {
  if (pad_dtype == CT_PAD_F32) return ((const float*)pad)[i];
  if (pad_dtype == CT_PAD_I32) return (float)((const int*)pad)[i];
  return 1.0f;
}

typedef float ct_f4 __attribute__((ext_vector_type(4)));

// Streaming (read-once / write-once) global accesses carry the non-temporal hint, so that what IS re-read inside a
// kernel — keys, and the second quad's g_out rows of the fused Slice backward — stays in the XCD's 4 MiB L2 instead
// of being pushed out by the streams (the re-reads then never reach the memory-side counters, nor HBM).
__device__ __forceinline__ float4 ld_stream4(const float* p) {
  const ct_f4 v = __builtin_nontemporal_load((const ct_f4*)p);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_stream4(float* p, float4 v) {
  const ct_f4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, (ct_f4*)p);
}
__device__ __forceinline__ float ld_stream(const float* p) { return __builtin_nontemporal_load(p); }

// hipGetLastError() is sticky across unrelated runtime calls of the host
// framework (e.g. a hipEventQuery that returned hipErrorNotReady), so every
// entry point clears it first (CT_CLEAR_ERROR) and checks after its launches.
#define CT_CLEAR_ERROR() ((void)hipGetLastError())
#define CT_CHECK_LAUNCH()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return CT_ELAUNCH;     \
  } while (0)
