// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 on gfx950 in the shapes the grouped-conv
// kernels use.  One workgroup per CU-slot, W waves per SIMD, NACC independent accumulators.
//   mode 0: bare MFMAs, constant operands
//   mode 1: B operand rewritten by a v_cndmask between MFMAs (the border masking of the conv kernels)
//   mode 2: B operand read from LDS one K-step ahead (register double buffer) + cndmask
// Prints cycles per MFMA per SIMD (32 = the 8-pass peak) and TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC, int MODE>
__global__ void __launch_bounds__(256) k(float* out, const float* in, int iters, int sel) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = in[i & 1023];
  __syncthreads();
  floatx4 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) acc[t] = floatx4{0, 0, 0, 0};
  float a = in[threadIdx.x & 63];
  float b[NACC], nb[NACC];
  const bool m0 = (threadIdx.x & 15) == sel, m1 = (threadIdx.x & 15) == sel + 1;
  const float* base = lds + (threadIdx.x & 15) * 260 + (threadIdx.x >> 4 & 3);
#pragma unroll
  for (int t = 0; t < NACC; ++t) b[t] = base[t * 3];
  __builtin_amdgcn_s_waitcnt(0xc07f);
  for (int it = 0; it < (MODE >= 3 ? 0 : iters); ++it) {
    if (MODE == 2) {
#pragma unroll
      for (int t = 0; t < NACC; ++t) nb[t] = base[((it + 1) & 7) * 4 + t * 3];
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      float bv = b[t];
      if (MODE >= 1) {
        if (t % 3 == 0) bv = m0 ? 0.0f : bv;
        if (t % 3 == 2) bv = m1 ? 0.0f : bv;
      }
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[t], 0, 0, 0);
    }
    if (MODE == 2) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
      for (int t = 0; t < NACC; ++t) b[t] = nb[t];
    }
  }
  if (MODE == 3 || MODE == 4) {
    // the ring kernel's order: 12 MFMAs per tile row on 3 accumulators (dependency distance 3; MODE 4: rows in pairs, distance 6)
    for (int it = 0; it < iters; ++it) {
      if (MODE == 3) {
#pragma unroll
        for (int r = 0; r < NACC / 3; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) acc[r * 3 + dx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j + dx], acc[r * 3 + dx], 0, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r + 1 < NACC / 3; r += 2)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
              for (int dx = 0; dx < 3; ++dx)
                acc[(r + rr) * 3 + dx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j + dx], acc[(r + rr) * 3 + dx], 0, 0, 0);
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int MODE>
void run(const char* name, int wgs_per_cu) {
  const int cus = 256, iters = 2000;
  float *out, *in;
  hipMalloc(&out, cus * wgs_per_cu * 256 * 4);
  hipMalloc(&in, 1024 * 4);
  hipMemset(in, 0, 1024 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, MODE><<<cus * wgs_per_cu, 256>>>(out, in, iters, 20);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC, MODE><<<cus * wgs_per_cu, 256>>>(out, in, iters, 20);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const int mult = MODE >= 3 ? 4 : 1;
  const double mfma_per_simd = (double)iters * NACC * mult * wgs_per_cu;      // one wave of each WG per SIMD
  const double total = (double)iters * NACC * mult * cus * wgs_per_cu * 4;
  printf("%-28s NACC %2d waves/SIMD %d: %.3f ms  %.1f ns/MFMA/SIMD (%.1f cyc @2.4GHz)  %.1f TFLOP/s\n", name, NACC, wgs_per_cu, ms,
         ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, total * 2048 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(in);
}

int main() {
  run<27, 0>("bare", 1);
  run<27, 0>("bare", 2);
  run<9, 0>("bare", 2);
  run<9, 0>("bare", 4);
  run<4, 0>("bare", 4);
  run<27, 1>("cndmask on B", 1);
  run<27, 1>("cndmask on B", 2);
  run<27, 3>("rows of 12 on 3 acc (x4)", 1);
  run<27, 3>("rows of 12 on 3 acc (x4)", 2);
  run<24, 4>("row pairs, 24 on 6 acc (x4)", 1);
  run<24, 4>("row pairs, 24 on 6 acc (x4)", 2);
  run<27, 2>("LDS prefetch + cndmask", 1);
  run<27, 2>("LDS prefetch + cndmask", 2);
  run<9, 2>("LDS prefetch + cndmask", 4);
  return 0;
}
