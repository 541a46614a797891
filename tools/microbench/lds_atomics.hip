// Micro-benchmark: throughput of LDS atomic flavours on gfx950 under the access
// pattern of the rasterizer (1024 threads / WG, 64 KiB tile, 2 WGs per CU).
//   mode 0: addresses uniformly random over the tile (data-like)
//   mode 1: conflict-free (lane i -> word base+i), distinct per wave instruction
// Prints ns per wave-instruction per CU (lower is better).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

enum Op { ADD_F32, ADD_U32, MAX_U32, ADD_U64, MAX_U64, ADD_F64, CAS_U32, RMW_PLAIN, READ_B32, WRITE_B32, PKADD_BF16 };
static const char* names[] = {"ds_add_f32", "ds_add_u32", "ds_max_u32", "ds_add_u64", "ds_max_u64", "ds_add_f64", "ds_cmpst_b32", "plain rmw f32", "ds_read_b32", "ds_write_b32", "ds_pk_add_bf16"};

template <int OP>
__global__ void __launch_bounds__(1024) k(int mode, float* out, int iters, int words) {
  extern __shared__ __align__(16) unsigned char lds[];
  float* f = (float*)lds;
  unsigned* u = (unsigned*)lds;
  unsigned long long* u64 = (unsigned long long*)lds;
  double* f64 = (double*)lds;
  for (int i = threadIdx.x; i < words; i += blockDim.x) u[i] = 0;
  __syncthreads();
  float acc = 0;
  // 16 addresses per thread, generated up front so that the timed loop is (almost) pure LDS traffic
  unsigned s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  unsigned wa[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    unsigned r = s >> 8;
    if (mode == 0) wa[j] = r % (unsigned)words;                                           // random word
    else if (mode == 1) wa[j] = (threadIdx.x + (r % 15u) * 1024u) % (unsigned)words;       // conflict-free
    else if (mode == 2) wa[j] = (r % 1024u) * 16u + (unsigned)j;                           // [cell][c]: lanes = points, same c
    else {                                                                                 // [cell][c]: lanes = (4 points x 16 c)
      unsigned pt = __shfl(r, threadIdx.x & ~15u, 64);                                     // one random cell per 16 lanes
      wa[j] = (pt % 1024u) * 16u + (threadIdx.x & 15u);
    }
  }
  for (int it = 0; it < iters / 16; ++it) {
#pragma unroll
   for (int j = 0; j < 16; ++j) {
    unsigned w = wa[j];
    float val = (float)(w & 7) + 0.5f;
    if (OP == ADD_F32) atomicAdd(&f[w], val);
    if (OP == ADD_U32) atomicAdd(&u[w], w);
    if (OP == MAX_U32) atomicMax(&u[w], w * 2654435761u);
    if (OP == ADD_U64) atomicAdd(&u64[w >> 1], (unsigned long long)w);
    if (OP == MAX_U64) atomicMax(&u64[w >> 1], (unsigned long long)w * 2654435761ull);
    if (OP == ADD_F64) atomicAdd(&f64[w >> 1], (double)val);
    if (OP == CAS_U32) acc += (float)atomicCAS(&u[w], 0u, w);
    if (OP == RMW_PLAIN) f[w] = f[w] + val;
    if (OP == READ_B32) acc += f[w];
    if (OP == WRITE_B32) f[w] = val;
    if (OP == PKADD_BF16) {
      typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
      bf2 v2 = {(__bf16)val, (__bf16)val};
      __builtin_amdgcn_ds_atomic_fadd_v2bf16((__attribute__((address_space(3))) bf2*)(__attribute__((address_space(3))) void*)&u[w], v2);
    }
   }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < words; i += blockDim.x) acc += f[i];
  if (acc == 123.456f) out[0] = acc;
}

template <int OP>
void run(int d_addr, float* d_out, int blocks, int iters, int words, const char* tag) {
  size_t lds = (size_t)words * 4;
  hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 1024, lds>>>(d_addr, d_out, iters, words);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<OP><<<blocks, 1024, lds>>>(d_addr, d_out, iters, words);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  hipError_t err = hipGetLastError();
  // wave-instructions per CU: blocks * 16 waves * iters / 256 CUs
  double wi_per_cu = (double)blocks * 16 * iters / 256.0;
  printf("%-16s %-12s %8.1f us   %7.2f ns / wave-instr / CU  (%s)\n", names[OP], tag, ms * 1e3, ms * 1e6 / wi_per_cu, hipGetErrorString(err));
}

int main() {
  const int blocks = 512, iters = 1024, words = 16384;  // 64 KiB
  float* d_out;
  hipMalloc(&d_out, 4);
  for (int mode = 0; mode < 4; ++mode) {
    int d_addr = mode;
    const char* tag = mode == 0 ? "random" : (mode == 1 ? "conflict-free" : (mode == 2 ? "cell*16+c" : "4pt x 16c"));
    run<ADD_F32>(d_addr, d_out, blocks, iters, words, tag);
    run<ADD_U32>(d_addr, d_out, blocks, iters, words, tag);
    run<MAX_U32>(d_addr, d_out, blocks, iters, words, tag);
    run<ADD_U64>(d_addr, d_out, blocks, iters, words, tag);
    run<MAX_U64>(d_addr, d_out, blocks, iters, words, tag);
    run<ADD_F64>(d_addr, d_out, blocks, iters, words, tag);
    run<CAS_U32>(d_addr, d_out, blocks, iters, words, tag);
    run<RMW_PLAIN>(d_addr, d_out, blocks, iters, words, tag);
    run<READ_B32>(d_addr, d_out, blocks, iters, words, tag);
    run<WRITE_B32>(d_addr, d_out, blocks, iters, words, tag);
    run<PKADD_BF16>(d_addr, d_out, blocks, iters, words, tag);
  }
  return 0;
}
