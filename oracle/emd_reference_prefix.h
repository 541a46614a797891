/* Prefix header of the `ref_emd` target of oracle/Makefile (TEST INFRASTRUCTURE ONLY).
 *
 * The reference's emd_linear/emd_cuda.cu:10-20 defines its own `atomicMax(float*, float)` (a compare-and-swap loop); the HIP
 * runtime header declares an overload of the same name and signature, so the reference's definition does not compile beside it.
 * The platform headers are included here FIRST (they are include-guarded: the source's own includes become no-ops), then the
 * name is redirected for the reference's translation unit, so that ITS definition and ITS three uses — unchanged — get a name of
 * their own.  Nothing of the reference's code is replaced: the compare-and-swap loop that runs is the reference's. */
#include <hip/hip_runtime.h>
#include <ATen/ATen.h>
#define atomicMax emd_reference_atomicMax
