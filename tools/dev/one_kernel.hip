// tools/dev/one_kernel.hip: ONE raster kernel instantiated alone (seconds instead of the two minutes of ct_raster.hip), for register
// and ISA studies:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -I include -I cloud_transformers_amd/csrc
//   -DCT_ONE='slice_bwd_sorted3_kernel<false, 8, true>' -Rpass-analysis=kernel-resource-usage -c tools/dev/one_kernel.hip -o /tmp/one.o
#include "ct_common.h"
#include <type_traits>
namespace ctdev {
#include "ct_raster_args.h"
#include "ct_raster_hot.h"
#include "ct_raster_hot3d.h"
#include "ct_raster_sorted.h"
#include "ct_raster_sorted3d.h"
}
#ifndef CT_ONE_ARGS
#define CT_ONE_ARGS RasterArgs, GridW<3>
#endif
#ifdef CT_ONE_2D
#undef CT_ONE_ARGS
#define CT_ONE_ARGS RasterArgs, GridW<2>
#endif
namespace ctdev {
template __global__ void CT_ONE(CT_ONE_ARGS);
}
