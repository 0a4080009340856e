// compile check of csrc/agx_water.h for gfx950 (tests/test_drinking.py): the header is not part of a kernel variant yet
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "agx_wave.h"
#include "../../include/agx_blob.h"
#include "agx_water.h"
extern "C" __global__ void __launch_bounds__(64) agx_water_kernel_check(const uint32_t* blob, const float* state, const float* trace, float* water, float* report, int nsub, int sw, int tw) {
  __shared__ float lds[agxw::LDS_WORDS];
  const int env = blockIdx.x;
  agxw::water_env(blob, state + (size_t)env * sw, trace + (size_t)env * tw, water + (size_t)env * 384, report + (size_t)env * 64, nsub, lds, threadIdx.x);
}
