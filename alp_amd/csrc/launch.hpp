// launch.hpp — host-side launch functions implemented by the *.hip kernel files (internal to libalpgpu.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/alpgpu.h"

namespace alpgpu {

// decode_kernels.hip
int launch_decode_column(hipStream_t stream, const alpgpu_column* col, double* d_out, int variant, int n_cus);

} // namespace alpgpu
