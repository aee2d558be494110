// pad_kernels.hip — tail padding: incomplete last vectors are padded with the tail vector's first value
// (PRIMITIVES.md:141-144, first strategy; the reference's drivers simply drop the tail,
// publication/source_code/bench_compression_ratio/alp.cpp:195).  The serialized form of a compressed column — the HBM layout of
// include/alpgpu.h laid end to end behind a 64-byte header whose n_values says where the data ends — is host code: api_container.hip,
// alpgpu_column_to_blob / alpgpu_column_from_blob.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/alpgpu.h"
#include "launch.hpp"

namespace alpgpu {

__global__ void k_pad_tail(double* __restrict__ data, uint64_t n_values) {
	const uint64_t first = n_values & ~1023ull; // first index of the incomplete vector
	const uint64_t i     = n_values + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i < first + 1024) { data[i] = data[first]; }
}

int launch_pad_tail(hipStream_t stream, double* d_in, uint64_t n_values) {
	if ((n_values & 1023ull) == 0) { return ALPGPU_OK; }
	hipLaunchKernelGGL(k_pad_tail, dim3(4), dim3(256), 0, stream, d_in, n_values);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

} // namespace alpgpu
