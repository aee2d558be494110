// dispatch_rate.hip — how fast does the chip start (and retire) workgroups that do nothing, or one dependent round trip, at the decode kernels'
// launch shapes?  hipcc --offload-arch=gfx950 -O3 tools/dispatch_rate.hip -o tools/dispatch_rate && tools/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int LDS_BYTES, int TRIPS>
__global__ void k_touch(const unsigned* __restrict__ in, unsigned* __restrict__ out, unsigned n_in) {
	__shared__ unsigned char lds[LDS_BYTES];
	unsigned                 x = blockIdx.x;
	if (TRIPS > 0) {
#pragma unroll
		for (int t = 0; t < TRIPS; ++t) { // TRIPS dependent loads: the first ones one word per workgroup (a "descriptor"), the last one 4 KiB contiguous per workgroup
			const unsigned at = (x * 2654435761u) % (n_in - 4096u);
			x                 = t == TRIPS - 1 ? in[(at & ~1023u) + threadIdx.x * (1024u / blockDim.x)] : in[at];
		}
		lds[threadIdx.x] = static_cast<unsigned char>(x);
		__syncthreads();
		if (threadIdx.x == 0 && lds[1] == 255 && x == 0xFFFFFFFFu) { out[blockIdx.x & 1023] = x; }
	} else {
		if (threadIdx.x == 0 && blockIdx.x == 0xFFFFFFF0u) { lds[0] = 1; out[0] = lds[0]; }
	}
}
__global__ void k_fill(unsigned* p) {
	const unsigned i = blockIdx.x * 256u + threadIdx.x;
	unsigned       h = i * 0x9E3779B1u;
	h ^= h >> 15, h *= 0x85EBCA77u, h ^= h >> 13;
	p[i] = h;
}
template <class K>
static void run(const char* name, K kernel, unsigned grid, unsigned block, const unsigned* in, unsigned* out, unsigned n_in) {
	hipEvent_t a, b;
	hipEventCreate(&a), hipEventCreate(&b);
	for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, 0, in, out, n_in); }
	hipEventRecord(a);
	const int reps = 10;
	for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, 0, in, out, n_in); }
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms = 0;
	hipEventElapsedTime(&ms, a, b);
	ms /= reps;
	printf("%-58s grid %8u x %3u: %.3f ms  = %.1f workgroups/us, %.1f wavefronts/us\n", name, grid, block, ms, grid / ms / 1e3, grid * (block / 64.0) / ms / 1e3);
}
int main() {
	const unsigned n_in = 1u << 28; // 1 GiB of words: the dependent loads miss every cache
	unsigned *in, *out;
	hipMalloc(&in, size_t(n_in) * 4), hipMalloc(&out, 4096);
	hipLaunchKernelGGL(k_fill, dim3(n_in / 256), dim3(256), 0, 0, in);
	run("empty, 256 threads, 20 KiB LDS (decode: 2 vectors/WG)", k_touch<20480, 0>, 1u << 19, 256, in, out, n_in);
	run("empty, 256 threads, 10 KiB LDS (decode: 1 vector/WG)", k_touch<10240, 0>, 1u << 20, 256, in, out, n_in);
	run("empty,  64 threads,  5 KiB LDS (one wavefront per vector)", k_touch<5120, 0>, 1u << 20, 64, in, out, n_in);
	run("1 round trip, 256 threads, 20 KiB LDS", k_touch<20480, 1>, 1u << 19, 256, in, out, n_in);
	run("2 dependent round trips, 256 threads, 20 KiB LDS", k_touch<20480, 2>, 1u << 19, 256, in, out, n_in);
	run("2 dependent round trips, 256 threads, 10 KiB LDS", k_touch<10240, 2>, 1u << 20, 256, in, out, n_in);
	run("2 dependent round trips,  64 threads,  5 KiB LDS", k_touch<5120, 2>, 1u << 20, 64, in, out, n_in);
	run("2 dependent round trips, 128 threads,  5 KiB LDS", k_touch<5120, 2>, 1u << 20, 128, in, out, n_in);
	return 0;
}
