// early_exit_probe.hip — does a wavefront that ends before the rest of its workgroup give its registers and its wavefront
// slot back?  Workgroups of 8 wavefronts at 128 VGPRs (4 wavefronts per SIMD: two workgroups per CU when all eight stay);
// wavefront 0 spins for SPIN cycles, the others either spin too (mode 0) or end at once (mode 1).  If ended wavefronts free
// their slots, mode 1 runs ~8 workgroups per CU at a time and finishes ~4x sooner.
//   hipcc --offload-arch=gfx950 -O3 -o early_exit_probe early_exit_probe.hip && ./early_exit_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void probe(int mode, long long spin, float* sink) {
	float acc[100]; // keep the allocation near 128 registers
#pragma unroll
	for (int i = 0; i < 100; ++i) { acc[i] = threadIdx.x * 0.5f + i; }
	if (mode == 1 && (threadIdx.x >> 6) != 0) { return; }
	const long long t0 = __builtin_readcyclecounter();
	while (__builtin_readcyclecounter() - t0 < spin) {
#pragma unroll
		for (int i = 0; i < 100; ++i) { acc[i] = acc[i] * 1.0001f + 0.5f; }
	}
	float s = 0;
#pragma unroll
	for (int i = 0; i < 100; ++i) { s += acc[i]; }
	if (s == 12345.678f) { sink[0] = s; }
}

int main() {
	float* sink;
	hipMalloc(&sink, 4);
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	for (int mode = 0; mode < 2; ++mode) {
		for (int rep = 0; rep < 3; ++rep) {
			hipEventRecord(a);
			hipLaunchKernelGGL(probe, dim3(256 * 64), dim3(512), 0, 0, mode, 20000LL, sink);
			hipEventRecord(b);
			hipEventSynchronize(b);
			float ms;
			hipEventElapsedTime(&ms, a, b);
			printf("mode %d (%s): %.3f ms for %d workgroups\n", mode, mode ? "7 of 8 wavefronts end at once" : "all 8 wavefronts spin", ms, 256 * 64);
		}
	}
	return 0;
}
