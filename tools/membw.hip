// tools/membw.hip — standalone HBM store/load structure micro-benchmark for the decode kernel's access pattern.
// hipcc --offload-arch=gfx950 -O3 -o tools/membw tools/membw.hip ; ./tools/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u2 __attribute__((ext_vector_type(2)));

// MODE bit0: NT stores; bit1: persistent grid-stride (else one wave per vector, grid covers all)
// READ_UNITS: 16-byte units read per lane per vector (0..8) from `in` (same index pattern as packed words)
template <int NT, int READ_UNITS>
__global__ void k_stream(const u2* __restrict__ in, d2* __restrict__ out, unsigned long long n_vec, int waves_per_wg) {
	extern __shared__ char smem[];
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned long long stride = (unsigned long long)gridDim.x * waves_per_wg;
	for (unsigned long long v = (unsigned long long)blockIdx.x * waves_per_wg + wave; v < n_vec; v += stride) {
		u2 acc = {v, 1};
#pragma unroll
		for (int j = 0; j < READ_UNITS; ++j) {
			u2 w = in[(v * READ_UNITS + j) * 64 + lane];
			acc += w;
		}
		d2* dst = out + v * 512;
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			d2 o;
			o.x = __longlong_as_double((long long)(acc.x + m));
			o.y = __longlong_as_double((long long)(acc.y));
			if (NT) __builtin_nontemporal_store(o, dst + 64 * m + lane);
			else dst[64 * m + lane] = o;
		}
	}
	if (smem[0] == 123 && lane == 77) out[0].x = 1.0; // keep the dynamic LDS allocation alive
}

template <int NT, int RU>
float run(const u2* in, d2* out, unsigned long long n, int wpw, int grid, int lds, int iters) {
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_stream<NT, RU>), dim3(grid), dim3(64 * wpw), lds, 0, in, out, n, wpw);
	std::vector<float> ts;
	for (int i = 0; i < iters; ++i) {
		hipEventRecord(a);
		hipLaunchKernelGGL((k_stream<NT, RU>), dim3(grid), dim3(64 * wpw), lds, 0, in, out, n, wpw);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms);
	}
	std::sort(ts.begin(), ts.end());
	return ts[ts.size() / 2];
}

int main() {
	const unsigned long long n = 1ull << 20;
	d2* out; u2* in;
	hipMalloc(&out, n * 8192); hipMalloc(&in, n * 8192);
	hipMemset(in, 1, n * 8192); hipMemset(out, 0, n * 8192);
	hipFuncSetAttribute((const void*)k_stream<1,0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	printf("store-only (8 KiB per wave per vector); GB/s of stores\n");
	for (int wpw : {1, 4, 8, 16}) {
		for (int lds_per_wave : {0, 4096, 10000}) {
			const int lds = lds_per_wave * wpw;
			if (lds > 160 * 1024) continue;
			for (int gmul : {0, 4, 8, 16, 32}) { // 0 = non-persistent (one wave per vector)
				const int waves_cu = lds_per_wave ? std::min(32, (160 * 1024) / lds_per_wave) : 32;
				int grid = gmul == 0 ? (int)(n / wpw) : 256 * gmul * 4 / wpw;
				if (grid < 1) continue;
				float t1 = run<1, 0>(in, out, n, wpw, grid, lds, 7);
				float t0 = run<0, 0>(in, out, n, wpw, grid, lds, 7);
				printf("wpw=%2d lds/wave=%5d (<=%2d waves/CU) grid=%7d  NT: %.3f ms %.0f GB/s   plain: %.3f ms %.0f GB/s\n", wpw, lds_per_wave, waves_cu,
				       grid, t1, n * 8192 / t1 / 1e6, t0, n * 8192 / t0 / 1e6);
			}
		}
	}
	printf("\nread R units (16 B/lane each = 1 KiB per wave) + store 8 KiB per vector; total traffic GB/s\n");
#define RUNR(RU) { float t1 = run<1, RU>(in, out, n, 4, 4096, 4 * 10000, 7); float t2 = run<1, RU>(in, out, n, 4, (int)(n / 4), 0, 7); float t3 = run<1, RU>(in, out, n, 4, 8192, 0, 7); \
	printf("read %d KiB: lds10000 grid4096: %.3f ms %.0f GB/s | nonpersistent lds0: %.3f ms %.0f GB/s | persistent lds0 grid8192: %.3f ms %.0f GB/s\n", RU, t1, n * (8192.0 + 1024 * RU) / t1 / 1e6, t2, n * (8192.0 + 1024 * RU) / t2 / 1e6, t3, n * (8192.0 + 1024 * RU) / t3 / 1e6); }
	RUNR(1) RUNR(2) RUNR(4) RUNR(8)
	return 0;
}
