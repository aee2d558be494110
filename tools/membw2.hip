// tools/membw2.hip — does a smaller in-flight footprint (several waves cooperating on one 8 KiB vector) raise
// the mixed read+write HBM rate?  hipcc --offload-arch=gfx950 -O3 -o tools/membw2 tools/membw2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u2 __attribute__((ext_vector_type(2)));

// WPV = waves cooperating per vector (1,2,4,8); each wave writes 8/WPV KiB of the vector and reads RU/WPV KiB
template <int WPV, int RU, int NT>
__global__ void k(const u2* __restrict__ in, d2* __restrict__ out, unsigned long long n_vec, int wpw, int persistent) {
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int vpw  = wpw / WPV;              // vectors per workgroup iteration
	const int sub  = wave % WPV;             // which part of the vector this wave owns
	const int vloc = wave / WPV;
	const unsigned long long stride = persistent ? (unsigned long long)gridDim.x * vpw : n_vec;
	for (unsigned long long v = (unsigned long long)blockIdx.x * vpw + vloc; v < n_vec; v += stride) {
		u2 acc = {v, 1};
		constexpr int RPW = RU * 64 / WPV;   // 16-byte units per wave... RU KiB = RU*64 units per vector
#pragma unroll
		for (int j = 0; j < RPW / 64; ++j) acc += in[v * (RU * 64) + sub * RPW + j * 64 + lane];
		if (RPW % 64) { if (lane < RPW % 64) acc += in[v * (RU * 64) + sub * RPW + (RPW / 64) * 64 + lane]; }
		d2* dst = out + v * 512;
#pragma unroll
		for (int mm = 0; mm < 8 / WPV; ++mm) {
			const int m = sub * (8 / WPV) + mm;
			d2 o; o.x = __longlong_as_double((long long)(acc.x + m)); o.y = __longlong_as_double((long long)acc.y);
			if (NT) __builtin_nontemporal_store(o, dst + 64 * m + lane); else dst[64 * m + lane] = o;
		}
	}
}
template <int WPV, int RU, int NT>
float run(const u2* in, d2* out, unsigned long long n, int wpw, int grid, int persistent) {
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	std::vector<float> ts;
	for (int i = 0; i < 9; ++i) {
		hipEventRecord(a);
		hipLaunchKernelGGL((k<WPV, RU, NT>), dim3(grid), dim3(64 * wpw), 0, 0, in, out, n, wpw, persistent);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b); if (i >= 2) ts.push_back(ms);
	}
	std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2];
}
template <int WPV, int RU>
void sweep(const u2* in, d2* out, unsigned long long n) {
	for (int wpw : {4, 8, 16}) {
		if (wpw < WPV) continue;
		const int vpw = wpw / WPV;
		for (int persistent : {0, 1}) {
			for (int occ : {8, 16, 32}) { // target resident waves per CU for the persistent grid
				if (!persistent && occ != 8) continue;
				int grid = persistent ? 256 * occ / wpw : (int)(n / vpw);
				float t0 = run<WPV, RU, 0>(in, out, n, wpw, grid, persistent);
				float t1 = run<WPV, RU, 1>(in, out, n, wpw, grid, persistent);
				printf("WPV=%d read=%dKiB wpw=%2d %s grid=%7d: plain %.3f ms %5.0f GB/s | NT %.3f ms %5.0f GB/s\n", WPV, RU, wpw,
				       persistent ? "persist" : "1-shot ", grid, t0, n * (8192.0 + 1024 * RU) / t0 / 1e6, t1, n * (8192.0 + 1024 * RU) / t1 / 1e6);
			}
		}
	}
}
int main() {
	const unsigned long long n = 1ull << 20;
	d2* out; u2* in; hipMalloc(&out, n * 8192); hipMalloc(&in, n * 8192);
	hipMemset(in, 1, n * 8192); hipMemset(out, 0, n * 8192);
	sweep<1, 0>(in, out, n); sweep<4, 0>(in, out, n); sweep<8, 0>(in, out, n);
	sweep<1, 2>(in, out, n); sweep<2, 2>(in, out, n); sweep<4, 2>(in, out, n); sweep<8, 2>(in, out, n);
	sweep<1, 4>(in, out, n); sweep<4, 4>(in, out, n);
	return 0;
}
