// tools/membw3.hip — small reads per vector (128 B * R128) + 8 KiB store, one 4-wave workgroup per vector, 1-shot.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
// VPW vectors per workgroup (each handled by 4 waves); read R128*128 bytes per vector (contiguous across vectors)
template <int VPW>
__global__ void k(const u2* __restrict__ in, d2* __restrict__ out, unsigned long long n_vec, int r128, int use_lds) {
	__shared__ u2 stage[VPW][640];
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int vl = wave / 4, sub = wave % 4, t = tid - vl * 256;
	const unsigned long long v = (unsigned long long)blockIdx.x * VPW + vl;
	if (v >= n_vec) return;
	const int n_units = 8 * r128;
	const u2* g = in + v * n_units;
	u2 acc = {v, 1};
	for (int c = t; c < n_units; c += 256) { u2 w = g[c]; if (use_lds) stage[vl][c] = w; else acc += w; }
	__syncthreads();
	if (use_lds && n_units) acc += stage[vl][(lane * 7) % n_units];
	d2* dst = out + v * 512;
#pragma unroll
	for (int mm = 0; mm < 2; ++mm) {
		const int m = 2 * sub + mm;
		d2 o; o.x = __longlong_as_double((long long)(acc.x + m)); o.y = __longlong_as_double((long long)acc.y);
		__builtin_nontemporal_store(o, dst + 64 * m + lane);
	}
}
template <int VPW>
float run(const u2* in, d2* out, unsigned long long n, int r128, int use_lds) {
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	std::vector<float> ts;
	for (int i = 0; i < 9; ++i) {
		hipEventRecord(a);
		hipLaunchKernelGGL((k<VPW>), dim3((unsigned)(n / VPW)), dim3(256 * VPW), 0, 0, in, out, n, r128, use_lds);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b); if (i >= 2) ts.push_back(ms);
	}
	std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2];
}
int main() {
	const unsigned long long n = 1ull << 20;
	d2* out; u2* in; hipMalloc(&out, n * 8192); hipMalloc(&in, n * 8192);
	hipMemset(in, 1, n * 8192); hipMemset(out, 0, n * 8192);
	for (int r128 : {0, 1, 2, 4, 6, 8, 10, 12, 16, 24, 32, 53}) {
		float t1 = run<1>(in, out, n, r128, 1), t1n = run<1>(in, out, n, r128, 0), t2 = run<2>(in, out, n, r128, 1), t4 = run<4>(in, out, n, r128, 1);
		double bytes = n * (8192.0 + 128.0 * r128);
		printf("read %2d*128B: VPW1 lds %.3f ms %5.0f GB/s | VPW1 nolds %.3f ms %5.0f | VPW2 %.3f ms %5.0f | VPW4 %.3f ms %5.0f\n", r128, t1, bytes / t1 / 1e6,
		       t1n, bytes / t1n / 1e6, t2, bytes / t2 / 1e6, t4, bytes / t4 / 1e6);
	}
	return 0;
}
