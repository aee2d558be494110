// tools/membw_encode.hip — the encode kernel's traffic without its arithmetic: every 1024-double vector (8 KiB) is read once and
// WK KiB are written at v * WK KiB; the stored data depend on everything that was read (as the packed words do).  How much of
// the kernel's time is the memory system's, and does the shape of the launch matter?
//   WPV waves share a vector (each reads 8/WPV KiB, writes WK/WPV KiB); wpw waves per workgroup; one workgroup per wpw/WPV vectors.
// hipcc --offload-arch=gfx950 -O3 -o tools/membw_encode tools/membw_encode.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned long long u2 __attribute__((ext_vector_type(2)));

template <int WPV, int WK, int NTL = 0, int NTS = 0>
__global__ void k(const u2* __restrict__ in, u2* __restrict__ out, unsigned long long n_vec, int wpw) {
	__shared__ unsigned long long part[16];
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int vpw  = wpw / WPV;
	const int sub  = wave % WPV;
	const unsigned long long v = (unsigned long long)blockIdx.x * vpw + wave / WPV;
	if (v >= n_vec) return;
	constexpr int RPW = 512 / WPV; // 16-byte units per wave
	u2 acc = {v, 1};
#pragma unroll
	for (int j = 0; j < RPW / 64; ++j) acc += NTL ? __builtin_nontemporal_load(in + v * 512 + sub * RPW + j * 64 + lane) : in[v * 512 + sub * RPW + j * 64 + lane];
	if (WPV > 1) { // the waves of a vector exchange one word, as a cross-wave min/max would
		if (lane == 0) part[wave] = acc.x;
		__syncthreads();
		acc.x += part[(wave / WPV) * WPV + ((sub + 1) % WPV)];
	}
	constexpr int WPWU = WK * 64 / WPV; // 16-byte units written per wave
	u2* dst = out + v * (WK * 64) + sub * WPWU;
#pragma unroll
	for (int j = 0; j < (WPWU + 63) / 64; ++j) {
		if (j * 64 + lane < WPWU) { u2 o = acc; o.x += j; if (NTS) __builtin_nontemporal_store(o, dst + j * 64 + lane); else dst[j * 64 + lane] = o; }
	}
}
template <int WPV, int WK, int NTL = 0, int NTS = 0>
void run(const u2* in, u2* out, unsigned long long n) {
	for (int wpw : {4, 8, 16}) {
		if (wpw < WPV) continue;
		const int vpw = wpw / WPV;
		hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
		std::vector<float> ts;
		for (int i = 0; i < 9; ++i) {
			hipEventRecord(a);
			hipLaunchKernelGGL((k<WPV, WK, NTL, NTS>), dim3((unsigned)((n + vpw - 1) / vpw)), dim3(64 * wpw), 0, 0, in, out, n, wpw);
			hipEventRecord(b); hipEventSynchronize(b);
			float ms; hipEventElapsedTime(&ms, a, b); if (i >= 2) ts.push_back(ms);
		}
		std::sort(ts.begin(), ts.end());
		const float t = ts[ts.size() / 2];
		printf("waves/vector=%d write=%dKiB waves/wg=%2d (%d vectors/wg)%s%s: %.3f ms  %5.0f GB/s read+write\n", WPV, WK, wpw, vpw, NTL ? " nt-loads" : "", NTS ? " nt-stores" : "", t, n * (8192.0 + 1024 * WK) / t / 1e6);
	}
}
int main() {
	const unsigned long long n = 1ull << 20;
	u2 *in, *out; hipMalloc(&in, n * 8192); hipMalloc(&out, n * 8192);
	hipMemset(in, 1, n * 8192); hipMemset(out, 0, n * 8192);
	run<1, 4>(in, out, n); run<2, 4>(in, out, n); run<4, 4>(in, out, n); run<8, 4>(in, out, n);
	run<1, 7>(in, out, n); run<4, 7>(in, out, n);
	run<1, 2>(in, out, n); run<4, 2>(in, out, n);
	run<1, 4, 0, 1>(in, out, n); run<1, 4, 1, 0>(in, out, n); run<1, 4, 1, 1>(in, out, n); run<1, 7, 1, 1>(in, out, n);
	return 0;
}
