// rw_phases.hip — what the HBM interface gives a kernel that writes a lot and reads a little, by the SHAPE of the launch (round 6, profiles/r06_float_decode.txt).
// Each "chunk" is W bytes written (16-byte stores, 1 KiB per wavefront and instruction) and R = W / 10 bytes read (contiguous, 16-byte loads).
//   persistent: G = CUs x wgs workgroups, workgroup b takes chunks b, b + G, ...;   dispatched: one workgroup per chunk, in order, `pad` KiB of unused LDS cap the residency
//   stores non-temporal or plain; reads none / in front of the chunk's stores
// build: hipcc --offload-arch=gfx950 -O3 -o build/rw_phases tools/micro/rw_phases.hip ; run: build/rw_phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <bool NT, bool PERSISTENT>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n_chunks, unsigned r_units, unsigned w_units, unsigned* sink) {
	extern __shared__ unsigned pad[];
	const size_t G = PERSISTENT ? gridDim.x : n_chunks;
	unsigned x = 0;
	for (size_t chunk = blockIdx.x; chunk < n_chunks; chunk += G) {
		if (r_units) {
			const u32x4* p = in + chunk * r_units;
			u32x4 acc = {0, 0, 0, 0};
			for (unsigned u = threadIdx.x; u < r_units; u += 256) { acc ^= p[u]; }
			x ^= acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
		}
		u32x4* q = out + chunk * w_units;
		const u32x4 v = {x, 1u, 2u, 3u};
		for (unsigned u = threadIdx.x; u < w_units; u += 256) {
			if (NT) { __builtin_nontemporal_store(v, q + u); } else { q[u] = v; }
		}
	}
	if (x == 0x12345u) { *sink = x; pad[0] = x; }
}
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %d line %d\n", (int)r_, __LINE__); return 1; } } while (0)
int main() {
	const size_t out_bytes = 4ull << 30;
	const int    cus = 256;
	unsigned* sink; CK(hipMalloc(&sink, 4));
	u32x4 *in, *out; CK(hipMalloc(&in, out_bytes)); CK(hipMalloc(&out, out_bytes)); CK(hipMemset(in, 1, out_bytes));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	auto run = [&](int kind, size_t grid, unsigned lds, size_t n_chunks, unsigned r_units, unsigned w_units) {
		float best = 1e9f;
		for (int rep = 0; rep < 4; ++rep) {
			(void)hipEventRecord(a);
			switch (kind) {
			case 0: hipLaunchKernelGGL((k<true, true>), dim3(grid), dim3(256), lds, 0, in, out, n_chunks, r_units, w_units, sink); break;
			case 1: hipLaunchKernelGGL((k<false, true>), dim3(grid), dim3(256), lds, 0, in, out, n_chunks, r_units, w_units, sink); break;
			case 2: hipLaunchKernelGGL((k<true, false>), dim3(grid), dim3(256), lds, 0, in, out, n_chunks, r_units, w_units, sink); break;
			default: hipLaunchKernelGGL((k<false, false>), dim3(grid), dim3(256), lds, 0, in, out, n_chunks, r_units, w_units, sink); break;
			}
			(void)hipEventRecord(b); (void)hipEventSynchronize(b);
			float ms; (void)hipEventElapsedTime(&ms, a, b);
			if (rep > 0 && ms < best) { best = ms; }
		}
		return best;
	};
	printf("fractions of 8 TB/s over the bytes written (+ read)\n");
	printf("W KiB | persistent, wgs/CU 1 2 4 8: nt-write-only plain-write-only nt+reads plain+reads (per wgs) | dispatched, WGs/CU capped at 8 6 4 2: the same four\n");
	for (unsigned w_kib : {4u, 8u, 16u, 32u, 64u}) {
		const unsigned w_units = w_kib * 64, r_units = w_units / 10;
		const size_t   n_chunks = out_bytes / ((size_t)w_units * 16);
		const double   wb = (double)n_chunks * w_units * 16, rb = (double)n_chunks * r_units * 16;
		printf("%5u |", w_kib);
		for (int wgs : {1, 2, 4, 8}) {
			const size_t G = (size_t)cus * wgs;
			printf("  %.3f %.3f %.3f %.3f", wb / run(0, G, 0, n_chunks, 0, w_units) / 8e9, wb / run(1, G, 0, n_chunks, 0, w_units) / 8e9, (wb + rb) / run(0, G, 0, n_chunks, r_units, w_units) / 8e9,
			       (wb + rb) / run(1, G, 0, n_chunks, r_units, w_units) / 8e9);
		}
		printf(" |");
		for (unsigned cap : {8u, 6u, 4u, 2u}) {
			const unsigned lds = cap >= 8 ? 0u : (160u * 1024u / cap) - 1024u;
			printf("  %.3f %.3f %.3f %.3f", wb / run(2, n_chunks, lds, n_chunks, 0, w_units) / 8e9, wb / run(3, n_chunks, lds, n_chunks, 0, w_units) / 8e9, (wb + rb) / run(2, n_chunks, lds, n_chunks, r_units, w_units) / 8e9,
			       (wb + rb) / run(3, n_chunks, lds, n_chunks, r_units, w_units) / 8e9);
		}
		printf("\n"); fflush(stdout);
	}
	return 0;
}
