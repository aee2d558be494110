// lds_dma_probe.hip — where does global_load_lds put its data when a workgroup has two wavefronts per SIMD?
// Every wavefront of an 8-wavefront workgroup DMA-loads P 1-KiB pieces of a source whose 32-bit words are their own index into its own
// 16.5-KiB slice of the workgroup's LDS (pre-filled with a sentinel), waits for vmcnt(0) (+ optional sleep / barrier), and the whole
// LDS image goes to global memory.  The host reports, per wavefront, how many of its words arrived where they were sent, and where
// every stray word is.   hipcc --offload-arch=gfx950 -O3 -o tools/lds_dma_probe tools/lds_dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int kWaves = 8, kSlice = 16512, kWords = kWaves * kSlice / 4;
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int MODE> // 0: vmcnt(0) only; 1: + __syncthreads before the dump's own barrier (always there); 2: size-4 loads
__global__ __launch_bounds__(512) void probe(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, int P, int stagger) {
	__shared__ uint32_t lds[kWords];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	for (int i = threadIdx.x; i < kWords; i += 512) { lds[i] = 0xDEADBEEFu; }
	__syncthreads();
	if (stagger) { for (int z = 0; z < wave * stagger; ++z) { __builtin_amdgcn_s_sleep(20); } }
	if (MODE == 5) { asm volatile("v_mov_b32 v199, 0" ::: "v199"); } // the kernel allocates 200 VGPRs: the second wavefront of a SIMD sits at registers 200..399
	const size_t gw = (size_t)blockIdx.x * kWaves + wave;
	uint8_t* base = reinterpret_cast<uint8_t*>(lds) + wave * kSlice;
	for (int p = 0; p < P; ++p) {
		if (MODE == 2) {
			for (int q = 0; q < 4; ++q) { __builtin_amdgcn_global_load_lds(src + (gw * P + p) * 256 + 64 * q + lane, reinterpret_cast<uint32_t*>(base + 1024 * p + 256 * q), 4, 0, 0); }
		} else if (MODE == 6) { // the last piece with only its first 8 lanes active, like a vector whose packed words end inside a piece
			if (p + 1 < P || lane < 8) { __builtin_amdgcn_global_load_lds(reinterpret_cast<const u4*>(src + (gw * P + p) * 256) + lane, reinterpret_cast<u4*>(base + 1024 * p), 16, 0, 0); }
		} else {
			__builtin_amdgcn_global_load_lds(reinterpret_cast<const u4*>(src + (gw * P + p) * 256) + lane, reinterpret_cast<u4*>(base + 1024 * p), 16, 0, 0);
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	if (MODE == 3 || MODE == 4 || MODE == 5 || MODE == 6) { // no barrier: every wavefront copies its own slice out right behind its own wait (asm reads: no compiler-inserted waits)
		float keep[40];
		if (MODE == 4) { // ~200 live VGPRs across the DMA: two wavefronts of a SIMD then use 400 of its 512 registers
#pragma unroll
			for (int i = 0; i < 40; ++i) { asm volatile("v_mov_b32 %0, %1" : "=v"(keep[i]) : "v"(lane + i)); }
		}
		typedef __attribute__((address_space(3))) uint8_t lb;
		const uint32_t a0 = (uint32_t)(uintptr_t)((lb*)base);
		for (int i = lane; i < kSlice / 4; i += 64) {
			uint32_t v;
			asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a0 + 4 * i) : "memory");
			out[(size_t)blockIdx.x * kWords + wave * (kSlice / 4) + i] = v;
		}
		if (MODE == 4) {
			float acc = 0;
#pragma unroll
			for (int i = 0; i < 40; ++i) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(keep[i])); }
			if (acc == 12345.f) { out[0] = 1; }
		}
		return;
	}
	__syncthreads();
	for (int i = threadIdx.x; i < kWords; i += 512) { out[(size_t)blockIdx.x * kWords + i] = lds[i]; }
}
int main(int argc, char** argv) {
	const int grid = argc > 1 ? atoi(argv[1]) : 17, P = argc > 2 ? atoi(argv[2]) : 4, mode = argc > 3 ? atoi(argv[3]) : 0, stagger = argc > 4 ? atoi(argv[4]) : 0, reps = 20;
	const int shift_words = argc > 5 ? atoi(argv[5]) / 4 : 0; // the source starts that many bytes into its buffer: pieces then straddle 4-KiB pages
	const size_t n_src = (size_t)grid * kWaves * P * 256 + 4096;
	std::vector<uint32_t> h(n_src);
	for (size_t i = 0; i < n_src; ++i) { h[i] = (uint32_t)(i - shift_words); }
	uint32_t *d_src, *d_out;
	hipMalloc(&d_src, n_src * 4);
	hipMalloc(&d_out, (size_t)grid * kWords * 4);
	hipMemcpy(d_src, h.data(), n_src * 4, hipMemcpyHostToDevice);
	std::vector<uint32_t> o((size_t)grid * kWords);
	long bad_by_wave[kWaves] = {0}, stray = 0, total_bad_runs = 0;
	for (int r = 0; r < reps; ++r) {
		if (mode == 2) { hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(512), 0, 0, d_src + shift_words, d_out, P, stagger); }
		else if (mode == 3) { hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(512), 0, 0, d_src + shift_words, d_out, P, stagger); }
		else if (mode == 5) { hipLaunchKernelGGL(probe<5>, dim3(grid), dim3(512), 0, 0, d_src + shift_words, d_out, P, stagger); }
		else if (mode == 6) { hipLaunchKernelGGL(probe<6>, dim3(grid), dim3(512), 0, 0, d_src + shift_words, d_out, P, stagger); }
		else if (mode == 4) { hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(512), 0, 0, d_src + shift_words, d_out, P, stagger); }
		else { hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 0, 0, d_src + shift_words, d_out, P, stagger); }
		hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost);
		bool any = false;
		for (int b = 0; b < grid; ++b) {
			for (int w = 0; w < kWaves; ++w) {
				const uint32_t* img = o.data() + (size_t)b * kWords + w * (kSlice / 4);
				for (int i = 0; i < kSlice / 4; ++i) {
					uint32_t want = i < P * 256 ? (uint32_t)(((size_t)b * kWaves + w) * P * 256 + i) : 0xDEADBEEFu;
					if (mode == 6 && i >= (P - 1) * 256 + 32) { want = 0xDEADBEEFu; }
					if (img[i] != want) {
						if (!any && total_bad_runs < 3) {
							printf("rep %d block %d wave %d word %d (piece %d lane %d): found %08x want %08x", r, b, w, i, i / 256, (i % 256) / 4, img[i], want);
							if (img[i] != 0xDEADBEEFu) { const uint32_t s = img[i]; printf("  = source word of block %u wave %u piece %u word %u", s / (kWaves * P * 256), (s / (P * 256)) % kWaves, (s / 256) % P, s % 256); }
							printf("\n");
						}
						any = true;
						if (i < P * 256) { bad_by_wave[w]++; } else { stray++; }
					}
				}
			}
		}
		total_bad_runs += any;
	}
	printf("grid %d P %d mode %d stagger %d: runs with errors %ld of %d; missing words by wave:", grid, P, mode, stagger, total_bad_runs, reps);
	for (int w = 0; w < kWaves; ++w) { printf(" %ld", bad_by_wave[w]); }
	printf("; words outside their destination: %ld\n", stray);
	return 0;
}
