// last_vgpr_probe.hip — stand-alone probe (no library) for a gfx950 behaviour found while chasing wrong sums in k_sink_direct
// (profiles/r03_consumers.txt): a 64-bit shift (v_lshrrev_b64 / v_lshlrev_b64) whose SHIFT AMOUNT — a single 32-bit register — sits in the
// LAST register of the wavefront's allocation computes with VGPR0 instead.  The operand seems to be range-checked as a register pair
// (v63:v64 of a 64-register allocation = out of range -> the documented substitution of VGPR0).
// Short-lived wavefronts (like k_sink_direct's) park a marker in the top register, run loads / LDS traffic / lane-masked regions that never
// touch it, read it back (always intact), and use it as the shift amount of 64-bit shifts (wrong for ~40 % of the lanes) and as the 32-bit
// operand of v_ldexp_f64, v_cvt_f64_u32, v_mad_u64_u32, v_lshl_add_u64 (never wrong).  With one more register allocated (-DMARGIN='"v64"')
// nothing is ever wrong.  One MI355X, ROCm 7.2:
//   ./last_vgpr_probe 2500 200    -> of 128000000 lanes 0 read back a different value, 51477462 got a wrong 64-bit shift by it
//   (MARGIN "v64")                -> 0, 0
//   hipcc --offload-arch=gfx950 -O3 -o last_vgpr_probe tools/last_vgpr_probe.hip && ./last_vgpr_probe [workgroups] [launches]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef TOP
#define TOP "v63" // the allocation's last register for __launch_bounds__(256, 8)
#endif
#ifndef MARGIN
#define MARGIN "v63" // "v64": one register more is allocated, TOP is no longer the last
#endif


// the top register as the shift amount of 64-bit shifts (what the failing builds of k_sink_direct kept there); the marker is put back after
#define SHIFT_TEST(STEP)                                                                                                              \
	{                                                                                                                             \
		const uint32_t sh = (lane * 5 + 3 + (STEP)) & 63;                                                                         \
		asm volatile("v_mov_b32 " TOP ", %0" ::"v"(sh) : TOP);                                                                    \
		const uint64_t xx = 0x0123456789ABCDEFull ^ (static_cast<uint64_t>(marker) << 20);                                        \
		uint64_t       r0, r1;                                                                                                    \
		asm volatile("v_lshrrev_b64 %0, " TOP ", %2\n\tv_lshlrev_b64 %1, " TOP ", %2" : "=&v"(r0), "=&v"(r1) : "v"(xx) : TOP);    \
		if (r0 != (xx >> sh) || r1 != (xx << sh)) {                                                                               \
			atomicAdd(bad + 1, 1u);                                                                                               \
			if (lane == 1) {                                                                                                      \
				uint32_t hw;                                                                                                      \
				asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                                \
				const uint32_t k = atomicAdd(bad + 3, 1u);                                                                        \
				if (k < 8) {                                                                                                      \
					bad[4 + 3 * k] = hw;                                                                                          \
					uint32_t eff   = 99;                                                                                          \
					for (uint32_t t = 0; t < 64; ++t) {                                                                           \
						if (r0 == (xx >> t)) { eff = t; }                                                                         \
					}                                                                                                             \
					bad[5 + 3 * k] = (eff << 16) | (sh << 8) | (STEP);                                                            \
					bad[6 + 3 * k] = static_cast<uint32_t>(r0);                                                                   \
				}                                                                                                                 \
			}                                                                                                                     \
		}                                                                                                                         \
		asm volatile("v_mov_b32 " TOP ", %0" ::"v"(marker) : TOP);                                                                \
	}

__global__ __launch_bounds__(256, 8) void k_probe(const uint4* __restrict__ in, size_t n_units, const uint16_t* __restrict__ pos, uint32_t* __restrict__ bad,
                                                  uint64_t* __restrict__ sink) {
	__shared__ uint32_t lds[4][1248]; // ~5 KB per wavefront, as k_sink_direct
	asm volatile("" ::: MARGIN);
	const int      lane   = threadIdx.x & 63;
	const int      wave   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t marker = 0xA5000000u | (blockIdx.x * 256u + threadIdx.x);
	asm volatile("v_mov_b32 " TOP ", %0" ::"v"(marker) : TOP);
	uint32_t* L   = lds[wave];
	uint64_t  acc = 0;
	const size_t v = static_cast<size_t>(blockIdx.x) * 4 + wave;
	// "exception prologue": LDS-DMA, zero, positions -> atomics, mask read
	const uint32_t* src = reinterpret_cast<const uint32_t*>(in + (v * 96) % n_units);
	__builtin_amdgcn_global_load_lds(src + lane, L + 64, 4, 0, 0);
	if (lane < 32) { L[lane] = 0u; }
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	const uint32_t p = pos[(v * 64 + lane) & 0xFFFF];
	atomicOr(&L[(p >> 5) & 31], 1u << (p & 31u));
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__builtin_amdgcn_wave_barrier();
	const uint32_t word = L[lane & 31];
	// eight "steps": two 16-byte loads each, a lookup out of LDS, a lane-masked patch
	const uint4* g = in + (v * 96) % n_units;
#pragma unroll
	for (int b = 0; b < 8; b += 4) {
		uint4 a[4], c[4];
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			a[i] = g[(8 * (b + i) + (lane & 7) + 8 * (lane >> 3)) % 88];
			c[i] = g[(8 * (b + i) + (lane & 7) + 8 * (lane >> 3)) % 88 + 8];
		}
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int      s  = (lane * 7 + (b + i) * 5) & 63;
			uint64_t       x  = ((static_cast<uint64_t>(a[i].y) << 32 | a[i].x) >> s) | ((static_cast<uint64_t>(c[i].y) << 32 | c[i].x) << 1 << (63 - s));
			const uint32_t w  = L[4 * (b + i) + (lane >> 4)];
			const uint32_t pf = L[64 + 4 * (b + i) + (lane >> 4)];
			if ((w >> (lane & 31)) & 1u) { x = reinterpret_cast<const uint64_t*>(L + 64)[(pf + lane) & 127]; }
			acc += x & 0xFFFFFFFull;
			SHIFT_TEST(b + i);
		}
	}
	uint32_t got;
	asm volatile("v_mov_b32 %0, " TOP : "=v"(got)::TOP);
	if (got != marker) { atomicAdd(bad, 1u); }
	// other 64-bit instructions with a 32-bit operand in the top register (counters 28..31)
	{
		const uint32_t k = (lane * 3 + 5) & 31;
		asm volatile("v_mov_b32 " TOP ", %0" ::"v"(k) : TOP);
		double   d, e;
		uint64_t m, a;
		const double   one = 1.5;
		const uint64_t big = 0x0000000100000003ull;
		asm volatile("v_ldexp_f64 %0, %1, " TOP : "=&v"(d) : "v"(one) : TOP);                   // src1 = exponent (32-bit)
		asm volatile("v_cvt_f64_u32 %0, " TOP : "=&v"(e) : : TOP);                               // 32-bit source, 64-bit result
		asm volatile("v_mad_u64_u32 %0, vcc, " TOP ", %1, %2" : "=&v"(m) : "v"(7u), "v"(big) : TOP, "vcc"); // 32 x 32 + 64
		asm volatile("v_lshl_add_u64 %0, %1, " TOP ", %2" : "=&v"(a) : "v"(big), "v"(big) : TOP);  // (64 << 32-bit) + 64  (shift amounts 0..4 only)
		if (d != __builtin_ldexp(1.5, static_cast<int>(k))) { atomicAdd(bad + 28, 1u); }
		if (e != static_cast<double>(k)) { atomicAdd(bad + 29, 1u); }
		if (m != static_cast<uint64_t>(k) * 7u + big) { atomicAdd(bad + 30, 1u); }
		asm volatile("v_mov_b32 " TOP ", %0" ::"v"(k & 3u) : TOP);
		asm volatile("v_lshl_add_u64 %0, %1, " TOP ", %2" : "=&v"(a) : "v"(big), "v"(big) : TOP);
		if (a != (big << (k & 3u)) + big) { atomicAdd(bad + 31, 1u); }
		asm volatile("v_mov_b32 " TOP ", %0" ::"v"(marker) : TOP);
	}
	SHIFT_TEST(8);
	if (acc == 0x123456789ull || word == 0xdeadbeefu) { sink[0] = acc; }
}

int main(int argc, char** argv) {
	const unsigned wgs = argc > 1 ? atoi(argv[1]) : 65536, reps = argc > 2 ? atoi(argv[2]) : 20;
	const size_t   n_units = 1u << 24; // 256 MiB
	uint4*         d_in;
	uint16_t*      d_pos;
	uint32_t*      d_bad;
	uint64_t*      d_sink;
	hipMalloc(&d_in, n_units * 16 + 4096);
	hipMalloc(&d_pos, 65536 * 2);
	hipMalloc(&d_bad, 128);
	hipMalloc(&d_sink, 8);
	hipMemset(d_in, 0x5a, n_units * 16 + 4096);
	std::vector<uint16_t> pos(65536);
	for (size_t i = 0; i < pos.size(); ++i) { pos[i] = static_cast<uint16_t>((i * 2654435761u) >> 22); }
	hipMemcpy(d_pos, pos.data(), pos.size() * 2, hipMemcpyHostToDevice);
	hipMemset(d_bad, 0, 128);
	for (unsigned r = 0; r < reps; ++r) { hipLaunchKernelGGL(k_probe, dim3(wgs), dim3(256), 0, 0, d_in, n_units - 256, d_pos, d_bad, d_sink); }
	hipDeviceSynchronize();
	uint32_t bad[32] = {0};
	hipMemcpy(bad, d_bad, 128, hipMemcpyDeviceToHost);
	printf("top register " TOP ", margin " MARGIN ", %u workgroups x %u launches: of %llu lanes %u read back a different value, %u got a wrong 64-bit shift by it"
	       " (%u), %u wavefront-steps (%s)\n",
	       wgs, reps, static_cast<unsigned long long>(wgs) * 256ull * reps, bad[0], bad[1], bad[2], bad[3], hipGetErrorString(hipGetLastError()));
	printf("   other 64-bit instructions with their 32-bit operand in " TOP ": v_ldexp_f64 %u wrong, v_cvt_f64_u32 %u, v_mad_u64_u32 %u, v_lshl_add_u64 %u\n", bad[28], bad[29], bad[30],
	       bad[31]);
	for (unsigned k = 0; k < bad[3] && k < 8; ++k) {
		const uint32_t h = bad[4 + 3 * k], e = bad[5 + 3 * k];
		printf("   step %u lane 1: shift wanted %u, result = shift by %u (99: by none); HW_ID %08x: wave %u simd %u cu %u se %u\n", e & 255u, (e >> 8) & 255u, e >> 16, h, h & 15u,
		       (h >> 4) & 3u, (h >> 8) & 15u, (h >> 13) & 7u);
	}
	return 0;
}
