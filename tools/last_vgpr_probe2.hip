// last_vgpr_probe2.hip — round 4: the "last allocated register" rule of gfx950 over EVERY kind of instruction the shipped kernels use in which a
// single 32-bit VGPR operand sits beside 64-bit (or wider) VGPR operands (opcode set: llvm-objdump of libalpgpu.so's code objects, tools/check_top_vgpr.py
// --opcodes).  Round 3's probe (tools/last_vgpr_probe.hip) convicted the two 64-bit shifts and cleared four others; the checker's list must be what the
// hardware does, not what happened to be tried.  One kernel per register budget: __launch_bounds__(256, 8) allocates 64 registers, TOP = v63 is the last.
// Every test writes a per-lane value into TOP, executes ONE instruction that reads (or writes) TOP as its 32-bit operand, and compares with the same
// instruction executed with the operand in a low register.  Counters per test in `bad`; with -DMARGIN='"v64"' (72 registers allocated) nothing may differ.
//   hipcc --offload-arch=gfx950 -O3 -o last_vgpr_probe2 tools/last_vgpr_probe2.hip && ./last_vgpr_probe2 [workgroups] [launches]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#ifndef TOP
#define TOP "v63"
#endif
#ifndef MARGIN
#define MARGIN "v63"
#endif

enum {
	T_LSHRREV_B64, T_LSHLREV_B64, T_ASHRREV_I64, T_LSHL_ADD_U64, T_LDEXP_F64, T_CVT_F64_U32, T_CVT_F64_I32, T_CVT_F64_F32, T_CVT_U32_F64_DST, T_CVT_I32_F64_DST,
	T_MAD_U64_U32, T_MAD_I64_I32, T_CMP_CLASS_F64, T_TRIG_PREOP_F64, T_DS_READ_B64, T_DS_READ_B128, T_DS_WRITE_B64, T_DS_OR_B64, T_GLOBAL_LOAD_X2, T_GLOBAL_LOAD_X4,
	T_GLOBAL_STORE_X4, T_BUFFER_LOAD_X4, T_FREXP_EXP_DST, T_CVT_F32_F64_DST, T_READBACK, T_COUNT
};
static const char* kNames[T_COUNT] = {
    "v_lshrrev_b64 (amount)", "v_lshlrev_b64 (amount)", "v_ashrrev_i64 (amount)", "v_lshl_add_u64 (amount)", "v_ldexp_f64 (exponent)", "v_cvt_f64_u32 (source)",
    "v_cvt_f64_i32 (source)", "v_cvt_f64_f32 (source)", "v_cvt_u32_f64 (32-bit DESTINATION)", "v_cvt_i32_f64 (32-bit DESTINATION)", "v_mad_u64_u32 (factor)",
    "v_mad_i64_i32 (factor)", "v_cmp_class_f64 (class mask)", "v_trig_preop_f64 (segment)", "ds_read_b64 (address)", "ds_read_b128 (address)", "ds_write_b64 (address)",
    "ds_or_b64 (address)", "global_load_dwordx2 saddr (offset)", "global_load_dwordx4 saddr (offset)", "global_store_dwordx4 saddr (offset)",
    "buffer_load_dwordx4 offen (offset)", "v_frexp_exp_i32_f64 (32-bit DESTINATION)", "v_cvt_f32_f64 (32-bit DESTINATION)", "marker read back"};

#define PUT_TOP(val) asm volatile("v_mov_b32 " TOP ", %0" ::"v"(val) : TOP)
#define MISS(t) atomicAdd(bad + (t), 1u)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256, 8) void k_probe2(const uint32_t* __restrict__ gmem, uint32_t* __restrict__ gout, uint32_t* __restrict__ bad, int rounds) {
	__shared__ uint64_t lds[4][160]; // 1280 B per wavefront
	asm volatile("" ::: MARGIN);
	const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	uint64_t*      L    = lds[wave];
	L[lane] = 0x1111111111111111ull * (lane & 15) + lane, L[lane + 64] = ~static_cast<uint64_t>(lane) * 0x9E3779B97F4A7C15ull;
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	const uint32_t lbase = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(L)); // LDS byte address of this wavefront's area
	// buffer resource over gmem (4 KiB window)
	const uint64_t ga = reinterpret_cast<uint64_t>(gmem);
	const u32x4    rsrc = {static_cast<uint32_t>(ga), static_cast<uint32_t>(ga >> 32) & 0xFFFFu, 1u << 20, 0x00020000u};
	for (int r = 0; r < rounds; ++r) {
		const uint32_t sh = (lane * 5 + 3 + r) & 63;
		const uint64_t xx = 0x0123456789ABCDEFull ^ (static_cast<uint64_t>(blockIdx.x * 256 + threadIdx.x + r) << 20);
		// ---- shifts ----
		uint64_t a, b;
		PUT_TOP(sh);
		asm volatile("v_lshrrev_b64 %0, " TOP ", %1" : "=&v"(a) : "v"(xx) : TOP);
		if (a != (xx >> sh)) { MISS(T_LSHRREV_B64); }
		asm volatile("v_lshlrev_b64 %0, " TOP ", %1" : "=&v"(a) : "v"(xx) : TOP);
		if (a != (xx << sh)) { MISS(T_LSHLREV_B64); }
		asm volatile("v_ashrrev_i64 %0, " TOP ", %1" : "=&v"(a) : "v"(xx) : TOP);
		if (a != static_cast<uint64_t>(static_cast<int64_t>(xx) >> sh)) { MISS(T_ASHRREV_I64); }
		PUT_TOP(sh & 3u);
		asm volatile("v_lshl_add_u64 %0, %1, " TOP ", %2" : "=&v"(a) : "v"(xx), "v"(xx) : TOP);
		if (a != (xx << (sh & 3u)) + xx) { MISS(T_LSHL_ADD_U64); }
		// ---- 32-bit source beside a 64-bit result ----
		double d;
		PUT_TOP(sh & 31u);
		asm volatile("v_ldexp_f64 %0, %1, " TOP : "=&v"(d) : "v"(1.5) : TOP);
		if (d != __builtin_ldexp(1.5, static_cast<int>(sh & 31u))) { MISS(T_LDEXP_F64); }
		PUT_TOP(sh * 77u + 5u);
		asm volatile("v_cvt_f64_u32 %0, " TOP : "=&v"(d) : : TOP);
		if (d != static_cast<double>(sh * 77u + 5u)) { MISS(T_CVT_F64_U32); }
		const int neg = -static_cast<int>(sh) * 31 - 7;
		PUT_TOP(neg);
		asm volatile("v_cvt_f64_i32 %0, " TOP : "=&v"(d) : : TOP);
		if (d != static_cast<double>(neg)) { MISS(T_CVT_F64_I32); }
		const float fl = 0.25f * static_cast<float>(sh) + 1.0f;
		PUT_TOP(fl);
		asm volatile("v_cvt_f64_f32 %0, " TOP : "=&v"(d) : : TOP);
		if (d != static_cast<double>(fl)) { MISS(T_CVT_F64_F32); }
		// ---- 64-bit source, 32-bit result IN the top register ----
		uint32_t got;
		const double big = 1000.0 * sh + 0.75;
		asm volatile("v_cvt_u32_f64 " TOP ", %1\n\tv_mov_b32 %0, " TOP : "=v"(got) : "v"(big) : TOP);
		if (got != static_cast<uint32_t>(big)) { MISS(T_CVT_U32_F64_DST); }
		asm volatile("v_cvt_i32_f64 " TOP ", %1\n\tv_mov_b32 %0, " TOP : "=v"(got) : "v"(-big) : TOP);
		if (static_cast<int>(got) != static_cast<int>(-big)) { MISS(T_CVT_I32_F64_DST); }
		asm volatile("v_frexp_exp_i32_f64 " TOP ", %1\n\tv_mov_b32 %0, " TOP : "=v"(got) : "v"(big) : TOP);
		{
			int e;
			(void)__builtin_frexp(big, &e);
			if (static_cast<int>(got) != e) { MISS(T_FREXP_EXP_DST); }
		}
		asm volatile("v_cvt_f32_f64 " TOP ", %1\n\tv_mov_b32 %0, " TOP : "=v"(got) : "v"(big) : TOP);
		if (__uint_as_float(got) != static_cast<float>(big)) { MISS(T_CVT_F32_F64_DST); }
		// ---- multiply-adds ----
		PUT_TOP(sh + 9u);
		asm volatile("v_mad_u64_u32 %0, vcc, " TOP ", %1, %2" : "=&v"(a) : "v"(7u), "v"(xx) : TOP, "vcc");
		if (a != static_cast<uint64_t>(sh + 9u) * 7u + xx) { MISS(T_MAD_U64_U32); }
		PUT_TOP(neg);
		asm volatile("v_mad_i64_i32 %0, vcc, " TOP ", %1, %2" : "=&v"(a) : "v"(11), "v"(xx) : TOP, "vcc");
		if (a != static_cast<uint64_t>(static_cast<int64_t>(neg) * 11 + static_cast<int64_t>(xx))) { MISS(T_MAD_I64_I32); }
		// ---- class test: 32-bit mask beside a 64-bit value ----
		{
			const uint32_t mask = (lane & 1) ? 0x3FFu : 0x200u; // all classes / +inf only
			PUT_TOP(mask);
			uint64_t res;
			asm volatile("v_cmp_class_f64 %0, %1, " TOP : "=s"(res) : "v"(big) : TOP);
			const bool mine = (res >> lane) & 1ull;
			if (mine != ((lane & 1) != 0)) { MISS(T_CMP_CLASS_F64); }
			PUT_TOP(lane & 3u);
			double lo, ref;
			asm volatile("v_trig_preop_f64 %0, %1, " TOP : "=&v"(lo) : "v"(big) : TOP);
			uint32_t seg = lane & 3u;
			asm volatile("v_trig_preop_f64 %0, %1, %2" : "=&v"(ref) : "v"(big), "v"(seg));
			if (__double_as_longlong(lo) != __double_as_longlong(ref)) { MISS(T_TRIG_PREOP_F64); }
		}
		// ---- LDS: 32-bit address in the top register, 64 / 128-bit data ----
		{
			const uint32_t addr = lbase + 8u * ((lane + r) & 63u);
			PUT_TOP(addr);
			asm volatile("ds_read_b64 %0, " TOP "\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a) : : TOP, "memory");
			if (a != L[(lane + r) & 63u]) { MISS(T_DS_READ_B64); }
			const uint32_t addr2 = lbase + 16u * ((lane + r) & 31u);
			PUT_TOP(addr2);
			u32x4 q;
			asm volatile("ds_read_b128 %0, " TOP "\n\ts_waitcnt lgkmcnt(0)" : "=&v"(q) : : TOP, "memory");
			const uint64_t w0 = L[2 * ((lane + r) & 31u)], w1 = L[2 * ((lane + r) & 31u) + 1];
			if (q.x != static_cast<uint32_t>(w0) || q.y != static_cast<uint32_t>(w0 >> 32) || q.z != static_cast<uint32_t>(w1) || q.w != static_cast<uint32_t>(w1 >> 32)) { MISS(T_DS_READ_B128); }
			const uint32_t addr3 = lbase + 8u * (128u + (lane & 31u));
			PUT_TOP(addr3);
			asm volatile("ds_write_b64 " TOP ", %0\n\ts_waitcnt lgkmcnt(0)" ::"v"(xx) : TOP, "memory");
			if (L[128 + (lane & 31u)] != xx && lane >= 32) { MISS(T_DS_WRITE_B64); } // (lanes l and l + 32 share a slot: the upper lane's value stays)
			L[128 + (lane & 31u)] = 0;
			__builtin_amdgcn_wave_barrier();
			PUT_TOP(addr3);
			asm volatile("ds_or_b64 " TOP ", %0\n\ts_waitcnt lgkmcnt(0)" ::"v"(1ull << lane) : TOP, "memory");
			__builtin_amdgcn_wave_barrier();
			if (L[128 + (lane & 31u)] != ((1ull << (lane & 31u)) | (1ull << ((lane & 31u) + 32)))) { MISS(T_DS_OR_B64); }
		}
		// ---- global / buffer: scalar base + 32-bit offset in the top register, 64 / 128-bit data ----
		{
			const uint32_t off = 16u * ((lane * 3 + r) & 127u);
			PUT_TOP(off);
			u32x2 g2;
			asm volatile("global_load_dwordx2 %0, " TOP ", %1\n\ts_waitcnt vmcnt(0)" : "=&v"(g2) : "s"(gmem) : TOP, "memory");
			if (g2.x != gmem[off / 4] || g2.y != gmem[off / 4 + 1]) { MISS(T_GLOBAL_LOAD_X2); }
			u32x4 g4;
			asm volatile("global_load_dwordx4 %0, " TOP ", %1\n\ts_waitcnt vmcnt(0)" : "=&v"(g4) : "s"(gmem) : TOP, "memory");
			if (g4.x != gmem[off / 4] || g4.w != gmem[off / 4 + 3]) { MISS(T_GLOBAL_LOAD_X4); }
			asm volatile("buffer_load_dwordx4 %0, " TOP ", %1, 0 offen\n\ts_waitcnt vmcnt(0)" : "=&v"(g4) : "s"(rsrc) : TOP, "memory");
			if (g4.x != gmem[off / 4] || g4.w != gmem[off / 4 + 3]) { MISS(T_BUFFER_LOAD_X4); }
			uint32_t* mine = gout + (static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
			const uint32_t soff = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(mine) - reinterpret_cast<uintptr_t>(gout));
			PUT_TOP(soff);
			const u32x4 sv = {lane, static_cast<uint32_t>(r), 0xABCD0000u + lane, soff};
			asm volatile("global_store_dwordx4 " TOP ", %0, %1\n\ts_waitcnt vmcnt(0)" ::"v"(sv), "s"(gout) : TOP, "memory");
			if (__builtin_nontemporal_load(mine + 2) != 0xABCD0000u + lane) { MISS(T_GLOBAL_STORE_X4); }
		}
		// the register itself
		const uint32_t marker = 0xA5000000u | (blockIdx.x * 256u + threadIdx.x);
		PUT_TOP(marker);
		asm volatile("s_nop 4\n\tv_mov_b32 %0, " TOP : "=v"(got)::TOP);
		if (got != marker) { MISS(T_READBACK); }
	}
}

int main(int argc, char** argv) {
	const unsigned wgs = argc > 1 ? atoi(argv[1]) : 2048, reps = argc > 2 ? atoi(argv[2]) : 20;
	uint32_t *d_g, *d_o, *d_bad;
	hipMalloc(&d_g, 1 << 20);
	hipMalloc(&d_o, static_cast<size_t>(wgs) * 256 * 16);
	hipMalloc(&d_bad, 4 * T_COUNT);
	uint32_t* h = static_cast<uint32_t*>(malloc(1 << 20));
	for (unsigned i = 0; i < (1u << 18); ++i) { h[i] = i * 2654435761u + 12345u; }
	hipMemcpy(d_g, h, 1 << 20, hipMemcpyHostToDevice);
	hipMemset(d_bad, 0, 4 * T_COUNT);
	for (unsigned r = 0; r < reps; ++r) { hipLaunchKernelGGL(k_probe2, dim3(wgs), dim3(256), 0, 0, d_g, d_o, d_bad, 8); }
	hipDeviceSynchronize();
	uint32_t bad[T_COUNT];
	hipMemcpy(bad, d_bad, 4 * T_COUNT, hipMemcpyDeviceToHost);
	const unsigned long long lanes = static_cast<unsigned long long>(wgs) * 256ull * reps * 8ull;
	printf("top register " TOP ", margin " MARGIN ": %u workgroups x %u launches x 8 rounds = %llu lane-tests per instruction (%s)\n", wgs, reps, lanes, hipGetErrorString(hipGetLastError()));
	for (int t = 0; t < T_COUNT; ++t) { printf("  %-44s %10u wrong%s\n", kNames[t], bad[t], bad[t] ? "   <-- CONVICTED" : ""); }
	return 0;
}
