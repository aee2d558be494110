// alp_device.hpp — shared device-side building blocks for the gfx950 ALP kernels.
//
// Execution model used by every kernel in this directory: ONE 64-lane wavefront owns ONE 1024-value
// vector.  Value i of the vector is handled by lane (i >> 1) & 63 in step m = i >> 7, i.e. a lane owns
// the value PAIR (128*m + 2*lane, +1) for m = 0..7, so that every global access to the doubles is a
// 16-byte-per-lane, 1-KiB-per-instruction contiguous transaction.
//
// FastLanes u64 layout (reference src/fastlanes_generated_ffor.cpp:7379-29749, closed form SURVEY.md §8 A9):
//   value i -> lane16 = i & 15, row = i >> 4; the lane's stream is the LSB-first concatenation of its 64
//   bw-bit fields; stream word k lives at packed[16*k + lane16].
// With the pair mapping a lane's two values share the row (8*m + (lane >> 3)) and sit in adjacent
// lane16 columns (2*(lane & 7), +1), so their packed words are one aligned 16-byte unit.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/alpgpu.h"

namespace alpgpu {

constexpr int kVec        = 1024;
constexpr int kRowgroup   = 100;
#ifndef ALPGPU_WAVES_PER_WG
#define ALPGPU_WAVES_PER_WG 4
#endif
constexpr int kWavesPerWg = ALPGPU_WAVES_PER_WG; // wavefronts (= vectors) per workgroup of the wave-per-vector kernels
// The single-pass encode kernels use tiles of 8 vectors: half as many tiles to order as with 4 (3.24 against 3.49 ms per 1 Mi
// vectors; 2: 4.56, 3: 3.82, 6: 4.22, 16: 4.03 — measured at 95 VGPRs, where two such workgroups fit a CU)
#ifndef ALPGPU_FUSED_WAVES
#define ALPGPU_FUSED_WAVES 8
#endif
constexpr int kFusedWaves = ALPGPU_FUSED_WAVES; // wavefronts (= vectors) per tile of k_encode_fused / k_encode_fused_f32

// ---- constants (reference include/alp/constants.hpp:66-154): bit-identical tables --------------------
__device__ __constant__ const double kFracArr[21] = {
    1.0,
    0.1,
    0.01,
    0.001,
    0.0001,
    0.00001,
    0.000001,
    0.0000001,
    0.00000001,
    0.000000001,
    0.0000000001,
    0.00000000001,
    0.000000000001,
    0.0000000000001,
    0.00000000000001,
    0.000000000000001,
    0.0000000000000001,
    0.00000000000000001,
    0.000000000000000001,
    0.0000000000000000001,
    0.00000000000000000001,
};
__device__ __constant__ const double kExpArr[24] = {
    1.0,
    10.0,
    100.0,
    1000.0,
    10000.0,
    100000.0,
    1000000.0,
    10000000.0,
    100000000.0,
    1000000000.0,
    10000000000.0,
    100000000000.0,
    1000000000000.0,
    10000000000000.0,
    100000000000000.0,
    1000000000000000.0,
    10000000000000000.0,
    100000000000000000.0,
    1000000000000000000.0,
    10000000000000000000.0,
    100000000000000000000.0,
    1000000000000000000000.0,
    10000000000000000000000.0,
    100000000000000000000000.0,
};
__device__ __constant__ const int64_t kFactArr[19] = {1LL,
                                                      10LL,
                                                      100LL,
                                                      1000LL,
                                                      10000LL,
                                                      100000LL,
                                                      1000000LL,
                                                      10000000LL,
                                                      100000000LL,
                                                      1000000000LL,
                                                      10000000000LL,
                                                      100000000000LL,
                                                      1000000000000LL,
                                                      10000000000000LL,
                                                      100000000000000LL,
                                                      1000000000000000LL,
                                                      10000000000000000LL,
                                                      100000000000000000LL,
                                                      1000000000000000000LL};

constexpr double kMagic      = 6755399441055744.0;     // 2^52 + 2^51, constants.hpp:70
constexpr double kUpperLimit = 9223372036854774784.0;  // constants.hpp:17
constexpr double kLowerLimit = -9223372036854774784.0; // constants.hpp:18

// ---- wave helpers -----------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & 63; }
__device__ __forceinline__ int wave_in_wg() { return __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6); }

// A predicate as a wave-uniform lane mask.  __ballot(int) of the HIP headers compares a VGPR with 0; the intrinsic takes the
// i1 as it is.  Ballot a COMPARE and combine masks as integers: the ballot of a compare is the compare's own SGPR result,
// while a predicate built from several compares is first materialised per lane (v_cndmask) and compared again.
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// base + number of set bits of `mask` below this lane (v_mbcnt_lo / v_mbcnt_hi: two instructions, the mask stays scalar)
__device__ __forceinline__ uint32_t mbcnt64(uint64_t mask, uint32_t base) {
	return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), base));
}

// Orders this wave's LDS traffic (and the compiler) around an intra-wave exchange.  The LDS unit executes
// one wave's DS operations in issue order, so no hardware barrier is needed between lanes of one wave.
__device__ __forceinline__ void wave_lds_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Inclusive add-scan over the 64 lanes as DPP row shifts + row broadcasts (register to register; lane 63 ends up with the total).
// A __shfl_up / __shfl_xor step is a ds_bpermute, i.e. a round trip through the LDS crossbar; chains of those sat on the
// critical paths of the rowgroup search and of the look-back.
__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t x) {
	int v = static_cast<int>(x);
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); // row_shr:1
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); // row_shr:2
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); // row_shr:4
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); // row_shr:8
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
	v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
	return static_cast<uint32_t>(v);
}

// 64 lane partials -> their sum at a balanced binary tree over adjacent lanes (DPP: row_shr 1, 2, 4, 8, then row broadcasts); the value
// of lane 63 is returned wave-uniform.  Lane 63's operands are, level by level, the sums of lanes {62,63}, {60..63}, {56..63}, {48..63},
// {32..63}, {0..63}: each level adds two neighbouring subtrees of equal size.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take_f64(double v) {
	const uint64_t b  = static_cast<uint64_t>(__double_as_longlong(v));
	int            lo = static_cast<int>(static_cast<uint32_t>(b)), hi = static_cast<int>(static_cast<uint32_t>(b >> 32));
	lo                = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
	hi                = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
	return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(static_cast<uint32_t>(hi)) << 32) | static_cast<uint32_t>(lo)));
}
__device__ __forceinline__ double wave_tree_sum_f64(double v) {
	v = v + dpp_take_f64<0x111, 0xf>(v); // row_shr:1
	v = v + dpp_take_f64<0x112, 0xf>(v); // row_shr:2
	v = v + dpp_take_f64<0x114, 0xf>(v); // row_shr:4
	v = v + dpp_take_f64<0x118, 0xf>(v); // row_shr:8   -> lane 15 of every row: the row's tree
	v = v + dpp_take_f64<0x142, 0xa>(v); // row_bcast:15 into rows 1 and 3
	v = v + dpp_take_f64<0x143, 0xc>(v); // row_bcast:31 into rows 2 and 3 -> lane 63
	const uint64_t b  = static_cast<uint64_t>(__double_as_longlong(v));
	const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(b), 63), hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(b >> 32), 63);
	return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(hi) << 32) | lo));
}

__device__ __forceinline__ uint64_t bw_mask(int bw) { return bw >= 64 ? ~0ULL : ((1ULL << bw) - 1ULL); }

__device__ __forceinline__ uint64_t uniform_u64(uint64_t x) {
	const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(x));
	const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(x >> 32));
	return (static_cast<uint64_t>(hi) << 32) | lo;
}

// The 32-byte rowgroup state, fetched ONCE per wavefront: one 32-byte read (lane i < 8 takes word i), then register moves.
// Read field by field from memory, the state cost the encode kernels a chain of dependent one-byte vector loads (scheme, then k,
// then the candidates ...), each a full memory round trip in front of the arithmetic.  The copy must only ever be indexed with
// compile-time constants (it lives in registers).
__device__ __forceinline__ alpgpu_rowgroup_state load_rowgroup_state(const alpgpu_rowgroup_state* __restrict__ p, int lane) {
	static_assert(sizeof(alpgpu_rowgroup_state) == 32, "eight words");
	const uint32_t mine = reinterpret_cast<const uint32_t*>(p)[lane & 7];
	uint32_t       w[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) { w[i] = __builtin_amdgcn_readlane(mine, i); }
	alpgpu_rowgroup_state s;
	s.scheme = static_cast<uint8_t>(w[0]);
	s.k      = static_cast<uint8_t>(w[0] >> 8);
#pragma unroll
	for (int i = 0; i < 10; ++i) { s.combos[i] = static_cast<uint8_t>(w[(2 + i) >> 2] >> (8 * ((2 + i) & 3))); }
	s.rd_rbw       = static_cast<uint8_t>(w[3]);
	s.rd_lbw       = static_cast<uint8_t>(w[3] >> 8);
	s.rd_dict_size = static_cast<uint8_t>(w[3] >> 16);
	s.pad          = static_cast<uint8_t>(w[3] >> 24);
#pragma unroll
	for (int i = 0; i < 8; ++i) { s.rd_dict[i] = static_cast<uint16_t>(w[4 + (i >> 1)] >> (16 * (i & 1))); }
	return s;
}

// The same from a state that another kernel — the persistent rowgroup search on a second stream (init_kernels.hip, ASYNC) — is
// publishing while this one runs.  The eight words are read with agent-scope loads (they bypass this CU's L1 and this XCD's L2, where
// stale lines of an earlier call could sit) in ONE 32-byte request; the tag (pad byte = kStateReady, stored last by the publisher after
// the other words had reached memory) says whether they are there.  Polls with s_sleep in between; gives up after spin_limit polls
// (ok = false: the caller raises the encode's stall flag and the recovery route re-encodes).  The returned state has pad = 0.
constexpr uint32_t kStateReady = 0xA5u;
__device__ __forceinline__ alpgpu_rowgroup_state unpack_rowgroup_state(uint32_t mine) {
	uint32_t w[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) { w[i] = __builtin_amdgcn_readlane(mine, i); }
	alpgpu_rowgroup_state s;
	s.scheme = static_cast<uint8_t>(w[0]);
	s.k      = static_cast<uint8_t>(w[0] >> 8);
#pragma unroll
	for (int i = 0; i < 10; ++i) { s.combos[i] = static_cast<uint8_t>(w[(2 + i) >> 2] >> (8 * ((2 + i) & 3))); }
	s.rd_rbw       = static_cast<uint8_t>(w[3]);
	s.rd_lbw       = static_cast<uint8_t>(w[3] >> 8);
	s.rd_dict_size = static_cast<uint8_t>(w[3] >> 16);
	s.pad          = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) { s.rd_dict[i] = static_cast<uint16_t>(w[4 + (i >> 1)] >> (16 * (i & 1))); }
	return s;
}
// Split in two so that the poll can be issued in front of the vector's own loads (vector-memory loads return in order: a poll issued behind
// them cannot be looked at before all 8 KiB have arrived).  begin: one agent-scope load per lane (word lane & 7).  finish: looks at it and polls
// on while the state is not there.
// "There" (nothing in the memory model says that the 32-byte request that saw the tag was not served BETWEEN the publisher's
// stores): the caller's memset leaves every unpublished state as all-ones bytes (kStateUnpublished); the publisher stores each of the four
// 8-byte words exactly once, the tagged one last; and no word of a real state is all-ones — word 0 starts with the scheme (1 or 2), words 2 and
// 3 hold four dictionary entries each, which are distinct where used and zero where not.  A read is therefore whole iff its tag is kStateReady
// AND none of its other words is still all-ones: a torn read is seen as such and simply polled again.  (A second, dependent read of the state
// after the tag — the first answer to the advice — cost 3 % of the encode: two far round trips in a row at the head of every wavefront.)
// (kStateUnpublished = 0xFF, the memset value of the states' buffer in front of a publishing search: launch.hpp)
__device__ __forceinline__ uint32_t rowgroup_state_poll_begin(const alpgpu_rowgroup_state* __restrict__ p, int lane) {
	return __hip_atomic_load(reinterpret_cast<const uint32_t*>(p) + (lane & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool rowgroup_state_is_whole(uint32_t mine) {
	const uint32_t w0 = __builtin_amdgcn_readlane(mine, 0), w1 = __builtin_amdgcn_readlane(mine, 1), w3 = __builtin_amdgcn_readlane(mine, 3);
	const uint32_t w4 = __builtin_amdgcn_readlane(mine, 4), w5 = __builtin_amdgcn_readlane(mine, 5), w6 = __builtin_amdgcn_readlane(mine, 6), w7 = __builtin_amdgcn_readlane(mine, 7);
	const bool     unpublished = ((w0 & w1) == 0xFFFFFFFFu) | ((w4 & w5) == 0xFFFFFFFFu) | ((w6 & w7) == 0xFFFFFFFFu);
	return ((w3 >> 24) == kStateReady) & !unpublished;
}
__device__ __forceinline__ alpgpu_rowgroup_state rowgroup_state_poll_finish(const alpgpu_rowgroup_state* __restrict__ p, uint32_t mine, int lane, uint32_t spin_limit, bool& ok) {
	const uint32_t* src   = reinterpret_cast<const uint32_t*>(p) + (lane & 7);
	uint32_t        spins = 0;
	ok                    = true;
	while (!rowgroup_state_is_whole(mine)) {
		if (++spins > spin_limit) {
			ok = false;
			break;
		}
		__builtin_amdgcn_s_sleep(32);
		mine = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	return unpack_rowgroup_state(mine);
}
__device__ __forceinline__ alpgpu_rowgroup_state load_rowgroup_state_async(const alpgpu_rowgroup_state* __restrict__ p, int lane, uint32_t spin_limit, bool& ok) {
	return rowgroup_state_poll_finish(p, rowgroup_state_poll_begin(p, lane), lane, spin_limit, ok);
}

// The ALP_RD dictionary of a vector's rowgroup (eight u16 entries = bytes 16..31 of the state) as two words.  The column decode
// kernels read it right after the descriptor, together with the packed words — read inside the decode it was a third dependent
// round trip, behind the barrier, for every ALP_RD vector.
struct RdDict {
	uint64_t lo, hi;
};
__device__ __forceinline__ RdDict load_rd_dict(const alpgpu_rowgroup_state* __restrict__ rgs, uint64_t v, bool is_rd) {
	RdDict dict {0ull, 0ull};
	if (is_rd) { // wave-uniform
		static_assert(offsetof(alpgpu_rowgroup_state, rd_dict) == 16, "dictionary = second half of the state");
		const uint32_t  rg = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v / kRowgroup));
		const uint64_t* dp = reinterpret_cast<const uint64_t*>(rgs + rg) + 2;
		dict.lo            = dp[0];
		dict.hi            = dp[1];
	}
	return dict;
}

// ---- scalar codec arithmetic --------------------------------------------------

// static_cast<int64_t>(double) with x86-64 semantics (cvttsd2si): NaN / |x| >= 2^63 -> INT64_MIN.
// The reference relies on this at include/alp/encoder.hpp:88.
__device__ __forceinline__ int64_t cast64_x86(double x) {
	const bool in_range = (x > -9223372036854775808.0) && (x < 9223372036854775808.0);
	return in_range ? static_cast<int64_t>(x) : INT64_MIN;
}

// include/alp/decoder.hpp:128-131 — wrap-around integer multiply, exact int64->double, one FP multiply
__device__ __forceinline__ double decode_value(int64_t enc, int64_t fact, double frac) {
	const int64_t m = static_cast<int64_t>(static_cast<uint64_t>(enc) * static_cast<uint64_t>(fact));
	return static_cast<double>(m) * frac;
}

// include/alp/encoder.hpp:81-89 with SAFE = false.  Compiled with -ffp-contract=off: two multiplies, one
// add, one subtract, each rounded.
__device__ __forceinline__ int64_t encode_value_unsafe(double v, double exp10, double frac10) {
	double t = v * exp10;
	t        = t * frac10;
	t        = t + kMagic;
	t        = t - kMagic;
	return cast64_x86(t);
}

// include/alp/encoder.hpp:75-89 with SAFE = true
__device__ __forceinline__ int64_t encode_value_safe(double v, double exp10, double frac10) {
	double t = v * exp10;
	t        = t * frac10;
	const bool impossible = !(__builtin_fabs(t) <= 1.7976931348623157e308) /* inf or nan */ || t > kUpperLimit ||
	                        t < kLowerLimit || (t == 0.0 && __builtin_signbit(t));
	if (impossible) { return static_cast<int64_t>(kUpperLimit); }
	t = t + kMagic;
	t = t - kMagic;
	return cast64_x86(t);
}

// min / max of doubles that ignore a quiet NaN operand (IEEE minNum / maxNum, which is what the instructions compute)
__device__ __forceinline__ double fmin_num(double a, double b) {
	double d;
	asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
	return d;
}
__device__ __forceinline__ double fmax_num(double a, double b) {
	double d;
	asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
	return d;
}
// include/alp/encoder.hpp:91-106
__device__ __forceinline__ int count_bits(int64_t mx, int64_t mn) {
	const uint64_t d = static_cast<uint64_t>(mx) - static_cast<uint64_t>(mn);
	return d == 0 ? 0 : 64 - __builtin_clzll(d);
}

// ---- FastLanes u64 unpack of one value pair from a staged copy of the vector's packed words -----------
// words16: the vector's packed stream viewed as 16-byte units (unit index = 8*k + a holds stream word k of
// lane16 columns 2a and 2a+1).  One extra unit row past the end must be readable (content irrelevant).
struct U64Pair {
	uint64_t x, y;
};

// packed words addressed as 16-byte units through a plain pointer (LDS or global)
struct UnitsPtr {
	const ulonglong2* p;
	__device__ __forceinline__ ulonglong2 operator()(int i) const { return p[i]; }
};

template <typename Units>
__device__ __forceinline__ U64Pair unpack_pair_u64(const Units& units, int bw, uint64_t mask, int row, int a) {
	const int p = row * bw;
	const int k = p >> 6;
	const int s = p & 63;
	const ulonglong2 w0 = units(8 * k + a);
	const ulonglong2 w1 = units(8 * k + 8 + a);
	U64Pair          r;
	// (w1 << (64 - s)) without the undefined shift by 64 when s == 0
	r.x = ((w0.x >> s) | ((w1.x << 1) << (63 - s))) & mask;
	r.y = ((w0.y >> s) | ((w1.y << 1) << (63 - s))) & mask;
	return r;
}

// A kernel argument read from the kernarg segment WHERE IT IS USED.  The compiler loads every argument a kernel names at its entry; arguments the
// single-pass encode kernels need only behind their wait for the ordered offset (descriptor / stream pointers, capacities: ten scalar registers)
// then live through the analysis and the pack, which run at the limit of the scalar register file — the allocator parked exactly these in VGPR
// lanes at the entry and fetched them back with v_readlane at the end (vector instructions).  byte_offset: the parameters in order, naturally aligned.
template <class T>
__device__ __forceinline__ T late_kernel_arg(int byte_offset) {
	typedef const __attribute__((address_space(4))) uint8_t* kernarg_t;
	kernarg_t ka = (kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
	return *reinterpret_cast<const __attribute__((address_space(4))) T*>(ka + byte_offset);
}
// k_encode_lean and k_encode_fused_f32 begin with the same seven pointers and two capacities
constexpr int kArgDescs = 16, kArgPacked = 24, kArgExcs = 32, kArgPackedCap = 56, kArgExcCap = 64;

} // namespace alpgpu
