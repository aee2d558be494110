// decode_kernels.hip — fused ALP / ALP_RD column decode for gfx950.
//
// Replaces, per vector (reference file:line relative to /root/reference):
//   ALP    : generated::falp::fallback::scalar::falp   include/alp/falp.hpp:10-26, src/falp.cpp:42440 (:114-121 per value)
//            + alp::decoder<double>::patch_exceptions   include/alp/decoder.hpp:141-149
//   ALP_RD : unffor::unffor (u64 right, u16 left)       src/fastlanes_generated_unffor.cpp:23010, :22846
//            + alp::rd_encoder<double>::decode          include/alp/rd.hpp:152-178
//
// Launch shape (measured: profiles/r01_membw_store_structure.txt, r01_membw2_waves_per_vector.txt,
// r01_decode_ablation.txt): the HBM subsystem of MI355X sustains ~5.2-5.4 TB/s of mixed read+write traffic from a
// persistent grid-stride kernel in which every wavefront streams its own 8 KiB vector (and the arithmetic is NOT
// the limiter there: dropping it changes nothing), but ~6.0-6.3 TB/s when the hardware dispatcher hands out ONE
// SMALL WORKGROUP PER VECTOR in order — the in-flight footprint stays compact and in address order.  So:
// grid = n_vectors, one 256-thread workgroup (4 wavefronts) per 1024-value vector, wavefront q owns steps
// m = 2q, 2q+1 (a contiguous 2 KiB quarter of the output):
//   1. one 32-byte descriptor load (wave-uniform scalar load),
//   2. the vector's packed words (128*bw bytes) go HBM -> LDS once, 16 bytes per thread per load,
//   3. the exception record (if any) becomes a 1024-bit position mask in LDS (+ values staged in LDS); every
//      wavefront keeps the 32-word exclusive prefix of the mask's popcounts in registers,
//   4. each lane unpacks its value pairs, applies base/FACT/FRAC (or the RD dictionary glue), substitutes
//      exceptions by rank, and stores 16 bytes -> every store instruction writes 1 KiB contiguous.
// What bounds it now (profiles/r01_time_waves.txt, r01_time_prefetch.txt, r01_time_v2.txt): a workgroup lives for two
// dependent HBM round trips (descriptor, then packed words) and 8 workgroups fit per CU (32-wave cap), i.e.
// ~2 vectors/us/CU regardless of bit width.  Keeping more vectors in flight (2 waves per vector, or 2 vectors per
// workgroup = ALPGPU_OPT_DECODE_VECTORS_PER_WG 2) lifts narrow widths (bw 4: 4.1 -> 5.2 TB/s, bw 16: 5.6 -> 6.1) but
// the larger in-flight window costs 5-8 % on wide widths and on the mixed-width benchmark column (0.72 vs 0.79 of
// peak); an L2 prefetch of the descriptor 2-16 Ki vectors ahead did not move anything.  Default: 1 vector, 4 waves.
// HBM traffic per vector is the algorithmic minimum: 32 B descriptor + 128*bw B packed + exception record read,
// 8192 B written (rocprofv3 FETCH_SIZE/WRITE_SIZE: profiles/*_pmc.json); no intermediate goes to HBM.
#include "alp_device.hpp"
#include "decode_policy.hpp"
#include "launch.hpp"
#include <cstdlib>

namespace alpgpu {

#ifndef ALPGPU_DEC_WAVES
#define ALPGPU_DEC_WAVES 4
#endif
constexpr int kDecWaves     = ALPGPU_DEC_WAVES; // wavefronts cooperating on one vector
constexpr int kStepsPerWave = 8 / kDecWaves;
#ifndef ALPGPU_DECODE_BATCH
#define ALPGPU_DECODE_BATCH 4 // steps whose words are requested together (decode_vector_quarters)
#endif
constexpr int kStageBytes   = 8704; // >= 63*128 (RD right) + 3*128 (RD left) + 128 pad, and >= 64*128 + 128 (ALP bw 64)
#ifndef ALPGPU_EXC_STAGE
#define ALPGPU_EXC_STAGE 128 // (A/B builds: 256 — round 6, profiles/r06_decode_policy.txt)
#endif
constexpr int kExcStage     = ALPGPU_EXC_STAGE;  // 8-byte exception values staged in LDS per vector (four times as many 2-byte ALP_RD ones); the rest are read from HBM on use
constexpr uint32_t kExcStageBytes = 8u * kExcStage;

template <int STAGE, int EXC = kExcStage>
struct __attribute__((aligned(16))) DecodeLdsT {
	static constexpr bool kPrefixInLds = false; // exception lookup by ds_bpermute (exception_hits)
	static constexpr int  kStage       = STAGE;
	static constexpr uint32_t kExcBytes = 8u * EXC;
	uint8_t  stage[STAGE];
	uint32_t mask[32];
	uint8_t  excv[kExcBytes]; // the head of the exception record as it lies in the stream (its values come first), brought in by LDS-DMA
	uint16_t rdict[8];        // ALP_RD: the rowgroup's dictionary, looked up by ds_read_u16 (round 6; decode_f32_kernels.hip: DecodeLdsF32)
};
using DecodeLds = DecodeLdsT<kStageBytes>;
// Round 6: columns whose vectors carry MORE exceptions than the 128-entry stage holds on average (bench.py's 10 %-exceptions column: 187 per vector) decode every
// value beyond the stage with a load from HBM in the unpack loop (the run-time form of decode_vector_quarters).  A 256-entry stage lifts that column from 0.72 to 0.77
// of the HBM peak — and costs every OTHER column 2-6 % (1 KiB more LDS per vector: city_temperature 0.78 -> 0.74, nyc29 0.81 -> 0.78; call 6), so it is an instance
// of its own, launched for columns whose hints say so (api_decode.hip: decode_variant_for, variant bit 6).
using DecodeLdsManyExc = DecodeLdsT<kStageBytes, 2 * kExcStage>;
// Round 4: FOUR narrow vectors per workgroup.  With two, a column of <= 16-bit vectors sits on a plateau of 0.69-0.73 of the HBM peak whatever its
// width (profiles/r04_decode_floor.txt): what bounds it is the bytes in flight per CU — 16 vectors, each two dependent round trips — and the full
// 8.5 KiB stage per vector is what caps a CU at two vectors x eight workgroups.  A stage of 2.25 KiB (17 bits + the unit row the unpack reads
// past the end) lets a workgroup take four.  A vector that does not fit it (the launch shape follows the column's AVERAGE width) is not staged:
// its lanes read their words straight from HBM through buffer loads, as the one-wavefront sinks do.
constexpr int kNarrowStageBytes = 17 * 128 + 128;
using DecodeLdsNarrow = DecodeLdsT<kNarrowStageBytes>;

// The exception mask (32 words) as seen by one wavefront: lane l < 32 holds word l and the number of exceptions in the
// words before it.  The prefix is a DPP row scan (register-to-register; a __shfl_up chain would be five dependent trips
// through the LDS crossbar on the critical path of a latency-bound workgroup), the per-step lookups below are readlanes.
struct ExcMask {
	uint32_t word; // mask word (lane & 31)
	int      excl; // exceptions in words 0 .. (lane & 31) - 1
};
template <class LDS>
__device__ __forceinline__ ExcMask load_exception_mask(const LDS& L, int lane) {
	ExcMask   m;
	m.word  = L.mask[lane & 31];
	const int c = lane < 32 ? __builtin_popcount(m.word) : 0;
	int       v = c;
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); // row_shr:1
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); // row_shr:2
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); // row_shr:4
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); // row_shr:8   -> inclusive scan inside each 16-lane row
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> row 1 += total of row 0 (lanes 0..31 done)
	m.excl = v - c;
	return m;
}

// the value of the exception of that rank: staged (LDS) or, past the stage, from HBM.  all_staged (wave-uniform: the vector's count fits the
// stage) keeps the common case to the LDS read alone — written as "LDS if rank < kStaged else HBM" the compiler selects between the two
// ADDRESSES and issues one flat load (slower, and it waits for every counter).
template <int VAL_BYTES, class LDS>
__device__ __forceinline__ uint64_t fetch_exception(const LDS& L, const uint8_t* __restrict__ rec, int rank, bool all_staged) {
	constexpr int kStaged = static_cast<int>(LDS::kExcBytes) / VAL_BYTES;
	const int     at      = rank < kStaged ? rank : kStaged - 1;
	uint64_t      v;
	if constexpr (VAL_BYTES == 8) {
		v = reinterpret_cast<const uint64_t*>(L.excv)[at];
		if constexpr (LDS::kPrefixInLds) {
			asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory"); // (k_sink_direct: as in exception_hits_lds) ... and keeps the two loads two loads
		} else {
			asm volatile("" : "+v"(v)); // keeps the two loads two loads
		}
		if (!all_staged) {
			if (rank >= kStaged) { v = reinterpret_cast<const uint64_t*>(rec)[rank]; }
		}
	} else {
		v = reinterpret_cast<const uint16_t*>(L.excv)[at];
		if constexpr (LDS::kPrefixInLds) {
			asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory");
		} else {
			asm volatile("" : "+v"(v)); // keeps the two loads two loads
		}
		if (!all_staged) {
			if (rank >= kStaged) { v = reinterpret_cast<const uint16_t*>(rec)[rank]; }
		}
	}
	return v;
}

// exception lookup for the pair (128*m + 2*lane, +1): 2-bit hit mask and the rank of the first hit.  m is wave-uniform.  The pair's mask
// word is 4m + (lane >> 4): word and prefix come from the lane that holds them by ds_bpermute (the LDS crossbar, no memory) — as eight
// readlanes and six selects per step this lookup was a quarter of the vector instructions of a vector with exceptions, and the consumers
// (SUM / COUNT sinks) are bound by exactly those (profiles/r03_consumers.txt).
// The same lookup out of the wavefront's LDS (mask word and its prefix as two ds_read_b32 of one address per 16 lanes) — what k_sink_direct
// uses.  SOME BUILDS of that kernel (another register budget, another batch size, two compiler pins ...) come back with a wrong extracted
// field in the step FOLLOWING an exception lookup — right words in the registers, wrong value — for ~5 % of the exception-carrying
// vectors of a long column, never the same ones; a build either does it in every run or never.  Three such builds became clean with this
// form instead of ds_bpermute; later one with this form was wrong again, clean with full LDS waits, and another wrong WITH them.  None of
// that was the cause: the wrong builds had a 64-bit shift's amount in the LAST register of their allocation, which gfx950 reads as VGPR0
// there (see k_sink_direct below; profiles/r03_consumers.txt has the way to it).  The LDS form stayed (same speed).  A build of this file
// ships only if tools/check_top_vgpr.py is clean and tests/test_decode_sum_gpu.py::test_exception_records_that_change_nothing and
// ...across_a_full_chip pass.  k_decode_column uses 44 of 48 allocated registers: the pattern cannot arise there.
template <class LDS>
__device__ __forceinline__ uint32_t exception_hits_lds(const LDS& L, int m, int lane, int& rank) {
	const int      q    = 4 * m + (lane >> 4);
	uint32_t       word = L.mask[q];
	int            pref = static_cast<int>(L.pref[q]);
	asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(word), "+v"(pref)::"memory"); // both words here before anything is computed from them (see above: not a cure by itself)
	const int      b0   = (2 * lane) & 31;
	const uint32_t hits = (word >> b0) & 3u;
	rank                = pref + __builtin_popcount(word & ((1u << b0) - 1u));
	return hits;
}
__device__ __forceinline__ uint32_t exception_hits(const ExcMask& em, int m, int lane, int& rank) {
	const int      src  = (4 * m + (lane >> 4)) << 2; // byte address of the source lane
	const uint32_t word = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src, static_cast<int>(em.word)));
	const int      pref = __builtin_amdgcn_ds_bpermute(src, em.excl);
	const int      b0   = (2 * lane) & 31;
	const uint32_t hits = (word >> b0) & 3u;
	rank                = pref + __builtin_popcount(word & ((1u << b0) - 1u));
	return hits;
}

template <bool NT_STORE>
__device__ __forceinline__ void store_pair(double2* __restrict__ p, double x, double y) {
	typedef double d2v __attribute__((ext_vector_type(2)));
	d2v o;
	o.x = x;
	o.y = y;
	if constexpr (NT_STORE) {
		__builtin_nontemporal_store(o, reinterpret_cast<d2v*>(p));
	} else {
		*reinterpret_cast<d2v*>(p) = o;
	}
}

// ---- one vector, after its packed words / exception mask are visible in L --------------------------------------
// SINK = kSinkStore: the decoded pair is stored (dst).  kSinkSum: it is added to `acc` instead, x then y, in step order —
// the fixed order tests/test_decode_sum_gpu.py reproduces on the host (decode fused into its
// consumer, the shape of publication/source_code/bench_end_to_end/src/benchmarks/alp/queries/q1.cpp:63-104, without the
// 8 KiB per vector of decoded doubles ever reaching HBM).
// kSinkCount: `acc` counts the values v with lo <= v <= hi (a predicate pushed into the decode; NaN never qualifies).
constexpr int kSinkStore = 0, kSinkSum = 1, kSinkCount = 2;
// kSinkProbe (alpgpu_debug_decode_probe): the consumers' memory traffic and latency chain without the unpack — descriptors, packed
// words into LDS, exceptions, the barrier — every thread then just adds up the staged 16-byte units it would have unpacked.
constexpr int kSinkProbe = 3;
template <int SINK>
__device__ __forceinline__ void consume_pair(double ox, double oy, double* acc, double lo, double hi) {
	if constexpr (SINK == kSinkSum) {
		*acc += ox;
		*acc += oy;
	} else {
		*acc += (ox >= lo && ox <= hi) ? 1.0 : 0.0; // small integers: exact in double
		*acc += (oy >= lo && oy <= hi) ? 1.0 : 0.0;
	}
}

// One table for everything the ALP decode reads by descriptor: entry i = {FACT_ARR[i], 10^i as a double, the conversion shortcut's bound for
// f = i, FRAC_ARR[i]} — one address computation on the scalar unit instead of four (the decode kernels run eight wavefronts per SIMD, each
// repeating the same wave-uniform prologue, and the scalar unit is what saturates first: profiles/r03_consumers.txt).  The bound is
// min(2^51 - 1, floor((2^63 - 1) / 10^i)): the largest |integer| that is exact as a double in the shortcut's range AND whose product with
// 10^i stays inside int64.  Literals as in alp_device.hpp (constants.hpp:66-154).
struct DecodeTabEntry {
	int64_t  fact;
	double   fact_d;
	uint64_t shortcut_bound;
	double   frac;
};
__device__ __constant__ const DecodeTabEntry kDecodeTab[21] = {
    {1ll, 1.0, 2251799813685247ull, 1.0},
    {10ll, 10.0, 2251799813685247ull, 0.1},
    {100ll, 100.0, 2251799813685247ull, 0.01},
    {1000ll, 1000.0, 2251799813685247ull, 0.001},
    {10000ll, 10000.0, 922337203685477ull, 0.0001},
    {100000ll, 100000.0, 92233720368547ull, 0.00001},
    {1000000ll, 1000000.0, 9223372036854ull, 0.000001},
    {10000000ll, 10000000.0, 922337203685ull, 0.0000001},
    {100000000ll, 100000000.0, 92233720368ull, 0.00000001},
    {1000000000ll, 1000000000.0, 9223372036ull, 0.000000001},
    {10000000000ll, 10000000000.0, 922337203ull, 0.0000000001},
    {100000000000ll, 100000000000.0, 92233720ull, 0.00000000001},
    {1000000000000ll, 1000000000000.0, 9223372ull, 0.000000000001},
    {10000000000000ll, 10000000000000.0, 922337ull, 0.0000000000001},
    {100000000000000ll, 100000000000000.0, 92233ull, 0.00000000000001},
    {1000000000000000ll, 1000000000000000.0, 9223ull, 0.000000000000001},
    {10000000000000000ll, 10000000000000000.0, 922ull, 0.0000000000000001},
    {100000000000000000ll, 100000000000000000.0, 92ull, 0.00000000000000001},
    {1000000000000000000ll, 1000000000000000000.0, 9ull, 0.000000000000000001},
    {0ll, 0.0, 0ull, 0.0000000000000000001},
    {0ll, 0.0, 0ull, 0.00000000000000000001},
};

// The per-vector constants, read in front of the barrier: ALP_RD = the rowgroup's dictionary (RdDict, alp_device.hpp); ALP vectors use
// the same two words for lo = FACT_ARR[f], hi = bits of FRAC_ARR[e] — table reads that depend only on the descriptor.
// (plus, for ALP, 10^f as a double and the no-wrap limit of the conversion shortcut: every table read that depends on the descriptor is in
// flight with the packed words instead of behind the barrier)
struct VectorConsts {
	uint64_t lo, hi;
	uint64_t fact_d_bits, shortcut_bound;
};
__device__ __forceinline__ VectorConsts load_vector_consts(const alpgpu_rowgroup_state* __restrict__ rgs, uint64_t v, const alpgpu_vector_desc& d) {
	if (d.scheme != ALPGPU_SCHEME_ALP) { // wave-uniform
		const RdDict r = load_rd_dict(rgs, v, true);
		return VectorConsts {r.lo, r.hi, 0ull, 0ull};
	}
	const DecodeTabEntry& by_f = kDecodeTab[d.f];
	return VectorConsts {static_cast<uint64_t>(by_f.fact), static_cast<uint64_t>(__double_as_longlong(kDecodeTab[d.e].frac)),
	                     static_cast<uint64_t>(__double_as_longlong(by_f.fact_d)), by_f.shortcut_bound};
}

// Where the packed words of a vector are read from: the workgroup's LDS stage (the column kernels), or HBM directly (k_sink_direct).
struct WordPair {
	ulonglong2 w0, w1;
};
struct StagedWords {
	const uint8_t* stage;
	__device__ __forceinline__ WordPair pair(int i) const { // units i and i + 8: stream words k and k + 1 of a column pair
#ifdef ALPGPU_STAGED_WORDS_FULL_WAIT
		WordPair w {reinterpret_cast<const ulonglong2*>(stage)[i], reinterpret_cast<const ulonglong2*>(stage)[i + 8]};
		asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w.w0.x), "+v"(w.w0.y), "+v"(w.w1.x), "+v"(w.w1.y)::"memory");
		return w;
#else
		return WordPair {reinterpret_cast<const ulonglong2*>(stage)[i], reinterpret_cast<const ulonglong2*>(stage)[i + 8]};
#endif
	}
	__device__ __forceinline__ uint2 left_pair(int rbw, int i) const { // left words i and i + 32
		return make_uint2(reinterpret_cast<const uint32_t*>(stage + 128 * rbw)[i], reinterpret_cast<const uint32_t*>(stage + 128 * rbw)[i + 32]);
	}
};
struct BufferWords { // HBM through buffer loads: the resources are sized to the vector's packed words, so reads past its last unit / last left
	// word return 0 (what the unpack reads there is masked off anyway) without a clamp per load, and the second unit of a pair is the
	// first one's address with an immediate offset
	__amdgpu_buffer_rsrc_t units, lefts;
	__device__ __forceinline__ WordPair pair(int i) const {
		typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
		const uint32_t at = static_cast<uint32_t>(i) * 16u;
		const u32x4    a = __builtin_amdgcn_raw_buffer_load_b128(units, at, 0, 0);
		const u32x4    b = __builtin_amdgcn_raw_buffer_load_b128(units, at, 128, 0); // unit i + 8: the same address register, a scalar offset
		return WordPair {make_ulonglong2((static_cast<uint64_t>(a[1]) << 32) | a[0], (static_cast<uint64_t>(a[3]) << 32) | a[2]),
		                 make_ulonglong2((static_cast<uint64_t>(b[1]) << 32) | b[0], (static_cast<uint64_t>(b[3]) << 32) | b[2])};
	}
	__device__ __forceinline__ uint2 left_pair(int, int i) const {
		const uint32_t at = static_cast<uint32_t>(i) * 4u;
		return make_uint2(__builtin_amdgcn_raw_buffer_load_b32(lefts, at, 0, 0), __builtin_amdgcn_raw_buffer_load_b32(lefts, at, 128, 0));
	}
};

// ---- exceptions patched AFTER the stores (round 5; the reference's own order: decode everything, then scatter, decoder.hpp:141-149) ---------------
// An ALP vector with a few exceptions is decoded as if it had none — no mask, no prefix, no per-value lookup (the +60 % vector instructions of
// profiles/r01_pmc_sq_decode_exceptions.txt) — and its exceptions are then written over the freshly stored values: exception j (positions ascend, so
// j is also its rank) by lane j of the wavefront that OWNS the quarter the position lies in, i.e. the same wavefront that stored that quarter a few
// instructions earlier.  A wavefront's vector stores reach memory in issue order (both go down the same L1 queue to the same L2 channel, same cache
// policy), so the 8-byte patch lands on the 16-byte store it overlaps; no barrier, no fence, no LDS.  Position and value are loaded (2 + 8 bytes
// per lane, straight from the record) together with the packed words and wait in two / three registers.  Vectors with more exceptions than
// `patch_max` (a kernel argument, <= 64: one lane each; 0 switches the arm off) and ALP_RD vectors (their exceptions replace the LEFT part only:
// the patched double needs the right part the lane no longer has) take the mask route below.  -DALPGPU_DECODE_PATCH_FENCE: s_waitcnt vmcnt(0)
// between a wavefront's stores and its patches (the conservative form, for A/B runs).
struct PatchRegs {
	uint32_t pos;
	uint64_t val;
};
__device__ __forceinline__ bool vector_patches_after(const alpgpu_vector_desc& d, uint32_t patch_max) { // wave-uniform
	return d.scheme == ALPGPU_SCHEME_ALP && d.exc_cnt != 0 && static_cast<uint32_t>(d.exc_cnt) <= patch_max;
}
__device__ __forceinline__ PatchRegs issue_patch_loads(const alpgpu_vector_desc& d, const uint8_t* __restrict__ rec, int lane) {
	PatchRegs   r {0u, 0ull};
	const int   cnt = d.exc_cnt;
	if (lane < cnt) {
		r.val = reinterpret_cast<const uint64_t*>(rec)[lane];
		r.pos = reinterpret_cast<const uint16_t*>(rec + 8u * static_cast<uint32_t>(cnt))[lane];
	}
	return r;
}
template <bool NT_STORE>
__device__ __forceinline__ void apply_patches(const PatchRegs& r, int cnt, double* __restrict__ out_vec, int wave, int lane) {
#ifdef ALPGPU_DECODE_PATCH_FENCE
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
	if (lane < cnt && static_cast<int>(r.pos >> 8) == wave) { // this wavefront stored values 256 wave .. 256 wave + 255 (steps 2 wave, 2 wave + 1)
		const double x = __longlong_as_double(static_cast<long long>(r.val));
#if defined(ALPGPU_DECODE_PATCH_NO_STORE) // timing experiment (wrong output): what the patch stores themselves cost
		asm volatile("" ::"v"(x), "v"(out_vec + r.pos));
#elif defined(ALPGPU_DECODE_PATCH_PLAIN) // A/B: the patch as an ordinary store whatever the vector's own stores are
		out_vec[r.pos] = x;
#else
		if constexpr (NT_STORE) {
			__builtin_nontemporal_store(x, out_vec + r.pos);
		} else {
			out_vec[r.pos] = x;
		}
#endif
	}
}
__device__ __forceinline__ alpgpu_vector_desc without_exceptions(alpgpu_vector_desc d) {
	d.exc_cnt = 0;
	return d;
}
// ALPGPU_DECODE_PATCH_MODE 1: as described above (a second, 8-byte store per exception).  2: the exceptions are put in their places IN REGISTERS, in
// front of the one store: each wavefront keeps a 256-byte slot table in LDS (its quarter of the vector's exception stage: byte i = 1 + the index of
// the exception at position 256 wave + i, 0 = none; zeroed behind the issue of the loads, filled once position and value have arrived), a lane reads
// the two bytes of its pair per step, and a hit fetches the value from the lane that holds it (exception j sits in lane j: ds_bpermute).  No
// partial-line write reaches HBM; no rank, no prefix, no mask.
// 0 (the default build since the measurement below): no patch arm is compiled in and `patch_max` is ignored.  Round 5's result (profiles/
// r05_decode_exceptions.txt): once the stores of a wavefront no longer wait for one another (ExcMode below), the mask route itself is 5-10 % FASTER
// than it was, mode 1 is slower than it at every width (its 8-byte stores reach HBM as read-modify-writes of lines the L2 has already let go:
// +20-30 % time on narrow vectors), mode 2 equals it on wide vectors and loses 5-10 % on narrow ones — and the mere presence of either arm's
// code and registers costs the two-vectors-per-workgroup kernel 8 % on exception-free columns of 3-6 bits.
#ifndef ALPGPU_DECODE_PATCH_MODE
#define ALPGPU_DECODE_PATCH_MODE 0
#endif
template <class LDS>
__device__ __forceinline__ void patch_table_zero(LDS& L, int wave, int lane) {
	reinterpret_cast<uint32_t*>(L.excv)[64 * wave + lane] = 0u;
}
template <class LDS>
__device__ __forceinline__ void patch_table_fill(LDS& L, const PatchRegs& r, int cnt, int wave, int lane) { // (behind the barrier that follows the loads)
	if (lane < cnt && static_cast<int>(r.pos >> 8) == wave) { L.excv[256 * wave + (r.pos & 255u)] = static_cast<uint8_t>(lane + 1); }
	wave_lds_sync();
}
// the pair of step m of this wavefront's quarter: exceptions (if any) over the decoded values
template <class LDS>
__device__ __forceinline__ void patch_pair_from_table(const LDS& L, const PatchRegs& r, int m, int wave, int lane, double& ox, double& oy) {
	const uint32_t t = reinterpret_cast<const uint16_t*>(L.excv)[128 * wave + 64 * (m & (kStepsPerWave - 1)) + lane];
	if (ballot64(t != 0u) != 0) { // wave-uniform: some lane's pair of this step holds an exception
		const int i0 = static_cast<int>(t & 255u) - 1, i1 = static_cast<int>(t >> 8) - 1;
		const int lo = static_cast<int>(static_cast<uint32_t>(r.val)), hi = static_cast<int>(static_cast<uint32_t>(r.val >> 32));
		const uint64_t v0 = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(4 * (i0 < 0 ? 0 : i0), hi))) << 32) |
		                    static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(4 * (i0 < 0 ? 0 : i0), lo));
		const uint64_t v1 = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(4 * (i1 < 0 ? 0 : i1), hi))) << 32) |
		                    static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(4 * (i1 < 0 ? 0 : i1), lo));
		ox = i0 >= 0 ? __longlong_as_double(static_cast<long long>(v0)) : ox;
		oy = i1 >= 0 ? __longlong_as_double(static_cast<long long>(v1)) : oy;
	}
}

template <int V>
struct ExcMode {
	static constexpr int value = V;
};
template <int A> // 0: the literal arithmetic, 1: the IEEE-product shortcut on 64-bit fields, 2: the same on fields of <= 32 bits (32-bit unpack)
struct ArithShortcut {
	static constexpr int value = A;
};
// N_Q consecutive quarters of one vector, from quarter q0 on (a quarter = steps kStepsPerWave q .. + kStepsPerWave - 1 = what one wavefront of a
// four-wavefront workgroup does; the sinks accumulate quarter q0 + i into acc[i]).  `em` = the vector's exception mask as this wavefront sees
// it (ignored when the vector has none), L = where staged exception values live.
// ONLY: 0 = either scheme (decided from the descriptor), 1 = the caller knows the vector is ALP, 2 = ALP_RD (only that arm is compiled in).
template <bool NT_STORE, int SINK, int N_Q, int ONLY = 0, class LDS, class WORDS>
__device__ __forceinline__ void decode_vector_quarters(const LDS& L, const WORDS& units, const alpgpu_vector_desc& d, const VectorConsts& dict, const ExcMask& em,
                                                      const uint8_t* __restrict__ rec, double2* __restrict__ dst, int q0, int lane, double* acc,
                                                       double range_lo, double range_hi, bool use_patch = false, PatchRegs patch = PatchRegs {0u, 0ull}) {
	// use_patch (ALPGPU_DECODE_PATCH_MODE 2; ALP store decode, one quarter per wavefront: q0 = the wavefront): d is the vector without its
	// exceptions, which are looked up in the wavefront's slot table instead (patch_pair_from_table)
	const int      bw       = d.bw;
	const int      cnt      = d.exc_cnt;
	const bool     all_staged = cnt <= static_cast<int>(LDS::kExcBytes) / (d.scheme == ALPGPU_SCHEME_ALP ? 8 : 2); // wave-uniform
	const int      a        = lane & 7;
	const int      r0       = lane >> 3;
	// The pair (row, columns 2a, 2a + 1) out of its two stream words (alp_device.hpp: unpack_pair_u64), in two halves: the reads of a whole
	// batch of steps are requested before the first of them is used — with the words in HBM (k_sink_direct) every step is otherwise its own
	// dependent round trip.
	constexpr int kSteps = N_Q * kStepsPerWave;
	constexpr int kBatch = kSteps < ALPGPU_DECODE_BATCH ? kSteps : ALPGPU_DECODE_BATCH;
	constexpr int kBatchRd = ONLY == 2 ? 1 : (kSteps < 2 ? kSteps : 2); // ALP_RD keeps the left words and the dictionary selects in registers as well
	auto request = [&](int width, int row) { return units.pair(8 * ((row * width) >> 6) + a); };
	auto extract = [&](int width, uint64_t wmask, int row, const WordPair w) {
		const int s = (row * width) & 63;
		U64Pair   r;
		r.x = ((w.w0.x >> s) | ((w.w1.x << 1) << (63 - s))) & wmask; // (w1 << (64 - s)) without the undefined shift by 64 when s == 0
		r.y = ((w.w0.y >> s) | ((w.w1.y << 1) << (63 - s))) & wmask;
		return r;
	};
	if (ONLY == 1 || (ONLY == 0 && d.scheme == ALPGPU_SCHEME_ALP)) {
		const uint64_t base = static_cast<uint64_t>(d.base);
		const int64_t  fact = static_cast<int64_t>(dict.lo);
		const double   frac = __longlong_as_double(static_cast<long long>(dict.hi));
		const uint64_t mask = bw_mask(bw);
		// Conversion shortcut, decided once per vector from its descriptor.  If every value base + digit of this vector
		// lies in (-2^51, 2^51) and times 10^f stays inside int64, then
		//   * (double)value is exact and equals  bitcast(bits(2^52+2^51) + value) - (2^52+2^51)   (one 64-bit add, one f64 add),
		//   * (double)(int64)(value * 10^f) is the correctly rounded product of two exactly representable doubles,
		//     i.e. the IEEE product  (double)value * 10^f,
		// which replaces the 64-bit integer multiply and the software int64->double conversion (src/falp.cpp:114-121 does
		// them per value) and yields the same bits.  Anything else takes the literal path.
		// (decided in integers on wave-uniform values against ONE table entry, kDecodeTab: as doubles — two software int64 -> double
		//  conversions, |.|, max, a product — the decision alone was ~35 vector instructions per wavefront and vector)
		const int64_t  lo       = d.base;
		const int64_t  bound    = static_cast<int64_t>(dict.shortcut_bound); // <= 2^51 - 1
		const double   fact_d   = __longlong_as_double(static_cast<long long>(dict.fact_d_bits));
		// |lo| <= bound first: lo + mask cannot overflow then (mask < 2^50); lo <= lo + mask, so the two ends bound everything between
		const bool     shortcut = bw <= 50 && lo >= -bound && lo <= bound && static_cast<int64_t>(static_cast<uint64_t>(lo) + mask) <= bound;
		const uint64_t kbits   = 0x4338000000000000ull + base;
		// what follows the conversion of a pair: exceptions patched in, then the sink.
		// EXC (compile time): 0 = the vector has no exceptions: nothing is looked up; 1 = it has some and all of their values are staged in LDS: no
		// memory load anywhere in the loop; 2 = decided per step at run time (cnt > 0, staged or not) — the form every caller had until round 5.
		// Why the first two exist (round 5): with the rare "value beyond the stage -> load it from HBM" arm inside the loop the compiler places
		// s_waitcnt vmcnt(0) at the join in front of EVERY step's store — on gfx9 stores count in vmcnt, so each store of a wavefront waited for
		// the acknowledgement of the one before it, exceptions or not — and every step was a basic block of its own.  The store decode now picks
		// the loop per vector (wave-uniform); the sinks (register budgets tuned around the old form) stay on 2.
		auto finish_pair = [&](auto exc_mode, int m, double ox, double oy, double* acc_q) {
			constexpr int EXC = decltype(exc_mode)::value;
			if constexpr (SINK == kSinkStore && N_Q == 1 && ALPGPU_DECODE_PATCH_MODE == 2) {
				if (use_patch) { patch_pair_from_table(L, patch, m, q0, lane, ox, oy); } // (wave-uniform)
			}
			if (EXC == 1 || (EXC == 2 && cnt > 0)) {
				const bool     staged = EXC == 1 ? true : all_staged;
				int            rank;
				uint32_t       hits;
				if constexpr (LDS::kPrefixInLds) { hits = exception_hits_lds(L, m, lane, rank); } else { hits = exception_hits(em, m, lane, rank); }
				if (hits & 1u) {
					ox = __longlong_as_double(static_cast<long long>(fetch_exception<8>(L, rec, rank, staged)));
					++rank;
				}
				if (hits & 2u) { oy = __longlong_as_double(static_cast<long long>(fetch_exception<8>(L, rec, rank, staged))); }
			}
			if constexpr (SINK != kSinkStore) {
				consume_pair<SINK>(ox, oy, acc_q, range_lo, range_hi);
			} else {
#ifdef ALPGPU_DECODE_STORE_WAIT // A/B: every store waits for the one before it, as every build did until round 5 (see above)
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
				store_pair<NT_STORE>(dst + 64 * m + lane, ox, oy);
			}
		};
		// ArithShortcut<2> (round 5; the store decode's vectors of <= 32 bits on the shortcut route — narrow vectors are bound by their vector
		// instructions, not by HBM: profiles/r05_decode_exceptions.txt): the field lies in three of the four dwords of its two stream words and
		// comes out with ONE 32-bit funnel shift (v_alignbit_b32; the amount is taken modulo 32) where the 64-bit form takes three 64-bit shifts;
		// and (double)(base + digit) is the double whose bits are {0x43380000, digit} = M + digit (M = 2^52 + 2^51) minus the wave-uniform double
		// M - base, whose bits are bits(M) - base (scalar integer arithmetic; |base| < 2^51 is the shortcut's condition) — every quantity an
		// integer below 2^53, the difference exact — one v_add_f64 where the 64-bit form adds the base in integers (two instructions) first.
		// The same value, hence the same bits, as ArithShortcut<1>: 7 vector instructions per value instead of 13.
		const uint32_t mask32   = static_cast<uint32_t>(mask);
		const double   magic_mb = __longlong_as_double(static_cast<long long>(0x4338000000000000ull - static_cast<uint64_t>(lo))); // M - base
		auto alp_steps = [&](auto exc_mode, auto shortcut_arith) {
			constexpr int SHORT = static_cast<int>(decltype(shortcut_arith)::value);
#pragma unroll
			for (int b = 0; b < kSteps; b += kBatch) {
				WordPair w[kBatch];
#pragma unroll
				for (int i = 0; i < kBatch; ++i) { w[i] = request(bw, 8 * (kStepsPerWave * q0 + b + i) + r0); }
#pragma unroll
				for (int i = 0; i < kBatch; ++i) {
					const int     m = kStepsPerWave * q0 + b + i;
					if constexpr (SHORT == 2) {
						const uint32_t sft = static_cast<uint32_t>((8 * m + r0) * bw) & 63u;
						const bool     low = sft < 32u; // the field begins in the first dword of word k (else in the second; it ends at most one dword later)
						const uint32_t x0 = static_cast<uint32_t>(w[i].w0.x), x1 = static_cast<uint32_t>(w[i].w0.x >> 32), x2 = static_cast<uint32_t>(w[i].w1.x);
						const uint32_t y0 = static_cast<uint32_t>(w[i].w0.y), y1 = static_cast<uint32_t>(w[i].w0.y >> 32), y2 = static_cast<uint32_t>(w[i].w1.y);
						const uint32_t ux = __builtin_amdgcn_alignbit(low ? x1 : x2, low ? x0 : x1, sft) & mask32;
						const uint32_t uy = __builtin_amdgcn_alignbit(low ? y1 : y2, low ? y0 : y1, sft) & mask32;
						const double   dx = __longlong_as_double(static_cast<long long>(0x4338000000000000ull | ux)) - magic_mb;
						const double   dy = __longlong_as_double(static_cast<long long>(0x4338000000000000ull | uy)) - magic_mb;
						finish_pair(exc_mode, m, (dx * fact_d) * frac, (dy * fact_d) * frac, acc + (b + i) / kStepsPerWave);
						continue;
					}
					const U64Pair u = extract(bw, mask, 8 * m + r0, w[i]);
					if constexpr (SHORT == 1) {
						finish_pair(exc_mode, m, ((__longlong_as_double(static_cast<long long>(u.x + kbits)) - kMagic) * fact_d) * frac,
						            ((__longlong_as_double(static_cast<long long>(u.y + kbits)) - kMagic) * fact_d) * frac, acc + (b + i) / kStepsPerWave);
					} else {
						finish_pair(exc_mode, m, decode_value(static_cast<int64_t>(u.x + base), fact, frac), decode_value(static_cast<int64_t>(u.y + base), fact, frac),
						            acc + (b + i) / kStepsPerWave);
					}
				}
			}
		};
		constexpr bool kPerVectorLoops = SINK == kSinkStore && N_Q == 1;
#ifdef ALPGPU_DECODE_NO_NARROW_ARITH // A/B: without the 32-bit form
		const bool narrow32 = false;
#else
		const bool narrow32 = kPerVectorLoops && bw <= 32;
#endif
		if (shortcut) { // wave-uniform; ONE branch per vector, not one per step (scalar instructions are the scarce ones here)
			if (kPerVectorLoops && cnt == 0) {
				if (narrow32) {
					alp_steps(ExcMode<0> {}, ArithShortcut<2> {});
				} else {
					alp_steps(ExcMode<0> {}, ArithShortcut<1> {});
				}
			} else if (kPerVectorLoops && all_staged) {
				if (narrow32) {
					alp_steps(ExcMode<1> {}, ArithShortcut<2> {});
				} else {
					alp_steps(ExcMode<1> {}, ArithShortcut<1> {});
				}
			} else {
#ifndef ALPGPU_SINK_NO_NARROW_ARITH // round 6, call 36: the SINKS' vectors of <= 32 bits through the 32-bit unpack too (7 vector instructions per value instead of 13; until
				// then the store decode's only): SUM of 2-24-bit columns 0.54-0.76 -> 0.45-0.69 ms per 1 Mi vectors (-9 to -15 %), with 20 exceptions per vector -8 to -13 %,
				// the benchmark column -3 %; same bits.  -DALPGPU_SINK_NO_NARROW_ARITH: the A/B.
				if (!kPerVectorLoops && bw <= 32) {
					alp_steps(ExcMode<2> {}, ArithShortcut<2> {});
				} else
#endif
				alp_steps(ExcMode<2> {}, ArithShortcut<1> {});
			}
		} else {
			alp_steps(ExcMode<2> {}, ArithShortcut<0> {});
		}
	} else {
		// ALP_RD: right parts = u64 lanes (bw = rbw, base 0); left parts = u16 lanes, 64 streams x 16 rows (value i ->
		// lane64 = i & 63, row = i >> 6, word k at left[64*k + lane64]); a lane's pair shares the row 2m + (lane >> 5)
		// and is one aligned u32 of the left stream.
		const int      rbw  = bw;
		const int      lbw  = d.lbw;
		const uint64_t mask = bw_mask(rbw);
		const uint32_t lmsk = (1u << lbw) - 1u;
		const uint64_t dlo = dict.lo, dhi = dict.hi;
		(void)dlo, (void)dhi;
		auto rd_steps = [&](auto exc_mode) {
			constexpr int EXC = decltype(exc_mode)::value; // as in the ALP arm
#pragma unroll
			for (int b = 0; b < kSteps; b += kBatchRd) {
				WordPair rw[kBatchRd];
				uint2    lw[kBatchRd];
#pragma unroll
				for (int i = 0; i < kBatchRd; ++i) {
					const int m = kStepsPerWave * q0 + b + i;
					rw[i]       = request(rbw, 8 * m + r0);
					lw[i]       = units.left_pair(rbw, 32 * (((2 * m + (lane >> 5)) * lbw) >> 4) + (lane & 31));
				}
#pragma unroll
				for (int i = 0; i < kBatchRd; ++i) {
					const int      m   = kStepsPerWave * q0 + b + i;
					double*        acc_q = acc + (b + i) / kStepsPerWave;
					U64Pair        u   = extract(rbw, mask, 8 * m + r0, rw[i]);
					if constexpr (ONLY == 2) { asm volatile("" : "+v"(u.x), "+v"(u.y)); } // (k_sink_direct) the right parts extracted HERE: their four words die before the left parts' work begins
					const int      s   = ((2 * m + (lane >> 5)) * lbw) & 15;
					const uint32_t w0 = lw[i].x, w1 = lw[i].y;
					const uint32_t i0  = (((w0 & 0xFFFFu) >> s) | ((w1 & 0xFFFFu) << (16 - s))) & lmsk;
					const uint32_t i1  = (((w0 >> 16) >> s) | ((w1 >> 16) << (16 - s))) & lmsk;
#ifdef ALPGPU_RD_DICT_IN_REGISTERS // A/B (round 6): the lookup as it was until round 5 — a 3-way select through a 64-bit shift per value
					uint64_t       l0  = ((i0 < 4 ? dlo >> (16 * i0) : dhi >> (16 * (i0 & 3))) & 0xFFFFull);
					uint64_t       l1  = ((i1 < 4 ? dlo >> (16 * i1) : dhi >> (16 * (i1 & 3))) & 0xFFFFull);
#else
					uint64_t       l0  = L.rdict[i0 & 7u];
					uint64_t       l1  = L.rdict[i1 & 7u];
#endif
					if (EXC == 1 || (EXC == 2 && cnt > 0)) {
						const bool     staged = EXC == 1 ? true : all_staged;
						int            rank;
						uint32_t       hits;
						if constexpr (LDS::kPrefixInLds) { hits = exception_hits_lds(L, m, lane, rank); } else { hits = exception_hits(em, m, lane, rank); }
						if (hits & 1u) {
							l0 = fetch_exception<2>(L, rec, rank, staged);
							++rank;
						}
						if (hits & 2u) { l1 = fetch_exception<2>(L, rec, rank, staged); }
					}
					const double ox = __longlong_as_double(static_cast<long long>((l0 << rbw) | u.x));
					const double oy = __longlong_as_double(static_cast<long long>((l1 << rbw) | u.y));
					if constexpr (SINK != kSinkStore) {
						consume_pair<SINK>(ox, oy, acc_q, range_lo, range_hi);
						if constexpr (ONLY == 2) { asm volatile("" : "+v"(*acc_q)); } // ... and the pair added HERE: left to itself the compiler keeps all sixteen results for the end (208 bytes of scratch, 3.7 x the time)
					} else {
						store_pair<NT_STORE>(dst + 64 * m + lane, ox, oy);
					}
				}
				if constexpr (ONLY == 2) { asm volatile("" ::: "memory"); } // (k_sink_direct) the next step's requests stay behind this one's use
			} // batch
		};
		constexpr bool kPerVectorLoops = SINK == kSinkStore && N_Q == 1;
		if (kPerVectorLoops && cnt == 0) {
			rd_steps(ExcMode<0> {});
		} else if (kPerVectorLoops && all_staged) {
			rd_steps(ExcMode<1> {});
		} else {
			rd_steps(ExcMode<2> {});
		}
	}
}

// one wavefront's share of a staged vector (the column kernels: four wavefronts per vector)
template <bool NT_STORE, int SINK = kSinkStore, class LDS = DecodeLds>
__device__ __forceinline__ void decode_staged_vector(const LDS& L, const alpgpu_vector_desc& d, const VectorConsts& dict,
                                                     const uint8_t* __restrict__ rec, double2* __restrict__ dst, int wave, int lane, double* acc = nullptr,
                                                     double range_lo = 0.0, double range_hi = 0.0, bool use_patch = false, PatchRegs patch = PatchRegs {0u, 0ull}) {
	ExcMask em {0u, 0};
	if (d.exc_cnt > 0) { em = load_exception_mask(L, lane); }
	decode_vector_quarters<NT_STORE, SINK, 1>(L, StagedWords {L.stage}, d, dict, em, rec, dst, wave, lane, acc, range_lo, range_hi, use_patch, patch);
}

// Issues every load of one vector: packed words and the values of its exceptions straight into LDS (global_load_lds, 16 resp. 4 bytes per
// lane, no VGPR round trip), exception positions into registers.  (Until round 3 the values went through registers too, behind a branch on the
// scheme for their width — and the compiler's wait-count pass put an s_waitcnt vmcnt(0) between the two arms, i.e. a whole HBM round trip in
// front of the packed loads of every vector with exceptions of one of the two schemes.)
// (a vector whose words do not fit the workgroup's stage — narrow stage only — is not staged: see vector_fits_stage)
template <class LDS>
__device__ __forceinline__ bool vector_fits_stage(const alpgpu_vector_desc& d) {
	return 128 * (static_cast<int>(d.bw) + (d.scheme == ALPGPU_SCHEME_ALP ? 0 : static_cast<int>(d.lbw))) + 128 <= LDS::kStage;
}
template <class LDS>
__device__ __forceinline__ uint32_t issue_vector_loads(LDS& L, const alpgpu_vector_desc& d, const uint8_t* __restrict__ packed,
                                                       const uint8_t* __restrict__ rec, int tid, int wave, bool record_elsewhere = false) {
	typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
	constexpr int T       = 64 * kDecWaves;
	const bool    is_alp  = d.scheme == ALPGPU_SCHEME_ALP;
	const int     n_units = vector_fits_stage<LDS>(d) ? 8 * (d.bw + (is_alp ? 0 : d.lbw)) : 0;
	const ull2*   g       = reinterpret_cast<const ull2*>(packed + d.packed_off);
	constexpr int kMaxUnits = (LDS::kStage - 128) / 16 < 528 ? (LDS::kStage - 128) / 16 : 528;
#pragma unroll
	for (int j = 0; j < (kMaxUnits + T - 1) / T; ++j) {
		const int c = tid + T * j;
		if (c < n_units) { // LDS destination = wave-uniform base + 16 * lane
			__builtin_amdgcn_global_load_lds(g + c, reinterpret_cast<ull2*>(L.stage) + (T * j + 64 * wave), 16, 0, 0);
		}
	}
	uint32_t  pos = 0u;
	const int cnt = record_elsewhere ? 0 : d.exc_cnt; // (record_elsewhere: the vector's exceptions are patched in after its stores, issue_patch_loads)
	if (cnt > 0) { // wave-uniform
		const uint32_t val_bytes = (is_alp ? 8u : 2u) * static_cast<uint32_t>(cnt);
		const int      dwords    = static_cast<int>(((val_bytes < LDS::kExcBytes ? val_bytes : LDS::kExcBytes) + 3u) >> 2); // (records are 8-byte multiples)
		static_assert(LDS::kExcBytes / 4 <= 2 * T, "two loads per thread cover the stage");
		if (tid < dwords) { __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(rec) + tid, reinterpret_cast<uint32_t*>(L.excv) + 64 * wave, 4, 0, 0); }
		if constexpr (LDS::kExcBytes / 4 > T) {
			if (tid + T < dwords) { __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(rec) + T + tid, reinterpret_cast<uint32_t*>(L.excv) + T + 64 * wave, 4, 0, 0); }
		}
		if (tid < cnt) { pos = reinterpret_cast<const uint16_t*>(rec + val_bytes)[tid]; }
	}
	return pos;
}

template <class LDS>
__device__ __forceinline__ void land_exceptions(LDS& L, const alpgpu_vector_desc& d, const uint8_t* __restrict__ rec, uint32_t pos, int tid) {
	constexpr int T   = 64 * kDecWaves;
	const int     cnt = d.exc_cnt;
	if (tid < cnt) { atomicOr(&L.mask[pos >> 5], 1u << (pos & 31)); }
	if (cnt > T) { // more exceptions than threads in one vector: rare
		const uint16_t* poss = reinterpret_cast<const uint16_t*>(rec + static_cast<size_t>(cnt) * (d.scheme == ALPGPU_SCHEME_ALP ? 8 : 2));
		for (int j = tid + T; j < cnt; j += T) {
			const uint32_t p = poss[j];
			atomicOr(&L.mask[p >> 5], 1u << (p & 31));
		}
	}
}

// V consecutive vectors per workgroup: all of their loads are in flight together, then they are unpacked one after the
// other by the same 4 wavefronts.  V = 2 doubles the bytes in flight per CU for the same residency (8 workgroups per CU).
template <int V, bool NT_STORE, int SINK = kSinkStore, class LDS = DecodeLds>
__global__ __launch_bounds__(64 * kDecWaves) void k_decode_column(const alpgpu_vector_desc* __restrict__ descs,
                                                                  const alpgpu_rowgroup_state* __restrict__ rgs,
                                                                  const uint8_t* __restrict__ packed,
                                                                  const uint8_t* __restrict__ excs, double* __restrict__ out,
                                                                  uint64_t n_vectors, uint64_t wg_offset, double lo, double hi, uint32_t patch_max,
                                                                  uint64_t* __restrict__ progress, uint64_t progress_tag) {
	static_assert(LDS::kStage == kStageBytes || SINK == kSinkStore, "the sinks pass their lane partials through a full stage");
	__shared__ LDS L[V];
	const int      tid  = static_cast<int>(threadIdx.x);
	const int      lane = tid & 63;
	const int      wave = wave_in_wg();
#ifdef ALPGPU_DEC_XCD_GROUP // experiment (round 6, call 40): consecutive workgroups go to consecutive XCDs; here each XCD takes ALPGPU_DEC_XCD_GROUP CONSECUTIVE workgroups' vectors of a span of 8 groups
	uint64_t b_idx = blockIdx.x;
	{
		constexpr uint64_t kG = ALPGPU_DEC_XCD_GROUP, kSpan = 8 * kG;
		const uint64_t     full = (static_cast<uint64_t>(gridDim.x) / kSpan) * kSpan;
		if (b_idx < full) {
			const uint64_t r = b_idx % kSpan;
			b_idx            = b_idx - r + (r & 7u) * kG + (r >> 3);
		}
	}
	const uint64_t v0 = (wg_offset + b_idx) * V;
#else
	const uint64_t v0   = (wg_offset + blockIdx.x) * V;
#endif
	if (v0 >= n_vectors) { return; }
	// An unhinted decode (api_decode.hip) launches every candidate shape; the plan kernel in front of them has written which one runs (bits 8.. of patch_max
	// = this launch's number, 0 = not a candidate: the usual launch).  One scalar load, taken by candidates only; a closed candidate costs its dispatch.
	if (SINK == kSinkStore && (patch_max >> 8) != 0u) { // (kernel argument: uniform)
		if (progress[kCtxWordShape] != static_cast<uint64_t>(patch_max >> 8)) { return; }
		patch_max &= 0xFFu;
	}
	// the read-ahead's pace (read_ahead_kernels.hip): workgroups are dispatched in ascending order, every 128th says where the launch is
	if (SINK == kSinkStore && progress != nullptr && (blockIdx.x & 127u) == 0 && tid == 0) {
		__hip_atomic_store(progress, progress_tag | v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}

	alpgpu_vector_desc d[V];
	uint32_t           pos[V];
#pragma unroll
	for (int i = 0; i < V; ++i) {
		const uint64_t v = v0 + i < n_vectors ? v0 + i : v0; // the odd tail vector is simply loaded twice
		d[i]             = descs[v];
	}
	VectorConsts dict[V];
#pragma unroll
	for (int i = 0; i < V; ++i) { dict[i] = load_vector_consts(rgs, v0 + i < n_vectors ? v0 + i : v0, d[i]); }
	// vectors whose (few) exceptions are written over their stored values afterwards (apply_patches): nothing of them goes through the mask
	bool      patched[V];
	PatchRegs pr[V];
#pragma unroll
	for (int i = 0; i < V; ++i) {
		patched[i] = ALPGPU_DECODE_PATCH_MODE != 0 && SINK == kSinkStore && vector_patches_after(d[i], patch_max); // workgroup-uniform
		pos[i]     = issue_vector_loads(L[i], d[i], packed, excs + d[i].exc_off, tid, wave, patched[i]);
		pr[i]      = PatchRegs {0u, 0ull};
		if (patched[i]) {
			pr[i] = issue_patch_loads(d[i], excs + d[i].exc_off, lane);
			if constexpr (ALPGPU_DECODE_PATCH_MODE == 2) { patch_table_zero(L[i], wave, lane); }
		}
	}
	// ALP_RD vectors: the dictionary into the vector's LDS (four lanes; visible behind the barrier below)
#pragma unroll
	for (int i = 0; i < V; ++i) {
		if (d[i].scheme != ALPGPU_SCHEME_ALP && tid < 4) { reinterpret_cast<uint32_t*>(L[i].rdict)[tid] = static_cast<uint32_t>((tid < 2 ? dict[i].lo : dict[i].hi) >> (32 * (tid & 1))); }
	}
	// Only a workgroup that has exceptions zeroes its masks, and it does so behind the issue of all its loads: the barrier that fences the
	// zeroes from the atomics waits for LDS only, so it falls into the shadow of the HBM round trip.  (Until round 3 every workgroup,
	// exceptions or not, began with the zeroing write and a barrier in front of its first load.)
	bool any_exc = false;
#pragma unroll
	for (int i = 0; i < V; ++i) { any_exc |= d[i].exc_cnt != 0 && !patched[i]; }
	if (any_exc) { // workgroup-uniform
		if (tid < 32 * V) { L[tid >> 5].mask[tid & 31] = 0; }
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // not __syncthreads(): its fence would wait for the loads in flight
#pragma unroll
		for (int i = 0; i < V; ++i) {
			if (!patched[i]) { land_exceptions(L[i], d[i], excs + d[i].exc_off, pos[i], tid); }
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // LDS-DMA completion is not tracked through the LDS for the compiler
	__syncthreads();
	if constexpr (SINK == kSinkStore && ALPGPU_DECODE_PATCH_MODE == 2) {
#pragma unroll
		for (int i = 0; i < V; ++i) {
			if (patched[i]) { patch_table_fill(L[i], pr[i], d[i].exc_cnt, wave, lane); }
		}
	}

	if constexpr (SINK != kSinkStore) {
		// Per-vector sums: lane partial (step order) in each of the four wavefronts -> the four partials of a lane position combined as
		// (w0 + w1) + (w2 + w3) -> ONE adjacent-lane tree over the 64 results, by one wavefront per vector.  (Counts take the same route; they
		// are small integers, so the order does not matter.)  Until late in round 3 every wavefront ran its own tree first: 18 of the ~125
		// vector instructions per wavefront and vector of a kernel whose VALU is 100 % busy (profiles/r03_consumers.txt).  The lane partials
		// travel through the vectors' packed-word stages, which every wavefront is done with behind the barrier below.
		double acc[V];
#pragma unroll
		for (int i = 0; i < V; ++i) {
			acc[i] = 0.0;
			if (v0 + i < n_vectors) {
				if constexpr (SINK == kSinkProbe) {
					const int       n_units = 8 * (d[i].bw + (d[i].scheme == ALPGPU_SCHEME_ALP ? 0 : d[i].lbw));
					const uint64_t* st      = reinterpret_cast<const uint64_t*>(L[i].stage);
					uint64_t        sum     = 0;
					for (int c = tid; c < n_units; c += 64 * kDecWaves) { sum += st[2 * c] + st[2 * c + 1]; }
					acc[i] = static_cast<double>(sum & 0xFFFFFu);
				} else {
					decode_staged_vector<NT_STORE, SINK>(L[i], d[i], dict[i], excs + d[i].exc_off, nullptr, wave, lane, &acc[i], lo, hi);
				}
			}
		}
		__syncthreads(); // nobody reads packed words any more
		static_assert(kStageBytes >= 8 * 64 * kDecWaves, "the lane partials of a vector fit its stage");
#pragma unroll
		for (int i = 0; i < V; ++i) { reinterpret_cast<double*>(L[i].stage)[64 * wave + lane] = acc[i]; }
		__syncthreads();
		if (wave < V && v0 + wave < n_vectors) { // wave-uniform: wavefront w finishes vector w
			const double* part = reinterpret_cast<const double*>(L[wave].stage) + lane;
#ifdef ALPGPU_EXPERIMENT_DEC_WAVES // timing experiments with another number of wavefronts per vector: the sums follow another order
			double total = 0.0;
			for (int w = 0; w < kDecWaves; ++w) { total += part[64 * w]; }
#else
			static_assert(kDecWaves == 4, "the documented summation order is for 4 wavefronts per vector");
			double total = (part[0] + part[64]) + (part[128] + part[192]);
#endif
			total = wave_tree_sum_f64(total); // balanced tree over adjacent lanes (DPP, alp_device.hpp)
			if (lane == 0) {
				if constexpr (SINK == kSinkCount) {
					reinterpret_cast<uint32_t*>(out)[v0 + wave] = static_cast<uint32_t>(total);
				} else {
					out[v0 + wave] = total;
				}
			}
		}
		return;
	}
#pragma unroll
	for (int i = 0; i < V; ++i) {
		if (v0 + i < n_vectors) {
			const alpgpu_vector_desc dd = patched[i] ? without_exceptions(d[i]) : d[i]; // (a patched vector unpacks as one without exceptions)
			if (LDS::kStage == kStageBytes || vector_fits_stage<LDS>(d[i])) { // (always, with the full stage)
				decode_staged_vector<NT_STORE, kSinkStore, LDS>(L[i], dd, dict[i], excs + d[i].exc_off, reinterpret_cast<double2*>(out + (v0 + i) * kVec), wave, lane, nullptr, 0.0, 0.0,
				                                                ALPGPU_DECODE_PATCH_MODE == 2 && patched[i], pr[i]);
			} else { // a wide vector in a narrow-stage launch: its words straight from HBM (bounded buffer loads, as in k_sink_direct)
				ExcMask em {0u, 0};
				if (dd.exc_cnt > 0) { em = load_exception_mask(L[i], lane); }
				uint8_t*          first      = const_cast<uint8_t*>(packed + d[i].packed_off);
				constexpr int     kRsrcFlags = 0x00020000;
				const bool        is_alp     = d[i].scheme == ALPGPU_SCHEME_ALP;
				const BufferWords words {__builtin_amdgcn_make_buffer_rsrc(first, 0, 128 * d[i].bw, kRsrcFlags),
				                         __builtin_amdgcn_make_buffer_rsrc(first + 128u * d[i].bw, 0, is_alp ? 0 : 128 * d[i].lbw, kRsrcFlags)};
				decode_vector_quarters<NT_STORE, kSinkStore, 1>(L[i], words, dd, dict[i], em, excs + d[i].exc_off, reinterpret_cast<double2*>(out + (v0 + i) * kVec), wave, lane,
				                                                nullptr, 0.0, 0.0, ALPGPU_DECODE_PATCH_MODE == 2 && patched[i], pr[i]);
			}
			if constexpr (ALPGPU_DECODE_PATCH_MODE == 1) {
				if (patched[i]) { apply_patches<NT_STORE>(pr[i], d[i].exc_cnt, out + (v0 + i) * kVec, wave, lane); }
			}
		}
	}
}

// ---- experiment (round 4): a workgroup that decides how to run its vectors from their descriptors ---------------------------------------------
// The launch shape of k_decode_column is one decision per COLUMN (from its size hints), but a column's rowgroups differ: on the benchmark column
// (widths 1..53 by rowgroup) the average says "one vector per workgroup" and the narrow third of the rowgroups runs in the shape that is 20-30 %
// slower for it when a whole column looks like that.  Here every workgroup owns TWO consecutive vectors, reads both descriptors at once and decides:
//   PAIRING 1: both narrow -> loads of both in flight together (k_decode_column<2>); else one after the other (the second one's descriptor
//              round trip is already behind it)
//   PAIRING 2: as 1, but the second vector's loads are issued as soon as the first one's have landed, in front of its unpack
//   PAIRING 3: no decision: three vectors per two workgroups (even workgroups two together, odd ones one) — 12 vectors in flight per CU, between
//              the 8 and the 16 of the two shapes of k_decode_column
// Not the default anywhere: tools/sweep_pairing.py measures them (profiles/r04_decode_floor.txt, section 4).
__device__ __forceinline__ bool vector_is_narrow(const alpgpu_vector_desc& d) {
	return d.scheme == ALPGPU_SCHEME_ALP && static_cast<int>(d.bw) <= (d.exc_cnt >= 2 ? 20 : 16);
}
__device__ __forceinline__ void prepare_exceptions(DecodeLds& L, const alpgpu_vector_desc& d, const uint8_t* __restrict__ rec, uint32_t pos, int tid) {
	if (d.exc_cnt != 0) { // workgroup-uniform
		if (tid < 32) { L.mask[tid] = 0; }
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
		land_exceptions(L, d, rec, pos, tid);
	}
}
__device__ __forceinline__ void loads_have_landed() {
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
}
template <bool NT_STORE, int PAIRING>
__global__ __launch_bounds__(64 * kDecWaves) void k_decode_pairs(const alpgpu_vector_desc* __restrict__ descs, const alpgpu_rowgroup_state* __restrict__ rgs,
                                                                 const uint8_t* __restrict__ packed, const uint8_t* __restrict__ excs, double* __restrict__ out,
                                                                 uint64_t n_vectors, uint64_t wg_offset, uint32_t patch_max) {
	__shared__ DecodeLds L[2];
	const int      tid  = static_cast<int>(threadIdx.x);
	const int      lane = tid & 63;
	const int      wave = wave_in_wg();
	const uint64_t g    = wg_offset + blockIdx.x;
	uint64_t       v0;
	int            n_here;
	if constexpr (PAIRING == 3) {
		v0     = 3 * (g >> 1) + ((g & 1) ? 2 : 0);
		n_here = (g & 1) ? 1 : 2;
	} else {
		v0     = 2 * g;
		n_here = 2;
	}
	if (v0 >= n_vectors) { return; }
	if (v0 + n_here > n_vectors) { n_here = 1; }
	const uint64_t           v1 = n_here == 2 ? v0 + 1 : v0;
	const alpgpu_vector_desc d0 = descs[v0], d1 = descs[v1];
	const VectorConsts       c0 = load_vector_consts(rgs, v0, d0), c1 = load_vector_consts(rgs, v1, d1);
	double2*                 o0 = reinterpret_cast<double2*>(out + v0 * kVec);
	double2*                 o1 = reinterpret_cast<double2*>(out + v1 * kVec);
	if (tid < 4) { // ALP_RD vectors: the dictionary into the vector's LDS (visible behind the barrier in front of the vector's unpack)
		if (d0.scheme != ALPGPU_SCHEME_ALP) { reinterpret_cast<uint32_t*>(L[0].rdict)[tid] = static_cast<uint32_t>((tid < 2 ? c0.lo : c0.hi) >> (32 * (tid & 1))); }
		if (d1.scheme != ALPGPU_SCHEME_ALP) { reinterpret_cast<uint32_t*>(L[1].rdict)[tid] = static_cast<uint32_t>((tid < 2 ? c1.lo : c1.hi) >> (32 * (tid & 1))); }
	}
	const bool together = n_here == 2 && (PAIRING == 3 || (vector_is_narrow(d0) && vector_is_narrow(d1))); // workgroup-uniform
	// vectors whose exceptions are patched in after their stores (apply_patches): decoded as if they had none
	// (ALPGPU_DECODE_PATCH_MODE 2 keeps its slot tables in k_decode_column only: here such vectors go through the mask)
	const bool               pa0 = ALPGPU_DECODE_PATCH_MODE == 1 && vector_patches_after(d0, patch_max), pa1 = ALPGPU_DECODE_PATCH_MODE == 1 && vector_patches_after(d1, patch_max); // (compile-time false in the default build)
	const alpgpu_vector_desc m0 = pa0 ? without_exceptions(d0) : d0, m1 = pa1 ? without_exceptions(d1) : d1;
	PatchRegs                r0 {0u, 0ull}, r1 {0u, 0ull};
	if (together) {
		const uint32_t p0 = issue_vector_loads(L[0], m0, packed, excs + d0.exc_off, tid, wave);
		const uint32_t p1 = issue_vector_loads(L[1], m1, packed, excs + d1.exc_off, tid, wave);
		if (pa0) { r0 = issue_patch_loads(d0, excs + d0.exc_off, lane); }
		if (pa1) { r1 = issue_patch_loads(d1, excs + d1.exc_off, lane); }
		if (m0.exc_cnt != 0 || m1.exc_cnt != 0) {
			if (tid < 64) { L[tid >> 5].mask[tid & 31] = 0; }
			asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
			land_exceptions(L[0], m0, excs + d0.exc_off, p0, tid);
			land_exceptions(L[1], m1, excs + d1.exc_off, p1, tid);
		}
		loads_have_landed();
		decode_staged_vector<NT_STORE, kSinkStore, DecodeLds>(L[0], m0, c0, excs + d0.exc_off, o0, wave, lane);
		if (pa0) { apply_patches<NT_STORE>(r0, d0.exc_cnt, out + v0 * kVec, wave, lane); }
		decode_staged_vector<NT_STORE, kSinkStore, DecodeLds>(L[1], m1, c1, excs + d1.exc_off, o1, wave, lane);
		if (pa1) { apply_patches<NT_STORE>(r1, d1.exc_cnt, out + v1 * kVec, wave, lane); }
		return;
	}
	const uint32_t p0 = issue_vector_loads(L[0], m0, packed, excs + d0.exc_off, tid, wave);
	if (pa0) { r0 = issue_patch_loads(d0, excs + d0.exc_off, lane); }
	prepare_exceptions(L[0], m0, excs + d0.exc_off, p0, tid);
	loads_have_landed();
	uint32_t p1 = 0;
	if (PAIRING == 2 && n_here == 2) {
		p1 = issue_vector_loads(L[1], m1, packed, excs + d1.exc_off, tid, wave);
		if (pa1) { r1 = issue_patch_loads(d1, excs + d1.exc_off, lane); }
	}
	decode_staged_vector<NT_STORE, kSinkStore, DecodeLds>(L[0], m0, c0, excs + d0.exc_off, o0, wave, lane);
	if (pa0) { apply_patches<NT_STORE>(r0, d0.exc_cnt, out + v0 * kVec, wave, lane); }
	if (n_here == 2) {
		if (PAIRING != 2) {
			p1 = issue_vector_loads(L[1], m1, packed, excs + d1.exc_off, tid, wave);
			if (pa1) { r1 = issue_patch_loads(d1, excs + d1.exc_off, lane); }
		}
		prepare_exceptions(L[1], m1, excs + d1.exc_off, p1, tid);
		loads_have_landed();
		decode_staged_vector<NT_STORE, kSinkStore, DecodeLds>(L[1], m1, c1, excs + d1.exc_off, o1, wave, lane);
		if (pa1) { apply_patches<NT_STORE>(r1, d1.exc_cnt, out + v1 * kVec, wave, lane); }
	}
}

// ---- the sinks with ONE wavefront per vector, packed words read from HBM as they are needed (no stage, no barrier) ---------------------
// What the four-wavefront sinks run out of is instruction issue — the scalar unit first, the VALU right behind — because every wavefront of a
// workgroup repeats the wave-uniform prologue of both its vectors (profiles/r03_consumers.txt).  Here a wavefront owns a vector: one
// prologue per vector, no cross-wavefront reduction, no barrier; without a stage its LDS is the exception mask and values only (1.2 KiB), so
// eight wavefronts per SIMD stay resident and hide the two dependent round trips (descriptor -> packed words) between them.  A lane reads
// the two 16-byte units of each of its eight pairs straight from HBM (buffer loads bounded to the vector's words): 16 loads per wavefront
// whatever the width; the 64 lanes of a load touch 8 runs of 128 bytes.  Results are bit-identical to k_decode_column's sinks: the lane keeps the four quarter partials p[q][L] apart,
// then (p0 + p1) + (p2 + p3), then the adjacent-lane tree (include/alpgpu.h).
#ifndef ALPGPU_SINK_DIRECT_OCC
#define ALPGPU_SINK_DIRECT_OCC 8 // wavefronts per SIMD the register budget is sized for (8 -> <= 64 VGPRs; measured against 5 and 6: profiles/r03_consumers.txt)
#endif
#ifndef ALPGPU_SINK_STAGE
#define ALPGPU_SINK_STAGE 3584 // bytes of packed words (bit widths <= 28) a wavefront of k_sink_direct stages in its LDS by LDS-DMA; with mask, values and prefixes 4992 B per wavefront = eight workgroups per CU (0: none)
#endif
#ifndef ALPGPU_SINK_STAGE_MAX_EXC
#define ALPGPU_SINK_STAGE_MAX_EXC 48 // ... only for vectors with at most this many exceptions
#endif
struct __attribute__((aligned(16))) SinkWaveLds {
	static constexpr bool kPrefixInLds = true; // exception_hits_lds
#if ALPGPU_SINK_STAGE > 0
	uint8_t  stage[ALPGPU_SINK_STAGE + 128]; // + the unit row past the end that the unpack reads and masks off
#endif
	uint32_t mask[32];
	static constexpr uint32_t kExcBytes = kExcStageBytes;
	uint8_t  excv[kExcStageBytes];
	uint32_t pref[32]; // exceptions in front of mask word i
	uint16_t rdict[8]; // ALP_RD: the rowgroup's dictionary (DecodeLdsT)
};
template <int SINK>
__global__ __launch_bounds__(64 * kDecWaves, ALPGPU_SINK_DIRECT_OCC) void k_sink_direct(const alpgpu_vector_desc* __restrict__ descs, const alpgpu_rowgroup_state* __restrict__ rgs,
                                                                   const uint8_t* __restrict__ packed, const uint8_t* __restrict__ excs, double* __restrict__ out,
                                                                   uint64_t n_vectors, uint64_t wg_offset, double lo, double hi) {
	// gfx950 range-checks the single-register AMOUNT of a 64-bit shift (v_lshrrev_b64 / v_lshlrev_b64, which this kernel lives on) as a register
	// PAIR: in the LAST register of the allocation it counts as out of range and VGPR0 is read instead (tools/last_vgpr_probe.hip).  Builds
	// of this kernel whose 64 registers were all in use with a shift amount in v63 returned wrong sums for ~5 % of some vectors; the
	// register allocator does not know the rule.  tools/check_top_vgpr.py (run by tests/test_build_rules.py) looks for the pattern in every
	// kernel of every build; -DALPGPU_SINK_REGISTER_MARGIN allocates one register more than the kernel uses instead (72 registers, 7
	// wavefronts per SIMD, 3-6 % slower: measured in profiles/r03_consumers.txt), which makes the pattern impossible here.
#ifdef ALPGPU_SINK_REGISTER_MARGIN
	asm volatile("" ::: "v64");
#endif
	__shared__ SinkWaveLds S[kDecWaves];
	const int      lane = static_cast<int>(threadIdx.x) & 63;
	const int      wave = wave_in_wg();
	const uint64_t v    = (wg_offset + blockIdx.x) * kDecWaves + wave;
	if (v >= n_vectors) { return; } // wave-uniform; no barrier anywhere in this kernel
	SinkWaveLds&             L    = S[wave];
	const alpgpu_vector_desc d    = descs[v];
	const VectorConsts       dict = load_vector_consts(rgs, v, d);
	const uint8_t*           rec  = excs + d.exc_off;
	const bool               is_alp = d.scheme == ALPGPU_SCHEME_ALP;
	const int                cnt  = d.exc_cnt;
	ExcMask                  em {0u, 0};
#if ALPGPU_SINK_STAGE > 0
	// a narrow ALP vector's words whole into the wavefront's LDS by LDS-DMA (1 KiB per instruction, no registers): ONE round trip for all of
	// them instead of two batches of register loads
	const bool staged = is_alp && 128u * d.bw <= static_cast<uint32_t>(ALPGPU_SINK_STAGE) && cnt <= ALPGPU_SINK_STAGE_MAX_EXC; // wave-uniform
	if (staged) {
		typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
		const ull2* g       = reinterpret_cast<const ull2*>(packed + d.packed_off);
		const int   n_units = 8 * d.bw;
		for (int j = 0; 64 * j < n_units; ++j) {
			if (64 * j + lane < n_units) { __builtin_amdgcn_global_load_lds(g + 64 * j + lane, reinterpret_cast<ull2*>(L.stage) + 64 * j, 16, 0, 0); }
		}
	}
#endif
	if (cnt > 0) { // wave-uniform: values of the first kExcStage exceptions by LDS-DMA, the mask from the positions
		const uint32_t val_bytes = (is_alp ? 8u : 2u) * static_cast<uint32_t>(cnt);
		const int      dwords    = static_cast<int>(((val_bytes < kExcStageBytes ? val_bytes : kExcStageBytes) + 3u) >> 2);
		for (int q = 0; 64 * q < dwords; ++q) { // wave-uniform trip count; LDS destination = wave-uniform base + 4 * lane
			if (64 * q + lane < dwords) { __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(rec) + 64 * q + lane, reinterpret_cast<uint32_t*>(L.excv) + 64 * q, 4, 0, 0); }
		}
		if (lane < 32) { L.mask[lane] = 0u; }
		wave_lds_sync();
		const uint16_t* poss = reinterpret_cast<const uint16_t*>(rec + val_bytes);
		for (int j = lane; j < cnt; j += 64) {
			const uint32_t p = poss[j];
			atomicOr(&L.mask[p >> 5], 1u << (p & 31u));
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the DMA'd values (LDS-DMA completion is not tracked through the LDS for the compiler)
		wave_lds_sync();
		em = load_exception_mask(L, lane);
		if (lane < 32) { L.pref[lane] = static_cast<uint32_t>(em.excl); }
		wave_lds_sync();
	}
	uint8_t*          first   = const_cast<uint8_t*>(packed + d.packed_off);
	constexpr int     kRsrcFlags = 0x00020000; // gfx9 raw buffer, 32-bit data format
	const BufferWords words {__builtin_amdgcn_make_buffer_rsrc(first, 0, 128 * d.bw, kRsrcFlags),
	                         __builtin_amdgcn_make_buffer_rsrc(first + 128u * d.bw, 0, is_alp ? 0 : 128 * d.lbw, kRsrcFlags)};
	if (!is_alp) { // wave-uniform: the dictionary into the wavefront's LDS
		if (lane < 4) { reinterpret_cast<uint32_t*>(L.rdict)[lane] = static_cast<uint32_t>((lane < 2 ? dict.lo : dict.hi) >> (32 * (lane & 1))); }
		wave_lds_sync();
	}
	double part[kDecWaves] = {0.0, 0.0, 0.0, 0.0};
#if ALPGPU_SINK_STAGE > 0
	if (staged) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		wave_lds_sync();
		decode_vector_quarters<false, SINK, kDecWaves, 1>(L, StagedWords {L.stage}, d, dict, em, rec, nullptr, 0, lane, part, lo, hi);
	} else
#endif
	if (is_alp) { // wave-uniform
		decode_vector_quarters<false, SINK, kDecWaves, 1>(L, words, d, dict, em, rec, nullptr, 0, lane, part, lo, hi);
	} else {
		// ALP_RD a quarter at a time: the right AND left words of eight steps in flight at once do not fit this kernel's 64 registers; with the two
		// pins in decode_vector_quarters' ALP_RD arm nothing is spilled (1.41 ms per 1 Mi vectors of an all-ALP_RD column, staged kernel 1.51;
		// without them 208 bytes of scratch and 5.4 ms)
#pragma unroll
		for (int q = 0; q < kDecWaves; ++q) {
			decode_vector_quarters<false, SINK, 1, 2>(L, words, d, dict, em, rec, nullptr, q, lane, &part[q], lo, hi);
			asm volatile("" ::: "memory"); // the next quarter's requests stay behind this one's use
		}
	}
	static_assert(kDecWaves == 4, "the documented summation order is for 4 quarters per vector");
	double total = (part[0] + part[1]) + (part[2] + part[3]);
	total        = wave_tree_sum_f64(total);
	if (lane == 0) {
		if constexpr (SINK == kSinkCount) {
			reinterpret_cast<uint32_t*>(out)[v] = static_cast<uint32_t>(total);
		} else {
			out[v] = total;
		}
	}
}

int launch_sink_direct(hipStream_t stream, const alpgpu_column* col, double lo, double hi, void* d_out, bool count) {
	const uint64_t n        = col->n_vectors;
	const uint64_t n_wg     = (n + kDecWaves - 1) / kDecWaves;
	const uint64_t kMaxGrid = 1ull << 30;
	// experiment (ALPGPU_SINK_PAD_LDS_KIB): unused dynamic LDS that caps the workgroups resident per CU, as launch_decode_column does by width
	static const unsigned pad_lds = std::getenv("ALPGPU_SINK_PAD_LDS_KIB") ? static_cast<unsigned>(std::atoi(std::getenv("ALPGPU_SINK_PAD_LDS_KIB"))) * 1024u : 0u;
	for (uint64_t off = 0; off < n_wg; off += kMaxGrid) {
		const dim3 grid(static_cast<unsigned>(n_wg - off < kMaxGrid ? n_wg - off : kMaxGrid)), block(64 * kDecWaves);
		if (count) {
			hipLaunchKernelGGL((k_sink_direct<kSinkCount>), grid, block, pad_lds, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, static_cast<double*>(d_out), n, off, lo, hi);
		} else {
			hipLaunchKernelGGL((k_sink_direct<kSinkSum>), grid, block, pad_lds, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, static_cast<double*>(d_out), n, off, 0.0, 0.0);
		}
	}
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

// whether this build has an arm that patches exceptions in after the stores (-DALPGPU_DECODE_PATCH_MODE=1 / 2; the default build has none: ALPGPU_OPT_DECODE_PATCH_AFTER is refused)
bool decode_patch_arm_compiled() { return ALPGPU_DECODE_PATCH_MODE != 0; }

int launch_decode_column(hipStream_t stream, const alpgpu_column* col, double* d_out, int variant, int n_cus, uint32_t patch_max, uint64_t* progress, uint64_t progress_tag, uint32_t gate) {
	(void)n_cus;
	if (patch_max > 64u) { patch_max = 64u; } // one lane per patched exception (apply_patches)
	if (gate != 0 && progress != nullptr) { patch_max |= gate << 8; } // (k_decode_column: a candidate launch of an unhinted decode)
	const uint64_t n = col->n_vectors;
	// variant bit 0: one vector per workgroup (default) instead of two; bit 1: plain instead of non-temporal stores; bit 2: FOUR vectors per
	// workgroup over the narrow stage (columns of <= 16-bit vectors)
	const bool     nt       = !(variant & 2);
	const int      pairing  = (variant >> 3) & 3; // experiment: k_decode_pairs
	if (pairing != 0) {
		const unsigned pad_lds_p = static_cast<unsigned>((variant >> 8) & 0xFF) * 1024u; // residency cap, as below
		const uint64_t n_wg_p   = pairing == 3 ? 2 * ((n + 2) / 3) : (n + 1) / 2;
		const uint64_t kMaxGridP = 1ull << 30;
		for (uint64_t off = 0; off < n_wg_p; off += kMaxGridP) {
			const dim3 grid(static_cast<unsigned>(n_wg_p - off < kMaxGridP ? n_wg_p - off : kMaxGridP)), block(64 * kDecWaves);
#define ALPGPU_LAUNCH_PAIRS(NT, P) hipLaunchKernelGGL((k_decode_pairs<NT, P>), grid, block, pad_lds_p, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, d_out, n, off, patch_max)
			if (pairing == 1) { if (nt) { ALPGPU_LAUNCH_PAIRS(true, 1); } else { ALPGPU_LAUNCH_PAIRS(false, 1); } }
			if (pairing == 2) { if (nt) { ALPGPU_LAUNCH_PAIRS(true, 2); } else { ALPGPU_LAUNCH_PAIRS(false, 2); } }
			if (pairing == 3) { if (nt) { ALPGPU_LAUNCH_PAIRS(true, 3); } else { ALPGPU_LAUNCH_PAIRS(false, 3); } }
#undef ALPGPU_LAUNCH_PAIRS
		}
		return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
	}
	const int      V        = (variant & 4) ? 4 : ((variant & 1) ? 1 : 2);
	// Unused dynamic LDS that caps the workgroups resident per CU (variant bits 8.. = KiB).  Wide vectors want FEWER
	// streams in flight per CU than the eight the wavefront slots allow: a column of 40-53-bit vectors decodes at 0.81 of the HBM peak with six
	// workgroups per CU and at 0.75 with eight, 34-38 bits like seven; up to 33 bits eight are best (tools/sweep_residency.py,
	// profiles/r04_decode_floor.txt section 6).  decode_variant_for (api_decode.hip) sets it from the column's size hints.
	const unsigned pad_lds  = static_cast<unsigned>((variant >> 8) & 0xFF) * 1024u;
	const uint64_t n_wg     = (n + V - 1) / V;
	const uint64_t kMaxGrid = 1ull << 30; // a grid dimension holds < 2^31 workgroups -> chunk very long columns
	for (uint64_t off = 0; off < n_wg; off += kMaxGrid) {
		const dim3 grid(static_cast<unsigned>(n_wg - off < kMaxGrid ? n_wg - off : kMaxGrid)), block(64 * kDecWaves);
		if (V == 4 && nt) {
			hipLaunchKernelGGL((k_decode_column<4, true, kSinkStore, DecodeLdsNarrow>), grid, block, pad_lds, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, d_out, n, off, 0.0, 0.0, patch_max, progress, progress_tag);
		} else if (V == 4) {
			hipLaunchKernelGGL((k_decode_column<4, false, kSinkStore, DecodeLdsNarrow>), grid, block, pad_lds, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, d_out, n, off, 0.0, 0.0, patch_max, progress, progress_tag);
		} else if (V == 2 && nt) {
			hipLaunchKernelGGL((k_decode_column<2, true>), grid, block, pad_lds, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, d_out, n, off, 0.0, 0.0, patch_max, progress, progress_tag);
		} else if (V == 2) {
			hipLaunchKernelGGL((k_decode_column<2, false>), grid, block, pad_lds, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, d_out, n, off, 0.0, 0.0, patch_max, progress, progress_tag);
		} else if ((variant & 64) && nt) { // one vector per workgroup, the 256-entry exception stage (columns of exception-heavy vectors)
			hipLaunchKernelGGL((k_decode_column<1, true, kSinkStore, DecodeLdsManyExc>), grid, block, pad_lds, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, d_out, n, off, 0.0, 0.0, patch_max, progress, progress_tag);
		} else if (nt) {
			hipLaunchKernelGGL((k_decode_column<1, true>), grid, block, pad_lds, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, d_out, n, off, 0.0, 0.0, patch_max, progress, progress_tag);
		} else {
			hipLaunchKernelGGL((k_decode_column<1, false>), grid, block, pad_lds, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, d_out, n, off, 0.0, 0.0, patch_max, progress, progress_tag);
		}
	}
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

// vectors_per_wg in {1, 2}.  There is no output stream to compete with, so two vectors in flight per workgroup is the default
// (measured on the benchmark column: 1.39 / 1.16 / 1.43 ms for 1 / 2 / 4 vectors per workgroup).
int launch_decode_sum(hipStream_t stream, const alpgpu_column* col, double* d_sums, int vectors_per_wg) {
	const uint64_t n        = col->n_vectors;
	const int      V        = vectors_per_wg == 1 ? 1 : 2;
	const uint64_t n_wg     = (n + V - 1) / V;
	const uint64_t kMaxGrid = 1ull << 30;
	for (uint64_t off = 0; off < n_wg; off += kMaxGrid) {
		const dim3 grid(static_cast<unsigned>(n_wg - off < kMaxGrid ? n_wg - off : kMaxGrid)), block(64 * kDecWaves);
		if (V == 2) {
			hipLaunchKernelGGL((k_decode_column<2, false, kSinkSum>), grid, block, 0, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, d_sums, n, off, 0.0, 0.0, 0u, static_cast<uint64_t*>(nullptr), 0ull);
		} else {
			hipLaunchKernelGGL((k_decode_column<1, false, kSinkSum>), grid, block, 0, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, d_sums, n, off, 0.0, 0.0, 0u, static_cast<uint64_t*>(nullptr), 0ull);
		}
	}
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

// measurement aid: the fused consumers' loads, barrier and reduction with the unpack left out (two vectors per workgroup, like them)
int launch_decode_probe(hipStream_t stream, const alpgpu_column* col, double* d_sums) {
	const uint64_t n        = col->n_vectors;
	const uint64_t n_wg     = (n + 1) / 2;
	const uint64_t kMaxGrid = 1ull << 30;
	for (uint64_t off = 0; off < n_wg; off += kMaxGrid) {
		const dim3 grid(static_cast<unsigned>(n_wg - off < kMaxGrid ? n_wg - off : kMaxGrid)), block(64 * kDecWaves);
		hipLaunchKernelGGL((k_decode_column<2, false, kSinkProbe>), grid, block, 0, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, d_sums, n, off, 0.0, 0.0, 0u, static_cast<uint64_t*>(nullptr), 0ull);
	}
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

// per-vector number of values in [lo, hi]: the same kernel with the predicate as its consumer (two vectors per workgroup)
int launch_decode_count_range(hipStream_t stream, const alpgpu_column* col, double lo, double hi, uint32_t* d_counts) {
	const uint64_t n        = col->n_vectors;
	const uint64_t n_wg     = (n + 1) / 2;
	const uint64_t kMaxGrid = 1ull << 30;
	for (uint64_t off = 0; off < n_wg; off += kMaxGrid) {
		const dim3 grid(static_cast<unsigned>(n_wg - off < kMaxGrid ? n_wg - off : kMaxGrid)), block(64 * kDecWaves);
		hipLaunchKernelGGL((k_decode_column<2, false, kSinkCount>), grid, block, 0, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc,
		                   reinterpret_cast<double*>(d_counts), n, off, lo, hi, 0u, static_cast<uint64_t*>(nullptr), 0ull);
	}
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

} // namespace alpgpu
