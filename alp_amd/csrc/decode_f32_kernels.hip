// decode_f32_kernels.hip — fused ALP / ALP_RD column decode for gfx950, single precision.
//
// Replaces, per vector (reference file:line relative to /root/reference):
//   ALP    : generated::falp::fallback::scalar::falp (float)  include/alp/falp.hpp:28-44, src/falp.cpp 32-bit section
//            + alp::decoder<float>::patch_exceptions          include/alp/decoder.hpp:141-149
//   ALP_RD : unffor::unffor (u32 right, u16 left)             include/fastlanes/unffor.hpp:7-15
//            + alp::rd_encoder<float>::decode                 include/alp/rd.hpp:152-178
//
// Same structure as decode_kernels.hip (the measurements that motivate it are listed there): the hardware dispatcher
// hands out small workgroups in vector order; the packed words go HBM -> LDS by LDS-DMA, the exception record becomes
// a 1024-bit mask + staged values, one barrier sits behind the memory latency.  A float vector is 4 KiB of output, so a
// 256-thread workgroup covers one vector with ONE 16-byte store per thread (thread t owns values 4t..4t+3 = one 16-byte
// unit of the FastLanes u32 layout, alp_device_f32.hpp), and V consecutive vectors per workgroup keep the bytes in flight
// per CU at the level of the double-precision kernel (V = 2 moves as many bytes per workgroup as one double vector).
#include "alp_device_f32.hpp"
#include "decode_f32_device.hpp"
#include "decode_policy.hpp"
#include "launch.hpp"
#include <cstdlib>

namespace alpgpu {

constexpr int kDecThreadsF  = 256;
constexpr int kStageBytesF  = 4480; // >= 31*128 (RD right) + 3*128 (RD left) + 128 pad, and >= 32*128 + 128 (ALP bw 32)
#ifndef ALPGPU_EXC_STAGE_F
#define ALPGPU_EXC_STAGE_F 256 // (round 5, late: 128 -> 256.  bench.py's decimal float column carries 94 exceptions per vector on average, 39 % of its vectors more than 128: decode 0.72-0.73 -> 0.75-0.76; the others unchanged)
#endif
constexpr int kExcStageF    = ALPGPU_EXC_STAGE_F; // exception values (4 bytes; ALP_RD: twice as many 2-byte ones) staged in LDS per vector

struct __attribute__((aligned(16))) DecodeLdsF32 {
	static constexpr bool kPrefixInLds = false; // exception lookup by ds_bpermute
	uint8_t  stage[kStageBytesF];
	uint32_t mask[32];
	uint8_t  excv[4 * kExcStageF]; // the head of the exception record as it lies in the stream (values first), brought in by LDS-DMA
	uint16_t rdict[8];             // ALP_RD: the rowgroup's dictionary, looked up by ds_read_u16 (round 6: the lookup in registers — a 3-way select through a 64-bit shift — was 13 of the ~27 vector instructions per ALP_RD value)
};

struct ExcMaskF {
	uint32_t word;
	int      excl;
};
// lane l < 32: mask word l and the number of exceptions in the words before it (DPP row scan, see decode_kernels.hip)
template <class LDS>
__device__ __forceinline__ ExcMaskF load_exception_mask_f32(const LDS& L, int lane) {
	ExcMaskF  m;
	m.word      = L.mask[lane & 31];
	const int c = lane < 32 ? __builtin_popcount(m.word) : 0;
	int       v = c;
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
	m.excl = v - c;
	return m;
}

// the value of the exception of that rank: staged (LDS) or, past the stage, from HBM.  all_staged (wave-uniform: the vector's count fits the stage)
// keeps the common case to the LDS read alone.  Until round 5 this read "LDS if rank < kStaged else HBM": the compiler selected between the two
// ADDRESSES and issued ONE FLAT load per exception (v_cndmask on pointers, src_shared_base) — slower than either, and it waits for every counter
// (decode_kernels.hip: fetch_exception had been cured of the same in round 3; found again in the float sink's ISA, profiles/r05_float_sink.txt).
template <int VAL_BYTES, class LDS>
__device__ __forceinline__ uint32_t fetch_exception_f32(const LDS& L, const uint8_t* __restrict__ rec, int rank, bool all_staged) {
	constexpr int kStaged = 4 * kExcStageF / VAL_BYTES;
	const int     at      = rank < kStaged ? rank : kStaged - 1;
	uint32_t      v;
	if constexpr (VAL_BYTES == 4) {
		v = reinterpret_cast<const uint32_t*>(L.excv)[at];
	} else {
		v = reinterpret_cast<const uint16_t*>(L.excv)[at];
	}
	asm volatile("" : "+v"(v)); // keeps the two loads two loads
	if (!all_staged) {
		if (rank >= kStaged) {
			if constexpr (VAL_BYTES == 4) {
				v = reinterpret_cast<const uint32_t*>(rec)[rank];
			} else {
				v = reinterpret_cast<const uint16_t*>(rec)[rank];
			}
#ifdef ALPGPU_F32_EXC_WAIT_IN_BRANCH // A/B (round 5): waited for HERE instead of at the join in front of the quad's store — measured 4 % SLOWER on columns whose vectors carry more exceptions than the stage holds (every such fetch then waits on its own)
			asm volatile("" : "+v"(v));
#endif
		}
	}
	return v;
}

template <bool NT_STORE>
__device__ __forceinline__ void store_quad(float* __restrict__ p, const u32x4& bits) {
	if constexpr (NT_STORE) {
		__builtin_nontemporal_store(bits, reinterpret_cast<u32x4*>(p));
	} else {
		*reinterpret_cast<u32x4*>(p) = bits;
	}
}

// packed words and exception values by LDS-DMA, positions into a register (decode_kernels.hip: issue_vector_loads)
__device__ __forceinline__ uint32_t issue_vector_loads_f32(DecodeLdsF32& L, const alpgpu_vector_desc& d, const uint8_t* __restrict__ packed,
                                                           const uint8_t* __restrict__ rec, int tid, int wave) {
	typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
	const bool  is_alp  = d.scheme == ALPGPU_SCHEME_ALP;
	const int   n_units = 8 * (d.bw + (is_alp ? 0 : d.lbw));
	const ull2* g       = reinterpret_cast<const ull2*>(packed + d.packed_off);
#pragma unroll
	for (int j = 0; j < 2; ++j) { // <= 272 units
		const int c = tid + kDecThreadsF * j;
		if (c < n_units) { __builtin_amdgcn_global_load_lds(g + c, reinterpret_cast<ull2*>(L.stage) + (kDecThreadsF * j + 64 * wave), 16, 0, 0); }
	}
	uint32_t  pos = 0u;
	const int cnt = d.exc_cnt;
	if (cnt > 0) { // wave-uniform
		const uint32_t val_bytes = (is_alp ? 4u : 2u) * static_cast<uint32_t>(cnt);
		const int      dwords    = static_cast<int>(((val_bytes < 4u * kExcStageF ? val_bytes : 4u * kExcStageF) + 3u) >> 2); // (records are 8-byte multiples)
		static_assert(kExcStageF <= kDecThreadsF, "one load per thread covers the stage");
		if (tid < dwords) { __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(rec) + tid, reinterpret_cast<uint32_t*>(L.excv) + 64 * wave, 4, 0, 0); }
		if (tid < cnt) { pos = reinterpret_cast<const uint16_t*>(rec + val_bytes)[tid]; }
	}
	return pos;
}

__device__ __forceinline__ void land_exceptions_f32(DecodeLdsF32& L, const alpgpu_vector_desc& d, const uint8_t* __restrict__ rec, uint32_t pos, int tid) {
	const int cnt = d.exc_cnt;
	if (tid < cnt) { atomicOr(&L.mask[pos >> 5], 1u << (pos & 31)); }
	if (cnt > kDecThreadsF) {
		const uint16_t* poss = reinterpret_cast<const uint16_t*>(rec + static_cast<size_t>(cnt) * (d.scheme == ALPGPU_SCHEME_ALP ? 4 : 2));
		for (int j = tid + kDecThreadsF; j < cnt; j += kDecThreadsF) {
			const uint32_t p = poss[j];
			atomicOr(&L.mask[p >> 5], 1u << (p & 31));
		}
	}
}

// The per-vector constants, read in front of the barrier: ALP_RD = the rowgroup's dictionary (RdDict, alp_device.hpp); ALP vectors use
// the same two words for lo = FACT_ARR[f], hi = bits of FRAC_ARR[e]
__device__ __forceinline__ RdDict load_vector_consts_f32(const alpgpu_rowgroup_state* __restrict__ rgs, uint64_t v, const alpgpu_vector_desc& d) {
	if (d.scheme != ALPGPU_SCHEME_ALP) { return load_rd_dict(rgs, v, true); } // wave-uniform
	return RdDict {static_cast<uint64_t>(kFactArrF[d.f]), static_cast<uint64_t>(__float_as_uint(kFracArrF[d.e]))};
}

// one quad, after its words have arrived: thread tid (of 256: wavefront `wave`, lane) owns values 4*tid .. 4*tid+3; `em` = the vector's
// exception mask as this wavefront sees it (ignored when the vector has none), L = where staged exception values live.
// SINK (as in decode_kernels.hip) = kSinkStoreF: the quad is stored.  kSinkSumF: its four values are widened to double
// (exact) and added to `acc` in index order.  kSinkCountF: `acc` counts the values v with lo <= v <= hi (NaN never does).
constexpr int kSinkStoreF = 0, kSinkSumF = 1, kSinkCountF = 2;
__device__ __constant__ const uint32_t kShortcutBoundF[11] = {16777216u, 16777216u, 16777216u, 2147483u, 214748u, 21474u, 2147u, 214u, 21u, 2u, 0u};
// HBM_EXC = false: the caller knows (wave-uniform) that every exception value of the vector is in the LDS stage — this instance has no load from HBM in it, so
// the compiler puts no s_waitcnt vmcnt(0) in front of the quad's store (stores count in vmcnt on gfx9: with the rare load in the code, every store of a
// wavefront waited for the one before it — decode_kernels.hip cured the double kernel of this in round 5, the float kernel followed late in the round).
template <bool NT_STORE, int SINK, class LDS, bool HBM_EXC = true>
__device__ __forceinline__ void finish_quad_f32(const LDS& L, const QuadWords& w, const alpgpu_vector_desc& d, const RdDict& dict, const ExcMaskF& em,
                                                const uint8_t* __restrict__ rec, float* __restrict__ dst, int tid, int wave, int lane, double* acc,
                                                float range_lo, float range_hi) {
	const int bw   = d.bw;
	const int cnt  = d.exc_cnt;
	const int a    = tid & 7;
	const int row  = tid >> 3;
	uint32_t  hits = 0;
	int       rank = 0;
	const bool all_staged = !HBM_EXC || cnt <= (d.scheme == ALPGPU_SCHEME_ALP ? kExcStageF : 2 * kExcStageF); // wave-uniform: every exception value of the vector is in the LDS stage
	// Round 6: the SINKS skip an ALP vector's exception positions here (the position counts as +0.0 / as outside the range) and take the exception VALUES in one
	// pass behind the quads (sink_exception_values_f32) — the per-value lookup (rank, four divergent blocks per quad, an LDS read and its wait each) was ~170 of
	// the ~540 vector instructions per vector of an exception-heavy column (profiles/r05_float_sink.txt); include/alpgpu.h documents the order.  ALP_RD vectors
	// (few exceptions, and an exception there is a LEFT part that needs its position's right part) keep the lookup in place.
	constexpr bool kApart = SINK != kSinkStoreF;
	if (kApart && cnt > 0 && d.scheme == ALPGPU_SCHEME_ALP) {
		const int wi = 8 * wave + (lane >> 3);
		uint32_t  word;
		if constexpr (LDS::kPrefixInLds) {
			word = L.mask[wi];
		} else {
			word = static_cast<uint32_t>(__shfl(static_cast<int>(em.word), wi));
		}
		hits = (word >> (4 * a)) & 0xFu;
	} else if (cnt > 0) { // the quad's four mask bits and the rank of its first exception
		const int wi = 8 * wave + (lane >> 3);
		uint32_t  word;
		int       pref;
		if constexpr (LDS::kPrefixInLds) { // k_sink_direct_f32: out of the wavefront's LDS, not by ds_bpermute while its loads are in flight (decode_kernels.hip: exception_hits_lds)
			word = L.mask[wi];
			pref = static_cast<int>(L.pref[wi]);
		} else {
			word = static_cast<uint32_t>(__shfl(static_cast<int>(em.word), wi));
			pref = __shfl(em.excl, wi);
		}
		const int      b0   = 4 * a;
		hits                = (word >> b0) & 0xFu;
		rank                = pref + __builtin_popcount(word & ((1u << b0) - 1u));
	}
	// FastLanes u32 unpack (alp_device_f32.hpp: unpack_quad_u32): (w1 << (32 - s)) without the undefined shift by 32 when s == 0
	// ... which is the 64-bit funnel {w1, w0} >> s: ONE v_alignbit_b32 per value (it takes the amount modulo 32, so s = 0 yields w0), where the
	// spelled-out form cost four (round 4: the one-wavefront sink is bound by its vector instructions — 489 per vector before this)
	const uint32_t s    = static_cast<uint32_t>(row * bw) & 31u;
	const uint32_t msk  = bw_mask32(bw);
	u32x4          q;
#pragma unroll
	for (int c = 0; c < 4; ++c) { q[c] = __builtin_amdgcn_alignbit(w.w1[c], w.w0[c], s) & msk; }
#if defined(ALPGPU_SINKF_STOP_AT) && ALPGPU_SINKF_STOP_AT == 4 // measurement builds (profiles/r05_float_sink.txt): the quad ends behind its unpack
	if constexpr (SINK != kSinkStoreF) {
		*acc += static_cast<double>(q[0] ^ q[1] ^ q[2] ^ q[3] ^ hits);
		return;
	}
#endif
	u32x4          out;
	if (d.scheme == ALPGPU_SCHEME_ALP) {
		const uint32_t base = static_cast<uint32_t>(d.base);
		const uint32_t fact = static_cast<uint32_t>(dict.lo);
		const float    frac = __uint_as_float(static_cast<uint32_t>(dict.hi));
		// Conversion shortcut, decided once per vector from its descriptor (wave-uniform, scalar arithmetic; the double kernels' kDecodeTab idea):
		// if every integer base + digit of the vector lies in [-2^24, 2^24] and times 10^f stays inside int32, then (float)(int32)(value * 10^f)
		// — a quarter-rate 32-bit integer multiply and a conversion — is the correctly rounded product of two exactly representable floats, i.e.
		// the IEEE product (float)value * 10^f (10^f = 2^f 5^f with 5^10 < 2^24: exact for every f <= 10): same bits, one full-rate multiply.
		// (the bound from a table — round 5: as a division it was ~25 scalar and vector instructions, and the one-wavefront sink, which runs this
		//  function four times per vector and is bound by its instruction issue, did it four times)
		const int64_t lo64 = static_cast<int64_t>(static_cast<int32_t>(base)), hi64 = lo64 + static_cast<int64_t>(bw_mask32(bw));
		const int64_t bnd  = static_cast<int64_t>(kShortcutBoundF[d.f]); // min(2^24, (2^31 - 1) / 10^f) for f <= 9; 0 for f = 10 (10^10 does not fit: literal path)
		const bool    shortcut = bw <= 24 && lo64 >= -bnd && hi64 <= bnd;
		if (shortcut) {
			const float fact_f = static_cast<float>(fact);
#pragma unroll
			for (int c = 0; c < 4; ++c) { out[c] = __float_as_uint((static_cast<float>(static_cast<int32_t>(q[c] + base)) * fact_f) * frac); }
		} else {
#pragma unroll
			for (int c = 0; c < 4; ++c) { out[c] = __float_as_uint(decode_value_f32(static_cast<int32_t>(q[c] + base), fact, frac)); }
		}
#if defined(ALPGPU_SINKF_STOP_AT) && ALPGPU_SINKF_STOP_AT == 5 // ... behind the conversion
		if constexpr (SINK != kSinkStoreF) {
			*acc += static_cast<double>(out[0] ^ out[1] ^ out[2] ^ out[3] ^ hits);
			return;
		}
#endif
		if constexpr (kApart) {
			// (the sinks: an exception position contributes +0.0f — (double)(+0.0f) added to a partial that is never -0.0 leaves it as it is — or, for COUNT, a NaN)
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				if (hits & (1u << c)) { out[c] = SINK == kSinkCountF ? 0x7FC00000u : 0u; }
			}
		} else if (hits) { // (a branch-free form — every lane reads the staged value it WOULD take, then selects — was measured in round 5: no difference, 64 VGPRs; profiles/r05_float_sink.txt)
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				if (hits & (1u << c)) {
					out[c] = fetch_exception_f32<4>(L, rec, rank, all_staged);
					++rank;
				}
			}
		}
	} else {
		// ALP_RD: right parts = u32 lanes (bw = rbw, base 0); left parts = u16 lanes, 64 streams x 16 rows (value i ->
		// lane64 = i & 63, row = i >> 6, word k at left[64*k + lane64]): the quad shares row tid >> 4 and is the aligned
		// 8-byte group (tid & 15) of every left word row.
		const int      rbw  = bw;
		const int      lbw  = d.lbw;
		const uint32_t lmsk = (1u << lbw) - 1u;
		const uint64_t dlo = dict.lo, dhi = dict.hi;
		(void)dlo, (void)dhi;
		const int      ls   = ((tid >> 4) * lbw) & 15;
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			const uint32_t f0  = static_cast<uint32_t>(w.l0 >> (16 * c)) & 0xFFFFu;
			const uint32_t f1  = static_cast<uint32_t>(w.l1 >> (16 * c)) & 0xFFFFu;
			const uint32_t idx = ((f0 >> ls) | (f1 << (16 - ls))) & lmsk;
#ifdef ALPGPU_F32_RD_DICT_IN_REGISTERS // A/B (round 6): the lookup as it was until round 5
			uint32_t       l   = static_cast<uint32_t>((idx < 4 ? dlo >> (16 * idx) : dhi >> (16 * (idx & 3))) & 0xFFFFull);
#else
			uint32_t       l   = L.rdict[idx & 7u];
#endif
			if (hits & (1u << c)) {
				l = fetch_exception_f32<2>(L, rec, rank, all_staged);
				++rank;
			}
			out[c] = (l << rbw) | q[c];
		}
	}
#if defined(ALPGPU_SINKF_STOP_AT) && ALPGPU_SINKF_STOP_AT == 6 // ... behind the exceptions, in front of the widening adds
	if constexpr (SINK != kSinkStoreF) {
		*acc += static_cast<double>(out[0] ^ out[1] ^ out[2] ^ out[3]);
		return;
	}
#endif
	if constexpr (SINK == kSinkSumF) {
#pragma unroll
		for (int c = 0; c < 4; ++c) { *acc += static_cast<double>(__uint_as_float(out[c])); }
	} else if constexpr (SINK == kSinkCountF) {
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			const float v = __uint_as_float(out[c]);
			*acc += (v >= range_lo && v <= range_hi) ? 1.0 : 0.0;
		}
	} else {
		store_quad<NT_STORE>(dst + 4 * tid, out);
		// (the two instances' stores must stay two stores: merged into one behind the join of their callers' branch, the wait for the rare load is back in front of it)
		if constexpr (HBM_EXC) {
			asm volatile("; quad stored (exception values beyond the stage possible)" ::: "memory");
		} else {
			asm volatile("; quad stored (exception values all staged)" ::: "memory");
		}
	}
}

// the sinks' second pass over an ALP vector with exceptions: lane L takes the values j = L, L + 64, ... in ascending order (staged ones from LDS, the rest from
// the stream) and adds them to `total` (SUM: widened; COUNT: those inside [lo, hi])
template <int SINK, class LDS>
__device__ __forceinline__ void sink_exception_values_f32(const LDS& L, const uint8_t* __restrict__ rec, int cnt, int lane, double& total, float range_lo, float range_hi) {
	for (int j = lane; j < cnt; j += 64) {
		uint32_t bits;
		if (j < kExcStageF) {
			bits = reinterpret_cast<const uint32_t*>(L.excv)[j];
		} else {
			bits = reinterpret_cast<const uint32_t*>(rec)[j];
		}
		const float v = __uint_as_float(bits);
		if constexpr (SINK == kSinkCountF) {
			total += (v >= range_lo && v <= range_hi) ? 1.0 : 0.0;
		} else {
			total += static_cast<double>(v);
		}
	}
}

// one vector of a staged workgroup, after its packed words / exception mask are visible in L
template <bool NT_STORE, int SINK = kSinkStoreF>
__device__ __forceinline__ void decode_staged_vector_f32(const DecodeLdsF32& L, const alpgpu_vector_desc& d, const RdDict& dict,
                                                         const uint8_t* __restrict__ rec, float* __restrict__ dst, int tid, int wave, int lane,
                                                         double* acc = nullptr, float range_lo = 0.0f, float range_hi = 0.0f) {
	ExcMaskF em {0u, 0};
	if (d.exc_cnt > 0) { em = load_exception_mask_f32(L, lane); }
	const QuadWords w = request_quad_f32(StagedWordsF {L.stage}, d, tid);
	if (SINK == kSinkStoreF && d.exc_cnt <= (d.scheme == ALPGPU_SCHEME_ALP ? kExcStageF : 2 * kExcStageF)) { // workgroup-uniform
		finish_quad_f32<NT_STORE, SINK, DecodeLdsF32, false>(L, w, d, dict, em, rec, dst, tid, wave, lane, acc, range_lo, range_hi);
	} else {
		finish_quad_f32<NT_STORE, SINK, DecodeLdsF32, true>(L, w, d, dict, em, rec, dst, tid, wave, lane, acc, range_lo, range_hi);
	}
}

// SINK != kSinkStoreF: `out` is the per-vector result array instead (double sums / uint32 counts)
template <int V, bool NT_STORE, int SINK = kSinkStoreF>
__global__ __launch_bounds__(kDecThreadsF) void k_decode_column_f32(const alpgpu_vector_desc* __restrict__ descs,
                                                                    const alpgpu_rowgroup_state* __restrict__ rgs, const uint8_t* __restrict__ packed,
                                                                    const uint8_t* __restrict__ excs, float* __restrict__ out, uint64_t n_vectors,
                                                                    uint64_t wg_offset, float range_lo, float range_hi, uint64_t* __restrict__ progress, uint64_t progress_tag,
                                                                    uint32_t gate) {
	__shared__ DecodeLdsF32 L[V];
	const int      tid  = static_cast<int>(threadIdx.x);
	const int      lane = tid & 63;
	const int      wave = wave_in_wg();
	const uint64_t v0   = (wg_offset + blockIdx.x) * V;
	if (v0 >= n_vectors) { return; }
	if constexpr (SINK == kSinkStoreF) {
		// a candidate launch of an unhinted decode runs only if the plan says so; the read-ahead's pace: every 128th workgroup says where the launch is
		// (decode_kernels.hip: k_decode_column; round 6: the float store decode reports too)
		if (gate != 0u && progress[kCtxWordShape] != static_cast<uint64_t>(gate)) { return; } // (kernel argument: uniform)
		if (progress != nullptr && (blockIdx.x & 127u) == 0 && tid == 0) { __hip_atomic_store(progress, progress_tag | v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
	}

#if defined(ALPGPU_F32_DISSECT) && ALPGPU_F32_DISSECT == 3 // measurement builds (profiles/r06_float_decode.txt): the launch as a plain fill
	if constexpr (SINK == kSinkStoreF) {
#pragma unroll
		for (int i = 0; i < V; ++i) {
			if (v0 + i < n_vectors) { store_quad<NT_STORE>(out + (v0 + i) * kVec + 4 * tid, u32x4 {1u, 2u, 3u, 4u}); }
		}
		return;
	}
#endif
	alpgpu_vector_desc d[V];
	uint32_t           pos[V];
#pragma unroll
	for (int i = 0; i < V; ++i) {
		const uint64_t v = v0 + i < n_vectors ? v0 + i : v0; // tail vectors of the last workgroup are simply loaded again
		d[i]             = descs[v];
	}
	RdDict dict[V];
#pragma unroll
	for (int i = 0; i < V; ++i) { dict[i] = load_vector_consts_f32(rgs, v0 + i < n_vectors ? v0 + i : v0, d[i]); }
#if defined(ALPGPU_F32_DISSECT) && ALPGPU_F32_DISSECT == 2 // ... a fill behind the descriptors' round trip
	if constexpr (SINK == kSinkStoreF) {
#pragma unroll
		for (int i = 0; i < V; ++i) {
			const uint32_t x = static_cast<uint32_t>(d[i].base) + d[i].bw + static_cast<uint32_t>(dict[i].lo ^ dict[i].hi);
			if (v0 + i < n_vectors) { store_quad<NT_STORE>(out + (v0 + i) * kVec + 4 * tid, u32x4 {x, x + tid, 3u, 4u}); }
		}
		return;
	}
#endif
#pragma unroll
	for (int i = 0; i < V; ++i) { pos[i] = issue_vector_loads_f32(L[i], d[i], packed, excs + d[i].exc_off, tid, wave); }
	// ALP_RD vectors: the dictionary into the vector's LDS (lanes 0..3 of wavefront 0; visible behind the barrier below)
#pragma unroll
	for (int i = 0; i < V; ++i) {
		if (d[i].scheme != ALPGPU_SCHEME_ALP && tid < 4) { reinterpret_cast<uint32_t*>(L[i].rdict)[tid] = static_cast<uint32_t>((tid < 2 ? dict[i].lo : dict[i].hi) >> (32 * (tid & 1))); }
	}
	// only a workgroup with exceptions zeroes its masks, behind the issue of its loads (decode_kernels.hip: k_decode_column)
	bool any_exc = false;
#pragma unroll
	for (int i = 0; i < V; ++i) { any_exc |= d[i].exc_cnt != 0; }
	if (any_exc) { // workgroup-uniform
		if (tid < 32 * V) { L[tid >> 5].mask[tid & 31] = 0; }
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // not __syncthreads(): its fence would wait for the loads in flight
#pragma unroll
		for (int i = 0; i < V; ++i) { land_exceptions_f32(L[i], d[i], excs + d[i].exc_off, pos[i], tid); }
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if constexpr (SINK != kSinkStoreF) {
		// per-vector result: thread partial over its quad (index order, in double) -> the four partials of a lane position combined as
		// (w0 + w1) + (w2 + w3) -> ONE adjacent-lane tree over the 64 results (decode_kernels.hip: the double sinks; the partials travel
		// through the vectors' spent stages); tests/test_decode_sum_gpu.py replays this order on the host
		double acc[V];
#pragma unroll
		for (int i = 0; i < V; ++i) {
			acc[i] = 0.0;
			if (v0 + i < n_vectors) {
				decode_staged_vector_f32<NT_STORE, SINK>(L[i], d[i], dict[i], excs + d[i].exc_off, nullptr, tid, wave, lane, &acc[i], range_lo,
				                                         range_hi);
			}
		}
		__syncthreads(); // nobody reads packed words any more
		static_assert(kStageBytesF >= 8 * kDecThreadsF && kDecThreadsF == 256, "the lane partials of a vector fit its stage; four wavefronts per vector");
#pragma unroll
		for (int i = 0; i < V; ++i) { reinterpret_cast<double*>(L[i].stage)[tid] = acc[i]; }
		__syncthreads();
		if (wave < V && v0 + wave < n_vectors) { // wave-uniform: wavefront w finishes vector w (V <= 4)
			const double* part  = reinterpret_cast<const double*>(L[wave].stage) + lane;
			double        total = (part[0] + part[64]) + (part[128] + part[192]);
#pragma unroll
			for (int i = 0; i < V; ++i) {
				if (wave == i && d[i].exc_cnt > 0 && d[i].scheme == ALPGPU_SCHEME_ALP) { sink_exception_values_f32<SINK>(L[i], excs + d[i].exc_off, d[i].exc_cnt, lane, total, range_lo, range_hi); }
			}
			total               = wave_tree_sum_f64(total);
			if (lane == 0) {
				if constexpr (SINK == kSinkCountF) {
					reinterpret_cast<uint32_t*>(out)[v0 + wave] = static_cast<uint32_t>(total);
				} else {
					reinterpret_cast<double*>(out)[v0 + wave] = total;
				}
			}
		}
		return;
	}
#if defined(ALPGPU_F32_DISSECT) && ALPGPU_F32_DISSECT == 1 // ... the staged words stored as they are (no unpack, no exceptions)
#pragma unroll
	for (int i = 0; i < V; ++i) {
		if (v0 + i < n_vectors) { store_quad<NT_STORE>(out + (v0 + i) * kVec + 4 * tid, reinterpret_cast<const u32x4*>(L[i].stage)[tid & 31]); }
	}
	return;
#endif
#pragma unroll
	for (int i = 0; i < V; ++i) {
		if (v0 + i < n_vectors) {
			decode_staged_vector_f32<NT_STORE>(L[i], d[i], dict[i], excs + d[i].exc_off, out + (v0 + i) * kVec, tid, wave, lane);
		}
	}
}

template <int V>
static void launch_v(hipStream_t stream, const alpgpu_column* col, float* d_out, bool nt, int pad_kib, uint64_t* progress, uint64_t tag, uint32_t gate) {
	const uint64_t n        = col->n_vectors;
	const uint64_t n_wg     = (n + V - 1) / V;
	const uint64_t kMaxGrid = 1ull << 30;
	// unused dynamic LDS that caps the workgroups resident per CU, as the double decode does by width (pad_kib < 0: ALPGPU_DECODE_F32_PAD_LDS_KIB, for experiments, else none)
	static const int env_pad = std::getenv("ALPGPU_DECODE_F32_PAD_LDS_KIB") ? std::atoi(std::getenv("ALPGPU_DECODE_F32_PAD_LDS_KIB")) : 0;
	const unsigned   pad_lds = static_cast<unsigned>(pad_kib >= 0 ? pad_kib : env_pad) * 1024u;
	if (gate != 0 && progress == nullptr) { gate = 0; }
	for (uint64_t off = 0; off < n_wg; off += kMaxGrid) {
		const dim3 grid(static_cast<unsigned>(n_wg - off < kMaxGrid ? n_wg - off : kMaxGrid)), block(kDecThreadsF);
		if (nt) {
			hipLaunchKernelGGL((k_decode_column_f32<V, true>), grid, block, pad_lds, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, d_out, n, off, 0.0f, 0.0f, progress, tag, gate);
		} else {
			hipLaunchKernelGGL((k_decode_column_f32<V, false>), grid, block, pad_lds, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, d_out, n, off, 0.0f, 0.0f, progress, tag, gate);
		}
	}
}

static int launch_store_direct_f32(hipStream_t stream, const alpgpu_column* col, float* d_out, int pad_kib, uint64_t* progress, uint64_t progress_tag); // (below, behind its kernel)
// vectors_per_wg in {1, 2, 4}; 8 = one wavefront per vector
int launch_decode_column_f32(hipStream_t stream, const alpgpu_column* col, float* d_out, int vectors_per_wg, bool plain_stores, int pad_kib, uint64_t* progress, uint64_t progress_tag,
                             uint32_t gate) {
	if (col->n_vectors == 0) { return ALPGPU_OK; }
	if (vectors_per_wg == 8) { return launch_store_direct_f32(stream, col, d_out, pad_kib, progress, progress_tag); } // (no gate: never a candidate of an unhinted decode)
	if (vectors_per_wg == 1) {
		launch_v<1>(stream, col, d_out, !plain_stores, pad_kib, progress, progress_tag, gate);
	} else if (vectors_per_wg == 2) {
		launch_v<2>(stream, col, d_out, !plain_stores, pad_kib, progress, progress_tag, gate);
	} else {
		launch_v<4>(stream, col, d_out, !plain_stores, pad_kib, progress, progress_tag, gate);
	}
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

template <int SINK, int V>
static int launch_sink_f32(hipStream_t stream, const alpgpu_column* col, void* d_result, float lo, float hi) {
	const uint64_t n = col->n_vectors;
	if (n == 0) { return ALPGPU_OK; }
	const uint64_t n_wg     = (n + V - 1) / V;
	const uint64_t kMaxGrid = 1ull << 30;
	for (uint64_t off = 0; off < n_wg; off += kMaxGrid) {
		const dim3 grid(static_cast<unsigned>(n_wg - off < kMaxGrid ? n_wg - off : kMaxGrid)), block(kDecThreadsF);
		hipLaunchKernelGGL((k_decode_column_f32<V, false, SINK>), grid, block, 0, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc,
		                   static_cast<float*>(d_result), n, off, lo, hi, static_cast<uint64_t*>(nullptr), 0ull, 0u);
	}
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

// ---- the float sinks with ONE wavefront per vector, packed words straight from HBM (decode_kernels.hip: k_sink_direct) ----------------
// Lane L does what threads L, 64 + L, 128 + L, 192 + L of the staged kernel do: four quads, their partials kept apart (p[w][L]), all four
// quads' words requested before the first is used; then (p0 + p1) + (p2 + p3) and the adjacent-lane tree: the same bits.
#ifndef ALPGPU_SINK_STAGE_F32
#define ALPGPU_SINK_STAGE_F32 3584 // bytes of packed words (bit widths <= 28) a wavefront of k_sink_direct_f32 stages in its LDS by LDS-DMA (0: none), as k_sink_direct does
#endif
#ifndef ALPGPU_SINK_STAGE_F32_MAX_EXC
#define ALPGPU_SINK_STAGE_F32_MAX_EXC 1024 // ... only for vectors with at most this many exceptions (round 6: any — it was 48 while the exception values were looked up per quad; with them taken behind the quads the stage pays at every count: +1.5 %, call 26)
#endif
struct __attribute__((aligned(16))) SinkWaveLdsF32 {
	static constexpr bool kPrefixInLds = true;
#if ALPGPU_SINK_STAGE_F32 > 0
	uint8_t  stage[ALPGPU_SINK_STAGE_F32 + 128]; // + the unit row past the end that the unpack reads and masks off
#endif
	uint32_t mask[32];
	uint8_t  excv[4 * kExcStageF];
	uint32_t pref[32]; // exceptions in front of mask word i
	uint16_t rdict[8]; // ALP_RD: the rowgroup's dictionary (DecodeLdsF32)
};
template <int SINK>
__global__ __launch_bounds__(kDecThreadsF, 8) void k_sink_direct_f32(const alpgpu_vector_desc* __restrict__ descs, const alpgpu_rowgroup_state* __restrict__ rgs,
                                                                      const uint8_t* __restrict__ packed, const uint8_t* __restrict__ excs, double* __restrict__ out,
                                                                      uint64_t n_vectors, uint64_t wg_offset, float lo, float hi, uint64_t* __restrict__ progress, uint64_t progress_tag) {
	#ifdef ALPGPU_SINK_REGISTER_MARGIN
	asm volatile("" ::: "v64"); // decode_kernels.hip: k_sink_direct
#endif
	__shared__ SinkWaveLdsF32 S[kDecThreadsF / 64];
	const int      lane = static_cast<int>(threadIdx.x) & 63;
	const int      wv   = wave_in_wg();
	const uint64_t v    = (wg_offset + blockIdx.x) * (kDecThreadsF / 64) + wv;
	if (v >= n_vectors) { return; } // wave-uniform; no barrier anywhere in this kernel
	if constexpr (SINK == kSinkStoreF) { // the store form reports where the launch is, for the read-ahead (k_decode_column_f32 does the same): every 32nd workgroup = every 128th vector
		if (progress != nullptr && (blockIdx.x & 31u) == 0 && threadIdx.x == 0) { __hip_atomic_store(progress, progress_tag | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
	}
	SinkWaveLdsF32&          L      = S[wv];
	const alpgpu_vector_desc d      = descs[v];
	const RdDict             dict   = load_vector_consts_f32(rgs, v, d);
	const uint8_t*           rec    = excs + d.exc_off;
	const bool               is_alp = d.scheme == ALPGPU_SCHEME_ALP;
	const int                cnt    = d.exc_cnt;
	ExcMaskF                 em {0u, 0};
#if defined(ALPGPU_SINKF_STOP_AT) && ALPGPU_SINKF_STOP_AT == 1 // measurement builds: every wavefront ends behind stage n with one dependent store
	if (lane == 0) { out[v] = static_cast<double>(d.bw + cnt) + static_cast<double>(dict.lo ^ dict.hi); }
	return;
#endif
#if ALPGPU_SINK_STAGE_F32 > 0
	// Round 4: a narrow ALP vector's words whole into the wavefront's LDS by LDS-DMA (1 KiB per instruction, no registers): ONE round trip for all
	// of them instead of eight 16-byte buffer loads per lane — what took the double sink from 0.85 to 0.73 ms (profiles/r03_consumers.txt) and what
	// the float one never had ("is the stage arm even taken for float?" — there was none).
	// (late round 4: ALP_RD vectors too — right words, then left words, as they lie in the stream; -DALPGPU_SINK_STAGE_F32_ALP_ONLY: the first form)
#ifdef ALPGPU_SINK_STAGE_F32_ALP_ONLY
	const int  stage_words = is_alp ? static_cast<int>(d.bw) : 1000;
#else
	const int  stage_words = static_cast<int>(d.bw) + (is_alp ? 0 : static_cast<int>(d.lbw));
#endif
	const bool staged = 128 * stage_words <= ALPGPU_SINK_STAGE_F32 && cnt <= ALPGPU_SINK_STAGE_F32_MAX_EXC; // wave-uniform
	if (staged) {
		typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
		const ull2* g       = reinterpret_cast<const ull2*>(packed + d.packed_off);
		const int   n_units = 8 * stage_words;
		for (int j = 0; 64 * j < n_units; ++j) {
			if (64 * j + lane < n_units) { __builtin_amdgcn_global_load_lds(g + 64 * j + lane, reinterpret_cast<ull2*>(L.stage) + 64 * j, 16, 0, 0); }
		}
	}
#endif
	if (cnt > 0) { // wave-uniform: values of the first kExcStageF exceptions by LDS-DMA, the mask from the positions
		const uint32_t val_bytes = (is_alp ? 4u : 2u) * static_cast<uint32_t>(cnt);
		const int      dwords    = static_cast<int>(((val_bytes < 4u * kExcStageF ? val_bytes : 4u * kExcStageF) + 3u) >> 2);
		for (int q = 0; 64 * q < dwords; ++q) {
			if (64 * q + lane < dwords) { __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(rec) + 64 * q + lane, reinterpret_cast<uint32_t*>(L.excv) + 64 * q, 4, 0, 0); }
		}
		if (lane < 32) { L.mask[lane] = 0u; }
		wave_lds_sync();
		const uint16_t* poss = reinterpret_cast<const uint16_t*>(rec + val_bytes);
		for (int j = lane; j < cnt; j += 64) {
			const uint32_t p = poss[j];
			atomicOr(&L.mask[p >> 5], 1u << (p & 31u));
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the DMA'd values
		wave_lds_sync();
		em = load_exception_mask_f32(L, lane);
		if (lane < 32) { L.pref[lane] = static_cast<uint32_t>(em.excl); }
		wave_lds_sync();
	}
	uint8_t*           first      = const_cast<uint8_t*>(packed + d.packed_off);
	constexpr int      kRsrcFlags = 0x00020000; // gfx9 raw buffer, 32-bit data format
	const BufferWordsF words {__builtin_amdgcn_make_buffer_rsrc(first, 0, 128 * d.bw, kRsrcFlags),
	                          __builtin_amdgcn_make_buffer_rsrc(first + 128u * d.bw, 0, is_alp ? 0 : 128 * d.lbw, kRsrcFlags)};
#if defined(ALPGPU_SINKF_STOP_AT) && ALPGPU_SINKF_STOP_AT == 2 // ... behind the issue of the loads and the exception mask
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	if (lane == 0) { out[v] = static_cast<double>(em.excl + static_cast<int>(em.word)); }
	return;
#endif
	if (!is_alp) { // wave-uniform: the dictionary into the wavefront's LDS, behind the issue of every load (its own scalar loads are waited for here, not in front of them)
		if (lane < 4) { reinterpret_cast<uint32_t*>(L.rdict)[lane] = static_cast<uint32_t>((lane < 2 ? dict.lo : dict.hi) >> (32 * (lane & 1))); }
		wave_lds_sync();
	}
	QuadWords w[4];
#if ALPGPU_SINK_STAGE_F32 > 0
	if (staged) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // LDS-DMA completion is not tracked through the LDS for the compiler
		wave_lds_sync();
		const StagedWordsF sw {L.stage};
#pragma unroll
		for (int q = 0; q < 4; ++q) { w[q] = request_quad_f32(sw, d, 64 * q + lane); }
	} else
#endif
	{
#pragma unroll
		for (int q = 0; q < 4; ++q) { w[q] = request_quad_f32(words, d, 64 * q + lane); }
	}
#if defined(ALPGPU_SINKF_STOP_AT) && ALPGPU_SINKF_STOP_AT == 3 // ... behind the arrival of the quads' words
	{
		uint32_t x = 0;
#pragma unroll
		for (int q = 0; q < 4; ++q) { x ^= w[q].w0[0] ^ w[q].w0[1] ^ w[q].w0[2] ^ w[q].w0[3] ^ w[q].w1[0] ^ w[q].w1[1] ^ w[q].w1[2] ^ w[q].w1[3] ^ static_cast<uint32_t>(w[q].l0 ^ w[q].l1); }
		out[v * 64 % n_vectors] = static_cast<double>(x);
		return;
	}
#endif
	if constexpr (SINK == kSinkStoreF) {
		// Round 6: the same wavefront as a STORE decoder (ALPGPU_OPT_DECODE_VECTORS_PER_WG = 8 for float columns): lane L's four quads are the 16-byte units L, 64 + L,
		// 128 + L, 192 + L of the vector — four store instructions of 1 KiB each, no barrier, one wave-uniform prologue per vector instead of four.
		float* dst = reinterpret_cast<float*>(out) + v * kVec;
		if (cnt <= (is_alp ? kExcStageF : 2 * kExcStageF)) { // wave-uniform: every exception value is staged — the instance without a load between the stores
#pragma unroll
			for (int q = 0; q < 4; ++q) { finish_quad_f32<true, kSinkStoreF, SinkWaveLdsF32, false>(L, w[q], d, dict, em, rec, dst, 64 * q + lane, q, lane, nullptr, 0.0f, 0.0f); }
		} else {
#pragma unroll
			for (int q = 0; q < 4; ++q) { finish_quad_f32<true, kSinkStoreF, SinkWaveLdsF32, true>(L, w[q], d, dict, em, rec, dst, 64 * q + lane, q, lane, nullptr, 0.0f, 0.0f); }
		}
		return;
	}
	double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
	for (int q = 0; q < 4; ++q) { finish_quad_f32<false, SINK>(L, w[q], d, dict, em, rec, nullptr, 64 * q + lane, q, lane, &part[q], lo, hi); }
	double total = (part[0] + part[1]) + (part[2] + part[3]);
	if (is_alp && cnt > 0) { sink_exception_values_f32<SINK>(L, rec, cnt, lane, total, lo, hi); }
	total        = wave_tree_sum_f64(total);
	if (lane == 0) {
		if constexpr (SINK == kSinkCountF) {
			reinterpret_cast<uint32_t*>(out)[v] = static_cast<uint32_t>(total);
		} else {
			out[v] = total;
		}
	}
}

// one wavefront per vector, four vectors per workgroup, no barrier: k_sink_direct_f32 as a store decoder (non-temporal stores)
static int launch_store_direct_f32(hipStream_t stream, const alpgpu_column* col, float* d_out, int pad_kib, uint64_t* progress, uint64_t progress_tag) {
	const uint64_t n = col->n_vectors, n_wg = (n + 3) / 4, kMaxGrid = 1ull << 30;
	for (uint64_t off = 0; off < n_wg; off += kMaxGrid) {
		hipLaunchKernelGGL((k_sink_direct_f32<kSinkStoreF>), dim3(static_cast<unsigned>(n_wg - off < kMaxGrid ? n_wg - off : kMaxGrid)), dim3(kDecThreadsF), pad_kib > 0 ? static_cast<unsigned>(pad_kib) * 1024u : 0u, stream,
		                   col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, reinterpret_cast<double*>(d_out), n, off, 0.0f, 0.0f, progress, progress_tag);
	}
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

int launch_sink_direct_f32(hipStream_t stream, const alpgpu_column* col, float lo, float hi, void* d_out, bool count) {
	const uint64_t n = col->n_vectors;
	if (n == 0) { return ALPGPU_OK; }
	const uint64_t per_wg   = kDecThreadsF / 64;
	const uint64_t n_wg     = (n + per_wg - 1) / per_wg;
	const uint64_t kMaxGrid = 1ull << 30;
	for (uint64_t off = 0; off < n_wg; off += kMaxGrid) {
		const dim3 grid(static_cast<unsigned>(n_wg - off < kMaxGrid ? n_wg - off : kMaxGrid)), block(kDecThreadsF);
		if (count) {
			hipLaunchKernelGGL((k_sink_direct_f32<kSinkCountF>), grid, block, 0, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, static_cast<double*>(d_out), n, off, lo, hi, static_cast<uint64_t*>(nullptr), 0ull);
		} else {
			hipLaunchKernelGGL((k_sink_direct_f32<kSinkSumF>), grid, block, 0, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, static_cast<double*>(d_out), n, off, 0.0f, 0.0f, static_cast<uint64_t*>(nullptr), 0ull);
		}
	}
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

// two vectors per workgroup: measured 1.57 / 1.30 / 1.37 ms per 1 Mi vectors for 1 / 2 / 4
int launch_decode_sum_f32(hipStream_t stream, const alpgpu_column* col, double* d_sums) { return launch_sink_f32<kSinkSumF, 2>(stream, col, d_sums, 0.0f, 0.0f); }

int launch_decode_count_range_f32(hipStream_t stream, const alpgpu_column* col, float lo, float hi, uint32_t* d_counts) {
	return launch_sink_f32<kSinkCountF, 2>(stream, col, d_counts, lo, hi);
}

} // namespace alpgpu
