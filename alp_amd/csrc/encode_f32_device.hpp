// encode_f32_device.hpp — wavefront-level building blocks of the single-precision ALP / ALP_RD vector encode (gfx950).
//
// One 64-lane wavefront per 1024-value vector; lane L holds the value quads i = 256*m + 4*L + j (m = 0..3, j = 0..3), so the
// 4 KiB input is read with four 1-KiB-contiguous 16-byte-per-lane loads; in the FastLanes u32 layout (alp_device_f32.hpp) a
// quad is one 16-byte unit: row = 8*m + (L >> 3), unit column a = L & 7.
// Reference functions restated here (file:line relative to /root/reference), PT = float:
//   second-level sampling  alp::encoder<float>::find_best_exponent_factor_from_combinations  include/alp/encoder.hpp:241-305
//   value encode + verify  alp::encoder<float>::encode_simdized                              include/alp/encoder.hpp:307-400
//   FOR analysis           alp::encoder<float>::analyze_ffor                                 include/alp/encoder.hpp:109-120
//   FFOR bit-packing       ffor::ffor (u32: src/fastlanes_generated_ffor.cpp:1776-7378)
#pragma once
#include "alp_device_f32.hpp"

namespace alpgpu {

struct __attribute__((aligned(16))) EncodeLdsF32 {
	uint32_t vals[kVec]; // (enc - base) or RD right parts, natural index order: 4 KiB
	float    smp[32];    // second-level samples
};

struct VecInF {
	f32x4 x[4];
};

// a wave-uniform lane mask (ballot) as this lane's predicate
__device__ __forceinline__ bool lane_in_f32(uint64_t ballot) { return __builtin_amdgcn_inverse_ballot_w64(ballot); }

__device__ __forceinline__ uint64_t lanemask_lt64(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }

__device__ __forceinline__ VecInF load_vector_f32(const float* __restrict__ in, uint64_t v, int lane) {
	const f32x4* p = reinterpret_cast<const f32x4*>(in + v * kVec);
	VecInF       r;
#pragma unroll
#ifdef ALPGPU_F32_PLAIN_POLICY // (A/B: plain loads and stores, as until late in round 4)
	for (int m = 0; m < 4; ++m) { r.x[m] = p[64 * m + lane]; }
#else
	for (int m = 0; m < 4; ++m) { r.x[m] = __builtin_nontemporal_load(p + 64 * m + lane); } // read once (the two-pass form's second read misses the caches of a long column anyway)
#endif
	return r;
}

// ---- second-level sampling (encoder.hpp:241-305), PT = float: exception cost 32 + 16 bits -----------------------
// sample s = input[32*s] = element j = 0 of step s >> 3 in lane 8*(s & 7).  All candidates are evaluated (two per round,
// one per half-wave); replaying the reference's sequential decision over the sizes gives its (e,f), early exit included.
__device__ __forceinline__ void second_level_select_f32(const VecInF& in, const alpgpu_rowgroup_state* __restrict__ rgp, EncodeLdsF32& L, int lane,
                                                        int& e_out, int& f_out) {
	const int k = rgp->k;
	if ((lane & 7) == 0) {
#pragma unroll
		for (int m = 0; m < 4; ++m) { L.smp[8 * m + (lane >> 3)] = in.x[m][0]; }
	}
	wave_lds_sync();
	const float sv   = L.smp[lane & 31];
	const int   half = lane >> 5;
	uint32_t    sizes[5];
#pragma unroll
	for (int i = 0; i < 5; ++i) { sizes[i] = 0xFFFFFFFFu; }
#pragma unroll
	for (int kk = 0; kk < 6; kk += 2) {
		if (kk < k) { // wave-uniform
			// candidate kk for the low half-wave, kk + 1 for the high one (candidate 0 again if there is none); rgp may be a
			// register copy of the state: constant indices only
			const int      kHi   = kk + 1 < 5 ? kk + 1 : 0; // a constant once the loop is unrolled
			const bool     hi_ok = half != 0 && kk + 1 < k;
			const int      e     = hi_ok ? rgp->combos[2 * kHi] : (half != 0 ? rgp->combos[0] : rgp->combos[2 * kk]);
			const int      f     = hi_ok ? rgp->combos[2 * kHi + 1] : (half != 0 ? rgp->combos[1] : rgp->combos[2 * kk + 1]);
			const int32_t  enc   = encode_value_f32(sv, kExpArrF[e], kFracArrF[f]);
			const float    dec   = decode_value_f32(enc, kFactArrF[f], kFracArrF[e]);
			const bool     ok    = dec == sv;
			const uint64_t bal   = ballot64(!ok);
			const uint32_t excs  = __builtin_popcount(static_cast<uint32_t>(half ? (bal >> 32) : bal));
			int32_t        mx    = ok ? enc : INT32_MIN;
			int32_t        mn    = ok ? enc : INT32_MAX;
			// min / max inside each 32-lane half by DPP (register to register): lane 31 / 63 end up with their half's result
#define ALPGPU_MINMAX_STEP(CTRL, ROWS)                                                                                  \
	{                                                                                                                   \
		const int32_t omx = __builtin_amdgcn_update_dpp(mx, mx, CTRL, ROWS, 0xf, false);                                \
		const int32_t omn = __builtin_amdgcn_update_dpp(mn, mn, CTRL, ROWS, 0xf, false);                                \
		mx                = omx > mx ? omx : mx;                                                                        \
		mn                = omn < mn ? omn : mn;                                                                        \
	}
			ALPGPU_MINMAX_STEP(0x111, 0xf) // row_shr:1
			ALPGPU_MINMAX_STEP(0x112, 0xf) // row_shr:2
			ALPGPU_MINMAX_STEP(0x114, 0xf) // row_shr:4
			ALPGPU_MINMAX_STEP(0x118, 0xf) // row_shr:8
			ALPGPU_MINMAX_STEP(0x142, 0xa) // row_bcast:15 into rows 1 and 3
#undef ALPGPU_MINMAX_STEP
			const uint32_t size = 32u * static_cast<uint32_t>(count_bits32(mx, mn)) + excs * 48u;
			sizes[kk]           = __builtin_amdgcn_readlane(size, 31);
			if (kk + 1 < 5) { sizes[kk + 1] = __builtin_amdgcn_readlane(size, 63); }
		}
	}
	int      best = 0, worse = 0;
	uint32_t best_size = sizes[0];
	bool     stopped   = false;
#pragma unroll
	for (int i = 1; i < 5; ++i) { // encoder.hpp:283-301
		if (i < k && !stopped) {
			if (sizes[i] >= best_size) {
				if (++worse == 2) { stopped = true; }
			} else {
				best_size = sizes[i];
				best      = i;
				worse     = 0;
			}
		}
	}
	e_out = rgp->combos[0];
	f_out = rgp->combos[1];
#pragma unroll
	for (int i = 1; i < 5; ++i) {
		e_out = best == i ? rgp->combos[2 * i] : e_out;
		f_out = best == i ? rgp->combos[2 * i + 1] : f_out;
	}
	wave_lds_sync();
}

// ---- encode_simdized + analyze_ffor for one vector held in registers --------------------------------------------------
struct AlpEncodedF {
	int32_t  enc[4][4];
	uint64_t ballot[4][4]; // exception slots of every (m, j) step as wave-uniform lane masks
	int      cnt;
	int32_t  base;
	int      bw;
};

__device__ __forceinline__ void encode_alp_registers_f32(const VecInF& in, int e, int f, int lane, AlpEncodedF& R) {
	(void)lane; // every cross-lane step below is a ballot, a DPP move or a readlane
	const float    exp10  = kExpArrF[e];
	const float    frac_f = kFracArrF[f];
	const uint32_t fact   = kFactArrF[f];
	const float    frac_e = kFracArrF[e];
	R.cnt   = 0;
#ifndef ALPGPU_F32_SHORTCUT_ANALYSIS // the default: every value through the literal conversions
#pragma unroll
	for (int m = 0; m < 4; ++m) {
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const float    v    = in.x[m][j];
			const uint32_t bits = __float_as_uint(v);
			// pass 1 (encoder.hpp:326-338): NaN, +-Inf and -0.0 are replaced by (float)ENCODING_UPPER_LIMIT
			const bool    special = ((bits & 0x7FFFFFFFu) >= 0x7F800000u) || bits == 0x80000000u;
			const float   vv      = special ? kUpperLimitF : v;
			const int32_t enc     = encode_value_f32(vv, exp10, frac_f);
			const float   dec     = decode_value_f32(enc, fact, frac_e);
			const bool    exc     = dec != vv;
			R.enc[m][j]           = enc;
			R.ballot[m][j]        = ballot64(exc);
			R.cnt += __builtin_popcountll(R.ballot[m][j]);
		}
	}
#else
	// -DALPGPU_F32_SHORTCUT_ANALYSIS (round 4, measured, NOT the default: bit-identical — 130 float / fuzz / reference tests — but 1093 instead of 1037 vector
	// instructions per vector and 2.45 instead of 2.43 ms per 1 Mi vectors: its twelve extra lane masks per value step push the kernel's
	// scalar registers over the edge, and every spilled mask comes back as v_readlane pairs — profiles/r04_float_encode.txt):
	// the double kernel's shortcut arithmetic (encode_device.hpp: encode_alp_registers), restated for float, on value PAIRS so that the
	// six roundings per value issue as packed instructions (v_pk_mul_f32 / v_pk_add_f32: two values per issue slot; -ffp-contract=off keeps every
	// one of them its own IEEE rounding).  The kernel runs eight wavefronts per SIMD with its VALU 85 % busy (profiles/r04_float_encode.txt): what
	// it executes per value is what it costs.  With t = (v * 10^e) * 10^-f, u = t + M, r = u - M (M = 2^23 + 2^22):
	//   * |t| < 2^22: u lies in [2^23, 2^24), where consecutive floats are consecutive integers: the encoded integer (int32)r is bits(u) - bits(M);
	//   * |r * 10^f| < 2^31: the int32 product does not wrap, and (float)(int32)(enc * 10^f) is the correctly rounded value of that integer — the
	//     IEEE product r * 10^f of two exactly representable floats (10^f = 2^f 5^f, 5^10 < 2^24);
	//   * |r * 10^f| > 2^31 (as a float: the exact product is then > 2^31 + 128): the product wraps.  Up to 2^32 it comes back with the opposite
	//     sign of v and not zero, beyond that with a magnitude below 2^31 <= half of |P| while v = P 10^-e (1 +- 2^-22): never equal — an exception
	//     without computing it;
	//   * NaN, +-Inf, -0.0 (pass 1 replaces them by 2^63, encoder.hpp:326-338): the cast of the huge t gives INT32_MIN, whose product with 10^f
	//     is 0 modulo 2^32 for f >= 1 and -2^31 for f = 0: decoded 0 or -2^31 10^-e, never 2^63 — always exceptions;
	//   * everything else (|t| >= 2^22, or |r * 10^f| == 2^31 exactly) redoes the value step literally, for the whole wavefront.
	// Bit-identical by construction; tests/test_float_gpu.py, test_fuzz_gpu.py and test_reference_gpu.py compare every stream byte.
	typedef float f32x2 __attribute__((ext_vector_type(2)));
	const float fact_f = static_cast<float>(fact); // exact (f <= 9; f = 10: fact is 10^10 mod 2^32, and then every non-zero integer is an exception on either route: see below)
#pragma unroll
	for (int m = 0; m < 4; ++m) {
		f32x2    vv[2], rr[2], dec[2], uu[2];
		uint64_t over_m[4], wide_m[4], sp_m[4];
		uint64_t any_wide = 0;
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const f32x2 v2 = {in.x[m][2 * h], in.x[m][2 * h + 1]};
			const f32x2 t2 = (v2 * exp10) * frac_f;
			const f32x2 u2 = t2 + kMagicF;
			const f32x2 r2 = u2 - kMagicF;
			const f32x2 p2 = r2 * fact_f;
			vv[h] = v2, rr[h] = r2, uu[h] = u2;
			dec[h] = p2 * frac_e;
#pragma unroll
			for (int c = 0; c < 2; ++c) {
				const int   j  = 2 * h + c;
				const float ap = __builtin_fabsf(p2[c]);
				sp_m[j]        = ballot64(__builtin_amdgcn_classf(v2[c], 0x227)); // sNaN, qNaN, -Inf, -0.0, +Inf
				over_m[j]      = ballot64(ap > 0x1p31f);
				wide_m[j]      = (ballot64(!(__builtin_fabsf(t2[c]) < 0x1p22f)) | (ballot64(!(ap < 0x1p31f)) & ~over_m[j])) & ~sp_m[j];
				any_wide |= wide_m[j];
			}
		}
		int32_t enc[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) { enc[j] = static_cast<int32_t>(__float_as_uint(uu[j >> 1][j & 1]) - 0x4B400000u); }
		float decs[4] = {dec[0][0], dec[0][1], dec[1][0], dec[1][1]};
		if (__builtin_expect(any_wide != 0, 0)) { // wave-uniform, rare: the literal conversions for the value steps that need them
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				if (wide_m[j] != 0) {
					enc[j]  = cast32_x86(rr[j >> 1][j & 1]);
					decs[j] = decode_value_f32(enc[j], fact, frac_e);
				}
			}
		}
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const uint64_t exc_m = ballot64(decs[j] != vv[j >> 1][j & 1]) | (over_m[j] & ~wide_m[j]) | sp_m[j];
			R.enc[m][j]          = enc[j];
			R.ballot[m][j]       = exc_m;
			R.cnt += __builtin_popcountll(exc_m);
		}
	}
#endif
	// filler = encoded value at the first non-exception position p (encoder.hpp:382-388); 0 when there is none or p == 1023
	int32_t filler = 0;
	bool    found  = false;
#pragma unroll
	for (int m = 0; m < 4; ++m) {
		const uint64_t any = ~(R.ballot[m][0] & R.ballot[m][1] & R.ballot[m][2] & R.ballot[m][3]);
		if (!found && any != 0) { // wave-uniform
			int best_pos = 1 << 20, best_j = 0, best_l = 0;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const uint64_t nj = ~R.ballot[m][j];
				const int      lj = nj ? __builtin_ctzll(nj) : 64;
				const int      pj = nj ? 4 * lj + j : (1 << 20);
				if (pj < best_pos) {
					best_pos = pj;
					best_j   = j;
					best_l   = lj;
				}
			}
			int32_t cand = 0;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				if (best_j == j) { cand = __builtin_amdgcn_readlane(R.enc[m][j], best_l); }
			}
			filler = (256 * m + best_pos == 1023) ? 0 : cand;
			found  = true;
		}
	}
	int32_t mn = INT32_MAX, mx = INT32_MIN;
#pragma unroll
	for (int m = 0; m < 4; ++m) {
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			R.enc[m][j] = lane_in_f32(R.ballot[m][j]) ? filler : R.enc[m][j];
			mn = R.enc[m][j] < mn ? R.enc[m][j] : mn;
			mx = R.enc[m][j] > mx ? R.enc[m][j] : mx;
		}
	}
	// wavefront min / max by DPP (register to register; lanes without a source lane keep their own value), result in lane 63
#define ALPGPU_MINMAX_STEP(CTRL, ROWS)                                                                                  \
	{                                                                                                                   \
		const int32_t omn = __builtin_amdgcn_update_dpp(mn, mn, CTRL, ROWS, 0xf, false);                                \
		const int32_t omx = __builtin_amdgcn_update_dpp(mx, mx, CTRL, ROWS, 0xf, false);                                \
		mn                = omn < mn ? omn : mn;                                                                        \
		mx                = omx > mx ? omx : mx;                                                                        \
	}
	ALPGPU_MINMAX_STEP(0x111, 0xf) // row_shr:1
	ALPGPU_MINMAX_STEP(0x112, 0xf) // row_shr:2
	ALPGPU_MINMAX_STEP(0x114, 0xf) // row_shr:4
	ALPGPU_MINMAX_STEP(0x118, 0xf) // row_shr:8
	ALPGPU_MINMAX_STEP(0x142, 0xa) // row_bcast:15 into rows 1 and 3
	ALPGPU_MINMAX_STEP(0x143, 0xc) // row_bcast:31 into rows 2 and 3
#undef ALPGPU_MINMAX_STEP
	mn     = __builtin_amdgcn_readlane(mn, 63);
	mx     = __builtin_amdgcn_readlane(mx, 63);
	R.base = mn;
	R.bw   = count_bits32(mx, mn);
}

// Calls emit(rank, m, j) on every lane whose (m, j) slot is set in `ballot`; rank = number of set slots at smaller positions
// (position = 256*m + 4*lane + j).  Steps without exceptions are skipped wave-uniformly; m and j are compile-time at the call.
template <class F>
__device__ __forceinline__ void for_each_exception_f32(const uint64_t (&ballot)[4][4], int lane, F&& emit) {
	const uint64_t lt   = lanemask_lt64(lane);
	int            soff = 0;
#pragma unroll
	for (int m = 0; m < 4; ++m) {
		const uint64_t any = ballot[m][0] | ballot[m][1] | ballot[m][2] | ballot[m][3];
		if (any != 0) {
			int rank = soff;
#pragma unroll
			for (int j = 0; j < 4; ++j) { rank += __builtin_popcountll(ballot[m][j] & lt); }
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				if (lane_in_f32(ballot[m][j])) {
					emit(rank, m, j);
					++rank;
				}
			}
#pragma unroll
			for (int j = 0; j < 4; ++j) { soff += __builtin_popcountll(ballot[m][j]); }
		}
	}
}

// ---- the same analysis with NO lane mask outliving its value step (late round 4) -----------------------------------------------------------
// At eight wavefronts per SIMD a wavefront may hold 80 scalar registers (800 per SIMD, 16 of each wavefront's share reserved for the trap
// handler); the sixteen exception masks of encode_alp_registers_f32 are 32 of them, live from the first value step to the exception record, and
// the single-pass kernel spilled 89 scalar registers into VGPR lanes — every spill a v_writelane, every use a v_readlane, ~170 of its ~1020
// vector instructions per vector (profiles/r04_float_encode.txt, late round 4, point d).  Here a lane keeps its OWN sixteen exception bits in one
// VGPR (one v_addc per value: bits = 2 * bits + exception) next to four scalar per-step counts; the filler is settled right behind the first value
// step (a vector that begins with 256 exceptions takes the old analysis: rare), so that every later step selects it straight from its compare;
// the record's stage asks the bits for each step's masks again (for_each_exception_bits_f32).  Same integers, same bytes.
struct AlpEncodedLeanF {
	int32_t  enc[4][4];   // exception slots already hold the filler
	uint32_t excbits;     // per lane: value k = 4 m + j at bit 15 - k
	uint32_t cnt_m[4];    // exceptions per value step (wave-uniform)
	int      cnt;
	int32_t  base;
	int      bw;
};
// bits = 2 * bits + (this lane in mask): one vector instruction (the mask goes in as the carry)
__device__ __forceinline__ uint32_t push_exception_bit(uint32_t bits, uint64_t mask) {
	uint32_t out;
	asm("v_addc_co_u32_e64 %0, vcc, %1, %1, %2" : "=v"(out) : "v"(bits), "s"(mask) : "vcc");
	return out;
}
// lane mask of value (m, j) out of the lanes' bits
__device__ __forceinline__ uint64_t exception_mask_of(uint32_t excbits, int m, int j) { return ballot64(((excbits >> (15 - (4 * m + j))) & 1u) != 0u); }

__device__ __forceinline__ void encode_alp_lean_f32(const VecInF& in, int e, int f, int lane, AlpEncodedLeanF& R) {
	const float    exp10  = kExpArrF[e];
	const float    frac_f = kFracArrF[f];
	const uint32_t fact   = kFactArrF[f];
	const float    frac_e = kFracArrF[e];
	R.cnt     = 0;
	R.excbits = 0;
	int32_t filler = 0;
	int32_t mn = INT32_MAX, mx = INT32_MIN;
	// one value step: encoded integers, the step's four lane masks (they die with the step), count, the lanes' bits
	auto value_step = [&](int m, uint64_t (&b)[4]) {
		uint32_t c = 0;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const float    v    = in.x[m][j];
			const uint32_t bits = __float_as_uint(v);
			// pass 1 (encoder.hpp:326-338): NaN, +-Inf and -0.0 are replaced by (float)ENCODING_UPPER_LIMIT
			const bool    special = ((bits & 0x7FFFFFFFu) >= 0x7F800000u) || bits == 0x80000000u;
			const float   vv      = special ? kUpperLimitF : v;
			const int32_t enc     = encode_value_f32(vv, exp10, frac_f);
			const float   dec     = decode_value_f32(enc, fact, frac_e);
			R.enc[m][j]           = enc;
			b[j]                  = ballot64(dec != vv);
			c += static_cast<uint32_t>(__builtin_popcountll(b[j]));
			R.excbits = push_exception_bit(R.excbits, b[j]);
		}
		R.cnt_m[m] = c;
		R.cnt += static_cast<int>(c);
	};
	auto fill_step = [&](int m, const uint64_t (&b)[4]) {
#pragma unroll
		for (int j = 0; j < 4; ++j) { R.enc[m][j] = lane_in_f32(b[j]) ? filler : R.enc[m][j]; }
	};
	uint64_t b0[4];
	value_step(0, b0);
	// filler = encoded value at the first non-exception position (encoder.hpp:382-388), 0 when that position is 1023.  It lies in step 0 unless the
	// vector begins with 256 exceptions (`late`, rare): then the steps run with a provisional 0 and the fix-up below settles it from the lanes' bits.
	const bool late = R.cnt_m[0] == 256u; // wave-uniform
	auto first_clean_of = [&](int m, const uint64_t (&b)[4], int& pos) {
		int     best_pos = 1 << 20, best_j = 0, best_l = 0;
		int32_t cand     = 0;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const uint64_t nj = ~b[j];
			const int      lj = nj ? __builtin_ctzll(nj) : 64;
			const int      pj = nj ? 4 * lj + j : (1 << 20);
			if (pj < best_pos) {
				best_pos = pj;
				best_j   = j;
				best_l   = lj;
			}
		}
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			if (best_j == j) { cand = __builtin_amdgcn_readlane(R.enc[m][j], best_l & 63); }
		}
		pos = 256 * m + best_pos;
		return cand;
	};
	if (!late) {
		int pos;
		filler = first_clean_of(0, b0, pos); // (pos < 256: never 1023)
	}
	fill_step(0, b0);
#pragma unroll
	for (int m = 1; m < 4; ++m) {
		uint64_t b[4];
		value_step(m, b);
		fill_step(m, b);
	}
	if (__builtin_expect(late, 0)) { // the non-exceptions are untouched, the exception slots hold the provisional 0: settle the filler, fill again
		bool found = false;
#pragma unroll
		for (int m = 1; m < 4; ++m) {
			if (!found && R.cnt_m[m] != 256u) {
				uint64_t b[4];
#pragma unroll
				for (int j = 0; j < 4; ++j) { b[j] = exception_mask_of(R.excbits, m, j); }
				int           pos;
				const int32_t cand = first_clean_of(m, b, pos);
				filler             = pos == 1023 ? 0 : cand;
				found              = true;
			}
		}
		if (filler != 0) {
#pragma unroll
			for (int m = 0; m < 4; ++m) {
#pragma unroll
				for (int j = 0; j < 4; ++j) { R.enc[m][j] = lane_in_f32(exception_mask_of(R.excbits, m, j)) ? filler : R.enc[m][j]; }
			}
		}
	}
	// (min / max behind the value steps, not inside them: two more live registers there and the kernel no longer fits its 64)
#pragma unroll
	for (int m = 0; m < 4; ++m) {
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			mn = R.enc[m][j] < mn ? R.enc[m][j] : mn;
			mx = R.enc[m][j] > mx ? R.enc[m][j] : mx;
		}
	}
#define ALPGPU_MINMAX_STEP(CTRL, ROWS)                                                                                  \
	{                                                                                                                   \
		const int32_t omn = __builtin_amdgcn_update_dpp(mn, mn, CTRL, ROWS, 0xf, false);                                \
		const int32_t omx = __builtin_amdgcn_update_dpp(mx, mx, CTRL, ROWS, 0xf, false);                                \
		mn                = omn < mn ? omn : mn;                                                                        \
		mx                = omx > mx ? omx : mx;                                                                        \
	}
	ALPGPU_MINMAX_STEP(0x111, 0xf) // row_shr:1
	ALPGPU_MINMAX_STEP(0x112, 0xf) // row_shr:2
	ALPGPU_MINMAX_STEP(0x114, 0xf) // row_shr:4
	ALPGPU_MINMAX_STEP(0x118, 0xf) // row_shr:8
	ALPGPU_MINMAX_STEP(0x142, 0xa) // row_bcast:15 into rows 1 and 3
	ALPGPU_MINMAX_STEP(0x143, 0xc) // row_bcast:31 into rows 2 and 3
#undef ALPGPU_MINMAX_STEP
	mn     = __builtin_amdgcn_readlane(mn, 63);
	mx     = __builtin_amdgcn_readlane(mx, 63);
	R.base = mn;
	R.bw   = count_bits32(mx, mn);
}

// for_each_exception_f32 over the lanes' bits: a step's four masks are asked for again (two vector instructions each) only if the step has exceptions
template <class F>
__device__ __forceinline__ void for_each_exception_bits_f32(uint32_t excbits, const uint32_t (&cnt_m)[4], int lane, F&& emit) {
	const uint64_t lt   = lanemask_lt64(lane);
	int            soff = 0;
#pragma unroll
	for (int m = 0; m < 4; ++m) {
		if (cnt_m[m] != 0) { // wave-uniform
			uint64_t b[4];
#pragma unroll
			for (int j = 0; j < 4; ++j) { b[j] = exception_mask_of(excbits, m, j); }
			int rank = soff;
#pragma unroll
			for (int j = 0; j < 4; ++j) { rank += __builtin_popcountll(b[j] & lt); }
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				if (lane_in_f32(b[j])) {
					emit(rank, m, j);
					++rank;
				}
			}
			soff += static_cast<int>(cnt_m[m]);
		}
	}
}


// ---- FFOR u32 pack from LDS: vals[i] = value - base (< 2^bw), natural order.  Output unit u = 8*k + a is the 16-byte group
// of stream word k for lane32 columns 4a..4a+3; lane handles units lane, lane + 64, ... -> 1-KiB contiguous stores.
// The same packing kept in registers (unit lane + 64*t in acc[t], t < ceil(8*bw / 64) <= 4): the single-pass encode packs
// before it waits for its output offset and only stores afterwards.
struct PackedUnitsF32 {
	u32x4 acc[4];
};
__device__ __forceinline__ void pack_u32_units(const EncodeLdsF32& L, int bw, int lane, PackedUnitsF32& P) {
	const u32x4* vals4   = reinterpret_cast<const u32x4*>(L.vals);
	const int    n_units = 8 * bw;
	// bit0 / bw without a per-lane division: bit0 <= 4032 and bw <= 64, so with M = floor(2^20 / bw) + 1 the error term
	// bit0 * (M * bw - 2^20) stays below 2^20 and (bit0 * M) >> 20 is the exact quotient (and fits 32 bits)
	const uint32_t inv_bw = bw > 0 ? (1u << 20) / static_cast<uint32_t>(bw) + 1u : 0u;
#pragma unroll
	for (int t = 0; t < 4; ++t) {
		u32x4     acc = {0u, 0u, 0u, 0u};
		const int u   = lane + 64 * t;
		if (64 * t < n_units && u < n_units) {
			const int k    = u >> 3;
			const int a    = u & 7;
			const int bit0 = 32 * k;
			// as pack_u64_units: the first row straddles the word's start (one right shift), the later ones start inside it
			int       r    = static_cast<int>((static_cast<uint32_t>(bit0) * inv_bw) >> 20); // = bit0 / bw
			const int sh   = bit0 - r * bw;                                                   // 0 <= sh < bw <= 32
			acc            = vals4[8 * r + a] >> static_cast<uint32_t>(sh);
			int   p        = bw - sh; // where row r + 1 starts inside this word (1..32)
			u32x4 nxt      = vals4[8 * (r + 1 < 32 ? r + 1 : 31) + a];
			++r;
			while (p < 32 && r < 32) {
				const u32x4 v = nxt;
				++r;
				nxt = vals4[8 * (r < 32 ? r : 31) + a];
				acc |= v << static_cast<uint32_t>(p);
				p += bw;
			}
		}
		P.acc[t] = acc;
	}
}
// The packing as a scatter from the registers that hold the values (encode_device.hpp: pack_u64_scatter): lane l holds, for m = 0..3, the
// quad (row 8m + l/8, columns 4(l%8) .. +3) — shifted to its place in stream word k = (row * bw) >> 5 and OR-ed into the wavefront-private
// image in LDS (ds_or_b32), upper bits of a straddling row into word k + 1; the image is read back as this lane's units.
// vals[m][j] = value - base (< 2^bw); `image`: 4 KiB of wavefront-private LDS.
__device__ __forceinline__ void pack_u32_scatter(uint32_t* image, const uint32_t (&vals)[4][4], int bw, int lane, PackedUnitsF32& P) {
	u32x4*    img4    = reinterpret_cast<u32x4*>(image);
	const int n_units = 8 * bw;
#pragma unroll
	for (int t = 0; t < 4; ++t) {
		if (64 * t < n_units) { img4[lane + 64 * t] = u32x4 {0u, 0u, 0u, 0u}; } // wave-uniform; whole 1-KiB blocks (the image has room for 256 units)
	}
	if (bw > 0) {
		const uint32_t p0 = static_cast<uint32_t>(lane >> 3) * static_cast<uint32_t>(bw);
		uint32_t*      wa = image + 4 * (lane & 7);
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			const uint32_t p = p0 + static_cast<uint32_t>(8 * m) * static_cast<uint32_t>(bw); // bit position of row 8m + l/8 in its column's stream
			const uint32_t k = p >> 5, s = p & 31u;
			uint32_t*      w = wa + 32 * k; // unit 8k + a
#pragma unroll
			for (int j = 0; j < 4; ++j) { __hip_atomic_fetch_or(w + j, vals[m][j] << s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
			if (s + static_cast<uint32_t>(bw) > 32u) { // the row's upper bits belong to word k + 1 (s >= 1 here)
#pragma unroll
				for (int j = 0; j < 4; ++j) { __hip_atomic_fetch_or(w + 32 + j, vals[m][j] >> (32u - s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
			}
		}
	}
#pragma unroll
	for (int t = 0; t < 4; ++t) {
		u32x4 acc = {0u, 0u, 0u, 0u};
		if (64 * t < n_units) { acc = img4[lane + 64 * t]; }
		P.acc[t] = acc;
	}
}
// The same scatter that LEAVES the image in LDS (round 4: nothing of the packed vector sits in registers while the wavefront waits for its
// ordered offset; the image's units are copied out afterwards, store_image_f32) — bytes [0, 128 * bw) of `image`.
__device__ __forceinline__ void pack_u32_scatter_image(uint32_t* image, const uint32_t (&vals)[4][4], int bw, int lane) {
	u32x4*    img4    = reinterpret_cast<u32x4*>(image);
	const int n_units = 8 * bw;
#pragma unroll
	for (int t = 0; t < 4; ++t) {
		if (64 * t < n_units) { img4[lane + 64 * t] = u32x4 {0u, 0u, 0u, 0u}; }
	}
	if (bw > 0) {
		const uint32_t p0 = static_cast<uint32_t>(lane >> 3) * static_cast<uint32_t>(bw);
		uint32_t*      wa = image + 4 * (lane & 7);
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			const uint32_t p = p0 + static_cast<uint32_t>(8 * m) * static_cast<uint32_t>(bw);
			const uint32_t k = p >> 5, s = p & 31u;
			uint32_t*      w = wa + 32 * k;
#pragma unroll
			for (int j = 0; j < 4; ++j) { __hip_atomic_fetch_or(w + j, vals[m][j] << s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
			if (s + static_cast<uint32_t>(bw) > 32u) {
#pragma unroll
				for (int j = 0; j < 4; ++j) { __hip_atomic_fetch_or(w + 32 + j, vals[m][j] >> (32u - s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
			}
		}
	}
	wave_lds_sync();
}
__device__ __forceinline__ void store_image_f32(const uint32_t* image, int bw, u32x4* __restrict__ out, int lane) {
	const u32x4* img4    = reinterpret_cast<const u32x4*>(image);
	const int    n_units = 8 * bw;
#pragma unroll
	for (int t = 0; t < 4; ++t) {
		const int u = lane + 64 * t;
#ifdef ALPGPU_F32_PLAIN_POLICY
		if (64 * t < n_units && u < n_units) { out[u] = img4[u]; }
#else
		if (64 * t < n_units && u < n_units) { __builtin_nontemporal_store(img4[u], out + u); } // written once
#endif
	}
}
__device__ __forceinline__ void store_packed_units_f32(const PackedUnitsF32& P, int bw, u32x4* __restrict__ out, int lane) {
	const int n_units = 8 * bw;
#pragma unroll
	for (int t = 0; t < 4; ++t) {
		const int u = lane + 64 * t;
		if (64 * t < n_units && u < n_units) { out[u] = P.acc[t]; }
	}
}

__device__ __forceinline__ void pack_u32_from_lds(const EncodeLdsF32& L, int bw, u32x4* __restrict__ out, int lane) {
	const u32x4* vals4   = reinterpret_cast<const u32x4*>(L.vals);
	const int    n_units = 8 * bw;
	// bit0 / bw without a per-lane division: bit0 <= 4032 and bw <= 64, so with M = floor(2^20 / bw) + 1 the error term
	// bit0 * (M * bw - 2^20) stays below 2^20 and (bit0 * M) >> 20 is the exact quotient (and fits 32 bits)
	const uint32_t inv_bw = bw > 0 ? (1u << 20) / static_cast<uint32_t>(bw) + 1u : 0u;
	for (int u = lane; u < n_units; u += 64) {
		const int k    = u >> 3;
		const int a    = u & 7;
		const int bit0 = 32 * k;
		int       r    = static_cast<int>((static_cast<uint32_t>(bit0) * inv_bw) >> 20); // = bit0 / bw
		int       p    = r * bw;
		u32x4     acc  = {0u, 0u, 0u, 0u};
		while (p < bit0 + 32 && r < 32) {
			const u32x4 v  = vals4[8 * r + a];
			const int   sh = p - bit0;
			acc |= sh >= 0 ? (v << static_cast<uint32_t>(sh)) : (v >> static_cast<uint32_t>(-sh));
			p += bw;
			++r;
		}
		out[u] = acc;
	}
}

} // namespace alpgpu
