// encode_device.hpp — wavefront-level building blocks of the ALP / ALP_RD vector ENCODE path (gfx950).
//
// Same ownership as the decode side: one 64-lane wavefront per 1024-value vector, lane L holds the value
// pairs i = 128*m + 2*L + j (m = 0..7, j = 0..1) in registers, so the 8 KiB input is read with eight
// 1-KiB-contiguous 16-B-per-lane loads and never re-read inside the kernel.
//
// Reference functions restated here (file:line relative to /root/reference):
//   second-level sampling  alp::encoder<double>::find_best_exponent_factor_from_combinations  include/alp/encoder.hpp:241-305
//   value encode + verify  alp::encoder<double>::encode_simdized                              include/alp/encoder.hpp:307-400
//   FOR analysis           alp::encoder<double>::analyze_ffor                                 include/alp/encoder.hpp:109-120
//   FFOR bit-packing       ffor::ffor (u64: src/fastlanes_generated_ffor.cpp:7379-29749; u16: :357-1775)
//   ALP_RD split           alp::rd_encoder<double>::encode                                    include/alp/rd.hpp:109-147
#pragma once
#include "alp_device.hpp"
#include "rd_exception_index.hpp"

namespace alpgpu {

#ifndef ALPGPU_ENCODE_GROUP
#define ALPGPU_ENCODE_GROUP 2
#endif

struct __attribute__((aligned(16))) EncodeLds {
	uint64_t vals[kVec]; // (enc - base) or RD right parts, natural index order: 8 KiB
	double   smp[32];    // second-level samples
};

struct VecIn {
	double2 x[8];
};

__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }

__device__ __forceinline__ int64_t wave_min_i64(int64_t v) {
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		const int64_t o = __shfl_xor(v, d);
		v               = o < v ? o : v;
	}
	return v;
}
__device__ __forceinline__ int64_t wave_max_i64(int64_t v) {
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		const int64_t o = __shfl_xor(v, d);
		v               = o > v ? o : v;
	}
	return v;
}

__device__ __forceinline__ int64_t readlane_i64(int64_t v, int src_lane) {
	const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(v), src_lane);
	const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32), src_lane);
	return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
}

// A wave-uniform lane mask (a ballot) as this lane's predicate: the SGPR pair is used directly as the select mask.
__device__ __forceinline__ bool lane_in(uint64_t ballot) { return __builtin_amdgcn_inverse_ballot_w64(ballot); }

// cross-lane copy of a double by DPP (register to register); lanes without a valid source lane keep their own value
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
	const uint64_t b  = static_cast<uint64_t>(__double_as_longlong(v));
	int            lo = static_cast<int>(static_cast<uint32_t>(b)), hi = static_cast<int>(static_cast<uint32_t>(b >> 32));
	lo                = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
	hi                = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
	return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(static_cast<uint32_t>(hi)) << 32) | static_cast<uint32_t>(lo)));
}
// the same where EVERY lane has a source lane (rotations inside a 16-lane row): no old value to keep, so no register copy in front
template <int CTRL>
__device__ __forceinline__ double dpp_all_f64(double v) {
	const uint64_t b  = static_cast<uint64_t>(__double_as_longlong(v));
	int            lo = static_cast<int>(static_cast<uint32_t>(b)), hi = static_cast<int>(static_cast<uint32_t>(b >> 32));
	lo                = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
	hi                = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
	return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(static_cast<uint32_t>(hi)) << 32) | static_cast<uint32_t>(lo)));
}
// NaN-ignoring minimum over each 16-lane row, left in every lane of the row (row_ror 1, 2, 4, 8: 3 instructions per step)
__device__ __forceinline__ double row_min_f64(double t) {
	t = fmin_num(t, dpp_all_f64<0x121>(t));
	t = fmin_num(t, dpp_all_f64<0x122>(t));
	t = fmin_num(t, dpp_all_f64<0x124>(t));
	t = fmin_num(t, dpp_all_f64<0x128>(t));
	return t;
}
__device__ __forceinline__ double f64_from_words(uint32_t lo, uint32_t hi) { return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(hi) << 32) | lo)); }

// Wavefront-wide NaN-ignoring min and max, returned wave-uniform.  ONE reduction chain for both: max = -min(-x); v_permlane32_swap (CDNA4)
// puts the minimum candidates of lane pairs (i, i + 32) side by side in lanes 0..31 and the negated maximum candidates in lanes 32..63, so a
// single v_min_f64 folds 64 -> 32 for both, four row rotations reduce every 16-lane row, and one row_bcast:15 joins rows (0,1) and (2,3):
// lane 31 holds the minimum, lane 63 minus the maximum.  25 vector instructions; two separate row_shr / row_bcast scans were 64, a third of
// them register copies (a DPP move that leaves some lanes unwritten is tied to its old value).
__device__ __forceinline__ void wave_minmax_f64(double& mn, double& mx) {
	const uint64_t a = static_cast<uint64_t>(__double_as_longlong(mn)), b = static_cast<uint64_t>(__double_as_longlong(mx)) ^ 0x8000000000000000ull;
	const auto     lo = __builtin_amdgcn_permlane32_swap(static_cast<uint32_t>(a), static_cast<uint32_t>(b), false, false);
	const auto     hi = __builtin_amdgcn_permlane32_swap(static_cast<uint32_t>(a >> 32), static_cast<uint32_t>(b >> 32), false, false);
	double         t  = fmin_num(f64_from_words(lo[0], hi[0]), f64_from_words(lo[1], hi[1]));
	t                 = row_min_f64(t);
	t                 = fmin_num(t, dpp_f64<0x142, 0xa>(t)); // row_bcast:15 into rows 1 and 3
	const uint64_t r  = static_cast<uint64_t>(__double_as_longlong(t));
	const uint32_t a0 = __builtin_amdgcn_readlane(static_cast<uint32_t>(r), 31), a1 = __builtin_amdgcn_readlane(static_cast<uint32_t>(r >> 32), 31);
	const uint32_t b0 = __builtin_amdgcn_readlane(static_cast<uint32_t>(r), 63), b1 = __builtin_amdgcn_readlane(static_cast<uint32_t>(r >> 32), 63);
	mn = f64_from_words(a0, a1);
	mx = f64_from_words(b0, b1 ^ 0x80000000u);
}

// The same reduction inside each 32-lane half: LANE 31 ends up with min and max of lanes 0..31, LANE 63 with those of lanes 32..63 (the other
// lanes hold nothing meaningful).  v_permlane16_swap folds rows (0,1) and (2,3): rows 0 / 2 then carry the halves' minimum candidates, rows
// 1 / 3 their negated maximum candidates; four row rotations; rows 1 / 3 fetch the minimum from the row below (row_bcast:15).
__device__ __forceinline__ void half_minmax_f64(double& mn, double& mx) {
	const uint64_t a = static_cast<uint64_t>(__double_as_longlong(mn)), b = static_cast<uint64_t>(__double_as_longlong(mx)) ^ 0x8000000000000000ull;
	const auto     lo = __builtin_amdgcn_permlane16_swap(static_cast<uint32_t>(a), static_cast<uint32_t>(b), false, false);
	const auto     hi = __builtin_amdgcn_permlane16_swap(static_cast<uint32_t>(a >> 32), static_cast<uint32_t>(b >> 32), false, false);
	double         t  = fmin_num(f64_from_words(lo[0], hi[0]), f64_from_words(lo[1], hi[1]));
	t                 = row_min_f64(t);
	mn                = dpp_f64<0x142, 0xa>(t);
	mx                = __longlong_as_double(static_cast<long long>(static_cast<uint64_t>(__double_as_longlong(t)) ^ 0x8000000000000000ull));
}

__device__ __forceinline__ VecIn load_vector(const double* __restrict__ in, uint64_t v, int lane) {
	const double2* p = reinterpret_cast<const double2*>(in + v * kVec);
	VecIn          r;
#pragma unroll
	for (int m = 0; m < 8; ++m) {
#ifdef ALPGPU_ENC_NT_LOAD
		typedef double d2v __attribute__((ext_vector_type(2)));
		const d2v q = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(p) + 64 * m + lane);
		r.x[m]      = make_double2(q.x, q.y);
#else
		r.x[m] = p[64 * m + lane];
#endif
	}
	return r;
}

// The same with a cache policy chosen per wavefront (k_encode_lean): `streaming` = non-temporal loads, for a vector that is read exactly once.
__device__ __forceinline__ VecIn load_vector_policy(const double* __restrict__ in, uint64_t v, int lane, bool streaming) {
	typedef double d2v __attribute__((ext_vector_type(2)));
	const d2v* p = reinterpret_cast<const d2v*>(in + v * kVec);
	VecIn      r;
	if (streaming) { // wave-uniform
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			const d2v q = __builtin_nontemporal_load(p + 64 * m + lane);
			r.x[m]      = make_double2(q.x, q.y);
		}
	} else {
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			const d2v q = p[64 * m + lane];
			r.x[m]      = make_double2(q.x, q.y);
		}
	}
	return r;
}

// ---- second-level sampling (encoder.hpp:241-305) --------------------------------------------------------
// 32 samples = input[32*s]; both half-waves evaluate one candidate each per round (lane & 31 = sample).
// All candidates are evaluated; the reference's early exit (two consecutive non-improvements) only stops
// evaluating, so replaying its sequential decision over the computed sizes gives the same (e,f).
// smp: 32 doubles of wavefront-private LDS
__device__ __forceinline__ void second_level_select(const VecIn& in, const alpgpu_rowgroup_state* __restrict__ rgp, double* smp, int lane,
                                                    int& e_out, int& f_out) {
	const int k = rgp->k;
	// sample s lives at index 32*s = 128*(s>>2) + 2*(16*(s&3)): element .x of step s>>2 in lane 16*(s&3)
	if ((lane & 15) == 0) {
#pragma unroll
		for (int m = 0; m < 8; ++m) { smp[4 * m + (lane >> 4)] = in.x[m].x; }
	}
	wave_lds_sync();
	const double sv   = smp[lane & 31];
	const int    half = lane >> 5;

	uint32_t sizes[5];
#pragma unroll
	for (int i = 0; i < 5; ++i) { sizes[i] = 0xFFFFFFFFu; }

#pragma unroll
	for (int kk = 0; kk < 6; kk += 2) {
		if (kk < k) { // wave-uniform
			// candidate kk for the low half-wave, kk + 1 for the high one (candidate 0 again if there is none).  Both candidates
			// are wave-uniform (rgp is a register copy of the state: constant indices only), so their table entries are scalar
			// reads and a lane merely selects its half's set.
			const int    kHi   = kk + 1 < 5 ? kk + 1 : 0; // a constant once the loop is unrolled
			const bool   has_b = kk + 1 < k;
			const int    e_a = rgp->combos[2 * kk], f_a = rgp->combos[2 * kk + 1];
			const int    e_b = has_b ? rgp->combos[2 * kHi] : rgp->combos[0], f_b = has_b ? rgp->combos[2 * kHi + 1] : rgp->combos[1];
			const bool   hi     = half != 0;
			const double exp10  = hi ? kExpArr[e_b] : kExpArr[e_a];
			const double frac_f = hi ? kFracArr[f_b] : kFracArr[f_a];
			const double frac_e = hi ? kFracArr[e_b] : kFracArr[e_a];
			const double fact_d = hi ? kExpArr[f_b] : kExpArr[f_a]; // 10^f, exact in double for f <= 18
			const double sentinel_from = (hi ? f_b : f_a) == 0 ? kUpperLimit : __builtin_inf();
			// the sample's round trip in doubles wherever that is provably the reference's arithmetic (the argument is spelled
			// out at PrecF64::step in init_kernels.hip); the two ambiguous cases send the whole wavefront down the literal path
			constexpr double k2p63 = 9223372036854775808.0, k2p64 = 18446744073709551616.0;
			const double t   = (sv * exp10) * frac_f;
			double       r   = __builtin_trunc((t + kMagic) - kMagic);
			const double p   = r * fact_d;
			const double ap  = __builtin_fabs(p);
			bool         ok  = (ap < k2p63) & (__double_as_longlong(p * frac_e) == __double_as_longlong(sv));
			const bool   lit = (ap == k2p63) | ((t > sentinel_from) & (t < k2p64));
			if (ballot64(lit) != 0) {
				const int64_t fact = hi ? kFactArr[f_b] : kFactArr[f_a];
				const int64_t enc  = encode_value_safe(sv, exp10, frac_f);
				ok                 = decode_value(enc, fact, frac_e) == sv;
				r                  = static_cast<double>(enc); // the cast of a double, the sentinel 2^63 - 1024 or -2^63: exact
			}
			const uint64_t bal  = ballot64(!ok);
			const uint32_t excs = __builtin_popcount(static_cast<uint32_t>(half ? (bal >> 32) : bal));
			const double   qnan = __longlong_as_double(0x7FF8000000000000ll);
			double         mn = ok ? r : qnan, mx = mn; // v_min / v_max_f64 ignore the quiet NaN of a failed sample
			half_minmax_f64(mn, mx);
			// no sample encodes: the reference compares INT64_MIN with INT64_MAX and gets a width of 1 (encoder.hpp:263-264,279)
			const bool     none = !(mx >= mn);
			const uint32_t size = 32u * static_cast<uint32_t>(count_bits(none ? INT64_MIN : cast64_x86(mx), none ? INT64_MAX : cast64_x86(mn))) + excs * 80u;
			const uint32_t s0   = __builtin_amdgcn_readlane(size, 31);
			const uint32_t s1   = __builtin_amdgcn_readlane(size, 63);
			sizes[kk] = s0;
			if (kk + 1 < 5) { sizes[kk + 1] = s1; }
		}
	}
	// the reference's sequential decision (encoder.hpp:283-301)
	int      best       = 0;
	uint32_t best_size  = sizes[0];
	int      worse      = 0;
	bool     stopped    = false;
#pragma unroll
	for (int i = 1; i < 5; ++i) {
		if (i < k && !stopped) {
			if (sizes[i] >= best_size) {
				worse++;
				if (worse == 2) { stopped = true; }
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

__device__ __forceinline__ void second_level_select(const VecIn& in, const alpgpu_rowgroup_state* __restrict__ rgp, EncodeLds& L, int lane,
                                                    int& e_out, int& f_out) {
	second_level_select(in, rgp, L.smp, lane, e_out, f_out);
}

// ---- encode_simdized + analyze_ffor for one vector held in registers -------------------------------------
// Outputs: enc[m][j] with exception slots overwritten by the filler; the exceptions of every (m,j) step as a
// wave-uniform lane mask (ballot); count; FOR base and bit width.
struct AlpEncoded {
	int64_t  enc[8][2];
	uint64_t ballot[8][2];
	int      cnt;
	int64_t  base;
	int      bw;
};

// Arithmetic shortcut (decided per value step, wave-uniformly).  Let t = (v * 10^e) * 10^-f as the reference computes it,
// u = t + M, r = u - M (M = 2^52 + 2^51).  If |t| < 2^51 then u lies in [2^52, 2^53], where consecutive doubles are
// consecutive integers: the encoded integer static_cast<int64_t>(r) equals bits(u) - bits(M) (one 64-bit subtract instead of
// a software double->int64 conversion and its x86 range check).  For the verification, (double)(int64)(enc * 10^f) is the
// correctly rounded value of an integer product that does not wrap when |enc * 10^f| < 2^63, i.e. exactly the IEEE product
// r * 10^f of two exactly representable doubles (one multiply instead of a 64-bit integer multiply and a software
// int64->double conversion).  A step in which any lane has |t| >= 2^51 (which includes +-Inf), a NaN, or |r * 10^f| == 2^63 is
// redone for the whole wavefront with the literal arithmetic; |r * 10^f| > 2^63 is an exception outright (argument at the
// test).  Results are bit-identical by construction.
__device__ __forceinline__ void encode_alp_registers(const VecIn& in, int e, int f, int lane, AlpEncoded& R) {
	const double  exp10  = kExpArr[e];
	const double  frac_f = kFracArr[f];
	const int64_t fact   = kFactArr[f];
	const double  fact_d = kExpArr[f]; // 10^f, exact in double for f <= 18
	const double  frac_e = kFracArr[e];
	R.cnt   = 0;
	// FOR analysis (encoder.hpp:109-120) rides along: the encoded integer of a non-exception equals r (an integer-valued
	// double, |r| <= 2^63) on both routes, so min / max are taken over r with exception lanes masked by a quiet NaN that
	// v_min_f64 / v_max_f64 ignore; exception slots later hold the filler, which is itself a non-exception's value.
	const double qnan = __longlong_as_double(0x7FF8000000000000ll);
	double       rmin = qnan, rmax = qnan;
	// Straight-line groups of kGroup value steps: all their floating-point chains first (independent of each other, so the
	// scheduler can interleave them), then their lane masks, then ONE wave-uniform test for the rare literal route.  A branch per
	// value step cuts the code into blocks of one 7-deep dependent chain each, and a wavefront with two or three neighbours on its
	// SIMD spends most of such a block waiting for its own previous result.
	constexpr int kGroup = ALPGPU_ENCODE_GROUP; // value steps (of 2 values) per group: 1, 2, 4 or 8
#pragma unroll
	for (int m0 = 0; m0 < 8; m0 += kGroup) {
		double   vv[kGroup][2], tt[kGroup][2], rr[kGroup][2], dec[kGroup][2];
		int64_t  enc[kGroup][2];
		uint64_t over_m[kGroup][2], wide_m[kGroup][2];
		uint64_t any_wide = 0;
#pragma unroll
		for (int g = 0; g < kGroup; ++g) {
#pragma unroll
			for (int j = 0; j < 2; ++j) {
				// pass 1 (encoder.hpp:326-338): for doubles only -0.0 matches (the mask literal evaluates to 0xFFE0000000000000,
				// SURVEY.md §8 A6) and is replaced by a value that cannot round-trip; NaN/Inf go through the arithmetic and fail the
				// compare.  Here -0.0 simply goes through as well (it encodes to 0 and decodes to +0.0, which == would accept) and is
				// made an exception by its bit pattern below.
				const double v = j == 0 ? in.x[m0 + g].x : in.x[m0 + g].y;
				double       t = v * exp10;
				t              = t * frac_f;
				const double u = t + kMagic;
				const double r = u - kMagic;
				const double prod = r * fact_d;
				vv[g][j]  = v;
				tt[g][j]  = t;
				rr[g][j]  = r;
				dec[g][j] = prod * frac_e;
				enc[g][j] = static_cast<int64_t>(static_cast<uint64_t>(__double_as_longlong(u)) - 0x4338000000000000ull);
				// |prod| > 2^63 (prod = fl(P), P = enc * 10^f exactly): the reference's int64 product wraps, and the wrapped value can
				// never decode to v — for |P| < 2^64 it has the opposite sign of v and is not zero, for larger |P| its magnitude is below
				// 2^63 while |v| 10^e > 0.96 * 2^64 — so the value is an exception without computing it.  Two-decimal values up to 10^5
				// land here all the time: the reference's search gives them (e,f) = (14,12), and |v| >= 92 233.72 wraps.  Only
				// |t| >= 2^51 (incl. Inf) and NaN take the literal route.
				// Every test goes straight from its compare into a lane mask (ballot of a compare = the compare's own SGPR result) and
				// the masks are combined as 64-bit integers on the scalar unit.
				const double ap = __builtin_fabs(prod);
				over_m[g][j]    = ballot64(ap > 0x1p63);                                                                   // the product wraps for sure
				// (round 4: until now a third compare also sent |prod| == 2^63 down the literal route.  It cannot happen on the shortcut route:
				//  there r is an integer with |r| <= 2^51, and r * 10^f rounds to +-2^63 only from [2^63 - 512, 2^63 + 1024], which holds no
				//  multiple of 10^f for f >= 4 and needs |r| > 2^51 for f <= 3 — tools/check_no_product_at_2p63.py enumerates it.)
				wide_m[g][j]    = ballot64(!(__builtin_fabs(t) < 0x1p51)); // |t| >= 2^51 (incl. Inf), NaN
				any_wide |= wide_m[g][j];
			}
		}
		if (__builtin_expect(any_wide != 0, 0)) { // wave-uniform, rare: the literal path of alp_device.hpp for the steps that need it
#pragma unroll
			for (int g = 0; g < kGroup; ++g) {
#pragma unroll
				for (int j = 0; j < 2; ++j) {
					if (wide_m[g][j] != 0) {
						enc[g][j] = cast64_x86(rr[g][j]);
						dec[g][j] = decode_value(enc[g][j], fact, frac_e);
					}
				}
			}
		}
#pragma unroll
		for (int g = 0; g < kGroup; ++g) {
#pragma unroll
			for (int j = 0; j < 2; ++j) {
				// The round trip holds iff the BITS agree: dec is never -0.0 (a product of an integer-valued double, or of a
				// converted int64, with positive powers of ten that cannot underflow) and never NaN on the literal route, which a NaN
				// input always takes; so -0.0 (encodes to 0, decodes to +0.0: pass 1 makes it an exception) and NaN fail the integer
				// compare exactly where the reference's float compare plus its special-value pass do.
				// (over_m decides only for lanes on the shortcut route: a lane of wide_m had enc / dec recomputed literally — wrap included
				//  — and is judged by the compare alone; with f = 0 and t = -2^63 - 2048 the reference's cast gives INT64_MIN, which
				//  decodes to -2^63 * 10^-e and CAN equal v)
				const uint64_t exc_m = ballot64(__double_as_longlong(dec[g][j]) != __double_as_longlong(vv[g][j])) | (over_m[g][j] & ~wide_m[g][j]);
				R.enc[m0 + g][j]     = enc[g][j];
				R.ballot[m0 + g][j]  = exc_m;
				R.cnt += __builtin_popcountll(exc_m);
				// min / max over r; a quiet NaN — only the high word is replaced — makes v_min / v_max_f64 skip an exception's lane
				const uint64_t rb = static_cast<uint64_t>(__double_as_longlong(rr[g][j]));
				const double   rm = __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(lane_in(exc_m) ? 0x7FF80000u : static_cast<uint32_t>(rb >> 32)) << 32) | (rb & 0xFFFFFFFFull)));
				rmin = fmin_num(rmin, rm);
				rmax = fmax_num(rmax, rm);
			}
		}
	}
	// filler = encoded value at the first non-exception position p (encoder.hpp:382-388); 0 when there is
	// none or when p == 1023 (the reference's index scan cannot see that case)
	int64_t filler = 0;
	bool    found  = false;
#pragma unroll
	for (int m = 0; m < 8; ++m) {
		const uint64_t n0 = ~R.ballot[m][0];
		const uint64_t n1 = ~R.ballot[m][1];
		if (!found && (n0 | n1) != 0) { // wave-uniform
			const int l0   = n0 ? __builtin_ctzll(n0) : 64;
			const int l1   = n1 ? __builtin_ctzll(n1) : 64;
			const int pos0 = 2 * l0, pos1 = 2 * l1 + 1;
			int       p;
			if (pos0 < pos1) {
				filler = readlane_i64(R.enc[m][0], l0);
				p      = 128 * m + pos0;
			} else {
				filler = readlane_i64(R.enc[m][1], l1);
				p      = 128 * m + pos1;
			}
			if (p == 1023) { filler = 0; }
			found = true;
		}
	}
#pragma unroll
	for (int m = 0; m < 8; ++m) {
#pragma unroll
		for (int j = 0; j < 2; ++j) { R.enc[m][j] = lane_in(R.ballot[m][j]) ? filler : R.enc[m][j]; }
	}
	wave_minmax_f64(rmin, rmax);
	// the filler takes part wherever there is an exception: normally it is one of the non-exception values, but it is 0 when
	// the only non-exception sits at position 1023 or when there is none
	const bool none = R.cnt == kVec;
	int64_t    mn   = none ? filler : cast64_x86(rmin);
	int64_t    mx   = none ? filler : cast64_x86(rmax);
	if (R.cnt > 0) {
		mn = filler < mn ? filler : mn;
		mx = filler > mx ? filler : mx;
	}
	R.base = mn;
	R.bw   = count_bits(mx, mn);
}

// Calls emit(rank, m, j) on every lane whose (m, j) slot is set in `ballot`; rank = number of set slots at smaller positions
// (position = 128*m + 2*lane + j), i.e. the slot's index in the ascending exception list.  m and j are compile-time at the call.
template <class F>
__device__ __forceinline__ void for_each_exception(const uint64_t (&ballot)[8][2], int lane, F&& emit) {
	(void)lane;
	int soff = 0;
#pragma unroll
	for (int m = 0; m < 8; ++m) {
		const uint64_t b0 = ballot[m][0], b1 = ballot[m][1];
		if ((b0 | b1) != 0) { // wave-uniform
			const int  before = static_cast<int>(mbcnt64(b1, mbcnt64(b0, static_cast<uint32_t>(soff))));
			const bool e0 = lane_in(b0), e1 = lane_in(b1);
			if (e0) { emit(before, m, 0); }
			if (e1) { emit(before + (e0 ? 1 : 0), m, 1); }
			soff += __builtin_popcountll(b0) + __builtin_popcountll(b1);
		}
	}
}

// ---- FFOR u64 pack from LDS (closed form of src/fastlanes_generated_ffor.cpp:7379-29749) -------------------
// vals[i] hold (value - base) & mask in natural index order.  Output unit u = 8*k + a is the 16-byte pair of
// stream word k for lane16 columns 2a, 2a+1; lane handles units lane, lane+64, ... -> 1-KiB contiguous stores.
// The same packing, kept in registers: unit lane + 64*t goes to acc[t] (t < ceil(8*bw / 64) <= 8).  The single-pass encode
// packs BEFORE it waits for its output offset, so that only the stores remain behind the wait (store_packed_units).
typedef unsigned long long ull2v __attribute__((ext_vector_type(2)));
struct PackedUnits {
	ull2v acc[8];
};
__device__ __forceinline__ void pack_u64_units(const uint64_t* vals, int bw, int lane, PackedUnits& P) {
	const ull2v* vals2   = reinterpret_cast<const ull2v*>(vals);
	const int    n_units = 8 * bw;
	// bit0 / bw without a per-lane division: bit0 <= 4032 and bw <= 64, so with M = floor(2^20 / bw) + 1 the error term
	// bit0 * (M * bw - 2^20) stays below 2^20 and (bit0 * M) >> 20 is the exact quotient (and fits 32 bits)
	const uint32_t inv_bw = bw > 0 ? (1u << 20) / static_cast<uint32_t>(bw) + 1u : 0u;
#pragma unroll
	for (int t = 0; t < 8; ++t) {
		ull2v     acc = {0ull, 0ull};
		const int u   = lane + 64 * t;
		if (64 * t < n_units && u < n_units) { // first test is wave-uniform
			// Word k of the column pair's stream = bits [64k, 64k + 64).  Its first contributing row r straddles the word's start
			// (its low `sh` bits belong to word k - 1): one right shift; every later row starts inside the word: left shifts only,
			// no direction to decide per row.  The next row's values are requested before the current ones are used.
			const int k    = u >> 3;
			const int a    = u & 7;
			const int bit0 = 64 * k;
			int       r    = static_cast<int>((static_cast<uint32_t>(bit0) * inv_bw) >> 20); // = bit0 / bw
			const int sh   = bit0 - r * bw;                                                   // 0 <= sh < bw <= 64
			acc            = vals2[8 * r + a] >> static_cast<unsigned long long>(sh);
			int   p        = bw - sh; // where row r + 1 starts inside this word (1..64)
			ull2v nxt      = vals2[8 * (r + 1 < 64 ? r + 1 : 63) + a];
			++r;
			while (p < 64 && r < 64) {
				const ull2v v = nxt;
				++r;
				nxt = vals2[8 * (r < 64 ? r : 63) + a];
				acc |= v << static_cast<unsigned long long>(p);
				p += bw;
			}
		}
		P.acc[t] = acc;
	}
}
// The same packing as a SCATTER from the registers that hold the values (no staging of the values, no per-word gather loop): lane l holds,
// for m = 0..7, the pair (row 8m + l/8, columns 2(l%8), 2(l%8) + 1) — one shift to its place in stream word k = (row * bw) >> 6, OR-ed into
// the wavefront-private output image in LDS (ds_or_b64; a row that straddles a word boundary adds its upper bits to word k + 1), and the
// image is read back as this lane's units.  ~11 vector instructions and 4 LDS atomics per pair; the gather form costs ~250 vector
// instructions per vector at 25 bits (profiles/r03_encode_levers.txt).  One wavefront's LDS operations execute in order: zero, OR, read.
// vals[m][j] = value - base (< 2^bw); `image`: 8 KiB of wavefront-private LDS.
__device__ __forceinline__ void pack_u64_scatter(uint64_t* image, const uint64_t (&vals)[8][2], int bw, int lane, PackedUnits& P) {
	ull2v*    img2    = reinterpret_cast<ull2v*>(image);
	const int n_units = 8 * bw;
#pragma unroll
	for (int t = 0; t < 8; ++t) {
		if (64 * t < n_units) { img2[lane + 64 * t] = ull2v {0ull, 0ull}; } // wave-uniform; whole 1-KiB blocks (the image has room for 512 units)
	}
	if (bw > 0) {
		const int      a  = lane & 7;
		const uint32_t p0 = static_cast<uint32_t>(lane >> 3) * static_cast<uint32_t>(bw);
		uint64_t*      wa = image + 2 * a;
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			const uint32_t p = p0 + static_cast<uint32_t>(8 * m) * static_cast<uint32_t>(bw); // bit position of row 8m + l/8 in its column's stream
			const uint32_t k = p >> 6, s = p & 63u;
			uint64_t*      w = wa + 16 * k; // unit 8k + a
			__hip_atomic_fetch_or(w, vals[m][0] << s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
			__hip_atomic_fetch_or(w + 1, vals[m][1] << s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
			if (s + static_cast<uint32_t>(bw) > 64u) { // the row's upper bits belong to word k + 1 (s >= 1 here)
				__hip_atomic_fetch_or(w + 16, vals[m][0] >> (64u - s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
				__hip_atomic_fetch_or(w + 17, vals[m][1] >> (64u - s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
			}
		}
	}
#pragma unroll
	for (int t = 0; t < 8; ++t) {
		ull2v acc = {0ull, 0ull};
		if (64 * t < n_units) { acc = img2[lane + 64 * t]; }
		P.acc[t] = acc;
	}
}
__device__ __forceinline__ void store_packed_units(const PackedUnits& P, int bw, ull2v* __restrict__ out, int lane) {
	const int n_units = 8 * bw;
#pragma unroll
	for (int t = 0; t < 8; ++t) {
		const int u = lane + 64 * t;
		if (64 * t < n_units && u < n_units) {
#ifndef ALPGPU_ENC_NT_STORE
			out[u] = P.acc[t];
#else
			__builtin_nontemporal_store(P.acc[t], out + u); // written once, read by nobody on this device soon
#endif
		}
	}
}

__device__ __forceinline__ void pack_u64_from_lds(const EncodeLds& L, int bw, ulonglong2* __restrict__ out, int lane) {
	const ulonglong2* vals2   = reinterpret_cast<const ulonglong2*>(L.vals);
	const int         n_units = 8 * bw;
	// bit0 / bw without a per-lane division: bit0 <= 4032 and bw <= 64, so with M = floor(2^20 / bw) + 1 the error term
	// bit0 * (M * bw - 2^20) stays below 2^20 and (bit0 * M) >> 20 is the exact quotient (and fits 32 bits)
	const uint32_t inv_bw = bw > 0 ? (1u << 20) / static_cast<uint32_t>(bw) + 1u : 0u;
	for (int u = lane; u < n_units; u += 64) {
		const int  k    = u >> 3;
		const int  a    = u & 7;
		const int  bit0 = 64 * k;
		int        r    = static_cast<int>((static_cast<uint32_t>(bit0) * inv_bw) >> 20); // = bit0 / bw
		int        p    = r * bw;
		ulonglong2 acc  = make_ulonglong2(0, 0);
		while (p < bit0 + 64 && r < 64) {
			const ulonglong2 v  = vals2[8 * r + a];
			const int        sh = p - bit0;
			if (sh >= 0) {
				acc.x |= v.x << sh;
				acc.y |= v.y << sh;
			} else {
				acc.x |= v.x >> (-sh);
				acc.y |= v.y >> (-sh);
			}
			p += bw;
			++r;
		}
		out[u] = acc;
	}
}

// ---- ALP_RD split of one vector held in registers (rd.hpp:109-147) ------------------------------------------
struct RdEncoded {
	uint64_t right[8][2];
	uint16_t left[8][2];  // original left parts
	uint8_t  idx[8][2];   // dictionary index; at exception slots the reference's map position (low 8 bits) or dict_size
	uint64_t ballot[8][2];
	int      cnt;
};

__device__ __forceinline__ void encode_rd_registers(const VecIn& in, const alpgpu_rowgroup_state& rg, int lane, RdEncoded& R,
                                                    const uint16_t* __restrict__ order_rg = nullptr, bool coherent = false) {
	const RdOrderView order = load_rd_order(order_rg, rg, lane, coherent);
	const int      rbw  = rg.rd_rbw;
	const uint64_t mask = bw_mask(rbw);
	const int      ds   = rg.rd_dict_size;
	R.cnt   = 0;
#pragma unroll
	for (int m = 0; m < 8; ++m) {
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			const double   v    = j == 0 ? in.x[m].x : in.x[m].y;
			const uint64_t bits = static_cast<uint64_t>(__double_as_longlong(v));
			R.right[m][j]       = bits & mask;
			const uint16_t left = static_cast<uint16_t>(bits >> rbw);
			int            idx  = ds;
#pragma unroll
			for (int d = 7; d >= 0; --d) {
				if (d < ds && rg.rd_dict[d] == left) { idx = d; }
			}
			const bool exc = idx == ds;
			R.left[m][j]   = left;
			R.ballot[m][j] = ballot64(exc);
			if (order.valid && R.ballot[m][j] != 0) { // rare, wave-uniform: the reference's index for a left part outside the dictionary
				const int ridx = rd_exception_index(order, left);
				idx            = exc ? ridx : idx;
			}
			R.idx[m][j] = static_cast<uint8_t>(idx); // only the low lbw (<= 3) bits are packed
			R.cnt += __builtin_popcountll(R.ballot[m][j]);
		}
	}
}

// FFOR u16 pack of the dictionary indices (64 lane-streams x 16 rows, words at out[64*k + lane64]);
// value i -> lane64 = i & 63, row = i >> 6.  A lane holds rows 2m + (lane>>5) of streams 2*(lane&31), +1;
// partner lane^32 holds the other 8 rows.  lbw <= 16 in general; the codec uses 1..3.
__device__ __forceinline__ void pack_left_u16(const uint16_t (&idx)[8][2], int lbw, uint32_t* __restrict__ out32, int lane) {
	if (lbw <= 4) {
		// whole stream (16 rows x lbw bits <= 64 bits) fits one u64 accumulator per stream
		const uint64_t lmask = (1ull << lbw) - 1ull;
		uint64_t       acc0 = 0, acc1 = 0;
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			const int row = 2 * m + (lane >> 5);
			acc0 |= (static_cast<uint64_t>(idx[m][0]) & lmask) << (row * lbw);
			acc1 |= (static_cast<uint64_t>(idx[m][1]) & lmask) << (row * lbw);
		}
		acc0 |= static_cast<uint64_t>(__shfl_xor(static_cast<long long>(acc0), 32));
		acc1 |= static_cast<uint64_t>(__shfl_xor(static_cast<long long>(acc1), 32));
		if (lane < 32) {
			for (int k = 0; k < lbw; ++k) {
				const uint32_t w0 = static_cast<uint32_t>(acc0 >> (16 * k)) & 0xFFFFu;
				const uint32_t w1 = static_cast<uint32_t>(acc1 >> (16 * k)) & 0xFFFFu;
				out32[32 * k + lane] = w0 | (w1 << 16);
			}
		}
	} else {
		// generic width: word by word
		const uint32_t lmask = lbw >= 16 ? 0xFFFFu : ((1u << lbw) - 1u);
		for (int k = 0; k < lbw; ++k) {
			uint32_t w0 = 0, w1 = 0;
#pragma unroll
			for (int m = 0; m < 8; ++m) {
				const int row = 2 * m + (lane >> 5);
				const int sh  = row * lbw - 16 * k; // position of this row's field relative to word k
				if (sh > -16 && sh < 16) {
					const uint32_t a = idx[m][0] & lmask, b = idx[m][1] & lmask;
					w0 |= sh >= 0 ? (a << sh) : (a >> (-sh));
					w1 |= sh >= 0 ? (b << sh) : (b >> (-sh));
				}
			}
			w0 = (w0 | __shfl_xor(w0, 32)) & 0xFFFFu;
			w1 = (w1 | __shfl_xor(w1, 32)) & 0xFFFFu;
			if (lane < 32) { out32[32 * k + lane] = w0 | (w1 << 16); }
		}
	}
}

} // namespace alpgpu
