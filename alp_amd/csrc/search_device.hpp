// search_device.hpp — the (e, f) candidate walk of the rowgroup search (alp::encoder<PT>::find_top_k_combinations, /root/reference include/alp/encoder.hpp:139-235),
// used by the search kernels (init_kernels.hip); a header of its own since round 6, when the double encode's tiles ran the walk as work items in their look-back
// wait (an experiment that lost: profiles/r06_encode_levers.txt, the code in profiles/r06_encode_experiments.diff).  The candidate order (e = 18..0, f = e..0) and the reference's update rule make the winner "the first
// candidate in that order with the minimum estimated size", i.e. the minimum of (size << 8 | candidate index).
#pragma once
#include "alp_device_f32.hpp"

namespace alpgpu {

struct ComboTable {
	uint8_t e[192];
	uint8_t f[192];
};
constexpr ComboTable make_combo_table(int max_exponent) {
	ComboTable t {};
	int        c = 0;
	for (int e = max_exponent; e >= 0; --e) {
		for (int f = e; f >= 0; --f) {
			t.e[c] = static_cast<uint8_t>(e);
			t.f[c] = static_cast<uint8_t>(f);
			++c;
		}
	}
	return t;
}
__device__ __constant__ const ComboTable kCombos64 = make_combo_table(18); // Constants<double>::MAX_EXPONENT: 190 candidates
__device__ __constant__ const ComboTable kCombos32 = make_combo_table(10); // Constants<float>::MAX_EXPONENT: 66 candidates

// What differs between alp::encoder<double> and alp::encoder<float> in the rowgroup search (constants.hpp:30-64 / :66-154).
// Encoded integers travel as int64 in both (sign-extended for float).
struct PrecF64 {
	using value_t = double;
	static constexpr int      kBits        = 64;
	static constexpr int      kNumCombos   = 190;
	static constexpr uint32_t kExcBits     = 64u;        // EXCEPTION_SIZE
	static constexpr uint32_t kRdThreshold = 48u * 32u;  // RD_SIZE_THRESHOLD_LIMIT
	static constexpr int64_t  kEncMin = INT64_MIN, kEncMax = INT64_MAX;
	struct Coef {
		double  exp10, frac_f, frac_e, fact_d, sentinel_from;
		int64_t fact;
	};
	static __device__ __forceinline__ const ComboTable& combos() { return kCombos64; }
	static __device__ __forceinline__ Coef coef(int e, int f) {
		return Coef {kExpArr[e], kFracArr[f], kFracArr[e], static_cast<double>(kFactArr[f]) /* 10^f <= 10^18: exact */,
		             f == 0 ? kUpperLimit : __builtin_inf(), kFactArr[f]};
	}
	// What one candidate accumulates over the samples of a vector: how many round-trip, and the range of their encodings.
	struct Acc {
		int     non_exc;
		int64_t mx, mn;   // literal path
		double  mxd, mnd; // double-only path (integers held exactly)
	};
	static __device__ __forceinline__ void start(Acc& a) {
		a.non_exc = 0;
		a.mx = kEncMin, a.mn = kEncMax;
		a.mxd = -__builtin_inf(), a.mnd = __builtin_inf();
	}
	// encode_value<true> + decode_value + compare for one sample (encoder.hpp:172-175).
	//
	// The literal arithmetic needs double -> int64, a 64-bit multiply and int64 -> double, none of which the hardware has as
	// one instruction.  All of it can stay in doubles whenever the integers involved are exactly representable:
	//   t = (v * 10^e) * 10^-f;  encodable (|t| <= 2^63 - 1024, not -0.0)  ->  r = trunc((t + M) - M)  is the encoded integer,
	//   exactly, as a double (that IS the reference's arithmetic; the cast truncates — which matters for -2^52 < t < -2^51, where
	//   t + M lands below 2^52 and keeps a half — and otherwise only changes the type);
	//   p = r * 10^f: if |p| < 2^63 the int64 product does not wrap and p = fl(r * 10^f) is what converting it to double gives
	//   (both are the correctly rounded exact integer), so the decoded value is p * 10^-e.
	// Every other case but two is a certain exception, so nothing has to be computed for it:
	//   * encodable, |p| > 2^63: the int64 product wraps.  Up to 2^64 it comes back with the opposite sign of v (and is not zero),
	//     beyond that the decoder's integer is below 2^63 in magnitude while |v| >= (2^64 - 10^f) * 10^-e (1 - 2^-50), more
	//     than 1.8 times the largest value it can decode to.
	//   * not encodable (the reference encodes the sentinel 2^63 - 1024): t NaN means v NaN; v = -0.0 (or an underflow to it)
	//     never equals the decoded sentinel, which is at least 10^-18 in magnitude; for f >= 1 or |t| >= 2^64 or t < 0 the
	//     decoded sentinel is <= 2^63 * 10^-e (1 + 2^-52) against |v| >= 10 * 2^63 * 10^-e (1 - 2^-50) (resp. 2 *, resp. the
	//     other sign).
	// Left for the literal arithmetic: |p| == 2^63 exactly (the exact product may be on either side of the wrap), and the sentinel
	// with f = 0 and 2^63 - 1024 < t < 2^64, where fl(fl((2^63 - 1024) * 10^-e) * 10^e) can come back as 2^63.  That branch is
	// skipped unless some lane of the wavefront is there.
	// Written with non-short-circuit logic on purpose: every `&&` here costs the scalar unit an exec-mask dance per sample.
	//   ok      : |p| < 2^63 implies |t| < 2^63 (encodable, not NaN); comparing BIT PATTERNS excludes t = -0.0 (v = -0.0 or an
	//             underflow to it: p and the decoded value are +0.0 then) and is the same as == everywhere else it can be true.
	//   literal : a superset of the two ambiguous cases is enough, the literal arithmetic is right wherever it runs, and it cannot
	//             overlap `ok` (|p| >= 2^63 in both).  c.sentinel_from = 2^63 - 1024 for f = 0, +inf otherwise.
	static __device__ __forceinline__ void step(Acc& a, double v, const Coef& c) {
		constexpr double k2p63 = 9223372036854775808.0, k2p64 = 18446744073709551616.0;
		const double t       = (v * c.exp10) * c.frac_f;
		const double r       = __builtin_trunc((t + kMagic) - kMagic); // not yet an integer for -2^52 < t < -2^51; the cast truncates
		const double p       = r * c.fact_d;
		const double ap      = __builtin_fabs(p);
		const double dec     = p * c.frac_e;
		const bool   ok      = (ap < k2p63) & (__double_as_longlong(dec) == __double_as_longlong(v));
		const bool   literal = (ap == k2p63) | ((t > c.sentinel_from) & (t < k2p64));
		// v_min/v_max_f64 ignore a quiet NaN: a failed sample only needs its high word replaced
		const uint64_t rb = static_cast<uint64_t>(__double_as_longlong(r));
		const double   rr = __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(ok ? static_cast<uint32_t>(rb >> 32) : 0x7FF80000u) << 32) | (rb & 0xFFFFFFFFull)));
		a.non_exc += ok ? 1 : 0;
		a.mxd = fmax_num(a.mxd, rr);
		a.mnd = fmin_num(a.mnd, rr);
		if (literal) {
			const int64_t enc = encode_value_safe(v, c.exp10, c.frac_f);
			if (decode_value(enc, c.fact, c.frac_e) == v) {
				++a.non_exc;
				a.mx = enc > a.mx ? enc : a.mx;
				a.mn = enc < a.mn ? enc : a.mn;
			}
		}
	}
	static __device__ __forceinline__ void finish(Acc& a) {
		if (a.mxd >= a.mnd) { // at least one sample went the double-only way: |values| < 2^63, the casts are exact
			const int64_t hi = static_cast<int64_t>(a.mxd), lo = static_cast<int64_t>(a.mnd);
			a.mx = hi > a.mx ? hi : a.mx;
			a.mn = lo < a.mn ? lo : a.mn;
		}
	}
	static __device__ __forceinline__ int      bits(int64_t mx, int64_t mn) { return count_bits(mx, mn); }
	static __device__ __forceinline__ uint64_t pattern(double v) { return static_cast<uint64_t>(__double_as_longlong(v)); }
};
struct PrecF32 {
	using value_t = float;
	static constexpr int      kBits        = 32;
	static constexpr int      kNumCombos   = 66;
	static constexpr uint32_t kExcBits     = 32u;
	static constexpr uint32_t kRdThreshold = 22u * 32u;
	static constexpr int64_t  kEncMin = INT32_MIN, kEncMax = INT32_MAX;
	struct Coef {
		float    exp10, frac_f, frac_e;
		uint32_t fact;
	};
	static __device__ __forceinline__ const ComboTable& combos() { return kCombos32; }
	static __device__ __forceinline__ Coef coef(int e, int f) { return Coef {kExpArrF[e], kFracArrF[f], kFracArrF[e], kFactArrF[f]}; }
	struct Acc {
		int     non_exc;
		int32_t mx32, mn32;
		int64_t mx, mn; // filled by finish()
	};
	static __device__ __forceinline__ void start(Acc& a) {
		a.non_exc = 0;
		a.mx32 = INT32_MIN, a.mn32 = INT32_MAX;
	}
	// every operation of the float arithmetic is one instruction; kept free of branches (see PrecF64::step)
	static __device__ __forceinline__ void step(Acc& a, float v, const Coef& c) {
		const int32_t q  = encode_value_f32(v, c.exp10, c.frac_f); // the SAFE branch does not exist as built (alp_device_f32.hpp)
		const bool    ok = decode_value_f32(q, c.fact, c.frac_e) == v;
		a.non_exc += ok ? 1 : 0;
		const int32_t hi = ok ? q : INT32_MIN, lo = ok ? q : INT32_MAX;
		a.mx32 = hi > a.mx32 ? hi : a.mx32;
		a.mn32 = lo < a.mn32 ? lo : a.mn32;
	}
	static __device__ __forceinline__ void finish(Acc& a) { a.mx = a.mx32, a.mn = a.mn32; }
	static __device__ __forceinline__ int      bits(int64_t mx, int64_t mn) { return count_bits32(static_cast<int32_t>(mx), static_cast<int32_t>(mn)); }
	static __device__ __forceinline__ uint64_t pattern(float v) { return static_cast<uint64_t>(__float_as_uint(v)); }
};

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		const uint32_t o = __shfl_xor(v, d);
		v                = o < v ? o : v;
	}
	return v;
}

} // namespace alpgpu
