// encode_lean_kernels.hip — the single-pass column encode (double) in the shape that keeps THREE tiles of eight vectors resident per CU.
//
// Same reference functions, same bytes and the same ordered-offset machinery as k_encode_fused (encode_kernels.hip; file:line relative to
// /root/reference: encoder.hpp:241-305 second-level sampling, :307-400 encode_simdized, :109-120 analyze_ffor,
// src/fastlanes_generated_ffor.cpp:7379-29749 ffor u64, :357-1775 ffor u16, rd.hpp:109-147 rd_encoder::encode).  What differs is what a
// wavefront HOLDS while it lives (residency is capped by the tile's 66 KiB of LDS as much as by its 97 VGPRs):
//
//   k_encode_fused                                              k_encode_lean
//   input 32 VGPRs + encoded integers 32 VGPRs, both live       input 32 VGPRs only: the analysis keeps lane masks, count, min / max — no integers;
//   through the arithmetic; packed units 32 VGPRs across the    the pack RECOMPUTES an integer (3 FP64 operations) right before it shifts it into
//   wait for the ordered offset                                  the image; the packed image and the exception record stay in LDS across the wait
//   8 KiB pack / exception image + 256 B per wavefront           6 KiB per wavefront: a 4 KiB image (stream words 0..31 of every column pair — all of a
//   (66 KiB per tile: two tiles per CU)                          vector of <= 32 bits) and the exception record behind it; a wider vector's words 32..63
//                                                                are packed in a second pass through the same 4 KiB AFTER its first half has been stored
//   97 VGPRs: four wavefronts per SIMD                           <= 72 VGPRs, 48.2 KiB per tile: three tiles = six wavefronts per SIMD, next to the
//                                                                persistent rowgroup search's one (6 x 72 + 64 <= 512 registers, 3 x 48.2 + 14 KiB LDS)
//
// A vector whose exception record does not fit behind its image (more than ~200 exceptions on a 32-bit vector) writes the record from its
// registers after the wait, as the two-pass form does.  Selected by ALPGPU_OPT_ENCODE_KERNEL (include/alpgpu.h); DESIGN.md §3.2 has the numbers.
#include "encode_device.hpp"
#include "encode_lookback.hpp"
#include "launch.hpp"

namespace alpgpu {

#ifndef ALPGPU_ENC_PRIO
#define ALPGPU_ENC_PRIO 1 // issue priority over the search workgroup that shares the CU: 1 and 2 alike on ALP columns (3.01 / 3.02 ms), 1 better on ALP_RD columns, whose search paces the encode (4.43 against 4.58 ms); 0: 3.16 / 4.46
#endif
#ifndef ALPGPU_LEAN_OCC
#define ALPGPU_LEAN_OCC 6 // __launch_bounds__ second argument (wavefronts per SIMD the register budget must admit): 6 -> <= 80 VGPRs (what three 48 KiB tiles per CU need), 7 -> <= 72, 8 -> <= 64
#endif

// Measurement builds (-DALPGPU_LEAN_STOP_AT=n): every wavefront ends behind stage n with ONE store that depends on what the stage produced, so the
// counters of successive builds (tools/pmc_busy.sh) difference into instructions per stage.  The column such a build writes is garbage.
#ifdef ALPGPU_LEAN_STOP_AT
#define ALPGPU_LEAN_STOP(n, expr)                                             \
	if (ALPGPU_LEAN_STOP_AT == (n)) {                                         \
		if (lane == 0) { LEAN_ARG_DESCS[v_read].base = static_cast<int64_t>(expr); }   \
		return;                                                               \
	}
#else
#define ALPGPU_LEAN_STOP(n, expr)
#endif

constexpr int kLeanImageWords = 32;                    // stream words per column pair that fit the image
constexpr int kLeanImageBytes = 128 * kLeanImageWords; // 4 KiB
constexpr int kLeanBufBytes   = 6144;                  // image + exception record, per wavefront

struct __attribute__((aligned(16))) LeanLds {
	uint64_t buf[kLeanBufBytes / 8];
};

// the encoded integer of one value, as the analysis computed it (encode_device.hpp: encode_alp_registers): bits(u) - bits(M) on the
// shortcut route, the x86 cast of (u - M) where the value step had taken the literal route (`wide`, wave-uniform)
// A wave-uniform double the optimiser cannot see through.  The pack recomputes (v * 10^e) * 10^-f + M from the input; handed the SAME
// multipliers as the analysis, the compiler recognises the expression and keeps all sixteen sums alive from the analysis to the pack instead
// (32 VGPRs: the very registers this kernel exists to give back).  With an opaque copy of the multiplier it is a new expression.
__device__ __forceinline__ double opaque_uniform(double x) {
	asm volatile("" : "+s"(x));
	return x;
}

__device__ __forceinline__ int64_t lean_encoded(double v, double exp10, double frac_f, bool wide) {
	const double t = (v * exp10) * frac_f;
	const double u = t + kMagic;
	if (wide) { return cast64_x86(u - kMagic); }
	return static_cast<int64_t>(static_cast<uint64_t>(__double_as_longlong(u)) - 0x4338000000000000ull);
}

// encode_simdized + analyze_ffor WITHOUT keeping the integers: exception lane masks per value step, their count, FOR base and width, the
// filler, and which value steps took the literal route (bit 2m + j of wide_steps).  Same decisions, value for value, as encode_alp_registers.
struct LeanAlp {
	uint64_t ballot[8][2];
	int      cnt;
	int64_t  base;
	int64_t  filler;
	int      bw;
	uint32_t wide_steps;
};
__device__ __forceinline__ void lean_analyze_alp(const VecIn& in, int e, int f, int lane, LeanAlp& R) {
	const double  exp10  = kExpArr[e];
	const double  frac_f = kFracArr[f];
	const int64_t fact   = kFactArr[f];
	const double  fact_d = kExpArr[f];
	const double  frac_e = kFracArr[e];
	R.cnt                = 0;
	R.wide_steps         = 0;
	const double qnan    = __longlong_as_double(0x7FF8000000000000ll);
	double       rmin = qnan, rmax = qnan;
	constexpr int kGroup = ALPGPU_ENCODE_GROUP;
#pragma unroll
	for (int m0 = 0; m0 < 8; m0 += kGroup) {
		double   vv[kGroup][2], rr[kGroup][2], dec[kGroup][2];
		uint64_t over_m[kGroup][2], wide_m[kGroup][2];
		uint64_t any_wide = 0;
#pragma unroll
		for (int g = 0; g < kGroup; ++g) {
#pragma unroll
			for (int j = 0; j < 2; ++j) {
				const double v = j == 0 ? in.x[m0 + g].x : in.x[m0 + g].y;
				double       t = v * exp10;
				t              = t * frac_f;
				const double u = t + kMagic;
				const double r = u - kMagic;
				const double prod = r * fact_d;
				vv[g][j]  = v;
				rr[g][j]  = r;
				dec[g][j] = prod * frac_e;
				const double ap = __builtin_fabs(prod);
				over_m[g][j]    = ballot64(ap > 0x1p63);
				wide_m[g][j]    = ballot64(!(__builtin_fabs(t) < 0x1p51)); // (|prod| == 2^63 cannot occur with |t| < 2^51: encode_device.hpp)
				any_wide |= wide_m[g][j];
			}
		}
		if (__builtin_expect(any_wide != 0, 0)) {
#pragma unroll
			for (int g = 0; g < kGroup; ++g) {
#pragma unroll
				for (int j = 0; j < 2; ++j) {
					if (wide_m[g][j] != 0) {
						dec[g][j] = decode_value(cast64_x86(rr[g][j]), fact, frac_e);
						R.wide_steps |= 1u << (2 * (m0 + g) + j);
					}
				}
			}
		}
#pragma unroll
		for (int g = 0; g < kGroup; ++g) {
#pragma unroll
			for (int j = 0; j < 2; ++j) {
				const uint64_t exc_m = ballot64(__double_as_longlong(dec[g][j]) != __double_as_longlong(vv[g][j])) | (over_m[g][j] & ~wide_m[g][j]);
				R.ballot[m0 + g][j]  = exc_m;
				R.cnt += __builtin_popcountll(exc_m);
				const uint64_t rb = static_cast<uint64_t>(__double_as_longlong(rr[g][j]));
				const double   rm = __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(lane_in(exc_m) ? 0x7FF80000u : static_cast<uint32_t>(rb >> 32)) << 32) | (rb & 0xFFFFFFFFull)));
				rmin = fmin_num(rmin, rm);
				rmax = fmax_num(rmax, rm);
			}
		}
	}
#if defined(ALPGPU_LEAN_STOP_AT) && ALPGPU_LEAN_STOP_AT == 3
	R.base = __double_as_longlong(rmin) ^ __double_as_longlong(rmax) ^ static_cast<int64_t>(R.ballot[0][0] ^ R.ballot[7][1] ^ R.ballot[3][0] ^ R.ballot[5][1]);
	R.filler = 0, R.bw = R.cnt;
	return;
#endif
	// filler = encoded value at the first non-exception position p (encoder.hpp:382-388); 0 when there is none or when p == 1023
	int64_t      filler  = 0;
	bool         found   = false;
	const double exp10_b = opaque_uniform(exp10); // recomputed, not carried over (see opaque_uniform)
#pragma unroll
	for (int m = 0; m < 8; ++m) {
		const uint64_t n0 = ~R.ballot[m][0];
		const uint64_t n1 = ~R.ballot[m][1];
		if (!found && (n0 | n1) != 0) { // wave-uniform, taken once
			const int  l0   = n0 ? __builtin_ctzll(n0) : 64;
			const int  l1   = n1 ? __builtin_ctzll(n1) : 64;
			const int  pos0 = 2 * l0, pos1 = 2 * l1 + 1;
			const bool first = pos0 < pos1;
			const int  l     = first ? l0 : l1;
			const int  p     = 128 * m + (first ? pos0 : pos1);
			const int64_t enc = first ? lean_encoded(in.x[m].x, exp10_b, frac_f, (R.wide_steps >> (2 * m)) & 1u)
			                          : lean_encoded(in.x[m].y, exp10_b, frac_f, (R.wide_steps >> (2 * m + 1)) & 1u);
			filler = p == 1023 ? 0 : readlane_i64(enc, l);
			found  = true;
		}
	}
	wave_minmax_f64(rmin, rmax);
	const bool none = R.cnt == kVec;
	int64_t    mn   = none ? filler : cast64_x86(rmin);
	int64_t    mx   = none ? filler : cast64_x86(rmax);
	if (R.cnt > 0) {
		mn = filler < mn ? filler : mn;
		mx = filler > mx ? filler : mx;
	}
	R.base   = mn;
	R.filler = filler;
	R.bw     = count_bits(mx, mn);
}

// rd_encoder::encode without keeping right / left parts: exception lane masks, count, and the packed left-index streams of this lane's two
// lane64 columns (16 rows x lbw <= 3 bits each).  Right parts are x & mask, left parts x >> rbw: recomputed from the input where needed.
struct LeanRd {
	uint64_t ballot[8][2];
	uint64_t acc0, acc1;
	int      cnt;
};
__device__ __forceinline__ void lean_analyze_rd(const VecIn& in, const alpgpu_rowgroup_state& rg, int lane, LeanRd& R, const uint16_t* __restrict__ order_rg, bool coherent) {
	const RdOrderView order = load_rd_order(order_rg, rg, lane, coherent);
	const int         rbw   = rg.rd_rbw;
	const int         ds    = rg.rd_dict_size;
	const int         lbw   = rg.rd_lbw;
	const uint64_t    lmask = (1ull << lbw) - 1ull;
	R.cnt  = 0;
	R.acc0 = R.acc1 = 0;
#pragma unroll
	for (int m = 0; m < 8; ++m) {
		const int row = 2 * m + (lane >> 5);
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			const double   v    = j == 0 ? in.x[m].x : in.x[m].y;
			const uint64_t bits = static_cast<uint64_t>(__double_as_longlong(v));
			const uint16_t left = static_cast<uint16_t>(bits >> rbw);
			int            idx  = ds;
#pragma unroll
			for (int d = 7; d >= 0; --d) {
				if (d < ds && rg.rd_dict[d] == left) { idx = d; }
			}
			const bool exc = idx == ds;
			R.ballot[m][j] = ballot64(exc);
			if (order.valid && R.ballot[m][j] != 0) { // rare, wave-uniform: the reference's index for a left part outside the dictionary
				const int ridx = rd_exception_index(order, left);
				idx            = exc ? ridx : idx;
			}
			R.cnt += __builtin_popcountll(R.ballot[m][j]);
			const uint64_t field = (static_cast<uint64_t>(static_cast<uint8_t>(idx)) & lmask) << (row * lbw);
			if (j == 0) {
				R.acc0 |= field;
			} else {
				R.acc1 |= field;
			}
		}
	}
	R.acc0 |= static_cast<uint64_t>(__shfl_xor(static_cast<long long>(R.acc0), 32));
	R.acc1 |= static_cast<uint64_t>(__shfl_xor(static_cast<long long>(R.acc1), 32));
}

// ---- the pack: stream words [k_lo, k_lo + n_words) of every column pair, OR-scattered into the (zeroed) image -------------------------------------
// Lane l holds, for m = 0..7, the pair (row 8m + l/8, columns 2(l%8), +1); its field starts at bit row*bw of the column's stream, i.e. in word
// k = (row*bw) >> 6 at shift s = (row*bw) & 63, and spills into word k + 1 when s + bw > 64.  Unit 8k + a (16 bytes) = word k of columns 2a, 2a + 1.
__device__ __forceinline__ void lean_zero_image(uint64_t* image, int n_words, int lane) {
	ull2v*    img2    = reinterpret_cast<ull2v*>(image);
	const int n_units = 8 * n_words;
#pragma unroll
	for (int t = 0; t < kLeanImageBytes / 1024; ++t) {
		if (64 * t < n_units) { img2[lane + 64 * t] = ull2v {0ull, 0ull}; } // wave-uniform; whole 1-KiB blocks
	}
}
// what a lane needs to place its pairs: the LDS byte address of its column pair's word 0 and the stream bit of its row in step 0
struct LeanPlace {
	uint32_t col_addr; // image + 16 * (lane % 8), as an LDS byte address
	uint32_t p0;       // (lane / 8) * bw
	uint32_t step;     // 8 * bw: bits between a lane's rows of consecutive steps (wave-uniform)
};
__device__ __forceinline__ LeanPlace lean_place(const uint64_t* image, int bw, int lane) {
	LeanPlace P;
	P.col_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(image)) + 16u * static_cast<uint32_t>(lane & 7);
	P.p0       = static_cast<uint32_t>(lane >> 3) * static_cast<uint32_t>(bw);
	P.step     = 8u * static_cast<uint32_t>(bw);
	return P;
}
__device__ __forceinline__ uint64_t* lean_lds_ptr(uint32_t byte_addr) {
	return reinterpret_cast<uint64_t*>(static_cast<uintptr_t>(byte_addr)); // (generic pointers to LDS carry the aperture; the address space is recovered below)
}
// one pair of step m.  WINDOWED = false: every word of the vector lies inside the image (bw <= 32): no range tests.
template <bool WINDOWED>
__device__ __forceinline__ void lean_scatter_pair(uint64_t* image, const LeanPlace& P, int m, int bw, uint64_t v0, uint64_t v1, int k_lo, int n_words, int lane) {
	const uint32_t p = P.p0 + static_cast<uint32_t>(m) * P.step;
	const int      k = static_cast<int>(p >> 6) - k_lo;
	const uint32_t s = p & 63u;
	uint64_t*      w = image + 2 * (lane & 7) + 16 * k;
	if (!WINDOWED || (k >= 0 && k < n_words)) {
		__hip_atomic_fetch_or(w, v0 << s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		__hip_atomic_fetch_or(w + 1, v1 << s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
	}
	if (s + static_cast<uint32_t>(bw) > 64u && (!WINDOWED || (k + 1 >= 0 && k + 1 < n_words))) { // (s >= 1 here)
		__hip_atomic_fetch_or(w + 16, v0 >> (64u - s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		__hip_atomic_fetch_or(w + 17, v1 >> (64u - s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
	}
}

// what the pack needs to know about an ALP vector (all wave-uniform, in scalar registers)
struct LeanAlpPack {
	double   exp10, frac_f;    // (the pack's own, opaque copies of the multipliers)
	uint64_t base_plus_magic;  // bits(M) + base: an integer minus the base is bits(u) minus this
	uint64_t fill_minus_base;  // what exception slots pack
	uint32_t wide_steps;       // value steps that took the literal route
};
// value (m, j) of an ALP vector, ready to pack: recomputed from the input
__device__ __forceinline__ uint64_t lean_alp_value(double v, const LeanAlpPack& A, uint64_t exc_ballot, bool wide) {
	uint64_t pv;
	if (__builtin_expect(wide, 0)) { // wave-uniform, almost never
		pv = static_cast<uint64_t>(lean_encoded(v, A.exp10, A.frac_f, true)) - (A.base_plus_magic - 0x4338000000000000ull);
	} else {
		const double u = ((v * A.exp10) * A.frac_f) + kMagic;
		pv             = static_cast<uint64_t>(__double_as_longlong(u)) - A.base_plus_magic;
	}
	return lane_in(exc_ballot) ? A.fill_minus_base : pv;
}
// all of an ALP vector's words that lie in [k_lo, k_lo + n_words)
template <bool WINDOWED>
__device__ __forceinline__ void lean_pack_alp(uint64_t* image, const VecIn& x, const LeanAlpPack& A, const uint64_t (&ballots)[8][2], int bw, int k_lo, int n_words, int lane) {
	lean_zero_image(image, n_words, lane);
	if (bw == 0) { return; }
	const LeanPlace P = lean_place(image, bw, lane);
	if (!WINDOWED && A.wide_steps == 0) {
		// the usual vector: <= 32 bits, no literal step.  The value minus the base fits 32 bits, and bits(M)'s low word is zero: the low word of
		// bits(u) minus the low word of the base IS the value (one subtract and one select per value instead of two each)
		const uint32_t base_lo = static_cast<uint32_t>(A.base_plus_magic), fill_lo = static_cast<uint32_t>(A.fill_minus_base);
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			const double   u0 = ((x.x[m].x * A.exp10) * A.frac_f) + kMagic, u1 = ((x.x[m].y * A.exp10) * A.frac_f) + kMagic;
			const uint32_t q0 = static_cast<uint32_t>(__double_as_longlong(u0)) - base_lo, q1 = static_cast<uint32_t>(__double_as_longlong(u1)) - base_lo;
			const uint64_t v0 = lane_in(ballots[m][0]) ? fill_lo : q0, v1 = lane_in(ballots[m][1]) ? fill_lo : q1;
			lean_scatter_pair<false>(image, P, m, bw, v0, v1, 0, n_words, lane);
		}
	} else {
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			const uint64_t v0 = lean_alp_value(x.x[m].x, A, ballots[m][0], (A.wide_steps >> (2 * m)) & 1u);
			const uint64_t v1 = lean_alp_value(x.x[m].y, A, ballots[m][1], (A.wide_steps >> (2 * m + 1)) & 1u);
			lean_scatter_pair<WINDOWED>(image, P, m, bw, v0, v1, k_lo, n_words, lane);
		}
	}
	wave_lds_sync();
}
// the right parts of an ALP_RD vector (always wider than the image: windowed)
__device__ __forceinline__ void lean_pack_rd(uint64_t* image, const VecIn& x, int bw, int k_lo, int n_words, int lane) {
	lean_zero_image(image, n_words, lane);
	const LeanPlace P    = lean_place(image, bw, lane);
	const uint64_t  mask = bw_mask(bw);
#pragma unroll
	for (int m = 0; m < 8; ++m) {
		lean_scatter_pair<true>(image, P, m, bw, static_cast<uint64_t>(__double_as_longlong(x.x[m].x)) & mask, static_cast<uint64_t>(__double_as_longlong(x.x[m].y)) & mask, k_lo,
		                        n_words, lane);
	}
	wave_lds_sync();
}

// the image's first n_units 16-byte units -> out[0 .. n_units): 1 KiB contiguous per instruction.
// All units are read into registers of their OWN before the first store (round 5, late): read and stored one by one, the compiler reuses one register quad, and the
// LDS read that overwrites it waits — s_waitcnt vmcnt(0), stores count in vmcnt on gfx9 — until the store before has taken its data: every store of a wavefront waited
// for the acknowledgement of the one before it (the "store acknowledgement" third of a wavefront's waits in profiles/r04_encode_levers.txt).
__device__ __forceinline__ void lean_store_image(const uint64_t* image, int n_units, ull2v* __restrict__ out, int lane) {
	const ull2v*  img2 = reinterpret_cast<const ull2v*>(image);
	constexpr int kT   = kLeanImageBytes / 1024;
#ifdef ALPGPU_LEAN_STORES_ONE_BY_ONE // (A/B: the form until late in round 5)
#pragma unroll
	for (int t = 0; t < kT; ++t) {
		const int u = lane + 64 * t;
		if (64 * t < n_units && u < n_units) { __builtin_nontemporal_store(img2[u], out + u); }
	}
#else
	ull2v r[kT];
#pragma unroll
	for (int t = 0; t < kT; ++t) { r[t] = img2[lane + 64 * t]; } // (inside the image whatever n_units is)
#pragma unroll
	for (int t = 0; t < kT; ++t) {
		const int u = lane + 64 * t;
#ifdef ALPGPU_LEAN_PLAIN_STORES // (A/B: until late in round 4 the lean kernel wrote its packed words with plain stores)
		if (64 * t < n_units && u < n_units) { out[u] = r[t]; }
#else
		if (64 * t < n_units && u < n_units) { __builtin_nontemporal_store(r[t], out + u); } // written once, read by nobody on this device soon
#endif
	}
#endif
}

// The five arguments needed only behind the wait are read from the kernarg segment where they are used (alp_device.hpp: late_kernel_arg):
// v_writelane 46 -> 24 and 1034 -> 981 vector instructions per vector (-DALPGPU_LEAN_ARGS_AT_ENTRY: the old form).
#ifndef ALPGPU_LEAN_ARGS_AT_ENTRY
#define ALPGPU_LEAN_LATE_ARGS 1
#endif

// UNORDERED (ALPGPU_OPT_ENCODE_UNORDERED, round 5): the tile does not wait for its predecessors' sizes.  Its last-arriving wavefront reserves the
// tile's packed / exception bytes with ONE agent-scope atomic add on a counter word behind the status words (same packing as a status word, so
// the value that comes back IS the tile's exclusive prefix) and the look-back is not run at all.  Every vector's bytes are what the ordered form
// writes, descriptor for descriptor; only WHERE a tile's bytes lie in the two streams follows the order in which tiles finished their analysis
// instead of vector order (the eight vectors of a tile stay together, in order).  A decoder never notices: descriptors carry offsets.
// (the first nine parameters are read by offset — alp_device.hpp: kArgDescs .. kArgExcCap — keep their order and types)
template <bool UNORDERED>
__global__ __launch_bounds__(64 * kFusedWaves, ALPGPU_LEAN_OCC) void k_encode_lean(const double* __restrict__ in, const alpgpu_rowgroup_state* __restrict__ rgs,
                                                                                    alpgpu_vector_desc* __restrict__ descs_entry, uint8_t* __restrict__ packed_entry,
                                                                                    uint8_t* __restrict__ excs_entry, uint64_t* __restrict__ status, uint64_t* __restrict__ totals,
                                                                                    uint64_t packed_capacity_entry, uint64_t exc_capacity_entry, uint64_t v_first, uint64_t n_vectors_launch,
                                                                                    const uint16_t* __restrict__ rd_order, uint32_t spin_limit, uint32_t async_states) {
	__builtin_amdgcn_s_setprio(ALPGPU_ENC_PRIO);
#ifdef ALPGPU_LEAN_LATE_ARGS
	(void)descs_entry, (void)packed_entry, (void)excs_entry, (void)packed_capacity_entry, (void)exc_capacity_entry;
#define LEAN_ARG_DESCS late_kernel_arg<alpgpu_vector_desc*>(kArgDescs)
#define LEAN_ARG_PACKED late_kernel_arg<uint8_t*>(kArgPacked)
#define LEAN_ARG_EXCS late_kernel_arg<uint8_t*>(kArgExcs)
#define LEAN_ARG_PACKED_CAP late_kernel_arg<uint64_t>(kArgPackedCap)
#define LEAN_ARG_EXC_CAP late_kernel_arg<uint64_t>(kArgExcCap)
#else
#define LEAN_ARG_DESCS descs_entry
#define LEAN_ARG_PACKED packed_entry
#define LEAN_ARG_EXCS excs_entry
#define LEAN_ARG_PACKED_CAP packed_capacity_entry
#define LEAN_ARG_EXC_CAP exc_capacity_entry
#endif
	__shared__ LeanLds  lds[kFusedWaves];
	__shared__ uint64_t s_size[kFusedWaves];
	__shared__ uint64_t s_excl;
	__shared__ uint32_t s_count;
	__shared__ uint32_t s_ready;
	const int           lane = lane_id();
	const int           wave = wave_in_wg();
	const uint64_t      tile = blockIdx.x;
#ifdef ALPGPU_LEAN_INIT_BARRIER_FIRST // (A/B: until late in round 4 the tile's two LDS words were set, and waited for, in front of the loads)
	if (threadIdx.x == 0) {
		s_count = 0;
		s_ready = 0;
		s_excl  = ~0ull;
	}
	__syncthreads();
#endif

	uint64_t*      buf  = lds[wave].buf;
	const uint64_t vl   = tile * kFusedWaves + wave;
	const bool     live = vl < n_vectors_launch;
	const uint64_t v    = v_first + vl;
	const uint64_t v_read = live ? v : v_first;
	// the first poll of the rowgroup's state (or its plain read) in FRONT of the 8 KiB everything waits for, the look at it behind them
	const alpgpu_rowgroup_state* rgp      = rgs + v_read / kRowgroup;
	const uint32_t               st_word  = async_states ? rowgroup_state_poll_begin(rgp, lane) : reinterpret_cast<const uint32_t*>(rgp)[lane & 7];
	// The input is read once and the packed words are written once: both non-temporal (round 4: 3.02 -> 2.91 ms per 1 Mi vectors on the mixed
	// column with both, 2.95 with the stores alone) — except the input of wavefront 0, which reads its vector a second time when it is wider
	// than the image (every ALP_RD vector) and should find it in the L2 / Infinity Cache then (all loads non-temporal: ALP_RD column 4.28 -> 4.41 ms).
#ifdef ALPGPU_LEAN_PREFETCH_TILES
	// Experiment: one 4-byte read per 128-byte line of the vector a wavefront ALPGPU_LEAN_PREFETCH_TILES tiles further on will load (about one
	// wavefront's life later): the line is then in the memory-side cache, a shorter trip than HBM under load.  Issued in front of this
	// wavefront's own loads and "used" right behind them, so that its register lives through the load phase only.
	uint32_t pf_word = 0;
	{
		const uint64_t vp = vl + static_cast<uint64_t>(ALPGPU_LEAN_PREFETCH_TILES) * kFusedWaves;
		if (vp < n_vectors_launch) { pf_word = *reinterpret_cast<const volatile uint32_t*>(reinterpret_cast<const uint8_t*>(in + (v_first + vp) * kVec) + 128 * lane); }
	}
#endif
#ifdef ALPGPU_LEAN_PLAIN_LOADS
	const VecIn                  x        = load_vector(in, v_read, lane);
#else
	const VecIn                  x        = load_vector_policy(in, v_read, lane, wave != 0);
#endif
#ifdef ALPGPU_LEAN_PREFETCH_TILES
	asm volatile("" ::"v"(pf_word));
#endif
#ifndef ALPGPU_LEAN_INIT_BARRIER_FIRST
	// the tile's two LDS words, set behind the ISSUE of the loads: the barrier that publishes them falls into the shadow of the input's round trip
	// (no wavefront touches them before its analysis is done).  s_excl starts as "stalled": should wavefront 0 give up on the rowgroup's state and
	// leave, it never runs the look-back, and the wavefronts the barrier below then releases must find ~0 there, not an earlier workgroup's
	// prefix: nothing of such a tile is written.
	if (threadIdx.x == 0) {
		s_count = 0;
		s_ready = 0;
		s_excl  = ~0ull;
	}
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // not __syncthreads(): its fence would wait for the loads in flight
#endif
	bool                         state_ok = true;
	const alpgpu_rowgroup_state  st       = async_states ? rowgroup_state_poll_finish(rgp, st_word, lane, spin_limit >> 4, state_ok) : unpack_rowgroup_state(st_word);
	if (!state_ok) { // wave-uniform
		if (lane == 0) { status_store(totals + 3, 1ull); }
		return;
	}
	ALPGPU_LEAN_STOP(1, __double_as_longlong(x.x[0].x + x.x[7].y) + st.k);
	alpgpu_vector_desc d;
	d.packed_off = d.exc_off = 0;
	d.base                   = 0;
	d.bw = d.e = d.f = d.lbw = 0;
	d.exc_cnt = d.scheme = 0;
	uint64_t ballots[8][2];
#pragma unroll
	for (int m = 0; m < 8; ++m) { ballots[m][0] = ballots[m][1] = 0; }
	int      cnt = 0;
	uint64_t acc0 = 0, acc1 = 0;     // ALP_RD: packed left streams
	uint64_t fill_minus_base = 0;    // ALP: what exception slots pack
	uint64_t base_plus_magic = 0;    // ALP: bits(M) + base — an integer is bits(u) minus this
	uint32_t wide_steps      = 0;
	double   exp10 = 1.0, frac_f = 1.0;
	if (live) {
		d.scheme = st.scheme;
		if (st.scheme == ALPGPU_SCHEME_ALP) {
			int e, f;
#ifdef ALPGPU_LEAN_NO_SECOND
			if (false) {
#else
			if (st.k > 1) {
#endif
				second_level_select(x, &st, reinterpret_cast<double*>(buf), lane, e, f);
			} else {
				e = st.combos[0];
				f = st.combos[1];
			}
			ALPGPU_LEAN_STOP(2, e * 32 + f);
			LeanAlp R;
			lean_analyze_alp(x, e, f, lane, R);
			ALPGPU_LEAN_STOP(3, R.base + R.bw);
			d.base = R.base, d.bw = static_cast<uint8_t>(R.bw), d.e = static_cast<uint8_t>(e), d.f = static_cast<uint8_t>(f);
			cnt             = R.cnt;
			wide_steps      = R.wide_steps;
			fill_minus_base = static_cast<uint64_t>(R.filler) - static_cast<uint64_t>(R.base);
			base_plus_magic = 0x4338000000000000ull + static_cast<uint64_t>(R.base);
			exp10           = opaque_uniform(kExpArr[e]); // the pack's own copy (see opaque_uniform)
			frac_f          = kFracArr[f];
#pragma unroll
			for (int m = 0; m < 8; ++m) { ballots[m][0] = R.ballot[m][0], ballots[m][1] = R.ballot[m][1]; }
		} else {
#ifndef ALPGPU_LEAN_NO_RD
			LeanRd R;
			lean_analyze_rd(x, st, lane, R, rd_order ? rd_order + (v / kRowgroup) * ALPGPU_RD_ORDER_STRIDE : nullptr, async_states != 0);
			d.bw = st.rd_rbw, d.lbw = st.rd_lbw;
			cnt  = R.cnt;
			acc0 = R.acc0, acc1 = R.acc1;
#pragma unroll
			for (int m = 0; m < 8; ++m) { ballots[m][0] = R.ballot[m][0], ballots[m][1] = R.ballot[m][1]; }
#endif
		}
		d.exc_cnt = static_cast<uint16_t>(cnt);
	}
	ALPGPU_LEAN_STOP(4, d.base + d.bw + cnt + static_cast<int64_t>(fill_minus_base ^ ballots[0][0] ^ ballots[7][1] ^ ballots[2][1] ^ ballots[5][0]) + wide_steps);
	uint64_t my_p = 0, my_e = 0; // bytes
	if (live) { record_sizes<8>(d, my_p, my_e); }
	uint64_t reserved = ~0ull; // UNORDERED: the tile's exclusive prefix, in the lane that asked for it
	bool     reserver = false;
	if (lane == 0) {
		s_size[wave] = status_pack(0, my_p >> 7, my_e >> 3);
		const uint32_t arrived = __hip_atomic_fetch_add(&s_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (arrived == kFusedWaves - 1) {
			uint64_t aggregate = 0;
#pragma unroll
			for (int w = 0; w < kFusedWaves; ++w) { aggregate += s_size[w]; }
			if constexpr (UNORDERED) {
				reserved = __hip_atomic_fetch_add(status + lookback_words(gridDim.x), aggregate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				reserver = true; // (the answer is looked at behind the pack and the record: one trip across the fabric in their shadow)
			} else {
				status_store(status + tile, kFlagAggregate | aggregate);
			}
		}
	}
	const uint64_t base_p = totals[0], base_e = totals[1];

	ALPGPU_LEAN_STOP(5, base_p + base_e + my_p + my_e);
	LookbackFirst look_first {0, 0};
	(void)look_first;
#ifdef ALPGPU_LEAN_LOOK_AHEAD_EARLY // experiment: in front of the pack already (the predecessors are mostly not there yet)
	if (wave == 0) { look_first = tile_lookback_begin(tile, status, lane); }
#endif
	// ---- pack (first window) and exception record: both into LDS, neither depends on where it will be stored ----
	const bool alp = d.scheme == ALPGPU_SCHEME_ALP;
	const int  bw  = d.bw;
	LeanAlpPack A;
	A.exp10 = exp10, A.frac_f = frac_f;
	A.base_plus_magic = uniform_u64(base_plus_magic), A.fill_minus_base = uniform_u64(fill_minus_base); // (wave-uniform by construction: scalar registers)
	A.wide_steps      = wide_steps;
	// A vector of <= 32 bits: all its words in the image.  A wider one (every ALP_RD vector): words 0..31 now; words 32.. after the wait, through
	// the same image.
	const int  words_a = bw < kLeanImageWords ? bw : kLeanImageWords; // stream words per column pair in the first window
	const bool wide_v  = bw > kLeanImageWords;                         // wave-uniform
#ifndef ALPGPU_LEAN_NO_PACK
	if (alp) {
		if (!wide_v) {
			lean_pack_alp<false>(buf, x, A, ballots, bw, 0, words_a, lane);
		} else {
			lean_pack_alp<true>(buf, x, A, ballots, bw, 0, kLeanImageWords, lane);
		}
	} else {
		lean_pack_rd(buf, x, bw, 0, words_a, lane);
	}
#endif
	ALPGPU_LEAN_STOP(6, buf[lane] + base_p);
	// Experiment (-DALPGPU_LEAN_LOOK_AHEAD): wavefront 0 asks for the status words of its look-back HERE, the exception record's stage below being
	// about one trip across the fabric long.  Measured: 3.04-3.09 ms against 2.90-2.95 without (mixed column) — the predecessors are in the same
	// stage at the same time, the early words are mostly not there yet and the round is simply spent twice.
#if defined(ALPGPU_LEAN_LOOK_AHEAD) && !defined(ALPGPU_LEAN_LOOK_AHEAD_EARLY)
	if (wave == 0) { look_first = tile_lookback_begin(tile, status, lane); }
#endif
	// exception record: cnt x value (8 B original bits, or the 2 B left part of an ALP_RD exception), then cnt x u16 position, pad zero; staged
	// behind the image when it fits there, else written from the registers after the wait
	const uint32_t rec_off    = static_cast<uint32_t>(128 * words_a);
	const bool     rec_staged = my_e <= kLeanBufBytes - rec_off;
	const uint32_t val_bytes  = alp ? 8u * static_cast<uint32_t>(cnt) : 2u * static_cast<uint32_t>(cnt);
	uint8_t*       img        = reinterpret_cast<uint8_t*>(buf) + rec_off;
#ifdef ALPGPU_LEAN_NO_EXC
	if (false) {
#else
	if (cnt > 0 && rec_staged) {
#endif
		if (lane == 0) { reinterpret_cast<uint64_t*>(img)[(my_e >> 3) - 1] = 0ull; } // the pad lives in the last word
		wave_lds_sync();
		for_each_exception(ballots, lane, [&](int r, int m, int j) {
			const uint64_t bits = static_cast<uint64_t>(__double_as_longlong(j == 0 ? x.x[m].x : x.x[m].y));
			if (alp) {
				reinterpret_cast<uint64_t*>(img)[r] = bits;
			} else {
				reinterpret_cast<uint16_t*>(img)[r] = static_cast<uint16_t>(bits >> bw);
			}
			reinterpret_cast<uint16_t*>(img + val_bytes)[r] = static_cast<uint16_t>(128 * m + 2 * lane + j);
		});
		wave_lds_sync();
	}

	ALPGPU_LEAN_STOP(7, buf[lane] + buf[600 + lane] + base_p);
	// ---- the ordered offset ----
	// Wavefront 0 finds the tile's offset; the others have nothing left to do but their stores, so they PARK at a workgroup barrier until it
	// arrives there too (k_encode_fused keeps its workers spinning on an LDS word: they used to pack meanwhile; here a spinning worker would
	// only take issue slots from the five other wavefronts of its SIMD — measured: with the pack or the record removed the kernel got SLOWER,
	// its wavefronts reached the spin earlier).  A wavefront that ended early (a stall) does not take part in the barrier any more.
#ifdef ALPGPU_LEAN_SPIN_WAIT
	if (wave == 0) { tile_lookback(tile, status, totals, s_size, &s_count, &s_excl, &s_ready, lane, spin_limit); }
	{
		uint32_t spins = 0;
		while (__hip_atomic_load(&s_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
			if (++spins > 64u * kSpinLimit) { return; }
			__builtin_amdgcn_s_sleep(ALPGPU_LEAN_SPIN_WAIT);
		}
	}
#else
#if defined(ALPGPU_LEAN_LOOK_AHEAD) || defined(ALPGPU_LEAN_LOOK_AHEAD_EARLY)
	if (wave == 0) { tile_lookback(tile, status, totals, s_size, &s_count, &s_excl, &s_ready, lane, spin_limit, &look_first); }
#else
	if constexpr (UNORDERED) {
		if (reserver) { s_excl = reserved; } // (one lane of one wavefront; every wavefront of the tile had posted its size when it asked)
	} else {
		if (wave == 0) { tile_lookback(tile, status, totals, s_size, &s_count, &s_excl, &s_ready, lane, spin_limit); }
	}
#endif
	__syncthreads();
#endif
	// sizes posted by the tile's earlier wavefronts: lane w < wave takes s_size[w], one DPP tree (both 31-bit fields stay apart: no carry between them)
	const uint64_t mine_sz = lane < wave ? s_size[lane & (kFusedWaves - 1)] : 0ull;
	const uint64_t local   = wave_sum_u64(mine_sz);
	const uint64_t excl    = s_excl;
	if (excl == ~0ull) { return; } // stalled: nothing of this tile is written
	const uint64_t pre = excl + local;
	d.packed_off       = base_p + ((pre >> 31) & 0x7FFFFFFFull) * 128ull;
	d.exc_off          = base_e + (pre & 0x7FFFFFFFull) * 8ull;
	if (!live) { return; }
	if (d.packed_off + my_p > LEAN_ARG_PACKED_CAP || d.exc_off + my_e > LEAN_ARG_EXC_CAP) {
		if (lane == 0) {
			__hip_atomic_store(totals + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			LEAN_ARG_DESCS[v] = empty_descriptor();
		}
		return;
	}
	ALPGPU_LEAN_STOP(8, d.packed_off + d.exc_off + buf[lane]);
	uint8_t* dst = LEAN_ARG_PACKED + d.packed_off;
	uint8_t* rec = LEAN_ARG_EXCS + d.exc_off;

	// ---- stores at the final offsets ----
	if (cnt > 0) {
		if (rec_staged) {
			const uint64_t* img64 = reinterpret_cast<const uint64_t*>(img);
			uint64_t*       rec64 = reinterpret_cast<uint64_t*>(rec);
			const int       n_w   = static_cast<int>(my_e >> 3);
#if defined(ALPGPU_LEAN_PLAIN_STORES)
			for (int w = lane; w < n_w; w += 64) { rec64[w] = img64[w]; }
#elif defined(ALPGPU_LEAN_STORES_ONE_BY_ONE)
			for (int w = lane; w < n_w; w += 64) { __builtin_nontemporal_store(img64[w], rec64 + w); }
#else
			for (int w0 = 0; w0 < n_w; w0 += 256) { // four words per lane and round in registers of their own (lean_store_image: no store waits for the one before it)
				uint64_t q[4];
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					const int w = w0 + 64 * k + lane;
					q[k]        = w < n_w ? img64[w] : 0ull;
				}
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					const int w = w0 + 64 * k + lane;
					if (w < n_w) { __builtin_nontemporal_store(q[k], rec64 + w); }
				}
			}
#endif
		} else { // a record larger than its staging room (rare): the vector is read again and the record written from it (as the two-pass form's pack kernel does)
			const VecIn xr   = load_vector(in, v, lane);
			uint16_t*   rpos = reinterpret_cast<uint16_t*>(rec + val_bytes);
			for_each_exception(ballots, lane, [&](int r, int m, int j) {
				const uint64_t bits = static_cast<uint64_t>(__double_as_longlong(j == 0 ? xr.x[m].x : xr.x[m].y));
				if (alp) {
					reinterpret_cast<uint64_t*>(rec)[r] = bits;
				} else {
					reinterpret_cast<uint16_t*>(rec)[r] = static_cast<uint16_t>(bits >> bw);
				}
				rpos[r] = static_cast<uint16_t>(128 * m + 2 * lane + j);
			});
			const int n_pos = static_cast<int>((my_e - val_bytes) >> 1);
			if (cnt + lane < n_pos) { rpos[cnt + lane] = 0; }
		}
	}
	lean_store_image(buf, 8 * words_a, reinterpret_cast<ull2v*>(dst), lane);
#ifndef ALPGPU_LEAN_NO_WIDE
	if (wide_v) { // words 32.. of every column pair: the image is free again once its reads above are issued (one wavefront's LDS operations run in order)
		// The input is still in the registers of wavefronts 1..7, which only slept on an LDS word meanwhile; wavefront 0 ran the look-back, whose
		// registers the input would not fit beside (the budget of three tiles per CU): it reads its vector again (an L2 / Infinity-Cache hit).
		VecIn xb = x;
		if (wave == 0) { xb = load_vector(in, v, lane); }
		wave_lds_sync();
		const int words_b = bw - kLeanImageWords;
		if (alp) {
			lean_pack_alp<true>(buf, xb, A, ballots, bw, kLeanImageWords, words_b, lane);
		} else {
			lean_pack_rd(buf, xb, bw, kLeanImageWords, words_b, lane);
		}
		lean_store_image(buf, 8 * words_b, reinterpret_cast<ull2v*>(dst + kLeanImageBytes), lane);
	}
#endif
	if (!alp && lane < 32) {
		uint32_t* out32 = reinterpret_cast<uint32_t*>(dst + 128ull * d.bw);
		for (int k = 0; k < d.lbw; ++k) {
			out32[32 * k + lane] = (static_cast<uint32_t>(acc0 >> (16 * k)) & 0xFFFFu) | ((static_cast<uint32_t>(acc1 >> (16 * k)) & 0xFFFFu) << 16);
		}
	}
	if (lane == 0) { LEAN_ARG_DESCS[v] = d; }
}

// the same launch sequence as launch_encode_fused_range (encode_kernels.hip) with the kernel above
void launch_k_encode_lean(hipStream_t stream, unsigned n_tiles, const double* d_in, const alpgpu_column* col, uint64_t* d_workspace, uint64_t first, uint64_t n_launch,
                          uint32_t spin_limit, uint32_t async_states, bool unordered) {
	if (unordered) {
		hipLaunchKernelGGL(k_encode_lean<true>, dim3(n_tiles), dim3(64 * kFusedWaves), 0, stream, d_in, col->d_rowgroups, col->d_vectors, col->d_packed, col->d_exc, d_workspace,
		                   col->d_totals, col->packed_capacity, col->exc_capacity, first, n_launch, col->d_rd_order, spin_limit, async_states);
	} else {
		hipLaunchKernelGGL(k_encode_lean<false>, dim3(n_tiles), dim3(64 * kFusedWaves), 0, stream, d_in, col->d_rowgroups, col->d_vectors, col->d_packed, col->d_exc, d_workspace,
		                   col->d_totals, col->packed_capacity, col->exc_capacity, first, n_launch, col->d_rd_order, spin_limit, async_states);
	}
}

} // namespace alpgpu
