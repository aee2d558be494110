// init_kernels.hip — per-rowgroup codec state for gfx950: sampling, (e,f) search, scheme decision, RD dictionary.
//
// Replaces, per rowgroup of <= 100 vectors (file:line relative to /root/reference):
//   alp::sampler::first_level_sample                 include/alp/sampler.hpp:14-52
//   alp::encoder<double>::find_top_k_combinations    include/alp/encoder.hpp:139-235
//   alp::encoder<double>::init                       include/alp/encoder.hpp:420-427
//   alp::rd_encoder<double>::init / find_best_dictionary / build_left_parts_dictionary
//                                                    include/alp/rd.hpp:180-185 / :89-104 / :33-87
//
// One workgroup of 9 wavefronts per rowgroup; wavefront w owns sampled vector w (rowgroup vectors 0,12,..,96),
// 32 samples each at stride 32 — the reference's sample set for whole-vector columns.  The (e,f) search maps
// the 190 candidates to lanes (3 per lane) and walks the 32 samples as LDS broadcasts, so there is no
// cross-lane traffic until the final arg-min.  The candidate order (e = 18..0, f = e..0) and the reference's
// update rule make the winner "the first candidate in that order with the minimum estimated size", i.e. the
// minimum of (size, candidate index).
#include <cstdlib>

#include "launch.hpp"
#include "rd_dictionary_order.hpp"
#include "search_device.hpp"

namespace alpgpu {

constexpr int kMaxSampledVectors = 9; // ceil(100 / 12)
constexpr int kInitThreads       = 64 * kMaxSampledVectors;
#ifndef ALPGPU_INIT_ASYNC_WAVES
#define ALPGPU_INIT_ASYNC_WAVES 4
#endif
// The persistent form beside two encode tiles (2 x 8 wavefronts x 96 VGPRs = 384 of a SIMD's 512 registers, 2 x 66 KiB of LDS): ONE
// wavefront per SIMD (4 per workgroup).  Measured (profiles/r03_async_init.txt): with the 128 VGPRs that exactly fill the SIMD the
// workgroup does NOT share the CU with two tiles (the encode beside it ran at half speed), with <= 96 it does; the 7 spilled dwords are
// outside the sample walk.  8 wavefronts of 64 VGPRs spill into the walk itself.
constexpr int kInitAsyncWaves        = ALPGPU_INIT_ASYNC_WAVES;
#ifndef ALPGPU_INIT_ASYNC_OCC
#define ALPGPU_INIT_ASYNC_OCC (ALPGPU_INIT_ASYNC_WAVES == 4 ? 5 : 8)
#endif
constexpr int kInitAsyncWavesPerSimd = ALPGPU_INIT_ASYNC_OCC; // __launch_bounds__ second argument: 4 -> <= 128 VGPRs, 5 -> <= 96, 8 -> <= 64

// (wave_scan_add_u32: alp_device.hpp)  The cut search below is a chain of dependent scans per 64-sample chunk; as __shfl_up steps
// (ds_bpermute, one LDS-crossbar round trip each) that chain was the latency that bounded ALP_RD rowgroups.
// values >= -1
__device__ __forceinline__ int wave_scan_max_i32(int v) {
	int t;
	t = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xf, false), v = t > v ? t : v;
	t = __builtin_amdgcn_update_dpp(-1, v, 0x112, 0xf, 0xf, false), v = t > v ? t : v;
	t = __builtin_amdgcn_update_dpp(-1, v, 0x114, 0xf, 0xf, false), v = t > v ? t : v;
	t = __builtin_amdgcn_update_dpp(-1, v, 0x118, 0xf, 0xf, false), v = t > v ? t : v;
	t = __builtin_amdgcn_update_dpp(-1, v, 0x142, 0xa, 0xf, false), v = t > v ? t : v;
	t = __builtin_amdgcn_update_dpp(-1, v, 0x143, 0xc, 0xf, false), v = t > v ? t : v;
	return v;
}

constexpr int kMaxSamples = kMaxSampledVectors * 32; // 288

// per-wavefront scratch of the ALP_RD cut search
struct __attribute__((aligned(16))) RdWaveScratch {
	uint32_t len[kMaxSamples];      // run length, stored at the run's first sorted position (0 elsewhere)
	uint32_t hist[kMaxSamples + 8]; // hist[L] = number of runs of length L
};

// The ALP_RD cut search only ever looks at the top 16 bits of a sample (cuts of 1..16 bits, rd.hpp:92), so the samples are
// sorted by a 32-bit composite key = (top 16 bits << 16) | sample index: unique, one compare per pair, and the sample index
// of every sorted entry comes along for free.
__device__ __forceinline__ uint32_t rd_left_of(uint32_t composite, int cut) { return (composite >> 16) >> (16 - cut); }

// Runs of equal left parts (the top `cut` bits) in the sorted samples: fills W.len, zeroes W.hist.  One wavefront.
// first != nullptr (the chosen cut only): first[p] = smallest original sample index inside the run that starts at sorted
// position p, i.e. the left part's first occurrence in the sample.
__device__ __forceinline__ void rd_build_runs(RdWaveScratch& W, uint32_t* first, const uint32_t* s_key, int n_smp, int cut, int lane) {
	for (int b = 0; b < kMaxSamples + 8; b += 64) {
		const int j = b + lane;
		if (j < kMaxSamples) {
			W.len[j] = 0u;
			if (first) { first[j] = 0xFFFFFFFFu; }
		}
		if (j < kMaxSamples + 8) { W.hist[j] = 0u; }
	}
	wave_lds_sync();
	int carry = 0; // start of the run that continues from the previous chunk
	for (int b = 0; b < n_smp; b += 64) {
		const int      j     = b + lane;
		const bool     valid = j < n_smp;
		const uint32_t left  = valid ? rd_left_of(s_key[j], cut) : 0u;
		const uint32_t prev  = (valid && j > 0) ? rd_left_of(s_key[j - 1], cut) : ~left;
		const bool     head  = valid && (j == 0 || left != prev);
		int            st    = wave_scan_max_i32(head ? j : -1); // start of the run this sample belongs to
		st                   = st < 0 ? carry : st;
		if (valid) {
			atomicAdd(&W.len[st], 1u);
			if (first) { atomicMin(&first[st], s_key[j] & 0xFFFFu); }
		}
		carry = __builtin_amdgcn_readlane(st, 63);
	}
	wave_lds_sync();
}

// The finished state goes to the column.  Stand-alone launches: a plain 32-byte store (the next kernel on the stream reads it).
// ASYNC: consumers on other CUs poll for it WHILE this kernel runs (load_rowgroup_state_async), so the words leave as agent-scope
// (write-through) stores and the word holding the tag — bytes 8..15, pad = kStateReady — leaves last, after the others (and, for
// ALP_RD, the rd_order table released in front of this call) are known to have reached memory: s_waitcnt vmcnt(0) in between
// (MI355X_MICROARCH.md, handoff-flag).  One lane.
template <bool ASYNC>
__device__ __forceinline__ void store_rowgroup_state(alpgpu_rowgroup_state* __restrict__ dst, const alpgpu_rowgroup_state& st) {
	if constexpr (!ASYNC) {
		*dst = st;
	} else {
		uint64_t w[4];
		__builtin_memcpy(w, &st, 32);
		w[1] = (w[1] & 0x00FFFFFFFFFFFFFFull) | (static_cast<uint64_t>(kStateReady) << 56);
		uint64_t* d64 = reinterpret_cast<uint64_t*>(dst);
		__hip_atomic_store(d64 + 0, w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store(d64 + 2, w[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store(d64 + 3, w[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__hip_atomic_store(d64 + 1, w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

// FROM_SAMPLES = false: `in` is the column, the kernel gathers the rowgroup's first-level sample itself.
// FROM_SAMPLES = true : `in` holds first-level samples gathered by the caller (what the reference's
// find_top_k_combinations / find_best_dictionary receive): workgroup b reads n_vectors (= its sample count, <= 288)
// doubles at in + 288*b.  Used by the per-rowgroup entry point behind include/alp.hpp, where the column may end in
// a partial vector and the sampler's index rules (sampler.hpp:29-44) are applied on the host.
//
// W = wavefronts per workgroup.  9 (one per sampled vector) for the stand-alone launches: one workgroup per rowgroup.
// ASYNC (W = 8): the PERSISTENT form that runs BESIDE the single-pass vector encode on a second stream — a grid of about one workgroup
// per CU walks the rowgroups rg_first + b, rg_first + b + G, ... in step and PUBLISHES every state for the encode tiles that poll for it
// (alp_device.hpp: load_rowgroup_state_async): state words as agent-scope write-through stores, the word that carries the tag (the pad
// byte = kStateReady) last, behind an s_waitcnt vmcnt(0); for an ALP_RD rowgroup its rd_order table is released first.  Eight
// wavefronts of 64 VGPRs are what fits a CU next to two encode tiles (2 x 8 wavefronts x 96 VGPRs, 2 x 66 KiB LDS): the search then
// runs in the issue slots the memory-bound encode leaves idle instead of 0.55 ms in front of it.
// Register budgets of the persistent forms: W = kInitAsyncWaves (4) beside two classic encode tiles: <= 96 VGPRs; W = kInitAsyncTileWaves (8) — a
// workgroup in the SHAPE of a lean encode tile (eight wavefronts, <= 80 VGPRs, 25 KiB of LDS): it takes one of a CU's three tile slots while the
// search lasts and gives it back afterwards, instead of keeping the third tile out for the whole encode the way a 96-register wavefront per
// SIMD would (6 x 80 + 96 > 512) — round 4: 3.13 -> 3.05 ms per 1 Mi vectors on the mixed column beside k_encode_lean.
constexpr int kInitAsyncTileWaves = 8;
template <bool ASYNC, int W, int BITS>
constexpr int init_waves_per_simd() { // (a float encode tile has 64 registers per lane: its tile-shaped search must fit 64 as well)
	return !ASYNC ? 1 : (W == kInitAsyncWaves ? kInitAsyncWavesPerSimd : (W == kInitAsyncTileWaves ? (BITS == 32 ? 8 : 6) : 1));
}
template <class P, bool FROM_SAMPLES, int W, bool ASYNC>
__global__ __launch_bounds__(64 * W, (init_waves_per_simd<ASYNC, W, P::kBits>())) void k_rowgroup_init(const typename P::value_t* __restrict__ in, uint64_t n_vectors,
                                                          alpgpu_rowgroup_state* __restrict__ rgs, int force_rd,
                                                          uint16_t* __restrict__ rd_order, uint64_t rg_first, uint64_t rg_end, double* __restrict__ cut_est_out,
                                                          uint32_t adaptive_base) {
	static_assert(!ASYNC || !FROM_SAMPLES, "the persistent form gathers its own samples");
	static_assert(W >= 2 && W <= kMaxSampledVectors, "the scratch arrays are sized for 9 wavefronts; wavefront 0 and the order replay use two of them");
	__shared__ typename P::value_t smp[kMaxSampledVectors * 32];
	__shared__ uint32_t best_key[kMaxSampledVectors];
	__shared__ uint32_t      s_key[kMaxSamples]; // (top 16 bits << 16 | sample index) of the samples, sorted (ALP_RD)
	// one scratch per wavefront for the cut search; afterwards wavefront 0 keeps its own and the order replay's arrays alias the others
	// (at least three of them).  With W = 4 that is 9 KiB instead of 21: the persistent form's whole LDS is 14 KiB, which leaves two 66 KiB
	// encode tiles room on the CU WHEREVER the allocator put it (measured with 26 KiB: placed between two tiles it kept the second tile
	// out for as long as it lived — the encode ran at half speed beside it).
	constexpr int kRdScratch = W > 4 ? W : 4;
#ifdef ALPGPU_EXPERIMENT_INIT_SMALL_LDS // residency experiment: no ALP_RD search, 3 KiB of LDS (ALP-only columns!)
	__shared__ RdWaveScratch s_rd[1];
#else
	__shared__ RdWaveScratch s_rd[kRdScratch];
#endif
	__shared__ uint32_t      s_first[kMaxSamples];
	__shared__ double        s_cut_est[17];
	__shared__ uint8_t       s_cut_ds[17];
	__shared__ int           s_best_cut;
	__shared__ int           s_scheme;
	__shared__ alpgpu_rowgroup_state s_state; // the state being built: written to the column (or published) when it is complete

	const int      lane    = lane_id();
	const int      wave    = wave_in_wg();
	const int      tid     = static_cast<int>(threadIdx.x);
	(void)rg_end;
	// How many workgroups walk the rowgroups (the persistent form only).  A column of ALP rowgroups wants ONE search workgroup per CU beside the
	// encode (it takes one of the CU's three tile slots for two thirds of the encode's time); a column of ALP_RD rowgroups, whose search is a
	// chain of dependent latencies the encode has to wait for anyway, wants all three (profiles/r04_encode_levers.txt, point 2: 3.02 / 4.58 ms with
	// one, 3.27 / 4.28 ms with three).  The launch brings three per CU; every workgroup looks at the states of the column's head (searched in
	// front of this kernel, on this stream: finished and visible) and those beyond `adaptive_base` leave unless ALP_RD is the majority there.
	uint32_t walkers = gridDim.x;
	if constexpr (ASYNC) {
		if (adaptive_base != 0 && adaptive_base < gridDim.x) { // (kernel argument: uniform)
			const bool     rd    = rgs[4 * lane].scheme == ALPGPU_SCHEME_ALP_RD; // 64 of the head's 256 rowgroups (api_encode.hip: kAsyncHeadRowgroups)
			const bool     heavy = __builtin_popcountll(ballot64(rd)) >= 32;
			walkers              = heavy ? gridDim.x : adaptive_base;
			if (blockIdx.x >= walkers) { return; }
		}
	}
	for (uint64_t rg = rg_first + blockIdx.x; ASYNC ? rg < rg_end : rg == rg_first + blockIdx.x; rg += ASYNC ? walkers : 1u << 30) {
	if (ASYNC) { __syncthreads(); } // the previous rowgroup's last readers of the shared arrays
	int n_sv, n_smp, samples_size;
	if constexpr (FROM_SAMPLES) {
		// encoder.hpp:140-143: ceil(n / 32) sampled "vectors" of min(n, 32) samples each
		n_smp        = static_cast<int>(n_vectors);
		n_sv         = (n_smp + 31) / 32;
		samples_size = n_smp < 32 ? n_smp : 32;
		for (int sv = wave; sv < n_sv; sv += W) {
			if (lane < 32) { smp[sv * 32 + lane] = in[288ull * blockIdx.x + sv * samples_size + (lane < samples_size ? lane : 0)]; }
		}
	} else {
		const uint64_t v_first = rg * kRowgroup;
		const int      nv      = static_cast<int>((n_vectors - v_first) < kRowgroup ? (n_vectors - v_first) : kRowgroup);
		n_sv                   = (nv + 11) / 12; // vectors with (index % 12) == 0, sampler.hpp:29-33
		n_smp                  = 32 * n_sv;
		samples_size           = 32;
		// first-level sample: values 32*s of each sampled vector (sampler.hpp:35-49 with full vectors)
		for (int sv = wave; sv < n_sv; sv += W) {
			if (lane < 32) { smp[sv * 32 + lane] = in[(v_first + 12ull * sv) * kVec + 32ull * lane]; }
		}
	}
	if (tid < kMaxSampledVectors) { best_key[tid] = 0xFFFFFFFFu; }
	__syncthreads();

	// ---- find_top_k_combinations: per sampled vector, arg-min over the 190 (e,f) candidates ----
	// (force_rd: rd_encoder::init called directly on these samples — the reference's rd.hpp:180-185 does not re-check
	//  the ALP threshold — so the search is skipped and every key stays "no valid candidate")
	// Work items = (sampled vector, round of 64 candidates) [+ (sampled vector, sideways tail) for float], dealt round-robin to the W
	// wavefronts; an item's best key goes into best_key[sampled vector] by an LDS atomic min (keys are unique: size << 8 | candidate).
	if (!force_rd) {
		// Float: 66 candidates are one full round of 64 and a tail of TWO — two lanes walking 32 samples while 62 idle, half of
		// the search's time.  The tail is turned on its side instead: half-wave h takes candidate 64 + h, its lanes one sample each;
		// one step, then the half-wave's count (a ballot) and min / max (DPP) — the same integers, reduced in another order.
		constexpr int  kTail       = P::kNumCombos % 64;
		constexpr bool kTailOnSide = P::kBits == 32 && kTail > 0 && kTail <= 2;
		constexpr int  kRounds     = kTailOnSide ? P::kNumCombos / 64 : (P::kNumCombos + 63) / 64;
		constexpr int  kItemsPerSv = kRounds + (kTailOnSide ? 1 : 0);
#pragma unroll 1
		for (int item = wave; item < n_sv * kItemsPerSv; item += W) {
			const int sv = item / kItemsPerSv;
			const int r  = item - sv * kItemsPerSv;
			uint32_t  my_key = 0xFFFFFFFFu;
			if (kTailOnSide && r == kRounds) { // wave-uniform
				const int  half   = lane >> 5, sidx = lane & 31;
				const int  c      = P::kNumCombos - kTail + (half < kTail ? half : 0);
				const bool active = half < kTail && sidx < samples_size;
				const typename P::Coef k = P::coef(P::combos().e[c], P::combos().f[c]);
				const float   v  = static_cast<float>(smp[sv * 32 + (sidx < samples_size ? sidx : 0)]);
				const int32_t q  = encode_value_f32(v, k.exp10, k.frac_f);
				const bool    ok = active & (decode_value_f32(q, k.fact, k.frac_e) == v);
				const uint64_t bal     = ballot64(ok);
				const int      non_exc = __builtin_popcount(static_cast<uint32_t>(half ? (bal >> 32) : bal));
				int32_t        mx = ok ? q : INT32_MIN, mn = ok ? q : INT32_MAX;
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
				ALPGPU_MINMAX_STEP(0x142, 0xa) // row_bcast:15 into rows 1 and 3: lanes 31 and 63 hold their half's result
#undef ALPGPU_MINMAX_STEP
				if (sidx == 31 && half < kTail && non_exc >= 2) { // encoder.hpp:182
					const uint32_t size = static_cast<uint32_t>(samples_size) * static_cast<uint32_t>(P::bits(mx, mn)) +
					                      static_cast<uint32_t>(samples_size - non_exc) * (P::kExcBits + 16u);
					my_key              = (size << 8) | static_cast<uint32_t>(c);
				}
			} else {
				const int c = lane + 64 * r;
				if (c < P::kNumCombos) {
					const typename P::Coef k = P::coef(P::combos().e[c], P::combos().f[c]);
					typename P::Acc        acc;
					P::start(acc);
					if constexpr (P::kBits == 64) {
						int s = 0; // four samples per trip, spelled out: the unroller declines this body (a rarely taken branch inside)
						for (; s + 4 <= samples_size; s += 4) {
							P::step(acc, smp[sv * 32 + s], k);
							P::step(acc, smp[sv * 32 + s + 1], k);
							P::step(acc, smp[sv * 32 + s + 2], k);
							P::step(acc, smp[sv * 32 + s + 3], k);
						}
						for (; s < samples_size; ++s) { P::step(acc, smp[sv * 32 + s], k); }
					} else {
#pragma unroll 4
						for (int s = 0; s < samples_size; ++s) { P::step(acc, smp[sv * 32 + s], k); }
					}
					P::finish(acc);
					const int     non_exc = acc.non_exc;
					const int64_t mx = acc.mx, mn = acc.mn;
					if (non_exc >= 2) { // encoder.hpp:182
						const uint32_t size = static_cast<uint32_t>(samples_size) * static_cast<uint32_t>(P::bits(mx, mn)) +
						                      static_cast<uint32_t>(samples_size - non_exc) * (P::kExcBits + 16u);
						my_key              = (size << 8) | static_cast<uint32_t>(c);
					}
				}
			}
			my_key = wave_min_u32(my_key);
			if (lane == 0 && my_key != 0xFFFFFFFFu) { atomicMin(&best_key[sv], my_key); }
		}
	}
	__syncthreads();

	// ---- vote, scheme decision, top-k (encoder.hpp:207-234) ----
	if (tid == 0) {
		uint32_t best_size = static_cast<uint32_t>(samples_size) * (P::kExcBits + 16u) + static_cast<uint32_t>(samples_size) * P::kExcBits; // encoder.hpp:147-149
		int      ce[kMaxSampledVectors], cf[kMaxSampledVectors], cn[kMaxSampledVectors];
		int      n_c = 0;
		for (int w = 0; w < n_sv; ++w) {
			int e = 0, f = 0;
			if (best_key[w] != 0xFFFFFFFFu) {
				const int      c    = static_cast<int>(best_key[w] & 0xFFu);
				const uint32_t size = best_key[w] >> 8;
				e                   = P::combos().e[c];
				f                   = P::combos().f[c];
				best_size           = size < best_size ? size : best_size;
			}
			int hit = -1;
			for (int i = 0; i < n_c; ++i) {
				if (ce[i] == e && cf[i] == f) { hit = i; }
			}
			if (hit < 0) {
				ce[n_c] = e, cf[n_c] = f, cn[n_c] = 1;
				++n_c;
			} else {
				++cn[hit];
			}
		}
		alpgpu_rowgroup_state st;
		st.k = 0;
		for (int i = 0; i < 10; ++i) { st.combos[i] = 0; }
		st.rd_rbw = st.rd_lbw = st.rd_dict_size = st.pad = 0;
		for (int i = 0; i < 8; ++i) { st.rd_dict[i] = 0; }
		if (force_rd || best_size >= P::kRdThreshold) { // RD_SIZE_THRESHOLD_LIMIT, encoder.hpp:213-216
			st.scheme = ALPGPU_SCHEME_ALP_RD;
		} else {
			st.scheme = ALPGPU_SCHEME_ALP;
			// insertion sort by (count desc, e desc, f desc): a strict total order (encoder.hpp:128-132)
			for (int i = 1; i < n_c; ++i) {
				const int e = ce[i], f = cf[i], n = cn[i];
				int       j = i - 1;
				while (j >= 0 && (n > cn[j] || (n == cn[j] && e > ce[j]) || (n == cn[j] && e == ce[j] && f > cf[j]))) {
					ce[j + 1] = ce[j], cf[j + 1] = cf[j], cn[j + 1] = cn[j];
					--j;
				}
				ce[j + 1] = e, cf[j + 1] = f, cn[j + 1] = n;
			}
			const int k = n_c < 5 ? n_c : 5;
			st.k        = static_cast<uint8_t>(k);
#pragma unroll
			for (int i = 0; i < 5; ++i) { // constant indices: `st` stays in registers (a dynamically indexed local is promoted to 18 KiB of LDS)
				st.combos[2 * i]     = i < k ? static_cast<uint8_t>(ce[i]) : 0;
				st.combos[2 * i + 1] = i < k ? static_cast<uint8_t>(cf[i]) : 0;
			}
		}
		s_scheme = st.scheme;
		if (st.scheme != ALPGPU_SCHEME_ALP_RD) {
			store_rowgroup_state<ASYNC>(rgs + rg, st);                                     // complete: an ALP rowgroup needs nothing more
			if (rd_order) { rd_order[rg * ALPGPU_RD_ORDER_STRIDE] = 0; }                   // no table for an ALP rowgroup (nobody reads it during the encode)
		} else {
			s_state = st; // completed by wavefront 0 below
		}
	}
	__syncthreads();
	if (s_scheme != ALPGPU_SCHEME_ALP_RD) { continue; }
#ifdef ALPGPU_EXPERIMENT_INIT_SMALL_LDS
	continue;
#endif

	// ---- ALP_RD: find the cut and the dictionary (rd.hpp:89-104, :33-87) ----
	// The samples are sorted ONCE (by their top 16 bits, see rd_left_of); for every cut position the equal left parts are then
	// contiguous runs of the sorted order.  One wavefront evaluates one cut at a time with wave-level primitives
	// only: run starts (max-scan), run lengths and first occurrences (LDS atomics), a histogram of run lengths and
	// a descending scan of it for the mass of the 8 most frequent left parts.  The size estimate does not depend on how
	// equally frequent left parts are ordered; the dictionary does, and is built afterwards for the chosen cut only, in the
	// reference's (libstdc++'s) order — rd_dictionary_order.hpp.
	// (sample t sits at smp[32 * (t / samples_size) + t % samples_size]; with 32-sample blocks that is smp[t])
	// Rank sort: the composite keys go to LDS in sample order first (W.* of the waves is free until the cut search); every
	// thread then counts the smaller keys, four broadcast keys per LDS read.
	uint32_t* s_unsorted = reinterpret_cast<uint32_t*>(&s_rd[0]); // 288 * 4 B <= sizeof(RdWaveScratch)
	static_assert(sizeof(RdWaveScratch) >= kMaxSamples * sizeof(uint32_t), "scratch reuse");
	static_assert(kMaxSamples % 4 == 0, "the rank loop reads four keys at a time");
	constexpr int kSortSlots = (kMaxSamples + 64 * W - 1) / (64 * W); // samples per thread (1 with >= 5 wavefronts, 2 with 4)
	for (int t = tid; t < kMaxSamples; t += 64 * W) {
		uint32_t c = 0xFFFFFFFFu; // padding sorts behind every sample
		if (t < n_smp) { c = (static_cast<uint32_t>(P::pattern(smp[32 * (t / samples_size) + (t % samples_size)]) >> (P::kBits - 16)) << 16) | static_cast<uint32_t>(t); }
		s_unsorted[t] = c;
	}
	__syncthreads();
	uint32_t my_key[kSortSlots];
	int      my_rank[kSortSlots];
#pragma unroll
	for (int u = 0; u < kSortSlots; ++u) {
		const int t = tid + u * 64 * W;
		my_key[u]   = 0;
		my_rank[u]  = 0;
		if (t < n_smp) {
			my_key[u]          = s_unsorted[t];
			const u32x4* quads = reinterpret_cast<const u32x4*>(s_unsorted);
			const int    n_q   = (n_smp + 3) / 4;
			for (int q = 0; q < n_q; ++q) {
				const u32x4 k = quads[q];
				my_rank[u] += (k[0] < my_key[u] ? 1 : 0) + (k[1] < my_key[u] ? 1 : 0) + (k[2] < my_key[u] ? 1 : 0) + (k[3] < my_key[u] ? 1 : 0);
			}
		}
	}
	__syncthreads(); // everybody is done reading s_unsorted (it aliases wave 0's scratch)
#pragma unroll
	for (int u = 0; u < kSortSlots; ++u) {
		if (tid + u * 64 * W < n_smp) { s_key[my_rank[u]] = my_key[u]; }
	}
	__syncthreads();

#ifdef ALPGPU_EXPERIMENT_RD_STOP_AFTER_SORT // timing experiments (profiles/r04_rd_search.txt): where the ALP_RD half of the search spends its time
	continue;
#endif
	RdWaveScratch& WS = s_rd[wave];
	// force_rd = 0x100 | cut: rd_encoder::build_left_parts_dictionary for ONE cut position (rd.hpp:33-87, called on its own): the other cuts
	// get an estimate nothing beats, so the dictionary below is that cut's
#ifdef ALPGPU_EXPERIMENT_RD_ONE_CUT
	const int forced_cut = 12;
#else
	const int forced_cut = (force_rd & 0x100) ? (force_rd & 0xFF) : 0;
#endif
	for (int cut = wave + 1; cut <= 16; cut += W) { // wave-uniform
		const int rbw = P::kBits - cut;
		if (forced_cut != 0 && cut != forced_cut) {
			if (lane == 0) {
				s_cut_est[cut] = 1.7976931348623157e308;
				s_cut_ds[cut]  = 0;
			}
			continue;
		}
		rd_build_runs(WS, nullptr, s_key, n_smp, cut, lane);
		// histogram of run lengths; number of distinct left parts
		int distinct = 0;
		for (int b = 0; b < n_smp; b += 64) {
			const int      j = b + lane;
			const uint32_t L = j < n_smp ? WS.len[j] : 0u;
			if (L) { atomicAdd(&WS.hist[L], 1u); }
			distinct += __builtin_popcountll(__ballot(L != 0));
		}
		wave_lds_sync();
		// mass of the 8 longest runs: walk run lengths from n_smp downwards
		uint32_t covered = 0;
		int      taken   = 0;
		for (int top = n_smp; top >= 1 && taken < 8; top -= 64) {
			const int      cval = top - lane;
			const uint32_t h    = cval >= 1 ? WS.hist[cval] : 0u;
			const uint32_t inc    = wave_scan_add_u32(h);
			const uint32_t before = static_cast<uint32_t>(taken) + inc - h;
			const uint32_t after  = static_cast<uint32_t>(taken) + inc;
			const uint32_t mine   = (after < 8u ? after : 8u) - (before < 8u ? before : 8u);
			const uint32_t part   = wave_scan_add_u32(mine * static_cast<uint32_t>(cval > 0 ? cval : 0));
			covered += static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(part), 63));
			taken += __builtin_amdgcn_readlane(static_cast<int>(inc), 63);
		}
		const uint32_t excs = static_cast<uint32_t>(n_smp) - covered;
		const int      ds   = distinct < 8 ? distinct : 8;
		const int      lbw  = ds <= 2 ? 1 : (ds <= 4 ? 2 : 3); // max(1, ceil(log2(ds)))
		if (lane == 0) {
			s_cut_est[cut] = static_cast<double>(rbw + lbw) + static_cast<double>(excs * 32u) / static_cast<double>(n_smp);
			s_cut_ds[cut]  = static_cast<uint8_t>(ds);
		}
		wave_lds_sync();
	}
	__syncthreads();
	if (tid == 0) { // first strictly smaller estimate in cut order 1..16 (rd.hpp:95-101)
		double best = 1.7976931348623157e308;
		int    bc   = 1;
		for (int cut = 1; cut <= 16; ++cut) {
			if (s_cut_est[cut] < best) {
				best = s_cut_est[cut];
				bc   = cut;
			}
		}
		s_best_cut = bc;
		if (cut_est_out != nullptr) { cut_est_out[blockIdx.x] = best; } // estimate_compression_size of the chosen (or forced) cut (rd.hpp:80-82)
	}
	__syncthreads();
	if (wave == 0) { // the dictionary of the chosen cut: the (<= 8) best-ranked runs
		// (the other wavefronts are past their cut searches: their scratch now holds the order replay's arrays, which keeps the
		//  workgroup's LDS small enough for six of them per CU — this single-wavefront tail then overlaps other rowgroups' searches)
		static_assert(sizeof(RdOrderLds) <= (kRdScratch - 1) * sizeof(RdWaveScratch), "order replay arrays alias s_rd[1..]");
		RdOrderLds& s_order = *reinterpret_cast<RdOrderLds*>(&s_rd[1]);
		const int best_cut = s_best_cut;
		const int rbw      = P::kBits - best_cut;
		const int ds       = s_cut_ds[best_cut];
		rd_build_runs(WS, s_first, s_key, n_smp, best_cut, lane);
		// distinct left parts in order of first occurrence in the sample, with their counts
		// (run starts are compacted into s_order.sorted[] = first occurrence, so the order loop walks D runs, not 288 samples)
		int distinct = 0;
		for (int b = 0; b < n_smp; b += 64) {
			const int      j    = b + lane;
			const bool     head = j < n_smp && WS.len[j] != 0;
			const uint64_t hb   = __ballot(head);
			if (head) { s_order.sorted[distinct + __builtin_popcountll(hb & ((1ull << lane) - 1ull))] = s_first[j]; }
			distinct += __builtin_popcountll(hb);
		}
		wave_lds_sync();
		for (int b = 0; b < n_smp; b += 64) {
			const int      j   = b + lane;
			const uint32_t L   = j < n_smp ? WS.len[j] : 0u;
			const uint32_t fo  = j < n_smp ? s_first[j] : 0u;
			int            ord = 0;
			for (int g = 0; g < distinct; ++g) { ord += s_order.sorted[g] < fo ? 1 : 0; } // broadcast reads
			if (L != 0) {
				s_order.okey[ord] = rd_left_of(s_key[j], best_cut);
				s_order.ocnt[ord] = L;
			}
		}
		wave_lds_sync();
		// the reference's order of equally frequent left parts is libstdc++'s (rd_dictionary_order.hpp): replayed by one lane
#ifndef ALPGPU_EXPERIMENT_SKIP_RD_ORDER
		if (lane == 0) { rd_reference_order(s_order, distinct); }
#endif
		wave_lds_sync();
		if (rd_order) { // the whole sorted order, for the encoders' exception-slot indices (alpgpu_column.d_rd_order)
			uint16_t* o = rd_order + rg * ALPGPU_RD_ORDER_STRIDE;
			for (int i = lane; i < distinct; i += 64) { o[1 + i] = static_cast<uint16_t>(s_order.sorted[i] & 0xFFFFu); }
			if (lane == 0) { o[0] = static_cast<uint16_t>(distinct); }
		}
		if constexpr (ASYNC) { // the table's (plain) stores of all lanes reach memory before the state that announces them
			if (rd_order) {
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			}
		}
		if (lane == 0) {
			const int             lbw = ds <= 2 ? 1 : (ds <= 4 ? 2 : 3);
			alpgpu_rowgroup_state st  = s_state;
			st.rd_rbw                 = static_cast<uint8_t>(rbw);
			st.rd_lbw                 = static_cast<uint8_t>(lbw);
			st.rd_dict_size           = static_cast<uint8_t>(ds);
#pragma unroll
			for (int i = 0; i < 8; ++i) { st.rd_dict[i] = i < ds ? static_cast<uint16_t>(s_order.sorted[i] & 0xFFFFu) : 0; }
			store_rowgroup_state<ASYNC>(rgs + rg, st);
		}
	}
	} // rowgroups of this workgroup
}

// rowgroups [rg_first, rg_first + rg_count) of a column of n_vectors vectors; rg_count = 0 means "to the end"
int launch_rowgroup_init(hipStream_t stream, const double* d_in, uint64_t n_vectors, alpgpu_rowgroup_state* d_rgs, uint16_t* d_rd_order, uint64_t rg_first,
                         uint64_t rg_count) {
	const uint64_t n_rg = (n_vectors + kRowgroup - 1) / kRowgroup;
	if (rg_first >= n_rg) { return ALPGPU_OK; }
	if (rg_count == 0 || rg_first + rg_count > n_rg) { rg_count = n_rg - rg_first; }
	// (4-wavefront workgroups, six of them per CU instead of three of nine wavefronts, take exactly as long — 0.578 ms per 1 Mi vectors, 1.00 ms all-ALP_RD:
	//  the search is bound by its arithmetic, not by its fixed latencies; profiles/r03_async_init.txt)
	hipLaunchKernelGGL((k_rowgroup_init<PrecF64, false, kMaxSampledVectors, false>), dim3(static_cast<unsigned>(rg_count)), dim3(kInitThreads), 0, stream, d_in,
	                   n_vectors, d_rgs, 0, d_rd_order, rg_first, rg_first + rg_count, static_cast<double*>(nullptr), 0u);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

// The persistent, publishing form for rowgroups [rg_first, rg_first + rg_count): `grid` workgroups of kInitAsyncWaves wavefronts.
template <class P>
static int launch_rowgroup_init_async_t(hipStream_t stream, const typename P::value_t* d_in, uint64_t n_vectors, alpgpu_rowgroup_state* d_rgs, uint16_t* d_rd_order,
                                        uint64_t rg_first, uint64_t rg_count, int grid, bool tile_shaped, uint32_t adaptive_base = 0) {
	if (rg_count == 0) { return ALPGPU_OK; }
	const uint64_t g = rg_count < static_cast<uint64_t>(grid) ? rg_count : static_cast<uint64_t>(grid);
	if (g == rg_count) { // one workgroup per rowgroup (the head in front of the encode): the 9-wavefront shape, publishing
		hipLaunchKernelGGL((k_rowgroup_init<P, false, kMaxSampledVectors, true>), dim3(static_cast<unsigned>(g)), dim3(kInitThreads), 0, stream, d_in, n_vectors, d_rgs, 0,
		                   d_rd_order, rg_first, rg_first + rg_count, static_cast<double*>(nullptr), 0u);
	} else if (tile_shaped) {
		hipLaunchKernelGGL((k_rowgroup_init<P, false, kInitAsyncTileWaves, true>), dim3(static_cast<unsigned>(g)), dim3(64 * kInitAsyncTileWaves), 0, stream, d_in, n_vectors,
		                   d_rgs, 0, d_rd_order, rg_first, rg_first + rg_count, static_cast<double*>(nullptr), adaptive_base);
	} else {
		hipLaunchKernelGGL((k_rowgroup_init<P, false, kInitAsyncWaves, true>), dim3(static_cast<unsigned>(g)), dim3(64 * kInitAsyncWaves), 0, stream, d_in, n_vectors, d_rgs,
		                   0, d_rd_order, rg_first, rg_first + rg_count, static_cast<double*>(nullptr), adaptive_base);
	}
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}
int launch_rowgroup_init_async(hipStream_t stream, const double* d_in, uint64_t n_vectors, alpgpu_rowgroup_state* d_rgs, uint16_t* d_rd_order, uint64_t rg_first,
                               uint64_t rg_count, int grid, bool tile_shaped, uint32_t adaptive_base) {
	return launch_rowgroup_init_async_t<PrecF64>(stream, d_in, n_vectors, d_rgs, d_rd_order, rg_first, rg_count, grid, tile_shaped, adaptive_base);
}
int launch_rowgroup_init_async_f32(hipStream_t stream, const float* d_in, uint64_t n_vectors, alpgpu_rowgroup_state* d_rgs, uint16_t* d_rd_order, uint64_t rg_first,
                                   uint64_t rg_count, int grid) {
	static const bool tile = std::getenv("ALPGPU_F32_SEARCH_TILE") != nullptr; // A/B: the eight-wavefront (64-register, spilling) form instead of four wavefronts of 96
	return launch_rowgroup_init_async_t<PrecF32>(stream, d_in, n_vectors, d_rgs, d_rd_order, rg_first, rg_count, grid, tile);
}

int launch_state_from_samples(hipStream_t stream, const double* d_samples, uint32_t n_samples, alpgpu_rowgroup_state* d_state, int force_rd, double* d_cut_estimate) {
	hipLaunchKernelGGL((k_rowgroup_init<PrecF64, true, kMaxSampledVectors, false>), dim3(1), dim3(kInitThreads), 0, stream, d_samples, static_cast<uint64_t>(n_samples),
	                   d_state, force_rd, static_cast<uint16_t*>(nullptr), 0ull, 1ull, d_cut_estimate, 0u);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

// single precision: alp::encoder<float>::init / alp::rd_encoder<float>::init
int launch_rowgroup_init_f32(hipStream_t stream, const float* d_in, uint64_t n_vectors, alpgpu_rowgroup_state* d_rgs, uint16_t* d_rd_order,
                             uint64_t rg_first, uint64_t rg_count) {
	const uint64_t n_rg = (n_vectors + kRowgroup - 1) / kRowgroup;
	if (rg_first >= n_rg) { return ALPGPU_OK; }
	if (rg_count == 0 || rg_first + rg_count > n_rg) { rg_count = n_rg - rg_first; }
	hipLaunchKernelGGL((k_rowgroup_init<PrecF32, false, kMaxSampledVectors, false>), dim3(static_cast<unsigned>(rg_count)), dim3(kInitThreads), 0, stream, d_in,
	                   n_vectors, d_rgs, 0, d_rd_order, rg_first, rg_first + rg_count, static_cast<double*>(nullptr), 0u);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

int launch_state_from_samples_f32(hipStream_t stream, const float* d_samples, uint32_t n_samples, alpgpu_rowgroup_state* d_state, int force_rd, double* d_cut_estimate) {
	hipLaunchKernelGGL((k_rowgroup_init<PrecF32, true, kMaxSampledVectors, false>), dim3(1), dim3(kInitThreads), 0, stream, d_samples, static_cast<uint64_t>(n_samples),
	                   d_state, force_rd, static_cast<uint16_t*>(nullptr), 0ull, 1ull, d_cut_estimate, 0u);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

} // namespace alpgpu
