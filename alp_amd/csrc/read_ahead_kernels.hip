// read_ahead_kernels.hip — the store decode's READ-AHEAD (round 5): a small persistent kernel on the context's second stream that reads the column's
// descriptors, packed words and exception records a bounded distance AHEAD of the decode kernel, so that the decode's two dependent reads (descriptor,
// then the words it points at) are served by the 256 MiB Infinity Cache instead of HBM.
//
// Why (tools/r05_mall_warm.py, profiles/r05_read_ahead.txt): a decode workgroup's life is two dependent HBM round trips under a write-dominated stream,
// then 8 KiB of stores.  With its inputs already in the Infinity Cache the SAME kernel runs at 0.90-0.94 of the HBM peak on 2-12-bit vectors (0.53-0.79
// cold), with 20 exceptions per vector at 0.88-0.95 (0.53-0.78 cold); a bulk read of the streams in front of the launch gives the same figures as a
// previous launch does.  The bytes still cross the HBM interface once — as long independent bursts of a kernel nobody waits for.
//
// Pacing: the decode kernel's workgroups are dispatched in ascending order; every 128th stores the index of its first vector to one word of context
// memory (decode_kernels.hip: k_decode_column, `progress`).  The read-ahead's wavefronts take batches of 64 vectors round-robin and keep a batch within
// [progress + lead_min, progress + lead_max): not so far ahead that the cache has dropped the lines again, not behind the decode.  The word carries
// the launch's tag in its top bits, so a stale value of an earlier decode reads as "not started".  Best effort by construction: a read-ahead that
// leaves early or never runs changes nothing but the decode's speed.
#include "decode_policy.hpp"   // the words of context memory an unhinted decode's plan arrives in
#include "encode_lookback.hpp" // status_load, record_sizes
#include "launch.hpp"

#include <cstdlib>

namespace alpgpu {

#ifndef ALPGPU_AHEAD_WAVES
#define ALPGPU_AHEAD_WAVES 8
#endif
constexpr int      kAheadWaves  = ALPGPU_AHEAD_WAVES; // wavefronts per read-ahead workgroup (few, fat workgroups: one lane of each polls)
constexpr int      kAheadUnroll = 16;       // 16-byte loads in flight per lane and round (16 KiB per wavefront)
constexpr uint64_t kAheadVecMask = (1ull << 40) - 1ull;
// Patience (round 6): how long a workgroup goes without news from the decode before it leaves.  Until round 5 a flat 50 ms — wherever
// the two kernels cannot run side by side (one hardware queue, AMD_SERIALIZE_KERNEL, a --pmc pass, a debugger) every decode of a narrow column then cost 50 ms.  Now
// from the column: the launcher passes the FIRST word's patience (the decode kernel is enqueued right behind this one: 200 us, or a quarter of the decode's own
// estimated duration if that is longer) and the steady state's (twice that estimate, at least 200 us) in ticks of wall_clock64().

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
	for (int s = 1; s < 64; s <<= 1) {
		const uint64_t o = __shfl_xor(v, s);
		v                = o < v ? o : v;
	}
	return v;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
	for (int s = 1; s < 64; s <<= 1) {
		const uint64_t o = __shfl_xor(v, s);
		v                = o > v ? o : v;
	}
	return v;
}
__device__ __forceinline__ uint64_t wave_add_u64(uint64_t v) {
#pragma unroll
	for (int s = 1; s < 64; s <<= 1) { v += __shfl_xor(v, s); }
	return v;
}

// the bytes [begin, end) of a stream, 16 per lane and load, kAheadUnroll loads in flight; returns something that depends on every word
__device__ __forceinline__ uint32_t touch_span(const uint8_t* __restrict__ stream, uint64_t begin, uint64_t end, int lane) {
	const uint64_t b = begin & ~15ull, e = end & ~15ull;
	const uint4*   p = reinterpret_cast<const uint4*>(stream + b);
	const uint64_t n = (e - b) >> 4;
	uint32_t       x = 0;
	for (uint64_t i = static_cast<uint64_t>(lane); i < n; i += 64ull * kAheadUnroll) {
		uint4 r[kAheadUnroll];
#pragma unroll
		for (int k = 0; k < kAheadUnroll; ++k) {
			const uint64_t j = i + 64ull * k;
			r[k]             = j < n ? p[j] : uint4 {0u, 0u, 0u, 0u};
		}
#pragma unroll
		for (int k = 0; k < kAheadUnroll; ++k) { x ^= r[k].x ^ r[k].w; }
	}
	return x;
}

// mode (experiments, ALPGPU_READ_AHEAD_MODE): bit 0 = pace only, read nothing; bit 1 = read without pacing; bit 2 = sleep ~1.5 ms and leave; bit 4 = the lead stays as given
// mode bit 3 (set by the launcher, not an experiment): the lead, the pace and the widest record come from the context's plan words (decode_policy.hpp: an UNHINTED
// decode, whose sizes were summed on this stream a moment ago; a lead of 0 there = this column gets no read-ahead: leave)
template <int VALUE_BYTES>
__global__ __launch_bounds__(64 * kAheadWaves) void k_read_ahead(const alpgpu_vector_desc* __restrict__ descs, const uint8_t* __restrict__ packed,
                                                                const uint8_t* __restrict__ excs, uint64_t n_vectors, uint64_t* __restrict__ ctx_words,
                                                                uint64_t tag, uint32_t lead_min, uint32_t lead_max, uint32_t ps_per_vector, uint32_t ps_per_tick, uint32_t max_bits, uint32_t mode,
                                                                uint32_t patience_first_ticks, uint32_t patience_ticks, uint32_t burst_ticks) {
	// One workgroup = kAheadWaves consecutive batches of 64 vectors per round; ONE lane of the workgroup reads the progress word, and while the round is
	// out of reach it does so every ~7 us only: the word lives on one memory channel, and every poll of every waiting wavefront is a trip to it.
	__shared__ uint64_t s_seen;
	__shared__ uint32_t s_go;
	__shared__ uint32_t s_lead;
	const uint64_t* progress = ctx_words + kCtxWordProgress;
	uint32_t*       hole     = reinterpret_cast<uint32_t*>(ctx_words + kCtxWordHole);
	const int      lane  = lane_id();
	const int      wave  = wave_in_wg();
	if (mode & 4u) { // experiment: resident for ~1.5 ms, touching nothing
		for (int k = 0; k < 440; ++k) { __builtin_amdgcn_s_sleep(127); }
		return;
	}
	if (mode & 8u) { // (uniform: kernel argument)
		const uint64_t lead = ctx_words[kCtxWordLead], pace = ctx_words[kCtxWordPace];
		lead_min = static_cast<uint32_t>(lead), lead_max = static_cast<uint32_t>(lead >> 32);
		ps_per_vector = static_cast<uint32_t>(pace), max_bits = static_cast<uint32_t>(pace >> 32);
		if (lead_max == 0 || ps_per_vector == 0) { return; }
		// (patience of an unhinted launch: the launcher does not know the decode's duration either — from the pace just read)
		const uint64_t est_ticks = n_vectors * static_cast<uint64_t>(ps_per_vector) / ps_per_tick;
		const uint64_t floor_t   = 200000000ull / ps_per_tick; // 200 us
		const uint64_t first = est_ticks / 4 > floor_t ? est_ticks / 4 : floor_t, steady = 2 * est_ticks > floor_t ? 2 * est_ticks : floor_t;
		patience_first_ticks = first > 0xFFFFFFFFull ? 0xFFFFFFFFu : static_cast<uint32_t>(first);
		patience_ticks       = steady > 0xFFFFFFFFull ? 0xFFFFFFFFu : static_cast<uint32_t>(steady);
	}
	const uint64_t round = static_cast<uint64_t>(gridDim.x) * kAheadWaves * 64;
	uint32_t       x     = 0;
	uint64_t       seen  = 0; // the decode's position as last read: first vector of a workgroup that has been dispatched
	// The lead stretches itself (round 6): a workgroup whose batch was asked for with less than an eighth of the lead to spare — the decode was nearly there
	// when the reads went out, the cliff of profiles/r05_read_ahead.txt, whose place moves from box to box — goes a quarter further ahead from its next
	// round on, up to three times the lead it was given.  Too long a lead decays slowly, too short a one falls off: this errs to the safe side.
	uint32_t       lead  = lead_max;
	for (uint64_t wg_first = static_cast<uint64_t>(blockIdx.x) * kAheadWaves * 64; wg_first < n_vectors; wg_first += round) {
		if (!(mode & 2u)) {
			if (threadIdx.x == 0) {
				// Sleep until the decode is lead vectors short of this round, by the clock: the decode advances at most one vector per ps_per_vector
				// (the host's estimate at the full HBM rate, so the sleep never overshoots), 0.9 of the way per nap, one poll per nap.
				uint32_t go      = 1;
				uint64_t changed = wall_clock64();
				for (;;) {
					const uint64_t w = status_load(progress);
					if ((w & ~kAheadVecMask) == tag && (w & kAheadVecMask) > seen) { seen = w & kAheadVecMask, changed = wall_clock64(); }
					if (wg_first < seen + lead) { break; }                          // in reach
					if (wall_clock64() - changed > (seen == 0 ? patience_first_ticks : patience_ticks)) { go = 0; break; } // the decode is not coming (it cannot run beside this kernel, or its launch failed): leave
					const uint64_t togo  = wg_first - (seen + lead) + 1;                       // vectors
					uint64_t       ticks = (togo * ps_per_vector / ps_per_tick) * 9ull / 10ull;        // ticks of wall_clock64() (100 MHz on MI355X; the host asks the runtime)
					const uint64_t t_min = 1000000ull / ps_per_tick, t_max = 200000000ull / ps_per_tick; // 1 us .. 200 us
					ticks                = ticks < t_min ? t_min : (ticks > t_max ? t_max : ticks);
					if (seen == 0 && ticks > patience_first_ticks / 2) { ticks = patience_first_ticks / 2 + 1; } // (before the first word: never sleep past the patience)
					const uint64_t until = wall_clock64() + ticks;
					while (wall_clock64() < until) { __builtin_amdgcn_s_sleep(32); }
				}
				// experiment (ALPGPU_READ_AHEAD_BURST_US): every workgroup's reads go out on the same beat of the device's wall clock — the memory controllers see reads in
				// bursts between the decode's writes instead of one here, one there (each of which turns a channel around on its own)
				if (go && burst_ticks != 0) {
					while (wall_clock64() % burst_ticks > burst_ticks / 8) { __builtin_amdgcn_s_sleep(8); }
				}
				if (go && !(mode & 16u) && seen != 0 && wg_first < seen + lead - lead / 8 * 7 && lead < 3 * lead_max) { lead += lead_max / 4; } // late: less than an eighth of the lead to spare
				s_seen = seen, s_go = go, s_lead = lead;
			}
			__syncthreads();
			seen                = s_seen;
			const uint32_t go   = s_go;
			lead                = s_lead;
			__syncthreads();
			if (!go) { break; }
			if (wg_first + kAheadWaves * 64 <= seen + lead_min) { continue; } // the decode is already there
		}
		if (threadIdx.x == 0) { __hip_atomic_fetch_add(ctx_words + kCtxWordBatches, static_cast<uint64_t>(kAheadWaves), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } // (debug counter; nobody waits for it)
		const uint64_t first = wg_first + static_cast<uint64_t>(wave) * 64;
		if (first >= n_vectors || (mode & 1u)) { continue; }
		const uint64_t     v = first + lane;
		alpgpu_vector_desc d = descs[v < n_vectors ? v : n_vectors - 1];
		uint64_t           pb, eb;
		record_sizes<VALUE_BYTES>(d, pb, eb);
		// max_bits: only vectors of at most this many packed bits per value are worth it (a column that mixes widths: the narrow vectors' two dependent reads are what
		// the decode waits for; the wide ones' bytes would only cross the fabric twice).  The descriptors themselves — the first of the two reads — are read for all.
		if (v >= n_vectors || pb > 128ull * max_bits) { pb = eb = 0; }
		x ^= static_cast<uint32_t>(d.base);
		// a column written in vector order: the batch's records are one span of each stream.  Otherwise (ALPGPU_OPT_ENCODE_UNORDERED; a blob stitched by hand)
		// record by record.
		const uint64_t p_lo = wave_min_u64(pb ? d.packed_off : ~0ull), p_hi = wave_max_u64(pb ? d.packed_off + pb : 0ull), p_sum = wave_add_u64(pb);
		const uint64_t e_lo = wave_min_u64(eb ? d.exc_off : ~0ull), e_hi = wave_max_u64(eb ? d.exc_off + eb : 0ull), e_sum = wave_add_u64(eb);
		if (p_sum != 0) {
			if (p_hi - p_lo <= 2 * p_sum) {
				x ^= touch_span(packed, p_lo, p_hi, lane);
			} else {
				for (int j = 0; j < 64; ++j) {
					const uint64_t o = __shfl(d.packed_off, j), s = __shfl(pb, j);
					if (s) { x ^= touch_span(packed, o, o + s, lane); }
				}
			}
		}
		if (e_sum != 0) {
			if (e_hi - e_lo <= 2 * e_sum + 4096) {
				x ^= touch_span(excs, e_lo, e_hi, lane);
			} else {
				for (int j = 0; j < 64; ++j) {
					const uint64_t o = __shfl(d.exc_off, j), s = __shfl(eb, j);
					if (s) { x ^= touch_span(excs, o, o + s, lane); }
				}
			}
		}
	}
	if (x == 0x9E3779B9u && n_vectors == ~0ull) { *hole = x; } // (never: what keeps the loads)
}

// lead_min / lead_max in vectors; value_bytes 8 (double column) or 4 (float column); from_plan: lead, pace and widest record from the context's plan words
int launch_read_ahead(hipStream_t stream, const alpgpu_column* col, int value_bytes, uint64_t* d_ctx_words, uint64_t tag, uint32_t lead_min, uint32_t lead_max,
                      uint32_t ps_per_vector, uint32_t ps_per_tick, uint32_t max_bits, int grid, bool from_plan) {
	if (col->n_vectors == 0 || grid <= 0) { return ALPGPU_OK; }
	static const uint32_t env_mode = std::getenv("ALPGPU_READ_AHEAD_MODE") ? static_cast<uint32_t>(std::atoi(std::getenv("ALPGPU_READ_AHEAD_MODE"))) : 0u;
	// (the self-stretching lead: built, measured — call 1b: +4-5 points where the lead is 20-30 % short, nothing where it is half, -1.5 % at the shipped lead on columns with
	//  exceptions — and OFF unless ALPGPU_READ_AHEAD_ADAPT=1: profiles/r06_decode_policy.txt)
	static const bool     no_adapt = !(std::getenv("ALPGPU_READ_AHEAD_ADAPT") && std::atoi(std::getenv("ALPGPU_READ_AHEAD_ADAPT")) != 0);
	const uint32_t mode = (env_mode & ~8u) | (from_plan ? 8u : 0u) | (no_adapt ? 16u : 0u);
	// patience: see above.  The decode's duration at the full HBM rate (ps_per_vector), in ticks
	const uint64_t est   = col->n_vectors * static_cast<uint64_t>(ps_per_vector) / (ps_per_tick ? ps_per_tick : 1u);
	const uint64_t floor_t = 200000000ull / (ps_per_tick ? ps_per_tick : 1u);
	const uint64_t first = est / 4 > floor_t ? est / 4 : floor_t, steady = 2 * est > floor_t ? 2 * est : floor_t;
	const uint32_t pf = first > 0xFFFFFFFFull ? 0xFFFFFFFFu : static_cast<uint32_t>(first), ps = steady > 0xFFFFFFFFull ? 0xFFFFFFFFu : static_cast<uint32_t>(steady);
	static const uint32_t burst_us = std::getenv("ALPGPU_READ_AHEAD_BURST_US") ? static_cast<uint32_t>(std::atoi(std::getenv("ALPGPU_READ_AHEAD_BURST_US"))) : 0u;
	const uint32_t burst = static_cast<uint32_t>(static_cast<uint64_t>(burst_us) * 1000000ull / (ps_per_tick ? ps_per_tick : 1u));
	if (value_bytes == 8) {
		hipLaunchKernelGGL((k_read_ahead<8>), dim3(static_cast<unsigned>(grid)), dim3(64 * kAheadWaves), 0, stream, col->d_vectors, col->d_packed, col->d_exc, col->n_vectors,
		                   d_ctx_words, tag, lead_min, lead_max, ps_per_vector, ps_per_tick, max_bits, mode, pf, ps, burst);
	} else {
		hipLaunchKernelGGL((k_read_ahead<4>), dim3(static_cast<unsigned>(grid)), dim3(64 * kAheadWaves), 0, stream, col->d_vectors, col->d_packed, col->d_exc, col->n_vectors,
		                   d_ctx_words, tag, lead_min, lead_max, ps_per_vector, ps_per_tick, max_bits, mode, pf, ps, burst);
	}
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

} // namespace alpgpu
