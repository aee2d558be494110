// encode_lookback.hpp — the ordered-offset machinery of the single-pass encode kernels (double: encode_kernels.hip,
// float: encode_f32_kernels.hip): one 64-bit status word per tile {flag | packed 128-byte units | exception 8-byte units}
// and the decoupled look-back that turns the tiles' sizes into vector-order stream offsets.  See the description above
// k_encode_fused in encode_kernels.hip.
#pragma once
#include "alp_device.hpp"

namespace alpgpu {

constexpr uint64_t kFusedMaxVectors = 1ull << 20; // 2^20 vectors * 1280 exception units < 2^31
constexpr uint64_t kFlagAggregate   = 1ull << 62;
constexpr uint64_t kFlagPrefix      = 2ull << 62;
constexpr uint32_t kSpinLimit       = 1u << 20;
#ifndef ALPGPU_LOOK_WINDOW
#define ALPGPU_LOOK_WINDOW 64
#endif
#ifndef ALPGPU_LOOK_SLEEP
#define ALPGPU_LOOK_SLEEP 16 // x64 cycles between polls of a window that still holds an unfinished tile (with 8-vector tiles: 8-24 -> 3.12 ms, 64 -> 3.22)
#endif
constexpr int      kLookWindow      = ALPGPU_LOOK_WINDOW; // status words examined per look-back round

__device__ __forceinline__ uint64_t status_pack(uint64_t flag, uint64_t packed_units, uint64_t exc_units) {
	return flag | (packed_units << 31) | exc_units;
}

// what a vector that did not fit its streams gets: a decoder stays inside the buffers (the column's content is unspecified)
__device__ __forceinline__ alpgpu_vector_desc empty_descriptor() {
	alpgpu_vector_desc z;
	z.packed_off = z.exc_off = 0;
	z.base                   = 0;
	z.bw = z.e = z.f = z.lbw = 0;
	z.exc_cnt                = 0;
	z.scheme                 = ALPGPU_SCHEME_ALP;
	return z;
}

// bytes a vector's record takes in the packed and in the exception stream; VALUE_BYTES = 8 (double column) or 4 (float column)
template <int VALUE_BYTES>
__device__ __forceinline__ void record_sizes(const alpgpu_vector_desc& d, uint64_t& packed, uint64_t& exc) {
	if (d.scheme == ALPGPU_SCHEME_ALP) {
		packed = 128ull * d.bw;
		exc    = ((VALUE_BYTES + 2ull) * d.exc_cnt + 7ull) & ~7ull; // cnt x value bits, then cnt x u16 positions
	} else {
		packed = 128ull * (static_cast<uint64_t>(d.bw) + d.lbw);
		exc    = (4ull * d.exc_cnt + 7ull) & ~7ull; // cnt x u16 left parts, then cnt x u16 positions
	}
}

// ---- two-level form (default) -------------------------------------------------------------------------------------------
// Measured on MI355X (tools/fused_phases.py, profiles/r01_fused_phases.txt): with the flat look-back below a wavefront spends
// ~37 % of its life waiting for its offset, and wider windows or faster polling make it worse — agent-scope loads of status
// words are served on the far side of the fabric (the XCDs' L2s are not coherent with each other), so the polling traffic
// itself (64 words x ~5 rounds per tile) is the cost.  Here tiles form blocks of kBlockTiles consecutive tiles:
//   level 1  status[tile]   = AGGREGATE | size of the tile, written once by the tile's last-arriving wavefront;
//            a tile reads its <= 63 predecessors INSIDE its block in one round (they were dispatched just before it);
//   level 2  bstatus[block] = AGGREGATE | size of the block, written by the tile that closes the block as soon as it knows
//            it, later replaced by PREFIX | inclusive prefix of the block; a tile finds its block's base with a decoupled
//            look-back over the (few, 64x sparser) block words, reading the tile words of a block directly while that block's
//            word is still missing (so nobody waits for another tile's look-back, only for tiles to finish encoding).
// Exclusive prefix of a tile = base of its block + sizes of its predecessors in the block.  ~2 rounds and ~130 status loads
// per tile instead of ~5 and ~320, and no positive feedback between look-back latency and look-back distance.
constexpr int kBlockTiles = 64;

__device__ __forceinline__ uint64_t status_load(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void     status_store(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Sum of the lanes' status payloads {packed units << 31 | exception units} (flags already masked off), wave-uniform.  The two
// 31-bit fields are reduced separately with DPP adds and joined by an ADD, which is what a 64-bit sum of the words gives; as
// six dependent 64-bit __shfl_xor steps (12 ds_bpermutes) this sum was ~1k cycles on the scout's path, two or three times per tile.
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
	const uint32_t lo = wave_scan_add_u32(static_cast<uint32_t>(v) & 0x7FFFFFFFu);
	const uint32_t hi = wave_scan_add_u32(static_cast<uint32_t>(v >> 31));
	const uint64_t l  = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(lo), 63));
	const uint64_t h  = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(hi), 63));
	return (h << 31) + l;
}

// status: [gridDim.x] tile words followed by [ceil(gridDim.x / kBlockTiles)] block words, all zero at launch.
// N_SIZES = vectors per tile (entries of s_size); s_count counts the tile's kFusedWaves wavefronts.
// spin_limit: unsuccessful polls before the tile gives up (kSpinLimit; 0 makes every tile that has to wait give up at once —
// the debug option that exercises the recovery route).
// The first round of both levels, issued ahead of time (k_encode_lean: in front of the exception record's stage, whose work then covers the
// trip across the fabric): the words a tile_lookback would begin with.  Words that are not there yet are simply polled again by tile_lookback.
struct LookbackFirst {
	uint64_t w1, w2;
};
__device__ __forceinline__ LookbackFirst tile_lookback_begin(uint64_t tile, const uint64_t* __restrict__ status, int lane) {
	const uint64_t* bstatus = status + gridDim.x;
	const uint64_t  block   = tile / kBlockTiles;
	const int       i       = static_cast<int>(tile % kBlockTiles);
	LookbackFirst   f;
	f.w1 = lane < i ? status_load(status + (tile - 1 - lane)) : kFlagAggregate;
	f.w2 = (block != 0 && static_cast<int64_t>(block) - 1 - lane >= 0) ? status_load(bstatus + (block - 1 - lane)) : kFlagPrefix;
	return f;
}

template <int N_SIZES = kFusedWaves>
__device__ __forceinline__ void tile_lookback(uint64_t tile, uint64_t* __restrict__ status, uint64_t* __restrict__ totals, const uint64_t* s_size,
                                              uint32_t* s_count, uint64_t* s_excl, uint32_t* s_ready, int lane, uint32_t spin_limit = kSpinLimit,
                                              const LookbackFirst* begun = nullptr) {
	uint64_t*      bstatus = status + gridDim.x;
	const uint64_t block   = tile / kBlockTiles;
	const int      i       = static_cast<int>(tile % kBlockTiles);
	const bool     closes  = i == kBlockTiles - 1 || tile == gridDim.x - 1; // this tile completes its block
	bool           stalled = false;
	uint32_t       spins   = 0;
	// the stall flag of another tile is a far-side read like the status words: looked at every 16th unsuccessful round only
	auto give_up = [&]() { return ++spins > spin_limit || ((spins & 15u) == 0 && status_load(totals + 3) != 0); };
#ifdef ALPGPU_ABLATE_LOOKBACK // timing experiment: worst-case strides instead of the scan (the output is NOT compact)
	if (lane == 0) {
		*s_excl = status_pack(0, tile * N_SIZES * 66, tile * N_SIZES * 1280);
		__hip_atomic_store(s_ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
	return;
#endif

	// level 1: the sizes of the i predecessors inside this block
	uint64_t local = 0;
	// both levels' first rounds are issued together: one trip across the fabric instead of two when nothing is late
	uint64_t first1, first2;
	if (begun != nullptr) { // (a compile-time fact at every call site)
		first1 = begun->w1, first2 = begun->w2;
	} else {
		first1 = lane < i ? status_load(status + (tile - 1 - lane)) : kFlagAggregate;
		first2 = (block != 0 && static_cast<int64_t>(block) - 1 - lane >= 0) ? status_load(bstatus + (block - 1 - lane)) : kFlagPrefix;
	}
	bool     fresh1 = true, fresh2 = true;
	if (i != 0) {
#ifdef ALPGPU_LOOK_REPOLL_MISSING // experiment (round 5, call 36): a word that has arrived is not read again — tile words are written once.  No difference (2.91-2.98 against 2.89-2.97 ms): off
		uint64_t st = first1;
		for (;;) {
			if (!fresh1 && lane < i && (st >> 62) == 0) { st = status_load(status + (tile - 1 - lane)); }
			fresh1 = false;
#else
		for (;;) {
			const uint64_t st = fresh1 ? first1 : (lane < i ? status_load(status + (tile - 1 - lane)) : kFlagAggregate);
			fresh1            = false;
#endif
			if (ballot64((st >> 62) == 0) == 0) {
				local = wave_sum_u64(st & ~(3ull << 62));
				break;
			}
			if (give_up()) {
				stalled = true;
				break;
			}
			__builtin_amdgcn_s_sleep(ALPGPU_LOOK_SLEEP);
		}
	}
	// a closing tile needs its own size as well (LDS only): it publishes the block's aggregate before looking further back
	uint64_t aggregate = 0;
	if (closes && !stalled) {
		uint32_t lspins = 0;
		while (__hip_atomic_load(s_count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != kFusedWaves) {
			if (++lspins > kSpinLimit) {
				stalled = true;
				break;
			}
			__builtin_amdgcn_s_sleep(2);
		}
#pragma unroll
		for (int w = 0; w < N_SIZES; ++w) { aggregate += s_size[w]; }
		if (!stalled && lane == 0) { status_store(bstatus + block, (block == 0 ? kFlagPrefix : kFlagAggregate) | (local + aggregate)); }
	}
	// level 2: the base of this block = everything before it
	uint64_t base = 0;
	if (block != 0 && !stalled) {
		int64_t look = static_cast<int64_t>(block) - 1; // nearest block not yet accounted for
#ifdef ALPGPU_LOOK_REPOLL_MISSING
		uint64_t st      = first2;
		int64_t  st_look = look; // the window st was read for
#endif
		while (look >= 0) {
			const int64_t  idx        = look - lane;
#ifdef ALPGPU_LOOK_REPOLL_MISSING
			if (!fresh2) {
				if (st_look != look) {
					st      = idx >= 0 ? status_load(bstatus + idx) : kFlagPrefix;
					st_look = look;
				} else if (idx >= 0 && (st >> 62) == 0) {
					st = status_load(bstatus + idx);
				}
			}
#else
			const uint64_t st         = fresh2 ? first2 : (idx >= 0 ? status_load(bstatus + idx) : kFlagPrefix);
#endif
			fresh2                    = false;
			const uint64_t fl         = st >> 62;
			const uint64_t has_prefix = ballot64(fl == 2);
			const uint64_t invalid    = ballot64(fl == 0);
			const int      first_p    = has_prefix ? __builtin_ctzll(has_prefix) : 64;
			const uint64_t upto       = first_p >= 63 ? ~0ull : ((2ull << first_p) - 1ull); // lanes 0..first_p
			uint64_t late = 0; // sizes of nearer blocks whose word is not there yet, summed from their 64 tile words
			if (invalid & upto) {
				// The closing tile of such a block is still waiting for the block's slowest tile or has not run its look-back yet;
				// the tile words themselves appear as soon as each tile has encoded, so read those instead of waiting a round.
				bool ready = true;
				for (uint64_t inv = invalid & upto; inv != 0 && ready; inv &= inv - 1) { // wave-uniform
					const uint64_t blk = static_cast<uint64_t>(look - __builtin_ctzll(inv));
					const uint64_t w   = status_load(status + blk * kBlockTiles + lane);
					ready              = ballot64((w >> 62) == 0) == 0;
					late += wave_sum_u64(w & ~(3ull << 62));
				}
				if (!ready) {
					if (give_up()) {
						stalled = true;
						break;
					}
					__builtin_amdgcn_s_sleep(ALPGPU_LOOK_SLEEP);
					continue;
				}
			}
			base += late + wave_sum_u64(((first_p == 64 || lane <= first_p) && fl != 0) ? (st & ~(3ull << 62)) : 0ull);
			if (first_p != 64) { break; }
			look -= 64;
		}
	}
	// the other wavefronts add up the sizes posted before theirs once released: every one of them must have posted (LDS only)
	if (!closes && !stalled) {
		uint32_t lspins = 0;
		while (__hip_atomic_load(s_count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != kFusedWaves) {
			if (++lspins > kSpinLimit) {
				stalled = true;
				break;
			}
			__builtin_amdgcn_s_sleep(2);
		}
	}
	if (lane == 0) {
		if (stalled) {
			status_store(totals + 3, 1ull);
			*s_excl = ~0ull;
		} else {
			const uint64_t excl = base + local;
			if (closes && block != 0) { status_store(bstatus + block, kFlagPrefix | (excl + aggregate)); }
			*s_excl = excl;
			if (tile == gridDim.x - 1) { // running totals of the column, published by the finish kernel
				const uint64_t incl = excl + aggregate;
				totals[4]           = totals[0] + ((incl >> 31) & 0x7FFFFFFFull) * 128ull;
				totals[5]           = totals[1] + (incl & 0x7FFFFFFFull) * 8ull;
			}
		}
		__hip_atomic_store(s_ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
}

// words of status workspace one launch over n_tiles tiles needs (tile words + block words)
__host__ __device__ inline uint64_t lookback_words(uint64_t n_tiles) { return n_tiles + (n_tiles + kBlockTiles - 1) / kBlockTiles; }

} // namespace alpgpu
