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
#define ALPGPU_LOOK_SLEEP 64 // x64 cycles between polls of a window that still holds an unfinished tile
#endif
constexpr int      kLookWindow      = ALPGPU_LOOK_WINDOW; // status words examined per look-back round

__device__ __forceinline__ uint64_t status_pack(uint64_t flag, uint64_t packed_units, uint64_t exc_units) {
	return flag | (packed_units << 31) | exc_units;
}

// The look-back of one tile (run by wavefront 0 after it has posted its own size): finds the tile's exclusive prefix,
// waits for the tile's aggregate, publishes the inclusive prefix and releases the workgroup through LDS.
__device__ __forceinline__ void tile_lookback(uint64_t tile, uint64_t* __restrict__ status, uint64_t* __restrict__ totals, const uint64_t* s_size,
                                              uint32_t* s_count, uint64_t* s_excl, uint32_t* s_ready, int lane) {
		uint64_t* my_status = status + tile;
		uint64_t  excl      = 0;
		bool      stalled   = false;
#ifdef ALPGPU_FUSED_NO_LOOKBACK // timing experiment only: worst-case strides instead of the scan (output is NOT compact)
		excl = status_pack(0, tile * kWavesPerWg * 66, tile * kWavesPerWg * 1280);
		if (false) {
#else
		if (tile != 0) {
#endif
			int64_t  look  = static_cast<int64_t>(tile) - 1; // nearest tile not yet accounted for
			uint32_t spins = 0;
			while (look >= 0) {
				uint64_t st[kLookWindow / 64];
#pragma unroll
				for (int k = 0; k < kLookWindow / 64; ++k) {
					const int64_t idx = look - (lane + 64 * k);
					st[k]             = idx >= 0 ? __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kFlagPrefix;
				}
				// entries are ordered nearest-first (k major): take everything up to and including the first prefix,
				// provided nothing before it is still invalid; otherwise poll again
				bool     done = false, retry = false;
				uint64_t part = 0;
#pragma unroll
				for (int k = 0; k < kLookWindow / 64; ++k) {
					if (!done && !retry) { // wave-uniform
						const uint64_t fl         = st[k] >> 62;
						const uint64_t has_prefix = __ballot(fl == 2);
						const uint64_t invalid    = __ballot(fl == 0);
						const int      first_p    = has_prefix ? __builtin_ctzll(has_prefix) : 64;
						const uint64_t upto       = first_p >= 63 ? ~0ull : ((2ull << first_p) - 1ull); // lanes 0..first_p
						if (invalid & upto) {
							retry = true;
						} else {
							part += (first_p == 64 || lane <= first_p) ? (st[k] & ~(3ull << 62)) : 0ull;
							done = first_p != 64;
						}
					}
				}
				if (retry) {
					if (++spins > kSpinLimit || __hip_atomic_load(totals + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
						stalled = true;
						break;
					}
					__builtin_amdgcn_s_sleep(ALPGPU_LOOK_SLEEP);
					continue;
				}
#pragma unroll
				for (int dd = 32; dd >= 1; dd >>= 1) { part += static_cast<uint64_t>(__shfl_xor(static_cast<long long>(part), dd)); }
				excl += part;
				if (done) { break; }
				look -= kLookWindow;
			}
		}
		// the tile's own aggregate: wait (LDS only) until every worker has posted its size
		uint32_t spins = 0;
		while (__hip_atomic_load(s_count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != kWavesPerWg) {
			if (++spins > kSpinLimit) {
				stalled = true;
				break;
			}
			__builtin_amdgcn_s_sleep(2);
		}
		uint64_t aggregate = 0;
#pragma unroll
		for (int w = 0; w < kWavesPerWg; ++w) { aggregate += s_size[w]; }
		if (lane == 0) {
			if (stalled) {
				__hip_atomic_store(totals + 3, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				*s_excl = ~0ull;
			} else {
				__hip_atomic_store(my_status, kFlagPrefix | (excl + aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				*s_excl = excl;
				if (tile == gridDim.x - 1) { // running totals of the column, published by k_fused_finish
					const uint64_t incl = excl + aggregate;
					totals[4]           = totals[0] + ((incl >> 31) & 0x7FFFFFFFull) * 128ull;
					totals[5]           = totals[1] + (incl & 0x7FFFFFFFull) * 8ull;
				}
			}
			__hip_atomic_store(s_ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
}

} // namespace alpgpu
