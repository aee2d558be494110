// decode_kernels.hip — fused ALP / ALP_RD column decode for gfx950.
//
// Replaces, per vector (reference file:line relative to /root/reference):
//   ALP    : generated::falp::fallback::scalar::falp   include/alp/falp.hpp:10-26, src/falp.cpp:42440 (:114-121 per value)
//            + alp::decoder<double>::patch_exceptions   include/alp/decoder.hpp:141-149
//   ALP_RD : unffor::unffor (u64 right, u16 left)       src/fastlanes_generated_unffor.cpp:23010, :22846
//            + alp::rd_encoder<double>::decode          include/alp/rd.hpp:152-178
//
// One wavefront decodes one vector per loop iteration (grid-stride over vectors):
//   1. one 32-byte descriptor load (wave-uniform),
//   2. the vector's packed words (128*bw bytes) are copied HBM -> LDS with 16-byte-per-lane loads,
//   3. the exception record (if any) becomes a 1024-bit position mask + per-word prefix counts in LDS, and
//      the exception values are staged in LDS (first kExcStage of them; the rest are read from HBM on use),
//   4. 8 steps; in each, a lane unpacks its value pair, applies base/FACT/FRAC (or the RD dictionary glue),
//      substitutes exceptions, and stores 16 bytes -> every store instruction writes 1 KiB contiguous.
// HBM traffic per vector is the algorithmic minimum: 32 B descriptor + 128*bw B packed + exception record
// read, 8192 B written; nothing is read twice and no intermediate goes to HBM.
#include "alp_device.hpp"
#include "launch.hpp"

namespace alpgpu {

constexpr int kStageBytes = 8704; // >= 63*128 (RD right) + 3*128 (RD left) + 128 pad, and >= 64*128 + 128 (ALP bw 64)
constexpr int kExcStage   = 128;  // exception values staged in LDS per vector

struct __attribute__((aligned(16))) DecodeLds {
	uint8_t  stage[kStageBytes];
	uint32_t mask[32];
	uint32_t pref[32];
	uint64_t excv[kExcStage];
};

using UnitsLds    = UnitsPtr;
using UnitsGlobal = UnitsPtr;

// Builds mask/pref (and stages values) for one vector's exception record.  VAL_BYTES = 8 (ALP) or 2 (RD).
template <int VAL_BYTES>
__device__ __forceinline__ void stage_exceptions(DecodeLds& L, const uint8_t* rec, int cnt, int lane) {
	if (lane < 32) { L.mask[lane] = 0; }
	wave_lds_sync();
	const uint16_t* poss = reinterpret_cast<const uint16_t*>(rec + static_cast<size_t>(cnt) * VAL_BYTES);
	for (int j = lane; j < cnt; j += 64) {
		const uint32_t p = poss[j];
		atomicOr(&L.mask[p >> 5], 1u << (p & 31));
		if (j < kExcStage) {
			if constexpr (VAL_BYTES == 8) {
				L.excv[j] = reinterpret_cast<const uint64_t*>(rec)[j];
			} else {
				L.excv[j] = reinterpret_cast<const uint16_t*>(rec)[j];
			}
		}
	}
	wave_lds_sync();
	// exclusive prefix of popcounts over the 32 mask words (lanes 32..63 carry zeros)
	const uint32_t w   = lane < 32 ? L.mask[lane] : 0u;
	const int      c   = __builtin_popcount(w);
	int            inc = c;
#pragma unroll
	for (int d = 1; d < 32; d <<= 1) {
		const int t = __shfl_up(inc, d);
		if (lane >= d) { inc += t; }
	}
	if (lane < 32) { L.pref[lane] = static_cast<uint32_t>(inc - c); }
	wave_lds_sync();
}

template <int VAL_BYTES>
__device__ __forceinline__ uint64_t fetch_exception(const DecodeLds& L, const uint8_t* rec, int rank) {
	if (rank < kExcStage) { return L.excv[rank]; }
	if constexpr (VAL_BYTES == 8) {
		return reinterpret_cast<const uint64_t*>(rec)[rank];
	} else {
		return reinterpret_cast<const uint16_t*>(rec)[rank];
	}
}

template <bool STAGE_LDS, bool NT_STORE>
__global__ __launch_bounds__(64 * kWavesPerWg) void k_decode_column(const alpgpu_vector_desc* __restrict__ descs,
                                                                    const alpgpu_rowgroup_state* __restrict__ rgs,
                                                                    const uint8_t* __restrict__ packed,
                                                                    const uint8_t* __restrict__ excs,
                                                                    double* __restrict__ out,
                                                                    uint64_t n_vectors) {
	__shared__ DecodeLds lds[kWavesPerWg];
	const int            lane = lane_id();
	const int            wave = wave_in_wg();
	DecodeLds&           L    = lds[wave];

	const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kWavesPerWg;
	for (uint64_t v = static_cast<uint64_t>(blockIdx.x) * kWavesPerWg + wave; v < n_vectors; v += stride) {
		const alpgpu_vector_desc d = descs[v];
		const int      bw     = d.bw;
		const int      cnt    = d.exc_cnt;
		const uint8_t* src    = packed + d.packed_off;
		const uint8_t* rec    = excs + d.exc_off;
		double2*       dst    = reinterpret_cast<double2*>(out + v * kVec);
		const bool     is_alp = d.scheme == ALPGPU_SCHEME_ALP;
		const int      lbw    = is_alp ? 0 : d.lbw;

		// -- stage packed words HBM -> LDS (16 B per lane per load) --
		if constexpr (STAGE_LDS) {
			const int         n_units = 8 * (bw + lbw);
			const ulonglong2* g       = reinterpret_cast<const ulonglong2*>(src);
			ulonglong2*       s       = reinterpret_cast<ulonglong2*>(L.stage);
			for (int c = lane; c < n_units; c += 64) { s[c] = g[c]; }
		}
		if (cnt > 0) {
			if (is_alp) {
				stage_exceptions<8>(L, rec, cnt, lane);
			} else {
				stage_exceptions<2>(L, rec, cnt, lane);
			}
		} else if constexpr (STAGE_LDS) {
			wave_lds_sync();
		}

		const int a  = lane & 7;
		const int r0 = lane >> 3;
		if (is_alp) {
			const uint64_t base = static_cast<uint64_t>(d.base);
			const int64_t  fact = kFactArr[d.f];
			const double   frac = kFracArr[d.e];
			const uint64_t mask = bw_mask(bw);
#pragma unroll
			for (int m = 0; m < 8; ++m) {
				U64Pair u;
				if constexpr (STAGE_LDS) {
					u = unpack_pair_u64(UnitsLds {reinterpret_cast<const ulonglong2*>(L.stage)}, bw, mask, 8 * m + r0, a);
				} else {
					u = unpack_pair_u64(UnitsGlobal {reinterpret_cast<const ulonglong2*>(src)}, bw, mask, 8 * m + r0, a);
				}
				double2 o;
				o.x = decode_value(static_cast<int64_t>(u.x + base), fact, frac);
				o.y = decode_value(static_cast<int64_t>(u.y + base), fact, frac);
				if (cnt > 0) {
					const int      wi   = 4 * m + (lane >> 4);
					const uint32_t word = L.mask[wi];
					const int      b0   = (2 * lane) & 31;
					const uint32_t hits = (word >> b0) & 3u;
					if (hits) {
						int rank = static_cast<int>(L.pref[wi]) + __builtin_popcount(word & ((1u << b0) - 1u));
						if (hits & 1u) {
							o.x = __longlong_as_double(static_cast<long long>(fetch_exception<8>(L, rec, rank)));
							++rank;
						}
						if (hits & 2u) { o.y = __longlong_as_double(static_cast<long long>(fetch_exception<8>(L, rec, rank))); }
					}
				}
				if constexpr (NT_STORE) {
					__builtin_nontemporal_store(o.x, &reinterpret_cast<double*>(dst + 64 * m + lane)[0]);
					__builtin_nontemporal_store(o.y, &reinterpret_cast<double*>(dst + 64 * m + lane)[1]);
				} else {
					dst[64 * m + lane] = o;
				}
			}
		} else {
			// ALP_RD: right parts = u64 lanes (bw = rbw, base 0); left parts = u16 lanes, 64 streams x 16 rows
			// (value i -> lane64 = i & 63, row = i >> 6, word k at left[64*k + lane64]); a lane's pair shares
			// the row 2m + (lane >> 5) and is one aligned u32 of the left stream.
			const alpgpu_rowgroup_state rg   = rgs[v / kRowgroup];
			const int                   rbw  = bw;
			const uint64_t              mask = bw_mask(rbw);
			const uint32_t              lmsk = (1u << lbw) - 1u;
			const uint64_t dlo = static_cast<uint64_t>(rg.rd_dict[0]) | (static_cast<uint64_t>(rg.rd_dict[1]) << 16) |
			                     (static_cast<uint64_t>(rg.rd_dict[2]) << 32) | (static_cast<uint64_t>(rg.rd_dict[3]) << 48);
			const uint64_t dhi = static_cast<uint64_t>(rg.rd_dict[4]) | (static_cast<uint64_t>(rg.rd_dict[5]) << 16) |
			                     (static_cast<uint64_t>(rg.rd_dict[6]) << 32) | (static_cast<uint64_t>(rg.rd_dict[7]) << 48);
			const uint8_t*  lsrc_b = STAGE_LDS ? (L.stage + 128 * rbw) : (src + 128 * rbw);
			const uint32_t* lsrc   = reinterpret_cast<const uint32_t*>(lsrc_b);
#pragma unroll
			for (int m = 0; m < 8; ++m) {
				U64Pair u;
				if constexpr (STAGE_LDS) {
					u = unpack_pair_u64(UnitsLds {reinterpret_cast<const ulonglong2*>(L.stage)}, rbw, mask, 8 * m + r0, a);
				} else {
					u = unpack_pair_u64(UnitsGlobal {reinterpret_cast<const ulonglong2*>(src)}, rbw, mask, 8 * m + r0, a);
				}
				const int      row = 2 * m + (lane >> 5);
				const int      p   = row * lbw;
				const int      k   = p >> 4;
				const int      s   = p & 15;
				const uint32_t w0  = lsrc[32 * k + (lane & 31)];
				const uint32_t w1  = lsrc[32 * k + 32 + (lane & 31)];
				const uint32_t i0  = (((w0 & 0xFFFFu) >> s) | ((w1 & 0xFFFFu) << (16 - s))) & lmsk;
				const uint32_t i1  = (((w0 >> 16) >> s) | ((w1 >> 16) << (16 - s))) & lmsk;
				uint64_t       l0  = ((i0 < 4 ? dlo >> (16 * i0) : dhi >> (16 * (i0 & 3))) & 0xFFFFull);
				uint64_t       l1  = ((i1 < 4 ? dlo >> (16 * i1) : dhi >> (16 * (i1 & 3))) & 0xFFFFull);
				if (cnt > 0) {
					const int      wi   = 4 * m + (lane >> 4);
					const uint32_t word = L.mask[wi];
					const int      b0   = (2 * lane) & 31;
					const uint32_t hits = (word >> b0) & 3u;
					if (hits) {
						int rank = static_cast<int>(L.pref[wi]) + __builtin_popcount(word & ((1u << b0) - 1u));
						if (hits & 1u) {
							l0 = fetch_exception<2>(L, rec, rank);
							++rank;
						}
						if (hits & 2u) { l1 = fetch_exception<2>(L, rec, rank); }
					}
				}
				double2 o;
				o.x = __longlong_as_double(static_cast<long long>((l0 << rbw) | u.x));
				o.y = __longlong_as_double(static_cast<long long>((l1 << rbw) | u.y));
				dst[64 * m + lane] = o;
			}
		}
		// the next iteration overwrites this wave's LDS: keep this iteration's reads ahead of those writes
		wave_lds_sync();
	}
}

int launch_decode_column(hipStream_t stream, const alpgpu_column* col, double* d_out, int variant, int n_cus) {
	if (col->n_vectors == 0) { return ALPGPU_OK; }
	const uint64_t wgs_needed = (col->n_vectors + kWavesPerWg - 1) / kWavesPerWg;
	const uint64_t wgs_cap    = static_cast<uint64_t>(n_cus) * 16; // persistent waves, grid-stride over vectors
	const unsigned grid       = static_cast<unsigned>(wgs_needed < wgs_cap ? wgs_needed : wgs_cap);
	const dim3     block(64 * kWavesPerWg);
	switch (variant) {
	case 1:
		hipLaunchKernelGGL((k_decode_column<false, false>), dim3(grid), block, 0, stream, col->d_vectors, col->d_rowgroups,
		                   col->d_packed, col->d_exc, d_out, col->n_vectors);
		break;
	case 2:
		hipLaunchKernelGGL((k_decode_column<true, true>), dim3(grid), block, 0, stream, col->d_vectors, col->d_rowgroups,
		                   col->d_packed, col->d_exc, d_out, col->n_vectors);
		break;
	default:
		hipLaunchKernelGGL((k_decode_column<true, false>), dim3(grid), block, 0, stream, col->d_vectors, col->d_rowgroups,
		                   col->d_packed, col->d_exc, d_out, col->n_vectors);
		break;
	}
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

} // namespace alpgpu
