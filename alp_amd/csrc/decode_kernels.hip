// decode_kernels.hip — fused ALP / ALP_RD column decode for gfx950.
//
// Replaces, per vector (reference file:line relative to /root/reference):
//   ALP    : generated::falp::fallback::scalar::falp   include/alp/falp.hpp:10-26, src/falp.cpp:42440 (:114-121 per value)
//            + alp::decoder<double>::patch_exceptions   include/alp/decoder.hpp:141-149
//   ALP_RD : unffor::unffor (u64 right, u16 left)       src/fastlanes_generated_unffor.cpp:23010, :22846
//            + alp::rd_encoder<double>::decode          include/alp/rd.hpp:152-178
//
// Launch shape (measured: profiles/r01_membw_store_structure.txt, r01_membw2_waves_per_vector.txt,
// r01_decode_ablation.txt): the HBM subsystem of MI355X sustains ~5.2-5.4 TB/s of mixed read+write traffic from a
// persistent grid-stride kernel in which every wavefront streams its own 8 KiB vector (and the arithmetic is NOT
// the limiter there: dropping it changes nothing), but ~6.0-6.3 TB/s when the hardware dispatcher hands out ONE
// SMALL WORKGROUP PER VECTOR in order — the in-flight footprint stays compact and in address order.  So:
// grid = n_vectors, one 256-thread workgroup (4 wavefronts) per 1024-value vector, wavefront q owns steps
// m = 2q, 2q+1 (a contiguous 2 KiB quarter of the output):
//   1. one 32-byte descriptor load (wave-uniform scalar load),
//   2. the vector's packed words (128*bw bytes) go HBM -> LDS once, 16 bytes per thread per load,
//   3. the exception record (if any) becomes a 1024-bit position mask in LDS (+ values staged in LDS); every
//      wavefront keeps the 32-word exclusive prefix of the mask's popcounts in registers,
//   4. each lane unpacks its value pairs, applies base/FACT/FRAC (or the RD dictionary glue), substitutes
//      exceptions by rank, and stores 16 bytes -> every store instruction writes 1 KiB contiguous.
// What bounds it now (profiles/r01_time_waves.txt, r01_time_prefetch.txt): a workgroup lives for two dependent HBM
// round trips (descriptor, then packed words) and 8 workgroups fit per CU (32-wave cap), i.e. ~2 vectors/us/CU
// regardless of bit width; 2 waves per vector doubles the vectors in flight and lifts narrow widths (bw 4: 4.1 ->
// 5.4 TB/s) but loses on wide ones and on the mixed-width benchmark column (0.71 vs 0.78 of peak), and an L2
// prefetch of the descriptor 2-16 Ki vectors ahead did not move anything.  4 waves per vector is the default.
// HBM traffic per vector is the algorithmic minimum: 32 B descriptor + 128*bw B packed + exception record read,
// 8192 B written (rocprofv3 FETCH_SIZE/WRITE_SIZE: profiles/*_pmc.json); no intermediate goes to HBM.
#include "alp_device.hpp"
#include "launch.hpp"

namespace alpgpu {

#ifndef ALPGPU_DEC_WAVES
#define ALPGPU_DEC_WAVES 4
#endif
constexpr int kDecWaves     = ALPGPU_DEC_WAVES; // wavefronts cooperating on one vector
constexpr int kStepsPerWave = 8 / kDecWaves;
constexpr int kStageBytes   = 8704; // >= 63*128 (RD right) + 3*128 (RD left) + 128 pad, and >= 64*128 + 128 (ALP bw 64)
constexpr int kExcStage     = 128;  // exception values staged in LDS per vector; the rest are read from HBM on use

struct __attribute__((aligned(16))) DecodeLds {
	uint8_t  stage[kStageBytes];
	uint32_t mask[32];
	uint64_t excv[kExcStage];
};

// exclusive prefix over the 32 mask words' popcounts, held by lanes 0..31 of every wavefront
__device__ __forceinline__ int mask_prefix(const DecodeLds& L, int lane) {
	const uint32_t w   = L.mask[lane & 31];
	const int      c   = lane < 32 ? __builtin_popcount(w) : 0;
	int            inc = c;
#pragma unroll
	for (int d = 1; d < 32; d <<= 1) {
		const int t = __shfl_up(inc, d);
		if (lane >= d) { inc += t; }
	}
	return inc - c;
}

template <int VAL_BYTES>
__device__ __forceinline__ uint64_t fetch_exception(const DecodeLds& L, const uint8_t* __restrict__ rec, int rank) {
	if (rank < kExcStage) { return L.excv[rank]; }
	if constexpr (VAL_BYTES == 8) {
		return reinterpret_cast<const uint64_t*>(rec)[rank];
	} else {
		return reinterpret_cast<const uint16_t*>(rec)[rank];
	}
}

// exception lookup for the pair (128*m + 2*lane, +1): 2-bit hit mask and the rank of the first hit
__device__ __forceinline__ uint32_t exception_hits(const DecodeLds& L, int pref_reg, int m, int lane, int& rank) {
	const int      wi   = 4 * m + (lane >> 4);
	const uint32_t word = L.mask[wi];
	const int      b0   = (2 * lane) & 31;
	const uint32_t hits = (word >> b0) & 3u;
	rank                = __shfl(pref_reg, wi) + __builtin_popcount(word & ((1u << b0) - 1u));
	return hits;
}

template <bool NT_STORE>
__device__ __forceinline__ void store_pair(double2* __restrict__ p, double x, double y) {
	typedef double d2v __attribute__((ext_vector_type(2)));
	d2v o;
	o.x = x;
	o.y = y;
	if constexpr (NT_STORE) {
		__builtin_nontemporal_store(o, reinterpret_cast<d2v*>(p));
	} else {
		*reinterpret_cast<d2v*>(p) = o;
	}
}

template <bool NT_STORE>
__global__ __launch_bounds__(64 * kDecWaves) void k_decode_column(const alpgpu_vector_desc* __restrict__ descs,
                                                                  const alpgpu_rowgroup_state* __restrict__ rgs,
                                                                  const uint8_t* __restrict__ packed,
                                                                  const uint8_t* __restrict__ excs, double* __restrict__ out,
                                                                  uint64_t n_vectors, uint64_t v_offset) {
	__shared__ DecodeLds L;
	const int            tid  = static_cast<int>(threadIdx.x);
	const int            lane = tid & 63;
	const int            wave = wave_in_wg();
	const uint64_t       v    = v_offset + blockIdx.x;
	if (v >= n_vectors) { return; }

	// The exception mask is zeroed and that write is fenced BEFORE anything waits on memory, so that the only
	// barrier that sits behind HBM latency is the single one after all of this vector's loads have landed.
	if (tid < 32) { L.mask[tid] = 0; }
	__syncthreads();

	const alpgpu_vector_desc d      = descs[v];
	const int                bw     = d.bw;
	const int                cnt    = d.exc_cnt;
	const bool               is_alp = d.scheme == ALPGPU_SCHEME_ALP;
	const int                lbw    = is_alp ? 0 : d.lbw;
	const uint8_t*           rec    = excs + d.exc_off;
	double2*                 dst    = reinterpret_cast<double2*>(out + v * kVec);

	// issue every load of this vector back to back: packed words (<= 3 x 16 B per thread), exception position + value
	typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
	const int       n_units = 8 * (bw + lbw);
	const ull2*     g       = reinterpret_cast<const ull2*>(packed + d.packed_off);
	const int       vb      = is_alp ? 8 : 2;
	const uint16_t* poss    = reinterpret_cast<const uint16_t*>(rec + static_cast<size_t>(cnt) * vb);
	ull2            pk0 = {0, 0}, pk1 = {0, 0}, pk2 = {0, 0};
	uint32_t        epos = 0;
	uint64_t        eval = 0;
	constexpr int T = 64 * kDecWaves;
	ull2          pk3 = {0, 0}, pk4 = {0, 0}, pk5 = {0, 0};
	if (tid < n_units) { pk0 = g[tid]; }
	if (tid + T < n_units) { pk1 = g[tid + T]; }
	if (tid + 2 * T < n_units) { pk2 = g[tid + 2 * T]; }
	if (T < 256) {
		if (tid + 3 * T < n_units) { pk3 = g[tid + 3 * T]; }
		if (tid + 4 * T < n_units) { pk4 = g[tid + 4 * T]; }
		if (tid + 5 * T < n_units) { pk5 = g[tid + 5 * T]; }
	}
	static_assert(kDecWaves >= 2, "exception staging below assumes T >= kExcStage");
	if (tid < cnt) {
		epos = poss[tid];
		if (tid < kExcStage) { eval = is_alp ? reinterpret_cast<const uint64_t*>(rec)[tid] : static_cast<uint64_t>(reinterpret_cast<const uint16_t*>(rec)[tid]); }
	}
	{
		ull2* s = reinterpret_cast<ull2*>(L.stage);
		if (tid < n_units) { s[tid] = pk0; }
		if (tid + T < n_units) { s[tid + T] = pk1; }
		if (tid + 2 * T < n_units) { s[tid + 2 * T] = pk2; }
		if (T < 256) {
			if (tid + 3 * T < n_units) { s[tid + 3 * T] = pk3; }
			if (tid + 4 * T < n_units) { s[tid + 4 * T] = pk4; }
			if (tid + 5 * T < n_units) { s[tid + 5 * T] = pk5; }
		}
	}
	if (tid < cnt) {
		atomicOr(&L.mask[epos >> 5], 1u << (epos & 31));
		if (tid < kExcStage) { L.excv[tid] = eval; }
	}
	for (int j = tid + T; j < cnt; j += T) { // more exceptions than threads in one vector: rare
		const uint32_t p = poss[j];
		atomicOr(&L.mask[p >> 5], 1u << (p & 31));
	}
	__syncthreads();
	const int pref_reg = cnt > 0 ? mask_prefix(L, lane) : 0;

	const int      a = lane & 7;
	const int      r0 = lane >> 3;
	const UnitsPtr units {reinterpret_cast<const ulonglong2*>(L.stage)};
	if (is_alp) {
		const uint64_t base = static_cast<uint64_t>(d.base);
		const int64_t  fact = kFactArr[d.f];
		const double   frac = kFracArr[d.e];
		const uint64_t mask = bw_mask(bw);
#pragma unroll
		for (int mm = 0; mm < kStepsPerWave; ++mm) {
			const int     m  = kStepsPerWave * wave + mm;
			const U64Pair u  = unpack_pair_u64(units, bw, mask, 8 * m + r0, a);
			double        ox = decode_value(static_cast<int64_t>(u.x + base), fact, frac);
			double        oy = decode_value(static_cast<int64_t>(u.y + base), fact, frac);
			if (cnt > 0) {
				int            rank;
				const uint32_t hits = exception_hits(L, pref_reg, m, lane, rank);
				if (hits & 1u) {
					ox = __longlong_as_double(static_cast<long long>(fetch_exception<8>(L, rec, rank)));
					++rank;
				}
				if (hits & 2u) { oy = __longlong_as_double(static_cast<long long>(fetch_exception<8>(L, rec, rank))); }
			}
			store_pair<NT_STORE>(dst + 64 * m + lane, ox, oy);
		}
	} else {
		// ALP_RD: right parts = u64 lanes (bw = rbw, base 0); left parts = u16 lanes, 64 streams x 16 rows (value i ->
		// lane64 = i & 63, row = i >> 6, word k at left[64*k + lane64]); a lane's pair shares the row 2m + (lane >> 5)
		// and is one aligned u32 of the left stream.
		const alpgpu_rowgroup_state* rgp  = rgs + v / kRowgroup;
		const int                    rbw  = bw;
		const uint64_t               mask = bw_mask(rbw);
		const uint32_t               lmsk = (1u << lbw) - 1u;
		const uint64_t dlo = static_cast<uint64_t>(rgp->rd_dict[0]) | (static_cast<uint64_t>(rgp->rd_dict[1]) << 16) |
		                     (static_cast<uint64_t>(rgp->rd_dict[2]) << 32) | (static_cast<uint64_t>(rgp->rd_dict[3]) << 48);
		const uint64_t dhi = static_cast<uint64_t>(rgp->rd_dict[4]) | (static_cast<uint64_t>(rgp->rd_dict[5]) << 16) |
		                     (static_cast<uint64_t>(rgp->rd_dict[6]) << 32) | (static_cast<uint64_t>(rgp->rd_dict[7]) << 48);
		const uint32_t* lsrc = reinterpret_cast<const uint32_t*>(L.stage + 128 * rbw);
#pragma unroll
		for (int mm = 0; mm < kStepsPerWave; ++mm) {
			const int      m   = kStepsPerWave * wave + mm;
			const U64Pair  u   = unpack_pair_u64(units, rbw, mask, 8 * m + r0, a);
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
				int            rank;
				const uint32_t hits = exception_hits(L, pref_reg, m, lane, rank);
				if (hits & 1u) {
					l0 = fetch_exception<2>(L, rec, rank);
					++rank;
				}
				if (hits & 2u) { l1 = fetch_exception<2>(L, rec, rank); }
			}
			store_pair<NT_STORE>(dst + 64 * m + lane, __longlong_as_double(static_cast<long long>((l0 << rbw) | u.x)),
			                     __longlong_as_double(static_cast<long long>((l1 << rbw) | u.y)));
		}
	}
}

int launch_decode_column(hipStream_t stream, const alpgpu_column* col, double* d_out, int variant, int n_cus) {
	(void)n_cus;
	const uint64_t n = col->n_vectors;
	// one workgroup per vector; a grid dimension holds < 2^31 workgroups -> chunk very long columns
	const uint64_t kMaxGrid = 1ull << 30;
	for (uint64_t off = 0; off < n; off += kMaxGrid) {
		const unsigned grid = static_cast<unsigned>(n - off < kMaxGrid ? n - off : kMaxGrid);
		if (variant == 1) {
			hipLaunchKernelGGL((k_decode_column<false>), dim3(grid), dim3(64 * kDecWaves), 0, stream, col->d_vectors, col->d_rowgroups,
			                   col->d_packed, col->d_exc, d_out, n, off);
		} else {
			hipLaunchKernelGGL((k_decode_column<true>), dim3(grid), dim3(64 * kDecWaves), 0, stream, col->d_vectors, col->d_rowgroups,
			                   col->d_packed, col->d_exc, d_out, n, off);
		}
	}
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

} // namespace alpgpu
