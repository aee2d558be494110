// encode_kernels.hip — whole-column ALP / ALP_RD vector encode for gfx950 (given the rowgroup states).
//
// Replaces, per vector (file:line relative to /root/reference):
//   alp::encoder<double>::encode          include/alp/encoder.hpp:402-418 (second-level sampling :241-305, encode_simdized :307-400)
//   alp::encoder<double>::analyze_ffor    include/alp/encoder.hpp:109-120
//   ffor::ffor (int64 / uint64 / uint16)  src/fastlanes_ffor.cpp:5-35 -> src/fastlanes_generated_ffor.cpp:29939, :29781
//   alp::rd_encoder<double>::encode       include/alp/rd.hpp:109-147
// and the caller's per-column loop (publication/source_code/bench_compression_ratio/alp.cpp:198-229).
//
// Variable-size output needs each vector's byte offsets.  Round-1 structure:
//   k_encode_analyze : one wave per vector; picks (e,f), counts exceptions, finds base/bw -> descriptor sizes
//   k_scan_tiles     : exclusive scan of the sizes inside tiles of 1024 vectors, tile totals to a workspace
//   k_scan_totals    : one workgroup scans the tile totals, writes the stream totals / overflow flag
//   k_encode_pack    : one wave per vector; re-encodes with the chosen (e,f) and writes packed words,
//                      exception record and the final descriptor at the now-known offsets
// Offsets are therefore in vector order and the output is byte-reproducible.  The input is read twice
// (analysis + pack); fusing the two passes with a decoupled look-back scan is the planned next step.
#include "encode_device.hpp"
#include "launch.hpp"

#include <cstdlib>

namespace alpgpu {

constexpr int kScanTile = 1024;

// ---- pass 1: analysis -----------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kWavesPerWg) void k_encode_analyze(const double* __restrict__ in,
                                                                     const alpgpu_rowgroup_state* __restrict__ rgs,
                                                                     alpgpu_vector_desc* __restrict__ descs, uint64_t n_vectors) {
	__shared__ EncodeLds lds[kWavesPerWg];
	const int            lane = lane_id();
	const int            wave = wave_in_wg();
	EncodeLds&           L    = lds[wave];
	const uint64_t       stride = static_cast<uint64_t>(gridDim.x) * kWavesPerWg;
	for (uint64_t v = static_cast<uint64_t>(blockIdx.x) * kWavesPerWg + wave; v < n_vectors; v += stride) {
		const alpgpu_rowgroup_state* rgp = rgs + v / kRowgroup;
		const VecIn                  x   = load_vector(in, v, lane);
		alpgpu_vector_desc           d;
		d.packed_off = 0;
		d.exc_off    = 0;
		d.scheme     = rgp->scheme;
		if (rgp->scheme == ALPGPU_SCHEME_ALP) {
			int e, f;
			if (rgp->k > 1) {
				second_level_select(x, rgp, L, lane, e, f);
			} else {
				e = rgp->combos[0];
				f = rgp->combos[1];
			}
			AlpEncoded R;
			encode_alp_registers(x, e, f, lane, R);
			d.base    = R.base;
			d.bw      = static_cast<uint8_t>(R.bw);
			d.e       = static_cast<uint8_t>(e);
			d.f       = static_cast<uint8_t>(f);
			d.lbw     = 0;
			d.exc_cnt = static_cast<uint16_t>(R.cnt);
		} else {
			RdEncoded R;
			encode_rd_registers(x, *rgp, lane, R);
			d.base    = 0;
			d.bw      = rgp->rd_rbw;
			d.e       = 0;
			d.f       = 0;
			d.lbw     = rgp->rd_lbw;
			d.exc_cnt = static_cast<uint16_t>(R.cnt);
		}
		if (lane == 0) { descs[v] = d; }
	}
}

// ---- pass 2: offsets ------------------------------------------------------------------------------------
__device__ __forceinline__ void desc_sizes(const alpgpu_vector_desc& d, uint64_t& packed, uint64_t& exc) {
	if (d.scheme == ALPGPU_SCHEME_ALP) {
		packed = 128ull * d.bw;
		exc    = (10ull * d.exc_cnt + 7ull) & ~7ull;
	} else {
		packed = 128ull * (static_cast<uint64_t>(d.bw) + d.lbw);
		exc    = (4ull * d.exc_cnt + 7ull) & ~7ull;
	}
}

// block = 256 threads, 4 consecutive descriptors per thread
__global__ __launch_bounds__(256) void k_scan_tiles(alpgpu_vector_desc* __restrict__ descs, uint64_t n_vectors,
                                                    uint64_t* __restrict__ tile_sums /* [n_tiles][2] */) {
	__shared__ uint64_t sp[256], se[256];
	const uint64_t      v0 = static_cast<uint64_t>(blockIdx.x) * kScanTile + 4ull * threadIdx.x;
	uint64_t            p[4], e[4];
	uint64_t            tp = 0, te = 0;
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		p[i] = e[i] = 0;
		if (v0 + i < n_vectors) { desc_sizes(descs[v0 + i], p[i], e[i]); }
		tp += p[i];
		te += e[i];
	}
	sp[threadIdx.x] = tp;
	se[threadIdx.x] = te;
	__syncthreads();
	// Hillis-Steele inclusive scan over 256 partials
	for (int d = 1; d < 256; d <<= 1) {
		uint64_t ap = 0, ae = 0;
		if (static_cast<int>(threadIdx.x) >= d) {
			ap = sp[threadIdx.x - d];
			ae = se[threadIdx.x - d];
		}
		__syncthreads();
		sp[threadIdx.x] += ap;
		se[threadIdx.x] += ae;
		__syncthreads();
	}
	uint64_t op = sp[threadIdx.x] - tp, oe = se[threadIdx.x] - te; // exclusive
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		if (v0 + i < n_vectors) {
			descs[v0 + i].packed_off = op;
			descs[v0 + i].exc_off    = oe;
		}
		op += p[i];
		oe += e[i];
	}
	if (threadIdx.x == 255) {
		tile_sums[2ull * blockIdx.x]     = sp[255];
		tile_sums[2ull * blockIdx.x + 1] = se[255];
	}
}

// single workgroup of 1024 threads: exclusive scan of tile sums in place; totals + overflow flag
__global__ __launch_bounds__(1024) void k_scan_totals(uint64_t* __restrict__ tile_sums, uint64_t n_tiles, uint64_t packed_capacity,
                                                      uint64_t exc_capacity, uint64_t* __restrict__ totals) {
	__shared__ uint64_t sp[1024], se[1024];
	__shared__ uint64_t carry[2];
	if (threadIdx.x == 0) { carry[0] = carry[1] = 0; }
	__syncthreads();
	for (uint64_t t0 = 0; t0 < n_tiles; t0 += 1024) {
		const uint64_t t  = t0 + threadIdx.x;
		const uint64_t vp = t < n_tiles ? tile_sums[2 * t] : 0;
		const uint64_t ve = t < n_tiles ? tile_sums[2 * t + 1] : 0;
		sp[threadIdx.x]   = vp;
		se[threadIdx.x]   = ve;
		__syncthreads();
		for (int d = 1; d < 1024; d <<= 1) {
			uint64_t ap = 0, ae = 0;
			if (static_cast<int>(threadIdx.x) >= d) {
				ap = sp[threadIdx.x - d];
				ae = se[threadIdx.x - d];
			}
			__syncthreads();
			sp[threadIdx.x] += ap;
			se[threadIdx.x] += ae;
			__syncthreads();
		}
		if (t < n_tiles) {
			tile_sums[2 * t]     = carry[0] + sp[threadIdx.x] - vp;
			tile_sums[2 * t + 1] = carry[1] + se[threadIdx.x] - ve;
		}
		__syncthreads();
		if (threadIdx.x == 1023) {
			carry[0] += sp[1023];
			carry[1] += se[1023];
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		totals[0] = carry[0];
		totals[1] = carry[1];
		totals[2] = (carry[0] > packed_capacity || carry[1] > exc_capacity) ? 1 : 0;
		totals[3] = 0;
	}
}

// ---- pass 3: pack -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kWavesPerWg) void k_encode_pack(const double* __restrict__ in,
                                                                  const alpgpu_rowgroup_state* __restrict__ rgs,
                                                                  alpgpu_vector_desc* __restrict__ descs,
                                                                  const uint64_t* __restrict__ tile_bases, uint8_t* __restrict__ packed,
                                                                  uint8_t* __restrict__ excs, const uint64_t* __restrict__ totals,
                                                                  uint64_t n_vectors) {
	__shared__ EncodeLds lds[kWavesPerWg];
	if (totals[2] != 0) { return; } // capacity overflow: write nothing (reported through alpgpu_column_totals)
	const int      lane   = lane_id();
	const int      wave   = wave_in_wg();
	EncodeLds&     L      = lds[wave];
	const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kWavesPerWg;
	for (uint64_t v = static_cast<uint64_t>(blockIdx.x) * kWavesPerWg + wave; v < n_vectors; v += stride) {
		const alpgpu_rowgroup_state* rgp  = rgs + v / kRowgroup;
		alpgpu_vector_desc           d    = descs[v];
		const uint64_t               tile = v / kScanTile;
		d.packed_off += tile_bases[2 * tile];
		d.exc_off += tile_bases[2 * tile + 1];
		const VecIn x   = load_vector(in, v, lane);
		uint8_t*    dst = packed + d.packed_off;
		uint8_t*    rec = excs + d.exc_off;
		if (d.scheme == ALPGPU_SCHEME_ALP) {
			AlpEncoded R;
			encode_alp_registers(x, d.e, d.f, lane, R);
			// exception record: cnt x f64 original bits, then cnt x u16 positions, ascending position order
			if (R.cnt > 0) {
				uint64_t* ev   = reinterpret_cast<uint64_t*>(rec);
				uint16_t* ep   = reinterpret_cast<uint16_t*>(rec + 8ull * R.cnt);
				int       soff = 0;
#pragma unroll
				for (int m = 0; m < 8; ++m) {
#pragma unroll
					for (int j = 0; j < 2; ++j) {
						if (R.flags & (1u << (2 * m + j))) {
							const int r = exception_rank(R.ballot, R.flags, m, j, lane, soff);
							ev[r]       = static_cast<uint64_t>(__double_as_longlong(j == 0 ? x.x[m].x : x.x[m].y));
							ep[r]       = static_cast<uint16_t>(128 * m + 2 * lane + j);
						}
					}
					soff += __builtin_popcountll(R.ballot[m][0]) + __builtin_popcountll(R.ballot[m][1]);
				}
			}
			// (enc - base) -> LDS in natural order, then FFOR pack
			ulonglong2*    lv   = reinterpret_cast<ulonglong2*>(L.vals);
			const uint64_t base = static_cast<uint64_t>(R.base);
#pragma unroll
			for (int m = 0; m < 8; ++m) {
				lv[64 * m + lane] = make_ulonglong2(static_cast<uint64_t>(R.enc[m][0]) - base, static_cast<uint64_t>(R.enc[m][1]) - base);
			}
			wave_lds_sync();
			pack_u64_from_lds(L, R.bw, reinterpret_cast<ulonglong2*>(dst), lane);
		} else {
			RdEncoded R;
			encode_rd_registers(x, *rgp, lane, R);
			if (R.cnt > 0) {
				uint16_t* ev   = reinterpret_cast<uint16_t*>(rec);
				uint16_t* ep   = reinterpret_cast<uint16_t*>(rec + 2ull * R.cnt);
				int       soff = 0;
#pragma unroll
				for (int m = 0; m < 8; ++m) {
#pragma unroll
					for (int j = 0; j < 2; ++j) {
						if (R.flags & (1u << (2 * m + j))) {
							const int r = exception_rank(R.ballot, R.flags, m, j, lane, soff);
							ev[r]       = R.left[m][j];
							ep[r]       = static_cast<uint16_t>(128 * m + 2 * lane + j);
						}
					}
					soff += __builtin_popcountll(R.ballot[m][0]) + __builtin_popcountll(R.ballot[m][1]);
				}
			}
			ulonglong2* lv = reinterpret_cast<ulonglong2*>(L.vals);
#pragma unroll
			for (int m = 0; m < 8; ++m) { lv[64 * m + lane] = make_ulonglong2(R.right[m][0], R.right[m][1]); }
			wave_lds_sync();
			pack_u64_from_lds(L, d.bw, reinterpret_cast<ulonglong2*>(dst), lane);
			uint16_t idx[8][2];
#pragma unroll
			for (int m = 0; m < 8; ++m) {
				idx[m][0] = R.idx[m][0];
				idx[m][1] = R.idx[m][1];
			}
			pack_left_u16(idx, d.lbw, reinterpret_cast<uint32_t*>(dst + 128ull * d.bw), lane);
		}
		if (lane == 0) { descs[v] = d; }
		wave_lds_sync(); // LDS reuse by the next iteration
	}
}

// One workgroup per kWavesPerWg consecutive vectors, handed out by the hardware dispatcher in order (the kernels' loops
// then run once): measured 15-25 % more HBM bandwidth than a persistent grid-stride launch for this access pattern
// (profiles/r01_membw2_waves_per_vector.txt).  ALPGPU_ENCODE_PERSISTENT=1 restores the capped grid for A/B runs.
static unsigned grid_for(uint64_t n_vectors, int n_cus, int wgs_per_cu) {
	const uint64_t need = (n_vectors + kWavesPerWg - 1) / kWavesPerWg;
	static const bool persistent = std::getenv("ALPGPU_ENCODE_PERSISTENT") != nullptr;
	const uint64_t cap  = persistent ? static_cast<uint64_t>(n_cus) * wgs_per_cu : (1ull << 30);
	return static_cast<unsigned>(need < cap ? (need ? need : 1) : cap);
}

uint64_t encode_workspace_bytes(uint64_t n_vectors) { return ((n_vectors + kScanTile - 1) / kScanTile) * 16 + 16; }

int launch_encode_vectors(hipStream_t stream, const double* d_in, uint64_t n_vectors, const alpgpu_column* col, uint64_t* d_workspace,
                          int n_cus) {
	if (n_vectors == 0) {
		(void)hipMemsetAsync(col->d_totals, 0, 32, stream);
		return ALPGPU_OK;
	}
	const uint64_t n_tiles = (n_vectors + kScanTile - 1) / kScanTile;
	const dim3     block(64 * kWavesPerWg);
	hipLaunchKernelGGL(k_encode_analyze, dim3(grid_for(n_vectors, n_cus, 16)), block, 0, stream, d_in, col->d_rowgroups, col->d_vectors,
	                   n_vectors);
	hipLaunchKernelGGL(k_scan_tiles, dim3(static_cast<unsigned>(n_tiles)), dim3(256), 0, stream, col->d_vectors, n_vectors, d_workspace);
	hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, stream, d_workspace, n_tiles, col->packed_capacity, col->exc_capacity,
	                   col->d_totals);
	hipLaunchKernelGGL(k_encode_pack, dim3(grid_for(n_vectors, n_cus, 16)), block, 0, stream, d_in, col->d_rowgroups, col->d_vectors,
	                   d_workspace, col->d_packed, col->d_exc, col->d_totals, n_vectors);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

} // namespace alpgpu
