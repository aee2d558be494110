// encode_kernels.hip — whole-column ALP / ALP_RD vector encode for gfx950 (given the rowgroup states).
//
// Replaces, per vector (file:line relative to /root/reference):
//   alp::encoder<double>::encode          include/alp/encoder.hpp:402-418 (second-level sampling :241-305, encode_simdized :307-400)
//   alp::encoder<double>::analyze_ffor    include/alp/encoder.hpp:109-120
//   ffor::ffor (int64 / uint64 / uint16)  src/fastlanes_ffor.cpp:5-35 -> src/fastlanes_generated_ffor.cpp:29939, :29781
//   alp::rd_encoder<double>::encode       include/alp/rd.hpp:109-147
// and the caller's per-column loop (publication/source_code/bench_compression_ratio/alp.cpp:198-229).
//
// Variable-size output needs each vector's byte offsets.  Two forms, byte-identical output:
//   default  k_encode_fused (+ k_fused_finish): one pass; offsets from the in-kernel two-level look-back of
//            encode_lookback.hpp (description above the kernel)
//   fallback (ALPGPU_OPT_ENCODE_TWO_PASS):
//     k_encode_analyze : one wave per vector; picks (e,f), counts exceptions, finds base/bw -> descriptor sizes
//     k_scan_tiles     : exclusive scan of the sizes inside tiles of 1024 vectors, tile totals to a workspace
//     k_scan_totals    : one workgroup scans the tile totals, writes the stream totals / overflow flag
//     k_encode_pack    : one wave per vector; re-encodes with the chosen (e,f) and writes packed words,
//                        exception record and the final descriptor at the now-known offsets
//     (no communication between workgroups; reads the input twice and repeats the arithmetic)
// Offsets are the exclusive scan of the record sizes in vector order, so the output is byte-reproducible.
#include "encode_device.hpp"
#include "encode_lookback.hpp"
#include "launch.hpp"

namespace alpgpu {

constexpr int kScanTile = 1024;
#ifndef ALPGPU_ENC_PRIO
#define ALPGPU_ENC_PRIO 2
#endif

// ---- pass 1: analysis -----------------------------------------------------------------------------------
// `gate` (all kernels of this form): nullptr, or a word that must be non-zero for the kernel to do anything — the recovery route
// of a stalled single pass is enqueued behind every single-pass encode and runs only when the stall flag was raised.
__global__ __launch_bounds__(64 * kWavesPerWg) void k_encode_analyze(const double* __restrict__ in,
                                                                     const alpgpu_rowgroup_state* __restrict__ rgs,
                                                                     alpgpu_vector_desc* __restrict__ descs, uint64_t n_vectors, const uint64_t* __restrict__ gate) {
	__shared__ EncodeLds lds[kWavesPerWg];
	if (gate != nullptr && *gate == 0) { return; }
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
// block = 256 threads, 4 consecutive descriptors per thread
template <int VALUE_BYTES>
__global__ __launch_bounds__(256) void k_scan_tiles(alpgpu_vector_desc* __restrict__ descs, uint64_t n_vectors,
                                                    uint64_t* __restrict__ tile_sums /* [n_tiles][2] */, const uint64_t* __restrict__ gate) {
	__shared__ uint64_t sp[256], se[256];
	if (gate != nullptr && *gate == 0) { return; }
	const uint64_t      v0 = static_cast<uint64_t>(blockIdx.x) * kScanTile + 4ull * threadIdx.x;
	uint64_t            p[4], e[4];
	uint64_t            tp = 0, te = 0;
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		p[i] = e[i] = 0;
		if (v0 + i < n_vectors) { record_sizes<VALUE_BYTES>(descs[v0 + i], p[i], e[i]); }
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
                                                      uint64_t exc_capacity, uint64_t* __restrict__ totals, const uint64_t* __restrict__ gate) {
	__shared__ uint64_t sp[1024], se[1024];
	__shared__ uint64_t carry[2];
	if (gate != nullptr && *gate == 0) { return; }
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
                                                                  uint64_t n_vectors, const uint16_t* __restrict__ rd_order, const uint64_t* __restrict__ gate) {
	__shared__ EncodeLds lds[kWavesPerWg];
	if (gate != nullptr && *gate == 0) { return; }
	const int      lane   = lane_id();
	const int      wave   = wave_in_wg();
	EncodeLds&     L      = lds[wave];
	const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kWavesPerWg;
	if (totals[2] != 0) { // capacity overflow (reported through alpgpu_column_totals): write no stream bytes, leave descriptors a decoder can follow
		for (uint64_t v = static_cast<uint64_t>(blockIdx.x) * kWavesPerWg + wave; v < n_vectors; v += stride) {
			if (lane == 0) { descs[v] = empty_descriptor(); }
		}
		return;
	}
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
				uint64_t* ev = reinterpret_cast<uint64_t*>(rec);
				uint16_t* ep = reinterpret_cast<uint16_t*>(rec + 8ull * R.cnt);
				for_each_exception(R.ballot, lane, [&](int r, int m, int j) {
					ev[r] = static_cast<uint64_t>(__double_as_longlong(j == 0 ? x.x[m].x : x.x[m].y));
					ep[r] = static_cast<uint16_t>(128 * m + 2 * lane + j);
				});
				// the pad up to the record's 8-byte size is zero, as in the single-pass encode
				const int n_pos = static_cast<int>((((10u * static_cast<uint32_t>(R.cnt) + 7u) & ~7u) - 8u * static_cast<uint32_t>(R.cnt)) >> 1);
				if (R.cnt + lane < n_pos) { ep[R.cnt + lane] = 0; }
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
			encode_rd_registers(x, *rgp, lane, R, rd_order ? rd_order + (v / kRowgroup) * ALPGPU_RD_ORDER_STRIDE : nullptr);
			if (R.cnt > 0) {
				uint16_t* ev = reinterpret_cast<uint16_t*>(rec);
				uint16_t* ep = reinterpret_cast<uint16_t*>(rec + 2ull * R.cnt);
				for_each_exception(R.ballot, lane, [&](int r, int m, int j) {
					ev[r] = R.left[m][j];
					ep[r] = static_cast<uint16_t>(128 * m + 2 * lane + j);
				});
				const int n_pos = static_cast<int>((((4u * static_cast<uint32_t>(R.cnt) + 7u) & ~7u) - 2u * static_cast<uint32_t>(R.cnt)) >> 1);
				if (R.cnt + lane < n_pos) { ep[R.cnt + lane] = 0; } // zero pad, as in the single-pass encode
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

// ---- single pass: analysis, ordered offsets (decoupled look-back) and pack in ONE kernel -------------------------------
// The two-pass form above reads the input twice and does the encode arithmetic twice.  Here a workgroup (tile) of kFusedWaves
// wavefronts = kFusedWaves consecutive vectors
//   1. encodes its vectors (registers -> LDS), knows their sizes,
//   2. publishes the tile's size as ONE 64-bit status word {flag | packed 128-B units | exception 8-B units},
//   3. packs every vector into registers and lays out its exception record in LDS (neither depends on where it will go),
//   4. wavefront 0 finds the tile's exclusive prefix with the two-level look-back of encode_lookback.hpp and hands it to the
//      other wavefronts through LDS words (no workgroup barrier is involved),
//   5. every wavefront stores its packed words / exception record / descriptor at the now-known offsets: contiguous stores only.
// A wavefront's life is a chain of dependent memory round trips (~1 us each under this streaming load), not arithmetic: the
// rowgroup state and the column's running totals are therefore read once, up front, next to the input.
// Offsets are therefore the same vector-order exclusive scan as in the two-pass form: the output is byte-identical.
// Status words are written with one agent-scope relaxed atomic store (the data IS the flag) and polled with agent-scope
// relaxed atomic loads (cdna_hip_programming.md §6 G16, recipe R2).  Forward progress needs the predecessor tiles to be
// resident or finished, which the in-order dispatch of a 1-D grid provides in practice but HIP does not promise:
// every spin is bounded, a stall sets totals[3] and nothing of a stalled tile is written; the two-pass kernels enqueued behind
// every single-pass encode (gated on that flag: launch_encode_recovery) then redo the column on the same stream.  The field widths bound one launch to kFusedMaxVectors vectors;
// longer columns are chained launch by launch through totals[0..1].  Where a wavefront's time goes: profiles/r01_fused_phases.txt.
#ifdef ALPGPU_FUSED_TIMING // experiment (tools/fused_phases.py): per-wavefront phase marks, 10 ns ticks since the wavefront started
__device__ int32_t* g_phase_buf = nullptr; // [n_vectors][8]
#define PHASE_WAIT_MEM() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define PHASE_MARK(k)                                                                                                        \
	do {                                                                                                                 \
		if (g_phase_buf != nullptr && lane == 0 && live) { g_phase_buf[8 * vl + (k)] = static_cast<int32_t>(wall_clock64() - phase_t0); } \
	} while (0)
#else
#define PHASE_WAIT_MEM()
#define PHASE_MARK(k)
#endif
__global__ __launch_bounds__(64 * kFusedWaves) void k_encode_fused(const double* __restrict__ in,
                                                                   const alpgpu_rowgroup_state* __restrict__ rgs,
                                                                   alpgpu_vector_desc* __restrict__ descs, uint8_t* __restrict__ packed,
                                                                   uint8_t* __restrict__ excs, uint64_t* __restrict__ status,
                                                                   uint64_t* __restrict__ totals, uint64_t packed_capacity,
                                                                   uint64_t exc_capacity, uint64_t v_first, uint64_t n_vectors_launch,
                                                                   const uint16_t* __restrict__ rd_order, uint32_t spin_limit, uint32_t async_states) {
	// instruction-issue priority over the persistent rowgroup search that may share the CU (alpgpu_encode_f64 on a long column): the
	// search then takes the issue slots this kernel leaves idle instead of competing for them (3.28 -> 3.14 ms per 1 Mi vectors)
	__builtin_amdgcn_s_setprio(ALPGPU_ENC_PRIO);
	__shared__ EncodeLds lds[kFusedWaves];
	__shared__ uint64_t  s_size[kFusedWaves]; // per vector: (packed units << 31) | exception units
	__shared__ uint64_t  s_excl;              // tile's exclusive prefix in the same packing, or ~0 on a stall
	__shared__ uint32_t  s_count;             // worker wavefronts that have posted their size
	__shared__ uint32_t  s_ready;             // set by the scout once s_excl is valid
	const int            lane = lane_id();
	const int            wave = wave_in_wg();
	const uint64_t       tile = blockIdx.x;
#ifdef ALPGPU_FUSED_TIMING
	const uint64_t phase_t0 = wall_clock64();
#endif
	if (threadIdx.x == 0) {
		s_count = 0;
		s_ready = 0;
	}
	__syncthreads(); // the only workgroup barrier: nothing waits on memory here

	// ---- the workers: one vector each -------------------------------------------------------------------------------
	EncodeLds&     L    = lds[wave];
	const uint64_t vl   = tile * kFusedWaves + wave; // vector index inside this launch
	const bool     live = vl < n_vectors_launch;
	const uint64_t v    = v_first + vl;
	// ---- 1. encode into registers / LDS ----
	VecIn              x;
	alpgpu_vector_desc d;
	uint64_t           acc0 = 0, acc1 = 0; // ALP_RD: packed left streams of this lane's two lane64 columns
	uint64_t           ballots[8][2];
	uint64_t           pvals[8][2]; // what gets packed: value - base (ALP), right parts (ALP_RD)
	int                cnt = 0;
#pragma unroll
	for (int m = 0; m < 8; ++m) { pvals[m][0] = pvals[m][1] = 0; }
	d.packed_off = d.exc_off = 0;
	d.base                   = 0;
	d.bw = d.e = d.f = d.lbw = 0;
	d.exc_cnt = d.scheme = 0;
	// The 8 KiB everything waits for are requested FIRST.  Requested behind the rowgroup state (whose unpacking into scalar
	// registers makes the compiler wait for it on the spot) and behind the scalar read of the running totals, the vector loads
	// started two memory round trips late — 20 % of a tile's life.  A wavefront past the end of the column reads the launch's
	// first vector instead (nothing of it is stored).
	const uint64_t v_read = live ? v : v_first;
	const alpgpu_rowgroup_state* rg_ptr  = rgs + v_read / kRowgroup;
	const uint32_t               st_word = async_states ? rowgroup_state_poll_begin(rg_ptr, lane) : reinterpret_cast<const uint32_t*>(rg_ptr)[lane & 7]; // in FRONT of the vector's loads (alp_device.hpp)
#ifdef ALPGPU_EXPERIMENT_ENC_WRAP_TRAFFIC // timing experiment (profiles/r03_encode_levers.txt): every vector's 8 KiB come from the first 8192 vectors and go
	// to the first 64 MiB / 8 MiB of the streams — the same arithmetic on a column that repeats its first 8192 vectors, without the HBM traffic
	x = load_vector(in, v_read & 8191ull, lane);
#else
	x                     = load_vector(in, v_read, lane);
#endif
	// the rowgroup's state, once, into registers (alp_device.hpp); its read is in flight together with the input's
	// async_states (alpgpu_encode_f64 on a long column): the states are being published by the persistent rowgroup search that runs
	// beside this kernel on a second stream; a wavefront polls its rowgroup's tag (nearly always set long before: the search runs
	// rowgroups ahead).  A state that does not arrive within the spin limit is a stall like a look-back that gives up: flag, no output
	// of this tile, the recovery route (behind both streams) re-encodes.
	bool                         state_ok = true;
	const alpgpu_rowgroup_state  st  = async_states ? rowgroup_state_poll_finish(rg_ptr, st_word, lane, spin_limit >> 4, state_ok) : unpack_rowgroup_state(st_word);
	const alpgpu_rowgroup_state* rgp = &st;
	if (!state_ok) { // wave-uniform
		if (lane == 0) { status_store(totals + 3, 1ull); }
		return; // (the tile's other wavefronts and its successors run into their own spin limits: status words never appear)
	}
	if (live) {
		PHASE_WAIT_MEM();
		PHASE_MARK(0);
		d.scheme = rgp->scheme;
		if (rgp->scheme == ALPGPU_SCHEME_ALP) {
			int e, f;
#ifdef ALPGPU_ABLATE_SECOND
			if (false) {
#else
			if (rgp->k > 1) {
#endif
				second_level_select(x, rgp, L, lane, e, f);
			} else {
				e = rgp->combos[0];
				f = rgp->combos[1];
			}
			PHASE_MARK(1);
			AlpEncoded R;
			encode_alp_registers(x, e, f, lane, R);
			PHASE_MARK(2);
			d.base = R.base, d.bw = static_cast<uint8_t>(R.bw), d.e = static_cast<uint8_t>(e), d.f = static_cast<uint8_t>(f);
			cnt = R.cnt;
			const uint64_t base = static_cast<uint64_t>(R.base);
#pragma unroll
			for (int m = 0; m < 8; ++m) {
				pvals[m][0]   = static_cast<uint64_t>(R.enc[m][0]) - base;
				pvals[m][1]   = static_cast<uint64_t>(R.enc[m][1]) - base;
				ballots[m][0] = R.ballot[m][0];
				ballots[m][1] = R.ballot[m][1];
			}
		} else {
			RdEncoded R;
			encode_rd_registers(x, *rgp, lane, R, rd_order ? rd_order + (v / kRowgroup) * ALPGPU_RD_ORDER_STRIDE : nullptr, async_states != 0);
			d.bw = rgp->rd_rbw, d.lbw = rgp->rd_lbw;
			cnt = R.cnt;
			const int      lbw   = d.lbw;
			const uint64_t lmask = (1ull << lbw) - 1ull;
#pragma unroll
			for (int m = 0; m < 8; ++m) {
				pvals[m][0]   = R.right[m][0];
				pvals[m][1]   = R.right[m][1];
				ballots[m][0] = R.ballot[m][0];
				ballots[m][1] = R.ballot[m][1];
				const int row     = 2 * m + (lane >> 5);
				acc0 |= (static_cast<uint64_t>(R.idx[m][0]) & lmask) << (row * lbw);
				acc1 |= (static_cast<uint64_t>(R.idx[m][1]) & lmask) << (row * lbw);
			}
			acc0 |= static_cast<uint64_t>(__shfl_xor(static_cast<long long>(acc0), 32));
			acc1 |= static_cast<uint64_t>(__shfl_xor(static_cast<long long>(acc1), 32));
		}
		d.exc_cnt = static_cast<uint16_t>(cnt);
	}
	uint64_t my_p = 0, my_e = 0; // bytes
	if (live) { record_sizes<8>(d, my_p, my_e); }
	// post this vector's size; the last worker to arrive publishes the tile's aggregate (so successors never wait for
	// this tile's own look-back), then everybody waits — on LDS words only — for the scout's exclusive prefix
	if (lane == 0) {
		s_size[wave] = status_pack(0, my_p >> 7, my_e >> 3);
		const uint32_t arrived = __hip_atomic_fetch_add(&s_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (arrived == kFusedWaves - 1) {
			uint64_t aggregate = 0;
#pragma unroll
			for (int w = 0; w < kFusedWaves; ++w) { aggregate += s_size[w]; }
			status_store(status + tile, kFlagAggregate | aggregate); // the tile word is written exactly once
		}
	}
	// bytes used by earlier launches of this column: constant while this launch runs (k_fused_finish updates them); read here, so
	// that the read is neither in front of the vector's loads nor behind the wait for the ordered offset
	const uint64_t base_p = totals[0], base_e = totals[1];
	// The packed words do not depend on where they will be stored: build them now, in registers, while the ordered offset is
	// still on its way (a wavefront otherwise idles ~40 % of its life here: profiles/r01_fused_phases.txt); wavefront 0 packs
	// first as well, its look-back then finds more of its predecessors already posted.
	PHASE_MARK(3);
	PackedUnits packed_units;
#ifdef ALPGPU_ABLATE_PACK
#pragma unroll
	for (int t = 0; t < 8; ++t) { packed_units.acc[t] = ull2v{0ull, 0ull}; }
#elif defined(ALPGPU_PACK_GATHER) // the round-2 form (A/B): values staged in natural order, every output word gathers its rows
	{
		ulonglong2* lv = reinterpret_cast<ulonglong2*>(L.vals);
#pragma unroll
		for (int m = 0; m < 8; ++m) { lv[64 * m + lane] = make_ulonglong2(pvals[m][0], pvals[m][1]); }
		wave_lds_sync();
		pack_u64_units(L.vals, d.bw, lane, packed_units);
	}
#else
	pack_u64_scatter(L.vals, pvals, d.bw, lane, packed_units);
#endif
	PHASE_MARK(4);
	// Likewise the exception record: its image is laid out in the (now free) staging area, so that after the wait it leaves as a
	// few contiguous 8-byte-per-lane stores instead of two one-lane stores per exception step, and the input values need not
	// stay in registers across the wait.  The values always fit (8 B x 1024); the positions follow them when the whole record
	// fits (<= 819 exceptions; always for ALP_RD), else they are written from the ballots after the wait.  Pad bytes are zero.
	const bool     alp_rec      = d.scheme == ALPGPU_SCHEME_ALP;
	const uint32_t val_bytes    = alp_rec ? 8u * static_cast<uint32_t>(cnt) : 2u * static_cast<uint32_t>(cnt);
	const bool     pos_staged   = my_e <= sizeof(L.vals);
	const uint32_t staged_bytes = pos_staged ? static_cast<uint32_t>(my_e) : val_bytes;
#ifdef ALPGPU_ABLATE_EXC
	if (false) {
#else
	if (cnt > 0) {
#endif
		uint8_t* img = reinterpret_cast<uint8_t*>(L.vals);
		wave_lds_sync(); // the pack's reads of the staging area are issued; the LDS executes one wavefront's operations in order
		if (lane == 0) { reinterpret_cast<uint64_t*>(img)[(staged_bytes >> 3) - 1] = 0ull; } // the pad lives in the last word
		wave_lds_sync();
		const int rbw = d.bw;
		for_each_exception(ballots, lane, [&](int r, int m, int j) {
			const uint64_t bits = static_cast<uint64_t>(__double_as_longlong(j == 0 ? x.x[m].x : x.x[m].y));
			const uint16_t pos  = static_cast<uint16_t>(128 * m + 2 * lane + j);
			if (alp_rec) {
				reinterpret_cast<uint64_t*>(img)[r] = bits;
			} else {
				reinterpret_cast<uint16_t*>(img)[r] = static_cast<uint16_t>(bits >> rbw);
			}
			if (pos_staged) { reinterpret_cast<uint16_t*>(img + val_bytes)[r] = pos; }
		});
		wave_lds_sync();
	}
	PHASE_MARK(5);
	if (wave == 0) { tile_lookback(tile, status, totals, s_size, &s_count, &s_excl, &s_ready, lane, spin_limit); }
	{
		uint32_t spins = 0;
		while (__hip_atomic_load(&s_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
			if (++spins > 64u * kSpinLimit) { return; }
			__builtin_amdgcn_s_sleep(2);
		}
	}
	uint64_t local = 0;
#pragma unroll
	for (int w = 0; w < kFusedWaves; ++w) { local += w < wave ? s_size[w] : 0; }
	const uint64_t excl = s_excl;
	if (excl == ~0ull) { return; } // stalled: nothing of this tile is written
	PHASE_MARK(6);

	// ---- 4. write at the final offsets ----
	const uint64_t pre    = excl + local;
	d.packed_off          = base_p + ((pre >> 31) & 0x7FFFFFFFull) * 128ull;
	d.exc_off             = base_e + (pre & 0x7FFFFFFFull) * 8ull;
	if (!live) { return; }
	if (d.packed_off + my_p > packed_capacity || d.exc_off + my_e > exc_capacity) {
		// this vector does not fit: report, write nothing past the buffers, and leave a descriptor that a decoder can follow
		// without leaving them (bit width 0, no exceptions: the column's content is unspecified, its extents are not)
		if (lane == 0) {
			__hip_atomic_store(totals + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			descs[v] = empty_descriptor();
		}
		return;
	}
#ifdef ALPGPU_EXPERIMENT_ENC_WRAP_TRAFFIC
	uint8_t* dst = packed + (d.packed_off & ((64ull << 20) - 1));
	uint8_t* rec = excs + (d.exc_off & ((8ull << 20) - 1));
#else
	uint8_t* dst = packed + d.packed_off;
	uint8_t* rec = excs + d.exc_off;
#endif
#ifdef ALPGPU_ABLATE_STORES // timing experiment: everything but the output stores
	if (packed_capacity == 1) {
#endif
#ifdef ALPGPU_ABLATE_EXC
	if (false) {
#else
	if (cnt > 0) {
#endif
		const uint64_t* img64 = reinterpret_cast<const uint64_t*>(L.vals);
		uint64_t*       rec64 = reinterpret_cast<uint64_t*>(rec);
		const int       n_w   = static_cast<int>(staged_bytes >> 3);
		for (int w = lane; w < n_w; w += 64) {
#ifndef ALPGPU_ENC_NT_STORE
			rec64[w] = img64[w];
#else
			__builtin_nontemporal_store(img64[w], rec64 + w);
#endif
		}
		if (!pos_staged) { // > 819 exceptions in an ALP vector: positions (and their pad) straight from the ballots
			uint16_t* rpos = reinterpret_cast<uint16_t*>(rec + val_bytes);
			for_each_exception(ballots, lane, [&](int r, int m, int j) { rpos[r] = static_cast<uint16_t>(128 * m + 2 * lane + j); });
			const int n_pos = static_cast<int>((my_e - val_bytes) >> 1);
			if (cnt + lane < n_pos) { rpos[cnt + lane] = 0; }
		}
	}
	store_packed_units(packed_units, d.bw, reinterpret_cast<ull2v*>(dst), lane);
	if (d.scheme != ALPGPU_SCHEME_ALP && lane < 32) {
		uint32_t* out32 = reinterpret_cast<uint32_t*>(dst + 128ull * d.bw);
		for (int k = 0; k < d.lbw; ++k) {
			out32[32 * k + lane] = (static_cast<uint32_t>(acc0 >> (16 * k)) & 0xFFFFu) | ((static_cast<uint32_t>(acc1 >> (16 * k)) & 0xFFFFu) << 16);
		}
	}
#ifdef ALPGPU_ABLATE_STORES
	}
#endif
	if (lane == 0) { descs[v] = d; }
	PHASE_MARK(7);
}

#ifdef ALPGPU_FUSED_TIMING
extern "C" __attribute__((visibility("default"))) int alpgpu_debug_fused_phases(void* buf) {
	return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

// publishes the running totals after a fused launch (single thread; keeps totals[0..1] stable while the launch runs) and
// latches the stall flag into totals[6], the gate of the recovery kernels (k_scan_totals clears totals[3] on its way)
// clear_rgs != nullptr (the last launch of an encode whose states were published beside it): the publishing tags (pad bytes) of the
// n_clear states are reset, so that the column's states are the reference's bytes (256 threads stride over them)
// reserve != nullptr (an UNORDERED launch, encode_lean_kernels.hip): nobody ran a look-back; the launch's bytes are what its tiles reserved
__global__ __launch_bounds__(256) void k_fused_finish(uint64_t* __restrict__ totals, alpgpu_rowgroup_state* __restrict__ clear_rgs, uint64_t n_clear,
                                                      const uint64_t* __restrict__ reserve) {
	if (threadIdx.x == 0) {
		if (reserve != nullptr) {
			const uint64_t incl = *reserve;
			totals[4]           = totals[0] + ((incl >> 31) & 0x7FFFFFFFull) * 128ull;
			totals[5]           = totals[1] + (incl & 0x7FFFFFFFull) * 8ull;
		}
		totals[0] = totals[4];
		totals[1] = totals[5];
		if (totals[3] != 0) { totals[6] = 1; }
	}
	if (clear_rgs != nullptr) {
		for (uint64_t i = threadIdx.x; i < n_clear; i += 256) { clear_rgs[i].pad = 0; }
	}
}

// ---- the single-pass encode's memory traffic and nothing else (alpgpu_debug_traffic_probe) ---------------------------------------
// Same launch shape (one wavefront per vector, kFusedWaves per workgroup, grid = tiles, in order): every vector is read once and
// `units` 16-byte units are written at v * units, the stored data depending on everything that was read.  bench.py quotes its
// rate as the measured ceiling for this read/write mix next to the nominal HBM peak.
__global__ __launch_bounds__(64 * kFusedWaves) void k_traffic_probe(const ull2v* __restrict__ in, ull2v* __restrict__ out, uint64_t n_vectors, uint32_t units) {
	const int      lane = lane_id();
	const uint64_t v    = static_cast<uint64_t>(blockIdx.x) * kFusedWaves + wave_in_wg();
	if (v >= n_vectors) { return; }
	ull2v acc = {v, 1ull};
#pragma unroll
	for (int m = 0; m < 8; ++m) { acc += in[v * 512 + 64 * m + lane]; }
	if (units == 0) { // read-only stream: every lane's loads feed ONE 8-byte store per vector (a store that depends on one lane only lets the compiler sink the loads into that lane)
		unsigned long long x = acc.x ^ acc.y;
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1) { x += __shfl_xor(x, d); }
		if (lane == 0) { reinterpret_cast<unsigned long long*>(out)[v] = x; }
		return;
	}
	ull2v* dst = out + v * units;
	for (uint32_t u = lane; u < units; u += 64) {
		ull2v o = acc;
		o.x += u;
		dst[u] = o;
	}
}
int launch_traffic_probe(hipStream_t stream, const void* d_in, void* d_out, uint64_t n_vectors, uint32_t write_bytes) {
	const uint64_t n_tiles = (n_vectors + kFusedWaves - 1) / kFusedWaves;
	hipLaunchKernelGGL(k_traffic_probe, dim3(static_cast<unsigned>(n_tiles)), dim3(64 * kFusedWaves), 0, stream, static_cast<const ull2v*>(d_in),
	                   static_cast<ull2v*>(d_out), n_vectors, write_bytes / 16u);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

uint64_t encode_workspace_bytes(uint64_t n_vectors) {
	const uint64_t two_pass = ((n_vectors + kScanTile - 1) / kScanTile) * 16 + 16;
	const uint64_t per_launch = n_vectors < kFusedMaxVectors ? n_vectors : kFusedMaxVectors;
	const uint64_t fused    = lookback_words((per_launch + kFusedWaves - 1) / kFusedWaves) * 8 + 64;
	return two_pass > fused ? two_pass : fused;
}

int launch_encode_reset_totals(hipStream_t stream, const alpgpu_column* col) {
	// d_totals: [0] packed bytes, [1] exception bytes, [2] overflow, [3] look-back stall, [4..5] running totals of the launch in flight,
	// [6] gate of the recovery kernels (a stall happened), [7] unused
	return hipMemsetAsync(col->d_totals, 0, 64, stream) == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

// vectors [v_first, v_first + n_range): continues the streams where d_totals[0..1] say the previous range ended
// async_head / async_join (with async_states): the events behind the head of the search and behind the persistent rest; the stream
// waits for the first in front of the first encode launch, for the second in front of the LAST finish kernel, which then clears the tags
void launch_k_encode_lean(hipStream_t stream, unsigned n_tiles, const double* d_in, const alpgpu_column* col, uint64_t* d_workspace, uint64_t first, uint64_t n_launch,
                          uint32_t spin_limit, uint32_t async_states, bool unordered); // encode_lean_kernels.hip
int launch_encode_fused_range(hipStream_t stream, const double* d_in, const alpgpu_column* col, uint64_t* d_workspace, uint64_t v_first, uint64_t n_range,
                              bool force_stall, bool async_states, hipEvent_t async_join, hipEvent_t async_head, int kernel) {
	for (uint64_t first = v_first; first < v_first + n_range; first += kFusedMaxVectors) {
		const uint64_t left     = v_first + n_range - first;
		const uint64_t n_launch = left < kFusedMaxVectors ? left : kFusedMaxVectors;
		const uint64_t n_tiles  = (n_launch + kFusedWaves - 1) / kFusedWaves;
		if (hipMemsetAsync(d_workspace, 0, (lookback_words(n_tiles) + 1) * 8, stream) != hipSuccess) { return ALPGPU_ERR_HIP; } // (+ the unordered form's counter word)
		if (async_states && first == v_first && async_head != nullptr) { // the head of the search (side stream) is what the first tiles need
			if (hipStreamWaitEvent(stream, async_head, 0) != hipSuccess) { return ALPGPU_ERR_HIP; }
		}
		const bool      unordered = kernel == (ALPGPU_ENCODE_KERNEL_LEAN | kEncodeUnorderedFlag); // (the lean kernel only; force_stall has nothing to stall there)
		const uint64_t* reserve   = unordered ? d_workspace + lookback_words(n_tiles) : nullptr;
		if ((kernel & ~kEncodeUnorderedFlag) == ALPGPU_ENCODE_KERNEL_LEAN) {
			launch_k_encode_lean(stream, static_cast<unsigned>(n_tiles), d_in, col, d_workspace, first, n_launch, force_stall ? 0u : kSpinLimit, async_states ? 1u : 0u, unordered);
		} else {
			hipLaunchKernelGGL(k_encode_fused, dim3(static_cast<unsigned>(n_tiles)), dim3(64 * kFusedWaves), 0, stream, d_in, col->d_rowgroups,
			                   col->d_vectors, col->d_packed, col->d_exc, d_workspace, col->d_totals, col->packed_capacity, col->exc_capacity, first,
			                   n_launch, col->d_rd_order, force_stall ? 0u : kSpinLimit, async_states ? 1u : 0u);
		}
		const bool last = first + n_launch >= v_first + n_range;
		if (async_states && last) {
			if (hipStreamWaitEvent(stream, async_join, 0) != hipSuccess) { return ALPGPU_ERR_HIP; }
			hipLaunchKernelGGL(k_fused_finish, dim3(1), dim3(256), 0, stream, col->d_totals, col->d_rowgroups, col->n_rowgroups, reserve);
		} else {
			hipLaunchKernelGGL(k_fused_finish, dim3(1), dim3(256), 0, stream, col->d_totals, static_cast<alpgpu_rowgroup_state*>(nullptr), 0ull, reserve);
		}
	}
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

int launch_encode_fused(hipStream_t stream, const double* d_in, uint64_t n_vectors, const alpgpu_column* col, uint64_t* d_workspace, bool force_stall,
                        bool async_states, hipEvent_t async_join, hipEvent_t async_head, int kernel) {
	if (launch_encode_reset_totals(stream, col) != ALPGPU_OK) { return ALPGPU_ERR_HIP; }
	return launch_encode_fused_range(stream, d_in, col, d_workspace, 0, n_vectors, force_stall, async_states, async_join, async_head, kernel);
}

// the scan of the two-pass form (shared with the float column kernels): descriptor sizes -> offsets, totals, overflow flag
int launch_scan_offsets(hipStream_t stream, const alpgpu_column* col, uint64_t n_vectors, uint64_t* d_workspace, bool f32, const uint64_t* gate) {
	const uint64_t n_tiles = (n_vectors + kScanTile - 1) / kScanTile;
	if (f32) {
		hipLaunchKernelGGL(k_scan_tiles<4>, dim3(static_cast<unsigned>(n_tiles)), dim3(256), 0, stream, col->d_vectors, n_vectors, d_workspace, gate);
	} else {
		hipLaunchKernelGGL(k_scan_tiles<8>, dim3(static_cast<unsigned>(n_tiles)), dim3(256), 0, stream, col->d_vectors, n_vectors, d_workspace, gate);
	}
	hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, stream, d_workspace, n_tiles, col->packed_capacity, col->exc_capacity, col->d_totals, gate);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

// The two-pass form.  gate == nullptr: unconditionally (ALPGPU_OPT_ENCODE_TWO_PASS), one workgroup per kWavesPerWg vectors.
// gate != nullptr: the recovery route behind a single-pass encode — a capped grid (the kernels stride over the column), so
// that the four launches cost microseconds when the gate is closed, which is always unless the look-back stalled.
int launch_encode_vectors(hipStream_t stream, const double* d_in, uint64_t n_vectors, const alpgpu_column* col, uint64_t* d_workspace,
                          int n_cus, const uint64_t* gate) {
	if (n_vectors == 0) {
		if (gate == nullptr) { (void)hipMemsetAsync(col->d_totals, 0, 64, stream); }
		return ALPGPU_OK;
	}
	const dim3     block(64 * kWavesPerWg);
	const uint64_t need = (n_vectors + kWavesPerWg - 1) / kWavesPerWg;
	const uint64_t cap  = gate != nullptr ? static_cast<uint64_t>(n_cus) * 16 : (1ull << 30);
	const dim3     grid(static_cast<unsigned>(need < cap ? need : cap));
	hipLaunchKernelGGL(k_encode_analyze, grid, block, 0, stream, d_in, col->d_rowgroups, col->d_vectors, n_vectors, gate);
	if (launch_scan_offsets(stream, col, n_vectors, d_workspace, false, gate) != ALPGPU_OK) { return ALPGPU_ERR_HIP; }
	hipLaunchKernelGGL(k_encode_pack, grid, block, 0, stream, d_in, col->d_rowgroups, col->d_vectors, d_workspace, col->d_packed, col->d_exc, col->d_totals,
	                   n_vectors, col->d_rd_order, gate);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

} // namespace alpgpu
