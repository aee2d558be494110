// encode_f32_kernels.hip — whole-column ALP / ALP_RD vector encode for gfx950, single precision.
//
// Replaces, per vector (file:line relative to /root/reference):
//   alp::encoder<float>::encode           include/alp/encoder.hpp:402-418 (second-level sampling :241-305, encode_simdized :307-400)
//   alp::encoder<float>::analyze_ffor     include/alp/encoder.hpp:109-120
//   ffor::ffor (int32 / uint32 / uint16)  include/fastlanes/ffor.hpp:7-15 -> src/fastlanes_generated_ffor.cpp:1776-7378, :357-1775
//   alp::rd_encoder<float>::encode        include/alp/rd.hpp:109-147
//
// Single pass, same scheme as k_encode_fused (encode_kernels.hip): one wavefront per 1024-value vector, kFusedWaves vectors per
// workgroup (tile), output offsets in vector order from the decoupled look-back of encode_lookback.hpp.  The same kernel body,
// compiled in two more modes, is the two-pass form (ALPGPU_OPT_ENCODE_TWO_PASS and the recovery route of a stalled single pass):
// ANALYZE stops after the sizes are known and leaves them in the descriptors, PACK takes its offsets from the scan of those
// (k_scan_tiles / k_scan_totals of encode_kernels.hip) instead of the look-back.
// Ownership: lane L holds the value quads i = 256*m + 4*L + j (m = 0..3, j = 0..3), so the 4 KiB input is read with four
// 1-KiB-contiguous 16-byte-per-lane loads; in the FastLanes u32 layout (alp_device_f32.hpp) a quad is one 16-byte unit:
// row = 8*m + (L >> 3), unit column a = L & 7.
#include "encode_f32_device.hpp"
#include "encode_lookback.hpp"
#include "rd_exception_index.hpp"
#include "launch.hpp"

namespace alpgpu {

enum FusedMode { kSinglePass = 0, kAnalyze = 1, kPack = 2 };
#ifndef ALPGPU_F32_ENC_OCC
#define ALPGPU_F32_ENC_OCC 8 // __launch_bounds__' second argument for the single pass: wavefronts per SIMD the register budget is sized for (round 4: 62 VGPRs, four tiles per CU)
#endif
constexpr int kScanTileF32 = 1024; // = kScanTile of encode_kernels.hip: vectors per tile of the two-pass scan

struct FusedSharedF32 {
	EncodeLdsF32 lds[kFusedWaves];
	uint64_t     s_size[kFusedWaves];
	uint64_t     s_excl;
	uint32_t     s_count;
	uint32_t     s_ready;
};

// one tile (kFusedWaves vectors, one per wavefront) of k_encode_fused_f32
// Measurement builds (-DALPGPU_F32_STOP_AT=n, as ALPGPU_LEAN_STOP_AT in encode_lean_kernels.hip): every wavefront of the single pass ends behind
// stage n with one dependent store; the column is garbage, the counters of successive builds difference into instructions per stage.
#ifdef ALPGPU_F32_STOP_AT
#define ALPGPU_F32_STOP(n, expr)                                                                    \
	if (MODE == kSinglePass && ALPGPU_F32_STOP_AT == (n)) {                                         \
		if (lane == 0) { arg_descs()[v_read].base = static_cast<int64_t>(expr); }                   \
		return;                                                                                     \
	}
#else
#define ALPGPU_F32_STOP(n, expr)
#endif
// UNORDERED (single pass only; ALPGPU_OPT_ENCODE_UNORDERED, encode_lean_kernels.hip): the tile reserves its bytes with one atomic add instead of the look-back
template <int MODE, bool UNORDERED = false>
__device__ __forceinline__ void encode_tile_f32(FusedSharedF32& S, const uint64_t tile, const float* __restrict__ in, const alpgpu_rowgroup_state* __restrict__ rgs,
                                                alpgpu_vector_desc* __restrict__ descs, uint8_t* __restrict__ packed, uint8_t* __restrict__ excs,
                                                uint64_t* __restrict__ status, uint64_t* __restrict__ totals, uint64_t packed_capacity, uint64_t exc_capacity,
                                                uint64_t v_first, uint64_t n_vectors_launch, const uint16_t* __restrict__ rd_order, uint32_t spin_limit,
                                                uint32_t async_states) {
	if (MODE == kPack && totals[2] != 0) { // capacity overflow (reported through alpgpu_column_totals): no stream bytes, descriptors a decoder can follow
		const uint64_t vo = v_first + tile * kFusedWaves + (threadIdx.x >> 6);
		if ((threadIdx.x & 63) == 0 && vo < v_first + n_vectors_launch) { descs[vo] = empty_descriptor(); }
		return;
	}
	// single pass: the arguments needed only behind the wait come from the kernarg segment where they are used (alp_device.hpp: late_kernel_arg;
	// -DALPGPU_F32_ARGS_AT_ENTRY: the old form).  The other two modes are not short of scalar registers.
#ifdef ALPGPU_F32_ARGS_AT_ENTRY
	constexpr bool kLate = false;
#else
	constexpr bool kLate = MODE == kSinglePass;
#endif
	auto arg_descs      = [&]() { if constexpr (kLate) { return late_kernel_arg<alpgpu_vector_desc*>(kArgDescs); } else { return descs; } };
	auto arg_packed     = [&]() { if constexpr (kLate) { return late_kernel_arg<uint8_t*>(kArgPacked); } else { return packed; } };
	auto arg_excs       = [&]() { if constexpr (kLate) { return late_kernel_arg<uint8_t*>(kArgExcs); } else { return excs; } };
	auto arg_packed_cap = [&]() { if constexpr (kLate) { return late_kernel_arg<uint64_t>(kArgPackedCap); } else { return packed_capacity; } };
	auto arg_exc_cap    = [&]() { if constexpr (kLate) { return late_kernel_arg<uint64_t>(kArgExcCap); } else { return exc_capacity; } };
	EncodeLdsF32 (&lds)[kFusedWaves] = S.lds;
	uint64_t (&s_size)[kFusedWaves]  = S.s_size;
	uint64_t& s_excl                 = S.s_excl;
	uint32_t& s_count                = S.s_count;
	uint32_t& s_ready                = S.s_ready;
	const int lane = lane_id();
	const int wave = wave_in_wg();
#ifdef ALPGPU_F32_INIT_BARRIER_FIRST // (A/B: until late in round 4 the tile's two LDS words were set, and waited for, in front of the loads)
	if (MODE == kSinglePass) { // (the other two modes share nothing between wavefronts)
		if (threadIdx.x == 0) {
			s_count = 0;
			s_ready = 0;
			s_excl  = ~0ull; // "stalled" until the look-back says otherwise (k_encode_lean: a wavefront 0 that gave up on its state never runs it)
		}
		__syncthreads();
	}
#endif

	EncodeLdsF32&  L    = lds[wave];
	const uint64_t vl   = tile * kFusedWaves + wave;
	const bool     live = vl < n_vectors_launch;
	const uint64_t v    = v_first + vl;
	VecInF             x;
	alpgpu_vector_desc d;
	uint64_t           lacc[4] = {0, 0, 0, 0}; // ALP_RD: packed left streams of this lane's four lane64 columns
	// Where the vector's exceptions are.  Default: sixteen lane masks (scalar register pairs), live from the value steps to the exception record.
	// -DALPGPU_F32_BITS_ANALYSIS (late round 4, measured, NOT the default): per lane its own sixteen bits in one VGPR plus the count of every value
	// step — no lane mask outlives its value step (encode_f32_device.hpp: encode_alp_lean_f32): 89 -> 41 spilled scalar registers and 1017 -> 964
	// vector instructions per vector, but the kernel then needs 66 VGPRs, seven of them go through scratch, and it is 3-4 % SLOWER
	// (profiles/r04_float_encode.txt, late round 4, point f).
#ifdef ALPGPU_F32_BITS_ANALYSIS
	uint32_t           excbits  = 0;
	uint32_t           cnt_m[4] = {0u, 0u, 0u, 0u};
#define ALPGPU_F32_EXC_WITNESS excbits
#define ALPGPU_F32_FOR_EACH_EXCEPTION(...) for_each_exception_bits_f32(excbits, cnt_m, lane, __VA_ARGS__)
#else
	uint64_t           ballots[4][4];
#define ALPGPU_F32_EXC_WITNESS static_cast<int64_t>(ballots[0][0] ^ ballots[3][3] ^ ballots[1][1])
#define ALPGPU_F32_FOR_EACH_EXCEPTION(...) for_each_exception_f32(ballots, lane, __VA_ARGS__)
#endif
	uint32_t           pvals[4][4]; // what gets packed: value - base (ALP), right parts (ALP_RD)
#pragma unroll
	for (int m = 0; m < 4; ++m) { pvals[m][0] = pvals[m][1] = pvals[m][2] = pvals[m][3] = 0u; }
	int                cnt = 0;
	d.packed_off = d.exc_off = 0;
	d.base                   = 0;
	d.bw = d.e = d.f = d.lbw = 0;
	d.exc_cnt = d.scheme = 0;
	// The rowgroup's state once, into registers; async_states: published by the persistent search beside this kernel (see k_encode_fused).  Its
	// (first) read is issued in FRONT of the vector's loads and looked at behind them: one round trip at the head of the wavefront, not two
	// (until round 4 the vector was requested only after the state had arrived).
	const uint64_t               v_read   = live ? v : v_first;
	const alpgpu_rowgroup_state* rg_ptr   = rgs + v_read / kRowgroup;
	const bool                   polling  = MODE == kSinglePass && async_states;
	const uint32_t               st_word  = polling ? rowgroup_state_poll_begin(rg_ptr, lane) : reinterpret_cast<const uint32_t*>(rg_ptr)[lane & 7];
	x                                     = load_vector_f32(in, v_read, lane);
#ifndef ALPGPU_F32_INIT_BARRIER_FIRST
	if (MODE == kSinglePass) { // the tile's two LDS words, set behind the ISSUE of the loads (k_encode_lean): the barrier falls into the shadow of their round trip
		if (threadIdx.x == 0) {
			s_count = 0;
			s_ready = 0;
			s_excl  = ~0ull; // "stalled" until the look-back says otherwise (k_encode_lean: a wavefront 0 that gave up on its state never runs it)
		}
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // not __syncthreads(): its fence would wait for the loads in flight
	}
#endif
	bool                         state_ok = true;
	const alpgpu_rowgroup_state  st       = polling ? rowgroup_state_poll_finish(rg_ptr, st_word, lane, spin_limit >> 4, state_ok) : unpack_rowgroup_state(st_word);
	const alpgpu_rowgroup_state* rgp      = &st;
	if (!state_ok) { // wave-uniform: a stall, like a look-back that gives up
		if (lane == 0) { status_store(totals + 3, 1ull); }
		return;
	}
	// bytes used by earlier launches of this column: constant while this launch runs (k_fused_finish updates them), read now so
	// that nothing but the ordered offset stands between the wait and the stores
	const uint64_t base_p = totals[0], base_e = totals[1];
	ALPGPU_F32_STOP(1, __float_as_uint(x.x[0][0] + x.x[3][3]) + st.k);
	if (live) {
		d.scheme = rgp->scheme;
		if (rgp->scheme == ALPGPU_SCHEME_ALP) {
			int e, f;
			if (rgp->k > 1) {
				second_level_select_f32(x, rgp, L, lane, e, f);
			} else {
				e = rgp->combos[0];
				f = rgp->combos[1];
			}
			ALPGPU_F32_STOP(2, e * 32 + f);
#ifdef ALPGPU_F32_BITS_ANALYSIS
			AlpEncodedLeanF R;
			encode_alp_lean_f32(x, e, f, lane, R);
			ALPGPU_F32_STOP(3, R.base + R.bw + R.cnt + R.excbits + R.enc[0][0] + R.enc[3][3] + R.enc[1][1] + R.enc[2][2]);
			excbits = R.excbits;
#else
			AlpEncodedF R;
			encode_alp_registers_f32(x, e, f, lane, R);
			ALPGPU_F32_STOP(3, R.base + R.bw + R.cnt + static_cast<int64_t>(R.ballot[0][0] ^ R.ballot[3][3] ^ R.ballot[1][2] ^ R.ballot[2][1]) + R.enc[0][0] + R.enc[3][3] + R.enc[1][1] + R.enc[2][2]);
#endif
			d.base = R.base, d.bw = static_cast<uint8_t>(R.bw), d.e = static_cast<uint8_t>(e), d.f = static_cast<uint8_t>(f);
			cnt = R.cnt;
			const uint32_t base = static_cast<uint32_t>(R.base);
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				u32x4 q;
#pragma unroll
				for (int j = 0; j < 4; ++j) { q[j] = static_cast<uint32_t>(R.enc[m][j]) - base; }
#ifdef ALPGPU_F32_BITS_ANALYSIS
				cnt_m[m] = R.cnt_m[m];
#else
#pragma unroll
				for (int j = 0; j < 4; ++j) { ballots[m][j] = R.ballot[m][j]; }
#endif
#pragma unroll
				for (int j = 0; j < 4; ++j) { pvals[m][j] = q[j]; }
			}
		} else {
			// rd.hpp:109-147: right = bits & mask, left = bits >> rbw; left -> dictionary index, not found = exception.
			// Left index streams: value i -> lane64 = i & 63 = 4*(lane & 15) + j, row = i >> 6 = 4*m + (lane >> 4).
			const int      rbw   = rgp->rd_rbw;
			const int      lbw   = rgp->rd_lbw;
			const int      ds    = rgp->rd_dict_size;
			const uint32_t rmask = bw_mask32(rbw);
			const uint64_t lmask = (1ull << lbw) - 1ull;
			d.bw = static_cast<uint8_t>(rbw), d.lbw = static_cast<uint8_t>(lbw);
			const RdOrderView order = load_rd_order(rd_order ? rd_order + (v / kRowgroup) * ALPGPU_RD_ORDER_STRIDE : nullptr, *rgp, lane, MODE == kSinglePass && async_states != 0);
			uint32_t dict[8]; // read once: left inside the loop, the compiler re-reads the dictionary from memory for every value
#pragma unroll
			for (int dd = 0; dd < 8; ++dd) { dict[dd] = rgp->rd_dict[dd]; }
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				u32x4     q;
				const int row = 4 * m + (lane >> 4);
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const uint32_t bits = __float_as_uint(x.x[m][j]);
					q[j]                = bits & rmask;
					const uint32_t left = (bits >> rbw) & 0xFFFFu;
					int            idx  = ds;
#pragma unroll
					for (int dd = 7; dd >= 0; --dd) {
						if (dd < ds && dict[dd] == left) { idx = dd; }
					}
					const bool     exc = idx == ds;
					const uint64_t bj  = ballot64(exc);
					if (order.valid && bj != 0) { // the reference's index for a left part outside the dictionary
						const int ridx = rd_exception_index(order, left);
						idx            = exc ? ridx : idx;
					}
#ifdef ALPGPU_F32_BITS_ANALYSIS
					cnt_m[m] += static_cast<uint32_t>(__builtin_popcountll(bj));
					excbits = push_exception_bit(excbits, bj);
#else
					ballots[m][j] = bj;
					cnt += __builtin_popcountll(bj);
#endif
					lacc[j] |= (static_cast<uint64_t>(idx) & lmask) << (row * lbw);
				}
#pragma unroll
				for (int j = 0; j < 4; ++j) { pvals[m][j] = q[j]; }
			}
#ifdef ALPGPU_F32_BITS_ANALYSIS
			cnt = static_cast<int>(cnt_m[0] + cnt_m[1] + cnt_m[2] + cnt_m[3]);
#endif
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				lacc[j] |= static_cast<uint64_t>(__shfl_xor(static_cast<long long>(lacc[j]), 16));
				lacc[j] |= static_cast<uint64_t>(__shfl_xor(static_cast<long long>(lacc[j]), 32));
			}
		}
		d.exc_cnt = static_cast<uint16_t>(cnt);
	}
	ALPGPU_F32_STOP(4, d.base + d.bw + cnt + pvals[0][0] + pvals[3][3] + pvals[1][2] + pvals[2][1] + ALPGPU_F32_EXC_WITNESS);
	uint64_t my_p = 0, my_e = 0;
	if (live) { record_sizes<4>(d, my_p, my_e); }
	if (MODE == kAnalyze) { // the sizes are all the scan needs
		if (live && lane == 0) { descs[v] = d; }
		return;
	}
	uint64_t reserved = ~0ull; // UNORDERED: the tile's exclusive prefix, in the lane that asked for it
	bool     reserver = false;
	if (MODE == kSinglePass && lane == 0) {
		s_size[wave]           = status_pack(0, my_p >> 7, my_e >> 3);
		const uint32_t arrived = __hip_atomic_fetch_add(&s_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (arrived == kFusedWaves - 1) {
			uint64_t aggregate = 0;
#pragma unroll
			for (int w = 0; w < kFusedWaves; ++w) { aggregate += s_size[w]; }
			if constexpr (UNORDERED) {
				reserved = __hip_atomic_fetch_add(status + lookback_words(gridDim.x), aggregate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				reserver = true;
			} else {
				status_store(status + tile, kFlagAggregate | aggregate); // the tile word is written exactly once
			}
		}
	}
	// Pack while the ordered offset is on its way (see k_encode_fused) — into the wavefront's 4 KiB image, which STAYS in LDS until the offset is
	// there (round 4; until then the image was read back into sixteen registers and the record took its place): nothing vector-sized but the
	// input itself is live across the wait.  The exception record is laid out behind the image's 128 * bw bytes when it fits what is left of the
	// 4 KiB, and leaves as contiguous stores after the wait; else (a wide vector with many exceptions) it is written from the registers then.
	// Pad bytes are zero.
	ALPGPU_F32_STOP(5, base_p + base_e + my_p + my_e + pvals[0][0] + pvals[3][3]);
	pack_u32_scatter_image(reinterpret_cast<uint32_t*>(L.vals), pvals, d.bw, lane);
	ALPGPU_F32_STOP(6, reinterpret_cast<uint32_t*>(L.vals)[lane] + base_p + ALPGPU_F32_EXC_WITNESS);
	const bool     alp_rec    = d.scheme == ALPGPU_SCHEME_ALP;
	const uint32_t val_bytes  = alp_rec ? 4u * static_cast<uint32_t>(cnt) : 2u * static_cast<uint32_t>(cnt);
	const uint32_t rec_off    = 128u * d.bw;
	const bool     rec_staged = rec_off + my_e <= sizeof(L.vals);
	uint8_t*       img        = reinterpret_cast<uint8_t*>(L.vals) + rec_off;
	if (cnt > 0 && rec_staged) {
		if (lane == 0) { reinterpret_cast<uint64_t*>(img)[(my_e >> 3) - 1] = 0ull; } // the pad lives in the last word
		wave_lds_sync();
		const int rbw = d.bw;
		ALPGPU_F32_FOR_EACH_EXCEPTION([&](int r, int m, int j) {
			const uint32_t bits = __float_as_uint(x.x[m][j]);
			if (alp_rec) {
				reinterpret_cast<uint32_t*>(img)[r] = bits;
			} else {
				reinterpret_cast<uint16_t*>(img)[r] = static_cast<uint16_t>(bits >> rbw);
			}
			reinterpret_cast<uint16_t*>(img + val_bytes)[r] = static_cast<uint16_t>(256 * m + 4 * lane + j);
		});
		wave_lds_sync();
	}
	ALPGPU_F32_STOP(7, reinterpret_cast<uint32_t*>(L.vals)[lane] + reinterpret_cast<uint32_t*>(L.vals)[960 + lane] + base_p);
	if (MODE == kSinglePass) {
		// wavefront 0 finds the tile's offset; the others park at a workgroup barrier meanwhile (see k_encode_lean: a worker that spins on an LDS
		// word takes issue slots from the wavefronts that still compute)
		if constexpr (UNORDERED) {
			if (reserver) { s_excl = reserved; }
		} else {
			if (wave == 0) { tile_lookback(tile, status, totals, s_size, &s_count, &s_excl, &s_ready, lane, spin_limit); }
		}
		__syncthreads();
		const uint64_t mine_sz = lane < wave ? s_size[lane & (kFusedWaves - 1)] : 0ull;
		const uint64_t local   = wave_sum_u64(mine_sz);
		const uint64_t excl    = s_excl;
		if (excl == ~0ull) { return; }
		const uint64_t pre = excl + local;
		d.packed_off       = base_p + ((pre >> 31) & 0x7FFFFFFFull) * 128ull;
		d.exc_off          = base_e + (pre & 0x7FFFFFFFull) * 8ull;
	} else if (live) { // kPack: the scan left the offsets inside the vector's scan tile in its descriptor
		const uint64_t st_ = v / kScanTileF32;
		d.packed_off       = descs[v].packed_off + status[2 * st_];
		d.exc_off          = descs[v].exc_off + status[2 * st_ + 1];
	}
	ALPGPU_F32_STOP(8, d.packed_off + d.exc_off + reinterpret_cast<uint32_t*>(L.vals)[lane]);
	if (!live) { return; }
	if (d.packed_off + my_p > arg_packed_cap() || d.exc_off + my_e > arg_exc_cap()) { // see k_encode_fused
		if (lane == 0) {
			__hip_atomic_store(totals + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			arg_descs()[v] = empty_descriptor();
		}
		return;
	}
	uint8_t* dst = arg_packed() + d.packed_off;
	uint8_t* rec = arg_excs() + d.exc_off;
	if (cnt > 0) {
		if (rec_staged) {
			const uint64_t* img64 = reinterpret_cast<const uint64_t*>(img);
			uint64_t*       rec64 = reinterpret_cast<uint64_t*>(rec);
			const int       n_w   = static_cast<int>(my_e >> 3);
			for (int w = lane; w < n_w; w += 64) { rec64[w] = img64[w]; }
		} else { // no room behind the image: values, positions and pad straight from the registers
			uint16_t* rpos = reinterpret_cast<uint16_t*>(rec + val_bytes);
			const int rbw  = d.bw;
			ALPGPU_F32_FOR_EACH_EXCEPTION([&](int r, int m, int j) {
				const uint32_t bits = __float_as_uint(x.x[m][j]);
				if (alp_rec) {
					reinterpret_cast<uint32_t*>(rec)[r] = bits;
				} else {
					reinterpret_cast<uint16_t*>(rec)[r] = static_cast<uint16_t>(bits >> rbw);
				}
				rpos[r] = static_cast<uint16_t>(256 * m + 4 * lane + j);
			});
			const int n_pos = static_cast<int>((my_e - val_bytes) >> 1);
			if (cnt + lane < n_pos) { rpos[cnt + lane] = 0; }
		}
	}
	store_image_f32(reinterpret_cast<const uint32_t*>(L.vals), d.bw, reinterpret_cast<u32x4*>(dst), lane);
	if (d.scheme != ALPGPU_SCHEME_ALP && lane < 16) {
		uint64_t* out64 = reinterpret_cast<uint64_t*>(dst + 128ull * d.bw);
		for (int k = 0; k < d.lbw; ++k) { // word k of lane64 columns 4*lane .. 4*lane+3
			uint64_t w = 0;
#pragma unroll
			for (int j = 0; j < 4; ++j) { w |= ((lacc[j] >> (16 * k)) & 0xFFFFull) << (16 * j); }
			out64[16 * k + lane] = w;
		}
	}
	if (lane == 0) { arg_descs()[v] = d; }
}

// kSinglePass: `status` = look-back words, one workgroup per tile.  kAnalyze / kPack: `status` = the scan's tile bases ([tile][2], kPack only),
// `gate` as in encode_kernels.hip, v_first = 0; the grid may be smaller than the number of tiles (the workgroups stride over them): behind a
// single-pass encode these two are launched with a closed gate every time, and a grid of one workgroup per tile cost 16 us per 256 Ki
// vectors just to find the gate closed (profiles/r03_float_encode.txt).
// (the single pass is held to 96 VGPRs — __launch_bounds__' second argument, wavefronts per SIMD: four of them per SIMD then leave the 96
// registers the persistent rowgroup search needs to share the CU)
// (the first nine parameters are read by offset in the single pass — alp_device.hpp: kArgDescs .. kArgExcCap — keep their order and types)
template <int MODE, bool UNORDERED = false>
__global__ __launch_bounds__(64 * kFusedWaves, MODE == kSinglePass ? ALPGPU_F32_ENC_OCC : 1) void k_encode_fused_f32(const float* __restrict__ in, const alpgpu_rowgroup_state* __restrict__ rgs,
                                                                       alpgpu_vector_desc* __restrict__ descs, uint8_t* __restrict__ packed,
                                                                       uint8_t* __restrict__ excs, uint64_t* __restrict__ status,
                                                                       uint64_t* __restrict__ totals, uint64_t packed_capacity, uint64_t exc_capacity,
                                                                       uint64_t v_first, uint64_t n_vectors_launch, const uint16_t* __restrict__ rd_order,
                                                                       uint32_t spin_limit, const uint64_t* __restrict__ gate, uint32_t async_states) {
	if (MODE != kSinglePass && gate != nullptr && *gate == 0) { return; }
	__builtin_amdgcn_s_setprio(2); // over the persistent rowgroup search that may share the CU (see k_encode_fused)
	__shared__ FusedSharedF32 S;
	if constexpr (MODE == kSinglePass) {
		encode_tile_f32<MODE, UNORDERED>(S, blockIdx.x, in, rgs, descs, packed, excs, status, totals, packed_capacity, exc_capacity, v_first, n_vectors_launch, rd_order, spin_limit,
		                                 async_states);
	} else {
		const uint64_t n_tiles = (n_vectors_launch + kFusedWaves - 1) / kFusedWaves;
		for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
			encode_tile_f32<MODE>(S, tile, in, rgs, descs, packed, excs, status, totals, packed_capacity, exc_capacity, v_first, n_vectors_launch, rd_order, spin_limit,
			                      async_states);
		}
	}
}

// (see k_fused_finish in encode_kernels.hip)
__global__ __launch_bounds__(256) void k_fused_finish_f32(uint64_t* __restrict__ totals, alpgpu_rowgroup_state* __restrict__ clear_rgs, uint64_t n_clear,
                                                          const uint64_t* __restrict__ reserve) {
	if (threadIdx.x == 0) {
		if (reserve != nullptr) { // an UNORDERED launch: its bytes are what its tiles reserved
			const uint64_t incl = *reserve;
			totals[4]           = totals[0] + ((incl >> 31) & 0x7FFFFFFFull) * 128ull;
			totals[5]           = totals[1] + (incl & 0x7FFFFFFFull) * 8ull;
		}
		totals[0] = totals[4];
		totals[1] = totals[5];
		if (totals[3] != 0) { totals[6] = 1; } // the gate of the recovery kernels
	}
	if (clear_rgs != nullptr) {
		for (uint64_t i = threadIdx.x; i < n_clear; i += 256) { clear_rgs[i].pad = 0; }
	}
}

int launch_encode_fused_range_f32(hipStream_t stream, const float* d_in, const alpgpu_column* col, uint64_t* d_workspace, uint64_t v_first, uint64_t n_range,
                                  bool force_stall, bool async_states, hipEvent_t async_join, hipEvent_t async_head, bool unordered) {
	for (uint64_t first = v_first; first < v_first + n_range; first += kFusedMaxVectors) {
		const uint64_t left     = v_first + n_range - first;
		const uint64_t n_launch = left < kFusedMaxVectors ? left : kFusedMaxVectors;
		const uint64_t n_tiles  = (n_launch + kFusedWaves - 1) / kFusedWaves;
		if (hipMemsetAsync(d_workspace, 0, (lookback_words(n_tiles) + 1) * 8, stream) != hipSuccess) { return ALPGPU_ERR_HIP; } // (+ the unordered form's counter word)
		if (async_states && first == v_first && async_head != nullptr) {
			if (hipStreamWaitEvent(stream, async_head, 0) != hipSuccess) { return ALPGPU_ERR_HIP; }
		}
		const uint64_t* reserve = unordered ? d_workspace + lookback_words(n_tiles) : nullptr;
		if (unordered) {
			hipLaunchKernelGGL((k_encode_fused_f32<kSinglePass, true>), dim3(static_cast<unsigned>(n_tiles)), dim3(64 * kFusedWaves), 0, stream, d_in, col->d_rowgroups,
			                   col->d_vectors, col->d_packed, col->d_exc, d_workspace, col->d_totals, col->packed_capacity, col->exc_capacity, first,
			                   n_launch, col->d_rd_order, force_stall ? 0u : kSpinLimit, static_cast<const uint64_t*>(nullptr), async_states ? 1u : 0u);
		} else {
			hipLaunchKernelGGL(k_encode_fused_f32<kSinglePass>, dim3(static_cast<unsigned>(n_tiles)), dim3(64 * kFusedWaves), 0, stream, d_in, col->d_rowgroups,
			                   col->d_vectors, col->d_packed, col->d_exc, d_workspace, col->d_totals, col->packed_capacity, col->exc_capacity, first,
			                   n_launch, col->d_rd_order, force_stall ? 0u : kSpinLimit, static_cast<const uint64_t*>(nullptr), async_states ? 1u : 0u);
		}
		if (async_states && first + n_launch >= v_first + n_range) {
			if (hipStreamWaitEvent(stream, async_join, 0) != hipSuccess) { return ALPGPU_ERR_HIP; }
			hipLaunchKernelGGL(k_fused_finish_f32, dim3(1), dim3(256), 0, stream, col->d_totals, col->d_rowgroups, col->n_rowgroups, reserve);
		} else {
			hipLaunchKernelGGL(k_fused_finish_f32, dim3(1), dim3(256), 0, stream, col->d_totals, static_cast<alpgpu_rowgroup_state*>(nullptr), 0ull, reserve);
		}
	}
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

int launch_encode_fused_f32(hipStream_t stream, const float* d_in, uint64_t n_vectors, const alpgpu_column* col, uint64_t* d_workspace, bool force_stall,
                            bool async_states, hipEvent_t async_join, hipEvent_t async_head, bool unordered) {
	if (hipMemsetAsync(col->d_totals, 0, 64, stream) != hipSuccess) { return ALPGPU_ERR_HIP; }
	return launch_encode_fused_range_f32(stream, d_in, col, d_workspace, 0, n_vectors, force_stall, async_states, async_join, async_head, unordered);
}

// the two-pass form for float columns (gate: see launch_encode_vectors in encode_kernels.hip)
int launch_encode_vectors_f32(hipStream_t stream, const float* d_in, uint64_t n_vectors, const alpgpu_column* col, uint64_t* d_workspace, const uint64_t* gate) {
	if (n_vectors == 0) {
		if (gate == nullptr) { (void)hipMemsetAsync(col->d_totals, 0, 64, stream); }
		return ALPGPU_OK;
	}
	// gate != nullptr: the recovery route behind a single-pass encode — a capped grid (the workgroups stride over the tiles), so that the
	// launches cost microseconds when the gate is closed, which is always unless the look-back stalled
	const uint64_t n_tiles = (n_vectors + kFusedWaves - 1) / kFusedWaves;
	const uint64_t cap     = gate != nullptr ? 4096ull : (1ull << 30);
	const dim3     grid(static_cast<unsigned>(n_tiles < cap ? n_tiles : cap)), block(64 * kFusedWaves);
	hipLaunchKernelGGL(k_encode_fused_f32<kAnalyze>, grid, block, 0, stream, d_in, col->d_rowgroups, col->d_vectors, col->d_packed, col->d_exc, d_workspace,
	                   col->d_totals, col->packed_capacity, col->exc_capacity, 0ull, n_vectors, col->d_rd_order, 0u, gate, 0u);
	if (launch_scan_offsets(stream, col, n_vectors, d_workspace, true, gate) != ALPGPU_OK) { return ALPGPU_ERR_HIP; }
	hipLaunchKernelGGL(k_encode_fused_f32<kPack>, grid, block, 0, stream, d_in, col->d_rowgroups, col->d_vectors, col->d_packed, col->d_exc, d_workspace,
	                   col->d_totals, col->packed_capacity, col->exc_capacity, 0ull, n_vectors, col->d_rd_order, 0u, gate, 0u);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

} // namespace alpgpu
