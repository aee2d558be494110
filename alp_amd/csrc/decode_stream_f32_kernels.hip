// decode_stream_f32_kernels.hip — the float store decode as PERSISTENT workgroups that stream the column (round 6).
//
// Replaces, per vector, what decode_f32_kernels.hip replaces (falp<float> + patch_exceptions, include/alp/falp.hpp:28-44, include/alp/decoder.hpp:141-149;
// ALP_RD: include/alp/rd.hpp:152-178) — same bytes, another schedule.
//
// Why (profiles/r06_float_decode.txt): with one small workgroup per two vectors, a 3-bit float column decodes at 0.49 of the HBM peak cold and 0.74 with its records
// already in the Infinity Cache; the same launch WITHOUT its packed-word loads runs at 0.84.  The 384 bytes a vector reads cost as much as 3 KiB of what it writes:
// every workgroup's life is descriptor round trip -> record round trip -> stores, one after the other, under a memory system saturated with writes, and every
// wavefront works out the same per-vector scalars again (200 scalar + 142 vector instructions per wavefront for two vectors).  Here
//   * ONE workgroup per CU owns chunks c, c + G, c + 2 G ... of C consecutive vectors.  Its D LOADING wavefronts issue every load — a chunk's descriptors a turn
//     ahead, then its packed words and exception records as two flat LDS-DMA copies into the chunk's arena (a column written in vector order: the chunk's records are
//     one span of each stream; otherwise vector by vector) — D chunks ahead of the decode, and are the only wavefronts that wait for memory;
//   * its NDEC DECODING wavefronts own whole vectors (lane L: the quads L, 64 + L, 128 + L, 192 + L) and wait for nothing but the chunk barrier: on gfx9 loads and
//     stores share one counter, and a wavefront that both prefetches and stores drains its stores — a round trip through a memory system saturated with writes —
//     once per chunk (the first form of this kernel did: 4 us per chunk of 8 vectors and workgroup);
//   * what a vector's decode needs besides its words — widths, base, 10^f, 10^-e, the conversion shortcut's verdict, slot offsets — is worked out ONCE per chunk,
//     one vector per LANE (not once per wavefront and vector in scalar code), and left in LDS as a 48-byte plan the unpack reads back;
//   * exceptions are found through a table per vector — quad t -> (index of its first exception, 4 hit bits) — written by the decoding wavefront from the (sorted)
//     positions and cleaned by the lanes that read it: no mask, no scan, no ds_bpermute, no zeroing pass.
// A vector whose record does not fit what is left of the arena is decoded from HBM directly (bounded buffer loads; positions and values from the stream).
// Shapes (template parameters; ALPGPU_OPT_DECODE_VECTORS_PER_WG 16-30): the rule uses <12, 24576, 2, 12> (decode_policy.hpp: policy_stream_f32).
#include "alp_device_f32.hpp"
#include "decode_f32_device.hpp"
#include "launch.hpp"
#include <cstdlib>

namespace alpgpu {


struct __attribute__((aligned(16))) StreamPlan {
	uint32_t src_off;  // arena byte offset of the packed words (right words, then ALP_RD left words, then the exception record)
	uint32_t exc_lds;  // arena byte offset of the exception record (values, then positions, as in the stream)
	uint32_t flags;    // bw | lbw << 8 | kPlan* bits
	uint32_t exc_cnt;
	uint32_t base, fact, frac_bits, pad;
	uint64_t packed_off, exc_off; // the descriptor's stream offsets (for vectors decoded from HBM directly)
};
constexpr uint32_t kPlanAlp = 1u << 16, kPlanShortcut = 1u << 17, kPlanDirect = 1u << 18;

template <int C, int ARENA, int D>
struct __attribute__((aligned(16))) StreamLds {
	uint8_t            arena[D + 1][ARENA + 128]; // chunk k lives in arena k mod (D + 1)  (+ the unit row past the last slot that the unpack reads and masks off)
	alpgpu_vector_desc desc[D][C];                // loading wavefront j: the descriptors of its next chunk
	StreamPlan         plan[D + 1][C];
	uint32_t           dict[D + 1][C][4]; // ALP_RD: the rowgroup's dictionary (8 x u16), by LDS-DMA
	uint16_t           table[2][C][256];  // chunk k: table k & 1.  Per quad of a vector: first exception index << 4 | hit bits (0: none)
	uint32_t           fact[12], frac[12], bound[12];
};

__device__ __constant__ const uint32_t kStreamShortcutBound[11] = {16777216u, 16777216u, 16777216u, 2147483u, 214748u, 21474u, 2147u, 214u, 21u, 2u, 0u}; // decode_f32_kernels.hip: kShortcutBoundF

__device__ __forceinline__ uint32_t row_exclusive_scan16(uint32_t v) { // over the 16 lanes of a DPP row
	uint32_t s = v;
	s += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(s), 0x111, 0xf, 0xf, false));
	s += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(s), 0x112, 0xf, 0xf, false));
	s += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(s), 0x114, 0xf, 0xf, false));
	s += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(s), 0x118, 0xf, 0xf, false));
	return s - v;
}
__device__ __forceinline__ uint32_t lane_value(uint32_t v, int l) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), l)); }

// the descriptors of chunk `c` -> S.desc[db] (8 dwords per vector, one LDS-DMA dword per lane; the loading wavefront)
template <int C, class LDS>
__device__ __forceinline__ void stream_load_descs(LDS& S, int db, const alpgpu_vector_desc* __restrict__ descs, uint64_t v_base, int n_here, int lane) {
#pragma unroll
	for (int u0 = 0; u0 < 8 * C; u0 += 64) {
		if (u0 + lane < 8 * n_here) {
			__builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(descs + v_base) + u0 + lane, reinterpret_cast<uint32_t*>(&S.desc[db][0]) + u0, 4, 0, 0);
		}
	}
}

// stage 2: the chunk's plan (one vector per lane) and the loads of its records into arena `buf`
template <int C, int ARENA, class LDS>
__device__ __forceinline__ void stream_issue_chunk(LDS& S, int db, int buf, const alpgpu_rowgroup_state* __restrict__ rgs, const uint8_t* __restrict__ packed,
                                                   const uint8_t* __restrict__ excs, uint64_t exc_limit, uint64_t v_base, int n_here, int lane) {
	typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
	const bool               valid  = lane < n_here;
	const alpgpu_vector_desc d      = S.desc[db][lane < C ? lane : 0];
	const bool               is_alp = d.scheme == ALPGPU_SCHEME_ALP;
	const uint32_t           words  = valid ? static_cast<uint32_t>(d.bw) + (is_alp ? 0u : static_cast<uint32_t>(d.lbw)) : 0u;
	const uint32_t           cnt    = valid ? static_cast<uint32_t>(d.exc_cnt) : 0u;
	const uint32_t           rec    = (cnt * (is_alp ? 6u : 4u) + 7u) & ~7u; // (records are 8-byte multiples)
	// A column written in vector order (every encode but the unordered one): the chunk's packed words are ONE span of the packed stream and its exception records one
	// span of the exception stream — two flat copies (a full KiB per LDS-DMA instruction) instead of a scalar loop per vector, which is what made the loading
	// wavefront the workgroup's pace (~750 instructions per chunk of 8, issued by one wavefront: call 12).  Layout then: packed words back to back, the records behind.
	const uint32_t pk      = 128u * words;
	const uint32_t pk_off  = row_exclusive_scan16(pk);
	const uint32_t rec_off = row_exclusive_scan16(rec);
	const uint64_t p0 = (static_cast<uint64_t>(lane_value(static_cast<uint32_t>(d.packed_off >> 32), 0)) << 32) | lane_value(static_cast<uint32_t>(d.packed_off), 0);
	const uint64_t e0 = (static_cast<uint64_t>(lane_value(static_cast<uint32_t>(d.exc_off >> 32), 0)) << 32) | lane_value(static_cast<uint32_t>(d.exc_off), 0);
	const int      last     = n_here - 1;
	const uint32_t pk_total = lane_value(pk_off + pk, last), rec_total = lane_value(rec_off + rec, last);
	// (the exception span is copied in 16-byte units from the 16-byte boundary below its first byte: the records then lie (e0 & 15) bytes into their part of the arena)
	const uint32_t rec_skew = static_cast<uint32_t>(e0) & 15u;
	const uint32_t rec_base = ((pk_total + 15u) & ~15u) + rec_skew;
	const bool     in_order = !valid || ((pk == 0u || d.packed_off == p0 + pk_off) && (rec == 0u || d.exc_off == e0 + rec_off));
	const bool     flat     = __builtin_amdgcn_ballot_w64(!in_order) == 0ull && rec_base + rec_total + 16u <= static_cast<uint32_t>(ARENA); // wave-uniform
	const uint32_t slot   = pk + ((rec + 15u) & ~15u);
	const uint32_t off_v  = row_exclusive_scan16(slot);
	const bool     direct = !flat && off_v + slot > static_cast<uint32_t>(ARENA);
	const uint32_t off    = flat ? pk_off : off_v;
	const uint32_t exc_at = flat ? rec_base + rec_off : off_v + pk;
	const uint32_t dirbit = direct ? 1u : 0u;
	const uint32_t plo = static_cast<uint32_t>(d.packed_off), phi = static_cast<uint32_t>(d.packed_off >> 32);
	const uint32_t elo = static_cast<uint32_t>(d.exc_off), ehi = static_cast<uint32_t>(d.exc_off >> 32);
	{
		if (lane < C) {
			const uint32_t f = d.f <= 10 ? d.f : 10, e = d.e <= 10 ? d.e : 10;
			const uint32_t base = static_cast<uint32_t>(d.base);
			// the conversion shortcut's verdict (decode_f32_kernels.hip: finish_quad_f32): every base + digit within [-2^24, 2^24] and, times 10^f, inside int32
			const int64_t lo64 = static_cast<int64_t>(static_cast<int32_t>(base)), hi64 = lo64 + static_cast<int64_t>(bw_mask32(d.bw));
			const int64_t bnd  = static_cast<int64_t>(S.bound[f]);
			const bool    shortcut = d.bw <= 24 && lo64 >= -bnd && hi64 <= bnd;
			StreamPlan    p;
			p.src_off    = off;
			p.exc_lds    = exc_at;
			p.flags      = static_cast<uint32_t>(d.bw) | (static_cast<uint32_t>(d.lbw) << 8) | (is_alp ? kPlanAlp : 0u) | (shortcut ? kPlanShortcut : 0u) | (direct ? kPlanDirect : 0u);
			p.exc_cnt    = cnt;
			p.base       = base;
			p.fact       = S.fact[f];
			p.frac_bits  = S.frac[e];
			p.pad        = 0;
			p.packed_off = d.packed_off;
			p.exc_off    = d.exc_off;
			S.plan[buf][lane] = p;
		}
		// ALP_RD vectors: their rowgroup's dictionary, dword (lane & 3) of vector lane >> 2
		static_assert(4 * C <= 64, "one wavefront covers a chunk's dictionaries");
		const int      vi = lane >> 2;
		const uint32_t rd = static_cast<uint32_t>(__shfl(static_cast<int>(valid && !is_alp ? 1u : 0u), vi));
		if (lane < 4 * C && rd != 0u) {
			const uint64_t rg = (v_base + static_cast<uint64_t>(vi)) / kRowgroup;
			__builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(rgs + rg) + 4 + (lane & 3), &S.dict[buf][0][0], 4, 0, 0);
		}
	}
#if defined(ALPGPU_STREAM_DISSECT) && ALPGPU_STREAM_DISSECT == 1 // measurement build: no record is loaded (the unpack reads what lies in the arena)
	return;
#endif
	if (flat) {
		const ull2* g = reinterpret_cast<const ull2*>(packed + p0);
		const int   n_units = static_cast<int>(pk_total >> 4);
		for (int u0 = 0; u0 < n_units; u0 += 64) {
			if (u0 + lane < n_units) { __builtin_amdgcn_global_load_lds(g + u0 + lane, reinterpret_cast<ull2*>(S.arena[buf]) + u0, 16, 0, 0); }
		}
		if (rec_total != 0u) {
			const uint64_t e_lo = e0 - rec_skew;
			const int      n_eu = static_cast<int>((rec_skew + rec_total + 15u) >> 4);
			if (e_lo + 16ull * static_cast<uint64_t>(n_eu) <= exc_limit) {
				const ull2* r = reinterpret_cast<const ull2*>(excs + e_lo);
				for (int u0 = 0; u0 < n_eu; u0 += 64) {
					if (u0 + lane < n_eu) { __builtin_amdgcn_global_load_lds(r + u0 + lane, reinterpret_cast<ull2*>(S.arena[buf] + rec_base - rec_skew) + u0, 16, 0, 0); }
				}
			} else { // (the stream's last bytes: never past what the caller allocated)
				const uint32_t* r    = reinterpret_cast<const uint32_t*>(excs + e0);
				const int       n_dw = static_cast<int>(rec_total >> 2);
				for (int u0 = 0; u0 < n_dw; u0 += 64) {
					if (u0 + lane < n_dw) { __builtin_amdgcn_global_load_lds(r + u0 + lane, reinterpret_cast<uint32_t*>(S.arena[buf] + rec_base) + u0, 4, 0, 0); }
				}
			}
		}
		return;
	}
#pragma unroll
	for (int i = 0; i < C; ++i) {
		const uint32_t w_i = lane_value(words, i), c_i = lane_value(cnt, i);
		if (lane_value(dirbit, i) != 0u || (w_i | c_i) == 0u) { continue; }
		const uint32_t off_i = lane_value(off, i);
		if (w_i != 0u) {
			const uint64_t po      = (static_cast<uint64_t>(lane_value(phi, i)) << 32) | lane_value(plo, i);
			const ull2*    g       = reinterpret_cast<const ull2*>(packed + po);
			const int      n_units = 8 * static_cast<int>(w_i);
			for (int u0 = 0; u0 < n_units; u0 += 64) {
				if (u0 + lane < n_units) { __builtin_amdgcn_global_load_lds(g + u0 + lane, reinterpret_cast<ull2*>(S.arena[buf] + off_i) + u0, 16, 0, 0); }
			}
		}
		if (c_i != 0u) {
			const uint64_t  eo   = (static_cast<uint64_t>(lane_value(ehi, i)) << 32) | lane_value(elo, i);
			const uint32_t* r    = reinterpret_cast<const uint32_t*>(excs + eo);
			const int       n_dw = static_cast<int>(lane_value(rec, i) >> 2);
			for (int u0 = 0; u0 < n_dw; u0 += 64) {
				if (u0 + lane < n_dw) { __builtin_amdgcn_global_load_lds(r + u0 + lane, reinterpret_cast<uint32_t*>(S.arena[buf] + off_i + 128u * w_i) + u0, 4, 0, 0); }
			}
		}
	}
}

// stage 3: the exception table of one vector from its (sorted) positions
template <class POS>
__device__ __forceinline__ void stream_build_table(uint16_t* __restrict__ table, POS pos, int cnt, int lane) {
	// (positions are taken modulo 1024: a malformed record cannot reach beyond the vector's table — the kernels follow a column's records as they find them)
	for (int j = lane; j < cnt; j += 64) {
		const uint32_t p     = pos[j] & 1023u;
		const uint32_t q     = p >> 2;
		const uint32_t prev  = j > 0 ? (pos[j - 1] & 1023u) : 0xFFFFu;
		if ((prev >> 2) != q) { // the first exception of its quad: it speaks for the (at most three) that follow
			uint32_t hits = 1u << (p & 3u);
#pragma unroll
			for (int t = 1; t < 4; ++t) {
				const uint32_t pn = pos[j + t < cnt ? j + t : cnt - 1] & 1023u;
				if ((pn >> 2) == q) { hits |= 1u << (pn & 3u); }
			}
			table[q] = static_cast<uint16_t>((static_cast<uint32_t>(j) << 4) | hits);
		}
	}
}

// stage 4: one whole vector by one wavefront — lane L owns the quads L, 64 + L, 128 + L, 192 + L (quad t = values 4 t .. 4 t + 3).  DIRECT: words and exception
// values from HBM, else from the arena
template <bool DIRECT, class LDS>
__device__ __forceinline__ void stream_decode_vector(LDS& S, int buf, int tb, int i, const uint8_t* __restrict__ packed, const uint8_t* __restrict__ excs, float* __restrict__ dst, int lane) {
	const StreamPlan P      = S.plan[buf][i];
	const uint32_t   flags  = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(P.flags)));
	const int        bw     = static_cast<int>(flags & 0xFFu);
	const int        lbw    = static_cast<int>((flags >> 8) & 0xFFu);
	const bool       is_alp = (flags & kPlanAlp) != 0u;
	const uint32_t   cnt    = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(P.exc_cnt)));
	alpgpu_vector_desc d; // (request_quad_f32 reads bw, lbw, scheme)
	d.bw     = static_cast<uint8_t>(bw);
	d.lbw    = static_cast<uint8_t>(lbw);
	d.scheme = is_alp ? ALPGPU_SCHEME_ALP : ALPGPU_SCHEME_ALP_RD;
	QuadWords      w[4];
	const uint8_t* rec_lds = nullptr;
	const uint8_t* rec_hbm = nullptr;
	if constexpr (DIRECT) {
		uint8_t*      first      = const_cast<uint8_t*>(packed + P.packed_off);
		constexpr int kRsrcFlags = 0x00020000;
		const BufferWordsF words {__builtin_amdgcn_make_buffer_rsrc(first, 0, 128 * bw, kRsrcFlags), __builtin_amdgcn_make_buffer_rsrc(first + 128u * bw, 0, is_alp ? 0 : 128 * lbw, kRsrcFlags)};
#pragma unroll
		for (int q = 0; q < 4; ++q) { w[q] = request_quad_f32(words, d, 64 * q + lane); }
		rec_hbm = excs + P.exc_off;
	} else {
		const uint32_t     src = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(P.src_off)));
		const StagedWordsF words {S.arena[buf] + src};
#pragma unroll
		for (int q = 0; q < 4; ++q) { w[q] = request_quad_f32(words, d, 64 * q + lane); }
		rec_lds = S.arena[buf] + static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(P.exc_lds)));
	}
	uint32_t entry[4] = {0u, 0u, 0u, 0u};
	if (cnt != 0u) { // wave-uniform
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			entry[q] = S.table[tb][i][64 * q + lane];
			if (entry[q] != 0u) { S.table[tb][i][64 * q + lane] = 0; } // the reader cleans up
		}
	}
	const uint32_t msk = bw_mask32(bw);
#pragma unroll
	for (int qd = 0; qd < 4; ++qd) {
		const int      tid  = 64 * qd + lane;
		const uint32_t hits = entry[qd] & 0xFu;
		uint32_t       rank = entry[qd] >> 4;
		const uint32_t s    = static_cast<uint32_t>((tid >> 3) * bw) & 31u;
		u32x4          q, out;
#pragma unroll
		for (int c = 0; c < 4; ++c) { q[c] = __builtin_amdgcn_alignbit(w[qd].w1[c], w[qd].w0[c], s) & msk; }
		if (is_alp) {
			const uint32_t base = P.base;
			const uint32_t fact = P.fact;
			const float    frac = __uint_as_float(P.frac_bits);
			if ((flags & kPlanShortcut) != 0u) {
				const float fact_f = static_cast<float>(fact);
#pragma unroll
				for (int c = 0; c < 4; ++c) { out[c] = __float_as_uint((static_cast<float>(static_cast<int32_t>(q[c] + base)) * fact_f) * frac); }
			} else {
#pragma unroll
				for (int c = 0; c < 4; ++c) { out[c] = __float_as_uint(decode_value_f32(static_cast<int32_t>(q[c] + base), fact, frac)); }
			}
			if (hits != 0u) {
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					if (hits & (1u << c)) {
						if constexpr (DIRECT) {
							out[c] = reinterpret_cast<const uint32_t*>(rec_hbm)[rank];
						} else {
							out[c] = reinterpret_cast<const uint32_t*>(rec_lds)[rank];
						}
						++rank;
					}
				}
			}
		} else {
			// ALP_RD (decode_f32_kernels.hip: finish_quad_f32): right parts = the u32 lanes, left parts = u16 lanes looked up in the dictionary
			const uint32_t lmsk = (1u << lbw) - 1u;
			const int      ls   = ((tid >> 4) * lbw) & 15;
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const uint32_t f0  = static_cast<uint32_t>(w[qd].l0 >> (16 * c)) & 0xFFFFu;
				const uint32_t f1  = static_cast<uint32_t>(w[qd].l1 >> (16 * c)) & 0xFFFFu;
				const uint32_t idx = ((f0 >> ls) | (f1 << (16 - ls))) & lmsk;
				uint32_t       l   = reinterpret_cast<const uint16_t*>(&S.dict[buf][i][0])[idx & 7u]; // (the dictionary looked up in LDS: decode_f32_kernels.hip)
				if (hits & (1u << c)) {
					if constexpr (DIRECT) {
						l = reinterpret_cast<const uint16_t*>(rec_hbm)[rank];
					} else {
						l = reinterpret_cast<const uint16_t*>(rec_lds)[rank];
					}
					++rank;
				}
				out[c] = (l << bw) | q[c];
			}
		}
#if defined(ALPGPU_STREAM_DISSECT) && ALPGPU_STREAM_DISSECT == 2 // measurement build: (almost) nothing is stored
		if ((out[0] ^ out[1] ^ out[2] ^ out[3]) == 0x13579BDFu) { __builtin_nontemporal_store(out, reinterpret_cast<u32x4*>(dst + 4 * tid)); }
#else
		__builtin_nontemporal_store(out, reinterpret_cast<u32x4*>(dst + 4 * tid));
#endif
	}
}

// the exception table of vector i of a landed chunk, by the wavefront that is about to decode it (the loading wavefronts built a chunk's tables one vector after the
// other in the first form: with 16 vectors per chunk that made them the workgroup's pace on every column with exceptions — call 18)
template <class LDS>
__device__ __forceinline__ void stream_build_vector_table(LDS& S, int buf, int tb, int i, const uint8_t* __restrict__ excs, int lane) {
	const StreamPlan& P   = S.plan[buf][i];
	const int         cnt = __builtin_amdgcn_readfirstlane(static_cast<int>(P.exc_cnt));
	if (cnt == 0) { return; }
	const uint32_t flags = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(P.flags)));
	const uint32_t vb    = (flags & kPlanAlp) ? 4u : 2u;
	if (flags & kPlanDirect) {
		stream_build_table(&S.table[tb][i][0], reinterpret_cast<const uint16_t*>(excs + P.exc_off + vb * cnt), cnt, lane);
	} else {
		const uint32_t at = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(P.exc_lds)));
		stream_build_table(&S.table[tb][i][0], reinterpret_cast<const uint16_t*>(S.arena[buf] + at + vb * cnt), cnt, lane);
	}
	wave_lds_sync();
}

// Wavefronts 0 .. 3 unpack and store; wavefront 4 + j loads the chunks k = j (mod D) of the workgroup: D chunks' records are in flight while one is decoded.
// Iteration k:   [loader k mod D: its chunk k has landed; tables of chunk k built]  barrier k  [decoders: chunk k]
//                                                                                              [loader k mod D: records of chunk k + D into the arena chunk k - 1 left, descriptors of chunk k + 2 D]
//                                                                                              [loader (k + 1) mod D: waits for chunk k + 1, builds its tables (the other table set)]
// NDEC wavefronts unpack and store (one whole vector at a time each), D load
template <int C, int ARENA, int D, int NDEC>
__global__ __launch_bounds__(64 * (NDEC + D)) void k_decode_stream_f32(const alpgpu_vector_desc* __restrict__ descs, const alpgpu_rowgroup_state* __restrict__ rgs,
                                                                                  const uint8_t* __restrict__ packed, const uint8_t* __restrict__ excs, float* __restrict__ out,
                                                                                  uint64_t n_vectors, uint64_t exc_limit, uint64_t* __restrict__ progress, uint64_t progress_tag) {
	static_assert(C <= 16, "a chunk's sizes are scanned inside one 16-lane row");
	constexpr int NBUF = D + 1;
	__shared__ StreamLds<C, ARENA, D> S;
	const int      tid  = static_cast<int>(threadIdx.x);
	const int      lane = tid & 63;
	const int      wave = wave_in_wg();
	const bool     loader   = wave >= NDEC;
	const int      j        = wave - NDEC; // (loaders)
	const uint64_t n_chunks = (n_vectors + C - 1) / C;
	const uint64_t G        = gridDim.x;
	const uint64_t c0       = blockIdx.x;
	if (c0 >= n_chunks) { return; }
	auto here  = [&](uint64_t chunk) { return static_cast<int>(n_vectors - chunk * C < static_cast<uint64_t>(C) ? n_vectors - chunk * C : static_cast<uint64_t>(C)); };
	auto chunk = [&](int k) { return c0 + static_cast<uint64_t>(k) * G; };
	// Two loops, one per kind of wavefront, with the same barriers: the compiler's wait-count pass follows the CODE, not the wavefronts — with the LDS-DMA loads
	// and the unpack in one loop it put s_waitcnt vmcnt(0) in front of the unpack's LDS reads ("an LDS-DMA may be in flight"), and on gfx9 that waits for the
	// wavefront's own STORES: every vector waited for the stores of the one before it (2 300 cycles per vector and wavefront; call 16).
	if (loader) {
		// every loader: the three small constant tables (the same values), its first chunk's descriptors and records
		if (lane < 11) {
			S.fact[lane]  = kFactArrF[lane];
			S.frac[lane]  = __float_as_uint(kFracArrF[lane]);
			S.bound[lane] = kStreamShortcutBound[lane];
		}
		if (chunk(j) < n_chunks) {
			stream_load_descs<C>(S, j, descs, chunk(j) * C, here(chunk(j)), lane);
			asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
			stream_issue_chunk<C, ARENA>(S, j, j % NBUF, rgs, packed, excs, exc_limit, chunk(j) * C, here(chunk(j)), lane);
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (the descriptors are read: their slot may be written again)
			if (chunk(j + D) < n_chunks) { stream_load_descs<C>(S, j, descs, chunk(j + D) * C, here(chunk(j + D)), lane); }
		}
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // (the decoders have cleaned the tables)
		if (j == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } // chunk 0 has landed
		for (int k = 0; chunk(k) < n_chunks; ++k) {
			asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // barrier k
			// the read-ahead's pace (read_ahead_kernels.hip): workgroup 0 says where the launch is — the workgroups move through the column side by side
			if (progress != nullptr && blockIdx.x == 0 && j == k % D && lane == 0) { __hip_atomic_store(progress, progress_tag | (chunk(k) * C), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
			if (j == k % D) { // chunk k - 1's arena is free: the records of chunk k + D into it, then the descriptors of chunk k + 2 D
				if (chunk(k + D) < n_chunks) {
					stream_issue_chunk<C, ARENA>(S, j, (k + D) % NBUF, rgs, packed, excs, exc_limit, chunk(k + D) * C, here(chunk(k + D)), lane);
					asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
					if (chunk(k + 2 * D) < n_chunks) { stream_load_descs<C>(S, j, descs, chunk(k + 2 * D) * C, here(chunk(k + 2 * D)), lane); }
				}
			}
			if (j == (k + 1) % D && chunk(k + 1) < n_chunks) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } // the next chunk is this wavefront's: it arrives at barrier k + 1 when the chunk has landed
		}
		return;
	}
	for (int t = tid; t < 2 * C * 128; t += 64 * NDEC) { reinterpret_cast<uint32_t*>(&S.table[0][0][0])[t] = 0u; }
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	for (int k = 0; chunk(k) < n_chunks; ++k) {
		const int buf = k % NBUF;
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // barrier k: chunk k's records, plan and tables are there
		const int n_here = here(chunk(k));
		float*    dst    = out + chunk(k) * C * kVec;
		for (int i = wave; i < n_here; i += NDEC) {
			stream_build_vector_table(S, buf, k & 1, i, excs, lane);
			const uint32_t flags = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(S.plan[buf][i].flags)));
			if (flags & kPlanDirect) {
				stream_decode_vector<true>(S, buf, k & 1, i, packed, excs, dst + i * kVec, lane);
			} else {
				stream_decode_vector<false>(S, buf, k & 1, i, packed, excs, dst + i * kVec, lane);
			}
		}
	}
}

template <int C, int ARENA, int D, int NDEC>
static int launch_stream(hipStream_t stream, const alpgpu_column* col, float* d_out, int n_cus, int wgs_per_cu, uint64_t* progress, uint64_t tag) {
	const uint64_t n        = col->n_vectors;
	const uint64_t n_chunks = (n + C - 1) / C;
	constexpr int  kThreads = 64 * (NDEC + D);
	// persistent workgroups: never more than are resident at once (a workgroup that waits for a slot starts its chunks when the others are half way through theirs)
	static const int resident = [] {
		int nb = 0;
		return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_decode_stream_f32<C, ARENA, D, NDEC>, kThreads, 0) == hipSuccess && nb > 0 ? nb : 1;
	}();
	if (wgs_per_cu > resident) { wgs_per_cu = resident; }
	const uint64_t cap      = static_cast<uint64_t>(n_cus) * static_cast<uint64_t>(wgs_per_cu);
	const unsigned grid     = static_cast<unsigned>(n_chunks < cap ? n_chunks : cap);
	hipLaunchKernelGGL((k_decode_stream_f32<C, ARENA, D, NDEC>), dim3(grid), dim3(kThreads), 0, stream, col->d_vectors, col->d_rowgroups, col->d_packed, col->d_exc, d_out, n, col->exc_capacity, progress, tag);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

// shape: 16 = chunks of 8 vectors over an 8 KiB arena; 17 = chunks of 16 over 16 KiB; 18 = chunks of 4 over 12 KiB (wide vectors)
int launch_decode_stream_f32(hipStream_t stream, const alpgpu_column* col, float* d_out, int shape, int n_cus, uint64_t* progress, uint64_t progress_tag) {
	if (col->n_vectors == 0) { return ALPGPU_OK; }
	static const int env_wgs = std::getenv("ALPGPU_STREAM_WGS_PER_CU") ? std::atoi(std::getenv("ALPGPU_STREAM_WGS_PER_CU")) : 0;
	if (n_cus <= 0) { n_cus = 256; }
	const int wgs = env_wgs > 0 ? env_wgs : 2;
	if (shape == 17) { return launch_stream<16, 16384, 2, 4>(stream, col, d_out, n_cus, wgs, progress, progress_tag); }
	if (shape == 18) { return launch_stream<4, 12288, 3, 4>(stream, col, d_out, n_cus, wgs, progress, progress_tag); }
	if (shape == 19) { return launch_stream<8, 8192, 1, 4>(stream, col, d_out, n_cus, wgs, progress, progress_tag); }
	if (shape == 20) { return launch_stream<8, 8192, 2, 4>(stream, col, d_out, n_cus, wgs, progress, progress_tag); }
	if (shape == 21) { return launch_stream<16, 16384, 2, 8>(stream, col, d_out, n_cus, env_wgs > 0 ? env_wgs : 1, progress, progress_tag); }
	if (shape == 22) { return launch_stream<8, 8192, 2, 8>(stream, col, d_out, n_cus, env_wgs > 0 ? env_wgs : 1, progress, progress_tag); }
	if (shape == 23) { return launch_stream<16, 16384, 2, 12>(stream, col, d_out, n_cus, env_wgs > 0 ? env_wgs : 1, progress, progress_tag); }
	if (shape == 24) { return launch_stream<12, 12288, 2, 12>(stream, col, d_out, n_cus, env_wgs > 0 ? env_wgs : 1, progress, progress_tag); }
	if (shape == 25) { return launch_stream<14, 14336, 2, 14>(stream, col, d_out, n_cus, env_wgs > 0 ? env_wgs : 1, progress, progress_tag); }
	if (shape == 27) { return launch_stream<12, 24576, 2, 12>(stream, col, d_out, n_cus, env_wgs > 0 ? env_wgs : 1, progress, progress_tag); }
	if (shape == 28) { return launch_stream<12, 49152, 1, 12>(stream, col, d_out, n_cus, env_wgs > 0 ? env_wgs : 1, progress, progress_tag); }
	if (shape == 29) { return launch_stream<12, 12288, 4, 12>(stream, col, d_out, n_cus, env_wgs > 0 ? env_wgs : 1, progress, progress_tag); }
	if (shape == 30) { return launch_stream<12, 12288, 1, 12>(stream, col, d_out, n_cus, env_wgs > 0 ? env_wgs : 1, progress, progress_tag); }
	if (shape == 26) { return launch_stream<16, 16384, 1, 14>(stream, col, d_out, n_cus, env_wgs > 0 ? env_wgs : 1, progress, progress_tag); }
	return launch_stream<8, 8192, 3, 4>(stream, col, d_out, n_cus, wgs, progress, progress_tag);
}

} // namespace alpgpu
