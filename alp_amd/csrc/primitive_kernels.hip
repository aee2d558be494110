// primitive_kernels.hip — the reference's per-vector primitives, n vectors at a time, fixed strides (gfx950).
//
// These are the entry points the source-compatible header include/alp.hpp forwards to (one vector per call
// there), and what a caller uses when it owns the column loop itself (benchmarks/benchmark.cpp:118-131 style).
// Reference functions (file:line relative to /root/reference):
//   ffor::ffor / unffor::unffor  u64: src/fastlanes_generated_ffor.cpp:29939 / _unffor.cpp:23010; u16: :29781 / :22846
//   falp                         src/falp.cpp:42440 (include/alp/falp.hpp:10-26)
//   decoder::decode              include/alp/decoder.hpp:134-138
//   decoder::patch_exceptions    include/alp/decoder.hpp:141-149
//   encoder::encode_simdized     include/alp/encoder.hpp:307-400
//   encoder::encode              include/alp/encoder.hpp:402-418
//   encoder::analyze_ffor        include/alp/encoder.hpp:109-120
//   rd_encoder::encode / decode  include/alp/rd.hpp:109-147 / :152-178
// One wavefront per vector, same lane ownership as the column kernels (encode_device.hpp / alp_device.hpp).
#include "encode_device.hpp"
#include "launch.hpp"

namespace alpgpu {

#define ALPGPU_VECTOR_LOOP(v, n)                                                                                        \
	for (uint64_t v = static_cast<uint64_t>(blockIdx.x) * kWavesPerWg + wave_in_wg(); v < (n);                            \
	     v += static_cast<uint64_t>(gridDim.x) * kWavesPerWg)

struct __attribute__((aligned(16))) UnpackLds {
	uint8_t stage[64 * 128 + 128];
};

// ---- FFOR / unFFOR, 64-bit lanes ---------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kWavesPerWg) void k_ffor_i64(const int64_t* __restrict__ in, int64_t* __restrict__ packed,
                                                               size_t packed_stride, const uint8_t* __restrict__ bws,
                                                               const int64_t* __restrict__ bases, uint64_t n) {
	__shared__ EncodeLds lds[kWavesPerWg];
	const int            lane = lane_id();
	EncodeLds&           L    = lds[wave_in_wg()];
	ALPGPU_VECTOR_LOOP(v, n) {
		const int bw = bws[v];
		if (bw >= 1 && bw <= 64) { // bw = 0 writes nothing; other widths are a no-op in the reference's switch
			const uint64_t    base = static_cast<uint64_t>(bases[v]);
			const uint64_t    mask = bw_mask(bw);
			const ulonglong2* src  = reinterpret_cast<const ulonglong2*>(in + v * kVec);
			ulonglong2*       lv   = reinterpret_cast<ulonglong2*>(L.vals);
#pragma unroll
			for (int m = 0; m < 8; ++m) {
				const ulonglong2 x = src[64 * m + lane];
				lv[64 * m + lane]  = make_ulonglong2((x.x - base) & mask, (x.y - base) & mask);
			}
			wave_lds_sync();
			pack_u64_from_lds(L, bw, reinterpret_cast<ulonglong2*>(packed + v * packed_stride), lane);
		}
		wave_lds_sync();
	}
}

// shared by unffor / falp: OUT = 0 integers (+base), OUT = 1 doubles (decode)
template <int OUT>
__global__ __launch_bounds__(64 * kWavesPerWg) void k_unffor_i64(const int64_t* __restrict__ packed, size_t packed_stride,
                                                                 void* __restrict__ out, const uint8_t* __restrict__ bws,
                                                                 const int64_t* __restrict__ bases, const uint8_t* __restrict__ facs,
                                                                 const uint8_t* __restrict__ exps, uint64_t n) {
	__shared__ UnpackLds lds[kWavesPerWg];
	const int            lane = lane_id();
	UnpackLds&           L    = lds[wave_in_wg()];
	ALPGPU_VECTOR_LOOP(v, n) {
		const int bw = bws[v];
		if (bw <= 64) {
			const ulonglong2* g = reinterpret_cast<const ulonglong2*>(packed + v * packed_stride);
			ulonglong2*       s = reinterpret_cast<ulonglong2*>(L.stage);
			for (int c = lane; c < 8 * bw; c += 64) { s[c] = g[c]; }
			wave_lds_sync();
			const uint64_t base = static_cast<uint64_t>(bases[v]);
			const uint64_t mask = bw_mask(bw);
			const int      a = lane & 7, r0 = lane >> 3;
			int64_t        fact = 1;
			double         frac = 1.0;
			if constexpr (OUT == 1) {
				fact = kFactArr[facs[v]];
				frac = kFracArr[exps[v]];
			}
#pragma unroll
			for (int m = 0; m < 8; ++m) {
				const U64Pair u = unpack_pair_u64(UnitsPtr {s}, bw, mask, 8 * m + r0, a);
				if constexpr (OUT == 1) {
					double2 o;
					o.x = decode_value(static_cast<int64_t>(u.x + base), fact, frac);
					o.y = decode_value(static_cast<int64_t>(u.y + base), fact, frac);
					reinterpret_cast<double2*>(static_cast<double*>(out) + v * kVec)[64 * m + lane] = o;
				} else {
					reinterpret_cast<ulonglong2*>(static_cast<int64_t*>(out) + v * kVec)[64 * m + lane] = make_ulonglong2(u.x + base, u.y + base);
				}
			}
		}
		wave_lds_sync();
	}
}

// ---- FFOR / unFFOR, 16-bit lanes: lane L owns stream L (values 64*row + L, row = 0..15) ------------------------
__global__ __launch_bounds__(64 * kWavesPerWg) void k_ffor_u16(const uint16_t* __restrict__ in, uint16_t* __restrict__ packed,
                                                               size_t packed_stride, const uint8_t* __restrict__ bws,
                                                               const uint16_t* __restrict__ bases, uint64_t n) {
	const int lane = lane_id();
	ALPGPU_VECTOR_LOOP(v, n) {
		const int bw = bws[v];
		if (bw < 1 || bw > 16) { continue; }
		const uint32_t base = bases ? bases[v] : 0;
		const uint32_t mask = bw == 16 ? 0xFFFFu : ((1u << bw) - 1u);
		const uint16_t* src = in + v * kVec;
		uint16_t*       dst = packed + v * packed_stride;
		// the stream is 16*bw bits <= 256 bits: four u64 accumulators
		uint64_t acc[4] = {0, 0, 0, 0};
#pragma unroll
		for (int row = 0; row < 16; ++row) {
			const uint64_t val = (static_cast<uint32_t>(src[64 * row + lane]) - base) & mask;
			const int      p   = row * bw;
			const int      q = p >> 6, s = p & 63;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				if (i == q) { acc[i] |= val << s; }
				if (i == q + 1 && s + bw > 64) { acc[i] |= val >> (64 - s); }
			}
		}
		for (int k = 0; k < bw; ++k) {
			uint64_t w = 0;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				if (i == (k >> 2)) { w = acc[i]; }
			}
			dst[64 * k + lane] = static_cast<uint16_t>(w >> (16 * (k & 3)));
		}
	}
}

__global__ __launch_bounds__(64 * kWavesPerWg) void k_unffor_u16(const uint16_t* __restrict__ packed, size_t packed_stride,
                                                                 uint16_t* __restrict__ out, const uint8_t* __restrict__ bws,
                                                                 const uint16_t* __restrict__ bases, uint64_t n) {
	const int lane = lane_id();
	ALPGPU_VECTOR_LOOP(v, n) {
		const int bw = bws[v];
		if (bw > 16) { continue; }
		const uint32_t  base = bases ? bases[v] : 0;
		const uint32_t  mask = bw == 16 ? 0xFFFFu : ((1u << bw) - 1u);
		const uint16_t* src  = packed + v * packed_stride;
		uint16_t*       dst  = out + v * kVec;
		uint64_t        acc[4] = {0, 0, 0, 0};
		for (int k = 0; k < bw; ++k) {
			const uint64_t w = src[64 * k + lane];
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				if (i == (k >> 2)) { acc[i] |= w << (16 * (k & 3)); }
			}
		}
#pragma unroll
		for (int row = 0; row < 16; ++row) {
			const int p = row * bw;
			const int q = p >> 6, s = p & 63;
			uint64_t  lo = 0, hi = 0;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				if (i == q) { lo = acc[i]; }
				if (i == q + 1) { hi = acc[i]; }
			}
			const uint64_t val = ((lo >> s) | ((hi << 1) << (63 - s))) & mask;
			dst[64 * row + lane] = static_cast<uint16_t>(static_cast<uint32_t>(val) + base);
		}
	}
}

// ---- FFOR / unFFOR, 8-bit lanes (src/fastlanes_generated_ffor.cpp:4-356): 128 lane-streams x 8 rows, value i -> stream i & 127,
// row i >> 7, stream word k at out[128*k + stream].  Lane L owns streams 2L and 2L+1: 16-bit accesses, 128 B per instruction.
// A stream is 8*bw <= 64 bits: one u64 accumulator each.
__global__ __launch_bounds__(64 * kWavesPerWg) void k_ffor_u8(const uint8_t* __restrict__ in, uint8_t* __restrict__ packed, size_t packed_stride,
                                                              const uint8_t* __restrict__ bws, const uint8_t* __restrict__ bases, uint64_t n) {
	const int lane = lane_id();
	ALPGPU_VECTOR_LOOP(v, n) {
		const int bw = bws[v];
		if (bw < 1 || bw > 8) { continue; }
		const uint32_t  base = bases ? bases[v] : 0;
		const uint32_t  mask = (1u << bw) - 1u;
		const uint16_t* src  = reinterpret_cast<const uint16_t*>(in + v * kVec);
		uint16_t*       dst  = reinterpret_cast<uint16_t*>(packed + v * packed_stride);
		uint64_t        a0 = 0, a1 = 0;
#pragma unroll
		for (int row = 0; row < 8; ++row) {
			const uint32_t w = src[64 * row + lane];
			a0 |= static_cast<uint64_t>(((w & 0xFFu) - base) & mask) << (row * bw);
			a1 |= static_cast<uint64_t>(((w >> 8) - base) & mask) << (row * bw);
		}
		for (int k = 0; k < bw; ++k) {
			dst[64 * k + lane] = static_cast<uint16_t>((static_cast<uint32_t>(a0 >> (8 * k)) & 0xFFu) | ((static_cast<uint32_t>(a1 >> (8 * k)) & 0xFFu) << 8));
		}
	}
}

__global__ __launch_bounds__(64 * kWavesPerWg) void k_unffor_u8(const uint8_t* __restrict__ packed, size_t packed_stride, uint8_t* __restrict__ out,
                                                                const uint8_t* __restrict__ bws, const uint8_t* __restrict__ bases, uint64_t n) {
	const int lane = lane_id();
	ALPGPU_VECTOR_LOOP(v, n) {
		const int bw = bws[v];
		if (bw > 8) { continue; }
		const uint32_t  base = bases ? bases[v] : 0;
		const uint32_t  mask = (1u << bw) - 1u;
		const uint16_t* src  = reinterpret_cast<const uint16_t*>(packed + v * packed_stride);
		uint16_t*       dst  = reinterpret_cast<uint16_t*>(out + v * kVec);
		uint64_t        a0 = 0, a1 = 0;
		for (int k = 0; k < bw; ++k) {
			const uint32_t w = src[64 * k + lane];
			a0 |= static_cast<uint64_t>(w & 0xFFu) << (8 * k);
			a1 |= static_cast<uint64_t>(w >> 8) << (8 * k);
		}
#pragma unroll
		for (int row = 0; row < 8; ++row) {
			const uint32_t x = (static_cast<uint32_t>(a0 >> (row * bw)) & mask) + base;
			const uint32_t y = (static_cast<uint32_t>(a1 >> (row * bw)) & mask) + base;
			dst[64 * row + lane] = static_cast<uint16_t>((x & 0xFFu) | ((y & 0xFFu) << 8));
		}
	}
}

// ---- decoder::decode, patch_exceptions, analyze_ffor ------------------------------------------------------------
__global__ __launch_bounds__(64 * kWavesPerWg) void k_decode_values(const int64_t* __restrict__ enc, double* __restrict__ out,
                                                                    const uint8_t* __restrict__ facs, const uint8_t* __restrict__ exps,
                                                                    uint64_t n) {
	const int lane = lane_id();
	ALPGPU_VECTOR_LOOP(v, n) {
		const int64_t     fact = kFactArr[facs[v]];
		const double      frac = kFracArr[exps[v]];
		const longlong2*  src  = reinterpret_cast<const longlong2*>(enc + v * kVec);
		double2*          dst  = reinterpret_cast<double2*>(out + v * kVec);
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			const longlong2 x = src[64 * m + lane];
			double2         o;
			o.x = decode_value(x.x, fact, frac);
			o.y = decode_value(x.y, fact, frac);
			dst[64 * m + lane] = o;
		}
	}
}

__global__ __launch_bounds__(64 * kWavesPerWg) void k_patch(double* __restrict__ out, const double* __restrict__ exc,
                                                            const uint16_t* __restrict__ pos, size_t exc_stride,
                                                            const uint16_t* __restrict__ cnts, uint64_t n) {
	const int lane = lane_id();
	ALPGPU_VECTOR_LOOP(v, n) {
		const int cnt = cnts[v];
		// positions are distinct in a well-formed exception list; duplicates resolve to the LAST entry like the
		// reference's sequential loop only if they fall in different 64-entry rounds — keep lists well-formed
		for (int j = lane; j < cnt; j += 64) { out[v * kVec + pos[v * exc_stride + j]] = exc[v * exc_stride + j]; }
	}
}

__global__ __launch_bounds__(64 * kWavesPerWg) void k_analyze_ffor(const int64_t* __restrict__ enc, uint8_t* __restrict__ bws,
                                                                   int64_t* __restrict__ bases, uint64_t n) {
	const int lane = lane_id();
	ALPGPU_VECTOR_LOOP(v, n) {
		const longlong2* src = reinterpret_cast<const longlong2*>(enc + v * kVec);
		int64_t          mn = INT64_MAX, mx = INT64_MIN;
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			const longlong2 x = src[64 * m + lane];
			mn                = x.x < mn ? x.x : mn;
			mn                = x.y < mn ? x.y : mn;
			mx                = x.x > mx ? x.x : mx;
			mx                = x.y > mx ? x.y : mx;
		}
		mn = wave_min_i64(mn);
		mx = wave_max_i64(mx);
		if (lane == 0) {
			bws[v]   = static_cast<uint8_t>(count_bits(mx, mn));
			bases[v] = mn;
		}
	}
}

// ---- encoder::encode_simdized / encoder::encode -------------------------------------------------------------------
// WITH_STATE = false: (fac, exp) given per vector.  true: rowgroup states + optional state index; second-level
// sampling picks (fac, exp), which are written back.
template <bool WITH_STATE>
__global__ __launch_bounds__(64 * kWavesPerWg) void k_encode_values(const double* __restrict__ in,
                                                                    const alpgpu_rowgroup_state* __restrict__ states,
                                                                    const uint32_t* __restrict__ state_idx, double* __restrict__ exc,
                                                                    uint16_t* __restrict__ pos, size_t exc_stride, uint16_t* __restrict__ cnts,
                                                                    int64_t* __restrict__ enc, uint8_t* __restrict__ facs,
                                                                    uint8_t* __restrict__ exps, uint64_t n) {
	__shared__ EncodeLds lds[kWavesPerWg];
	const int            lane = lane_id();
	EncodeLds&           L    = lds[wave_in_wg()];
	ALPGPU_VECTOR_LOOP(v, n) {
		const VecIn x = load_vector(in, v, lane);
		int         e, f;
		if constexpr (WITH_STATE) {
			const alpgpu_rowgroup_state* rgp = states + (state_idx ? static_cast<uint64_t>(state_idx[v]) : v / kRowgroup);
			if (rgp->k > 1) {
				second_level_select(x, rgp, L, lane, e, f);
			} else {
				e = rgp->combos[0];
				f = rgp->combos[1];
			}
			if (lane == 0) {
				facs[v] = static_cast<uint8_t>(f);
				exps[v] = static_cast<uint8_t>(e);
			}
		} else {
			f = facs[v];
			e = exps[v];
		}
		AlpEncoded R;
		encode_alp_registers(x, e, f, lane, R);
		longlong2* dst = reinterpret_cast<longlong2*>(enc + v * kVec);
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			longlong2 o;
			o.x = R.enc[m][0];
			o.y = R.enc[m][1];
			dst[64 * m + lane] = o;
		}
		for_each_exception(R.ballot, lane, [&](int r, int m, int j) {
			exc[v * exc_stride + r] = j == 0 ? x.x[m].x : x.x[m].y;
			pos[v * exc_stride + r] = static_cast<uint16_t>(128 * m + 2 * lane + j);
		});
		if (lane == 0) { cnts[v] = static_cast<uint16_t>(R.cnt); }
	}
}

// ---- rd_encoder::encode / decode on unpacked right/left arrays ---------------------------------------------------------
__global__ __launch_bounds__(64 * kWavesPerWg) void k_rd_encode(const double* __restrict__ in,
                                                                const alpgpu_rowgroup_state* __restrict__ states,
                                                                const uint32_t* __restrict__ state_idx, uint16_t* __restrict__ exc,
                                                                uint16_t* __restrict__ pos, size_t exc_stride, uint16_t* __restrict__ cnts,
                                                                uint64_t* __restrict__ right, uint16_t* __restrict__ left, uint64_t n) {
	const int lane = lane_id();
	ALPGPU_VECTOR_LOOP(v, n) {
		const alpgpu_rowgroup_state* rgp = states + (state_idx ? static_cast<uint64_t>(state_idx[v]) : v / kRowgroup);
		const VecIn                  x   = load_vector(in, v, lane);
		RdEncoded                    R;
		encode_rd_registers(x, *rgp, lane, R);
		ulonglong2* rdst = reinterpret_cast<ulonglong2*>(right + v * kVec);
		uint32_t*   ldst = reinterpret_cast<uint32_t*>(left + v * kVec);
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			rdst[64 * m + lane] = make_ulonglong2(R.right[m][0], R.right[m][1]);
			ldst[64 * m + lane] = static_cast<uint32_t>(R.idx[m][0]) | (static_cast<uint32_t>(R.idx[m][1]) << 16);
		}
		for_each_exception(R.ballot, lane, [&](int r, int m, int j) {
			exc[v * exc_stride + r] = R.left[m][j];
			pos[v * exc_stride + r] = static_cast<uint16_t>(128 * m + 2 * lane + j);
		});
		if (lane == 0) { cnts[v] = static_cast<uint16_t>(R.cnt); }
	}
}

__global__ __launch_bounds__(64 * kWavesPerWg) void k_rd_decode(double* __restrict__ out, const uint64_t* __restrict__ right,
                                                                const uint16_t* __restrict__ left,
                                                                const alpgpu_rowgroup_state* __restrict__ states,
                                                                const uint32_t* __restrict__ state_idx, const uint16_t* __restrict__ exc,
                                                                const uint16_t* __restrict__ pos, size_t exc_stride,
                                                                const uint16_t* __restrict__ cnts, uint64_t n) {
	const int lane = lane_id();
	ALPGPU_VECTOR_LOOP(v, n) {
		const alpgpu_rowgroup_state* rgp = states + (state_idx ? static_cast<uint64_t>(state_idx[v]) : v / kRowgroup);
		const int                    rbw = rgp->rd_rbw;
		const ulonglong2*            rs  = reinterpret_cast<const ulonglong2*>(right + v * kVec);
		const uint32_t*              ls  = reinterpret_cast<const uint32_t*>(left + v * kVec);
		ulonglong2*                  dst = reinterpret_cast<ulonglong2*>(out + v * kVec);
#pragma unroll
		for (int m = 0; m < 8; ++m) {
			const ulonglong2 r  = rs[64 * m + lane];
			const uint32_t   l  = ls[64 * m + lane];
			const uint32_t   i0 = l & 0xFFFFu, i1 = l >> 16;
			const uint64_t   l0 = i0 < 8 ? rgp->rd_dict[i0] : 0; // indices >= 8 occur only at exception slots (patched below)
			const uint64_t   l1 = i1 < 8 ? rgp->rd_dict[i1] : 0;
			dst[64 * m + lane]  = make_ulonglong2((l0 << rbw) | r.x, (l1 << rbw) | r.y);
		}
		// exceptions overwrite (rd.hpp:171-176): other lanes' 16-byte stores above must have completed first
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		const int cnt = cnts[v];
		for (int j = lane; j < cnt; j += 64) {
			const uint32_t p                                  = pos[v * exc_stride + j];
			const uint64_t u                                  = (static_cast<uint64_t>(exc[v * exc_stride + j]) << rbw) | right[v * kVec + p];
			reinterpret_cast<uint64_t*>(out + v * kVec)[p] = u;
		}
	}
}

// ---- launchers -------------------------------------------------------------------------------------------------------------
static unsigned prim_grid(uint64_t n, int n_cus) {
	const uint64_t need = (n + kWavesPerWg - 1) / kWavesPerWg;
	const uint64_t cap  = static_cast<uint64_t>(n_cus) * 16;
	return static_cast<unsigned>(need < cap ? (need ? need : 1) : cap);
}
#define PRIM_LAUNCH(kernel, ...)                                                                                           \
	do {                                                                                                                   \
		if (n == 0) { return ALPGPU_OK; }                                                                                  \
		hipLaunchKernelGGL(kernel, dim3(prim_grid(n, n_cus)), dim3(64 * kWavesPerWg), 0, stream, __VA_ARGS__);             \
		return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;                                               \
	} while (0)

int launch_ffor_i64(hipStream_t stream, int n_cus, const int64_t* in, int64_t* packed, size_t stride, const uint8_t* bw,
                    const int64_t* base, uint64_t n) {
	PRIM_LAUNCH(k_ffor_i64, in, packed, stride, bw, base, n);
}
int launch_unffor_i64(hipStream_t stream, int n_cus, const int64_t* packed, size_t stride, int64_t* out, const uint8_t* bw,
                      const int64_t* base, uint64_t n) {
	PRIM_LAUNCH((k_unffor_i64<0>), packed, stride, static_cast<void*>(out), bw, base, static_cast<const uint8_t*>(nullptr),
	            static_cast<const uint8_t*>(nullptr), n);
}
int launch_falp(hipStream_t stream, int n_cus, const int64_t* packed, size_t stride, double* out, const uint8_t* bw,
                const int64_t* base, const uint8_t* fac, const uint8_t* exp, uint64_t n) {
	PRIM_LAUNCH((k_unffor_i64<1>), packed, stride, static_cast<void*>(out), bw, base, fac, exp, n);
}
int launch_ffor_u16(hipStream_t stream, int n_cus, const uint16_t* in, uint16_t* packed, size_t stride, const uint8_t* bw,
                    const uint16_t* base, uint64_t n) {
	PRIM_LAUNCH(k_ffor_u16, in, packed, stride, bw, base, n);
}
int launch_unffor_u16(hipStream_t stream, int n_cus, const uint16_t* packed, size_t stride, uint16_t* out, const uint8_t* bw,
                      const uint16_t* base, uint64_t n) {
	PRIM_LAUNCH(k_unffor_u16, packed, stride, out, bw, base, n);
}
int launch_ffor_u8(hipStream_t stream, int n_cus, const uint8_t* in, uint8_t* packed, size_t stride, const uint8_t* bw, const uint8_t* base, uint64_t n) {
	PRIM_LAUNCH(k_ffor_u8, in, packed, stride, bw, base, n);
}
int launch_unffor_u8(hipStream_t stream, int n_cus, const uint8_t* packed, size_t stride, uint8_t* out, const uint8_t* bw, const uint8_t* base, uint64_t n) {
	PRIM_LAUNCH(k_unffor_u8, packed, stride, out, bw, base, n);
}
// alp::encoder<double>::encode_value<SAFE> (include/alp/encoder.hpp:81-89) on n VALUES with one (factor, exponent) pair: the scalar helper of
// the reference's public header, offered for source compatibility (include/alp/encoder.hpp here) — element-wise, one value per thread
__global__ __launch_bounds__(256) void k_encode_value(const double* __restrict__ in, int64_t* __restrict__ enc, int fac, int exp, int safe, uint64_t n) {
	const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
	if (i >= n) { return; }
	enc[i] = safe ? encode_value_safe(in[i], kExpArr[exp], kFracArr[fac]) : encode_value_unsafe(in[i], kExpArr[exp], kFracArr[fac]);
}
int launch_encode_value(hipStream_t stream, const double* in, int64_t* enc, int fac, int exp, int safe, uint64_t n) {
	if (n == 0) { return ALPGPU_OK; }
	hipLaunchKernelGGL(k_encode_value, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, in, enc, fac, exp, safe, n);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

int launch_decode_values(hipStream_t stream, int n_cus, const int64_t* enc, double* out, const uint8_t* fac, const uint8_t* exp,
                         uint64_t n) {
	PRIM_LAUNCH(k_decode_values, enc, out, fac, exp, n);
}
int launch_patch(hipStream_t stream, int n_cus, double* out, const double* exc, const uint16_t* pos, size_t stride,
                 const uint16_t* cnt, uint64_t n) {
	PRIM_LAUNCH(k_patch, out, exc, pos, stride, cnt, n);
}
int launch_analyze_ffor(hipStream_t stream, int n_cus, const int64_t* enc, uint8_t* bw, int64_t* base, uint64_t n) {
	PRIM_LAUNCH(k_analyze_ffor, enc, bw, base, n);
}
int launch_encode_simdized(hipStream_t stream, int n_cus, const double* in, double* exc, uint16_t* pos, size_t stride, uint16_t* cnt,
                           int64_t* enc, const uint8_t* fac, const uint8_t* exp, uint64_t n) {
	PRIM_LAUNCH((k_encode_values<false>), in, static_cast<const alpgpu_rowgroup_state*>(nullptr), static_cast<const uint32_t*>(nullptr),
	            exc, pos, stride, cnt, enc, const_cast<uint8_t*>(fac), const_cast<uint8_t*>(exp), n);
}
int launch_encode_values(hipStream_t stream, int n_cus, const double* in, const alpgpu_rowgroup_state* states, const uint32_t* idx,
                         double* exc, uint16_t* pos, size_t stride, uint16_t* cnt, int64_t* enc, uint8_t* fac, uint8_t* exp, uint64_t n) {
	PRIM_LAUNCH((k_encode_values<true>), in, states, idx, exc, pos, stride, cnt, enc, fac, exp, n);
}
int launch_rd_encode(hipStream_t stream, int n_cus, const double* in, const alpgpu_rowgroup_state* states, const uint32_t* idx,
                     uint16_t* exc, uint16_t* pos, size_t stride, uint16_t* cnt, uint64_t* right, uint16_t* left, uint64_t n) {
	PRIM_LAUNCH(k_rd_encode, in, states, idx, exc, pos, stride, cnt, right, left, n);
}
int launch_rd_decode(hipStream_t stream, int n_cus, double* out, const uint64_t* right, const uint16_t* left,
                     const alpgpu_rowgroup_state* states, const uint32_t* idx, const uint16_t* exc, const uint16_t* pos, size_t stride,
                     const uint16_t* cnt, uint64_t n) {
	PRIM_LAUNCH(k_rd_decode, out, right, left, states, idx, exc, pos, stride, cnt, n);
}

} // namespace alpgpu
