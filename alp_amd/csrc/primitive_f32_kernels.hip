// primitive_f32_kernels.hip — the reference's per-vector primitives for float, n vectors at a time, fixed strides (gfx950).
//
// Entry points behind the float instantiations of include/alp.hpp.  Reference functions (file:line relative to /root/reference):
//   ffor::ffor / unffor::unffor  (int32/uint32)  include/fastlanes/ffor.hpp:7-15, unffor.hpp:7-15 (src/fastlanes_generated_ffor.cpp:1776-7378)
//   falp (float)                 include/alp/falp.hpp:28-44
//   decoder<float>::decode / patch_exceptions    include/alp/decoder.hpp:134-149
//   encoder<float>::encode_simdized / encode / analyze_ffor   include/alp/encoder.hpp:307-400 / :402-418 / :109-120
//   rd_encoder<float>::encode / decode           include/alp/rd.hpp:109-147 / :152-178
// One wavefront per vector, lane ownership of encode_f32_device.hpp.
#include "encode_f32_device.hpp"
#include "launch.hpp"

namespace alpgpu {

#define ALPGPU_VECTOR_LOOP(v, n)                                                                                        \
	for (uint64_t v = static_cast<uint64_t>(blockIdx.x) * kWavesPerWg + wave_in_wg(); v < (n);                            \
	     v += static_cast<uint64_t>(gridDim.x) * kWavesPerWg)

struct __attribute__((aligned(16))) UnpackLdsF32 {
	uint8_t stage[32 * 128 + 128];
};

__global__ __launch_bounds__(64 * kWavesPerWg) void k_ffor_i32(const int32_t* __restrict__ in, int32_t* __restrict__ packed, size_t packed_stride,
                                                               const uint8_t* __restrict__ bws, const int32_t* __restrict__ bases, uint64_t n) {
	__shared__ EncodeLdsF32 lds[kWavesPerWg];
	const int               lane = lane_id();
	EncodeLdsF32&           L    = lds[wave_in_wg()];
	ALPGPU_VECTOR_LOOP(v, n) {
		const int bw = bws[v];
		if (bw >= 1 && bw <= 32) { // bw = 0 writes nothing; other widths are a no-op in the reference's switch
			const uint32_t base = static_cast<uint32_t>(bases[v]);
			const uint32_t mask = bw_mask32(bw);
			const u32x4*   src  = reinterpret_cast<const u32x4*>(in + v * kVec);
			u32x4*         lv   = reinterpret_cast<u32x4*>(L.vals);
#pragma unroll
			for (int m = 0; m < 4; ++m) { lv[64 * m + lane] = (src[64 * m + lane] - base) & mask; }
			wave_lds_sync();
			pack_u32_from_lds(L, bw, reinterpret_cast<u32x4*>(packed + v * packed_stride), lane);
		}
		wave_lds_sync();
	}
}

// OUT = 0 integers (+base), OUT = 1 floats (decode)
template <int OUT>
__global__ __launch_bounds__(64 * kWavesPerWg) void k_unffor_i32(const int32_t* __restrict__ packed, size_t packed_stride, void* __restrict__ out,
                                                                 const uint8_t* __restrict__ bws, const int32_t* __restrict__ bases,
                                                                 const uint8_t* __restrict__ facs, const uint8_t* __restrict__ exps, uint64_t n) {
	__shared__ UnpackLdsF32 lds[kWavesPerWg];
	const int               lane = lane_id();
	UnpackLdsF32&           L    = lds[wave_in_wg()];
	ALPGPU_VECTOR_LOOP(v, n) {
		const int bw = bws[v];
		if (bw <= 32) {
			const u32x4* g = reinterpret_cast<const u32x4*>(packed + v * packed_stride);
			u32x4*       s = reinterpret_cast<u32x4*>(L.stage);
			for (int c = lane; c < 8 * bw; c += 64) { s[c] = g[c]; }
			wave_lds_sync();
			const uint32_t base = static_cast<uint32_t>(bases[v]);
			const uint32_t mask = bw_mask32(bw);
			const int      a = lane & 7, r0 = lane >> 3;
			uint32_t       fact = 1;
			float          frac = 1.0f;
			if constexpr (OUT == 1) {
				fact = kFactArrF[facs[v]];
				frac = kFracArrF[exps[v]];
			}
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				const u32x4 u = unpack_quad_u32(s, bw, mask, 8 * m + r0, a) + base;
				if constexpr (OUT == 1) {
					u32x4 o;
#pragma unroll
					for (int c = 0; c < 4; ++c) { o[c] = __float_as_uint(decode_value_f32(static_cast<int32_t>(u[c]), fact, frac)); }
					reinterpret_cast<u32x4*>(static_cast<float*>(out) + v * kVec)[64 * m + lane] = o;
				} else {
					reinterpret_cast<u32x4*>(static_cast<int32_t*>(out) + v * kVec)[64 * m + lane] = u;
				}
			}
		}
		wave_lds_sync();
	}
}

__global__ __launch_bounds__(64 * kWavesPerWg) void k_decode_values_f32(const int32_t* __restrict__ enc, float* __restrict__ out,
                                                                        const uint8_t* __restrict__ facs, const uint8_t* __restrict__ exps, uint64_t n) {
	const int lane = lane_id();
	ALPGPU_VECTOR_LOOP(v, n) {
		const uint32_t fact = kFactArrF[facs[v]];
		const float    frac = kFracArrF[exps[v]];
		const u32x4*   src  = reinterpret_cast<const u32x4*>(enc + v * kVec);
		u32x4*         dst  = reinterpret_cast<u32x4*>(out + v * kVec);
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			const u32x4 x = src[64 * m + lane];
			u32x4       o;
#pragma unroll
			for (int c = 0; c < 4; ++c) { o[c] = __float_as_uint(decode_value_f32(static_cast<int32_t>(x[c]), fact, frac)); }
			dst[64 * m + lane] = o;
		}
	}
}

__global__ __launch_bounds__(64 * kWavesPerWg) void k_patch_f32(float* __restrict__ out, const float* __restrict__ exc, const uint16_t* __restrict__ pos,
                                                                size_t exc_stride, const uint16_t* __restrict__ cnts, uint64_t n) {
	const int lane = lane_id();
	ALPGPU_VECTOR_LOOP(v, n) {
		const int cnt = cnts[v];
		for (int j = lane; j < cnt; j += 64) { out[v * kVec + pos[v * exc_stride + j]] = exc[v * exc_stride + j]; }
	}
}

__global__ __launch_bounds__(64 * kWavesPerWg) void k_analyze_ffor_i32(const int32_t* __restrict__ enc, uint8_t* __restrict__ bws,
                                                                       int32_t* __restrict__ bases, uint64_t n) {
	typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
	const int lane = lane_id();
	ALPGPU_VECTOR_LOOP(v, n) {
		const i32x4* src = reinterpret_cast<const i32x4*>(enc + v * kVec);
		int32_t      mn = INT32_MAX, mx = INT32_MIN;
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			const i32x4 x = src[64 * m + lane];
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				mn = x[c] < mn ? x[c] : mn;
				mx = x[c] > mx ? x[c] : mx;
			}
		}
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1) {
			const int32_t omn = __shfl_xor(mn, d);
			const int32_t omx = __shfl_xor(mx, d);
			mn                = omn < mn ? omn : mn;
			mx                = omx > mx ? omx : mx;
		}
		if (lane == 0) {
			bws[v]   = static_cast<uint8_t>(count_bits32(mx, mn));
			bases[v] = mn;
		}
	}
}

template <bool WITH_STATE>
__global__ __launch_bounds__(64 * kWavesPerWg) void k_encode_values_f32(const float* __restrict__ in, const alpgpu_rowgroup_state* __restrict__ states,
                                                                        const uint32_t* __restrict__ state_idx, float* __restrict__ exc,
                                                                        uint16_t* __restrict__ pos, size_t exc_stride, uint16_t* __restrict__ cnts,
                                                                        int32_t* __restrict__ enc, uint8_t* __restrict__ facs, uint8_t* __restrict__ exps,
                                                                        uint64_t n) {
	__shared__ EncodeLdsF32 lds[kWavesPerWg];
	const int               lane = lane_id();
	EncodeLdsF32&           L    = lds[wave_in_wg()];
	ALPGPU_VECTOR_LOOP(v, n) {
		const VecInF x = load_vector_f32(in, v, lane);
		int          e, f;
		if constexpr (WITH_STATE) {
			const alpgpu_rowgroup_state* rgp = states + (state_idx ? static_cast<uint64_t>(state_idx[v]) : v / kRowgroup);
			if (rgp->k > 1) {
				second_level_select_f32(x, rgp, L, lane, e, f);
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
		AlpEncodedF R;
		encode_alp_registers_f32(x, e, f, lane, R);
		u32x4* dst = reinterpret_cast<u32x4*>(enc + v * kVec);
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			u32x4 o;
#pragma unroll
			for (int j = 0; j < 4; ++j) { o[j] = static_cast<uint32_t>(R.enc[m][j]); }
			dst[64 * m + lane] = o;
		}
		for_each_exception_f32(R.ballot, lane, [&](int r, int m, int j) {
			exc[v * exc_stride + r] = x.x[m][j];
			pos[v * exc_stride + r] = static_cast<uint16_t>(256 * m + 4 * lane + j);
		});
		if (lane == 0) { cnts[v] = static_cast<uint16_t>(R.cnt); }
	}
}

// rd_encoder<float>::encode on unpacked arrays: right [n][1024] u32, left [n][1024] u16 (dictionary index; dict_size at exceptions)
__global__ __launch_bounds__(64 * kWavesPerWg) void k_rd_encode_f32(const float* __restrict__ in, const alpgpu_rowgroup_state* __restrict__ states,
                                                                    const uint32_t* __restrict__ state_idx, uint16_t* __restrict__ exc,
                                                                    uint16_t* __restrict__ pos, size_t exc_stride, uint16_t* __restrict__ cnts,
                                                                    uint32_t* __restrict__ right, uint16_t* __restrict__ left, uint64_t n) {
	const int lane = lane_id();
	ALPGPU_VECTOR_LOOP(v, n) {
		const alpgpu_rowgroup_state* rgp   = states + (state_idx ? static_cast<uint64_t>(state_idx[v]) : v / kRowgroup);
		const VecInF                 x     = load_vector_f32(in, v, lane);
		const int                    rbw   = rgp->rd_rbw;
		const int                    ds    = rgp->rd_dict_size;
		const uint32_t               rmask = bw_mask32(rbw);
		u32x4*                       rdst  = reinterpret_cast<u32x4*>(right + v * kVec);
		uint64_t*                    ldst  = reinterpret_cast<uint64_t*>(left + v * kVec);
		int                          soff  = 0;
		uint32_t                     dict[8]; // read once (see k_encode_fused_f32)
#pragma unroll
		for (int dd = 0; dd < 8; ++dd) { dict[dd] = rgp->rd_dict[dd]; }
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			u32x4    q;
			uint64_t lw = 0;
			uint64_t bal[4];
			uint32_t mine = 0;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const uint32_t bits = __float_as_uint(x.x[m][j]);
				q[j]                = bits & rmask;
				const uint32_t l    = (bits >> rbw) & 0xFFFFu;
				int            idx  = ds;
#pragma unroll
				for (int dd = 7; dd >= 0; --dd) {
					if (dd < ds && dict[dd] == l) { idx = dd; }
				}
				lw |= static_cast<uint64_t>(idx) << (16 * j);
				bal[j] = __ballot(idx == ds);
				mine |= (idx == ds) ? (1u << j) : 0u;
			}
			rdst[64 * m + lane] = q;
			ldst[64 * m + lane] = lw;
			const uint64_t lt   = lanemask_lt64(lane);
			int            r    = soff + __builtin_popcountll(bal[0] & lt) + __builtin_popcountll(bal[1] & lt) + __builtin_popcountll(bal[2] & lt) +
			        __builtin_popcountll(bal[3] & lt);
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				if (mine & (1u << j)) {
					exc[v * exc_stride + r] = static_cast<uint16_t>(__float_as_uint(x.x[m][j]) >> rbw);
					pos[v * exc_stride + r] = static_cast<uint16_t>(256 * m + 4 * lane + j);
					++r;
				}
			}
			soff += __builtin_popcountll(bal[0]) + __builtin_popcountll(bal[1]) + __builtin_popcountll(bal[2]) + __builtin_popcountll(bal[3]);
		}
		if (lane == 0) { cnts[v] = static_cast<uint16_t>(soff); }
	}
}

__global__ __launch_bounds__(64 * kWavesPerWg) void k_rd_decode_f32(float* __restrict__ out, const uint32_t* __restrict__ right, const uint16_t* __restrict__ left,
                                                                    const alpgpu_rowgroup_state* __restrict__ states, const uint32_t* __restrict__ state_idx,
                                                                    const uint16_t* __restrict__ exc, const uint16_t* __restrict__ pos, size_t exc_stride,
                                                                    const uint16_t* __restrict__ cnts, uint64_t n) {
	const int lane = lane_id();
	ALPGPU_VECTOR_LOOP(v, n) {
		const alpgpu_rowgroup_state* rgp = states + (state_idx ? static_cast<uint64_t>(state_idx[v]) : v / kRowgroup);
		const int                    rbw = rgp->rd_rbw;
		const u32x4*                 rs  = reinterpret_cast<const u32x4*>(right + v * kVec);
		const uint64_t*              ls  = reinterpret_cast<const uint64_t*>(left + v * kVec);
		u32x4*                       dst = reinterpret_cast<u32x4*>(out + v * kVec);
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			const u32x4    r = rs[64 * m + lane];
			const uint64_t l = ls[64 * m + lane];
			u32x4          o;
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const uint32_t i = static_cast<uint32_t>(l >> (16 * c)) & 0xFFFFu;
				const uint32_t d = i < 8 ? rgp->rd_dict[i] : 0; // indices >= 8 occur only at exception slots (patched below)
				o[c]             = (d << rbw) | r[c];
			}
			dst[64 * m + lane] = o;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		const int cnt = cnts[v];
		for (int j = lane; j < cnt; j += 64) {
			const uint32_t p                               = pos[v * exc_stride + j];
			reinterpret_cast<uint32_t*>(out + v * kVec)[p] = (static_cast<uint32_t>(exc[v * exc_stride + j]) << rbw) | right[v * kVec + p];
		}
	}
}

static unsigned prim_grid_f32(uint64_t n, int n_cus) {
	const uint64_t need = (n + kWavesPerWg - 1) / kWavesPerWg;
	const uint64_t cap  = static_cast<uint64_t>(n_cus) * 16;
	return static_cast<unsigned>(need < cap ? (need ? need : 1) : cap);
}
#define PRIM_LAUNCH(kernel, ...)                                                                                           \
	do {                                                                                                                   \
		if (n == 0) { return ALPGPU_OK; }                                                                                  \
		hipLaunchKernelGGL(kernel, dim3(prim_grid_f32(n, n_cus)), dim3(64 * kWavesPerWg), 0, stream, __VA_ARGS__);         \
		return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;                                               \
	} while (0)

int launch_ffor_i32(hipStream_t stream, int n_cus, const int32_t* in, int32_t* packed, size_t stride, const uint8_t* bw, const int32_t* base, uint64_t n) {
	PRIM_LAUNCH(k_ffor_i32, in, packed, stride, bw, base, n);
}
int launch_unffor_i32(hipStream_t stream, int n_cus, const int32_t* packed, size_t stride, int32_t* out, const uint8_t* bw, const int32_t* base, uint64_t n) {
	PRIM_LAUNCH((k_unffor_i32<0>), packed, stride, static_cast<void*>(out), bw, base, static_cast<const uint8_t*>(nullptr),
	            static_cast<const uint8_t*>(nullptr), n);
}
int launch_falp_f32(hipStream_t stream, int n_cus, const int32_t* packed, size_t stride, float* out, const uint8_t* bw, const int32_t* base,
                    const uint8_t* fac, const uint8_t* exp, uint64_t n) {
	PRIM_LAUNCH((k_unffor_i32<1>), packed, stride, static_cast<void*>(out), bw, base, fac, exp, n);
}
// alp::encoder<float>::encode_value<SAFE> on n values with one (factor, exponent) pair (as built by the reference's compiler the SAFE branch
// does not exist for float: alp_device_f32.hpp)
__global__ __launch_bounds__(256) void k_encode_value_f32(const float* __restrict__ in, int32_t* __restrict__ enc, int fac, int exp, uint64_t n) {
	const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
	if (i >= n) { return; }
	enc[i] = encode_value_f32(in[i], kExpArrF[exp], kFracArrF[fac]);
}
int launch_encode_value_f32(hipStream_t stream, const float* in, int32_t* enc, int fac, int exp, uint64_t n) {
	if (n == 0) { return ALPGPU_OK; }
	hipLaunchKernelGGL(k_encode_value_f32, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, in, enc, fac, exp, n);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

int launch_decode_values_f32(hipStream_t stream, int n_cus, const int32_t* enc, float* out, const uint8_t* fac, const uint8_t* exp, uint64_t n) {
	PRIM_LAUNCH(k_decode_values_f32, enc, out, fac, exp, n);
}
int launch_patch_f32(hipStream_t stream, int n_cus, float* out, const float* exc, const uint16_t* pos, size_t stride, const uint16_t* cnt, uint64_t n) {
	PRIM_LAUNCH(k_patch_f32, out, exc, pos, stride, cnt, n);
}
int launch_analyze_ffor_i32(hipStream_t stream, int n_cus, const int32_t* enc, uint8_t* bw, int32_t* base, uint64_t n) {
	PRIM_LAUNCH(k_analyze_ffor_i32, enc, bw, base, n);
}
int launch_encode_simdized_f32(hipStream_t stream, int n_cus, const float* in, float* exc, uint16_t* pos, size_t stride, uint16_t* cnt, int32_t* enc,
                               const uint8_t* fac, const uint8_t* exp, uint64_t n) {
	PRIM_LAUNCH((k_encode_values_f32<false>), in, static_cast<const alpgpu_rowgroup_state*>(nullptr), static_cast<const uint32_t*>(nullptr), exc, pos,
	            stride, cnt, enc, const_cast<uint8_t*>(fac), const_cast<uint8_t*>(exp), n);
}
int launch_encode_values_f32(hipStream_t stream, int n_cus, const float* in, const alpgpu_rowgroup_state* states, const uint32_t* idx, float* exc,
                             uint16_t* pos, size_t stride, uint16_t* cnt, int32_t* enc, uint8_t* fac, uint8_t* exp, uint64_t n) {
	PRIM_LAUNCH((k_encode_values_f32<true>), in, states, idx, exc, pos, stride, cnt, enc, fac, exp, n);
}
int launch_rd_encode_f32(hipStream_t stream, int n_cus, const float* in, const alpgpu_rowgroup_state* states, const uint32_t* idx, uint16_t* exc,
                         uint16_t* pos, size_t stride, uint16_t* cnt, uint32_t* right, uint16_t* left, uint64_t n) {
	PRIM_LAUNCH(k_rd_encode_f32, in, states, idx, exc, pos, stride, cnt, right, left, n);
}
int launch_rd_decode_f32(hipStream_t stream, int n_cus, float* out, const uint32_t* right, const uint16_t* left, const alpgpu_rowgroup_state* states,
                         const uint32_t* idx, const uint16_t* exc, const uint16_t* pos, size_t stride, const uint16_t* cnt, uint64_t n) {
	PRIM_LAUNCH(k_rd_decode_f32, out, right, left, states, idx, exc, pos, stride, cnt, n);
}

__global__ void k_pad_tail_f32(float* __restrict__ data, uint64_t n_values) {
	const uint64_t first = n_values & ~1023ull;
	const uint64_t i     = n_values + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i < first + 1024) { data[i] = data[first]; }
}
int launch_pad_tail_f32(hipStream_t stream, float* d_in, uint64_t n_values) {
	if ((n_values & 1023ull) == 0) { return ALPGPU_OK; }
	hipLaunchKernelGGL(k_pad_tail_f32, dim3(4), dim3(256), 0, stream, d_in, n_values);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

} // namespace alpgpu
