// decode_f32_device.hpp — what the float decode kernels share (decode_f32_kernels.hip: one small workgroup per few vectors; decode_stream_f32_kernels.hip:
// persistent workgroups that stream the column): where a quad's packed words come from and how they are requested.
#pragma once
#include "alp_device_f32.hpp"

namespace alpgpu {

// Where a quad's words come from: the workgroup's LDS stage (k_decode_column_f32), or HBM directly through buffer loads bounded to the
// vector's words (k_sink_direct_f32; decode_kernels.hip: BufferWords).
struct QuadWords {
	u32x4    w0, w1; // units 8k + a and 8k + 8 + a: stream words k, k + 1 of the quad's four columns
	uint64_t l0, l1; // ALP_RD: left words (16 k' + group) and + 16
};
struct StagedWordsF {
	const uint8_t* stage;
	__device__ __forceinline__ void units(int i, u32x4& w0, u32x4& w1) const {
		w0 = reinterpret_cast<const u32x4*>(stage)[i];
		w1 = reinterpret_cast<const u32x4*>(stage)[i + 8];
	}
	__device__ __forceinline__ void lefts(int rbw, int i, uint64_t& l0, uint64_t& l1) const {
		l0 = reinterpret_cast<const uint64_t*>(stage + 128 * rbw)[i];
		l1 = reinterpret_cast<const uint64_t*>(stage + 128 * rbw)[i + 16];
	}
};
struct BufferWordsF {
	__amdgpu_buffer_rsrc_t right, left;
	__device__ __forceinline__ void units(int i, u32x4& w0, u32x4& w1) const {
		const uint32_t at = static_cast<uint32_t>(i) * 16u;
		w0 = __builtin_amdgcn_raw_buffer_load_b128(right, at, 0, 0);
		w1 = __builtin_amdgcn_raw_buffer_load_b128(right, at, 128, 0);
	}
	__device__ __forceinline__ void lefts(int, int i, uint64_t& l0, uint64_t& l1) const {
		typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
		const uint32_t at = static_cast<uint32_t>(i) * 8u;
		const u32x2    a = __builtin_amdgcn_raw_buffer_load_b64(left, at, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b64(left, at, 128, 0);
		l0 = (static_cast<uint64_t>(a[1]) << 32) | a[0];
		l1 = (static_cast<uint64_t>(b[1]) << 32) | b[0];
	}
};
// the requests of the quad 4 tid .. 4 tid + 3 (row = tid >> 3, a = tid & 7; ALP_RD: left row tid >> 4, group tid & 15)
template <class WORDS>
__device__ __forceinline__ QuadWords request_quad_f32(const WORDS& words, const alpgpu_vector_desc& d, int tid) {
	QuadWords q;
	q.l0 = q.l1 = 0;
	words.units(8 * (((tid >> 3) * d.bw) >> 5) + (tid & 7), q.w0, q.w1);
	if (d.scheme != ALPGPU_SCHEME_ALP) { words.lefts(d.bw, 16 * (((tid >> 4) * d.lbw) >> 4) + (tid & 15), q.l0, q.l1); }
	return q;
}

} // namespace alpgpu
