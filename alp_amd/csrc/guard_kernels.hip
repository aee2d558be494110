// guard_kernels.hip — opt-in validation of a device-resident column's descriptors (alpgpu_column_validate, include/alpgpu.h).
//
// The decode kernels follow the descriptors as they find them, like the reference's decoder follows the bit width, exception count and
// positions its caller hands it (include/alp/decoder.hpp:141-149 writes out[pos[i]] unchecked; src/falp.cpp reads 16*bw words).  Columns
// from alpgpu_encode_* are well-formed by construction and blobs are validated on the host (api_container.hip: validate_blob_vectors); this kernel
// is the same set of checks for descriptors that reached HBM some other way.  One thread per vector.
#include "alp_device.hpp"
#include "decode_policy.hpp"
#include "launch.hpp"

namespace alpgpu {

__global__ __launch_bounds__(256) void k_validate_column(const alpgpu_vector_desc* __restrict__ descs, const alpgpu_rowgroup_state* __restrict__ rgs,
                                                         const uint8_t* __restrict__ excs, uint64_t n_vectors, uint64_t packed_capacity, uint64_t exc_capacity,
                                                         uint32_t value_bytes, unsigned long long* __restrict__ first_bad) {
	const uint64_t v = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
	if (v >= n_vectors) { return; }
	const alpgpu_vector_desc    d  = descs[v];
	const alpgpu_rowgroup_state rg = rgs[v / kRowgroup];
	const uint32_t vbits = 8u * value_bytes;
	const uint32_t max_e = value_bytes == 8 ? 18u : 10u;
	const bool     alp = d.scheme == ALPGPU_SCHEME_ALP, rd = d.scheme == ALPGPU_SCHEME_ALP_RD;
	bool           ok  = (alp || rd) && rg.scheme == d.scheme;
	const uint64_t psz = 128ull * (d.bw + (rd ? d.lbw : 0));
	const uint64_t esz = ((alp ? value_bytes + 2ull : 4ull) * d.exc_cnt + 7ull) & ~7ull;
	ok = ok && d.bw <= vbits && d.exc_cnt <= 1024 && (!alp || (d.e <= max_e && d.f <= d.e)) &&
	     (!rd || (d.lbw >= 1 && d.lbw <= 3 && d.bw <= vbits - 1 && d.bw == rg.rd_rbw && d.lbw == rg.rd_lbw));
	ok = ok && (d.packed_off & 127ull) == 0 && (d.exc_off & 7ull) == 0 && d.packed_off <= packed_capacity && psz <= packed_capacity - d.packed_off &&
	     d.exc_off <= exc_capacity && esz <= exc_capacity - d.exc_off;
	if (ok && d.exc_cnt) { // only now is the record known to lie inside the stream
		const uint16_t* pos = reinterpret_cast<const uint16_t*>(excs + d.exc_off + static_cast<uint64_t>(alp ? value_bytes : 2u) * d.exc_cnt);
		uint32_t        any = 0;
		for (uint32_t j = 0; j < d.exc_cnt; ++j) { any |= pos[j]; }
		ok = any < 1024;
	}
	if (!ok) { atomicMin(first_bad, static_cast<unsigned long long>(v)); }
}

// How many rowgroups of the column are ALP_RD: one workgroup, *count written once (alpgpu_column_totals -> alpgpu_column::alp_rd_rowgroups_hint)
__global__ __launch_bounds__(256) void k_count_rd_rowgroups(const alpgpu_rowgroup_state* __restrict__ rgs, uint64_t n_rowgroups, uint64_t* __restrict__ count) {
	__shared__ unsigned int s_n;
	if (threadIdx.x == 0) { s_n = 0u; }
	__syncthreads();
	unsigned int mine = 0u;
	for (uint64_t r = threadIdx.x; r < n_rowgroups; r += 256) { mine += rgs[r].scheme == ALPGPU_SCHEME_ALP_RD ? 1u : 0u; }
	if (mine) { atomicAdd(&s_n, mine); }
	__syncthreads();
	if (threadIdx.x == 0) { *count = s_n; }
}
int launch_count_rd_rowgroups(hipStream_t stream, const alpgpu_column* col, uint64_t* d_count) {
	hipLaunchKernelGGL(k_count_rd_rowgroups, dim3(1), dim3(256), 0, stream, col->d_rowgroups, col->n_rowgroups, d_count);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

// Per-segment sums over the descriptors (alpgpu_column_totals -> the context's segment table, api_decode.hip): bytes of packed records, exceptions, vectors of ALP_RD rowgroups
// of every segment of seg_vectors consecutive vectors.  kSegmentSplit workgroups per segment, each over its share, added into out[3 s .. 3 s + 2] (zeroed by the launcher).
constexpr unsigned kSegmentSplit = 16;
__global__ __launch_bounds__(256) void k_segment_sums(const alpgpu_vector_desc* __restrict__ descs, uint64_t n_vectors, uint64_t seg_vectors, unsigned long long* __restrict__ out) {
	__shared__ unsigned long long s_sum[3];
	if (threadIdx.x < 3) { s_sum[threadIdx.x] = 0ull; }
	__syncthreads();
	const uint64_t seg   = blockIdx.x / kSegmentSplit, part = blockIdx.x % kSegmentSplit;
	const uint64_t share = (seg_vectors + kSegmentSplit - 1) / kSegmentSplit;
	const uint64_t s0    = seg * seg_vectors;
	const uint64_t s1    = s0 + seg_vectors < n_vectors ? s0 + seg_vectors : n_vectors;
	const uint64_t v0    = s0 + part * share;
	const uint64_t v1    = v0 + share < s1 ? v0 + share : s1;
	unsigned long long p = 0, e = 0, r = 0;
	for (uint64_t v = v0 + threadIdx.x; v < v1; v += 256) {
		const alpgpu_vector_desc d  = descs[v];
		const bool               rd = d.scheme == ALPGPU_SCHEME_ALP_RD;
		p += 128ull * (d.bw + (rd ? d.lbw : 0));
		e += d.exc_cnt;
		r += rd ? 1ull : 0ull;
	}
	atomicAdd(&s_sum[0], p);
	atomicAdd(&s_sum[1], e);
	atomicAdd(&s_sum[2], r);
	__syncthreads();
	if (threadIdx.x < 3 && s_sum[threadIdx.x] != 0ull) { atomicAdd(&out[3 * seg + threadIdx.x], s_sum[threadIdx.x]); }
}
int launch_segment_sums(hipStream_t stream, const alpgpu_column* col, uint64_t seg_vectors, uint32_t n_seg, uint64_t* d_out) {
	if (hipMemsetAsync(d_out, 0, 24ull * n_seg, stream) != hipSuccess) { return ALPGPU_ERR_HIP; }
	hipLaunchKernelGGL(k_segment_sums, dim3(n_seg * kSegmentSplit), dim3(256), 0, stream, col->d_vectors, col->n_vectors, seg_vectors, reinterpret_cast<unsigned long long*>(d_out));
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

// The plan of an UNHINTED decode (round 6; api_decode.hip: decode_unhinted): the column's sizes are not known on the host — it was encoded a moment ago on this
// stream and nobody has called alpgpu_column_totals, a host synchronisation — so the rule (decode_policy.hpp: policy_unhinted, the same numbers the host's rule
// uses) is evaluated here, from the segment sums k_segment_sums has just left in the context's words, and its answer goes into the plan words: which of the
// candidate launches enqueued behind this kernel runs, and whether, how far ahead and at what pace the read-ahead beside them reads.  One wavefront.
__global__ __launch_bounds__(64) void k_unhinted_plan(uint64_t* __restrict__ words, uint32_t n_seg, uint64_t n_vectors, int value_bytes, int read_ahead_option, int lead_us_option,
                                                      uint32_t max_bits) {
	if (threadIdx.x != 0) { return; }
	uint64_t packed = 0, exceptions = 0, rd = 0;
	for (uint32_t s = 0; s < n_seg; ++s) {
		packed += words[kCtxWordSegments + 3 * s], exceptions += words[kCtxWordSegments + 3 * s + 1], rd += words[kCtxWordSegments + 3 * s + 2];
	}
	const UnhintedChoice c = policy_unhinted(n_vectors, static_cast<double>(packed), static_cast<double>(exceptions), static_cast<double>(rd), value_bytes, read_ahead_option);
	uint64_t lead = 0, pace = 0;
	if (c.ahead) {
		const ReadAheadPace p = policy_read_ahead_pace(static_cast<double>(n_vectors), static_cast<double>(packed), (value_bytes + 2.0) * static_cast<double>(exceptions), value_bytes, lead_us_option);
		lead = (static_cast<uint64_t>(p.lead_max) << 32) | p.lead_min;
		pace = (static_cast<uint64_t>(max_bits) << 32) | p.ps_per_vector;
	}
	words[kCtxWordTotals] = packed, words[kCtxWordTotals + 1] = exceptions, words[kCtxWordTotals + 2] = rd;
	words[kCtxWordLead]  = lead;
	words[kCtxWordPace]  = pace;
	words[kCtxWordShape] = static_cast<uint64_t>(c.shape);
}
int launch_unhinted_plan(hipStream_t stream, uint64_t* d_ctx_words, uint32_t n_seg, uint64_t n_vectors, int value_bytes, int read_ahead_option, int lead_us_option, uint32_t max_bits) {
	hipLaunchKernelGGL(k_unhinted_plan, dim3(1), dim3(64), 0, stream, d_ctx_words, n_seg, n_vectors, value_bytes, read_ahead_option, lead_us_option, max_bits);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

int launch_validate_column(hipStream_t stream, const alpgpu_column* col, uint32_t value_bytes, unsigned long long* d_first_bad) {
	const uint64_t n = col->n_vectors;
	hipLaunchKernelGGL(k_validate_column, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, col->d_vectors, col->d_rowgroups, col->d_exc, n,
	                   col->packed_capacity, col->exc_capacity, value_bytes, d_first_bad);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

} // namespace alpgpu
