// api_encode.hip — rowgroup search and vector encode entry points of include/alpgpu.h, and the encode's measurement probes (see host_ctx.hpp for the map).
#include "host_ctx.hpp"

extern "C" {

int alpgpu_rowgroup_init_f64(alpgpu_ctx* ctx, const double* d_in, uint64_t n_vectors, alpgpu_column* col) {
	ALPGPU_CHECK_CTX(ctx);
	if (!d_in && n_vectors) { return fail(ALPGPU_ERR_INVALID, "null input"); }
	if (int rc = check_column(col, n_vectors)) { return rc; }
	if (n_vectors == 0) { return ALPGPU_OK; }
	if (alpgpu::launch_rowgroup_init(ctx->stream, d_in, n_vectors, col->d_rowgroups, col->d_rd_order) != ALPGPU_OK) {
		return fail(ALPGPU_ERR_HIP, "rowgroup init launch failed", hipGetLastError());
	}
	return ALPGPU_OK;
}

static int state_from_samples(alpgpu_ctx* ctx, const double* d_samples, uint32_t n_samples, alpgpu_rowgroup_state* d_state, int force_rd, double* d_estimate = nullptr) {
	ALPGPU_CHECK_CTX(ctx);
	if (!d_samples || !d_state) { return fail(ALPGPU_ERR_INVALID, "null samples or state"); }
	if (n_samples == 0 || n_samples > 288) { return fail(ALPGPU_ERR_INVALID, "n_samples must be 1..288 (9 sampled vectors x 32)"); }
	if (alpgpu::launch_state_from_samples(ctx->stream, d_samples, n_samples, d_state, force_rd, d_estimate) != ALPGPU_OK) {
		return fail(ALPGPU_ERR_HIP, "state-from-samples launch failed", hipGetLastError());
	}
	return ALPGPU_OK;
}

int alpgpu_state_from_samples_f64(alpgpu_ctx* ctx, const double* d_samples, uint32_t n_samples, alpgpu_rowgroup_state* d_state) {
	return state_from_samples(ctx, d_samples, n_samples, d_state, 0);
}

int alpgpu_rd_state_from_samples_f64(alpgpu_ctx* ctx, const double* d_samples, uint32_t n_samples, alpgpu_rowgroup_state* d_state) {
	return state_from_samples(ctx, d_samples, n_samples, d_state, 1);
}
// rd_encoder::build_left_parts_dictionary for ONE cut (rd.hpp:33-87): the kernel of alpgpu_rd_state_from_samples with every other cut ruled out
int alpgpu_rd_dictionary_for_cut_f64(alpgpu_ctx* ctx, const double* d_samples, uint32_t n_samples, uint8_t right_bit_width, alpgpu_rowgroup_state* d_state, double* d_estimate) {
	if (right_bit_width < 48 || right_bit_width > 63) { return fail(ALPGPU_ERR_INVALID, "right_bit_width must be 48..63 (a cut of 1..16 bits, rd.hpp:92)"); }
	return state_from_samples(ctx, d_samples, n_samples, d_state, 0x100 | (64 - right_bit_width), d_estimate);
}

// Single pass, with the recovery route enqueued behind it: the two-pass kernels, gated on the stall flag the single pass
// raises when its look-back gives up (d_totals[6]).  No host synchronisation; when nothing stalled — always, in practice —
// the four gated launches cost a few microseconds.  A column is therefore complete whenever this returns ALPGPU_OK and the
// stream has drained, whatever the dispatch order of the single pass was.
// async_states: the rowgroup states are being published by the persistent search on ctx->init_stream (recorded in ctx->ev_join); the
// single pass polls for them, and everything that reads the states plainly — the tag clean-up, the recovery route — waits for that stream
static int encode_vectors_f64(alpgpu_ctx* ctx, const double* d_in, uint64_t n_vectors, alpgpu_column* col, bool async_states) {
	ALPGPU_CHECK_CTX(ctx);
	if (!d_in && n_vectors) { return fail(ALPGPU_ERR_INVALID, "null input"); }
	if (int rc = check_column(col, n_vectors)) { return rc; }
	col->alp_rd_rowgroups_hint = 0; // the column is being rewritten: unknown until alpgpu_column_totals counts again
	if (n_vectors == 0) {
		if (col->d_totals) { ALPGPU_HIP(hipMemsetAsync(col->d_totals, 0, 64, ctx->stream)); }
		return ALPGPU_OK;
	}
	if (int rc = ensure_workspace(ctx, alpgpu::encode_workspace_bytes(n_vectors))) { return rc; }
	uint64_t* ws = static_cast<uint64_t*>(ctx->workspace);
	int       rc;
	if (ctx->encode_two_pass) {
		rc = alpgpu::launch_encode_vectors(ctx->stream, d_in, n_vectors, col, ws, ctx->n_cus);
	} else {
		const int kernel = ctx->encode_kernel | ((ctx->encode_unordered && ctx->encode_kernel == ALPGPU_ENCODE_KERNEL_LEAN) ? alpgpu::kEncodeUnorderedFlag : 0);
		rc = alpgpu::launch_encode_fused(ctx->stream, d_in, n_vectors, col, ws, ctx->force_stall != 0, async_states, ctx->ev_join, ctx->ev_head, kernel); // (waits for / joins the search's stream)
		if (rc == ALPGPU_OK) { rc = alpgpu::launch_encode_vectors(ctx->stream, d_in, n_vectors, col, ws, ctx->n_cus, col->d_totals + 6); }
	}
	if (rc != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "encode launch failed", hipGetLastError()); }
	return workspace_used(ctx);
}
int alpgpu_encode_vectors_f64(alpgpu_ctx* ctx, const double* d_in, uint64_t n_vectors, alpgpu_column* col) {
	segment_table_forget(ctx, col);
	return encode_vectors_f64(ctx, d_in, n_vectors, col, false);
}

// Rowgroup search + vector encode.
// Short columns (and ALPGPU_OPT_ENCODE_ASYNC_INIT = 0, and the two-pass form): one after the other on the context's stream.
// Long columns: the search of the first kAsyncHeadRowgroups rowgroups runs in front (a few tens of microseconds); the rest of the search
// is the PERSISTENT kernel (about one 4-wavefront workgroup per CU, init_kernels.hip) on the context's second stream, started together
// with the single-pass vector encode, which polls for each rowgroup's state as its tiles reach it.  The search is VALU-bound and touches
// 3 % of the bytes, the vector encode is memory-bound with idle issue slots: side by side on the same CUs — one search wavefront per
// SIMD fits beside two encode tiles, registers and LDS — the 0.55 ms the search took in front of a 1 Mi-vector encode disappear into it.
// (Round 1 tried the search of the NEXT chunk as a full-width grid on a second stream: 26 % slower — its 9-wavefront, 62 KiB workgroups
// displaced encode tiles.)  Everything rejoins the context's stream: callers see one stream, as before.
constexpr uint64_t kAsyncHeadRowgroups = 256;
static_assert(kAsyncHeadRowgroups >= 4 * 64, "the persistent search looks at the states of rowgroups 0, 4, .., 252 of the head (init_kernels.hip: walkers)");  // searched in front: what the persistent search needs to get ahead of the encode's front
constexpr uint64_t kAsyncMinRowgroups  = 1024; // shorter columns are not worth two streams
static int encode_vectors_f32(alpgpu_ctx* ctx, const float* d_in, uint64_t n_vectors, alpgpu_column* col, bool async_states);
extern "C++" {
template <class T>
static int encode_with_side_search(alpgpu_ctx* ctx, const T* d_in, uint64_t n_vectors, alpgpu_column* col) {
	constexpr bool f32  = sizeof(T) == 4;
	const uint64_t n_rg = (n_vectors + 99) / 100;
	ALPGPU_CHECK_CTX(ctx);
	if (!d_in) { return fail(ALPGPU_ERR_INVALID, "null input"); }
	if (int rc = check_column(col, n_vectors)) { return rc; }
	static const bool serial = std::getenv("ALPGPU_ASYNC_SERIAL") != nullptr; // experiment: the publishing search IN FRONT of the polling encode, one stream
	hipStream_t       side   = serial ? ctx->stream : ctx->init_stream;
	ALPGPU_HIP(hipMemsetAsync(col->d_rowgroups, alpgpu::kStateUnpublished, 32ull * n_rg, ctx->stream)); // "unpublished": no tag, every word all-ones (alp_device.hpp)
	ALPGPU_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
	if (!serial) { ALPGPU_HIP(hipStreamWaitEvent(ctx->init_stream, ctx->ev_fork, 0)); }
	// the head of the search and, behind it, the persistent rest: both on the side stream; the context's stream meanwhile clears its
	// totals and status words and then waits for the head only
	auto search = [&](uint64_t first, uint64_t count, int grid, uint32_t adaptive_base = 0) {
		if constexpr (f32) {
			return alpgpu::launch_rowgroup_init_async_f32(side, d_in, n_vectors, col->d_rowgroups, col->d_rd_order, first, count, grid);
		} else {
			return alpgpu::launch_rowgroup_init_async(side, d_in, n_vectors, col->d_rowgroups, col->d_rd_order, first, count, grid,
			                                          ctx->encode_kernel == ALPGPU_ENCODE_KERNEL_LEAN && count > static_cast<uint64_t>(grid), adaptive_base);
		}
	};
	if (search(0, kAsyncHeadRowgroups, static_cast<int>(kAsyncHeadRowgroups)) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "rowgroup init launch failed", hipGetLastError()); }
	ALPGPU_HIP(hipEventRecord(ctx->ev_head, side));
	// double columns beside the lean kernel: three search workgroups per CU are launched, two of them leave at once unless the column's head is
	// mostly ALP_RD (whose latency-bound search the encode would wait for anyway; k_rowgroup_init: `walkers`)
	const bool adaptive = !f32 && ctx->async_init_adaptive && ctx->encode_kernel == ALPGPU_ENCODE_KERNEL_LEAN && ctx->async_init_wg_per_cu < 3;
	const int  wg_per_cu = adaptive ? 3 : ctx->async_init_wg_per_cu;
	if (search(kAsyncHeadRowgroups, n_rg - kAsyncHeadRowgroups, ctx->n_cus * wg_per_cu, adaptive ? static_cast<uint32_t>(ctx->n_cus * ctx->async_init_wg_per_cu) : 0u) != ALPGPU_OK) {
		return fail(ALPGPU_ERR_HIP, "rowgroup init launch failed", hipGetLastError());
	}
	ALPGPU_HIP(hipEventRecord(ctx->ev_join, side));
	if constexpr (f32) {
		return encode_vectors_f32(ctx, d_in, n_vectors, col, true);
	} else {
		return encode_vectors_f64(ctx, d_in, n_vectors, col, true);
	}
}
} // extern "C++"

int alpgpu_encode_f64(alpgpu_ctx* ctx, const double* d_in, uint64_t n_vectors, alpgpu_column* col) {
	segment_table_forget(ctx, col);
	const uint64_t n_rg = (n_vectors + 99) / 100;
	if (!ctx || !ctx->async_init || ctx->encode_two_pass || n_rg < kAsyncMinRowgroups) {
		if (int rc = alpgpu_rowgroup_init_f64(ctx, d_in, n_vectors, col)) { return rc; }
		return alpgpu_encode_vectors_f64(ctx, d_in, n_vectors, col);
	}
	return encode_with_side_search(ctx, d_in, n_vectors, col);
}
int alpgpu_rowgroup_init_f32(alpgpu_ctx* ctx, const float* d_in, uint64_t n_vectors, alpgpu_column* col) {
	ALPGPU_CHECK_CTX(ctx);
	if (!d_in && n_vectors) { return fail(ALPGPU_ERR_INVALID, "null input"); }
	if (int rc = check_column(col, n_vectors)) { return rc; }
	if (n_vectors == 0) { return ALPGPU_OK; }
	if (alpgpu::launch_rowgroup_init_f32(ctx->stream, d_in, n_vectors, col->d_rowgroups, col->d_rd_order) != ALPGPU_OK) {
		return fail(ALPGPU_ERR_HIP, "rowgroup init launch failed", hipGetLastError());
	}
	return ALPGPU_OK;
}

static int state_from_samples_f32(alpgpu_ctx* ctx, const float* d_samples, uint32_t n_samples, alpgpu_rowgroup_state* d_state, int force_rd, double* d_estimate = nullptr) {
	ALPGPU_CHECK_CTX(ctx);
	if (!d_samples || !d_state) { return fail(ALPGPU_ERR_INVALID, "null samples or state"); }
	if (n_samples == 0 || n_samples > 288) { return fail(ALPGPU_ERR_INVALID, "n_samples must be 1..288 (9 sampled vectors x 32)"); }
	if (alpgpu::launch_state_from_samples_f32(ctx->stream, d_samples, n_samples, d_state, force_rd, d_estimate) != ALPGPU_OK) {
		return fail(ALPGPU_ERR_HIP, "state-from-samples launch failed", hipGetLastError());
	}
	return ALPGPU_OK;
}
int alpgpu_state_from_samples_f32(alpgpu_ctx* ctx, const float* d_samples, uint32_t n_samples, alpgpu_rowgroup_state* d_state) {
	return state_from_samples_f32(ctx, d_samples, n_samples, d_state, 0);
}
int alpgpu_rd_state_from_samples_f32(alpgpu_ctx* ctx, const float* d_samples, uint32_t n_samples, alpgpu_rowgroup_state* d_state) {
	return state_from_samples_f32(ctx, d_samples, n_samples, d_state, 1);
}
int alpgpu_rd_dictionary_for_cut_f32(alpgpu_ctx* ctx, const float* d_samples, uint32_t n_samples, uint8_t right_bit_width, alpgpu_rowgroup_state* d_state, double* d_estimate) {
	if (right_bit_width < 16 || right_bit_width > 31) { return fail(ALPGPU_ERR_INVALID, "right_bit_width must be 16..31 (a cut of 1..16 bits, rd.hpp:92)"); }
	return state_from_samples_f32(ctx, d_samples, n_samples, d_state, 0x100 | (32 - right_bit_width), d_estimate);
}

static int encode_vectors_f32(alpgpu_ctx* ctx, const float* d_in, uint64_t n_vectors, alpgpu_column* col, bool async_states) { // see encode_vectors_f64
	ALPGPU_CHECK_CTX(ctx);
	if (!d_in && n_vectors) { return fail(ALPGPU_ERR_INVALID, "null input"); }
	if (int rc = check_column(col, n_vectors)) { return rc; }
	col->alp_rd_rowgroups_hint = 0; // the column is being rewritten: unknown until alpgpu_column_totals counts again
	if (n_vectors == 0) {
		if (col->d_totals) { ALPGPU_HIP(hipMemsetAsync(col->d_totals, 0, 64, ctx->stream)); }
		return ALPGPU_OK;
	}
	if (int rc = ensure_workspace(ctx, alpgpu::encode_workspace_bytes(n_vectors))) { return rc; }
	uint64_t* ws = static_cast<uint64_t*>(ctx->workspace);
	int       rc;
	if (ctx->encode_two_pass) {
		rc = alpgpu::launch_encode_vectors_f32(ctx->stream, d_in, n_vectors, col, ws);
	} else {
		rc = alpgpu::launch_encode_fused_f32(ctx->stream, d_in, n_vectors, col, ws, ctx->force_stall != 0, async_states, ctx->ev_join, ctx->ev_head, ctx->encode_unordered != 0);
		if (rc == ALPGPU_OK) { rc = alpgpu::launch_encode_vectors_f32(ctx->stream, d_in, n_vectors, col, ws, col->d_totals + 6); }
	}
	if (rc != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "encode launch failed", hipGetLastError()); }
	return workspace_used(ctx);
}
int alpgpu_encode_vectors_f32(alpgpu_ctx* ctx, const float* d_in, uint64_t n_vectors, alpgpu_column* col) {
	segment_table_forget(ctx, col);
	return encode_vectors_f32(ctx, d_in, n_vectors, col, false);
}

// Float columns keep the search in front unless ALPGPU_OPT_ENCODE_ASYNC_INIT = 2: the float single pass needs 77 VGPRs, THREE of its
// tiles fit a CU, and the persistent search's wavefront takes one of them away for as long as it lives — 3.53 against 3.38 ms per 1 Mi
// vectors (profiles/r03_async_init.txt); beside the double kernel's two tiles it fits in what they leave.
int alpgpu_encode_f32(alpgpu_ctx* ctx, const float* d_in, uint64_t n_vectors, alpgpu_column* col) {
	segment_table_forget(ctx, col);
	const uint64_t n_rg = (n_vectors + 99) / 100;
	if (!ctx || ctx->async_init < 2 || ctx->encode_two_pass || n_rg < kAsyncMinRowgroups) {
		if (int rc = alpgpu_rowgroup_init_f32(ctx, d_in, n_vectors, col)) { return rc; }
		return alpgpu_encode_vectors_f32(ctx, d_in, n_vectors, col);
	}
	return encode_with_side_search(ctx, d_in, n_vectors, col);
}
// measurement aid: the single-pass encode's loads and stores without its arithmetic (encode_kernels.hip: k_traffic_probe)
int alpgpu_debug_traffic_probe(alpgpu_ctx* ctx, const void* d_in, void* d_out, uint64_t n_vectors, uint32_t write_bytes_per_vector) {
	ALPGPU_CHECK_CTX(ctx);
	if (n_vectors == 0) { return ALPGPU_OK; }
	if (!d_in || !d_out || write_bytes_per_vector % 16u != 0 || write_bytes_per_vector > 8192u) { return fail(ALPGPU_ERR_INVALID, "bad probe arguments"); }
	if (alpgpu::launch_traffic_probe(ctx->stream, d_in, d_out, n_vectors, write_bytes_per_vector) != ALPGPU_OK) {
		return fail(ALPGPU_ERR_HIP, "traffic probe launch failed", hipGetLastError());
	}
	return ALPGPU_OK;
}

// measurement aid: the same probe with the encode's rowgroup search running BESIDE it exactly as beside alpgpu_encode_f64 (head in front on the
// side stream, then the persistent kernel, same grid and adaptive rule) — what the encode's loads and stores cost when they share the CUs with
// the search.  The probe does not read the states; `scratch` receives them (d_rowgroups, d_rd_order of a column of n_vectors vectors).
int alpgpu_debug_traffic_probe_with_search(alpgpu_ctx* ctx, const double* d_in, void* d_out, uint64_t n_vectors, uint32_t write_bytes_per_vector, alpgpu_column* scratch) {
	ALPGPU_CHECK_CTX(ctx);
	const uint64_t n_rg = (n_vectors + 99) / 100;
	if (!d_in || !d_out || !scratch || !scratch->d_rowgroups || write_bytes_per_vector % 16u != 0 || write_bytes_per_vector > 8192u) { return fail(ALPGPU_ERR_INVALID, "bad probe arguments"); }
	if (n_rg < kAsyncMinRowgroups) { return fail(ALPGPU_ERR_INVALID, "the search runs beside the encode from 1024 rowgroups on only"); }
	ALPGPU_HIP(hipMemsetAsync(scratch->d_rowgroups, alpgpu::kStateUnpublished, 32ull * n_rg, ctx->stream));
	ALPGPU_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
	ALPGPU_HIP(hipStreamWaitEvent(ctx->init_stream, ctx->ev_fork, 0));
	const bool lean = ctx->encode_kernel == ALPGPU_ENCODE_KERNEL_LEAN;
	if (alpgpu::launch_rowgroup_init_async(ctx->init_stream, d_in, n_vectors, scratch->d_rowgroups, scratch->d_rd_order, 0, kAsyncHeadRowgroups, static_cast<int>(kAsyncHeadRowgroups), false, 0) != ALPGPU_OK) {
		return fail(ALPGPU_ERR_HIP, "rowgroup init launch failed", hipGetLastError());
	}
	ALPGPU_HIP(hipEventRecord(ctx->ev_head, ctx->init_stream));
	const bool adaptive  = ctx->async_init_adaptive && lean && ctx->async_init_wg_per_cu < 3;
	const int  wg_per_cu = adaptive ? 3 : ctx->async_init_wg_per_cu;
	if (alpgpu::launch_rowgroup_init_async(ctx->init_stream, d_in, n_vectors, scratch->d_rowgroups, scratch->d_rd_order, kAsyncHeadRowgroups, n_rg - kAsyncHeadRowgroups, ctx->n_cus * wg_per_cu,
	                                       lean, adaptive ? static_cast<uint32_t>(ctx->n_cus * ctx->async_init_wg_per_cu) : 0u) != ALPGPU_OK) {
		return fail(ALPGPU_ERR_HIP, "rowgroup init launch failed", hipGetLastError());
	}
	ALPGPU_HIP(hipEventRecord(ctx->ev_join, ctx->init_stream));
	ALPGPU_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_head, 0));
	if (alpgpu::launch_traffic_probe(ctx->stream, d_in, d_out, n_vectors, write_bytes_per_vector) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "traffic probe launch failed", hipGetLastError()); }
	ALPGPU_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
	return ALPGPU_OK;
}
// ---- tail padding + blob container ---------------------------------------------------------------------------------
int alpgpu_pad_tail_f64(alpgpu_ctx* ctx, double* d_in, uint64_t n_values) {
	ALPGPU_CHECK_CTX(ctx);
	if (!d_in && n_values) { return fail(ALPGPU_ERR_INVALID, "null input"); }
	if (alpgpu::launch_pad_tail(ctx->stream, d_in, n_values) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "pad launch failed", hipGetLastError()); }
	return ALPGPU_OK;
}
int alpgpu_pad_tail_f32(alpgpu_ctx* ctx, float* d_in, uint64_t n_values) {
	ALPGPU_CHECK_CTX(ctx);
	if (!d_in && n_values) { return fail(ALPGPU_ERR_INVALID, "null input"); }
	if (alpgpu::launch_pad_tail_f32(ctx->stream, d_in, n_values) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "pad launch failed", hipGetLastError()); }
	return ALPGPU_OK;
}

} // extern "C"
