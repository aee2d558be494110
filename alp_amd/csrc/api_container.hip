// api_container.hip — the serialized column (blob) and the descriptor checks of include/alpgpu.h (see host_ctx.hpp for the map).
#include "host_ctx.hpp"

extern "C" {

uint64_t alpgpu_blob_size(uint64_t n_vectors, uint64_t packed_bytes, uint64_t exc_bytes) {
	return sizeof(alpgpu_blob_header) + 32ull * ((n_vectors + 99) / 100) + 32ull * n_vectors + align8(packed_bytes) + align8(exc_bytes);
}

static int column_to_blob(alpgpu_ctx* ctx, const alpgpu_column* col, uint64_t n_values, void* h_blob, uint64_t capacity, uint64_t* written,
                          uint64_t value_bytes) {
	ALPGPU_CHECK_CTX(ctx);
	if (!col || !h_blob) { return fail(ALPGPU_ERR_INVALID, "null column or blob"); }
	if (n_values > col->n_vectors * 1024ull || n_values + 1024ull <= col->n_vectors * 1024ull) {
		return fail(ALPGPU_ERR_INVALID, "n_values must lie in the column's last vector");
	}
	uint64_t t[4] = {0, 0, 0, 0};
	if (col->n_vectors) {
		ALPGPU_HIP(hipMemcpyAsync(t, col->d_totals, sizeof(t), hipMemcpyDeviceToHost, ctx->stream));
		ALPGPU_HIP(hipStreamSynchronize(ctx->stream));
	}
	if (t[2]) { return fail(ALPGPU_ERR_CAPACITY, "the column overflowed its streams; nothing to serialise"); }
	if (t[3]) { return fail(ALPGPU_ERR_HIP, "the column's encode did not complete (look-back stall and failed recovery); nothing to serialise"); }
	const uint64_t need = alpgpu_blob_size(col->n_vectors, t[0], t[1]);
	if (written) { *written = need; }
	if (capacity < need) { return fail(ALPGPU_ERR_CAPACITY, "blob buffer too small (size returned in *written)"); }
	alpgpu_blob_header h;
	std::memset(&h, 0, sizeof(h));
	std::memcpy(h.magic, "ALPGPU1", 8);
	h.version = 1, h.header_bytes = sizeof(h), h.n_values = n_values, h.n_vectors = col->n_vectors, h.n_rowgroups = col->n_rowgroups;
	h.packed_bytes = t[0], h.exc_bytes = t[1];
	h.reserved     = value_bytes == 8 ? 0 : value_bytes; // double blobs keep the version-1 encoding (0)
	uint8_t* p = static_cast<uint8_t*>(h_blob);
	std::memcpy(p, &h, sizeof(h));
	p += sizeof(h);
	if (col->n_vectors) {
		ALPGPU_HIP(hipMemcpyAsync(p, col->d_rowgroups, 32ull * col->n_rowgroups, hipMemcpyDeviceToHost, ctx->stream));
		p += 32ull * col->n_rowgroups;
		ALPGPU_HIP(hipMemcpyAsync(p, col->d_vectors, 32ull * col->n_vectors, hipMemcpyDeviceToHost, ctx->stream));
		p += 32ull * col->n_vectors;
		if (t[0]) { ALPGPU_HIP(hipMemcpyAsync(p, col->d_packed, t[0], hipMemcpyDeviceToHost, ctx->stream)); }
		p += align8(t[0]);
		if (t[1]) { ALPGPU_HIP(hipMemcpyAsync(p, col->d_exc, t[1], hipMemcpyDeviceToHost, ctx->stream)); }
		ALPGPU_HIP(hipStreamSynchronize(ctx->stream));
	}
	return ALPGPU_OK;
}

int alpgpu_column_to_blob(alpgpu_ctx* ctx, const alpgpu_column* col, uint64_t n_values, void* h_blob, uint64_t capacity, uint64_t* written) {
	return column_to_blob(ctx, col, n_values, h_blob, capacity, written, 8);
}
int alpgpu_column_to_blob_f32(alpgpu_ctx* ctx, const alpgpu_column* col, uint64_t n_values, void* h_blob, uint64_t capacity, uint64_t* written) {
	return column_to_blob(ctx, col, n_values, h_blob, capacity, written, 4);
}

// header of a serialized column: identity, value type, consistent counts, not truncated
int validate_blob_header(const void* h_blob, uint64_t size, uint64_t value_bytes, alpgpu_blob_header& h) {
	if (size < sizeof(alpgpu_blob_header)) { return fail(ALPGPU_ERR_INVALID, "blob shorter than its header"); }
	std::memcpy(&h, h_blob, sizeof(h));
	if (std::memcmp(h.magic, "ALPGPU1", 8) != 0 || h.version != 1 || h.header_bytes != sizeof(h)) { return fail(ALPGPU_ERR_INVALID, "not an ALPGPU v1 blob"); }
	if ((h.reserved == 0 ? 8ull : h.reserved) != value_bytes) { return fail(ALPGPU_ERR_INVALID, "blob holds a column of the other value type"); }
	if (h.n_rowgroups != (h.n_vectors + 99) / 100 || h.n_values > h.n_vectors * 1024ull || (h.n_vectors && h.n_values + 1024ull <= h.n_vectors * 1024ull)) {
		return fail(ALPGPU_ERR_INVALID, "inconsistent blob header");
	}
	if (h.packed_bytes > (1ull << 56) || h.exc_bytes > (1ull << 56) || size < alpgpu_blob_size(h.n_vectors, h.packed_bytes, h.exc_bytes)) {
		return fail(ALPGPU_ERR_INVALID, "blob truncated");
	}
	return ALPGPU_OK;
}

// vectors [v_begin, v_end) of a blob whose header passed: every extent a kernel will dereference is checked here, so a corrupt
// blob cannot make the decoder read out of bounds
// window (optional) = {p0, p1, e0, e1}: the byte ranges of the two streams that will be resident when these vectors are decoded (the
// chunked host route uploads [p0, p1) / [e0, e1) only) — every record must lie inside them, not merely inside the whole streams.
int validate_blob_vectors(const void* h_blob, const alpgpu_blob_header& h, uint64_t value_bytes, uint64_t v_begin, uint64_t v_end,
                                 const uint64_t* window) {
	const unsigned vbits = static_cast<unsigned>(8 * value_bytes); // 64 or 32
	const unsigned max_e = value_bytes == 8 ? 18u : 10u;
	const uint8_t* p   = static_cast<const uint8_t*>(h_blob) + sizeof(h);
	const auto*    rgs = reinterpret_cast<const alpgpu_rowgroup_state*>(p);
	const auto*    vds = reinterpret_cast<const alpgpu_vector_desc*>(p + 32ull * h.n_rowgroups);
	for (uint64_t v = v_begin; v < v_end; ++v) {
		alpgpu_vector_desc d;
		std::memcpy(&d, vds + v, sizeof(d));
		alpgpu_rowgroup_state rg;
		std::memcpy(&rg, rgs + v / 100, sizeof(rg));
		const bool alp = d.scheme == ALPGPU_SCHEME_ALP, rd = d.scheme == ALPGPU_SCHEME_ALP_RD;
		if ((!alp && !rd) || rg.scheme != d.scheme) { return fail(ALPGPU_ERR_INVALID, "blob: bad scheme in a descriptor"); }
		const uint64_t psz = 128ull * (d.bw + (rd ? d.lbw : 0));
		const uint64_t esz = align8((alp ? value_bytes + 2ull : 4ull) * d.exc_cnt);
		if (d.bw > vbits || d.exc_cnt > 1024 || (alp && (d.e > max_e || d.f > d.e)) ||
		    (rd && (d.lbw < 1 || d.lbw > 3 || d.bw > vbits - 1 || d.bw != rg.rd_rbw || d.lbw != rg.rd_lbw))) {
			return fail(ALPGPU_ERR_INVALID, "blob: descriptor field out of range");
		}
		if ((d.packed_off & 127ull) || (d.exc_off & 7ull) || d.packed_off > h.packed_bytes || psz > h.packed_bytes - d.packed_off || d.exc_off > h.exc_bytes ||
		    esz > h.exc_bytes - d.exc_off) {
			return fail(ALPGPU_ERR_INVALID, "blob: descriptor extent outside its stream");
		}
		if (window != nullptr && ((psz != 0 && (d.packed_off < window[0] || d.packed_off + psz > window[1])) ||
		                          (esz != 0 && (d.exc_off < window[2] || d.exc_off + esz > window[3])))) {
			return fail(ALPGPU_ERR_INVALID, "blob: a vector's record lies outside its chunk's stream range (offsets must ascend with the vector index)");
		}
		if (d.exc_cnt) { // positions must be < 1024
			const uint8_t*  rec = p + 32ull * h.n_rowgroups + 32ull * h.n_vectors + align8(h.packed_bytes) + d.exc_off;
			const uint16_t* pos = reinterpret_cast<const uint16_t*>(rec + (alp ? value_bytes : 2ull) * d.exc_cnt);
			uint32_t        any = 0; // positions are 16-bit: OR them and look at the bits above 1023 once
			for (uint32_t j = 0; j < d.exc_cnt; ++j) {
				uint16_t q;
				std::memcpy(&q, pos + j, 2);
				any |= q;
			}
			if (any >= 1024) { return fail(ALPGPU_ERR_INVALID, "blob: exception position out of range"); }
		}
	}
	return ALPGPU_OK;
}

static int validate_blob(const void* h_blob, uint64_t size, uint64_t value_bytes, alpgpu_blob_header& h) {
	if (int rc = validate_blob_header(h_blob, size, value_bytes, h)) { return rc; }
	return validate_blob_vectors(h_blob, h, value_bytes, 0, h.n_vectors);
}

static int column_from_blob(alpgpu_ctx* ctx, const void* h_blob, uint64_t size, alpgpu_column* col, uint64_t* n_values, uint64_t value_bytes) {
	ALPGPU_CHECK_CTX(ctx);
	if (!h_blob || !col) { return fail(ALPGPU_ERR_INVALID, "null blob or column"); }
	alpgpu_blob_header h;
	if (int rc = validate_blob(h_blob, size, value_bytes, h)) { return rc; }
	if (col->n_vectors != h.n_vectors || col->n_rowgroups != h.n_rowgroups) { return fail(ALPGPU_ERR_INVALID, "column was allocated for a different vector count"); }
	if (col->packed_capacity < h.packed_bytes || col->exc_capacity < h.exc_bytes) { return fail(ALPGPU_ERR_CAPACITY, "column streams too small for the blob"); }
	const uint8_t* p   = static_cast<const uint8_t*>(h_blob) + sizeof(h);
	const auto*    rgs = reinterpret_cast<const alpgpu_rowgroup_state*>(p);
	const auto*    vds = reinterpret_cast<const alpgpu_vector_desc*>(p + 32ull * h.n_rowgroups);
	if (h.n_vectors) {
		ALPGPU_HIP(hipMemcpyAsync(col->d_rowgroups, rgs, 32ull * h.n_rowgroups, hipMemcpyHostToDevice, ctx->stream));
		ALPGPU_HIP(hipMemcpyAsync(col->d_vectors, vds, 32ull * h.n_vectors, hipMemcpyHostToDevice, ctx->stream));
		const uint8_t* ps = p + 32ull * h.n_rowgroups + 32ull * h.n_vectors;
		if (h.packed_bytes) { ALPGPU_HIP(hipMemcpyAsync(col->d_packed, ps, h.packed_bytes, hipMemcpyHostToDevice, ctx->stream)); }
		if (h.exc_bytes) { ALPGPU_HIP(hipMemcpyAsync(col->d_exc, ps + align8(h.packed_bytes), h.exc_bytes, hipMemcpyHostToDevice, ctx->stream)); }
	}
	const uint64_t t[4] = {h.packed_bytes, h.exc_bytes, 0, 0};
	ALPGPU_HIP(hipMemcpyAsync(col->d_totals, t, sizeof(t), hipMemcpyHostToDevice, ctx->stream));
	ALPGPU_HIP(hipStreamSynchronize(ctx->stream));
	col->packed_bytes_hint = h.packed_bytes;
	col->exc_bytes_hint    = h.exc_bytes;
	{
		uint64_t n_rd = 0;
		const alpgpu_rowgroup_state* rgs_h = reinterpret_cast<const alpgpu_rowgroup_state*>(p);
		for (uint64_t r = 0; r < h.n_rowgroups; ++r) { n_rd += rgs_h[r].scheme == ALPGPU_SCHEME_ALP_RD ? 1 : 0; }
		col->alp_rd_rowgroups_hint = 1 + n_rd;
	}
	segment_table_forget(ctx, col);
	if (h.n_vectors >= 2 * kSegmentMinVectors) { // the decode's launch plan (plan_decode_runs): the same sums alpgpu_column_totals takes on the device
		SegmentTable* seg = segment_table_new(ctx, col, h.packed_bytes, h.exc_bytes);
		for (uint32_t i = 0; i < seg->n_seg; ++i) {
			uint64_t       pk = 0, ec = 0, rd = 0;
			const uint64_t v1 = (i + 1) * seg->seg_vectors < h.n_vectors ? (i + 1) * seg->seg_vectors : h.n_vectors;
			for (uint64_t v = i * seg->seg_vectors; v < v1; ++v) {
				const bool is_rd = vds[v].scheme == ALPGPU_SCHEME_ALP_RD;
				pk += 128ull * (vds[v].bw + (is_rd ? vds[v].lbw : 0));
				ec += vds[v].exc_cnt;
				rd += is_rd ? 1 : 0;
			}
			seg->packed[i] = pk, seg->exc_cnt[i] = ec, seg->rd_vectors[i] = rd;
		}
	}
	if (n_values) { *n_values = h.n_values; }
	return ALPGPU_OK;
}

int alpgpu_column_from_blob(alpgpu_ctx* ctx, const void* h_blob, uint64_t size, alpgpu_column* col, uint64_t* n_values) {
	return column_from_blob(ctx, h_blob, size, col, n_values, 8);
}
int alpgpu_column_from_blob_f32(alpgpu_ctx* ctx, const void* h_blob, uint64_t size, alpgpu_column* col, uint64_t* n_values) {
	return column_from_blob(ctx, h_blob, size, col, n_values, 4);
}

int alpgpu_column_validate(alpgpu_ctx* ctx, const alpgpu_column* col, int value_bytes, uint64_t* first_bad) {
	ALPGPU_CHECK_CTX(ctx);
	if (!col || (value_bytes != 8 && value_bytes != 4)) { return fail(ALPGPU_ERR_INVALID, "null column, or value_bytes not 8 / 4"); }
	if (first_bad) { *first_bad = ~0ull; }
	if (col->n_vectors == 0) { return ALPGPU_OK; }
	if (col->n_rowgroups != (col->n_vectors + 99) / 100 || !col->d_vectors || !col->d_rowgroups || (!col->d_exc && col->exc_capacity)) {
		return fail(ALPGPU_ERR_INVALID, "column has no descriptors, or n_rowgroups != ceil(n_vectors / 100)");
	}
	if (int rc = ensure_workspace(ctx, 64)) { return rc; }
	unsigned long long* d_bad = static_cast<unsigned long long*>(ctx->workspace);
	ALPGPU_HIP(hipMemsetAsync(d_bad, 0xFF, 8, ctx->stream));
	if (alpgpu::launch_validate_column(ctx->stream, col, static_cast<uint32_t>(value_bytes), d_bad) != ALPGPU_OK) {
		return fail(ALPGPU_ERR_HIP, "validate launch failed", hipGetLastError());
	}
	unsigned long long bad = 0;
	ALPGPU_HIP(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
	ALPGPU_HIP(hipStreamSynchronize(ctx->stream));
	if (first_bad) { *first_bad = bad; }
	return bad == ~0ull ? ALPGPU_OK : fail(ALPGPU_ERR_INVALID, "column: a descriptor is malformed or points outside its stream (index in *first_bad)");
}

} // extern "C"
