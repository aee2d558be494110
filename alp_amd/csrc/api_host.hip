// api_host.hip — columns that live in HOST memory: the chunked two-stream pipelines over one or several contexts (include/alpgpu.h; see host_ctx.hpp for the map).
#include "host_ctx.hpp"

extern "C" {

// ==== host-resident columns ============================================================================================
// alpgpu_compress_host_* / alpgpu_decompress_host_*: the column AND its serialized form live in host memory (the reference's callers,
// publication/source_code/bench_compression_ratio/alp.cpp:198-229, hold both there).  Two streams, two chunk slots of whole rowgroups:
// while chunk i is copied up on one stream, chunk i-1 is encoded on the other.  A chunk is encoded into its slot's column by the
// ordinary (self-healing) encode; its packed bytes come straight down to their place in the blob (known chunk by chunk), its exception
// bytes are appended to the column's exception stream in HBM (its place in the blob depends on the packed stream's final size) and come
// down in one copy at the end; its descriptors go straight to their place in the blob and are shifted by the bytes of the chunks
// before on the host (a column cut at rowgroup boundaries is the sum of its parts: tests/test_sharding*.py).  Decompression mirrors it:
// the streams go up chunk by chunk, each chunk is decoded from a view of the column (descriptors hold absolute offsets), the doubles
// come down on the chunk's stream while the next chunk is on its way up.
extern "C++" {
namespace {

constexpr uint64_t kHostChunkVectors = 12800; // 128 rowgroups: 100 MiB of doubles per copy

struct HostPipe { // everything a call allocates, released on every return path
	hipStream_t stream[2] = {nullptr, nullptr};
	void*       d_in[2]   = {nullptr, nullptr};
	void*       dev[16]   = {nullptr};
	int         n_dev     = 0;
	hipStream_t saved     = nullptr;
	int         saved_unordered = 0;
	int         saved_read_ahead = 0, saved_unhinted = 0;
	alpgpu_ctx* ctx       = nullptr;
	// The pipeline alternates ctx->stream between its two streams, so two chunks' decodes are in flight at once: they must not share the context's ONE progress word,
	// fork / join events and side stream.  Chunks are far below the read-ahead's thresholds today; the pipeline says so instead of relying on it.
	void bind(alpgpu_ctx* c) {
		ctx              = c;
		saved            = c->stream;
		saved_unordered  = c->encode_unordered;
		saved_read_ahead = c->read_ahead, saved_unhinted = c->decode_unhinted;
		c->read_ahead = 0, c->decode_unhinted = 0;
	}
	~HostPipe() {
		if (ctx) {
			ctx->stream           = saved;
			ctx->encode_unordered = saved_unordered;
			ctx->read_ahead = saved_read_ahead, ctx->decode_unhinted = saved_unhinted;
		}
		for (int k = 0; k < 2; ++k) {
			if (stream[k]) { (void)hipStreamSynchronize(stream[k]); }
		}
		for (int i = 0; i < n_dev; ++i) { (void)hipFree(dev[i]); }
		for (int k = 0; k < 2; ++k) {
			if (d_in[k]) { (void)hipFree(d_in[k]); }
			if (stream[k]) { (void)hipStreamDestroy(stream[k]); }
		}
	}
	int alloc(void** p, uint64_t bytes) {
		if (hipMalloc(p, bytes ? bytes : 8) != hipSuccess) { return fail(ALPGPU_ERR_HIP, "hipMalloc (host pipeline)", hipGetLastError()); }
		dev[n_dev++] = *p;
		return ALPGPU_OK;
	}
};

// One contiguous piece of a column (whole rowgroups; the column's last piece may end in an incomplete vector) through the pipeline of
// ctx: rowgroup states -> out_rg, descriptors -> out_vec (offsets counted from the piece's own streams), packed stream -> out_str, exception
// stream -> out_str + align8(packed bytes).  out_cap = bytes available at out_str.  *pb / *eb = the streams' sizes (also when they do not
// fit: ALPGPU_ERR_CAPACITY, nothing usable written).  A whole column is one piece (compress_host); N pieces on N contexts are N of these
// side by side (compress_host_multi).
template <int VALUE_BYTES>
int compress_host_piece(alpgpu_ctx* ctx, const void* h_in, uint64_t n_values, uint8_t* blob_rg, uint8_t* blob_vec, uint8_t* blob_str, uint64_t out_cap,
                        uint64_t* out_pb, uint64_t* out_eb) {
	ALPGPU_CHECK_CTX(ctx);
	const uint64_t n        = (n_values + 1023) / 1024;
	const uint64_t VB       = 1024ull * VALUE_BYTES;
	const uint64_t capacity = out_cap; // of the streams
	HostPipe P;
	P.bind(ctx);
	ctx->encode_unordered = 0; // a blob's streams are in vector order (byte for byte the reference's; the chunked decompression relies on it)
	uint64_t total_p = 0, total_e = 0;
	bool     blob_full = false;
	std::vector<uint64_t> chunk_p, chunk_e; // bytes in front of every chunk
	void*    d_exc_all = nullptr;
	uint64_t cap_e_all = 0;
	if (n) {
		const uint64_t chunk   = n < kHostChunkVectors ? n : kHostChunkVectors;
		const uint64_t c_nrg   = (chunk + 99) / 100;
		const uint64_t c_cap_p = VALUE_BYTES == 8 ? alpgpu_packed_capacity(chunk) : alpgpu_packed_capacity_f32(chunk);
		const uint64_t c_cap_e = VALUE_BYTES == 8 ? alpgpu_exc_capacity(chunk) : alpgpu_exc_capacity_f32(chunk);
		alpgpu_column col[2];
		for (int k = 0; k < 2; ++k) {
			ALPGPU_HIP(hipStreamCreateWithFlags(&P.stream[k], hipStreamNonBlocking));
			if (hipMalloc(&P.d_in[k], chunk * VB) != hipSuccess) { return fail(ALPGPU_ERR_HIP, "hipMalloc (chunk buffer)", hipGetLastError()); }
			std::memset(&col[k], 0, sizeof(col[k]));
			col[k].packed_capacity = c_cap_p, col[k].exc_capacity = c_cap_e;
			if (int rc = P.alloc(reinterpret_cast<void**>(&col[k].d_rowgroups), 32ull * c_nrg)) { return rc; }
			if (int rc = P.alloc(reinterpret_cast<void**>(&col[k].d_vectors), 32ull * chunk)) { return rc; }
			if (int rc = P.alloc(reinterpret_cast<void**>(&col[k].d_packed), c_cap_p)) { return rc; }
			if (int rc = P.alloc(reinterpret_cast<void**>(&col[k].d_exc), c_cap_e)) { return rc; }
			if (int rc = P.alloc(reinterpret_cast<void**>(&col[k].d_totals), 64)) { return rc; }
			// the reference's sorted order per rowgroup: ALP_RD streams byte-identical to the reference's even at exception slots
			if (int rc = P.alloc(reinterpret_cast<void**>(&col[k].d_rd_order), 2ull * ALPGPU_RD_ORDER_STRIDE * c_nrg)) { return rc; }
		}
		// the column's exception stream in HBM: its worst case (1.25 x the input) is taken when that is a small part of the free
		// memory, else a quarter of the input (an error, not an overrun, if a column ever needs more)
		size_t free_b = 0, total_b = 0;
		ALPGPU_HIP(hipMemGetInfo(&free_b, &total_b));
		cap_e_all = n * (VALUE_BYTES == 8 ? 10240ull : 6144ull) + 64;
		if (cap_e_all > free_b / 4) { cap_e_all = n * VB / 4 + 4096; }
		if (int rc = P.alloc(&d_exc_all, cap_e_all)) { return rc; }
		const uint64_t n_chunks = (n + chunk - 1) / chunk;
		// chunk i: copy up, (pad,) encode — everything that needs nothing from the host
		auto send_up = [&](uint64_t i) -> int {
			const int      k   = static_cast<int>(i & 1);
			const uint64_t v0  = i * chunk;
			const uint64_t cnt = n - v0 < chunk ? n - v0 : chunk;
			const uint64_t val = (v0 + cnt) * 1024 <= n_values ? cnt * 1024 : n_values - v0 * 1024; // values of this chunk present in h_in
			ctx->stream        = P.stream[k];
			ALPGPU_HIP(hipMemcpyAsync(P.d_in[k], static_cast<const uint8_t*>(h_in) + v0 * VB, val * VALUE_BYTES, hipMemcpyHostToDevice, P.stream[k]));
			if (val != cnt * 1024) { // the column's last vector is incomplete: padded with its first value (alpgpu_pad_tail_*)
				const int rc = VALUE_BYTES == 8 ? alpgpu::launch_pad_tail(P.stream[k], static_cast<double*>(P.d_in[k]), val)
				                                : alpgpu::launch_pad_tail_f32(P.stream[k], static_cast<float*>(P.d_in[k]), val);
				if (rc != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "pad launch failed", hipGetLastError()); }
			}
			col[k].n_vectors = cnt, col[k].n_rowgroups = (cnt + 99) / 100;
			return VALUE_BYTES == 8 ? alpgpu_encode_f64(ctx, static_cast<const double*>(P.d_in[k]), cnt, &col[k])
			                        : alpgpu_encode_f32(ctx, static_cast<const float*>(P.d_in[k]), cnt, &col[k]);
		};
		if (int rc = send_up(0)) { return rc; }
		for (uint64_t i = 0; i < n_chunks; ++i) {
			const int      k   = static_cast<int>(i & 1);
			const uint64_t v0  = i * chunk;
			const uint64_t cnt = n - v0 < chunk ? n - v0 : chunk;
			// The next chunk is on its way BEFORE the host blocks on this one's sizes: the link never waits for the host.  (Its slot
			// was last used by chunk i - 1, whose copies down are ahead of it on the same stream.)
			if (i + 1 < n_chunks) {
				if (int rc = send_up(i + 1)) { return rc; }
			}
			ctx->stream = P.stream[k];
			uint64_t pb = 0, eb = 0;
			int      ov = 0;
			if (int rc2 = alpgpu_column_totals(ctx, &col[k], &pb, &eb, &ov)) { return rc2; } // waits for this chunk only
			if (total_e + eb > cap_e_all) {
				// The reserve was a guess (a quarter of the input when the worst case does not fit comfortably): grow it to what the rest of
				// the column can need at most, keep what has been collected.  Only when even that cannot be had does the call fail — with
				// its own text, and *written = the size that always suffices, so that a caller's "retry with a larger blob" stops here.
				const uint64_t rest = n - v0 - cnt;
				const uint64_t want = total_e + eb + rest * (VALUE_BYTES == 8 ? 10240ull : 6144ull) + 64;
				void*          grown = nullptr;
				ALPGPU_HIP(hipStreamSynchronize(P.stream[0])); // copies into the old reserve
				ALPGPU_HIP(hipStreamSynchronize(P.stream[1]));
				if (hipMalloc(&grown, want) != hipSuccess) {
					(void)hipGetLastError();
					*out_pb = total_p + pb, *out_eb = want;
					return fail(ALPGPU_ERR_HIP, "the column's exception stream outgrew the pipeline's reserve and HBM has no room for a larger one (encode the column in pieces)");
				}
				if (total_e) { ALPGPU_HIP(hipMemcpy(grown, d_exc_all, total_e, hipMemcpyDeviceToDevice)); }
				for (int q = 0; q < P.n_dev; ++q) {
					if (P.dev[q] == d_exc_all) { P.dev[q] = grown; }
				}
				(void)hipFree(d_exc_all);
				d_exc_all = grown;
				cap_e_all = want;
			}
			chunk_p.push_back(total_p);
			chunk_e.push_back(total_e);
			// the packed stream's place in the blob is known chunk by chunk: it comes down at once, under the next chunk's copy up and
			// encode; the exception stream's place depends on the packed stream's final size, so it collects in HBM
			if (total_p + pb <= capacity) {
				if (pb) { ALPGPU_HIP(hipMemcpyAsync(blob_str + total_p, col[k].d_packed, pb, hipMemcpyDeviceToHost, P.stream[k])); }
			} else {
				blob_full = true; // keep counting: the caller learns the size it needs
			}
			if (eb) { ALPGPU_HIP(hipMemcpyAsync(static_cast<uint8_t*>(d_exc_all) + total_e, col[k].d_exc, eb, hipMemcpyDeviceToDevice, P.stream[k])); }
			ALPGPU_HIP(hipMemcpyAsync(blob_rg + 32ull * (v0 / 100), col[k].d_rowgroups, 32ull * col[k].n_rowgroups, hipMemcpyDeviceToHost, P.stream[k]));
			ALPGPU_HIP(hipMemcpyAsync(blob_vec + 32ull * v0, col[k].d_vectors, 32ull * cnt, hipMemcpyDeviceToHost, P.stream[k]));
			total_p += pb;
			total_e += eb;
		}
		ALPGPU_HIP(hipStreamSynchronize(P.stream[0]));
		ALPGPU_HIP(hipStreamSynchronize(P.stream[1]));
	}
	*out_pb = total_p, *out_eb = total_e;
	if (blob_full || align8(total_p) + total_e > capacity) { return fail(ALPGPU_ERR_CAPACITY, "blob buffer too small (size returned in *written)"); }
	if (n) {
		if (total_e) { ALPGPU_HIP(hipMemcpyAsync(blob_str + align8(total_p), d_exc_all, total_e, hipMemcpyDeviceToHost, P.stream[1])); }
		// meanwhile: the chunks' descriptors become the column's (offsets continue where the chunks before ended)
		const uint64_t chunk = n < kHostChunkVectors ? n : kHostChunkVectors;
		for (uint64_t v = chunk; v < n; ++v) { // chunk 0 is in place already
			alpgpu_vector_desc d;
			std::memcpy(&d, blob_vec + 32ull * v, sizeof(d));
			d.packed_off += chunk_p[v / chunk];
			d.exc_off += chunk_e[v / chunk];
			std::memcpy(blob_vec + 32ull * v, &d, sizeof(d));
		}
		if (total_p != align8(total_p)) { std::memset(blob_str + total_p, 0, align8(total_p) - total_p); }
		ALPGPU_HIP(hipStreamSynchronize(P.stream[0]));
		ALPGPU_HIP(hipStreamSynchronize(P.stream[1]));
	}
	return ALPGPU_OK;
}

template <int VALUE_BYTES>
uint64_t worst_case_blob(uint64_t n) {
	return VALUE_BYTES == 8 ? alpgpu_blob_size(n, alpgpu_packed_capacity(n), alpgpu_exc_capacity(n)) : alpgpu_blob_size(n, alpgpu_packed_capacity_f32(n), alpgpu_exc_capacity_f32(n));
}

void write_blob_header(void* h_blob, uint64_t n_values, uint64_t n, uint64_t total_p, uint64_t total_e, int value_bytes) {
	alpgpu_blob_header h;
	std::memset(&h, 0, sizeof(h));
	std::memcpy(h.magic, "ALPGPU1", 8);
	h.version = 1, h.header_bytes = sizeof(h), h.n_values = n_values, h.n_vectors = n, h.n_rowgroups = (n + 99) / 100;
	h.packed_bytes = total_p, h.exc_bytes = total_e;
	h.reserved     = value_bytes == 8 ? 0 : static_cast<uint64_t>(value_bytes);
	std::memcpy(h_blob, &h, sizeof(h));
}

template <int VALUE_BYTES>
int compress_host(alpgpu_ctx* ctx, const void* h_in, uint64_t n_values, void* h_blob, uint64_t capacity, uint64_t* written) {
	ALPGPU_CHECK_CTX(ctx);
	if ((!h_in && n_values) || !h_blob) { return fail(ALPGPU_ERR_INVALID, "null input or blob"); }
	const uint64_t n   = (n_values + 1023) / 1024;
	const uint64_t nrg = (n + 99) / 100;
	// the blob must at least hold its fixed part before anything is produced
	if (capacity < alpgpu_blob_size(n, 0, 0)) {
		if (written) { *written = worst_case_blob<VALUE_BYTES>(n); } // nothing has been encoded yet: the size that always suffices
		return fail(ALPGPU_ERR_CAPACITY, "blob buffer too small for the column's descriptors");
	}
	uint8_t* blob     = static_cast<uint8_t*>(h_blob);
	uint8_t* blob_rg  = blob + sizeof(alpgpu_blob_header);
	uint8_t* blob_vec = blob_rg + 32ull * nrg;
	uint8_t* blob_str = blob_vec + 32ull * n;
	uint64_t pb = 0, eb = 0;
	const int rc = compress_host_piece<VALUE_BYTES>(ctx, h_in, n_values, blob_rg, blob_vec, blob_str, capacity - static_cast<uint64_t>(blob_str - blob), &pb, &eb);
	if (written) { *written = alpgpu_blob_size(n, pb, eb); }
	if (rc != ALPGPU_OK) { return rc; }
	write_blob_header(h_blob, n_values, n, pb, eb, VALUE_BYTES);
	return ALPGPU_OK;
}

// whole-rowgroup shards of a column of n vectors: (first vector, vectors) of piece i of k — the rule of alp_amd/sharding.py: rowgroup_shard
void shard_of(uint64_t n, int i, int k, uint64_t* first, uint64_t* count) {
	const uint64_t nrg = (n + 99) / 100, base = nrg / k, extra = nrg % k;
	const uint64_t first_rg = i * base + (static_cast<uint64_t>(i) < extra ? i : extra);
	const uint64_t my_rg    = base + (static_cast<uint64_t>(i) < extra ? 1 : 0);
	const uint64_t last     = (first_rg + my_rg) * 100 < n ? (first_rg + my_rg) * 100 : n;
	*first                  = first_rg * 100 < n ? first_rg * 100 : n; // an empty trailing shard (more pieces than rowgroups) starts at the column's end, not past it
	*count                  = last > *first ? last - *first : 0;
}

// N contexts (normally one per GPU of the node; several on one device work too): the column is cut into N whole-rowgroup shards, shard i
// runs through ctxs[i]'s two-stream pipeline on its own host thread — every GPU has its own PCIe link, so the shards travel side by side —
// and the pieces are joined into ONE blob by the concat_shards rule (descriptor offsets shifted by the bytes of the shards before): byte
// for byte the blob a single context writes for the column.  Rowgroup states and descriptors go straight to their places (their sizes
// are known up front); a shard's streams first land in its own region of the caller's buffer (regions in proportion to the shards'
// vector counts: the worst-case capacity always suffices) and are then moved together.
template <int VALUE_BYTES>
int compress_host_multi(alpgpu_ctx* const* ctxs, int k, const void* h_in, uint64_t n_values, void* h_blob, uint64_t capacity, uint64_t* written) {
	if (!ctxs || k < 1) { return fail(ALPGPU_ERR_INVALID, "no contexts"); }
	for (int i = 0; i < k; ++i) {
		if (!ctxs[i]) { return fail(ALPGPU_ERR_INVALID, "null context"); }
		for (int j = 0; j < i; ++j) {
			if (ctxs[j] == ctxs[i]) { return fail(ALPGPU_ERR_INVALID, "the same context twice: one pipeline per context"); }
		}
	}
	if (k == 1) { return compress_host<VALUE_BYTES>(ctxs[0], h_in, n_values, h_blob, capacity, written); }
	if ((!h_in && n_values) || !h_blob) { return fail(ALPGPU_ERR_INVALID, "null input or blob"); }
	const uint64_t n = (n_values + 1023) / 1024, nrg = (n + 99) / 100, VB = 1024ull * VALUE_BYTES;
	const uint64_t fixed = alpgpu_blob_size(n, 0, 0);
	if (capacity < fixed + 64ull * k) {
		if (written) { *written = worst_case_blob<VALUE_BYTES>(n); }
		return fail(ALPGPU_ERR_CAPACITY, "blob buffer too small for the column's descriptors");
	}
	uint8_t* blob     = static_cast<uint8_t*>(h_blob);
	uint8_t* blob_rg  = blob + sizeof(alpgpu_blob_header);
	uint8_t* blob_vec = blob_rg + 32ull * nrg;
	uint8_t* blob_str = blob_vec + 32ull * n;
	const uint64_t room = capacity - fixed;
	std::vector<uint64_t> first(k), count(k), reg_off(k + 1), pb(k, 0), eb(k, 0);
	std::vector<int>         rcs(k, ALPGPU_OK);
	std::vector<std::string> errs(k);
	for (int i = 0; i < k; ++i) { shard_of(n, i, k, &first[i], &count[i]); }
	for (int i = 0; i <= k; ++i) { // region i = [reg_off[i], reg_off[i+1]) of the stream area, 8-byte aligned, in proportion to the vectors
		const uint64_t upto = (i < k && first[i] < n) ? first[i] : n; // never past the column: a region must end inside the caller's buffer
		reg_off[i]          = n ? (static_cast<uint64_t>(static_cast<unsigned __int128>(room) * upto / n) & ~7ull) : 0;
	}
	if (reg_off[k] > room) { return fail(ALPGPU_ERR_INVALID, "internal: shard regions exceed the blob buffer"); } // (cannot happen: upto <= n)
	std::vector<std::thread> th;
	for (int i = 0; i < k; ++i) {
		th.emplace_back([&, i]() {
			if (count[i] == 0) { return; }
			const uint64_t v_end  = first[i] + count[i];
			const uint64_t values = v_end * 1024 <= n_values ? count[i] * 1024 : n_values - first[i] * 1024;
			rcs[i] = compress_host_piece<VALUE_BYTES>(ctxs[i], static_cast<const uint8_t*>(h_in) + first[i] * VB, values, blob_rg + 32ull * (first[i] / 100),
			                                          blob_vec + 32ull * first[i], blob_str + reg_off[i], reg_off[i + 1] - reg_off[i], &pb[i], &eb[i]);
			if (rcs[i] != ALPGPU_OK) { errs[i] = alpgpu_last_error(); }
		});
	}
	for (auto& t : th) { t.join(); }
	uint64_t total_p = 0, total_e = 0;
	for (int i = 0; i < k; ++i) { total_p += pb[i], total_e += eb[i]; }
	int bad = -1;
	for (int i = 0; i < k; ++i) {
		if (rcs[i] != ALPGPU_OK && (bad < 0 || rcs[bad] == ALPGPU_ERR_CAPACITY)) { bad = i; } // a failure other than "too small" is reported first
	}
	if (bad >= 0) {
		if (written) {
			*written = alpgpu_blob_size(n, total_p, total_e);
			if (rcs[bad] == ALPGPU_ERR_CAPACITY) {
				// "Too small" here means: some shard did not fit ITS region (regions are proportional to the shards' vector counts).  Every piece
				// has counted its true sizes, so the capacity that makes every region large enough is known: the largest
				// (bytes of shard i) * n / (vectors of shard i), plus the fixed part — at most the worst-case size, which always suffices.
				unsigned __int128 need_room = 0;
				for (int i = 0; i < k; ++i) {
					if (count[i] == 0) { continue; }
					const unsigned __int128 r = (static_cast<unsigned __int128>(align8(pb[i]) + align8(eb[i]) + 16) * n + count[i] - 1) / count[i];
					need_room                 = r > need_room ? r : need_room;
				}
				const uint64_t worst = worst_case_blob<VALUE_BYTES>(n);
				const unsigned __int128 want = static_cast<unsigned __int128>(fixed) + need_room + 64ull * k;
				*written = want < worst ? static_cast<uint64_t>(want) : worst;
			}
		}
		return fail(rcs[bad], errs[bad].c_str());
	}
	if (written) { *written = alpgpu_blob_size(n, total_p, total_e); }
	// exception streams aside, packed streams together (each moves towards the front: ascending order never overwrites what is still to move),
	// exception streams behind them; then the descriptors' offsets continue where the shards before ended
	std::vector<uint8_t> exc_all(total_e);
	{
		uint64_t e_at = 0;
		for (int i = 0; i < k; ++i) {
			if (eb[i]) { std::memcpy(exc_all.data() + e_at, blob_str + reg_off[i] + align8(pb[i]), eb[i]); }
			e_at += eb[i];
		}
		uint64_t p_at = 0;
		for (int i = 0; i < k; ++i) {
			if (pb[i] && reg_off[i] != p_at) {
				uint8_t *dst = blob_str + p_at, *src = blob_str + reg_off[i];
				if (dst + pb[i] <= src && pb[i] >= (64ull << 20)) { // disjoint (the usual case from the second shard on): k threads copy a slice each
					std::vector<std::thread> movers;
					for (int t = 0; t < k; ++t) {
						const uint64_t a = pb[i] * t / k, b = pb[i] * (t + 1) / k;
						movers.emplace_back([=]() { std::memcpy(dst + a, src + a, b - a); });
					}
					for (auto& m : movers) { m.join(); }
				} else {
					std::memmove(dst, src, pb[i]);
				}
			}
			p_at += pb[i];
		}
		if (total_p != align8(total_p)) { std::memset(blob_str + total_p, 0, align8(total_p) - total_p); }
		if (total_e) { std::memcpy(blob_str + align8(total_p), exc_all.data(), total_e); }
	}
	{
		uint64_t p_at = pb[0], e_at = eb[0];
		for (int i = 1; i < k; ++i) {
			for (uint64_t v = first[i]; v < first[i] + count[i]; ++v) {
				alpgpu_vector_desc d;
				std::memcpy(&d, blob_vec + 32ull * v, sizeof(d));
				d.packed_off += p_at;
				d.exc_off += e_at;
				std::memcpy(blob_vec + 32ull * v, &d, sizeof(d));
			}
			p_at += pb[i], e_at += eb[i];
		}
	}
	write_blob_header(h_blob, n_values, n, total_p, total_e, VALUE_BYTES);
	return ALPGPU_OK;
}

// vectors [v_begin, v_end) (whole rowgroups, or to the column's end) of a blob whose header has passed validate_blob_header, decoded by
// ctx's two-stream pipeline into h_out (the column's first value at h_out[0]).  Device buffers hold this range's records and bytes only;
// descriptors keep their absolute stream offsets, so the streams' device pointers are biased by the range's first offsets.
template <int VALUE_BYTES>
int decompress_host_range(alpgpu_ctx* ctx, const void* h_blob, const alpgpu_blob_header& h, void* h_out, uint64_t v_begin, uint64_t v_end) {
	ALPGPU_CHECK_CTX(ctx);
	const uint64_t n = h.n_vectors;
	if (v_begin >= v_end) { return ALPGPU_OK; }
	const uint64_t VB       = 1024ull * VALUE_BYTES;
	const uint8_t* blob_rg  = static_cast<const uint8_t*>(h_blob) + sizeof(h);
	const uint8_t* blob_vec = blob_rg + 32ull * h.n_rowgroups;
	const uint8_t* blob_p   = blob_vec + 32ull * n;
	const uint8_t* blob_e   = blob_p + align8(h.packed_bytes);
	auto desc_at = [&](uint64_t v) {
		alpgpu_vector_desc d;
		std::memcpy(&d, blob_vec + 32ull * v, sizeof(d));
		return d;
	};
	const uint64_t nr       = v_end - v_begin;
	const uint64_t rg_begin = v_begin / 100, rg_end = (v_end + 99) / 100;
	// the range's stream extents (offsets ascend with the vector index: checked chunk by chunk below)
	const uint64_t P0 = desc_at(v_begin).packed_off, E0 = desc_at(v_begin).exc_off;
	const uint64_t P1 = v_end < n ? desc_at(v_end).packed_off : h.packed_bytes;
	const uint64_t E1 = v_end < n ? desc_at(v_end).exc_off : h.exc_bytes;
	if (P0 > P1 || E0 > E1 || P1 > h.packed_bytes || E1 > h.exc_bytes) { return fail(ALPGPU_ERR_INVALID, "blob: stream offsets do not ascend with the vector index"); }
	HostPipe P;
	P.bind(ctx);
	alpgpu_column col;
	std::memset(&col, 0, sizeof(col));
	col.n_vectors = nr, col.n_rowgroups = rg_end - rg_begin, col.packed_capacity = h.packed_bytes, col.exc_capacity = h.exc_bytes;
	uint8_t *d_p = nullptr, *d_e = nullptr;
	if (int rc = P.alloc(reinterpret_cast<void**>(&col.d_rowgroups), 32ull * (rg_end - rg_begin))) { return rc; }
	if (int rc = P.alloc(reinterpret_cast<void**>(&col.d_vectors), 32ull * nr)) { return rc; }
	if (int rc = P.alloc(reinterpret_cast<void**>(&d_p), P1 - P0 + 128)) { return rc; }
	if (int rc = P.alloc(reinterpret_cast<void**>(&d_e), E1 - E0 + 64)) { return rc; }
	col.d_packed = reinterpret_cast<uint8_t*>(reinterpret_cast<uintptr_t>(d_p) - P0); // absolute offset o of the stream lives at d_p + (o - P0)
	col.d_exc    = reinterpret_cast<uint8_t*>(reinterpret_cast<uintptr_t>(d_e) - E0);
	const uint64_t chunk = nr < kHostChunkVectors ? nr : kHostChunkVectors;
	for (int k = 0; k < 2; ++k) {
		ALPGPU_HIP(hipStreamCreateWithFlags(&P.stream[k], hipStreamNonBlocking));
		if (hipMalloc(&P.d_in[k], chunk * VB) != hipSuccess) { return fail(ALPGPU_ERR_HIP, "hipMalloc (chunk buffer)", hipGetLastError()); }
	}
	// the descriptors and rowgroup states first (small), then stream by stream, chunk by chunk
	ALPGPU_HIP(hipMemcpyAsync(col.d_rowgroups, blob_rg + 32ull * rg_begin, 32ull * (rg_end - rg_begin), hipMemcpyHostToDevice, P.stream[0]));
	ALPGPU_HIP(hipMemcpyAsync(col.d_vectors, blob_vec + 32ull * v_begin, 32ull * nr, hipMemcpyHostToDevice, P.stream[0]));
	ALPGPU_HIP(hipStreamSynchronize(P.stream[0]));
	// is the output page-locked memory the device can address?
	uint8_t* out_dev = nullptr;
	{
		hipPointerAttribute_t attr;
		std::memset(&attr, 0, sizeof(attr));
		if (hipPointerGetAttributes(&attr, h_out) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer != nullptr) {
			out_dev = static_cast<uint8_t*>(attr.devicePointer);
		} else {
			(void)hipGetLastError(); // pageable memory: not an error
		}
	}
	const uint64_t n_chunks = (nr + chunk - 1) / chunk;
	// Validation (every descriptor, every exception position: a pass over ~15 % of the blob) runs ahead of the pipeline on a few host
	// threads, chunk by chunk; nothing of a chunk is launched before that chunk has passed.  A worker that finds a fault stops at it;
	// the main thread validates that chunk again itself, for the error code and text.
	std::vector<std::atomic<int>> verdict(n_chunks); // 0 pending, 1 passed, 2 failed
	for (auto& f : verdict) { f.store(0, std::memory_order_relaxed); }
	const unsigned          n_workers = n_chunks >= 4 ? 4u : 1u;
	std::vector<std::thread> workers;
	struct Joiner {
		std::vector<std::thread>& t;
		~Joiner() {
			for (auto& w : t) {
				if (w.joinable()) { w.join(); }
			}
		}
	} joiner {workers};
	// the byte ranges a chunk's decode will find in HBM: from its first vector's offsets to the next chunk's (the streams are exclusive
	// scans in vector order); every record of the chunk is checked against them
	struct Window {
		uint64_t w[4];
		bool     ok;
	};
	auto window_of = [&](uint64_t c) {
		const uint64_t v0 = v_begin + c * chunk, v1 = v_end - v0 < chunk ? v_end : v0 + chunk;
		Window W;
		W.w[0] = desc_at(v0).packed_off, W.w[2] = desc_at(v0).exc_off;
		W.w[1] = v1 < n ? desc_at(v1).packed_off : h.packed_bytes;
		W.w[3] = v1 < n ? desc_at(v1).exc_off : h.exc_bytes;
		W.ok   = P0 <= W.w[0] && W.w[0] <= W.w[1] && W.w[1] <= P1 && E0 <= W.w[2] && W.w[2] <= W.w[3] && W.w[3] <= E1;
		return W;
	};
	for (unsigned w = 0; w < n_workers; ++w) {
		workers.emplace_back([&, w]() {
			for (uint64_t c = w; c < n_chunks; c += n_workers) {
				const uint64_t b = v_begin + c * chunk, e = v_end - b < chunk ? v_end : b + chunk;
				const Window   W = window_of(c);
				const int      rc = W.ok ? validate_blob_vectors(h_blob, h, VALUE_BYTES, b, e, W.w) : ALPGPU_ERR_INVALID;
				verdict[c].store(rc == ALPGPU_OK ? 1 : 2, std::memory_order_release);
				if (rc != ALPGPU_OK) { return; }
			}
		});
	}
	for (uint64_t i = 0; i < n_chunks; ++i) {
		const int      k   = static_cast<int>(i & 1);
		const uint64_t v0  = v_begin + i * chunk;
		const uint64_t cnt = v_end - v0 < chunk ? v_end - v0 : chunk;
		int            vd;
		while ((vd = verdict[i].load(std::memory_order_acquire)) == 0) { std::this_thread::yield(); }
		const Window W = window_of(i);
		if (!W.ok) { return fail(ALPGPU_ERR_INVALID, "blob: stream offsets do not ascend with the vector index"); }
		if (vd != 1) { return validate_blob_vectors(h_blob, h, VALUE_BYTES, v0, v0 + cnt, W.w); }
		// the chunk's bytes: its ranges end where the next chunk's begin, and every record of the chunk lies inside them (validated above)
		const uint64_t p0 = W.w[0], p1 = W.w[1], e0 = W.w[2], e1 = W.w[3];
		if (p1 > p0) { ALPGPU_HIP(hipMemcpyAsync(d_p + (p0 - P0), blob_p + p0, p1 - p0, hipMemcpyHostToDevice, P.stream[k])); }
		if (e1 > e0) { ALPGPU_HIP(hipMemcpyAsync(d_e + (e0 - E0), blob_e + e0, e1 - e0, hipMemcpyHostToDevice, P.stream[k])); }
		alpgpu_column view = col; // descriptors hold absolute stream offsets: a view of whole rowgroups decodes on its own
		view.n_vectors     = cnt;
		view.n_rowgroups   = (cnt + 99) / 100;
		view.d_vectors     = col.d_vectors + (v0 - v_begin);
		view.d_rowgroups   = col.d_rowgroups + (v0 / 100 - rg_begin);
		view.packed_bytes_hint = p1 - p0, view.exc_bytes_hint = e1 - e0;
		ctx->stream        = P.stream[k];
		const uint64_t val = (v0 + cnt) * 1024 <= h.n_values ? cnt * 1024 : h.n_values - v0 * 1024;
		// Page-locked output the device can address: the decode kernel stores straight into it — the doubles cross the link as the
		// kernel's own (non-temporal) stores while the copy engine brings the next chunk's bytes up, so the two directions overlap;
		// through the chunk buffer and a copy down they take turns on this system.  (A chunk that ends inside a vector goes through the
		// buffer: the kernel writes whole vectors.)
		void* const direct = (out_dev != nullptr && val == cnt * 1024) ? static_cast<void*>(out_dev + v0 * VB) : nullptr;
		void* const target = direct ? direct : P.d_in[k];
		const int   rc     = VALUE_BYTES == 8 ? alpgpu_decode_f64(ctx, &view, static_cast<double*>(target)) : alpgpu_decode_f32(ctx, &view, static_cast<float*>(target));
		if (rc != ALPGPU_OK) { return rc; }
		if (!direct) { ALPGPU_HIP(hipMemcpyAsync(static_cast<uint8_t*>(h_out) + v0 * VB, P.d_in[k], val * VALUE_BYTES, hipMemcpyDeviceToHost, P.stream[k])); }
	}
	ALPGPU_HIP(hipStreamSynchronize(P.stream[0]));
	ALPGPU_HIP(hipStreamSynchronize(P.stream[1]));
	return ALPGPU_OK;
}

// k = 1: the whole column on ctxs[0].  k > 1: whole-rowgroup shards of the column (shard_of), shard i decoded by ctxs[i] on its own host thread
// into its part of h_out — the mirror of compress_host_multi.
template <int VALUE_BYTES>
int decompress_host_multi(alpgpu_ctx* const* ctxs, int k, const void* h_blob, uint64_t size, void* h_out, uint64_t out_capacity_values, uint64_t* n_values) {
	if (!ctxs || k < 1) { return fail(ALPGPU_ERR_INVALID, "no contexts"); }
	for (int i = 0; i < k; ++i) {
		if (!ctxs[i]) { return fail(ALPGPU_ERR_INVALID, "null context"); }
		for (int j = 0; j < i; ++j) {
			if (ctxs[j] == ctxs[i]) { return fail(ALPGPU_ERR_INVALID, "the same context twice: one pipeline per context"); }
		}
	}
	if (!h_blob) { return fail(ALPGPU_ERR_INVALID, "null blob"); }
	alpgpu_blob_header h;
	if (int rc = validate_blob_header(h_blob, size, VALUE_BYTES, h)) { return rc; } // the vectors are validated chunk by chunk, while the chunk before is in flight
	if (n_values) { *n_values = h.n_values; }
	if (h.n_values > out_capacity_values) { return fail(ALPGPU_ERR_CAPACITY, "output buffer too small (value count returned in *n_values)"); }
	if (!h_out && h.n_values) { return fail(ALPGPU_ERR_INVALID, "null output"); }
	if (h.n_vectors == 0) { return ALPGPU_OK; }
	if (k == 1) { return decompress_host_range<VALUE_BYTES>(ctxs[0], h_blob, h, h_out, 0, h.n_vectors); }
	std::vector<int>         rcs(k, ALPGPU_OK);
	std::vector<std::string> errs(k);
	std::vector<std::thread> th;
	for (int i = 0; i < k; ++i) {
		th.emplace_back([&, i]() {
			uint64_t first = 0, count = 0;
			shard_of(h.n_vectors, i, k, &first, &count);
			rcs[i] = decompress_host_range<VALUE_BYTES>(ctxs[i], h_blob, h, h_out, first, first + count);
			if (rcs[i] != ALPGPU_OK) { errs[i] = alpgpu_last_error(); }
		});
	}
	for (auto& t : th) { t.join(); }
	for (int i = 0; i < k; ++i) {
		if (rcs[i] != ALPGPU_OK) { return fail(rcs[i], errs[i].c_str()); }
	}
	return ALPGPU_OK;
}

template <int VALUE_BYTES>
int decompress_host(alpgpu_ctx* ctx, const void* h_blob, uint64_t size, void* h_out, uint64_t out_capacity_values, uint64_t* n_values) {
	if (!ctx) { return fail(ALPGPU_ERR_INVALID, "null context"); }
	alpgpu_ctx* one[1] = {ctx};
	return decompress_host_multi<VALUE_BYTES>(one, 1, h_blob, size, h_out, out_capacity_values, n_values);
}

} // namespace
} // extern "C++"

int alpgpu_compress_host_f64(alpgpu_ctx* ctx, const double* h_in, uint64_t n_values, void* h_blob, uint64_t capacity, uint64_t* written) {
	return compress_host<8>(ctx, h_in, n_values, h_blob, capacity, written);
}
int alpgpu_compress_host_f32(alpgpu_ctx* ctx, const float* h_in, uint64_t n_values, void* h_blob, uint64_t capacity, uint64_t* written) {
	return compress_host<4>(ctx, h_in, n_values, h_blob, capacity, written);
}
int alpgpu_decompress_host_f64(alpgpu_ctx* ctx, const void* h_blob, uint64_t size, double* h_out, uint64_t out_capacity_values, uint64_t* n_values) {
	return decompress_host<8>(ctx, h_blob, size, h_out, out_capacity_values, n_values);
}
int alpgpu_decompress_host_f32(alpgpu_ctx* ctx, const void* h_blob, uint64_t size, float* h_out, uint64_t out_capacity_values, uint64_t* n_values) {
	return decompress_host<4>(ctx, h_blob, size, h_out, out_capacity_values, n_values);
}
int alpgpu_compress_host_multi_f64(alpgpu_ctx* const* ctxs, int n_ctx, const double* h_in, uint64_t n_values, void* h_blob, uint64_t capacity, uint64_t* written) {
	return compress_host_multi<8>(ctxs, n_ctx, h_in, n_values, h_blob, capacity, written);
}
int alpgpu_compress_host_multi_f32(alpgpu_ctx* const* ctxs, int n_ctx, const float* h_in, uint64_t n_values, void* h_blob, uint64_t capacity, uint64_t* written) {
	return compress_host_multi<4>(ctxs, n_ctx, h_in, n_values, h_blob, capacity, written);
}
int alpgpu_decompress_host_multi_f64(alpgpu_ctx* const* ctxs, int n_ctx, const void* h_blob, uint64_t size, double* h_out, uint64_t out_capacity_values,
                                     uint64_t* n_values) {
	return decompress_host_multi<8>(ctxs, n_ctx, h_blob, size, h_out, out_capacity_values, n_values);
}
int alpgpu_decompress_host_multi_f32(alpgpu_ctx* const* ctxs, int n_ctx, const void* h_blob, uint64_t size, float* h_out, uint64_t out_capacity_values,
                                     uint64_t* n_values) {
	return decompress_host_multi<4>(ctxs, n_ctx, h_blob, size, h_out, out_capacity_values, n_values);
}

} // extern "C"
