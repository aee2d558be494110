// api_decode.hip — the store decode's LAUNCH POLICY (vectors per workgroup, residency, read-ahead, region by region), the decode entry points, the fused
// consumers and alpgpu_column_totals (which records what the policy needs) of include/alpgpu.h (see host_ctx.hpp for the map).
#include "decode_policy.hpp"
#include "host_ctx.hpp"

using alpgpu::kReadAheadBits;
using alpgpu::kReadAheadBitsExc;
using alpgpu::kReadAheadVectors;

extern "C" {

// "the column's vectors carry exceptions" as far as the decode's launch shape is concerned: about two or more per vector — unless they are
// patched in after the stores (ALPGPU_OPT_DECODE_PATCH_AFTER: an average of at most half the arm's limit, 10 bytes of record each), which
// costs a wavefront a handful of instructions: such a column behaves like one without exceptions
static bool column_decodes_with_exceptions(const alpgpu_ctx* ctx, const alpgpu_column* col) {
	const double n = static_cast<double>(col->n_vectors);
	const double e = static_cast<double>(col->exc_bytes_hint);
	if (e < 16.0 * n) { return false; }
	if (ctx->decode_patch_shape && ctx->decode_patch_max > 0 && e <= 5.0 * static_cast<double>(ctx->decode_patch_max) * n) { return false; } // (builds with a patch arm only: decode_patch_max is 0 otherwise)
	return true;
}

// The store decode of this column runs with the read-ahead (read_ahead_kernels.hip).  Asked for (1): any column long enough to be worth a second launch whose
// sizes are known (the lead is in vectors per microsecond).  Left to the library (-1, the default): columns of NARROW vectors only — there the decode is bound
// by its two dependent reads under a write-dominated stream (0.68-0.72 of the HBM peak at 2-6 bits, 0.60-0.69 with exceptions) and gains from finding them in
// the Infinity Cache: +3-14 % up to 6 bits (one vector per workgroup), with exceptions +5-24 % up to 7 bits (two per workgroup).  At 7 bits without exceptions
// it is even across six closing runs (-1 %), and an extension to 11 / 9 bits that single-column A/B runs suggested (+2-7 %, call 49) lost 3-6 % in the bench line of
// another box (call 50): the limits are where the gain is robust.  Beyond, the second stream of reads costs more than the hits save (benchmark column 0.77 -> 0.73).
// The lead that goes with a width: alpgpu_decode_f64.  tools/r05_read_ahead*.py, profiles/r05_read_ahead.txt.
// (float columns, round 6: the same kernel beside k_decode_column_f32, limits of their own — decode_policy.hpp, profiles/r06_float_decode.txt)
static bool read_ahead_for(const alpgpu_ctx* ctx, const alpgpu_column* col, int value_bytes = 8) {
	if (ctx->read_ahead == 0 || col->packed_bytes_hint == 0 || ctx->d_progress == nullptr) { return false; }
	if (ctx->read_ahead > 0) { return col->n_vectors >= 32768; }
	if (ctx->streams_serialize) { return false; } // (the two kernels cannot run side by side in this process: api_context.hip)
	return alpgpu::policy_read_ahead_auto(col->n_vectors, static_cast<double>(col->packed_bytes_hint), column_decodes_with_exceptions(ctx, col), value_bytes);
}

static int decode_variant_for(const alpgpu_ctx* ctx, const alpgpu_column* col) {
	int variant = ctx->decode_variant;
	if (ctx->decode_auto) { // pick the launch shape from the host-side size hints, if any
		const double n        = static_cast<double>(col->n_vectors);
		const bool   hinted   = col->packed_bytes_hint != 0 || col->exc_bytes_hint != 0;
		// More vectors per workgroup = more bytes in flight per CU, which is what narrow vectors lack (two dependent round trips for 8 KiB of
		// output) and what wide ones pay for.  Crossovers measured at 1-bit resolution on 1 Mi-vector columns (tools/sweep_vpw_fine.py,
		// profiles/r04_decode_floor.txt): without exceptions one vector per workgroup wins from 17 bits on (16 itself — whole KiB per vector —
		// still prefers more), with ~2 or more exceptions per vector from 21 bits on; every ALP_RD column is far beyond either.  Up to
		// kNarrowAutoBits bits FOUR vectors share a workgroup over the narrow stage (round 4).
		const bool   with_exc = column_decodes_with_exceptions(ctx, col);
		const double bits     = static_cast<double>(col->packed_bytes_hint) / (128.0 * (n > 0 ? n : 1.0));
		const bool   narrow   = bits <= (with_exc ? alpgpu::kTwoVectorsBitsExc : alpgpu::kTwoVectorsBits); // (17.5: with the residency caps below two vectors per workgroup win through 17 bits; with exceptions through 22: round 5)
		const double four_max = with_exc ? ctx->decode_four_bits_exc : ctx->decode_four_bits; // (0 = never: the four-vector shape lost at every width, it is chosen by tuning runs only)
		const bool   four     = four_max > 0.0 && bits <= four_max;
		variant               = (variant & ~5) | ((hinted && narrow) ? 0 : 1) | ((hinted && four) ? 4 : 0);
		// narrow vectors under the read-ahead: their reads hit the Infinity Cache, and ONE vector per workgroup — the shape that suffers most from the two round
		// trips (0.53 at 2-6 bits) — becomes the best one (0.75-0.80); with exceptions two per workgroup stay ahead (0.68-0.74 against 0.64-0.68)
		if (ctx->read_ahead < 0 && read_ahead_for(ctx, col) && !with_exc) { variant = (variant & ~5) | 1; }
		// (almost) nothing but 0-bit vectors — a pure stream of stores, e.g. the gov26 shape: one vector per workgroup (and six workgroups per CU, below): 0.71 -> 0.82 (call 2)
		if (hinted && bits <= alpgpu::kEmptyVectorsBits) { variant = (variant & ~5) | 1; }
	}
	// Narrow vectors WITH exceptions: the pair kernel (k_decode_pairs, both vectors' loads in flight together when both are narrow, one after the
	// other otherwise) is 1-4 % ahead of k_decode_column<2> up to 18 bits (tools/sweep_pairing.py, profiles/r04_decode_floor.txt section 4); without
	// exceptions it is not.  ALPGPU_OPT_DECODE_PAIRING overrides.
	int pairing = ctx->decode_pairing;
	if (ctx->decode_auto && pairing == 0 && ctx->decode_pairs_auto && col->packed_bytes_hint != 0) {
		const double n = static_cast<double>(col->n_vectors);
		if (column_decodes_with_exceptions(ctx, col) && static_cast<double>(col->packed_bytes_hint) <= 18.0 * 128.0 * n) { pairing = 1; }
	}
	// Residency by width (decode_kernels.hip: launch_decode_column; unused dynamic LDS): what a CU wants is a certain amount of bytes in flight, not a
	// certain number of workgroups.  One vector per workgroup: eight workgroups per CU up to 33 bits, seven up to 35, six beyond; seven for ALP_RD
	// columns.  Two vectors per workgroup: eight / seven / six workgroups by width.  ALPGPU_DECODE_PAD_LDS_KIB overrides (A/B runs; 0 = never cap).
	const int pad_env = ctx->decode_pad_kib; // ALPGPU_OPT_DECODE_RESIDENCY_PAD / ALPGPU_DECODE_PAD_LDS_KIB: -1 = by the rule below
	int pad_kib = pad_env >= 0 ? pad_env : 0;
	if (pad_env < 0 && ctx->decode_auto && pairing == 0 && col->packed_bytes_hint != 0 && col->n_vectors != 0 && !(ctx->read_ahead < 0 && read_ahead_for(ctx, col))) { // (under the read-ahead no cap helps)
		const double n        = static_cast<double>(col->n_vectors);
		const double bits     = static_cast<double>(col->packed_bytes_hint) / (128.0 * n);
		const bool   with_exc = column_decodes_with_exceptions(ctx, col);
		const bool   mostly_rd = col->alp_rd_rowgroups_hint != 0 && 2.0 * static_cast<double>(col->alp_rd_rowgroups_hint - 1) * 100.0 > n;
		// (re-measured in round 5 with the per-vector decode loops — a workgroup's stores no longer wait for one another, workgroups live shorter and a CU
		//  wants somewhat fewer of them: tools/r05_decode_resid.py, profiles/r05_decode_exceptions.txt)
		if ((variant & 5) == 1) {
			// one vector per workgroup: up to ~30 bits a 6 KiB pad (ten workgroups' worth of LDS for eight: 0.77-0.80 -> 0.79-0.81, with exceptions
			// 0.75-0.79 -> 0.78-0.82 up to 38 bits); 30-38 bits without exceptions none; from 38 bits on seven, then six workgroups per CU
			// (+5-7 %); ALP_RD columns — more arithmetic per value — seven
			if (mostly_rd) {
				pad_kib = 11;
			} else if (bits <= alpgpu::kEmptyVectorsBits) {
				pad_kib = 14; // a pure stream of stores wants few workgroups per CU, like wide vectors
			} else if (with_exc) {
				pad_kib = bits >= 46.0 ? 14 : (bits >= 38.0 ? 11 : 6);
			} else {
				pad_kib = bits >= 38.0 ? 14 : (bits >= 30.0 ? 0 : 6);
			}
		} else if ((variant & 5) == 0 && !with_exc) {
			// two vectors per workgroup, no exceptions: sixteen vectors in flight per CU up to 8 bits, fourteen (seven workgroups) beyond.  With
			// exceptions the caps lose.
			pad_kib = bits > 8.5 ? 3 : 0;
		}
	}
	// exception-heavy columns (more exceptions per vector than the 128-entry stage holds, on average): the instance with the 256-entry stage (decode_kernels.hip:
	// DecodeLdsManyExc; one vector per workgroup, non-temporal stores): 0.72 -> 0.77 on bench.py's 10 %-exceptions column (call 6)
	int many_exc = 0;
	if (ctx->decode_auto && pairing == 0 && (variant & 7) == 1 && col->n_vectors != 0 && static_cast<double>(col->exc_bytes_hint) >= 10.0 * 128.0 * static_cast<double>(col->n_vectors) &&
	    !(col->alp_rd_rowgroups_hint != 0 && 2.0 * static_cast<double>(col->alp_rd_rowgroups_hint - 1) * 100.0 > static_cast<double>(col->n_vectors))) {
		many_exc = 64;
	}
	return (variant & 7) | (pairing << 3) | many_exc | (pad_kib << 8);
}

// Float columns (round 6): vectors per workgroup — 2 (the bytes in flight of one double vector), FOUR for columns of narrow vectors whose sizes are known, what
// ALPGPU_OPT_DECODE_VECTORS_PER_WG says otherwise — and the residency pad (ALPGPU_OPT_DECODE_RESIDENCY_PAD, else none).  tools/sweep_f32_decode.py,
// profiles/r06_float_decode.txt.  Returns vectors per workgroup | pad KiB << 8 (pad 0xFF: the kernel's environment knob, for experiments).
static int decode_shape_f32(const alpgpu_ctx* ctx, const alpgpu_column* col) {
	int vpw = ctx->decode_vpw ? ctx->decode_vpw : 2;
	if (ctx->decode_vpw == 0 && ctx->decode_auto && col->packed_bytes_hint != 0 && col->n_vectors != 0) {
		const double bits = static_cast<double>(col->packed_bytes_hint) / (128.0 * static_cast<double>(col->n_vectors));
		const double four = column_decodes_with_exceptions(ctx, col) ? alpgpu::kFourVectorsBitsExcF32 : alpgpu::kFourVectorsBitsF32; // (0: never — decode_policy.hpp)
		if (four > 0.0 && bits <= four) { vpw = 4; }
		// narrow vectors without exceptions: streamed by persistent workgroups (decode_policy.hpp: policy_stream_f32)
		if (alpgpu::policy_stream_f32(col->n_vectors, static_cast<double>(col->packed_bytes_hint), column_decodes_with_exceptions(ctx, col))) { vpw = alpgpu::kStreamShapeF32; }
	}
	return vpw | ((ctx->decode_pad_kib >= 0 ? ctx->decode_pad_kib : 0xFF) << 8);
}

// ---- a launch rule that sees more than the column's averages (round 5) -------------------------------------------------------------------------
// alpgpu_column_totals and alpgpu_column_from_blob record, per segment of the column, what they record for the whole: packed bytes, exceptions, ALP_RD
// vectors.  alpgpu_decode_f64 of that column (same context, same descriptor buffer) merges adjacent segments of the same KIND — by packed width: up to the
// read-ahead's 7 bits / up to the two-vectors-per-workgroup limit / below 38 bits / beyond; with or without exceptions — into runs and decodes run by run,
// each through the rule above with the run's own sizes: a column whose first half is 6-bit vectors with exceptions and whose second half is 44-bit vectors
// (bench.py: decode_bimodal) gets two vectors per workgroup + the read-ahead for the first and one per workgroup, six workgroups per CU, for the second,
// instead of the shape of their average.  One kind, or more than kMaxRuns runs (a column that changes every few thousand vectors is served by its average): the
// whole column in one launch, as before.  ALPGPU_OPT_DECODE_SEGMENTS = 0: never.  Launch shapes only: the bytes cannot differ.
constexpr int      kMaxRuns           = 8;
struct DecodeRun {
	uint64_t v0, n, packed, exc_bytes, rd_vectors;
};

uint64_t segment_vectors_for(uint64_t n_vectors) {
	uint64_t sv = (n_vectors + kMaxSegments - 1) / kMaxSegments;
	sv          = (sv + 399) / 400 * 400;
	return sv < kSegmentMinVectors ? kSegmentMinVectors : sv;
}

static SegmentTable* segment_table_of(alpgpu_ctx* ctx, const alpgpu_column* col) {
	for (auto& t : ctx->seg_tables) {
		if (t.key != nullptr && t.key == col->d_vectors && t.d_packed == col->d_packed && t.n_vectors == col->n_vectors && t.packed_bytes == col->packed_bytes_hint &&
		    t.exc_bytes == col->exc_bytes_hint) {
			return &t;
		}
	}
	return nullptr;
}
static void segment_tables_drop(alpgpu_ctx* ctx, const alpgpu_column* col) {
	for (auto& t : ctx->seg_tables) {
		if (t.key == col->d_vectors) { t.key = nullptr; }
	}
}
void segment_table_forget(alpgpu_ctx* ctx, const alpgpu_column* col) {
	if (!ctx || !col) { return; }
	segment_tables_drop(ctx, col);
	for (auto& l : ctx->learn) { // ... and what an unhinted decode of the old content left behind (decode_unhinted)
		if (l.state != 0 && l.key == col->d_vectors) {
			if (l.state == 1) { (void)hipEventSynchronize(l.ev); } // (its copy targets the slot's host words: let it land before the slot is reused)
			l.state = 0;
		}
	}
}
// (packed_bytes / exc_bytes: the stream sizes the caller is about to write into the column's hints)
SegmentTable* segment_table_new(alpgpu_ctx* ctx, const alpgpu_column* col, uint64_t packed_bytes, uint64_t exc_bytes) {
	segment_tables_drop(ctx, col);
	SegmentTable* t = nullptr;
	for (auto& c : ctx->seg_tables) {
		if (c.key == nullptr) { t = &c; break; }
	}
	if (!t) {
		t             = &ctx->seg_tables[ctx->seg_next];
		ctx->seg_next = (ctx->seg_next + 1) % 4;
	}
	t->key         = col->d_vectors;
	t->d_packed    = col->d_packed;
	t->n_vectors   = col->n_vectors;
	t->packed_bytes = packed_bytes, t->exc_bytes = exc_bytes;
	t->seg_vectors = segment_vectors_for(col->n_vectors);
	t->n_seg       = static_cast<uint32_t>((col->n_vectors + t->seg_vectors - 1) / t->seg_vectors);
	return t;
}

// the kind of a stretch of vectors, from its sums (see above); float columns: up to the read-ahead's limit / up to the four-vectors limit / beyond
static int stretch_kind(const alpgpu_ctx* ctx, uint64_t n, uint64_t packed, uint64_t exc_bytes, int value_bytes) {
	alpgpu_column v {};
	v.n_vectors = n, v.packed_bytes_hint = packed ? packed : 1, v.exc_bytes_hint = exc_bytes;
	const bool   with_exc = column_decodes_with_exceptions(ctx, &v);
	const double bits     = static_cast<double>(packed) / (128.0 * static_cast<double>(n));
	int          band;
	if (value_bytes == 8) {
		band = bits <= (with_exc ? kReadAheadBitsExc : kReadAheadBits) ? 0 : (bits <= (with_exc ? alpgpu::kTwoVectorsBitsExc : alpgpu::kTwoVectorsBits) ? 1 : (bits < 38.0 ? 2 : 3));
	} else {
		band = bits <= (with_exc ? alpgpu::kReadAheadBitsExcF32 : alpgpu::kReadAheadBitsF32) ? 0 : (bits <= (with_exc ? alpgpu::kFourVectorsBitsExcF32 : alpgpu::kFourVectorsBitsF32) ? 1 : 2);
	}
	return 2 * band + (with_exc ? 1 : 0);
}

// runs[0 .. return) cover the column; 0 = no plan (decode the column whole)
static int plan_decode_runs(alpgpu_ctx* ctx, const alpgpu_column* col, DecodeRun* runs, int value_bytes = 8) {
	if (!ctx->decode_segments || !ctx->decode_auto || ctx->decode_pad_kib >= 0 || ctx->decode_pairing != 0) { return 0; } // (a forced shape is a forced shape)
	if (value_bytes == 4 && ctx->decode_vpw != 0) { return 0; }
	const SegmentTable* t = segment_table_of(ctx, col);
	if (!t || t->n_seg < 2) { return 0; }
	int n_runs = 0, kind = -1;
	for (uint32_t s = 0; s < t->n_seg; ++s) {
		const uint64_t v0 = s * t->seg_vectors;
		const uint64_t n  = v0 + t->seg_vectors < t->n_vectors ? t->seg_vectors : t->n_vectors - v0;
		const uint64_t eb = (value_bytes + 2ull) * t->exc_cnt[s]; // (ALP: the value + a 2-byte position; ALP_RD records are smaller and their vectors wide anyway)
		const int      k  = stretch_kind(ctx, n, t->packed[s], eb, value_bytes);
		if (k != kind) {
			if (n_runs == kMaxRuns) { return 0; }
			runs[n_runs++] = DecodeRun {v0, 0, 0, 0, 0};
			kind           = k;
		}
		DecodeRun& r = runs[n_runs - 1];
		r.n += n, r.packed += t->packed[s], r.exc_bytes += eb, r.rd_vectors += t->rd_vectors[s];
	}
	return n_runs >= 2 ? n_runs : 0;
}

// the view of a run: descriptors hold absolute stream offsets, so a stretch of whole rowgroups decodes on its own
static alpgpu_column run_view(const alpgpu_column* col, const DecodeRun& r) {
	alpgpu_column v   = *col;
	v.n_vectors       = r.n;
	v.n_rowgroups     = (r.n + 99) / 100;
	v.d_vectors       = col->d_vectors + r.v0;
	v.d_rowgroups     = col->d_rowgroups + r.v0 / 100;
	v.packed_bytes_hint = r.packed ? r.packed : 1;
	v.exc_bytes_hint    = r.exc_bytes;
	v.alp_rd_rowgroups_hint = 1 + r.rd_vectors / 100;
	return v;
}

// what alpgpu_decode_f64 / _f32 would launch for this column right now (option + size hints): vectors per decode workgroup
int alpgpu_decode_vectors_per_wg(alpgpu_ctx* ctx, const alpgpu_column* col, int is_f32) {
	if (!ctx || !col) { return fail(ALPGPU_ERR_INVALID, "null context or column"); }
	if (is_f32) { return decode_shape_f32(ctx, col) & 0xFF; }
	const int variant = decode_variant_for(ctx, col);
	if ((variant >> 3) & 3) { return 2; } // (the pair kernel: two vectors per workgroup, run together or one after the other)
	return (variant & 4) ? 4 : ((variant & 1) ? 1 : 2);
}

// ... and whether it would start the read-ahead beside the decode kernel (ALPGPU_OPT_DECODE_READ_AHEAD): 1 / 0; negative on bad arguments
int alpgpu_decode_reads_ahead(alpgpu_ctx* ctx, const alpgpu_column* col, int is_f32) {
	if (!ctx || !col) { return fail(ALPGPU_ERR_INVALID, "null context or column"); }
	if (is_f32 && (decode_shape_f32(ctx, col) & 0xFF) >= 16) { return 0; } // (streamed by persistent workgroups: no read-ahead beside it)
	return read_ahead_for(ctx, col, is_f32 ? 4 : 8) ? 1 : 0;
}

// measurement aid: what alpgpu_decode_sum_f64 costs with its unpack arithmetic left out (decode_kernels.hip: kSinkProbe)
int alpgpu_debug_decode_probe_f64(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_out) {
	ALPGPU_CHECK_CTX(ctx);
	if (!col || (!d_out && col->n_vectors)) { return fail(ALPGPU_ERR_INVALID, "null column or output"); }
	if (col->n_vectors == 0) { return ALPGPU_OK; }
	if (!col->d_vectors || !col->d_rowgroups) { return fail(ALPGPU_ERR_INVALID, "column has no descriptors"); }
	if (alpgpu::launch_decode_probe(ctx->stream, col, d_out) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "decode probe launch failed", hipGetLastError()); }
	return ALPGPU_OK;
}
// which kernel computes the per-vector sums (ALPGPU_OPT_CONSUMER_PIPELINED)
// 0 (default) and 2: one wavefront per vector (k_sink_direct) whatever the column holds — ahead of the staged four-wavefront kernel on ALP
// columns (0.81 against 0.91 ms per 1 Mi vectors of the benchmark column) and, since its ALP_RD arm stopped spilling, on ALP_RD columns too
// (1.41 against 1.51 ms); 1: the persistent LDS-ring kernel; 3: the staged four-wavefront kernel.  (alp_rd_rowgroups_hint, which chose
// between the two while the ALP_RD arm spilled, is still kept up to date in the column for callers that want to know.)
static bool use_direct_sink(const alpgpu_ctx* ctx, const alpgpu_column*) { return ctx->pipelined_consumer == 0 || ctx->pipelined_consumer == 2; }
static int sum_launch(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_sums) {
	if (ctx->pipelined_consumer == 1) { return alpgpu::launch_consume_sum(ctx->stream, col, d_sums, ctx->n_cus); }
	if (use_direct_sink(ctx, col)) { return alpgpu::launch_sink_direct(ctx->stream, col, 0.0, 0.0, d_sums, false); }
	return alpgpu::launch_decode_sum(ctx->stream, col, d_sums, 2);
}

// Float columns: the one-wavefront kernel whatever the column holds (0.99 against 1.08 ms for the staged kernel on the decimal column, 1.04
// against 1.06 on an all-ALP_RD one); options 1 and 3 select the staged kernel (there is no ring kernel).
static bool use_direct_sink_f32(const alpgpu_ctx* ctx, const alpgpu_column*) { return ctx->pipelined_consumer == 0 || ctx->pipelined_consumer == 2; }
static int  sum_launch_f32(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_sums) {
	if (use_direct_sink_f32(ctx, col)) { return alpgpu::launch_sink_direct_f32(ctx->stream, col, 0.0f, 0.0f, d_sums, false); }
	return alpgpu::launch_decode_sum_f32(ctx->stream, col, d_sums);
}

int alpgpu_decode_sum_f64(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_sums) {
	ALPGPU_CHECK_CTX(ctx);
	if (!col || (!d_sums && col->n_vectors)) { return fail(ALPGPU_ERR_INVALID, "null column or output"); }
	if (col->n_vectors == 0) { return ALPGPU_OK; }
	if (!col->d_vectors || !col->d_rowgroups) { return fail(ALPGPU_ERR_INVALID, "column has no descriptors"); }
	const int rc = sum_launch(ctx, col, d_sums);
	if (rc != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "decode-sum launch failed", hipGetLastError()); }
	return ALPGPU_OK;
}

// The whole column's total (the reference's consumer keeps ONE accumulator across vectors, q1.cpp:91-100): per-vector sums into the
// context's workspace, then the documented tree over them.  Everything stays on the stream; *d_total is device memory.
static int column_sum(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_total, bool f32) {
	ALPGPU_CHECK_CTX(ctx);
	if (!col || !d_total) { return fail(ALPGPU_ERR_INVALID, "null column or output"); }
	const uint64_t n = col->n_vectors;
	if (n == 0) {
		ALPGPU_HIP(hipMemsetAsync(d_total, 0, sizeof(double), ctx->stream));
		return ALPGPU_OK;
	}
	if (!col->d_vectors || !col->d_rowgroups) { return fail(ALPGPU_ERR_INVALID, "column has no descriptors"); }
	const uint64_t l1 = (n + 1023) / 1024;
	if (int rc = ensure_workspace(ctx, 8ull * (n + 2 * l1) + 64)) { return rc; }
	double* sums = static_cast<double*>(ctx->workspace);
	int     rc   = f32 ? sum_launch_f32(ctx, col, sums)
	                   : sum_launch(ctx, col, sums);
	if (rc == ALPGPU_OK) { rc = alpgpu::launch_tree_sum(ctx->stream, sums, n, sums + n, d_total); }
	if (rc != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "column-sum launch failed", hipGetLastError()); }
	return workspace_used(ctx);
}
int alpgpu_column_sum_f64(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_total) { return column_sum(ctx, col, d_total, false); }
int alpgpu_column_sum_f32(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_total) { return column_sum(ctx, col, d_total, true); }
// the same tree over any device array of doubles (e.g. the per-vector sums a caller already has)
int alpgpu_tree_sum_f64(alpgpu_ctx* ctx, const double* d_in, uint64_t n, double* d_total) {
	ALPGPU_CHECK_CTX(ctx);
	if ((!d_in && n) || !d_total) { return fail(ALPGPU_ERR_INVALID, "null input or output"); }
	const uint64_t l1 = (n + 1023) / 1024;
	if (int rc = ensure_workspace(ctx, 16ull * l1 + 64)) { return rc; }
	if (alpgpu::launch_tree_sum(ctx->stream, d_in, n, static_cast<double*>(ctx->workspace), d_total) != ALPGPU_OK) {
		return fail(ALPGPU_ERR_HIP, "tree-sum launch failed", hipGetLastError());
	}
	return workspace_used(ctx);
}

int alpgpu_decode_count_range_f64(alpgpu_ctx* ctx, const alpgpu_column* col, double lo, double hi, uint32_t* d_counts) {
	ALPGPU_CHECK_CTX(ctx);
	if (!col || (!d_counts && col->n_vectors)) { return fail(ALPGPU_ERR_INVALID, "null column or output"); }
	if (col->n_vectors == 0) { return ALPGPU_OK; }
	if (!col->d_vectors || !col->d_rowgroups) { return fail(ALPGPU_ERR_INVALID, "column has no descriptors"); }
	const int rc = ctx->pipelined_consumer == 1 ? alpgpu::launch_consume_count_range(ctx->stream, col, lo, hi, d_counts, ctx->n_cus)
	               : use_direct_sink(ctx, col)  ? alpgpu::launch_sink_direct(ctx->stream, col, lo, hi, d_counts, true)
	                                            : alpgpu::launch_decode_count_range(ctx->stream, col, lo, hi, d_counts);
	if (rc != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "decode-count launch failed", hipGetLastError()); }
	return ALPGPU_OK;
}

extern "C++" {
// the next tag of the context's progress word (its top 24 bits: a stale value of an earlier launch reads as "not started")
static uint64_t next_progress_tag(alpgpu_ctx* ctx) {
	ctx->progress_gen = (ctx->progress_gen + 1) & 0xFFFFFFull;
	if (ctx->progress_gen == 0) { ctx->progress_gen = 1; }
	return ctx->progress_gen << 40;
}

// one launch of the store decode over a column or a run of it (+ the read-ahead beside it where the rule wants one); VB = 8 (double) or 4 (float)
template <int VB>
static int decode_one(alpgpu_ctx* ctx, const alpgpu_column* col, void* d_out) {
	// The read-ahead (read_ahead_kernels.hip): a few persistent workgroups on the context's second stream pull the column's streams into the Infinity
	// Cache a bounded distance ahead of the decode kernel, which tells them where it is.  Started first so that it is ahead from the first workgroup on.
	// (a float column streamed by persistent workgroups prefetches for itself; the read-ahead beside it changed nothing: call 14)
#ifdef ALPGPU_STREAM_WITH_READ_AHEAD // measurement build: the read-ahead beside the streamed float decode too
	const bool ahead = read_ahead_for(ctx, col, VB);
#else
	const bool ahead = read_ahead_for(ctx, col, VB) && !(VB == 4 && (decode_shape_f32(ctx, col) & 0xFF) >= 16);
#endif
	uint64_t   tag   = 0;
	if (ahead) {
		tag = next_progress_tag(ctx);
		ALPGPU_HIP(hipEventRecord(ctx->ev_fork, ctx->stream)); // (the read-ahead may start where the decode may)
	}
	// The decode kernel is enqueued FIRST (round 6; until round 5 the read-ahead was): whatever delays the host between the two launches — the lazy load of a code
	// object on a process's first decode takes milliseconds — then costs the read-ahead a late start instead of its patience (a read-ahead that waits for a decode
	// that is not even enqueued leaves, and that decode then runs without one).  The side stream has the higher priority: its few workgroups are placed at once.
	int rc;
	if constexpr (VB == 8) {
		rc = alpgpu::launch_decode_column(ctx->stream, col, static_cast<double*>(d_out), decode_variant_for(ctx, col), ctx->n_cus, static_cast<uint32_t>(ctx->decode_patch_max), ahead ? ctx->d_progress : nullptr, tag);
	} else {
		const int shape = decode_shape_f32(ctx, col);
		if ((shape & 0xFF) >= 16) { // the column streamed by persistent workgroups (decode_stream_f32_kernels.hip): no read-ahead beside it, nothing to report
			rc = alpgpu::launch_decode_stream_f32(ctx->stream, col, static_cast<float*>(d_out), shape & 0xFF, ctx->n_cus, ahead ? ctx->d_progress : nullptr, tag);
		} else {
			rc = alpgpu::launch_decode_column_f32(ctx->stream, col, static_cast<float*>(d_out), shape & 0xFF, (ctx->decode_variant & 2) != 0, (shape >> 8) == 0xFF ? -1 : (shape >> 8), ahead ? ctx->d_progress : nullptr, tag);
		}
	}
	if (ahead) {
		const alpgpu::ReadAheadPace pace = alpgpu::policy_read_ahead_pace(static_cast<double>(col->n_vectors), static_cast<double>(col->packed_bytes_hint),
		                                                                  static_cast<double>(col->exc_bytes_hint), VB, ctx->read_ahead_us); // (decode_policy.hpp: the lead is a time)
		ALPGPU_HIP(hipStreamWaitEvent(ctx->init_stream, ctx->ev_fork, 0));
		if (alpgpu::launch_read_ahead(ctx->init_stream, col, VB, ctx->d_progress, tag, pace.lead_min, pace.lead_max, pace.ps_per_vector, ctx->wall_tick_ps, static_cast<uint32_t>(ctx->read_ahead_bits), ctx->read_ahead_grid) != ALPGPU_OK) {
			return fail(ALPGPU_ERR_HIP, "read-ahead launch failed", hipGetLastError());
		}
		ALPGPU_HIP(hipEventRecord(ctx->ev_join, ctx->init_stream));
	}
	if (ahead) { ALPGPU_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0)); } // (the read-ahead leaves on its own once its last batch is in reach or the decode never shows up)
	if (rc != ALPGPU_OK) { return fail(rc, "decode launch failed", hipGetLastError()); }
	return ALPGPU_OK;
}

// ---- a column whose sizes the host does not know (round 6) ------------------------------------------------------------------------
// alpgpu_encode_* leaves the stream sizes in device memory; the host learns them from alpgpu_column_totals, a synchronisation.  A caller that encodes and
// decodes without it used to get one vector per workgroup and no read-ahead (0.50 on 4-bit vectors where the hinted decode reaches 0.72-0.76).  Now, for columns of
// at least kUnhintedMinVectors vectors: the per-segment sums are taken on the stream (k_segment_sums, 17 us per 1 Mi vectors), k_unhinted_plan evaluates the rule on
// the device (decode_policy.hpp) and the read-ahead beside the decode takes its lead and pace — or "not for this column" — from the plan's words.  The sums also
// travel to page-locked host memory behind an event nobody waits for: the NEXT decode of the same column (same descriptor and packed buffers, same length, not
// encoded again in between) finds them there and is planned on the host like a hinted one, launch shape and regions included (call 1b: 1.00-1.05 of the hinted
// figure from the second decode on).  The launch shape of the FIRST decode: see decode_unhinted.  ALPGPU_OPT_DECODE_UNHINTED = 0: the old behaviour.
constexpr uint64_t kUnhintedMinVectors = 65536;

static LearnSlot* learn_slot_of(alpgpu_ctx* ctx, const alpgpu_column* col, int value_bytes) {
	for (auto& l : ctx->learn) {
		if (l.state != 0 && l.key == col->d_vectors && l.d_packed == col->d_packed && l.n_vectors == col->n_vectors && l.value_bytes == value_bytes) { return &l; }
	}
	return nullptr;
}

// the column with the sizes an earlier unhinted decode of it left behind, if they have arrived; false: nothing known (yet)
static bool learned_hints(alpgpu_ctx* ctx, const alpgpu_column* col, int value_bytes, alpgpu_column* hinted) {
	LearnSlot* l = learn_slot_of(ctx, col, value_bytes);
	if (!l) { return false; }
	if (l->state == 1) {
		if (hipEventQuery(l->ev) != hipSuccess) {
			(void)hipGetLastError(); // not ready: not an error
			return false;
		}
		const uint64_t* sums = ctx->h_learn + static_cast<size_t>(l - ctx->learn) * 3 * kMaxSegments;
		l->packed = l->exceptions = l->rd_vectors = 0;
		for (uint32_t i = 0; i < l->n_seg; ++i) { l->packed += sums[3 * i], l->exceptions += sums[3 * i + 1], l->rd_vectors += sums[3 * i + 2]; }
		l->state = 2;
		if (l->n_seg >= 2) { // the decode's launch plan (plan_decode_runs), as alpgpu_column_totals would have left it
			alpgpu_column c = *col;
			SegmentTable* t = segment_table_new(ctx, &c, l->packed ? l->packed : 1, (value_bytes + 2ull) * l->exceptions);
			for (uint32_t i = 0; i < t->n_seg && i < l->n_seg; ++i) { t->packed[i] = sums[3 * i], t->exc_cnt[i] = sums[3 * i + 1], t->rd_vectors[i] = sums[3 * i + 2]; }
		}
	}
	*hinted = *col;
	hinted->packed_bytes_hint     = l->packed ? l->packed : 1; // (a column of nothing but 0-bit vectors: "known, and narrow")
	hinted->exc_bytes_hint        = (value_bytes + 2ull) * l->exceptions;
	hinted->alp_rd_rowgroups_hint = 1 + l->rd_vectors / 100;
	return true;
}

template <int VB>
static int decode_unhinted(alpgpu_ctx* ctx, const alpgpu_column* col, void* d_out) {
	uint64_t* words = ctx->d_progress;
	const uint64_t seg_vectors = segment_vectors_for(col->n_vectors);
	const uint32_t n_seg       = static_cast<uint32_t>((col->n_vectors + seg_vectors - 1) / seg_vectors);
	if (alpgpu::launch_segment_sums(ctx->stream, col, seg_vectors, n_seg, words + alpgpu::kCtxWordSegments) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "segment sums launch failed", hipGetLastError()); }
	if (alpgpu::launch_unhinted_plan(ctx->stream, words, n_seg, col->n_vectors, VB, (ctx->read_ahead < 0 && ctx->streams_serialize) ? 0 : ctx->read_ahead, ctx->read_ahead_us, static_cast<uint32_t>(ctx->read_ahead_bits)) != ALPGPU_OK) {
		return fail(ALPGPU_ERR_HIP, "plan launch failed", hipGetLastError());
	}
	const uint64_t tag        = next_progress_tag(ctx);
	const bool     with_ahead = ctx->read_ahead > 0 || (ctx->read_ahead < 0 && !ctx->streams_serialize);
	if (with_ahead) { ALPGPU_HIP(hipEventRecord(ctx->ev_fork, ctx->stream)); }
	const bool plain = (ctx->decode_variant & 2) != 0;
	int        rc    = ALPGPU_OK;
	// Which launch.  A closed candidate is not free: the dispatcher hands out ~5.5 workgroups per nanosecond, so 1 Mi empty workgroups cost 0.19 ms — 10-17 % of a
	// decode (call 1b: first unhinted decode 0.59-0.68 of peak against 0.72-0.81 hinted with all three candidates launched).  Default (1) therefore: ONE launch in the
	// shape a column without hints always got (double: one vector per workgroup, 6 KiB pad; float: two), un-gated, and the device-side plan steers the read-ahead only —
	// which is what narrow columns gain most from — while the sizes travel to the host for the next decode.  2: every candidate, gated (the measured alternative).
	if constexpr (VB == 8) { // decode_policy.hpp: 1 = one vector per workgroup + 6 KiB, 2 = two per workgroup, 3 = one per workgroup + 11 KiB
		if (ctx->decode_unhinted == 2) {
			const int variants[alpgpu::kUnhintedShapesF64] = {1 | (6 << 8), 0, 1 | (11 << 8)};
			for (int c = 0; c < alpgpu::kUnhintedShapesF64 && rc == ALPGPU_OK; ++c) {
				rc = alpgpu::launch_decode_column(ctx->stream, col, static_cast<double*>(d_out), variants[c] | (plain ? 2 : 0), ctx->n_cus, 0u, words, tag, static_cast<uint32_t>(c + 1));
			}
		} else {
			rc = alpgpu::launch_decode_column(ctx->stream, col, static_cast<double*>(d_out), 1 | (6 << 8) | (plain ? 2 : 0), ctx->n_cus, 0u, words, tag, 0u);
		}
	} else {
		rc = alpgpu::launch_decode_column_f32(ctx->stream, col, static_cast<float*>(d_out), 2, plain, 0, words, tag, 0u);
	}
	if (with_ahead) { // behind the decode's launch (decode_one says why); the kernel leaves at once when the plan says "no read-ahead for this column"
		ALPGPU_HIP(hipStreamWaitEvent(ctx->init_stream, ctx->ev_fork, 0));
		if (alpgpu::launch_read_ahead(ctx->init_stream, col, VB, words, tag, 0, 0, 0, ctx->wall_tick_ps, 0, ctx->read_ahead_grid, true) != ALPGPU_OK) {
			return fail(ALPGPU_ERR_HIP, "read-ahead launch failed", hipGetLastError());
		}
		ALPGPU_HIP(hipEventRecord(ctx->ev_join, ctx->init_stream));
	}
	// the sums, for the next decode of this column: to page-locked memory behind an event that is only ever queried (enqueued BEHIND the decode: the copy delays nothing)
	if (ctx->h_learn != nullptr) {
		LearnSlot* l = learn_slot_of(ctx, col, VB);
		if (!l) {
			l               = &ctx->learn[ctx->learn_next];
			ctx->learn_next = (ctx->learn_next + 1) % kLearnSlots;
		}
		// (a slot whose copy is still in flight — the same column decoded again before its sizes arrived, or a fifth unhinted column within microseconds — is left
		//  alone: no host synchronisation here either; the earlier copy lands, or this column simply stays unhinted a little longer)
		const bool busy = l->state == 1 && hipEventQuery(l->ev) != hipSuccess;
		if (busy) {
			(void)hipGetLastError();
		} else {
			l->key = col->d_vectors, l->d_packed = col->d_packed, l->n_vectors = col->n_vectors, l->value_bytes = VB, l->n_seg = n_seg;
			l->state = 0;
			if (hipMemcpyAsync(ctx->h_learn + static_cast<size_t>(l - ctx->learn) * 3 * kMaxSegments, words + alpgpu::kCtxWordSegments, 24ull * n_seg, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
			    hipEventRecord(l->ev, ctx->stream) == hipSuccess) {
				l->state = 1;
			} else {
				(void)hipGetLastError(); // best effort: the column simply stays unhinted
			}
		}
	}
	if (with_ahead) { ALPGPU_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0)); }
	if (rc != ALPGPU_OK) { return fail(rc, "decode launch failed", hipGetLastError()); }
	return ALPGPU_OK;
}

template <int VB>
static int decode_column(alpgpu_ctx* ctx, const alpgpu_column* col, void* d_out) {
	ALPGPU_CHECK_CTX(ctx);
	if (!col || (!d_out && col->n_vectors)) { return fail(ALPGPU_ERR_INVALID, "null column or output"); }
	if (col->n_vectors == 0) { return ALPGPU_OK; }
	if (!col->d_vectors || !col->d_rowgroups) { return fail(ALPGPU_ERR_INVALID, "column has no descriptors"); }
	alpgpu_column hinted;
	const bool    unhinted = col->packed_bytes_hint == 0 && col->exc_bytes_hint == 0;
	const bool    free_shape = ctx->decode_auto && ctx->decode_pairing == 0 && ctx->decode_pad_kib < 0 && (VB == 8 || ctx->decode_vpw == 0);
	if (unhinted && ctx->decode_unhinted && free_shape && ctx->d_progress != nullptr && col->n_vectors >= kUnhintedMinVectors) {
		if (learned_hints(ctx, col, VB, &hinted)) {
			col = &hinted; // an earlier decode of this column took its sizes: planned on the host from here on
		} else {
			return decode_unhinted<VB>(ctx, col, d_out);
		}
	}
	DecodeRun runs[kMaxRuns];
	const int n_runs = plan_decode_runs(ctx, col, runs, VB);
	if (n_runs == 0) { return decode_one<VB>(ctx, col, d_out); }
	for (int i = 0; i < n_runs; ++i) { // regions of different kinds, each with its own launch shape (plan_decode_runs)
		const alpgpu_column view = run_view(col, runs[i]);
		if (const int rc = decode_one<VB>(ctx, &view, static_cast<uint8_t*>(d_out) + runs[i].v0 * 1024 * VB)) { return rc; }
	}
	return ALPGPU_OK;
}
} // extern "C++"

int alpgpu_decode_f64(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_out) { return decode_column<8>(ctx, col, d_out); }
int alpgpu_decode_f32(alpgpu_ctx* ctx, const alpgpu_column* col, float* d_out) { return decode_column<4>(ctx, col, d_out); }

// debug aids (tests): batches of 64 vectors the context's read-aheads have read so far; the plan words of its last unhinted decode
// (out[0] shape, [1] lead_min | lead_max << 32, [2] ps per vector | max bits << 32, [3..5] packed bytes, exceptions, ALP_RD vectors).  Both wait for the stream.
int alpgpu_debug_read_ahead_batches(alpgpu_ctx* ctx, uint64_t* batches) {
	ALPGPU_CHECK_CTX(ctx);
	if (!batches || !ctx->d_progress) { return fail(ALPGPU_ERR_INVALID, "null output"); }
	ALPGPU_HIP(hipStreamSynchronize(ctx->stream));
	ALPGPU_HIP(hipStreamSynchronize(ctx->init_stream));
	ALPGPU_HIP(hipMemcpy(batches, ctx->d_progress + alpgpu::kCtxWordBatches, 8, hipMemcpyDeviceToHost));
	return ALPGPU_OK;
}
int alpgpu_debug_forget_column(alpgpu_ctx* ctx, const alpgpu_column* col) {
	ALPGPU_CHECK_CTX(ctx);
	if (!col) { return fail(ALPGPU_ERR_INVALID, "null column"); }
	segment_table_forget(ctx, col);
	return ALPGPU_OK;
}
int alpgpu_debug_unhinted_plan(alpgpu_ctx* ctx, uint64_t* out6) {
	ALPGPU_CHECK_CTX(ctx);
	if (!out6 || !ctx->d_progress) { return fail(ALPGPU_ERR_INVALID, "null output"); }
	ALPGPU_HIP(hipStreamSynchronize(ctx->stream));
	uint64_t w[16];
	ALPGPU_HIP(hipMemcpy(w, ctx->d_progress, sizeof(w), hipMemcpyDeviceToHost));
	out6[0] = w[alpgpu::kCtxWordShape], out6[1] = w[alpgpu::kCtxWordLead], out6[2] = w[alpgpu::kCtxWordPace];
	out6[3] = w[alpgpu::kCtxWordTotals], out6[4] = w[alpgpu::kCtxWordTotals + 1], out6[5] = w[alpgpu::kCtxWordTotals + 2];
	return ALPGPU_OK;
}

// how many launches alpgpu_decode_f64 would make for this column now: 1, or the number of runs of plan_decode_runs; negative on bad arguments
int alpgpu_decode_runs(alpgpu_ctx* ctx, const alpgpu_column* col) {
	if (!ctx || !col) { return fail(ALPGPU_ERR_INVALID, "null context or column"); }
	DecodeRun runs[kMaxRuns];
	const int n_runs = plan_decode_runs(ctx, col, runs, 8);
	return n_runs == 0 ? 1 : n_runs;
}
int alpgpu_decode_runs_f32(alpgpu_ctx* ctx, const alpgpu_column* col) {
	if (!ctx || !col) { return fail(ALPGPU_ERR_INVALID, "null context or column"); }
	DecodeRun runs[kMaxRuns];
	const int n_runs = plan_decode_runs(ctx, col, runs, 4);
	return n_runs == 0 ? 1 : n_runs;
}
int alpgpu_column_totals(alpgpu_ctx* ctx, alpgpu_column* col, uint64_t* packed_bytes, uint64_t* exc_bytes, int* overflow) {
	ALPGPU_CHECK_CTX(ctx);
	if (!col) { return fail(ALPGPU_ERR_INVALID, "null column"); }
	uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	if (col->d_totals) {
		const bool count_rd = col->d_rowgroups != nullptr && col->n_rowgroups != 0;
		if (count_rd && alpgpu::launch_count_rd_rowgroups(ctx->stream, col, col->d_totals + 7) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "rowgroup count launch failed", hipGetLastError()); }
		// per-segment sums for the decode's launch plan (plan_decode_runs): columns long enough to have two segments
		// (the table is filled first and gets its key last: an early return leaves no half-made entry that a later decode could match — ADVICE round 5)
		uint64_t seg_sums[3 * kMaxSegments];
		uint64_t seg_vectors = 0;
		uint32_t n_seg       = 0;
		segment_table_forget(ctx, col);
		if (col->d_vectors && ctx->d_progress && col->n_vectors >= 2 * kSegmentMinVectors) {
			seg_vectors = segment_vectors_for(col->n_vectors);
			n_seg       = static_cast<uint32_t>((col->n_vectors + seg_vectors - 1) / seg_vectors);
			if (alpgpu::launch_segment_sums(ctx->stream, col, seg_vectors, n_seg, ctx->d_progress + alpgpu::kCtxWordSegments) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "segment sums launch failed", hipGetLastError()); }
			ALPGPU_HIP(hipMemcpyAsync(seg_sums, ctx->d_progress + alpgpu::kCtxWordSegments, 24ull * n_seg, hipMemcpyDeviceToHost, ctx->stream));
		}
		ALPGPU_HIP(hipMemcpyAsync(t, col->d_totals, sizeof(t), hipMemcpyDeviceToHost, ctx->stream));
		ALPGPU_HIP(hipStreamSynchronize(ctx->stream));
		if (n_seg != 0 && !t[2] && !t[3]) { // (an overflowed or unrecovered column: no plans)
			SegmentTable* seg = segment_table_new(ctx, col, t[0], t[1]);
			for (uint32_t i = 0; i < seg->n_seg && i < n_seg; ++i) { seg->packed[i] = seg_sums[3 * i], seg->exc_cnt[i] = seg_sums[3 * i + 1], seg->rd_vectors[i] = seg_sums[3 * i + 2]; }
		}
		col->alp_rd_rowgroups_hint = count_rd ? 1 + t[7] : (col->n_vectors == 0 ? 1 : 0);
	} else if (col->n_vectors != 0) {
		return fail(ALPGPU_ERR_INVALID, "column without d_totals");
	}
	if (packed_bytes) { *packed_bytes = t[0]; }
	if (exc_bytes) { *exc_bytes = t[1]; }
	if (overflow) { *overflow = static_cast<int>(t[2]); }
	col->packed_bytes_hint = t[0];
	col->exc_bytes_hint    = t[1];
	// cannot happen through alpgpu_encode_*: the recovery route clears the flag (alpgpu_encode_vectors_f64)
	if (t[3]) { return fail(ALPGPU_ERR_HIP, "single-pass encode stalled in its offset look-back and was not recovered"); }
	return t[2] ? fail(ALPGPU_ERR_CAPACITY, "an output stream overflowed its capacity") : ALPGPU_OK;
}
int alpgpu_decode_sum_f32(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_sums) {
	ALPGPU_CHECK_CTX(ctx);
	if (!col || (!d_sums && col->n_vectors)) { return fail(ALPGPU_ERR_INVALID, "null column or output"); }
	if (col->n_vectors == 0) { return ALPGPU_OK; }
	if (!col->d_vectors || !col->d_rowgroups) { return fail(ALPGPU_ERR_INVALID, "column has no descriptors"); }
	if (sum_launch_f32(ctx, col, d_sums) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "decode-sum launch failed", hipGetLastError()); }
	return ALPGPU_OK;
}

int alpgpu_decode_count_range_f32(alpgpu_ctx* ctx, const alpgpu_column* col, float lo, float hi, uint32_t* d_counts) {
	ALPGPU_CHECK_CTX(ctx);
	if (!col || (!d_counts && col->n_vectors)) { return fail(ALPGPU_ERR_INVALID, "null column or output"); }
	if (col->n_vectors == 0) { return ALPGPU_OK; }
	if (!col->d_vectors || !col->d_rowgroups) { return fail(ALPGPU_ERR_INVALID, "column has no descriptors"); }
	if ((use_direct_sink_f32(ctx, col) ? alpgpu::launch_sink_direct_f32(ctx->stream, col, lo, hi, d_counts, true) : alpgpu::launch_decode_count_range_f32(ctx->stream, col, lo, hi, d_counts)) != ALPGPU_OK) {
		return fail(ALPGPU_ERR_HIP, "decode-count launch failed", hipGetLastError());
	}
	return ALPGPU_OK;
}

} // extern "C"
