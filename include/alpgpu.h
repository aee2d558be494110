/*
 * alpgpu.h — C ABI of libalpgpu.so: the MI355X (gfx950) ALP / ALP_RD vector codec.
 *
 * This is the drop-in boundary of the repository.  Everything behind it is hand-written HIP for CDNA4;
 * there is NO CPU fallback: every entry point fails with ALPGPU_ERR_NO_DEVICE when no gfx950 device is
 * usable.  The C++ header include/alp.hpp (source-compatible with the reference's include/alp.hpp:4-13)
 * and the Python harness (alp_amd/capi.py) are both thin callers of these functions.
 *
 * The reference (cwida/ALP) has no FFI layer; its boundary is the C++ vector API.  Each batch entry
 * point below names the reference function(s) it replaces, file:line relative to /root/reference:
 *
 *   alpgpu_rowgroup_init_f64   alp::encoder<double>::init            include/alp/encoder.hpp:420-427
 *                              (sampler::first_level_sample           include/alp/sampler.hpp:14-52,
 *                               find_top_k_combinations               include/alp/encoder.hpp:139-235)
 *                              alp::rd_encoder<double>::init          include/alp/rd.hpp:180-185 (:33-104)
 *   alpgpu_state_from_samples_f64  find_top_k_combinations (:139-235) + find_best_dictionary (rd.hpp:89-104) on given samples
 *   alpgpu_encode_f64          alp::encoder<double>::encode           include/alp/encoder.hpp:402-418
 *                              (find_best_exponent_factor_from_combinations :241-305, encode_simdized :307-400)
 *                              alp::encoder<double>::analyze_ffor     include/alp/encoder.hpp:109-120
 *                              ffor::ffor(int64/uint64/uint16)        include/fastlanes/ffor.hpp:7-15
 *                                                                     (src/fastlanes_generated_ffor.cpp:29781,29939)
 *                              alp::rd_encoder<double>::encode        include/alp/rd.hpp:109-147
 *   alpgpu_decode_f64          generated::falp::fallback::scalar::falp include/alp/falp.hpp:10-26 (src/falp.cpp:42440)
 *                              alp::decoder<double>::patch_exceptions include/alp/decoder.hpp:141-149
 *                              unffor::unffor(uint64/uint16)          include/fastlanes/unffor.hpp:7-15
 *                              alp::rd_encoder<double>::decode        include/alp/rd.hpp:152-178
 *   alpgpu_decode_sum_f64      falp + patch_exceptions fused with a SUM consumer (bench_end_to_end .../queries/q1.cpp:63-104)
 *   alpgpu_ffor_i64 / alpgpu_unffor_i64 / alpgpu_ffor_u16 / alpgpu_unffor_u16 / alpgpu_ffor_u8 / alpgpu_unffor_u8 (+ _i32, float section)
 *                              ffor::ffor / unffor::unffor            include/fastlanes/{ffor,unffor}.hpp:7-15
 *   alpgpu_falp_f64            falp (no exception patching)           include/alp/falp.hpp:10-26
 *   alpgpu_decode_values_f64   alp::decoder<double>::decode           include/alp/decoder.hpp:134-138
 *   alpgpu_patch_f64           alp::decoder<double>::patch_exceptions include/alp/decoder.hpp:141-149
 *   alpgpu_encode_simdized_f64 alp::encoder<double>::encode_simdized  include/alp/encoder.hpp:307-400
 *   alpgpu_analyze_ffor_i64    alp::encoder<double>::analyze_ffor     include/alp/encoder.hpp:109-120
 *   alpgpu_rd_encode_vectors_f64 / alpgpu_rd_decode_vectors_f64
 *                              alp::rd_encoder<double>::encode/decode include/alp/rd.hpp:109-147 / :152-178
 *
 * Conventions
 *   - plain C types; every pointer named d_* is a DEVICE pointer (HBM) unless stated otherwise;
 *   - a "vector" is 1024 values, a "rowgroup" is 100 vectors (reference include/alp/config.hpp:11-15);
 *     rowgroup r owns vectors 100r .. 100r+99 and one state (scheme, (e,f) candidates / RD dictionary);
 *   - all work is enqueued on the context's stream (alpgpu_set_stream) and is asynchronous; call
 *     alpgpu_synchronize (or synchronise the stream yourself) before reading results on the host;
 *   - return value: ALPGPU_OK or a negative ALPGPU_ERR_*; alpgpu_last_error() gives the text (thread-local);
 *   - no C++ exceptions cross this boundary; one context per device, used by one host thread at a time.
 */
#ifndef ALPGPU_H
#define ALPGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALPGPU_VECTOR_SIZE 1024u
#define ALPGPU_ROWGROUP_VECTORS 100u
#define ALPGPU_MAX_COMBINATIONS 5u
#define ALPGPU_RD_DICT_SIZE 8u

/* same numeric values as the reference's alp::Scheme (include/alp/constants.hpp:10-14) */
#define ALPGPU_SCHEME_INVALID 0u
#define ALPGPU_SCHEME_ALP_RD 1u
#define ALPGPU_SCHEME_ALP 2u

#define ALPGPU_OK 0
#define ALPGPU_ERR_NO_DEVICE (-1)   /* no usable gfx950 device / HIP runtime failure at start-up */
#define ALPGPU_ERR_INVALID (-2)     /* bad argument */
#define ALPGPU_ERR_HIP (-3)         /* a HIP call failed; see alpgpu_last_error() */
#define ALPGPU_ERR_CAPACITY (-4)    /* an output stream was too small (reported by alpgpu_column_totals) */

typedef struct alpgpu_ctx alpgpu_ctx;

/* Per-rowgroup codec state in HBM (32 bytes).  Mirror of the fields of alp::state<double>
 * (reference include/alp/encoder.hpp:35-62) that outlive init(). */
typedef struct alpgpu_rowgroup_state {
	uint8_t  scheme;      /* ALPGPU_SCHEME_ALP | ALPGPU_SCHEME_ALP_RD */
	uint8_t  k;           /* ALP: number of (e,f) candidates, 1..5 (state.k_combinations) */
	uint8_t  combos[10];  /* ALP: combos[2i] = exponent e, combos[2i+1] = factor f (best_k_combinations) */
	uint8_t  rd_rbw;      /* ALP_RD: right_bit_width (48..63) */
	uint8_t  rd_lbw;      /* ALP_RD: left_bit_width (1..3) */
	uint8_t  rd_dict_size;/* ALP_RD: actual_dictionary_size (1..8) */
	uint8_t  pad;
	uint16_t rd_dict[8];  /* ALP_RD: left_parts_dict */
} alpgpu_rowgroup_state;

/* Per-vector descriptor in HBM (32 bytes, 32-byte aligned; one scalar load per wavefront).
 * Modelled on the reference's per-vector metadata (state.exp/fac/bit_width/for_base + exceptions_count;
 * the 48-byte alp_m record of publication/source_code/bench_end_to_end/include/encoding/helper.hpp:36-67). */
typedef struct alpgpu_vector_desc {
	uint64_t packed_off;  /* byte offset of this vector's bit-packed words in the packed stream (multiple of 128).
	                         ALP: 128*bw bytes.  ALP_RD: 128*rbw bytes (right, u64 lanes) then 128*lbw (left, u16 lanes) */
	uint64_t exc_off;     /* byte offset of this vector's exception record in the exception stream (multiple of 8).
	                         ALP: cnt*8 B values (f64 bits) then cnt*2 B positions.  ALP_RD: cnt*2 B left parts then cnt*2 B positions.
	                         record size is rounded up to 8 bytes; both encode forms write the pad bytes as zero (the streams are byte-reproducible into a dirty buffer) */
	int64_t  base;        /* ALP: frame-of-reference base (for_base).  ALP_RD: 0 */
	uint8_t  bw;          /* ALP: bit width 0..64.  ALP_RD: right bit width */
	uint8_t  e;           /* ALP: exponent index (state.exp) */
	uint8_t  f;           /* ALP: factor index (state.fac) */
	uint8_t  lbw;         /* ALP_RD: left bit width; ALP: 0 */
	uint16_t exc_cnt;     /* number of exceptions 0..1024 */
	uint16_t scheme;      /* copy of the rowgroup's scheme */
} alpgpu_vector_desc;

/* A compressed column resident in HBM.  All buffers are caller-allocated (alpgpu_malloc or any HIP
 * allocator, e.g. a torch tensor) and caller-owned, like every buffer of the reference API. */
typedef struct alpgpu_column {
	uint64_t               n_vectors;
	uint64_t               n_rowgroups;     /* ceil(n_vectors / 100) */
	alpgpu_rowgroup_state* d_rowgroups;     /* [n_rowgroups] */
	alpgpu_vector_desc*    d_vectors;       /* [n_vectors] */
	uint8_t*               d_packed;        /* packed stream, 128-byte aligned */
	uint64_t               packed_capacity; /* bytes; worst case n_vectors * 8448 */
	uint8_t*               d_exc;           /* exception stream, 8-byte aligned */
	uint64_t               exc_capacity;    /* bytes; worst case n_vectors * 10240 */
	uint64_t*              d_totals;        /* [8]: [0] packed bytes used, [1] exception bytes used, [2] overflow flag, [3] look-back stall flag of the
	                                           single-pass encode (always 0 once alpgpu_encode_* has drained: the recovery route clears it),
	                                           [4..5] running totals of the encode launch in flight, [6] recovery gate (latched from [3]),
	                                           [7] ALP_RD rowgroups as last counted by alpgpu_column_totals */
	/* host-side hints (0 = unknown): stream sizes as last seen by the host.  Filled by alpgpu_column_totals and
	 * alpgpu_column_from_blob; decode uses them only to pick its launch shape (ALPGPU_OPT_DECODE_VECTORS_PER_WG = 0 "auto") */
	uint64_t               packed_bytes_hint;
	uint64_t               exc_bytes_hint;
	/* optional (NULL = absent; ABI version 2), ALP_RD rowgroups only: ALPGPU_RD_ORDER_STRIDE u16 words per rowgroup =
	 * { D, then the D distinct sampled left parts in the reference's sorted order (left_parts_sorted_repetitions,
	 * include/alp/rd.hpp:47-60) }.  alpgpu_rowgroup_init_* writes it; alpgpu_encode_vectors_* reads it to pack, at exception
	 * slots, the left index the reference packs there (its map position, rd.hpp:69-77, :130-135).  Without it those slots
	 * carry the dictionary size.  No decoder reads these bits; a table that does not start with the rowgroup's dictionary
	 * (e.g. states supplied by the caller) is ignored. */
	uint16_t*              d_rd_order;
	/* host-side hint (ABI version 3; zero-initialise it): 0 = unknown, else 1 + the number of ALP_RD rowgroups of the column, as counted by
	 * alpgpu_column_totals (a small kernel over d_rowgroups) or alpgpu_column_from_blob.  Informational: the fused consumers chose their
	 * kernel by it while the one-wavefront kernel's ALP_RD arm spilled; they no longer read it. */
	uint64_t               alp_rd_rowgroups_hint;
} alpgpu_column;
#define ALPGPU_RD_ORDER_STRIDE 296u

/* ---- context / plumbing ------------------------------------------------------------------------- */
int         alpgpu_ctx_create(int device, alpgpu_ctx** out_ctx);
int         alpgpu_init(int device, alpgpu_ctx** out_ctx); /* the name SURVEY.md §8(b) uses for the same call */
void        alpgpu_ctx_destroy(alpgpu_ctx* ctx);
const char* alpgpu_last_error(void);
int         alpgpu_abi_version(void);
/* run on a caller-provided hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = the device's
 * default (null) stream.  A new context runs on its own non-blocking stream until this is called. */
int         alpgpu_set_stream(alpgpu_ctx* ctx, void* hip_stream);
int         alpgpu_use_own_stream(alpgpu_ctx* ctx);
int         alpgpu_synchronize(alpgpu_ctx* ctx);
/* tuning knobs (never change results).  ALPGPU_OPT_DECODE_VECTORS_PER_WG: 1 or 2 consecutive vectors per decode
 * workgroup, or 0 (default) = choose from the column's size hints: 2 keeps twice the bytes in flight and is faster for
 * narrow columns (average packed width <= 17 bits; <= 22 bits when there are about two or more exceptions per vector: crossovers
 * re-measured at one-bit resolution in rounds 4 and 5), 1 for wider ones — every ALP_RD column.  A column without size hints: ALPGPU_OPT_DECODE_UNHINTED.
 * Float columns: 1, 2 or 4; 8 = one wavefront per vector; 16..30 = the column STREAMED by persistent workgroups (decode_stream_f32_kernels.hip: loading
 * wavefronts several chunks ahead, decoding wavefronts that own whole vectors; the values name chunk size / arena / wavefront counts, 27 = chunks of 12 vectors,
 * 24 KiB of records per chunk, 12 decoding + 2 loading wavefronts, one workgroup per CU).  0 = 2, and 27 for hinted columns of >= 32 768 vectors with more
 * than 1.5 and at most 8.5 packed bits per value and fewer than ~2 exceptions per vector (round 6: 0.65-0.72 of the HBM peak against 0.60-0.62).
 * 4 (double columns; round 4) = four vectors per workgroup over a 2.25 KiB stage, vectors wider than 17 bits read straight from HBM:
 * built to lift the narrow widths' floor, measured SLOWER than 2 at every width (profiles/r04_decode_floor.txt), never chosen by 0.
 * ALPGPU_OPT_DECODE_PLAIN_STORES: 1 = ordinary instead of non-temporal stores. */
#define ALPGPU_OPT_DECODE_VECTORS_PER_WG 1
#define ALPGPU_OPT_DECODE_PLAIN_STORES 2
/* ALPGPU_OPT_ENCODE_TWO_PASS: 1 = analysis pass + scan + pack pass (reads the input twice) instead of the default
 * single-pass encode whose output offsets come from an in-kernel look-back; both give byte-identical columns, double and
 * float.  The single pass needs its predecessor workgroups to run; should its look-back ever give up (HIP promises no
 * dispatch order), the two-pass kernels enqueued behind it on the same stream redo the column: callers see a complete column. */
#define ALPGPU_OPT_ENCODE_TWO_PASS 3
/* ALPGPU_OPT_DEBUG_FORCE_STALL: 1 = every look-back of the single pass that has to wait gives up at once (tests of the
 * recovery route; the result is still a complete, byte-identical column). */
#define ALPGPU_OPT_DEBUG_FORCE_STALL 4
/* ALPGPU_OPT_ENCODE_ASYNC_INIT: 1 (default) = alpgpu_encode_f64 of a column of >= 1024 rowgroups runs the rowgroup search as a
 * persistent kernel on a second, internal stream BESIDE the single-pass vector encode, which polls for each rowgroup's state (the two
 * rejoin the context's stream before the call's last launches: callers still see one stream); 0 = search, then vectors, on one stream;
 * 2 = beside the encode for float columns as well (measured slower there: the float tiles leave the search no room).  Results are
 * identical in every mode. */
#define ALPGPU_OPT_ENCODE_ASYNC_INIT 6
/* ALPGPU_OPT_CONSUMER_PIPELINED: which kernel runs alpgpu_decode_sum_f64 / alpgpu_decode_count_range_f64 / alpgpu_column_sum_f64.
 * 0 (default) and 2 = ONE wavefront per vector (no barrier, one wave-uniform prologue per vector, eight wavefronts per SIMD), whatever the
 *     column holds: narrow ALP vectors (bit width <= 28, <= 48 exceptions) staged whole in the wavefront's LDS by LDS-DMA, every other
 *     vector's packed words read straight from HBM with bounded buffer loads;
 * 3 = the staged kernel (four wavefronts per vector, one short-lived workgroup per two vectors); same bits as 0 / 2 (the order documented
 *     at alpgpu_decode_sum_f64);
 * 1 = the persistent, software-pipelined kernel of alp_amd/csrc/consume_kernels.hip (one wavefront per vector, packed words, exception
 *     records and descriptors prefetched into per-wavefront LDS rings by LDS-DMA).  Its summation order is its own: lane L adds its 16
 *     values 128m + 2L, 128m + 2L + 1 (m = 0..7) in ascending order from +0.0, then the adjacent-lane tree over the 64 lane sums.
 * Measured in round 3 (profiles/r03_consumers.txt): what the staged kernel runs out of is instruction issue, the scalar unit first; the
 * one-wavefront kernel does a quarter of its scalar work per vector and is 7-30 % faster on ALP and ALP_RD columns alike (while its ALP_RD
 * arm still spilled, the default chose between the two by alpgpu_column::alp_rd_rowgroups_hint); the ring kernel is slower than both. */
#define ALPGPU_OPT_CONSUMER_PIPELINED 5
/* ALPGPU_OPT_ENCODE_KERNEL: which single-pass kernel alpgpu_encode_f64 / alpgpu_encode_vectors_f64 launch (same bytes either way).
 * ALPGPU_ENCODE_KERNEL_LEAN (default): the input is the only vector-sized thing a wavefront holds — the analysis keeps lane masks, the pack
 * recomputes each integer as it shifts it into a 4 KiB LDS image that also waits for the ordered offset — 6 KiB of LDS and <= 72 VGPRs per
 * wavefront, three 8-vector tiles per CU.  ALPGPU_ENCODE_KERNEL_CLASSIC: round 3's kernel (input + integers + packed units in registers,
 * 8 KiB image, two tiles per CU).  DESIGN.md §3.2. */
#define ALPGPU_OPT_ENCODE_KERNEL 7
#define ALPGPU_ENCODE_KERNEL_LEAN 0
#define ALPGPU_ENCODE_KERNEL_CLASSIC 1
/* ALPGPU_OPT_DECODE_PAIRING (double store decode only; 0 = the library's choice, the default — since round 5 never this kernel): workgroups that own two consecutive vectors and choose how to
 * run them from the two descriptors — 1: together when both are narrow, else one after the other; 2: as 1 with the second vector's loads issued in front
 * of the first one's unpack; 3: three vectors per two workgroups.  Same output bytes as every other shape (tests/test_decode_gpu.py). */
#define ALPGPU_OPT_DECODE_PAIRING 8
/* ALPGPU_OPT_DECODE_PATCH_AFTER (double store decode; round 5; an EXPERIMENT that lost — effective only in builds of decode_kernels.hip with
 * -DALPGPU_DECODE_PATCH_MODE=1 or 2, ignored by the default build; default 0): an ALP vector with 1..value exceptions (value <= 64) is unpacked as
 * if it had none — no exception mask, no rank lookup — and its exceptions are put in afterwards: mode 1 by 8-byte stores over the stored values (the
 * reference's own order, include/alp/decoder.hpp:141-149), mode 2 in registers through a per-wavefront slot table.  Same output bytes
 * (tests/test_decode_gpu.py runs every exception count 0..1024 under every limit); measured against the mask route in profiles/r05_decode_exceptions.txt. */
#define ALPGPU_OPT_DECODE_PATCH_AFTER 9
/* ALPGPU_OPT_ENCODE_UNORDERED (double columns under ALPGPU_ENCODE_KERNEL_LEAN, and float columns; round 5; default 0): 1 = alpgpu_encode_* / alpgpu_encode_vectors_* do not
 * assign stream offsets in vector order.  Each 8-vector tile reserves its packed / exception bytes with ONE atomic add when its analysis is done,
 * instead of waiting for the sizes of every tile before it (the ordered form's look-back).  Every vector's record — descriptor fields, packed words,
 * exception values and positions — is byte for byte what the ordered form writes; what changes is WHERE in d_packed / d_exc a tile's records lie
 * (tiles in the order they finished; the eight vectors of a tile stay adjacent, in order), so the two streams as a whole are a permutation of
 * the reference's by tiles and differ from run to run.  Every decoder and consumer of this library follows the descriptors' offsets and
 * does not care; alpgpu_column_to_blob serializes such a column as it is (alpgpu_column_from_blob's validation accepts it: records may lie
 * anywhere inside the streams as long as they do not leave them); alpgpu_decompress_host_* — which uploads a blob's streams chunk by chunk and relies on
 * offsets that ascend with the vector index — refuses such a blob (ALPGPU_ERR_INVALID): decode it with alpgpu_column_from_blob + alpgpu_decode_*.
 * The host pipeline (alpgpu_compress_host_*) always uses the ordered form.
 * If the rowgroup search beside the encode stalls, the recovery route rewrites the column in vector order. */
#define ALPGPU_OPT_ENCODE_UNORDERED 10
/* ALPGPU_OPT_DECODE_RESIDENCY_PAD (tuning aid): KiB of unused dynamic LDS every double store-decode workgroup asks for, which caps the workgroups
 * resident per CU (160 KiB / (its own 9.6 or 19.3 KiB + this)); -1 (default) = chosen from the column's size hints (what a CU wants
 * is an amount of bytes in flight).  Never changes results (tests/test_decode_gpu.py: 0 .. 120 KiB at one and two vectors per workgroup); a pad that does not fit
 * beside the workgroup's own LDS any more (150 KiB with two vectors per workgroup) makes the launch fail: alpgpu_decode_f64 returns ALPGPU_ERR_HIP, nothing is written. */
#define ALPGPU_OPT_DECODE_RESIDENCY_PAD 11
/* ALPGPU_OPT_DECODE_READ_AHEAD (store decode, double since round 5, float since round 6): alpgpu_decode_f64 / _f32 start a READ-AHEAD beside the decode kernel — a few persistent workgroups
 * on the context's second stream that pull descriptors, packed words and exception records into the Infinity Cache ALPGPU_OPT_DECODE_READ_AHEAD_US
 * microseconds (default 0 = by the vectors' width: 18 us at 1 bit .. 60 us from 8 bits on) ahead of the decode kernel, which tells them where it is; the decode's two dependent reads then hit the cache.
 *   -1 (default)  columns of >= 262144 vectors of at most 6 packed bits per value on average (7 when they have exceptions) whose size hints are set (+3-24 % there; wider columns lose);
 *    0            never;   1  every column of >= 32768 vectors whose size hints are set (measurements).
 * Same output bytes.  The two kernels are joined on the context's stream: work enqueued behind alpgpu_decode_f64 waits for both.
 * Where kernels of two streams cannot run side by side (GPU_MAX_HW_QUEUES=1, AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING, a counter-collection pass) the read-ahead
 * leaves when no word from the decode arrives (round 6: within max(200 us, a quarter of the decode's own estimated duration); until round 5 a flat 50 ms), and a context
 * created with one of those three variables set does not start it on its own (-1 behaves as 0).  One context runs ONE store decode at a time (one progress word). */
#define ALPGPU_OPT_DECODE_READ_AHEAD 12
#define ALPGPU_OPT_DECODE_READ_AHEAD_US 13
/* ALPGPU_OPT_DECODE_SEGMENTS (double store decode; round 5; default 1): alpgpu_column_totals and alpgpu_column_from_blob remember, in the context, the sizes of up to 32
 * segments of the column; alpgpu_decode_f64 of that column through the same context merges adjacent segments of the same kind (by packed width and exceptions) into at most 8 runs
 * and decodes run by run, each with the launch shape (and read-ahead) its own sizes call for — a column whose regions differ is no longer decoded in the shape of its average.
 * Columns of one kind, columns without the call, forced launch shapes: one launch, as before.  0 = never.  alpgpu_decode_runs tells.  Same bytes. */
#define ALPGPU_OPT_DECODE_SEGMENTS 14
/* ALPGPU_OPT_DECODE_UNHINTED (round 6; default 1): a column of >= 65536 vectors whose packed_bytes_hint and exc_bytes_hint are both 0 — encoded a moment ago, nobody called
 * alpgpu_column_totals (a host synchronisation) — no longer decodes blind: its sizes are summed on the stream and the launch rule is evaluated on the DEVICE.
 *   1  the decode kernel is launched in the shape such a column always got (one vector per workgroup; float: two) and the read-ahead beside it takes "whether", its lead
 *      and its pace from the device-side plan (long narrow columns: 0.50 -> 0.60 of the HBM peak on the first decode); the sums travel to page-locked host memory behind an
 *      event that is only queried, and the next alpgpu_decode_* of the same column (same buffers and length, not encoded again through this context in between) is planned
 *      on the host like a hinted one — shape, residency, regions (0.72-0.76);
 *   2  as 1, and the first decode launches EVERY candidate shape gated on the plan's word (closed candidates cost their dispatch: ~0.19 ms per 1 Mi empty workgroups,
 *      which is why this is not the default: profiles/r06_decode_policy.txt);
 *   0  one vector per workgroup (float: two), no read-ahead, nothing learned: as before round 6.
 * No host synchronisation in any mode.  Same bytes. */
#define ALPGPU_OPT_DECODE_UNHINTED 15
int         alpgpu_set_option(alpgpu_ctx* ctx, int option, int64_t value);
/* the launch shape alpgpu_decode_f64 (is_f32 = 0) or alpgpu_decode_f32 (1) would use for this column now: vectors per decode
 * workgroup (1, 2 or 4), from ALPGPU_OPT_DECODE_VECTORS_PER_WG and the column's size hints; negative on bad arguments */
int         alpgpu_decode_vectors_per_wg(alpgpu_ctx* ctx, const alpgpu_column* col, int is_f32);
/* ... and whether that decode would run with the read-ahead beside it (ALPGPU_OPT_DECODE_READ_AHEAD): 1 / 0; negative on bad arguments */
int         alpgpu_decode_reads_ahead(alpgpu_ctx* ctx, const alpgpu_column* col, int is_f32);
/* ... and in how many launches alpgpu_decode_f64 (alpgpu_decode_runs_f32: alpgpu_decode_f32) would decode it (ALPGPU_OPT_DECODE_SEGMENTS): 1, or the number of runs; negative
 * on bad arguments.  The three functions above describe the plan of a column whose sizes the host knows, for the column as a WHOLE: a column decoded in several runs gets
 * a shape per run (alpgpu_decode_runs > 1), an unhinted one the shape its device-side plan picks (alpgpu_debug_unhinted_plan). */
int         alpgpu_decode_runs(alpgpu_ctx* ctx, const alpgpu_column* col);
int         alpgpu_decode_runs_f32(alpgpu_ctx* ctx, const alpgpu_column* col);
/* debug aids (tests; both wait for the context's streams): batches of 64 vectors this context's read-aheads have read since it was created; and the plan of its last
 * unhinted decode: out6[0] the candidate that ran (double: 1 = one vector per workgroup, 2 = two, 3 = one with seven workgroups per CU; float: 1 = two, 2 = four),
 * [1] read-ahead lead_min | lead_max << 32 in vectors (0: none), [2] picoseconds per vector | widest record read ahead << 32, [3..5] packed bytes, exceptions, ALP_RD vectors */
int         alpgpu_debug_read_ahead_batches(alpgpu_ctx* ctx, uint64_t* batches);
int         alpgpu_debug_unhinted_plan(alpgpu_ctx* ctx, uint64_t* out6);
/* ... and: forget what the context remembers about this column (its segments, what an unhinted decode learned), as every alpgpu_encode_* into it does */
int         alpgpu_debug_forget_column(alpgpu_ctx* ctx, const alpgpu_column* col);
/* ---- host-resident columns -----------------------------------------------------------------------------------------------
 * The reference's callers (publication/source_code/bench_compression_ratio/alp.cpp:198-229) hold the column and what they
 * compress it into in host memory.  These entry points take it from there: n_values values at h_in (the last vector may be
 * incomplete: it is padded as alpgpu_pad_tail_* does) become a serialized column at h_blob — byte for byte the blob that
 * alpgpu_encode_* + alpgpu_column_to_blob produce for the same values — and back.  Inside, chunks of whole rowgroups travel up on
 * one stream while the previous chunk is encoded on another and its packed bytes come down; the exception stream collects in HBM and follows at the end
 * (decompression: streams up chunk by chunk, each chunk decoded from a view of the column, doubles down on the chunk's stream).
 * Page-locked h_in / h_blob / h_out (alpgpu_malloc_host) give the link's rate (measured host to host: ~40 GB/s of doubles each
 * way); pageable memory works, at the runtime's staging rate.  Synchronous: everything has arrived when the call returns.
 * Capacity: alpgpu_blob_size(n_vectors, alpgpu_packed_capacity(n_vectors), alpgpu_exc_capacity(n_vectors)) always suffices; a
 * smaller buffer returns ALPGPU_ERR_CAPACITY with the needed size in *written.  The context's stream is left as it was. */
int alpgpu_compress_host_f64(alpgpu_ctx* ctx, const double* h_in, uint64_t n_values, void* h_blob, uint64_t capacity, uint64_t* written);
int alpgpu_compress_host_f32(alpgpu_ctx* ctx, const float* h_in, uint64_t n_values, void* h_blob, uint64_t capacity, uint64_t* written);
/* *n_values receives the column's value count (also when h_out is too small: ALPGPU_ERR_CAPACITY); the blob is validated first */
int alpgpu_decompress_host_f64(alpgpu_ctx* ctx, const void* h_blob, uint64_t size, double* h_out, uint64_t out_capacity_values, uint64_t* n_values);
int alpgpu_decompress_host_f32(alpgpu_ctx* ctx, const void* h_blob, uint64_t size, float* h_out, uint64_t out_capacity_values, uint64_t* n_values);

/* The same over SEVERAL contexts — normally one per GPU of the node, each with its own PCIe link (several contexts on one device work
 * too): the column is cut into n_ctx contiguous whole-rowgroup shards (sizes differ by at most one rowgroup, the rule of
 * alp_amd/sharding.py: rowgroup_shard), shard i runs through ctxs[i]'s pipeline on its own host thread, and the pieces become ONE blob —
 * byte for byte the blob alpgpu_compress_host_* writes with a single context (the descriptors' stream offsets continue where the
 * shards before ended: no collective, a host-side concatenation is the only exchange).  This is how a C / C++ caller that holds a
 * host column (publication/source_code/bench_compression_ratio/alp.cpp:198-229; the worker loop of
 * publication/source_code/bench_end_to_end/src/benchmarks/alp/run_query.cpp:233-305) uses all GPUs of a node from one process.
 * Capacity: regions of the caller's buffer, proportional to the shards' vector counts, serve as the shards' staging, so a shard that
 * compresses worse than the column's average can fail with ALPGPU_ERR_CAPACITY although the sum would fit; *written is then the capacity with
 * which every shard fits its region (never more than the worst-case size alpgpu_blob_size(n, packed_capacity(n), exc_capacity(n)), which
 * always suffices — the capacities' constant terms cover the regions' 8-byte rounding for n_ctx <= 64); on success *written is the blob's size.
 * A context must not be used by anything else during the call; ctxs[i] must be distinct. */
int alpgpu_compress_host_multi_f64(alpgpu_ctx* const* ctxs, int n_ctx, const double* h_in, uint64_t n_values, void* h_blob, uint64_t capacity, uint64_t* written);
int alpgpu_compress_host_multi_f32(alpgpu_ctx* const* ctxs, int n_ctx, const float* h_in, uint64_t n_values, void* h_blob, uint64_t capacity, uint64_t* written);
int alpgpu_decompress_host_multi_f64(alpgpu_ctx* const* ctxs, int n_ctx, const void* h_blob, uint64_t size, double* h_out, uint64_t out_capacity_values,
                                     uint64_t* n_values);
int alpgpu_decompress_host_multi_f32(alpgpu_ctx* const* ctxs, int n_ctx, const void* h_blob, uint64_t size, float* h_out, uint64_t out_capacity_values,
                                     uint64_t* n_values);

/* Measurement aid, not part of the codec: launches the memory traffic of the single-pass encode without its arithmetic — the same
 * launch shape, every 8 KiB vector of d_in read once, write_bytes_per_vector (a multiple of 16, <= 8192) written per vector at
 * d_out + v * write_bytes_per_vector, stored data depending on all loaded data.  bench.py times it to put a measured ceiling for the
 * encode's read/write mix next to the nominal HBM peak. */
int         alpgpu_debug_traffic_probe(alpgpu_ctx* ctx, const void* d_in, void* d_out, uint64_t n_vectors, uint32_t write_bytes_per_vector);
/* ... and with the rowgroup search of alpgpu_encode_f64 (the search of /root/reference include/alp/encoder.hpp:139-235 and rd.hpp:89-104, over d_in, a column
 * of doubles) running beside it on the context's second stream, launched exactly as the encode launches it; its states go to scratch->d_rowgroups
 * (and d_rd_order).  n_vectors >= 102400 (the search runs beside the encode from 1024 rowgroups on).  write_bytes_per_vector = 0 makes either probe a
 * read-only stream of the input.  Round 5: the measured speed of light of "the encode's bytes + the search's instructions" for bench.py. */
int         alpgpu_debug_traffic_probe_with_search(alpgpu_ctx* ctx, const double* d_in, void* d_out, uint64_t n_vectors, uint32_t write_bytes_per_vector,
                                                   alpgpu_column* scratch);
/* Measurement aid, not part of the codec: alpgpu_decode_sum_f64 with the unpack arithmetic left out — the same descriptor and packed-word
 * loads into LDS, the same barrier and reduction, one double per vector written to d_out (its value means nothing).  bench.py times it
 * to say how much of the fused consumers' time is their chain of dependent loads. */
int         alpgpu_debug_decode_probe_f64(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_out);
/* device properties the bench reports: [0]=CU count, [1]=LDS bytes/CU... see alp_amd/capi.py */
int         alpgpu_device_info(alpgpu_ctx* ctx, char* name_out, size_t name_cap, int* cu_count, uint64_t* hbm_bytes);

/* device memory helpers for hosts that do not link a HIP runtime themselves (include/alp.hpp) */
int alpgpu_malloc(alpgpu_ctx* ctx, void** d_ptr, size_t bytes);
int alpgpu_free(alpgpu_ctx* ctx, void* d_ptr);
int alpgpu_memcpy_h2d(alpgpu_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int alpgpu_memcpy_d2h(alpgpu_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
int alpgpu_memset(alpgpu_ctx* ctx, void* d_dst, int value, size_t bytes);
/* page-locked host memory (copies to and from it are plain DMA) and a host-to-device copy that is enqueued but not waited for:
 * h_src must stay untouched until a later synchronous call on this context (alpgpu_memcpy_d2h, alpgpu_synchronize) has returned.
 * include/alp/batch.hpp moves a whole rowgroup's arrays with one copy each way through such a buffer. */
int alpgpu_malloc_host(alpgpu_ctx* ctx, void** h_ptr, size_t bytes);
int alpgpu_free_host(alpgpu_ctx* ctx, void* h_ptr);
int alpgpu_memcpy_h2d_async(alpgpu_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);

/* worst-case stream capacities for n_vectors (bytes) */
uint64_t alpgpu_packed_capacity(uint64_t n_vectors);
uint64_t alpgpu_exc_capacity(uint64_t n_vectors);

/* ---- whole-column hot path ----------------------------------------------------------------------- */

/* Rowgroup init for every rowgroup of the column: first-level sampling, (e,f) top-k search, scheme
 * decision, and for ALP_RD rowgroups the cut/dictionary search.  Writes col->d_rowgroups[0..n_rowgroups). */
int alpgpu_rowgroup_init_f64(alpgpu_ctx* ctx, const double* d_in, uint64_t n_vectors, alpgpu_column* col);

/* The same decision for ONE rowgroup from first-level samples the caller gathered (what the reference's
 * find_top_k_combinations, include/alp/encoder.hpp:139-235, and find_best_dictionary, include/alp/rd.hpp:89-104,
 * receive): d_samples holds n_samples (1..288) doubles, blocks of min(n_samples, 32) per sampled vector. */
int alpgpu_state_from_samples_f64(alpgpu_ctx* ctx, const double* d_samples, uint32_t n_samples, alpgpu_rowgroup_state* d_state);
/* rd_encoder::init's half alone (include/alp/rd.hpp:180-185): cut + dictionary for these samples, no ALP/ALP_RD re-decision */
int alpgpu_rd_state_from_samples_f64(alpgpu_ctx* ctx, const double* d_samples, uint32_t n_samples, alpgpu_rowgroup_state* d_state);
/* rd_encoder::build_left_parts_dictionary for ONE cut position (include/alp/rd.hpp:33-87, called by find_best_dictionary :89-104 for every cut
 * and once more for the chosen one): right_bit_width = 64 - cut for doubles (48..63), 32 - cut for floats (16..31).  *d_state receives the ALP_RD
 * state of that cut (bit widths, dictionary size, dictionary in the reference's order), *d_estimate the value the reference returns —
 * estimate_compression_size (rd.hpp:23-31): right + left bit width + 32 bits per sampled exception / sample.  Both in device memory. */
int alpgpu_rd_dictionary_for_cut_f64(alpgpu_ctx* ctx, const double* d_samples, uint32_t n_samples, uint8_t right_bit_width, alpgpu_rowgroup_state* d_state,
                                     double* d_estimate);
int alpgpu_rd_dictionary_for_cut_f32(alpgpu_ctx* ctx, const float* d_samples, uint32_t n_samples, uint8_t right_bit_width, alpgpu_rowgroup_state* d_state,
                                     double* d_estimate);

/* Vector encode of the whole column given col->d_rowgroups (from alpgpu_rowgroup_init_f64 or supplied by
 * the caller): second-level sampling, encode + exception compaction, analyze_ffor, FFOR pack (ALP);
 * split + dictionary encode + FFOR pack of right/left (ALP_RD).  Single pass over the input; output
 * offsets are assigned in vector order.  Writes d_vectors, d_packed, d_exc, d_totals.  If a stream is too small the
 * overflow flag is raised, nothing is written past the buffers and the column content is unspecified — but every vector
 * that did not fit gets an empty descriptor, so decoding such a column stays inside the buffers. */
int alpgpu_encode_vectors_f64(alpgpu_ctx* ctx, const double* d_in, uint64_t n_vectors, alpgpu_column* col);

/* alpgpu_rowgroup_init_f64 followed by alpgpu_encode_vectors_f64 */
int alpgpu_encode_f64(alpgpu_ctx* ctx, const double* d_in, uint64_t n_vectors, alpgpu_column* col);

/* Fused decode of the whole column: ALP vectors = falp (unFFOR + int->double) + patch_exceptions;
 * ALP_RD vectors = unFFOR(right,left) + dictionary glue + patch.  d_out receives n_vectors*1024 doubles.
 * TRUST: alpgpu_decode_* / alpgpu_decode_sum_* / alpgpu_decode_count_range_* / alpgpu_column_sum_* follow col->d_vectors as they
 * find it — offsets, widths and exception counts are NOT checked by the kernels (a descriptor with a bad packed_off reads out of
 * bounds).  Columns written by alpgpu_encode_* and columns loaded by alpgpu_column_from_blob / alpgpu_decompress_host_* (which
 * validate every descriptor on the host) are safe; for descriptors from anywhere else call alpgpu_column_validate first. */
int alpgpu_decode_f64(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_out);

/* Decode fused into a consumer (the SCAN/SUM shape of the reference's end-to-end bench,
 * publication/source_code/bench_end_to_end/src/benchmarks/alp/queries/q1.cpp:63-104): d_sums[v] = sum of the 1024 decoded
 * values of vector v, exceptions patched in; the doubles themselves never reach HBM.  Summation order (so that the result can
 * be reproduced bit for bit): wavefront q of 4 owns values 256q..256q+255; lane L of it adds its values 256q+2L, +1, 256q+128+2L,
 * +1 in that order starting from +0.0, giving p[q][L]; then s[L] = (p[0][L] + p[1][L]) + (p[2][L] + p[3][L]) for L = 0..63; then the
 * 64 s[L] combine by a balanced binary tree over ADJACENT lanes — (0,1), (2,3), ...; then (0..1, 2..3), ...; six levels.
 * (Earlier in round 3 each wavefront ran that tree over its own 64 partials first and the four results were combined last; round 2
 * used a butterfly.  The order is a property of the library version: tests/test_decode_sum_gpu.py holds the host replica.) */
int alpgpu_decode_sum_f64(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_sums);
/* The other consumer named there, a predicate pushed into the scan: d_counts[v] = number of decoded values x of vector v with
 * lo <= x <= hi (exceptions patched in; NaN never qualifies; -0.0 == 0.0 as in C).  Nothing but 4 bytes per vector is written. */
int alpgpu_decode_count_range_f64(alpgpu_ctx* ctx, const alpgpu_column* col, double lo, double hi, uint32_t* d_counts);

/* The column's total (the reference's consumer adds every vector into ONE accumulator, q1.cpp:91-100): *d_total (device memory,
 * one double) = the per-vector sums of alpgpu_decode_sum_* combined by a balanced binary tree over adjacent elements, absent
 * elements counting as +0.0: level by level, s'[i] = s[2i] + s[2i+1] over the array padded with +0.0 to a multiple of 1024, ten
 * levels per kernel launch, until one element is left.  (A chain of a million dependent adds has no parallel form with the same
 * rounding; this order is the documented replacement, reproduced on the host by tests/test_decode_sum_gpu.py.)  0.0 for an
 * empty column.  alpgpu_tree_sum_f64 is that tree over any device array of n doubles.  Asynchronous on the context's stream. */
int alpgpu_column_sum_f64(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_total);
int alpgpu_column_sum_f32(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_total);
int alpgpu_tree_sum_f64(alpgpu_ctx* ctx, const double* d_in, uint64_t n, double* d_total);

/* Opt-in guard for device-resident columns of unknown origin: one pass over the descriptors on the device checks, for every vector,
 * scheme (and that it is its rowgroup's), widths, exponent / factor, exception count, alignment, that its packed words and its
 * exception record lie inside packed_capacity / exc_capacity, and that every exception position is < 1024.  value_bytes = 8
 * (double column) or 4 (float column).  Synchronises the stream; ALPGPU_OK, or ALPGPU_ERR_INVALID with *first_bad = the lowest
 * offending vector index (optional). */
int alpgpu_column_validate(alpgpu_ctx* ctx, const alpgpu_column* col, int value_bytes, uint64_t* first_bad);

/* host copy of d_totals after the stream has drained: packed bytes, exception bytes, overflow flag */
int alpgpu_column_totals(alpgpu_ctx* ctx, alpgpu_column* col, uint64_t* packed_bytes, uint64_t* exc_bytes, int* overflow);

/* ---- tail padding and a serialized container ---------------------------------------------
 * The codec works on whole 1024-value vectors (reference PRIMITIVES.md:141-144 leaves incomplete last vectors to the
 * caller; its drivers drop them).  alpgpu_pad_tail_f64 fills d_in[n_values .. next multiple of 1024) with the first
 * value of the incomplete vector; d_in must have room for that many doubles.  Encode ceil(n_values/1024) vectors. */
int alpgpu_pad_tail_f64(alpgpu_ctx* ctx, double* d_in, uint64_t n_values);

/* Blob = 64-byte header, rowgroup states, vector descriptors, packed stream, exception stream (all little-endian,
 * exactly the HBM records of this header).  Host memory.  Both calls synchronise the context's stream. */
typedef struct alpgpu_blob_header {
	char     magic[8];      /* "ALPGPU1\0" */
	uint32_t version;       /* 1 */
	uint32_t header_bytes;  /* 64 */
	uint64_t n_values;      /* values that are data (<= n_vectors * 1024; the rest is tail padding) */
	uint64_t n_vectors;
	uint64_t n_rowgroups;
	uint64_t packed_bytes;
	uint64_t exc_bytes;
	uint64_t reserved;       /* value type: 0 or 8 = double column, 4 = float column */
} alpgpu_blob_header;
uint64_t alpgpu_blob_size(uint64_t n_vectors, uint64_t packed_bytes, uint64_t exc_bytes);
int      alpgpu_column_to_blob(alpgpu_ctx* ctx, const alpgpu_column* col, uint64_t n_values, void* h_blob, uint64_t capacity,
                               uint64_t* written);
/* validates the blob (sizes, every descriptor's extents) and copies it into the caller-allocated column buffers */
int      alpgpu_column_from_blob(alpgpu_ctx* ctx, const void* h_blob, uint64_t size, alpgpu_column* col, uint64_t* n_values);

/* ---- vector primitives on batches (fixed strides; the reference's per-vector API, n at a time) ------
 * Each processes n_vectors independent vectors.  "stride" arguments are in ELEMENTS of the pointed type
 * between consecutive vectors (the reference's callers use 1024-element buffers for everything). */

/* ffor::ffor / unffor::unffor, 64-bit lanes: in/out [n][1024] int64, packed [n][packed_stride] (>= 16*bw words
 * used per vector), bw/base per vector */
int alpgpu_ffor_i64(alpgpu_ctx* ctx, const int64_t* d_in, int64_t* d_packed, size_t packed_stride,
                    const uint8_t* d_bw, const int64_t* d_base, uint64_t n_vectors);
int alpgpu_unffor_i64(alpgpu_ctx* ctx, const int64_t* d_packed, size_t packed_stride, int64_t* d_out,
                      const uint8_t* d_bw, const int64_t* d_base, uint64_t n_vectors);
/* 16-bit lanes (ALP_RD left parts) */
int alpgpu_ffor_u16(alpgpu_ctx* ctx, const uint16_t* d_in, uint16_t* d_packed, size_t packed_stride,
                    const uint8_t* d_bw, const uint16_t* d_base, uint64_t n_vectors);
int alpgpu_unffor_u16(alpgpu_ctx* ctx, const uint16_t* d_packed, size_t packed_stride, uint16_t* d_out,
                      const uint8_t* d_bw, const uint16_t* d_base, uint64_t n_vectors);

/* 8-bit lanes (128 lane-streams x 8 rows; not used by the codec, part of the reference's ffor/unffor API: include/fastlanes/ffor.hpp:10) */
int alpgpu_ffor_u8(alpgpu_ctx* ctx, const uint8_t* d_in, uint8_t* d_packed, size_t packed_stride, const uint8_t* d_bw,
                   const uint8_t* d_base, uint64_t n_vectors);
int alpgpu_unffor_u8(alpgpu_ctx* ctx, const uint8_t* d_packed, size_t packed_stride, uint8_t* d_out, const uint8_t* d_bw,
                     const uint8_t* d_base, uint64_t n_vectors);

/* falp without exception patching: packed -> doubles */
int alpgpu_falp_f64(alpgpu_ctx* ctx, const int64_t* d_packed, size_t packed_stride, double* d_out,
                    const uint8_t* d_bw, const int64_t* d_base, const uint8_t* d_fac, const uint8_t* d_exp,
                    uint64_t n_vectors);
/* decoder::decode: encoded integers -> doubles */
int alpgpu_decode_values_f64(alpgpu_ctx* ctx, const int64_t* d_enc, double* d_out, const uint8_t* d_fac,
                             const uint8_t* d_exp, uint64_t n_vectors);
/* decoder::patch_exceptions: out[pos[j]] = exc[j]; exc/pos [n][exc_stride] */
int alpgpu_patch_f64(alpgpu_ctx* ctx, double* d_out, const double* d_exc, const uint16_t* d_pos, size_t exc_stride,
                     const uint16_t* d_cnt, uint64_t n_vectors);
/* encoder::encode_simdized with a given (fac, exp) per vector: -> enc [n][1024], exc/pos [n][exc_stride], cnt [n] */
int alpgpu_encode_simdized_f64(alpgpu_ctx* ctx, const double* d_in, double* d_exc, uint16_t* d_pos, size_t exc_stride,
                               uint16_t* d_cnt, int64_t* d_enc, const uint8_t* d_fac, const uint8_t* d_exp,
                               uint64_t n_vectors);
/* encoder::encode (second-level sampling when state.k > 1, then encode_simdized); one state per vector
 * (d_state_idx[v] indexes d_states; NULL = vector v uses state v/100); chosen (fac, exp) are returned */
int alpgpu_encode_values_f64(alpgpu_ctx* ctx, const double* d_in, const alpgpu_rowgroup_state* d_states,
                             const uint32_t* d_state_idx, double* d_exc, uint16_t* d_pos, size_t exc_stride,
                             uint16_t* d_cnt, int64_t* d_enc, uint8_t* d_fac, uint8_t* d_exp, uint64_t n_vectors);
/* encoder::encode_value<SAFE> (include/alp/encoder.hpp:81-89), the scalar helper of the reference's header, on n_values VALUES (not vectors)
 * with ONE (factor, exponent) pair: d_enc[i] = the encoded integer of d_in[i]; safe != 0 gives the sampling form (the sentinel
 * ENCODING_UPPER_LIMIT for values that cannot be encoded), 0 the plain one (x86 cast semantics).  _f32: `safe` is ignored, the float SAFE
 * branch does not exist in the reference as built. */
int alpgpu_encode_value_f64(alpgpu_ctx* ctx, const double* d_in, int64_t* d_enc, uint8_t fac, uint8_t exp, int safe, uint64_t n_values);
int alpgpu_encode_value_f32(alpgpu_ctx* ctx, const float* d_in, int32_t* d_enc, uint8_t fac, uint8_t exp, int safe, uint64_t n_values);
/* encoder::analyze_ffor */
int alpgpu_analyze_ffor_i64(alpgpu_ctx* ctx, const int64_t* d_enc, uint8_t* d_bw, int64_t* d_base, uint64_t n_vectors);

/* rd_encoder::encode / decode with per-vector state index as above */
int alpgpu_rd_encode_vectors_f64(alpgpu_ctx* ctx, const double* d_in, const alpgpu_rowgroup_state* d_states,
                                 const uint32_t* d_state_idx, uint16_t* d_exc, uint16_t* d_pos, size_t exc_stride,
                                 uint16_t* d_cnt, uint64_t* d_right, uint16_t* d_left, uint64_t n_vectors);
int alpgpu_rd_decode_vectors_f64(alpgpu_ctx* ctx, double* d_out, const uint64_t* d_right, const uint16_t* d_left,
                                 const alpgpu_rowgroup_state* d_states, const uint32_t* d_state_idx,
                                 const uint16_t* d_exc, const uint16_t* d_pos, size_t exc_stride,
                                 const uint16_t* d_cnt, uint64_t n_vectors);
/* the same two under the names SURVEY.md §8(b) lists (alpgpu_rd_encode_f64 / alpgpu_rd_decode_f64) */
int alpgpu_rd_encode_f64(alpgpu_ctx* ctx, const double* d_in, const alpgpu_rowgroup_state* d_states, const uint32_t* d_state_idx,
                         uint16_t* d_exc, uint16_t* d_pos, size_t exc_stride, uint16_t* d_cnt, uint64_t* d_right, uint16_t* d_left,
                         uint64_t n_vectors);
int alpgpu_rd_decode_f64(alpgpu_ctx* ctx, double* d_out, const uint64_t* d_right, const uint16_t* d_left,
                         const alpgpu_rowgroup_state* d_states, const uint32_t* d_state_idx, const uint16_t* d_exc,
                         const uint16_t* d_pos, size_t exc_stride, const uint16_t* d_cnt, uint64_t n_vectors);

/* ==== single precision ====================================================================
 * The float instantiation of the same API: alp::encoder<float> / decoder<float> / rd_encoder<float> (same files and
 * lines as the double entries above), Constants<float> include/alp/constants.hpp:30-64 (66 (e,f) candidates, exception
 * cost 32+16 bits, ALP_RD threshold 22*32), falp float include/alp/falp.hpp:28-44, 32-bit FFOR
 * src/fastlanes_generated_ffor.cpp:1776-7378 (32 lanes x 32 rows).  The column records are the same structs with
 *   packed stream : ALP 128*bw bytes (bw 0..32); ALP_RD 128*rbw (right, u32 lanes, rbw 16..31) then 128*lbw (left, u16 lanes)
 *   exception rec.: ALP cnt*4 B values (f32 bits) then cnt*2 B positions; ALP_RD cnt*2 B left parts then cnt*2 B positions;
 *                   rounded up to 8 bytes
 *   vector_desc.base: the int32 frame-of-reference base, sign-extended.
 * A column encoded by the _f32 functions must be decoded by alpgpu_decode_f32 (the records do not carry the value type;
 * blobs do: alpgpu_blob_header.reserved = 4 for float columns).
 * Behaviour where the reference's float code is undefined in C++ (out-of-range float->int32 casts, the SAFE branch of
 * encode_value, FACT_ARR[10]) follows the reference AS BUILT with Clang, pinned by oracle/alp_oracle_f32.c (see its header). */
uint64_t alpgpu_packed_capacity_f32(uint64_t n_vectors); /* worst case n_vectors * 4352 */
uint64_t alpgpu_exc_capacity_f32(uint64_t n_vectors);    /* worst case n_vectors * 6144 */
int alpgpu_rowgroup_init_f32(alpgpu_ctx* ctx, const float* d_in, uint64_t n_vectors, alpgpu_column* col);
int alpgpu_state_from_samples_f32(alpgpu_ctx* ctx, const float* d_samples, uint32_t n_samples, alpgpu_rowgroup_state* d_state);
int alpgpu_rd_state_from_samples_f32(alpgpu_ctx* ctx, const float* d_samples, uint32_t n_samples, alpgpu_rowgroup_state* d_state);
int alpgpu_encode_vectors_f32(alpgpu_ctx* ctx, const float* d_in, uint64_t n_vectors, alpgpu_column* col);
int alpgpu_encode_f32(alpgpu_ctx* ctx, const float* d_in, uint64_t n_vectors, alpgpu_column* col);
/* ALPGPU_OPT_DECODE_VECTORS_PER_WG applies with values 0 (auto), 1, 2, 4 (float: four vectors over the full stage), 8 and 16..30 (see the option) */
int alpgpu_decode_f32(alpgpu_ctx* ctx, const alpgpu_column* col, float* d_out);
/* The fused consumers of alpgpu_decode_sum_f64 / alpgpu_decode_count_range_f64 for float columns.  Sums accumulate in double
 * (every float widens exactly): thread t = 64 w + L of 256 adds values 4t, 4t+1, 4t+2, 4t+3 in that order starting from 0, giving
 * p[w][L]; then s[L] = (p[0][L] + p[1][L]) + (p[2][L] + p[3][L]); then the balanced tree over adjacent lanes over the 64 s[L] (as for
 * double; earlier in round 3 each wavefront ran the tree first).  ALPGPU_OPT_CONSUMER_PIPELINED: 0 / 2 = one wavefront per vector (the
 * default for float columns whatever they hold), 1 / 3 = the staged four-wavefront kernel; same bits.
 * Round 6: in an ALP vector the exception POSITIONS are skipped in the quads (they contribute nothing to p[w][L]) and the exception VALUES
 * join behind them, in the order of the record: s[L] += (double)exc[j] for j = L, L + 64, L + 128, ... before the tree.  ALP_RD vectors: as
 * before, every value in its place.  (tests/test_decode_sum_gpu.py: host_sums_f32 is the replica.) */
int alpgpu_decode_sum_f32(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_sums);
int alpgpu_decode_count_range_f32(alpgpu_ctx* ctx, const alpgpu_column* col, float lo, float hi, uint32_t* d_counts);
int alpgpu_pad_tail_f32(alpgpu_ctx* ctx, float* d_in, uint64_t n_values);
int alpgpu_column_to_blob_f32(alpgpu_ctx* ctx, const alpgpu_column* col, uint64_t n_values, void* h_blob, uint64_t capacity, uint64_t* written);
int alpgpu_column_from_blob_f32(alpgpu_ctx* ctx, const void* h_blob, uint64_t size, alpgpu_column* col, uint64_t* n_values);
/* batch primitives, 32-bit lanes (same conventions as the 64-bit ones above) */
int alpgpu_ffor_i32(alpgpu_ctx* ctx, const int32_t* d_in, int32_t* d_packed, size_t packed_stride, const uint8_t* d_bw,
                    const int32_t* d_base, uint64_t n_vectors);
int alpgpu_unffor_i32(alpgpu_ctx* ctx, const int32_t* d_packed, size_t packed_stride, int32_t* d_out, const uint8_t* d_bw,
                      const int32_t* d_base, uint64_t n_vectors);
int alpgpu_falp_f32(alpgpu_ctx* ctx, const int32_t* d_packed, size_t packed_stride, float* d_out, const uint8_t* d_bw,
                    const int32_t* d_base, const uint8_t* d_fac, const uint8_t* d_exp, uint64_t n_vectors);
int alpgpu_decode_values_f32(alpgpu_ctx* ctx, const int32_t* d_enc, float* d_out, const uint8_t* d_fac, const uint8_t* d_exp,
                             uint64_t n_vectors);
int alpgpu_patch_f32(alpgpu_ctx* ctx, float* d_out, const float* d_exc, const uint16_t* d_pos, size_t exc_stride,
                     const uint16_t* d_cnt, uint64_t n_vectors);
int alpgpu_encode_simdized_f32(alpgpu_ctx* ctx, const float* d_in, float* d_exc, uint16_t* d_pos, size_t exc_stride, uint16_t* d_cnt,
                               int32_t* d_enc, const uint8_t* d_fac, const uint8_t* d_exp, uint64_t n_vectors);
int alpgpu_encode_values_f32(alpgpu_ctx* ctx, const float* d_in, const alpgpu_rowgroup_state* d_states, const uint32_t* d_state_idx,
                             float* d_exc, uint16_t* d_pos, size_t exc_stride, uint16_t* d_cnt, int32_t* d_enc, uint8_t* d_fac,
                             uint8_t* d_exp, uint64_t n_vectors);
int alpgpu_analyze_ffor_i32(alpgpu_ctx* ctx, const int32_t* d_enc, uint8_t* d_bw, int32_t* d_base, uint64_t n_vectors);
int alpgpu_rd_encode_vectors_f32(alpgpu_ctx* ctx, const float* d_in, const alpgpu_rowgroup_state* d_states, const uint32_t* d_state_idx,
                                 uint16_t* d_exc, uint16_t* d_pos, size_t exc_stride, uint16_t* d_cnt, uint32_t* d_right,
                                 uint16_t* d_left, uint64_t n_vectors);
int alpgpu_rd_decode_vectors_f32(alpgpu_ctx* ctx, float* d_out, const uint32_t* d_right, const uint16_t* d_left,
                                 const alpgpu_rowgroup_state* d_states, const uint32_t* d_state_idx, const uint16_t* d_exc,
                                 const uint16_t* d_pos, size_t exc_stride, const uint16_t* d_cnt, uint64_t n_vectors);

#ifdef __cplusplus
}
#endif
#endif /* ALPGPU_H */
