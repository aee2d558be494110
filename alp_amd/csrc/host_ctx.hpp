// host_ctx.hpp — what the translation units behind include/alpgpu.h share (internal to libalpgpu.so): the context, the error text, the argument-check
// macros and the few helpers more than one of them uses.  The extern "C" surface is spread over
//   api_context.hip    contexts, options, streams, memory, the scan workspace
//   api_encode.hip     rowgroup search + vector encode entry points (and the encode's measurement probes)
//   api_decode.hip     the store decode's LAUNCH POLICY (shape, residency, read-ahead, region by region), the fused consumers, alpgpu_column_totals
//   api_primitives.hip the reference's per-vector primitives in batch form
//   api_container.hip  the serialized column (blob) and the descriptor checks
//   api_host.hip       columns that live in host memory: the chunked two-stream pipelines, one or several contexts
// No codec arithmetic lives in any of them and there is no CPU path: without a gfx950 device alpgpu_ctx_create fails and nothing else can be called.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/alpgpu.h"
#include "launch.hpp"

#define ALPGPU_INTERNAL __attribute__((visibility("hidden")))

// Per-segment sizes of a column (segments of seg_vectors consecutive vectors, a multiple of 400): what a launch rule that sees more than the column's
// averages needs.  Host-side, in the context, keyed by the descriptor buffer: struct alpgpu_column stays as it is (ABI 3).
constexpr int kMaxSegments = 32;
struct SegmentTable {
	const void* key;         // col->d_vectors (nullptr: empty slot)
	const void* d_packed;    // ... and the rest of what a column must share with the one the sums were taken from: a caching allocator hands the same
	uint64_t    n_vectors;   //     descriptor buffer to the next column of the same length, whose stream sizes then differ
	uint64_t    packed_bytes, exc_bytes;
	uint64_t    seg_vectors;
	uint32_t    n_seg;
	uint64_t    packed[kMaxSegments];  // bytes of packed records
	uint64_t    exc_cnt[kMaxSegments]; // exceptions
	uint64_t    rd_vectors[kMaxSegments]; // vectors of ALP_RD rowgroups
};

// What an UNHINTED decode learned about a column (api_decode.hip: decode_unhinted): the per-segment sums taken on the stream travel to page-locked host words behind
// an event that is only ever queried; once there, the next decode of the same column is planned on the host.  state 0: empty, 1: copy in flight, 2: sizes known.
constexpr int kLearnSlots = 4;
struct LearnSlot {
	const void* key;       // col->d_vectors
	const void* d_packed;
	uint64_t    n_vectors;
	int         value_bytes;
	uint32_t    n_seg;
	int         state;
	hipEvent_t  ev;
	uint64_t    packed, exceptions, rd_vectors;
};

struct alpgpu_ctx {
	int         device;
	hipStream_t own_stream;
	hipStream_t stream;
	int         n_cus;
	int         decode_variant;
	int         decode_auto;     // 1: vectors per decode workgroup chosen from the column's size hints
	double      decode_four_bits, decode_four_bits_exc; // auto rule: four vectors per workgroup up to this many packed bits per value (without / with exceptions)
	int         decode_vpw;      // the value last given to ALPGPU_OPT_DECODE_VECTORS_PER_WG (0 auto, 1, 2, 4); float decode reads this
	char        name[128];
	uint64_t    hbm_bytes;
	int         encode_two_pass; // 0 (default): single-pass encode with look-back offsets; 1: analysis + scan + pack
	int         force_stall;     // debug: the single pass gives up in its look-back, the recovery route re-encodes
	int         async_init_wg_per_cu; // persistent search workgroups per CU (1; ALPGPU_ASYNC_INIT_WG_PER_CU for experiments)
	int         async_init_adaptive;  // three per CU when the column's head is mostly ALP_RD (default; ALPGPU_ASYNC_INIT_ADAPTIVE=0 for A/B runs)
	int         async_init;      // 1 (default): alpgpu_encode_* of a long column runs the rowgroup search BESIDE the vector encode (second stream)
	hipStream_t init_stream;     // ... on this stream (highest priority: its few workgroups are placed first)
	hipEvent_t  ev_fork, ev_head, ev_join;
	int         encode_kernel;   // ALPGPU_ENCODE_KERNEL_LEAN (default) / _CLASSIC
	int         encode_unordered; // ALPGPU_OPT_ENCODE_UNORDERED: tiles reserve their stream bytes with one atomic add (lean kernel, device columns only)
	int         decode_pairing;  // ALPGPU_OPT_DECODE_PAIRING: 0 auto, 1..3 -> k_decode_pairs
	int         decode_pairs_auto; // the auto rule may pick the pair kernel (ALPGPU_DECODE_PAIRS_AUTO=0 for A/B runs)
	int         decode_pad_kib;    // ALPGPU_OPT_DECODE_RESIDENCY_PAD: KiB of unused dynamic LDS per decode workgroup (-1: chosen from the column's hints)
	int         decode_patch_max;  // ALPGPU_OPT_DECODE_PATCH_AFTER: ALP vectors with 1..this many exceptions are patched after their stores (0: never; <= 64)
	int         decode_patch_shape; // 1 (default): a column whose vectors are patched picks its launch shape like a column without exceptions (ALPGPU_DECODE_PATCH_SHAPE=0 for A/B runs)
	int         read_ahead;        // ALPGPU_OPT_DECODE_READ_AHEAD: the store decode runs with a read-ahead into the Infinity Cache on the second stream (read_ahead_kernels.hip)
	int         streams_serialize; // the process runs with GPU_MAX_HW_QUEUES=1 / AMD_SERIALIZE_KERNEL / HIP_LAUNCH_BLOCKING: left to itself (-1) the library starts no read-ahead
	int         read_ahead_us;     // ... this many microseconds ahead of the decode kernel (0: 12 + 6.5 us per packed bit of the vectors, at most 60)
	int         read_ahead_grid;   // ... by this many eight-wavefront workgroups
	int         read_ahead_bits;   // ... records of vectors of at most this many packed bits per value (the descriptors of all)
	uint64_t*   d_progress;        // ... paced by this word of device memory (2 KiB; the map of its words: decode_policy.hpp)
	uint64_t    progress_gen;      // ... whose tag changes with every launch
	uint32_t    wall_tick_ps;      // picoseconds per tick of the device's wall_clock64() (the read-ahead's naps)
	int         decode_segments;   // ALPGPU_OPT_DECODE_SEGMENTS: a column whose regions differ is decoded region by region, each with its own launch shape (1, default)
	SegmentTable seg_tables[4];    // ... from the per-segment sizes alpgpu_column_totals / alpgpu_column_from_blob last saw (host-side, keyed by the descriptor buffer)
	int         seg_next;
	int         decode_unhinted;   // ALPGPU_OPT_DECODE_UNHINTED (1, default): a column whose sizes the host does not know gets them summed on the stream and its launch shape chosen on the device
	LearnSlot   learn[kLearnSlots]; // ... and what those sums said, for the next decode of the same column
	int         learn_next;
	uint64_t*   h_learn;           // page-locked: kLearnSlots x 3 x kMaxSegments words
	int         pipelined_consumer; // 1: the fused consumers through the persistent LDS-ring kernel (consume_kernels.hip; its own summation order)
	void*       workspace;       // scan workspace (tile sums / tile status words), grown on demand
	uint64_t    workspace_bytes;
	hipEvent_t  ws_event;        // recorded behind the last encode that used the workspace ...
	hipStream_t ws_stream;       // ... on this stream
	int         ws_busy;
};

namespace alpgpu_host {
extern thread_local char g_err[512];
ALPGPU_INTERNAL int fail(int code, const char* what, hipError_t e = hipSuccess);
} // namespace alpgpu_host
using alpgpu_host::fail;
using alpgpu_host::g_err;

#define ALPGPU_HIP(call)                                                                                                \
	do {                                                                                                                \
		hipError_t e_ = (call);                                                                                         \
		if (e_ != hipSuccess) { return fail(ALPGPU_ERR_HIP, #call, e_); }                                               \
	} while (0)

#define ALPGPU_CHECK_CTX(ctx)                                                                                           \
	do {                                                                                                                \
		if (!(ctx)) { return fail(ALPGPU_ERR_INVALID, "null context"); }                                                \
		ALPGPU_HIP(hipSetDevice((ctx)->device));                                                                        \
	} while (0)

#define ALPGPU_PRIM(cond_ok, call)                                                                                      \
	do {                                                                                                                \
		ALPGPU_CHECK_CTX(ctx);                                                                                          \
		if (n_vectors && !(cond_ok)) { return fail(ALPGPU_ERR_INVALID, "null pointer argument"); }                      \
		if ((call) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "kernel launch failed", hipGetLastError()); }            \
		return ALPGPU_OK;                                                                                               \
	} while (0)

// ---- helpers shared between the translation units (hidden: not part of the ABI) ----
extern "C" {
// api_context.hip: one scan / status workspace per context, ordered behind its previous user
ALPGPU_INTERNAL int ensure_workspace(alpgpu_ctx* ctx, uint64_t bytes);
ALPGPU_INTERNAL int workspace_used(alpgpu_ctx* ctx);
ALPGPU_INTERNAL int check_column(const alpgpu_column* col, uint64_t n_vectors);
// api_decode.hip: the per-segment sizes the decode's launch plan is made from (a column that is encoded again has new regions)
constexpr uint64_t kSegmentMinVectors = 32800; // a multiple of 400: runs begin on rowgroup boundaries and on even vectors
ALPGPU_INTERNAL void          segment_table_forget(alpgpu_ctx* ctx, const alpgpu_column* col);
ALPGPU_INTERNAL uint64_t      segment_vectors_for(uint64_t n_vectors);
ALPGPU_INTERNAL SegmentTable* segment_table_new(alpgpu_ctx* ctx, const alpgpu_column* col, uint64_t packed_bytes, uint64_t exc_bytes);
// api_container.hip: a serialized column's header and descriptors, checked on the host
ALPGPU_INTERNAL int  validate_blob_header(const void* h_blob, uint64_t size, uint64_t value_bytes, alpgpu_blob_header& h);
ALPGPU_INTERNAL int  validate_blob_vectors(const void* h_blob, const alpgpu_blob_header& h, uint64_t value_bytes, uint64_t v_begin, uint64_t v_end, const uint64_t* window = nullptr);
} // extern "C"
inline uint64_t align8(uint64_t x) { return (x + 7ull) & ~7ull; }
