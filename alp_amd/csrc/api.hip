// api.hip — the extern "C" surface declared in include/alpgpu.h.  Thin: argument checks, stream selection,
// kernel launches.  No codec arithmetic lives here and there is no CPU path: without a gfx950 device
// alpgpu_ctx_create fails and nothing else can be called.
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
	int         read_ahead_us;     // ... this many microseconds ahead of the decode kernel (0: 12 + 6.5 us per packed bit of the vectors, at most 60)
	int         read_ahead_grid;   // ... by this many eight-wavefront workgroups
	int         read_ahead_bits;   // ... records of vectors of at most this many packed bits per value (the descriptors of all)
	uint64_t*   d_progress;        // ... paced by this word of device memory (2 KiB: [0] the decode's position, tagged; [1] never written; [64..159] alpgpu_column_totals' segment sums)
	uint64_t    progress_gen;      // ... whose tag changes with every launch
	uint32_t    wall_tick_ps;      // picoseconds per tick of the device's wall_clock64() (the read-ahead's naps)
	int         decode_segments;   // ALPGPU_OPT_DECODE_SEGMENTS: a column whose regions differ is decoded region by region, each with its own launch shape (1, default)
	SegmentTable seg_tables[4];    // ... from the per-segment sizes alpgpu_column_totals / alpgpu_column_from_blob last saw (host-side, keyed by the descriptor buffer)
	int         seg_next;
	int         pipelined_consumer; // 1: the fused consumers through the persistent LDS-ring kernel (consume_kernels.hip; its own summation order)
	void*       workspace;       // scan workspace (tile sums / tile status words), grown on demand
	uint64_t    workspace_bytes;
	hipEvent_t  ws_event;        // recorded behind the last encode that used the workspace ...
	hipStream_t ws_stream;       // ... on this stream
	int         ws_busy;
};

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* what, hipError_t e = hipSuccess) {
	if (e != hipSuccess) {
		std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
	} else {
		std::snprintf(g_err, sizeof(g_err), "%s", what);
	}
	return code;
}

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

} // namespace

extern "C" {

static void segment_table_forget(alpgpu_ctx* ctx, const alpgpu_column* col); // (a column that is encoded again has new regions: below, with the decode's launch plan)

int alpgpu_abi_version(void) { return 3; } // 2: alpgpu_column.d_rd_order; 3: alpgpu_column.alp_rd_rowgroups_hint

const char* alpgpu_last_error(void) { return g_err; }

int alpgpu_ctx_create(int device, alpgpu_ctx** out_ctx) {
	if (!out_ctx) { return fail(ALPGPU_ERR_INVALID, "out_ctx is null"); }
	*out_ctx  = nullptr;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
		return fail(ALPGPU_ERR_NO_DEVICE, "no HIP device visible: libalpgpu has no CPU fallback");
	}
	if (device < 0 || device >= count) { return fail(ALPGPU_ERR_INVALID, "device index out of range"); }
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) { return fail(ALPGPU_ERR_NO_DEVICE, "hipGetDeviceProperties failed"); }
	if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !std::getenv("ALPGPU_ALLOW_ANY_ARCH")) {
		std::snprintf(g_err, sizeof(g_err), "device %d is %s; libalpgpu is built for gfx950 only", device, prop.gcnArchName);
		return ALPGPU_ERR_NO_DEVICE;
	}
	ALPGPU_HIP(hipSetDevice(device));
	alpgpu_ctx* ctx = new (std::nothrow) alpgpu_ctx();
	if (!ctx) { return fail(ALPGPU_ERR_INVALID, "out of host memory"); }
	ctx->device = device;
	if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
		delete ctx;
		return fail(ALPGPU_ERR_HIP, "hipStreamCreate failed");
	}
	ctx->stream         = ctx->own_stream;
	{
		int least = 0, greatest = 0;
		(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
		const char* pr = std::getenv("ALPGPU_INIT_STREAM_PRIO"); // experiments: "low" / "normal"; default: highest
		const int   prio = pr && pr[0] == 'l' ? least : (pr && pr[0] == 'n' ? 0 : greatest);
		if (hipStreamCreateWithPriority(&ctx->init_stream, hipStreamNonBlocking, prio) != hipSuccess ||
		    hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&ctx->ev_head, hipEventDisableTiming) != hipSuccess) {
			(void)hipStreamDestroy(ctx->own_stream);
			delete ctx;
			return fail(ALPGPU_ERR_HIP, "hipStreamCreate / hipEventCreate failed");
		}
	}
	ctx->async_init     = std::getenv("ALPGPU_ENCODE_SYNC_INIT") ? 0 : 1;
	ctx->async_init_wg_per_cu = std::getenv("ALPGPU_ASYNC_INIT_WG_PER_CU") ? std::atoi(std::getenv("ALPGPU_ASYNC_INIT_WG_PER_CU")) : 1;
	if (ctx->async_init_wg_per_cu < 1) { ctx->async_init_wg_per_cu = 1; }
	ctx->async_init_adaptive = std::getenv("ALPGPU_ASYNC_INIT_ADAPTIVE") ? std::atoi(std::getenv("ALPGPU_ASYNC_INIT_ADAPTIVE")) : 1;
	ctx->n_cus          = prop.multiProcessorCount;
	ctx->hbm_bytes      = prop.totalGlobalMem;
	ctx->decode_variant  = 1; // bit 0: one vector per decode workgroup, bit 1: plain stores
	ctx->decode_auto     = 1;
	ctx->decode_vpw      = 0;
	ctx->decode_four_bits     = std::getenv("ALPGPU_DECODE_FOUR_BITS") ? std::atof(std::getenv("ALPGPU_DECODE_FOUR_BITS")) : 0.0;     // (tuning runs; defaults set from the sweep)
	ctx->decode_four_bits_exc = std::getenv("ALPGPU_DECODE_FOUR_BITS_EXC") ? std::atof(std::getenv("ALPGPU_DECODE_FOUR_BITS_EXC")) : 0.0;
	ctx->encode_two_pass = std::getenv("ALPGPU_ENCODE_TWO_PASS") ? 1 : 0;
	ctx->force_stall     = 0;
	ctx->pipelined_consumer = 0;
	// (round 5: 0.  With the per-vector loops k_decode_column<2> is 10-20 % ahead of k_decode_pairs on narrow vectors with exceptions: profiles/r05_decode_exceptions.txt)
	ctx->decode_pairs_auto = std::getenv("ALPGPU_DECODE_PAIRS_AUTO") ? std::atoi(std::getenv("ALPGPU_DECODE_PAIRS_AUTO")) : 0;
	ctx->decode_pad_kib     = std::getenv("ALPGPU_DECODE_PAD_LDS_KIB") ? std::atoi(std::getenv("ALPGPU_DECODE_PAD_LDS_KIB")) & 0xFF : -1;
	// the patch arm exists in -DALPGPU_DECODE_PATCH_MODE=1 / 2 builds of decode_kernels.hip only (measured slower than the mask route: profiles/r05_decode_exceptions.txt);
	// the default build ignores the limit, and the launch rule must not count on an arm that is not there: 0 unless asked for
	ctx->decode_patch_max   = std::getenv("ALPGPU_DECODE_PATCH_AFTER") ? std::atoi(std::getenv("ALPGPU_DECODE_PATCH_AFTER")) : 0; // (A/B runs)
	if (ctx->decode_patch_max < 0 || ctx->decode_patch_max > 64) { ctx->decode_patch_max = 64; }
	ctx->decode_patch_shape = std::getenv("ALPGPU_DECODE_PATCH_SHAPE") ? std::atoi(std::getenv("ALPGPU_DECODE_PATCH_SHAPE")) : 1;
	ctx->decode_pairing  = std::getenv("ALPGPU_DECODE_PAIRING") ? (std::atoi(std::getenv("ALPGPU_DECODE_PAIRING")) & 3) : 0; // (A/B runs)
	ctx->encode_kernel   = std::getenv("ALPGPU_ENCODE_KERNEL") ? std::atoi(std::getenv("ALPGPU_ENCODE_KERNEL")) : ALPGPU_ENCODE_KERNEL_LEAN; // (A/B runs)
	ctx->encode_unordered = std::getenv("ALPGPU_ENCODE_UNORDERED") ? std::atoi(std::getenv("ALPGPU_ENCODE_UNORDERED")) : 0; // (A/B runs)
	ctx->read_ahead      = std::getenv("ALPGPU_DECODE_READ_AHEAD") ? std::atoi(std::getenv("ALPGPU_DECODE_READ_AHEAD")) : -1; // -1: by the column (read_ahead_for)
	ctx->read_ahead_us   = std::getenv("ALPGPU_READ_AHEAD_US") ? std::atoi(std::getenv("ALPGPU_READ_AHEAD_US")) : 0; // 0: by the vectors' width (alpgpu_decode_f64)
	ctx->read_ahead_grid = std::getenv("ALPGPU_READ_AHEAD_GRID") ? std::atoi(std::getenv("ALPGPU_READ_AHEAD_GRID")) : 64;
	ctx->decode_segments = std::getenv("ALPGPU_DECODE_SEGMENTS") ? std::atoi(std::getenv("ALPGPU_DECODE_SEGMENTS")) : 1;
	for (auto& t : ctx->seg_tables) { t.key = nullptr; }
	ctx->seg_next        = 0;
	ctx->read_ahead_bits = std::getenv("ALPGPU_READ_AHEAD_BITS") ? std::atoi(std::getenv("ALPGPU_READ_AHEAD_BITS")) : 128;
	ctx->d_progress      = nullptr;
	ctx->progress_gen    = 0;
	{
		int khz = 0;
		if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) { khz = 100000; } // 100 MHz: gfx9's s_memrealtime
		ctx->wall_tick_ps = static_cast<uint32_t>(1000000000ll / khz);
		if (ctx->wall_tick_ps == 0) { ctx->wall_tick_ps = 1; }
	}
	ctx->workspace       = nullptr;
	ctx->workspace_bytes = 0;
	ctx->ws_stream       = nullptr;
	ctx->ws_busy         = 0;
	if (hipEventCreateWithFlags(&ctx->ws_event, hipEventDisableTiming) != hipSuccess) {
		(void)hipStreamDestroy(ctx->own_stream);
		delete ctx;
		return fail(ALPGPU_ERR_HIP, "hipEventCreate failed");
	}
	if (hipMalloc(reinterpret_cast<void**>(&ctx->d_progress), 2048) != hipSuccess || hipMemset(ctx->d_progress, 0, 2048) != hipSuccess) {
		if (ctx->d_progress) { (void)hipFree(ctx->d_progress); }
		(void)hipEventDestroy(ctx->ws_event);
		(void)hipStreamDestroy(ctx->init_stream);
		(void)hipStreamDestroy(ctx->own_stream);
		delete ctx;
		return fail(ALPGPU_ERR_HIP, "hipMalloc of the context's progress word failed");
	}
	if (const char* v = std::getenv("ALPGPU_DECODE_VARIANT")) { // A/B runs
		ctx->decode_variant = std::atoi(v);
		ctx->decode_auto    = 0;
	}
	std::snprintf(ctx->name, sizeof(ctx->name), "%s (%s)", prop.name, prop.gcnArchName);
	*out_ctx = ctx;
	return ALPGPU_OK;
}

int alpgpu_init(int device, alpgpu_ctx** out_ctx) { return alpgpu_ctx_create(device, out_ctx); }

void alpgpu_ctx_destroy(alpgpu_ctx* ctx) {
	if (!ctx) { return; }
	(void)hipSetDevice(ctx->device);
	if (ctx->ws_busy) { (void)hipEventSynchronize(ctx->ws_event); }
	(void)hipEventDestroy(ctx->ws_event);
	(void)hipStreamSynchronize(ctx->init_stream);
	(void)hipEventDestroy(ctx->ev_fork);
	(void)hipEventDestroy(ctx->ev_head);
	(void)hipEventDestroy(ctx->ev_join);
	(void)hipStreamDestroy(ctx->init_stream);
	(void)hipStreamDestroy(ctx->own_stream);
	if (ctx->workspace) { (void)hipFree(ctx->workspace); }
	if (ctx->d_progress) { (void)hipFree(ctx->d_progress); }
	delete ctx;
}

int alpgpu_set_stream(alpgpu_ctx* ctx, void* hip_stream) {
	if (!ctx) { return fail(ALPGPU_ERR_INVALID, "null context"); }
	ctx->stream = static_cast<hipStream_t>(hip_stream); // NULL is the device's legacy default stream
	return ALPGPU_OK;
}

int alpgpu_use_own_stream(alpgpu_ctx* ctx) {
	if (!ctx) { return fail(ALPGPU_ERR_INVALID, "null context"); }
	ctx->stream = ctx->own_stream;
	return ALPGPU_OK;
}

int alpgpu_set_option(alpgpu_ctx* ctx, int option, int64_t value) {
	if (!ctx) { return fail(ALPGPU_ERR_INVALID, "null context"); }
	switch (option) {
	case ALPGPU_OPT_DECODE_VECTORS_PER_WG:
		if (value != 0 && value != 1 && value != 2 && value != 4) {
			return fail(ALPGPU_ERR_INVALID, "decode vectors per workgroup must be 0 (auto), 1, 2 or (float columns) 4");
		}
		ctx->decode_auto    = value == 0;
		ctx->decode_vpw     = static_cast<int>(value);
		ctx->decode_variant = (ctx->decode_variant & ~5) | (value >= 2 ? 0 : 1) | (value == 4 ? 4 : 0); // (4: four vectors over the narrow stage)
		return ALPGPU_OK;
	case ALPGPU_OPT_ENCODE_TWO_PASS:
		ctx->encode_two_pass = value ? 1 : 0;
		return ALPGPU_OK;
	case ALPGPU_OPT_DECODE_PLAIN_STORES:
		ctx->decode_variant = (ctx->decode_variant & ~2) | (value ? 2 : 0);
		return ALPGPU_OK;
	case ALPGPU_OPT_DEBUG_FORCE_STALL:
		ctx->force_stall = value ? 1 : 0;
		return ALPGPU_OK;
	case ALPGPU_OPT_ENCODE_ASYNC_INIT:
		if (value < 0 || value > 2) { return fail(ALPGPU_ERR_INVALID, "async init: 0 (off), 1 (double columns: default) or 2 (float columns too)"); }
		ctx->async_init = static_cast<int>(value);
		return ALPGPU_OK;
	case ALPGPU_OPT_DECODE_PAIRING:
		if (value < 0 || value > 3) { return fail(ALPGPU_ERR_INVALID, "decode pairing: 0 (off) .. 3"); }
		ctx->decode_pairing = static_cast<int>(value);
		return ALPGPU_OK;
	case ALPGPU_OPT_ENCODE_KERNEL:
		if (value != ALPGPU_ENCODE_KERNEL_LEAN && value != ALPGPU_ENCODE_KERNEL_CLASSIC) { return fail(ALPGPU_ERR_INVALID, "encode kernel: 0 (lean) or 1 (classic)"); }
		ctx->encode_kernel = static_cast<int>(value);
		return ALPGPU_OK;
	case ALPGPU_OPT_DECODE_RESIDENCY_PAD:
		if (value < -1 || value > 150) { return fail(ALPGPU_ERR_INVALID, "decode residency pad: -1 (by the library's rule) or 0..150 KiB"); }
		ctx->decode_pad_kib = static_cast<int>(value);
		return ALPGPU_OK;
	case ALPGPU_OPT_ENCODE_UNORDERED:
		ctx->encode_unordered = value ? 1 : 0;
		return ALPGPU_OK;
	case ALPGPU_OPT_DECODE_PATCH_AFTER:
		if (value < 0 || value > 64) { return fail(ALPGPU_ERR_INVALID, "decode patch-after: 0 (never) .. 64 exceptions per vector"); }
		ctx->decode_patch_max = static_cast<int>(value);
		return ALPGPU_OK;
	case ALPGPU_OPT_DECODE_READ_AHEAD:
		if (value < -1 || value > 1) { return fail(ALPGPU_ERR_INVALID, "decode read-ahead: -1 (columns of narrow vectors: the default), 0 (off) or 1 (on)"); }
		ctx->read_ahead = value;
		return ALPGPU_OK;
	case ALPGPU_OPT_DECODE_READ_AHEAD_US:
		if (value < 0 || value > 10000) { return fail(ALPGPU_ERR_INVALID, "decode read-ahead lead: 0 (by the vectors' width) or 1..10000 microseconds"); }
		ctx->read_ahead_us = value;
		return ALPGPU_OK;
	case ALPGPU_OPT_DECODE_SEGMENTS:
		if (value < 0 || value > 1) { return fail(ALPGPU_ERR_INVALID, "decode by segments: 0 (off) or 1 (on)"); }
		ctx->decode_segments = value;
		return ALPGPU_OK;
	case ALPGPU_OPT_CONSUMER_PIPELINED:
		if (value < 0 || value > 3) { return fail(ALPGPU_ERR_INVALID, "consumer kernel: 0 (chosen per column), 1 (persistent LDS-ring kernel), 2 (one wavefront per vector, no stage) or 3 (four wavefronts per vector)"); }
		ctx->pipelined_consumer = static_cast<int>(value);
		return ALPGPU_OK;
	default:
		return fail(ALPGPU_ERR_INVALID, "unknown option");
	}
}

int alpgpu_synchronize(alpgpu_ctx* ctx) {
	ALPGPU_CHECK_CTX(ctx);
	ALPGPU_HIP(hipStreamSynchronize(ctx->stream));
	return ALPGPU_OK;
}

int alpgpu_device_info(alpgpu_ctx* ctx, char* name_out, size_t name_cap, int* cu_count, uint64_t* hbm_bytes) {
	if (!ctx) { return fail(ALPGPU_ERR_INVALID, "null context"); }
	if (name_out && name_cap) { std::snprintf(name_out, name_cap, "%s", ctx->name); }
	if (cu_count) { *cu_count = ctx->n_cus; }
	if (hbm_bytes) { *hbm_bytes = ctx->hbm_bytes; }
	return ALPGPU_OK;
}

int alpgpu_malloc(alpgpu_ctx* ctx, void** d_ptr, size_t bytes) {
	ALPGPU_CHECK_CTX(ctx);
	if (!d_ptr) { return fail(ALPGPU_ERR_INVALID, "d_ptr is null"); }
	ALPGPU_HIP(hipMalloc(d_ptr, bytes ? bytes : 8));
	return ALPGPU_OK;
}

int alpgpu_free(alpgpu_ctx* ctx, void* d_ptr) {
	ALPGPU_CHECK_CTX(ctx);
	ALPGPU_HIP(hipFree(d_ptr));
	return ALPGPU_OK;
}

int alpgpu_memcpy_h2d(alpgpu_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
	ALPGPU_CHECK_CTX(ctx);
	ALPGPU_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
	ALPGPU_HIP(hipStreamSynchronize(ctx->stream));
	return ALPGPU_OK;
}

int alpgpu_memcpy_d2h(alpgpu_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
	ALPGPU_CHECK_CTX(ctx);
	ALPGPU_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
	ALPGPU_HIP(hipStreamSynchronize(ctx->stream));
	return ALPGPU_OK;
}

// page-locked host memory: copies to and from it are DMA transfers the runtime need not stage, and may be left asynchronous
int alpgpu_malloc_host(alpgpu_ctx* ctx, void** h_ptr, size_t bytes) {
	ALPGPU_CHECK_CTX(ctx);
	if (!h_ptr) { return fail(ALPGPU_ERR_INVALID, "h_ptr is null"); }
	ALPGPU_HIP(hipHostMalloc(h_ptr, bytes ? bytes : 8, hipHostMallocDefault));
	return ALPGPU_OK;
}

int alpgpu_free_host(alpgpu_ctx* ctx, void* h_ptr) {
	ALPGPU_CHECK_CTX(ctx);
	ALPGPU_HIP(hipHostFree(h_ptr));
	return ALPGPU_OK;
}

// enqueued on the context's stream, NOT waited for: h_src must stay untouched until a later synchronous call on this context returns
int alpgpu_memcpy_h2d_async(alpgpu_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
	ALPGPU_CHECK_CTX(ctx);
	ALPGPU_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
	return ALPGPU_OK;
}

int alpgpu_memset(alpgpu_ctx* ctx, void* d_dst, int value, size_t bytes) {
	ALPGPU_CHECK_CTX(ctx);
	ALPGPU_HIP(hipMemsetAsync(d_dst, value, bytes, ctx->stream));
	return ALPGPU_OK;
}

// worst case per vector: ALP bw=64 -> 8192 B; ALP_RD rbw=63,lbw=3 -> 8448 B.  +1 KiB slack at the end.
uint64_t alpgpu_packed_capacity(uint64_t n_vectors) { return n_vectors * 8448ull + 1024ull; }
// worst case per vector: 1024 exceptions x (8 B value + 2 B position)
uint64_t alpgpu_exc_capacity(uint64_t n_vectors) { return n_vectors * 10240ull + 64ull; }

// One scan / status workspace per context.  Work that uses it is ordered behind the previous user: same stream = stream
// order; another stream (alpgpu_set_stream between two encodes, e.g. a torch stream switch) waits on the event recorded
// behind the previous encode; growing the buffer waits for that event on the host before the old buffer is freed.
static int ensure_workspace(alpgpu_ctx* ctx, uint64_t bytes) {
	if (ctx->ws_busy && ctx->ws_stream != ctx->stream) { ALPGPU_HIP(hipStreamWaitEvent(ctx->stream, ctx->ws_event, 0)); }
	if (ctx->workspace_bytes >= bytes) { return ALPGPU_OK; }
	if (ctx->ws_busy) { ALPGPU_HIP(hipEventSynchronize(ctx->ws_event)); }
	ctx->ws_busy = 0;
	if (ctx->workspace) { ALPGPU_HIP(hipFree(ctx->workspace)); }
	ctx->workspace       = nullptr;
	ctx->workspace_bytes = 0;
	const uint64_t want  = bytes < (1ull << 20) ? (1ull << 20) : bytes * 2;
	ALPGPU_HIP(hipMalloc(&ctx->workspace, want));
	ctx->workspace_bytes = want;
	return ALPGPU_OK;
}
static int workspace_used(alpgpu_ctx* ctx) {
	ALPGPU_HIP(hipEventRecord(ctx->ws_event, ctx->stream));
	ctx->ws_stream = ctx->stream;
	ctx->ws_busy   = 1;
	return ALPGPU_OK;
}

static int check_column(const alpgpu_column* col, uint64_t n_vectors) {
	if (!col) { return fail(ALPGPU_ERR_INVALID, "null column"); }
	if (col->n_vectors != n_vectors) { return fail(ALPGPU_ERR_INVALID, "column.n_vectors does not match n_vectors"); }
	if (col->n_rowgroups != (n_vectors + 99) / 100) { return fail(ALPGPU_ERR_INVALID, "column.n_rowgroups must be ceil(n_vectors/100)"); }
	if (n_vectors && (!col->d_rowgroups || !col->d_vectors || !col->d_packed || !col->d_exc || !col->d_totals)) {
		return fail(ALPGPU_ERR_INVALID, "column buffers must be allocated by the caller");
	}
	return ALPGPU_OK; // (an empty column may have no buffers at all: every entry point returns early for it)
}

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
constexpr double   kReadAheadBits    = 6.5; // without exceptions
constexpr double   kReadAheadBitsExc = 7.5; // with exceptions
constexpr uint64_t kReadAheadVectors = 262144; // shorter columns: the cold start and the join of the second stream eat the gain (cold columns: even at 131072 vectors)
static bool read_ahead_for(const alpgpu_ctx* ctx, const alpgpu_column* col) {
	if (ctx->read_ahead == 0 || col->packed_bytes_hint == 0 || ctx->d_progress == nullptr) { return false; }
	if (ctx->read_ahead > 0) { return col->n_vectors >= 32768; }
	const double limit = column_decodes_with_exceptions(ctx, col) ? kReadAheadBitsExc : kReadAheadBits;
	return col->n_vectors >= kReadAheadVectors && static_cast<double>(col->packed_bytes_hint) <= limit * 128.0 * static_cast<double>(col->n_vectors);
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
		const bool   narrow   = bits <= (with_exc ? 22.0 : 17.5); // (17.5: with the residency caps below two vectors per workgroup win through 17 bits; with exceptions through 22: round 5)
		const double four_max = with_exc ? ctx->decode_four_bits_exc : ctx->decode_four_bits; // (0 = never: the four-vector shape lost at every width, it is chosen by tuning runs only)
		const bool   four     = four_max > 0.0 && bits <= four_max;
		variant               = (variant & ~5) | ((hinted && narrow) ? 0 : 1) | ((hinted && four) ? 4 : 0);
		// narrow vectors under the read-ahead: their reads hit the Infinity Cache, and ONE vector per workgroup — the shape that suffers most from the two round
		// trips (0.53 at 2-6 bits) — becomes the best one (0.75-0.80); with exceptions two per workgroup stay ahead (0.68-0.74 against 0.64-0.68)
		if (ctx->read_ahead < 0 && read_ahead_for(ctx, col) && !with_exc) { variant = (variant & ~5) | 1; }
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
	return (variant & 7) | (pairing << 3) | (pad_kib << 8);
}

// ---- a launch rule that sees more than the column's averages (round 5) -------------------------------------------------------------------------
// alpgpu_column_totals and alpgpu_column_from_blob record, per segment of the column, what they record for the whole: packed bytes, exceptions, ALP_RD
// vectors.  alpgpu_decode_f64 of that column (same context, same descriptor buffer) merges adjacent segments of the same KIND — by packed width: up to the
// read-ahead's 7 bits / up to the two-vectors-per-workgroup limit / below 38 bits / beyond; with or without exceptions — into runs and decodes run by run,
// each through the rule above with the run's own sizes: a column whose first half is 6-bit vectors with exceptions and whose second half is 44-bit vectors
// (bench.py: decode_bimodal) gets two vectors per workgroup + the read-ahead for the first and one per workgroup, six workgroups per CU, for the second,
// instead of the shape of their average.  One kind, or more than kMaxRuns runs (a column that changes every few thousand vectors is served by its average): the
// whole column in one launch, as before.  ALPGPU_OPT_DECODE_SEGMENTS = 0: never.  Launch shapes only: the bytes cannot differ.
constexpr uint64_t kSegmentMinVectors = 32800; // a multiple of 400: runs begin on rowgroup boundaries and on even vectors
constexpr int      kMaxRuns           = 8;
struct DecodeRun {
	uint64_t v0, n, packed, exc_bytes, rd_vectors;
};

static uint64_t segment_vectors_for(uint64_t n_vectors) {
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
static void segment_table_forget(alpgpu_ctx* ctx, const alpgpu_column* col) {
	if (!ctx || !col) { return; }
	for (auto& t : ctx->seg_tables) {
		if (t.key == col->d_vectors) { t.key = nullptr; }
	}
}
// (packed_bytes / exc_bytes: the stream sizes the caller is about to write into the column's hints)
static SegmentTable* segment_table_new(alpgpu_ctx* ctx, const alpgpu_column* col, uint64_t packed_bytes, uint64_t exc_bytes) {
	segment_table_forget(ctx, col);
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

// the kind of a stretch of vectors, from its sums (see above)
static int stretch_kind(const alpgpu_ctx* ctx, uint64_t n, uint64_t packed, uint64_t exc_bytes) {
	alpgpu_column v {};
	v.n_vectors = n, v.packed_bytes_hint = packed ? packed : 1, v.exc_bytes_hint = exc_bytes;
	const bool   with_exc = column_decodes_with_exceptions(ctx, &v);
	const double bits     = static_cast<double>(packed) / (128.0 * static_cast<double>(n));
	const int    band     = bits <= (with_exc ? kReadAheadBitsExc : kReadAheadBits) ? 0 : (bits <= (with_exc ? 22.0 : 17.5) ? 1 : (bits < 38.0 ? 2 : 3));
	return 2 * band + (with_exc ? 1 : 0);
}

// runs[0 .. return) cover the column; 0 = no plan (decode the column whole)
static int plan_decode_runs(alpgpu_ctx* ctx, const alpgpu_column* col, DecodeRun* runs) {
	if (!ctx->decode_segments || !ctx->decode_auto || ctx->decode_pad_kib >= 0 || ctx->decode_pairing != 0) { return 0; } // (a forced shape is a forced shape)
	const SegmentTable* t = segment_table_of(ctx, col);
	if (!t || t->n_seg < 2) { return 0; }
	int n_runs = 0, kind = -1;
	for (uint32_t s = 0; s < t->n_seg; ++s) {
		const uint64_t v0 = s * t->seg_vectors;
		const uint64_t n  = v0 + t->seg_vectors < t->n_vectors ? t->seg_vectors : t->n_vectors - v0;
		const uint64_t eb = 10ull * t->exc_cnt[s]; // (ALP: 8-byte value + 2-byte position; ALP_RD records are smaller and their vectors wide anyway)
		const int      k  = stretch_kind(ctx, n, t->packed[s], eb);
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
	if (is_f32) { return ctx->decode_vpw ? ctx->decode_vpw : 2; }
	const int variant = decode_variant_for(ctx, col);
	if ((variant >> 3) & 3) { return 2; } // (the pair kernel: two vectors per workgroup, run together or one after the other)
	return (variant & 4) ? 4 : ((variant & 1) ? 1 : 2);
}

// ... and whether it would start the read-ahead beside the decode kernel (ALPGPU_OPT_DECODE_READ_AHEAD): 1 / 0; negative on bad arguments
int alpgpu_decode_reads_ahead(alpgpu_ctx* ctx, const alpgpu_column* col, int is_f32) {
	if (!ctx || !col) { return fail(ALPGPU_ERR_INVALID, "null context or column"); }
	return (!is_f32 && read_ahead_for(ctx, col)) ? 1 : 0;
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

// one launch of the store decode over a column or a run of it (+ the read-ahead beside it where the rule wants one)
static int decode_one_f64(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_out) {
	const int variant = decode_variant_for(ctx, col);
	// The read-ahead (read_ahead_kernels.hip): a few persistent workgroups on the context's second stream pull the column's streams into the Infinity
	// Cache a bounded distance ahead of the decode kernel, which tells them where it is.  Started first so that it is ahead from the first workgroup on.
	const bool ahead = read_ahead_for(ctx, col);
	uint64_t   tag   = 0;
	if (ahead) {
		ctx->progress_gen = (ctx->progress_gen + 1) & 0xFFFFFFull;
		if (ctx->progress_gen == 0) { ctx->progress_gen = 1; }
		tag = ctx->progress_gen << 40;
		// The lead is a TIME (ALPGPU_OPT_DECODE_READ_AHEAD_US): what the read-ahead brings into the Infinity Cache stays there for some tens of
		// microseconds only (the decode's own stores stream through it), and it has to be there before the decode asks.  In vectors: that time at the rate of
		// a decode running at the full HBM bandwidth (an upper bound of the true rate: the read-ahead's naps by it never overshoot).
		const double   n        = static_cast<double>(col->n_vectors);
		const double   per_vec  = (static_cast<double>(col->packed_bytes_hint) + static_cast<double>(col->exc_bytes_hint)) / n + 32.0;
		const double   ps_vec   = (8192.0 + per_vec) / 8.0;                                    // picoseconds per vector at 8 TB/s
		// ... how long: by the vectors' width unless set — the wider the vectors, the longer a read-ahead workgroup's round takes.  Best leads measured per width
		// (calls 47, 49): 15 / 20 / 30 / 30 / 35 / 40 / 50 us at 1 .. 7 bits, 50-70 us at 8-11, with and without exceptions; too short falls off a cliff whose place
		// moves a little from box to box, too long decays slowly: a little above the optimum, 12 + 6.5 us per bit, at most 60.
		const double   bits     = static_cast<double>(col->packed_bytes_hint) / (128.0 * n);
		const double   lead_by_width = 12.0 + 6.5 * bits;
		const double   lead_us  = ctx->read_ahead_us > 0 ? static_cast<double>(ctx->read_ahead_us) : (lead_by_width > 60.0 ? 60.0 : lead_by_width);
		const double   lead     = lead_us * 1.0e6 / ps_vec * 0.78; // ... vectors per lead time at the decode's usual 0.78 of the full rate
		const uint32_t lead_max = static_cast<uint32_t>(lead < 4096.0 ? 4096.0 : (lead > 4.0e9 ? 4.0e9 : lead));
		const uint32_t lead_min = 2048; // about what is resident when a workgroup reports: those vectors' reads are under way
		ALPGPU_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
		ALPGPU_HIP(hipStreamWaitEvent(ctx->init_stream, ctx->ev_fork, 0));
		if (alpgpu::launch_read_ahead(ctx->init_stream, col, 8, ctx->d_progress, tag, lead_min, lead_max, static_cast<uint32_t>(ps_vec), ctx->wall_tick_ps, static_cast<uint32_t>(ctx->read_ahead_bits), ctx->read_ahead_grid) != ALPGPU_OK) {
			return fail(ALPGPU_ERR_HIP, "read-ahead launch failed", hipGetLastError());
		}
		ALPGPU_HIP(hipEventRecord(ctx->ev_join, ctx->init_stream));
	}
	const int rc = alpgpu::launch_decode_column(ctx->stream, col, d_out, variant, ctx->n_cus, static_cast<uint32_t>(ctx->decode_patch_max), ahead ? ctx->d_progress : nullptr, tag);
	if (ahead) { ALPGPU_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0)); } // (the read-ahead leaves on its own once its last batch is in reach or the decode never shows up)
	if (rc != ALPGPU_OK) { return fail(rc, "decode launch failed", hipGetLastError()); }
	return ALPGPU_OK;
}

int alpgpu_decode_f64(alpgpu_ctx* ctx, const alpgpu_column* col, double* d_out) {
	ALPGPU_CHECK_CTX(ctx);
	if (!col || (!d_out && col->n_vectors)) { return fail(ALPGPU_ERR_INVALID, "null column or output"); }
	if (col->n_vectors == 0) { return ALPGPU_OK; }
	if (col->n_vectors && (!col->d_vectors || !col->d_rowgroups)) { return fail(ALPGPU_ERR_INVALID, "column has no descriptors"); }
	DecodeRun runs[kMaxRuns];
	const int n_runs = plan_decode_runs(ctx, col, runs);
	if (n_runs == 0) { return decode_one_f64(ctx, col, d_out); }
	for (int i = 0; i < n_runs; ++i) { // regions of different kinds, each with its own launch shape (plan_decode_runs)
		const alpgpu_column view = run_view(col, runs[i]);
		if (const int rc = decode_one_f64(ctx, &view, d_out + runs[i].v0 * 1024)) { return rc; }
	}
	return ALPGPU_OK;
}

// how many launches alpgpu_decode_f64 would make for this column now: 1, or the number of runs of plan_decode_runs; negative on bad arguments
int alpgpu_decode_runs(alpgpu_ctx* ctx, const alpgpu_column* col) {
	if (!ctx || !col) { return fail(ALPGPU_ERR_INVALID, "null context or column"); }
	DecodeRun runs[kMaxRuns];
	const int n_runs = plan_decode_runs(ctx, col, runs);
	return n_runs == 0 ? 1 : n_runs;
}

// ---- vector primitives on batches ---------------------------------------------------------------------------
#define ALPGPU_PRIM(cond_ok, call)                                                                                      \
	do {                                                                                                                \
		ALPGPU_CHECK_CTX(ctx);                                                                                          \
		if (n_vectors && !(cond_ok)) { return fail(ALPGPU_ERR_INVALID, "null pointer argument"); }                      \
		if ((call) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "kernel launch failed", hipGetLastError()); }            \
		return ALPGPU_OK;                                                                                               \
	} while (0)

int alpgpu_ffor_i64(alpgpu_ctx* ctx, const int64_t* d_in, int64_t* d_packed, size_t packed_stride, const uint8_t* d_bw,
                    const int64_t* d_base, uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_packed && d_bw && d_base,
	            alpgpu::launch_ffor_i64(ctx->stream, ctx->n_cus, d_in, d_packed, packed_stride, d_bw, d_base, n_vectors));
}
int alpgpu_unffor_i64(alpgpu_ctx* ctx, const int64_t* d_packed, size_t packed_stride, int64_t* d_out, const uint8_t* d_bw,
                      const int64_t* d_base, uint64_t n_vectors) {
	ALPGPU_PRIM(d_packed && d_out && d_bw && d_base,
	            alpgpu::launch_unffor_i64(ctx->stream, ctx->n_cus, d_packed, packed_stride, d_out, d_bw, d_base, n_vectors));
}
int alpgpu_ffor_u16(alpgpu_ctx* ctx, const uint16_t* d_in, uint16_t* d_packed, size_t packed_stride, const uint8_t* d_bw,
                    const uint16_t* d_base, uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_packed && d_bw,
	            alpgpu::launch_ffor_u16(ctx->stream, ctx->n_cus, d_in, d_packed, packed_stride, d_bw, d_base, n_vectors));
}
int alpgpu_unffor_u16(alpgpu_ctx* ctx, const uint16_t* d_packed, size_t packed_stride, uint16_t* d_out, const uint8_t* d_bw,
                      const uint16_t* d_base, uint64_t n_vectors) {
	ALPGPU_PRIM(d_packed && d_out && d_bw,
	            alpgpu::launch_unffor_u16(ctx->stream, ctx->n_cus, d_packed, packed_stride, d_out, d_bw, d_base, n_vectors));
}
int alpgpu_ffor_u8(alpgpu_ctx* ctx, const uint8_t* d_in, uint8_t* d_packed, size_t packed_stride, const uint8_t* d_bw, const uint8_t* d_base,
                   uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_packed && d_bw, alpgpu::launch_ffor_u8(ctx->stream, ctx->n_cus, d_in, d_packed, packed_stride, d_bw, d_base, n_vectors));
}
int alpgpu_unffor_u8(alpgpu_ctx* ctx, const uint8_t* d_packed, size_t packed_stride, uint8_t* d_out, const uint8_t* d_bw, const uint8_t* d_base,
                     uint64_t n_vectors) {
	ALPGPU_PRIM(d_packed && d_out && d_bw, alpgpu::launch_unffor_u8(ctx->stream, ctx->n_cus, d_packed, packed_stride, d_out, d_bw, d_base, n_vectors));
}
int alpgpu_falp_f64(alpgpu_ctx* ctx, const int64_t* d_packed, size_t packed_stride, double* d_out, const uint8_t* d_bw,
                    const int64_t* d_base, const uint8_t* d_fac, const uint8_t* d_exp, uint64_t n_vectors) {
	ALPGPU_PRIM(d_packed && d_out && d_bw && d_base && d_fac && d_exp,
	            alpgpu::launch_falp(ctx->stream, ctx->n_cus, d_packed, packed_stride, d_out, d_bw, d_base, d_fac, d_exp, n_vectors));
}
int alpgpu_decode_values_f64(alpgpu_ctx* ctx, const int64_t* d_enc, double* d_out, const uint8_t* d_fac, const uint8_t* d_exp,
                             uint64_t n_vectors) {
	ALPGPU_PRIM(d_enc && d_out && d_fac && d_exp, alpgpu::launch_decode_values(ctx->stream, ctx->n_cus, d_enc, d_out, d_fac, d_exp, n_vectors));
}
int alpgpu_patch_f64(alpgpu_ctx* ctx, double* d_out, const double* d_exc, const uint16_t* d_pos, size_t exc_stride,
                     const uint16_t* d_cnt, uint64_t n_vectors) {
	ALPGPU_PRIM(d_out && d_exc && d_pos && d_cnt, alpgpu::launch_patch(ctx->stream, ctx->n_cus, d_out, d_exc, d_pos, exc_stride, d_cnt, n_vectors));
}
int alpgpu_encode_simdized_f64(alpgpu_ctx* ctx, const double* d_in, double* d_exc, uint16_t* d_pos, size_t exc_stride,
                               uint16_t* d_cnt, int64_t* d_enc, const uint8_t* d_fac, const uint8_t* d_exp, uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_exc && d_pos && d_cnt && d_enc && d_fac && d_exp,
	            alpgpu::launch_encode_simdized(ctx->stream, ctx->n_cus, d_in, d_exc, d_pos, exc_stride, d_cnt, d_enc, d_fac, d_exp, n_vectors));
}
int alpgpu_encode_values_f64(alpgpu_ctx* ctx, const double* d_in, const alpgpu_rowgroup_state* d_states, const uint32_t* d_state_idx,
                             double* d_exc, uint16_t* d_pos, size_t exc_stride, uint16_t* d_cnt, int64_t* d_enc, uint8_t* d_fac,
                             uint8_t* d_exp, uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_states && d_exc && d_pos && d_cnt && d_enc && d_fac && d_exp,
	            alpgpu::launch_encode_values(ctx->stream, ctx->n_cus, d_in, d_states, d_state_idx, d_exc, d_pos, exc_stride, d_cnt, d_enc,
	                                         d_fac, d_exp, n_vectors));
}
int alpgpu_encode_value_f64(alpgpu_ctx* ctx, const double* d_in, int64_t* d_enc, uint8_t fac, uint8_t exp, int safe, uint64_t n_values) {
	ALPGPU_CHECK_CTX(ctx);
	if (n_values && (!d_in || !d_enc)) { return fail(ALPGPU_ERR_INVALID, "null pointer argument"); }
	if (exp > 18 || fac > exp) { return fail(ALPGPU_ERR_INVALID, "factor / exponent out of range (0 <= factor <= exponent <= 18)"); }
	if (alpgpu::launch_encode_value(ctx->stream, d_in, d_enc, fac, exp, safe, n_values) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "kernel launch failed", hipGetLastError()); }
	return ALPGPU_OK;
}
int alpgpu_encode_value_f32(alpgpu_ctx* ctx, const float* d_in, int32_t* d_enc, uint8_t fac, uint8_t exp, int safe, uint64_t n_values) {
	(void)safe; // the reference's float SAFE branch does not exist as built (include/alpgpu.h, single precision)
	ALPGPU_CHECK_CTX(ctx);
	if (n_values && (!d_in || !d_enc)) { return fail(ALPGPU_ERR_INVALID, "null pointer argument"); }
	if (exp > 10 || fac > exp) { return fail(ALPGPU_ERR_INVALID, "factor / exponent out of range (0 <= factor <= exponent <= 10)"); }
	if (alpgpu::launch_encode_value_f32(ctx->stream, d_in, d_enc, fac, exp, n_values) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "kernel launch failed", hipGetLastError()); }
	return ALPGPU_OK;
}
int alpgpu_analyze_ffor_i64(alpgpu_ctx* ctx, const int64_t* d_enc, uint8_t* d_bw, int64_t* d_base, uint64_t n_vectors) {
	ALPGPU_PRIM(d_enc && d_bw && d_base, alpgpu::launch_analyze_ffor(ctx->stream, ctx->n_cus, d_enc, d_bw, d_base, n_vectors));
}
int alpgpu_rd_encode_vectors_f64(alpgpu_ctx* ctx, const double* d_in, const alpgpu_rowgroup_state* d_states,
                                 const uint32_t* d_state_idx, uint16_t* d_exc, uint16_t* d_pos, size_t exc_stride, uint16_t* d_cnt,
                                 uint64_t* d_right, uint16_t* d_left, uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_states && d_exc && d_pos && d_cnt && d_right && d_left,
	            alpgpu::launch_rd_encode(ctx->stream, ctx->n_cus, d_in, d_states, d_state_idx, d_exc, d_pos, exc_stride, d_cnt, d_right,
	                                     d_left, n_vectors));
}
int alpgpu_rd_decode_vectors_f64(alpgpu_ctx* ctx, double* d_out, const uint64_t* d_right, const uint16_t* d_left,
                                 const alpgpu_rowgroup_state* d_states, const uint32_t* d_state_idx, const uint16_t* d_exc,
                                 const uint16_t* d_pos, size_t exc_stride, const uint16_t* d_cnt, uint64_t n_vectors) {
	ALPGPU_PRIM(d_out && d_right && d_left && d_states && d_exc && d_pos && d_cnt,
	            alpgpu::launch_rd_decode(ctx->stream, ctx->n_cus, d_out, d_right, d_left, d_states, d_state_idx, d_exc, d_pos, exc_stride,
	                                     d_cnt, n_vectors));
}

int alpgpu_rd_encode_f64(alpgpu_ctx* ctx, const double* d_in, const alpgpu_rowgroup_state* d_states, const uint32_t* d_state_idx, uint16_t* d_exc,
                         uint16_t* d_pos, size_t exc_stride, uint16_t* d_cnt, uint64_t* d_right, uint16_t* d_left, uint64_t n_vectors) {
	return alpgpu_rd_encode_vectors_f64(ctx, d_in, d_states, d_state_idx, d_exc, d_pos, exc_stride, d_cnt, d_right, d_left, n_vectors);
}
int alpgpu_rd_decode_f64(alpgpu_ctx* ctx, double* d_out, const uint64_t* d_right, const uint16_t* d_left, const alpgpu_rowgroup_state* d_states,
                         const uint32_t* d_state_idx, const uint16_t* d_exc, const uint16_t* d_pos, size_t exc_stride, const uint16_t* d_cnt,
                         uint64_t n_vectors) {
	return alpgpu_rd_decode_vectors_f64(ctx, d_out, d_right, d_left, d_states, d_state_idx, d_exc, d_pos, exc_stride, d_cnt, n_vectors);
}

// ---- tail padding + blob container ---------------------------------------------------------------------------------
int alpgpu_pad_tail_f64(alpgpu_ctx* ctx, double* d_in, uint64_t n_values) {
	ALPGPU_CHECK_CTX(ctx);
	if (!d_in && n_values) { return fail(ALPGPU_ERR_INVALID, "null input"); }
	if (alpgpu::launch_pad_tail(ctx->stream, d_in, n_values) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "pad launch failed", hipGetLastError()); }
	return ALPGPU_OK;
}

static uint64_t align8(uint64_t x) { return (x + 7ull) & ~7ull; }

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
static int validate_blob_header(const void* h_blob, uint64_t size, uint64_t value_bytes, alpgpu_blob_header& h) {
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
static int validate_blob_vectors(const void* h_blob, const alpgpu_blob_header& h, uint64_t value_bytes, uint64_t v_begin, uint64_t v_end,
                                 const uint64_t* window = nullptr) {
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

int alpgpu_column_totals(alpgpu_ctx* ctx, alpgpu_column* col, uint64_t* packed_bytes, uint64_t* exc_bytes, int* overflow) {
	ALPGPU_CHECK_CTX(ctx);
	if (!col) { return fail(ALPGPU_ERR_INVALID, "null column"); }
	uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	if (col->d_totals) {
		const bool count_rd = col->d_rowgroups != nullptr && col->n_rowgroups != 0;
		if (count_rd && alpgpu::launch_count_rd_rowgroups(ctx->stream, col, col->d_totals + 7) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "rowgroup count launch failed", hipGetLastError()); }
		// per-segment sums for the decode's launch plan (plan_decode_runs): columns long enough to have two segments
		SegmentTable* seg = nullptr;
		uint64_t      seg_sums[3 * kMaxSegments];
		segment_table_forget(ctx, col);
		if (col->d_vectors && ctx->d_progress && col->n_vectors >= 2 * kSegmentMinVectors) {
			seg = segment_table_new(ctx, col, 0, 0); // (the sizes: below, once they are here)
			if (alpgpu::launch_segment_sums(ctx->stream, col, seg->seg_vectors, seg->n_seg, ctx->d_progress + 64) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "segment sums launch failed", hipGetLastError()); }
			ALPGPU_HIP(hipMemcpyAsync(seg_sums, ctx->d_progress + 64, 24ull * seg->n_seg, hipMemcpyDeviceToHost, ctx->stream));
		}
		ALPGPU_HIP(hipMemcpyAsync(t, col->d_totals, sizeof(t), hipMemcpyDeviceToHost, ctx->stream));
		ALPGPU_HIP(hipStreamSynchronize(ctx->stream));
		if (seg) {
			for (uint32_t i = 0; i < seg->n_seg; ++i) { seg->packed[i] = seg_sums[3 * i], seg->exc_cnt[i] = seg_sums[3 * i + 1], seg->rd_vectors[i] = seg_sums[3 * i + 2]; }
			seg->packed_bytes = t[0], seg->exc_bytes = t[1];
			if (t[2] || t[3]) { seg->key = nullptr; } // an overflowed or unrecovered column: no plans
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

// ==== single precision =================================================================================================
// worst case per vector: ALP bw=32 -> 4096 B; ALP_RD rbw=31,lbw=3 -> 4352 B.  +1 KiB slack at the end.
uint64_t alpgpu_packed_capacity_f32(uint64_t n_vectors) { return n_vectors * 4352ull + 1024ull; }
// worst case per vector: 1024 exceptions x (4 B value + 2 B position)
uint64_t alpgpu_exc_capacity_f32(uint64_t n_vectors) { return n_vectors * 6144ull + 64ull; }

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

int alpgpu_decode_f32(alpgpu_ctx* ctx, const alpgpu_column* col, float* d_out) {
	ALPGPU_CHECK_CTX(ctx);
	if (!col || (!d_out && col->n_vectors)) { return fail(ALPGPU_ERR_INVALID, "null column or output"); }
	if (col->n_vectors == 0) { return ALPGPU_OK; }
	if (!col->d_vectors || !col->d_rowgroups) { return fail(ALPGPU_ERR_INVALID, "column has no descriptors"); }
	const int vpw = ctx->decode_vpw ? ctx->decode_vpw : 2; // a float vector is 4 KiB: two per workgroup = the bytes of one double vector
	if (alpgpu::launch_decode_column_f32(ctx->stream, col, d_out, vpw, (ctx->decode_variant & 2) != 0) != ALPGPU_OK) {
		return fail(ALPGPU_ERR_HIP, "decode launch failed", hipGetLastError());
	}
	return ALPGPU_OK;
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

int alpgpu_pad_tail_f32(alpgpu_ctx* ctx, float* d_in, uint64_t n_values) {
	ALPGPU_CHECK_CTX(ctx);
	if (!d_in && n_values) { return fail(ALPGPU_ERR_INVALID, "null input"); }
	if (alpgpu::launch_pad_tail_f32(ctx->stream, d_in, n_values) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "pad launch failed", hipGetLastError()); }
	return ALPGPU_OK;
}

int alpgpu_ffor_i32(alpgpu_ctx* ctx, const int32_t* d_in, int32_t* d_packed, size_t packed_stride, const uint8_t* d_bw, const int32_t* d_base,
                    uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_packed && d_bw && d_base, alpgpu::launch_ffor_i32(ctx->stream, ctx->n_cus, d_in, d_packed, packed_stride, d_bw, d_base, n_vectors));
}
int alpgpu_unffor_i32(alpgpu_ctx* ctx, const int32_t* d_packed, size_t packed_stride, int32_t* d_out, const uint8_t* d_bw, const int32_t* d_base,
                      uint64_t n_vectors) {
	ALPGPU_PRIM(d_packed && d_out && d_bw && d_base, alpgpu::launch_unffor_i32(ctx->stream, ctx->n_cus, d_packed, packed_stride, d_out, d_bw, d_base, n_vectors));
}
int alpgpu_falp_f32(alpgpu_ctx* ctx, const int32_t* d_packed, size_t packed_stride, float* d_out, const uint8_t* d_bw, const int32_t* d_base,
                    const uint8_t* d_fac, const uint8_t* d_exp, uint64_t n_vectors) {
	ALPGPU_PRIM(d_packed && d_out && d_bw && d_base && d_fac && d_exp,
	            alpgpu::launch_falp_f32(ctx->stream, ctx->n_cus, d_packed, packed_stride, d_out, d_bw, d_base, d_fac, d_exp, n_vectors));
}
int alpgpu_decode_values_f32(alpgpu_ctx* ctx, const int32_t* d_enc, float* d_out, const uint8_t* d_fac, const uint8_t* d_exp, uint64_t n_vectors) {
	ALPGPU_PRIM(d_enc && d_out && d_fac && d_exp, alpgpu::launch_decode_values_f32(ctx->stream, ctx->n_cus, d_enc, d_out, d_fac, d_exp, n_vectors));
}
int alpgpu_patch_f32(alpgpu_ctx* ctx, float* d_out, const float* d_exc, const uint16_t* d_pos, size_t exc_stride, const uint16_t* d_cnt,
                     uint64_t n_vectors) {
	ALPGPU_PRIM(d_out && d_exc && d_pos && d_cnt, alpgpu::launch_patch_f32(ctx->stream, ctx->n_cus, d_out, d_exc, d_pos, exc_stride, d_cnt, n_vectors));
}
int alpgpu_encode_simdized_f32(alpgpu_ctx* ctx, const float* d_in, float* d_exc, uint16_t* d_pos, size_t exc_stride, uint16_t* d_cnt, int32_t* d_enc,
                               const uint8_t* d_fac, const uint8_t* d_exp, uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_exc && d_pos && d_cnt && d_enc && d_fac && d_exp,
	            alpgpu::launch_encode_simdized_f32(ctx->stream, ctx->n_cus, d_in, d_exc, d_pos, exc_stride, d_cnt, d_enc, d_fac, d_exp, n_vectors));
}
int alpgpu_encode_values_f32(alpgpu_ctx* ctx, const float* d_in, const alpgpu_rowgroup_state* d_states, const uint32_t* d_state_idx, float* d_exc,
                             uint16_t* d_pos, size_t exc_stride, uint16_t* d_cnt, int32_t* d_enc, uint8_t* d_fac, uint8_t* d_exp, uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_states && d_exc && d_pos && d_cnt && d_enc && d_fac && d_exp,
	            alpgpu::launch_encode_values_f32(ctx->stream, ctx->n_cus, d_in, d_states, d_state_idx, d_exc, d_pos, exc_stride, d_cnt, d_enc, d_fac,
	                                             d_exp, n_vectors));
}
int alpgpu_analyze_ffor_i32(alpgpu_ctx* ctx, const int32_t* d_enc, uint8_t* d_bw, int32_t* d_base, uint64_t n_vectors) {
	ALPGPU_PRIM(d_enc && d_bw && d_base, alpgpu::launch_analyze_ffor_i32(ctx->stream, ctx->n_cus, d_enc, d_bw, d_base, n_vectors));
}
int alpgpu_rd_encode_vectors_f32(alpgpu_ctx* ctx, const float* d_in, const alpgpu_rowgroup_state* d_states, const uint32_t* d_state_idx,
                                 uint16_t* d_exc, uint16_t* d_pos, size_t exc_stride, uint16_t* d_cnt, uint32_t* d_right, uint16_t* d_left,
                                 uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_states && d_exc && d_pos && d_cnt && d_right && d_left,
	            alpgpu::launch_rd_encode_f32(ctx->stream, ctx->n_cus, d_in, d_states, d_state_idx, d_exc, d_pos, exc_stride, d_cnt, d_right, d_left,
	                                         n_vectors));
}
int alpgpu_rd_decode_vectors_f32(alpgpu_ctx* ctx, float* d_out, const uint32_t* d_right, const uint16_t* d_left, const alpgpu_rowgroup_state* d_states,
                                 const uint32_t* d_state_idx, const uint16_t* d_exc, const uint16_t* d_pos, size_t exc_stride, const uint16_t* d_cnt,
                                 uint64_t n_vectors) {
	ALPGPU_PRIM(d_out && d_right && d_left && d_states && d_exc && d_pos && d_cnt,
	            alpgpu::launch_rd_decode_f32(ctx->stream, ctx->n_cus, d_out, d_right, d_left, d_states, d_state_idx, d_exc, d_pos, exc_stride, d_cnt,
	                                         n_vectors));
}

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
	alpgpu_ctx* ctx       = nullptr;
	~HostPipe() {
		if (ctx) {
			ctx->stream           = saved;
			ctx->encode_unordered = saved_unordered;
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
	P.ctx   = ctx;
	P.saved = ctx->stream;
	P.saved_unordered     = ctx->encode_unordered;
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
	P.ctx   = ctx;
	P.saved = ctx->stream;
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
