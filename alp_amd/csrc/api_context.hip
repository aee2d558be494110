// api_context.hip — contexts, options, streams, device memory and the per-context scan workspace of include/alpgpu.h (see host_ctx.hpp for the map).
#include "host_ctx.hpp"

namespace alpgpu_host {
thread_local char g_err[512] = "";
int fail(int code, const char* what, hipError_t e) {
	if (e != hipSuccess) {
		std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
	} else {
		std::snprintf(g_err, sizeof(g_err), "%s", what);
	}
	return code;
}
} // namespace alpgpu_host

extern "C" {

// One scan / status workspace per context.  Work that uses it is ordered behind the previous user: same stream = stream
// order; another stream (alpgpu_set_stream between two encodes, e.g. a torch stream switch) waits on the event recorded
// behind the previous encode; growing the buffer waits for that event on the host before the old buffer is freed.
int ensure_workspace(alpgpu_ctx* ctx, uint64_t bytes) {
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
int workspace_used(alpgpu_ctx* ctx) {
	ALPGPU_HIP(hipEventRecord(ctx->ws_event, ctx->stream));
	ctx->ws_stream = ctx->stream;
	ctx->ws_busy   = 1;
	return ALPGPU_OK;
}

int check_column(const alpgpu_column* col, uint64_t n_vectors) {
	if (!col) { return fail(ALPGPU_ERR_INVALID, "null column"); }
	if (col->n_vectors != n_vectors) { return fail(ALPGPU_ERR_INVALID, "column.n_vectors does not match n_vectors"); }
	if (col->n_rowgroups != (n_vectors + 99) / 100) { return fail(ALPGPU_ERR_INVALID, "column.n_rowgroups must be ceil(n_vectors/100)"); }
	if (n_vectors && (!col->d_rowgroups || !col->d_vectors || !col->d_packed || !col->d_exc || !col->d_totals)) {
		return fail(ALPGPU_ERR_INVALID, "column buffers must be allocated by the caller");
	}
	return ALPGPU_OK; // (an empty column may have no buffers at all: every entry point returns early for it)
}

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
	ctx->decode_patch_max   = (std::getenv("ALPGPU_DECODE_PATCH_AFTER") && alpgpu::decode_patch_arm_compiled()) ? std::atoi(std::getenv("ALPGPU_DECODE_PATCH_AFTER")) : 0; // (A/B runs)
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
	ctx->decode_unhinted = std::getenv("ALPGPU_DECODE_UNHINTED") ? std::atoi(std::getenv("ALPGPU_DECODE_UNHINTED")) : 1;
	for (auto& l : ctx->learn) { l.state = 0, l.key = nullptr, l.ev = nullptr; }
	ctx->learn_next      = 0;
	ctx->h_learn         = nullptr;
	// Where kernels of two streams cannot run side by side the read-ahead can only wait for a decode that starts after it has left (its patience: a few hundred
	// microseconds per decode, read_ahead_kernels.hip): left to itself (-1) the library then does not start it at all.  What the runtime documents for that:
	{
		const char* ser = std::getenv("AMD_SERIALIZE_KERNEL");
		const char* blk = std::getenv("HIP_LAUNCH_BLOCKING");
		const char* q   = std::getenv("GPU_MAX_HW_QUEUES");
		ctx->streams_serialize = ((ser && std::atoi(ser) != 0) || (blk && std::atoi(blk) != 0) || (q && std::atoi(q) == 1)) ? 1 : 0;
	}
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
	// what unhinted decodes learn about their columns (api_decode.hip): page-locked words + one event per slot; without them such columns simply stay unhinted
	if (hipHostMalloc(reinterpret_cast<void**>(&ctx->h_learn), sizeof(uint64_t) * kLearnSlots * 3 * kMaxSegments, hipHostMallocDefault) != hipSuccess) {
		(void)hipGetLastError();
		ctx->h_learn = nullptr;
	} else {
		for (auto& l : ctx->learn) {
			if (hipEventCreateWithFlags(&l.ev, hipEventDisableTiming) != hipSuccess) {
				(void)hipGetLastError();
				for (auto& m : ctx->learn) {
					if (m.ev) { (void)hipEventDestroy(m.ev); m.ev = nullptr; }
				}
				(void)hipHostFree(ctx->h_learn);
				ctx->h_learn = nullptr;
				break;
			}
		}
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
	for (auto& l : ctx->learn) {
		if (l.ev) {
			if (l.state == 1) { (void)hipEventSynchronize(l.ev); }
			(void)hipEventDestroy(l.ev);
		}
	}
	if (ctx->h_learn) { (void)hipHostFree(ctx->h_learn); }
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
		if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && !(value >= 16 && value <= 30)) {
			return fail(ALPGPU_ERR_INVALID, "decode vectors per workgroup must be 0 (auto), 1, 2, 4 or (float columns) 8: one wavefront per vector, 16-30: streamed by persistent workgroups");
		}
		ctx->decode_auto    = value == 0;
		ctx->decode_vpw     = static_cast<int>(value);
		ctx->decode_variant = (ctx->decode_variant & ~5) | ((value == 2 || value == 4) ? 0 : 1) | (value == 4 ? 4 : 0); // (4: four vectors over the narrow stage; 8: float columns only, double columns take 1)
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
	case ALPGPU_OPT_DECODE_UNHINTED:
		if (value < 0 || value > 2) { return fail(ALPGPU_ERR_INVALID, "unhinted decode: 0 (as before round 6), 1 (sizes summed on the stream, read-ahead planned on the device, learned for the next decode) or 2 (1 + every candidate shape launched, gated)"); }
		ctx->decode_unhinted = static_cast<int>(value);
		return ALPGPU_OK;
	case ALPGPU_OPT_DECODE_PATCH_AFTER:
		if (value < 0 || value > 64) { return fail(ALPGPU_ERR_INVALID, "decode patch-after: 0 (never) .. 64 exceptions per vector"); }
		// the arm exists in -DALPGPU_DECODE_PATCH_MODE=1 / 2 builds only: a build without it accepts the option and IGNORES it — the launch rule (shape, residency pad,
		// read-ahead limit, segment kinds: column_decodes_with_exceptions) must not count on an arm that is not there
		if (!alpgpu::decode_patch_arm_compiled()) { value = 0; }
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
// ==== single precision =================================================================================================
// worst case per vector: ALP bw=32 -> 4096 B; ALP_RD rbw=31,lbw=3 -> 4352 B.  +1 KiB slack at the end.
uint64_t alpgpu_packed_capacity_f32(uint64_t n_vectors) { return n_vectors * 4352ull + 1024ull; }
// worst case per vector: 1024 exceptions x (4 B value + 2 B position)
uint64_t alpgpu_exc_capacity_f32(uint64_t n_vectors) { return n_vectors * 6144ull + 64ull; }

} // extern "C"
