// alp/batch.hpp — the reference's per-vector loop, a rowgroup at a time.
//
// The per-vector functions of this header set (alp::encoder<PT>::encode, analyze_ffor, ffor::ffor, falp, patch_exceptions)
// ship ONE 1024-value vector to the GPU and back per call: 2 uploads, 1-3 kernels and 4-5 synchronous downloads per vector,
// bound by PCIe round trips (INTEGRATION.md has the measured rates).  A caller that keeps the reference's loop shape
//
//     for each rowgroup:  encoder::init(...)                                  (include/alp/encoder.hpp:402-427 upstream,
//         for each vector:  encoder::encode(...); analyze_ffor(...); ffor(...)  test/test_alp_sample.cpp:137-166)
//
// replaces the inner loop by ONE call of alp::gpu::rowgroup<PT>::encode (and the decode loop falp + patch_exceptions by
// ::decode): one upload, three kernels over all vectors of the rowgroup, one download — literally one copy each way: the arrays of
// a call lie next to each other in one device buffer and in a page-locked host mirror of it.  The outputs are the per-vector
// outputs of the reference, at a fixed stride of 1024 elements per vector: the FFOR-packed words (16 * bit_width words
// used), bit widths, bases, (factor, exponent), exceptions, positions, counts.  Bit-identical to calling the per-vector
// functions in a loop (tests/cpp/batch_test.cpp checks that on the GPU).  ALP_RD rowgroups have ::encode_rd / ::decode_rd.
#ifndef ALP_BATCH_HPP
#define ALP_BATCH_HPP
#include <cstring>
#include <stdexcept>
#include <mutex>
#include <utility>
#include <vector>

#include "alp/config.hpp"
#include "alp/decoder.hpp"
#include "alp/encoder.hpp"
#include "alp/gpu_bridge.hpp"

namespace alp { namespace gpu {

// Per-thread scratch of the batched calls: one device buffer and a page-locked host mirror of it.  Every call lays the arrays it
// needs out next to each other (sized for ITS n vectors), so that whatever goes up is one contiguous copy and whatever comes
// down is another; user arrays are copied into / out of the mirror with memcpy.
//   [ idx: cap x u32 zeros (every vector uses state 0) ][ state: 64 B ][ the call's arrays ... ]
struct batch_scratch {
	uint8_t* dev {nullptr};
	uint8_t* host {nullptr};
	size_t   cap {0};   // vectors the zero index array covers
	size_t   bytes {0}; // size of both buffers
	~batch_scratch() {
		if (dev) { alpgpu_free(context(), dev); }
		if (host) { alpgpu_free_host(context(), host); }
	}
	size_t off_state() const { return (cap * 4 + 63) & ~size_t(63); }
	size_t off_arrays() const { return off_state() + 64; }
	// room for n vectors whose arrays take at most per_vector bytes each
	void ensure(size_t n, size_t per_vector) {
		const size_t want_cap = n < 100 ? 100 : n;
		const size_t need     = ((want_cap * 4 + 63) & ~size_t(63)) + 64 + want_cap * per_vector + 64;
		if (want_cap <= cap && need <= bytes) { return; }
		if (dev) { check(alpgpu_free(context(), dev), "alpgpu_free"); }
		if (host) { check(alpgpu_free_host(context(), host), "alpgpu_free_host"); }
		dev = host = nullptr;
		cap   = want_cap > cap ? want_cap : cap;
		bytes = ((cap * 4 + 63) & ~size_t(63)) + 64 + cap * per_vector + 64;
		if (bytes < need) { bytes = need; }
		check(alpgpu_malloc(context(), reinterpret_cast<void**>(&dev), bytes), "alpgpu_malloc");
		check(alpgpu_malloc_host(context(), reinterpret_cast<void**>(&host), bytes), "alpgpu_malloc_host");
		check(alpgpu_memset(context(), dev, 0, off_state()), "alpgpu_memset");
	}
	template <class T>
	T* d(size_t off) const {
		return reinterpret_cast<T*>(dev + off);
	}
	uint8_t* h(size_t off) const { return host + off; }
	void     up(size_t off, size_t n) const { check(alpgpu_memcpy_h2d_async(context(), dev + off, host + off, n), "alpgpu_memcpy_h2d_async"); }
	void     down(size_t off, size_t n) const { check(alpgpu_memcpy_d2h(context(), host + off, dev + off, n), "alpgpu_memcpy_d2h"); } // synchronous
};
inline batch_scratch& batch_tls() {
	static thread_local batch_scratch s;
	return s;
}

template <class PT>
struct rowgroup {
	using ST = typename inner_t<PT>::st;
	using UT = typename inner_t<PT>::ut;
	static constexpr size_t V  = config::VECTOR_SIZE;
	static constexpr size_t VB = V * sizeof(PT); // bytes of one vector of values / encoded integers / right parts

	// the arrays of the ALP calls, in the order they lie in the buffers (n = vectors of this call)
	struct alp_layout {
		size_t in, exc, packed, pos, base, cnt, bw, fac, exp, enc, end;
		alp_layout(const batch_scratch& s, size_t n) {
			in     = s.off_arrays();       // n x VB   values in (encode) / out (decode)
			exc    = in + n * VB;          // n x VB   exception values            -+
			packed = exc + n * VB;         // n x VB   FFOR words                    |
			pos    = packed + n * VB;      // n x 2 KiB exception positions         |  one block: encode's output,
			base   = pos + n * 2048;       // n x 8                                 |  decode's input
			cnt    = base + n * 8;         // n x 2                                 |
			bw     = cnt + n * 2;          // n                                     |
			fac    = bw + n;               // n                                     |
			exp    = fac + n;              // n                                    -+
			enc    = (exp + n + 15) & ~size_t(15); // n x VB   encoded integers (device only)
			end    = enc + n * VB;
		}
	};
	static constexpr size_t kAlpPerVector = 4 * VB + 2048 + 8 + 2 + 3 + 1;
	// ... and of the ALP_RD calls
	struct rd_layout {
		size_t in, right, left, exc, pos, cnt, end;
		rd_layout(const batch_scratch& s, size_t n) {
			in    = s.off_arrays();   // n x VB    values in (encode) / out (decode)
			right = in + n * VB;      // n x VB    right parts          -+
			left  = right + n * VB;   // n x 2 KiB dictionary indices    |  one block: encode's output, decode's input
			exc   = left + n * 2048;  // n x 2 KiB exceptions            |
			pos   = exc + n * 2048;   // n x 2 KiB positions             |
			cnt   = pos + n * 2048;   // n x 2                          -+
			end   = cnt + n * 2;
		}
	};
	static constexpr size_t kRdPerVector = 2 * VB + 3 * 2048 + 2 + 1;
	static constexpr size_t kPerVector   = kAlpPerVector > kRdPerVector ? kAlpPerVector : kRdPerVector;

	//! ALP rowgroup: second-level sampling + encode, analyze_ffor, ffor for n_vectors vectors sharing `stt` (from encoder<PT>::init).
	//! All outputs are host arrays at a stride of 1024 elements per vector (counts / bit_widths / bases / facs / exps: one per vector).
	static void encode(const PT* vectors, size_t n_vectors, const state<PT>& stt, ST* ffor_packed, bw_t* bit_widths, ST* bases, uint8_t* facs, uint8_t* exps,
	                   PT* exceptions, exp_p_t* positions, exp_c_t* counts) {
		if (n_vectors == 0) { return; }
		auto& s = batch_tls();
		s.ensure(n_vectors, kPerVector);
		alpgpu_ctx*                 c = context();
		const alpgpu_rowgroup_state d = to_device_state(stt);
		const uint64_t              n = n_vectors;
		const alp_layout            L(s, n);
		std::memcpy(s.h(s.off_state()), &d, sizeof(d));
		std::memcpy(s.h(L.in), vectors, n * VB);
		s.up(s.off_state(), L.in + n * VB - s.off_state()); // state + values: one copy
		auto* st  = s.d<alpgpu_rowgroup_state>(s.off_state());
		auto* idx = s.d<uint32_t>(0);
		if constexpr (sizeof(PT) == 8) {
			check(alpgpu_encode_values_f64(c, s.d<double>(L.in), st, idx, s.d<double>(L.exc), s.d<uint16_t>(L.pos), V, s.d<uint16_t>(L.cnt), s.d<int64_t>(L.enc),
			                               s.d<uint8_t>(L.fac), s.d<uint8_t>(L.exp), n), "alpgpu_encode_values_f64");
			check(alpgpu_analyze_ffor_i64(c, s.d<int64_t>(L.enc), s.d<uint8_t>(L.bw), s.d<int64_t>(L.base), n), "alpgpu_analyze_ffor_i64");
			check(alpgpu_ffor_i64(c, s.d<int64_t>(L.enc), s.d<int64_t>(L.packed), V, s.d<uint8_t>(L.bw), s.d<int64_t>(L.base), n), "alpgpu_ffor_i64");
		} else {
			check(alpgpu_encode_values_f32(c, s.d<float>(L.in), st, idx, s.d<float>(L.exc), s.d<uint16_t>(L.pos), V, s.d<uint16_t>(L.cnt), s.d<int32_t>(L.enc),
			                               s.d<uint8_t>(L.fac), s.d<uint8_t>(L.exp), n), "alpgpu_encode_values_f32");
			check(alpgpu_analyze_ffor_i32(c, s.d<int32_t>(L.enc), s.d<uint8_t>(L.bw), s.d<int32_t>(L.base), n), "alpgpu_analyze_ffor_i32");
			check(alpgpu_ffor_i32(c, s.d<int32_t>(L.enc), s.d<int32_t>(L.packed), V, s.d<uint8_t>(L.bw), s.d<int32_t>(L.base), n), "alpgpu_ffor_i32");
		}
		s.down(L.exc, L.exp + n - L.exc); // everything the caller gets back: one copy
		std::memcpy(exceptions, s.h(L.exc), n * VB);
		std::memcpy(ffor_packed, s.h(L.packed), n * VB);
		std::memcpy(positions, s.h(L.pos), n * 2048);
		// bases travel as 8 bytes per vector; the float API's are int32
		if constexpr (sizeof(ST) == 8) {
			std::memcpy(bases, s.h(L.base), n * 8);
		} else {
			std::memcpy(bases, s.h(L.base), n * 4);
		}
		std::memcpy(counts, s.h(L.cnt), n * 2);
		std::memcpy(bit_widths, s.h(L.bw), n);
		std::memcpy(facs, s.h(L.fac), n);
		std::memcpy(exps, s.h(L.exp), n);
	}

	//! falp + patch_exceptions for n_vectors vectors (inputs as produced by encode())
	static void decode(const ST* ffor_packed, const bw_t* bit_widths, const ST* bases, const uint8_t* facs, const uint8_t* exps, const PT* exceptions,
	                   const exp_p_t* positions, const exp_c_t* counts, size_t n_vectors, PT* out) {
		if (n_vectors == 0) { return; }
		auto& s = batch_tls();
		s.ensure(n_vectors, kPerVector);
		alpgpu_ctx*      c = context();
		const uint64_t   n = n_vectors;
		const alp_layout L(s, n);
		std::memcpy(s.h(L.exc), exceptions, n * VB);
		std::memcpy(s.h(L.packed), ffor_packed, n * VB);
		std::memcpy(s.h(L.pos), positions, n * 2048);
		std::memcpy(s.h(L.base), bases, n * sizeof(ST));
		std::memcpy(s.h(L.cnt), counts, n * 2);
		std::memcpy(s.h(L.bw), bit_widths, n);
		std::memcpy(s.h(L.fac), facs, n);
		std::memcpy(s.h(L.exp), exps, n);
		s.up(L.exc, L.exp + n - L.exc); // one copy
		if constexpr (sizeof(PT) == 8) {
			check(alpgpu_falp_f64(c, s.d<int64_t>(L.packed), V, s.d<double>(L.in), s.d<uint8_t>(L.bw), s.d<int64_t>(L.base), s.d<uint8_t>(L.fac), s.d<uint8_t>(L.exp), n),
			      "alpgpu_falp_f64");
			check(alpgpu_patch_f64(c, s.d<double>(L.in), s.d<double>(L.exc), s.d<uint16_t>(L.pos), V, s.d<uint16_t>(L.cnt), n), "alpgpu_patch_f64");
		} else {
			check(alpgpu_falp_f32(c, s.d<int32_t>(L.packed), V, s.d<float>(L.in), s.d<uint8_t>(L.bw), s.d<int32_t>(L.base), s.d<uint8_t>(L.fac), s.d<uint8_t>(L.exp), n),
			      "alpgpu_falp_f32");
			check(alpgpu_patch_f32(c, s.d<float>(L.in), s.d<float>(L.exc), s.d<uint16_t>(L.pos), V, s.d<uint16_t>(L.cnt), n), "alpgpu_patch_f32");
		}
		s.down(L.in, n * VB);
		std::memcpy(out, s.h(L.in), n * VB);
	}

	//! ALP_RD rowgroup (rd_encoder<PT>::encode per vector, include/alp/rd.hpp:109-147 upstream): right parts, left dictionary
	//! indices, exceptions (left parts), positions, counts — unpacked, at a stride of 1024 per vector, as the reference returns them
	static void encode_rd(const PT* vectors, size_t n_vectors, const state<PT>& stt, UT* right_parts, uint16_t* left_parts, uint16_t* exceptions, exp_p_t* positions,
	                      exp_c_t* counts) {
		if (n_vectors == 0) { return; }
		auto& s = batch_tls();
		s.ensure(n_vectors, kPerVector);
		alpgpu_ctx*                 c = context();
		const alpgpu_rowgroup_state d = to_device_state(stt);
		const uint64_t              n = n_vectors;
		const rd_layout             L(s, n);
		std::memcpy(s.h(s.off_state()), &d, sizeof(d));
		std::memcpy(s.h(L.in), vectors, n * VB);
		s.up(s.off_state(), L.in + n * VB - s.off_state());
		auto* st  = s.d<alpgpu_rowgroup_state>(s.off_state());
		auto* idx = s.d<uint32_t>(0);
		if constexpr (sizeof(PT) == 8) {
			check(alpgpu_rd_encode_vectors_f64(c, s.d<double>(L.in), st, idx, s.d<uint16_t>(L.exc), s.d<uint16_t>(L.pos), V, s.d<uint16_t>(L.cnt), s.d<uint64_t>(L.right),
			                                   s.d<uint16_t>(L.left), n), "alpgpu_rd_encode_vectors_f64");
		} else {
			check(alpgpu_rd_encode_vectors_f32(c, s.d<float>(L.in), st, idx, s.d<uint16_t>(L.exc), s.d<uint16_t>(L.pos), V, s.d<uint16_t>(L.cnt), s.d<uint32_t>(L.right),
			                                   s.d<uint16_t>(L.left), n), "alpgpu_rd_encode_vectors_f32");
		}
		s.down(L.right, L.cnt + n * 2 - L.right);
		std::memcpy(right_parts, s.h(L.right), n * VB);
		std::memcpy(left_parts, s.h(L.left), n * 2048);
		std::memcpy(exceptions, s.h(L.exc), n * 2048);
		std::memcpy(positions, s.h(L.pos), n * 2048);
		std::memcpy(counts, s.h(L.cnt), n * 2);
	}

	//! rd_encoder<PT>::decode per vector (rd.hpp:152-178 upstream) for n_vectors vectors
	static void decode_rd(const UT* right_parts, const uint16_t* left_parts, const uint16_t* exceptions, const exp_p_t* positions, const exp_c_t* counts,
	                      const state<PT>& stt, size_t n_vectors, PT* out) {
		if (n_vectors == 0) { return; }
		auto& s = batch_tls();
		s.ensure(n_vectors, kPerVector);
		alpgpu_ctx*                 c = context();
		const alpgpu_rowgroup_state d = to_device_state(stt);
		const uint64_t              n = n_vectors;
		const rd_layout             L(s, n);
		std::memcpy(s.h(s.off_state()), &d, sizeof(d));
		std::memcpy(s.h(L.right), right_parts, n * VB);
		std::memcpy(s.h(L.left), left_parts, n * 2048);
		std::memcpy(s.h(L.exc), exceptions, n * 2048);
		std::memcpy(s.h(L.pos), positions, n * 2048);
		std::memcpy(s.h(L.cnt), counts, n * 2);
		s.up(s.off_state(), 64);
		s.up(L.right, L.cnt + n * 2 - L.right);
		auto* st  = s.d<alpgpu_rowgroup_state>(s.off_state());
		auto* idx = s.d<uint32_t>(0);
		if constexpr (sizeof(PT) == 8) {
			check(alpgpu_rd_decode_vectors_f64(c, s.d<double>(L.in), s.d<uint64_t>(L.right), s.d<uint16_t>(L.left), st, idx, s.d<uint16_t>(L.exc), s.d<uint16_t>(L.pos), V,
			                                   s.d<uint16_t>(L.cnt), n), "alpgpu_rd_decode_vectors_f64");
		} else {
			check(alpgpu_rd_decode_vectors_f32(c, s.d<float>(L.in), s.d<uint32_t>(L.right), s.d<uint16_t>(L.left), st, idx, s.d<uint16_t>(L.exc), s.d<uint16_t>(L.pos), V,
			                                   s.d<uint16_t>(L.cnt), n), "alpgpu_rd_decode_vectors_f32");
		}
		s.down(L.in, n * VB);
		std::memcpy(out, s.h(L.in), n * VB);
	}
};

// A whole column that lives in host memory, to and from its serialized form (include/alpgpu.h: alpgpu_compress_host_* — chunks of whole
// rowgroups go up on one stream while the previous chunk is encoded on another; ~35 GB/s of doubles from page-locked memory, ~10 GB/s
// from pageable memory like the std::vectors used here; callers that care hand page-locked buffers to the C functions directly).
// This is what replaces the reference's caller loop (publication/source_code/bench_compression_ratio/alp.cpp:198-229) as a whole.
// A pool of contexts for a list of devices (one each; the same device may be listed more than once — e.g. in tests on a one-GPU box):
// what alp::gpu::column<PT>::compress / decompress over a device list run on.  Contexts live until the process ends.
// A context serves ONE pipeline at a time (include/alpgpu.h: "a context must not be used by anything else during the call"), so the pool
// hands contexts out as a LEASE: entries are marked in use until the lease is destroyed, and a second thread that asks for the same devices
// meanwhile gets other (new) contexts.  column<PT>::compress / decompress over a device list are thereby safe to call from several threads.
struct context_pool {
	struct entry {
		int         device;
		alpgpu_ctx* ctx;
		bool        in_use;
	};
	std::mutex         mu;
	std::vector<entry> entries; // in creation order; contexts live until the process ends
	static context_pool& instance() {
		static context_pool p;
		return p;
	}
};
class context_lease {
	std::vector<alpgpu_ctx*> ctxs_;
	std::vector<size_t>      held_;

public:
	explicit context_lease(const std::vector<int>& devices) {
		context_pool&               P = context_pool::instance();
		std::lock_guard<std::mutex> lock(P.mu);
		try {
			for (int dev : devices) {
				size_t at = P.entries.size();
				for (size_t i = 0; i < P.entries.size(); ++i) {
					if (!P.entries[i].in_use && P.entries[i].device == dev) {
						at = i;
						break;
					}
				}
				if (at == P.entries.size()) {
					alpgpu_ctx* c = nullptr;
					check(alpgpu_ctx_create(dev, &c), "alpgpu_ctx_create");
					P.entries.push_back({dev, c, false});
				}
				P.entries[at].in_use = true;
				held_.push_back(at);
				ctxs_.push_back(P.entries[at].ctx);
			}
		} catch (...) {
			for (size_t i : held_) { P.entries[i].in_use = false; }
			throw;
		}
	}
	~context_lease() {
		context_pool&               P = context_pool::instance();
		std::lock_guard<std::mutex> lock(P.mu);
		for (size_t i : held_) { P.entries[i].in_use = false; }
	}
	context_lease(const context_lease&)            = delete;
	context_lease& operator=(const context_lease&) = delete;
	const std::vector<alpgpu_ctx*>& contexts() const { return ctxs_; }
};

template <class PT>
struct column {
	// The column cut into whole-rowgroup shards over `devices` (alpgpu_compress_host_multi_*): every GPU of the node works on its shard
	// over its own PCIe link, the result is the one-device blob byte for byte.
	static std::vector<uint8_t> compress(const PT* values, size_t n_values, const std::vector<int>& devices) {
		const context_lease             lease(devices); // the contexts are this call's until it returns
		const std::vector<alpgpu_ctx*>& ctxs = lease.contexts();
		const uint64_t                 n    = (n_values + config::VECTOR_SIZE - 1) / config::VECTOR_SIZE;
		uint64_t                       cap  = alpgpu_blob_size(n, n * config::VECTOR_SIZE * sizeof(PT) + 1024 * ctxs.size(), 0);
		std::vector<uint8_t>           blob;
		for (int attempt = 0; attempt < 2; ++attempt) {
			blob.resize(cap);
			uint64_t  written = 0;
			const int rc = sizeof(PT) == 8 ? alpgpu_compress_host_multi_f64(ctxs.data(), static_cast<int>(ctxs.size()), reinterpret_cast<const double*>(values), n_values, blob.data(), cap, &written)
			                               : alpgpu_compress_host_multi_f32(ctxs.data(), static_cast<int>(ctxs.size()), reinterpret_cast<const float*>(values), n_values, blob.data(), cap, &written);
			if (rc == ALPGPU_ERR_CAPACITY && attempt == 0 && written > cap) {
				cap = written;
				continue;
			}
			check(rc, "alpgpu_compress_host_multi");
			blob.resize(written);
			break;
		}
		return blob;
	}
	static std::vector<PT> decompress(const uint8_t* blob, size_t size, const std::vector<int>& devices) {
		const context_lease             lease(devices);
		const std::vector<alpgpu_ctx*>& ctxs = lease.contexts();
		uint64_t  n_values = 0;
		const int probe    = sizeof(PT) == 8 ? alpgpu_decompress_host_multi_f64(ctxs.data(), 1, blob, size, nullptr, 0, &n_values)
		                                     : alpgpu_decompress_host_multi_f32(ctxs.data(), 1, blob, size, nullptr, 0, &n_values);
		if (probe != ALPGPU_OK && probe != ALPGPU_ERR_CAPACITY) { check(probe, "alpgpu_decompress_host_multi (header)"); }
		std::vector<PT> out(n_values);
		if (n_values == 0) { return out; }
		if constexpr (sizeof(PT) == 8) {
			check(alpgpu_decompress_host_multi_f64(ctxs.data(), static_cast<int>(ctxs.size()), blob, size, reinterpret_cast<double*>(out.data()), out.size(), &n_values), "alpgpu_decompress_host_multi_f64");
		} else {
			check(alpgpu_decompress_host_multi_f32(ctxs.data(), static_cast<int>(ctxs.size()), blob, size, reinterpret_cast<float*>(out.data()), out.size(), &n_values), "alpgpu_decompress_host_multi_f32");
		}
		return out;
	}
	static std::vector<uint8_t> compress(const PT* values, size_t n_values) {
		const uint64_t n = (n_values + config::VECTOR_SIZE - 1) / config::VECTOR_SIZE;
		// a compressed column is almost never larger than the column: start there, and take the size the library asks for otherwise
		// (the worst case — every value an exception — is 2.3 times the input)
		uint64_t             cap = alpgpu_blob_size(n, n * config::VECTOR_SIZE * sizeof(PT) + 1024, 0);
		std::vector<uint8_t> blob;
		for (int attempt = 0; attempt < 2; ++attempt) {
			blob.resize(cap);
			uint64_t written = 0;
			const int rc = sizeof(PT) == 8 ? alpgpu_compress_host_f64(context(), reinterpret_cast<const double*>(values), n_values, blob.data(), cap, &written)
			                               : alpgpu_compress_host_f32(context(), reinterpret_cast<const float*>(values), n_values, blob.data(), cap, &written);
			if (rc == ALPGPU_ERR_CAPACITY && attempt == 0 && written > cap) {
				cap = written;
				continue;
			}
			check(rc, "alpgpu_compress_host");
			blob.resize(written);
			break;
		}
		return blob;
	}
	static std::vector<PT> decompress(const uint8_t* blob, size_t size) {
		// the value count comes from the library AFTER it has validated the header (a call with no output capacity returns it with
		// ALPGPU_ERR_CAPACITY): nothing is allocated on the word of a corrupt or hostile blob
		uint64_t n_values = 0;
		const int probe   = sizeof(PT) == 8 ? alpgpu_decompress_host_f64(context(), blob, size, nullptr, 0, &n_values)
		                                    : alpgpu_decompress_host_f32(context(), blob, size, nullptr, 0, &n_values);
		if (probe != ALPGPU_OK && probe != ALPGPU_ERR_CAPACITY) { check(probe, "alpgpu_decompress_host (header)"); }
		std::vector<PT> out(n_values);
		if (n_values == 0) { return out; }
		if constexpr (sizeof(PT) == 8) {
			check(alpgpu_decompress_host_f64(context(), blob, size, out.data(), out.size(), &n_values), "alpgpu_decompress_host_f64");
		} else {
			check(alpgpu_decompress_host_f32(context(), blob, size, out.data(), out.size(), &n_values), "alpgpu_decompress_host_f32");
		}
		return out;
	}
};

}} // namespace alp::gpu
#endif
