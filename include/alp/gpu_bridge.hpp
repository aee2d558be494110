// alp/gpu_bridge.hpp — the only piece of include/alp.hpp that talks to libalpgpu.so (include/alpgpu.h).
//
// One process-wide context (device chosen by ALPGPU_DEVICE, default 0) and one set of device scratch buffers per
// host thread, so the per-vector API of the reference stays re-entrant (the reference made its scratch
// thread_local for the same reason, include/alp/encoder.hpp:314-319).  Any failure of the GPU path throws
// std::runtime_error: there is no CPU implementation to fall back to.
#ifndef ALP_GPU_BRIDGE_HPP
#define ALP_GPU_BRIDGE_HPP

#include "alpgpu.h"
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>

namespace alp { namespace gpu {

inline void check(int rc, const char* what) {
	if (rc != ALPGPU_OK) { throw std::runtime_error(std::string(what) + ": " + alpgpu_last_error()); }
}

inline alpgpu_ctx* context() {
	static alpgpu_ctx*    ctx = nullptr;
	static std::once_flag once;
	std::call_once(once, [] {
		const char* dev = std::getenv("ALPGPU_DEVICE");
		check(alpgpu_ctx_create(dev ? std::atoi(dev) : 0, &ctx), "alpgpu_ctx_create");
	});
	return ctx;
}

// Per-thread device scratch: room for one vector of every array the per-vector API touches.
struct scratch {
	uint8_t* base {nullptr};
	// byte offsets inside base (all 16-byte aligned)
	static constexpr size_t IN = 0, ENC = 8192, PACKED = 16384, OUT = 24832, EXC = 33024, POS = 41216, LEFT = 43264,
	                        PACKED_LEFT = 45312, META = 47360, STATE = 47616, SAMPLES = 47680, TOTAL = 50176;
	scratch() { check(alpgpu_malloc(context(), reinterpret_cast<void**>(&base), TOTAL), "alpgpu_malloc"); }
	~scratch() {
		if (base) { alpgpu_free(context(), base); }
	}
	template <class T>
	T* at(size_t off) const {
		return reinterpret_cast<T*>(base + off);
	}
	// META layout: [0] bw u8, [1] fac u8, [2] exp u8, [8..15] base i64 (or i32 in [8..11]), [16..17] cnt u16, [24..25] base16 u16
	uint8_t*  bw() const { return base + META; }
	uint8_t*  fac() const { return base + META + 1; }
	uint8_t*  exp() const { return base + META + 2; }
	int64_t*  ffor_base() const { return at<int64_t>(META + 8); }
	uint16_t* cnt() const { return at<uint16_t>(META + 16); }
	uint16_t* base16() const { return at<uint16_t>(META + 24); }
};

inline scratch& tls() {
	static thread_local scratch s;
	return s;
}

// The C ABI has one entry point per precision (…_f64 / …_f32, include/alpgpu.h); the templates of this header pick theirs here.
template <class PT>
struct abi;
template <>
struct abi<double> {
	using ST = int64_t;
	using UT = uint64_t;
	static constexpr size_t VEC_BYTES = 8192;
	static int state_from_samples(const double* smp, uint32_t n, alpgpu_rowgroup_state* st) { return alpgpu_state_from_samples_f64(context(), smp, n, st); }
	static int rd_state_from_samples(const double* smp, uint32_t n, alpgpu_rowgroup_state* st) { return alpgpu_rd_state_from_samples_f64(context(), smp, n, st); }
	static int rd_dictionary_for_cut(const double* smp, uint32_t n, uint8_t rbw, alpgpu_rowgroup_state* st, double* est) { return alpgpu_rd_dictionary_for_cut_f64(context(), smp, n, rbw, st, est); }
	static int encode_values(const double* in, const alpgpu_rowgroup_state* st, double* exc, uint16_t* pos, uint16_t* cnt, ST* enc, uint8_t* fac, uint8_t* exp) {
		return alpgpu_encode_values_f64(context(), in, st, nullptr, exc, pos, 1024, cnt, enc, fac, exp, 1);
	}
	static int encode_simdized(const double* in, double* exc, uint16_t* pos, uint16_t* cnt, ST* enc, const uint8_t* fac, const uint8_t* exp) {
		return alpgpu_encode_simdized_f64(context(), in, exc, pos, 1024, cnt, enc, fac, exp, 1);
	}
	static int analyze_ffor(const ST* enc, uint8_t* bw, ST* base) { return alpgpu_analyze_ffor_i64(context(), enc, bw, base, 1); }
	static int decode_values(const ST* enc, double* out, const uint8_t* fac, const uint8_t* exp) { return alpgpu_decode_values_f64(context(), enc, out, fac, exp, 1); }
	static int patch(double* out, const double* exc, const uint16_t* pos, const uint16_t* cnt) { return alpgpu_patch_f64(context(), out, exc, pos, 1024, cnt, 1); }
	static int rd_encode(const double* in, const alpgpu_rowgroup_state* st, uint16_t* exc, uint16_t* pos, uint16_t* cnt, UT* right, uint16_t* left) {
		return alpgpu_rd_encode_vectors_f64(context(), in, st, nullptr, exc, pos, 1024, cnt, right, left, 1);
	}
	static int rd_decode(double* out, const UT* right, const uint16_t* left, const alpgpu_rowgroup_state* st, const uint16_t* exc, const uint16_t* pos,
	                     const uint16_t* cnt) {
		return alpgpu_rd_decode_vectors_f64(context(), out, right, left, st, nullptr, exc, pos, 1024, cnt, 1);
	}
};
template <>
struct abi<float> {
	using ST = int32_t;
	using UT = uint32_t;
	static constexpr size_t VEC_BYTES = 4096;
	static int state_from_samples(const float* smp, uint32_t n, alpgpu_rowgroup_state* st) { return alpgpu_state_from_samples_f32(context(), smp, n, st); }
	static int rd_state_from_samples(const float* smp, uint32_t n, alpgpu_rowgroup_state* st) { return alpgpu_rd_state_from_samples_f32(context(), smp, n, st); }
	static int rd_dictionary_for_cut(const float* smp, uint32_t n, uint8_t rbw, alpgpu_rowgroup_state* st, double* est) { return alpgpu_rd_dictionary_for_cut_f32(context(), smp, n, rbw, st, est); }
	static int encode_values(const float* in, const alpgpu_rowgroup_state* st, float* exc, uint16_t* pos, uint16_t* cnt, ST* enc, uint8_t* fac, uint8_t* exp) {
		return alpgpu_encode_values_f32(context(), in, st, nullptr, exc, pos, 1024, cnt, enc, fac, exp, 1);
	}
	static int encode_simdized(const float* in, float* exc, uint16_t* pos, uint16_t* cnt, ST* enc, const uint8_t* fac, const uint8_t* exp) {
		return alpgpu_encode_simdized_f32(context(), in, exc, pos, 1024, cnt, enc, fac, exp, 1);
	}
	static int analyze_ffor(const ST* enc, uint8_t* bw, ST* base) { return alpgpu_analyze_ffor_i32(context(), enc, bw, base, 1); }
	static int decode_values(const ST* enc, float* out, const uint8_t* fac, const uint8_t* exp) { return alpgpu_decode_values_f32(context(), enc, out, fac, exp, 1); }
	static int patch(float* out, const float* exc, const uint16_t* pos, const uint16_t* cnt) { return alpgpu_patch_f32(context(), out, exc, pos, 1024, cnt, 1); }
	static int rd_encode(const float* in, const alpgpu_rowgroup_state* st, uint16_t* exc, uint16_t* pos, uint16_t* cnt, UT* right, uint16_t* left) {
		return alpgpu_rd_encode_vectors_f32(context(), in, st, nullptr, exc, pos, 1024, cnt, right, left, 1);
	}
	static int rd_decode(float* out, const UT* right, const uint16_t* left, const alpgpu_rowgroup_state* st, const uint16_t* exc, const uint16_t* pos,
	                     const uint16_t* cnt) {
		return alpgpu_rd_decode_vectors_f32(context(), out, right, left, st, nullptr, exc, pos, 1024, cnt, 1);
	}
};

inline void h2d(void* d, const void* h, size_t n) { check(alpgpu_memcpy_h2d(context(), d, h, n), "alpgpu_memcpy_h2d"); }
inline void d2h(void* h, const void* d, size_t n) { check(alpgpu_memcpy_d2h(context(), h, d, n), "alpgpu_memcpy_d2h"); }

}} // namespace alp::gpu
#endif
