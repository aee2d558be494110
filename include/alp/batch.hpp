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
// ::decode): one upload, three kernels over all vectors of the rowgroup, one download.  The outputs are the per-vector
// outputs of the reference, at a fixed stride of 1024 elements per vector: the FFOR-packed words (16 * bit_width words
// used), bit widths, bases, (factor, exponent), exceptions, positions, counts.  Bit-identical to calling the per-vector
// functions in a loop (tests/cpp/batch_test.cpp checks that on the GPU).  ALP_RD rowgroups have ::encode_rd / ::decode_rd.
#ifndef ALP_BATCH_HPP
#define ALP_BATCH_HPP
#include "alp/config.hpp"
#include "alp/decoder.hpp"
#include "alp/encoder.hpp"
#include "alp/gpu_bridge.hpp"

namespace alp { namespace gpu {

// per-thread device scratch for n vectors of every array the batched calls touch, grown on demand
struct batch_scratch {
	uint8_t* base {nullptr};
	size_t   cap {0};    // vectors
	size_t   vb {0};     // bytes per vector of the value type the scratch was sized for
	~batch_scratch() {
		if (base) { alpgpu_free(context(), base); }
	}
	// per-vector slices: A, B, C, D = four value-sized arrays; then 2-byte arrays P, L, Q, X; then bytes / small words
	size_t off_A() const { return 0; }
	size_t off_B() const { return cap * vb; }
	size_t off_C() const { return 2 * cap * vb; }
	size_t off_D() const { return 3 * cap * vb; }
	size_t off_P() const { return 4 * cap * vb; }
	size_t off_L() const { return off_P() + cap * 2048; }
	size_t off_Q() const { return off_L() + cap * 2048; }
	size_t off_X() const { return off_Q() + cap * 2048; }
	size_t off_base() const { return off_X() + cap * 2048; }          // cap x 8
	size_t off_idx() const { return off_base() + cap * 8; }           // cap x 4 (zeros: every vector uses state 0)
	size_t off_cnt() const { return off_idx() + cap * 4; }            // cap x 2
	size_t off_bw() const { return off_cnt() + cap * 2; }             // cap
	size_t off_fac() const { return off_bw() + cap; }                 // cap
	size_t off_exp() const { return off_fac() + cap; }                // cap
	size_t off_state() const { return (off_exp() + cap + 63) & ~size_t(63); }
	size_t total() const { return off_state() + 64; }
	void   ensure(size_t n, size_t value_bytes) {
        if (n <= cap && value_bytes * 1024 == vb) { return; }
        if (base) { check(alpgpu_free(context(), base), "alpgpu_free"); }
        base = nullptr;
        cap  = n < 100 ? 100 : n;
        vb   = value_bytes * 1024;
        check(alpgpu_malloc(context(), reinterpret_cast<void**>(&base), total()), "alpgpu_malloc");
        check(alpgpu_memset(context(), base + off_idx(), 0, cap * 4), "alpgpu_memset");
	}
	template <class T>
	T* at(size_t off) const {
		return reinterpret_cast<T*>(base + off);
	}
};
inline batch_scratch& batch_tls() {
	static thread_local batch_scratch s;
	return s;
}

template <class PT>
struct rowgroup {
	using ST = typename inner_t<PT>::st;
	using UT = typename inner_t<PT>::ut;
	static constexpr size_t V = config::VECTOR_SIZE;

	//! ALP rowgroup: second-level sampling + encode, analyze_ffor, ffor for n_vectors vectors sharing `stt` (from encoder<PT>::init).
	//! All outputs are host arrays at a stride of 1024 elements per vector (counts / bit_widths / bases / facs / exps: one per vector).
	static void encode(const PT* vectors, size_t n_vectors, const state<PT>& stt, ST* ffor_packed, bw_t* bit_widths, ST* bases, uint8_t* facs, uint8_t* exps,
	                   PT* exceptions, exp_p_t* positions, exp_c_t* counts) {
		if (n_vectors == 0) { return; }
		auto& s = batch_tls();
		s.ensure(n_vectors, sizeof(PT));
		alpgpu_ctx*                 c = context();
		const alpgpu_rowgroup_state d = to_device_state(stt);
		const uint64_t              n = n_vectors;
		h2d(s.at<PT>(s.off_A()), vectors, n * V * sizeof(PT));
		h2d(s.at<uint8_t>(s.off_state()), &d, sizeof(d));
		auto* st  = s.at<alpgpu_rowgroup_state>(s.off_state());
		auto* idx = s.at<uint32_t>(s.off_idx());
		if constexpr (sizeof(PT) == 8) {
			check(alpgpu_encode_values_f64(c, s.at<double>(s.off_A()), st, idx, s.at<double>(s.off_C()), s.at<uint16_t>(s.off_P()), V, s.at<uint16_t>(s.off_cnt()),
			                               s.at<int64_t>(s.off_B()), s.at<uint8_t>(s.off_fac()), s.at<uint8_t>(s.off_exp()), n), "alpgpu_encode_values_f64");
			check(alpgpu_analyze_ffor_i64(c, s.at<int64_t>(s.off_B()), s.at<uint8_t>(s.off_bw()), s.at<int64_t>(s.off_base()), n), "alpgpu_analyze_ffor_i64");
			check(alpgpu_ffor_i64(c, s.at<int64_t>(s.off_B()), s.at<int64_t>(s.off_D()), V, s.at<uint8_t>(s.off_bw()), s.at<int64_t>(s.off_base()), n), "alpgpu_ffor_i64");
		} else {
			check(alpgpu_encode_values_f32(c, s.at<float>(s.off_A()), st, idx, s.at<float>(s.off_C()), s.at<uint16_t>(s.off_P()), V, s.at<uint16_t>(s.off_cnt()),
			                               s.at<int32_t>(s.off_B()), s.at<uint8_t>(s.off_fac()), s.at<uint8_t>(s.off_exp()), n), "alpgpu_encode_values_f32");
			check(alpgpu_analyze_ffor_i32(c, s.at<int32_t>(s.off_B()), s.at<uint8_t>(s.off_bw()), s.at<int32_t>(s.off_base()), n), "alpgpu_analyze_ffor_i32");
			check(alpgpu_ffor_i32(c, s.at<int32_t>(s.off_B()), s.at<int32_t>(s.off_D()), V, s.at<uint8_t>(s.off_bw()), s.at<int32_t>(s.off_base()), n), "alpgpu_ffor_i32");
		}
		d2h(ffor_packed, s.at<ST>(s.off_D()), n * V * sizeof(ST));
		d2h(exceptions, s.at<PT>(s.off_C()), n * V * sizeof(PT));
		d2h(positions, s.at<uint16_t>(s.off_P()), n * V * 2);
		d2h(counts, s.at<uint16_t>(s.off_cnt()), n * 2);
		d2h(bit_widths, s.at<uint8_t>(s.off_bw()), n);
		d2h(bases, s.at<ST>(s.off_base()), n * sizeof(ST));
		d2h(facs, s.at<uint8_t>(s.off_fac()), n);
		d2h(exps, s.at<uint8_t>(s.off_exp()), n);
	}

	//! falp + patch_exceptions for n_vectors vectors (inputs as produced by encode())
	static void decode(const ST* ffor_packed, const bw_t* bit_widths, const ST* bases, const uint8_t* facs, const uint8_t* exps, const PT* exceptions,
	                   const exp_p_t* positions, const exp_c_t* counts, size_t n_vectors, PT* out) {
		if (n_vectors == 0) { return; }
		auto& s = batch_tls();
		s.ensure(n_vectors, sizeof(PT));
		alpgpu_ctx*    c = context();
		const uint64_t n = n_vectors;
		h2d(s.at<ST>(s.off_D()), ffor_packed, n * V * sizeof(ST));
		h2d(s.at<PT>(s.off_C()), exceptions, n * V * sizeof(PT));
		h2d(s.at<uint16_t>(s.off_P()), positions, n * V * 2);
		h2d(s.at<uint16_t>(s.off_cnt()), counts, n * 2);
		h2d(s.at<uint8_t>(s.off_bw()), bit_widths, n);
		h2d(s.at<ST>(s.off_base()), bases, n * sizeof(ST));
		h2d(s.at<uint8_t>(s.off_fac()), facs, n);
		h2d(s.at<uint8_t>(s.off_exp()), exps, n);
		if constexpr (sizeof(PT) == 8) {
			check(alpgpu_falp_f64(c, s.at<int64_t>(s.off_D()), V, s.at<double>(s.off_A()), s.at<uint8_t>(s.off_bw()), s.at<int64_t>(s.off_base()), s.at<uint8_t>(s.off_fac()),
			                      s.at<uint8_t>(s.off_exp()), n), "alpgpu_falp_f64");
			check(alpgpu_patch_f64(c, s.at<double>(s.off_A()), s.at<double>(s.off_C()), s.at<uint16_t>(s.off_P()), V, s.at<uint16_t>(s.off_cnt()), n), "alpgpu_patch_f64");
		} else {
			check(alpgpu_falp_f32(c, s.at<int32_t>(s.off_D()), V, s.at<float>(s.off_A()), s.at<uint8_t>(s.off_bw()), s.at<int32_t>(s.off_base()), s.at<uint8_t>(s.off_fac()),
			                      s.at<uint8_t>(s.off_exp()), n), "alpgpu_falp_f32");
			check(alpgpu_patch_f32(c, s.at<float>(s.off_A()), s.at<float>(s.off_C()), s.at<uint16_t>(s.off_P()), V, s.at<uint16_t>(s.off_cnt()), n), "alpgpu_patch_f32");
		}
		d2h(out, s.at<PT>(s.off_A()), n * V * sizeof(PT));
	}

	//! ALP_RD rowgroup (rd_encoder<PT>::encode per vector, include/alp/rd.hpp:109-147 upstream): right parts, left dictionary
	//! indices, exceptions (left parts), positions, counts — unpacked, at a stride of 1024 per vector, as the reference returns them
	static void encode_rd(const PT* vectors, size_t n_vectors, const state<PT>& stt, UT* right_parts, uint16_t* left_parts, uint16_t* exceptions, exp_p_t* positions,
	                      exp_c_t* counts) {
		if (n_vectors == 0) { return; }
		auto& s = batch_tls();
		s.ensure(n_vectors, sizeof(PT));
		alpgpu_ctx*                 c = context();
		const alpgpu_rowgroup_state d = to_device_state(stt);
		const uint64_t              n = n_vectors;
		h2d(s.at<PT>(s.off_A()), vectors, n * V * sizeof(PT));
		h2d(s.at<uint8_t>(s.off_state()), &d, sizeof(d));
		auto* st  = s.at<alpgpu_rowgroup_state>(s.off_state());
		auto* idx = s.at<uint32_t>(s.off_idx());
		if constexpr (sizeof(PT) == 8) {
			check(alpgpu_rd_encode_vectors_f64(c, s.at<double>(s.off_A()), st, idx, s.at<uint16_t>(s.off_Q()), s.at<uint16_t>(s.off_P()), V, s.at<uint16_t>(s.off_cnt()),
			                                   s.at<uint64_t>(s.off_B()), s.at<uint16_t>(s.off_L()), n), "alpgpu_rd_encode_vectors_f64");
		} else {
			check(alpgpu_rd_encode_vectors_f32(c, s.at<float>(s.off_A()), st, idx, s.at<uint16_t>(s.off_Q()), s.at<uint16_t>(s.off_P()), V, s.at<uint16_t>(s.off_cnt()),
			                                   s.at<uint32_t>(s.off_B()), s.at<uint16_t>(s.off_L()), n), "alpgpu_rd_encode_vectors_f32");
		}
		d2h(right_parts, s.at<UT>(s.off_B()), n * V * sizeof(UT));
		d2h(left_parts, s.at<uint16_t>(s.off_L()), n * V * 2);
		d2h(exceptions, s.at<uint16_t>(s.off_Q()), n * V * 2);
		d2h(positions, s.at<uint16_t>(s.off_P()), n * V * 2);
		d2h(counts, s.at<uint16_t>(s.off_cnt()), n * 2);
	}

	//! rd_encoder<PT>::decode per vector (rd.hpp:152-178 upstream) for n_vectors vectors
	static void decode_rd(const UT* right_parts, const uint16_t* left_parts, const uint16_t* exceptions, const exp_p_t* positions, const exp_c_t* counts,
	                      const state<PT>& stt, size_t n_vectors, PT* out) {
		if (n_vectors == 0) { return; }
		auto& s = batch_tls();
		s.ensure(n_vectors, sizeof(PT));
		alpgpu_ctx*                 c = context();
		const alpgpu_rowgroup_state d = to_device_state(stt);
		const uint64_t              n = n_vectors;
		h2d(s.at<UT>(s.off_B()), right_parts, n * V * sizeof(UT));
		h2d(s.at<uint16_t>(s.off_L()), left_parts, n * V * 2);
		h2d(s.at<uint16_t>(s.off_Q()), exceptions, n * V * 2);
		h2d(s.at<uint16_t>(s.off_P()), positions, n * V * 2);
		h2d(s.at<uint16_t>(s.off_cnt()), counts, n * 2);
		h2d(s.at<uint8_t>(s.off_state()), &d, sizeof(d));
		auto* st  = s.at<alpgpu_rowgroup_state>(s.off_state());
		auto* idx = s.at<uint32_t>(s.off_idx());
		if constexpr (sizeof(PT) == 8) {
			check(alpgpu_rd_decode_vectors_f64(c, s.at<double>(s.off_A()), s.at<uint64_t>(s.off_B()), s.at<uint16_t>(s.off_L()), st, idx, s.at<uint16_t>(s.off_Q()),
			                                   s.at<uint16_t>(s.off_P()), V, s.at<uint16_t>(s.off_cnt()), n), "alpgpu_rd_decode_vectors_f64");
		} else {
			check(alpgpu_rd_decode_vectors_f32(c, s.at<float>(s.off_A()), s.at<uint32_t>(s.off_B()), s.at<uint16_t>(s.off_L()), st, idx, s.at<uint16_t>(s.off_Q()),
			                                   s.at<uint16_t>(s.off_P()), V, s.at<uint16_t>(s.off_cnt()), n), "alpgpu_rd_decode_vectors_f32");
		}
		d2h(out, s.at<PT>(s.off_A()), n * V * sizeof(PT));
	}
};

}} // namespace alp::gpu
#endif
