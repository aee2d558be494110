// alp/decoder.hpp — alp::decoder<PT> with the reference's signatures (include/alp/decoder.hpp:122-149), computed on the GPU.
#ifndef ALP_DECODER_HPP
#define ALP_DECODER_HPP
#include "alp/common.hpp"
#include "alp/config.hpp"
#include "alp/gpu_bridge.hpp"
#include <cstring>

namespace alp {

template <typename PT>
struct inner_t {};
template <>
struct inner_t<double> {
	using ut = uint64_t;
	using st = int64_t;
};
template <>
struct inner_t<float> {
	using ut = uint32_t;
	using st = int32_t;
};

template <class PT>
struct decoder {
	using UT = typename inner_t<PT>::ut;
	using ST = typename inner_t<PT>::st;

	//! decode of a whole vector: output[i] = PT(encoded[i] * FACT[fac]) * FRAC[exp]   (decoder.hpp:134-138)
	static inline void decode(const ST* encoded_integers, const uint8_t fac_idx, const uint8_t exp_idx, PT* output) {
		auto& s = gpu::tls();
		gpu::h2d(s.at<ST>(s.ENC), encoded_integers, gpu::abi<PT>::VEC_BYTES);
		const uint8_t fe[2] = {fac_idx, exp_idx};
		gpu::h2d(s.fac(), fe, 2);
		gpu::check(gpu::abi<PT>::decode_values(s.at<ST>(s.ENC), s.at<PT>(s.OUT), s.fac(), s.exp()), "alpgpu_decode_values");
		gpu::d2h(output, s.at<PT>(s.OUT), gpu::abi<PT>::VEC_BYTES);
	}

	//! single value (decoder.hpp:128-131); goes through the vector kernel like everything else in this header
	static inline PT decode_value(const ST encoded_value, const uint8_t factor, const uint8_t exponent) {
		ST in[config::VECTOR_SIZE] = {encoded_value};
		PT out[config::VECTOR_SIZE];
		decode(in, factor, exponent, out);
		return out[0];
	}

	//! out[pos[i]] = exceptions[i]   (decoder.hpp:141-149)
	static inline void
	patch_exceptions(PT* out, const PT* exceptions, const exp_p_t* exceptions_positions, const exp_c_t* exceptions_count) {
		const exp_c_t n = exceptions_count[0];
		if (n == 0) { return; }
		auto& s = gpu::tls();
		gpu::h2d(s.at<PT>(s.OUT), out, gpu::abi<PT>::VEC_BYTES);
		gpu::h2d(s.at<PT>(s.EXC), exceptions, static_cast<size_t>(n) * sizeof(PT));
		gpu::h2d(s.at<exp_p_t>(s.POS), exceptions_positions, static_cast<size_t>(n) * 2);
		gpu::h2d(s.cnt(), &n, 2);
		gpu::check(gpu::abi<PT>::patch(s.at<PT>(s.OUT), s.at<PT>(s.EXC), s.at<exp_p_t>(s.POS), s.cnt()), "alpgpu_patch");
		gpu::d2h(out, s.at<PT>(s.OUT), gpu::abi<PT>::VEC_BYTES);
	}
};

} // namespace alp
#endif
