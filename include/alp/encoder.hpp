// alp/encoder.hpp — alp::state<PT> and alp::encoder<PT> with the reference's signatures
// (include/alp/encoder.hpp:35-62, :109-120, :307-427), computed on the GPU through include/alpgpu.h.
#ifndef ALP_ENCODER_HPP
#define ALP_ENCODER_HPP
#include "alp/common.hpp"
#include "alp/config.hpp"
#include "alp/constants.hpp"
#include "alp/decoder.hpp"
#include "alp/gpu_bridge.hpp"
#include <cmath>
#include <stdexcept>
#include "alp/sampler.hpp"
#include <unordered_map>
#include <utility>
#include <vector>

namespace alp {

//! per-rowgroup (and, for exp/fac/exceptions_count, per-vector) state; field-for-field the reference's struct
template <typename PT>
struct state {
	using UT = typename inner_t<PT>::ut;
	using ST = typename inner_t<PT>::st;

	Scheme   scheme {Scheme::INVALID};
	uint16_t vector_size {config::VECTOR_SIZE};
	uint16_t exceptions_count {0};
	size_t   sampled_values_n {0};

	// ALP
	uint16_t                         k_combinations {5};
	std::vector<std::pair<int, int>> best_k_combinations; // (exponent, factor)
	uint8_t                          exp {};
	uint8_t                          fac {};
	bw_t                             bit_width {};
	ST                               for_base {};

	// ALP_RD
	bw_t                                   right_bit_width {0};
	bw_t                                   left_bit_width {0};
	UT                                     right_for_base {0};
	uint16_t                               left_for_base {0};
	uint16_t                               left_parts_dict[config::MAX_RD_DICTIONARY_SIZE] {};
	uint8_t                                actual_dictionary_size {};
	uint32_t                               actual_dictionary_size_bytes {};
	std::unordered_map<uint16_t, uint16_t> left_parts_dict_map;
};

namespace gpu {
//! host state -> the 32-byte device record of include/alpgpu.h
template <typename PT>
inline alpgpu_rowgroup_state to_device_state(const state<PT>& stt) {
	alpgpu_rowgroup_state d {};
	d.scheme = static_cast<uint8_t>(stt.scheme);
	d.k      = static_cast<uint8_t>(stt.k_combinations);
	for (size_t i = 0; i < stt.best_k_combinations.size() && i < config::MAX_K_COMBINATIONS; ++i) {
		d.combos[2 * i]     = static_cast<uint8_t>(stt.best_k_combinations[i].first);
		d.combos[2 * i + 1] = static_cast<uint8_t>(stt.best_k_combinations[i].second);
	}
	d.rd_rbw       = stt.right_bit_width;
	d.rd_lbw       = stt.left_bit_width;
	d.rd_dict_size = stt.actual_dictionary_size;
	for (size_t i = 0; i < config::MAX_RD_DICTIONARY_SIZE; ++i) { d.rd_dict[i] = stt.left_parts_dict[i]; }
	return d;
}

//! runs the rowgroup decision on the GPU for samples already gathered into sample_arr
template <typename PT>
inline alpgpu_rowgroup_state state_from_samples(const PT* sample_arr, size_t n_samples) {
	auto&        s       = tls();
	const size_t n_block = (n_samples + config::SAMPLES_PER_VECTOR - 1) / config::SAMPLES_PER_VECTOR;
	const size_t n_up    = n_samples < config::SAMPLES_PER_VECTOR ? n_samples : n_block * config::SAMPLES_PER_VECTOR;
	h2d(s.template at<PT>(s.SAMPLES), sample_arr, n_up * sizeof(PT));
	check(abi<PT>::state_from_samples(s.template at<PT>(s.SAMPLES), static_cast<uint32_t>(n_samples), s.template at<alpgpu_rowgroup_state>(s.STATE)),
	      "alpgpu_state_from_samples");
	alpgpu_rowgroup_state d {};
	d2h(&d, s.template at<alpgpu_rowgroup_state>(s.STATE), sizeof(d));
	return d;
}
} // namespace gpu

template <typename PT>
struct encoder {
	using UT = typename inner_t<PT>::ut;
	using ST = typename inner_t<PT>::st;

	static constexpr uint8_t EXACT_TYPE_BIT_SIZE = sizeof(UT) * 8;

	//! rowgroup init (encoder.hpp:420-427): sample, rank the (e,f) candidates, decide ALP vs ALP_RD
	static inline void
	init(const PT* data_column, const size_t column_offset, const size_t tuples_count, PT* sample_arr, state<PT>& stt) {
		stt.scheme           = Scheme::ALP;
		stt.sampled_values_n = sampler::first_level_sample<PT>(data_column, column_offset, tuples_count, sample_arr);
		stt.k_combinations   = config::MAX_K_COMBINATIONS;
		stt.best_k_combinations.clear();
		find_top_k_combinations(sample_arr, stt);
	}

	//! encoder.hpp:139-235
	static inline void find_top_k_combinations(const PT* smp_arr, state<PT>& stt) {
		const alpgpu_rowgroup_state d = gpu::state_from_samples(smp_arr, stt.sampled_values_n);
		if (d.scheme == ALPGPU_SCHEME_ALP_RD) {
			stt.scheme = Scheme::ALP_RD;
			return;
		}
		if (d.k < stt.k_combinations) { stt.k_combinations = d.k; }
		for (uint16_t i = 0; i < stt.k_combinations; ++i) { stt.best_k_combinations.emplace_back(d.combos[2 * i], d.combos[2 * i + 1]); }
	}

	//! vector encode (encoder.hpp:402-418): second-level sampling when k > 1, then encode_simdized
	static inline void encode(const PT*  input_vector,
	                          PT*        exceptions,
	                          uint16_t*  exceptions_positions,
	                          uint16_t*  exceptions_count,
	                          ST*        encoded_integers,
	                          state<PT>& stt) {
		auto&                       s = gpu::tls();
		const alpgpu_rowgroup_state d = gpu::to_device_state(stt);
		gpu::h2d(s.at<PT>(s.IN), input_vector, gpu::abi<PT>::VEC_BYTES);
		gpu::h2d(s.at<alpgpu_rowgroup_state>(s.STATE), &d, sizeof(d));
		gpu::check(gpu::abi<PT>::encode_values(s.at<PT>(s.IN), s.at<alpgpu_rowgroup_state>(s.STATE), s.at<PT>(s.EXC), s.at<uint16_t>(s.POS), s.cnt(),
		                                       s.at<ST>(s.ENC), s.fac(), s.exp()),
		           "alpgpu_encode_values");
		fetch_encoded(s, exceptions, exceptions_positions, exceptions_count, encoded_integers);
		uint8_t fe[2];
		gpu::d2h(fe, s.fac(), 2);
		stt.fac = fe[0];
		stt.exp = fe[1];
	}

	//! second-level sampling on its own (encoder.hpp:241-305): the best of the rowgroup's top_k candidates for this vector.  It is the first half
	//! of the device's vector encode (alpgpu_encode_values_*: candidate choice, then encode_simdized with the winner), whose choice is returned;
	//! the 32 samples are the values at stride 32 of a whole vector, which is what the reference's encode() passes (stt.vector_size = 1024) —
	//! another input_vector_size has no device form and throws std::invalid_argument.
	static inline void find_best_exponent_factor_from_combinations(const std::vector<std::pair<int, int>>& top_combinations,
	                                                               const uint8_t                           top_k,
	                                                               const PT*                               input_vector,
	                                                               const uint16_t                          input_vector_size,
	                                                               uint8_t&                                factor,
	                                                               uint8_t&                                exponent) {
		if (input_vector_size != config::VECTOR_SIZE) { throw std::invalid_argument("alp::encoder::find_best_exponent_factor_from_combinations: whole vectors (1024 values) only"); }
		if (top_k == 0 || top_k > config::MAX_K_COMBINATIONS || top_combinations.size() < top_k) {
			if (top_k == 0) { // the reference's loop does not run: both stay 0
				factor = exponent = 0;
				return;
			}
			throw std::invalid_argument("alp::encoder::find_best_exponent_factor_from_combinations: top_k must be 1..5 and covered by top_combinations");
		}
		auto&                 s = gpu::tls();
		alpgpu_rowgroup_state d {};
		d.scheme = ALPGPU_SCHEME_ALP;
		d.k      = top_k;
		for (size_t i = 0; i < top_k; ++i) {
			d.combos[2 * i]     = static_cast<uint8_t>(top_combinations[i].first);
			d.combos[2 * i + 1] = static_cast<uint8_t>(top_combinations[i].second);
		}
		gpu::h2d(s.at<PT>(s.IN), input_vector, gpu::abi<PT>::VEC_BYTES);
		gpu::h2d(s.at<alpgpu_rowgroup_state>(s.STATE), &d, sizeof(d));
		gpu::check(gpu::abi<PT>::encode_values(s.at<PT>(s.IN), s.at<alpgpu_rowgroup_state>(s.STATE), s.at<PT>(s.EXC), s.at<uint16_t>(s.POS), s.cnt(),
		                                       s.at<ST>(s.ENC), s.fac(), s.exp()),
		           "alpgpu_encode_values");
		uint8_t fe[2];
		gpu::d2h(fe, s.fac(), 2);
		factor   = fe[0];
		exponent = fe[1];
	}

	//! encoder.hpp:307-400 with an explicit (factor, exponent)
	static inline void encode_simdized(const PT*            input_vector,
	                                   PT*                  exceptions,
	                                   exp_p_t*             exceptions_positions,
	                                   exp_c_t*             exceptions_count,
	                                   ST*                  encoded_integers,
	                                   const factor_idx_t   factor_idx,
	                                   const exponent_idx_t exponent_idx) {
		auto&         s     = gpu::tls();
		const uint8_t fe[2] = {factor_idx, exponent_idx};
		gpu::h2d(s.at<PT>(s.IN), input_vector, gpu::abi<PT>::VEC_BYTES);
		gpu::h2d(s.fac(), fe, 2);
		gpu::check(gpu::abi<PT>::encode_simdized(s.at<PT>(s.IN), s.at<PT>(s.EXC), s.at<uint16_t>(s.POS), s.cnt(), s.at<ST>(s.ENC), s.fac(), s.exp()),
		           "alpgpu_encode_simdized");
		fetch_encoded(s, exceptions, exceptions_positions, exceptions_count, encoded_integers);
	}

	//! encoder.hpp:81-89: one value -> its encoded integer.  Like decode_value in alp/decoder.hpp it goes through the device (one value up, one
	//! integer down: a compatibility helper, not a fast path); SAFE = the sampling form with the ENCODING_UPPER_LIMIT sentinel.
	template <bool SAFE = true>
	static inline ST encode_value(const PT value, const uint8_t factor_idx, const uint8_t exponent_idx) {
		auto& s = gpu::tls();
		gpu::h2d(s.at<PT>(s.IN), &value, sizeof(PT));
		if constexpr (sizeof(PT) == 8) {
			gpu::check(alpgpu_encode_value_f64(gpu::context(), s.at<double>(s.IN), s.at<int64_t>(s.ENC), factor_idx, exponent_idx, SAFE ? 1 : 0, 1), "alpgpu_encode_value_f64");
		} else {
			gpu::check(alpgpu_encode_value_f32(gpu::context(), s.at<float>(s.IN), s.at<int32_t>(s.ENC), factor_idx, exponent_idx, SAFE ? 1 : 0, 1), "alpgpu_encode_value_f32");
		}
		ST out;
		gpu::d2h(&out, s.at<ST>(s.ENC), sizeof(ST));
		return out;
	}
	//! encoder.hpp:75-78
	static inline bool is_impossible_to_encode(const PT n) {
		return !(n == n) || n - n != PT(0) || n > ENCODING_UPPER_LIMIT || n < ENCODING_LOWER_LIMIT || (n == PT(0) && std::signbit(n));
	}
	//! encoder.hpp:91-106
	template <typename T>
	static inline uint8_t count_bits(T x) {
		uint8_t res = 0;
		while (x) {
			x >>= 1;
			++res;
		}
		return res;
	}
	template <typename T>
	static inline uint8_t count_bits(T max, T min) {
		const uint64_t delta = (static_cast<uint64_t>(max) - static_cast<uint64_t>(min));
		return count_bits<uint64_t>(delta);
	}

	//! min/max -> frame-of-reference base and bit width (encoder.hpp:109-120)
	static inline void analyze_ffor(const ST* input_vector, bw_t& bit_width, ST* base_for) {
		auto& s = gpu::tls();
		gpu::h2d(s.at<ST>(s.ENC), input_vector, gpu::abi<PT>::VEC_BYTES);
		gpu::check(gpu::abi<PT>::analyze_ffor(s.at<ST>(s.ENC), s.bw(), s.at<ST>(s.META + 8)), "alpgpu_analyze_ffor");
		gpu::d2h(&bit_width, s.bw(), 1);
		gpu::d2h(base_for, s.at<ST>(s.META + 8), sizeof(ST));
	}

private:
	static inline void fetch_encoded(gpu::scratch& s, PT* exceptions, uint16_t* positions, uint16_t* count, ST* encoded) {
		uint16_t n = 0;
		gpu::d2h(&n, s.cnt(), 2);
		gpu::d2h(encoded, s.at<ST>(s.ENC), gpu::abi<PT>::VEC_BYTES);
		if (n) {
			gpu::d2h(exceptions, s.at<PT>(s.EXC), static_cast<size_t>(n) * sizeof(PT));
			gpu::d2h(positions, s.at<uint16_t>(s.POS), static_cast<size_t>(n) * 2);
		}
		*count = n;
	}
};

} // namespace alp
#endif
