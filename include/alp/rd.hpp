// alp/rd.hpp — alp::rd_encoder<PT> with the reference's signatures (include/alp/rd.hpp:109-185), computed on the GPU.
//
// Dictionary order.  The reference builds the left-part histogram in a std::unordered_map and std::sort()s it by count
// only, so the order of equally frequent left parts is whatever libstdc++ produces.  The device builder
// replays those container internals (alp_amd/csrc/rd_dictionary_order.hpp) and returns the reference's dictionary entry
// by entry.  Through this per-vector header, left parts outside the dictionary get index = dictionary size (the reference
// stores a map position there, which no decoder reads; the batch C ABI reproduces it, see alpgpu_column.d_rd_order).  A state produced by the reference itself can be passed to encode()/decode() unchanged.
#ifndef ALP_RD_HPP
#define ALP_RD_HPP
#include "alp/common.hpp"
#include "alp/constants.hpp"
#include "alp/encoder.hpp"
#include "alp/gpu_bridge.hpp"
#include "alp/sampler.hpp"
#include <stdexcept>

namespace alp {

template <class PT>
struct rd_encoder {
	using UT                                     = typename inner_t<PT>::ut;
	static constexpr uint8_t EXACT_TYPE_BIT_SIZE = sizeof(UT) * 8;

	//! rd.hpp:23-31: the bits per value a cut is estimated to cost within a sample (plain arithmetic on four numbers: no codec work in it)
	static inline double estimate_compression_size(const bw_t right_bit_width, const bw_t left_bit_width, const exp_c_t exceptions_count, const uint64_t sample_count) {
		const double exceptions_size = exceptions_count * (RD_EXCEPTION_POSITION_SIZE + RD_EXCEPTION_SIZE);
		return right_bit_width + left_bit_width + (exceptions_size / static_cast<double>(sample_count));
	}

	//! rd.hpp:33-87: the dictionary of ONE cut position and its estimated size; PERSIST_DICT also writes it into the state.  The device runs the
	//! kernel of find_best_dictionary with every other cut ruled out (alpgpu_rd_dictionary_for_cut_*): same left-part histogram, same
	//! libstdc++ order of equally frequent parts.  The cut must be one find_best_dictionary can choose (1..16 bits: rd.hpp:92); anything else
	//! throws std::invalid_argument (the reference would shift by whatever it is given).
	template <bool PERSIST_DICT>
	static double build_left_parts_dictionary(const PT* in_p, bw_t right_bit_width, state<PT>& stt) {
		if (right_bit_width >= EXACT_TYPE_BIT_SIZE || EXACT_TYPE_BIT_SIZE - right_bit_width > config::CUTTING_LIMIT) {
			throw std::invalid_argument("alp::rd_encoder::build_left_parts_dictionary: right_bit_width must leave a left part of 1..16 bits");
		}
		auto&        s       = gpu::tls();
		const size_t n       = stt.sampled_values_n;
		gpu::h2d(s.at<PT>(s.SAMPLES), in_p, n * sizeof(PT)); // exactly the stt.sampled_values_n values the reference reads (rd.hpp:40): the kernel is given n and reads no further
		gpu::check(gpu::abi<PT>::rd_dictionary_for_cut(s.at<PT>(s.SAMPLES), static_cast<uint32_t>(n), right_bit_width, s.at<alpgpu_rowgroup_state>(s.STATE),
		                                               s.at<double>(s.META + 32)),
		           "alpgpu_rd_dictionary_for_cut");
		double estimate = 0.0;
		gpu::d2h(&estimate, s.at<double>(s.META + 32), sizeof(double));
		if (PERSIST_DICT) {
			alpgpu_rowgroup_state d {};
			gpu::d2h(&d, s.at<alpgpu_rowgroup_state>(s.STATE), sizeof(d));
			persist(d, stt);
		}
		return estimate;
	}

	//! rd.hpp:180-185: sample the rowgroup, choose the cut and the dictionary
	static inline void init(const PT* data_column, size_t column_offset, size_t tuples_count, PT* sample_arr, state<PT>& stt) {
		stt.scheme           = Scheme::ALP_RD;
		stt.sampled_values_n = sampler::first_level_sample<PT>(data_column, column_offset, tuples_count, sample_arr);
		find_best_dictionary(sample_arr, stt);
	}

	//! rd.hpp:89-104
	static inline void find_best_dictionary(const PT* smp_arr, state<PT>& stt) {
		// the device decision covers ALP and ALP_RD in one pass; ask it to judge these samples as an RD rowgroup
		auto&        s       = gpu::tls();
		const size_t n       = stt.sampled_values_n;
		const size_t n_block = (n + config::SAMPLES_PER_VECTOR - 1) / config::SAMPLES_PER_VECTOR;
		const size_t n_up    = n < config::SAMPLES_PER_VECTOR ? n : n_block * config::SAMPLES_PER_VECTOR;
		gpu::h2d(s.at<PT>(s.SAMPLES), smp_arr, n_up * sizeof(PT));
		gpu::check(gpu::abi<PT>::rd_state_from_samples(s.at<PT>(s.SAMPLES), static_cast<uint32_t>(n), s.at<alpgpu_rowgroup_state>(s.STATE)),
		           "alpgpu_rd_state_from_samples");
		alpgpu_rowgroup_state d {};
		gpu::d2h(&d, s.at<alpgpu_rowgroup_state>(s.STATE), sizeof(d));
		persist(d, stt);
	}

private:
	//! the fields build_left_parts_dictionary<true> leaves in the state (rd.hpp:69-84)
	static inline void persist(const alpgpu_rowgroup_state& d, state<PT>& stt) {
		stt.right_bit_width              = d.rd_rbw;
		stt.left_bit_width               = d.rd_lbw;
		stt.actual_dictionary_size       = d.rd_dict_size;
		stt.actual_dictionary_size_bytes = d.rd_dict_size * DICTIONARY_ELEMENT_SIZE_BYTES;
		stt.left_parts_dict_map.clear();
		for (size_t i = 0; i < config::MAX_RD_DICTIONARY_SIZE; ++i) {
			stt.left_parts_dict[i] = d.rd_dict[i];
			if (i < d.rd_dict_size) { stt.left_parts_dict_map.insert({d.rd_dict[i], static_cast<uint16_t>(i)}); }
		}
	}

public:

	//! rd.hpp:109-147
	static inline void encode(const PT*  dbl_arr,
	                          uint16_t*  exceptions,
	                          uint16_t*  exception_positions,
	                          uint16_t*  exceptions_count_p,
	                          UT*        right_parts,
	                          uint16_t*  left_parts,
	                          state<PT>& stt) {
		auto&                       s = gpu::tls();
		const alpgpu_rowgroup_state d = gpu::to_device_state(stt);
		gpu::h2d(s.at<PT>(s.IN), dbl_arr, gpu::abi<PT>::VEC_BYTES);
		gpu::h2d(s.at<alpgpu_rowgroup_state>(s.STATE), &d, sizeof(d));
		gpu::check(gpu::abi<PT>::rd_encode(s.at<PT>(s.IN), s.at<alpgpu_rowgroup_state>(s.STATE), s.at<uint16_t>(s.EXC), s.at<uint16_t>(s.POS), s.cnt(),
		                                   s.at<UT>(s.ENC), s.at<uint16_t>(s.LEFT)),
		           "alpgpu_rd_encode_vectors");
		uint16_t n = 0;
		gpu::d2h(&n, s.cnt(), 2);
		gpu::d2h(right_parts, s.at<UT>(s.ENC), gpu::abi<PT>::VEC_BYTES);
		gpu::d2h(left_parts, s.at<uint16_t>(s.LEFT), 2048);
		if (n) {
			gpu::d2h(exceptions, s.at<uint16_t>(s.EXC), static_cast<size_t>(n) * 2);
			gpu::d2h(exception_positions, s.at<uint16_t>(s.POS), static_cast<size_t>(n) * 2);
		}
		stt.exceptions_count  = n;
		exceptions_count_p[0] = n;
	}

	//! rd.hpp:152-178
	static inline void decode(PT*        a_out,
	                          UT*        unffor_right_arr,
	                          uint16_t*  unffor_left_arr,
	                          uint16_t*  exceptions,
	                          uint16_t*  exceptions_positions,
	                          uint16_t*  exceptions_count,
	                          state<PT>& stt) {
		auto&                       s = gpu::tls();
		const alpgpu_rowgroup_state d = gpu::to_device_state(stt);
		const uint16_t              n = exceptions_count[0];
		gpu::h2d(s.at<UT>(s.ENC), unffor_right_arr, gpu::abi<PT>::VEC_BYTES);
		gpu::h2d(s.at<uint16_t>(s.LEFT), unffor_left_arr, 2048);
		gpu::h2d(s.at<alpgpu_rowgroup_state>(s.STATE), &d, sizeof(d));
		gpu::h2d(s.cnt(), &n, 2);
		if (n) {
			gpu::h2d(s.at<uint16_t>(s.EXC), exceptions, static_cast<size_t>(n) * 2);
			gpu::h2d(s.at<uint16_t>(s.POS), exceptions_positions, static_cast<size_t>(n) * 2);
		}
		gpu::check(gpu::abi<PT>::rd_decode(s.at<PT>(s.OUT), s.at<UT>(s.ENC), s.at<uint16_t>(s.LEFT), s.at<alpgpu_rowgroup_state>(s.STATE),
		                                   s.at<uint16_t>(s.EXC), s.at<uint16_t>(s.POS), s.cnt()),
		           "alpgpu_rd_decode_vectors");
		gpu::d2h(a_out, s.at<PT>(s.OUT), gpu::abi<PT>::VEC_BYTES);
	}
};

} // namespace alp
#endif
