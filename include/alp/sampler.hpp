// alp/sampler.hpp — first-level sampling: WHICH values of a rowgroup are looked at (reference include/alp/sampler.hpp:14-52).
// Pure index arithmetic on the caller's host column (no codec arithmetic), kept on the host so that columns ending in
// a partial vector follow the reference's rules; the samples are then judged on the GPU (alpgpu_state_from_samples_f64).
// The batch path (alpgpu_rowgroup_init_f64) gathers the same positions on the device for whole-vector columns.
#ifndef ALP_SAMPLER_HPP
#define ALP_SAMPLER_HPP
#include "alp/config.hpp"
#include <cstddef>

namespace alp::sampler {

template <class T>
inline size_t first_level_sample(const T* data, const size_t data_offset, const size_t data_size, T* data_sample) {
	const size_t left     = data_size - data_offset;
	const size_t portion  = left < config::ROWGROUP_SIZE ? left : config::ROWGROUP_SIZE;
	const size_t n_vec    = (portion + config::VECTOR_SIZE - 1) / config::VECTOR_SIZE;
	size_t       n_taken  = 0;
	size_t       cursor   = data_offset;
	for (size_t v = 0; v < n_vec; ++v) {
		const size_t remaining = data_size - cursor;
		const size_t n_values  = remaining < config::VECTOR_SIZE ? remaining : config::VECTOR_SIZE;
		const bool   selected  = (v % config::ROWGROUP_SAMPLES_JUMP) == 0;
		// short tail vectors are skipped unless nothing has been sampled yet
		const bool   too_short = n_values < config::SAMPLES_PER_VECTOR && n_taken != 0;
		if (selected && !too_short) {
			size_t step = (n_values + config::SAMPLES_PER_VECTOR - 1) / config::SAMPLES_PER_VECTOR;
			if (step < 1) { step = 1; }
			for (size_t i = 0; i < n_values; i += step) { data_sample[n_taken++] = data[cursor + i]; }
		}
		cursor += n_values;
	}
	return n_taken;
}

} // namespace alp::sampler
#endif
