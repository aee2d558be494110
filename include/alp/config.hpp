// alp/config.hpp — the codec's compile-time parameters under the names callers of the reference use
// (alp::config::VECTOR_SIZE, ...).  The values are part of the format: the device code (alp_amd/csrc/alp_device.hpp,
// init_kernels.hip) hard-codes the same numbers, and tests/test_abi.py checks the two stay in step.
#ifndef ALP_CONFIG_HPP
#define ALP_CONFIG_HPP
#include <cstddef>

// How the GPU path uses them:
//   * one wavefront (64 lanes) owns one VECTOR_SIZE-value vector; a lane holds 16 of its values (8 pairs);
//   * one workgroup of the init kernel owns one rowgroup and samples vectors 0, 12, ..., 96 of it (9 sampled vectors x 32
//     values = 288 samples) — exactly the positions alp::sampler::first_level_sample picks for whole-vector columns;
//   * descriptors store the rowgroup's (exponent, factor) candidates in 10 bytes (MAX_K_COMBINATIONS pairs) and the ALP_RD
//     dictionary in 16 bytes (MAX_RD_DICTIONARY_SIZE entries of 16 bits): struct alpgpu_rowgroup_state in alpgpu.h.
namespace alp::config {

// geometry: 1024-value vectors, 100 of them per rowgroup
inline constexpr size_t VECTOR_SIZE            = 1u << 10;
inline constexpr size_t N_VECTORS_PER_ROWGROUP = 100;
inline constexpr size_t ROWGROUP_SIZE          = VECTOR_SIZE * N_VECTORS_PER_ROWGROUP;

// first-level sampling: 8 equidistant vectors of a full rowgroup -> every 12th vector, 32 values from each
inline constexpr size_t ROWGROUP_VECTOR_SAMPLES = 8;
inline constexpr size_t ROWGROUP_SAMPLES_JUMP   = ROWGROUP_SIZE / ROWGROUP_VECTOR_SAMPLES / VECTOR_SIZE;
inline constexpr size_t SAMPLES_PER_VECTOR      = 32;
static_assert(ROWGROUP_SAMPLES_JUMP == 12, "the GPU sampler strides by 12 vectors");

// at most 5 (exponent, factor) candidates survive the rowgroup search
inline constexpr size_t MAX_K_COMBINATIONS = 5;

// ALP_RD: up to 16 cut positions are tried; the left-part dictionary has at most 2^3 entries
inline constexpr size_t CUTTING_LIMIT          = 16;
inline constexpr size_t MAX_RD_DICT_BIT_WIDTH  = 3;
inline constexpr size_t MAX_RD_DICTIONARY_SIZE = size_t {1} << MAX_RD_DICT_BIT_WIDTH;

} // namespace alp::config
#endif // ALP_CONFIG_HPP
