// alp/config.hpp — compile-time codec parameters; same names and values as the reference's
// include/alp/config.hpp:11-26 (they are part of the format).
#ifndef ALP_CONFIG_HPP
#define ALP_CONFIG_HPP
#include <cstddef>

namespace alp::config {
inline constexpr size_t VECTOR_SIZE             = 1024;
inline constexpr size_t N_VECTORS_PER_ROWGROUP  = 100;
inline constexpr size_t ROWGROUP_SIZE           = N_VECTORS_PER_ROWGROUP * VECTOR_SIZE;
inline constexpr size_t ROWGROUP_VECTOR_SAMPLES = 8;
inline constexpr size_t ROWGROUP_SAMPLES_JUMP   = (ROWGROUP_SIZE / ROWGROUP_VECTOR_SAMPLES) / VECTOR_SIZE; // 12
inline constexpr size_t SAMPLES_PER_VECTOR      = 32;
inline constexpr size_t MAX_K_COMBINATIONS      = 5;
inline constexpr size_t CUTTING_LIMIT           = 16;
inline constexpr size_t MAX_RD_DICT_BIT_WIDTH   = 3;
inline constexpr size_t MAX_RD_DICTIONARY_SIZE  = (1 << MAX_RD_DICT_BIT_WIDTH);
} // namespace alp::config
#endif
