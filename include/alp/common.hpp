// alp/common.hpp — index / counter typedefs (reference include/alp/common.hpp:8-16).
#ifndef ALP_COMMON_HPP
#define ALP_COMMON_HPP
#include <cstdint>
namespace alp {
using bw_t           = uint8_t;  // bit width
using exp_c_t        = uint16_t; // exception count
using exp_p_t        = uint16_t; // exception position
using factor_idx_t   = uint8_t;
using exponent_idx_t = uint8_t;
} // namespace alp
#endif
