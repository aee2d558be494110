// alp/common.hpp — small integer aliases used across the vector API (same names as the reference's common.hpp so that
// caller code compiles unchanged).
#ifndef ALP_COMMON_HPP
#define ALP_COMMON_HPP
#include <cstdint>

namespace alp {

// per-vector metadata
using bw_t           = uint8_t;  //!< FFOR bit width, 0..64
using factor_idx_t   = uint8_t;  //!< index into the 10^-f table used on encode / 10^f on decode
using exponent_idx_t = uint8_t;  //!< index into the 10^e table

// exception bookkeeping: a vector holds at most 1024 exceptions, positions are 0..1023
using exp_p_t = uint16_t; //!< position of an exception inside its vector
using exp_c_t = uint16_t; //!< number of exceptions of a vector

static_assert(sizeof(exp_p_t) == 2 && sizeof(exp_c_t) == 2, "the exception record layout of include/alpgpu.h uses 16-bit positions");

} // namespace alp
#endif // ALP_COMMON_HPP
