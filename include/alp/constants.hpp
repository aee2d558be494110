// alp/constants.hpp — public constants of the codec (reference include/alp/constants.hpp:10-25, :30-64 float, :66-154 double).
// Host code only needs them for sizing and bookkeeping; the arithmetic tables live in the device code
// (alp_amd/csrc/alp_device.hpp) and are bit-identical.
#ifndef ALP_CONSTANTS_HPP
#define ALP_CONSTANTS_HPP
#include "alp/config.hpp"
#include <cstdint>

namespace alp {

enum class Scheme : uint8_t { INVALID, ALP_RD, ALP };

inline constexpr uint8_t SAMPLING_EARLY_EXIT_THRESHOLD    = 2;
inline constexpr double  ENCODING_UPPER_LIMIT             = 9223372036854774784.0;
inline constexpr double  ENCODING_LOWER_LIMIT             = -9223372036854774784.0;
inline constexpr uint8_t DICTIONARY_ELEMENT_SIZE_BYTES    = 2;
inline constexpr uint8_t RD_EXCEPTION_POSITION_SIZE       = 16;
inline constexpr uint8_t RD_EXCEPTION_POSITION_SIZE_BYTES = RD_EXCEPTION_POSITION_SIZE / 8;
inline constexpr uint8_t EXCEPTION_POSITION_SIZE          = 16;
inline constexpr uint8_t EXCEPTION_POSITION_SIZE_BYTES    = EXCEPTION_POSITION_SIZE / 8;
inline constexpr uint8_t RD_EXCEPTION_SIZE                = 16;
inline constexpr uint8_t RD_EXCEPTION_SIZE_BYTES          = RD_EXCEPTION_SIZE / 8;

template <class T>
struct Constants {};

template <>
struct Constants<double> {
	static inline constexpr size_t   RD_SIZE_THRESHOLD_LIMIT = 48 * config::SAMPLES_PER_VECTOR;
	static inline constexpr double   MAGIC_NUMBER            = 6755399441055744.0; // 2^52 + 2^51
	static inline constexpr uint8_t  EXCEPTION_SIZE          = 64;
	static inline constexpr uint8_t  EXCEPTION_SIZE_BYTES    = EXCEPTION_SIZE / 8;
	static inline constexpr uint8_t  MAX_EXPONENT            = 18;
	static inline constexpr uint64_t NEGATIVE_ZERO           = 0x8000000000000000ULL;
	static inline constexpr uint64_t POSITIVE_INF            = 0x7FF0000000000000ULL;
	static inline constexpr uint64_t NEGATIVE_INF            = 0xFFF0000000000000ULL;
	static inline constexpr uint64_t SIGN_BIT_MASK           = 0x7FFFFFFFFFFFFFFFULL;
};

template <>
struct Constants<float> {
	static inline constexpr size_t   RD_SIZE_THRESHOLD_LIMIT = 22 * config::SAMPLES_PER_VECTOR;
	static inline constexpr float    MAGIC_NUMBER            = 12582912.0f; // 2^23 + 2^22
	static inline constexpr uint8_t  EXCEPTION_SIZE          = 32;
	static inline constexpr uint8_t  EXCEPTION_SIZE_BYTES    = EXCEPTION_SIZE / 8;
	static inline constexpr uint8_t  MAX_EXPONENT            = 10;
	static inline constexpr uint32_t NEGATIVE_ZERO           = 0x80000000u;
	static inline constexpr uint32_t POSITIVE_INF            = 0x7F800000u;
	static inline constexpr uint32_t NEGATIVE_INF            = 0xFF800000u;
	static inline constexpr uint32_t SIGN_BIT_MASK           = 0x7FFFFFFFu;
};

} // namespace alp
#endif
