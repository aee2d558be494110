// alp/constants.hpp — public constants of the codec (reference include/alp/constants.hpp:10-25, :30-64 float, :66-154 double).
// Host code only needs them for sizing and bookkeeping; the arithmetic runs on the device, whose tables (alp_amd/csrc/alp_device.hpp,
// alp_device_f32.hpp) are what the kernels use — FRAC_ARR / EXP_ARR / FACT_ARR below are host copies for source compatibility, the same
// numbers (powers of ten as correctly rounded literals / exact integers), compared with the device's by tests/test_dropin_gpu.py.
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
	static inline constexpr uint64_t EXPONENTIAL_BITS_MASK   = 0xFFE0000000000000ULL; // what the reference's 65-digit literal evaluates to
	// 10^-i (i = 0..20), 10^i (i = 0..23), 10^i as integers (i = 0..18)
	static inline constexpr double  FRAC_ARR[] = {1.0,   1e-1,  1e-2,  1e-3,  1e-4,  1e-5,  1e-6,  1e-7,  1e-8,  1e-9, 1e-10,
	                                              1e-11, 1e-12, 1e-13, 1e-14, 1e-15, 1e-16, 1e-17, 1e-18, 1e-19, 1e-20};
	static inline constexpr double  EXP_ARR[]  = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
	                                              1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22, 1e23};
	static inline constexpr int64_t FACT_ARR[] = {1LL,
	                                              10LL,
	                                              100LL,
	                                              1000LL,
	                                              10000LL,
	                                              100000LL,
	                                              1000000LL,
	                                              10000000LL,
	                                              100000000LL,
	                                              1000000000LL,
	                                              10000000000LL,
	                                              100000000000LL,
	                                              1000000000000LL,
	                                              10000000000000LL,
	                                              100000000000000LL,
	                                              1000000000000000LL,
	                                              10000000000000000LL,
	                                              100000000000000000LL,
	                                              1000000000000000000LL};
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
	// 10^-i, 10^i (i = 0..10) as floats, 10^i as int32 (i = 0..9: the reference's table has ten entries; (e, f) = (10, 10) reads past it,
	// alp_amd/csrc/alp_device_f32.hpp)
	static inline constexpr float   FRAC_ARR[] = {1.0f, 0.1f, 0.01f, 0.001f, 0.0001f, 0.00001f, 0.000001f, 0.0000001f, 0.00000001f, 0.000000001f, 0.0000000001f};
	static inline constexpr float   EXP_ARR[]  = {1.0f, 10.0f, 100.0f, 1000.0f, 10000.0f, 100000.0f, 1000000.0f, 10000000.0f, 100000000.0f, 1000000000.0f, 10000000000.0f};
	static inline constexpr int32_t FACT_ARR[] = {1, 10, 100, 1000, 10000, 100000, 1000000, 10000000, 100000000, 1000000000};
};

} // namespace alp
#endif
