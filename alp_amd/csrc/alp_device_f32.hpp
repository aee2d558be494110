// alp_device_f32.hpp — single-precision counterparts of alp_device.hpp.
//
// Reference: Constants<float> include/alp/constants.hpp:30-64; alp::encoder<float>::encode_value include/alp/encoder.hpp:81-89;
// alp::decoder<float>::decode_value include/alp/decoder.hpp:128-131; 32-bit FFOR src/fastlanes_generated_ffor.cpp:1776-7378.
//
// As-built behaviour of the reference that is restated here (it is undefined behaviour in C++; the reference builds with Clang
// only, CMakeLists.txt:50-52, and oracle/alp_oracle_f32.c pins what that build does — see its header, U1..U4):
//   * static_cast<int32_t>(float) out of range / NaN = INT32_MIN (cvttss2si);
//   * the SAFE branch of encode_value<true> is compiled away for float, so the sampling estimators use the same
//     arithmetic as the encoder proper (a -0.0f sample encodes to 0 and counts as encodable);
//   * int32 products wrap modulo 2^32;
//   * FACT_ARR[10] is read out of bounds for (e,f) = (10,10); no observable result depends on it (every value whose
//     encoded integer is non-zero is an exception for any multiplier); this file uses 10^10 mod 2^32.
//
// FastLanes u32 layout: value i -> lane32 = i & 31, row = i >> 5; the lane's stream is the LSB-first concatenation of its
// 32 bw-bit fields; stream word k lives at packed[32*k + lane32].  Four consecutive values 4t..4t+3 share the row t >> 3
// and are the 16-byte unit (t & 7) of every stream-word row: unit index 8*k + (t & 7).
#pragma once
#include "alp_device.hpp"

namespace alpgpu {

__device__ __constant__ const float kFracArrF[11] = {1.0f, 0.1f, 0.01f, 0.001f, 0.0001f, 0.00001f, 0.000001f, 0.0000001f,
                                                     0.00000001f, 0.000000001f, 0.0000000001f};
__device__ __constant__ const float kExpArrF[11]  = {1.0f, 10.0f, 100.0f, 1000.0f, 10000.0f, 100000.0f, 1000000.0f, 10000000.0f,
                                                     100000000.0f, 1000000000.0f, 10000000000.0f};
__device__ __constant__ const uint32_t kFactArrF[11] = {1u, 10u, 100u, 1000u, 10000u, 100000u, 1000000u, 10000000u,
                                                        100000000u, 1000000000u, 1410065408u};

constexpr float kMagicF      = 12582912.0f;             // 2^23 + 2^22, constants.hpp:34
constexpr float kUpperLimitF = 9223372036854775808.0f;  // (float)ENCODING_UPPER_LIMIT, encoder.hpp:334

__device__ __forceinline__ uint32_t bw_mask32(int bw) { return bw >= 32 ? ~0u : ((1u << bw) - 1u); }

// static_cast<int32_t>(float) as x86-64 executes it
__device__ __forceinline__ int32_t cast32_x86(float x) {
	const bool in_range = (x > -2147483904.0f) && (x < 2147483648.0f); // NaN fails both
	return in_range ? static_cast<int32_t>(x) : INT32_MIN;
}

// include/alp/encoder.hpp:81-89, PT = float (SAFE and !SAFE coincide, see header).  -ffp-contract=off: each op rounds.
__device__ __forceinline__ int32_t encode_value_f32(float v, float exp10, float frac10) {
	float t = v * exp10;
	t       = t * frac10;
	t       = t + kMagicF;
	t       = t - kMagicF;
	return cast32_x86(t);
}

// include/alp/decoder.hpp:128-131, PT = float
__device__ __forceinline__ float decode_value_f32(int32_t enc, uint32_t fact, float frac) {
	const int32_t m = static_cast<int32_t>(static_cast<uint32_t>(enc) * fact);
	return static_cast<float>(m) * frac;
}

// include/alp/encoder.hpp:91-106, 32-bit branch
__device__ __forceinline__ int count_bits32(int32_t mx, int32_t mn) {
	const uint32_t d = static_cast<uint32_t>(mx) - static_cast<uint32_t>(mn);
	return d == 0 ? 0 : 32 - __builtin_clz(d);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float    f32x4 __attribute__((ext_vector_type(4)));

// FastLanes u32 unpack of the four values 4t..4t+3 (row = t >> 3, a = t & 7) from a staged copy of the vector's packed
// words viewed as 16-byte units; one extra unit row past the end must be readable (content irrelevant).
__device__ __forceinline__ u32x4 unpack_quad_u32(const u32x4* __restrict__ units, int bw, uint32_t mask, int row, int a) {
	const int   p  = row * bw;
	const int   k  = p >> 5;
	const int   s  = p & 31;
	const u32x4 w0 = units[8 * k + a];
	const u32x4 w1 = units[8 * k + 8 + a];
	// the 64-bit funnel {w1, w0} >> s as one v_alignbit_b32 per value (amount modulo 32: s = 0 yields w0)
	u32x4 r;
#pragma unroll
	for (int c = 0; c < 4; ++c) { r[c] = __builtin_amdgcn_alignbit(w1[c], w0[c], static_cast<uint32_t>(s)) & mask; }
	return r;
}

} // namespace alpgpu
