// fastlanes/ffor.hpp — ffor::ffor with the reference's signatures (include/fastlanes/ffor.hpp:7-15) for the word sizes
// of the reference: 64-bit (double: ALP integers, ALP_RD right parts), 32-bit (float: the same), 16-bit (ALP_RD left parts) and
// 8-bit (unused by the codec, part of the API).  GPU-backed.
#ifndef FASTLANES_FFOR_HPP
#define FASTLANES_FFOR_HPP
#include "alp/gpu_bridge.hpp"
#include <cstdint>

namespace fastlanes::generated::ffor::fallback::scalar {

inline void ffor(const uint64_t* __restrict in, uint64_t* __restrict out, uint8_t bw, const uint64_t* __restrict a_base_p) {
	if (bw == 0 || bw > 64) { return; } // the reference writes nothing for bw 0 and ignores unknown widths
	auto& s = alp::gpu::tls();
	alp::gpu::h2d(s.at<uint64_t>(s.ENC), in, 8192);
	alp::gpu::h2d(s.bw(), &bw, 1);
	alp::gpu::h2d(s.ffor_base(), a_base_p, 8);
	alp::gpu::check(alpgpu_ffor_i64(alp::gpu::context(), s.at<int64_t>(s.ENC), s.at<int64_t>(s.PACKED), 1024, s.bw(), s.ffor_base(), 1),
	                "alpgpu_ffor_i64");
	alp::gpu::d2h(out, s.at<uint64_t>(s.PACKED), static_cast<size_t>(bw) * 128);
}
inline void ffor(const int64_t* __restrict in, int64_t* __restrict out, uint8_t bw, const int64_t* __restrict a_base_p) {
	ffor(reinterpret_cast<const uint64_t*>(in), reinterpret_cast<uint64_t*>(out), bw, reinterpret_cast<const uint64_t*>(a_base_p));
}

inline void ffor(const uint16_t* __restrict in, uint16_t* __restrict out, uint8_t bw, const uint16_t* __restrict a_base_p) {
	if (bw == 0 || bw > 16) { return; }
	auto& s = alp::gpu::tls();
	alp::gpu::h2d(s.at<uint16_t>(s.LEFT), in, 2048);
	alp::gpu::h2d(s.bw(), &bw, 1);
	alp::gpu::h2d(s.base16(), a_base_p, 2);
	alp::gpu::check(alpgpu_ffor_u16(alp::gpu::context(), s.at<uint16_t>(s.LEFT), s.at<uint16_t>(s.PACKED_LEFT), 1024, s.bw(), s.base16(), 1),
	                "alpgpu_ffor_u16");
	alp::gpu::d2h(out, s.at<uint16_t>(s.PACKED_LEFT), static_cast<size_t>(bw) * 128);
}
inline void ffor(const int16_t* __restrict in, int16_t* __restrict out, uint8_t bw, const int16_t* __restrict a_base_p) {
	ffor(reinterpret_cast<const uint16_t*>(in), reinterpret_cast<uint16_t*>(out), bw, reinterpret_cast<const uint16_t*>(a_base_p));
}

inline void ffor(const uint32_t* __restrict in, uint32_t* __restrict out, uint8_t bw, const uint32_t* __restrict a_base_p) {
	if (bw == 0 || bw > 32) { return; }
	auto& s = alp::gpu::tls();
	alp::gpu::h2d(s.at<uint32_t>(s.ENC), in, 4096);
	alp::gpu::h2d(s.bw(), &bw, 1);
	alp::gpu::h2d(s.at<int32_t>(s.META + 8), a_base_p, 4);
	alp::gpu::check(alpgpu_ffor_i32(alp::gpu::context(), s.at<int32_t>(s.ENC), s.at<int32_t>(s.PACKED), 1024, s.bw(), s.at<int32_t>(s.META + 8), 1),
	                "alpgpu_ffor_i32");
	alp::gpu::d2h(out, s.at<uint32_t>(s.PACKED), static_cast<size_t>(bw) * 128);
}
inline void ffor(const int32_t* __restrict in, int32_t* __restrict out, uint8_t bw, const int32_t* __restrict a_base_p) {
	ffor(reinterpret_cast<const uint32_t*>(in), reinterpret_cast<uint32_t*>(out), bw, reinterpret_cast<const uint32_t*>(a_base_p));
}

inline void ffor(const uint8_t* __restrict in, uint8_t* __restrict out, uint8_t bw, const uint8_t* __restrict a_base_p) {
	if (bw == 0 || bw > 8) { return; }
	auto& s = alp::gpu::tls();
	alp::gpu::h2d(s.at<uint8_t>(s.LEFT), in, 1024);
	alp::gpu::h2d(s.bw(), &bw, 1);
	alp::gpu::h2d(s.at<uint8_t>(s.META + 24), a_base_p, 1);
	alp::gpu::check(alpgpu_ffor_u8(alp::gpu::context(), s.at<uint8_t>(s.LEFT), s.at<uint8_t>(s.PACKED_LEFT), 1024, s.bw(), s.at<uint8_t>(s.META + 24), 1),
	                "alpgpu_ffor_u8");
	alp::gpu::d2h(out, s.at<uint8_t>(s.PACKED_LEFT), static_cast<size_t>(bw) * 128);
}
inline void ffor(const int8_t* __restrict in, int8_t* __restrict out, uint8_t bw, const int8_t* __restrict a_base_p) {
	ffor(reinterpret_cast<const uint8_t*>(in), reinterpret_cast<uint8_t*>(out), bw, reinterpret_cast<const uint8_t*>(a_base_p));
}

} // namespace fastlanes::generated::ffor::fallback::scalar

namespace ffor = fastlanes::generated::ffor::fallback::scalar;
#endif
