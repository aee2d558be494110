// fastlanes/unffor.hpp — unffor::unffor with the reference's signatures (include/fastlanes/unffor.hpp:7-15), 64-, 32-, 16- and 8-bit lanes.
#ifndef FASTLANES_UNFFOR_HPP
#define FASTLANES_UNFFOR_HPP
#include "alp/gpu_bridge.hpp"
#include <cstdint>

namespace fastlanes::generated::unffor::fallback::scalar {

inline void unffor(const uint64_t* __restrict in, uint64_t* __restrict out, uint8_t bw, const uint64_t* __restrict a_base_p) {
	if (bw > 64) { return; }
	auto& s = alp::gpu::tls();
	if (bw) { alp::gpu::h2d(s.at<uint64_t>(s.PACKED), in, static_cast<size_t>(bw) * 128); }
	alp::gpu::h2d(s.bw(), &bw, 1);
	alp::gpu::h2d(s.ffor_base(), a_base_p, 8);
	alp::gpu::check(alpgpu_unffor_i64(alp::gpu::context(), s.at<int64_t>(s.PACKED), 1024, s.at<int64_t>(s.ENC), s.bw(), s.ffor_base(), 1),
	                "alpgpu_unffor_i64");
	alp::gpu::d2h(out, s.at<uint64_t>(s.ENC), 8192);
}
inline void unffor(const int64_t* __restrict in, int64_t* __restrict out, uint8_t bw, const int64_t* __restrict a_base_p) {
	unffor(reinterpret_cast<const uint64_t*>(in), reinterpret_cast<uint64_t*>(out), bw, reinterpret_cast<const uint64_t*>(a_base_p));
}

inline void unffor(const uint16_t* __restrict in, uint16_t* __restrict out, uint8_t bw, const uint16_t* __restrict a_base_p) {
	if (bw > 16) { return; }
	auto& s = alp::gpu::tls();
	if (bw) { alp::gpu::h2d(s.at<uint16_t>(s.PACKED_LEFT), in, static_cast<size_t>(bw) * 128); }
	alp::gpu::h2d(s.bw(), &bw, 1);
	alp::gpu::h2d(s.base16(), a_base_p, 2);
	alp::gpu::check(alpgpu_unffor_u16(alp::gpu::context(), s.at<uint16_t>(s.PACKED_LEFT), 1024, s.at<uint16_t>(s.LEFT), s.bw(), s.base16(), 1),
	                "alpgpu_unffor_u16");
	alp::gpu::d2h(out, s.at<uint16_t>(s.LEFT), 2048);
}
inline void unffor(const int16_t* __restrict in, int16_t* __restrict out, uint8_t bw, const int16_t* __restrict a_base_p) {
	unffor(reinterpret_cast<const uint16_t*>(in), reinterpret_cast<uint16_t*>(out), bw, reinterpret_cast<const uint16_t*>(a_base_p));
}

inline void unffor(const uint32_t* __restrict in, uint32_t* __restrict out, uint8_t bw, const uint32_t* __restrict a_base_p) {
	if (bw > 32) { return; }
	auto& s = alp::gpu::tls();
	if (bw) { alp::gpu::h2d(s.at<uint32_t>(s.PACKED), in, static_cast<size_t>(bw) * 128); }
	alp::gpu::h2d(s.bw(), &bw, 1);
	alp::gpu::h2d(s.at<int32_t>(s.META + 8), a_base_p, 4);
	alp::gpu::check(alpgpu_unffor_i32(alp::gpu::context(), s.at<int32_t>(s.PACKED), 1024, s.at<int32_t>(s.ENC), s.bw(), s.at<int32_t>(s.META + 8), 1),
	                "alpgpu_unffor_i32");
	alp::gpu::d2h(out, s.at<uint32_t>(s.ENC), 4096);
}
inline void unffor(const int32_t* __restrict in, int32_t* __restrict out, uint8_t bw, const int32_t* __restrict a_base_p) {
	unffor(reinterpret_cast<const uint32_t*>(in), reinterpret_cast<uint32_t*>(out), bw, reinterpret_cast<const uint32_t*>(a_base_p));
}

inline void unffor(const uint8_t* __restrict in, uint8_t* __restrict out, uint8_t bw, const uint8_t* __restrict a_base_p) {
	if (bw > 8) { return; }
	auto& s = alp::gpu::tls();
	if (bw) { alp::gpu::h2d(s.at<uint8_t>(s.PACKED_LEFT), in, static_cast<size_t>(bw) * 128); }
	alp::gpu::h2d(s.bw(), &bw, 1);
	alp::gpu::h2d(s.at<uint8_t>(s.META + 24), a_base_p, 1);
	alp::gpu::check(alpgpu_unffor_u8(alp::gpu::context(), s.at<uint8_t>(s.PACKED_LEFT), 1024, s.at<uint8_t>(s.LEFT), s.bw(), s.at<uint8_t>(s.META + 24), 1),
	                "alpgpu_unffor_u8");
	alp::gpu::d2h(out, s.at<uint8_t>(s.LEFT), 1024);
}
inline void unffor(const int8_t* __restrict in, int8_t* __restrict out, uint8_t bw, const int8_t* __restrict a_base_p) {
	unffor(reinterpret_cast<const uint8_t*>(in), reinterpret_cast<uint8_t*>(out), bw, reinterpret_cast<const uint8_t*>(a_base_p));
}

} // namespace fastlanes::generated::unffor::fallback::scalar

namespace unffor = fastlanes::generated::unffor::fallback::scalar;
#endif
