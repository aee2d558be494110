// alp/falp.hpp — generated::falp::fallback::scalar::falp with the reference's signatures (include/alp/falp.hpp:10-26 double,
// :28-44 float): fused unFFOR + decode of one vector, on the GPU (alpgpu_falp_f64 / alpgpu_falp_f32).  The widest case
// (bw = 64, bw = 32 for float) follows unffor + decode: the reference's generated kernels for it are broken and
// unreachable through the codec (tests/test_oracle_f32.py records the float one).
#ifndef ALP_FALP_HPP
#define ALP_FALP_HPP
#include "alp/gpu_bridge.hpp"
#include <cstdint>

namespace generated { namespace falp { namespace fallback { namespace scalar {

inline void falp(const uint64_t* __restrict in, double* __restrict out, uint8_t bw, const uint64_t* __restrict a_base_p, uint8_t factor,
                 uint8_t exponent) {
	if (bw > 64) { return; }
	auto&         s       = alp::gpu::tls();
	const uint8_t meta[3] = {bw, factor, exponent};
	alp::gpu::h2d(s.bw(), meta, 3);
	alp::gpu::h2d(s.ffor_base(), a_base_p, 8);
	if (bw) { alp::gpu::h2d(s.at<uint64_t>(s.PACKED), in, static_cast<size_t>(bw) * 128); }
	alp::gpu::check(alpgpu_falp_f64(alp::gpu::context(), s.at<int64_t>(s.PACKED), 1024, s.at<double>(s.OUT), s.bw(), s.ffor_base(), s.fac(),
	                                s.exp(), 1),
	                "alpgpu_falp_f64");
	alp::gpu::d2h(out, s.at<double>(s.OUT), 8192);
}

inline void falp(const int64_t* __restrict in, double* __restrict out, uint8_t bw, const int64_t* __restrict base, uint8_t factor,
                 uint8_t exponent) {
	falp(reinterpret_cast<const uint64_t*>(in), out, bw, reinterpret_cast<const uint64_t*>(base), factor, exponent);
}

inline void falp(const uint32_t* __restrict in, float* __restrict out, uint8_t bw, const uint32_t* __restrict a_base_p, uint8_t factor,
                 uint8_t exponent) {
	if (bw > 32) { return; }
	auto&         s       = alp::gpu::tls();
	const uint8_t meta[3] = {bw, factor, exponent};
	alp::gpu::h2d(s.bw(), meta, 3);
	alp::gpu::h2d(s.at<int32_t>(s.META + 8), a_base_p, 4);
	if (bw) { alp::gpu::h2d(s.at<uint32_t>(s.PACKED), in, static_cast<size_t>(bw) * 128); }
	alp::gpu::check(alpgpu_falp_f32(alp::gpu::context(), s.at<int32_t>(s.PACKED), 1024, s.at<float>(s.OUT), s.bw(), s.at<int32_t>(s.META + 8),
	                                s.fac(), s.exp(), 1),
	                "alpgpu_falp_f32");
	alp::gpu::d2h(out, s.at<float>(s.OUT), 4096);
}

inline void falp(const int32_t* __restrict in, float* __restrict out, uint8_t bw, const int32_t* __restrict base, uint8_t factor, uint8_t exponent) {
	falp(reinterpret_cast<const uint32_t*>(in), out, bw, reinterpret_cast<const uint32_t*>(base), factor, exponent);
}

}}}} // namespace generated::falp::fallback::scalar
#endif
