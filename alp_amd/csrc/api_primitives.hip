// api_primitives.hip — the reference's per-vector primitives on batches (include/alpgpu.h; see host_ctx.hpp for the map).
#include "host_ctx.hpp"

extern "C" {

int alpgpu_ffor_i64(alpgpu_ctx* ctx, const int64_t* d_in, int64_t* d_packed, size_t packed_stride, const uint8_t* d_bw,
                    const int64_t* d_base, uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_packed && d_bw && d_base,
	            alpgpu::launch_ffor_i64(ctx->stream, ctx->n_cus, d_in, d_packed, packed_stride, d_bw, d_base, n_vectors));
}
int alpgpu_unffor_i64(alpgpu_ctx* ctx, const int64_t* d_packed, size_t packed_stride, int64_t* d_out, const uint8_t* d_bw,
                      const int64_t* d_base, uint64_t n_vectors) {
	ALPGPU_PRIM(d_packed && d_out && d_bw && d_base,
	            alpgpu::launch_unffor_i64(ctx->stream, ctx->n_cus, d_packed, packed_stride, d_out, d_bw, d_base, n_vectors));
}
int alpgpu_ffor_u16(alpgpu_ctx* ctx, const uint16_t* d_in, uint16_t* d_packed, size_t packed_stride, const uint8_t* d_bw,
                    const uint16_t* d_base, uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_packed && d_bw,
	            alpgpu::launch_ffor_u16(ctx->stream, ctx->n_cus, d_in, d_packed, packed_stride, d_bw, d_base, n_vectors));
}
int alpgpu_unffor_u16(alpgpu_ctx* ctx, const uint16_t* d_packed, size_t packed_stride, uint16_t* d_out, const uint8_t* d_bw,
                      const uint16_t* d_base, uint64_t n_vectors) {
	ALPGPU_PRIM(d_packed && d_out && d_bw,
	            alpgpu::launch_unffor_u16(ctx->stream, ctx->n_cus, d_packed, packed_stride, d_out, d_bw, d_base, n_vectors));
}
int alpgpu_ffor_u8(alpgpu_ctx* ctx, const uint8_t* d_in, uint8_t* d_packed, size_t packed_stride, const uint8_t* d_bw, const uint8_t* d_base,
                   uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_packed && d_bw, alpgpu::launch_ffor_u8(ctx->stream, ctx->n_cus, d_in, d_packed, packed_stride, d_bw, d_base, n_vectors));
}
int alpgpu_unffor_u8(alpgpu_ctx* ctx, const uint8_t* d_packed, size_t packed_stride, uint8_t* d_out, const uint8_t* d_bw, const uint8_t* d_base,
                     uint64_t n_vectors) {
	ALPGPU_PRIM(d_packed && d_out && d_bw, alpgpu::launch_unffor_u8(ctx->stream, ctx->n_cus, d_packed, packed_stride, d_out, d_bw, d_base, n_vectors));
}
int alpgpu_falp_f64(alpgpu_ctx* ctx, const int64_t* d_packed, size_t packed_stride, double* d_out, const uint8_t* d_bw,
                    const int64_t* d_base, const uint8_t* d_fac, const uint8_t* d_exp, uint64_t n_vectors) {
	ALPGPU_PRIM(d_packed && d_out && d_bw && d_base && d_fac && d_exp,
	            alpgpu::launch_falp(ctx->stream, ctx->n_cus, d_packed, packed_stride, d_out, d_bw, d_base, d_fac, d_exp, n_vectors));
}
int alpgpu_decode_values_f64(alpgpu_ctx* ctx, const int64_t* d_enc, double* d_out, const uint8_t* d_fac, const uint8_t* d_exp,
                             uint64_t n_vectors) {
	ALPGPU_PRIM(d_enc && d_out && d_fac && d_exp, alpgpu::launch_decode_values(ctx->stream, ctx->n_cus, d_enc, d_out, d_fac, d_exp, n_vectors));
}
int alpgpu_patch_f64(alpgpu_ctx* ctx, double* d_out, const double* d_exc, const uint16_t* d_pos, size_t exc_stride,
                     const uint16_t* d_cnt, uint64_t n_vectors) {
	ALPGPU_PRIM(d_out && d_exc && d_pos && d_cnt, alpgpu::launch_patch(ctx->stream, ctx->n_cus, d_out, d_exc, d_pos, exc_stride, d_cnt, n_vectors));
}
int alpgpu_encode_simdized_f64(alpgpu_ctx* ctx, const double* d_in, double* d_exc, uint16_t* d_pos, size_t exc_stride,
                               uint16_t* d_cnt, int64_t* d_enc, const uint8_t* d_fac, const uint8_t* d_exp, uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_exc && d_pos && d_cnt && d_enc && d_fac && d_exp,
	            alpgpu::launch_encode_simdized(ctx->stream, ctx->n_cus, d_in, d_exc, d_pos, exc_stride, d_cnt, d_enc, d_fac, d_exp, n_vectors));
}
int alpgpu_encode_values_f64(alpgpu_ctx* ctx, const double* d_in, const alpgpu_rowgroup_state* d_states, const uint32_t* d_state_idx,
                             double* d_exc, uint16_t* d_pos, size_t exc_stride, uint16_t* d_cnt, int64_t* d_enc, uint8_t* d_fac,
                             uint8_t* d_exp, uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_states && d_exc && d_pos && d_cnt && d_enc && d_fac && d_exp,
	            alpgpu::launch_encode_values(ctx->stream, ctx->n_cus, d_in, d_states, d_state_idx, d_exc, d_pos, exc_stride, d_cnt, d_enc,
	                                         d_fac, d_exp, n_vectors));
}
int alpgpu_encode_value_f64(alpgpu_ctx* ctx, const double* d_in, int64_t* d_enc, uint8_t fac, uint8_t exp, int safe, uint64_t n_values) {
	ALPGPU_CHECK_CTX(ctx);
	if (n_values && (!d_in || !d_enc)) { return fail(ALPGPU_ERR_INVALID, "null pointer argument"); }
	if (exp > 18 || fac > exp) { return fail(ALPGPU_ERR_INVALID, "factor / exponent out of range (0 <= factor <= exponent <= 18)"); }
	if (alpgpu::launch_encode_value(ctx->stream, d_in, d_enc, fac, exp, safe, n_values) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "kernel launch failed", hipGetLastError()); }
	return ALPGPU_OK;
}
int alpgpu_encode_value_f32(alpgpu_ctx* ctx, const float* d_in, int32_t* d_enc, uint8_t fac, uint8_t exp, int safe, uint64_t n_values) {
	(void)safe; // the reference's float SAFE branch does not exist as built (include/alpgpu.h, single precision)
	ALPGPU_CHECK_CTX(ctx);
	if (n_values && (!d_in || !d_enc)) { return fail(ALPGPU_ERR_INVALID, "null pointer argument"); }
	if (exp > 10 || fac > exp) { return fail(ALPGPU_ERR_INVALID, "factor / exponent out of range (0 <= factor <= exponent <= 10)"); }
	if (alpgpu::launch_encode_value_f32(ctx->stream, d_in, d_enc, fac, exp, n_values) != ALPGPU_OK) { return fail(ALPGPU_ERR_HIP, "kernel launch failed", hipGetLastError()); }
	return ALPGPU_OK;
}
int alpgpu_analyze_ffor_i64(alpgpu_ctx* ctx, const int64_t* d_enc, uint8_t* d_bw, int64_t* d_base, uint64_t n_vectors) {
	ALPGPU_PRIM(d_enc && d_bw && d_base, alpgpu::launch_analyze_ffor(ctx->stream, ctx->n_cus, d_enc, d_bw, d_base, n_vectors));
}
int alpgpu_rd_encode_vectors_f64(alpgpu_ctx* ctx, const double* d_in, const alpgpu_rowgroup_state* d_states,
                                 const uint32_t* d_state_idx, uint16_t* d_exc, uint16_t* d_pos, size_t exc_stride, uint16_t* d_cnt,
                                 uint64_t* d_right, uint16_t* d_left, uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_states && d_exc && d_pos && d_cnt && d_right && d_left,
	            alpgpu::launch_rd_encode(ctx->stream, ctx->n_cus, d_in, d_states, d_state_idx, d_exc, d_pos, exc_stride, d_cnt, d_right,
	                                     d_left, n_vectors));
}
int alpgpu_rd_decode_vectors_f64(alpgpu_ctx* ctx, double* d_out, const uint64_t* d_right, const uint16_t* d_left,
                                 const alpgpu_rowgroup_state* d_states, const uint32_t* d_state_idx, const uint16_t* d_exc,
                                 const uint16_t* d_pos, size_t exc_stride, const uint16_t* d_cnt, uint64_t n_vectors) {
	ALPGPU_PRIM(d_out && d_right && d_left && d_states && d_exc && d_pos && d_cnt,
	            alpgpu::launch_rd_decode(ctx->stream, ctx->n_cus, d_out, d_right, d_left, d_states, d_state_idx, d_exc, d_pos, exc_stride,
	                                     d_cnt, n_vectors));
}

int alpgpu_rd_encode_f64(alpgpu_ctx* ctx, const double* d_in, const alpgpu_rowgroup_state* d_states, const uint32_t* d_state_idx, uint16_t* d_exc,
                         uint16_t* d_pos, size_t exc_stride, uint16_t* d_cnt, uint64_t* d_right, uint16_t* d_left, uint64_t n_vectors) {
	return alpgpu_rd_encode_vectors_f64(ctx, d_in, d_states, d_state_idx, d_exc, d_pos, exc_stride, d_cnt, d_right, d_left, n_vectors);
}
int alpgpu_rd_decode_f64(alpgpu_ctx* ctx, double* d_out, const uint64_t* d_right, const uint16_t* d_left, const alpgpu_rowgroup_state* d_states,
                         const uint32_t* d_state_idx, const uint16_t* d_exc, const uint16_t* d_pos, size_t exc_stride, const uint16_t* d_cnt,
                         uint64_t n_vectors) {
	return alpgpu_rd_decode_vectors_f64(ctx, d_out, d_right, d_left, d_states, d_state_idx, d_exc, d_pos, exc_stride, d_cnt, n_vectors);
}
int alpgpu_ffor_i32(alpgpu_ctx* ctx, const int32_t* d_in, int32_t* d_packed, size_t packed_stride, const uint8_t* d_bw, const int32_t* d_base,
                    uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_packed && d_bw && d_base, alpgpu::launch_ffor_i32(ctx->stream, ctx->n_cus, d_in, d_packed, packed_stride, d_bw, d_base, n_vectors));
}
int alpgpu_unffor_i32(alpgpu_ctx* ctx, const int32_t* d_packed, size_t packed_stride, int32_t* d_out, const uint8_t* d_bw, const int32_t* d_base,
                      uint64_t n_vectors) {
	ALPGPU_PRIM(d_packed && d_out && d_bw && d_base, alpgpu::launch_unffor_i32(ctx->stream, ctx->n_cus, d_packed, packed_stride, d_out, d_bw, d_base, n_vectors));
}
int alpgpu_falp_f32(alpgpu_ctx* ctx, const int32_t* d_packed, size_t packed_stride, float* d_out, const uint8_t* d_bw, const int32_t* d_base,
                    const uint8_t* d_fac, const uint8_t* d_exp, uint64_t n_vectors) {
	ALPGPU_PRIM(d_packed && d_out && d_bw && d_base && d_fac && d_exp,
	            alpgpu::launch_falp_f32(ctx->stream, ctx->n_cus, d_packed, packed_stride, d_out, d_bw, d_base, d_fac, d_exp, n_vectors));
}
int alpgpu_decode_values_f32(alpgpu_ctx* ctx, const int32_t* d_enc, float* d_out, const uint8_t* d_fac, const uint8_t* d_exp, uint64_t n_vectors) {
	ALPGPU_PRIM(d_enc && d_out && d_fac && d_exp, alpgpu::launch_decode_values_f32(ctx->stream, ctx->n_cus, d_enc, d_out, d_fac, d_exp, n_vectors));
}
int alpgpu_patch_f32(alpgpu_ctx* ctx, float* d_out, const float* d_exc, const uint16_t* d_pos, size_t exc_stride, const uint16_t* d_cnt,
                     uint64_t n_vectors) {
	ALPGPU_PRIM(d_out && d_exc && d_pos && d_cnt, alpgpu::launch_patch_f32(ctx->stream, ctx->n_cus, d_out, d_exc, d_pos, exc_stride, d_cnt, n_vectors));
}
int alpgpu_encode_simdized_f32(alpgpu_ctx* ctx, const float* d_in, float* d_exc, uint16_t* d_pos, size_t exc_stride, uint16_t* d_cnt, int32_t* d_enc,
                               const uint8_t* d_fac, const uint8_t* d_exp, uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_exc && d_pos && d_cnt && d_enc && d_fac && d_exp,
	            alpgpu::launch_encode_simdized_f32(ctx->stream, ctx->n_cus, d_in, d_exc, d_pos, exc_stride, d_cnt, d_enc, d_fac, d_exp, n_vectors));
}
int alpgpu_encode_values_f32(alpgpu_ctx* ctx, const float* d_in, const alpgpu_rowgroup_state* d_states, const uint32_t* d_state_idx, float* d_exc,
                             uint16_t* d_pos, size_t exc_stride, uint16_t* d_cnt, int32_t* d_enc, uint8_t* d_fac, uint8_t* d_exp, uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_states && d_exc && d_pos && d_cnt && d_enc && d_fac && d_exp,
	            alpgpu::launch_encode_values_f32(ctx->stream, ctx->n_cus, d_in, d_states, d_state_idx, d_exc, d_pos, exc_stride, d_cnt, d_enc, d_fac,
	                                             d_exp, n_vectors));
}
int alpgpu_analyze_ffor_i32(alpgpu_ctx* ctx, const int32_t* d_enc, uint8_t* d_bw, int32_t* d_base, uint64_t n_vectors) {
	ALPGPU_PRIM(d_enc && d_bw && d_base, alpgpu::launch_analyze_ffor_i32(ctx->stream, ctx->n_cus, d_enc, d_bw, d_base, n_vectors));
}
int alpgpu_rd_encode_vectors_f32(alpgpu_ctx* ctx, const float* d_in, const alpgpu_rowgroup_state* d_states, const uint32_t* d_state_idx,
                                 uint16_t* d_exc, uint16_t* d_pos, size_t exc_stride, uint16_t* d_cnt, uint32_t* d_right, uint16_t* d_left,
                                 uint64_t n_vectors) {
	ALPGPU_PRIM(d_in && d_states && d_exc && d_pos && d_cnt && d_right && d_left,
	            alpgpu::launch_rd_encode_f32(ctx->stream, ctx->n_cus, d_in, d_states, d_state_idx, d_exc, d_pos, exc_stride, d_cnt, d_right, d_left,
	                                         n_vectors));
}
int alpgpu_rd_decode_vectors_f32(alpgpu_ctx* ctx, float* d_out, const uint32_t* d_right, const uint16_t* d_left, const alpgpu_rowgroup_state* d_states,
                                 const uint32_t* d_state_idx, const uint16_t* d_exc, const uint16_t* d_pos, size_t exc_stride, const uint16_t* d_cnt,
                                 uint64_t n_vectors) {
	ALPGPU_PRIM(d_out && d_right && d_left && d_states && d_exc && d_pos && d_cnt,
	            alpgpu::launch_rd_decode_f32(ctx->stream, ctx->n_cus, d_out, d_right, d_left, d_states, d_state_idx, d_exc, d_pos, exc_stride, d_cnt,
	                                         n_vectors));
}

} // extern "C"
