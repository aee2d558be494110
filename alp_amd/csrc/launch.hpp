// launch.hpp — host-side launch functions implemented by the *.hip kernel files (internal to libalpgpu.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/alpgpu.h"

namespace alpgpu {

// what alpgpu_encode_* memsets the rowgroup states to in front of a search that PUBLISHES them beside the encode: every byte 0xFF = "not there"
// (alp_device.hpp: rowgroup_state_is_whole — no word of a real state is all-ones, so a torn read of a state being published is recognised)
constexpr int kStateUnpublished = 0xFF;

// decode_kernels.hip
// patch_max: ALP vectors with 1..patch_max (<= 64) exceptions are decoded without any per-value lookup and patched after their stores (0: never)
// progress (nullable): a word of device memory the kernel's workgroups report their position to, tagged (read_ahead_kernels.hip)
// gate (unhinted decode, api_decode.hip): != 0 -> the launch runs only if the context's shape word (progress[kCtxWordShape], decode_policy.hpp) holds this value
int launch_decode_column(hipStream_t stream, const alpgpu_column* col, double* d_out, int variant, int n_cus, uint32_t patch_max, uint64_t* progress = nullptr,
                         uint64_t progress_tag = 0, uint32_t gate = 0);
bool decode_patch_arm_compiled();
// read_ahead_kernels.hip: the column's descriptors, packed words and exception records read into the Infinity Cache a bounded distance ahead of the decode
// kernel that reports to d_progress with this tag (lead_min / lead_max in vectors; value_bytes 8 or 4; grid workgroups of four wavefronts)
// ps_per_vector: picoseconds the decode needs per vector AT LEAST (the read-ahead's workgroups sleep by it between looks at the progress word);
// ps_per_tick: of wall_clock64() on this device (hipDeviceAttributeWallClockRate)
// d_ctx_words: the context's 2 KiB of device words (decode_policy.hpp); from_plan: lead, pace and max_bits are read from its plan words (an unhinted decode)
int launch_read_ahead(hipStream_t stream, const alpgpu_column* col, int value_bytes, uint64_t* d_ctx_words, uint64_t tag, uint32_t lead_min, uint32_t lead_max,
                      uint32_t ps_per_vector, uint32_t ps_per_tick, uint32_t max_bits, int grid, bool from_plan = false); // max_bits: records of wider vectors are left alone (their descriptors are read)
int launch_decode_sum(hipStream_t stream, const alpgpu_column* col, double* d_sums, int vectors_per_wg);
int launch_decode_count_range(hipStream_t stream, const alpgpu_column* col, double lo, double hi, uint32_t* d_counts);
// the same sinks, one wavefront per vector, packed words straight from HBM (no stage, no barrier); count = false: per-vector sums (double), true: counts (u32)
int launch_sink_direct(hipStream_t stream, const alpgpu_column* col, double lo, double hi, void* d_out, bool count);
int launch_decode_sum_f32(hipStream_t stream, const alpgpu_column* col, double* d_sums);
int launch_decode_count_range_f32(hipStream_t stream, const alpgpu_column* col, float lo, float hi, uint32_t* d_counts);
int launch_sink_direct_f32(hipStream_t stream, const alpgpu_column* col, float lo, float hi, void* d_out, bool count);

// consume_kernels.hip: decode fused into SUM / COUNT consumers (persistent, software-pipelined), and the column total's tree
int launch_consume_sum(hipStream_t stream, const alpgpu_column* col, double* d_sums, int n_cus);
int launch_consume_count_range(hipStream_t stream, const alpgpu_column* col, double lo, double hi, uint32_t* d_counts, int n_cus);
int launch_tree_sum(hipStream_t stream, const double* d_in, uint64_t n, double* d_scratch, double* d_total);

// guard_kernels.hip
int launch_validate_column(hipStream_t stream, const alpgpu_column* col, uint32_t value_bytes, unsigned long long* d_first_bad);
int launch_count_rd_rowgroups(hipStream_t stream, const alpgpu_column* col, uint64_t* d_count);
// d_out[3 s .. 3 s + 2] = {packed bytes, exceptions, ALP_RD vectors} of segment s (seg_vectors consecutive vectors each)
int launch_segment_sums(hipStream_t stream, const alpgpu_column* col, uint64_t seg_vectors, uint32_t n_seg, uint64_t* d_out);
// the unhinted decode's plan (decode_policy.hpp: policy_unhinted) from the segment sums just taken, into the context's plan words; one small workgroup
int launch_unhinted_plan(hipStream_t stream, uint64_t* d_ctx_words, uint32_t n_seg, uint64_t n_vectors, int value_bytes, int read_ahead_option, int lead_us_option, uint32_t max_bits);

// init_kernels.hip
int launch_rowgroup_init(hipStream_t stream, const double* d_in, uint64_t n_vectors, alpgpu_rowgroup_state* d_rgs, uint16_t* d_rd_order, uint64_t rg_first = 0,
                         uint64_t rg_count = 0);

int launch_state_from_samples(hipStream_t stream, const double* d_samples, uint32_t n_samples, alpgpu_rowgroup_state* d_state, int force_rd, double* d_cut_estimate = nullptr);
// the persistent, publishing form (runs beside the single-pass encode on a second stream) and the tag clean-up behind it
// tile_shaped: eight-wavefront workgroups that fit a lean encode tile's slot (init_kernels.hip), else four wavefronts beside two classic tiles
int launch_rowgroup_init_async(hipStream_t stream, const double* d_in, uint64_t n_vectors, alpgpu_rowgroup_state* d_rgs, uint16_t* d_rd_order, uint64_t rg_first,
                               uint64_t rg_count, int grid, bool tile_shaped = false, uint32_t adaptive_base = 0);
// adaptive_base != 0: `grid` workgroups are launched, those beyond adaptive_base leave at once unless the column's head is mostly ALP_RD (init_kernels.hip)
int launch_rowgroup_init_async_f32(hipStream_t stream, const float* d_in, uint64_t n_vectors, alpgpu_rowgroup_state* d_rgs, uint16_t* d_rd_order, uint64_t rg_first,
                                   uint64_t rg_count, int grid);

// encode_kernels.hip
uint64_t encode_workspace_bytes(uint64_t n_vectors);
// or-ed into `kernel` (lean kernel only): tiles reserve their bytes with one atomic add instead of waiting for their predecessors' sizes
// (ALPGPU_OPT_ENCODE_UNORDERED; encode_lean_kernels.hip)
constexpr int kEncodeUnorderedFlag = 0x100;
// single pass (force_stall: debug, every look-back that has to wait gives up — exercises the recovery route)
// kernel: ALPGPU_ENCODE_KERNEL_LEAN (encode_lean_kernels.hip: 6 KiB of LDS and <= 72 VGPRs per wavefront, three tiles per CU) or _CLASSIC (k_encode_fused)
int launch_encode_fused(hipStream_t stream, const double* d_in, uint64_t n_vectors, const alpgpu_column* col, uint64_t* d_workspace, bool force_stall = false,
                        bool async_states = false, hipEvent_t async_join = nullptr, hipEvent_t async_head = nullptr, int kernel = ALPGPU_ENCODE_KERNEL_LEAN);
// pieces of the above for a caller that interleaves other work: zero d_totals, then vector ranges in ascending order
int launch_encode_reset_totals(hipStream_t stream, const alpgpu_column* col);
int launch_encode_fused_range(hipStream_t stream, const double* d_in, const alpgpu_column* col, uint64_t* d_workspace, uint64_t v_first, uint64_t n_range,
                              bool force_stall = false, bool async_states = false, hipEvent_t async_join = nullptr, hipEvent_t async_head = nullptr,
                              int kernel = ALPGPU_ENCODE_KERNEL_LEAN);
// two pass; gate = nullptr: unconditionally, else a device word that must be non-zero for the kernels to do anything (d_totals + 6)
int launch_encode_vectors(hipStream_t stream, const double* d_in, uint64_t n_vectors, const alpgpu_column* col, uint64_t* d_workspace,
                          int n_cus, const uint64_t* gate = nullptr);
int launch_scan_offsets(hipStream_t stream, const alpgpu_column* col, uint64_t n_vectors, uint64_t* d_workspace, bool f32, const uint64_t* gate);

// pad_kernels.hip
int launch_pad_tail(hipStream_t stream, double* d_in, uint64_t n_values);

// primitive_kernels.hip
int launch_ffor_i64(hipStream_t stream, int n_cus, const int64_t* in, int64_t* packed, size_t stride, const uint8_t* bw,
                    const int64_t* base, uint64_t n);
int launch_unffor_i64(hipStream_t stream, int n_cus, const int64_t* packed, size_t stride, int64_t* out, const uint8_t* bw,
                      const int64_t* base, uint64_t n);
int launch_falp(hipStream_t stream, int n_cus, const int64_t* packed, size_t stride, double* out, const uint8_t* bw,
                const int64_t* base, const uint8_t* fac, const uint8_t* exp, uint64_t n);
int launch_ffor_u16(hipStream_t stream, int n_cus, const uint16_t* in, uint16_t* packed, size_t stride, const uint8_t* bw,
                    const uint16_t* base, uint64_t n);
int launch_unffor_u16(hipStream_t stream, int n_cus, const uint16_t* packed, size_t stride, uint16_t* out, const uint8_t* bw,
                      const uint16_t* base, uint64_t n);
int launch_ffor_u8(hipStream_t stream, int n_cus, const uint8_t* in, uint8_t* packed, size_t stride, const uint8_t* bw, const uint8_t* base, uint64_t n);
int launch_unffor_u8(hipStream_t stream, int n_cus, const uint8_t* packed, size_t stride, uint8_t* out, const uint8_t* bw, const uint8_t* base, uint64_t n);
int launch_decode_values(hipStream_t stream, int n_cus, const int64_t* enc, double* out, const uint8_t* fac, const uint8_t* exp,
                         uint64_t n);
int launch_patch(hipStream_t stream, int n_cus, double* out, const double* exc, const uint16_t* pos, size_t stride,
                 const uint16_t* cnt, uint64_t n);
int launch_analyze_ffor(hipStream_t stream, int n_cus, const int64_t* enc, uint8_t* bw, int64_t* base, uint64_t n);
int launch_encode_simdized(hipStream_t stream, int n_cus, const double* in, double* exc, uint16_t* pos, size_t stride, uint16_t* cnt,
                           int64_t* enc, const uint8_t* fac, const uint8_t* exp, uint64_t n);
int launch_encode_value(hipStream_t stream, const double* in, int64_t* enc, int fac, int exp, int safe, uint64_t n);
int launch_encode_value_f32(hipStream_t stream, const float* in, int32_t* enc, int fac, int exp, uint64_t n);
int launch_decode_probe(hipStream_t stream, const alpgpu_column* col, double* d_sums);
int launch_traffic_probe(hipStream_t stream, const void* d_in, void* d_out, uint64_t n_vectors, uint32_t write_bytes);
int launch_encode_values(hipStream_t stream, int n_cus, const double* in, const alpgpu_rowgroup_state* states, const uint32_t* idx,
                         double* exc, uint16_t* pos, size_t stride, uint16_t* cnt, int64_t* enc, uint8_t* fac, uint8_t* exp, uint64_t n);
int launch_rd_encode(hipStream_t stream, int n_cus, const double* in, const alpgpu_rowgroup_state* states, const uint32_t* idx,
                     uint16_t* exc, uint16_t* pos, size_t stride, uint16_t* cnt, uint64_t* right, uint16_t* left, uint64_t n);
int launch_rd_decode(hipStream_t stream, int n_cus, double* out, const uint64_t* right, const uint16_t* left,
                     const alpgpu_rowgroup_state* states, const uint32_t* idx, const uint16_t* exc, const uint16_t* pos, size_t stride,
                     const uint16_t* cnt, uint64_t n);


// ---- single precision (decode_f32_kernels.hip, encode_f32_kernels.hip, init_kernels.hip, primitive_f32_kernels.hip) ----
// pad_kib: unused dynamic LDS per workgroup (residency cap); progress / tag / gate: as launch_decode_column
int launch_decode_column_f32(hipStream_t stream, const alpgpu_column* col, float* d_out, int vectors_per_wg, bool plain_stores, int pad_kib = -1, uint64_t* progress = nullptr,
                             uint64_t progress_tag = 0, uint32_t gate = 0);
// decode_stream_f32_kernels.hip: persistent workgroups, three chunks in flight each; shape 16: chunks of 8 vectors / 8 KiB of records, 17: 16 / 16 KiB, 18: 4 / 12 KiB
int launch_decode_stream_f32(hipStream_t stream, const alpgpu_column* col, float* d_out, int shape, int n_cus, uint64_t* progress = nullptr, uint64_t progress_tag = 0);
int launch_rowgroup_init_f32(hipStream_t stream, const float* d_in, uint64_t n_vectors, alpgpu_rowgroup_state* d_rgs, uint16_t* d_rd_order,
                             uint64_t rg_first = 0, uint64_t rg_count = 0);
int launch_state_from_samples_f32(hipStream_t stream, const float* d_samples, uint32_t n_samples, alpgpu_rowgroup_state* d_state, int force_rd, double* d_cut_estimate = nullptr);
int launch_encode_fused_f32(hipStream_t stream, const float* d_in, uint64_t n_vectors, const alpgpu_column* col, uint64_t* d_workspace, bool force_stall = false,
                            bool async_states = false, hipEvent_t async_join = nullptr, hipEvent_t async_head = nullptr, bool unordered = false);
int launch_encode_fused_range_f32(hipStream_t stream, const float* d_in, const alpgpu_column* col, uint64_t* d_workspace, uint64_t v_first, uint64_t n_range,
                                  bool force_stall = false, bool async_states = false, hipEvent_t async_join = nullptr, hipEvent_t async_head = nullptr,
                                  bool unordered = false);
int launch_encode_vectors_f32(hipStream_t stream, const float* d_in, uint64_t n_vectors, const alpgpu_column* col, uint64_t* d_workspace, const uint64_t* gate = nullptr);
int launch_pad_tail_f32(hipStream_t stream, float* d_in, uint64_t n_values);
int launch_ffor_i32(hipStream_t stream, int n_cus, const int32_t* in, int32_t* packed, size_t stride, const uint8_t* bw, const int32_t* base, uint64_t n);
int launch_unffor_i32(hipStream_t stream, int n_cus, const int32_t* packed, size_t stride, int32_t* out, const uint8_t* bw, const int32_t* base, uint64_t n);
int launch_falp_f32(hipStream_t stream, int n_cus, const int32_t* packed, size_t stride, float* out, const uint8_t* bw, const int32_t* base,
                    const uint8_t* fac, const uint8_t* exp, uint64_t n);
int launch_decode_values_f32(hipStream_t stream, int n_cus, const int32_t* enc, float* out, const uint8_t* fac, const uint8_t* exp, uint64_t n);
int launch_patch_f32(hipStream_t stream, int n_cus, float* out, const float* exc, const uint16_t* pos, size_t stride, const uint16_t* cnt, uint64_t n);
int launch_analyze_ffor_i32(hipStream_t stream, int n_cus, const int32_t* enc, uint8_t* bw, int32_t* base, uint64_t n);
int launch_encode_simdized_f32(hipStream_t stream, int n_cus, const float* in, float* exc, uint16_t* pos, size_t stride, uint16_t* cnt, int32_t* enc,
                               const uint8_t* fac, const uint8_t* exp, uint64_t n);
int launch_encode_values_f32(hipStream_t stream, int n_cus, const float* in, const alpgpu_rowgroup_state* states, const uint32_t* idx, float* exc,
                             uint16_t* pos, size_t stride, uint16_t* cnt, int32_t* enc, uint8_t* fac, uint8_t* exp, uint64_t n);
int launch_rd_encode_f32(hipStream_t stream, int n_cus, const float* in, const alpgpu_rowgroup_state* states, const uint32_t* idx, uint16_t* exc,
                         uint16_t* pos, size_t stride, uint16_t* cnt, uint32_t* right, uint16_t* left, uint64_t n);
int launch_rd_decode_f32(hipStream_t stream, int n_cus, float* out, const uint32_t* right, const uint16_t* left, const alpgpu_rowgroup_state* states,
                         const uint32_t* idx, const uint16_t* exc, const uint16_t* pos, size_t stride, const uint16_t* cnt, uint64_t n);

} // namespace alpgpu
