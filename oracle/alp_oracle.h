/* oracle/alp_oracle.h — TEST INFRASTRUCTURE ONLY (see alp_oracle.c header). */
#ifndef ALP_ORACLE_H
#define ALP_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ALPO_SCHEME_INVALID = 0, ALPO_SCHEME_ALP_RD = 1, ALPO_SCHEME_ALP = 2 };

/* Per-rowgroup state — plain-C mirror of alp::state<double> (reference include/alp/encoder.hpp:35-62). */
typedef struct {
	int      scheme;
	size_t   sampled_values_n;
	int      k_combinations;
	int      combos[5][2]; /* [i][0] = exponent, [i][1] = factor */
	uint8_t  exp, fac;
	/* ALP_RD */
	uint8_t  right_bit_width, left_bit_width;
	uint16_t left_parts_dict[8];
	uint8_t  actual_dictionary_size;
	/* full "sorted by repetitions" list of the persisted dictionary build (rd.hpp:47-77): entry i is the
	 * left part at sorted position i; entries > actual_dictionary_size are in the reference's
	 * left_parts_dict_map with value i; entry == actual_dictionary_size is (quirk, rd.hpp:75) NOT. */
	uint16_t rd_sorted_left[1024];
	uint16_t rd_sorted_count;
} alpo_state;

int64_t alpo_cast64(double x);
int64_t alpo_encode_value_safe(double v, int fac, int exp);
int64_t alpo_encode_value_unsafe(double v, int fac, int exp);
double  alpo_decode_value(int64_t enc, int fac, int exp);
int     alpo_count_bits(int64_t max, int64_t min);

size_t alpo_first_level_sample(const double* data, size_t data_offset, size_t data_size, double* data_sample);
void   alpo_find_top_k(const double* smp, alpo_state* st);
void   alpo_encoder_init(const double* col, size_t off, size_t n, double* sample_arr, alpo_state* st);
void   alpo_find_best_ef(const alpo_state* st, const double* in, int vector_size, uint8_t* fac, uint8_t* exp);
void   alpo_encode_simdized(const double* in, double* exc, uint16_t* pos, uint16_t* cnt, int64_t* enc, int fac, int exp);
void   alpo_encode(const double* in, double* exc, uint16_t* pos, uint16_t* cnt, int64_t* enc, alpo_state* st);
void   alpo_analyze_ffor(const int64_t* in, uint8_t* bw, int64_t* base);

void alpo_ffor_u64(const uint64_t* in, uint64_t* out, int bw, uint64_t base);
void alpo_unffor_u64(const uint64_t* in, uint64_t* out, int bw, uint64_t base);
void alpo_ffor_u16(const uint16_t* in, uint16_t* out, int bw, uint16_t base);
void alpo_unffor_u16(const uint16_t* in, uint16_t* out, int bw, uint16_t base);

/* 8-bit lanes (not used by the codec; part of the reference's ffor/unffor API, include/fastlanes/ffor.hpp:10) */
void alpo_ffor_u8(const uint8_t* in, uint8_t* out, int bw, uint8_t base);
void alpo_unffor_u8(const uint8_t* in, uint8_t* out, int bw, uint8_t base);

void alpo_decode(const int64_t* enc, int fac, int exp, double* out);
void alpo_falp(const uint64_t* in, double* out, int bw, uint64_t base, int fac, int exp);
void alpo_patch(double* out, const double* exc, const uint16_t* pos, uint16_t cnt);

void alpo_rd_init(const double* col, size_t off, size_t n, double* sample_arr, alpo_state* st);
void alpo_rd_encode(const double* in, uint16_t* exc, uint16_t* pos, uint16_t* cnt, uint64_t* right, uint16_t* left,
                    const alpo_state* st);
void alpo_rd_decode(double* out, const uint64_t* right, const uint16_t* left, const uint16_t* exc, const uint16_t* pos,
                    uint16_t cnt, const alpo_state* st);

/* whole-column driver, same output contract as oracle/ref_harness.cpp:ref_encode_column */
void alpo_encode_column(const double* column, size_t n_vectors, uint8_t* scheme, uint8_t* e, uint8_t* f, uint8_t* bw,
                        uint8_t* lbw, int64_t* base, uint16_t* exc_cnt, int64_t* packed, uint16_t* packed_left,
                        double* exc, uint16_t* pos, uint16_t* dict, uint8_t* dict_size, uint8_t* k_out,
                        int* combos_out);

/* whole-column decode from the fixed-stride layout above */
void alpo_decode_column(size_t n_vectors, const uint8_t* scheme, const uint8_t* e, const uint8_t* f, const uint8_t* bw,
                        const uint8_t* lbw, const int64_t* base, const uint16_t* exc_cnt, const int64_t* packed,
                        const uint16_t* packed_left, const double* exc, const uint16_t* pos, const uint16_t* dict,
                        double* out);

double alpo_time_falp_column(const int64_t* packed, size_t stride_words, const uint8_t* bw, const uint8_t* e,
                             const uint8_t* f, const int64_t* base, const uint16_t* exc_cnt, const double* exc,
                             const uint16_t* pos, size_t exc_stride, size_t n_vectors, double* out, int reps);

/* ---- single precision (alp_oracle_f32.c): alp::encoder<float> / decoder<float> / rd_encoder<float>, 32-bit FFOR -- */
int32_t alpof_cast32(float x);
int32_t alpof_encode_value(float v, int fac, int exp);
float   alpof_decode_value(int32_t enc, int fac, int exp);
int     alpof_count_bits(int32_t max, int32_t min);
size_t  alpof_first_level_sample(const float* data, size_t data_offset, size_t data_size, float* data_sample);
void    alpof_find_top_k(const float* smp, alpo_state* st);
void    alpof_encoder_init(const float* col, size_t off, size_t n, float* sample_arr, alpo_state* st);
void    alpof_find_best_ef(const alpo_state* st, const float* in, int vector_size, uint8_t* fac, uint8_t* exp);
void    alpof_encode_simdized(const float* in, float* exc, uint16_t* pos, uint16_t* cnt, int32_t* enc, int fac, int exp);
void    alpof_encode(const float* in, float* exc, uint16_t* pos, uint16_t* cnt, int32_t* enc, alpo_state* st);
void    alpof_analyze_ffor(const int32_t* in, uint8_t* bw, int32_t* base);
void    alpof_ffor_u32(const uint32_t* in, uint32_t* out, int bw, uint32_t base);
void    alpof_unffor_u32(const uint32_t* in, uint32_t* out, int bw, uint32_t base);
void    alpof_decode(const int32_t* enc, int fac, int exp, float* out);
void    alpof_falp(const uint32_t* in, float* out, int bw, uint32_t base, int fac, int exp);
void    alpof_patch(float* out, const float* exc, const uint16_t* pos, uint16_t cnt);
void    alpof_rd_init(const float* col, size_t off, size_t n, float* sample_arr, alpo_state* st);
void    alpof_rd_encode(const float* in, uint16_t* exc, uint16_t* pos, uint16_t* cnt, uint32_t* right, uint16_t* left,
                        const alpo_state* st);
void    alpof_rd_decode(float* out, const uint32_t* right, const uint16_t* left, const uint16_t* exc, const uint16_t* pos,
                        uint16_t cnt, const alpo_state* st);
/* fixed-stride column layout with 32-bit words: packed [n*1024] int32, packed_left [n*1024] u16, exc [n*1024] float */
void alpof_encode_column(const float* column, size_t n_vectors, uint8_t* scheme, uint8_t* e, uint8_t* f, uint8_t* bw,
                         uint8_t* lbw, int64_t* base, uint16_t* exc_cnt, int32_t* packed, uint16_t* packed_left, float* exc,
                         uint16_t* pos, uint16_t* dict, uint8_t* dict_size, uint8_t* k_out, int* combos_out);
void alpof_decode_column(size_t n_vectors, const uint8_t* scheme, const uint8_t* e, const uint8_t* f, const uint8_t* bw,
                         const uint8_t* lbw, const int64_t* base, const uint16_t* exc_cnt, const int32_t* packed,
                         const uint16_t* packed_left, const float* exc, const uint16_t* pos, const uint16_t* dict, float* out);

#ifdef __cplusplus
}
#endif
#endif
