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

#ifdef __cplusplus
}
#endif
#endif
