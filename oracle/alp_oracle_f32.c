/*
 * oracle/alp_oracle_f32.c — TEST INFRASTRUCTURE ONLY.  NOT PRODUCT CODE.  (See alp_oracle.c for the rules.)
 *
 * Plain-C restatement of the reference's single-precision instantiation: alp::encoder<float>, alp::decoder<float>,
 * alp::rd_encoder<float>, the 32-bit FFOR kernels and the float falp.  Structure mirrors alp_oracle.c; only what
 * differs from the double path is restated here (constants, 32-bit casts, 66 (e,f) combinations, 32 x 32 FastLanes
 * layout); the libstdc++ order emulation is shared (libstdcxx_order.h).
 *
 * PINNING: tests/test_oracle_f32.py checks every function value-for-value against the real reference's float
 * instantiation compiled in place (oracle/_ref, reff_* shims in ref_harness.cpp) on the reference's own float test
 * columns (data/float/test_{0..3}.csv with the bit widths data/include/float/test.hpp:10-14 asserts, and
 * data/edge_case/avx512dq.csv with data/include/float/edge_case.hpp:10's (exceptions = 192, bw = 0)) and on random
 * inputs; tests/golden/float_vectors.npz holds fixtures generated from that build.
 *
 * AS-BUILT BEHAVIOUR pinned here (the reference has undefined behaviour at these points; its build system accepts
 * only Clang — CMakeLists.txt:50-52 — and this is what Clang -O2/-O3 produces; oracle/Makefile uses the same compiler):
 *  (U1) encoder.hpp:84-86 returns the double 9223372036854774784 converted to int32_t: undefined, so the optimiser
 *       drops the whole SAFE branch and encode_value<true> == encode_value<false>.  Consequence: in the two sampling
 *       estimators a -0.0f sample encodes to 0 and is NOT counted as an exception (0.0f == -0.0f), unlike doubles.
 *  (U2) static_cast<int32_t>(float) outside int32 range / NaN = 0x80000000 (cvttss2si / vcvttps2dq).
 *  (U3) int32 * int32 in decode_value wraps modulo 2^32 (imul / vpmulld).
 *  (U4) Constants<float>::FACT_ARR has 10 entries but (e,f) = (10,10) is searched (MAX_EXPONENT = 10): FACT_ARR[10]
 *       is an out-of-bounds read.  Its value never changes an observable result: with f = 10 every value whose
 *       encoded integer d != 0 has |v| >= 0.5 while |decoded| <= 2^31 * 1e-10 < 0.5, so it is an exception for ANY
 *       multiplier, and d == 0 decodes to 0 for any multiplier.  This file uses 10^10 mod 2^32.
 */
#define _POSIX_C_SOURCE 200809L
#include "alp_oracle.h"
#include "libstdcxx_order.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define VECTOR_SIZE 1024
#define N_VECTORS_PER_ROWGROUP 100
#define ROWGROUP_SIZE (N_VECTORS_PER_ROWGROUP * VECTOR_SIZE)
#define ROWGROUP_SAMPLES_JUMP 12
#define SAMPLES_PER_VECTOR 32
#define MAX_K_COMBINATIONS 5
#define CUTTING_LIMIT 16
#define MAX_RD_DICTIONARY_SIZE 8
#define SAMPLING_EARLY_EXIT_THRESHOLD 2
#define EXCEPTION_POSITION_SIZE 16
#define RD_EXCEPTION_SIZE 16
#define RD_EXCEPTION_POSITION_SIZE 16
/* include/alp/constants.hpp:30-64 */
#define F32_EXCEPTION_SIZE 32
#define F32_RD_SIZE_THRESHOLD_LIMIT (22 * SAMPLES_PER_VECTOR)
#define F32_MAX_EXPONENT 10

static const float F32_MAGIC = 12582912.0f; /* 2^23 + 2^22 */
static const float F32_FRAC[11] = {1.0f, 0.1f, 0.01f, 0.001f, 0.0001f, 0.00001f, 0.000001f, 0.0000001f, 0.00000001f,
                                   0.000000001f, 0.0000000001f};
static const float F32_EXP[11]  = {1.0f, 10.0f, 100.0f, 1000.0f, 10000.0f, 100000.0f, 1000000.0f, 10000000.0f,
                                   100000000.0f, 1000000000.0f, 10000000000.0f};
/* entries 0..9: constants.hpp:62; entry 10: see (U4) */
static const uint32_t F32_FACT[11] = {1u, 10u, 100u, 1000u, 10000u, 100000u, 1000000u, 10000000u, 100000000u,
                                      1000000000u, 1410065408u};

/* (U2) */
int32_t alpof_cast32(float x) {
	if (!(x > -2147483904.0f && x < 2147483648.0f)) { return INT32_MIN; } /* NaN fails both; -2147483904 = prev float below -2^31 */
	return (int32_t)x;
}

/* include/alp/encoder.hpp:81-89; SAFE and !SAFE coincide (U1).  float * float * float, then + MAGIC - MAGIC. */
int32_t alpof_encode_value(float v, int fac, int exp) {
	float t = v * F32_EXP[exp];
	t       = t * F32_FRAC[fac];
	t       = t + F32_MAGIC;
	t       = t - F32_MAGIC;
	return alpof_cast32(t);
}

/* include/alp/decoder.hpp:128-131 with (U3), (U4) */
float alpof_decode_value(int32_t enc, int fac, int exp) {
	const int32_t m = (int32_t)((uint32_t)enc * F32_FACT[fac]);
	return (float)m * F32_FRAC[exp];
}

/* include/alp/encoder.hpp:91-106 (32-bit branch) */
int alpof_count_bits(int32_t max, int32_t min) {
	const uint32_t delta = (uint32_t)max - (uint32_t)min;
	if (delta == 0) { return 0; }
	return 32 - __builtin_clz(delta);
}

/* include/alp/sampler.hpp:14-52, PT = float */
size_t alpof_first_level_sample(const float* data, size_t data_offset, size_t data_size, float* data_sample) {
	const size_t left_in_data      = data_size - data_offset;
	const size_t portion_to_sample = left_in_data < (size_t)ROWGROUP_SIZE ? left_in_data : (size_t)ROWGROUP_SIZE;
	const size_t available_vectors = (size_t)ceil((double)portion_to_sample / VECTOR_SIZE);
	size_t       sample_idx = 0, data_idx = data_offset;
	for (size_t vector_idx = 0; vector_idx < available_vectors; vector_idx++) {
		const size_t rem = data_size - data_idx;
		const size_t n   = rem < (size_t)VECTOR_SIZE ? rem : (size_t)VECTOR_SIZE;
		if ((vector_idx % ROWGROUP_SAMPLES_JUMP) != 0) {
			data_idx += n;
			continue;
		}
		int32_t inc = (int32_t)ceil((double)n / SAMPLES_PER_VECTOR);
		if (inc < 1) { inc = 1; }
		if (n < SAMPLES_PER_VECTOR && sample_idx != 0) {
			data_idx += n;
			continue;
		}
		for (size_t i = 0; i < n; i += (size_t)inc) { data_sample[sample_idx++] = data[data_idx + i]; }
		data_idx += n;
	}
	return sample_idx;
}

typedef struct {
	int e, f, count;
} combo32_t;
static int combo32_before(const combo32_t* a, const combo32_t* b) { /* encoder.hpp:128-132 */
	return (a->count > b->count) || (a->count == b->count && b->e < a->e) ||
	       (a->count == b->count && b->e == a->e && b->f < a->f);
}

/* include/alp/encoder.hpp:139-235, PT = float: 66 combinations, exception cost 32 + 16 bits, RD threshold 22 * 32 */
void alpof_find_top_k(const float* smp, alpo_state* st) {
	const uint64_t n_vectors_to_sample = (uint64_t)ceilf((float)st->sampled_values_n / SAMPLES_PER_VECTOR);
	const uint64_t samples_size = st->sampled_values_n < SAMPLES_PER_VECTOR ? st->sampled_values_n : SAMPLES_PER_VECTOR;
	combo32_t      global[16];
	int            n_global   = 0;
	uint64_t       smp_offset = 0;
	const uint64_t worst = samples_size * (F32_EXCEPTION_SIZE + EXCEPTION_POSITION_SIZE) + samples_size * F32_EXCEPTION_SIZE;
	uint64_t       best_size = worst;
	for (uint64_t smp_n = 0; smp_n < n_vectors_to_sample; smp_n++) {
		int      found_factor = 0, found_exponent = 0;
		uint64_t sample_size = worst;
		for (int e = F32_MAX_EXPONENT; e >= 0; e--) {
			for (int f = e; f >= 0; f--) {
				uint16_t exc = 0, non_exc = 0;
				int32_t  mx = INT32_MIN, mn = INT32_MAX;
				for (uint64_t i = 0; i < samples_size; i++) {
					const float   v   = smp[smp_offset + i];
					const int32_t enc = alpof_encode_value(v, f, e);
					const float   dec = alpof_decode_value(enc, f, e);
					if (dec == v) {
						non_exc++;
						if (enc > mx) { mx = enc; }
						if (enc < mn) { mn = enc; }
					} else {
						exc++;
					}
				}
				if (non_exc < 2) { continue; }
				const uint32_t bits = (uint32_t)alpof_count_bits(mx, mn);
				const uint64_t size = samples_size * bits + (uint64_t)exc * (F32_EXCEPTION_SIZE + EXCEPTION_POSITION_SIZE);
				if ((size < sample_size) || (size == sample_size && found_exponent < e) ||
				    (size == sample_size && found_exponent == e && found_factor < f)) {
					sample_size    = size;
					found_exponent = e;
					found_factor   = f;
					if (sample_size < best_size) { best_size = sample_size; }
				}
			}
		}
		int hit = -1;
		for (int i = 0; i < n_global; i++) {
			if (global[i].e == found_exponent && global[i].f == found_factor) { hit = i; }
		}
		if (hit < 0) {
			global[n_global].e = found_exponent, global[n_global].f = found_factor, global[n_global].count = 1;
			n_global++;
		} else {
			global[hit].count++;
		}
		smp_offset += samples_size;
	}
	if (best_size >= F32_RD_SIZE_THRESHOLD_LIMIT) {
		st->scheme = ALPO_SCHEME_ALP_RD;
		return;
	}
	for (int i = 1; i < n_global; i++) { /* strict total order: any sort reproduces std::sort, encoder.hpp:228 */
		combo32_t c = global[i];
		int       j = i - 1;
		while (j >= 0 && combo32_before(&c, &global[j])) {
			global[j + 1] = global[j];
			j--;
		}
		global[j + 1] = c;
	}
	if (n_global < st->k_combinations) { st->k_combinations = n_global; }
	for (int i = 0; i < st->k_combinations; i++) { st->combos[i][0] = global[i].e, st->combos[i][1] = global[i].f; }
}

/* include/alp/encoder.hpp:420-427 */
void alpof_encoder_init(const float* col, size_t off, size_t n, float* sample_arr, alpo_state* st) {
	st->scheme           = ALPO_SCHEME_ALP;
	st->sampled_values_n = alpof_first_level_sample(col, off, n, sample_arr);
	st->k_combinations   = MAX_K_COMBINATIONS;
	for (int i = 0; i < 5; i++) { st->combos[i][0] = st->combos[i][1] = -1; }
	alpof_find_top_k(sample_arr, st);
}

/* include/alp/encoder.hpp:241-305 */
void alpof_find_best_ef(const alpo_state* st, const float* in, int vector_size, uint8_t* fac, uint8_t* exp) {
	uint8_t  found_exponent = 0, found_factor = 0, worse_count = 0;
	uint64_t best_size = 0;
	int32_t  inc       = vector_size / SAMPLES_PER_VECTOR;
	if (inc < 1) { inc = 1; }
	for (int k = 0; k < st->k_combinations; k++) {
		const int e = st->combos[k][0], f = st->combos[k][1];
		uint32_t  exc = 0;
		int32_t   mx = INT32_MIN, mn = INT32_MAX;
		for (int i = 0; i < vector_size; i += inc) {
			const float   v   = in[i];
			const int32_t enc = alpof_encode_value(v, f, e);
			const float   dec = alpof_decode_value(enc, f, e);
			if (dec == v) {
				if (enc > mx) { mx = enc; }
				if (enc < mn) { mn = enc; }
			} else {
				exc++;
			}
		}
		const uint32_t bits = (uint32_t)alpof_count_bits(mx, mn); /* no encodable sample: bits(INT32_MIN, INT32_MAX) = 1 */
		const uint64_t size = (uint64_t)SAMPLES_PER_VECTOR * bits + (uint64_t)exc * (F32_EXCEPTION_SIZE + EXCEPTION_POSITION_SIZE);
		if (k == 0) {
			best_size = size, found_factor = (uint8_t)f, found_exponent = (uint8_t)e;
			continue;
		}
		if (size >= best_size) {
			if (++worse_count == SAMPLING_EARLY_EXIT_THRESHOLD) { break; }
			continue;
		}
		best_size = size, found_factor = (uint8_t)f, found_exponent = (uint8_t)e, worse_count = 0;
	}
	*exp = found_exponent;
	*fac = found_factor;
}

/* include/alp/encoder.hpp:307-400 (scalar compaction branch).  The float masks are correct (constants.hpp:41-46):
 * NaN, +-Inf and -0.0 are replaced by (float)ENCODING_UPPER_LIMIT = 9223372036854775808.0f before encoding. */
void alpof_encode_simdized(const float* in, float* exc, uint16_t* pos, uint16_t* cnt, int32_t* enc, int fac, int exp) {
	static _Thread_local float    dec_arr[VECTOR_SIZE];
	static _Thread_local float    val_arr[VECTOR_SIZE];
	static _Thread_local uint32_t idx_arr[VECTOR_SIZE + 1];
	const float upper = (float)9223372036854774784.0;
	for (int i = 0; i < VECTOR_SIZE; i++) {
		uint32_t b;
		memcpy(&b, &in[i], 4);
		const int special = ((b & 0x7FFFFFFFu) >= 0x7F800000u) || b == 0x80000000u;
		val_arr[i]        = special ? upper : in[i];
	}
	for (int i = 0; i < VECTOR_SIZE; i++) {
		enc[i]     = alpof_encode_value(val_arr[i], fac, exp);
		dec_arr[i] = alpof_decode_value(enc[i], fac, exp);
	}
	uint32_t n = 0;
	for (int i = 0; i < VECTOR_SIZE; i++) {
		const int is_exc = dec_arr[i] != val_arr[i];
		idx_arr[n]       = (uint32_t)i;
		n += (uint32_t)is_exc;
	}
	/* filler search, encoder.hpp:382-388; entries above n stand in for stale thread_local contents exactly as in
	 * alp_oracle.c:alpo_encode_simdized */
	int32_t filler = 0;
	for (uint32_t i = n + 1; i < VECTOR_SIZE; i++) { idx_arr[i] = i; }
	for (uint32_t i = 0; i < VECTOR_SIZE; i++) {
		if (i != idx_arr[i]) {
			filler = enc[i];
			break;
		}
	}
	uint16_t c = 0;
	for (uint32_t j = 0; j < n; j++) {
		const uint32_t i = idx_arr[j];
		enc[i]           = filler;
		exc[c]           = in[i];
		pos[c]           = (uint16_t)i;
		c++;
	}
	*cnt = c;
}

/* include/alp/encoder.hpp:402-418 */
void alpof_encode(const float* in, float* exc, uint16_t* pos, uint16_t* cnt, int32_t* enc, alpo_state* st) {
	if (st->k_combinations > 1) {
		alpof_find_best_ef(st, in, VECTOR_SIZE, &st->fac, &st->exp);
	} else {
		st->exp = (uint8_t)st->combos[0][0];
		st->fac = (uint8_t)st->combos[0][1];
	}
	alpof_encode_simdized(in, exc, pos, cnt, enc, st->fac, st->exp);
}

/* include/alp/encoder.hpp:109-120 */
void alpof_analyze_ffor(const int32_t* in, uint8_t* bw, int32_t* base) {
	int32_t mn = INT32_MAX, mx = INT32_MIN;
	for (int i = 0; i < VECTOR_SIZE; i++) {
		if (in[i] < mn) { mn = in[i]; }
		if (in[i] > mx) { mx = in[i]; }
	}
	*bw   = (uint8_t)alpof_count_bits(mx, mn);
	*base = mn;
}

/* ---- 32-bit FFOR, closed form of src/fastlanes_generated_ffor.cpp:1776-7378 (dispatch :29825 ff.) and
 * src/fastlanes_generated_unffor.cpp (32-bit section): 32 lane-streams x 32 rows; value i -> lane = i % 32,
 * row = i / 32; the lane's stream is the LSB-first concatenation of its 32 bw-bit fields; stream word k lives at
 * out[32*k + lane].  bw = 0 writes nothing; bw = 32 does not mask; bw > 32 is a no-op. */
void alpof_ffor_u32(const uint32_t* in, uint32_t* out, int bw, uint32_t base) {
	if (bw <= 0 || bw > 32) { return; }
	const uint32_t mask = bw == 32 ? 0xFFFFFFFFu : ((1u << bw) - 1u);
	for (int w = 0; w < 32 * bw; w++) { out[w] = 0; }
	for (int i = 0; i < VECTOR_SIZE; i++) {
		const int      lane = i % 32, row = i / 32;
		const uint32_t v = (in[i] - base) & mask;
		const int      p = row * bw, k = p / 32, s = p % 32;
		out[32 * k + lane] |= v << s;
		if (s + bw > 32) { out[32 * (k + 1) + lane] |= v >> (32 - s); }
	}
}

void alpof_unffor_u32(const uint32_t* in, uint32_t* out, int bw, uint32_t base) {
	if (bw < 0 || bw > 32) { return; }
	if (bw == 0) {
		for (int i = 0; i < VECTOR_SIZE; i++) { out[i] = base; }
		return;
	}
	const uint32_t mask = bw == 32 ? 0xFFFFFFFFu : ((1u << bw) - 1u);
	for (int i = 0; i < VECTOR_SIZE; i++) {
		const int lane = i % 32, row = i / 32;
		const int p = row * bw, k = p / 32, s = p % 32;
		uint32_t  v = in[32 * k + lane] >> s;
		if (s + bw > 32) { v |= in[32 * (k + 1) + lane] << (32 - s); }
		out[i] = (v & mask) + base;
	}
}

/* include/alp/decoder.hpp:134-138 */
void alpof_decode(const int32_t* enc, int fac, int exp, float* out) {
	for (int i = 0; i < VECTOR_SIZE; i++) { out[i] = alpof_decode_value(enc[i], fac, exp); }
}

/* include/alp/falp.hpp:28-44 -> src/falp.cpp (32-bit section, falp_<bw>bw_32ow_32crw_1uf): unffor + decode fused */
void alpof_falp(const uint32_t* in, float* out, int bw, uint32_t base, int fac, int exp) {
	static _Thread_local uint32_t tmp[VECTOR_SIZE];
	alpof_unffor_u32(in, tmp, bw, base);
	for (int i = 0; i < VECTOR_SIZE; i++) { out[i] = alpof_decode_value((int32_t)tmp[i], fac, exp); }
}

/* include/alp/decoder.hpp:141-149 */
void alpof_patch(float* out, const float* exc, const uint16_t* pos, uint16_t cnt) {
	for (uint16_t i = 0; i < cnt; i++) { out[pos[i]] = exc[i]; }
}

/* ---- ALP_RD, UT = uint32_t: include/alp/rd.hpp ---------------------------------------------------------------- */
static double rd32_estimate(int rbw, int lbw, uint32_t exceptions_count, uint64_t sample_count) { /* rd.hpp:23-31 */
	const double exceptions_size = (double)(exceptions_count * (RD_EXCEPTION_POSITION_SIZE + RD_EXCEPTION_SIZE));
	return (double)rbw + (double)lbw + (exceptions_size / (double)sample_count);
}

static double rd32_build_dictionary(const float* in_p, int right_bit_width, alpo_state* st, int persist) { /* rd.hpp:33-87 */
	static _Thread_local ht_t  h;
	static _Thread_local rep_t sorted[VECTOR_SIZE];
	ht_init(&h);
	for (size_t i = 0; i < st->sampled_values_n; i++) {
		uint32_t b;
		memcpy(&b, &in_p[i], 4);
		ht_increment(&h, (uint64_t)(b >> right_bit_width));
	}
	int n_sorted = 0;
	for (int p = h.bb_next; p != HT_NULL; p = h.nodes[p].next) {
		sorted[n_sorted].first  = h.nodes[p].val;
		sorted[n_sorted].second = h.nodes[p].key;
		n_sorted++;
	}
	ss_sort(sorted, sorted + n_sorted);
	uint32_t exceptions_count = 0;
	for (int i = MAX_RD_DICTIONARY_SIZE; i < n_sorted; i++) { exceptions_count += (uint32_t)sorted[i].first; }
	const int dict_size = n_sorted < MAX_RD_DICTIONARY_SIZE ? n_sorted : MAX_RD_DICTIONARY_SIZE;
	int       lbw       = (int)ceil(log2((double)dict_size));
	if (lbw < 1) { lbw = 1; }
	if (persist) {
		for (int i = 0; i < 8; i++) { st->left_parts_dict[i] = 0; }
		for (int i = 0; i < dict_size; i++) { st->left_parts_dict[i] = (uint16_t)sorted[i].second; }
		st->left_bit_width         = (uint8_t)lbw;
		st->right_bit_width        = (uint8_t)right_bit_width;
		st->actual_dictionary_size = (uint8_t)dict_size;
		st->rd_sorted_count        = (uint16_t)n_sorted;
		for (int i = 0; i < n_sorted; i++) { st->rd_sorted_left[i] = (uint16_t)sorted[i].second; }
	}
	return rd32_estimate(right_bit_width, lbw, exceptions_count, st->sampled_values_n);
}

/* rd.hpp:89-104, :180-185: cuts 32-1 .. 32-16 */
void alpof_rd_init(const float* col, size_t off, size_t n, float* sample_arr, alpo_state* st) {
	st->scheme           = ALPO_SCHEME_ALP_RD;
	st->sampled_values_n = alpof_first_level_sample(col, off, n, sample_arr);
	int    right_bit_width = 0;
	double best            = 1.7976931348623157e308;
	for (int i = 1; i <= CUTTING_LIMIT; i++) {
		const int    cand = 32 - i;
		const double est  = rd32_build_dictionary(sample_arr, cand, st, 0);
		if (est < best) {
			right_bit_width = cand;
			best            = est;
		}
	}
	rd32_build_dictionary(sample_arr, right_bit_width, st, 1);
}

/* rd.hpp:109-147 */
void alpof_rd_encode(const float* in, uint16_t* exc, uint16_t* pos, uint16_t* cnt, uint32_t* right, uint16_t* left,
                     const alpo_state* st) {
	const int      rbw  = st->right_bit_width;
	const uint32_t mask = (uint32_t)((1ULL << rbw) - 1);
	uint16_t       c    = 0;
	for (int i = 0; i < VECTOR_SIZE; i++) {
		uint32_t b;
		memcpy(&b, &in[i], 4);
		right[i]           = b & mask;
		const uint16_t key = (uint16_t)(b >> rbw);
		uint16_t       idx = st->actual_dictionary_size;
		for (int d = 0; d < st->actual_dictionary_size; d++) {
			if (st->left_parts_dict[d] == key) {
				idx = (uint16_t)d;
				break;
			}
		}
		if (idx == st->actual_dictionary_size) {
			for (int d = st->actual_dictionary_size + 1; d < st->rd_sorted_count; d++) {
				if (st->rd_sorted_left[d] == key) {
					idx = (uint16_t)d;
					break;
				}
			}
		}
		left[i] = idx;
		if (idx >= st->actual_dictionary_size) {
			exc[c] = key;
			pos[c] = (uint16_t)i;
			c++;
		}
	}
	*cnt = c;
}

/* rd.hpp:152-178 */
void alpof_rd_decode(float* out, const uint32_t* right, const uint16_t* left, const uint16_t* exc, const uint16_t* pos,
                     uint16_t cnt, const alpo_state* st) {
	const int rbw = st->right_bit_width;
	for (int i = 0; i < VECTOR_SIZE; i++) {
		const uint16_t l = left[i] < 8 ? st->left_parts_dict[left[i]] : 0;
		const uint32_t u = ((uint32_t)l << rbw) | right[i];
		memcpy(&out[i], &u, 4);
	}
	for (uint16_t j = 0; j < cnt; j++) {
		const uint32_t u = ((uint32_t)exc[j] << rbw) | right[pos[j]];
		memcpy(&out[pos[j]], &u, 4);
	}
}

/* ---- whole-column drivers, same contract as ref_harness.cpp:reff_encode_column ----------------------------------- */
void alpof_encode_column(const float* column, size_t n_vectors, uint8_t* scheme, uint8_t* e, uint8_t* f, uint8_t* bw,
                         uint8_t* lbw, int64_t* base, uint16_t* exc_cnt, int32_t* packed, uint16_t* packed_left, float* exc,
                         uint16_t* pos, uint16_t* dict, uint8_t* dict_size, uint8_t* k_out, int* combos_out) {
	const size_t n_values = n_vectors * VECTOR_SIZE;
	alpo_state*  st       = (alpo_state*)calloc(1, sizeof(alpo_state));
	float*       sample   = (float*)calloc(VECTOR_SIZE, sizeof(float));
	int32_t*     enc      = (int32_t*)malloc(VECTOR_SIZE * sizeof(int32_t));
	uint32_t*    right    = (uint32_t*)malloc(VECTOR_SIZE * sizeof(uint32_t));
	uint16_t*    left     = (uint16_t*)malloc(VECTOR_SIZE * sizeof(uint16_t));
	uint16_t*    rd_exc   = (uint16_t*)malloc(VECTOR_SIZE * sizeof(uint16_t));
	for (size_t v = 0; v < n_vectors; v++) {
		const size_t rg = v / N_VECTORS_PER_ROWGROUP;
		if (v % N_VECTORS_PER_ROWGROUP == 0) {
			memset(st, 0, sizeof(*st));
			alpof_encoder_init(column, rg * ROWGROUP_SIZE, n_values, sample, st);
			if (st->scheme == ALPO_SCHEME_ALP_RD) { alpof_rd_init(column, rg * ROWGROUP_SIZE, n_values, sample, st); }
			if (dict) {
				memcpy(dict + rg * 8, st->left_parts_dict, 16);
				dict_size[rg] = st->scheme == ALPO_SCHEME_ALP_RD ? st->actual_dictionary_size : 0;
			}
			if (k_out) {
				k_out[rg] = st->scheme == ALPO_SCHEME_ALP ? (uint8_t)st->k_combinations : 0;
				for (int i = 0; i < 5; i++) {
					const int have                  = st->scheme == ALPO_SCHEME_ALP && i < st->k_combinations;
					combos_out[rg * 10 + 2 * i]     = have ? st->combos[i][0] : -1;
					combos_out[rg * 10 + 2 * i + 1] = have ? st->combos[i][1] : -1;
				}
			}
		}
		const float* in = column + v * VECTOR_SIZE;
		scheme[v]       = (uint8_t)st->scheme;
		uint16_t cnt    = 0;
		if (st->scheme == ALPO_SCHEME_ALP) {
			uint8_t b  = 0;
			int32_t bs = 0;
			alpof_encode(in, exc + v * VECTOR_SIZE, pos + v * VECTOR_SIZE, &cnt, enc, st);
			alpof_analyze_ffor(enc, &b, &bs);
			memset(packed + v * VECTOR_SIZE, 0, 4096);
			alpof_ffor_u32((const uint32_t*)enc, (uint32_t*)(packed + v * VECTOR_SIZE), b, (uint32_t)bs);
			e[v] = st->exp, f[v] = st->fac, bw[v] = b, lbw[v] = 0, base[v] = bs, exc_cnt[v] = cnt;
		} else {
			alpof_rd_encode(in, rd_exc, pos + v * VECTOR_SIZE, &cnt, right, left, st);
			memset(packed + v * VECTOR_SIZE, 0, 4096);
			memset(packed_left + v * VECTOR_SIZE, 0, 2048);
			alpof_ffor_u32(right, (uint32_t*)(packed + v * VECTOR_SIZE), st->right_bit_width, 0);
			alpo_ffor_u16(left, packed_left + v * VECTOR_SIZE, st->left_bit_width, 0);
			memcpy(exc + v * VECTOR_SIZE, rd_exc, (size_t)cnt * 2);
			e[v] = 0, f[v] = 0, bw[v] = st->right_bit_width, lbw[v] = st->left_bit_width, base[v] = 0, exc_cnt[v] = cnt;
		}
	}
	free(st), free(sample), free(enc), free(right), free(left), free(rd_exc);
}

void alpof_decode_column(size_t n_vectors, const uint8_t* scheme, const uint8_t* e, const uint8_t* f, const uint8_t* bw,
                         const uint8_t* lbw, const int64_t* base, const uint16_t* exc_cnt, const int32_t* packed,
                         const uint16_t* packed_left, const float* exc, const uint16_t* pos, const uint16_t* dict, float* out) {
	uint32_t*   right = (uint32_t*)malloc(VECTOR_SIZE * sizeof(uint32_t));
	uint16_t*   left  = (uint16_t*)malloc(VECTOR_SIZE * sizeof(uint16_t));
	alpo_state* st    = (alpo_state*)calloc(1, sizeof(alpo_state));
	for (size_t v = 0; v < n_vectors; v++) {
		float* o = out + v * VECTOR_SIZE;
		if (scheme[v] == ALPO_SCHEME_ALP) {
			alpof_falp((const uint32_t*)(packed + v * VECTOR_SIZE), o, bw[v], (uint32_t)(int32_t)base[v], f[v], e[v]);
			alpof_patch(o, exc + v * VECTOR_SIZE, pos + v * VECTOR_SIZE, exc_cnt[v]);
		} else {
			st->right_bit_width = bw[v];
			st->left_bit_width  = lbw[v];
			memcpy(st->left_parts_dict, dict + (v / N_VECTORS_PER_ROWGROUP) * 8, 16);
			alpof_unffor_u32((const uint32_t*)(packed + v * VECTOR_SIZE), right, bw[v], 0);
			alpo_unffor_u16(packed_left + v * VECTOR_SIZE, left, lbw[v], 0);
			alpof_rd_decode(o, right, left, (const uint16_t*)(exc + v * VECTOR_SIZE), pos + v * VECTOR_SIZE, exc_cnt[v], st);
		}
	}
	free(right), free(left), free(st);
}
