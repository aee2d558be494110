/*
 * oracle/alp_oracle.c — TEST INFRASTRUCTURE ONLY.  NOT PRODUCT CODE.
 *
 * A plain-C, scalar CPU restatement of the cwida/ALP vector hot path (double precision): rowgroup
 * sampling, (exponent,factor) search, decimal->integer encode with exception compaction, FFOR
 * bit-(un)packing in the FastLanes interleaved layout, fused decode, and the ALP_RD left-dictionary
 * split.  It exists so that the HIP kernels in alp_amd/csrc can be checked bit-for-bit on any input.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * library (libalpgpu.so) never links it and has no CPU fallback.
 *
 * PINNING: this restatement is pinned (tests/test_oracle_vs_ref.py, tests/test_golden.py) against
 *   (1) the real reference compiled in this container from /root/reference (oracle/_ref/libalp_ref.so,
 *       recipe: oracle/Makefile), value-for-value on the reference's own test data and random inputs, and
 *   (2) golden fixtures in tests/golden/ generated from that build by tools/make_golden.py, which
 *       include the known-answer (bit width, exception count) pairs the reference's own unit test
 *       asserts (/root/reference/test/test_alp_sample.cpp:174-179, data/include/double/alp_dataset.hpp,
 *       data/include/generated_columns.hpp, data/include/edge_case.hpp).
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * All arithmetic is IEEE-754 binary64, round-to-nearest-even, no FMA contraction (-ffp-contract=off).
 */
#define _POSIX_C_SOURCE 200809L
#include "alp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- constants: include/alp/constants.hpp:66-154, :16-18; include/alp/config.hpp:11-26 ------------- */
#define VECTOR_SIZE 1024
#define N_VECTORS_PER_ROWGROUP 100
#define ROWGROUP_SIZE (N_VECTORS_PER_ROWGROUP * VECTOR_SIZE)
#define ROWGROUP_SAMPLES_JUMP 12 /* (ROWGROUP_SIZE / 8) / VECTOR_SIZE, config.hpp:19 */
#define SAMPLES_PER_VECTOR 32
#define MAX_K_COMBINATIONS 5
#define CUTTING_LIMIT 16
#define MAX_RD_DICTIONARY_SIZE 8
#define SAMPLING_EARLY_EXIT_THRESHOLD 2
#define EXCEPTION_SIZE 64
#define EXCEPTION_POSITION_SIZE 16
#define RD_EXCEPTION_SIZE 16
#define RD_EXCEPTION_POSITION_SIZE 16
#define RD_SIZE_THRESHOLD_LIMIT (48 * SAMPLES_PER_VECTOR)
#define MAX_EXPONENT 18

static const double ENCODING_UPPER_LIMIT = 9223372036854774784.0;
static const double ENCODING_LOWER_LIMIT = -9223372036854774784.0;
static const double MAGIC_NUMBER         = 6755399441055744.0; /* 0x0018000000000000 = 2^52 + 2^51 */

static const double FRAC_ARR[21] = {
    1.0,
    0.1,
    0.01,
    0.001,
    0.0001,
    0.00001,
    0.000001,
    0.0000001,
    0.00000001,
    0.000000001,
    0.0000000001,
    0.00000000001,
    0.000000000001,
    0.0000000000001,
    0.00000000000001,
    0.000000000000001,
    0.0000000000000001,
    0.00000000000000001,
    0.000000000000000001,
    0.0000000000000000001,
    0.00000000000000000001,
};

static const double EXP_ARR[24] = {
    1.0,
    10.0,
    100.0,
    1000.0,
    10000.0,
    100000.0,
    1000000.0,
    10000000.0,
    100000000.0,
    1000000000.0,
    10000000000.0,
    100000000000.0,
    1000000000000.0,
    10000000000000.0,
    100000000000000.0,
    1000000000000000.0,
    10000000000000000.0,
    100000000000000000.0,
    1000000000000000000.0,
    10000000000000000000.0,
    100000000000000000000.0,
    1000000000000000000000.0,
    10000000000000000000000.0,
    100000000000000000000000.0,
};

static const int64_t FACT_ARR[19] = {1LL,
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

/* ---- scalar kernels ------------------------------------------------------------------------------- */

/* static_cast<int64_t>(double) as x86-64 executes it (cvttsd2si / vcvttpd2qq): truncate toward zero;
 * NaN or |x| >= 2^63 -> 0x8000000000000000.  The reference relies on this (UB in C++) at
 * include/alp/encoder.hpp:88; SURVEY.md hard part H2. */
int64_t alpo_cast64(double x) {
	if (!(x > -9223372036854775808.0 && x < 9223372036854775808.0)) { /* NaN fails both */
		return INT64_MIN;
	}
	/* x in (-2^63, 2^63): -2^63 itself also maps to INT64_MIN, consistent with the branch above */
	return (int64_t)x;
}

/* include/alp/encoder.hpp:75-78 */
static int is_impossible_to_encode(double n) {
	return !isfinite(n) || isnan(n) || n > ENCODING_UPPER_LIMIT || n < ENCODING_LOWER_LIMIT || (n == 0.0 && signbit(n));
}

/* include/alp/encoder.hpp:81-89, SAFE = true */
int64_t alpo_encode_value_safe(double v, int fac, int exp) {
	double t = v * EXP_ARR[exp];
	t        = t * FRAC_ARR[fac];
	if (is_impossible_to_encode(t)) { return (int64_t)ENCODING_UPPER_LIMIT; }
	t = t + MAGIC_NUMBER;
	t = t - MAGIC_NUMBER;
	return alpo_cast64(t);
}

/* include/alp/encoder.hpp:81-89, SAFE = false */
int64_t alpo_encode_value_unsafe(double v, int fac, int exp) {
	double t = v * EXP_ARR[exp];
	t        = t * FRAC_ARR[fac];
	t        = t + MAGIC_NUMBER;
	t        = t - MAGIC_NUMBER;
	return alpo_cast64(t);
}

/* include/alp/decoder.hpp:128-131: int64 * int64 (wraps), then int64 -> double, then one FP multiply */
double alpo_decode_value(int64_t enc, int fac, int exp) {
	const int64_t m = (int64_t)((uint64_t)enc * (uint64_t)FACT_ARR[fac]);
	return (double)m * FRAC_ARR[exp];
}

/* include/alp/encoder.hpp:91-106 */
int alpo_count_bits(int64_t max, int64_t min) {
	const uint64_t delta = (uint64_t)max - (uint64_t)min;
	if (delta == 0) { return 0; }
	return 64 - __builtin_clzll(delta);
}

/* ---- first-level sampler: include/alp/sampler.hpp:14-52 -------------------------------------------- */
size_t alpo_first_level_sample(const double* data, size_t data_offset, size_t data_size, double* data_sample) {
	const size_t left_in_data      = data_size - data_offset;
	const size_t portion_to_sample = left_in_data < (size_t)ROWGROUP_SIZE ? left_in_data : (size_t)ROWGROUP_SIZE;
	const size_t available_vectors = (size_t)ceil((double)portion_to_sample / VECTOR_SIZE);
	size_t       sample_idx        = 0;
	size_t       data_idx          = data_offset;

	for (size_t vector_idx = 0; vector_idx < available_vectors; vector_idx++) {
		const size_t rem = data_size - data_idx;
		const size_t n   = rem < (size_t)VECTOR_SIZE ? rem : (size_t)VECTOR_SIZE;
		if ((vector_idx % ROWGROUP_SAMPLES_JUMP) != 0) { /* sampler.hpp:29-33 */
			data_idx += n;
			continue;
		}
		int32_t inc = (int32_t)ceil((double)n / SAMPLES_PER_VECTOR); /* sampler.hpp:35-37 */
		if (inc < 1) { inc = 1; }
		if (n < SAMPLES_PER_VECTOR && sample_idx != 0) { /* sampler.hpp:39-44 */
			data_idx += n;
			continue;
		}
		for (size_t i = 0; i < n; i += (size_t)inc) { /* sampler.hpp:46-49 */
			data_sample[sample_idx++] = data[data_idx + i];
		}
		data_idx += n;
	}
	return sample_idx;
}

/* ---- first-level (e,f) search + scheme decision: include/alp/encoder.hpp:139-235 ---------------------- */
typedef struct {
	int e, f, count;
} combo_t;

/* include/alp/encoder.hpp:128-132 — a strict total order (count desc, exponent desc, factor desc) */
static int combo_before(const combo_t* a, const combo_t* b) {
	return (a->count > b->count) || (a->count == b->count && b->e < a->e) ||
	       (a->count == b->count && b->e == a->e && b->f < a->f);
}

void alpo_find_top_k(const double* smp, alpo_state* st) {
	const uint64_t n_vectors_to_sample = (uint64_t)ceil((double)st->sampled_values_n / SAMPLES_PER_VECTOR);
	const uint64_t samples_size = st->sampled_values_n < SAMPLES_PER_VECTOR ? st->sampled_values_n : SAMPLES_PER_VECTOR;
	combo_t        global[16];
	int            n_global   = 0;
	uint64_t       smp_offset = 0;

	uint64_t best_size = samples_size * (EXCEPTION_SIZE + EXCEPTION_POSITION_SIZE) + samples_size * EXCEPTION_SIZE;
	for (uint64_t smp_n = 0; smp_n < n_vectors_to_sample; smp_n++) {
		int      found_factor = 0, found_exponent = 0;
		uint64_t sample_size = samples_size * (EXCEPTION_SIZE + EXCEPTION_POSITION_SIZE) + samples_size * EXCEPTION_SIZE;
		for (int e = MAX_EXPONENT; e >= 0; e--) {
			for (int f = e; f >= 0; f--) {
				uint16_t exc = 0, non_exc = 0;
				int64_t  mx = INT64_MIN, mn = INT64_MAX;
				for (uint64_t i = 0; i < samples_size; i++) {
					const double  v   = smp[smp_offset + i];
					const int64_t enc = alpo_encode_value_safe(v, f, e);
					const double  dec = alpo_decode_value(enc, f, e);
					if (dec == v) {
						non_exc++;
						if (enc > mx) { mx = enc; }
						if (enc < mn) { mn = enc; }
					} else {
						exc++;
					}
				}
				if (non_exc < 2) { continue; } /* encoder.hpp:182 */
				const uint32_t bits = (uint32_t)alpo_count_bits(mx, mn);
				uint64_t       size = samples_size * bits + (uint64_t)exc * (EXCEPTION_SIZE + EXCEPTION_POSITION_SIZE);
				if ((size < sample_size) || (size == sample_size && found_exponent < e) ||
				    (size == sample_size && found_exponent == e && found_factor < f)) { /* encoder.hpp:190-197 */
					sample_size    = size;
					found_exponent = e;
					found_factor   = f;
					if (sample_size < best_size) { best_size = sample_size; }
				}
			}
		}
		int hit = -1; /* global_combinations[(e,f)]++, encoder.hpp:207-208 */
		for (int i = 0; i < n_global; i++) {
			if (global[i].e == found_exponent && global[i].f == found_factor) { hit = i; }
		}
		if (hit < 0) {
			global[n_global].e     = found_exponent;
			global[n_global].f     = found_factor;
			global[n_global].count = 1;
			n_global++;
		} else {
			global[hit].count++;
		}
		smp_offset += samples_size;
	}

	if (best_size >= RD_SIZE_THRESHOLD_LIMIT) { /* encoder.hpp:213-216 */
		st->scheme = ALPO_SCHEME_ALP_RD;
		return;
	}
	/* sort (total order => any algorithm gives the reference's result), encoder.hpp:228 */
	for (int i = 1; i < n_global; i++) {
		combo_t c = global[i];
		int     j = i - 1;
		while (j >= 0 && combo_before(&c, &global[j])) {
			global[j + 1] = global[j];
			j--;
		}
		global[j + 1] = c;
	}
	if (n_global < st->k_combinations) { st->k_combinations = n_global; }
	for (int i = 0; i < st->k_combinations; i++) {
		st->combos[i][0] = global[i].e;
		st->combos[i][1] = global[i].f;
	}
}

/* include/alp/encoder.hpp:420-427 */
void alpo_encoder_init(const double* col, size_t off, size_t n, double* sample_arr, alpo_state* st) {
	st->scheme           = ALPO_SCHEME_ALP;
	st->sampled_values_n = alpo_first_level_sample(col, off, n, sample_arr);
	st->k_combinations   = MAX_K_COMBINATIONS;
	for (int i = 0; i < 5; i++) { st->combos[i][0] = st->combos[i][1] = -1; }
	alpo_find_top_k(sample_arr, st);
}

/* ---- second-level sampling: include/alp/encoder.hpp:241-305 ----------------------------------------- */
void alpo_find_best_ef(const alpo_state* st, const double* in, int vector_size, uint8_t* fac, uint8_t* exp) {
	uint8_t  found_exponent = 0, found_factor = 0;
	uint64_t best_size   = 0;
	uint8_t  worse_count = 0;
	int32_t  inc         = vector_size / SAMPLES_PER_VECTOR; /* integer division inside ceil(), encoder.hpp:253 */
	if (inc < 1) { inc = 1; }
	for (int k = 0; k < st->k_combinations; k++) {
		const int e = st->combos[k][0], f = st->combos[k][1];
		uint32_t  exc = 0;
		int64_t   mx = INT64_MIN, mn = INT64_MAX;
		for (int i = 0; i < vector_size; i += inc) {
			const double  v   = in[i];
			const int64_t enc = alpo_encode_value_safe(v, f, e);
			const double  dec = alpo_decode_value(enc, f, e);
			if (dec == v) {
				if (enc > mx) { mx = enc; }
				if (enc < mn) { mn = enc; }
			} else {
				exc++;
			}
		}
		const uint32_t bits = (uint32_t)alpo_count_bits(mx, mn); /* all-exception quirk: bits(INT64_MIN,INT64_MAX)=1 */
		const uint64_t size = (uint64_t)SAMPLES_PER_VECTOR * bits + (uint64_t)exc * (EXCEPTION_SIZE + EXCEPTION_POSITION_SIZE);
		if (k == 0) {
			best_size      = size;
			found_factor   = (uint8_t)f;
			found_exponent = (uint8_t)e;
			continue;
		}
		if (size >= best_size) {
			worse_count++;
			if (worse_count == SAMPLING_EARLY_EXIT_THRESHOLD) { break; }
			continue;
		}
		best_size      = size;
		found_factor   = (uint8_t)f;
		found_exponent = (uint8_t)e;
		worse_count    = 0;
	}
	*exp = found_exponent;
	*fac = found_factor;
}

/* ---- vector encode: include/alp/encoder.hpp:307-400 (scalar compaction branch :373-379 = canonical) -- */
void alpo_encode_simdized(const double* in, double* exc, uint16_t* pos, uint16_t* cnt, int64_t* enc, int fac, int exp) {
	static _Thread_local double   dec_arr[VECTOR_SIZE];
	static _Thread_local double   val_arr[VECTOR_SIZE];
	static _Thread_local uint64_t idx_arr[VECTOR_SIZE + 1];
	/* pass 1 (encoder.hpp:326-338).  For doubles the EXPONENTIAL_BITS_MASK literal has 65 binary digits
	 * (constants.hpp:82-83) and evaluates to 0xFFE0000000000000, so "(bits & SIGN_BIT_MASK) >= mask" is
	 * never true: only -0.0 is replaced by the sentinel.  NaN/Inf flow through and become exceptions. */
	for (int i = 0; i < VECTOR_SIZE; i++) {
		uint64_t b;
		memcpy(&b, &in[i], 8);
		const int special = ((b & 0x7FFFFFFFFFFFFFFFULL) >= 0xFFE0000000000000ULL) || b == 0x8000000000000000ULL;
		val_arr[i]        = special ? ENCODING_UPPER_LIMIT : in[i];
	}
	/* pass 2 (encoder.hpp:341-349) */
	for (int i = 0; i < VECTOR_SIZE; i++) {
		enc[i]     = alpo_encode_value_unsafe(val_arr[i], fac, exp);
		dec_arr[i] = alpo_decode_value(enc[i], fac, exp);
	}
	/* pass 3, scalar branch (encoder.hpp:373-379): branch-free compaction, writes idx_arr[n] for every i */
	uint64_t n = 0;
	for (int i = 0; i < VECTOR_SIZE; i++) {
		const int is_exc = dec_arr[i] != val_arr[i];
		idx_arr[n]       = (uint64_t)i;
		n += (uint64_t)is_exc;
	}
	/* pass 4 (encoder.hpp:382-388): first i with idx_arr[i] != i.  After the scalar pass 3, idx_arr[0..n-1]
	 * are the exception positions and idx_arr[n] == 1023 when n < 1024 and position 1023 is not an
	 * exception... more precisely idx_arr[n] holds the last i written while the cursor stood at n.
	 * Entries above n are never reached because the scan stops at or before index n, EXCEPT when
	 * idx_arr[i] == i for all i <= n, which needs position n..1023 pattern analysed below; we reproduce
	 * the loop literally, and keep entries above n equal to their index (a neutral stand-in for the
	 * reference's stale thread_local contents, which the loop can only reach when n == 1024 or when
	 * idx_arr[n] == n, i.e. n == 1023 and positions 0..1022 are the exceptions). */
	int64_t filler = 0;
	for (uint64_t i = n + 1; i < VECTOR_SIZE; i++) { idx_arr[i] = i; }
	for (uint64_t i = 0; i < VECTOR_SIZE; i++) {
		if (i != idx_arr[i]) {
			filler = enc[i];
			break;
		}
	}
	/* pass 5 (encoder.hpp:390-399) */
	uint16_t c = 0;
	for (uint64_t j = 0; j < n; j++) {
		const uint64_t i = idx_arr[j];
		enc[i]           = filler;
		exc[c]           = in[i];
		pos[c]           = (uint16_t)i;
		c++;
	}
	*cnt = c;
}

/* include/alp/encoder.hpp:402-418 */
void alpo_encode(const double* in, double* exc, uint16_t* pos, uint16_t* cnt, int64_t* enc, alpo_state* st) {
	if (st->k_combinations > 1) {
		alpo_find_best_ef(st, in, VECTOR_SIZE, &st->fac, &st->exp);
	} else {
		st->exp = (uint8_t)st->combos[0][0];
		st->fac = (uint8_t)st->combos[0][1];
	}
	alpo_encode_simdized(in, exc, pos, cnt, enc, st->fac, st->exp);
}

/* include/alp/encoder.hpp:109-120 */
void alpo_analyze_ffor(const int64_t* in, uint8_t* bw, int64_t* base) {
	int64_t mn = INT64_MAX, mx = INT64_MIN;
	for (int i = 0; i < VECTOR_SIZE; i++) {
		if (in[i] < mn) { mn = in[i]; }
		if (in[i] > mx) { mx = in[i]; }
	}
	*bw   = (uint8_t)alpo_count_bits(mx, mn);
	*base = mn;
}

/* ---- FFOR, closed form of the generated kernels ------------------------------------------------------
 * u64: src/fastlanes_generated_ffor.cpp:7379-29749 (dispatch :29939), src/fastlanes_generated_unffor.cpp
 * :5905-22811 (dispatch :23010).  16 lane-streams x 64 rows; value i -> lane = i%16, row = i/16; the
 * lane's stream is the LSB-first concatenation of its 64 bw-bit fields; stream word k lives at
 * out[16*k + lane].  bw = 0 writes nothing; bw = 64 does not mask; bw > 64 is a no-op (switch w/o default). */
void alpo_ffor_u64(const uint64_t* in, uint64_t* out, int bw, uint64_t base) {
	if (bw <= 0 || bw > 64) { return; }
	const uint64_t mask = bw == 64 ? ~0ULL : ((1ULL << bw) - 1);
	for (int w = 0; w < 16 * bw; w++) { out[w] = 0; }
	for (int i = 0; i < VECTOR_SIZE; i++) {
		const int      lane = i & 15, row = i >> 4;
		const uint64_t v = (in[i] - base) & mask;
		const int      p = row * bw, k = p >> 6, s = p & 63;
		out[16 * k + lane] |= v << s;
		if (s + bw > 64) { out[16 * (k + 1) + lane] |= v >> (64 - s); }
	}
}

void alpo_unffor_u64(const uint64_t* in, uint64_t* out, int bw, uint64_t base) {
	if (bw < 0 || bw > 64) { return; }
	if (bw == 0) { /* generated_unffor.cpp:5905-5935: fill with base */
		for (int i = 0; i < VECTOR_SIZE; i++) { out[i] = base; }
		return;
	}
	const uint64_t mask = bw == 64 ? ~0ULL : ((1ULL << bw) - 1);
	for (int i = 0; i < VECTOR_SIZE; i++) {
		const int lane = i & 15, row = i >> 4;
		const int p = row * bw, k = p >> 6, s = p & 63;
		uint64_t  v = in[16 * k + lane] >> s;
		if (s + bw > 64) { v |= in[16 * (k + 1) + lane] << (64 - s); }
		out[i] = (v & mask) + base;
	}
}

/* u16: generated_ffor.cpp:357-1775 (dispatch :29781), generated_unffor.cpp:347-1525 (dispatch :22846).
 * 64 lane-streams x 16 rows; lane = i%64, row = i/64; words are u16 at out[64*k + lane]. */
void alpo_ffor_u16(const uint16_t* in, uint16_t* out, int bw, uint16_t base) {
	if (bw <= 0 || bw > 16) { return; }
	const uint16_t mask = bw == 16 ? 0xFFFF : (uint16_t)((1u << bw) - 1);
	for (int w = 0; w < 64 * bw; w++) { out[w] = 0; }
	for (int i = 0; i < VECTOR_SIZE; i++) {
		const int      lane = i & 63, row = i >> 6;
		const uint16_t v = (uint16_t)((uint16_t)(in[i] - base) & mask);
		const int      p = row * bw, k = p >> 4, s = p & 15;
		out[64 * k + lane] |= (uint16_t)(v << s);
		if (s + bw > 16) { out[64 * (k + 1) + lane] |= (uint16_t)(v >> (16 - s)); }
	}
}

void alpo_unffor_u16(const uint16_t* in, uint16_t* out, int bw, uint16_t base) {
	if (bw < 0 || bw > 16) { return; }
	if (bw == 0) {
		for (int i = 0; i < VECTOR_SIZE; i++) { out[i] = base; }
		return;
	}
	const uint16_t mask = bw == 16 ? 0xFFFF : (uint16_t)((1u << bw) - 1);
	for (int i = 0; i < VECTOR_SIZE; i++) {
		const int lane = i & 63, row = i >> 6;
		const int p = row * bw, k = p >> 4, s = p & 15;
		uint32_t  v = (uint32_t)in[64 * k + lane] >> s;
		if (s + bw > 16) { v |= (uint32_t)in[64 * (k + 1) + lane] << (16 - s); }
		out[i] = (uint16_t)((v & mask) + base);
	}
}

/* ---- decode ------------------------------------------------------------------------------------------ */
/* u8: src/fastlanes_generated_ffor.cpp:4-356 (dispatch :29749 ff.).  128 lane-streams x 8 rows; value i -> lane = i % 128,
 * row = i / 128; stream word k lives at out[128*k + lane].  bw = 0 writes nothing; bw = 8 does not mask; bw > 8 is a no-op. */
void alpo_ffor_u8(const uint8_t* in, uint8_t* out, int bw, uint8_t base) {
	if (bw <= 0 || bw > 8) { return; }
	const unsigned mask = (1u << bw) - 1u;
	for (int w = 0; w < 128 * bw; w++) { out[w] = 0; }
	for (int i = 0; i < VECTOR_SIZE; i++) {
		const int      lane = i % 128, row = i / 128;
		const unsigned v = ((unsigned)(uint8_t)(in[i] - base)) & mask;
		const int      p = row * bw, k = p / 8, s = p % 8;
		out[128 * k + lane] |= (uint8_t)(v << s);
		if (s + bw > 8) { out[128 * (k + 1) + lane] |= (uint8_t)(v >> (8 - s)); }
	}
}

void alpo_unffor_u8(const uint8_t* in, uint8_t* out, int bw, uint8_t base) {
	if (bw < 0 || bw > 8) { return; }
	if (bw == 0) {
		for (int i = 0; i < VECTOR_SIZE; i++) { out[i] = base; }
		return;
	}
	const unsigned mask = (1u << bw) - 1u;
	for (int i = 0; i < VECTOR_SIZE; i++) {
		const int lane = i % 128, row = i / 128;
		const int p = row * bw, k = p / 8, s = p % 8;
		unsigned  v = (unsigned)in[128 * k + lane] >> s;
		if (s + bw > 8) { v |= (unsigned)in[128 * (k + 1) + lane] << (8 - s); }
		out[i] = (uint8_t)((v & mask) + base);
	}
}

/* include/alp/decoder.hpp:134-138 */
void alpo_decode(const int64_t* enc, int fac, int exp, double* out) {
	for (int i = 0; i < VECTOR_SIZE; i++) { out[i] = alpo_decode_value(enc[i], fac, exp); }
}

/* src/falp.cpp:42440-42643 dispatch -> falp_<bw>bw_64ow_64crw_1uf (:11-33761); per value :114-121:
 * unpack, + base, * FACT (u64 wrap), (int64) -> double, * FRAC.  bw = 64 in the reference is broken
 * (falp.cpp:33313-33319 multiplies the u64 by the double) and unreachable through the
 * codec (such data goes to ALP_RD); here bw = 64 follows unffor+decode. */
void alpo_falp(const uint64_t* in, double* out, int bw, uint64_t base, int fac, int exp) {
	static _Thread_local uint64_t tmp[VECTOR_SIZE];
	if (bw < 0 || bw > 64) { return; }
	alpo_unffor_u64(in, tmp, bw, base);
	for (int i = 0; i < VECTOR_SIZE; i++) { out[i] = alpo_decode_value((int64_t)tmp[i], fac, exp); }
}

/* include/alp/decoder.hpp:141-149 */
void alpo_patch(double* out, const double* exc, const uint16_t* pos, uint16_t cnt) {
	for (uint16_t i = 0; i < cnt; i++) { out[pos[i]] = exc[i]; }
}

/* ---- ALP_RD ------------------------------------------------------------------------------------------
 * include/alp/rd.hpp:33-87 builds the left-part histogram in a std::unordered_map<uint64_t,int32_t>,
 * copies it (in the map's iteration order) to a vector and std::sort()s it by count only.  Ties are
 * therefore resolved by libstdc++ implementation details.  To be able to pin this
 * restatement against the reference AS BUILT HERE (libstdc++ 11: identity hash, _Mod_range_hashing,
 * _Prime_rehash_policy, singly-linked node list; introsort + final insertion sort), the two containers'
 * observable orders are emulated below.  This is deliberately literal. */

#include "libstdcxx_order.h"

/* include/alp/rd.hpp:23-31 */
static double rd_estimate(int rbw, int lbw, uint32_t exceptions_count, uint64_t sample_count) {
	const double exceptions_size = (double)(exceptions_count * (RD_EXCEPTION_POSITION_SIZE + RD_EXCEPTION_SIZE));
	return (double)rbw + (double)lbw + (exceptions_size / (double)sample_count);
}

/* include/alp/rd.hpp:33-87 */
static double rd_build_dictionary(const double* in_p, int right_bit_width, alpo_state* st, int persist) {
	static _Thread_local ht_t  h;
	static _Thread_local rep_t sorted[VECTOR_SIZE];
	ht_init(&h);
	for (size_t i = 0; i < st->sampled_values_n; i++) {
		uint64_t b;
		memcpy(&b, &in_p[i], 8);
		ht_increment(&h, b >> right_bit_width);
	}
	int n_sorted = 0;
	for (int p = h.bb_next; p != HT_NULL; p = h.nodes[p].next) { /* map iteration order */
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
	return rd_estimate(right_bit_width, lbw, exceptions_count, st->sampled_values_n);
}

/* include/alp/rd.hpp:89-104 and :180-185 */
void alpo_rd_init(const double* col, size_t off, size_t n, double* sample_arr, alpo_state* st) {
	st->scheme           = ALPO_SCHEME_ALP_RD;
	st->sampled_values_n = alpo_first_level_sample(col, off, n, sample_arr);
	int    right_bit_width = 0;
	double best            = 1.7976931348623157e308;
	for (int i = 1; i <= CUTTING_LIMIT; i++) {
		const int    cand = 64 - i;
		const double est  = rd_build_dictionary(sample_arr, cand, st, 0);
		if (est < best) {
			right_bit_width = cand;
			best            = est;
		}
	}
	rd_build_dictionary(sample_arr, right_bit_width, st, 1);
}

/* include/alp/rd.hpp:109-147.  Index of a left part: its dictionary position; else, if the reference's
 * left_parts_dict_map holds it (sorted positions > dict_size, rd.hpp:75), that position; else dict_size. */
void alpo_rd_encode(const double* in, uint16_t* exc, uint16_t* pos, uint16_t* cnt, uint64_t* right, uint16_t* left,
                    const alpo_state* st) {
	const int      rbw  = st->right_bit_width;
	const uint64_t mask = (1ULL << rbw) - 1;
	uint16_t       c    = 0;
	for (int i = 0; i < VECTOR_SIZE; i++) {
		uint64_t b;
		memcpy(&b, &in[i], 8);
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

/* include/alp/rd.hpp:152-178 */
void alpo_rd_decode(double* out, const uint64_t* right, const uint16_t* left, const uint16_t* exc, const uint16_t* pos,
                    uint16_t cnt, const alpo_state* st) {
	const int rbw = st->right_bit_width;
	for (int i = 0; i < VECTOR_SIZE; i++) {
		/* left[i] may exceed 7 only at exception slots; the reference indexes past the dictionary there
		 * (rd.hpp:166) and overwrites the slot afterwards; guard the read, same observable result */
		const uint16_t l = left[i] < 8 ? st->left_parts_dict[left[i]] : 0;
		const uint64_t u = ((uint64_t)l << rbw) | right[i];
		memcpy(&out[i], &u, 8);
	}
	for (uint16_t j = 0; j < cnt; j++) {
		const uint64_t u = ((uint64_t)exc[j] << rbw) | right[pos[j]];
		memcpy(&out[pos[j]], &u, 8);
	}
}

/* ---- whole-column drivers (loop shape: publication/source_code/bench_compression_ratio/alp.cpp:198-229,
 * with rowgroup r <-> vectors 100r..100r+99 and a per-rowgroup scheme) ------------------ */
void alpo_encode_column(const double* column, size_t n_vectors, uint8_t* scheme, uint8_t* e, uint8_t* f, uint8_t* bw,
                        uint8_t* lbw, int64_t* base, uint16_t* exc_cnt, int64_t* packed, uint16_t* packed_left,
                        double* exc, uint16_t* pos, uint16_t* dict, uint8_t* dict_size, uint8_t* k_out,
                        int* combos_out) {
	const size_t n_values = n_vectors * VECTOR_SIZE;
	alpo_state*  st       = (alpo_state*)calloc(1, sizeof(alpo_state));
	double*      sample   = (double*)calloc(VECTOR_SIZE, sizeof(double));
	int64_t*     enc      = (int64_t*)malloc(VECTOR_SIZE * sizeof(int64_t));
	uint64_t*    right    = (uint64_t*)malloc(VECTOR_SIZE * sizeof(uint64_t));
	uint16_t*    left     = (uint16_t*)malloc(VECTOR_SIZE * sizeof(uint16_t));
	uint16_t*    rd_exc   = (uint16_t*)malloc(VECTOR_SIZE * sizeof(uint16_t));
	for (size_t v = 0; v < n_vectors; v++) {
		const size_t rg = v / N_VECTORS_PER_ROWGROUP;
		if (v % N_VECTORS_PER_ROWGROUP == 0) {
			memset(st, 0, sizeof(*st));
			alpo_encoder_init(column, rg * ROWGROUP_SIZE, n_values, sample, st);
			if (st->scheme == ALPO_SCHEME_ALP_RD) { alpo_rd_init(column, rg * ROWGROUP_SIZE, n_values, sample, st); }
			if (dict) {
				memcpy(dict + rg * 8, st->left_parts_dict, 16);
				dict_size[rg] = st->scheme == ALPO_SCHEME_ALP_RD ? st->actual_dictionary_size : 0;
			}
			if (k_out) {
				k_out[rg] = st->scheme == ALPO_SCHEME_ALP ? (uint8_t)st->k_combinations : 0;
				for (int i = 0; i < 5; i++) {
					const int have            = st->scheme == ALPO_SCHEME_ALP && i < st->k_combinations;
					combos_out[rg * 10 + 2 * i]     = have ? st->combos[i][0] : -1;
					combos_out[rg * 10 + 2 * i + 1] = have ? st->combos[i][1] : -1;
				}
			}
		}
		const double* in = column + v * VECTOR_SIZE;
		scheme[v]        = (uint8_t)st->scheme;
		uint16_t cnt     = 0;
		if (st->scheme == ALPO_SCHEME_ALP) {
			uint8_t b  = 0;
			int64_t bs = 0;
			alpo_encode(in, exc + v * VECTOR_SIZE, pos + v * VECTOR_SIZE, &cnt, enc, st);
			alpo_analyze_ffor(enc, &b, &bs);
			memset(packed + v * VECTOR_SIZE, 0, 8192);
			alpo_ffor_u64((const uint64_t*)enc, (uint64_t*)(packed + v * VECTOR_SIZE), b, (uint64_t)bs);
			e[v] = st->exp, f[v] = st->fac, bw[v] = b, lbw[v] = 0, base[v] = bs, exc_cnt[v] = cnt;
		} else {
			alpo_rd_encode(in, rd_exc, pos + v * VECTOR_SIZE, &cnt, right, left, st);
			memset(packed + v * VECTOR_SIZE, 0, 8192);
			memset(packed_left + v * VECTOR_SIZE, 0, 2048);
			alpo_ffor_u64(right, (uint64_t*)(packed + v * VECTOR_SIZE), st->right_bit_width, 0);
			alpo_ffor_u16(left, packed_left + v * VECTOR_SIZE, st->left_bit_width, 0);
			memcpy(exc + v * VECTOR_SIZE, rd_exc, (size_t)cnt * 2);
			e[v] = 0, f[v] = 0, bw[v] = st->right_bit_width, lbw[v] = st->left_bit_width, base[v] = 0, exc_cnt[v] = cnt;
		}
	}
	free(st), free(sample), free(enc), free(right), free(left), free(rd_exc);
}

void alpo_decode_column(size_t n_vectors, const uint8_t* scheme, const uint8_t* e, const uint8_t* f, const uint8_t* bw,
                        const uint8_t* lbw, const int64_t* base, const uint16_t* exc_cnt, const int64_t* packed,
                        const uint16_t* packed_left, const double* exc, const uint16_t* pos, const uint16_t* dict,
                        double* out) {
	uint64_t*   right = (uint64_t*)malloc(VECTOR_SIZE * sizeof(uint64_t));
	uint16_t*   left  = (uint16_t*)malloc(VECTOR_SIZE * sizeof(uint16_t));
	alpo_state* st    = (alpo_state*)calloc(1, sizeof(alpo_state));
	for (size_t v = 0; v < n_vectors; v++) {
		double* o = out + v * VECTOR_SIZE;
		if (scheme[v] == ALPO_SCHEME_ALP) {
			alpo_falp((const uint64_t*)(packed + v * VECTOR_SIZE), o, bw[v], (uint64_t)base[v], f[v], e[v]);
			alpo_patch(o, exc + v * VECTOR_SIZE, pos + v * VECTOR_SIZE, exc_cnt[v]);
		} else {
			st->right_bit_width = bw[v];
			st->left_bit_width  = lbw[v];
			memcpy(st->left_parts_dict, dict + (v / N_VECTORS_PER_ROWGROUP) * 8, 16);
			alpo_unffor_u64((const uint64_t*)(packed + v * VECTOR_SIZE), right, bw[v], 0);
			alpo_unffor_u16(packed_left + v * VECTOR_SIZE, left, lbw[v], 0);
			alpo_rd_decode(o, right, left, (const uint16_t*)(exc + v * VECTOR_SIZE), pos + v * VECTOR_SIZE, exc_cnt[v], st);
		}
	}
	free(right), free(left), free(st);
}

/* ---- timing loop (bench.py cpu_baseline kind "port"); decode = falp + patch, single thread ------------ */
double alpo_time_falp_column(const int64_t* packed, size_t stride_words, const uint8_t* bw, const uint8_t* e,
                             const uint8_t* f, const int64_t* base, const uint16_t* exc_cnt, const double* exc,
                             const uint16_t* pos, size_t exc_stride, size_t n_vectors, double* out, int reps) {
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (int r = 0; r < reps; r++) {
		for (size_t v = 0; v < n_vectors; v++) {
			alpo_falp((const uint64_t*)(packed + v * stride_words), out + v * VECTOR_SIZE, bw[v], (uint64_t)base[v], f[v], e[v]);
			alpo_patch(out + v * VECTOR_SIZE, exc + v * exc_stride, pos + v * exc_stride, exc_cnt[v]);
		}
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
