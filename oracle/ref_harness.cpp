// oracle/ref_harness.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A thin extern "C" shim (our own code) that is compiled TOGETHER WITH the reference's own sources,
// in place, from /root/reference (see oracle/Makefile): /root/reference/src/*.cpp and
// -I/root/reference/include.  Nothing from the reference is copied into this repository; the build
// output goes to oracle/_ref/ (git-ignored).  The shim lets Python (ctypes) drive the REAL reference
// implementation so that
//   * the C restatement in oracle/alp_oracle.c can be pinned against it value-for-value,
//   * golden fixtures under tests/golden/ can be generated (tools/make_golden.py),
//   * bench.py can time the reference's CPU path on the GPU box's host cores ("cpu_baseline").
//
// Reference entry points wrapped (file:line in /root/reference):
//   alp::encoder<double>::init             include/alp/encoder.hpp:420
//   alp::encoder<double>::encode           include/alp/encoder.hpp:402
//   alp::encoder<double>::encode_simdized  include/alp/encoder.hpp:307
//   alp::encoder<double>::analyze_ffor     include/alp/encoder.hpp:109
//   alp::decoder<double>::decode           include/alp/decoder.hpp:134
//   alp::decoder<double>::patch_exceptions include/alp/decoder.hpp:141
//   alp::rd_encoder<double>::init/encode/decode   include/alp/rd.hpp:180 / :109 / :152
//   alp::sampler::first_level_sample       include/alp/sampler.hpp:14
//   ffor::ffor / unffor::unffor (u64,u16)  include/fastlanes/ffor.hpp:7-15, unffor.hpp:7-15
//   generated::falp::fallback::scalar::falp include/alp/falp.hpp:10
#include "alp.hpp"

#include <chrono>
#include <cstdint>
#include <cstring>
#include <vector>

using dstate = alp::state<double>;

extern "C" {

// ---- state handle ---------------------------------------------------------------------------
void* ref_state_new() { return new dstate(); }
void  ref_state_free(void* s) { delete static_cast<dstate*>(s); }

int ref_state_scheme(void* s) { return static_cast<int>(static_cast<dstate*>(s)->scheme); }
int ref_state_k(void* s) { return static_cast<dstate*>(s)->k_combinations; }
int ref_state_exp(void* s) { return static_cast<dstate*>(s)->exp; }
int ref_state_fac(void* s) { return static_cast<dstate*>(s)->fac; }
int ref_state_sampled_n(void* s) { return static_cast<int>(static_cast<dstate*>(s)->sampled_values_n); }
int ref_state_rbw(void* s) { return static_cast<dstate*>(s)->right_bit_width; }
int ref_state_lbw(void* s) { return static_cast<dstate*>(s)->left_bit_width; }
int ref_state_dict_size(void* s) { return static_cast<dstate*>(s)->actual_dictionary_size; }
void ref_state_dict(void* s, uint16_t* out8) {
	std::memcpy(out8, static_cast<dstate*>(s)->left_parts_dict, 8 * sizeof(uint16_t));
}
// combos_out[2*i] = exponent, combos_out[2*i+1] = factor; returns number of pairs
int ref_state_combos(void* s, int* combos_out, int cap) {
	auto& v = static_cast<dstate*>(s)->best_k_combinations;
	int   n = 0;
	for (auto& p : v) {
		if (n >= cap) break;
		combos_out[2 * n]     = p.first;
		combos_out[2 * n + 1] = p.second;
		++n;
	}
	return n;
}
// left-part -> index map as the reference builds it (rd.hpp:69-77); returns entry count
int ref_state_dict_map(void* s, uint16_t* keys, uint16_t* vals, int cap) {
	auto& m = static_cast<dstate*>(s)->left_parts_dict_map;
	int   n = 0;
	for (auto& kv : m) {
		if (n >= cap) break;
		keys[n] = kv.first;
		vals[n] = kv.second;
		++n;
	}
	return n;
}

// ---- rowgroup init ---------------------------------------------------------------------------
// encoder::init and, when it resolves to ALP_RD, rd_encoder::init — the sequence of
// test/test_alp_sample.cpp:137-141.  Returns the scheme (1 = ALP_RD, 2 = ALP).
int ref_init(void* s, const double* column, size_t column_offset, size_t tuples_count, double* sample_arr) {
	auto& stt = *static_cast<dstate*>(s);
	alp::encoder<double>::init(column, column_offset, tuples_count, sample_arr, stt);
	if (stt.scheme == alp::Scheme::ALP_RD) {
		alp::rd_encoder<double>::init(column, column_offset, tuples_count, sample_arr, stt);
	}
	return static_cast<int>(stt.scheme);
}

size_t ref_first_level_sample(const double* data, size_t off, size_t n, double* out) {
	return alp::sampler::first_level_sample<double>(data, off, n, out);
}

// ---- ALP vector primitives -------------------------------------------------------------------
void ref_encode(void* s, const double* in, double* exc, uint16_t* pos, uint16_t* cnt, int64_t* enc) {
	alp::encoder<double>::encode(in, exc, pos, cnt, enc, *static_cast<dstate*>(s));
}
void ref_encode_simdized(
    const double* in, double* exc, uint16_t* pos, uint16_t* cnt, int64_t* enc, uint8_t fac, uint8_t exp) {
	alp::encoder<double>::encode_simdized(in, exc, pos, cnt, enc, fac, exp);
}
void ref_analyze_ffor(const int64_t* in, uint8_t* bw, int64_t* base) {
	alp::bw_t b = 0;
	alp::encoder<double>::analyze_ffor(in, b, base);
	*bw = b;
}
void ref_ffor_i64(const int64_t* in, int64_t* out, uint8_t bw, const int64_t* base) { ffor::ffor(in, out, bw, base); }
void ref_unffor_i64(const int64_t* in, int64_t* out, uint8_t bw, const int64_t* base) {
	unffor::unffor(in, out, bw, base);
}
void ref_ffor_u64(const uint64_t* in, uint64_t* out, uint8_t bw, const uint64_t* base) {
	ffor::ffor(in, out, bw, base);
}
void ref_unffor_u64(const uint64_t* in, uint64_t* out, uint8_t bw, const uint64_t* base) {
	unffor::unffor(in, out, bw, base);
}
void ref_ffor_u16(const uint16_t* in, uint16_t* out, uint8_t bw, const uint16_t* base) {
	ffor::ffor(in, out, bw, base);
}
void ref_unffor_u16(const uint16_t* in, uint16_t* out, uint8_t bw, const uint16_t* base) {
	unffor::unffor(in, out, bw, base);
}
void ref_ffor_u8(const uint8_t* in, uint8_t* out, uint8_t bw, const uint8_t* base) { ffor::ffor(in, out, bw, base); }
void ref_unffor_u8(const uint8_t* in, uint8_t* out, uint8_t bw, const uint8_t* base) { unffor::unffor(in, out, bw, base); }
void ref_falp(const int64_t* in, double* out, uint8_t bw, const int64_t* base, uint8_t fac, uint8_t exp) {
	generated::falp::fallback::scalar::falp(in, out, bw, base, fac, exp);
}
void ref_decode(const int64_t* enc, uint8_t fac, uint8_t exp, double* out) {
	alp::decoder<double>::decode(enc, fac, exp, out);
}
void ref_patch(double* out, const double* exc, const uint16_t* pos, const uint16_t* cnt) {
	alp::decoder<double>::patch_exceptions(out, exc, pos, cnt);
}

// ---- ALP_RD vector primitives ----------------------------------------------------------------
void ref_rd_encode(
    void* s, const double* in, uint16_t* exc, uint16_t* pos, uint16_t* cnt, uint64_t* right, uint16_t* left) {
	alp::rd_encoder<double>::encode(in, exc, pos, cnt, right, left, *static_cast<dstate*>(s));
}
void ref_rd_decode(
    void* s, double* out, uint64_t* right, uint16_t* left, uint16_t* exc, uint16_t* pos, uint16_t* cnt) {
	alp::rd_encoder<double>::decode(out, right, left, exc, pos, cnt, *static_cast<dstate*>(s));
}

// ---- whole-column driver (our loop around the reference primitives) ---------------------------
// Semantics of the batch API this repo exposes: rowgroup r = vectors 100r..100r+99,
// init(column, r*102400, n_values) at each rowgroup start; per-rowgroup scheme; whole vectors only.
// Per-vector outputs (SoA): scheme/e/f/bw/base/exc_cnt, and for RD rbw/lbw/dict; packed streams are
// written at fixed 8 KiB (ALP, RD right) / 2 KiB (RD left) strides so the caller can compare bytes.
void ref_encode_column(const double* column,
                       size_t        n_vectors,
                       uint8_t*      scheme,    // [n_vectors]
                       uint8_t*      e,         // [n_vectors]
                       uint8_t*      f,         // [n_vectors]
                       uint8_t*      bw,        // [n_vectors]  (ALP: bit width; RD: right bit width)
                       uint8_t*      lbw,       // [n_vectors]  (RD only)
                       int64_t*      base,      // [n_vectors]
                       uint16_t*     exc_cnt,   // [n_vectors]
                       int64_t*      packed,    // [n_vectors*1024] (ALP ffor output / RD right ffor output)
                       uint16_t*     packed_left, // [n_vectors*1024] (RD left ffor output)
                       double*       exc,       // [n_vectors*1024] (ALP exceptions; RD: u16 left values in low bits)
                       uint16_t*     pos,       // [n_vectors*1024]
                       uint16_t*     dict,      // [n_rowgroups*8]
                       uint8_t*      dict_size, // [n_rowgroups]
                       uint8_t*      k_out,     // [n_rowgroups]
                       int*          combos_out // [n_rowgroups*10]
) {
	const size_t        n_values = n_vectors * 1024;
	dstate              stt;
	std::vector<double> sample(1024);
	std::vector<int64_t>  enc(1024);
	std::vector<uint64_t> right(1024);
	std::vector<uint16_t> left(1024);
	std::vector<uint16_t> rd_exc(1024);
	for (size_t v = 0; v < n_vectors; ++v) {
		const size_t rg = v / 100;
		if (v % 100 == 0) {
			stt = dstate();
			alp::encoder<double>::init(column, rg * 102400, n_values, sample.data(), stt);
			if (stt.scheme == alp::Scheme::ALP_RD) {
				alp::rd_encoder<double>::init(column, rg * 102400, n_values, sample.data(), stt);
			}
			if (dict) {
				std::memcpy(dict + rg * 8, stt.left_parts_dict, 16);
				dict_size[rg] = stt.scheme == alp::Scheme::ALP_RD ? stt.actual_dictionary_size : 0;
			}
			if (k_out) {
				k_out[rg] = stt.scheme == alp::Scheme::ALP ? static_cast<uint8_t>(stt.k_combinations) : 0;
				for (int i = 0; i < 5; ++i) {
					const bool have = stt.scheme == alp::Scheme::ALP && i < static_cast<int>(stt.best_k_combinations.size());
					combos_out[rg * 10 + 2 * i]     = have ? stt.best_k_combinations[i].first : -1;
					combos_out[rg * 10 + 2 * i + 1] = have ? stt.best_k_combinations[i].second : -1;
				}
			}
		}
		const double* in = column + v * 1024;
		scheme[v]        = static_cast<uint8_t>(stt.scheme);
		if (stt.scheme == alp::Scheme::ALP) {
			uint16_t cnt = 0;
			alp::encoder<double>::encode(in, exc + v * 1024, pos + v * 1024, &cnt, enc.data(), stt);
			alp::bw_t b = 0;
			int64_t   bs = 0;
			alp::encoder<double>::analyze_ffor(enc.data(), b, &bs);
			std::memset(packed + v * 1024, 0, 8192);
			ffor::ffor(enc.data(), packed + v * 1024, b, &bs);
			e[v] = stt.exp, f[v] = stt.fac, bw[v] = b, lbw[v] = 0, base[v] = bs, exc_cnt[v] = cnt;
		} else {
			uint16_t cnt = 0;
			alp::rd_encoder<double>::encode(in, rd_exc.data(), pos + v * 1024, &cnt, right.data(), left.data(), stt);
			std::memset(packed + v * 1024, 0, 8192);
			std::memset(packed_left + v * 1024, 0, 2048);
			ffor::ffor(right.data(), reinterpret_cast<uint64_t*>(packed + v * 1024), stt.right_bit_width, &stt.right_for_base);
			ffor::ffor(left.data(), packed_left + v * 1024, stt.left_bit_width, &stt.left_for_base);
			auto* exc16 = reinterpret_cast<uint16_t*>(exc + v * 1024);
			std::memcpy(exc16, rd_exc.data(), cnt * 2);
			e[v] = 0, f[v] = 0, bw[v] = stt.right_bit_width, lbw[v] = stt.left_bit_width, base[v] = 0, exc_cnt[v] = cnt;
		}
	}
}

// ---- timing loops for bench.py's cpu_baseline (single thread; wall seconds returned) -----------
// decode = falp + patch_exceptions per vector (test/test_alp_sample.cpp:169-170), inputs at fixed strides.
double ref_time_falp_column(const int64_t*  packed,  // [n*stride_words]
                            size_t          stride_words,
                            const uint8_t*  bw,
                            const uint8_t*  e,
                            const uint8_t*  f,
                            const int64_t*  base,
                            const uint16_t* exc_cnt,
                            const double*   exc,  // [n*exc_stride]
                            const uint16_t* pos,  // [n*exc_stride]
                            size_t          exc_stride,
                            size_t          n_vectors,
                            double*         out,  // [n*1024]
                            int             reps) {
	auto t0 = std::chrono::steady_clock::now();
	for (int r = 0; r < reps; ++r) {
		for (size_t v = 0; v < n_vectors; ++v) {
			generated::falp::fallback::scalar::falp(packed + v * stride_words, out + v * 1024, bw[v], base + v, f[v], e[v]);
			alp::decoder<double>::patch_exceptions(out + v * 1024, exc + v * exc_stride, pos + v * exc_stride, exc_cnt + v);
		}
	}
	auto t1 = std::chrono::steady_clock::now();
	return std::chrono::duration<double>(t1 - t0).count();
}

// encode = (init per rowgroup) + encode + analyze_ffor + ffor per vector (test_alp_sample.cpp:137,164-166)
double ref_time_encode_column(const double* column, size_t n_vectors, int64_t* packed_scratch /*[1024]*/, int reps, uint64_t* sum_bw) {
	const size_t         n_values = n_vectors * 1024;
	std::vector<double>  sample(1024), exc(1024);
	std::vector<int64_t> enc(1024);
	std::vector<uint16_t> pos(1024), left(1024), rd_exc(1024);
	std::vector<uint64_t> right(1024);
	uint64_t             acc = 0;
	auto                 t0  = std::chrono::steady_clock::now();
	for (int r = 0; r < reps; ++r) {
		dstate stt;
		for (size_t v = 0; v < n_vectors; ++v) {
			if (v % 100 == 0) {
				stt = dstate();
				alp::encoder<double>::init(column, (v / 100) * 102400, n_values, sample.data(), stt);
				if (stt.scheme == alp::Scheme::ALP_RD) {
					alp::rd_encoder<double>::init(column, (v / 100) * 102400, n_values, sample.data(), stt);
				}
			}
			const double* in = column + v * 1024;
			uint16_t      cnt = 0;
			if (stt.scheme == alp::Scheme::ALP) {
				alp::encoder<double>::encode(in, exc.data(), pos.data(), &cnt, enc.data(), stt);
				alp::bw_t b  = 0;
				int64_t   bs = 0;
				alp::encoder<double>::analyze_ffor(enc.data(), b, &bs);
				ffor::ffor(enc.data(), packed_scratch, b, &bs);
				acc += b;
			} else {
				alp::rd_encoder<double>::encode(in, rd_exc.data(), pos.data(), &cnt, right.data(), left.data(), stt);
				ffor::ffor(right.data(), reinterpret_cast<uint64_t*>(packed_scratch), stt.right_bit_width, &stt.right_for_base);
				ffor::ffor(left.data(), reinterpret_cast<uint16_t*>(exc.data()), stt.left_bit_width, &stt.left_for_base);
				acc += stt.right_bit_width + stt.left_bit_width;
			}
		}
	}
	auto t1 = std::chrono::steady_clock::now();
	if (sum_bw) *sum_bw = acc;
	return std::chrono::duration<double>(t1 - t0).count();
}

} // extern "C"

// =====================================================================================================================
// float (32-bit) path of the reference — same shim, second instantiation.
//   alp::encoder<float>::{init,encode,encode_simdized,analyze_ffor}, alp::decoder<float>::{decode,patch_exceptions},
//   alp::rd_encoder<float>::{init,encode,decode}, ffor/unffor (int32/uint32), falp (float): include/alp/falp.hpp:28-44
// =====================================================================================================================
using fstate = alp::state<float>;

extern "C" {

void reff_ffor_u32(const uint32_t* in, uint32_t* out, uint8_t bw, const uint32_t* base) { ffor::ffor(in, out, bw, base); }
void reff_unffor_u32(const uint32_t* in, uint32_t* out, uint8_t bw, const uint32_t* base) { unffor::unffor(in, out, bw, base); }
void reff_falp(const int32_t* in, float* out, uint8_t bw, const int32_t* base, uint8_t fac, uint8_t exp) {
	generated::falp::fallback::scalar::falp(in, out, bw, base, fac, exp);
}
void reff_unffor_decode(const int32_t* in, float* out, uint8_t bw, const int32_t* base, uint8_t fac, uint8_t exp) {
	int32_t tmp[1024];
	unffor::unffor(in, tmp, bw, base);
	alp::decoder<float>::decode(tmp, fac, exp, out);
}
void reff_encode_simdized(const float* in, float* exc, uint16_t* pos, uint16_t* cnt, int32_t* enc, uint8_t fac, uint8_t exp) {
	alp::encoder<float>::encode_simdized(in, exc, pos, cnt, enc, fac, exp);
}
void reff_analyze_ffor(const int32_t* in, uint8_t* bw, int32_t* base) {
	alp::bw_t b = 0;
	alp::encoder<float>::analyze_ffor(in, b, base);
	*bw = b;
}
float reff_decode_value(int32_t enc, uint8_t fac, uint8_t exp) { return alp::decoder<float>::decode_value(enc, fac, exp); }
int32_t reff_encode_value(float v, uint8_t fac, uint8_t exp) { return alp::encoder<float>::encode_value<true>(v, fac, exp); }

// whole-column driver, float: same contract as ref_encode_column with 32-bit words
// (packed: [n*1024] int32, base: int32 widened to int64, exc: [n*1024] float, RD exceptions as u16 in the low bytes of exc)
void reff_encode_column(const float* column, size_t n_vectors, uint8_t* scheme, uint8_t* e, uint8_t* f, uint8_t* bw, uint8_t* lbw,
                        int64_t* base, uint16_t* exc_cnt, int32_t* packed, uint16_t* packed_left, float* exc, uint16_t* pos,
                        uint16_t* dict, uint8_t* dict_size, uint8_t* k_out, int* combos_out) {
	const size_t          n_values = n_vectors * 1024;
	fstate                stt;
	std::vector<float>    sample(1024);
	std::vector<int32_t>  enc(1024);
	std::vector<uint32_t> right(1024);
	std::vector<uint16_t> left(1024);
	std::vector<uint16_t> rd_exc(1024);
	for (size_t v = 0; v < n_vectors; ++v) {
		const size_t rg = v / 100;
		if (v % 100 == 0) {
			stt = fstate();
			alp::encoder<float>::init(column, rg * 102400, n_values, sample.data(), stt);
			if (stt.scheme == alp::Scheme::ALP_RD) { alp::rd_encoder<float>::init(column, rg * 102400, n_values, sample.data(), stt); }
			std::memcpy(dict + rg * 8, stt.left_parts_dict, 16);
			dict_size[rg] = stt.scheme == alp::Scheme::ALP_RD ? stt.actual_dictionary_size : 0;
			k_out[rg]     = stt.scheme == alp::Scheme::ALP ? static_cast<uint8_t>(stt.k_combinations) : 0;
			for (int i = 0; i < 5; ++i) {
				const bool have = stt.scheme == alp::Scheme::ALP && i < static_cast<int>(stt.best_k_combinations.size());
				combos_out[rg * 10 + 2 * i]     = have ? stt.best_k_combinations[i].first : -1;
				combos_out[rg * 10 + 2 * i + 1] = have ? stt.best_k_combinations[i].second : -1;
			}
		}
		const float* in = column + v * 1024;
		scheme[v]       = static_cast<uint8_t>(stt.scheme);
		uint16_t cnt    = 0;
		if (stt.scheme == alp::Scheme::ALP) {
			alp::encoder<float>::encode(in, exc + v * 1024, pos + v * 1024, &cnt, enc.data(), stt);
			alp::bw_t b  = 0;
			int32_t   bs = 0;
			alp::encoder<float>::analyze_ffor(enc.data(), b, &bs);
			std::memset(packed + v * 1024, 0, 4096);
			ffor::ffor(enc.data(), packed + v * 1024, b, &bs);
			e[v] = stt.exp, f[v] = stt.fac, bw[v] = b, lbw[v] = 0, base[v] = bs, exc_cnt[v] = cnt;
		} else {
			alp::rd_encoder<float>::encode(in, rd_exc.data(), pos + v * 1024, &cnt, right.data(), left.data(), stt);
			std::memset(packed + v * 1024, 0, 4096);
			std::memset(packed_left + v * 1024, 0, 2048);
			ffor::ffor(right.data(), reinterpret_cast<uint32_t*>(packed + v * 1024), stt.right_bit_width, &stt.right_for_base);
			ffor::ffor(left.data(), packed_left + v * 1024, stt.left_bit_width, &stt.left_for_base);
			std::memcpy(exc + v * 1024, rd_exc.data(), cnt * 2);
			e[v] = 0, f[v] = 0, bw[v] = stt.right_bit_width, lbw[v] = stt.left_bit_width, base[v] = 0, exc_cnt[v] = cnt;
		}
	}
}

} // extern "C"
