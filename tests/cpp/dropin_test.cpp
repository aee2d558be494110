// tests/cpp/dropin_test.cpp — drives include/alp.hpp (the source-compatible drop-in) the way the reference's own unit
// test drives the reference (test/test_alp_sample.cpp:97-187 `test_column`): init -> scheme switch -> encode -> pack ->
// decode -> bit-pattern-aware equality, plus the (exceptions_count, bit_width) known answers.  Our own harness code; the
// column data and expectations come from tests/golden (exported by tests/test_dropin_gpu.py as raw files).
//
// usage: dropin_test <dir>   where <dir>/columns.txt lists: type name n_values expected_bw expected_exc expect_rd
//        (type = f64 | f32) and <dir>/<name>.<type> holds the raw values.  Both precisions run through the same
//        template, like the reference's test_column<PT>.
#include "alp.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <string>
#include <vector>

template <class PT>
static bool same_value(PT original, PT decoded) {
	if (std::isnan(original)) { return std::isnan(decoded); }
	return std::memcmp(&original, &decoded, sizeof(PT)) == 0; // covers -0.0 vs 0.0
}

template <class PT>
struct Buffers {
	using ST = typename alp::inner_t<PT>::st;
	using UT = typename alp::inner_t<PT>::ut;
	std::vector<PT>       input, exceptions, decoded, sample, glue;
	std::vector<uint16_t> rd_exc, pos, exc_c, left, ffor_left, unffor_left;
	std::vector<ST>       ffor_buf, base, encoded;
	std::vector<UT>       right, ffor_right, unffor_right;
	Buffers()
	    : input(1024), exceptions(1024), decoded(1024), sample(1024), glue(1024), rd_exc(1024), pos(1024), exc_c(1024), left(1024),
	      ffor_left(1024), unffor_left(1024), ffor_buf(1024), base(1024), encoded(1024), right(1024), ffor_right(1024), unffor_right(1024) {}
};

template <class PT>
static int test_column(const std::string& dir, const std::string& name, size_t n_values, int want_bw, int want_exc, int want_rd) {
	using ST                 = typename alp::inner_t<PT>::st;
	constexpr size_t VEC_BYTES = sizeof(PT) * 1024;
	Buffers<PT>      B;
	std::vector<PT>  column(n_values);
	std::ifstream    f(dir + "/" + name + (sizeof(PT) == 8 ? ".f64" : ".f32"), std::ios::binary);
	f.read(reinterpret_cast<char*>(column.data()), static_cast<std::streamsize>(n_values * sizeof(PT)));
	if (!f) {
		std::printf("FAIL %s: cannot read input\n", name.c_str());
		return 1;
	}
	int failures = 0;
	const size_t n_vectors = n_values / alp::config::VECTOR_SIZE;

	alp::state<PT> stt;
	for (size_t v = 0; v < n_vectors; ++v) {
		const size_t offset = v * alp::config::VECTOR_SIZE;
		if (v % alp::config::N_VECTORS_PER_ROWGROUP == 0) {
			stt = alp::state<PT>();
			alp::encoder<PT>::init(column.data(), offset, n_values, B.sample.data(), stt);
			if (stt.scheme == alp::Scheme::ALP_RD) {
				alp::rd_encoder<PT>::init(column.data(), offset, n_values, B.sample.data(), stt);
				// the public statics find_best_dictionary is made of (rd.hpp:23-104): the 16 cuts one by one give the cut it chose (first strictly
				// smaller estimate), persisting that cut gives its dictionary, and an estimate is estimate_compression_size of what it counted
				alp::state<PT> probe = stt;
				double         best  = 1e300;
				alp::bw_t      chosen = 0;
				for (size_t i = 1; i <= alp::config::CUTTING_LIMIT; ++i) {
					const alp::bw_t rbw = static_cast<alp::bw_t>(sizeof(PT) * 8 - i);
					const double    est = alp::rd_encoder<PT>::template build_left_parts_dictionary<false>(B.sample.data(), rbw, probe);
					if (est < rbw + 1) { std::printf("FAIL %s: estimate %.3f below its own bit widths\n", name.c_str(), est), ++failures; }
					if (est < best) { best = est, chosen = rbw; }
				}
				if (chosen != stt.right_bit_width) { std::printf("FAIL %s: cut-by-cut search picks rbw %d, init picked %d\n", name.c_str(), chosen, stt.right_bit_width), ++failures; }
				const double again = alp::rd_encoder<PT>::template build_left_parts_dictionary<true>(B.sample.data(), chosen, probe);
				if (again != best || probe.left_bit_width != stt.left_bit_width || probe.actual_dictionary_size != stt.actual_dictionary_size ||
				    std::memcmp(probe.left_parts_dict, stt.left_parts_dict, sizeof(stt.left_parts_dict)) != 0) {
					std::printf("FAIL %s: build_left_parts_dictionary<true> of the chosen cut differs from find_best_dictionary\n", name.c_str()), ++failures;
				}
				const double lo = alp::rd_encoder<PT>::estimate_compression_size(chosen, stt.left_bit_width, 0, stt.sampled_values_n);
				if (again < lo || alp::rd_encoder<PT>::estimate_compression_size(10, 3, 4, 64) != 13.0 + 4 * 32 / 64.0) {
					std::printf("FAIL %s: estimate_compression_size\n", name.c_str()), ++failures;
				}
			}
			if (v == 0 && want_rd >= 0 && (stt.scheme == alp::Scheme::ALP_RD) != (want_rd == 1)) {
				std::printf("FAIL %s: scheme %d, expected rd=%d\n", name.c_str(), static_cast<int>(stt.scheme), want_rd);
				++failures;
			}
		}
		std::memcpy(B.input.data(), column.data() + offset, VEC_BYTES);
		const PT* out = nullptr;
		if (stt.scheme == alp::Scheme::ALP_RD) {
			alp::rd_encoder<PT>::encode(B.input.data(), B.rd_exc.data(), B.pos.data(), B.exc_c.data(), B.right.data(), B.left.data(), stt);
			ffor::ffor(B.right.data(), B.ffor_right.data(), stt.right_bit_width, &stt.right_for_base);
			ffor::ffor(B.left.data(), B.ffor_left.data(), stt.left_bit_width, &stt.left_for_base);
			unffor::unffor(B.ffor_right.data(), B.unffor_right.data(), stt.right_bit_width, &stt.right_for_base);
			unffor::unffor(B.ffor_left.data(), B.unffor_left.data(), stt.left_bit_width, &stt.left_for_base);
			alp::rd_encoder<PT>::decode(B.glue.data(), B.unffor_right.data(), B.unffor_left.data(), B.rd_exc.data(), B.pos.data(), B.exc_c.data(), stt);
			out = B.glue.data();
		} else {
			alp::bw_t bit_width = 0;
			alp::encoder<PT>::encode(B.input.data(), B.exceptions.data(), B.pos.data(), B.exc_c.data(), B.encoded.data(), stt);
			if (v % 7 == 0) { // second-level sampling called on its own (encoder.hpp:241-305) chooses what encode() chose
				uint8_t fac = 255, exp = 255;
				alp::encoder<PT>::find_best_exponent_factor_from_combinations(stt.best_k_combinations, static_cast<uint8_t>(stt.k_combinations), B.input.data(),
				                                                              alp::config::VECTOR_SIZE, fac, exp);
				if (fac != stt.fac || exp != stt.exp) { std::printf("FAIL %s v%zu: second-level sampling alone gives (%d,%d), encode() (%d,%d)\n", name.c_str(), v, exp, fac, stt.exp, stt.fac), ++failures; }
			}
			alp::encoder<PT>::analyze_ffor(B.encoded.data(), bit_width, B.base.data());
			ffor::ffor(B.encoded.data(), B.ffor_buf.data(), bit_width, B.base.data());
			generated::falp::fallback::scalar::falp(B.ffor_buf.data(), B.decoded.data(), bit_width, B.base.data(), stt.fac, stt.exp);
			alp::decoder<PT>::patch_exceptions(B.decoded.data(), B.exceptions.data(), B.pos.data(), B.exc_c.data());
			out = B.decoded.data();
			// the unfused path must agree with the fused one (benchmarks/benchmark.cpp:129-131)
			std::vector<ST> unpacked(1024);
			std::vector<PT> dec2(1024);
			unffor::unffor(B.ffor_buf.data(), unpacked.data(), bit_width, B.base.data());
			alp::decoder<PT>::decode(unpacked.data(), stt.fac, stt.exp, dec2.data());
			alp::decoder<PT>::patch_exceptions(dec2.data(), B.exceptions.data(), B.pos.data(), B.exc_c.data());
			if (std::memcmp(dec2.data(), out, VEC_BYTES) != 0) {
				std::printf("FAIL %s v%zu: unffor+decode differs from falp\n", name.c_str(), v);
				++failures;
			}
			if (v == 0 && want_bw >= 0 && (bit_width != want_bw || B.exc_c[0] != want_exc)) {
				std::printf("FAIL %s: bit_width %d exceptions %d, expected %d %d\n", name.c_str(), bit_width, B.exc_c[0], want_bw, want_exc);
				++failures;
			}
		}
		for (size_t i = 0; i < alp::config::VECTOR_SIZE; ++i) {
			if (!same_value(B.input[i], out[i])) {
				std::printf("FAIL %s v%zu: value %zu differs after the round trip\n", name.c_str(), v, i);
				++failures;
				break;
			}
		}
	}
	if (!failures) { std::printf("ok   %s (%zu vectors, scheme %s)\n", name.c_str(), n_vectors, stt.scheme == alp::Scheme::ALP ? "ALP" : "ALP_RD"); }
	return failures;
}

// 8-bit lanes of ffor / unffor (not used by the codec; part of the reference's API): pack, unpack, compare
static int test_u8_lanes() {
	int failures = 0;
	for (int bw = 0; bw <= 8; ++bw) {
		std::vector<uint8_t> in(1024), packed(1024, 0), out(1024, 0xAA);
		const uint8_t        base = static_cast<uint8_t>(17 * bw + 3);
		for (int i = 0; i < 1024; ++i) { in[i] = static_cast<uint8_t>(base + ((i * 37 + 11) & ((1 << bw) - 1))); }
		ffor::ffor(in.data(), packed.data(), static_cast<uint8_t>(bw), &base);
		unffor::unffor(packed.data(), out.data(), static_cast<uint8_t>(bw), &base);
		if (in != out) {
			std::printf("FAIL u8 lanes bw=%d\n", bw);
			++failures;
		}
	}
	if (!failures) { std::printf("ok   u8 ffor/unffor bw 0..8\n"); }
	return failures;
}

// the small public helpers of alp::encoder / alp::decoder (encoder.hpp:75-106, decoder.hpp:128-131): known answers from their definitions
template <class PT>
static int test_helpers(const char* name) {
	using E  = alp::encoder<PT>;
	using ST = typename alp::inner_t<PT>::st;
	int failures = 0;
	auto expect = [&](bool ok, const char* what) {
		if (!ok) { std::printf("FAIL helpers<%s>: %s\n", name, what), ++failures; }
	};
	expect(E::template count_bits<uint64_t>(0) == 0 && E::template count_bits<uint64_t>(1) == 1 && E::template count_bits<uint64_t>(255) == 8 && E::template count_bits<uint64_t>(256) == 9, "count_bits(x)");
	expect(E::template count_bits<uint64_t>(1ull << 63) == 64 && E::template count_bits<uint32_t>(0x80000000u) == 32, "count_bits of the top bit");
	expect(E::template count_bits<ST>(ST(5), ST(-3)) == 4 && E::template count_bits<ST>(ST(7), ST(7)) == 0 && E::template count_bits<ST>(ST(1023), ST(0)) == 10, "count_bits(max, min)");
	expect(E::template count_bits<ST>(std::numeric_limits<ST>::max(), std::numeric_limits<ST>::min()) == sizeof(ST) * 8, "count_bits over the whole range wraps like the unsigned type");
	expect(E::is_impossible_to_encode(std::numeric_limits<PT>::quiet_NaN()) && E::is_impossible_to_encode(std::numeric_limits<PT>::infinity()) &&
	           E::is_impossible_to_encode(-std::numeric_limits<PT>::infinity()) && E::is_impossible_to_encode(PT(-0.0)) && E::is_impossible_to_encode(PT(1e19)) &&
	           E::is_impossible_to_encode(PT(-1e19)),
	       "is_impossible_to_encode of NaN, infinities, -0.0 and values beyond the int64 limits");
	expect(!E::is_impossible_to_encode(PT(0.0)) && !E::is_impossible_to_encode(PT(1.5)) && !E::is_impossible_to_encode(PT(-123456.0)) && !E::is_impossible_to_encode(PT(9.0e18)), "is_impossible_to_encode of ordinary values");
	// decode_value = (PT)(integer) * FACT[f] * FRAC[e], one value through the vector kernel
	const PT dv = alp::decoder<PT>::decode_value(ST(12345), 2, 3);
	const PT want = static_cast<PT>(static_cast<ST>(12345) * static_cast<ST>(100)) * alp::Constants<PT>::FRAC_ARR[3];
	expect(std::memcmp(&dv, &want, sizeof(PT)) == 0, "decode_value(12345, f = 2, e = 3)");
	expect(E::template encode_value<true>(PT(12.5), 0, 1) == ST(125) && E::template encode_value<false>(PT(-7.25), 0, 2) == ST(-725), "encode_value of exact decimals");
	if (!failures) { std::printf("ok   helpers<%s>\n", name); }
	return failures;
}

int main(int argc, char** argv) {
	if (argc < 2) { return 2; }
	const std::string dir = argv[1];
	std::ifstream     list(dir + "/columns.txt");
	std::string       type, name;
	size_t            n_values;
	int               bw, exc, rd, failures = 0, n = 0;
	while (list >> type >> name >> n_values >> bw >> exc >> rd) {
		failures += type == "f32" ? test_column<float>(dir, name, n_values, bw, exc, rd) : test_column<double>(dir, name, n_values, bw, exc, rd);
		++n;
	}
	failures += test_u8_lanes();
	failures += test_helpers<double>("double") + test_helpers<float>("float");
	std::printf("%d columns, %d failures\n", n, failures);
	return failures ? 1 : 0;
}
