// constants_dump.cpp — prints alp::Constants<double / float>::{FRAC_ARR, EXP_ARR, FACT_ARR} of include/alp/constants.hpp as bit patterns, one per
// line ("<table> <index> <hex>"), and uses alp::encoder<PT>::encode_value<SAFE> on the values given as arguments ("ev64|ev32 <value> <fac> <exp>").
// tests/test_dropin_gpu.py compares the tables with what the DEVICE computes with its own tables, and encode_value with the oracle.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "alp.hpp"

template <class T, class U>
static void dump(const char* name, const T* arr, size_t n) {
	for (size_t i = 0; i < n; ++i) {
		U u;
		std::memcpy(&u, &arr[i], sizeof(U));
		std::printf("%s %zu %016" PRIx64 "\n", name, i, static_cast<uint64_t>(u));
	}
}

int main(int argc, char** argv) {
	using D = alp::Constants<double>;
	using F = alp::Constants<float>;
	dump<double, uint64_t>("frac64", D::FRAC_ARR, sizeof(D::FRAC_ARR) / sizeof(double));
	dump<double, uint64_t>("exp64", D::EXP_ARR, sizeof(D::EXP_ARR) / sizeof(double));
	dump<int64_t, uint64_t>("fact64", D::FACT_ARR, sizeof(D::FACT_ARR) / sizeof(int64_t));
	dump<float, uint32_t>("frac32", F::FRAC_ARR, sizeof(F::FRAC_ARR) / sizeof(float));
	dump<float, uint32_t>("exp32", F::EXP_ARR, sizeof(F::EXP_ARR) / sizeof(float));
	dump<int32_t, uint32_t>("fact32", F::FACT_ARR, sizeof(F::FACT_ARR) / sizeof(int32_t));
	for (int i = 1; i + 3 < argc; i += 4) {
		const int fac = std::atoi(argv[i + 2]), exp = std::atoi(argv[i + 3]);
		if (std::strcmp(argv[i], "ev64") == 0) {
			const double v = std::strtod(argv[i + 1], nullptr);
			std::printf("ev64 %" PRId64 " %" PRId64 "\n", alp::encoder<double>::encode_value<true>(v, fac, exp), alp::encoder<double>::encode_value<false>(v, fac, exp));
		} else {
			const float v = std::strtof(argv[i + 1], nullptr);
			std::printf("ev32 %d %d\n", alp::encoder<float>::encode_value<true>(v, fac, exp), alp::encoder<float>::encode_value<false>(v, fac, exp));
		}
	}
	return 0;
}
