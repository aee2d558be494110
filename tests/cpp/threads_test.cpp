// tests/cpp/threads_test.cpp — the per-vector drop-in API (include/alp.hpp) called from several host threads at once.
//
// The reference's vector functions are re-entrant (its one known race, static scratch arrays in encode_simdized, was fixed with thread_local:
// include/alp/encoder.hpp:314-319, issue #41), and its end-to-end benchmark decodes from morsel-driven worker threads
// (publication/source_code/bench_end_to_end/src/benchmarks/alp/run_query.cpp:233-305).  Here the functions share ONE process-wide device context
// and stream and keep their device scratch per host thread (include/alp/gpu_bridge.hpp).  Every thread runs the reference's own loop
// (test/test_alp_sample.cpp:97-187: init -> encode -> analyze_ffor -> ffor -> falp -> patch_exceptions, or the ALP_RD chain) over a column of
// its own; what it produces — state, encoded integers, widths, bases, packed words, exception lists, decoded values — must be, byte for byte,
// what the same loop produced alone.  Our own harness; synthetic columns made here.
//
//   g++ -std=c++17 -O1 -pthread -Iinclude tests/cpp/threads_test.cpp -Lalp_amd -lalpgpu && ./a.out [threads] [vectors per column] [rounds]
#include "alp.hpp"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

static uint64_t fnv(uint64_t h, const void* p, size_t n) {
	const unsigned char* b = static_cast<const unsigned char*>(p);
	for (size_t i = 0; i < n; ++i) { h = (h ^ b[i]) * 1099511628211ull; }
	return h;
}

template <class PT>
static std::vector<PT> make_column(int kind, size_t n_vectors, uint32_t seed) {
	std::mt19937_64                        rng(seed);
	std::uniform_real_distribution<double> uni(0.0, 1.0);
	std::vector<PT>                        c(n_vectors * 1024);
	for (size_t i = 0; i < c.size(); ++i) {
		const double u = uni(rng);
		double       v;
		switch (kind % 4) {
		case 0: v = std::round(u * 2.0e4 - 1.0e4) / 100.0; break;                                  // two decimals
		case 1: v = u; break;                                                                      // full precision: ALP_RD
		case 2: v = (i % 97 == 0) ? u * 3.14159265358979 : std::round(u * 1.0e3) / 10.0; break;     // one decimal + ~1 % exceptions
		default: v = (i % 1024 < 3) ? (i % 2 ? NAN : -0.0) : std::round(u * 5.0e5) / 1.0e4; break; // four decimals, specials at the head of every vector
		}
		c[i] = static_cast<PT>(v);
	}
	return c;
}

// the reference's loop over one column; returns a digest of everything it produced, -1 round-trip failures through `bad`
template <class PT>
static uint64_t run_column(const std::vector<PT>& column, int& bad) {
	using ST = typename alp::inner_t<PT>::st;
	using UT = typename alp::inner_t<PT>::ut;
	constexpr size_t      VEC_BYTES = sizeof(PT) * 1024;
	std::vector<PT>       input(1024), exceptions(1024), decoded(1024), sample(1024), glue(1024);
	std::vector<uint16_t> rd_exc(1024), pos(1024), exc_c(1024), left(1024), ffor_left(1024), unffor_left(1024);
	std::vector<ST>       ffor_buf(1024), base(1024), encoded(1024);
	std::vector<UT>       right(1024), ffor_right(1024), unffor_right(1024);
	const size_t          n_values = column.size(), n_vectors = n_values / 1024;
	uint64_t              h = 1469598103934665603ull;
	alp::state<PT>        stt;
	for (size_t v = 0; v < n_vectors; ++v) {
		const size_t offset = v * 1024;
		if (v % alp::config::N_VECTORS_PER_ROWGROUP == 0) {
			stt = alp::state<PT>();
			alp::encoder<PT>::init(column.data(), offset, n_values, sample.data(), stt);
			if (stt.scheme == alp::Scheme::ALP_RD) { alp::rd_encoder<PT>::init(column.data(), offset, n_values, sample.data(), stt); }
			const int scheme = static_cast<int>(stt.scheme);
			h = fnv(h, &scheme, sizeof(scheme));
			h = fnv(h, &stt.k_combinations, sizeof(stt.k_combinations));
		}
		std::memcpy(input.data(), column.data() + offset, VEC_BYTES);
		const PT* out = nullptr;
		if (stt.scheme == alp::Scheme::ALP_RD) {
			alp::rd_encoder<PT>::encode(input.data(), rd_exc.data(), pos.data(), exc_c.data(), right.data(), left.data(), stt);
			ffor::ffor(right.data(), ffor_right.data(), stt.right_bit_width, &stt.right_for_base);
			ffor::ffor(left.data(), ffor_left.data(), stt.left_bit_width, &stt.left_for_base);
			unffor::unffor(ffor_right.data(), unffor_right.data(), stt.right_bit_width, &stt.right_for_base);
			unffor::unffor(ffor_left.data(), unffor_left.data(), stt.left_bit_width, &stt.left_for_base);
			alp::rd_encoder<PT>::decode(glue.data(), unffor_right.data(), unffor_left.data(), rd_exc.data(), pos.data(), exc_c.data(), stt);
			out = glue.data();
			h   = fnv(h, &stt.right_bit_width, 1), h = fnv(h, &stt.left_bit_width, 1);
			h   = fnv(h, ffor_right.data(), static_cast<size_t>(stt.right_bit_width) * 128);
			h   = fnv(h, ffor_left.data(), static_cast<size_t>(stt.left_bit_width) * 128);
			h   = fnv(h, &exc_c[0], 2), h = fnv(h, rd_exc.data(), 2u * exc_c[0]), h = fnv(h, pos.data(), 2u * exc_c[0]);
		} else {
			alp::bw_t bit_width = 0;
			alp::encoder<PT>::encode(input.data(), exceptions.data(), pos.data(), exc_c.data(), encoded.data(), stt);
			alp::encoder<PT>::analyze_ffor(encoded.data(), bit_width, base.data());
			ffor::ffor(encoded.data(), ffor_buf.data(), bit_width, base.data());
			generated::falp::fallback::scalar::falp(ffor_buf.data(), decoded.data(), bit_width, base.data(), stt.fac, stt.exp);
			alp::decoder<PT>::patch_exceptions(decoded.data(), exceptions.data(), pos.data(), exc_c.data());
			out = decoded.data();
			h   = fnv(h, &stt.fac, 1), h = fnv(h, &stt.exp, 1), h = fnv(h, &bit_width, 1), h = fnv(h, &base[0], sizeof(ST));
			h   = fnv(h, encoded.data(), 1024 * sizeof(ST));
			h   = fnv(h, ffor_buf.data(), static_cast<size_t>(bit_width) * 128); // 1024 values x bit_width bits
			h   = fnv(h, &exc_c[0], 2), h = fnv(h, exceptions.data(), sizeof(PT) * exc_c[0]), h = fnv(h, pos.data(), 2u * exc_c[0]);
		}
		h = fnv(h, out, VEC_BYTES);
		if (std::memcmp(out, input.data(), VEC_BYTES) != 0) { // bit patterns: NaN payloads and -0.0 survive the codec
			++bad;
		}
	}
	return h;
}

int main(int argc, char** argv) {
	const int    n_threads = argc > 1 ? std::atoi(argv[1]) : 6;
	const size_t n_vectors = argc > 2 ? static_cast<size_t>(std::atoi(argv[2])) : 130; // one full rowgroup + a partial one
	const int    rounds    = argc > 3 ? std::atoi(argv[3]) : 2;
	int          failures  = 0;
	// columns: thread t gets kind t (double for even t, float for odd t)
	std::vector<std::vector<double>> cd(n_threads);
	std::vector<std::vector<float>>  cf(n_threads);
	std::vector<uint64_t>            alone(n_threads);
	for (int t = 0; t < n_threads; ++t) {
		int bad = 0;
		if (t % 2 == 0) {
			cd[t]    = make_column<double>(t / 2, n_vectors, 100 + t);
			alone[t] = run_column<double>(cd[t], bad);
		} else {
			cf[t]    = make_column<float>(t / 2, n_vectors, 100 + t);
			alone[t] = run_column<float>(cf[t], bad);
		}
		if (bad) { std::printf("FAIL column %d alone: %d vectors do not round-trip\n", t, bad), ++failures; }
	}
	for (int r = 0; r < rounds; ++r) {
		std::vector<uint64_t>    got(n_threads, 0);
		std::vector<int>         bad(n_threads, 0);
		std::vector<std::string> err(n_threads);
		std::atomic<int>         ready {0};
		std::vector<std::thread> th;
		for (int t = 0; t < n_threads; ++t) {
			th.emplace_back([&, t]() {
				ready.fetch_add(1);
				while (ready.load() < n_threads) {} // all threads enter the API together
				try {
					got[t] = (t % 2 == 0) ? run_column<double>(cd[t], bad[t]) : run_column<float>(cf[t], bad[t]);
				} catch (const std::exception& e) { err[t] = e.what(); }
			});
		}
		for (auto& x : th) { x.join(); }
		for (int t = 0; t < n_threads; ++t) {
			if (!err[t].empty()) { std::printf("FAIL round %d thread %d: %s\n", r, t, err[t].c_str()), ++failures; }
			if (bad[t]) { std::printf("FAIL round %d thread %d: %d vectors do not round-trip\n", r, t, bad[t]), ++failures; }
			if (got[t] != alone[t]) { std::printf("FAIL round %d thread %d: digest %016llx, alone %016llx\n", r, t, (unsigned long long)got[t], (unsigned long long)alone[t]), ++failures; }
		}
	}
	std::printf("threads_test: %d threads x %zu vectors x %d rounds, %d failures\n", n_threads, n_vectors, rounds, failures);
	return failures ? 1 : 0;
}
