// multi_test.cpp — alpgpu_compress_host_multi_* / alp::gpu::column<PT> over a device list (include/alp/batch.hpp): ONE host column cut into
// whole-rowgroup shards over several contexts, one host thread and one two-stream pipeline each, joined into one blob.  The reference's
// caller this replaces is a worker loop over a host column (publication/source_code/bench_end_to_end/src/benchmarks/alp/run_query.cpp:233-305;
// publication/source_code/bench_compression_ratio/alp.cpp:198-229).  Checked here on the contexts given on the command line (device indices;
// "0 0 0" = three contexts on one GPU): the multi-context blob equals the one-context blob byte for byte, decompression over the same
// contexts returns the input bits, for double and float columns with ALP and ALP_RD stretches, an incomplete last vector, fewer rowgroups
// than contexts, an empty column; a blob buffer that is too small is refused with the size to use.
//   g++ -std=c++17 -O1 -Iinclude tests/cpp/multi_test.cpp -Lalp_amd -lalpgpu -pthread && ./a.out 0 0
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "alp.hpp"
#include "alp/batch.hpp"

static int failures = 0;
#define EXPECT(cond, ...)                                                                                              \
	do {                                                                                                               \
		if (!(cond)) {                                                                                                 \
			++failures;                                                                                                \
			std::printf("FAIL %s:%d: ", __FILE__, __LINE__);                                                           \
			std::printf(__VA_ARGS__);                                                                                  \
			std::printf("\n");                                                                                         \
		}                                                                                                              \
	} while (0)

template <class PT>
std::vector<PT> make_column(size_t n_values, unsigned seed) {
	std::mt19937_64                        rng(seed);
	std::uniform_real_distribution<double> uni(-1e4, 1e4), unit(0.0, 1.0);
	std::vector<PT>                        v(n_values);
	for (size_t i = 0; i < n_values; ++i) {
		const size_t rg = i / (100 * 1024);
		if (rg % 7 == 3) {
			v[i] = static_cast<PT>(unit(rng)); // full-precision stretch: ALP_RD rowgroups
		} else {
			const double scale = (rg % 3 == 0) ? 10.0 : ((rg % 3 == 1) ? 100.0 : 1000.0);
			v[i]               = static_cast<PT>(std::round(uni(rng) * scale) / scale);
			if ((rng() & 255) == 0) { v[i] = static_cast<PT>(uni(rng) * 3.14159265358979); }
		}
	}
	return v;
}

template <class PT>
void run(const char* name, size_t n_values, const std::vector<int>& devices) {
	const std::vector<PT> col = make_column<PT>(n_values, static_cast<unsigned>(n_values % 1000 + 11));
	const auto t0   = std::chrono::steady_clock::now();
	const auto one  = alp::gpu::column<PT>::compress(col.data(), col.size(), std::vector<int> {devices[0]});
	const auto t1   = std::chrono::steady_clock::now();
	const auto many = alp::gpu::column<PT>::compress(col.data(), col.size(), devices);
	const auto t2   = std::chrono::steady_clock::now();
	EXPECT(one.size() == many.size(), "%s: blob sizes %zu vs %zu", name, one.size(), many.size());
	if (one.size() == many.size()) {
		size_t first = one.size();
		for (size_t i = 0; i < one.size(); ++i) {
			if (one[i] != many[i]) {
				first = i;
				break;
			}
		}
		EXPECT(first == one.size(), "%s: blobs differ at byte %zu of %zu", name, first, one.size());
	}
	const auto back = alp::gpu::column<PT>::decompress(many.data(), many.size(), devices);
	const auto t3   = std::chrono::steady_clock::now();
	EXPECT(back.size() == col.size(), "%s: %zu values back of %zu", name, back.size(), col.size());
	EXPECT(back.size() != col.size() || col.empty() || std::memcmp(back.data(), col.data(), col.size() * sizeof(PT)) == 0, "%s: decompressed values differ", name);
	const auto back1 = alp::gpu::column<PT>::decompress(one.data(), one.size(), std::vector<int> {devices[0]});
	EXPECT(back1.size() == col.size() && (col.empty() || std::memcmp(back1.data(), col.data(), col.size() * sizeof(PT)) == 0), "%s: one-context round trip", name);
	auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
	const double gb = col.size() * sizeof(PT) / 1e9;
	std::printf("%-28s %9zu values: blob %zu B (%.2f bits/value); compress 1 ctx %.1f ms, %zu ctx %.1f ms (%.1f GB/s); decompress %zu ctx %.1f ms (%.1f GB/s)\n", name,
	            col.size(), one.size(), col.empty() ? 0.0 : one.size() * 8.0 / col.size(), ms(t0, t1), devices.size(), ms(t1, t2), gb / (ms(t1, t2) / 1e3), devices.size(),
	            ms(t2, t3), gb / (ms(t2, t3) / 1e3));
}

int main(int argc, char** argv) {
	std::vector<int> devices;
	for (int i = 1; i < argc; ++i) { devices.push_back(std::atoi(argv[i])); }
	if (devices.size() < 2) { devices = {0, 0}; }
	run<double>("f64 warm-up", 300 * 1024, devices);
	run<double>("f64 3 chunks per shard", 80000ull * 1024 + 517, devices); // 0.66 GB: several pipeline chunks per shard, incomplete last vector
	run<double>("f64 one rowgroup", 60 * 1024, devices);                   // fewer rowgroups than contexts: the others get nothing
	run<double>("f64 one value", 1, devices);
	run<double>("f64 empty", 0, devices);
	run<float>("f32 2 chunks per shard", 50000ull * 1024 + 3, devices);
	run<float>("f32 rowgroups = contexts", devices.size() * 100 * 1024, devices);
	// a buffer that is too small: refused, with the size that works
	{
		const std::vector<double>      col  = make_column<double>(4000 * 1024, 5);
		const alp::gpu::context_lease   lease(devices);
		const std::vector<alpgpu_ctx*>& ctxs = lease.contexts();
		std::vector<uint8_t>           small(alpgpu_blob_size(4000, 0, 0) + 4096 * devices.size());
		uint64_t                       written = 0;
		const int rc = alpgpu_compress_host_multi_f64(ctxs.data(), static_cast<int>(ctxs.size()), col.data(), col.size(), small.data(), small.size(), &written);
		EXPECT(rc == ALPGPU_ERR_CAPACITY && written > small.size(), "too-small buffer: rc %d written %llu", rc, static_cast<unsigned long long>(written));
		std::vector<uint8_t> right(written);
		uint64_t             w2 = 0;
		EXPECT(alpgpu_compress_host_multi_f64(ctxs.data(), static_cast<int>(ctxs.size()), col.data(), col.size(), right.data(), right.size(), &w2) == ALPGPU_OK && w2 <= written,
		       "the size returned must work");
		alpgpu_ctx* twice[2] = {ctxs[0], ctxs[0]};
		EXPECT(alpgpu_compress_host_multi_f64(twice, 2, col.data(), col.size(), right.data(), right.size(), &w2) == ALPGPU_ERR_INVALID, "the same context twice is refused");
	}
	// two host threads over the SAME device list at the same time: the pool leases each of them its own contexts (a context serves one
	// pipeline at a time), both blobs are the one-context blob
	{
		const std::vector<double> col  = make_column<double>(30000ull * 1024 + 77, 9);
		const auto                want = alp::gpu::column<double>::compress(col.data(), col.size(), std::vector<int> {devices[0]});
		std::vector<uint8_t>      got[2];
		std::vector<double>       back[2];
		std::thread               th[2];
		for (int t = 0; t < 2; ++t) {
			th[t] = std::thread([&, t]() {
				got[t]  = alp::gpu::column<double>::compress(col.data(), col.size(), devices);
				back[t] = alp::gpu::column<double>::decompress(got[t].data(), got[t].size(), devices);
			});
		}
		for (auto& x : th) { x.join(); }
		for (int t = 0; t < 2; ++t) {
			EXPECT(got[t] == want, "concurrent compress, thread %d: blob differs from the one-context blob", t);
			EXPECT(back[t].size() == col.size() && std::memcmp(back[t].data(), col.data(), col.size() * 8) == 0, "concurrent round trip, thread %d", t);
		}
		// a lease held here keeps its contexts out of another lease's hands
		const alp::gpu::context_lease a(devices), b(devices);
		for (alpgpu_ctx* x : a.contexts()) {
			for (alpgpu_ctx* y : b.contexts()) { EXPECT(x != y, "two live leases share a context"); }
		}
	}
	std::printf("multi_test: %d failures\n", failures);
	return failures ? 1 : 0;
}
