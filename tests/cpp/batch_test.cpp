// tests/cpp/batch_test.cpp — alp::gpu::rowgroup<PT> (include/alp/batch.hpp) against the per-vector functions of the same
// header set, which tests/cpp/dropin_test.cpp checks against the reference's columns: one call per rowgroup must produce,
// byte for byte, what the reference-shaped loop (encoder::encode, analyze_ffor, ffor::ffor per vector; falp +
// patch_exceptions per vector; rd_encoder::encode / decode per vector) produces — and is timed next to it.
//
// usage: batch_test <n_rowgroups>     prints "... 0 failures" and the vectors/s of both shapes
#include "alp.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static uint64_t lcg(uint64_t& s) {
	s = s * 6364136223846793005ull + 1442695040888963407ull;
	return s >> 11;
}

template <class PT>
static std::vector<PT> make_column(size_t n_values, bool rd, uint64_t seed) {
	std::vector<PT> c(n_values);
	for (size_t i = 0; i < n_values; ++i) {
		const double u = static_cast<double>(lcg(seed)) / 9007199254740992.0; // [0, 1)
		if (rd) {
			c[i] = static_cast<PT>(u);
		} else {
			// two decimals below 1000 (double) / one decimal below 1000 (float: two would need more bits than its ALP cut-off allows)
			c[i] = sizeof(PT) == 8 ? static_cast<PT>(std::round(u * 100000.0) / 100.0) : static_cast<PT>(std::round(u * 10000.0) / 10.0);
			if (lcg(seed) % 97 == 0) { c[i] = static_cast<PT>(u * 3.14159265358979); } // ~1 % exceptions
		}
	}
	return c;
}

template <class PT>
static int run(size_t n_rowgroups, bool rd) {
	using ST = typename alp::inner_t<PT>::st;
	using UT = typename alp::inner_t<PT>::ut;
	constexpr size_t V = alp::config::VECTOR_SIZE, R = alp::config::N_VECTORS_PER_ROWGROUP;
	const size_t     n_vectors = n_rowgroups * R, n_values = n_vectors * V;
	const auto       column = make_column<PT>(n_values, rd, 12345 + (rd ? 1 : 0) + sizeof(PT));
	int              failures = 0;
	std::vector<PT>  sample(V);
	// per-vector outputs, whole column (stride 1024)
	std::vector<ST>       a_packed(n_values), b_packed(n_values), enc(V), bases_a(n_vectors), bases_b(n_vectors);
	std::vector<PT>       a_exc(n_values), b_exc(n_values), a_out(n_values), b_out(n_values);
	std::vector<uint16_t> a_pos(n_values), b_pos(n_values), a_cnt(n_vectors), b_cnt(n_vectors);
	std::vector<uint8_t>  a_bw(n_vectors), b_bw(n_vectors), a_fac(n_vectors), b_fac(n_vectors), a_exp(n_vectors), b_exp(n_vectors);
	std::vector<UT>       a_right(n_values), b_right(n_values);
	std::vector<uint16_t> a_left(n_values), b_left(n_values), a_rexc(n_values), b_rexc(n_values);
	std::vector<alp::state<PT>> states(n_rowgroups);
	for (size_t g = 0; g < n_rowgroups; ++g) {
		alp::state<PT>& stt = states[g];
		alp::encoder<PT>::init(column.data(), g * R * V, n_values, sample.data(), stt);
		if (stt.scheme == alp::Scheme::ALP_RD) { alp::rd_encoder<PT>::init(column.data(), g * R * V, n_values, sample.data(), stt); }
		if ((stt.scheme == alp::Scheme::ALP_RD) != rd) {
			std::printf("FAIL: rowgroup %zu resolved to scheme %d, column built for rd=%d\n", g, static_cast<int>(stt.scheme), rd ? 1 : 0);
			return 1;
		}
	}
	// ---- (A) the reference-shaped per-vector loop
	auto t0 = std::chrono::steady_clock::now();
	for (size_t v = 0; v < n_vectors; ++v) {
		alp::state<PT> stt = states[v / R];
		const PT*      in  = column.data() + v * V;
		if (!rd) {
			alp::encoder<PT>::encode(in, a_exc.data() + v * V, a_pos.data() + v * V, a_cnt.data() + v, enc.data(), stt);
			alp::bw_t bw = 0;
			alp::encoder<PT>::analyze_ffor(enc.data(), bw, bases_a.data() + v);
			ffor::ffor(enc.data(), a_packed.data() + v * V, bw, bases_a.data() + v);
			a_bw[v] = bw, a_fac[v] = stt.fac, a_exp[v] = stt.exp;
		} else {
			alp::rd_encoder<PT>::encode(in, a_rexc.data() + v * V, a_pos.data() + v * V, a_cnt.data() + v, a_right.data() + v * V, a_left.data() + v * V, stt);
		}
	}
	auto t1 = std::chrono::steady_clock::now();
	for (size_t v = 0; v < n_vectors; ++v) {
		alp::state<PT> stt = states[v / R];
		if (!rd) {
			generated::falp::fallback::scalar::falp(a_packed.data() + v * V, a_out.data() + v * V, a_bw[v], bases_a.data() + v, a_fac[v], a_exp[v]);
			alp::decoder<PT>::patch_exceptions(a_out.data() + v * V, a_exc.data() + v * V, a_pos.data() + v * V, a_cnt.data() + v);
		} else {
			alp::rd_encoder<PT>::decode(a_out.data() + v * V, a_right.data() + v * V, a_left.data() + v * V, a_rexc.data() + v * V, a_pos.data() + v * V, a_cnt.data() + v, stt);
		}
	}
	// ---- (B) one call per rowgroup; the first call of a process pays the scratch allocation and the lazy load of three
	// kernels, so rowgroup 0 is run once before the clock starts (the per-vector loop above has warmed its own kernels)
	{
		if (!rd) {
			alp::gpu::rowgroup<PT>::encode(column.data(), R, states[0], b_packed.data(), b_bw.data(), bases_b.data(), b_fac.data(), b_exp.data(), b_exc.data(), b_pos.data(),
			                               b_cnt.data());
			alp::gpu::rowgroup<PT>::decode(b_packed.data(), b_bw.data(), bases_b.data(), b_fac.data(), b_exp.data(), b_exc.data(), b_pos.data(), b_cnt.data(), R, b_out.data());
		} else {
			alp::gpu::rowgroup<PT>::encode_rd(column.data(), R, states[0], b_right.data(), b_left.data(), b_rexc.data(), b_pos.data(), b_cnt.data());
			alp::gpu::rowgroup<PT>::decode_rd(b_right.data(), b_left.data(), b_rexc.data(), b_pos.data(), b_cnt.data(), states[0], R, b_out.data());
		}
	}
	auto t2 = std::chrono::steady_clock::now();
	for (size_t g = 0; g < n_rowgroups; ++g) {
		const size_t o = g * R * V, ov = g * R;
		if (!rd) {
			alp::gpu::rowgroup<PT>::encode(column.data() + o, R, states[g], b_packed.data() + o, b_bw.data() + ov, bases_b.data() + ov, b_fac.data() + ov, b_exp.data() + ov,
			                               b_exc.data() + o, b_pos.data() + o, b_cnt.data() + ov);
		} else {
			alp::gpu::rowgroup<PT>::encode_rd(column.data() + o, R, states[g], b_right.data() + o, b_left.data() + o, b_rexc.data() + o, b_pos.data() + o, b_cnt.data() + ov);
		}
	}
	auto t3 = std::chrono::steady_clock::now();
	for (size_t g = 0; g < n_rowgroups; ++g) {
		const size_t o = g * R * V, ov = g * R;
		if (!rd) {
			alp::gpu::rowgroup<PT>::decode(b_packed.data() + o, b_bw.data() + ov, bases_b.data() + ov, b_fac.data() + ov, b_exp.data() + ov, b_exc.data() + o, b_pos.data() + o,
			                               b_cnt.data() + ov, R, b_out.data() + o);
		} else {
			alp::gpu::rowgroup<PT>::decode_rd(b_right.data() + o, b_left.data() + o, b_rexc.data() + o, b_pos.data() + o, b_cnt.data() + ov, states[g], R, b_out.data() + o);
		}
	}
	auto t4 = std::chrono::steady_clock::now();
	// ---- compare
	for (size_t v = 0; v < n_vectors; ++v) {
		const size_t o = v * V;
		bool         ok = a_cnt[v] == b_cnt[v] && std::memcmp(a_pos.data() + o, b_pos.data() + o, 2 * a_cnt[v]) == 0;
		if (!rd) {
			ok = ok && a_bw[v] == b_bw[v] && bases_a[v] == bases_b[v] && a_fac[v] == b_fac[v] && a_exp[v] == b_exp[v] &&
			     std::memcmp(a_packed.data() + o, b_packed.data() + o, 16 * a_bw[v] * sizeof(ST) * (sizeof(ST) == 8 ? 1 : 2)) == 0 &&
			     std::memcmp(a_exc.data() + o, b_exc.data() + o, sizeof(PT) * a_cnt[v]) == 0;
		} else {
			ok = ok && std::memcmp(a_right.data() + o, b_right.data() + o, V * sizeof(UT)) == 0 && std::memcmp(a_left.data() + o, b_left.data() + o, V * 2) == 0 &&
			     std::memcmp(a_rexc.data() + o, b_rexc.data() + o, 2 * a_cnt[v]) == 0;
		}
		ok = ok && std::memcmp(a_out.data() + o, column.data() + o, V * sizeof(PT)) == 0 && std::memcmp(b_out.data() + o, column.data() + o, V * sizeof(PT)) == 0;
		if (!ok) {
			if (failures < 5) { std::printf("FAIL %s %s vector %zu\n", sizeof(PT) == 8 ? "f64" : "f32", rd ? "rd" : "alp", v); }
			++failures;
		}
	}
	auto   sec = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
	std::printf("%s %-3s %zu vectors: per-vector encode %.0f vec/s decode %.0f vec/s | per-rowgroup encode %.0f vec/s decode %.0f vec/s | %d failures\n",
	            sizeof(PT) == 8 ? "f64" : "f32", rd ? "rd" : "alp", n_vectors, n_vectors / sec(t0, t1), n_vectors / sec(t1, t2), n_vectors / sec(t2, t3), n_vectors / sec(t3, t4), failures);
	return failures;
}

// alp::gpu::column<PT>: a whole host column to its serialized form and back (the column ends inside a vector on purpose)
template <class PT>
static int whole_column(size_t n_values) {
	std::vector<PT> col = make_column<PT>(n_values + 1024, false, 777 + sizeof(PT));
	col.resize(n_values);
	const auto t0   = std::chrono::steady_clock::now();
	const auto blob = alp::gpu::column<PT>::compress(col.data(), col.size());
	const auto t1   = std::chrono::steady_clock::now();
	const auto back = alp::gpu::column<PT>::decompress(blob.data(), blob.size());
	const auto t2   = std::chrono::steady_clock::now();
	const bool ok   = back.size() == col.size() && std::memcmp(back.data(), col.data(), col.size() * sizeof(PT)) == 0;
	auto       sec  = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
	std::printf("%s column of %zu values: %.2f bits/value, compress %.2f GB/s, decompress %.2f GB/s (pageable std::vector) | %s\n", sizeof(PT) == 8 ? "f64" : "f32",
	            n_values, 8.0 * static_cast<double>(blob.size()) / static_cast<double>(n_values), n_values * sizeof(PT) / sec(t0, t1) / 1e9,
	            n_values * sizeof(PT) / sec(t1, t2) / 1e9, ok ? "round trip ok" : "FAIL");
	return ok ? 0 : 1;
}

int main(int argc, char** argv) {
	const size_t n_rg = argc > 1 ? static_cast<size_t>(std::atoi(argv[1])) : 3;
	int          f    = 0;
	f += run<double>(n_rg, false);
	f += run<double>(n_rg, true);
	f += run<float>(n_rg, false);
	f += run<float>(n_rg, true);
	f += whole_column<double>(30000 * 1024 + 517); // three chunks of the host pipeline
	f += whole_column<float>(13000 * 1024 + 3);
	std::printf("batch_test: %d failures\n", f);
	return f ? 1 : 0;
}
