/* tests/c/abi_roundtrip.c — the C ABI of include/alpgpu.h used from plain C (C11, no C++ and no HIP headers on the caller's side):
 * allocate a column with alpgpu_malloc, encode + decode doubles and floats that live in HBM, read totals, serialise to a blob and
 * back.  Built and run by tests/test_abi_c_gpu.py; prints "ok" lines and returns the number of failures. */
#include "alpgpu.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(call)                                                                                                     \
	do {                                                                                                                \
		int rc_ = (call);                                                                                               \
		if (rc_ != ALPGPU_OK) {                                                                                         \
			printf("FAIL %s -> %d: %s\n", #call, rc_, alpgpu_last_error());                                             \
			return 1;                                                                                                   \
		}                                                                                                               \
	} while (0)

static int make_column(alpgpu_ctx* ctx, uint64_t n, int is_float, alpgpu_column* col) {
	memset(col, 0, sizeof(*col));
	col->n_vectors   = n;
	col->n_rowgroups = (n + 99) / 100;
	CHECK(alpgpu_malloc(ctx, (void**)&col->d_rowgroups, col->n_rowgroups * sizeof(alpgpu_rowgroup_state)));
	CHECK(alpgpu_malloc(ctx, (void**)&col->d_vectors, n * sizeof(alpgpu_vector_desc)));
	col->packed_capacity = is_float ? alpgpu_packed_capacity_f32(n) : alpgpu_packed_capacity(n);
	col->exc_capacity    = is_float ? alpgpu_exc_capacity_f32(n) : alpgpu_exc_capacity(n);
	CHECK(alpgpu_malloc(ctx, (void**)&col->d_packed, col->packed_capacity));
	CHECK(alpgpu_malloc(ctx, (void**)&col->d_exc, col->exc_capacity));
	CHECK(alpgpu_malloc(ctx, (void**)&col->d_totals, 64));
	CHECK(alpgpu_malloc(ctx, (void**)&col->d_rd_order, col->n_rowgroups * ALPGPU_RD_ORDER_STRIDE * sizeof(uint16_t)));
	return 0;
}

static void free_column(alpgpu_ctx* ctx, alpgpu_column* col) {
	alpgpu_free(ctx, col->d_rowgroups);
	alpgpu_free(ctx, col->d_vectors);
	alpgpu_free(ctx, col->d_packed);
	alpgpu_free(ctx, col->d_exc);
	alpgpu_free(ctx, col->d_totals);
	alpgpu_free(ctx, col->d_rd_order);
}

static int roundtrip_f64(alpgpu_ctx* ctx) {
	const uint64_t n      = 250; /* 2.5 rowgroups */
	const size_t   values = (size_t)n * 1024;
	double*        h_in   = (double*)malloc(values * 8);
	double*        h_out  = (double*)malloc(values * 8);
	uint64_t       seed   = 42;
	for (size_t i = 0; i < values; ++i) {
		seed = seed * 6364136223846793005ull + 1442695040888963407ull;
		const double u = (double)(seed >> 11) / 9007199254740992.0;
		if (i >= 150 * 1024 && i < 200 * 1024) {
			h_in[i] = u; /* full-precision values: an ALP_RD stretch */
		} else {
			h_in[i] = round(u * 1e7) / 100.0;
			if ((seed & 1023) == 0) { h_in[i] = -0.0; }
			if ((seed & 4095) == 1) { h_in[i] = NAN; }
		}
	}
	double *d_in, *d_out;
	CHECK(alpgpu_malloc(ctx, (void**)&d_in, values * 8));
	CHECK(alpgpu_malloc(ctx, (void**)&d_out, values * 8));
	CHECK(alpgpu_memcpy_h2d(ctx, d_in, h_in, values * 8));
	alpgpu_column col;
	if (make_column(ctx, n, 0, &col)) { return 1; }
	CHECK(alpgpu_encode_f64(ctx, d_in, n, &col));
	CHECK(alpgpu_decode_f64(ctx, &col, d_out));
	uint64_t pb, eb;
	int      ov;
	CHECK(alpgpu_column_totals(ctx, &col, &pb, &eb, &ov));
	CHECK(alpgpu_memcpy_d2h(ctx, h_out, d_out, values * 8));
	if (memcmp(h_in, h_out, values * 8) != 0) {
		printf("FAIL f64: decode(encode(x)) differs from x\n");
		return 1;
	}
	/* blob round trip into a second column */
	const uint64_t size = alpgpu_blob_size(n, pb, eb);
	void*          blob = malloc(size);
	uint64_t       written = 0, n_values = 0;
	CHECK(alpgpu_column_to_blob(ctx, &col, values, blob, size, &written));
	alpgpu_column col2;
	if (make_column(ctx, n, 0, &col2)) { return 1; }
	CHECK(alpgpu_column_from_blob(ctx, blob, written, &col2, &n_values));
	CHECK(alpgpu_memset(ctx, d_out, 0, values * 8));
	CHECK(alpgpu_decode_f64(ctx, &col2, d_out));
	CHECK(alpgpu_memcpy_d2h(ctx, h_out, d_out, values * 8));
	if (n_values != values || memcmp(h_in, h_out, values * 8) != 0) {
		printf("FAIL f64: blob round trip\n");
		return 1;
	}
	printf("ok   f64: %llu vectors, %.2f bits/value, blob %llu bytes\n", (unsigned long long)n, (double)(pb + eb + 32 * n) * 8.0 / (double)values,
	       (unsigned long long)written);
	/* the host-to-host entry points: page-locked buffers, the same blob byte for byte, and back */
	{
		double*  p_in = NULL;
		double*  p_out = NULL;
		void*    p_blob = NULL;
		uint64_t w2 = 0, nv2 = 0;
		CHECK(alpgpu_malloc_host(ctx, (void**)&p_in, values * 8));
		CHECK(alpgpu_malloc_host(ctx, (void**)&p_out, values * 8));
		CHECK(alpgpu_malloc_host(ctx, &p_blob, size));
		memcpy(p_in, h_in, values * 8);
		CHECK(alpgpu_compress_host_f64(ctx, p_in, values, p_blob, size, &w2));
		if (w2 != written || memcmp(p_blob, blob, written) != 0) {
			printf("FAIL f64: alpgpu_compress_host_f64 does not reproduce the column's blob\n");
			return 1;
		}
		CHECK(alpgpu_decompress_host_f64(ctx, p_blob, w2, p_out, values, &nv2));
		if (nv2 != values || memcmp(p_in, p_out, values * 8) != 0) {
			printf("FAIL f64: host round trip\n");
			return 1;
		}
		if (alpgpu_compress_host_f64(ctx, p_in, values, p_blob, 4096, &w2) != ALPGPU_ERR_CAPACITY || w2 == 0) {
			printf("FAIL f64: a short blob buffer must be refused with the needed size\n");
			return 1;
		}
		CHECK(alpgpu_free_host(ctx, p_in));
		CHECK(alpgpu_free_host(ctx, p_out));
		CHECK(alpgpu_free_host(ctx, p_blob));
		printf("ok   f64: host-to-host compress / decompress reproduce the blob and the values\n");
	}
	free(blob);
	free_column(ctx, &col);
	free_column(ctx, &col2);
	alpgpu_free(ctx, d_in);
	alpgpu_free(ctx, d_out);
	free(h_in);
	free(h_out);
	return 0;
}

static int roundtrip_f32(alpgpu_ctx* ctx) {
	const uint64_t n      = 130;
	const size_t   values = (size_t)n * 1024;
	float*         h_in   = (float*)malloc(values * 4);
	float*         h_out  = (float*)malloc(values * 4);
	uint64_t       seed   = 7;
	for (size_t i = 0; i < values; ++i) {
		seed    = seed * 6364136223846793005ull + 1442695040888963407ull;
		h_in[i] = (float)(round((double)(seed >> 40) / 16777216.0 * 1e4) / 10.0);
	}
	float *d_in, *d_out;
	CHECK(alpgpu_malloc(ctx, (void**)&d_in, values * 4));
	CHECK(alpgpu_malloc(ctx, (void**)&d_out, values * 4));
	CHECK(alpgpu_memcpy_h2d(ctx, d_in, h_in, values * 4));
	alpgpu_column col;
	if (make_column(ctx, n, 1, &col)) { return 1; }
	CHECK(alpgpu_encode_f32(ctx, d_in, n, &col));
	CHECK(alpgpu_decode_f32(ctx, &col, d_out));
	uint64_t pb, eb;
	int      ov;
	CHECK(alpgpu_column_totals(ctx, &col, &pb, &eb, &ov));
	CHECK(alpgpu_memcpy_d2h(ctx, h_out, d_out, values * 4));
	if (memcmp(h_in, h_out, values * 4) != 0) {
		printf("FAIL f32: decode(encode(x)) differs from x\n");
		return 1;
	}
	printf("ok   f32: %llu vectors, %.2f bits/value\n", (unsigned long long)n, (double)(pb + eb + 32 * n) * 8.0 / (double)values);
	free_column(ctx, &col);
	alpgpu_free(ctx, d_in);
	alpgpu_free(ctx, d_out);
	free(h_in);
	free(h_out);
	return 0;
}

int main(void) {
	alpgpu_ctx* ctx = NULL;
	if (alpgpu_ctx_create(0, &ctx) != ALPGPU_OK) {
		printf("FAIL alpgpu_ctx_create: %s\n", alpgpu_last_error());
		return 1;
	}
	char     name[128];
	int      cus = 0;
	uint64_t hbm = 0;
	alpgpu_device_info(ctx, name, sizeof(name), &cus, &hbm);
	printf("device: %s, %d CUs, ABI %d\n", name, cus, alpgpu_abi_version());
	int failures = roundtrip_f64(ctx) + roundtrip_f32(ctx);
	/* error behaviour: a NULL column is refused, the message is retrievable */
	if (alpgpu_decode_f64(ctx, NULL, NULL) != ALPGPU_ERR_INVALID || strlen(alpgpu_last_error()) == 0) {
		printf("FAIL error reporting\n");
		++failures;
	}
	alpgpu_ctx_destroy(ctx);
	printf("%d failures\n", failures);
	return failures;
}
