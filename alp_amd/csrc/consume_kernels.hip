// consume_kernels.hip — decode fused into a consumer (per-vector SUM, per-vector COUNT of lo <= x <= hi) for gfx950, double columns.
//
// Replaces the scan + aggregate shape of the reference's end-to-end benchmark
//   publication/source_code/bench_end_to_end/src/benchmarks/alp/queries/q1.cpp:63-100   (alp decode of a vector, then aggr_plus over it)
// on top of, per vector,
//   ALP    : generated::falp::fallback::scalar::falp + alp::decoder<double>::patch_exceptions   src/falp.cpp:114-121, include/alp/decoder.hpp:141-149
//   ALP_RD : unffor (u64 right, u16 left) + alp::rd_encoder<double>::decode                     include/alp/rd.hpp:152-178
// The decoded doubles never reach HBM: what is read is 128*bw + 10*exc + 13 bytes per vector, what is written 8 (or 4).
//
// Launch shape.  The storing decode (decode_kernels.hip) is one short-lived workgroup per vector because its 8 KiB STORES want to
// leave in address order; that shape costs two dependent round trips (descriptor, packed words) per workgroup life and tops out at
// ~0.47 of the HBM peak when there is nothing to store (profiles/r02: decode_sum 0.41, its loads-only probe 0.475).  A consumer has no
// output stream to keep in order, so this kernel is PERSISTENT and software-pipelined instead:
//   * grid = a few workgroups per CU, every wavefront on its own: wavefront g of G takes vectors g, g + G, g + 2G, ... (the
//     wavefronts of the chip sweep the column together, so the bytes in flight stay a compact window of the packed stream);
//   * each wavefront owns an LDS ring of 1-KiB pieces; a vector's packed words AND its exception record go HBM -> LDS by LDS-DMA
//     (global_load_lds, no VGPR round trip), up to kPrefetchMax vectors ahead of the one being unpacked; descriptors are scalar loads,
//     one more vector ahead;
//   * loads return in issue order, so "vector k has landed" is one s_waitcnt vmcnt(N) with N = the loads issued after k's;
//   * no workgroup barrier anywhere: a wavefront never waits for another one.
// Summation order (include/alpgpu.h, reproduced by tests/test_decode_sum_gpu.py): lane L adds its 16 values in ascending index order
// (128m + 2L, 128m + 2L + 1 for m = 0..7) starting from +0.0; the 64 lane partials combine by a balanced binary tree over adjacent
// lanes (pairs (0,1), (2,3), ...; then pairs of pairs, ...): six levels.
#include "alp_device.hpp"
#include "launch.hpp"

namespace alpgpu {

#ifndef ALPGPU_CONS_WAVES
#define ALPGPU_CONS_WAVES 8
#endif
#ifndef ALPGPU_CONS_RING
#define ALPGPU_CONS_RING 8
#endif
#ifndef ALPGPU_CONS_PREFETCH
#define ALPGPU_CONS_PREFETCH 4
#endif
#ifndef ALPGPU_CONS_WG_PER_CU
#define ALPGPU_CONS_WG_PER_CU 2
#endif
constexpr int      kConsWaves     = ALPGPU_CONS_WAVES;    // wavefronts per workgroup (no cooperation between them: LDS bookkeeping only)
constexpr int      kRingPieces    = ALPGPU_CONS_RING;     // 1-KiB pieces per wavefront ring (a power of two).  A vector whose packed words + exception
                                                          // record need more pieces than the ring has (8: ALP wider than 56 bits with exceptions, a few
                                                          // ALP_RD shapes) is unpacked straight from HBM, without the ring — rare, correct, not fast
constexpr uint32_t kRingBytes     = 1024u * kRingPieces;
constexpr int      kPrefetchMax   = ALPGPU_CONS_PREFETCH; // vectors in flight behind the one being unpacked (ring space permitting)
constexpr uint32_t kExcStageBytes = 1024; // of a vector's exception record that travels with its packed words (ALP: the values of <= 128
                                          // exceptions, the whole record up to 102; ALP_RD: whole records up to 256 exceptions)
static_assert((kRingPieces & (kRingPieces - 1)) == 0 && kRingPieces >= 4, "ring = power of two");
constexpr int kConsWavesPerSimd = kConsWaves * ALPGPU_CONS_WG_PER_CU / 4; // the occupancy the register budget is sized for

constexpr int kDescBatch = 8;  // descriptors requested together (one 32-byte LDS-DMA load each)
constexpr int kDescSlots = 32; // descriptor ring: the batch in use, the two ahead of it, and the one being replaced
struct __attribute__((aligned(16))) ConsumeLds {
	uint8_t  ring[kRingBytes];
	uint32_t mask[32];
	uint32_t dring[kDescSlots][8]; // descriptors of this wavefront's next vectors, brought in like the data: no scalar load in the loop
};

// ---- LDS access of the main loop: inline assembly on purpose ---------------------------------------------------------------------
// The compiler's wait-count pass knows that global_load_lds writes LDS but cannot tell WHICH bytes: in front of every LDS access it can
// see it waits for ALL outstanding LDS-DMA loads (s_waitcnt vmcnt(0)), which would drain the prefetch queue once per vector.  Accesses
// written as inline assembly carry no memory operand for that pass; their ordering is ours to keep: the LDS unit executes one
// wavefront's operations in issue order, every asm here is volatile with a memory clobber (program order among them is kept), and a
// read's s_waitcnt lgkmcnt(0) sits INSIDE the statement that issues it.
typedef unsigned long long                           ull2v_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) uint8_t    lds_byte_t;
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) { return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const lds_byte_t*)p)); }
__device__ __forceinline__ void     lds_write_b32(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void     lds_or_b32(uint32_t addr, uint32_t v) { asm volatile("ds_or_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ uint32_t lds_read_b32_now(uint32_t addr) {
	uint32_t r;
	asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
	return r;
}
__device__ __forceinline__ uint32_t lds_read_u16_now(uint32_t addr) {
	uint32_t r;
	asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
	return r;
}
__device__ __forceinline__ uint64_t lds_read_b64_now(uint32_t addr) {
	uint64_t r;
	asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
	return r;
}
// Eight reads and the wait for them in ONE statement: a result leaves the statement only when it has arrived.  (Reads and wait as
// separate statements were wrong: the compiler may copy a read's destination registers before the statement that waits — it did, into
// the registers the wait's tied operands were assigned — and such a copy takes whatever the registers held.  The younger wavefront of
// a SIMD, whose LDS returns come later, then summed garbage now and then.)
__device__ __forceinline__ void lds_read8_b128(ull2v_t (&r)[8], const uint32_t (&a)[8]) {
	asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
	             "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15\n\ts_waitcnt lgkmcnt(0)"
	             : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
	             : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])
	             : "memory");
}
__device__ __forceinline__ void lds_read8_b32(uint32_t (&r)[8], const uint32_t (&a)[8]) {
	asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %9\n\tds_read_b32 %2, %10\n\tds_read_b32 %3, %11\n\t"
	             "ds_read_b32 %4, %12\n\tds_read_b32 %5, %13\n\tds_read_b32 %6, %14\n\tds_read_b32 %7, %15\n\ts_waitcnt lgkmcnt(0)"
	             : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
	             : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])
	             : "memory");
}
// s_waitcnt vmcnt(N) for a wave-uniform N known at run time (the instruction takes an immediate).  N = loads issued after the ones
// waited for; a smaller immediate than N is always safe (it waits for more).
__device__ __forceinline__ void wait_vmcnt_le(uint32_t n) {
#define ALPGPU_WAIT_CASE(K)                                                                                            \
	case K: asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory"); break;
	switch (n) {
		ALPGPU_WAIT_CASE(0)
		ALPGPU_WAIT_CASE(1)
		ALPGPU_WAIT_CASE(2)
		ALPGPU_WAIT_CASE(3)
		ALPGPU_WAIT_CASE(4)
		ALPGPU_WAIT_CASE(5)
		ALPGPU_WAIT_CASE(6)
		ALPGPU_WAIT_CASE(7)
		ALPGPU_WAIT_CASE(8)
		ALPGPU_WAIT_CASE(9)
		ALPGPU_WAIT_CASE(10)
		ALPGPU_WAIT_CASE(11)
		ALPGPU_WAIT_CASE(12)
		ALPGPU_WAIT_CASE(13)
		ALPGPU_WAIT_CASE(14)
		ALPGPU_WAIT_CASE(15)
	default:
		if (n >= 32) {
			asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
		} else if (n >= 24) {
			asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
		} else {
			asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
		}
	}
#undef ALPGPU_WAIT_CASE
}

// the same with few cases (waits for somewhat more than asked): for waits that are almost never waits
__device__ __forceinline__ void wait_vmcnt_coarse(uint32_t n) {
	if (n >= 32) {
		asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
	} else if (n >= 16) {
		asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
	} else if (n >= 8) {
		asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
	} else if (n >= 4) {
		asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
	} else {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
}

constexpr int kConsumeSum = 1, kConsumeCount = 2;

// exception mask of the vector as seen by this wavefront (decode_kernels.hip: ExcMask): lane l < 32 holds mask word l and the number of
// exceptions in the words before it
struct ConsExcMask {
	uint32_t word;
	int      excl;
};
__device__ __forceinline__ uint32_t cons_exception_hits(const ConsExcMask& em, int m, int lane, int& rank) {
	const int g    = lane >> 4; // the pair's mask word is 4m + g
	uint32_t  word = __builtin_amdgcn_readlane(em.word, 4 * m);
	int       pref = __builtin_amdgcn_readlane(em.excl, 4 * m);
#pragma unroll
	for (int k = 1; k < 4; ++k) {
		const uint32_t wk = __builtin_amdgcn_readlane(em.word, 4 * m + k);
		const int      pk = __builtin_amdgcn_readlane(em.excl, 4 * m + k);
		word              = g == k ? wk : word;
		pref              = g == k ? pk : pref;
	}
	const int      b0   = (2 * lane) & 31;
	const uint32_t hits = (word >> b0) & 3u;
	rank                = pref + __builtin_popcount(word & ((1u << b0) - 1u));
	return hits;
}

template <int SINK>
__device__ __forceinline__ void consume_one(double& acc, double x, double lo, double hi) {
	if constexpr (SINK == kConsumeSum) {
		acc += x;
	} else {
		acc += (x >= lo && x <= hi) ? 1.0 : 0.0; // small integers: exact in double, any order
	}
}

// Descriptors and ALP_RD dictionaries through the constant address space: a wave-uniform read there is a scalar load whatever the
// kernel has stored in between (behind the LDS-DMA "stores" and the memory clobbers of the loop the compiler would otherwise fall back
// to vector loads — whose s_waitcnt vmcnt(0) drains the prefetch queue once per vector).  Both are read-only for the kernel's lifetime.
typedef __attribute__((address_space(4))) const uint32_t const_u32_t;
__device__ __forceinline__ alpgpu_vector_desc load_desc_scalar(const alpgpu_vector_desc* __restrict__ descs, uint64_t v) {
	const_u32_t* p = (const_u32_t*)(reinterpret_cast<uintptr_t>(descs + v));
	uint32_t     w[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) { w[i] = p[i]; }
	alpgpu_vector_desc d;
	d.packed_off = (static_cast<uint64_t>(w[1]) << 32) | w[0];
	d.exc_off    = (static_cast<uint64_t>(w[3]) << 32) | w[2];
	d.base       = static_cast<int64_t>((static_cast<uint64_t>(w[5]) << 32) | w[4]);
	d.bw         = static_cast<uint8_t>(w[6]);
	d.e          = static_cast<uint8_t>(w[6] >> 8);
	d.f          = static_cast<uint8_t>(w[6] >> 16);
	d.lbw        = static_cast<uint8_t>(w[6] >> 24);
	d.exc_cnt    = static_cast<uint16_t>(w[7]);
	d.scheme     = static_cast<uint16_t>(w[7] >> 16);
	return d;
}
// the same record from LDS: lane l < 8 (every lane l & 7, in fact) holds its dword l & 7
__device__ __forceinline__ alpgpu_vector_desc desc_from_lanes(uint32_t mine) {
	uint32_t w[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) { w[i] = __builtin_amdgcn_readlane(mine, i); }
	alpgpu_vector_desc d;
	d.packed_off = (static_cast<uint64_t>(w[1]) << 32) | w[0];
	d.exc_off    = (static_cast<uint64_t>(w[3]) << 32) | w[2];
	d.base       = static_cast<int64_t>((static_cast<uint64_t>(w[5]) << 32) | w[4]);
	d.bw         = static_cast<uint8_t>(w[6]);
	d.e          = static_cast<uint8_t>(w[6] >> 8);
	d.f          = static_cast<uint8_t>(w[6] >> 16);
	d.lbw        = static_cast<uint8_t>(w[6] >> 24);
	d.exc_cnt    = static_cast<uint16_t>(w[7]);
	d.scheme     = static_cast<uint16_t>(w[7] >> 16);
	return d;
}
static_assert(offsetof(alpgpu_vector_desc, bw) == 24 && offsetof(alpgpu_vector_desc, exc_cnt) == 28 && sizeof(alpgpu_vector_desc) == 32, "descriptor layout");
__device__ __forceinline__ RdDict load_rd_dict_scalar(const alpgpu_rowgroup_state* __restrict__ rgs, uint64_t v, bool is_rd) {
	RdDict dict {0ull, 0ull};
	if (is_rd) { // wave-uniform
		const_u32_t* p = (const_u32_t*)(reinterpret_cast<uintptr_t>(rgs + v / kRowgroup) + 16);
		dict.lo        = (static_cast<uint64_t>(p[1]) << 32) | p[0];
		dict.hi        = (static_cast<uint64_t>(p[3]) << 32) | p[2];
	}
	return dict;
}

// FastLanes u64 unpack of one value pair from the two 16-byte units that hold it (alp_device.hpp: unpack_pair_u64, on words already read)
__device__ __forceinline__ U64Pair unpack_pair_words(const ull2v_t& w0, const ull2v_t& w1, int s, uint64_t mask) {
	U64Pair r;
	r.x = ((w0.x >> s) | ((w1.x << 1) << (63 - s))) & mask;
	r.y = ((w0.y >> s) | ((w1.y << 1) << (63 - s))) & mask;
	return r;
}

template <int SINK>
__global__ __launch_bounds__(64 * kConsWaves, kConsWavesPerSimd) void k_consume_column(const alpgpu_vector_desc* __restrict__ descs,
                                                                    const alpgpu_rowgroup_state* __restrict__ rgs,
                                                                    const uint8_t* __restrict__ packed, const uint8_t* __restrict__ excs,
                                                                    void* __restrict__ out, uint64_t n_vectors, double lo, double hi) {
	typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
	__shared__ ConsumeLds lds[kConsWaves];
	const int      lane   = lane_id();
	const int      wave   = wave_in_wg();
	ConsumeLds&    L      = lds[wave];
	const uint64_t gw     = static_cast<uint64_t>(blockIdx.x) * kConsWaves + wave;
	const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kConsWaves;
	if (gw >= n_vectors) { return; }
	const uint32_t my_cnt   = static_cast<uint32_t>((n_vectors - gw + stride - 1) / stride);
	const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(L.ring)); // LDS byte address of this wavefront's ring
	const uint32_t mask_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(L.mask));

	// the three tables of the ALP decode, one entry per lane, fetched once: a vector's constants are then readlanes, not memory reads.
	// (The empty asm makes the loads complete HERE: a first use inside the loop would cost an s_waitcnt vmcnt(0) per vector.)
	double  t_frac = kFracArr[lane < 21 ? lane : 0];
	double  t_expd = kExpArr[lane < 24 ? lane : 0];
	int64_t t_fact = kFactArr[lane < 19 ? lane : 0];
	asm volatile("" : "+v"(t_frac), "+v"(t_expd), "+v"(t_fact));
	auto lane_u64 = [&](uint64_t b, int idx) {
		const uint32_t l = __builtin_amdgcn_readlane(static_cast<uint32_t>(b), idx), h = __builtin_amdgcn_readlane(static_cast<uint32_t>(b >> 32), idx);
		return (static_cast<uint64_t>(h) << 32) | l;
	};

	uint32_t head = 0, tail = 0; // ring pieces allocated / released so far (running counts; position = count mod kRingPieces)
	uint32_t issued = 0;         // LDS-DMA loads issued so far
	uint32_t next_issue = 0;     // first vector (of this wavefront's sequence) whose loads are not issued yet
	// the vectors in flight, oldest first (scalar registers: the queue is shifted by one per consumed vector)
	constexpr int Q = kPrefetchMax + 1;
	uint32_t      q_start[Q], q_end[Q];
#pragma unroll
	for (int i = 0; i < Q; ++i) { q_start[i] = q_end[i] = 0; }
	// Descriptors travel like the data: HBM -> LDS by LDS-DMA, kDescBatch at a time, two batches ahead of the one in use.  (As scalar
	// loads they cost a cold miss per vector on the critical path: scalar loads return out of order, so every s_waitcnt lgkmcnt — the
	// LDS reads' included — waited for the descriptor requested last: 2.4 us per vector, whatever its width.)
	const uint32_t desc_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(L.dring));
	uint32_t       issued_batches = 0, confirmed_batches = 0; // descriptor batches requested / known to have landed
	uint32_t       bend[3]        = {0u, 0u, 0u};             // `issued` behind each batch that is requested but not yet confirmed, oldest first
	auto issue_desc_batch = [&]() {
		const uint32_t j0 = issued_batches * kDescBatch;
#pragma unroll
		for (int i = 0; i < kDescBatch; ++i) {
			const uint32_t j = j0 + static_cast<uint32_t>(i);
			if (j < my_cnt) { // wave-uniform
				const uint32_t* gd = reinterpret_cast<const uint32_t*>(descs + (gw + static_cast<uint64_t>(j) * stride));
				if (lane < 8) { __builtin_amdgcn_global_load_lds(gd + lane, &L.dring[j & (kDescSlots - 1)][0], 4, 0, 0); }
				++issued;
			}
		}
		const uint32_t slot = issued_batches - confirmed_batches;
#pragma unroll
		for (int i = 0; i < 3; ++i) { bend[i] = slot == static_cast<uint32_t>(i) ? issued : bend[i]; }
		++issued_batches;
	};
	auto ensure_desc = [&](uint32_t j) { // descriptor j is readable from the ring
		while (j / kDescBatch >= confirmed_batches) {
			wait_vmcnt_coarse(issued - bend[0]); // requested two batches (>= 8 vectors) ago: landed long since
			bend[0] = bend[1], bend[1] = bend[2];
			++confirmed_batches;
		}
	};
	auto read_desc = [&](uint32_t j) {
		ensure_desc(j);
		return desc_from_lanes(lds_read_b32_now(desc_lds + 32u * (j & (kDescSlots - 1)) + 4u * static_cast<uint32_t>(lane & 7)));
	};
	issue_desc_batch();
	if (issued_batches * kDescBatch < my_cnt) { issue_desc_batch(); }
	alpgpu_vector_desc d_issue = read_desc(0);

	auto pieces_of = [](const alpgpu_vector_desc& d, uint32_t& pk_pieces, uint32_t& exc_loads) {
		const bool     is_alp  = d.scheme == ALPGPU_SCHEME_ALP;
		const uint32_t n_units = 8u * (static_cast<uint32_t>(d.bw) + (is_alp ? 0u : static_cast<uint32_t>(d.lbw)));
		const uint32_t rec     = ((is_alp ? 10u : 4u) * static_cast<uint32_t>(d.exc_cnt) + 7u) & ~7u;
		const uint32_t staged  = rec < kExcStageBytes ? rec : kExcStageBytes;
		pk_pieces              = (n_units + 63u) >> 6;
		exc_loads              = (staged + 255u) >> 8; // 256 bytes per load (4 bytes per lane)
		return n_units;
	};

	for (uint32_t k = 0; k < my_cnt; ++k) {
		// descriptors: two batches ahead of the batch vector k is in
		while (issued_batches * kDescBatch < my_cnt && issued_batches < k / kDescBatch + 3u) { issue_desc_batch(); }
		// keep the ring full: vectors k .. k + kPrefetchMax, ring space permitting (vector k itself always fits an empty ring)
		while (next_issue < my_cnt && next_issue - k <= static_cast<uint32_t>(kPrefetchMax)) {
			uint32_t       pk_pieces, exc_loads;
			const uint32_t n_units = pieces_of(d_issue, pk_pieces, exc_loads);
			uint32_t       pieces  = pk_pieces + (exc_loads ? 1u : 0u);
			if (pieces > static_cast<uint32_t>(kRingPieces)) { pieces = pk_pieces = exc_loads = 0u; } // does not fit the ring at all: read directly when its turn comes
			if (head + pieces - tail > static_cast<uint32_t>(kRingPieces)) { break; }
			const ull2* g = reinterpret_cast<const ull2*>(packed + d_issue.packed_off);
			for (uint32_t j = 0; j < pk_pieces; ++j) { // wave-uniform trip count
				const uint32_t c   = 64u * j + static_cast<uint32_t>(lane);
				uint8_t*       dst = L.ring + (((head + j) & (kRingPieces - 1u)) << 10); // wave-uniform base; the hardware adds 16 * lane
#ifdef ALPGPU_CONS_NO_DMA // experiment: through registers
				if (c < n_units) { reinterpret_cast<ull2*>(dst)[lane] = g[c]; }
#else
				if (c < n_units) { __builtin_amdgcn_global_load_lds(g + c, reinterpret_cast<ull2*>(dst), 16, 0, 0); }
#endif
			}
			if (exc_loads) {
				const bool      is_alp = d_issue.scheme == ALPGPU_SCHEME_ALP;
				const uint32_t  rec    = ((is_alp ? 10u : 4u) * static_cast<uint32_t>(d_issue.exc_cnt) + 7u) & ~7u;
				const uint32_t  dwords = (rec < kExcStageBytes ? rec : kExcStageBytes) >> 2;
				const uint32_t* ge     = reinterpret_cast<const uint32_t*>(excs + d_issue.exc_off);
				uint8_t*        dst    = L.ring + (((head + pk_pieces) & (kRingPieces - 1u)) << 10);
				for (uint32_t q = 0; q < exc_loads; ++q) {
					const uint32_t c = 64u * q + static_cast<uint32_t>(lane);
#ifdef ALPGPU_CONS_NO_DMA
					if (c < dwords) { reinterpret_cast<uint32_t*>(dst + 256u * q)[lane] = ge[c]; }
#else
					if (c < dwords) { __builtin_amdgcn_global_load_lds(ge + c, reinterpret_cast<uint32_t*>(dst + 256u * q), 4, 0, 0); }
#endif
				}
			}
			issued += pk_pieces + exc_loads;
			const uint32_t slot = next_issue - k;
#pragma unroll
			for (int i = 0; i < Q; ++i) {
				q_start[i] = slot == static_cast<uint32_t>(i) ? head : q_start[i];
				q_end[i]   = slot == static_cast<uint32_t>(i) ? issued : q_end[i];
			}
			head += pieces;
			++next_issue;
			if (next_issue < my_cnt) { d_issue = read_desc(next_issue); }
		}
		const uint64_t           v = gw + static_cast<uint64_t>(k) * stride;
		const alpgpu_vector_desc d = read_desc(k);
		const bool               is_alp = d.scheme == ALPGPU_SCHEME_ALP;
		const RdDict             dict   = load_rd_dict_scalar(rgs, v, !is_alp);
		const uint32_t start_piece = q_start[0];
		const uint32_t end_count   = q_end[0];
#pragma unroll
		for (int i = 0; i + 1 < Q; ++i) {
			q_start[i] = q_start[i + 1];
			q_end[i]   = q_end[i + 1];
		}
		uint32_t pk_pieces, exc_loads;
		(void)pieces_of(d, pk_pieces, exc_loads);
		const bool direct = pk_pieces + (exc_loads ? 1u : 0u) > static_cast<uint32_t>(kRingPieces); // not in the ring: everything from HBM
		if (direct) { pk_pieces = exc_loads = 0u; }
		const int      cnt      = d.exc_cnt;
		const uint32_t val_b    = is_alp ? 8u : 2u;
		const uint32_t rec      = ((val_b + 2u) * static_cast<uint32_t>(cnt) + 7u) & ~7u;
		const bool     whole    = !direct && rec <= kExcStageBytes;                                       // positions staged too
		const int      n_staged = direct ? 0 : (whole ? cnt : static_cast<int>(kExcStageBytes / val_b)); // values readable from the ring
		const uint8_t* rec_g    = excs + d.exc_off;
		const uint32_t rec_l    = ring_lds + (((start_piece + pk_pieces) & (kRingPieces - 1u)) << 10);
		if (cnt > 0 && lane < 32) { lds_write_b32(mask_lds + 4u * static_cast<uint32_t>(lane), 0u); }
		wait_vmcnt_le(issued - end_count); // everything of vector k has landed in the ring
#ifdef ALPGPU_CONS_DELAY // experiment
		for (int zz = 0; zz < ALPGPU_CONS_DELAY; ++zz) { __builtin_amdgcn_s_sleep(127); }
#endif
#ifdef ALPGPU_CONS_VERIFY // experiment: is the ring what was asked for?  out[v] = mismatching 16-byte units + 1000 * (first bad unit + 1)
		{
			const uint32_t n_units_k = 8u * (static_cast<uint32_t>(d.bw) + (is_alp ? 0u : static_cast<uint32_t>(d.lbw)));
			const ull2v_t* gk        = reinterpret_cast<const ull2v_t*>(packed + d.packed_off);
			int            bad = 0, first = 1 << 20;
			for (uint32_t c = lane; c < n_units_k; c += 64) {
				const ull2v_t want = gk[c];
				ull2v_t       have;
				asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(have) : "v"(ring_lds + ((((start_piece & (kRingPieces - 1u)) << 10) + 16u * c) & (kRingBytes - 1u))) : "memory");
				if (want.x != have.x || want.y != have.y) {
					++bad;
					first = first < static_cast<int>(c) ? first : static_cast<int>(c);
				}
			}
			for (int dd = 32; dd >= 1; dd >>= 1) {
				bad += __shfl_xor(bad, dd);
				const int o = __shfl_xor(first, dd);
				first       = o < first ? o : first;
			}
			if (lane == 0) { static_cast<double*>(out)[v] = bad == 0 ? 0.0 : static_cast<double>(bad + 1000 * (first + 1)); }
			tail = start_piece + pk_pieces + (exc_loads ? 1u : 0u);
			continue;
		}
#endif
		ConsExcMask em {0u, 0};
		if (cnt > 0) { // wave-uniform
			if (whole) {
				const uint32_t pos_l = rec_l + val_b * static_cast<uint32_t>(cnt);
				for (int j = lane; j < cnt; j += 64) {
					const uint32_t p = lds_read_u16_now(pos_l + 2u * static_cast<uint32_t>(j));
					lds_or_b32(mask_lds + 4u * (p >> 5), 1u << (p & 31u));
				}
			} else { // a record larger than the stage: positions straight from HBM (rare: > 102 ALP / 256 ALP_RD exceptions in one vector)
				const uint16_t* pos = reinterpret_cast<const uint16_t*>(rec_g + val_b * static_cast<uint32_t>(cnt));
				for (int j = lane; j < cnt; j += 64) {
					const uint32_t p = pos[j];
					lds_or_b32(mask_lds + 4u * (p >> 5), 1u << (p & 31u));
				}
			}
			em.word     = lds_read_b32_now(mask_lds + 4u * static_cast<uint32_t>(lane & 31));
			const int c = lane < 32 ? __builtin_popcount(em.word) : 0;
			int       s = c;
			s += __builtin_amdgcn_update_dpp(0, s, 0x111, 0xf, 0xf, false);
			s += __builtin_amdgcn_update_dpp(0, s, 0x112, 0xf, 0xf, false);
			s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xf, false);
			s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xf, false);
			s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xa, 0xf, false);
			em.excl = s - c;
		}
		auto exception_bits = [&](int rank) -> uint64_t { // the value of the exception of that rank: from the ring, or (past the stage) from HBM
			uint64_t r;
			if (rank < n_staged) {
				r = is_alp ? lds_read_b64_now(rec_l + 8u * static_cast<uint32_t>(rank)) : static_cast<uint64_t>(lds_read_u16_now(rec_l + 2u * static_cast<uint32_t>(rank)));
			} else {
				r = is_alp ? reinterpret_cast<const uint64_t*>(rec_g)[rank] : static_cast<uint64_t>(reinterpret_cast<const uint16_t*>(rec_g)[rank]);
			}
			return r;
		};

		// This lane's packed words, four value steps (= eight 16-byte units) at a time: pair (m, lane) sits in row 8m + (lane >> 3), unit
		// column a = lane & 7.  (All sixteen units at once would hold 64 registers: over the 128 that four wavefronts per SIMD leave.)
		const uint32_t vstart = (start_piece & (kRingPieces - 1u)) << 10;
		const int      bw     = d.bw;
		const int      a      = lane & 7;
		const int      r0     = lane >> 3;
		const uint64_t mask   = bw_mask(bw);
		// ALP: the vector's constants, and the conversion shortcut of decode_kernels.hip (decode_staged_vector) decided once from its descriptor
		const uint64_t base   = static_cast<uint64_t>(d.base);
		const int64_t  fact   = static_cast<int64_t>(lane_u64(static_cast<uint64_t>(t_fact), is_alp ? d.f : 0));
		const double   frac   = __longlong_as_double(static_cast<long long>(lane_u64(static_cast<uint64_t>(__double_as_longlong(t_frac)), is_alp ? d.e : 0)));
		const double   fact_d = __longlong_as_double(static_cast<long long>(lane_u64(static_cast<uint64_t>(__double_as_longlong(t_expd)), is_alp ? d.f : 0)));
		const int64_t  blo      = d.base;
		const bool     narrow   = is_alp && bw <= 50 && blo > -(1ll << 51) && blo < (1ll << 51) && blo + static_cast<int64_t>(mask) < (1ll << 51);
		const double   maxabs   = narrow ? __builtin_fmax(__builtin_fabs(static_cast<double>(blo)), __builtin_fabs(static_cast<double>(blo + static_cast<int64_t>(mask)))) : 0.0;
		const bool     shortcut = narrow && maxabs * fact_d < 9.2233720368547e18;
		const uint64_t kbits    = 0x4338000000000000ull + base;
		// ALP_RD (decode_kernels.hip): right parts = u64 lanes (bw = rbw, base 0); left parts = u16 lanes right behind them, 64 streams x 16
		// rows: a lane's pair shares the row 2m + (lane >> 5) and is one aligned u32 of the left stream
		const int      lbw   = d.lbw;
		const uint32_t lmsk  = (1u << lbw) - 1u;
		const uint64_t dlo = dict.lo, dhi = dict.hi;
		const uint32_t lbase = vstart + 128u * static_cast<uint32_t>(bw);
		double         acc   = 0.0;
#pragma unroll 1 // (code size: the two halves share their instructions; unrolled, the kernel is twice as long for nothing)
		for (int h = 0; h < 2; ++h) {
			ull2v_t w[8]; // w[2i], w[2i + 1]: the unit of step m = 4h + i and the unit one stream word further
			int     sh[4];
			{
				uint32_t ad[8];
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const int      p   = (8 * (4 * h + i) + r0) * bw;
					const uint32_t off = 16u * static_cast<uint32_t>(8 * (p >> 6) + a);
					sh[i]              = p & 63;
					ad[2 * i]          = ring_lds + ((vstart + off) & (kRingBytes - 1u));
					ad[2 * i + 1]      = ring_lds + ((vstart + off + 128u) & (kRingBytes - 1u));
				}
				if (!direct) { // wave-uniform
					lds_read8_b128(w, ad);
				} else {
					const ull2v_t* gu = reinterpret_cast<const ull2v_t*>(packed + d.packed_off);
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						const int u  = 8 * (((8 * (4 * h + i) + r0) * bw) >> 6) + a;
						w[2 * i]     = gu[u];
						w[2 * i + 1] = gu[u + 8]; // one unit row past the vector's end may be read (content irrelevant; the streams carry that slack)
					}
				}
			}
			if (is_alp) {
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const int     m = 4 * h + i;
					const U64Pair u = unpack_pair_words(w[2 * i], w[2 * i + 1], sh[i], mask);
					double        ox, oy;
					if (shortcut) { // wave-uniform
						ox = ((__longlong_as_double(static_cast<long long>(u.x + kbits)) - kMagic) * fact_d) * frac;
						oy = ((__longlong_as_double(static_cast<long long>(u.y + kbits)) - kMagic) * fact_d) * frac;
					} else {
						ox = decode_value(static_cast<int64_t>(u.x + base), fact, frac);
						oy = decode_value(static_cast<int64_t>(u.y + base), fact, frac);
					}
					if (cnt > 0) {
						int            rank;
						const uint32_t hits = cons_exception_hits(em, m, lane, rank);
						if (hits & 1u) {
							ox = __longlong_as_double(static_cast<long long>(exception_bits(rank)));
							++rank;
						}
						if (hits & 2u) { oy = __longlong_as_double(static_cast<long long>(exception_bits(rank))); }
					}
					consume_one<SINK>(acc, ox, lo, hi);
					consume_one<SINK>(acc, oy, lo, hi);
				}
			} else {
				uint32_t lw[8]; // lw[2i], lw[2i + 1]: the left-stream words of step 4h + i
				{
					uint32_t ad[8];
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						const int      p   = (2 * (4 * h + i) + (lane >> 5)) * lbw;
						const uint32_t off = 4u * static_cast<uint32_t>(32 * (p >> 4) + (lane & 31));
						ad[2 * i]          = ring_lds + ((lbase + off) & (kRingBytes - 1u));
						ad[2 * i + 1]      = ring_lds + ((lbase + off + 128u) & (kRingBytes - 1u));
					}
					if (!direct) {
						lds_read8_b32(lw, ad);
					} else {
						const uint32_t* gl = reinterpret_cast<const uint32_t*>(packed + d.packed_off + 128ull * static_cast<uint32_t>(bw));
#pragma unroll
						for (int i = 0; i < 4; ++i) {
							const int j   = 32 * (((2 * (4 * h + i) + (lane >> 5)) * lbw) >> 4) + (lane & 31);
							lw[2 * i]     = gl[j];
							lw[2 * i + 1] = gl[j + 32];
						}
					}
				}
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const int      m  = 4 * h + i;
					const U64Pair  u  = unpack_pair_words(w[2 * i], w[2 * i + 1], sh[i], mask);
					const int      s  = ((2 * m + (lane >> 5)) * lbw) & 15;
					const uint32_t i0 = (((lw[2 * i] & 0xFFFFu) >> s) | ((lw[2 * i + 1] & 0xFFFFu) << (16 - s))) & lmsk;
					const uint32_t i1 = (((lw[2 * i] >> 16) >> s) | ((lw[2 * i + 1] >> 16) << (16 - s))) & lmsk;
					uint64_t       l0 = ((i0 < 4 ? dlo >> (16 * i0) : dhi >> (16 * (i0 & 3))) & 0xFFFFull);
					uint64_t       l1 = ((i1 < 4 ? dlo >> (16 * i1) : dhi >> (16 * (i1 & 3))) & 0xFFFFull);
					if (cnt > 0) {
						int            rank;
						const uint32_t hits = cons_exception_hits(em, m, lane, rank);
						if (hits & 1u) {
							l0 = exception_bits(rank);
							++rank;
						}
						if (hits & 2u) { l1 = exception_bits(rank); }
					}
					consume_one<SINK>(acc, __longlong_as_double(static_cast<long long>((l0 << bw) | u.x)), lo, hi);
					consume_one<SINK>(acc, __longlong_as_double(static_cast<long long>((l1 << bw) | u.y)), lo, hi);
				}
			}
		}
		// this vector's ring reads have all returned (every one of them was waited for): its pieces are free for the next loads
		tail = start_piece + pk_pieces + (exc_loads ? 1u : 0u);
		const double total = wave_tree_sum_f64(acc);
		if (lane == 0) {
			if constexpr (SINK == kConsumeCount) {
				static_cast<uint32_t*>(out)[v] = static_cast<uint32_t>(total);
			} else {
				static_cast<double*>(out)[v] = total;
			}
		}
	}
}

// ---- the column's total: per-vector sums combined by a balanced binary tree over adjacent elements ----------------------------------
// (the reference's consumer adds every vector into one accumulator, q1.cpp:91-100; a sequential chain of a million dependent adds has
// no parallel form with the same rounding, so the order is this documented tree instead.)  One level of the reduction: block b adds
// elements [1024 b, 1024 b + 1024) of `in` (absent elements count as +0.0) by adjacent pairs, pairs of pairs, ...; out[b] = that sum.
__global__ __launch_bounds__(256) void k_tree_sum(const double* __restrict__ in, uint64_t n, double* __restrict__ out) {
	__shared__ double s_w[4];
	const uint64_t    i0 = static_cast<uint64_t>(blockIdx.x) * 1024 + 4ull * threadIdx.x;
	double            e[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) { e[j] = i0 + j < n ? in[i0 + j] : 0.0; }
	const double w = wave_tree_sum_f64((e[0] + e[1]) + (e[2] + e[3]));
	if ((threadIdx.x & 63) == 0) { s_w[threadIdx.x >> 6] = w; }
	__syncthreads();
	if (threadIdx.x == 0) { out[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]); }
}

static unsigned consume_grid(int n_cus, uint64_t n_vectors) {
	const uint64_t want = static_cast<uint64_t>(n_cus > 0 ? n_cus : 256) * ALPGPU_CONS_WG_PER_CU;
	const uint64_t need = (n_vectors + kConsWaves - 1) / kConsWaves;
	return static_cast<unsigned>(need < want ? need : want);
}

int launch_consume_sum(hipStream_t stream, const alpgpu_column* col, double* d_sums, int n_cus) {
	hipLaunchKernelGGL((k_consume_column<kConsumeSum>), dim3(consume_grid(n_cus, col->n_vectors)), dim3(64 * kConsWaves), 0, stream, col->d_vectors, col->d_rowgroups,
	                   col->d_packed, col->d_exc, static_cast<void*>(d_sums), col->n_vectors, 0.0, 0.0);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

int launch_consume_count_range(hipStream_t stream, const alpgpu_column* col, double lo, double hi, uint32_t* d_counts, int n_cus) {
	hipLaunchKernelGGL((k_consume_column<kConsumeCount>), dim3(consume_grid(n_cus, col->n_vectors)), dim3(64 * kConsWaves), 0, stream, col->d_vectors, col->d_rowgroups,
	                   col->d_packed, col->d_exc, static_cast<void*>(d_counts), col->n_vectors, lo, hi);
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

// d_scratch: two buffers of ceil(n / 1024) doubles back to back; *d_total receives the tree's root (0.0 for n = 0)
int launch_tree_sum(hipStream_t stream, const double* d_in, uint64_t n, double* d_scratch, double* d_total) {
	if (n == 0) { return hipMemsetAsync(d_total, 0, sizeof(double), stream) == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP; }
	const uint64_t l1  = (n + 1023) / 1024;
	double*        buf[2] = {d_scratch, d_scratch + l1};
	const double*  src = d_in;
	int            t   = 0;
	while (true) {
		const uint64_t blocks = (n + 1023) / 1024;
		double*        dst    = blocks == 1 ? d_total : buf[t];
		hipLaunchKernelGGL(k_tree_sum, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, src, n, dst);
		if (blocks == 1) { break; }
		src = dst;
		n   = blocks;
		t ^= 1;
	}
	return hipGetLastError() == hipSuccess ? ALPGPU_OK : ALPGPU_ERR_HIP;
}

} // namespace alpgpu
