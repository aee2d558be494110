// decode_policy.hpp — the numbers of the store decode's launch rule that BOTH sides use: the host (api_decode.hpp: decode_variant_for, read_ahead_for,
// decode_one) when a column's sizes are known there (packed_bytes_hint / exc_bytes_hint, alpgpu_column_totals), and the device (guard_kernels.hip:
// k_unhinted_plan) when they are not — a column that is decoded right behind its encode, without the host synchronisation alpgpu_column_totals is:
// the sizes are then summed on the stream and the rule is evaluated THERE; what it decides reaches the decode launches as a word of context memory
// (round 6).  Every threshold is a measured crossover: DESIGN.md §3.1, profiles/r05_read_ahead.txt, profiles/r06_decode_policy.txt.
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace alpgpu {

// ---- words of the context's device memory (alpgpu_ctx::d_progress, 2 KiB = 256 words) ----
constexpr int kCtxWordProgress  = 0;  // the running decode's position, tagged (read_ahead_kernels.hip)
constexpr int kCtxWordHole      = 1;  // never written (what keeps the read-ahead's loads)
constexpr int kCtxWordBatches   = 2;  // batches of 64 vectors the read-ahead has read since the context was created (debug counter: alpgpu_debug_read_ahead_batches)
constexpr int kCtxWordShape     = 3;  // unhinted decode: the candidate launch that runs (1-based; the others leave at once)
constexpr int kCtxWordLead      = 4;  // unhinted decode: the read-ahead's lead_min (low word) and lead_max (high word), in vectors; 0 = no read-ahead
constexpr int kCtxWordPace      = 5;  // unhinted decode: picoseconds per vector (low word), widest record read ahead in packed bits (high word)
constexpr int kCtxWordTotals    = 8;  // unhinted decode: packed bytes, exceptions, ALP_RD vectors of the whole column (three words)
constexpr int kCtxWordSegments  = 64; // per-segment sums: 3 words each, up to 32 segments (guard_kernels.hip: k_segment_sums)

// ---- double columns ----
constexpr double   kReadAheadBits    = 6.5;    // the read-ahead pays up to this many packed bits per value on average without exceptions ...
constexpr double   kReadAheadBitsExc = 7.5;    // ... and with exceptions
constexpr uint64_t kReadAheadVectors = 262144; // shorter columns: the cold start and the join of the second stream eat the gain
constexpr double   kTwoVectorsBits    = 17.5;  // two vectors per workgroup up to here without exceptions ...
constexpr double   kTwoVectorsBitsExc = 22.0;  // ... and with (about two or more per vector)
constexpr double   kEmptyVectorsBits    = 0.75; // columns of (almost) nothing but 0-bit vectors: one vector per workgroup, six workgroups per CU, no read-ahead (call 2, the gov26 column)
// ---- float columns (a vector is 4 KiB; two per workgroup are the bytes in flight of one double vector): tools/sweep_f32_decode.py, profiles/r06_float_decode.txt ----
// Measured (call 1b, 1 Mi float vectors, fractions of 8 TB/s): cold, two vectors per workgroup are ahead of one at every width and of four wherever there are exceptions; without
// exceptions four are 1-2 points ahead at 6-14 bits only (and at 1 bit): not worth a rule — always two.  The read-ahead lifts 2-7-bit columns from 0.49-0.60 to
// 0.60-0.63 (with exceptions 0.44-0.54 -> 0.51-0.54) and loses from 8 bits on and at 1 bit; leads of ~20 us beat the double rule's 30-50.
#ifndef ALPGPU_F32_READ_AHEAD_BITS
#define ALPGPU_F32_READ_AHEAD_BITS 7.5
#endif
#ifndef ALPGPU_F32_READ_AHEAD_BITS_EXC
#define ALPGPU_F32_READ_AHEAD_BITS_EXC 7.5
#endif
#ifndef ALPGPU_F32_FOUR_VECTORS_BITS
#define ALPGPU_F32_FOUR_VECTORS_BITS 0.0
#endif
#ifndef ALPGPU_F32_FOUR_VECTORS_BITS_EXC
#define ALPGPU_F32_FOUR_VECTORS_BITS_EXC 0.0
#endif
// Round 6, late: float columns of 2-8-bit vectors WITHOUT exceptions are streamed by persistent workgroups (decode_stream_f32_kernels.hip; shape 27: chunks of 12 vectors,
// 12 decoding + 2 loading wavefronts, one workgroup per CU) — 0.65-0.73 of the HBM peak against 0.60-0.65 for two vectors per workgroup + the read-ahead (calls 19-22).
// With ~2 or more exceptions per vector it is level or behind cold (0.51-0.53 against 0.53-0.55; ahead warm), from 9 bits on it sits on a plateau of 0.69-0.70 that
// the small workgroups pass, and a column of 1-bit vectors is faster as it was (0.76): the rule below.  ALPGPU_F32_STREAM_BITS=0: never.
#ifndef ALPGPU_F32_STREAM_BITS
#define ALPGPU_F32_STREAM_BITS 8.5
#endif
constexpr double   kStreamBitsF32      = ALPGPU_F32_STREAM_BITS;
constexpr double   kStreamLeastBitsF32 = 1.5;
constexpr uint64_t kStreamVectorsF32   = 32768; // shorter columns: a few chunks per workgroup only
constexpr int      kStreamShapeF32     = 27;
__host__ __device__ inline bool policy_stream_f32(uint64_t n_vectors, double packed_bytes, bool with_exc) {
	const double n = static_cast<double>(n_vectors);
	return kStreamBitsF32 > 0.0 && n_vectors >= kStreamVectorsF32 && !with_exc && packed_bytes > kStreamLeastBitsF32 * 128.0 * n && packed_bytes <= kStreamBitsF32 * 128.0 * n;
}
constexpr double kReadAheadBitsF32     = ALPGPU_F32_READ_AHEAD_BITS;
constexpr double kReadAheadBitsExcF32  = ALPGPU_F32_READ_AHEAD_BITS_EXC;
constexpr double kFourVectorsBitsF32    = ALPGPU_F32_FOUR_VECTORS_BITS;     // four float vectors per workgroup up to here (two beyond)
constexpr double kFourVectorsBitsExcF32 = ALPGPU_F32_FOUR_VECTORS_BITS_EXC;

// "the column's vectors carry exceptions" as far as the launch shape is concerned: about two or more per vector (16 bytes of record)
__host__ __device__ inline bool policy_with_exceptions(double n_vectors, double exc_bytes) { return exc_bytes >= 16.0 * n_vectors; }

// the read-ahead on its own (ALPGPU_OPT_DECODE_READ_AHEAD = -1): long columns of narrow vectors only
__host__ __device__ inline bool policy_read_ahead_auto(uint64_t n_vectors, double packed_bytes, bool with_exc, int value_bytes) {
	const double limit = value_bytes == 8 ? (with_exc ? kReadAheadBitsExc : kReadAheadBits) : (with_exc ? kReadAheadBitsExcF32 : kReadAheadBitsF32);
	// (float columns of 1-bit vectors: 0.75 cold, 0.72 with the second stream of reads; double columns of almost nothing but 0-bit vectors — the gov26 shape, a pure stream of
	//  stores: 0.82 with six workgroups per CU and no second stream, 0.71-0.72 with it: kEmptyVectorsBits)
	// (float columns of 1-bit vectors WITH exceptions: 0.44 -> 0.52 with the read-ahead, call 3 — the 1.5-bit floor is for exception-free ones)
	const double least = value_bytes == 8 ? kEmptyVectorsBits : (with_exc ? 0.5 : 1.5);
	return n_vectors >= kReadAheadVectors && packed_bytes <= limit * 128.0 * static_cast<double>(n_vectors) && packed_bytes >= least * 128.0 * static_cast<double>(n_vectors);
}

// The read-ahead's pace and lead.  The lead is a TIME: what the read-ahead brings into the Infinity Cache stays there for some tens of microseconds only (the
// decode's own stores stream through it) and has to be there before the decode asks.  In vectors: that time at the rate of a decode running at 0.78 of the full
// HBM bandwidth.  How long: by the vectors' width unless set — best leads measured per width: 15 / 20 / 30 / 30 / 35 / 40 / 50 us at 1 .. 7 bits, 50-70 us at
// 8-11, with and without exceptions; too short falls off a cliff whose place moves a little from box to box, too long decays slowly: a little above the optimum,
// 12 + 6.5 us per bit, at most 60 (profiles/r05_read_ahead.txt).  The read-ahead itself stretches the lead when its batches arrive late (read_ahead_kernels.hip).
struct ReadAheadPace {
	uint32_t lead_min, lead_max, ps_per_vector;
};
__host__ __device__ inline ReadAheadPace policy_read_ahead_pace(double n_vectors, double packed_bytes, double exc_bytes, int value_bytes, int lead_us_option) {
	const double out_bytes = 1024.0 * value_bytes;
	const double per_vec   = (packed_bytes + exc_bytes) / n_vectors + 32.0;
	const double ps_vec    = (out_bytes + per_vec) / 8.0; // picoseconds per vector at 8 TB/s: an upper bound of the decode's rate (the read-ahead's naps by it never overshoot)
	const double bits      = packed_bytes / (128.0 * n_vectors);
	const double by_width  = value_bytes == 8 ? 12.0 + 6.5 * bits : 12.0 + 3.0 * bits; // (float columns: ~20 us at 3-6 bits, call 1b)
	const double lead_us   = lead_us_option > 0 ? static_cast<double>(lead_us_option) : (by_width > 60.0 ? 60.0 : by_width);
	const double lead      = lead_us * 1.0e6 / ps_vec * 0.78;
	ReadAheadPace p;
	p.lead_max      = static_cast<uint32_t>(lead < 4096.0 ? 4096.0 : (lead > 4.0e9 ? 4.0e9 : lead));
	p.lead_min      = 2048; // about what is resident when a workgroup reports: those vectors' reads are under way
	p.ps_per_vector = static_cast<uint32_t>(ps_vec);
	return p;
}

// ---- the unhinted decode's candidate launches (api_decode.hip launches all of them; k_unhinted_plan says which one runs) ----
// double: 1 = one vector per workgroup, 6 KiB pad (up to ~38 bits); 2 = two vectors per workgroup (narrow vectors); 3 = one per workgroup, 11 KiB pad
//         (seven workgroups per CU: vectors of 38 bits and more, ALP_RD columns)
// float : 1 = two vectors per workgroup (the only shape the rule uses: four never pay, see above)
constexpr int kUnhintedShapesF64 = 3;
constexpr int kUnhintedShapesF32 = 1;
struct UnhintedChoice {
	int  shape; // 1-based
	bool ahead;
};
__host__ __device__ inline UnhintedChoice policy_unhinted(uint64_t n_vectors, double packed_bytes, double exceptions, double rd_vectors, int value_bytes, int read_ahead_option) {
	const double n         = static_cast<double>(n_vectors);
	const double exc_bytes = (value_bytes + 2.0) * exceptions;
	const bool   with_exc  = policy_with_exceptions(n, exc_bytes);
	const double bits      = packed_bytes / (128.0 * n);
	UnhintedChoice c;
	c.ahead = read_ahead_option > 0 ? n_vectors >= 32768 : (read_ahead_option < 0 && policy_read_ahead_auto(n_vectors, packed_bytes, with_exc, value_bytes));
	if (value_bytes == 8) {
		const bool narrow = bits <= (with_exc ? kTwoVectorsBitsExc : kTwoVectorsBits) && bits > kEmptyVectorsBits;
		const bool rd     = 2.0 * rd_vectors > n;
		c.shape           = narrow ? 2 : ((rd || bits >= 38.0 || bits <= kEmptyVectorsBits) ? 3 : 1);
		if (read_ahead_option < 0 && c.ahead && !with_exc) { c.shape = 1; } // under the read-ahead one vector per workgroup is the best shape without exceptions
	} else {
		c.shape = 1;
	}
	return c;
}

} // namespace alpgpu
