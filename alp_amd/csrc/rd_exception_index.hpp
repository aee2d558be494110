// rd_exception_index.hpp — the left index the reference packs at ALP_RD exception slots.
//
// include/alp/rd.hpp:69-77 keeps, next to the dictionary, a map from every sampled left part at sorted position
// i > dictionary_size to i; rd.hpp:130-135 stores map[left] as the vector's left index when the left part is not in the
// dictionary (dictionary_size when the map does not hold it), and FFOR keeps its low left_bit_width bits.  No decoder reads
// those bits (the slot is patched from the exception list), but they are stream bytes: with the sorted order of the
// rowgroup at hand (alpgpu_column.d_rd_order, written by the rowgroup-init kernel) the encoders reproduce them.
#pragma once
#include "alp_device.hpp"

namespace alpgpu {

struct RdOrderView {
	uint32_t keys[5]; // lane l of keys[k] = sorted left part 64k + l (0xFFFFFFFF past the end)
	int      count;   // D
	int      ds;      // dictionary size
	bool     valid;
};

// one wavefront; order_rg = this rowgroup's table or nullptr
// coherent: the table was written by ANOTHER kernel that is still running (the persistent rowgroup search, released before the state
// that announced it): agent-scope loads, which bypass this CU's L1 and this XCD's L2
__device__ __forceinline__ RdOrderView load_rd_order(const uint16_t* __restrict__ order_rg, const alpgpu_rowgroup_state& rg, int lane, bool coherent = false) {
	RdOrderView V;
	V.ds    = rg.rd_dict_size;
	V.count = 0;
	V.valid = false;
#pragma unroll
	for (int k = 0; k < 5; ++k) { V.keys[k] = 0xFFFFFFFFu; }
	if (order_rg == nullptr) { return V; }
	// count and entries are read together (the table's stride covers 1 + 288 entries whatever the count): one round trip
	uint32_t  raw[5];
	int D;
	if (!coherent) { // wave-uniform
		D = order_rg[0];
#pragma unroll
		for (int k = 0; k < 5; ++k) {
			const int i = 64 * k + lane;
			raw[k]      = i < 288 ? order_rg[1 + i] : 0xFFFFu;
		}
	} else {
		D = __hip_atomic_load(order_rg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
		for (int k = 0; k < 5; ++k) {
			const int i = 64 * k + lane;
			raw[k]      = i < 288 ? static_cast<uint32_t>(__hip_atomic_load(order_rg + 1 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0xFFFFu;
		}
	}
	if (D < V.ds || D > 288) { return V; }
#pragma unroll
	for (int k = 0; k < 5; ++k) {
		if (64 * k + lane < D) { V.keys[k] = raw[k]; }
	}
	// the table must belong to this state: its first entries are the dictionary
	uint32_t mine = 0u; // rg may be a register copy of the state: constant indices only
#pragma unroll
	for (int i = 0; i < 8; ++i) { mine = lane == i ? rg.rd_dict[i] : mine; }
	const bool     bad  = lane < V.ds && V.keys[0] != mine;
	V.count = D;
	V.valid = ballot64(bad) == 0;
	return V;
}

// per lane: the index of a left part that is NOT in the dictionary
__device__ __forceinline__ int rd_exception_index(const RdOrderView& V, uint32_t left) {
	int idx = V.ds;
#pragma unroll
	for (int k = 0; k < 5; ++k) {
		const int lo = V.ds + 1 > 64 * k ? V.ds + 1 : 64 * k;
		const int hi = V.count < 64 * k + 64 ? V.count : 64 * k + 64;
		for (int i = lo; i < hi; ++i) { // wave-uniform bounds
			const uint32_t key = __builtin_amdgcn_readlane(V.keys[k], i - 64 * k);
			idx                = key == left ? i : idx;
		}
	}
	return idx;
}

} // namespace alpgpu
