// rd_dictionary_order.hpp — the order in which the reference lists the distinct left parts of an ALP_RD sample.
//
// include/alp/rd.hpp:33-60 counts the left parts in a std::unordered_map<UT,int32_t> (filled in sample order), copies it in
// iteration order into a std::vector and std::sort()s that by count only.  Equal counts therefore come out in an order that
// is decided by libstdc++ internals: the node list of _Hashtable (identity hash, modulo bucket index, prime
// rehash policy 13 -> 29 -> 59 -> 127 -> 257 -> 541 buckets, new nodes go to the front of their bucket, a node entering an
// empty bucket goes to the front of the whole list) and introsort + final insertion sort (unstable above 16 elements).
// The dictionary is the first eight entries of that order, so to produce the reference's dictionaries bit for bit the two
// containers are replayed here — our own restatement of their published algorithms, the same one the CPU oracle carries
// (oracle/libstdcxx_order.h), written for ONE lane working on small LDS arrays.
//
// Only insertions of NEW keys change the node list (operator[] on an existing key just increments its count, and the rehash
// check sits on the insertion path), so the replay takes the D distinct left parts in order of first occurrence with their
// final counts: D insertions + at most 2D node moves in rehashes + a sort of D elements, D <= 288 (typically 10-40).
#pragma once
#include "alp_device.hpp"

namespace alpgpu {

constexpr int kRdMaxDistinct = 288;
constexpr int kRdMaxBuckets  = 544; // 541 buckets hold up to 541 nodes before the next rehash

struct RdOrderLds {
	uint32_t okey[kRdMaxDistinct];   // in: distinct left parts in first-occurrence order
	uint32_t ocnt[kRdMaxDistinct];   // in: their counts
	uint32_t sorted[kRdMaxDistinct]; // out: (count << 16) | left part, in the reference's sorted order
	int16_t  next[kRdMaxDistinct];
	int16_t  bkt[kRdMaxBuckets];
	int16_t  nb[kRdMaxBuckets];
	int16_t  stack[3 * 24];
};

namespace rdorder {
constexpr int kNull = -1; // end of list / empty bucket
constexpr int kHead = -2; // "the node before the first node" (_M_before_begin)

__device__ inline int next_bucket_count(int n) { // _Prime_rehash_policy::_M_next_bkt for the sizes this replay can reach
	if (n <= 13) { return 13; }                  // (the first insertion always asks for >= 12 -> 13)
	const int primes[] = {17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 103, 109, 113, 127, 137, 139, 149,
	                      157, 167, 179, 193, 199, 211, 227, 241, 257, 277, 293, 313, 337, 359, 383, 409, 439, 467, 503, 541, 577, 619};
	for (int i = 0; i < static_cast<int>(sizeof(primes) / sizeof(primes[0])); ++i) {
		if (primes[i] >= n) { return primes[i]; }
	}
	return 619;
}

struct Table {
	RdOrderLds* X;
	int         head;        // first node of the list
	int         n_buckets;
	int         next_resize;
	__device__ int  next_of(int prev) const { return prev == kHead ? head : X->next[prev]; }
	__device__ void set_next(int prev, int nx) {
		if (prev == kHead) {
			head = nx;
		} else {
			X->next[prev] = static_cast<int16_t>(nx);
		}
	}
	// _M_rehash_aux(n, unique keys)
	__device__ void rehash(int n) {
		for (int i = 0; i < n; ++i) { X->nb[i] = kNull; }
		int p         = head;
		int first_bkt = 0;
		head          = kNull;
		while (p != kNull) {
			const int nx = X->next[p];
			const int b  = static_cast<int>(X->okey[p] % static_cast<uint32_t>(n));
			if (X->nb[b] == kNull) {
				X->next[p] = static_cast<int16_t>(head);
				head       = p;
				X->nb[b]   = kHead;
				if (X->next[p] != kNull) { X->nb[first_bkt] = static_cast<int16_t>(p); }
				first_bkt = b;
			} else {
				const int before = X->nb[b];
				X->next[p]       = static_cast<int16_t>(next_of(before));
				set_next(before, p);
			}
			p = nx;
		}
		for (int i = 0; i < n; ++i) { X->bkt[i] = X->nb[i]; }
		n_buckets = n;
	}
	// the insertion half of operator[] for a key that is not in the table yet; node index = number of nodes so far
	__device__ void insert_new(int node) {
		const uint32_t key = X->okey[node];
		int            b   = static_cast<int>(key % static_cast<uint32_t>(n_buckets));
		if (node + 1 > next_resize) { // _M_need_rehash(n_bkt, n_elt = node, n_ins = 1), max_load_factor 1
			int need = node + 1;
			if (next_resize == 0 && need < 11) { need = 11; }
			if (need >= n_buckets) {
				const int want = need + 1, grow = n_buckets * 2;
				const int nbk  = next_bucket_count(want > grow ? want : grow);
				next_resize    = nbk;
				if (nbk != n_buckets) { rehash(nbk); }
				b = static_cast<int>(key % static_cast<uint32_t>(n_buckets));
			} else {
				next_resize = n_buckets;
			}
		}
		if (X->bkt[b] != kNull) { // _M_insert_bucket_begin
			const int before = X->bkt[b];
			X->next[node]    = static_cast<int16_t>(next_of(before));
			set_next(before, node);
		} else {
			X->next[node] = static_cast<int16_t>(head);
			head          = node;
			if (X->next[node] != kNull) { X->bkt[X->okey[X->next[node]] % static_cast<uint32_t>(n_buckets)] = static_cast<int16_t>(node); }
			X->bkt[b] = kHead;
		}
	}
};

// ---- std::sort(first, last, by count descending) as libstdc++ implements it, on uint32 (count << 16 | key) -------------
__device__ inline bool before(uint32_t a, uint32_t b) { return (a >> 16) > (b >> 16); }

__device__ inline void push_heap(uint32_t* a, int hole, int top, uint32_t value) {
	int parent = (hole - 1) / 2;
	while (hole > top && before(a[parent], value)) {
		a[hole] = a[parent];
		hole    = parent;
		parent  = (hole - 1) / 2;
	}
	a[hole] = value;
}
__device__ inline void adjust_heap(uint32_t* a, int hole, int len, uint32_t value) {
	const int top    = hole;
	int       second = hole;
	while (second < (len - 1) / 2) {
		second = 2 * (second + 1);
		if (before(a[second], a[second - 1])) { second--; }
		a[hole] = a[second];
		hole    = second;
	}
	if ((len & 1) == 0 && second == (len - 2) / 2) {
		second  = 2 * (second + 1);
		a[hole] = a[second - 1];
		hole    = second - 1;
	}
	push_heap(a, hole, top, value);
}
__device__ inline void heap_sort(uint32_t* a, int len) { // __partial_sort(first, last, last)
	if (len >= 2) {
		for (int parent = (len - 2) / 2;; --parent) {
			adjust_heap(a, parent, len, a[parent]);
			if (parent == 0) { break; }
		}
	}
	for (int last = len; last > 1;) {
		--last;
		const uint32_t v = a[last];
		a[last]          = a[0];
		adjust_heap(a, 0, last, v);
	}
}
__device__ inline void swap32(uint32_t& x, uint32_t& y) {
	const uint32_t t = x;
	x                = y;
	y                = t;
}
__device__ inline void move_median_to_first(uint32_t* s, int result, int a, int b, int c) {
	if (before(s[a], s[b])) {
		if (before(s[b], s[c])) {
			swap32(s[result], s[b]);
		} else if (before(s[a], s[c])) {
			swap32(s[result], s[c]);
		} else {
			swap32(s[result], s[a]);
		}
	} else if (before(s[a], s[c])) {
		swap32(s[result], s[a]);
	} else if (before(s[b], s[c])) {
		swap32(s[result], s[c]);
	} else {
		swap32(s[result], s[b]);
	}
}
__device__ inline int unguarded_partition(uint32_t* s, int first, int last, int pivot) {
	for (;;) {
		while (before(s[first], s[pivot])) { ++first; }
		--last;
		while (before(s[pivot], s[last])) { --last; }
		if (!(first < last)) { return first; }
		swap32(s[first], s[last]);
		++first;
	}
}
__device__ inline void unguarded_linear_insert(uint32_t* s, int last) {
	const uint32_t val  = s[last];
	int            next = last - 1;
	while (next >= 0 && before(val, s[next])) { // (the guard never fires where libstdc++ relies on a sentinel)
		s[last] = s[next];
		last    = next;
		--next;
	}
	s[last] = val;
}
__device__ inline void insertion_sort(uint32_t* s, int first, int last) {
	if (first == last) { return; }
	for (int i = first + 1; i != last; ++i) {
		if (before(s[i], s[first])) {
			const uint32_t val = s[i];
			for (int k = i; k > first; --k) { s[k] = s[k - 1]; }
			s[first] = val;
		} else {
			unguarded_linear_insert(s, i);
		}
	}
}
__device__ inline void sort(uint32_t* s, int n, int16_t* stack) {
	if (n == 0) { return; }
	// __introsort_loop; the recursive call on [cut, last) and the continuing loop on [first, cut) touch disjoint ranges,
	// so an explicit stack of pending ranges replays it exactly
	int sp        = 0;
	stack[sp++]   = 0;
	stack[sp++]   = static_cast<int16_t>(n);
	stack[sp++]   = static_cast<int16_t>(2 * (31 - __builtin_clz(static_cast<unsigned>(n))));
	while (sp > 0) {
		int depth = stack[--sp];
		int last  = stack[--sp];
		int first = stack[--sp];
		while (last - first > 16) {
			if (depth == 0) {
				heap_sort(s + first, last - first);
				break;
			}
			--depth;
			const int mid = first + (last - first) / 2;
			move_median_to_first(s, first, first + 1, mid, last - 1);
			const int cut = unguarded_partition(s, first + 1, last, first);
			stack[sp++]   = static_cast<int16_t>(cut);
			stack[sp++]   = static_cast<int16_t>(last);
			stack[sp++]   = static_cast<int16_t>(depth);
			last          = cut;
		}
	}
	if (n > 16) { // __final_insertion_sort
		insertion_sort(s, 0, 16);
		for (int i = 16; i != n; ++i) { unguarded_linear_insert(s, i); }
	} else {
		insertion_sort(s, 0, n);
	}
}
} // namespace rdorder

// One lane.  In: X.okey / X.ocnt [0, D).  Out: X.sorted[0, D) = the reference's left_parts_sorted_repetitions order.
__device__ inline void rd_reference_order(RdOrderLds& X, int D) {
	rdorder::Table T {&X, rdorder::kNull, 1, 0};
	X.bkt[0] = rdorder::kNull;
	for (int node = 0; node < D; ++node) { T.insert_new(node); }
	int i = 0;
	for (int p = T.head; p != rdorder::kNull && i < D; p = X.next[p]) { X.sorted[i++] = (X.ocnt[p] << 16) | X.okey[p]; }
	rdorder::sort(X.sorted, D, X.stack);
}

} // namespace alpgpu
