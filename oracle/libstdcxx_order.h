/*
 * oracle/libstdcxx_order.h — TEST INFRASTRUCTURE ONLY (part of the CPU oracle, never linked into the product).
 *
 * Observable iteration / sort orders of the two libstdc++ containers the reference's ALP_RD dictionary builder goes
 * through (include/alp/rd.hpp:33-60: std::unordered_map<UT,int32_t> filled in sample order, copied in iteration order
 * into a std::vector, std::sort()ed by count only).  Ties are decided by these implementation details,
 * so they are emulated literally: libstdc++ 11 _Hashtable (identity hash, _Mod_range_hashing, _Prime_rehash_policy,
 * singly linked node list) and std::sort (introsort + final insertion sort).  Shared by the double and float oracles;
 * keys are widened to 64 bits (identity hash of a uint32_t and of the same value as uint64_t coincide).
 */
#ifndef ALP_ORACLE_LIBSTDCXX_ORDER_H
#define ALP_ORACLE_LIBSTDCXX_ORDER_H
#include <stddef.h>
#include <stdint.h>
#include <math.h>
#include <string.h>

#define ALPO_HT_CAPACITY 1024

/* -- libstdc++ _Hashtable<uint64_t, pair<const uint64_t,int32_t>> (unique keys), observable order only */
typedef struct {
	uint64_t key;
	int32_t  val;
	int      next; /* node index or -1 */
} ht_node;
#define HT_BB (-2) /* &_M_before_begin */
#define HT_NULL (-1)
typedef struct {
	ht_node nodes[ALPO_HT_CAPACITY];
	int     n_nodes;
	int     bb_next;       /* _M_before_begin._M_nxt */
	int     buckets[2048]; /* "node before the bucket's first node": node idx, HT_BB or HT_NULL */
	size_t  n_buckets;
	size_t  next_resize; /* _Prime_rehash_policy::_M_next_resize */
} ht_t;

static const unsigned long PRIME_LIST[] = {
    2ul,   3ul,   5ul,   7ul,   11ul,  13ul,  17ul,  19ul,  23ul,  29ul,  31ul,  37ul,  41ul,  43ul,  47ul,  53ul,
    59ul,  61ul,  67ul,  71ul,  73ul,  79ul,  83ul,  89ul,  97ul,  103ul, 109ul, 113ul, 127ul, 137ul, 139ul, 149ul,
    157ul, 167ul, 179ul, 193ul, 199ul, 211ul, 227ul, 241ul, 257ul, 277ul, 293ul, 313ul, 337ul, 359ul, 383ul, 409ul,
    439ul, 467ul, 503ul, 541ul, 577ul, 619ul, 661ul, 709ul, 761ul, 823ul, 887ul, 953ul, 1031ul, 1109ul, 1193ul, 1289ul,
    1381ul, 1493ul, 1613ul, 1741ul, 1879ul, 2029ul};

static size_t ht_next_bkt(ht_t* h, size_t n) { /* _Prime_rehash_policy::_M_next_bkt, max_load_factor 1.0 */
	static const unsigned char fast_bkt[] = {2, 2, 2, 3, 5, 5, 7, 7, 11, 11, 11, 11, 13, 13};
	if (n < sizeof(fast_bkt)) {
		if (n == 0) { return 1; }
		h->next_resize = fast_bkt[n];
		return fast_bkt[n];
	}
	const size_t n_primes = sizeof(PRIME_LIST) / sizeof(PRIME_LIST[0]);
	size_t       i        = 6; /* lower_bound over __prime_list + 6 .. */
	while (i < n_primes - 1 && PRIME_LIST[i] < n) { i++; }
	h->next_resize = PRIME_LIST[i];
	return PRIME_LIST[i];
}

static void ht_init(ht_t* h) {
	h->n_nodes     = 0;
	h->bb_next     = HT_NULL;
	h->n_buckets   = 1;
	h->buckets[0]  = HT_NULL;
	h->next_resize = 0;
}

static int ht_node_next(const ht_t* h, int prev) { return prev == HT_BB ? h->bb_next : h->nodes[prev].next; }
static void ht_set_next(ht_t* h, int prev, int nx) {
	if (prev == HT_BB) {
		h->bb_next = nx;
	} else {
		h->nodes[prev].next = nx;
	}
}

static void ht_rehash(ht_t* h, size_t n) { /* _M_rehash_aux(n, true_type) */
	static _Thread_local int nb[2048];
	for (size_t i = 0; i < n; i++) { nb[i] = HT_NULL; }
	int    p          = h->bb_next;
	size_t bbegin_bkt = 0;
	h->bb_next        = HT_NULL;
	while (p != HT_NULL) {
		const int    next = h->nodes[p].next;
		const size_t bkt  = (size_t)(h->nodes[p].key % n);
		if (nb[bkt] == HT_NULL) {
			h->nodes[p].next = h->bb_next;
			h->bb_next       = p;
			nb[bkt]          = HT_BB;
			if (h->nodes[p].next != HT_NULL) { nb[bbegin_bkt] = p; }
			bbegin_bkt = bkt;
		} else {
			h->nodes[p].next = ht_node_next(h, nb[bkt]);
			ht_set_next(h, nb[bkt], p);
		}
		p = next;
	}
	for (size_t i = 0; i < n; i++) { h->buckets[i] = nb[i]; }
	h->n_buckets = n;
}

/* operator[](key)++ */
static void ht_increment(ht_t* h, uint64_t key) {
	size_t bkt = (size_t)(key % h->n_buckets);
	/* _M_find_before_node */
	int prev = h->buckets[bkt];
	if (prev != HT_NULL) {
		int p = ht_node_next(h, prev);
		for (;;) {
			if (h->nodes[p].key == key) {
				h->nodes[p].val++;
				return;
			}
			const int nx = h->nodes[p].next;
			if (nx == HT_NULL || (size_t)(h->nodes[nx].key % h->n_buckets) != bkt) { break; }
			p = nx;
		}
	}
	/* _M_insert_unique_node: rehash check (_M_need_rehash(n_bkt, n_elt, 1)) */
	const size_t n_elt = (size_t)h->n_nodes;
	if (n_elt + 1 > h->next_resize) {
		/* hashtable_c++0x.cc: max(n_elt + n_ins, _M_next_resize ? 0 : 11) / max_load_factor */
		size_t need = n_elt + 1;
		if (h->next_resize == 0 && need < 11) { need = 11; }
		double min_bkts = (double)need / 1.0;
		if (min_bkts >= (double)h->n_buckets) {
			size_t want = (size_t)floor(min_bkts) + 1;
			size_t grow = h->n_buckets * 2;
			size_t nbk  = ht_next_bkt(h, want > grow ? want : grow);
			if (nbk != h->n_buckets) { ht_rehash(h, nbk); }
			bkt = (size_t)(key % h->n_buckets);
		} else {
			h->next_resize = (size_t)floor((double)h->n_buckets * 1.0);
		}
	}
	const int node      = h->n_nodes++;
	h->nodes[node].key  = key;
	h->nodes[node].val  = 1; /* value-initialised 0 then ++ */
	/* _M_insert_bucket_begin */
	if (h->buckets[bkt] != HT_NULL) {
		h->nodes[node].next = ht_node_next(h, h->buckets[bkt]);
		ht_set_next(h, h->buckets[bkt], node);
	} else {
		h->nodes[node].next = h->bb_next;
		h->bb_next          = node;
		if (h->nodes[node].next != HT_NULL) {
			const int nx                                              = h->nodes[node].next;
			h->buckets[(size_t)(h->nodes[nx].key % h->n_buckets)] = node;
		}
		h->buckets[bkt] = HT_BB;
	}
}

/* -- libstdc++ std::sort on pair<int,uint64_t> with comp(a,b) = (uint16_t)a.first > (uint16_t)b.first */
typedef struct {
	int      first;
	uint64_t second;
} rep_t;
static int rep_comp(const rep_t* a, const rep_t* b) { return (uint16_t)a->first > (uint16_t)b->first; }
static void rep_swap(rep_t* a, rep_t* b) {
	rep_t t = *a;
	*a      = *b;
	*b      = t;
}
static void ss_push_heap(rep_t* first, long hole, long top, rep_t value) {
	long parent = (hole - 1) / 2;
	while (hole > top && rep_comp(&first[parent], &value)) {
		first[hole] = first[parent];
		hole        = parent;
		parent      = (hole - 1) / 2;
	}
	first[hole] = value;
}
static void ss_adjust_heap(rep_t* first, long hole, long len, rep_t value) {
	const long top    = hole;
	long       second = hole;
	while (second < (len - 1) / 2) {
		second = 2 * (second + 1);
		if (rep_comp(&first[second], &first[second - 1])) { second--; }
		first[hole] = first[second];
		hole        = second;
	}
	if ((len & 1) == 0 && second == (len - 2) / 2) {
		second      = 2 * (second + 1);
		first[hole] = first[second - 1];
		hole        = second - 1;
	}
	ss_push_heap(first, hole, top, value);
}
static void ss_heap_sort(rep_t* first, rep_t* last) { /* __partial_sort(first,last,last) */
	const long len = last - first;
	if (len >= 2) {
		long parent = (len - 2) / 2;
		for (;;) {
			rep_t v = first[parent];
			ss_adjust_heap(first, parent, len, v);
			if (parent == 0) { break; }
			parent--;
		}
	}
	while (last - first > 1) {
		--last;
		rep_t v = *last;
		*last   = *first;
		ss_adjust_heap(first, 0, last - first, v);
	}
}
static void ss_move_median_to_first(rep_t* result, rep_t* a, rep_t* b, rep_t* c) {
	if (rep_comp(a, b)) {
		if (rep_comp(b, c)) {
			rep_swap(result, b);
		} else if (rep_comp(a, c)) {
			rep_swap(result, c);
		} else {
			rep_swap(result, a);
		}
	} else if (rep_comp(a, c)) {
		rep_swap(result, a);
	} else if (rep_comp(b, c)) {
		rep_swap(result, c);
	} else {
		rep_swap(result, b);
	}
}
static rep_t* ss_unguarded_partition(rep_t* first, rep_t* last, rep_t* pivot) {
	for (;;) {
		while (rep_comp(first, pivot)) { ++first; }
		--last;
		while (rep_comp(pivot, last)) { --last; }
		if (!(first < last)) { return first; }
		rep_swap(first, last);
		++first;
	}
}
static void ss_introsort_loop(rep_t* first, rep_t* last, long depth_limit) {
	while (last - first > 16) {
		if (depth_limit == 0) {
			ss_heap_sort(first, last);
			return;
		}
		--depth_limit;
		rep_t* mid = first + (last - first) / 2;
		ss_move_median_to_first(first, first + 1, mid, last - 1);
		rep_t* cut = ss_unguarded_partition(first + 1, last, first);
		ss_introsort_loop(cut, last, depth_limit);
		last = cut;
	}
}
static void ss_unguarded_linear_insert(rep_t* last) {
	rep_t  val  = *last;
	rep_t* next = last - 1;
	while (rep_comp(&val, next)) {
		*last = *next;
		last  = next;
		--next;
	}
	*last = val;
}
static void ss_insertion_sort(rep_t* first, rep_t* last) {
	if (first == last) { return; }
	for (rep_t* i = first + 1; i != last; ++i) {
		if (rep_comp(i, first)) {
			rep_t val = *i;
			memmove(first + 1, first, (size_t)(i - first) * sizeof(rep_t));
			*first = val;
		} else {
			ss_unguarded_linear_insert(i);
		}
	}
}
static void ss_sort(rep_t* first, rep_t* last) {
	if (first == last) { return; }
	const long n = last - first;
	ss_introsort_loop(first, last, (long)(63 - __builtin_clzll((unsigned long long)n)) * 2);
	if (n > 16) {
		ss_insertion_sort(first, first + 16);
		for (rep_t* i = first + 16; i != last; ++i) { ss_unguarded_linear_insert(i); }
	} else {
		ss_insertion_sort(first, last);
	}
}

#endif
