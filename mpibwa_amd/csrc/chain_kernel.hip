// chain_kernel.hip — seeds -> chains -> filtered chains on the device, one lane per read.
//
// Device counterpart of mem_chain (src/bwamem.c:251-315), test_and_merge (:190-211), mem_chain_weight (:213-237) and
// mem_chain_flt (:327-385).  One body (chain_read) over two kinds of storage and two statements of the reference's ordered map:
//  * the reads the reference's own data structure keeps trivial: its map is a B-tree whose root holds up to 9 keys
//    (src/kbtree.h, node size 512 B), so a read that never has more than 9 chains lives in one sorted array — MapArray, with the
//    B-tree's rules for equal keys (a new key goes right behind the FIRST equal one; a lookup that hits equal keys returns the
//    first), state in LDS (StoreLds).  Three launches with growing LDS footprints (chain_kernel<seeds, chains, ...>);
//  * the reads with more chains: the B-tree itself (MapBtree: kb_intervalp / kb_putp with t = 5), state in an HBM slice per read
//    (StoreGen), up to 255 seeds and chains (chain_general_kernel).
// Reads with more than 255 seeds, or long enough for mem_flt_chained_seeds to act (l_query >= ~700 bp), are flagged
// (n_chains = -1) and take the host path (host_chain.cpp).
//
// The unstable sort of mem_chain_flt is ks_introsort (src/ksort.h:176-226) statement by statement (ck_introsort).
// Floating-point compares (mask_level, drop_ratio, frac_rep) are IEEE single precision in the same expression shapes as
// the reference (no contraction, no fast-math).
//
// Output per read, in the slots the read's seeds already own (seed_off[r] .. seed_off[r] + n_seeds[r]): the kept chains
// (DevChain, in mem_chain_flt's order, with the reference window of mem_chain2aln), their seeds in the order
// mem_chain2aln visits them (ascending (score, index) — src/bwamem.c:662-667), and the identity visiting order.
// Latency-bound integer work: ~1.5 k instructions per ordinary read, a few hundred bytes of HBM per read.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>
#include "device.h"

namespace mbw {

#define HIP_OK(call)                                                                                             \
	do {                                                                                                         \
		hipError_t e_ = (call);                                                                                  \
		if (e_ != hipSuccess) die("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);    \
	} while (0)

namespace {

typedef long long i64;

#define CK_MAXCH 9
#define CK_MAXCH_SMALL 4
#define CK_MAXSEEDS_SMALL 16   // first launch: every read, up to this many seeds and CK_MAXCH_SMALL chains
#define CK_MAXSEEDS 64         // second launch: what the first declined, up to this many seeds
#define CK_MAXSEEDS_BIG 255    // third launch (and the B-tree kernel): reads with more seeds than that (seed numbers are bytes)
#define CK_MAXCH_GEN 255       // third launch: chains per read (chain numbers are bytes, 255 = none)
#define CK_MAXNODES 128        // ... and B-tree nodes per read (255 keys in nodes of >= 4 need 64 leaves + their parents)
enum { F_POS_LO = 0, F_POS_HI, F_FIRST_Q, F_LAST_R_LO, F_LAST_R_HI, F_LAST_Q, F_LAST_LEN, F_RID, F_N, F_W, F_KEPT, F_FIRSTOV, F_NFIELDS,
       F_HEAD = F_NFIELDS, F_TAIL, F_NFIELDS_GEN };

// ---- where a read's working set lives ----
// StoreLds: the reads the reference's ordered map keeps in ONE node (at most 9 chains): everything in LDS, [..][lane].
// StoreGen: up to 255 chains: the same fields in a per-read slice of an HBM scratch buffer, plus the seeds of a chain as a
//           linked list in arrival order (the weight and emission loops of a read with 150 seeds and 50 chains would otherwise
//           scan all seeds once per chain) and the nodes of the B-tree.
template <int MAXCH_>
struct StoreLds {
	static constexpr int MAXCH = MAXCH_;
	static constexpr bool GENERAL = false;
	uint32_t *tab;    // [F_NFIELDS * MAXCH][64]
	uint8_t *cid_;    // [MAXS + 1][64]     chain id of every seed (255 = in no chain)
	uint8_t *ord_;    // [16][64]           chain ids in tree order, later in filter order
	uint8_t *tmp_;    // [MAXS + 1][64]     scratch: members of one chain
	int lane;
	__device__ __forceinline__ uint32_t &f(int field, int id) const { return tab[(field * MAXCH + id) * 64 + lane]; }
	__device__ __forceinline__ uint8_t &cid(int k) const { return cid_[k * 64 + lane]; }
	__device__ __forceinline__ uint8_t &ord(int k) const { return ord_[k * 64 + lane]; }
	__device__ __forceinline__ uint8_t &tmp(int k) const { return tmp_[k * 64 + lane]; }
};
struct GenNode { uint8_t internal, n, key[2 * 5 - 1], child[2 * 5], pad[3]; };   // the reference's node: t = 5, up to 9 keys (src/kbtree.h, 512-byte nodes of 40-byte chains)
struct StoreGen {
	static constexpr int MAXCH = CK_MAXCH_GEN;
	static constexpr bool GENERAL = true;
	uint32_t *tab;    // [F_NFIELDS_GEN][CK_MAXCH_GEN]
	uint8_t *cid_, *ord_, *tmp_, *next_;   // [256] each
	GenNode *nodes;   // [CK_MAXNODES]
	__device__ __forceinline__ uint32_t &f(int field, int id) const { return tab[field * CK_MAXCH_GEN + id]; }
	__device__ __forceinline__ uint8_t &cid(int k) const { return cid_[k]; }
	__device__ __forceinline__ uint8_t &ord(int k) const { return ord_[k]; }
	__device__ __forceinline__ uint8_t &tmp(int k) const { return tmp_[k]; }
	__device__ __forceinline__ uint8_t &next(int k) const { return next_[k]; }
};
constexpr size_t CK_GEN_BYTES = ((size_t)F_NFIELDS_GEN * CK_MAXCH_GEN * 4 + 4 * 256 + CK_MAXNODES * sizeof(GenNode) + 255) & ~(size_t)255;

template <class S> __device__ __forceinline__ i64 st_pos(const S &L, int id) { return (i64)((unsigned long long)L.f(F_POS_HI, id) << 32 | L.f(F_POS_LO, id)); }
template <class S> __device__ __forceinline__ i64 st_last_r(const S &L, int id) { return (i64)((unsigned long long)L.f(F_LAST_R_HI, id) << 32 | L.f(F_LAST_R_LO, id)); }

// ---- the ordered map of mem_chain (src/bwamem.c:263, 288-300; src/kbtree.h) ----
// One node: a sorted array with the B-tree's rules for equal keys (a new key goes right behind the first key that is not
// smaller ... stepped back; a lookup that hits equal keys returns the first of them).
struct MapArray {
	int n = 0;
	template <class S> __device__ __forceinline__ int lower(const S &L, i64 pos) const   // closest chain at or before pos, or -1
	{
		int f = 0;
		while (f < n && st_pos(L, L.ord(f)) < pos) ++f;
		const int li = (f < n && st_pos(L, L.ord(f)) == pos) ? f : f - 1;
		return li >= 0 ? (int)L.ord(li) : -1;
	}
	template <class S> __device__ __forceinline__ bool put(const S &L, int id, i64 pos)
	{
		if (n == S::MAXCH) return false;   // the reference's root node would split here
		int f = 0;
		while (f < n && st_pos(L, L.ord(f)) < pos) ++f;
		const int li = (f < n && st_pos(L, L.ord(f)) == pos) ? f : f - 1;
		for (int t = n; t > li + 1; --t) L.ord(t) = L.ord(t - 1);
		L.ord(li + 1) = (uint8_t)id;
		++n;
		return true;
	}
	template <class S> __device__ __forceinline__ void in_order(const S &) const {}   // ord already is
};
// The B-tree itself (kb_intervalp / kb_putp of src/kbtree.h with t = 5), because with equal keys what a lookup returns
// depends on where the splits put them.
struct MapBtree {
	int n = 0, n_nodes = 0, root = 0;
	bool full = false;
	template <class S> __device__ __forceinline__ int make(const S &L)
	{
		if (n_nodes == CK_MAXNODES) { full = true; return 0; }
		GenNode &x = L.nodes[n_nodes];
		x.internal = 0; x.n = 0;
		return n_nodes++;
	}
	// index of the last key <= pos in the node (-1: none); *r = sign(pos - that key's successor rule) as __kb_getp_aux
	template <class S> __device__ __forceinline__ int locate(const S &L, const GenNode &x, i64 pos, int *r) const
	{
		int begin = 0, end = x.n;
		if (x.n == 0) return -1;
		while (begin < end) {
			const int mid = (begin + end) >> 1;
			if (st_pos(L, x.key[mid]) < pos) begin = mid + 1;
			else end = mid;
		}
		if (begin == x.n) { *r = 1; return x.n - 1; }
		const i64 kp = st_pos(L, x.key[begin]);
		*r = (kp < pos) - (pos < kp);
		if (*r < 0) --begin;
		return begin;
	}
	template <class S> __device__ __forceinline__ int lower(const S &L, i64 pos) const
	{
		if (n == 0) return -1;
		int low = -1, r = 0, xi = root;
		for (;;) {
			const GenNode &x = L.nodes[xi];
			const int i = locate(L, x, pos, &r);
			if (i >= 0 && r == 0) return x.key[i];
			if (i >= 0) low = x.key[i];
			if (!x.internal) return low;
			xi = x.child[i + 1];
		}
	}
	template <class S> __device__ __forceinline__ void split(const S &L, int xi, int i, int yi)
	{
		const int zi = make(L);
		if (full) return;
		GenNode &x = L.nodes[xi], &y = L.nodes[yi], &z = L.nodes[zi];
		z.internal = y.internal;
		z.n = 4;
		for (int k = 0; k < 4; ++k) z.key[k] = y.key[5 + k];
		if (y.internal) for (int k = 0; k < 5; ++k) z.child[k] = y.child[5 + k];
		y.n = 4;
		for (int k = x.n; k > i; --k) x.child[k + 1] = x.child[k];
		x.child[i + 1] = (uint8_t)zi;
		for (int k = x.n - 1; k >= i; --k) x.key[k + 1] = x.key[k];
		x.key[i] = y.key[4];
		++x.n;
	}
	template <class S> __device__ __forceinline__ bool put(const S &L, int id, i64 pos)
	{
		if (n == S::MAXCH) return false;
		if (n_nodes == 0) root = make(L);
		int r = root, rr = 0;
		if (L.nodes[r].n == 9) {
			const int si = make(L);
			if (full) return false;
			root = si; L.nodes[si].internal = 1; L.nodes[si].n = 0;
			L.nodes[si].child[0] = (uint8_t)r;
			split(L, si, 0, r);
			if (full) return false;
			r = si;
		}
		int xi = r;
		while (L.nodes[xi].internal) {
			int i = locate(L, L.nodes[xi], pos, &rr) + 1;
			if (L.nodes[L.nodes[xi].child[i]].n == 9) {
				split(L, xi, i, L.nodes[xi].child[i]);
				if (full) return false;
				if (st_pos(L, L.nodes[xi].key[i]) < pos) ++i;
			}
			xi = L.nodes[xi].child[i];
		}
		GenNode &x = L.nodes[xi];
		const int i = locate(L, x, pos, &rr);
		for (int k = x.n - 1; k > i; --k) x.key[k + 1] = x.key[k];
		x.key[i + 1] = (uint8_t)id;
		++x.n;
		++n;
		return true;
	}
	template <class S> __device__ __forceinline__ void in_order(const S &L) const   // chain ids in key order into ord[]
	{
		if (n == 0) return;
		int sn[8], si[8], sp = 0, out = 0;   // node, children already entered (a tree of <= 255 keys with >= 4 per node is at most 4 levels deep)
		sn[0] = root; si[0] = 0;
		while (sp >= 0) {
			const GenNode &x = L.nodes[sn[sp]];
			if (!x.internal) {
				for (int k = 0; k < x.n; ++k) L.ord(out++) = x.key[k];
				--sp;
				continue;
			}
			const int i = si[sp];
			if (i > 0 && i <= x.n) L.ord(out++) = x.key[i - 1];   // back from child i - 1: its separator key
			if (i <= x.n) { si[sp] = i + 1; ++sp; sn[sp] = x.child[i]; si[sp] = 0; }
			else --sp;
		}
	}
};

__device__ __forceinline__ int ck_pos2rid(const i64 *__restrict__ ann_off, int n_seqs, i64 l_pac, i64 pos_f)
{
	if (pos_f >= l_pac) return -1;
	int left = 0, mid = 0, right = n_seqs;
	while (left < right) {   // src/bntseq.c:349-363
		mid = (left + right) >> 1;
		if (pos_f >= ann_off[mid]) {
			if (mid == n_seqs - 1 || pos_f < ann_off[mid + 1]) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}
__device__ __forceinline__ i64 ck_depos(i64 l_pac, i64 pos) { return pos >= l_pac ? (l_pac << 1) - 1 - pos : pos; }

// the unstable sort of mem_chain_flt for any n: ks_introsort (src/ksort.h:176-226) on the chain ids in ord[0, n), "less" =
// heavier first — median-of-three quicksort with an explicit stack, ranges of <= 16 left to one final insertion sort, comb
// sort when the depth budget runs out (sortutil.h is the host's statement of the same)
template <class S, class LT>
__device__ __forceinline__ void ck_insertion(const S &L, int s, int t, LT lt)   // [s, t)
{
	for (int i = s + 1; i < t; ++i)
		for (int j = i; j > s && lt(L.ord(j), L.ord(j - 1)); --j) { const uint8_t x = L.ord(j); L.ord(j) = L.ord(j - 1); L.ord(j - 1) = x; }
}
template <class S, class LT>
__device__ __forceinline__ void ck_comb(const S &L, int a, int n, LT lt)
{
	const double shrink = 1.2473309501039786540366528676643;
	int gap = n;
	bool swapped;
	do {
		if (gap > 2) {
			gap = (int)((double)gap / shrink);
			if (gap == 9 || gap == 10) gap = 11;
		}
		swapped = false;
		for (int i = a; i < a + n - gap; ++i) {
			const int j = i + gap;
			if (lt(L.ord(j), L.ord(i))) { const uint8_t x = L.ord(i); L.ord(i) = L.ord(j); L.ord(j) = x; swapped = true; }
		}
	} while (swapped || gap > 2);
	if (gap != 1) ck_insertion(L, a, a + n, lt);
}
template <class S, class LT>
__device__ __forceinline__ void ck_introsort(const S &L, int n, LT lt)
{
	if (n < 2) return;
	if (n == 2) {
		if (lt(L.ord(1), L.ord(0))) { const uint8_t x = L.ord(0); L.ord(0) = L.ord(1); L.ord(1) = x; }
		return;
	}
	int d = 2;
	while ((1 << d) < n) ++d;
	int fs[16], ft[16], fd[16], sp = 0;   // only ranges of more than 16 elements are pushed, the smaller side is worked first
	int s = 0, t = n - 1;
	d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) {
				ck_comb(L, s, t - s + 1, lt);
				t = s;
				continue;
			}
			int i = s, j = t, k = i + ((j - i) >> 1) + 1;
			if (lt(L.ord(k), L.ord(i))) { if (lt(L.ord(k), L.ord(j))) k = j; }
			else k = lt(L.ord(j), L.ord(i)) ? i : j;
			const uint8_t pivot = L.ord(k);
			if (k != t) { const uint8_t x = L.ord(k); L.ord(k) = L.ord(t); L.ord(t) = x; }
			for (;;) {
				do ++i; while (lt(L.ord(i), pivot));
				do --j; while (i <= j && lt(pivot, L.ord(j)));
				if (j <= i) break;
				const uint8_t x = L.ord(i); L.ord(i) = L.ord(j); L.ord(j) = x;
			}
			{ const uint8_t x = L.ord(i); L.ord(i) = L.ord(t); L.ord(t) = x; }
			if (i - s > t - i) {
				if (i - s > 16) { fs[sp] = s; ft[sp] = i - 1; fd[sp] = d; ++sp; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { fs[sp] = i + 1; ft[sp] = t; fd[sp] = d; ++sp; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (sp == 0) { ck_insertion(L, 0, n, lt); return; }
			--sp;
			s = fs[sp]; t = ft[sp]; d = fd[sp];
		}
	}
}

// mem_chain + mem_chain_flt + emission for ONE read (one lane).  Returns the number of kept chains, or -1: host path.
template <class S, class MAP>
__device__ __forceinline__ int chain_read(const S &L, const ChainParams &P, int rd, int ns, int lq, i64 so, const int *__restrict__ l_rep,
                                          const unsigned long long *__restrict__ sa, const int32_t *__restrict__ qbl, const i64 *__restrict__ ann_off,
                                          const uint8_t *__restrict__ ann_alt, int n_seqs, const int *__restrict__ gap, DevChain *__restrict__ chains,
                                          DevSeed *__restrict__ seeds, unsigned int *__restrict__ srt)
{
	const i64 l_pac = P.l_pac;
	auto S_R = [&](int k) -> i64 { return (i64)sa[so + k]; };
	auto S_Q = [&](int k) -> int { return qbl[2 * (so + k)]; };
	auto S_L = [&](int k) -> int { return qbl[2 * (so + k) + 1]; };
	// the seeds of chain `id` in arrival order
	auto members = [&](int id, auto fn) {
		if constexpr (S::GENERAL) {
			for (int k = (int)L.f(F_HEAD, id); k != 255; k = L.next(k)) fn(k);
		} else {
			for (int k = 0; k < ns; ++k)
				if (L.cid(k) == id) fn(k);
		}
	};

	// ---------------- mem_chain: seeds into the ordered map ----------------
	MAP map;
	int n_ch = 0;
	i64 c_lo = 0, c_hi = -1;
	int c_rid = -1;
	for (int k = 0; k < ns; ++k) {
		const i64 rb = S_R(k);
		const int qb = S_Q(k), len = S_L(k);
		L.cid(k) = 255;
		int rid;
		if (rb >= c_lo && rb + len <= c_hi && len > 0) rid = c_rid;
		else {   // bns_intv2rid, src/bntseq.c:365-376
			if (rb < l_pac && rb + len > l_pac) rid = -2;
			else {
				const int rid_b = ck_pos2rid(ann_off, n_seqs, l_pac, ck_depos(l_pac, rb));
				const int rid_e = len > 0 ? ck_pos2rid(ann_off, n_seqs, l_pac, ck_depos(l_pac, rb + len - 1)) : rid_b;
				rid = rid_b == rid_e ? rid_b : -1;
			}
			if (rid >= 0) {
				const i64 o = ann_off[rid], l = ann_off[rid + 1] - o;
				if (rb < l_pac) { c_lo = o; c_hi = o + l; }
				else { c_lo = (l_pac << 1) - o - l; c_hi = (l_pac << 1) - o; }
				c_rid = rid;
			}
		}
		if (rid < 0) continue;   // bridges two contigs or the strand boundary
		const int id0 = map.lower(L, rb);   // closest chain at or before the seed (kb_intervalp)
		bool merged = false;
		if (id0 >= 0) {   // test_and_merge
			const int id = id0;
			const i64 first_r = st_pos(L, id), last_r = st_last_r(L, id);
			const int first_q = (int)L.f(F_FIRST_Q, id), last_q = (int)L.f(F_LAST_Q, id), last_len = (int)L.f(F_LAST_LEN, id);
			const int qend = last_q + last_len;
			const i64 rend = last_r + last_len;
			if ((int)L.f(F_RID, id) == rid) {
				if (qb >= first_q && qb + len <= qend && rb >= first_r && rb + len <= rend) merged = true;   // contained: dropped
				else if (!((last_r < l_pac || first_r < l_pac) && rb >= l_pac)) {                            // never chain across strands
					const i64 x = qb - last_q, y = rb - last_r;
					if (y >= 0 && x - y <= P.w && y - x <= P.w && x - last_len < P.max_chain_gap && y - last_len < P.max_chain_gap) {
						L.f(F_LAST_R_LO, id) = (uint32_t)rb; L.f(F_LAST_R_HI, id) = (uint32_t)((unsigned long long)rb >> 32);
						L.f(F_LAST_Q, id) = (uint32_t)qb; L.f(F_LAST_LEN, id) = (uint32_t)len;
						L.f(F_N, id) += 1;
						L.cid(k) = (uint8_t)id;
						if constexpr (S::GENERAL) { L.next((int)L.f(F_TAIL, id)) = (uint8_t)k; L.next(k) = 255; L.f(F_TAIL, id) = (uint32_t)k; }
						merged = true;
					}
				}
			}
		}
		if (merged) continue;
		const int id = n_ch;
		if (id >= S::MAXCH) return -1;
		// (the record first: the map compares positions through it)
		L.f(F_POS_LO, id) = (uint32_t)rb; L.f(F_POS_HI, id) = (uint32_t)((unsigned long long)rb >> 32);
		L.f(F_FIRST_Q, id) = (uint32_t)qb;
		L.f(F_LAST_R_LO, id) = (uint32_t)rb; L.f(F_LAST_R_HI, id) = (uint32_t)((unsigned long long)rb >> 32);
		L.f(F_LAST_Q, id) = (uint32_t)qb; L.f(F_LAST_LEN, id) = (uint32_t)len;
		L.f(F_RID, id) = (uint32_t)rid; L.f(F_N, id) = 1;
		if constexpr (S::GENERAL) { L.f(F_HEAD, id) = L.f(F_TAIL, id) = (uint32_t)k; L.next(k) = 255; }
		if (!map.put(L, id, rb)) return -1;   // more chains (or tree nodes) than this launch keeps: host path
		L.cid(k) = (uint8_t)id;
		++n_ch;
	}
	map.in_order(L);

	// ---------------- mem_chain_flt ----------------
	int n = 0;
	for (int t = 0; t < n_ch; ++t) {
		const int id = L.ord(t);
		// mem_chain_weight: seed coverage of the query, of the reference, the smaller one
		i64 end = 0;
		int w = 0, wq;
		members(id, [&](int k) {
			const int qb = S_Q(k), len = S_L(k);
			if (qb >= end) w += len;
			else if (qb + len > end) w += (int)(qb + len - end);
			end = end > qb + len ? end : qb + len;
		});
		wq = w; w = 0; end = 0;
		members(id, [&](int k) {
			const i64 rb = S_R(k);
			const int len = S_L(k);
			if (rb >= end) w += len;
			else if (rb + len > end) w += (int)(rb + len - end);
			end = end > rb + len ? end : rb + len;
		});
		w = w < wq ? w : wq;
		w = w < 1 << 30 ? w : (1 << 30) - 1;
		const uint32_t w29 = (uint32_t)w & 0x1fffffffu;
		L.f(F_W, id) = w29; L.f(F_KEPT, id) = 0; L.f(F_FIRSTOV, id) = 0xffffffffu;
		if ((int)w29 < P.min_chain_weight) continue;
		L.ord(n++) = (uint8_t)id;
	}
	if (n == 0) return 0;
	ck_introsort(L, n, [&](int a_id, int b_id) -> bool { return (int)L.f(F_W, a_id) > (int)L.f(F_W, b_id); });   // "less" of the descending sort
	// pairwise overlap marking; tmp holds the positions (in ord) of the chains kept so far
	auto BEG = [&](int t) -> int { return (int)L.f(F_FIRST_Q, L.ord(t)); };
	auto END = [&](int t) -> int { const int id = L.ord(t); return (int)L.f(F_LAST_Q, id) + (int)L.f(F_LAST_LEN, id); };
	auto ALT = [&](int t) -> int { return ann_alt[L.f(F_RID, L.ord(t))]; };
	int n_kept = 0;
	L.f(F_KEPT, L.ord(0)) = 3;
	L.tmp(n_kept++) = 0;
	for (int i = 1; i < n; ++i) {
		bool large_ovlp = false;
		int kk;
		for (kk = 0; kk < n_kept; ++kk) {
			const int j = L.tmp(kk);
			const int b_max = BEG(j) > BEG(i) ? BEG(j) : BEG(i);
			const int e_min = END(j) < END(i) ? END(j) : END(i);
			if (e_min > b_max && (!ALT(j) || ALT(i))) {
				const int li_ = END(i) - BEG(i), lj_ = END(j) - BEG(j);
				const int min_l = li_ < lj_ ? li_ : lj_;
				if ((float)(e_min - b_max) >= (float)min_l * P.mask_level && min_l < P.max_chain_gap) {
					large_ovlp = true;
					if (L.f(F_FIRSTOV, L.ord(j)) == 0xffffffffu) L.f(F_FIRSTOV, L.ord(j)) = (uint32_t)i;
					const int wi = (int)L.f(F_W, L.ord(i)), wj = (int)L.f(F_W, L.ord(j));
					if ((float)wi < (float)wj * P.drop_ratio && wj - wi >= P.min_seed_len << 1) break;
				}
			}
		}
		if (kk == n_kept) {
			L.tmp(n_kept++) = (uint8_t)i;
			L.f(F_KEPT, L.ord(i)) = large_ovlp ? 2 : 3;
		}
	}
	for (int kk = 0; kk < n_kept; ++kk) {
		const uint32_t fo = L.f(F_FIRSTOV, L.ord(L.tmp(kk)));
		if (fo != 0xffffffffu) L.f(F_KEPT, L.ord(fo)) = 1;
	}
	{
		int i = 0, cnt = 0;
		for (; i < n; ++i) {   // at most max_chain_extend chains with kept = 1 or 2 are extended
			const uint32_t kp = L.f(F_KEPT, L.ord(i));
			if (kp == 0 || kp == 3) continue;
			if (++cnt >= P.max_chain_extend) break;
		}
		for (; i < n; ++i)
			if (L.f(F_KEPT, L.ord(i)) < 3) L.f(F_KEPT, L.ord(i)) = 0;
	}

	// ---------------- emit the kept chains and their seeds ----------------
	const float frac_rep = (float)l_rep[rd] / (float)lq;
	int n_out = 0;
	i64 cursor = so;
	for (int t = 0; t < n; ++t) {
		const int id = L.ord(t);
		if (L.f(F_KEPT, id) == 0) continue;
		// members in arrival order, then sorted by (length, arrival index): keys are distinct
		int cs = 0;
		members(id, [&](int k) { L.tmp(cs++) = (uint8_t)k; });
		// tmp was also the kept list: it is no longer needed at this point
		for (int a = 1; a < cs; ++a) {
			const uint8_t v = L.tmp(a);
			const int lv = S_L(v);
			int b = a;
			while (b > 0 && S_L(L.tmp(b - 1)) > lv) { L.tmp(b) = L.tmp(b - 1); --b; }   // stable: equal lengths keep arrival order
			L.tmp(b) = v;
		}
		const int rid = (int)L.f(F_RID, id);
		const i64 first_r = st_pos(L, id);
		i64 lo = l_pac << 1, hi = 0;
		for (int a = 0; a < cs; ++a) {
			const int k = L.tmp(a);
			DevSeed ds;
			ds.rbeg = S_R(k); ds.qbeg = S_Q(k); ds.len = S_L(k);
			seeds[cursor + a] = ds;
			srt[cursor + a] = (unsigned int)a;
			// widest reference span any seed of the chain could reach (src/bwamem.c:642-658)
			const i64 b = ds.rbeg - (ds.qbeg + gap[ds.qbeg]);
			const int tail = lq - ds.qbeg - ds.len;
			const i64 e = ds.rbeg + ds.len + (tail + gap[tail]);
			lo = b < lo ? b : lo;
			hi = e > hi ? e : hi;
		}
		DevChain d;
		i64 fb = ann_off[rid], fe = ann_off[rid + 1];
		if (first_r >= l_pac) { const i64 tt = fb; fb = (l_pac << 1) - fe; fe = (l_pac << 1) - tt; }
		d.far_beg = fb; d.far_end = fe;
		d.seed_beg = (int)cursor; d.n_seeds = cs; d.rid = rid; d.frac_rep = frac_rep;
		d.rmax0 = lo > 0 ? lo : 0;
		d.rmax1 = hi < l_pac << 1 ? hi : l_pac << 1;
		if (d.rmax0 < l_pac && l_pac < d.rmax1) {   // never cross the strand boundary
			if (first_r < l_pac) d.rmax1 = l_pac;
			else d.rmax0 = l_pac;
		}
		d.rmax0 = d.rmax0 > fb ? d.rmax0 : fb;       // bns_fetch_seq clamps to the contig
		d.rmax1 = d.rmax1 < fe ? d.rmax1 : fe;
		chains[so + n_out] = d;
		++n_out;
		cursor += cs;
	}
	return n_out;
}

// Three launches with growing LDS footprints, because a lane's working set decides how many waves a CU holds and the kernel
// lives on latency hiding: <16 seeds, 4 chains> takes every read first (15.5 KB per wave: 10 waves per CU; nine reads in ten stay
// here), <64, 9> (37 KB: 4 waves) retries what that declined, <255, 9> (62 KB) the reads with 65-255 seeds.
// MAXS / MAXCH: seeds / chains per read the instantiation has LDS for.  LO >= 0: a retry — only reads an earlier launch declined
// (n_chains = -1) with more than LO seeds.
template <int MAXS, int MAXCH, int LO>
__global__ void __launch_bounds__(64)
chain_kernel(ChainParams P, int n_reads, const int *__restrict__ lens, const int *__restrict__ n_seeds, const int *__restrict__ l_rep,
             const i64 *__restrict__ seed_off, const unsigned long long *__restrict__ sa, const int32_t *__restrict__ qbl,
             const i64 *__restrict__ ann_off, const uint8_t *__restrict__ ann_alt, int n_seqs, const int *__restrict__ tab, int tab_stride,
             DevChain *__restrict__ chains, DevSeed *__restrict__ seeds, unsigned int *__restrict__ srt, int *__restrict__ n_chains,
             const int *__restrict__ list, const unsigned int *__restrict__ list_n)
{
	extern __shared__ uint32_t lds_raw[];
	StoreLds<MAXCH> L;
	L.lane = threadIdx.x;
	L.tab = lds_raw;
	L.cid_ = (uint8_t *)(lds_raw + F_NFIELDS * MAXCH * 64);
	L.ord_ = L.cid_ + (MAXS + 1) * 64;
	L.tmp_ = L.ord_ + 16 * 64;
	// a retry walks the list of the reads it is for (chain_pick_kernel), so that its waves are full: handed the whole batch, nearly
	// every wave would hold one or two such reads and run the whole serial path for them
	int rd = blockIdx.x * 64 + threadIdx.x;
	if (list) { if (rd >= (int)*list_n) return; rd = list[rd]; }
	if (rd >= n_reads) return;
	const int ns = n_seeds[rd], lq = lens[rd];
	const int *gap = tab, *noflt = tab + 5 * tab_stride;
	if (LO >= 0) { if (n_chains[rd] != -1 || ns <= LO) return; }
	else if (ns == 0) { n_chains[rd] = 0; return; }
	if (ns > MAXS || !noflt[lq]) { n_chains[rd] = -1; return; }
	n_chains[rd] = chain_read<StoreLds<MAXCH>, MapArray>(L, P, rd, ns, lq, seed_off[rd], l_rep, sa, qbl, ann_off, ann_alt, n_seqs, gap, chains, seeds, srt);
}

// chain_pick_kernel lists the reads a later launch is for: declined so far (n_chains = -1), lo < seeds <= hi, at most cap of them.
// Last launch: the reads all single-node launches declined (more than 9 chains) are chained with the reference's B-tree, a lane per
// read, the working set in an HBM slice.
__global__ void __launch_bounds__(256)
chain_pick_kernel(int n_reads, const int *__restrict__ lens, const int *__restrict__ n_seeds, const int *__restrict__ n_chains,
                  const int *__restrict__ noflt, int lo, int hi, int cap, int *__restrict__ list, unsigned int *__restrict__ count)
{
	const int rd = blockIdx.x * 256 + threadIdx.x;
	const bool mine = rd < n_reads && n_chains[rd] == -1 && n_seeds[rd] > lo && n_seeds[rd] <= hi && noflt[lens[rd]] != 0;
	const unsigned long long m = __ballot(mine);
	if (!m) return;
	const int lane = threadIdx.x & 63, lead = __ffsll((long long)m) - 1;
	unsigned int base = 0;
	if (lane == lead) base = atomicAdd(count, (unsigned int)__popcll(m));
	base = __shfl(base, lead);
	const unsigned int at = base + (unsigned int)__popcll(m & ((1ull << lane) - 1));
	if (mine && at < (unsigned int)cap) list[at] = rd;
}

__global__ void __launch_bounds__(64)
chain_general_kernel(ChainParams P, const int *__restrict__ list, const unsigned int *__restrict__ count, int cap, const int *__restrict__ lens,
                     const int *__restrict__ n_seeds, const int *__restrict__ l_rep, const i64 *__restrict__ seed_off,
                     const unsigned long long *__restrict__ sa, const int32_t *__restrict__ qbl, const i64 *__restrict__ ann_off,
                     const uint8_t *__restrict__ ann_alt, int n_seqs, const int *__restrict__ tab, uint8_t *__restrict__ scratch,
                     DevChain *__restrict__ chains, DevSeed *__restrict__ seeds, unsigned int *__restrict__ srt, int *__restrict__ n_chains)
{
	const int t = blockIdx.x * 64 + threadIdx.x;
	const int n_list = (int)(*count < (unsigned int)cap ? *count : (unsigned int)cap);
	if (t >= n_list) return;
	const int rd = list[t];
	uint8_t *base = scratch + (size_t)t * CK_GEN_BYTES;
	StoreGen L;
	L.tab = (uint32_t *)base;
	L.cid_ = base + (size_t)F_NFIELDS_GEN * CK_MAXCH_GEN * 4;
	L.ord_ = L.cid_ + 256; L.tmp_ = L.ord_ + 256; L.next_ = L.tmp_ + 256;
	L.nodes = (GenNode *)(L.next_ + 256);
	const int r = chain_read<StoreGen, MapBtree>(L, P, rd, n_seeds[rd], lens[rd], seed_off[rd], l_rep, sa, qbl, ann_off, ann_alt, n_seqs, tab, chains,
	                                              seeds, srt);
	n_chains[rd] = r < 0 ? -2 : r;   // (-2: declined for good — the next round of this kernel must not pick the read again)
}

// ---------------------------------------------------------------------------------------------------------------------
// chain_heavy_kernel (round 4): the reads with MORE than 255 seeds — reads of high-copy repeats: max_occ (500) occurrences
// of each of several intervals, hundreds to a few thousand seeds and nearly as many chains — one WAVEFRONT per read.
//   * mem_chain (src/bwamem.c:251-315) is sequential in the seeds (every seed meets the ordered map as the earlier ones left it), but
//     what a seed does to the map is a job for 64 lanes: the map is a sorted array in LDS, the closest chain at or before the seed
//     is one or two ballots, a new chain makes room with all lanes (while the chain positions of a read are distinct this is the
//     reference's B-tree to the letter; a read with two chains at one position goes to the host's restatement of the tree);
//   * mem_chain_weight (:213-237) runs a chain per lane;
//   * mem_chain_flt's unstable sort (:341) is ks_introsort on (weight, chain) words in LDS, lane 0;
//   * its pairwise pass (:346-367) — every chain against every chain kept so far, n^2 / 2 tests for a repeat whose chains all
//     overlap and all survive: the host spent 300 k cycles per such read here — runs 64 kept chains per step, the columns of the
//     kept chains (query span, weight, ALT flag, first shadowed chain) in the LDS the tree no longer needs, a break found by ballot;
//   * emission as chain_read above, a kept chain per lane, output positions by a wave prefix sum.
// Persistent waves take reads from a list (chain_pick_kernel) through a counter.  CAP = seeds (hence chains) a read may have in
// the instantiation: 256 (7.5 KB of LDS per wave: the reads of up to 255 seeds with more than 9 chains), 1024 (26 KB) and 4096 (98 KB);
// more seeds than that, or reads long enough for mem_flt_chained_seeds to act, stay with the host.
// ---------------------------------------------------------------------------------------------------------------------
#define HV_NONE 0xFFFFu
// LDS while the seeds are walked: the ordered map (chain positions in ascending order, 8 bytes, and whose they are, 2) and per chain
// the last seed's offset from its position (4), first_q / last_q / last_len / tail / nmem (5 x 2) — everything test_and_merge reads;
// afterwards the same bytes hold the filter's columns
__host__ __device__ constexpr size_t hv_lds_bytes(int cap) { return (size_t)(cap + 64) * 24; }   // (+ 64: the map's shifts write one entry past its end)
__host__ __device__ constexpr size_t hv_scratch_bytes(int cap) { return (size_t)cap * (4 + 7 * 2 + 8 + 4) + 512; }
struct HvStore {   // what only the later phases read (HBM, the wave's slice): contig and first seed of every chain, the seeds' links, the tree order
	uint32_t *rid;
	uint16_t *head, *cid, *next, *ord;
	// LDS
	uint32_t *last_off;
	uint16_t *first_q, *last_q, *last_len, *tail, *nmem;
};
// ks_introsort (src/ksort.h:176-226) of n words, "less" = the heavier chain first (the weight is the word's upper half)
__device__ __forceinline__ bool hv_lt(uint32_t a, uint32_t b) { return (a >> 16) > (b >> 16); }
__device__ __forceinline__ void hv_insertion(uint32_t *a, int s, int t)
{
	for (int i = s + 1; i < t; ++i)
		for (int j = i; j > s && hv_lt(a[j], a[j - 1]); --j) { const uint32_t x = a[j]; a[j] = a[j - 1]; a[j - 1] = x; }
}
__device__ __forceinline__ void hv_comb(uint32_t *a, int s0, int n)
{
	const double shrink = 1.2473309501039786540366528676643;
	int gap = n;
	bool swapped;
	do {
		if (gap > 2) {
			gap = (int)((double)gap / shrink);
			if (gap == 9 || gap == 10) gap = 11;
		}
		swapped = false;
		for (int i = s0; i < s0 + n - gap; ++i) {
			const int j = i + gap;
			if (hv_lt(a[j], a[i])) { const uint32_t x = a[i]; a[i] = a[j]; a[j] = x; swapped = true; }
		}
	} while (swapped || gap > 2);
	if (gap != 1) hv_insertion(a, s0, s0 + n);
}
__device__ __forceinline__ void hv_introsort(uint32_t *a, int n)
{
	if (n < 2) return;
	if (n == 2) {
		if (hv_lt(a[1], a[0])) { const uint32_t x = a[0]; a[0] = a[1]; a[1] = x; }
		return;
	}
	int d = 2;
	while ((1 << d) < n) ++d;
	int fs[32], ft[32], fd[32], sp = 0;
	int s = 0, t = n - 1;
	d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) {
				hv_comb(a, s, t - s + 1);
				t = s;
				continue;
			}
			int i = s, j = t, k = i + ((j - i) >> 1) + 1;
			if (hv_lt(a[k], a[i])) { if (hv_lt(a[k], a[j])) k = j; }
			else k = hv_lt(a[j], a[i]) ? i : j;
			const uint32_t pivot = a[k];
			if (k != t) { const uint32_t x = a[k]; a[k] = a[t]; a[t] = x; }
			for (;;) {
				do ++i; while (hv_lt(a[i], pivot));
				do --j; while (i <= j && hv_lt(pivot, a[j]));
				if (j <= i) break;
				const uint32_t x = a[i]; a[i] = a[j]; a[j] = x;
			}
			{ const uint32_t x = a[i]; a[i] = a[t]; a[t] = x; }
			if (i - s > t - i) {
				if (i - s > 16) { fs[sp] = s; ft[sp] = i - 1; fd[sp] = d; ++sp; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { fs[sp] = i + 1; ft[sp] = t; fd[sp] = d; ++sp; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (sp == 0) { hv_insertion(a, 0, n); return; }
			--sp;
			s = fs[sp]; t = ft[sp]; d = fd[sp];
		}
	}
}
__device__ __forceinline__ void hv_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int hv_bcast(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int CAP>
__global__ void __launch_bounds__(64)
chain_heavy_kernel(ChainParams P, const int *__restrict__ list, const unsigned int *__restrict__ list_n, unsigned int *work, int list_cap,
                   const int *__restrict__ lens, const int *__restrict__ n_seeds, const int *__restrict__ l_rep, const i64 *__restrict__ seed_off,
                   const unsigned long long *__restrict__ sa, const int32_t *__restrict__ qbl, const i64 *__restrict__ ann_off,
                   const uint8_t *__restrict__ ann_alt, int n_seqs, const int *__restrict__ gap, uint8_t *__restrict__ scratch,
                   DevChain *__restrict__ chains, DevSeed *__restrict__ seeds, unsigned int *__restrict__ srt, int *__restrict__ n_chains)
{
	extern __shared__ uint8_t hv_lds[];
	const int lane = threadIdx.x;
	// phase A: the tree
	i64 *spos = (i64 *)hv_lds;                                    // [CAP + 64] chain positions in ascending order
	uint16_t *sid = (uint16_t *)(hv_lds + (size_t)(CAP + 64) * 22);   // [CAP + 64] ... and whose they are
	// phases C-D (the tree is dead by then): sort words, then the columns of the chains in sorted order and of the kept list
	uint32_t *skey = (uint32_t *)hv_lds;                 // [CAP]
	uint16_t *cbeg = (uint16_t *)(skey + CAP);           // [CAP] each
	uint16_t *cend = cbeg + CAP, *kidx = cend + CAP, *kfirst = kidx + CAP;
	uint8_t *calt = (uint8_t *)(kfirst + CAP), *ckept = calt + CAP;   // [CAP] each: 4 + 8 + 2 = 14 bytes per chain <= the 20.5 of the tree
	uint8_t *sb = scratch + (size_t)blockIdx.x * hv_scratch_bytes(CAP);
	HvStore S;
	S.rid = (uint32_t *)sb;
	S.head = (uint16_t *)(S.rid + CAP); S.cid = S.head + CAP; S.next = S.cid + CAP; S.ord = S.next + CAP;
	S.last_off = (uint32_t *)(hv_lds + (size_t)(CAP + 64) * 8);
	S.first_q = (uint16_t *)(S.last_off + CAP); S.last_q = S.first_q + CAP; S.last_len = S.last_q + CAP; S.tail = S.last_len + CAP; S.nmem = S.tail + CAP;
	// (the filter reads first_q / last_q / last_len / nmem of the chains after the tree is gone: they are copied to the wave's HBM slice
	// before the LDS is reused — hv_spill below)
	uint16_t *g_first_q = S.ord + CAP, *g_end_q = g_first_q + CAP, *g_nmem = g_end_q + CAP;
	i64 *g_clo = (i64 *)(sb + (((size_t)CAP * 18 + 15) & ~(size_t)15));   // per seed: start of its contig on its strand (-1: no contig), its contig
	int *g_rid = (int *)(g_clo + CAP);
	const i64 l_pac = P.l_pac;
	const int n_list = (int)(*list_n < (unsigned int)list_cap ? *list_n : (unsigned int)list_cap);
	for (;;) {
		int t = 0;
		if (lane == 0) t = (int)atomicAdd(work, 1u);
		t = hv_bcast(t);
		if (t >= n_list) break;
		const int rd = hv_bcast(list[t]);
		const int ns = hv_bcast(n_seeds[rd]), lq = hv_bcast(lens[rd]);
		const i64 so = seed_off[rd];
		auto S_R = [&](int k) -> i64 { return (i64)sa[so + k]; };
		auto S_Q = [&](int k) -> int { return qbl[2 * (so + k)]; };
		auto S_L = [&](int k) -> int { return qbl[2 * (so + k) + 1]; };
		if (ns > CAP) continue;   // (the list only holds reads that fit; n_chains stays -1: host)
		// ---------------- mem_chain: the whole wave per seed ----------------
		// The ordered map of src/bwamem.c:263 is only ever asked for the closest chain at or before a position and walked in order at
		// the end; while all chain positions of the read are DISTINCT, a sorted array answers both exactly like the reference's B-tree
		// (whose shape only shows when keys are equal).  The array lives in LDS; the wave finds the predecessor with one or two ballots
		// and makes room for a new chain with all lanes.  A new chain at the position of an existing one ends the attempt: such a read
		// (rare: two seeds at one reference position more than w apart on the read) is chained by the host's restatement of the tree.
		int n_ch = 0, ok = 1;
		{
			// the contig of every seed first (bns_intv2rid, src/bntseq.c:365-376: two binary searches over the contig table in HBM), a
			// seed per lane — the seeds of a repeat hop from contig to contig, and ten dependent loads per seed in the walk below were
			// most of its time: the walk gets each seed's contig and where that contig starts on the seed's strand along with the seed
			for (int k = lane; k < ns; k += 64) {
				const i64 rb = S_R(k);
				const int len = S_L(k);
				int rid;
				if (rb < l_pac && rb + len > l_pac) rid = -2;
				else {
					const int rid_b = ck_pos2rid(ann_off, n_seqs, l_pac, ck_depos(l_pac, rb));
					const int rid_e = len > 0 ? ck_pos2rid(ann_off, n_seqs, l_pac, ck_depos(l_pac, rb + len - 1)) : rid_b;
					rid = rid_b == rid_e ? rid_b : -1;
				}
				i64 lo = -1;
				if (rid >= 0) lo = rb < l_pac ? ann_off[rid] : (l_pac << 1) - ann_off[rid + 1];
				g_rid[k] = rid; g_clo[k] = lo;
			}
			hv_sync();
			i64 nx_rb = S_R(0), nx_clo = g_clo[0];   // (the next seed is fetched while this one meets the map: the walk itself only touches LDS)
			int nx_qb = S_Q(0), nx_len = S_L(0), nx_rid = g_rid[0];
			for (int k = 0; k < ns; ++k) {
				const i64 rb = nx_rb, c_lo = nx_clo;
				const int qb = nx_qb, len = nx_len, rid = nx_rid;
				if (k + 1 < ns) { nx_rb = S_R(k + 1); nx_qb = S_Q(k + 1); nx_len = S_L(k + 1); nx_rid = g_rid[k + 1]; nx_clo = g_clo[k + 1]; }
				if (rid < 0) { if (lane == 0) S.cid[k] = HV_NONE; continue; }
				// chains at or before rb: a prefix of the sorted array -> its length
				int cnt;
				if (n_ch <= 64) cnt = __popcll(__ballot(lane < n_ch && spos[lane] <= rb));
				else {
					const int stride = (n_ch + 63) >> 6;
					const int at = lane * stride;
					const int blk = __popcll(__ballot(at < n_ch && spos[at] <= rb));   // samples 0, stride, 2 stride, ... at or before rb
					if (blk == 0) cnt = 0;
					else {
						const int base = (blk - 1) * stride, e = base + lane;       // the answer lies in [base, base + stride)
						cnt = base + __popcll(__ballot(lane < stride && e < n_ch && spos[e] <= rb));
					}
				}
				bool merged = false;
				int id0 = -1;
				if (cnt > 0) {
					id0 = sid[cnt - 1];
					const i64 first_r = spos[cnt - 1];
					// (a chain lies in the contig of its first seed, hence of its position: the chain found is in this seed's contig — and on its
					// strand — exactly when its position is not before the contig's start)
					if (first_r >= c_lo) {   // test_and_merge, src/bwamem.c:190-211
						const i64 last_r = first_r + (i64)S.last_off[id0];
						const int first_q = S.first_q[id0], last_q = S.last_q[id0], last_len = S.last_len[id0];
						const int qend = last_q + last_len;
						const i64 rend = last_r + last_len;
						if (qb >= first_q && qb + len <= qend && rb >= first_r && rb + len <= rend) { merged = true; if (lane == 0) S.cid[k] = HV_NONE; }
						else if (!((last_r < l_pac || first_r < l_pac) && rb >= l_pac)) {
							const i64 x = qb - last_q, y = rb - last_r;
							if (y >= 0 && x - y <= P.w && y - x <= P.w && x - last_len < P.max_chain_gap && y - last_len < P.max_chain_gap) {
								if (lane == 0) {
									S.last_off[id0] = (uint32_t)(rb - first_r); S.last_q[id0] = (uint16_t)qb; S.last_len[id0] = (uint16_t)len;
									S.nmem[id0] += 1;
									S.cid[k] = (uint16_t)id0;
									S.next[S.tail[id0]] = (uint16_t)k; S.next[k] = HV_NONE; S.tail[id0] = (uint16_t)k;
								}
								merged = true;
							}
						}
					}
				}
				if (!merged) {
					if (cnt > 0 && spos[cnt - 1] == rb) { ok = 0; break; }   // equal keys: the reference's tree decides, on the host
					const int id = n_ch;
					// room at position cnt: the entries behind it move up by one, the top 64 first
					for (int hi = n_ch; hi > cnt; hi -= 64) {
						const int e = hi - 1 - lane;
						i64 v = 0; uint16_t w = 0;
						if (e >= cnt) { v = spos[e]; w = sid[e]; }
						hv_sync();
						if (e >= cnt) { spos[e + 1] = v; sid[e + 1] = w; }
						hv_sync();
					}
					if (lane == 0) {
						spos[cnt] = rb; sid[cnt] = (uint16_t)id;
						S.last_off[id] = 0; S.rid[id] = (uint32_t)rid;
						S.first_q[id] = (uint16_t)qb; S.last_q[id] = (uint16_t)qb; S.last_len[id] = (uint16_t)len; S.nmem[id] = 1;
						S.head[id] = (uint16_t)k; S.tail[id] = (uint16_t)k; S.next[k] = HV_NONE;
						S.cid[k] = (uint16_t)id;
					}
					++n_ch;
				}
				hv_sync();
			}
			if (ok)
				for (int t0 = lane; t0 < n_ch; t0 += 64) S.ord[t0] = sid[t0];   // the chains in position order (kb_itr of src/bwamem.c:304)
		}
		ok = hv_bcast(ok); n_ch = hv_bcast(n_ch);
		hv_sync();
		if (!ok) { if (lane == 0) n_chains[rd] = -2; continue; }
		for (int id = lane; id < n_ch; id += 64) {   // hv_spill: query span and size of every chain, out of the LDS that is about to be reused
			g_first_q[id] = S.first_q[id]; g_end_q[id] = (uint16_t)(S.last_q[id] + S.last_len[id]); g_nmem[id] = S.nmem[id];
		}
		hv_sync();
		// ---------------- mem_chain_weight: a chain per lane; the chains that weigh enough, in tree order, as sort words ----------------
		int n = 0;
		for (int t0 = 0; t0 < n_ch; t0 += 64) {
			const int tt = t0 + lane;
			int w = 0, id = 0;
			if (tt < n_ch) {
				id = S.ord[tt];
				i64 end = 0;
				int wq;
				for (int k = S.head[id]; k != HV_NONE; k = S.next[k]) {
					const int qb = S_Q(k), len = S_L(k);
					if (qb >= end) w += len;
					else if (qb + len > end) w += (int)(qb + len - end);
					end = end > qb + len ? end : qb + len;
				}
				wq = w; w = 0; end = 0;
				for (int k = S.head[id]; k != HV_NONE; k = S.next[k]) {
					const i64 rb = S_R(k);
					const int len = S_L(k);
					if (rb >= end) w += len;
					else if (rb + len > end) w += (int)(rb + len - end);
					end = end > rb + len ? end : rb + len;
				}
				w = w < wq ? w : wq;
			}
			const bool pass = tt < n_ch && w >= P.min_chain_weight && w < 65536;
			if (__ballot(tt < n_ch && w >= 65536)) ok = 0;   // (a weight beyond the sort word: reads of more than 65 kbp never come here)
			const unsigned long long m = __ballot(pass);
			if (pass) skey[n + __popcll(m & ((1ull << lane) - 1))] = (uint32_t)w << 16 | (uint32_t)id;
			n += __popcll(m);
		}
		hv_sync();
		if (!ok) { if (lane == 0) n_chains[rd] = -2; continue; }
		if (n == 0) { if (lane == 0) n_chains[rd] = 0; continue; }
		if (lane == 0) hv_introsort(skey, n);
		hv_sync();
		// ---------------- mem_chain_flt's pairwise pass: 64 kept chains per step ----------------
		for (int t0 = 0; t0 < n; t0 += 64) {
			const int tt = t0 + lane;
			if (tt < n) {
				const int id = skey[tt] & 0xffff;
				cbeg[tt] = g_first_q[id];
				cend[tt] = g_end_q[id];
				calt[tt] = ann_alt[S.rid[id]];
				ckept[tt] = 0;
			}
		}
		if (lane == 0) { ckept[0] = 3; kidx[0] = 0; kfirst[0] = HV_NONE; }
		hv_sync();
		int nk = 1;
		for (int i = 1; i < n; ++i) {
			const int bi = cbeg[i], ei = cend[i], li = ei - bi, wi = (int)(skey[i] >> 16), alt_i = calt[i];
			bool large = false, broke = false;
			for (int k0 = 0; k0 < nk && !broke; k0 += 64) {
				const int kk = k0 + lane;
				bool sig = false, brk = false;
				if (kk < nk) {
					const int j = kidx[kk];
					const int bj = cbeg[j], ej = cend[j];
					const int b_max = bj > bi ? bj : bi, e_min = ej < ei ? ej : ei;
					if (e_min > b_max && (!calt[j] || alt_i)) {
						const int lj = ej - bj, min_l = li < lj ? li : lj;
						if ((float)(e_min - b_max) >= (float)min_l * P.mask_level && min_l < P.max_chain_gap) {
							sig = true;
							const int wj = (int)(skey[j] >> 16);
							brk = (float)wi < (float)wj * P.drop_ratio && wj - wi >= P.min_seed_len << 1;
						}
					}
				}
				unsigned long long sm = __ballot(sig);
				const unsigned long long bm = __ballot(brk);
				if (bm) {   // the reference stops at the first such chain: what lies behind it is not looked at
					const int fb = __ffsll((long long)bm) - 1;
					sm &= fb == 63 ? ~0ull : ((2ull << fb) - 1);
					broke = true;
				}
				if (sm) {
					large = true;
					if ((sm >> lane) & 1) { if (kfirst[kk] == HV_NONE) kfirst[kk] = (uint16_t)i; }
				}
			}
			if (!broke) {
				if (lane == 0) { kidx[nk] = (uint16_t)i; kfirst[nk] = HV_NONE; ckept[i] = large ? 2 : 3; }
				++nk;
			}
			hv_sync();
		}
		for (int k0 = 0; k0 < nk; k0 += 64) {
			const int kk = k0 + lane;
			if (kk < nk && kfirst[kk] != HV_NONE) ckept[kfirst[kk]] = 1;
		}
		hv_sync();
		if (P.max_chain_extend < n) {   // at most max_chain_extend chains with kept = 1 or 2 are extended (src/bwamem.c:372-379)
			if (lane == 0) {
				int i = 0, cnt = 0;
				for (; i < n; ++i) {
					if (ckept[i] == 0 || ckept[i] == 3) continue;
					if (++cnt >= P.max_chain_extend) break;
				}
				for (; i < n; ++i)
					if (ckept[i] < 3) ckept[i] = 0;
			}
			hv_sync();
		}
		// ---------------- emission: a kept chain per lane, where it goes by prefix sums ----------------
		const float frac_rep = (float)l_rep[rd] / (float)lq;
		int n_out = 0;
		i64 cursor = so;
		for (int t0 = 0; t0 < n; t0 += 64) {
			const int tt = t0 + lane;
			const bool kept = tt < n && ckept[tt] != 0;
			const int id = kept ? (int)(skey[tt] & 0xffff) : 0;
			const int cs = kept ? (int)g_nmem[id] : 0;
			int pre = cs;   // inclusive prefix sum over the lanes
			for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(pre, o); if (lane >= o) pre += v; }
			const unsigned long long km = __ballot(kept);
			const i64 my_cur = cursor + (pre - cs);
			const int my_out = n_out + __popcll(km & ((1ull << lane) - 1));
			if (kept) {
				// members in arrival order, then (stable) by length: the order mem_chain2aln visits them backwards in (src/bwamem.c:662-667)
				int a = 0;
				for (int k = S.head[id]; k != HV_NONE; k = S.next[k], ++a) {
					DevSeed ds;
					ds.rbeg = S_R(k); ds.qbeg = S_Q(k); ds.len = S_L(k);
					int b = a;
					while (b > 0 && seeds[my_cur + b - 1].len > ds.len) { seeds[my_cur + b] = seeds[my_cur + b - 1]; --b; }
					seeds[my_cur + b] = ds;
				}
				i64 lo = l_pac << 1, hi = 0;
				for (a = 0; a < cs; ++a) {
					const DevSeed ds = seeds[my_cur + a];
					srt[my_cur + a] = (unsigned int)a;
					const i64 b = ds.rbeg - (ds.qbeg + gap[ds.qbeg]);
					const int tail = lq - ds.qbeg - ds.len;
					const i64 e = ds.rbeg + ds.len + (tail + gap[tail]);
					lo = b < lo ? b : lo;
					hi = e > hi ? e : hi;
				}
				const int rid = (int)S.rid[id];
				const i64 first_r = S_R(S.head[id]);
				DevChain d;
				i64 fb = ann_off[rid], fe = ann_off[rid + 1];
				if (first_r >= l_pac) { const i64 x = fb; fb = (l_pac << 1) - fe; fe = (l_pac << 1) - x; }
				d.far_beg = fb; d.far_end = fe;
				d.seed_beg = (int)my_cur; d.n_seeds = cs; d.rid = rid; d.frac_rep = frac_rep;
				d.rmax0 = lo > 0 ? lo : 0;
				d.rmax1 = hi < l_pac << 1 ? hi : l_pac << 1;
				if (d.rmax0 < l_pac && l_pac < d.rmax1) {
					if (first_r < l_pac) d.rmax1 = l_pac;
					else d.rmax0 = l_pac;
				}
				d.rmax0 = d.rmax0 > fb ? d.rmax0 : fb;
				d.rmax1 = d.rmax1 < fe ? d.rmax1 : fe;
				chains[so + my_out] = d;
			}
			cursor += __shfl(pre, 63);
			n_out += __popcll(km);
		}
		if (lane == 0) n_chains[rd] = n_out;
		hv_sync();   // (the next read reuses the LDS)
	}
}

// regions from their per-read slots into one dense array (reg_pos = exclusive prefix of n_regs)
// (a read whose chains were extended as independent units, c2a_groups.hip: its regions lie per chain, and go back into the read's
// chain order — the order the serial walk of src/bwamem.c:632-786 produces them in)
__global__ void reg_pack_kernel(int n_reads, const int *__restrict__ reg_beg, const int *__restrict__ n_regs, const int *__restrict__ reg_pos,
                                const DevReg *__restrict__ regs, DevReg *__restrict__ packed, C2aUnits U, const int *__restrict__ chain_beg,
                                const int *__restrict__ chain_cnt)
{
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	const int m = n_regs[r];
	DevReg *dst = packed + reg_pos[r];
	if (U.max_units > 0 && chain_cnt[r] > U.heavy_t) {
		int k = 0;
		for (int ci = chain_beg[r]; ci < chain_beg[r] + chain_cnt[r]; ++ci) {
			const DevReg *src = regs + U.c_rabs[ci];
			for (int j = 0; j < U.c_rcnt[ci]; ++j) dst[k++] = src[j];
		}
		return;
	}
	const DevReg *src = regs + reg_beg[r];
	for (int k = 0; k < m; ++k) dst[k] = src[k];
}

} // namespace

size_t reg_pack_tmp_bytes(int n_reads)
{
	size_t t = 0;
	HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, t, (const int *)nullptr, (int *)nullptr, n_reads + 1));
	return t + 256;
}

// d_nregs must hold n_reads + 1 entries (the last one is ignored and may be anything); d_reg_pos gets n_reads + 1
// entries, the last one being the total.  Everything is queued on `stream`: no host round trip between c2a and the copy.
void launch_reg_pack(void *stream, int n_reads, const int *d_reg_beg, const int *d_nregs, int *d_reg_pos, const DevReg *d_regs, DevReg *d_packed,
                     void *d_tmp, size_t tmp_bytes, const C2aUnits *units, const int *d_chain_beg, const int *d_chain_cnt)
{
	C2aUnits U;
	if (units) U = *units;
	if (n_reads <= 0) return;
	HIP_OK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_nregs, d_reg_pos, n_reads + 1, (hipStream_t)stream));
	hipLaunchKernelGGL(reg_pack_kernel, dim3((n_reads + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_reads, d_reg_beg, d_nregs, d_reg_pos, d_regs,
	                   d_packed, U, d_chain_beg, d_chain_cnt);
	HIP_OK(hipGetLastError());
}

static size_t chain_lds_bytes(int maxs, int maxch) { return (size_t)F_NFIELDS * maxch * 64 * 4 + (size_t)(2 * (maxs + 1) + 16) * 64; }

// scratch of launch_chain: [the B-tree kernel's slices for `cap` reads][its list][two retry lists][counters][two lists of reads with more
// than 255 seeds][the slices of chain_heavy_kernel's persistent waves]
#define HV_WAVES_T 4096   // waves of the 256-seed instantiation (5.3 KB of LDS each): the reads with more than 9 chains
#define HV_WAVES_S 1024   // ... of the 1024-seed instantiation (21 KB: 7 per CU)
#define HV_WAVES_L 256    // ... of the 4096-seed one (84 KB: one per CU)
static size_t heavy_scratch_bytes() { return HV_WAVES_T * hv_scratch_bytes(256) + HV_WAVES_S * hv_scratch_bytes(1024) + HV_WAVES_L * hv_scratch_bytes(4096); }
size_t chain_general_bytes(int cap, int n_reads)
{
	return (size_t)cap * CK_GEN_BYTES + (size_t)cap * 4 + (size_t)n_reads * 8 + 512 + (size_t)n_reads * 8 + 256 + heavy_scratch_bytes();
}

// d_gen / gen_cap: scratch of chain_general_bytes(gen_cap) for the third launch (reads with more than 9 chains, up to 255
// seeds and chains, at most gen_cap of them per call); null / 0: those reads keep n_chains = -1 (host path).
void launch_chain(void *stream, const ChainParams &P, int n_reads, const int *d_len, const int *d_nseeds, const int *d_lrep,
                  const int64_t *d_seed_off, const uint64_t *d_sa, const int32_t *d_qbl, const int64_t *d_ann_off, const uint8_t *d_ann_alt,
                  int n_seqs, const int *d_tab, int tab_stride, DevChain *d_chains, DevSeed *d_seeds, unsigned int *d_srt, int *d_nchains,
                  void *d_gen, int gen_cap)
{
	if (n_reads <= 0) return;
	const size_t lds_s = chain_lds_bytes(CK_MAXSEEDS_SMALL, CK_MAXCH_SMALL), lds = chain_lds_bytes(CK_MAXSEEDS, CK_MAXCH), lds_big = chain_lds_bytes(CK_MAXSEEDS_BIG, CK_MAXCH);
	static bool s_attr = false;
	if (!s_attr) {
		if (lds > 64 * 1024) HIP_OK(hipFuncSetAttribute((const void *)chain_kernel<CK_MAXSEEDS, CK_MAXCH, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		if (lds_big > 64 * 1024) HIP_OK(hipFuncSetAttribute((const void *)chain_kernel<CK_MAXSEEDS_BIG, CK_MAXCH, CK_MAXSEEDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big));
		s_attr = true;
	}
	hipStream_t st = (hipStream_t)stream;
	const dim3 grid((n_reads + 63) / 64), block(64);
#define CHAIN_ARGS P, n_reads, d_len, d_nseeds, d_lrep, (const i64 *)d_seed_off, (const unsigned long long *)d_sa, d_qbl, (const i64 *)d_ann_off, d_ann_alt, n_seqs, d_tab, \
	               tab_stride, d_chains, d_seeds, d_srt, d_nchains
	const int big = getenv("MPIBWA_CHAIN_BIG") ? atoi(getenv("MPIBWA_CHAIN_BIG")) : 2;   // 0: up to 64 seeds only, 1: + 255 seeds, 2: + the B-tree kernel
	const int *noflt = d_tab + 5 * tab_stride;
	hipLaunchKernelGGL((chain_kernel<CK_MAXSEEDS_SMALL, CK_MAXCH_SMALL, -1>), grid, block, lds_s, st, CHAIN_ARGS, (const int *)nullptr, (const unsigned int *)nullptr);
	if (!d_gen || gen_cap <= 0) {   // no scratch for the lists: the retries look at every read
		hipLaunchKernelGGL((chain_kernel<CK_MAXSEEDS, CK_MAXCH, 0>), grid, block, lds, st, CHAIN_ARGS, (const int *)nullptr, (const unsigned int *)nullptr);
		if (big >= 1) hipLaunchKernelGGL((chain_kernel<CK_MAXSEEDS_BIG, CK_MAXCH, CK_MAXSEEDS>), grid, block, lds_big, st, CHAIN_ARGS, (const int *)nullptr, (const unsigned int *)nullptr);
	} else {
		uint8_t *scratch = (uint8_t *)d_gen;
		int *list_g = (int *)(scratch + (size_t)gen_cap * CK_GEN_BYTES);
		int *list_a = list_g + gen_cap, *list_b = list_a + n_reads;
		unsigned int *count = (unsigned int *)(list_b + n_reads);   // [0] general, [1] 64-seed retry, [2] 255-seed retry, [4..7] heavy: list sizes and work counters
		int *list_h1 = (int *)((uint8_t *)count + 512), *list_h2 = list_h1 + n_reads;
		uint8_t *hv_scr = (uint8_t *)(((uintptr_t)(list_h2 + n_reads) + 255) & ~(uintptr_t)255);
		HIP_OK(hipMemsetAsync(count, 0, 64, st));
		const dim3 pgrid((n_reads + 255) / 256), pblock(256);
		hipLaunchKernelGGL(chain_pick_kernel, pgrid, pblock, 0, st, n_reads, d_len, d_nseeds, (const int *)d_nchains, noflt, 0, CK_MAXSEEDS, n_reads, list_a, count + 1);
		hipLaunchKernelGGL((chain_kernel<CK_MAXSEEDS, CK_MAXCH, 0>), grid, block, lds, st, CHAIN_ARGS, (const int *)list_a, (const unsigned int *)(count + 1));
		if (big >= 1) {
			hipLaunchKernelGGL(chain_pick_kernel, pgrid, pblock, 0, st, n_reads, d_len, d_nseeds, (const int *)d_nchains, noflt, CK_MAXSEEDS, CK_MAXSEEDS_BIG, n_reads, list_b,
			                   count + 2);
			hipLaunchKernelGGL((chain_kernel<CK_MAXSEEDS_BIG, CK_MAXCH, CK_MAXSEEDS>), grid, block, lds_big, st, CHAIN_ARGS, (const int *)list_b, (const unsigned int *)(count + 2));
		}
		static const bool use_general = getenv("MPIBWA_CHAIN_GENERAL") && atoi(getenv("MPIBWA_CHAIN_GENERAL")) != 0;
		bool launch_t = false;
		if (big >= 2 && !use_general) {
			// the reads with more than 9 chains (up to 255 seeds): a wavefront per read with the B-tree in LDS (on a low-complexity
			// reference a third of the reads are of this kind; a lane per read with the tree in HBM — chain_general_kernel, rounds 2-3 —
			// took 51 ms per chunk there)
			int *list_t = list_a;   // (the 64-seed retry is done with it)
			hipLaunchKernelGGL(chain_pick_kernel, pgrid, pblock, 0, st, n_reads, d_len, d_nseeds, (const int *)d_nchains, noflt, 0, CK_MAXSEEDS_BIG, n_reads, list_t, count);
			launch_t = true;   // (launched below, next to the instantiations for more seeds)
		}
		if (big >= 2 && use_general) {
			// the scratch holds gen_cap reads at a time: several rounds over it (on a low-complexity reference a third of the reads have
			// more than 9 chains; a round that finds nothing left is two empty launches)
			static const int rounds = getenv("MPIBWA_CHAIN_ROUNDS") ? std::max(1, atoi(getenv("MPIBWA_CHAIN_ROUNDS"))) : 4;
			for (int r = 0; r < rounds; ++r) {
				if (r) HIP_OK(hipMemsetAsync(count, 0, 4, st));
				hipLaunchKernelGGL(chain_pick_kernel, pgrid, pblock, 0, st, n_reads, d_len, d_nseeds, (const int *)d_nchains, noflt, 0, CK_MAXSEEDS_BIG, gen_cap, list_g, count);
				hipLaunchKernelGGL(chain_general_kernel, dim3((gen_cap + 63) / 64), dim3(64), 0, st, P, (const int *)list_g, (const unsigned int *)count, gen_cap, d_len,
				                   d_nseeds, d_lrep, (const i64 *)d_seed_off, (const unsigned long long *)d_sa, d_qbl, (const i64 *)d_ann_off, d_ann_alt, n_seqs,
				                   d_tab, scratch, d_chains, d_seeds, d_srt, d_nchains);
			}
		}
		if (big >= 2 && !(getenv("MPIBWA_CHAIN_HEAVY") && atoi(getenv("MPIBWA_CHAIN_HEAVY")) == 0)) {
			// reads with more than 255 seeds (high-copy repeats): a wavefront per read, two LDS footprints
			static bool s_attr2 = false;
			if (!s_attr2) {
				HIP_OK(hipFuncSetAttribute((const void *)chain_heavy_kernel<4096>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hv_lds_bytes(4096)));
				s_attr2 = true;
			}
			uint8_t *scr_s = hv_scr + HV_WAVES_T * hv_scratch_bytes(256);
			hipLaunchKernelGGL(chain_pick_kernel, pgrid, pblock, 0, st, n_reads, d_len, d_nseeds, (const int *)d_nchains, noflt, CK_MAXSEEDS_BIG, 1024, n_reads, list_h1, count + 4);
			hipLaunchKernelGGL(chain_pick_kernel, pgrid, pblock, 0, st, n_reads, d_len, d_nseeds, (const int *)d_nchains, noflt, 1024, 4096, n_reads, list_h2, count + 6);
			// The instantiations are latency-bound launches of a few hundred to a few thousand waves each: side by side on two side streams
			// of the call's stream (created once per stream) instead of one after the other — 23 -> 10 ms per sub-batch with one call in flight.
			struct Side { hipStream_t s[2]; hipEvent_t fork, join[2]; };
			static std::mutex side_mu;
			static std::vector<std::pair<hipStream_t, Side>> sides;
			Side sd;
			{
				std::lock_guard<std::mutex> lk(side_mu);
				bool found = false;
				for (auto &e : sides) if (e.first == st) { sd = e.second; found = true; break; }
				if (!found) {
					for (int k = 0; k < 2; ++k) { HIP_OK(hipStreamCreateWithFlags(&sd.s[k], hipStreamNonBlocking)); HIP_OK(hipEventCreateWithFlags(&sd.join[k], hipEventDisableTiming)); }
					HIP_OK(hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming));
					sides.emplace_back(st, sd);
				}
			}
			HIP_OK(hipEventRecord(sd.fork, st));
			for (int k = 0; k < 2; ++k) HIP_OK(hipStreamWaitEvent(sd.s[k], sd.fork, 0));
			hipLaunchKernelGGL((chain_heavy_kernel<4096>), dim3(HV_WAVES_L), dim3(64), hv_lds_bytes(4096), sd.s[0], P, (const int *)list_h2, (const unsigned int *)(count + 6), count + 7,
			                   n_reads, d_len, d_nseeds, d_lrep, (const i64 *)d_seed_off, (const unsigned long long *)d_sa, d_qbl, (const i64 *)d_ann_off, d_ann_alt, n_seqs, d_tab,
			                   scr_s + HV_WAVES_S * hv_scratch_bytes(1024), d_chains, d_seeds, d_srt, d_nchains);
			hipLaunchKernelGGL((chain_heavy_kernel<1024>), dim3(HV_WAVES_S), dim3(64), hv_lds_bytes(1024), sd.s[1], P, (const int *)list_h1, (const unsigned int *)(count + 4), count + 5,
			                   n_reads, d_len, d_nseeds, d_lrep, (const i64 *)d_seed_off, (const unsigned long long *)d_sa, d_qbl, (const i64 *)d_ann_off, d_ann_alt, n_seqs, d_tab,
			                   scr_s, d_chains, d_seeds, d_srt, d_nchains);
			if (launch_t) {
				hipLaunchKernelGGL((chain_heavy_kernel<256>), dim3(HV_WAVES_T), dim3(64), hv_lds_bytes(256), st, P, (const int *)list_a, (const unsigned int *)count, count + 3,
				                   n_reads, d_len, d_nseeds, d_lrep, (const i64 *)d_seed_off, (const unsigned long long *)d_sa, d_qbl, (const i64 *)d_ann_off, d_ann_alt, n_seqs, d_tab,
				                   hv_scr, d_chains, d_seeds, d_srt, d_nchains);
				launch_t = false;
			}
			for (int k = 0; k < 2; ++k) { HIP_OK(hipEventRecord(sd.join[k], sd.s[k])); HIP_OK(hipStreamWaitEvent(st, sd.join[k], 0)); }
		}
		if (launch_t)   // (MPIBWA_CHAIN_HEAVY=0: the 256-seed instantiation on its own)
			hipLaunchKernelGGL((chain_heavy_kernel<256>), dim3(HV_WAVES_T), dim3(64), hv_lds_bytes(256), st, P, (const int *)list_a, (const unsigned int *)count, count + 3,
			                   n_reads, d_len, d_nseeds, d_lrep, (const i64 *)d_seed_off, (const unsigned long long *)d_sa, d_qbl, (const i64 *)d_ann_off, d_ann_alt, n_seqs, d_tab,
			                   hv_scr, d_chains, d_seeds, d_srt, d_nchains);
	}
#undef CHAIN_ARGS
	HIP_OK(hipGetLastError());
}

} // namespace mbw
