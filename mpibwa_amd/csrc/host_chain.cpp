// host_chain.cpp — reference geometry helpers, seed chaining and chain filtering (host stages).
//
//   bns_pos2rid / bns_intv2rid / bns_get_seq / bns_fetch_seq     src/bntseq.c:349-446
//   mem_chain (seeds -> chains through an ordered map)           src/bwamem.c:251-315
//   test_and_merge, mem_chain_weight                             src/bwamem.c:190-237
//   mem_chain_flt, mem_flt_chained_seeds, mem_seed_sw            src/bwamem.c:327-385, 571-617
//
// The ordered map is the reference's B-tree (src/kbtree.h, node size 512 B =>
// at most 9 keys per node).  Duplicate keys are possible (two chains anchored
// at the same reference position) and the tree shape decides which one a
// predecessor query meets and in which order they are traversed, so the same
// B-tree insertion / search rules are restated here rather than substituting
// std::map.
#include "host.h"
#include "device.h"
#include "sortutil.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <memory>

namespace mbw {

// ---------------------------------------------------------------------------
// reference coordinates
// ---------------------------------------------------------------------------
int bns_pos2rid(const bntseq_t *bns, int64_t pos_f)
{
	if (pos_f >= bns->l_pac) return -1;
	int left = 0, mid = 0, right = bns->n_seqs;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= bns->anns[mid].offset) {
			if (mid == bns->n_seqs - 1 || pos_f < bns->anns[mid + 1].offset) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}

int bns_intv2rid(const bntseq_t *bns, int64_t rb, int64_t re)
{
	int is_rev;
	if (rb < bns->l_pac && re > bns->l_pac) return -2;
	int rid_b = bns_pos2rid(bns, bns_depos(bns, rb, &is_rev));
	int rid_e = rb < re ? bns_pos2rid(bns, bns_depos(bns, re - 1, &is_rev)) : rid_b;
	return rid_b == rid_e ? rid_b : -1;
}

static inline int pac_base(const uint8_t *pac, int64_t l) { return pac[l >> 2] >> ((~l & 3) << 1) & 3; }

// bases of [beg,end) in the doubled (forward + reverse-complement) coordinate; empty + !ok when the
// interval bridges the strand boundary
std::vector<uint8_t> bns_get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, bool *ok)
{
	std::vector<uint8_t> seq;
	if (end < beg) std::swap(beg, end);
	if (end > l_pac << 1) end = l_pac << 1;
	if (beg < 0) beg = 0;
	*ok = beg >= l_pac || end <= l_pac;
	if (!*ok) return seq;
	seq.resize(end - beg);
	if (beg >= l_pac) {
		int64_t beg_f = (l_pac << 1) - 1 - end, end_f = (l_pac << 1) - 1 - beg, l = 0;
		for (int64_t k = end_f; k > beg_f; --k) seq[l++] = 3 - pac_base(pac, k);
	} else {
		int64_t l = 0;
		for (int64_t k = beg; k < end; ++k) seq[l++] = pac_base(pac, k);
	}
	return seq;
}

std::vector<uint8_t> bns_fetch_seq(const bntseq_t *bns, const uint8_t *pac, int64_t *beg, int64_t mid, int64_t *end, int *rid)
{
	int is_rev;
	if (*end < *beg) std::swap(*beg, *end);
	assert(*beg <= mid && mid < *end);
	*rid = bns_pos2rid(bns, bns_depos(bns, mid, &is_rev));
	int64_t far_beg = bns->anns[*rid].offset, far_end = far_beg + bns->anns[*rid].len;
	if (is_rev) {
		int64_t t = far_beg;
		far_beg = (bns->l_pac << 1) - far_end;
		far_end = (bns->l_pac << 1) - t;
	}
	*beg = std::max(*beg, far_beg);
	*end = std::min(*end, far_end);
	bool ok;
	std::vector<uint8_t> seq = bns_get_seq(bns->l_pac, pac, *beg, *end, &ok);
	if (!ok || (int64_t)seq.size() != *end - *beg) die("bns_fetch_seq: inconsistent interval [%lld,%lld)", (long long)*beg, (long long)*end);
	return seq;
}

// ---------------------------------------------------------------------------
// B-tree keyed by chain position (kbtree.h semantics, t = 5)
// ---------------------------------------------------------------------------
namespace {

struct BTree {
	static const int T = 5, MAXK = 2 * T - 1;
	struct Node {
		bool internal;
		int n;
		int key[MAXK];
		int child[MAXK + 1];   // indices into `nodes`
	};
	std::vector<Node> &nodes;  // recycled across reads by the caller
	int root;
	int n_keys = 0;
	const std::vector<HChain> &ch;

	BTree(const std::vector<HChain> &chains, std::vector<Node> &pool) : nodes(pool), ch(chains) { nodes.clear(); root = make(); }
	int make()
	{
		nodes.emplace_back();
		nodes.back().internal = false; nodes.back().n = 0;
		return (int)nodes.size() - 1;
	}
	int cmp(int64_t a, int64_t b) const { return (b < a) - (a < b); }
	// index of the first key >= pos, stepped back by one when that key is > pos;
	// *r = sign(pos - key[...]) exactly as the reference's __kb_getp_aux
	int locate(const Node *x, int64_t pos, int *r) const
	{
		int tr, begin = 0, end = x->n;
		if (!r) r = &tr;
		if (x->n == 0) return -1;
		while (begin < end) {
			int mid = (begin + end) >> 1;
			if (cmp(ch[x->key[mid]].pos, pos) < 0) begin = mid + 1;
			else end = mid;
		}
		if (begin == x->n) { *r = 1; return x->n - 1; }
		if ((*r = cmp(pos, ch[x->key[begin]].pos)) < 0) --begin;
		return begin;
	}
	// closest key <= pos on the search path (or -1)
	int lower(int64_t pos) const
	{
		int low = -1, r = 0, xi = root;
		for (;;) {
			const Node *x = &nodes[xi];
			int i = locate(x, pos, &r);
			if (i >= 0 && r == 0) return x->key[i];
			if (i >= 0) low = x->key[i];
			if (!x->internal) return low;
			xi = x->child[i + 1];
		}
	}
	void split(int xi, int i, int yi)
	{
		int zi = make();   // may move `nodes`: take the pointers afterwards
		Node *x = &nodes[xi], *y = &nodes[yi], *z = &nodes[zi];
		z->internal = y->internal;
		z->n = T - 1;
		memcpy(z->key, y->key + T, sizeof(int) * (T - 1));
		if (y->internal) memcpy(z->child, y->child + T, sizeof(int) * T);
		y->n = T - 1;
		memmove(x->child + i + 2, x->child + i + 1, sizeof(int) * (x->n - i));
		x->child[i + 1] = zi;
		memmove(x->key + i + 1, x->key + i, sizeof(int) * (x->n - i));
		x->key[i] = y->key[T - 1];
		++x->n;
	}
	void put_nonfull(int xi, int k)
	{
		int64_t pos = ch[k].pos;
		while (nodes[xi].internal) {
			int i = locate(&nodes[xi], pos, 0) + 1;
			if (nodes[nodes[xi].child[i]].n == MAXK) {
				split(xi, i, nodes[xi].child[i]);
				if (cmp(pos, ch[nodes[xi].key[i]].pos) > 0) ++i;
			}
			xi = nodes[xi].child[i];
		}
		Node *x = &nodes[xi];
		int i = locate(x, pos, 0);
		if (i != x->n - 1) memmove(x->key + i + 2, x->key + i + 1, (x->n - i - 1) * sizeof(int));
		x->key[i + 1] = k;
		++x->n;
	}
	void put(int k)
	{
		++n_keys;
		int r = root;
		if (nodes[r].n == MAXK) {
			int si = make();
			root = si; nodes[si].internal = true; nodes[si].n = 0;
			nodes[si].child[0] = r;
			split(si, 0, r);
			r = si;
		}
		put_nonfull(r, k);
	}
	void inorder(int xi, std::vector<int> &out) const
	{
		const Node *x = &nodes[xi];
		for (int i = 0; i < x->n; ++i) {
			if (x->internal) inorder(x->child[i], out);
			out.push_back(nodes[xi].key[i]);
			x = &nodes[xi];
		}
		if (x->internal) inorder(x->child[x->n], out);
	}
};

// 1 when seed p was absorbed by (or is already covered by) chain c
bool test_and_merge(const mem_opt_t *opt, int64_t l_pac, HChain &c, const HSeed &p, int seed_rid)
{
	const HSeed &last = c.seeds.back(), &first = c.seeds.front();
	int64_t qend = last.qbeg + last.len, rend = last.rbeg + last.len;
	if (seed_rid != c.rid) return false;
	if (p.qbeg >= first.qbeg && p.qbeg + p.len <= qend && p.rbeg >= first.rbeg && p.rbeg + p.len <= rend) return true;
	if ((last.rbeg < l_pac || first.rbeg < l_pac) && p.rbeg >= l_pac) return false;   // never chain across strands
	int64_t x = p.qbeg - last.qbeg, y = p.rbeg - last.rbeg;
	if (y >= 0 && x - y <= opt->w && y - x <= opt->w && x - last.len < opt->max_chain_gap && y - last.len < opt->max_chain_gap) {
		c.seeds.push_back(p);
		return true;
	}
	return false;
}

int chain_weight(const HChain &c)
{
	int64_t end = 0;
	int w = 0, tmp;
	for (const HSeed &s : c.seeds) {
		if (s.qbeg >= end) w += s.len;
		else if (s.qbeg + s.len > end) w += s.qbeg + s.len - end;
		end = end > s.qbeg + s.len ? end : s.qbeg + s.len;
	}
	tmp = w; w = 0; end = 0;
	for (const HSeed &s : c.seeds) {
		if (s.rbeg >= end) w += s.len;
		else if (s.rbeg + s.len > end) w += s.rbeg + s.len - end;
		end = end > s.rbeg + s.len ? end : s.rbeg + s.len;
	}
	w = w < tmp ? w : tmp;
	return w < 1 << 30 ? w : (1 << 30) - 1;
}

} // namespace

// Scratch recycled from read to read by one thread: chain objects (their seed vectors keep their capacity), B-tree nodes.
struct ChainScratch::Impl {
	std::vector<HChain> pool;
	size_t used = 0;
	std::vector<BTree::Node> nodes;
	std::vector<int> order, kept, kcol;
};
ChainScratch::ChainScratch() : p(new Impl()) {}
ChainScratch::~ChainScratch() { delete p; }

void chains_from_seeds(const mem_opt_t *opt, const bntseq_t *bns, int l_query, const HSeed *seeds, int n_seeds, int l_rep,
                       ChainScratch &S, std::vector<HChain *> &chains)
{
	ChainScratch::Impl &W = *S.p;
	chains.clear();
	if ((int)W.pool.size() < n_seeds) W.pool.resize(n_seeds);   // at most one chain per seed; pointers stay valid during this read
	W.used = 0;
	BTree tree(W.pool, W.nodes);
	// consecutive seeds mostly fall into the contig (and strand) of the previous one: remember its span in the doubled
	// coordinate instead of two binary searches per seed
	int64_t c_lo = 0, c_hi = -1;
	int c_rid = -1;
	for (int k = 0; k < n_seeds; ++k) {
		const HSeed &s = seeds[k];
		int rid;
		if (s.rbeg >= c_lo && s.rbeg + s.len <= c_hi && s.len > 0) rid = c_rid;
		else {
			rid = bns_intv2rid(bns, s.rbeg, s.rbeg + s.len);
			if (rid >= 0) {
				const int64_t o = bns->anns[rid].offset, l = bns->anns[rid].len;
				if (s.rbeg < bns->l_pac) { c_lo = o; c_hi = o + l; }
				else { c_lo = (bns->l_pac << 1) - o - l; c_hi = (bns->l_pac << 1) - o; }
				c_rid = rid;
			}
		}
		if (rid < 0) continue;   // bridges two contigs or the strand boundary
		bool add = true;
		if (tree.n_keys) {
			int low = tree.lower(s.rbeg);
			if (low >= 0 && test_and_merge(opt, bns->l_pac, W.pool[low], s, rid)) add = false;
		}
		if (add) {
			HChain &c = W.pool[W.used];
			c.pos = s.rbeg; c.rid = rid; c.is_alt = !!bns->anns[rid].is_alt;
			c.first = -1; c.w = 0; c.kept = 0;
			c.seeds.clear();
			c.seeds.push_back(s);
			tree.put((int)W.used);
			++W.used;
		}
	}
	W.order.clear();
	tree.inorder(tree.root, W.order);
	float frac = (float)l_rep / l_query;
	for (int id : W.order) {
		W.pool[id].frac_rep = frac;
		chains.push_back(&W.pool[id]);
	}
}

#define CHN_BEG(c) ((c)->seeds.front().qbeg)
#define CHN_END(c) ((c)->seeds.back().qbeg + (c)->seeds.back().len)

// The reference sorts the chain structs themselves with its unstable introsort; sorting pointers with the same
// comparator performs the same comparisons and swaps, hence the same order, without copying seed arrays.
void chain_filter(const mem_opt_t *opt, ChainScratch &S, std::vector<HChain *> &a)
{
	if (a.empty()) return;
	size_t k = 0;
	for (size_t i = 0; i < a.size(); ++i) {
		HChain *c = a[i];
		c->first = -1; c->kept = 0;
		c->w = (uint32_t)chain_weight(*c) & 0x1fffffffu;   // 29-bit field in the reference
		if ((int)c->w < opt->min_chain_weight) continue;
		a[k++] = c;
	}
	a.resize(k);
	int n = (int)a.size();
	if (n == 0) return;   // (the reference would index a[0] here; min_chain_weight = 0 by default so this cannot happen there)
	ks_introsort((size_t)n, a.data(), [](const HChain *x, const HChain *y) { return x->w > y->w; });
	// The chains that survived so far, as columns (query span, weight, ALT flag, the first chain each one shadows): a read of a
	// high-copy repeat has hundreds of chains of equal weight that all overlap and all survive, so the test below runs n^2 / 2
	// times — over contiguous ints instead of through every chain's seed vector (same comparisons, same order, same result).
	ChainScratch::Impl &W = *S.p;
	std::vector<int> &kept = W.kept;
	kept.clear();
	W.kcol.resize((size_t)5 * n);
	int *kbeg = W.kcol.data(), *kend = kbeg + n, *kw = kend + n, *kalt = kw + n, *kfirst = kalt + n;
	int nk = 0;
	auto keep = [&](int i) {
		kept.push_back(i);
		kbeg[nk] = CHN_BEG(a[i]); kend[nk] = CHN_END(a[i]); kw[nk] = (int)a[i]->w; kalt[nk] = a[i]->is_alt ? 1 : 0; kfirst[nk] = -1;
		++nk;
	};
	a[0]->kept = 3;
	keep(0);
	const float mask_level = opt->mask_level, drop_ratio = opt->drop_ratio;
	const int max_chain_gap = opt->max_chain_gap, two_seeds = opt->min_seed_len << 1;
	for (int i = 1; i < n; ++i) {
		bool large_ovlp = false;
		const int bi = CHN_BEG(a[i]), ei = CHN_END(a[i]), li = ei - bi, wi = (int)a[i]->w, alt_i = a[i]->is_alt ? 1 : 0;
		int kk;
		for (kk = 0; kk < nk; ++kk) {
			const int b_max = kbeg[kk] > bi ? kbeg[kk] : bi;
			const int e_min = kend[kk] < ei ? kend[kk] : ei;
			if (e_min > b_max && (!kalt[kk] || alt_i)) {
				const int lj = kend[kk] - kbeg[kk];
				const int min_l = li < lj ? li : lj;
				if (e_min - b_max >= min_l * mask_level && min_l < max_chain_gap) {
					large_ovlp = true;
					if (kfirst[kk] < 0) kfirst[kk] = i;
					if (wi < kw[kk] * drop_ratio && kw[kk] - wi >= two_seeds) break;
				}
			}
		}
		if (kk == nk) {
			a[i]->kept = large_ovlp ? 2 : 3;
			keep(i);
		}
	}
	for (int kk = 0; kk < nk; ++kk) {
		a[kept[kk]]->first = kfirst[kk];
		if (kfirst[kk] >= 0) a[kfirst[kk]]->kept = 1;
	}
	int i = 0, cnt = 0;
	for (; i < n; ++i) {   // at most max_chain_extend chains with kept = 1 or 2 are extended
		if (a[i]->kept == 0 || a[i]->kept == 3) continue;
		if (++cnt >= opt->max_chain_extend) break;
	}
	for (; i < n; ++i)
		if (a[i]->kept < 3) a[i]->kept = 0;
	k = 0;
	for (int q = 0; q < n; ++q) {
		if (a[q]->kept == 0) continue;
		a[k++] = a[q];
	}
	a.resize(k);
}

// ---- short-seed rescoring for long reads (no-op below ~700 bp) ----
static int seed_sw(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, int l_query, const uint8_t *query, const HSeed &s)
{
	const int SHORT_EXT = 50, SHORT_LEN = 200;
	int64_t l_pac = bns->l_pac;
	if (s.len >= SHORT_LEN) return -1;
	int qb = s.qbeg, qe = s.qbeg + s.len, rid;
	int64_t rb = s.rbeg, re = s.rbeg + s.len, mid = (rb + re) >> 1;
	qb -= SHORT_EXT; qb = qb > 0 ? qb : 0;
	qe += SHORT_EXT; qe = qe < l_query ? qe : l_query;
	rb -= SHORT_EXT; rb = rb > 0 ? rb : 0;
	re += SHORT_EXT; re = re < l_pac << 1 ? re : l_pac << 1;
	if (rb < l_pac && l_pac < re) {
		if (mid < l_pac) re = l_pac;
		else rb = l_pac;
	}
	if (qe - qb >= SHORT_LEN || re - rb >= SHORT_LEN) return -1;
	std::vector<uint8_t> rseq = bns_fetch_seq(bns, pac, &rb, mid, &re, &rid);
	std::vector<uint8_t> q(query + qb, query + qe);
	KswResult x = ksw_align2(qe - qb, q.data(), (int)(re - rb), rseq.data(), opt->mat, opt->o_del, opt->e_del, opt->o_ins,
	                         opt->e_ins, KSW_XSTART);
	return x.score;
}

void filter_chained_seeds(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, int l_query, const uint8_t *query,
                          std::vector<HChain *> &chains)
{
	double min_l = opt->min_chain_weight ? 1.1f * opt->min_chain_weight : 5.5f * log(l_query);
	int min_HSP_score = (int)(opt->a * min_l + .499);
	if (min_l > 0.05f * l_query) return;   // short reads: nothing to do
	for (HChain *cp : chains) {
		HChain &c = *cp;
		size_t k = 0;
		for (size_t j = 0; j < c.seeds.size(); ++j) {
			HSeed &s = c.seeds[j];
			s.score = seed_sw(opt, bns, pac, l_query, query, s);
			if (s.score < 0 || s.score >= min_HSP_score) {
				s.score = s.score < 0 ? s.len * opt->a : s.score;
				c.seeds[k++] = s;
			}
		}
		c.seeds.resize(k);
	}
}

// One filtered chain in the layout the extension kernels read: the chain's contig bounds and reference window
// (src/bwamem.c:642-661) and its seeds in the order mem_chain2aln visits them — the reference sorts (score << 32 | index)
// ascending and walks from the end (src/bwamem.c:662-667); keys are distinct, so any sort gives that order.
void pack_chain_for_device(const bntseq_t *bns, const HChain &ch, int l_query, const int *gap_h, std::vector<uint64_t> &key, DevChain &d, DevSeed *osd)
{
	const int cs = (int)ch.seeds.size();
	d.n_seeds = cs; d.rid = ch.rid; d.frac_rep = ch.frac_rep;
	d.far_beg = d.far_end = 0;
	if (cs) {
		int is_rev;
		bns_depos(bns, ch.seeds[0].rbeg, &is_rev);
		int64_t fb = bns->anns[ch.rid].offset, fe = fb + bns->anns[ch.rid].len;
		if (is_rev) { int64_t t = fb; fb = (bns->l_pac << 1) - fe; fe = (bns->l_pac << 1) - t; }
		d.far_beg = fb; d.far_end = fe;
	}
	key.resize(cs);
	const HSeed *hsd = ch.seeds.data();
	for (int k = 0; k < cs; ++k) key[k] = (uint64_t)hsd[k].score << 32 | (uint32_t)k;
	if (cs == 2) { if (key[1] < key[0]) std::swap(key[0], key[1]); }
	else if (cs > 2) std::sort(key.begin(), key.end());
	int64_t lo = bns->l_pac << 1, hi = 0;
	for (int k = 0; k < cs; ++k) {
		const HSeed &t = hsd[(uint32_t)key[k]];
		osd[k].rbeg = t.rbeg; osd[k].qbeg = t.qbeg; osd[k].len = t.len;
		// widest reference span any seed of the chain could reach (src/bwamem.c:642-658)
		const int64_t b = t.rbeg - (t.qbeg + gap_h[t.qbeg]);
		const int tail = l_query - t.qbeg - t.len;
		const int64_t e = t.rbeg + t.len + (tail + gap_h[tail]);
		lo = b < lo ? b : lo;
		hi = e > hi ? e : hi;
	}
	d.rmax0 = lo > 0 ? lo : 0;
	d.rmax1 = hi < bns->l_pac << 1 ? hi : bns->l_pac << 1;
	if (cs) {
		if (d.rmax0 < bns->l_pac && bns->l_pac < d.rmax1) {   // never cross the strand boundary
			if (ch.seeds[0].rbeg < bns->l_pac) d.rmax1 = bns->l_pac;
			else d.rmax0 = bns->l_pac;
		}
		d.rmax0 = d.rmax0 > d.far_beg ? d.rmax0 : d.far_beg;   // bns_fetch_seq clamps to the contig
		d.rmax1 = d.rmax1 < d.far_end ? d.rmax1 : d.far_end;
	}
}

int cal_max_gap(const mem_opt_t *opt, int qlen)
{
	int l_del = (int)((double)(qlen * opt->a - opt->o_del) / opt->e_del + 1.);
	int l_ins = (int)((double)(qlen * opt->a - opt->o_ins) / opt->e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < opt->w << 1 ? l : opt->w << 1;
}

} // namespace mbw
