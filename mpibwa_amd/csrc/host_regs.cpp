// host_regs.cpp — alignment-region post-processing and SAM record generation (host stages).
//
//   mem_sort_dedup_patch, mem_patch_reg            src/bwamem.c:406-489
//   mem_mark_primary_se(_core)                     src/bwamem.c:493-569
//   mem_approx_mapq_se, mem_reorder_primary5       src/bwamem.c:952-1001
//   bwa_gen_cigar2 (CIGAR + MD + NM)               src/bwa.c:121-207
//   mem_reg2aln, infer_bw                          src/bwamem.c:792-800, 1089-1159
//   mem_aln2sam, add_cigar, mem_reg2sam            src/bwamem.c:812-946, 1003-1049
//   mem_gen_alt (XA tag)                           src/bwamem_extra.c:91-140
#include "host.h"
#include "sortutil.h"
#include "hprof.h"

#include <algorithm>
#include <cassert>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace mbw {

// ---------------------------------------------------------------------------
// redundant-hit removal and colinear-hit patching
// ---------------------------------------------------------------------------
static int patch_reg(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, uint8_t *query, const HReg *a, const HReg *b, int *_w)
{
	if (bns == 0 || pac == 0 || query == 0) return 0;
	assert(a->rid == b->rid && a->rb <= b->rb);
	if (a->rb < bns->l_pac && b->rb >= bns->l_pac) return 0;
	if (a->qb >= b->qb || a->qe >= b->qe || a->re >= b->re) return 0;   // not colinear
	int w = (int)((a->re - b->rb) - (a->qe - b->qb));
	w = w > 0 ? w : -w;
	double r = (double)(a->re - b->rb) / (b->re - a->rb) - (double)(a->qe - b->qb) / (b->qe - a->qb);
	r = r > 0. ? r : -r;
	if (a->re < b->rb || a->qe < b->qb) {
		if (w > opt->w << 1 || r >= 0.05f) return 0;
	} else if (w > opt->w << 2 || r >= 0.05f * 2) return 0;
	w += a->w + b->w;
	w = w < opt->w << 2 ? w : opt->w << 2;
	int score = 0;
	gen_cigar2(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w, bns->l_pac, pac, b->qe - a->qb, query + a->qb, a->rb, b->re,
	           &score, 0, 0, 0);
	int q_s = (int)((double)(b->qe - a->qb) / ((b->qe - b->qb) + (a->qe - a->qb)) * (b->score + a->score) + .499);
	int r_s = (int)((double)(b->re - a->rb) / ((b->re - b->rb) + (a->re - a->rb)) * (b->score + a->score) + .499);
	if ((double)score / (q_s > r_s ? q_s : r_s) < 0.90f) return 0;
	*_w = w;
	return score;
}

int sort_dedup_patch(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, uint8_t *query, HRegV &v)
{
	int n = (int)v.size();
	v.settled = true;    // (unless two hits are merged below: the merged hit has not been compared with its new neighbours)
	if (n <= 1) return n;
	HProf hp_(HP_DEDUP);
	HReg *a = v.data();
	ks_introsort((size_t)n, a, [](const HReg &x, const HReg &y) { return x.re < y.re; });   // by END position
	for (int i = 0; i < n; ++i) a[i].n_comp = 1;
	for (int i = 1; i < n; ++i) {
		HReg *p = &a[i];
		if (p->rid != a[i - 1].rid || p->rb >= a[i - 1].re + opt->max_chain_gap) continue;
		for (int j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + opt->max_chain_gap; --j) {
			HReg *q = &a[j];
			if (q->qe == q->qb) continue;   // already excluded
			int64_t orr = q->re - p->rb;
			int64_t oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
			int64_t mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
			int64_t mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
			int score, w;
			if (orr > opt->mask_level_redun * mr && oq > opt->mask_level_redun * mq) {   // one of the two is redundant
				if (p->score < q->score) { p->qe = p->qb; break; }
				else q->qe = q->qb;
			} else if (q->rb < p->rb && (score = patch_reg(opt, bns, pac, query, q, p, &w)) > 0) {   // merge q into p
				p->n_comp += q->n_comp + 1;
				p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
				p->sub = p->sub > q->sub ? p->sub : q->sub;
				p->csub = p->csub > q->csub ? p->csub : q->csub;
				p->qb = q->qb; p->rb = q->rb;
				p->truesc = p->score = score;
				p->w = w;
				q->qb = q->qe;
				v.settled = false;
			}
		}
	}
	int m = 0;
	for (int i = 0; i < n; ++i)
		if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
	n = m;
	ks_introsort((size_t)n, a, [](const HReg &x, const HReg &y) {
		return x.score > y.score || (x.score == y.score && (x.rb < y.rb || (x.rb == y.rb && x.qb < y.qb)));
	});
	for (int i = 1; i < n; ++i)
		if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb) a[i].qe = a[i].qb;
	m = n > 0 ? 1 : 0;
	for (int i = 1; i < n; ++i)
		if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
	v.resize(m);
	return m;
}

// ---------------------------------------------------------------------------
// primary / secondary marking
// ---------------------------------------------------------------------------
static void mark_primary_core(const mem_opt_t *opt, int n, HReg *a, std::vector<int> &z)
{
	int tmp = opt->a + opt->b;
	tmp = opt->o_del + opt->e_del > tmp ? opt->o_del + opt->e_del : tmp;
	tmp = opt->o_ins + opt->e_ins > tmp ? opt->o_ins + opt->e_ins : tmp;
	z.clear();
	z.push_back(0);
	for (int i = 1; i < n; ++i) {
		size_t k;
		for (k = 0; k < z.size(); ++k) {
			int j = z[k];
			int b_max = a[j].qb > a[i].qb ? a[j].qb : a[i].qb;
			int e_min = a[j].qe < a[i].qe ? a[j].qe : a[i].qe;
			if (e_min > b_max) {
				int min_l = a[i].qe - a[i].qb < a[j].qe - a[j].qb ? a[i].qe - a[i].qb : a[j].qe - a[j].qb;
				if (e_min - b_max >= min_l * opt->mask_level) {   // significant overlap on the query
					if (a[j].sub == 0) a[j].sub = a[i].score;
					if (a[j].score - a[i].score <= tmp && (a[j].is_alt || !a[i].is_alt)) ++a[j].sub_n;
					break;
				}
			}
		}
		if (k == z.size()) z.push_back(i);
		else a[i].secondary = z[k];
	}
}

int mark_primary_se(const mem_opt_t *opt, HRegV &v, int64_t id)
{
	int n = (int)v.size(), n_pri = 0;
	if (n == 0) return 0;
	HProf hp_(HP_MARK);
	HReg *a = v.data();
	static thread_local std::vector<int> z;   // recycled from read to read
	for (int i = 0; i < n; ++i) {
		a[i].sub = a[i].alt_sc = 0; a[i].secondary = a[i].secondary_all = -1; a[i].hash = hash_64(id + i);
		if (!a[i].is_alt) ++n_pri;
	}
	ks_introsort((size_t)n, a, [](const HReg &x, const HReg &y) {
		return x.score > y.score || (x.score == y.score && (x.is_alt < y.is_alt || (x.is_alt == y.is_alt && x.hash < y.hash)));
	});
	mark_primary_core(opt, n, a, z);
	for (int i = 0; i < n; ++i) {
		HReg *p = &a[i];
		p->secondary_all = i;   // rank in the first round
		if (!p->is_alt && p->secondary >= 0 && a[p->secondary].is_alt) p->alt_sc = a[p->secondary].score;
	}
	if (n_pri >= 0 && n_pri < n) {
		z.resize(n);
		if (n_pri > 0)
			ks_introsort((size_t)n, a, [](const HReg &x, const HReg &y) {
				return x.is_alt < y.is_alt || (x.is_alt == y.is_alt && (x.score > y.score || (x.score == y.score && x.hash < y.hash)));
			});
		for (int i = 0; i < n; ++i) z[a[i].secondary_all] = i;
		for (int i = 0; i < n; ++i) {
			if (a[i].secondary >= 0) {
				a[i].secondary_all = z[a[i].secondary];
				if (a[i].is_alt) a[i].secondary = INT_MAX;
			} else a[i].secondary_all = -1;
		}
		if (n_pri > 0) {   // second round over the primary-assembly hits only
			for (int i = 0; i < n_pri; ++i) { a[i].sub = 0; a[i].secondary = -1; }
			mark_primary_core(opt, n_pri, a, z);
		}
	} else {
		for (int i = 0; i < n; ++i) a[i].secondary_all = a[i].secondary;
	}
	return n_pri;
}

// Single-end mapping quality of a hit from its score, the best competing score and how much of it its seeds cover.  The
// double-precision expressions are the reference's, operation for operation (src/bwamem.c:952-975): a product evaluated in another
// order rounds differently and moves a MAPQ by one.
int approx_mapq_se(const mem_opt_t *opt, const HReg *a)
{
	// the competitor: the best overlapping hit, at least a minimal seed's worth, or the best other chain of the same region
	int rival = a->sub ? a->sub : opt->min_seed_len * opt->a;
	if (a->csub > rival) rival = a->csub;
	if (rival >= a->score) return 0;
	const int q_span = a->qe - a->qb, r_span = (int)(a->re - a->rb);
	const int span = q_span > r_span ? q_span : r_span;
	const double identity = 1. - (double)(span * opt->a - a->score) / (opt->a + opt->b) / span;
	int q = 0;
	if (a->score != 0) {
		if (opt->mapQ_coef_len > 0) {   // -Q: the length-scaled form
			double f = span < opt->mapQ_coef_len ? 1. : opt->mapQ_coef_fac / log(span);
			f *= identity * identity;
			q = (int)(6.02 * (a->score - rival) / opt->a * f * f + .499);
		} else {
			q = (int)(30.0 * (1. - (double)rival / a->score) * log(a->seedcov) + .499);
			if (identity < 0.95) q = (int)(q * identity * identity + .499);
		}
	}
	if (a->sub_n > 0) q -= (int)(4.343 * log(a->sub_n + 1) + .499);   // many equally good competitors
	q = q > 60 ? 60 : q < 0 ? 0 : q;
	return (int)(q * (1. - a->frac_rep) + .499);                       // seeds in repeats count for less
}

// -5: of the primary hits worth reporting (not ALT, score >= T) the one that starts first in the read becomes a[0]; every
// `secondary` link that named one of the two exchanged slots follows it (src/bwamem.c:977-1001).
void reorder_primary5(int T, HRegV &a)
{
	const int n = (int)a.size();
	auto reportable_primary = [&](const HReg &r) { return r.secondary < 0 && !r.is_alt && r.score >= T; };
	int count = 0, first = -1;
	for (int k = 0; k < n; ++k) {
		if (!reportable_primary(a[k])) continue;
		++count;
		if (first < 0 || a[k].qb < a[first].qb) first = k;   // the earliest of equal starts stays
	}
	if (count <= 1 || first == 0) return;
	std::swap(a[0], a[first]);
	auto relink = [&](int &link) { if (link == 0) link = first; else if (link == first) link = 0; };
	for (int k = 1; k < n; ++k) { relink(a[k].secondary); relink(a[k].secondary_all); }
}

// ---------------------------------------------------------------------------
// CIGAR / MD / NM
// ---------------------------------------------------------------------------
static void put_int(std::string &s, long v)
{
	char buf[32];
	int l = 0;
	if (v == 0) { s.push_back('0'); return; }
	unsigned long x = v < 0 ? (unsigned long)(-v) : (unsigned long)v;
	while (x) { buf[l++] = '0' + x % 10; x /= 10; }
	if (v < 0) buf[l++] = '-';
	while (l) s.push_back(buf[--l]);
}

// Global re-alignment of query against [rb,re); returns false when the interval is unusable
// (the reference then returns a NULL cigar and leaves *score untouched).
bool gen_cigar2(const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, int w_, int64_t l_pac, const uint8_t *pac, int l_query,
                uint8_t *query, int64_t rb, int64_t re, int *score, std::vector<uint32_t> *cigar, std::string *md, int *NM)
{
	if (cigar) cigar->clear();
	if (NM) *NM = -1;
	if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return false;
	bool ok;
	std::vector<uint8_t> rseq = bns_get_seq(l_pac, pac, rb, re, &ok);
	int64_t rlen = ok ? (int64_t)rseq.size() : 0;
	if (re - rb != rlen) return false;
	if (rb >= l_pac) {   // reverse both so that gaps end up left-aligned on the forward strand
		std::reverse(query, query + l_query);
		std::reverse(rseq.begin(), rseq.end());
	}
	if (l_query == re - rb && w_ == 0) {
		if (cigar) cigar->push_back((uint32_t)l_query << 4 | 0);
		*score = 0;
		for (int i = 0; i < l_query; ++i) *score += mat[rseq[i] * 5 + query[i]];
	} else {
		int max_ins = (int)((double)(((l_query + 1) >> 1) * mat[0] - o_ins) / e_ins + 1.);
		int max_del = (int)((double)(((l_query + 1) >> 1) * mat[0] - o_del) / e_del + 1.);
		int max_gap = max_ins > max_del ? max_ins : max_del;
		max_gap = max_gap > 1 ? max_gap : 1;
		int w = (max_gap + abs((int)rlen - l_query) + 1) >> 1;
		w = w < w_ ? w : w_;
		int min_w = abs((int)rlen - l_query) + 3;
		w = w > min_w ? w : min_w;
		HProf hp_(HP_GLOBAL2);
		*score = ksw_global2(l_query, query, (int)rlen, rseq.data(), mat, o_del, e_del, o_ins, e_ins, w, cigar);
	}
	if (NM && cigar) {
		const char *int2base = rb < l_pac ? "ACGTN" : "TGCAN";
		int x = 0, y = 0, u = 0, n_mm = 0, n_gap = 0, nc = (int)cigar->size();
		md->clear();
		for (int k = 0; k < nc; ++k) {
			int op = (*cigar)[k] & 0xf, len = (*cigar)[k] >> 4;
			if (op == 0) {
				for (int i = 0; i < len; ++i) {
					if (query[x + i] != rseq[y + i]) {
						put_int(*md, u);
						md->push_back(int2base[rseq[y + i]]);
						++n_mm; u = 0;
					} else ++u;
				}
				x += len; y += len;
			} else if (op == 2) {
				if (k > 0 && k < nc - 1) {   // leading / trailing deletions are squeezed out later
					put_int(*md, u);
					md->push_back('^');
					for (int i = 0; i < len; ++i) md->push_back(int2base[rseq[y + i]]);
					u = 0; n_gap += len;
				}
				y += len;
			} else if (op == 1) { x += len; n_gap += len; }
		}
		put_int(*md, u);
		*NM = n_mm + n_gap;
	}
	if (rb >= l_pac) std::reverse(query, query + l_query);
	return true;
}

static inline int infer_bw(int l1, int l2, int score, int a, int q, int r)
{
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;   // equal lengths need at least two gaps
	int w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
	if (w < abs(l1 - l2)) w = abs(l1 - l2);
	return w;
}

HAln reg2aln(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, int l_query, const char *query_, const HReg *ar,
             AlnCtx *ctx, int read_idx, bool need_mapq)
{
	HProf hp_(HP_REG2ALN);
	HAln a;
	if (ar == 0 || ar->rb < 0 || ar->re < 0) {   // unmapped record
		a.rid = -1; a.pos = -1; a.flag |= 0x4;
		return a;
	}
	int qb = ar->qb, qe = ar->qe, is_rev, NM = -1, score = 0, last_sc = -(1 << 30);
	int64_t rb = ar->rb, re = ar->re;
	const bool collecting = ctx && ctx->mode == AlnCtx::COLLECT;
	if (!collecting && need_mapq) a.mapq = (ar->secondary < 0 ? approx_mapq_se(opt, ar) : 0) & 0xff;   // (logs: not needed to list the request)
	if (ar->secondary >= 0) a.flag |= 0x100;
	int tmp = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt->a, opt->o_del, opt->e_del);
	int w2 = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt->a, opt->o_ins, opt->e_ins);
	w2 = w2 > tmp ? w2 : tmp;
	if (w2 > opt->w) w2 = w2 < ar->w ? w2 : ar->w;
	if (collecting) {   // only note that this region needs its CIGAR; decisions do not depend on it
		AlnReqH rq;
		rq.rb = rb; rq.re = re; rq.read = read_idx; rq.qb = qb; rq.qe = qe; rq.w2 = w2; rq.truesc = ar->truesc; rq.pad = 0;
		ctx->reqs->push_back(rq);
		a.rid = ar->rid;
		a.score = ar->score; a.sub = ar->sub > ar->csub ? ar->sub : ar->csub;
		a.is_alt = ar->is_alt; a.alt_sc = ar->alt_sc;
		return a;
	}
	bool have = false;
	if (ctx && ctx->mode == AlnCtx::REPLAY) {
		const AlnHdrH &h = ctx->hdr[ctx->cursor++];
		if (h.flags == 0) {
			const uint32_t *cg = (const uint32_t *)(ctx->pool + (size_t)h.pool_off * 4);
			a.cigar.assign(cg, cg + h.n_cigar);
			a.md.assign((const char *)(cg + h.n_cigar), h.md_len);
			NM = h.NM; score = h.score;
			have = true;
		}
	}
	if (!have) {
		std::vector<uint8_t> query(l_query);
		for (int i = 0; i < l_query; ++i) query[i] = query_[i] < 5 ? query_[i] : nt4_table[(uint8_t)query_[i]];
		int i = 0;
		std::vector<uint32_t> cg;
		do {
			w2 = w2 < opt->w << 2 ? w2 : opt->w << 2;
			gen_cigar2(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w2, bns->l_pac, pac, qe - qb, &query[qb], rb, re, &score,
			           &cg, &a.md, &NM);
			a.cigar.assign(cg.data(), cg.data() + cg.size());
			if (score == last_sc || w2 == opt->w << 2) break;   // global and local scores may legitimately differ
			last_sc = score;
			w2 <<= 1;
		} while (++i < 3 && score < ar->truesc - opt->a);
	}
	a.NM = (uint32_t)NM & 0x3fffff;
	int64_t pos = bns_depos(bns, rb < bns->l_pac ? rb : re - 1, &is_rev);
	a.is_rev = is_rev;
	if (!a.cigar.empty()) {   // squeeze out a leading or trailing deletion
		if ((a.cigar[0] & 0xf) == 2) {
			pos += a.cigar[0] >> 4;
			a.cigar.pop_front();
		} else if ((a.cigar.back() & 0xf) == 2) a.cigar.pop_back();
	}
	if (qb != 0 || qe != l_query) {   // soft clips
		int clip5 = is_rev ? l_query - qe : qb, clip3 = is_rev ? qb : l_query - qe;
		if (clip5) a.cigar.push_front((uint32_t)clip5 << 4 | 3);
		if (clip3) a.cigar.push_back((uint32_t)clip3 << 4 | 3);
	}
	a.rid = bns_pos2rid(bns, pos);
	assert(a.rid == ar->rid);
	a.pos = pos - bns->anns[a.rid].offset;
	a.score = ar->score; a.sub = ar->sub > ar->csub ? ar->sub : ar->csub;
	a.is_alt = ar->is_alt; a.alt_sc = ar->alt_sc;
	return a;
}

// ---------------------------------------------------------------------------
// SAM text
// ---------------------------------------------------------------------------
static inline int get_rlen(const CigarV &cigar)
{
	int l = 0;
	for (uint32_t c : cigar)
		if ((c & 0xf) == 0 || (c & 0xf) == 2) l += c >> 4;
	return l;
}

// SEQ / QUAL columns: nt4 codes -> letters (complemented and reversed for reverse-strand records), 16 bases at a time
#if defined(__SSE2__)
static inline __m128i rev16(__m128i x)
{
	x = _mm_shuffle_epi32(x, 0x1B);
	x = _mm_shufflelo_epi16(x, 0xB1);
	x = _mm_shufflehi_epi16(x, 0xB1);
	return _mm_or_si128(_mm_srli_epi16(x, 8), _mm_slli_epi16(x, 8));
}
static inline __m128i letters16(__m128i c, const char *lut)
{
	__m128i r = _mm_set1_epi8(lut[4]);
	for (int k = 0; k < 4; ++k) {
		const __m128i m = _mm_cmpeq_epi8(c, _mm_set1_epi8((char)k));
		r = _mm_or_si128(_mm_andnot_si128(m, r), _mm_and_si128(m, _mm_set1_epi8(lut[k])));
	}
	return r;
}
#endif
static inline void put_seq_fwd(char *dst, const char *codes, int n)
{
	int i = 0;
#if defined(__SSE2__)
	for (; i + 16 <= n; i += 16) _mm_storeu_si128((__m128i *)(dst + i), letters16(_mm_loadu_si128((const __m128i *)(codes + i)), "ACGTN"));
#endif
	for (; i < n; ++i) dst[i] = "ACGTN"[(int)codes[i]];
}
static inline void put_seq_rev(char *dst, const char *codes, int n)   // dst[k] = comp(codes[n-1-k])
{
	int i = 0;
#if defined(__SSE2__)
	for (; i + 16 <= n; i += 16)
		_mm_storeu_si128((__m128i *)(dst + i), letters16(rev16(_mm_loadu_si128((const __m128i *)(codes + n - 16 - i))), "TGCAN"));
#endif
	for (; i < n; ++i) dst[i] = "TGCAN"[(int)codes[n - 1 - i]];
}
static inline void put_rev_bytes(char *dst, const char *src, int n)    // dst[k] = src[n-1-k]
{
	int i = 0;
#if defined(__SSE2__)
	for (; i + 16 <= n; i += 16) _mm_storeu_si128((__m128i *)(dst + i), rev16(_mm_loadu_si128((const __m128i *)(src + n - 16 - i))));
#endif
	for (; i < n; ++i) dst[i] = src[n - 1 - i];
}

// the fields of a record that mem_aln2sam adjusts on its private copies of the read's and the mate's alignment
struct AlnView {
	const HAln *a;
	int64_t pos;
	int rid, flag;
	uint32_t is_rev, is_alt, mapq, NM;
	int score, sub, alt_sc;
	bool has_xa, no_cigar;
	explicit AlnView(const HAln *h)
	    : a(h), pos(h->pos), rid(h->rid), flag(h->flag), is_rev(h->is_rev), is_alt(h->is_alt), mapq(h->mapq), NM(h->NM), score(h->score),
	      sub(h->sub), alt_sc(h->alt_sc), has_xa(h->has_xa), no_cigar(false) {}
	int n_cigar() const { return no_cigar ? 0 : (int)a->cigar.size(); }
	const CigarV &cigar() const { return a->cigar; }
};

static void add_cigar(const mem_opt_t *opt, const AlnView *p, std::string &str, int which)
{
	if (p->n_cigar()) {
		for (uint32_t cg : p->cigar()) {
			int c = cg & 0xf;
			if (!(opt->flag & MEM_F_SOFTCLIP) && !p->is_alt && (c == 3 || c == 4)) c = which ? 4 : 3;   // hard clips on supplementary lines
			put_int(str, cg >> 4);
			str.push_back("MIDSH"[c]);
		}
	} else str.push_back('*');
}

static void aln2sam(const mem_opt_t *opt, const bntseq_t *bns, std::string &str, const bseq1_t *s, int n, const HAln *const *list, int which,
                    const HAln *m_)
{
	HProf hp_(HP_ALN2SAM);
	AlnView ptmp(list[which]), *p = &ptmp;
	HAln none;
	AlnView mtmp(m_ ? m_ : &none), *m = m_ ? &mtmp : 0;
	p->flag |= m ? 0x1 : 0;
	p->flag |= p->rid < 0 ? 0x4 : 0;
	p->flag |= m && m->rid < 0 ? 0x8 : 0;
	if (p->rid < 0 && m && m->rid >= 0) { p->rid = m->rid; p->pos = m->pos; p->is_rev = m->is_rev; p->no_cigar = true; }   // unmapped read placed at its mate
	if (m && m->rid < 0 && p->rid >= 0) { m->rid = p->rid; m->pos = p->pos; m->is_rev = p->is_rev; m->no_cigar = true; }
	p->flag |= p->is_rev ? 0x10 : 0;
	p->flag |= m && m->is_rev ? 0x20 : 0;

	str += s->name; str.push_back('\t');
	put_int(str, (p->flag & 0xffff) | (p->flag & 0x10000 ? 0x100 : 0)); str.push_back('\t');
	if (p->rid >= 0) {
		str += bns->anns[p->rid].name; str.push_back('\t');
		put_int(str, p->pos + 1); str.push_back('\t');
		put_int(str, p->mapq); str.push_back('\t');
		add_cigar(opt, p, str, which);
	} else str += "*\t0\t0\t*";
	str.push_back('\t');

	if (m && m->rid >= 0) {   // mate fields
		if (p->rid == m->rid) str.push_back('=');
		else str += bns->anns[m->rid].name;
		str.push_back('\t');
		put_int(str, m->pos + 1); str.push_back('\t');
		if (p->rid == m->rid) {
			int64_t p0 = p->pos + (p->is_rev ? (p->no_cigar ? 0 : get_rlen(p->cigar())) - 1 : 0);
			int64_t p1 = m->pos + (m->is_rev ? (m->no_cigar ? 0 : get_rlen(m->cigar())) - 1 : 0);
			if (m->n_cigar() == 0 || p->n_cigar() == 0) str.push_back('0');
			else put_int(str, -(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
		} else str.push_back('0');
	} else str += "*\t0\t0";
	str.push_back('\t');

	if (p->flag & 0x100) str += "*\t*";   // secondary: no SEQ/QUAL
	else {
		int qb = 0, qe = s->l_seq;
		bool trim = p->n_cigar() && which && !(opt->flag & MEM_F_SOFTCLIP) && !p->is_alt;
		if (!p->is_rev) {
			if (trim) {
				if ((p->cigar()[0] & 0xf) == 4 || (p->cigar()[0] & 0xf) == 3) qb += p->cigar()[0] >> 4;
				if ((p->cigar().back() & 0xf) == 4 || (p->cigar().back() & 0xf) == 3) qe -= p->cigar().back() >> 4;
			}
			size_t at = str.size();
			str.resize(at + (qe - qb));
			put_seq_fwd(&str[at], s->seq + qb, qe - qb);
			str.push_back('\t');
			if (s->qual) str.append(s->qual + qb, qe - qb);
			else str.push_back('*');
		} else {
			if (trim) {
				if ((p->cigar()[0] & 0xf) == 4 || (p->cigar()[0] & 0xf) == 3) qe -= p->cigar()[0] >> 4;
				if ((p->cigar().back() & 0xf) == 4 || (p->cigar().back() & 0xf) == 3) qb += p->cigar().back() >> 4;
			}
			size_t at = str.size();
			str.resize(at + (qe - qb));
			put_seq_rev(&str[at], s->seq + qb, qe - qb);
			str.push_back('\t');
			if (s->qual) {
				at = str.size();
				str.resize(at + (qe - qb));
				put_rev_bytes(&str[at], s->qual + qb, qe - qb);
			} else str.push_back('*');
		}
	}

	if (p->n_cigar()) {
		str += "\tNM:i:"; put_int(str, p->NM);
		str += "\tMD:Z:"; str += p->a->md;
	}
	if (m && m->n_cigar()) { str += "\tMC:Z:"; add_cigar(opt, m, str, which); }
	if (p->score >= 0) { str += "\tAS:i:"; put_int(str, p->score); }
	if (p->sub >= 0) { str += "\tXS:i:"; put_int(str, p->sub); }
	if (bwa_rg_id[0]) { str += "\tRG:Z:"; str += bwa_rg_id; }
	if (!(p->flag & 0x100)) {
		int i;
		for (i = 0; i < n; ++i)
			if (i != which && !(list[i]->flag & 0x100)) break;
		if (i < n) {   // other non-secondary lines of this read: SA tag
			str += "\tSA:Z:";
			for (i = 0; i < n; ++i) {
				const HAln *r = list[i];
				if (i == which || (r->flag & 0x100)) continue;
				str += bns->anns[r->rid].name; str.push_back(',');
				put_int(str, r->pos + 1); str.push_back(',');
				str.push_back("+-"[r->is_rev]); str.push_back(',');
				for (uint32_t cg : r->cigar) { put_int(str, cg >> 4); str.push_back("MIDSH"[cg & 0xf]); }
				str.push_back(','); put_int(str, r->mapq);
				str.push_back(','); put_int(str, r->NM);
				str.push_back(';');
			}
		}
		if (p->alt_sc > 0) {
			char buf[64];
			snprintf(buf, sizeof buf, "\tpa:f:%.3f", (double)p->score / p->alt_sc);
			str += buf;
		}
	}
	if (p->has_xa) { str += "\tXA:Z:"; str += p->a->xa; }
	if (s->comment) { str.push_back('\t'); str += s->comment; }
	if ((opt->flag & MEM_F_REF_HDR) && p->rid >= 0 && bns->anns[p->rid].anno != 0 && bns->anns[p->rid].anno[0] != 0) {
		str += "\tXR:Z:";
		size_t at = str.size();
		str += bns->anns[p->rid].anno;
		for (size_t i = at; i < str.size(); ++i)
			if (str[i] == '\t') str[i] = ' ';
	}
	str.push_back('\n');
}

void aln2sam_pub(const mem_opt_t *opt, const bntseq_t *bns, std::string &str, const bseq1_t *s, int n, const HAln *const *list, int which,
                 const HAln *m)
{
	aln2sam(opt, bns, str, s, n, list, which, m);
}

// XA strings per region (only valid after mark_primary_se); returns false when no region has alternatives
bool gen_alt(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const HRegV &a, int l_query, const char *query,
             std::vector<std::string> &xa, std::vector<char> &has, AlnCtx *ctx, int read_idx)
{
	HProf hp_(HP_GENALT);
	int n = (int)a.size(), tot = 0;
	const double ratio = opt->XA_drop_ratio;
	auto pri_idx = [&](int i) {
		int k = a[i].secondary_all;
		if (k >= 0 && a[i].score >= a[k].score * ratio) return k;
		return -1;
	};
	bool any = false;
	for (int i = 0; i < n && !any; ++i) any = pri_idx(i) >= 0;
	if (!any) return false;   // the usual case: no secondary hit close enough to its primary
	std::vector<int> cnt(n, 0);
	std::vector<char> has_alt(n, 0);
	xa.assign(n, std::string());
	has.assign(n, 0);
	for (int i = 0; i < n; ++i) {
		int r = pri_idx(i);
		if (r >= 0) {
			++cnt[r]; ++tot;
			if (a[i].is_alt) has_alt[r] = 1;
		}
	}
	if (tot == 0) return false;
	for (int i = 0; i < n; ++i) {
		int r = pri_idx(i);
		if (r < 0) continue;
		if (cnt[r] > opt->max_XA_hits_alt || (!has_alt[r] && cnt[r] > opt->max_XA_hits)) continue;
		HAln t = reg2aln(opt, bns, pac, l_query, query, &a[i], ctx, read_idx, false);   // XA entries carry no MAPQ
		has[r] = 1;
		if (ctx && !ctx->text()) continue;
		std::string &s = xa[r];
		s += bns->anns[t.rid].name;
		s.push_back(','); s.push_back("+-"[t.is_rev]); put_int(s, t.pos + 1);
		s.push_back(',');
		for (uint32_t cg : t.cigar) { put_int(s, cg >> 4); s.push_back("MIDSHN"[cg & 0xf]); }
		s.push_back(','); put_int(s, t.NM);
		s.push_back(';');
	}
	return true;
}

static char *to_c(const std::string &s)
{
	char *p = (char *)malloc(s.size() + 1);   // ownership passes to the caller of mem_process_seqs, which free()s it
	if (!p) die("out of memory");
	memcpy(p, s.data(), s.size());
	p[s.size()] = 0;
	return p;
}
char *sam_to_c(const std::string &s) { return to_c(s); }

void reg2sam(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, bseq1_t *s, HRegV &a, int extra_flag, const HAln *m,
             AlnCtx *ctx, int read_idx)
{
	const bool text = !ctx || ctx->text();
	std::vector<std::string> xa;
	std::vector<char> has;
	bool have_xa = false;
	if (!(opt->flag & MEM_F_ALL)) have_xa = gen_alt(opt, bns, pac, a, s->l_seq, s->seq, xa, has, ctx, read_idx);
	std::vector<HAln> aa;
	static thread_local std::string str;   // keeps its capacity from read to read
	str.clear();
	int l = 0;
	for (size_t k = 0; k < a.size(); ++k) {
		HReg *p = &a[k];
		if (p->score < opt->T) continue;
		if (p->secondary >= 0 && (p->is_alt || !(opt->flag & MEM_F_ALL))) continue;
		if (p->secondary >= 0 && p->secondary < INT_MAX && p->score < a[p->secondary].score * opt->drop_ratio) continue;
		aa.push_back(reg2aln(opt, bns, pac, s->l_seq, s->seq, p, ctx, read_idx));
		HAln *q = &aa.back();
		if (have_xa && has[k]) { q->has_xa = true; q->xa = xa[k]; }
		q->flag |= extra_flag;
		if (p->secondary >= 0) q->sub = -1;
		if (l && p->secondary < 0) q->flag |= (opt->flag & MEM_F_NO_MULTI) ? 0x10000 : 0x800;   // supplementary
		if (!(opt->flag & MEM_F_KEEP_SUPP_MAPQ) && l && !p->is_alt && q->mapq > aa[0].mapq) q->mapq = aa[0].mapq;
		++l;
	}
	if (!text) return;
	if (aa.empty()) {
		HAln t = reg2aln(opt, bns, pac, s->l_seq, s->seq, 0);
		t.flag |= extra_flag;
		const HAln *one = &t;
		aln2sam(opt, bns, str, s, 1, &one, 0, m);
	} else {
		const HAln *small[8];
		std::vector<const HAln *> big;
		const HAln **list = small;
		if (aa.size() > 8) { big.resize(aa.size()); list = big.data(); }
		for (size_t k = 0; k < aa.size(); ++k) list[k] = &aa[k];
		for (size_t k = 0; k < aa.size(); ++k) aln2sam(opt, bns, str, s, (int)aa.size(), list, (int)k, m);
	}
	s->sam = to_c(str);
}

} // namespace mbw
