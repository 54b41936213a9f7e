// host_pair.cpp — paired-end stages (host):
//   mem_pestat   insert-size distribution over the whole batch    src/bwamem_pair.c:46-109
//   mem_matesw   mate rescue by local alignment in the window     src/bwamem_pair.c:111-180
//   mem_pair     best consistent pair                             src/bwamem_pair.c:182-243
//   mem_sam_pe   pairing decision, MAPQ and the two SAM records   src/bwamem_pair.c:250-393
#include "host.h"
#include <thread>
#include <algorithm>
#include "sortutil.h"
#include "hprof.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace mbw {

void aln2sam_pub(const mem_opt_t *opt, const bntseq_t *bns, std::string &str, const bseq1_t *s, int n, const HAln *const *list, int which,
                 const HAln *m);
bool gen_alt(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const HRegV &a, int l_query, const char *query,
             std::vector<std::string> &xa, std::vector<char> &has, AlnCtx *ctx, int read_idx);
char *sam_to_c(const std::string &s);

// orientation (0 FF, 1 FR, 2 RF, 3 RR) and distance of two hits given in the doubled coordinate
static inline int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)
{
	int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
	int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;   // mate projected on read 1's strand
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

static int cal_sub(const mem_opt_t *opt, const HRegV &r)
{
	size_t j;
	for (j = 1; j < r.size(); ++j) {
		int b_max = r[j].qb > r[0].qb ? r[j].qb : r[0].qb;
		int e_min = r[j].qe < r[0].qe ? r[j].qe : r[0].qe;
		if (e_min > b_max) {
			int min_l = r[j].qe - r[j].qb < r[0].qe - r[0].qb ? r[j].qe - r[j].qb : r[0].qe - r[0].qb;
			if (e_min - b_max >= min_l * opt->mask_level) break;
		}
	}
	return j < r.size() ? r[j].score : opt->min_seed_len * opt->a;
}

// Votes of the pairs [lo, hi) (pair index) for the insert-size statistics, as histograms over the insert size per
// orientation (4 x (max_ins + 1) counters) — usable when max_ins is small enough to count (pestat_can_count).  The votes
// are only ever used sorted, so they can be gathered in any order, by any number of threads, sub-batch by sub-batch.
bool pestat_can_count(const mem_opt_t *opt) { return opt->max_ins > 0 && opt->max_ins <= (1 << 20); }

void pestat_gather(const mem_opt_t *opt, int64_t l_pac, int lo, int hi, const HRegV *regs, uint64_t *hist)
{
	const size_t stride = (size_t)opt->max_ins + 1;
	for (int i = lo; i < hi; ++i) {
		const HRegV &r0 = regs[i << 1 | 0], &r1 = regs[i << 1 | 1];
		if (r0.empty() || r1.empty()) continue;
		if (cal_sub(opt, r0) > 0.8 * r0[0].score) continue;   // only unique hits vote
		if (cal_sub(opt, r1) > 0.8 * r1[0].score) continue;
		if (r0[0].rid != r1[0].rid) continue;
		int64_t is;
		int dir = infer_dir(l_pac, r0[0].rb, r1[0].rb, &is);
		if (is && is <= opt->max_ins) ++hist[dir * stride + is];
	}
}

static void pestat_from_sorted(const mem_opt_t *opt, std::vector<uint64_t> isize[4], mem_pestat_t pes[4]);

void pestat_from_hist(const mem_opt_t *opt, const uint64_t *hist, mem_pestat_t pes[4])
{
	const size_t stride = (size_t)opt->max_ins + 1;
	std::vector<uint64_t> isize[4];
	for (int d = 0; d < 4; ++d) {
		size_t tot = 0;
		for (size_t v = 0; v < stride; ++v) tot += hist[d * stride + v];
		isize[d].reserve(tot);
		for (size_t v = 0; v < stride; ++v) isize[d].insert(isize[d].end(), hist[d * stride + v], (uint64_t)v);
	}
	pestat_from_sorted(opt, isize, pes);
}

void pestat(const mem_opt_t *opt, int64_t l_pac, int n, const HRegV *regs, mem_pestat_t pes[4], int n_threads)
{
	std::vector<uint64_t> isize[4];
	// The votes are only ever used sorted, so they can be gathered by several threads in any order; with the usual
	// max_ins (10 000) sorting is a counting sort over the insert sizes.
	const int np = n >> 1;
	const bool counting = pestat_can_count(opt);
	int nt = std::max(1, std::min(n_threads, np / 4096));
	if (counting) {
		const size_t stride = (size_t)opt->max_ins + 1;
		std::vector<std::vector<uint64_t>> part(nt, std::vector<uint64_t>());
		auto gather = [&](int t) {
			part[t].assign(4 * stride, 0);
			pestat_gather(opt, l_pac, (int)((int64_t)np * t / nt), (int)((int64_t)np * (t + 1) / nt), regs, part[t].data());
		};
		std::vector<std::thread> th;
		for (int t = 1; t < nt; ++t) th.emplace_back(gather, t);
		gather(0);
		for (auto &t : th) t.join();
		for (int t = 1; t < nt; ++t)
			for (size_t v = 0; v < 4 * stride; ++v) part[0][v] += part[t][v];
		pestat_from_hist(opt, part[0].data(), pes);
		return;
	}
	std::vector<std::vector<uint64_t>> part(nt * 4);
	auto gather = [&](int t) {
		int lo = (int)((int64_t)np * t / nt), hi = (int)((int64_t)np * (t + 1) / nt);
		std::vector<uint64_t> *out = &part[t * 4];
		for (int i = lo; i < hi; ++i) {
			const HRegV &r0 = regs[i << 1 | 0], &r1 = regs[i << 1 | 1];
			if (r0.empty() || r1.empty()) continue;
			if (cal_sub(opt, r0) > 0.8 * r0[0].score) continue;   // only unique hits vote
			if (cal_sub(opt, r1) > 0.8 * r1[0].score) continue;
			if (r0[0].rid != r1[0].rid) continue;
			int64_t is;
			int dir = infer_dir(l_pac, r0[0].rb, r1[0].rb, &is);
			if (is && is <= opt->max_ins) out[dir].push_back(is);
		}
	};
	{
		std::vector<std::thread> th;
		for (int t = 1; t < nt; ++t) th.emplace_back(gather, t);
		gather(0);
		for (auto &t : th) t.join();
	}
	for (int d = 0; d < 4; ++d) {
		for (int t = 0; t < nt; ++t) isize[d].insert(isize[d].end(), part[t * 4 + d].begin(), part[t * 4 + d].end());
		std::sort(isize[d].begin(), isize[d].end());   // plain integers: every sort gives the same array
	}
	pestat_from_sorted(opt, isize, pes);
}

static void pestat_from_sorted(const mem_opt_t *opt, std::vector<uint64_t> isize[4], mem_pestat_t pes[4])
{
	memset(pes, 0, 4 * sizeof(mem_pestat_t));
	if (bwa_verbose >= 3)
		fprintf(stderr, "[M::%s] # candidate unique pairs for (FF, FR, RF, RR): (%ld, %ld, %ld, %ld)\n", "mem_pestat",
		        (long)isize[0].size(), (long)isize[1].size(), (long)isize[2].size(), (long)isize[3].size());
	for (int d = 0; d < 4; ++d) {
		mem_pestat_t *r = &pes[d];
		std::vector<uint64_t> &q = isize[d];
		if (q.size() < 10) {
			fprintf(stderr, "[M::%s] skip orientation %c%c as there are not enough pairs\n", "mem_pestat", "FR"[d >> 1 & 1], "FR"[d & 1]);
			r->failed = 1;
			continue;
		} else fprintf(stderr, "[M::%s] analyzing insert size distribution for orientation %c%c...\n", "mem_pestat", "FR"[d >> 1 & 1], "FR"[d & 1]);
		int p25 = (int)q[(int)(.25 * q.size() + .499)];
		int p50 = (int)q[(int)(.50 * q.size() + .499)];
		int p75 = (int)q[(int)(.75 * q.size() + .499)];
		r->low = (int)(p25 - 2.0 * (p75 - p25) + .499);
		if (r->low < 1) r->low = 1;
		r->high = (int)(p75 + 2.0 * (p75 - p25) + .499);
		fprintf(stderr, "[M::%s] (25, 50, 75) percentile: (%d, %d, %d)\n", "mem_pestat", p25, p50, p75);
		fprintf(stderr, "[M::%s] low and high boundaries for computing mean and std.dev: (%d, %d)\n", "mem_pestat", r->low, r->high);
		size_t x = 0;
		r->avg = 0;
		for (uint64_t v : q)
			if (v >= (uint64_t)r->low && v <= (uint64_t)r->high) { r->avg += v; ++x; }
		r->avg /= x;
		r->std = 0;
		for (uint64_t v : q)
			if (v >= (uint64_t)r->low && v <= (uint64_t)r->high) r->std += (v - r->avg) * (v - r->avg);
		r->std = sqrt(r->std / x);
		fprintf(stderr, "[M::%s] mean and std.dev: (%.2f, %.2f)\n", "mem_pestat", r->avg, r->std);
		r->low = (int)(p25 - 3.0 * (p75 - p25) + .499);
		r->high = (int)(p75 + 3.0 * (p75 - p25) + .499);
		if (r->low > r->avg - 4.0 * r->std) r->low = (int)(r->avg - 4.0 * r->std + .499);
		if (r->high < r->avg + 4.0 * r->std) r->high = (int)(r->avg + 4.0 * r->std + .499);
		if (r->low < 1) r->low = 1;
		fprintf(stderr, "[M::%s] low and high boundaries for proper pairs: (%d, %d)\n", "mem_pestat", r->low, r->high);
	}
	size_t max = 0;
	for (int d = 0; d < 4; ++d) max = max > isize[d].size() ? max : isize[d].size();
	for (int d = 0; d < 4; ++d)
		if (pes[d].failed == 0 && isize[d].size() < max * 0.05) {
			pes[d].failed = 1;
			fprintf(stderr, "[M::%s] skip orientation %c%c\n", "mem_pestat", "FR"[d >> 1 & 1], "FR"[d & 1]);
		}
}

// Window of the reference in which the mate of hit `a` is searched for orientation r (src/bwamem_pair.c:131-149), clipped
// to the contig the way bns_fetch_seq does (src/bntseq.c:423-446).  False when mem_matesw would not align there.
static bool matesw_window(const mem_opt_t *opt, const bntseq_t *bns, const mem_pestat_t pes[4], const HReg *a, int l_ms, int r, int64_t *rb_,
                          int64_t *re_, int *is_rev_)
{
	const int64_t l_pac = bns->l_pac;
	int is_rev = (r >> 1 != (r & 1));   // mate must be reverse-complemented
	int is_larger = !(r >> 1);          // mate lies at the larger coordinate
	int64_t rb, re;
	if (!is_rev) {
		rb = is_larger ? a->rb + pes[r].low : a->rb - pes[r].high;
		re = (is_larger ? a->rb + pes[r].high : a->rb - pes[r].low) + l_ms;
	} else {
		rb = (is_larger ? a->rb + pes[r].low : a->rb - pes[r].high) - l_ms;
		re = is_larger ? a->rb + pes[r].high : a->rb - pes[r].low;
	}
	if (rb < 0) rb = 0;
	if (re > l_pac << 1) re = l_pac << 1;
	*is_rev_ = is_rev;
	if (rb >= re) return false;
	int rev_mid;
	const int64_t mid = (rb + re) >> 1;
	int rid = bns_pos2rid(bns, bns_depos(bns, mid, &rev_mid));
	int64_t far_beg = bns->anns[rid].offset, far_end = far_beg + bns->anns[rid].len;
	if (rev_mid) {
		int64_t t = far_beg;
		far_beg = (l_pac << 1) - far_end;
		far_end = (l_pac << 1) - t;
	}
	rb = std::max(rb, far_beg);
	re = std::min(re, far_end);
	*rb_ = rb; *re_ = re;
	return a->rid == rid && re - rb >= opt->min_seed_len;
}

static inline void matesw_skips(const bntseq_t *bns, const mem_pestat_t pes[4], const HReg *a, const HRegV &ma, int skip[4])
{
	for (int r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
	for (size_t i = 0; i < ma.size(); ++i) {   // orientations already explained by an existing mate hit
		int64_t dist;
		int r = infer_dir(bns->l_pac, a->rb, ma[i].rb, &dist);
		if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
	}
}

// mem_sort_dedup_patch(opt, 0, 0, 0, ...) (src/bwamem.c:437-489, as src/bwamem_pair.c:176 calls it) on "a list that is a fixed
// point of that pass + one new hit b", without sorting: returns false when the outcome depends on how the reference's unstable
// sort orders hits with equal end positions (the caller then runs the pass itself).
// Why this is the same list.  Among the hits of a fixed point no two are redundant (every pair within reach of each other was
// compared while both were alive, :451-459), and nothing is merged without the reference sequence (mem_patch_reg returns 0,
// :408), so every event of the pass involves b.  The pass walks the hits by increasing end position; a hit p looks back over
// the hits that end before it (same contig, p->rb < q->re + max_chain_gap: a contiguous run) and, per redundant pair, the
// lower score dies, the earlier one on a tie (:456-459), and a hit that dies stops looking.  So: the partners R of b are found
// in one scan; if all of them score less than b they all die and b stays, whatever the order; otherwise b first meets those that
// end before it, nearest first (killing them until one scores more, which kills b), then those that end after it meet b in
// turn (dying until one scores at least b's, which kills b) — an order that is only defined by the end positions when these are
// distinct.  What is left is already sorted by (score desc, rb, qb) except for b, and two hits with equal (score, rb, qb)
// are always redundant, so the second half of the pass (:475-487) reduces to putting b in its place.
static bool insert_into_settled(const mem_opt_t *opt, HRegV &ma, const HReg &b)
{
	const int n = (int)ma.size();
	int R_[48], nR = 0;
	bool any_ge = false, tie = false;
	auto redundant = [&](const HReg *q, const HReg *p) -> bool {   // q ends first; p looks back at it (src/bwamem.c:448-455)
		if (!(p->rb < q->re + opt->max_chain_gap)) return false;
		const int64_t orr = q->re - p->rb;
		const int64_t oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
		const int64_t mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
		const int64_t mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
		return orr > opt->mask_level_redun * mr && oq > opt->mask_level_redun * mq;
	};
	for (int i = 0; i < n; ++i) {
		const HReg &e = ma[i];
		if (e.rid != b.rid) continue;
		// (equal end positions: either may be the one the pass visits first)
		const bool red = e.re < b.re ? redundant(&e, &b) : e.re > b.re ? redundant(&b, &e) : (redundant(&e, &b) || redundant(&b, &e));
		if (!red) continue;
		if (nR == 48) return false;
		R_[nR++] = i;
		if (e.score >= b.score) any_ge = true;
		if (e.re == b.re) tie = true;
	}
	bool b_alive = true;
	uint64_t dead[4] = {0, 0, 0, 0};   // positions in R_
	if (nR && !any_ge) {
		for (int k = 0; k < nR; ++k) dead[k >> 6] |= 1ull << (k & 63);
	} else if (nR) {
		if (tie) return false;
		// R by end position (insertion sort of a handful of indices); equal end positions among them: the order is the sort's
		int o[48];
		for (int k = 0; k < nR; ++k) o[k] = k;
		for (int a = 1; a < nR; ++a)
			for (int c = a; c > 0 && ma[R_[o[c]]].re < ma[R_[o[c - 1]]].re; --c) std::swap(o[c], o[c - 1]);
		for (int a = 1; a < nR; ++a)
			if (ma[R_[o[a]]].re == ma[R_[o[a - 1]]].re) return false;
		int first_after = 0;
		while (first_after < nR && ma[R_[o[first_after]]].re < b.re) ++first_after;
		for (int a = first_after - 1; a >= 0 && b_alive; --a) {   // b looks back, nearest first
			if (b.score < ma[R_[o[a]]].score) b_alive = false;
			else dead[o[a] >> 6] |= 1ull << (o[a] & 63);
		}
		for (int a = first_after; a < nR && b_alive; ++a) {        // the hits behind b meet it in turn
			if (ma[R_[o[a]]].score < b.score) dead[o[a] >> 6] |= 1ull << (o[a] & 63);
			else b_alive = false;
		}
	}
	// compact the survivors (they keep their order), then b into its place by (score desc, rb, qb)
	if (dead[0] | dead[1] | dead[2] | dead[3]) {
		int m = 0, k = 0;
		for (int i = 0; i < n; ++i) {
			if (k < nR && R_[k] == i) {
				const bool d = (dead[k >> 6] >> (k & 63)) & 1;
				++k;
				if (d) continue;
			}
			if (m != i) ma[m] = ma[i];
			++m;
		}
		ma.resize(m);
	}
	if (b_alive) {
		size_t at = 0;
		while (at < ma.size() && (ma[at].score > b.score || (ma[at].score == b.score && (ma[at].rb < b.rb || (ma[at].rb == b.rb && ma[at].qb < b.qb))))) ++at;
		HReg nb = b;
		nb.n_comp = 1;
		ma.insert(ma.begin() + at, nb);
	}
	ma.settled = true;
	return true;
}

static int matesw(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4], const HReg *a, int l_ms,
                  const uint8_t *ms, HRegV &ma, const MswCtx *mctx, int mate_read)
{
	HProf hp_(HP_MATESW);
	int64_t l_pac = bns->l_pac;
	int skip[4], n = 0;
	matesw_skips(bns, pes, a, ma, skip);
	if (skip[0] + skip[1] + skip[2] + skip[3] == 4) return 0;
	for (int r = 0; r < 4; ++r) {
		if (skip[r]) continue;
		int is_rev;
		int64_t rb = 0, re = 0;
		if (matesw_window(opt, bns, pes, a, l_ms, r, &rb, &re, &is_rev)) {
			KswResult aln;
			const MswResH *hit = nullptr;
			if (mctx)
				for (int k = 0; k < mctx->n; ++k)
					if (mctx->req[k].read == mate_read && mctx->req[k].rb == rb && mctx->req[k].re == re && mctx->req[k].is_rev == is_rev) {
						if (mctx->res[k].flags == 0) hit = &mctx->res[k];
						break;
					}
			if (hit) {   // computed on the device (msw_kernel.hip)
				aln.score = hit->score; aln.te = hit->te; aln.qe = hit->qe; aln.score2 = hit->score2; aln.te2 = hit->te2; aln.tb = hit->tb; aln.qb = hit->qb;
			} else {
				HProf hp2_(HP_ALIGN2);
				std::vector<uint8_t> seq(ms, ms + l_ms);
				if (is_rev)
					for (int i = 0; i < l_ms; ++i) seq[l_ms - 1 - i] = ms[i] < 4 ? 3 - ms[i] : 4;
				int rid;
				int64_t fb = rb, fe = re;
				std::vector<uint8_t> ref = bns_fetch_seq(bns, pac, &fb, (rb + re) >> 1, &fe, &rid);
				int xtra = KSW_XSUBO | KSW_XSTART | (l_ms * opt->a < 250 ? KSW_XBYTE : 0) | (opt->min_seed_len * opt->a);
				aln = ksw_align2(l_ms, seq.data(), (int)(re - rb), ref.data(), opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, xtra);
			}
			if (aln.score >= opt->min_seed_len && aln.qb >= 0) {
				HReg b;
				b.rid = a->rid;
				b.is_alt = a->is_alt;
				b.qb = is_rev ? l_ms - (aln.qe + 1) : aln.qb;
				b.qe = is_rev ? l_ms - aln.qb : aln.qe + 1;
				b.rb = is_rev ? (l_pac << 1) - (rb + aln.te + 1) : rb + aln.tb;
				b.re = is_rev ? (l_pac << 1) - (rb + aln.tb) : rb + aln.te + 1;
				b.score = aln.score;
				b.csub = aln.score2;
				b.secondary = -1;
				b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
				// A list that is a fixed point of the redundancy pass takes the new hit without the pass being run (insert_into_settled);
				// otherwise: keep `ma` ordered by score (insert before the first strictly lower score) for the pass below
				if (!(ma.settled && insert_into_settled(opt, ma, b))) {
					size_t at = 0;
					while (at < ma.size() && !(ma[at].score < b.score)) ++at;
					ma.insert(ma.begin() + at, b);
					ma.settled = false;
				}
			}
			++n;
		}
		// The reference re-runs mem_sort_dedup_patch(opt, 0, 0, 0, ...) after every orientation once an alignment has been
		// attempted (src/bwamem_pair.c:176).  On a list that is already a fixed point of that pass — every pair of survivors was
		// compared while both were alive, nothing is patched without the reference (bns = 0), the final order is total on
		// (score, rb, qb) — the call returns the list as it is: it is only made when a hit was added since the last pass (or
		// phase 1 merged two hits, HRegV::settled).
		if (n && !ma.settled) sort_dedup_patch(opt, 0, 0, 0, ma);
	}
	return n;
}

// The alignments mem_sam_pe's rescue loop will ask for, judged from the hits as they are before any rescue: a request for
// every (candidate hit, orientation) that is not already explained by a mate hit.  Later rescues of the same pair can
// only make some of them unnecessary (or, through de-duplication, very rarely need one that is not here: matesw then
// computes it on the host), so plugging the device results into the sequential logic changes nothing.
void sam_pe_msw_collect(const mem_opt_t *opt, const bntseq_t *bns, const mem_pestat_t pes[4], const bseq1_t s[2], const HRegV a[2], int read0,
                        int max_tlen, std::vector<MswReqH> &out)
{
	if (opt->flag & MEM_F_NO_RESCUE) return;
	for (int i = 0; i < 2; ++i) {
		if (a[i].empty()) continue;
		const int l_ms = s[!i].l_seq;
		int nb = 0;
		for (size_t j = 0; j < a[i].size() && nb < opt->max_matesw; ++j) {
			if (a[i][j].score < a[i][0].score - opt->pen_unpaired) continue;
			++nb;
			int skip[4];
			matesw_skips(bns, pes, &a[i][j], a[!i], skip);
			for (int r = 0; r < 4; ++r) {
				if (skip[r]) continue;
				MswReqH q;
				int is_rev;
				if (!matesw_window(opt, bns, pes, &a[i][j], l_ms, r, &q.rb, &q.re, &is_rev)) continue;
				if (q.re - q.rb > max_tlen) continue;
				q.read = read0 + !i; q.is_rev = is_rev;
				out.push_back(q);
			}
		}
	}
}

struct Pair64 { uint64_t x, y; };
static inline bool pair_lt(const Pair64 &a, const Pair64 &b) { return a.x < b.x || (a.x == b.x && a.y < b.y); }

static int pair_hits(const mem_opt_t *opt, const bntseq_t *bns, const mem_pestat_t pes[4], HRegV a[2], int id, int *sub, int *n_sub,
                     int z[2], int n_pri[2])
{
	HProf hp_(HP_PAIR);
	static thread_local std::vector<Pair64> v, u;   // recycled from pair to pair
	v.clear(); u.clear();
	int y[4], ret;
	int64_t l_pac = bns->l_pac;
	for (int r = 0; r < 2; ++r)
		for (int i = 0; i < n_pri[r]; ++i) {
			const HReg *e = &a[r][i];
			Pair64 key;
			key.x = e->rb < l_pac ? e->rb : (l_pac << 1) - 1 - e->rb;   // forward-strand position
			key.x = (uint64_t)e->rid << 32 | (key.x - bns->anns[e->rid].offset);
			key.y = (uint64_t)e->score << 32 | i << 2 | (e->rb >= l_pac) << 1 | r;
			v.push_back(key);
		}
	ks_introsort(v.size(), v.data(), pair_lt);
	y[0] = y[1] = y[2] = y[3] = -1;
	for (size_t i = 0; i < v.size(); ++i) {
		for (int r = 0; r < 2; ++r) {
			int dir = r << 1 | (v[i].y >> 1 & 1), which;
			if (pes[dir].failed) continue;
			which = r << 1 | ((v[i].y & 1) ^ 1);
			if (y[which] < 0) continue;
			for (int k = y[which]; k >= 0; --k) {
				if ((int)(v[k].y & 3) != which) continue;
				int64_t dist = (int64_t)v[i].x - v[k].x;
				if (dist > pes[dir].high) break;
				if (dist < pes[dir].low) continue;
				double ns = (dist - pes[dir].avg) / pes[dir].std;
				int q = (int)((v[i].y >> 32) + (v[k].y >> 32) + .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)) * opt->a + .499);
				if (q < 0) q = 0;
				Pair64 p;
				p.y = (uint64_t)k << 32 | i;
				p.x = (uint64_t)q << 32 | (hash_64(p.y ^ id << 8) & 0xffffffffU);
				u.push_back(p);
			}
		}
		y[v[i].y & 3] = (int)i;
	}
	if (!u.empty()) {
		int tmp = opt->a + opt->b;
		tmp = tmp > opt->o_del + opt->e_del ? tmp : opt->o_del + opt->e_del;
		tmp = tmp > opt->o_ins + opt->e_ins ? tmp : opt->o_ins + opt->e_ins;
		ks_introsort(u.size(), u.data(), pair_lt);
		int i = (int)(u.back().y >> 32), k = (int)(u.back().y << 32 >> 32);
		z[v[i].y & 1] = (int)(v[i].y << 32 >> 34);
		z[v[k].y & 1] = (int)(v[k].y << 32 >> 34);
		ret = (int)(u.back().x >> 32);
		*sub = u.size() > 1 ? (int)(u[u.size() - 2].x >> 32) : 0;
		*n_sub = 0;
		for (long j = (long)u.size() - 2; j >= 0; --j)
			if (*sub - (int)(u[j].x >> 32) <= tmp) ++*n_sub;
	} else { ret = 0; *sub = 0; *n_sub = 0; }
	return ret;
}

#define RAW_MAPQ(diff, a) ((int)(6.02 * (diff) / (a) + .499))

// ---- decisions: mate rescue, primary marking, pairing, MAPQ (everything that mutates a[]) ----
void sam_pe_plan(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4], uint64_t id, bseq1_t s[2],
                 HRegV a[2], PairPlan &P, const MswCtx *mctx, int read0)
{
	int n = 0, o, subo = 0, n_sub = 0;
	P = PairPlan();
	if (!(opt->flag & MEM_F_NO_RESCUE)) {   // mate rescue from the best hits of each end
		HRegV b[2];
		HReg bbuf[2][6];   // the copies usually fit here (one or two candidates per end): no heap block per pair
		b[0].attach(bbuf[0], 6); b[1].attach(bbuf[1], 6);
		for (int i = 0; i < 2; ++i)
			for (size_t j = 0; j < a[i].size(); ++j)
				if (a[i][j].score >= a[i][0].score - opt->pen_unpaired) b[i].push_back(a[i][j]);
		for (int i = 0; i < 2; ++i)
			for (size_t j = 0; j < b[i].size() && (int)j < opt->max_matesw; ++j)
				n += matesw(opt, bns, pac, pes, &b[i][j], s[!i].l_seq, (uint8_t *)s[!i].seq, a[!i], mctx, read0 + !i);
	}
	P.n_rescue = n;
	P.n_pri[0] = mark_primary_se(opt, a[0], id << 1 | 0);
	P.n_pri[1] = mark_primary_se(opt, a[1], id << 1 | 1);
	if (opt->flag & MEM_F_PRIMARY5) {
		reorder_primary5(opt->T, a[0]);
		reorder_primary5(opt->T, a[1]);
	}
	if ((opt->flag & MEM_F_NOPAIRING) || !P.n_pri[0] || !P.n_pri[1]) return;
	int *z = P.z;
	if ((o = pair_hits(opt, bns, pes, a, (int)id, &subo, &n_sub, z, P.n_pri)) <= 0) return;
	int is_multi[2], q_pe, score_un, *q_se = P.q_se;
	for (int i = 0; i < 2; ++i) {   // an end with several good primary hits is left to the single-end logic
		int j;
		for (j = 1; j < P.n_pri[i]; ++j)
			if (a[i][j].secondary < 0 && a[i][j].score >= opt->T) break;
		is_multi[i] = j < P.n_pri[i] ? 1 : 0;
	}
	if (is_multi[0] || is_multi[1]) return;
	P.paired = true;
	score_un = a[0][0].score + a[1][0].score - opt->pen_unpaired;
	subo = subo > score_un ? subo : score_un;
	q_pe = RAW_MAPQ(o - subo, opt->a);
	if (n_sub > 0) q_pe -= (int)(4.343 * log(n_sub + 1) + .499);
	if (q_pe < 0) q_pe = 0;
	if (q_pe > 60) q_pe = 60;
	q_pe = (int)(q_pe * (1. - .5 * (a[0][0].frac_rep + a[1][0].frac_rep)) + .499);
	if (o > score_un) {   // the pair beats the two best single-end hits
		HReg *c[2] = {&a[0][z[0]], &a[1][z[1]]};
		for (int i = 0; i < 2; ++i) {
			if (c[i]->secondary >= 0) { c[i]->sub = a[i][c[i]->secondary].score; c[i]->secondary = -2; }
			q_se[i] = approx_mapq_se(opt, c[i]);
		}
		q_se[0] = q_se[0] > q_pe ? q_se[0] : q_pe < q_se[0] + 40 ? q_pe : q_se[0] + 40;
		q_se[1] = q_se[1] > q_pe ? q_se[1] : q_pe < q_se[1] + 40 ? q_pe : q_se[1] + 40;
		P.extra_flag |= 2;
		// cap by the tandem-repeat score
		q_se[0] = q_se[0] < RAW_MAPQ(c[0]->score - c[0]->csub, opt->a) ? q_se[0] : RAW_MAPQ(c[0]->score - c[0]->csub, opt->a);
		q_se[1] = q_se[1] < RAW_MAPQ(c[1]->score - c[1]->csub, opt->a) ? q_se[1] : RAW_MAPQ(c[1]->score - c[1]->csub, opt->a);
	} else {
		z[0] = z[1] = 0;
		q_se[0] = approx_mapq_se(opt, &a[0][0]);
		q_se[1] = approx_mapq_se(opt, &a[1][0]);
	}
	for (int i = 0; i < 2; ++i) {
		int k = a[i][z[i]].secondary_all;
		if (k >= 0 && k < P.n_pri[i]) {   // the chosen hit was secondary: swap roles with its parent
			for (size_t j = 0; j < a[i].size(); ++j)
				if (a[i][j].secondary_all == k || (int)j == k) a[i][j].secondary_all = z[i];
			a[i][z[i]].secondary_all = -1;
		}
	}
}

// ---- emission: CIGARs (through ctx) and SAM text; a[] and the plan are only read ----
void sam_pe_emit(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4], bseq1_t s[2], HRegV a[2],
                 const PairPlan &P, AlnCtx *ctx, int read0)
{
	const bool text = !ctx || ctx->text();
	const int *z = P.z, *n_pri = P.n_pri;
	HAln h[2];
	const size_t req0 = (ctx && ctx->reqs) ? ctx->reqs->size() : 0;   // COLLECT: the pair's first CIGAR request
	if (P.paired) {
		std::vector<std::string> xa[2];
		std::vector<char> has[2];
		bool have_xa[2] = {false, false};
		if (!(opt->flag & MEM_F_ALL))
			for (int i = 0; i < 2; ++i) have_xa[i] = gen_alt(opt, bns, pac, a[i], s[i].l_seq, s[i].seq, xa[i], has[i], ctx, read0 + i);
		const HAln *aa[2][2];   // the lines of each read: the chosen hit, then possibly its best ALT hit
		int n_aa[2] = {0, 0};
		HAln g[2];
		int h_req[2] = {-1, -1};   // COLLECT: the chosen hits' CIGAR requests, counted from the pair's first request
		for (int i = 0; i < 2; ++i) {
			if (!text && ctx->reqs) h_req[i] = (int)(ctx->reqs->size() - req0);
			h[i] = reg2aln(opt, bns, pac, s[i].l_seq, s[i].seq, &a[i][z[i]], ctx, read0 + i, false);   // MAPQ comes from the pairing
			h[i].mapq = P.q_se[i] & 0xff;
			h[i].flag |= 0x40 << i | P.extra_flag;
			if (text && have_xa[i] && has[i][z[i]]) { h[i].has_xa = true; h[i].xa = xa[i][z[i]]; }
			aa[i][n_aa[i]++] = &h[i];
			if (n_pri[i] < (int)a[i].size()) {   // the read also has ALT hits
				HReg *p = &a[i][n_pri[i]];
				if (p->score < opt->T || p->secondary >= 0 || !p->is_alt) continue;
				g[i] = reg2aln(opt, bns, pac, s[i].l_seq, s[i].seq, p, ctx, read0 + i);
				g[i].flag |= 0x800 | 0x40 << i | P.extra_flag;
				if (text && have_xa[i] && has[i][n_pri[i]]) { g[i].has_xa = true; g[i].xa = xa[i][n_pri[i]]; }
				aa[i][n_aa[i]++] = &g[i];
			}
		}
		if (!text) {
			// One line per read, nothing but the standard tags: the device writes these records (sam_kernel.hip).  What it
			// needs besides the CIGAR results: the region, the flag bits and the numbers decided above.
			if (ctx->desc && n_aa[0] == 1 && n_aa[1] == 1 && !h[0].has_xa && !h[1].has_xa && !(have_xa[0] && has[0][z[0]]) &&
			    !(have_xa[1] && has[1][z[1]]) && h[0].alt_sc <= 0 && h[1].alt_sc <= 0 && !s[0].comment && !s[1].comment &&
			    !(opt->flag & (MEM_F_ALL | MEM_F_REF_HDR)) && strcmp(s[0].name, s[1].name) == 0) {
				for (int i = 0; i < 2; ++i) {
					const HReg &r = a[i][z[i]];
					SamDescH &d = ctx->desc[i];
					d.rb = r.rb; d.re = r.re; d.qb = r.qb; d.qe = r.qe; d.rid = r.rid;
					d.req = h_req[i];
					d.flag = h[i].flag; d.mapq = (int32_t)h[i].mapq; d.score = h[i].score; d.sub = h[i].sub;
				}
			}
			return;
		}
		static thread_local std::string str;   // keeps its capacity from pair to pair
		str.clear();
		for (int i = 0; i < n_aa[0]; ++i) aln2sam_pub(opt, bns, str, &s[0], n_aa[0], aa[0], i, &h[1]);
		s[0].sam = sam_to_c(str);
		str.clear();
		for (int i = 0; i < n_aa[1]; ++i) aln2sam_pub(opt, bns, str, &s[1], n_aa[1], aa[1], i, &h[0]);
		s[1].sam = sam_to_c(str);
		if (strcmp(s[0].name, s[1].name) != 0) die("paired reads have different names: \"%s\", \"%s\"", s[0].name, s[1].name);
		return;
	}
	// no usable pair: report the ends independently
	int extra_flag = 1;
	for (int i = 0; i < 2; ++i) {
		int which = -1;
		if (!a[i].empty()) {
			if (a[i][0].score >= opt->T) which = 0;
			else if (n_pri[i] < (int)a[i].size() && a[i][n_pri[i]].score >= opt->T) which = n_pri[i];
		}
		h[i] = reg2aln(opt, bns, pac, s[i].l_seq, s[i].seq, which >= 0 ? &a[i][which] : 0, ctx, read0 + i);
	}
	if (!(opt->flag & MEM_F_NOPAIRING) && h[0].rid == h[1].rid && h[0].rid >= 0) {   // still flag a proper pair if the top hits form one
		int64_t dist;
		int d = infer_dir(bns->l_pac, a[0][0].rb, a[1][0].rb, &dist);
		if (!pes[d].failed && dist >= pes[d].low && dist <= pes[d].high) extra_flag |= 2;
	}
	reg2sam(opt, bns, pac, &s[0], a[0], 0x41 | extra_flag, &h[1], ctx, read0);
	reg2sam(opt, bns, pac, &s[1], a[1], 0x81 | extra_flag, &h[0], ctx, read0 + 1);
	if (text && strcmp(s[0].name, s[1].name) != 0) die("paired reads have different names: \"%s\", \"%s\"", s[0].name, s[1].name);
}

int sam_pe(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4], uint64_t id, bseq1_t s[2], HRegV a[2])
{
	PairPlan P;
	sam_pe_plan(opt, bns, pac, pes, id, s, a, P);
	sam_pe_emit(opt, bns, pac, pes, s, a, P, nullptr, 0);
	return P.n_rescue;
}

} // namespace mbw

// ---- host-logic test hooks (no GPU involved): the library's host stages on the reference's own records ----
namespace {
struct Ref88 {   // mem_alnreg_t as the reference lays it out (src/bwamem.h:59-77)
	int64_t rb, re;
	int32_t qb, qe, rid, score, truesc, sub, alt_sc, csub, sub_n, w, seedcov, secondary, secondary_all, seedlen0;
	int32_t n_comp_is_alt;   // int n_comp:30, is_alt:2
	float frac_rep;
	uint64_t hash;
};
static_assert(sizeof(Ref88) == 88, "mem_alnreg_t is 88 bytes");
void regs_in(const void *regs, int n, mbw::HRegV &v)
{
	const Ref88 *r = (const Ref88 *)regs;
	for (int j = 0; j < n; ++j) {
		const Ref88 &x = r[j];
		mbw::HReg h;
		h.rb = x.rb; h.re = x.re; h.qb = x.qb; h.qe = x.qe; h.rid = x.rid; h.score = x.score; h.truesc = x.truesc; h.sub = x.sub;
		h.alt_sc = x.alt_sc; h.csub = x.csub; h.sub_n = x.sub_n; h.w = x.w; h.seedcov = x.seedcov; h.secondary = x.secondary;
		h.secondary_all = x.secondary_all; h.seedlen0 = x.seedlen0;
		h.n_comp = (int32_t)((uint32_t)x.n_comp_is_alt << 2) >> 2; h.is_alt = (x.n_comp_is_alt >> 30) & 3;
		h.frac_rep = x.frac_rep; h.hash = x.hash;
		v.push_back(h);
	}
}
void regs_out(const mbw::HRegV &v, void *regs)
{
	Ref88 *r = (Ref88 *)regs;
	for (size_t j = 0; j < v.size(); ++j) {
		const mbw::HReg &h = v[j];
		Ref88 &x = r[j];
		x.rb = h.rb; x.re = h.re; x.qb = h.qb; x.qe = h.qe; x.rid = h.rid; x.score = h.score; x.truesc = h.truesc; x.sub = h.sub;
		x.alt_sc = h.alt_sc; x.csub = h.csub; x.sub_n = h.sub_n; x.w = h.w; x.seedcov = h.seedcov; x.secondary = h.secondary;
		x.secondary_all = h.secondary_all; x.seedlen0 = h.seedlen0;
		x.n_comp_is_alt = (int32_t)(((uint32_t)h.n_comp & 0x3fffffffu) | ((uint32_t)h.is_alt & 3u) << 30);
		x.frac_rep = h.frac_rep; x.hash = h.hash;
	}
}
} // namespace

// mem_sam_pe (src/bwamem_pair.c:250-393) as the library's host path runs it — the rescue loop with the host's ksw_align2,
// mem_mark_primary_se, mem_pair, the MAPQ arithmetic, mem_reg2sam with the host's global alignment — on the two ends' regions given as the
// reference's own mem_alnreg_t records, i.e. what the reference's mem_align1_core returns.  s[k].seq holds nt4 codes; s[k].sam is set
// (malloc family).  Returns the number of rescued hits, as the reference does.
extern "C" int mi355x_host_sam_pe(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4], uint64_t id, bseq1_t s[2],
                                  const void *regs0, int n0, const void *regs1, int n1)
{
	mbw::HRegV a[2];
	regs_in(regs0, n0, a[0]);
	regs_in(regs1, n1, a[1]);
	return mbw::sam_pe(opt, bns, pac, pes, id, s, a);
}

// mem_sort_dedup_patch (src/bwamem.c:437-489, with mem_patch_reg :406-435) on the n regions mem_chain2aln left for one read (query: nt4
// codes); the kept regions are written back to regs in their new order.  Returns their number.
extern "C" int mi355x_host_sort_dedup_patch(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, uint8_t *query, void *regs, int n)
{
	mbw::HRegV v;
	regs_in(regs, n, v);
	const int m = mbw::sort_dedup_patch(opt, bns, pac, query, v);
	regs_out(v, regs);
	return m;
}

// the single-end half of worker2 (src/bwamem.c:1187-1203): mem_mark_primary_se with id, mem_reorder_primary5 under -5, mem_reg2sam.
// s->seq: nt4 codes; s->sam is set.
extern "C" void mi355x_host_reg2sam_se(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, bseq1_t *s, const void *regs, int n, int64_t id)
{
	mbw::HRegV v;
	regs_in(regs, n, v);
	mbw::mark_primary_se(opt, v, id);
	if (opt->flag & MEM_F_PRIMARY5) mbw::reorder_primary5(opt->T, v);
	mbw::reg2sam(opt, bns, pac, s, v, 0, nullptr);
}

// mem_pestat (src/bwamem_pair.c:46-109) of the library's host path on the regions of n reads (mates interleaved) given as the reference's
// records: regs = all reads' regions one after the other, n_regs[i] = how many read i has.  The messages go to stderr as the reference's do.
extern "C" void mi355x_host_pestat(const mem_opt_t *opt, int64_t l_pac, int n, const void *regs, const int *n_regs, mem_pestat_t pes[4], int n_threads)
{
	std::vector<mbw::HRegV> v((size_t)n);
	const Ref88 *r = (const Ref88 *)regs;
	for (int i = 0; i < n; ++i) { regs_in(r, n_regs[i], v[(size_t)i]); r += n_regs[i]; }
	mbw::pestat(opt, l_pac, n, v.data(), pes, n_threads);
}

// mem_flt_chained_seeds with mem_seed_sw (src/bwamem.c:571-617) of the library's host path — the rescoring of short seeds for reads of
// ~700 bp and more — on chains given by their seeds alone: seeds = (rbeg, qbeg, len, score) records of 24 bytes (mem_seed_t), n_seeds[c] of
// chain c, one chain after the other.  The kept seeds of every chain are written back in place (with their scores), n_seeds[c] updated.
extern "C" void mi355x_host_flt_chained_seeds(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, int l_query, const uint8_t *query, int n_chains,
                                              void *seeds, int *n_seeds)
{
	static_assert(sizeof(mbw::HSeed) == 24, "mem_seed_t is 24 bytes");
	std::vector<mbw::HChain> chains((size_t)n_chains);
	std::vector<mbw::HChain *> ptr;
	mbw::HSeed *sd = (mbw::HSeed *)seeds;
	size_t at = 0;
	for (int c = 0; c < n_chains; ++c) {
		chains[(size_t)c].seeds.assign(sd + at, sd + at + n_seeds[c]);
		at += (size_t)n_seeds[c];
		ptr.push_back(&chains[(size_t)c]);
	}
	mbw::filter_chained_seeds(opt, bns, pac, l_query, query, ptr);
	at = 0;
	for (int c = 0; c < n_chains; ++c) {
		const size_t had = (size_t)n_seeds[c];
		n_seeds[c] = (int)chains[(size_t)c].seeds.size();
		for (size_t j = 0; j < chains[(size_t)c].seeds.size(); ++j) sd[at + j] = chains[(size_t)c].seeds[j];
		at += had;
	}
}
