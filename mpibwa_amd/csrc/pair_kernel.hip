// pair_kernel.hip — the pairing decisions of mem_sam_pe for the pairs that need none of its machinery, a pair per lane.
//
// Device counterpart, for ONE shape of pair, of
//   mem_sam_pe            src/bwamem_pair.c:250-393   (rescue loop, primary marking, pairing, MAPQ, the two records)
//   mem_matesw            src/bwamem_pair.c:111-128   (only its test "is every orientation already explained or failed?")
//   mem_pair              src/bwamem_pair.c:182-243   (one hit per end: one candidate pair)
//   mem_approx_mapq_se    src/bwamem.c:952-976
//   mem_reg2aln           src/bwamem.c:1089-1105      (the band of the final global alignment: infer_bw, :792-800)
//   mem_sort_dedup_patch  src/bwamem.c:437-489        (up to four regions per end, as long as no two of them get as far as
//                                                       mem_patch_reg's global alignment, :406-435)
//   mem_mark_primary_se   src/bwamem.c:493-569        (reads without ALT hits)
// The shape: each end has at most four regions (typically the hit and a few 19-25 bp chance matches), none on an ALT
// contig; mem_matesw would return without aligning for every candidate hit; mem_pair finds a pair; no end has a second
// good primary hit (the single-end logic's case) and no secondary hit is close enough to its primary for an XA entry
// (src/bwamem_extra.c:91-110).  The unstable sorts (ks_introsort) and the hash tie-breaks are the reference's.  That is the bulk of a chunk (three pairs in four on the bench
// workload).  For such a pair the kernel writes what the host's COLLECT pass would have produced — the two requests for
// aln_kernel and the two line descriptors for sam_emit_kernel — so its SAM records are made without the host touching the
// pair at all; every other pair is left to the host (status 0) with its full logic.
//
// Floating point: the reference decides in double (and two float adds).  The expressions are evaluated here in the same
// types and order (the library is built with -ffp-contract=off, IEEE division); the two transcendental sites are host-built
// tables: .721 * log(2 * erfc(|ns| / sqrt 2)) * a per insert size and orientation (src/bwamem_pair.c:218-219), and
// mapQ_coef_fac / log(l) per length (src/bwamem.c:964).
#include <hip/hip_runtime.h>
#include "device.h"

namespace mbw {

typedef long long i64;
typedef unsigned long long u64;

// per read: its first PR_MAXREG regions and how many it has
__global__ void first_reg_kernel(int n, const int *__restrict__ reg_pos, const int *__restrict__ nregs, const DevReg *__restrict__ packed,
                                 DevReg *__restrict__ first, int *__restrict__ nfirst)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int m = nregs[i];
	nfirst[i] = m;
	for (int j = 0; j < m && j < PR_MAXREG; ++j) first[(size_t)i * PR_MAXREG + j] = packed[reg_pos[i] + j];
}
void launch_first_reg(void *stream, int n, const int *d_reg_pos, const int *d_nregs, const DevReg *d_packed, DevReg *d_first, int *d_nfirst)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(first_reg_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, d_reg_pos, d_nregs, d_packed, d_first, d_nfirst);
}

// mem_sort_dedup_patch (src/bwamem.c:437-489) for n <= PR_MAXREG regions of one read.  Returns the number of regions left, or -1 when two regions pass the
// cheap tests of mem_patch_reg (:411-423) and the reference would go on to align across them (the host's case).
// ks_introsort (src/ksort.h:176-226) for 3 <= n <= 16 is ONE median-of-three partition of the whole range followed by an
// insertion sort (no sub-range is long enough to be pushed); its order of equal keys is that of these very swaps.  `o` holds
// the element numbers, lt compares two of them.
template <class LT>
__device__ __forceinline__ void small_introsort(int n, int *o, LT lt)
{
	if (n < 2) return;
	if (n == 2) {
		if (lt(o[1], o[0])) { const int x = o[0]; o[0] = o[1]; o[1] = x; }
		return;
	}
	{
		const int t = n - 1;
		int i = 0, j = t, k = i + ((j - i) >> 1) + 1;
		if (lt(o[k], o[i])) { if (lt(o[k], o[j])) k = j; }
		else k = lt(o[j], o[i]) ? i : j;
		const int pivot = o[k];
		if (k != t) { const int x = o[k]; o[k] = o[t]; o[t] = x; }
		for (;;) {
			do ++i; while (lt(o[i], pivot));
			do --j; while (i <= j && lt(pivot, o[j]));
			if (j <= i) break;
			const int x = o[i]; o[i] = o[j]; o[j] = x;
		}
		const int x = o[i]; o[i] = o[t]; o[t] = x;
	}
	for (int i = 1; i < n; ++i)
		for (int j = i; j > 0 && lt(o[j], o[j - 1]); --j) { const int x = o[j]; o[j] = o[j - 1]; o[j - 1] = x; }
}
template <class LT>
__device__ __forceinline__ void sort_regs(int n, DevReg *a, LT lt)
{
	int o[PR_MAXREG];
	DevReg t[PR_MAXREG];
	for (int i = 0; i < n; ++i) { o[i] = i; t[i] = a[i]; }
	small_introsort(n, o, [&](int x, int y) { return lt(t[x], t[y]); });
	for (int i = 0; i < n; ++i) a[i] = t[o[i]];
}

__device__ __forceinline__ int dedup_small(const PairParams &P, DevReg *a, int n)
{
	if (n <= 1) return n;
	sort_regs(n, a, [](const DevReg &x, const DevReg &y) { return x.re < y.re; });   // by END position
	for (int i = 1; i < n; ++i) {
		DevReg *p = &a[i];
		if (p->rid != a[i - 1].rid || p->rb >= a[i - 1].re + P.max_chain_gap) continue;
		for (int j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + P.max_chain_gap; --j) {
			DevReg *q = &a[j];
			if (q->qe == q->qb) continue;   // already excluded
			const i64 orr = q->re - p->rb;
			const i64 oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
			const i64 mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
			const i64 mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
			if (orr > P.mask_level_redun * mr && oq > P.mask_level_redun * mq) {   // one of the two is redundant
				if (p->score < q->score) { p->qe = p->qb; break; }
				else q->qe = q->qb;
			} else if (q->rb < p->rb) {   // mem_patch_reg(q, p): would it align?
				const DevReg *x = q, *y = p;
				if (x->rb < P.l_pac && y->rb >= P.l_pac) continue;
				if (x->qb >= y->qb || x->qe >= y->qe || x->re >= y->re) continue;   // not colinear
				int w = (int)((x->re - y->rb) - (x->qe - y->qb));
				w = w > 0 ? w : -w;
				double r = (double)(x->re - y->rb) / (y->re - x->rb) - (double)(x->qe - y->qb) / (y->qe - x->qb);
				r = r > 0. ? r : -r;
				if (x->re < y->rb || x->qe < y->qb) {
					if (w > P.w << 1 || r >= 0.05f) continue;
				} else if (w > P.w << 2 || r >= 0.05f * 2) continue;
				return -1;
			}
		}
	}
	int m = 0;
	for (int i = 0; i < n; ++i)
		if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
	n = m;
	sort_regs(n, a, [](const DevReg &x, const DevReg &y) {   // by score, then position
		return x.score > y.score || (x.score == y.score && (x.rb < y.rb || (x.rb == y.rb && x.qb < y.qb)));
	});
	for (int i = 1; i < n; ++i)
		if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb) a[i].qe = a[i].qb;
	m = n > 0 ? 1 : 0;
	for (int i = 1; i < n; ++i)
		if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
	return m;
}

// orientation (0 FF, 1 FR, 2 RF, 3 RR) and distance of two hits given in the doubled coordinate (src/bwamem_pair.c:23-30)
__device__ __forceinline__ int infer_dir(i64 l_pac, i64 b1, i64 b2, i64 *dist)
{
	const int r1 = b1 >= l_pac, r2 = b2 >= l_pac;
	const i64 p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}
// a region with the fields the pairing stage adds to it (mem_alnreg_t: sub, sub_n, secondary, secondary_all, hash)
struct PReg {
	DevReg d;
	int sub, sub_n, secondary, secondary_all;
	u64 hash;
};
__device__ __forceinline__ u64 hash_64(u64 key)   // src/utils.h:98-109
{
	key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
	key += (key << 3);   key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
	return key;
}
// true when mem_matesw(hit, the mate's hits) returns at once: every orientation failed or explained by a mate hit (:118-128)
__device__ __forceinline__ bool no_rescue_needed(const PairParams &P, const DevReg &hit, const PReg *ma, int n_ma)
{
	int skip[4];
	for (int r = 0; r < 4; ++r) skip[r] = P.failed[r] ? 1 : 0;
	for (int i = 0; i < n_ma; ++i) {
		i64 dist;
		const int r = infer_dir(P.l_pac, hit.rb, ma[i].d.rb, &dist);
		if (dist >= P.low[r] && dist <= P.high[r]) skip[r] = 1;
	}
	return skip[0] + skip[1] + skip[2] + skip[3] == 4;
}
#define RAW_MAPQ(diff, a) ((int)(6.02 * (diff) / (a) + .499))
// mem_approx_mapq_se (src/bwamem.c:952-976) with csub = 0 (no hit of these reads comes from mate rescue)
__device__ __forceinline__ int mapq_se(const PairParams &P, const PReg &r, const double *__restrict__ ltab)
{
	const int sub = r.sub ? r.sub : P.min_seed_len * P.a;
	if (sub >= r.d.score) return 0;
	const int l = r.d.qe - r.d.qb > r.d.re - r.d.rb ? r.d.qe - r.d.qb : (int)(r.d.re - r.d.rb);
	const double identity = 1. - (double)(l * P.a - r.d.score) / (P.a + P.b) / l;
	int mapq;
	if (r.d.score == 0) mapq = 0;
	else {
		double tmp = ltab[l];
		tmp *= identity * identity;
		mapq = (int)(6.02 * (r.d.score - sub) / P.a * tmp * tmp + .499);
	}
	if (r.sub_n > 0) mapq -= P.lnq[r.sub_n];
	if (mapq > 60) mapq = 60;
	if (mapq < 0) mapq = 0;
	mapq = (int)(mapq * (1. - r.d.frac_rep) + .499);
	return mapq;
}
__device__ __forceinline__ int infer_bw(int l1, int l2, int score, int a, int q, int r)
{
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	int w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
	const int d = l1 > l2 ? l1 - l2 : l2 - l1;
	if (w < d) w = d;
	return w;
}
// mem_mark_primary_se (src/bwamem.c:521-569) for a read without ALT hits; mem_mark_primary_se_core :493-519
__device__ __forceinline__ void mark_primary(const PairParams &P, PReg *a, int n, u64 id)
{
	for (int i = 0; i < n; ++i) { a[i].sub = 0; a[i].secondary = a[i].secondary_all = -1; a[i].hash = hash_64(id + i); }
	{
		int o[PR_MAXREG];
		PReg t[PR_MAXREG];
		for (int i = 0; i < n; ++i) { o[i] = i; t[i] = a[i]; }
		small_introsort(n, o, [&](int x, int y) { return t[x].d.score > t[y].d.score || (t[x].d.score == t[y].d.score && t[x].hash < t[y].hash); });
		for (int i = 0; i < n; ++i) a[i] = t[o[i]];
	}
	int tmp = P.a + P.b;
	tmp = P.o_del + P.e_del > tmp ? P.o_del + P.e_del : tmp;
	tmp = P.o_ins + P.e_ins > tmp ? P.o_ins + P.e_ins : tmp;
	int z[PR_MAXREG], nz = 1;
	z[0] = 0;
	for (int i = 1; i < n; ++i) {
		int k;
		for (k = 0; k < nz; ++k) {
			const int j = z[k];
			const int b_max = a[j].d.qb > a[i].d.qb ? a[j].d.qb : a[i].d.qb;
			const int e_min = a[j].d.qe < a[i].d.qe ? a[j].d.qe : a[i].d.qe;
			if (e_min > b_max) {
				const int min_l = a[i].d.qe - a[i].d.qb < a[j].d.qe - a[j].d.qb ? a[i].d.qe - a[i].d.qb : a[j].d.qe - a[j].d.qb;
				if (e_min - b_max >= min_l * P.mask_level) {   // significant overlap on the query
					if (a[j].sub == 0) a[j].sub = a[i].d.score;
					if (a[j].d.score - a[i].d.score <= tmp) ++a[j].sub_n;
					break;
				}
			}
		}
		if (k == nz) z[nz++] = i;
		else a[i].secondary = z[k];
	}
	for (int i = 0; i < n; ++i) a[i].secondary_all = a[i].secondary;
}

#define PR_MAXPAIR 16
struct Pair64 { u64 x, y; };
__device__ __forceinline__ bool pair_lt(const Pair64 &a, const Pair64 &b) { return a.x < b.x || (a.x == b.x && a.y < b.y); }
template <int CAP>
__device__ __forceinline__ void sort_pairs(int n, Pair64 *v)
{
	int o[CAP];
	Pair64 t[CAP];
	for (int i = 0; i < n; ++i) { o[i] = i; t[i] = v[i]; }
	small_introsort(n, o, [&](int x, int y) { return pair_lt(t[x], t[y]); });
	for (int i = 0; i < n; ++i) v[i] = t[o[i]];
}

// status[k]: 1 = decided here; 0 / 2..: the host's pair (the number says which test sent it there: statistics only)
__global__ void __launch_bounds__(64)
pair_simple_kernel(PairParams P, int n_pairs, const DevReg *__restrict__ first, const int *__restrict__ nfirst, const uint8_t *__restrict__ pair_ok,
                   const i64 *__restrict__ ann_off, const uint8_t *__restrict__ ann_alt, const double *__restrict__ ptab, const double *__restrict__ ltab,
                   uint8_t *__restrict__ status, AlnReq *__restrict__ reqs, SamDesc *__restrict__ desc)
{
	const int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n_pairs) return;
	AlnReq none;
	none.rb = none.re = 0; none.read = -1; none.qb = none.qe = none.w2 = none.truesc = none.pad = 0;
	reqs[2 * k] = none; reqs[2 * k + 1] = none;
	desc[2 * k].req = -1; desc[2 * k + 1].req = -1;
	status[k] = 0;
	int n[2] = {nfirst[2 * k], nfirst[2 * k + 1]};
	if (!pair_ok[k]) { status[k] = 2; return; }
	if (n[0] == 0 && n[1] == 0) {   // no hit on either end: the two "unmapped" records need no decision at all (src/bwamem_pair.c:363-391, flags 77 / 141)
		for (int e = 0; e < 2; ++e) {
			SamDesc d;
			d.rb = d.re = 0; d.qb = d.qe = 0; d.req = -3; d.rid = -1;
			d.flag = 0x1 | 0x4 | 0x8 | 0x40 << e; d.mapq = 0; d.score = 0; d.sub = 0;
			desc[2 * k + e] = d;
		}
		status[k] = 1;
		return;
	}
	if (n[0] < 1 || n[1] < 1) { status[k] = 2; return; }
	if (n[0] > PR_MAXREG || n[1] > PR_MAXREG) { status[k] = 3; return; }
	PReg a[2][PR_MAXREG];
	for (int e = 0; e < 2; ++e) {
		DevReg r[PR_MAXREG];
		for (int j = 0; j < n[e]; ++j) r[j] = first[(size_t)(2 * k + e) * PR_MAXREG + j];
		n[e] = dedup_small(P, r, n[e]);
		if (n[e] < 0) { status[k] = 4; return; }   // two hits the host has to try to patch
		for (int j = 0; j < n[e]; ++j) {
			a[e][j].d = r[j]; a[e][j].sub = a[e][j].sub_n = 0; a[e][j].secondary = a[e][j].secondary_all = -1; a[e][j].hash = 0;
			if (ann_alt[r[j].rid]) { status[k] = 6; return; }
			const int l = r[j].qe - r[j].qb > r[j].re - r[j].rb ? r[j].qe - r[j].qb : (int)(r[j].re - r[j].rb);
			if (l >= P.ltab_n || l <= 0) { status[k] = 6; return; }
		}
	}
	// the rescue loop would not align anything (src/bwamem_pair.c:263-272): every candidate hit is explained by the mate's hits
	if (!P.no_rescue)
		for (int e = 0; e < 2; ++e) {
			int nb = 0;
			for (int j = 0; j < n[e] && nb < P.max_matesw; ++j) {
				if (a[e][j].d.score < a[e][0].d.score - P.pen_unpaired) continue;
				++nb;
				if (!no_rescue_needed(P, a[e][j].d, a[!e], n[!e])) { status[k] = 7; return; }
			}
		}
	const u64 id = P.id0 + (u64)k;
	mark_primary(P, a[0], n[0], id << 1 | 0);
	mark_primary(P, a[1], n[1], id << 1 | 1);
	// mem_pair (src/bwamem_pair.c:182-243)
	Pair64 v[2 * PR_MAXREG], u[PR_MAXPAIR];
	int nv = 0, nu = 0;
	for (int r = 0; r < 2; ++r)
		for (int i = 0; i < n[r]; ++i) {
			const DevReg &e = a[r][i].d;
			Pair64 key;
			key.x = (u64)(e.rb < P.l_pac ? e.rb : (P.l_pac << 1) - 1 - e.rb);
			key.x = (u64)e.rid << 32 | (key.x - (u64)ann_off[e.rid]);
			key.y = (u64)e.score << 32 | (u64)(i << 2 | (e.rb >= P.l_pac) << 1 | r);
			v[nv++] = key;
		}
	sort_pairs<2 * PR_MAXREG>(nv, v);
	int y[4] = {-1, -1, -1, -1};
	const int idi = (int)((unsigned)(int)id << 8);
	for (int i = 0; i < nv; ++i) {
		for (int r = 0; r < 2; ++r) {
			const int dir = r << 1 | (int)(v[i].y >> 1 & 1);
			if (P.failed[dir]) continue;
			const int which = r << 1 | (int)((v[i].y & 1) ^ 1);
			if (y[which] < 0) continue;
			for (int kk = y[which]; kk >= 0; --kk) {
				if ((int)(v[kk].y & 3) != which) continue;
				const i64 dist = (i64)v[i].x - (i64)v[kk].x;
				if (dist > P.high[dir]) break;
				if (dist < P.low[dir]) continue;
				int q = (int)((double)((v[i].y >> 32) + (v[kk].y >> 32)) + ptab[P.tab_off[dir] + (int)(dist - P.low[dir])] + .499);
				if (q < 0) q = 0;
				Pair64 p;
				p.y = (u64)kk << 32 | (u64)i;
				p.x = (u64)q << 32 | (hash_64(p.y ^ (u64)(i64)idi) & 0xffffffffU);
				if (nu < PR_MAXPAIR) u[nu] = p;
				++nu;
			}
		}
		y[v[i].y & 3] = i;
	}
	if (nu > PR_MAXPAIR) { status[k] = 3; return; }   // (small_introsort: at most 16 elements)
	if (nu == 0) { status[k] = 8; return; }   // no pair in a proper orientation and distance: the host reports the ends independently
	int tmp = P.a + P.b;
	tmp = tmp > P.o_del + P.e_del ? tmp : P.o_del + P.e_del;
	tmp = tmp > P.o_ins + P.e_ins ? tmp : P.o_ins + P.e_ins;
	sort_pairs<PR_MAXPAIR>(nu, u);
	int z[2];
	{
		const int i = (int)(u[nu - 1].y >> 32), kk = (int)(u[nu - 1].y << 32 >> 32);
		z[v[i].y & 1] = (int)(v[i].y << 32 >> 34);
		z[v[kk].y & 1] = (int)(v[kk].y << 32 >> 34);
	}
	const int o = (int)(u[nu - 1].x >> 32);
	int subo = nu > 1 ? (int)(u[nu - 2].x >> 32) : 0, n_sub = 0;
	for (int j = nu - 2; j >= 0; --j)
		if (subo - (int)(u[j].x >> 32) <= tmp) ++n_sub;
	if (o <= 0) { status[k] = 9; return; }
	for (int e = 0; e < 2; ++e)   // an end with several good primary hits is left to the single-end logic (src/bwamem_pair.c:303-309)
		for (int j = 1; j < n[e]; ++j)
			if (a[e][j].secondary < 0 && a[e][j].d.score >= P.T) { status[k] = 10; return; }
	const int score_un = a[0][0].d.score + a[1][0].d.score - P.pen_unpaired;
	subo = subo > score_un ? subo : score_un;
	int q_pe = RAW_MAPQ(o - subo, P.a);
	if (n_sub > 0) q_pe -= P.lnq[n_sub];
	if (q_pe < 0) q_pe = 0;
	if (q_pe > 60) q_pe = 60;
	q_pe = (int)(q_pe * (1. - .5 * (a[0][0].d.frac_rep + a[1][0].d.frac_rep)) + .499);
	int q_se[2], extra_flag = 1;   // (0x1: PairPlan::extra_flag starts at 1)
	if (o > score_un) {   // the pair beats the two best single-end hits
		for (int e = 0; e < 2; ++e) {
			PReg &c = a[e][z[e]];
			if (c.secondary >= 0) { c.sub = a[e][c.secondary].d.score; c.secondary = -2; }
			q_se[e] = mapq_se(P, c, ltab);
		}
		q_se[0] = q_se[0] > q_pe ? q_se[0] : q_pe < q_se[0] + 40 ? q_pe : q_se[0] + 40;
		q_se[1] = q_se[1] > q_pe ? q_se[1] : q_pe < q_se[1] + 40 ? q_pe : q_se[1] + 40;
		extra_flag |= 2;
		const int ca = RAW_MAPQ(a[0][z[0]].d.score, P.a), cb = RAW_MAPQ(a[1][z[1]].d.score, P.a);   // the tandem-repeat cap with csub = 0
		q_se[0] = q_se[0] < ca ? q_se[0] : ca;
		q_se[1] = q_se[1] < cb ? q_se[1] : cb;
	} else {
		z[0] = z[1] = 0;
		q_se[0] = mapq_se(P, a[0][0], ltab);
		q_se[1] = mapq_se(P, a[1][0], ltab);
	}
	for (int e = 0; e < 2; ++e) {   // the chosen hit was secondary: swap roles with its parent (src/bwamem_pair.c:332-339)
		const int kk = a[e][z[e]].secondary_all;
		if (kk >= 0 && kk < n[e]) {
			for (int j = 0; j < n[e]; ++j)
				if (a[e][j].secondary_all == kk || j == kk) a[e][j].secondary_all = z[e];
			a[e][z[e]].secondary_all = -1;
		}
	}
	// a secondary hit close enough to its primary gets an XA entry (src/bwamem_extra.c:91-110): the host's kind of record
	for (int e = 0; e < 2; ++e)
		for (int j = 0; j < n[e]; ++j) {
			const int kk = a[e][j].secondary_all;
			if (kk >= 0 && a[e][j].d.score >= a[e][kk].d.score * (double)P.XA_drop_ratio) { status[k] = 11; return; }
		}
	for (int e = 0; e < 2; ++e) {
		const PReg &R = a[e][z[e]];
		const int l1 = R.d.qe - R.d.qb, l2 = (int)(R.d.re - R.d.rb);
		const int t2 = infer_bw(l1, l2, R.d.truesc, P.a, P.o_del, P.e_del);
		int w2 = infer_bw(l1, l2, R.d.truesc, P.a, P.o_ins, P.e_ins);
		w2 = w2 > t2 ? w2 : t2;
		if (w2 > P.w) w2 = w2 < R.d.w ? w2 : R.d.w;
		AlnReq q;
		q.rb = R.d.rb; q.re = R.d.re; q.read = 2 * k + e; q.qb = R.d.qb; q.qe = R.d.qe; q.w2 = w2; q.truesc = R.d.truesc; q.pad = 0;
		reqs[2 * k + e] = q;
		SamDesc d;
		d.rb = R.d.rb; d.re = R.d.re; d.qb = R.d.qb; d.qe = R.d.qe; d.req = e; d.rid = R.d.rid;
		d.flag = 0x40 << e | extra_flag; d.mapq = q_se[e] & 0xff; d.score = R.d.score; d.sub = R.sub;
		desc[2 * k + e] = d;
	}
	status[k] = 1;
}

void launch_pair_simple(void *stream, const PairParams &P, int n_pairs, const DevReg *d_first, const int *d_nfirst, const uint8_t *d_ok,
                        const int64_t *d_ann_off, const uint8_t *d_ann_alt, const double *d_ptab, const double *d_ltab, uint8_t *d_status,
                        AlnReq *d_reqs, SamDesc *d_desc)
{
	if (n_pairs <= 0) return;
	hipLaunchKernelGGL(pair_simple_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, (hipStream_t)stream, P, n_pairs, d_first, d_nfirst, d_ok,
	                   (const i64 *)d_ann_off, d_ann_alt, d_ptab, d_ltab, d_status, d_reqs, d_desc);
}

// the descriptors of the pairs decided on the device go over the (empty) ones the host uploaded for them
__global__ void desc_overlay_kernel(int n_pairs, const uint8_t *__restrict__ status, const SamDesc *__restrict__ from, SamDesc *__restrict__ to)
{
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= 2 * n_pairs) return;
	if (status[r >> 1] == 1) to[r] = from[r];
}
void launch_desc_overlay(void *stream, int n_pairs, const uint8_t *d_status, const SamDesc *d_from, SamDesc *d_to)
{
	if (n_pairs <= 0) return;
	hipLaunchKernelGGL(desc_overlay_kernel, dim3((2 * n_pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_pairs, d_status, d_from, d_to);
}

} // namespace mbw
