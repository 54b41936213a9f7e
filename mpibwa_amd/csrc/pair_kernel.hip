// pair_kernel.hip — the pairing decisions of mem_sam_pe for the pairs that need none of its machinery, a pair per lane.
//
// Device counterpart, for ONE shape of pair, of
//   mem_sam_pe            src/bwamem_pair.c:250-393   (rescue loop, primary marking, pairing, MAPQ, the two records)
//   mem_matesw            src/bwamem_pair.c:111-128   (only its test "is every orientation already explained or failed?")
//   mem_pair              src/bwamem_pair.c:182-243   (one hit per end: one candidate pair)
//   mem_approx_mapq_se    src/bwamem.c:952-976
//   mem_reg2aln           src/bwamem.c:1089-1105      (the band of the final global alignment: infer_bw, :792-800)
// The shape: each end has exactly ONE region (mem_sort_dedup_patch, mem_mark_primary_se and the XA tag have nothing to
// do: src/bwamem.c:439, :521, src/bwamem_extra.c:100), neither lies on an ALT contig, mem_matesw would return without
// aligning for both ends, and mem_pair finds the pair.  That is the bulk of a chunk (three pairs in four on the bench
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

// one record per read: its first region and how many it has
__global__ void first_reg_kernel(int n, const int *__restrict__ reg_pos, const int *__restrict__ nregs, const DevReg *__restrict__ packed,
                                 DevReg *__restrict__ first, int *__restrict__ nfirst)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int m = nregs[i];
	nfirst[i] = m;
	if (m > 0) first[i] = packed[reg_pos[i]];
}
void launch_first_reg(void *stream, int n, const int *d_reg_pos, const int *d_nregs, const DevReg *d_packed, DevReg *d_first, int *d_nfirst)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(first_reg_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, d_reg_pos, d_nregs, d_packed, d_first, d_nfirst);
}

// orientation (0 FF, 1 FR, 2 RF, 3 RR) and distance of two hits given in the doubled coordinate (src/bwamem_pair.c:23-30)
__device__ __forceinline__ int infer_dir(i64 l_pac, i64 b1, i64 b2, i64 *dist)
{
	const int r1 = b1 >= l_pac, r2 = b2 >= l_pac;
	const i64 p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}
// true when mem_matesw(hit, mate list = {mate}) returns at once: every orientation failed or explained (:118-128)
__device__ __forceinline__ bool no_rescue_needed(const PairParams &P, const DevReg &hit, const DevReg &mate)
{
	int skip[4];
	for (int r = 0; r < 4; ++r) skip[r] = P.failed[r] ? 1 : 0;
	i64 dist;
	const int r = infer_dir(P.l_pac, hit.rb, mate.rb, &dist);
	if (dist >= P.low[r] && dist <= P.high[r]) skip[r] = 1;
	return skip[0] + skip[1] + skip[2] + skip[3] == 4;
}
#define RAW_MAPQ(diff, a) ((int)(6.02 * (diff) / (a) + .499))
// mem_approx_mapq_se for a hit with sub = csub = sub_n = 0 (the only hit of its read)
__device__ __forceinline__ int mapq_se(const PairParams &P, const DevReg &r, const double *__restrict__ ltab)
{
	const int sub = P.min_seed_len * P.a;
	if (sub >= r.score) return 0;
	const int l = r.qe - r.qb > r.re - r.rb ? r.qe - r.qb : (int)(r.re - r.rb);
	const double identity = 1. - (double)(l * P.a - r.score) / (P.a + P.b) / l;
	int mapq;
	if (r.score == 0) mapq = 0;
	else {
		double tmp = ltab[l];
		tmp *= identity * identity;
		mapq = (int)(6.02 * (r.score - sub) / P.a * tmp * tmp + .499);
	}
	if (mapq > 60) mapq = 60;
	if (mapq < 0) mapq = 0;
	mapq = (int)(mapq * (1. - r.frac_rep) + .499);
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

__global__ void __launch_bounds__(256)
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
	if (!pair_ok[k] || nfirst[2 * k] != 1 || nfirst[2 * k + 1] != 1) return;
	const DevReg A = first[2 * k], B = first[2 * k + 1];
	if (ann_alt[A.rid] || ann_alt[B.rid]) return;
	const int la = A.qe - A.qb > A.re - A.rb ? A.qe - A.qb : (int)(A.re - A.rb), lb = B.qe - B.qb > B.re - B.rb ? B.qe - B.qb : (int)(B.re - B.rb);
	if (la >= P.ltab_n || lb >= P.ltab_n || la <= 0 || lb <= 0) return;
	// the rescue loop would not align anything (src/bwamem_pair.c:263-272)
	if (!P.no_rescue && !(no_rescue_needed(P, A, B) && no_rescue_needed(P, B, A))) return;
	// mem_pair with one hit per end: the two keys in sorted order, then the one candidate pair (:192-224)
	const int reva = A.rb >= P.l_pac, revb = B.rb >= P.l_pac;
	const u64 pa = (u64)(reva ? (P.l_pac << 1) - 1 - A.rb : A.rb), pb = (u64)(revb ? (P.l_pac << 1) - 1 - B.rb : B.rb);
	const u64 xa = (u64)A.rid << 32 | (pa - (u64)ann_off[A.rid]), xb = (u64)B.rid << 32 | (pb - (u64)ann_off[B.rid]);
	const u64 ya = (u64)A.score << 32 | (u64)(reva << 1 | 0), yb = (u64)B.score << 32 | (u64)(revb << 1 | 1);
	const bool a_first = xa < xb || (xa == xb && ya < yb);
	const u64 x0 = a_first ? xa : xb, x1 = a_first ? xb : xa, y0 = a_first ? ya : yb, y1 = a_first ? yb : ya;
	const int dir = (int)(y0 >> 1 & 1) << 1 | (int)(y1 >> 1 & 1);
	if (P.failed[dir]) return;
	const i64 dist = (i64)x1 - (i64)x0;
	if (dist > P.high[dir] || dist < P.low[dir]) return;
	int o = (int)((double)((y1 >> 32) + (y0 >> 32)) + ptab[P.tab_off[dir] + (int)(dist - P.low[dir])] + .499);
	if (o < 0) o = 0;
	if (o <= 0) return;   // no usable pair: the host reports the ends independently
	// (no end has several primary hits; sub = n_sub = 0)
	const int score_un = A.score + B.score - P.pen_unpaired;
	const int subo = 0 > score_un ? 0 : score_un;
	int q_pe = RAW_MAPQ(o - subo, P.a);
	if (q_pe < 0) q_pe = 0;
	if (q_pe > 60) q_pe = 60;
	q_pe = (int)(q_pe * (1. - .5 * (A.frac_rep + B.frac_rep)) + .499);
	int q_se[2] = {mapq_se(P, A, ltab), mapq_se(P, B, ltab)}, extra_flag = 1;   // (0x1: PairPlan::extra_flag starts at 1)
	if (o > score_un) {   // the pair beats the two single-end hits
		q_se[0] = q_se[0] > q_pe ? q_se[0] : q_pe < q_se[0] + 40 ? q_pe : q_se[0] + 40;
		q_se[1] = q_se[1] > q_pe ? q_se[1] : q_pe < q_se[1] + 40 ? q_pe : q_se[1] + 40;
		extra_flag |= 2;
		const int ca = RAW_MAPQ(A.score, P.a), cb = RAW_MAPQ(B.score, P.a);   // the tandem-repeat cap with csub = 0
		q_se[0] = q_se[0] < ca ? q_se[0] : ca;
		q_se[1] = q_se[1] < cb ? q_se[1] : cb;
	}
	for (int e = 0; e < 2; ++e) {
		const DevReg &R = e ? B : A;
		const int l1 = R.qe - R.qb, l2 = (int)(R.re - R.rb);
		const int tmp = infer_bw(l1, l2, R.truesc, P.a, P.o_del, P.e_del);
		int w2 = infer_bw(l1, l2, R.truesc, P.a, P.o_ins, P.e_ins);
		w2 = w2 > tmp ? w2 : tmp;
		if (w2 > P.w) w2 = w2 < R.w ? w2 : R.w;
		AlnReq q;
		q.rb = R.rb; q.re = R.re; q.read = 2 * k + e; q.qb = R.qb; q.qe = R.qe; q.w2 = w2; q.truesc = R.truesc; q.pad = 0;
		reqs[2 * k + e] = q;
		SamDesc d;
		d.rb = R.rb; d.re = R.re; d.qb = R.qb; d.qe = R.qe; d.req = e; d.rid = R.rid;
		d.flag = 0x40 << e | extra_flag; d.mapq = q_se[e] & 0xff; d.score = R.score; d.sub = 0;
		desc[2 * k + e] = d;
	}
	status[k] = 1;
}

void launch_pair_simple(void *stream, const PairParams &P, int n_pairs, const DevReg *d_first, const int *d_nfirst, const uint8_t *d_ok,
                        const int64_t *d_ann_off, const uint8_t *d_ann_alt, const double *d_ptab, const double *d_ltab, uint8_t *d_status,
                        AlnReq *d_reqs, SamDesc *d_desc)
{
	if (n_pairs <= 0) return;
	hipLaunchKernelGGL(pair_simple_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, n_pairs, d_first, d_nfirst, d_ok,
	                   (const i64 *)d_ann_off, d_ann_alt, d_ptab, d_ltab, d_status, d_reqs, d_desc);
}

// the descriptors of the pairs decided on the device go over the (empty) ones the host uploaded for them
__global__ void desc_overlay_kernel(int n_pairs, const uint8_t *__restrict__ status, const SamDesc *__restrict__ from, SamDesc *__restrict__ to)
{
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= 2 * n_pairs) return;
	if (status[r >> 1]) to[r] = from[r];
}
void launch_desc_overlay(void *stream, int n_pairs, const uint8_t *d_status, const SamDesc *d_from, SamDesc *d_to)
{
	if (n_pairs <= 0) return;
	hipLaunchKernelGGL(desc_overlay_kernel, dim3((2 * n_pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_pairs, d_status, d_from, d_to);
}

} // namespace mbw
