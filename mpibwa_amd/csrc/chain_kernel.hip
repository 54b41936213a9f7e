// chain_kernel.hip — seeds -> chains -> filtered chains on the device, one lane per read.
//
// Device counterpart of mem_chain (src/bwamem.c:251-315), test_and_merge (:190-211), mem_chain_weight (:213-237) and
// mem_chain_flt (:327-385) for the reads the reference's own data structure keeps trivial: its ordered map is a B-tree
// whose root holds up to 9 keys (src/kbtree.h, node size 512 B), so a read that never has more than 9 chains lives in one
// sorted array — which is what this kernel keeps, with the B-tree's rules for equal keys (a new key goes right behind the
// FIRST equal one; a lookup that hits equal keys returns the first).  Reads with more chains, more than CK_MAXSEEDS seeds,
// or long enough for mem_flt_chained_seeds to act (l_query >= ~700 bp) are flagged (n_chains = -1) and take the host
// path (host_chain.cpp); on 2x150 bp data that is ~2 % of the reads.
//
// The unstable sort of mem_chain_flt (ks_introsort, src/ksort.h:176-226) is reproduced for what n <= 9 exercises of
// it: n == 2 is a compare-and-swap, 3 <= n <= 17 is ONE median-of-three partition pass followed by an insertion sort.
// Floating-point compares (mask_level, drop_ratio, frac_rep) are IEEE single precision in the same expression shapes as
// the reference (no contraction, no fast-math).
//
// Output per read, in the slots the read's seeds already own (seed_off[r] .. seed_off[r] + n_seeds[r]): the kept chains
// (DevChain, in mem_chain_flt's order, with the reference window of mem_chain2aln), their seeds in the order
// mem_chain2aln visits them (ascending (score, index) — src/bwamem.c:662-667), and the identity visiting order.
// Latency-bound integer work: ~1.5 k instructions per read, a few hundred bytes of HBM per read.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
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
#define CK_MAXSEEDS 64         // first launch: every read, up to this many seeds
#define CK_MAXSEEDS_BIG 255    // second launch: the reads the first one declined for their seed count only (seed numbers are bytes)
enum { F_POS_LO = 0, F_POS_HI, F_FIRST_Q, F_LAST_R_LO, F_LAST_R_HI, F_LAST_Q, F_LAST_LEN, F_RID, F_N, F_W, F_KEPT, F_FIRSTOV, F_NFIELDS };

struct Lds {
	uint32_t *tab;    // [F_NFIELDS * CK_MAXCH][64]
	uint8_t *cid;     // [MAXS + 1][64]     chain id of every seed (255 = in no chain)
	uint8_t *ord;     // [16][64]           chain ids in tree order, later in filter order
	uint8_t *tmp;     // [MAXS + 1][64]     scratch: members of one chain
	int lane;
	__device__ __forceinline__ uint32_t &f(int field, int id) const { return tab[(field * CK_MAXCH + id) * 64 + lane]; }
	__device__ __forceinline__ i64 pos(int id) const { return (i64)((unsigned long long)f(F_POS_HI, id) << 32 | f(F_POS_LO, id)); }
	__device__ __forceinline__ i64 last_r(int id) const { return (i64)((unsigned long long)f(F_LAST_R_HI, id) << 32 | f(F_LAST_R_LO, id)); }
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

// MAXS: seeds per read the instantiation has LDS for.  RETRY: second launch — only the reads the first launch declined
// because they have more than CK_MAXSEEDS seeds (a few per cent of 2x150 bp reads, with ~150 seeds each: they used to be
// a quarter of the host's CPU time per chunk); reads with more than 9 chains stay with the host's B-tree either way.
template <int MAXS, bool RETRY>
__global__ void __launch_bounds__(64)
chain_kernel(ChainParams P, int n_reads, const int *__restrict__ lens, const int *__restrict__ n_seeds, const int *__restrict__ l_rep,
             const i64 *__restrict__ seed_off, const unsigned long long *__restrict__ sa, const int32_t *__restrict__ qbl,
             const i64 *__restrict__ ann_off, const uint8_t *__restrict__ ann_alt, int n_seqs, const int *__restrict__ tab, int tab_stride,
             DevChain *__restrict__ chains, DevSeed *__restrict__ seeds, unsigned int *__restrict__ srt, int *__restrict__ n_chains)
{
	extern __shared__ uint32_t lds_raw[];
	Lds L;
	L.lane = threadIdx.x;
	L.tab = lds_raw;
	L.cid = (uint8_t *)(lds_raw + F_NFIELDS * CK_MAXCH * 64);
	L.ord = L.cid + (MAXS + 1) * 64;
	L.tmp = L.ord + 16 * 64;
	const int lane = threadIdx.x;
	const int rd = blockIdx.x * 64 + lane;
	if (rd >= n_reads) return;
	const int ns = n_seeds[rd], lq = lens[rd];
	const i64 so = seed_off[rd];
	const int *gap = tab, *noflt = tab + 5 * tab_stride;
	if (RETRY) { if (n_chains[rd] != -1 || ns <= CK_MAXSEEDS) return; }
	else if (ns == 0) { n_chains[rd] = 0; return; }
	if (ns > MAXS || !noflt[lq]) { n_chains[rd] = -1; return; }
	const i64 l_pac = P.l_pac;
	auto S_R = [&](int k) -> i64 { return (i64)sa[so + k]; };
	auto S_Q = [&](int k) -> int { return qbl[2 * (so + k)]; };
	auto S_L = [&](int k) -> int { return qbl[2 * (so + k) + 1]; };
	auto CID = [&](int k) -> uint8_t & { return L.cid[k * 64 + lane]; };
	auto ORD = [&](int k) -> uint8_t & { return L.ord[k * 64 + lane]; };
	auto TMP = [&](int k) -> uint8_t & { return L.tmp[k * 64 + lane]; };

	// ---------------- mem_chain: seeds into the ordered map ----------------
	int n_ch = 0;
	i64 c_lo = 0, c_hi = -1;
	int c_rid = -1;
	for (int k = 0; k < ns; ++k) {
		const i64 rb = S_R(k);
		const int qb = S_Q(k), len = S_L(k);
		CID(k) = 255;
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
		// closest chain at or before the seed: first key >= pos, stepped back unless equal (kb_intervalp, single node)
		int f = 0;
		while (f < n_ch && L.pos(ORD(f)) < rb) ++f;
		const int li = (f < n_ch && L.pos(ORD(f)) == rb) ? f : f - 1;
		bool merged = false;
		if (li >= 0) {   // test_and_merge
			const int id = ORD(li);
			const i64 first_r = L.pos(id), last_r = L.last_r(id);
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
						CID(k) = (uint8_t)id;
						merged = true;
					}
				}
			}
		}
		if (merged) continue;
		if (n_ch == CK_MAXCH) { n_chains[rd] = -1; return; }   // the reference's root node would split here: host path
		const int id = n_ch;
		L.f(F_POS_LO, id) = (uint32_t)rb; L.f(F_POS_HI, id) = (uint32_t)((unsigned long long)rb >> 32);
		L.f(F_FIRST_Q, id) = (uint32_t)qb;
		L.f(F_LAST_R_LO, id) = (uint32_t)rb; L.f(F_LAST_R_HI, id) = (uint32_t)((unsigned long long)rb >> 32);
		L.f(F_LAST_Q, id) = (uint32_t)qb; L.f(F_LAST_LEN, id) = (uint32_t)len;
		L.f(F_RID, id) = (uint32_t)rid; L.f(F_N, id) = 1;
		CID(k) = (uint8_t)id;
		for (int t = n_ch; t > li + 1; --t) ORD(t) = ORD(t - 1);   // the new key goes right behind position li
		ORD(li + 1) = (uint8_t)id;
		++n_ch;
	}

	// ---------------- mem_chain_flt ----------------
	int n = 0;
	for (int t = 0; t < n_ch; ++t) {
		const int id = ORD(t);
		// mem_chain_weight: seed coverage of the query, of the reference, the smaller one
		i64 end = 0;
		int w = 0, wq;
		for (int k = 0; k < ns; ++k) {
			if (CID(k) != id) continue;
			const int qb = S_Q(k), len = S_L(k);
			if (qb >= end) w += len;
			else if (qb + len > end) w += (int)(qb + len - end);
			end = end > qb + len ? end : qb + len;
		}
		wq = w; w = 0; end = 0;
		for (int k = 0; k < ns; ++k) {
			if (CID(k) != id) continue;
			const i64 rb = S_R(k);
			const int len = S_L(k);
			if (rb >= end) w += len;
			else if (rb + len > end) w += (int)(rb + len - end);
			end = end > rb + len ? end : rb + len;
		}
		w = w < wq ? w : wq;
		w = w < 1 << 30 ? w : (1 << 30) - 1;
		const uint32_t w29 = (uint32_t)w & 0x1fffffffu;
		L.f(F_W, id) = w29; L.f(F_KEPT, id) = 0; L.f(F_FIRSTOV, id) = 0xffffffffu;
		if ((int)w29 < P.min_chain_weight) continue;
		ORD(n++) = (uint8_t)id;
	}
	if (n == 0) { n_chains[rd] = 0; return; }
	auto W = [&](int t) -> int { return (int)L.f(F_W, ORD(t)); };
	auto LT = [&](int a_id, int b_id) -> bool { return (int)L.f(F_W, a_id) > (int)L.f(F_W, b_id); };   // "less" of the descending sort
	if (n == 2) {
		if (LT(ORD(1), ORD(0))) { uint8_t t = ORD(0); ORD(0) = ORD(1); ORD(1) = t; }
	} else if (n > 2) {
		// one median-of-three partition pass over the whole range (what ks_introsort does before it hands ranges of <= 17
		// elements to the insertion sort)
		{
			int i = 0, j = n - 1, k = i + ((j - i) >> 1) + 1;
			if (LT(ORD(k), ORD(i))) { if (LT(ORD(k), ORD(j))) k = j; }
			else k = LT(ORD(j), ORD(i)) ? i : j;
			const uint8_t pivot = ORD(k);
			if (k != n - 1) { uint8_t t = ORD(k); ORD(k) = ORD(n - 1); ORD(n - 1) = t; }
			for (;;) {
				do ++i; while (LT(ORD(i), pivot));
				do --j; while (i <= j && LT(pivot, ORD(j)));
				if (j <= i) break;
				uint8_t t = ORD(i); ORD(i) = ORD(j); ORD(j) = t;
			}
			uint8_t t = ORD(i); ORD(i) = ORD(n - 1); ORD(n - 1) = t;
		}
		for (int i = 1; i < n; ++i)   // __ks_insertsort
			for (int j = i; j > 0 && LT(ORD(j), ORD(j - 1)); --j) { uint8_t t = ORD(j); ORD(j) = ORD(j - 1); ORD(j - 1) = t; }
	}
	(void)W;
	// pairwise overlap marking; TMP holds the positions (in ORD) of the chains kept so far
	auto BEG = [&](int t) -> int { return (int)L.f(F_FIRST_Q, ORD(t)); };
	auto END = [&](int t) -> int { const int id = ORD(t); return (int)L.f(F_LAST_Q, id) + (int)L.f(F_LAST_LEN, id); };
	auto ALT = [&](int t) -> int { return ann_alt[L.f(F_RID, ORD(t))]; };
	int n_kept = 0;
	L.f(F_KEPT, ORD(0)) = 3;
	TMP(n_kept++) = 0;
	for (int i = 1; i < n; ++i) {
		bool large_ovlp = false;
		int kk;
		for (kk = 0; kk < n_kept; ++kk) {
			const int j = TMP(kk);
			const int b_max = BEG(j) > BEG(i) ? BEG(j) : BEG(i);
			const int e_min = END(j) < END(i) ? END(j) : END(i);
			if (e_min > b_max && (!ALT(j) || ALT(i))) {
				const int li_ = END(i) - BEG(i), lj_ = END(j) - BEG(j);
				const int min_l = li_ < lj_ ? li_ : lj_;
				if ((float)(e_min - b_max) >= (float)min_l * P.mask_level && min_l < P.max_chain_gap) {
					large_ovlp = true;
					if (L.f(F_FIRSTOV, ORD(j)) == 0xffffffffu) L.f(F_FIRSTOV, ORD(j)) = (uint32_t)i;
					const int wi = (int)L.f(F_W, ORD(i)), wj = (int)L.f(F_W, ORD(j));
					if ((float)wi < (float)wj * P.drop_ratio && wj - wi >= P.min_seed_len << 1) break;
				}
			}
		}
		if (kk == n_kept) {
			TMP(n_kept++) = (uint8_t)i;
			L.f(F_KEPT, ORD(i)) = large_ovlp ? 2 : 3;
		}
	}
	for (int kk = 0; kk < n_kept; ++kk) {
		const uint32_t fo = L.f(F_FIRSTOV, ORD(TMP(kk)));
		if (fo != 0xffffffffu) L.f(F_KEPT, ORD(fo)) = 1;
	}
	{
		int i = 0, cnt = 0;
		for (; i < n; ++i) {   // at most max_chain_extend chains with kept = 1 or 2 are extended
			const uint32_t kp = L.f(F_KEPT, ORD(i));
			if (kp == 0 || kp == 3) continue;
			if (++cnt >= P.max_chain_extend) break;
		}
		for (; i < n; ++i)
			if (L.f(F_KEPT, ORD(i)) < 3) L.f(F_KEPT, ORD(i)) = 0;
	}

	// ---------------- emit the kept chains and their seeds ----------------
	const float frac_rep = (float)l_rep[rd] / (float)lq;
	int n_out = 0;
	i64 cursor = so;
	for (int t = 0; t < n; ++t) {
		const int id = ORD(t);
		if (L.f(F_KEPT, id) == 0) continue;
		// members in arrival order, then sorted by (length, arrival index): keys are distinct
		int cs = 0;
		for (int k = 0; k < ns; ++k)
			if (CID(k) == id) TMP(cs++) = (uint8_t)k;
		// TMP was also the kept list: it is no longer needed at this point
		for (int a = 1; a < cs; ++a) {
			const uint8_t v = TMP(a);
			const int lv = S_L(v);
			int b = a;
			while (b > 0 && S_L(TMP(b - 1)) > lv) { TMP(b) = TMP(b - 1); --b; }   // stable: equal lengths keep arrival order
			TMP(b) = v;
		}
		const int rid = (int)L.f(F_RID, id);
		const i64 first_r = L.pos(id);
		i64 lo = l_pac << 1, hi = 0;
		for (int a = 0; a < cs; ++a) {
			const int k = TMP(a);
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
	n_chains[rd] = n_out;
}

// regions from their per-read slots into one dense array (reg_pos = exclusive prefix of n_regs)
__global__ void reg_pack_kernel(int n_reads, const int *__restrict__ reg_beg, const int *__restrict__ n_regs, const int *__restrict__ reg_pos,
                                const DevReg *__restrict__ regs, DevReg *__restrict__ packed)
{
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	const int m = n_regs[r];
	const DevReg *src = regs + reg_beg[r];
	DevReg *dst = packed + reg_pos[r];
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
                     void *d_tmp, size_t tmp_bytes)
{
	if (n_reads <= 0) return;
	HIP_OK(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_nregs, d_reg_pos, n_reads + 1, (hipStream_t)stream));
	hipLaunchKernelGGL(reg_pack_kernel, dim3((n_reads + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_reads, d_reg_beg, d_nregs, d_reg_pos, d_regs,
	                   d_packed);
	HIP_OK(hipGetLastError());
}

static size_t chain_lds_bytes(int maxs) { return (size_t)F_NFIELDS * CK_MAXCH * 64 * 4 + (size_t)(2 * (maxs + 1) + 16) * 64; }

void launch_chain(void *stream, const ChainParams &P, int n_reads, const int *d_len, const int *d_nseeds, const int *d_lrep,
                  const int64_t *d_seed_off, const uint64_t *d_sa, const int32_t *d_qbl, const int64_t *d_ann_off, const uint8_t *d_ann_alt,
                  int n_seqs, const int *d_tab, int tab_stride, DevChain *d_chains, DevSeed *d_seeds, unsigned int *d_srt, int *d_nchains)
{
	if (n_reads <= 0) return;
	const size_t lds = chain_lds_bytes(CK_MAXSEEDS), lds_big = chain_lds_bytes(CK_MAXSEEDS_BIG);
	static bool s_attr = false;
	if (!s_attr) {
		if (lds > 64 * 1024) HIP_OK(hipFuncSetAttribute((const void *)chain_kernel<CK_MAXSEEDS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		if (lds_big > 64 * 1024) HIP_OK(hipFuncSetAttribute((const void *)chain_kernel<CK_MAXSEEDS_BIG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big));
		s_attr = true;
	}
	hipLaunchKernelGGL((chain_kernel<CK_MAXSEEDS, false>), dim3((n_reads + 63) / 64), dim3(64), lds, (hipStream_t)stream, P, n_reads, d_len, d_nseeds, d_lrep,
	                   (const i64 *)d_seed_off, (const unsigned long long *)d_sa, d_qbl, (const i64 *)d_ann_off, d_ann_alt, n_seqs, d_tab, tab_stride,
	                   d_chains, d_seeds, d_srt, d_nchains);
	static const bool retry = getenv("MPIBWA_CHAIN_BIG") == nullptr || atoi(getenv("MPIBWA_CHAIN_BIG")) != 0;
	if (retry)
		hipLaunchKernelGGL((chain_kernel<CK_MAXSEEDS_BIG, true>), dim3((n_reads + 63) / 64), dim3(64), lds_big, (hipStream_t)stream, P, n_reads, d_len, d_nseeds,
		                   d_lrep, (const i64 *)d_seed_off, (const unsigned long long *)d_sa, d_qbl, (const i64 *)d_ann_off, d_ann_alt, n_seqs, d_tab,
		                   tab_stride, d_chains, d_seeds, d_srt, d_nchains);
	HIP_OK(hipGetLastError());
}

} // namespace mbw
