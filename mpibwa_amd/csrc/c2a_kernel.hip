// c2a_kernel.hip — chain -> alignment regions on the device, one wavefront per read.
//
// Device counterpart of mem_chain2aln (src/bwamem.c:632-786) including its
// calls to bns_fetch_seq (the reference window is never materialised: target
// bases are read straight from the 2-bit pac in HBM, src/bntseq.c:398-446)
// and ksw_extend2 (wave_ext.cuh).  The seed loop of one read is sequentially
// dependent (a seed is skipped when an earlier extension of ANY chain of the
// read already covers it, src/bwamem.c:671-706), so one wavefront owns one
// read: the scalar control flow is wave-uniform and every inner loop (the
// containment tests over earlier regions, the overlap test over the other
// seeds, seed coverage, and the DP rows) runs across the 64 lanes.
//
// Floating-point decisions of the reference are resolved on the host into
// integer tables indexed by length (gap[], bound5[], bound3[], ceil95[],
// thr10[]), so the device does integer work only.
#include <hip/hip_runtime.h>
#include "device.h"
#include "wave_ext.cuh"

namespace mbw {

typedef long long i64;

__device__ __forceinline__ i64 wave_min_i64(i64 v)
{
	for (int o = 32; o > 0; o >>= 1) { i64 t = __shfl_xor(v, o); v = t < v ? t : v; }
	return v;
}
__device__ __forceinline__ i64 wave_max_i64(i64 v)
{
	for (int o = 32; o > 0; o >>= 1) { i64 t = __shfl_xor(v, o); v = t > v ? t : v; }
	return v;
}
__device__ __forceinline__ int wave_sum_i32(int v)
{
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	return v;
}

// base at position p of the doubled reference (forward strand followed by its reverse complement)
__device__ __forceinline__ int ref_base(const uint8_t *pac, i64 l_pac, i64 p)
{
	if (p >= l_pac) {
		i64 f = (l_pac << 1) - 1 - p;
		return 3 - ((pac[f >> 2] >> ((~f & 3) << 1)) & 3);
	}
	return (pac[p >> 2] >> ((~p & 3) << 1)) & 3;
}

// wave-uniform values that arrive through vector loads / cross-lane reductions -> scalar registers
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ i64 uni64(i64 v)
{
	const unsigned long long u = (unsigned long long)v;
	return (i64)((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32 | (unsigned)__builtin_amdgcn_readfirstlane((int)u));
}

#define C2A_WAVES 1
#define SRT_MARK 0xFFFFFFFFu

__global__ void __launch_bounds__(64 * C2A_WAVES)
c2a_kernel(C2aParams P, WxParams X, int n_reads, const uint8_t *__restrict__ seq, const int64_t *__restrict__ off,
           const int *__restrict__ lens, const int *__restrict__ chain_beg, const int *__restrict__ chain_cnt, const DevChain *__restrict__ chains,
           const DevSeed *__restrict__ seeds, unsigned int *srt, const int *__restrict__ reg_beg, DevReg *regs, int *n_regs, const int *__restrict__ tab,
           int tab_stride, const uint8_t *__restrict__ pac, unsigned long long *counters, int max_len, const int *__restrict__ order)
{
	extern __shared__ int lds[];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int slot = blockIdx.x * C2A_WAVES + wave;
	if (slot >= n_reads) return;
	// reads with the most seeds are started first, so that the kernel does not end on a few long-running reads
	const int rd = uni(order ? order[slot] : slot);
	const WxLds L = wx_lds(lds + (size_t)wave * wx_lds_ints(max_len), max_len);
	const int *gap = tab, *bound5 = tab + tab_stride, *bound3 = tab + 2 * tab_stride, *ceil95 = tab + 3 * tab_stride,
	          *thr10 = tab + 4 * tab_stride;
	const uint8_t *q = seq + off[rd];
	const int lq = uni(lens[rd]);
	DevReg *av = regs + reg_beg[rd];
	int nav = 0;
	unsigned long long cells = 0, n_ext = 0, n_diff = 0;
	const i64 l_pac = P.l_pac;
	int max_sc = 1;   // largest entry of the scoring matrix: what one more column can add at most
	for (int t = 0; t < 25; ++t) max_sc = X.mat[t] > max_sc ? X.mat[t] : max_sc;

	// one ksw_extend2.  P.early: 1 = stop a row loop as soon as nothing the code below reads can change (wave_ext.cuh);
	// 0 = every row the reference computes; 2 = both, and count the extensions whose used outputs differ
	auto extend = [&](int qlen, auto qf, int tlen, auto tf, int wc, int h0, int clip) -> WxResult {
		++n_ext;
		for (int j = lane; j < qlen; j += 64) L.Qs[j] = (uint8_t)qf(j);
		__builtin_amdgcn_wave_barrier();
		if (P.early == 0) return wave_extend<false>(qlen, tlen, tf, X, wc, h0, L, cells, max_sc);
		const WxResult r = wave_extend<true>(qlen, tlen, tf, X, wc, h0, L, cells, max_sc, clip);
		if (P.early == 2) {
			unsigned long long c2 = 0;
			const WxResult f = wave_extend<false>(qlen, tlen, tf, X, wc, h0, L, c2, max_sc);
			const bool loc_r = r.gscore <= 0 || r.gscore <= r.score - clip, loc_f = f.gscore <= 0 || f.gscore <= f.score - clip;
			bool same = f.score == r.score && f.qle == r.qle && f.tle == r.tle && f.max_off == r.max_off && loc_r == loc_f;
			if (same && !loc_f) same = f.gtle == r.gtle && f.gscore == r.gscore;
			if (!same) ++n_diff;
		}
		return r;
	};

	const int ci_beg = uni(chain_beg[rd]), ci_cnt = uni(chain_cnt[rd]);
	const int ci_end = ci_beg + (ci_cnt > 0 ? ci_cnt : 0);
	for (int ci = ci_beg; ci < ci_end; ++ci) {
		DevChain C = chains[ci];
		C.n_seeds = uni(C.n_seeds); C.seed_beg = uni(C.seed_beg); C.rid = uni(C.rid);
		C.far_beg = uni64(C.far_beg); C.far_end = uni64(C.far_end);
		const int n = C.n_seeds;
		if (n == 0) continue;
		const DevSeed *sd = seeds + C.seed_beg;
		unsigned int *ord = srt + C.seed_beg;
		// widest reference span any seed of the chain could reach (src/bwamem.c:642-658)
		i64 lo = l_pac << 1, hi = 0;
		for (int i = lane; i < n; i += 64) {
			const DevSeed t = sd[i];
			i64 b = t.rbeg - (t.qbeg + gap[t.qbeg]);
			int tail = lq - t.qbeg - t.len;
			i64 e = t.rbeg + t.len + (tail + gap[tail]);
			lo = b < lo ? b : lo;
			hi = e > hi ? e : hi;
		}
		i64 rmax0 = uni64(wave_min_i64(lo)), rmax1 = uni64(wave_max_i64(hi));
		rmax0 = rmax0 > 0 ? rmax0 : 0;
		rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
		if (rmax0 < l_pac && l_pac < rmax1) {   // never cross the strand boundary
			if (sd[0].rbeg < l_pac) rmax1 = l_pac;
			else rmax0 = l_pac;
		}
		// bns_fetch_seq clamps to the contig that holds the first seed
		rmax0 = rmax0 > C.far_beg ? rmax0 : C.far_beg;
		rmax1 = rmax1 < C.far_end ? rmax1 : C.far_end;

		for (int k = n - 1; k >= 0; --k) {
			DevSeed s = sd[uni((int)ord[k])];
			s.rbeg = uni64(s.rbeg); s.qbeg = uni(s.qbeg); s.len = uni(s.len);
			// ---- is the seed already inside an earlier extension of this read? ----
			bool hit = false;
			for (int i0 = 0; i0 < nav && !hit; i0 += 64) {
				int i = i0 + lane;
				bool h = false;
				if (i < nav) {
					const DevReg p = av[i];
					if (!(s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) &&
					    !(s.len - p.seedlen0 > thr10[lq])) {
						int qd = s.qbeg - p.qb; i64 rd_ = s.rbeg - p.rb;
						int mg = gap[qd < rd_ ? qd : (int)rd_];
						int w = mg < p.w ? mg : p.w;
						if (qd - rd_ < w && rd_ - qd < w) h = true;
						else {
							qd = p.qe - (s.qbeg + s.len); rd_ = p.re - (s.rbeg + s.len);
							mg = gap[qd < rd_ ? qd : (int)rd_];
							w = mg < p.w ? mg : p.w;
							if (qd - rd_ < w && rd_ - qd < w) h = true;
						}
					}
				}
				hit = __ballot(h) != 0;
			}
			if (hit) {   // extend anyway if a long overlapping seed of the chain sits on another diagonal
				bool other = false;
				for (int i0 = k + 1; i0 < n && !other; i0 += 64) {
					int i = i0 + lane;
					bool h = false;
					if (i < n && ord[i] != SRT_MARK) {
						const DevSeed t = sd[ord[i]];
						if (!(t.len < ceil95[s.len])) {
							if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) h = true;
							if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) h = true;
						}
					}
					other = __ballot(h) != 0;
				}
				if (!other) {
					if (lane == 0) ord[k] = SRT_MARK;
					continue;
				}
			}
			// ---- new region ----
			DevReg a;
			a.rb = a.re = 0; a.qb = a.qe = 0;
			a.rid = C.rid; a.score = a.truesc = -1; a.w = P.w; a.seedcov = 0; a.seedlen0 = s.len; a.frac_rep = C.frac_rep;
			int aw0 = P.w, aw1 = P.w;
			if (s.qbeg) {   // left extension: both sequences reversed
				const int qlen = s.qbeg;
				const i64 tlen64 = s.rbeg - rmax0;
				const int tlen = (int)tlen64;
				WxResult r{};
				for (int i = 0; i < 2; ++i) {
					int prev = a.score;
					aw0 = P.w << i;
					int wc = aw0 < bound5[qlen] ? aw0 : bound5[qlen];
					r = extend(qlen, [&](int j) { return q[s.qbeg - 1 - j]; }, tlen, [&](int t) { return ref_base(pac, l_pac, s.rbeg - 1 - t); }, wc, s.len * P.a, P.pen_clip5);
					a.score = r.score;
					if (a.score == prev || r.max_off < (aw0 >> 1) + (aw0 >> 2)) break;
				}
				if (r.gscore <= 0 || r.gscore <= a.score - P.pen_clip5) {   // local
					a.qb = s.qbeg - r.qle; a.rb = s.rbeg - r.tle; a.truesc = a.score;
				} else {                                                    // reaches the read start
					a.qb = 0; a.rb = s.rbeg - r.gtle; a.truesc = r.gscore;
				}
			} else { a.score = a.truesc = s.len * P.a; a.qb = 0; a.rb = s.rbeg; }
			if (s.qbeg + s.len != lq) {   // right extension
				const int sc0 = a.score, qe = s.qbeg + s.len;
				const i64 re = s.rbeg + s.len - rmax0;
				const int qlen = lq - qe, tlen = (int)(rmax1 - rmax0 - re);
				WxResult r{};
				for (int i = 0; i < 2; ++i) {
					int prev = a.score;
					aw1 = P.w << i;
					int wc = aw1 < bound3[qlen] ? aw1 : bound3[qlen];
					r = extend(qlen, [&](int j) { return q[qe + j]; }, tlen, [&](int t) { return ref_base(pac, l_pac, s.rbeg + s.len + t); }, wc, sc0, P.pen_clip3);
					a.score = r.score;
					if (a.score == prev || r.max_off < (aw1 >> 1) + (aw1 >> 2)) break;
				}
				if (r.gscore <= 0 || r.gscore <= a.score - P.pen_clip3) {
					a.qe = qe + r.qle; a.re = rmax0 + re + r.tle; a.truesc += a.score - sc0;
				} else {
					a.qe = lq; a.re = rmax0 + re + r.gtle; a.truesc += r.gscore - sc0;
				}
			} else { a.qe = lq; a.re = s.rbeg + s.len; }
			// seed coverage
			int cov = 0;
			for (int i = lane; i < n; i += 64) {
				const DevSeed t = sd[i];
				if (t.qbeg >= a.qb && t.qbeg + t.len <= a.qe && t.rbeg >= a.rb && t.rbeg + t.len <= a.re) cov += t.len;
			}
			a.seedcov = wave_sum_i32(cov);
			a.w = aw0 > aw1 ? aw0 : aw1;
			if (lane == 0) av[nav] = a;
			++nav;
			__builtin_amdgcn_wave_barrier();
		}
	}
	if (lane == 0) {
		n_regs[rd] = nav;
		atomicAdd(&counters[0], cells);
		atomicAdd(&counters[1], n_ext);
		if (n_diff) atomicAdd(&counters[3], n_diff);
	}
}

void launch_c2a(void *stream, const C2aParams &P, const ExtParams &ep, int n_reads, const uint8_t *d_seq, const int64_t *d_off,
                const int *d_len, const int *d_chain_beg, const int *d_chain_cnt, const DevChain *d_chains, const DevSeed *d_seeds, unsigned int *d_srt,
                const int *d_reg_beg, DevReg *d_regs, int *d_nregs, const int *d_tab, int tab_stride, const uint8_t *d_pac, unsigned long long *d_counters,
                int max_len, const int *d_order)
{
	WxParams X;
	for (int i = 0; i < 25; ++i) X.mat[i] = ep.mat[i];
	X.o_del = ep.o_del; X.e_del = ep.e_del; X.o_ins = ep.o_ins; X.e_ins = ep.e_ins; X.zdrop = ep.zdrop;
	size_t shmem = (size_t)C2A_WAVES * wx_lds_ints(max_len) * sizeof(int);
	int n_blocks = (n_reads + C2A_WAVES - 1) / C2A_WAVES;
	hipLaunchKernelGGL(c2a_kernel, dim3(n_blocks), dim3(64 * C2A_WAVES), shmem, (hipStream_t)stream, P, X, n_reads, d_seq, d_off,
	                   d_len, d_chain_beg, d_chain_cnt, d_chains, d_seeds, d_srt, d_reg_beg, d_regs, d_nregs, d_tab, tab_stride, d_pac, d_counters,
	                   max_len, d_order);
}

} // namespace mbw
