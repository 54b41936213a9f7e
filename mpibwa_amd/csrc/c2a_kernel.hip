// c2a_kernel.hip — chain -> alignment regions on the device, one wavefront per read.
//
// Device counterpart of mem_chain2aln (src/bwamem.c:632-786) including its calls to bns_fetch_seq (src/bntseq.c:398-446: the
// chain's reference window is decoded from the 2-bit pac into LDS once per chain) and ksw_extend2 (wave_ext.cuh; not run at all
// for a flank with at most one mismatch, see `ungapped` below).  The seed loop of one read is sequentially dependent (a seed is
// skipped when an earlier extension of ANY chain of the read already covers it, src/bwamem.c:671-706), so one wavefront owns one
// read: the scalar control flow is wave-uniform and every inner loop (the containment tests over earlier regions, the overlap
// test over the other seeds, seed coverage, and the DP rows) runs across the 64 lanes.  What the wave works on — the read,
// its chains and seeds, the regions found so far — is staged in LDS at the start (C2A_CAP_*).
//
// Floating-point decisions of the reference are resolved on the host into integer tables indexed by length (gap[], bound5[],
// bound3[], ceil95[], thr10[]), so the device does integer work only.
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

// What a read's wave would otherwise fetch from HBM one dependent load at a time — its bases, its chains, their seeds and visiting
// order, the regions found so far, the reference window of the chain in hand — is staged in LDS with a few coalesced loads.
// Without this the kernel spent two thirds of its time outside the DP (16 ms of 24 per chunk with the DP switched off): a wave
// went through a dozen serial 2-4 us round trips per seed (order -> seed -> regions -> query bases -> window bases -> seeds again).
// Reads beyond the caps (more than C2A_CAP_C chains, C2A_CAP_S seeds, C2A_CAP_R regions, a window longer than the staging
// buffer) use the arrays in HBM for the part that does not fit.
#define C2A_CAP_C 8
#define C2A_CAP_S 64
#define C2A_CAP_R 8
__host__ __device__ inline int c2a_win_cap(int max_len) { return (3 * max_len + 128 + 15) & ~15; }
__host__ __device__ inline size_t c2a_lds_bytes(int max_len)
{
	return (((size_t)wx_lds_ints(max_len) * 4 + 15) & ~(size_t)15) + ((max_len + 15) & ~15) + C2A_CAP_C * sizeof(DevChain) + C2A_CAP_S * (sizeof(DevSeed) + 4) +
	       C2A_CAP_R * sizeof(DevReg) + c2a_win_cap(max_len);
}

__global__ void __launch_bounds__(64 * C2A_WAVES)
c2a_kernel(C2aParams P, WxParams X, int n_reads, const uint8_t *__restrict__ seq, const int64_t *__restrict__ off,
           const int *__restrict__ lens, const int *__restrict__ chain_beg, const int *__restrict__ chain_cnt, const DevChain *__restrict__ chains,
           const DevSeed *__restrict__ seeds, unsigned int *srt, const int *__restrict__ reg_beg, DevReg *regs, int *n_regs, const int *__restrict__ tab,
           int tab_stride, const uint8_t *__restrict__ pac, unsigned long long *counters, int max_len, const int *__restrict__ order, C2aUnits U)
{
	extern __shared__ int lds[];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	int slot = blockIdx.x * C2A_WAVES + wave;
	// the first U.max_units slots are the units of the reads with many chains (c2a_groups.hip): a group of their chains each,
	// independent of the read's other groups; they go first (their reads are the long-running ones)
	const bool unit_mode = slot < U.max_units;
	int u_beg = 0, u_cnt = 0, u_av = 0;
	int rd;
	if (unit_mode) {
		if ((unsigned)slot >= *U.n_units) return;
		u_beg = uni(U.ustart[slot]); u_cnt = uni(U.ustart[slot + 1]) - u_beg;
		rd = uni(U.unit_rd[slot]); u_av = uni(U.unit_av[slot]);
	} else {
		slot -= U.max_units;
		if (slot >= n_reads) return;
		// reads with the most seeds are started first, so that the kernel does not end on a few long-running reads
		rd = uni(order ? order[slot] : slot);
		if (U.max_units > 0 && chain_cnt[rd] > U.heavy_t) return;   // worked off as units
	}
	uint8_t *const lbase = (uint8_t *)lds + (size_t)wave * c2a_lds_bytes(max_len);
	const WxLds L = wx_lds((int *)lbase, max_len);
	uint8_t *const rdl = lbase + (((size_t)wx_lds_ints(max_len) * 4 + 15) & ~(size_t)15);   // the read
	DevChain *const chl = (DevChain *)(rdl + ((max_len + 15) & ~15));               // its chains
	DevSeed *const sdl = (DevSeed *)(chl + C2A_CAP_C);                              // their seeds
	unsigned int *const ordl = (unsigned int *)(sdl + C2A_CAP_S);                   // ... and visiting order (with the "skipped" marks)
	DevReg *const avl = (DevReg *)(ordl + C2A_CAP_S);                               // the regions found so far
	uint8_t *const winl = (uint8_t *)(avl + C2A_CAP_R);                             // reference window of the chain in hand, one base per byte
	const int win_cap = c2a_win_cap(max_len);
	const int *gap = tab, *bound5 = tab + tab_stride, *bound3 = tab + 2 * tab_stride, *ceil95 = tab + 3 * tab_stride,
	          *thr10 = tab + 4 * tab_stride;
	const uint8_t *qg = seq + off[rd];
	const int lq = uni(lens[rd]);
	DevReg *av = regs + (unit_mode ? u_av : reg_beg[rd]);
	const int ci_beg = unit_mode ? 0 : uni(chain_beg[rd]), ci_cnt = unit_mode ? u_cnt : uni(chain_cnt[rd]);
	const int ci_end = ci_beg + (ci_cnt > 0 ? ci_cnt : 0);
	auto CI = [&](int t) -> int { return unit_mode ? uni(U.clist[u_beg + t]) : t; };   // the chain walked at step t
	// stage the read (slots are 16-byte aligned and padded) and the chains
	for (int j = lane * 4; j < lq; j += 256) *(uint32_t *)(rdl + j) = *(const uint32_t *)(qg + j);
	const bool st_c = ci_cnt > 0 && ci_cnt <= C2A_CAP_C;
	if (st_c) {
		if (unit_mode) {
			const int W = (int)(sizeof(DevChain) / 4);
			for (int k = lane; k < ci_cnt * W; k += 64) ((uint32_t *)chl)[k] = ((const uint32_t *)(chains + U.clist[u_beg + k / W]))[k % W];
		} else
			for (int k = lane; k < ci_cnt * (int)(sizeof(DevChain) / 4); k += 64) ((uint32_t *)chl)[k] = ((const uint32_t *)(chains + ci_beg))[k];
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	const uint8_t *q = rdl;
	// the seeds of a read's chains lie back to back in the flat arrays: one range
	int s0 = 0, n_tot = 0;
	bool st_s = false;
	if (st_c) {
		s0 = uni(chl[0].seed_beg);
		int sum = 0;
		bool contiguous = true;
		for (int k = 0; k < ci_cnt; ++k) {
			const int sb = uni(chl[k].seed_beg), sn = uni(chl[k].n_seeds);
			contiguous = contiguous && sb == s0 + sum;
			sum += sn;
		}
		n_tot = sum;
		st_s = contiguous && n_tot <= C2A_CAP_S;
		if (st_s && lane < n_tot) { sdl[lane] = seeds[s0 + lane]; ordl[lane] = srt[s0 + lane]; }
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
	int nav = 0;
	unsigned long long cells = 0, n_ext = 0, n_diff = 0, n_closed = 0;
	const i64 l_pac = P.l_pac;
	int max_sc = 1;   // largest entry of the scoring matrix: what one more column can add at most
	for (int t = 0; t < 25; ++t) max_sc = X.mat[t] > max_sc ? X.mat[t] : max_sc;

	// ---- extensions that need no DP ----
	// When the flank matches the reference along the diagonal with at most ONE mismatch (no ambiguous base, the window at least
	// as long as the flank), every off-diagonal cell of ksw_extend2 stays strictly below the diagonal cell of its row: a cell at
	// distance d from the diagonal has paid for a gap of d and has at most as many match columns, so it is worth at most
	// h0 + a (min(i, j) + 1) - (o + e d), against h0 + a (i + 1) - (a + b) on the diagonal — strictly less whenever
	// a + b < min(o_del, o_ins) + min(e_del, e_ins) (5 < 7 with the default scores).  Then every row maximum is the diagonal
	// cell, max_off = 0, the score at the query end is the diagonal's, reached in row qlen - 1 and never again, and the best cell is
	// the end of the flank or the peak in front of the mismatch, whichever is higher (the earlier one on a tie: the reference
	// updates on "greater" only).  The diagonal must stay alive (h0 > b) and the z-drop test must not fire at the mismatch
	// (a + b <= zdrop).  The flank right behind a seed nearly always starts with the mismatch that ended the seed, so a read with
	// one or two sequencing errors is extended without any DP.  tests/csrc/ungapped_extend_check.c restates the rule on the CPU
	// against the oracle's ksw_extend2 (tests/test_ungapped_extend.py); MPIBWA_C2A_EARLY=2 checks every use in the kernel.
	const int sc_a = X.mat[0], sc_b = -X.mat[1];
	bool plain = sc_a > 0 && sc_b > 0;
	for (int i = 0; i < 4; ++i)
		for (int j = 0; j < 4; ++j) plain = plain && X.mat[i * 5 + j] == (i == j ? sc_a : -sc_b);
	const int g1 = (X.o_del < X.o_ins ? X.o_del : X.o_ins) + (X.e_del < X.e_ins ? X.e_del : X.e_ins);
	auto ungapped = [&](int qlen, int tlen, auto tf, int h0, WxResult &r) -> bool {
		if (!plain || tlen < qlen || qlen <= 0) return false;
		int mm = 0, p = -1;
		bool bad = false;
		for (int j0 = 0; j0 < qlen && !bad && mm <= 1; j0 += 64) {
			const int j = j0 + lane;
			int qb = 0, tb = 0;
			if (j < qlen) { qb = (int)L.Qs[j]; tb = (int)tf(j); }
			bad = __ballot(qb > 3 || tb > 3) != 0;
			const unsigned long long mis = __ballot(qb != tb);
			if (mis) {
				if (p < 0) p = j0 + __ffsll((long long)mis) - 1;
				mm += __popcll(mis);
			}
		}
		if (bad || mm > 1) return false;
		if (mm == 1 && !(sc_a + sc_b < g1 && h0 > sc_b && (X.zdrop <= 0 || sc_a + sc_b <= X.zdrop))) return false;
		const int end = h0 + sc_a * qlen - (mm ? sc_a + sc_b : 0);
		int best = h0, bl = 0;
		if (mm == 0) { best = end; bl = qlen; }
		else {
			if (p > 0) { best = h0 + sc_a * p; bl = p; }
			if (end > best) { best = end; bl = qlen; }
		}
		r.score = best; r.qle = bl; r.tle = bl; r.gtle = qlen; r.gscore = end; r.max_off = 0;
		return true;
	};

	// one ksw_extend2.  P.early: 1 = no DP when the closed form above applies, and row loops that stop as soon as nothing the code
	// below reads can change (wave_ext.cuh); 0 = every row the reference computes; 2 = both, and count the extensions whose outputs differ
	auto extend = [&](int qlen, auto qf, int tlen, auto tf, int wc, int h0, int clip) -> WxResult {
		++n_ext;
		for (int j = lane; j < qlen; j += 64) L.Qs[j] = (uint8_t)qf(j);
		__builtin_amdgcn_wave_barrier();
		if (P.early == 0) return wave_extend<false>(qlen, tlen, tf, X, wc, h0, L, cells, max_sc);
		WxResult r;
		const bool closed = ungapped(qlen, tlen, tf, h0, r);
		if (closed) ++n_closed;
		else r = wave_extend<true>(qlen, tlen, tf, X, wc, h0, L, cells, max_sc, clip);
		if (P.early == 2) {
			unsigned long long c2 = 0;
			const WxResult f = wave_extend<false>(qlen, tlen, tf, X, wc, h0, L, c2, max_sc);
			const bool loc_r = r.gscore <= 0 || r.gscore <= r.score - clip, loc_f = f.gscore <= 0 || f.gscore <= f.score - clip;
			bool same = f.score == r.score && f.qle == r.qle && f.tle == r.tle && f.max_off == r.max_off && loc_r == loc_f;
			if (same && (!loc_f || closed)) same = f.gtle == r.gtle && f.gscore == r.gscore;   // the closed form claims all six
			if (!same) ++n_diff;
		}
		return r;
	};

	for (int ct = ci_beg; ct < ci_end; ++ct) {
		const int ci = CI(ct);
		DevChain C = st_c ? chl[ct - ci_beg] : chains[ci];
		C.n_seeds = uni(C.n_seeds); C.seed_beg = uni(C.seed_beg); C.rid = uni(C.rid);
		C.far_beg = uni64(C.far_beg); C.far_end = uni64(C.far_end);
		const int n = C.n_seeds;
		const int nav0 = nav;
		if (unit_mode && lane == 0) { U.c_rabs[ci] = u_av + nav0; U.c_rcnt[ci] = 0; }
		if (n == 0) continue;
		// the chain's seeds and order: the staged copies, or the flat arrays
		const DevSeed *sdg = seeds + C.seed_beg;
		unsigned int *ordg = srt + C.seed_beg;
		int sl = C.seed_beg - s0;   // first staged seed of the chain
		bool st_sc = st_s;
		if (unit_mode && !st_s && n <= C2A_CAP_S) {   // a unit whose chains do not lie back to back stages them one at a time
			if (lane < n) { sdl[lane] = sdg[lane]; ordl[lane] = ordg[lane]; }
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			st_sc = true; sl = 0;
		}
		auto SD = [&](int i) -> DevSeed { return st_sc ? sdl[sl + i] : sdg[i]; };
		auto ORD = [&](int i) -> unsigned int { return st_sc ? ordl[sl + i] : ordg[i]; };
		// the reference window any seed of the chain could reach (src/bwamem.c:642-661; computed with the chain) and its bases
		const i64 rmax0 = uni64(C.rmax0), rmax1 = uni64(C.rmax1);
		const int wlen = (int)(rmax1 - rmax0);
		const bool st_w = wlen <= win_cap;
		if (st_w) {
			for (int t = lane; t < wlen; t += 64) winl[t] = (uint8_t)ref_base(pac, l_pac, rmax0 + t);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		}
		auto TB = [&](i64 pos) -> int { return st_w ? (int)winl[pos - rmax0] : ref_base(pac, l_pac, pos); };

		for (int k = n - 1; k >= 0; --k) {
			DevSeed s = SD(uni((int)ORD(k)));
			s.rbeg = uni64(s.rbeg); s.qbeg = uni(s.qbeg); s.len = uni(s.len);
			// ---- is the seed already inside an earlier extension of this read? ----
			bool hit = false;
			for (int i0 = 0; i0 < nav && !hit; i0 += 64) {
				int i = i0 + lane;
				bool h = false;
				if (i < nav) {
					const DevReg p = i < C2A_CAP_R ? avl[i] : av[i];
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
					if (i < n && ORD(i) != SRT_MARK) {
						const DevSeed t = SD((int)ORD(i));
						if (!(t.len < ceil95[s.len])) {
							if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) h = true;
							if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) h = true;
						}
					}
					other = __ballot(h) != 0;
				}
				if (!other) {
					if (lane == 0) { if (st_sc) ordl[sl + k] = SRT_MARK; else ordg[k] = SRT_MARK; }
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
					r = extend(qlen, [&](int j) { return q[s.qbeg - 1 - j]; }, tlen, [&](int t) { return TB(s.rbeg - 1 - t); }, wc, s.len * P.a, P.pen_clip5);
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
					r = extend(qlen, [&](int j) { return q[qe + j]; }, tlen, [&](int t) { return TB(s.rbeg + s.len + t); }, wc, sc0, P.pen_clip3);
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
				const DevSeed t = SD(i);
				if (t.qbeg >= a.qb && t.qbeg + t.len <= a.qe && t.rbeg >= a.rb && t.rbeg + t.len <= a.re) cov += t.len;
			}
			a.seedcov = wave_sum_i32(cov);
			a.w = aw0 > aw1 ? aw0 : aw1;
			if (lane == 0) { av[nav] = a; if (nav < C2A_CAP_R) avl[nav] = a; }
			++nav;
			__builtin_amdgcn_wave_barrier();
		}
		if (unit_mode && lane == 0) U.c_rcnt[ci] = nav - nav0;
	}
	if (lane == 0) {
		if (unit_mode) { if (nav) atomicAdd(&n_regs[rd], nav); }
		else n_regs[rd] = nav;
		// statistics only — but an atomic on ONE address costs ~10 ns of a queue the whole chip shares (MI355X_MICROARCH.md: same-
		// address atomics), and two per read made that queue, not the DP, the length of this kernel (222 000 reads x 2 = 4.9 of
		// 5.4 ms per launch, DP switched off or not): the counters are spread over C2A_STAT_SLOTS cache lines, the host adds them up
		unsigned long long *st = counters + (size_t)(blockIdx.x % C2A_STAT_SLOTS) * 8;
		if (cells) atomicAdd(&st[0], cells);
		if (n_ext) atomicAdd(&st[1], n_ext);
		if (n_closed) atomicAdd(&st[2], n_closed);
		if (n_diff) atomicAdd(&st[3], n_diff);
	}
}

void launch_c2a(void *stream, const C2aParams &P, const ExtParams &ep, int n_reads, const uint8_t *d_seq, const int64_t *d_off,
                const int *d_len, const int *d_chain_beg, const int *d_chain_cnt, const DevChain *d_chains, const DevSeed *d_seeds, unsigned int *d_srt,
                const int *d_reg_beg, DevReg *d_regs, int *d_nregs, const int *d_tab, int tab_stride, const uint8_t *d_pac, unsigned long long *d_counters,
                int max_len, const int *d_order, const C2aUnits *units)
{
	C2aUnits U;
	if (units) U = *units;
	WxParams X;
	for (int i = 0; i < 25; ++i) X.mat[i] = ep.mat[i];
	X.o_del = ep.o_del; X.e_del = ep.e_del; X.o_ins = ep.o_ins; X.e_ins = ep.e_ins; X.zdrop = ep.zdrop;
	size_t shmem = (size_t)C2A_WAVES * c2a_lds_bytes(max_len);
	if (shmem > 64 * 1024 && hipFuncSetAttribute((const void *)c2a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess)
		die("c2a_kernel: cannot reserve %zu bytes of LDS", shmem);
	int n_blocks = (U.max_units + n_reads + C2A_WAVES - 1) / C2A_WAVES;
	hipLaunchKernelGGL(c2a_kernel, dim3(n_blocks), dim3(64 * C2A_WAVES), shmem, (hipStream_t)stream, P, X, n_reads, d_seq, d_off,
	                   d_len, d_chain_beg, d_chain_cnt, d_chains, d_seeds, d_srt, d_reg_beg, d_regs, d_nregs, d_tab, tab_stride, d_pac, d_counters,
	                   max_len, d_order, U);
}

} // namespace mbw
