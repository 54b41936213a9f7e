// c2a_lane.hip — chain -> alignment regions on the device, one LANE per read.
//
// Same contract as c2a_kernel.hip (mem_chain2aln, src/bwamem.c:632-786, with bns_fetch_seq folded into direct reads of
// the 2-bit pac and ksw_extend2, src/ksw.c:380-479, as the DP), different mapping.  The seed loop of a read is strictly
// sequential and its extensions are small (tens of columns), so a wavefront spread over the columns of one row keeps
// most lanes idle and pays a prefix scan per row.  Here every lane owns a read and runs the reference's scalar
// recurrence as it is written; the 64 lanes of a wave advance 64 unrelated extensions row by row:
//   * the DP row state of a lane — eh[j] = {H(i-1,j-1), E(i,j)} plus the query base of column j, 13 + 13 + 3 bits —
//     is one dword per column in LDS, laid out cell[j][lane] (64 consecutive dwords per j: no bank conflicts whatever
//     j each lane is at); the top 3 bits of cell[x] hold base x of the read for the whole life of the read, so that
//     setting up an extension (query reversed for the left one) never goes back to HBM;
//   * the unit of lock-step is a handful of DP cells, not a row: every lane carries its own (row, column) position, and
//     row boundaries, band retries, seed / chain / read changes are per-lane state transitions between cell steps, so a
//     lane with 10-column rows never waits for a neighbour with 130-column rows (measured: lock-stepping whole rows left
//     ~10 % of the lanes busy);
//   * lanes fetch their next read from an atomic counter (persistent grid), which evens out reads with many
//     extensions against reads with none.
// Integer work only; the floating-point decisions of the reference come in as tables indexed by length (see the host).
// HBM traffic: the read (once), its chains and seeds, one pac byte per DP row, the regions.  VALU-bound: ~30 ops per cell.
#include <hip/hip_runtime.h>
#include "device.h"

namespace mbw {

#define HIP_OK(call)                                                                                             \
	do {                                                                                                         \
		hipError_t e_ = (call);                                                                                  \
		if (e_ != hipSuccess) die("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);    \
	} while (0)

namespace {

typedef long long i64;

struct LxParams {
	uint32_t slo[5];   // scores of target base t against query codes 0..3, one byte each
	int s4[5];         // ... against query code 4
	int o_del, e_del, o_ins, e_ins, zdrop;
};

__device__ __forceinline__ int lx_ref_base(const uint8_t *__restrict__ pac, i64 l_pac, i64 p)
{
	if (p >= l_pac) {
		i64 f = (l_pac << 1) - 1 - p;
		return 3 - ((pac[f >> 2] >> ((~f & 3) << 1)) & 3);
	}
	return (pac[p >> 2] >> ((~p & 3) << 1)) & 3;
}

enum { ST_CHAIN = 0, ST_SEED, ST_RIGHT, ST_EXT_DONE, ST_FIN, ST_FETCH = 99, ST_ROW = 100, ST_DONE = 101 };
#define LX_MARK 0xFFFFFFFFu
#define LX_TOP 0xE0000000u    // base of the read at this position
#define LX_KEEP 0xFC000000u   // ... and the query base of this column

__global__ void __launch_bounds__(64)
c2a_lane_kernel(C2aParams P, LxParams X, int n_reads, const uint8_t *__restrict__ seq, const int64_t *__restrict__ off,
                const int *__restrict__ lens, const int *__restrict__ chain_beg, const int *__restrict__ chain_cnt, const DevChain *__restrict__ chains,
                const DevSeed *__restrict__ seeds, unsigned int *srt, const int *__restrict__ reg_beg, DevReg *regs, int *n_regs,
                const int *__restrict__ tab, int tab_stride, const uint8_t *__restrict__ pac, unsigned long long *counters)
{
	extern __shared__ uint32_t cell[];   // [max_len + 5][64]
	const int lane = threadIdx.x;
	const int *gap = tab, *bound5 = tab + tab_stride, *bound3 = tab + 2 * tab_stride, *ceil95 = tab + 3 * tab_stride,
	          *thr10 = tab + 4 * tab_stride;
	const i64 l_pac = P.l_pac;
	const int oe_del = X.o_del + X.e_del, oe_ins = X.o_ins + X.e_ins, e_del = X.e_del, e_ins = X.e_ins;
	unsigned long long cells = 0, n_ext = 0;

	int st = ST_FETCH;   // every lane starts by asking for a read
	// read
	int rd = 0, lq = 0, nav = 0, ci = 0, ci_end = 0;
	DevReg *av = regs;
	// chain
	int n = 0, k = -1, c_rid = 0;
	float c_frac = 0;
	const DevSeed *sd = seeds;
	unsigned int *ord = srt;
	i64 rmax0 = 0, rmax1 = 0;
	// seed / region under construction
	DevSeed s;
	s.rbeg = 0; s.qbeg = 0; s.len = 0;
	DevReg a;
	a.rb = a.re = 0; a.qb = a.qe = 0; a.rid = 0; a.score = a.truesc = -1; a.w = 0; a.seedcov = 0; a.seedlen0 = 0; a.frac_rep = 0; a.pad = 0;
	int aw0 = 0, aw1 = 0;
	// extension
	int dir = 0, tri = 0, qlen = 0, tlen = 0, w = 0, h0 = 0, prev = 0, sc0 = 0, qe0 = 0;
	i64 re0 = 0, tpos = 0;   // tpos: doubled-coordinate position of target row 0; rows go tpos + i * tstep
	int tstep = 1;
	int i = 0, beg = 0, end = 0, best = 0, best_i = -1, best_j = -1, best_ie = -1, gscore = -1, max_off = 0, tb_next = 0;
	// the row a lane is in (lanes do not wait for each other at row boundaries)
	bool row_open = false;
	int j = 0, jend = 0, h1 = 0, f = 0, key = -1, first_nz = 0x7fffffff, last_nz = -1, s4 = 0;
	uint32_t slo = 0;

	// set up one call of ksw_extend2: first row into LDS (src/ksw.c:389-393), counters reset
	auto ext_begin = [&]() {
		if (dir == 0) {
			prev = a.score;
			qlen = s.qbeg; tlen = (int)(s.rbeg - rmax0);
			aw0 = P.w << tri;
			w = aw0 < bound5[qlen] ? aw0 : bound5[qlen];
			h0 = s.len * P.a;
			tpos = s.rbeg - 1; tstep = -1;
		} else {
			prev = a.score;
			if (tri == 0) { sc0 = a.score; qe0 = s.qbeg + s.len; re0 = s.rbeg + s.len - rmax0; }
			qlen = lq - qe0; tlen = (int)(rmax1 - rmax0 - re0);
			aw1 = P.w << tri;
			w = aw1 < bound3[qlen] ? aw1 : bound3[qlen];
			h0 = sc0;
			tpos = s.rbeg + s.len; tstep = 1;
		}
		int hrow = h0;   // eh[j].h of the first row
		for (int j = 0; j <= qlen; ++j) {
			uint32_t qb = 0;
			if (j < qlen) {
				const int x = dir == 0 ? s.qbeg - 1 - j : qe0 + j;
				qb = cell[x * 64 + lane] >> 29;
			}
			if (j == 1) hrow = h0 > oe_ins ? h0 - oe_ins : 0;
			else if (j >= 2) hrow = hrow > e_ins ? hrow - e_ins : 0;
			cell[j * 64 + lane] = (cell[j * 64 + lane] & LX_TOP) | qb << 26 | (uint32_t)hrow;
		}
		i = 0; beg = 0; end = qlen;
		best = h0; best_i = best_j = -1; best_ie = -1; gscore = -1; max_off = 0;
		++n_ext;
		row_open = false;
		if (tlen > 0) { tb_next = lx_ref_base(pac, l_pac, tpos); st = ST_ROW; }
		else st = ST_EXT_DONE;
	};

	for (;;) {
		// ---------------- DP: lanes whose row is complete close it (src/ksw.c:447-469) ----------------
		if (st == ST_ROW && row_open && j >= jend) {
			row_open = false;
			cell[end * 64 + lane] = (cell[end * 64 + lane] & LX_KEEP) | (uint32_t)h1;   // eh[end] = {h1, 0}
			const int jfin = beg < end ? end : beg;   // the reference's column counter after its loop
			if (jfin == qlen) {
				if (h1 >= gscore) best_ie = i;
				if (h1 > gscore) gscore = h1;
			}
			const int rowmax = key < 0 ? 0 : key >> 13, rowmax_j = key < 0 ? -1 : key & 8191;
			bool stop = rowmax == 0;
			if (!stop) {
				if (rowmax > best) {
					best = rowmax; best_i = i; best_j = rowmax_j;
					int d = rowmax_j - i; d = d < 0 ? -d : d;
					if (d > max_off) max_off = d;
				} else if (X.zdrop > 0) {
					const int di = i - best_i, dj = rowmax_j - best_j;
					if (di > dj) { if (best - rowmax - (di - dj) * e_del > X.zdrop) stop = true; }
					else { if (best - rowmax - (dj - di) * e_ins > X.zdrop) stop = true; }
				}
			}
			if (!stop) {
				// live range of the next row (src/ksw.c:466-469)
				const int nb = first_nz < end ? first_nz : end;
				int ne = h1 != 0 ? end : last_nz;
				if (ne < nb) ne = nb - 1;
				beg = nb;
				end = ne + 2 < qlen ? ne + 2 : qlen;
				if (++i >= tlen) stop = true;
			}
			if (stop) st = ST_EXT_DONE;
		}
		// ---------------- new reads: the whole wave loads the read of every lane that asks for one ----------------
		if (__any(st == ST_FETCH)) {
			if (st == ST_FETCH) {
				rd = (int)atomicAdd(&counters[2], 1ull);
				if (rd >= n_reads) st = ST_DONE;
			}
			unsigned long long need = __ballot(st == ST_FETCH);
			while (need) {
				const int l = __ffsll((long long)need) - 1;
				need &= need - 1;
				const int r = __shfl(rd, l);
				const int len = lens[r];
				const uint8_t *q = seq + off[r];   // 16-byte aligned slot, padded to a multiple of 16
				for (int base = 0; base < len; base += 256) {
					const int x0 = base + lane * 4;
					if (x0 < len) {
						const uint32_t v = *(const uint32_t *)(q + x0);   // coalesced: 64 lanes x 4 bases
#pragma unroll
						for (int u = 0; u < 4; ++u) {
							uint32_t c = (v >> (8 * u)) & 0xff;
							if (x0 + u < len) cell[(x0 + u) * 64 + l] = (c > 4 ? 4u : c) << 29;
						}
					}
				}
				if (lane == 0) cell[len * 64 + l] = 0;
			}
			if (st == ST_FETCH) {
				lq = lens[rd];
				av = regs + reg_beg[rd];
				nav = 0;
				ci = chain_beg[rd]; ci_end = ci + (chain_cnt[rd] > 0 ? chain_cnt[rd] : 0);
				st = ST_CHAIN;
			}
		}
		// ---------------- control: every lane runs until it has a DP row to do, needs a new read, or is done ----------------
		while (st < ST_FETCH) {
			if (st == ST_CHAIN) {
				if (ci >= ci_end) { n_regs[rd] = nav; st = ST_FETCH; break; }
				const DevChain C = chains[ci++];
				n = C.n_seeds;
				if (n == 0) continue;
				sd = seeds + C.seed_beg;
				ord = srt + C.seed_beg;
				c_rid = C.rid; c_frac = C.frac_rep;
				rmax0 = C.rmax0; rmax1 = C.rmax1;   // the window of src/bwamem.c:642-661, worked out while the host packed the seeds
				k = n - 1;
				st = ST_SEED;
			} else if (st == ST_SEED) {
				if (k < 0) { st = ST_CHAIN; continue; }
				s = sd[k];   // seeds come sorted in visiting order; ord[] only carries the "skipped" marks
				// is the seed already inside an earlier extension of this read? (src/bwamem.c:671-689)
				bool hit = false;
				for (int r_ = 0; r_ < nav && !hit; ++r_) {
					const DevReg p = av[r_];
					if (s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) continue;
					if (s.len - p.seedlen0 > thr10[lq]) continue;
					int qd = s.qbeg - p.qb; i64 rdd = s.rbeg - p.rb;
					int mg = gap[qd < rdd ? qd : (int)rdd];
					int ww = mg < p.w ? mg : p.w;
					if (qd - rdd < ww && rdd - qd < ww) { hit = true; break; }
					qd = p.qe - (s.qbeg + s.len); rdd = p.re - (s.rbeg + s.len);
					mg = gap[qd < rdd ? qd : (int)rdd];
					ww = mg < p.w ? mg : p.w;
					if (qd - rdd < ww && rdd - qd < ww) hit = true;
				}
				if (hit) {   // extend anyway if a long overlapping seed of the chain sits on another diagonal (src/bwamem.c:690-706)
					bool other = false;
					for (int t_ = k + 1; t_ < n && !other; ++t_) {
						if (ord[t_] == LX_MARK) continue;
						const DevSeed t = sd[t_];
						if (t.len < ceil95[s.len]) continue;
						if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) other = true;
						if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) other = true;
					}
					if (!other) { ord[k] = LX_MARK; --k; continue; }
				}
				// new region
				a.rb = a.re = 0; a.qb = a.qe = 0;
				a.rid = c_rid; a.score = a.truesc = -1; a.w = P.w; a.seedcov = 0; a.seedlen0 = s.len; a.frac_rep = c_frac; a.pad = 0;
				aw0 = aw1 = P.w;
				if (s.qbeg) { dir = 0; tri = 0; ext_begin(); }   // left extension: both sequences reversed
				else { a.score = a.truesc = s.len * P.a; a.qb = 0; a.rb = s.rbeg; st = ST_RIGHT; }
			} else if (st == ST_RIGHT) {
				if (s.qbeg + s.len != lq) { dir = 1; tri = 0; ext_begin(); }
				else { a.qe = lq; a.re = s.rbeg + s.len; st = ST_FIN; }
			} else if (st == ST_EXT_DONE) {
				const int r_score = best, r_qle = best_j + 1, r_tle = best_i + 1, r_gtle = best_ie + 1, r_gscore = gscore;
				a.score = r_score;
				const int aw = dir == 0 ? aw0 : aw1;
				if (tri == 0 && !(a.score == prev || max_off < (aw >> 1) + (aw >> 2))) { tri = 1; ext_begin(); continue; }   // retry, band doubled
				if (dir == 0) {
					if (r_gscore <= 0 || r_gscore <= a.score - P.pen_clip5) {   // local
						a.qb = s.qbeg - r_qle; a.rb = s.rbeg - r_tle; a.truesc = a.score;
					} else {                                                    // reaches the read start
						a.qb = 0; a.rb = s.rbeg - r_gtle; a.truesc = r_gscore;
					}
					st = ST_RIGHT;
				} else {
					if (r_gscore <= 0 || r_gscore <= a.score - P.pen_clip3) {
						a.qe = qe0 + r_qle; a.re = rmax0 + re0 + r_tle; a.truesc += a.score - sc0;
					} else {
						a.qe = lq; a.re = rmax0 + re0 + r_gtle; a.truesc += r_gscore - sc0;
					}
					st = ST_FIN;
				}
			} else {   // ST_FIN: seed coverage, store the region
				int cov = 0;
				for (int t_ = 0; t_ < n; ++t_) {
					const DevSeed t = sd[t_];
					if (t.qbeg >= a.qb && t.qbeg + t.len <= a.qe && t.rbeg >= a.rb && t.rbeg + t.len <= a.re) cov += t.len;
				}
				a.seedcov = cov;
				a.w = aw0 > aw1 ? aw0 : aw1;
				av[nav++] = a;
				--k;
				st = ST_SEED;
			}
		}
		if (!__any(st == ST_ROW || st == ST_FETCH)) break;

		// ---------------- DP: lanes that are at a row boundary open their next row ... ----------------
		if (st == ST_ROW && !row_open) {
			const int tb = tb_next;
			if (i + 1 < tlen) tb_next = lx_ref_base(pac, l_pac, tpos + (i64)(i + 1) * tstep);   // in flight during this row
			slo = tb == 0 ? X.slo[0] : tb == 1 ? X.slo[1] : tb == 2 ? X.slo[2] : X.slo[3];
			s4 = tb == 0 ? X.s4[0] : tb == 1 ? X.s4[1] : tb == 2 ? X.s4[2] : X.s4[3];
			if (beg < i - w) beg = i - w;
			if (end > i + w + 1) end = i + w + 1;
			if (end > qlen) end = qlen;
			h1 = 0;
			if (beg == 0) { h1 = h0 - (X.o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
			f = 0; key = -1; first_nz = 0x7fffffff; last_nz = -1;
			j = beg; jend = end;
			if (beg < end) cells += (unsigned long long)(end - beg);
			row_open = true;
		}
		// ---------------- ... and every lane inside a row advances by up to 8 cells of ksw_extend2's inner loop ----------------
		const int jlim = st == ST_ROW ? jend : 0;
#pragma unroll
		for (int g = 0; g < 2; ++g) {
			uint32_t wv[4];
			const int jb = st == ST_ROW ? j : 0;
#pragma unroll
			for (int u = 0; u < 4; ++u) wv[u] = cell[(jb + u) * 64 + lane];   // reads past the row's end stay inside the lane's column (padding)
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				if (j < jlim) {
					const int m0 = wv[u] & 0x1fff;
					int e = (wv[u] >> 13) & 0x1fff;
					const uint32_t qb = (wv[u] >> 26) & 7;
					int sc = (int)(int8_t)(slo >> ((qb & 3) << 3));
					sc = qb < 4 ? sc : s4;
					const int M = m0 ? m0 + sc : 0;   // M and H kept apart: no "100M3I3D20M"
					int h = max(max(M, e), f);
					const uint32_t back = (wv[u] & LX_KEEP) | (uint32_t)h1;   // eh[j].h = H(i,j-1)
					key = max(key, h << 13 | j);      // row maximum, largest column wins ties
					int t = M - oe_del; t = t > 0 ? t : 0;
					e = max(e - e_del, t);
					t = M - oe_ins; t = t > 0 ? t : 0;
					f = max(f - e_ins, t);
					if ((h1 | e) != 0) { first_nz = min(first_nz, j); last_nz = j; }
					cell[j * 64 + lane] = back | (uint32_t)e << 13;
					h1 = h;
					++j;
				}
			}
		}
	}
	// totals
	for (int o = 32; o; o >>= 1) {
		cells += __shfl_xor(cells, o);
		n_ext += __shfl_xor(n_ext, o);
	}
	if (lane == 0) {
		atomicAdd(&counters[0], cells);
		atomicAdd(&counters[1], n_ext);
	}
}

} // namespace

bool c2a_lane_fits(int max_len, int a)
{
	return (int64_t)max_len * a + a * 64 < 8000 && (size_t)(max_len + 5) * 256 <= 160 * 1024;
}

void launch_c2a_lane(void *stream, const C2aParams &P, const ExtParams &ep, int n_reads, const uint8_t *d_seq, const int64_t *d_off,
                     const int *d_len, const int *d_chain_beg, const int *d_chain_cnt, const DevChain *d_chains, const DevSeed *d_seeds,
                     unsigned int *d_srt, const int *d_reg_beg, DevReg *d_regs, int *d_nregs, const int *d_tab, int tab_stride, const uint8_t *d_pac,
                     unsigned long long *d_counters, int max_len)
{
	if (n_reads <= 0) return;
	LxParams X;
	for (int t = 0; t < 5; ++t) {
		X.slo[t] = 0;
		for (int q = 0; q < 4; ++q) X.slo[t] |= (uint32_t)(uint8_t)ep.mat[t * 5 + q] << (8 * q);
		X.s4[t] = ep.mat[t * 5 + 4];
	}
	X.o_del = ep.o_del; X.e_del = ep.e_del; X.o_ins = ep.o_ins; X.e_ins = ep.e_ins; X.zdrop = ep.zdrop;
	const size_t lds = (size_t)(max_len + 5) * 256;
	static size_t s_attr = 0;
	static int s_cus = 0;
	if (!s_cus) {
		int dev = 0;
		HIP_OK(hipGetDevice(&dev));
		HIP_OK(hipDeviceGetAttribute(&s_cus, hipDeviceAttributeMultiprocessorCount, dev));
	}
	if (lds > s_attr) {
		HIP_OK(hipFuncSetAttribute((const void *)c2a_lane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		s_attr = lds;
	}
	const int per_cu = (int)std::max<size_t>(1, (160 * 1024) / lds);
	int blocks = s_cus * per_cu;                       // persistent: as many waves as fit at once
	blocks = std::min(blocks, (n_reads + 63) / 64);
	hipLaunchKernelGGL(c2a_lane_kernel, dim3(blocks), dim3(64), lds, (hipStream_t)stream, P, X, n_reads, d_seq, d_off, d_len, d_chain_beg, d_chain_cnt,
	                   d_chains, d_seeds, d_srt, d_reg_beg, d_regs, d_nregs, d_tab, tab_stride, d_pac, d_counters);
	HIP_OK(hipGetLastError());
}

} // namespace mbw
