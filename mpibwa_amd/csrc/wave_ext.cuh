// wave_ext.cuh — banded affine-gap seed extension by ONE wavefront (device code).
//
// Bit-exact counterpart of ksw_extend2 (src/ksw.c:380-479).  The reference's
// decisions are row-sequential (the live column range of row i+1 depends on
// the zeros of row i, z-drop and the best cell are tested per row), so rows
// stay sequential here and the 64 lanes work across the columns of a row:
//   * H(i-1,j-1)/E(i,j) live in an LDS array exactly like the reference's
//     eh[] (including the stale cells it re-reads when the range re-grows);
//   * the F recurrence  F(i,j+1) = max(F(i,j) - e_ins, max(M(i,j)-oe_ins, 0))
//     only depends on M, so it is a max-plus prefix scan: with
//     g_k = t_k + k*e_ins,  F(i,j) = max(A, max_{k<j} g_k + e_ins) - j*e_ins,
//     computed with DPP row shifts / row broadcasts (no LDS traffic);
//   * the row maximum with the reference's tie rule (largest j wins) is a
//     wave max followed by a ballot.
// No MFMA: integer DP with a data-dependent band is not a contraction.
#ifndef MBW_WAVE_EXT_CUH
#define MBW_WAVE_EXT_CUH
#include <hip/hip_runtime.h>

namespace mbw {

#define WX_NEG (-(1 << 29))

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int wx_dpp(int old, int v)
{
	return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}

// inclusive prefix max over the 64 lanes: six v_max_i32 with a DPP source operand.  A lane that receives nothing in a
// step (shifted in from outside its row, or masked by row_mask) is simply not written, so it keeps its own value — no
// identity constant, no separate v_mov_dpp.  (The builtin form costs three VALU ops per step: the compiler does not
// fold it.)  s_nop 1: a DPP operand must not be read within two wait states of the VALU write that produced it.
__device__ __forceinline__ int wx_scan_max(int v)
{
	asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
	             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
	             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
	             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
	             "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
	             "s_nop 1"
	             : "+v"(v));
	return v;
}
// value of the previous lane (lane 0 gets `first`)
__device__ __forceinline__ int wx_prev_lane(int v, int first)
{
	return wx_dpp<0x138, 0xf>(first, v);         // wave_shr:1
}

struct WxParams {
	int8_t mat[25];
	int o_del, e_del, o_ins, e_ins, zdrop;
};

struct WxResult { int score, qle, tle, gtle, gscore, max_off; };

// QF(j)  -> query base 0..4 at column j  (evaluated per lane)
// TF(i)  -> target base 0..4 at row i    (evaluated per lane for prefetching 64 rows at a time)
// H, E   -> LDS (or global) arrays of qlen+1 ints owned by this wave
// `w` must already be clamped like src/ksw.c:395-407 (see wx_clamp_band on the host side).
//
// EARLY: stop as soon as no later row can change any output.  Every alignment path that reaches a later row leaves row i
// through one of the stored cells — diagonally from eh[j].h into column j, or vertically from eh[j].e in column j — and
// gains at most max_sc per remaining column, so
//     B = max_j max(eh[j].h + (qlen - j) * max_sc,  eh[j].e + (qlen - 1 - j) * max_sc)        (stale cells included)
// bounds every later H, in particular every later row maximum and every later score at the query end.  When B <= max
// and B < gscore the reference's remaining rows update neither (max, max_i, max_j, max_off) nor (gscore, max_ie); it only
// goes on until the scores have decayed to zero, typically for as many rows again as the alignment itself.  The test
// runs every 4th row once the query end has been reached.  Off: the row count (and the cell counter) match the reference.
template <bool EARLY = false, typename QF, typename TF>
__device__ __forceinline__ WxResult wave_extend(int qlen, QF qf, int tlen, TF tf, const WxParams &P, int w, int h0,
                                                int *H, int *E, unsigned long long &cells, int max_sc = 1)
{
	const int lane = threadIdx.x & 63;
	// The extension's shape is the same in every lane, but it was read through vector loads: pin it to scalar registers,
	// or the row loop's whole bookkeeping (band limits, live range, strip counts, loop tests) is compiled into vector
	// instructions with exec-mask branches around them.
	qlen = __builtin_amdgcn_readfirstlane(qlen);
	tlen = __builtin_amdgcn_readfirstlane(tlen);
	w = __builtin_amdgcn_readfirstlane(w);
	h0 = __builtin_amdgcn_readfirstlane(h0);
	max_sc = __builtin_amdgcn_readfirstlane(max_sc);
	const int oe_del = P.o_del + P.e_del, oe_ins = P.o_ins + P.e_ins, e_del = P.e_del, e_ins = P.e_ins;
	// first row (src/ksw.c:389-393)
	for (int j = lane; j <= qlen; j += 64) {
		int v = 0;
		if (j == 0) v = h0;
		else {
			int t = h0 - oe_ins - (j - 1) * e_ins;   // H[j] while the chain H[j-1] > e_ins holds
			// H[1] = max(h0-oe_ins,0); H[j] = H[j-1]-e_ins as long as H[j-1] > e_ins
			int h1 = h0 > oe_ins ? h0 - oe_ins : 0;
			if (j == 1) v = h1;
			else v = (h1 - (j - 2) * e_ins > e_ins) ? t : 0;
		}
		H[j] = v; E[j] = 0;
	}
	__builtin_amdgcn_wave_barrier();
	int best = h0, best_i = -1, best_j = -1, best_ie = -1, gscore = -1, max_off = 0;
	int beg = 0, end = qlen;
	int tv = 0, tv_base = -64;
	// the five rows of the scoring matrix as packed bytes in scalar registers: byte q of {hi, lo} = mat[t][q], so the
	// score of a cell is one v_perm_b32 (+ sign extension) instead of a chain of four compares and selects
	uint32_t plo[5], phi[5];
#pragma unroll
	for (int t = 0; t < 5; ++t) {
		plo[t] = (uint32_t)(uint8_t)P.mat[t * 5] | (uint32_t)(uint8_t)P.mat[t * 5 + 1] << 8 | (uint32_t)(uint8_t)P.mat[t * 5 + 2] << 16 |
		         (uint32_t)(uint8_t)P.mat[t * 5 + 3] << 24;
		phi[t] = (uint32_t)(uint8_t)P.mat[t * 5 + 4];
	}
	for (int i = 0; i < tlen; ++i) {
		if (i - tv_base >= 64) { tv_base = i; tv = (i + lane < tlen) ? (int)tf(i + lane) : 4; }
		const int tb = __builtin_amdgcn_readlane(tv, i - tv_base);
		const uint32_t slo = tb == 0 ? plo[0] : tb == 1 ? plo[1] : tb == 2 ? plo[2] : tb == 3 ? plo[3] : plo[4];
		const uint32_t shi = tb == 0 ? phi[0] : tb == 1 ? phi[1] : tb == 2 ? phi[2] : tb == 3 ? phi[3] : phi[4];
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		int hleft0 = 0;
		if (beg == 0) { hleft0 = h0 - (P.o_del + e_del * (i + 1)); if (hleft0 < 0) hleft0 = 0; }
		int rowkey = -1;                           // (row maximum) << 13 | its largest column
		int lanekey = -1;                          // ... per lane over the strips of the row; one wave reduction per row
		int A = beg * e_ins;                       // F(i,beg) = 0
		int h_carry = 0, h_last = hleft0;          // h of column 64*s-1; H[end] after the row
		int first_nz = end, last_nz = -1;          // first / last cell of [beg,end] with H or E non-zero after the row
		if (beg < end) {
			const int s0 = beg >> 6, s1 = (end - 1) >> 6;
			int diag0 = H[s0 << 6];                // pre-read of the strip's first cell (see below)
			for (int s = s0; s <= s1; ++s) {
				const int j = (s << 6) + lane;
				const bool act = j >= beg && j < end;
				int diag = lane == 0 ? diag0 : (j <= qlen ? H[j] : 0);
				int e = j <= qlen ? E[j] : 0;
				// the last lane writes H[64(s+1)], which is the next strip's first diagonal: read it first
				const int nxt = (s + 1) << 6;
				if (nxt <= qlen) diag0 = H[nxt];
				int qb = act ? (int)qf(j) : 4;
				const int sc = (int)(int8_t)__builtin_amdgcn_perm(shi, slo, (uint32_t)qb | 0x0c0c0c00u);
				int M = diag ? diag + sc : 0;
				int tI = M - oe_ins; tI = tI > 0 ? tI : 0;
				int g = act ? tI + j * e_ins : WX_NEG;
				int incl = wx_scan_max(g);
				int excl = wx_prev_lane(incl, WX_NEG);
				int f = max(A, excl + e_ins) - j * e_ins;
				int h = max(max(M, e), f);
				int tD = M - oe_del; tD = tD > 0 ? tD : 0;
				int en = max(e - e_del, tD);
				if (j == beg) H[j] = hleft0;
				if (act) { H[j + 1] = h; E[j] = en; }
				A = max(A, __builtin_amdgcn_readlane(incl, 63) + e_ins);
				// row maximum; the largest column wins ties: maximum of (h << 13 | column)
				lanekey = max(lanekey, act ? (h << 13 | j) : -1);
				// cells of this strip that are non-zero after the row: H[c] = h of column c-1, E[c] = en of column c
				const unsigned long long bh = __ballot(act && h != 0), be = __ballot(act && en != 0);
				unsigned long long nz = (bh << 1) | be;
				if (h_carry != 0 && (s << 6) > beg) nz |= 1ull;
				if ((beg >> 6) == s && hleft0 != 0) nz |= 1ull << (beg & 63);
				if (nz) {
					const int lo = (s << 6) + __ffsll((long long)nz) - 1, hi = (s << 6) + 63 - __clzll(nz);
					if (lo < end && lo < first_nz) first_nz = lo;
					if (hi > last_nz) last_nz = hi;
				}
				h_carry = __builtin_amdgcn_readlane(h, 63);
				if (s == s1) h_last = __builtin_amdgcn_readlane(h, (end - 1) & 63);
			}
			rowkey = __builtin_amdgcn_readlane(wx_scan_max(lanekey), 63);
			if ((end & 63) == 0 && h_carry != 0 && end > last_nz) last_nz = end;   // cell `end` opens the next strip
			cells += (unsigned long long)(end - beg);
		} else {
			if (lane == 0) H[end] = hleft0;        // empty range: eh[end].h = h1 (src/ksw.c:447)
			if (hleft0 != 0) last_nz = end;
		}
		if (lane == 0) E[end] = 0;
		__builtin_amdgcn_wave_barrier();
		const int h1 = h_last;
		const int jfin = beg < end ? end : beg;    // value of the reference's column counter after its loop
		if (jfin == qlen) {
			if (h1 >= gscore) best_ie = i;
			if (h1 > gscore) gscore = h1;
		}
		const int rowmax = rowkey < 0 ? 0 : rowkey >> 13, rowmax_j = rowkey < 0 ? -1 : rowkey & 8191;
		if (rowmax == 0) break;
		if (rowmax > best) {
			best = rowmax; best_i = i; best_j = rowmax_j;
			int d = rowmax_j - i; d = d < 0 ? -d : d;
			if (d > max_off) max_off = d;
		} else if (P.zdrop > 0) {
			int di = i - best_i, dj = rowmax_j - best_j;
			if (di > dj) { if (best - rowmax - (di - dj) * e_del > P.zdrop) break; }
			else { if (best - rowmax - (dj - di) * e_ins > P.zdrop) break; }
		}
		// live range of the next row (src/ksw.c:466-469): first non-zero cell of [beg,end) (else end), then the last
		// non-zero cell of [that,end] (else one before it)
		const int nb = first_nz;
		const int ne = last_nz >= nb ? last_nz : nb - 1;
		beg = nb;
		end = ne + 2 < qlen ? ne + 2 : qlen;
		if (EARLY && gscore >= 0 && (i & 3) == 3) {
			int b = WX_NEG;
			for (int j = lane; j <= qlen; j += 64) {
				const int hv = j < qlen ? H[j] + (qlen - j) * max_sc : WX_NEG;
				const int ev = E[j] + (qlen - 1 - j) * max_sc;
				b = max(b, max(hv, ev));
			}
			b = __builtin_amdgcn_readlane(wx_scan_max(b), 63);
			if (b < gscore && b <= best) break;
		}
	}
	WxResult r;
	r.score = best; r.qle = best_j + 1; r.tle = best_i + 1; r.gtle = best_ie + 1; r.gscore = gscore; r.max_off = max_off;
	return r;
}

} // namespace mbw
#endif
