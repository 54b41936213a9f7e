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
// minimum of two wave-uniform values on the scalar unit
__device__ __forceinline__ int wx_smin(int a, int b)
{
	int r;
	asm("s_min_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc");
	return r;
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

// LDS of one wave: the row state eh[] as {h, e} pairs like the reference's eh_t — one 8-byte load and one 8-byte store per lane
// and strip (qlen + 2 cells, and a strip of pad so that the lanes of a row read and write a full strip without bounds tests),
// the query bases (staged by the caller: one byte per column, + pad), and one write-only cell.
struct WxLds { int2 *HE; uint8_t *Qs; int2 *dummy; };
__host__ __device__ inline int wx_lds_ints(int max_qlen) { return 2 * (max_qlen + 2 + 64) + (((max_qlen + 64 + 7) & ~7) >> 2) + 2; }   // (even: waves stay 8-byte aligned)
__device__ __forceinline__ WxLds wx_lds(int *base, int max_qlen)
{
	WxLds L;
	L.HE = (int2 *)base;
	L.Qs = (uint8_t *)(L.HE + (max_qlen + 2 + 64));
	L.dummy = (int2 *)(L.Qs + ((max_qlen + 64 + 7) & ~7));
	return L;
}

// L.Qs[j] -> query base 0..4 at column j (staged by the caller)
// TF(i)   -> target base 0..4 at row i    (evaluated per lane, 64 rows at a time)
// `w` must already be clamped like src/ksw.c:395-407 (see wx_clamp_band on the host side).
//
// A row is worked in strips of 64 lanes counted from its first live column `beg`.  Lane L of a strip owns column
// j = beg + L: it reads eh[j] = {H(i-1,j-1), E(i,j)}, and writes eh[j].h = h of column j - 1 (lane 0 of the row: the first-column
// value h1 of src/ksw.c:420-424; the lane one past the last column: eh[end].h, src/ksw.c:447) and eh[j].e = its own E(i+1,j)
// (one past the last column: 0) — so a strip reads and writes the same 64 cells, strips never touch each other's, and no
// lane needs a bounds test: what lies past the row goes to the write-only cell.  The scoring-matrix rows of the next 64
// target bases sit one per lane, a row takes its own with two v_readlane.  (The row loop was bound by the scalar unit:
// 250 scalar instructions per row against 100 vector ones; this form has 50.)
//
// EARLY: stop as soon as no later row can change any output the caller uses.  Every alignment path that reaches a later row
// leaves row i through one of the stored cells — diagonally from eh[j].h into column j, or vertically from eh[j].e in column j
// — and gains at most max_sc per remaining column, so
//     B = max_j max(eh[j].h + (qlen - j) * max_sc,  eh[j].e + (qlen - 1 - j) * max_sc)        (stale cells included)
// bounds every later H, in particular every later row maximum and every later score at the query end.
//   (1) B <= max and B < gscore: the reference's remaining rows update neither (max, max_i, max_j, max_off) nor
//       (gscore, max_ie); it only goes on until the scores have decayed to zero, typically for as many rows again as the
//       alignment itself.
//   (2) B <= max - clip and gscore <= max - clip (clip = the caller's clipping penalty, src/bwamem.c:722,755): the best cell is
//       final, and whatever the score at the query end becomes it stays at or below max - clip, so the caller takes its
//       local-alignment branch and reads score, qle, tle and max_off only — gscore and gtle may differ from the reference's,
//       nobody looks.  This is what ends the extensions into sequence that does not match (a chimeric or clipped read, a
//       seed in the wrong copy of a repeat): the reference computes rows until a score of ~100 has decayed to zero.
// The test runs every 4th row once it can succeed.  Off (the default): row count, cell count and all six outputs are the reference's.
template <bool EARLY = false, typename TF>
__device__ __forceinline__ WxResult wave_extend(int qlen, int tlen, TF tf, const WxParams &P, int w, int h0, const WxLds &L,
                                                unsigned long long &cells, int max_sc = 1, int clip = 0x3fffffff)
{
	const int lane = threadIdx.x & 63;
	int2 *const HE = L.HE;
	const uint8_t *const Qs = L.Qs;
	// The extension's shape is the same in every lane, but it was read through vector loads: pin it to scalar registers,
	// or the row loop's whole bookkeeping (band limits, live range, strip counts, loop tests) is compiled into vector
	// instructions with exec-mask branches around them.
	qlen = __builtin_amdgcn_readfirstlane(qlen);
	tlen = __builtin_amdgcn_readfirstlane(tlen);
	w = __builtin_amdgcn_readfirstlane(w);
	h0 = __builtin_amdgcn_readfirstlane(h0);
	max_sc = __builtin_amdgcn_readfirstlane(max_sc);
	clip = __builtin_amdgcn_readfirstlane(clip);
	const int oe_del = P.o_del + P.e_del, oe_ins = P.o_ins + P.e_ins, e_del = P.e_del, e_ins = P.e_ins;
	int lane_e = lane * e_ins;
	asm volatile("" : "+v"(lane_e));   // (opaque: or the compiler folds it back into a multiplication per row, two in fact: +j e and -j e)
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
		HE[j] = make_int2(v, 0);
	}
	__builtin_amdgcn_wave_barrier();
	int best = h0, best_i = -1, best_j = -1, best_ie = -1, gscore = -1, max_off = 0;
	int beg = 0, end = qlen;
	int tv_base = -64;
	// the five rows of the scoring matrix as packed bytes: byte q of {hi, lo} = mat[t][q], so the score of a cell is one
	// v_perm_b32 (+ sign extension) instead of a chain of four compares and selects
	int vslo = 0, vshi = 0;
	uint32_t plo[5], phi[5];
#pragma unroll
	for (int t = 0; t < 5; ++t) {
		plo[t] = (uint32_t)(uint8_t)P.mat[t * 5] | (uint32_t)(uint8_t)P.mat[t * 5 + 1] << 8 | (uint32_t)(uint8_t)P.mat[t * 5 + 2] << 16 |
		         (uint32_t)(uint8_t)P.mat[t * 5 + 3] << 24;
		phi[t] = (uint32_t)(uint8_t)P.mat[t * 5 + 4];
	}
	for (int i = 0; i < tlen; ++i) {
		if (i - tv_base >= 64) {
			tv_base = i;
			const int tb = (i + lane < tlen) ? (int)tf(i + lane) : 4;
			vslo = (int)(tb == 0 ? plo[0] : tb == 1 ? plo[1] : tb == 2 ? plo[2] : tb == 3 ? plo[3] : plo[4]);
			vshi = (int)(tb == 0 ? phi[0] : tb == 1 ? phi[1] : tb == 2 ? phi[2] : tb == 3 ? phi[3] : phi[4]);
		}
		const uint32_t slo = (uint32_t)__builtin_amdgcn_readlane(vslo, i - tv_base), shi = (uint32_t)__builtin_amdgcn_readlane(vshi, i - tv_base);
		if (beg < i - w) beg = i - w;
		end = wx_smin(wx_smin(end, i + w + 1), qlen);   // (left to itself the compiler does this three-way minimum on the vector unit and reads it back)
		int hleft0 = 0;
		if (beg == 0) { hleft0 = h0 - (P.o_del + e_del * (i + 1)); if (hleft0 < 0) hleft0 = 0; }
		const int n_col = end - beg;               // negative: the band has moved past the live cells, the row is empty
		int lanekey = -1;                          // (h << 13 | column), per lane over the strips of the row
		int A = beg * e_ins;                       // F(i,beg) = 0
		int h_carry = hleft0, h_cin = hleft0;      // h of the column before the strip (h_cin: before the row's last strip)
		int h_strip = 0;                           // the h of the row's last strip, a column per lane
		int first_nz = end, last_nz = -1;          // first / last cell of [beg,end] with H or E non-zero after the row
		const int jb = beg + lane;
		int je = lane_e + beg * e_ins;             // j * e_ins of the strip in hand (a multiplication per lane and row costs four additions' time)
		for (int s0 = 0; s0 <= n_col; s0 += 64, je += 64 * e_ins) {
			const int j = jb + s0;
			const bool act = j < end, wr = j <= end;
			// The lanes past the row's last column (they only exist in the row's LAST strip, to the right of every live lane) run on
			// whatever the pad holds (stale cells of earlier rows, any byte for a base): M, h and e' are forced to 0 for them — e' has to
			// be a real 0 in the one lane behind the last column (eh[end].e, src/ksw.c:447) — their g = j * e_ins is never seen by a live
			// lane (the scans run left to right; A and h_carry are only read off lanes that are live when they matter), and
			// their key (0 << 13 | j) can only win when every live h is 0: then the row maximum is 0 and the loop ends before anybody
			// looks at the column.
			const int2 he = HE[j];
			const int diag = he.x, e = he.y;
			const int sc = (int)(int8_t)__builtin_amdgcn_perm(shi, slo, (uint32_t)Qs[j] | 0x0c0c0c00u);
			const int M = (act && diag) ? diag + sc : 0;
			int tI = M - oe_ins; tI = tI > 0 ? tI : 0;
			const int g = tI + je;
			const int incl = wx_scan_max(g);
			const int excl = wx_prev_lane(incl, WX_NEG);
			const int f = max(A, excl + e_ins) - je;
			const int h = act ? max(max(M, e), f) : 0;
			int tD = M - oe_del; tD = tD > 0 ? tD : 0;
			const int en = act ? max(e - e_del, tD) : 0;
			const int h_prev = wx_prev_lane(h, h_carry);
			*(wr ? HE + j : L.dummy) = make_int2(h_prev, en);
			A = max(A, __builtin_amdgcn_readlane(incl, 63) + e_ins);
			lanekey = max(lanekey, h << 13 | j);   // the largest column wins ties
			// (behind the lane that writes eh[end], h_prev and e' are 0 by the masks above: no test for "a lane that writes")
			const unsigned long long nz = __builtin_amdgcn_ballot_w64((h_prev | en) != 0);
			if (nz) {
				const int lo = beg + s0 + __ffsll((long long)nz) - 1, hi = beg + s0 + 63 - __clzll(nz);
				if (lo < end && lo < first_nz) first_nz = lo;
				if (hi > last_nz) last_nz = hi;
			}
			h_cin = h_carry;
			h_carry = __builtin_amdgcn_readlane(h, 63);
			h_strip = h;
		}
		// h of the row's last column: in the last strip when that has live lanes, else what was carried into it (also: an empty row)
		const int h_last = (n_col > 0 && (n_col & 63)) ? __builtin_amdgcn_readlane(h_strip, (n_col - 1) & 63) : n_col > 0 ? h_cin : hleft0;
		int rowkey = -1;
		if (n_col > 0) {
			rowkey = __builtin_amdgcn_readlane(wx_scan_max(lanekey), 63);
			cells += (unsigned long long)n_col;
		} else if (n_col < 0) {
			if (lane == 0) HE[end] = make_int2(hleft0, 0);   // src/ksw.c:447 with an empty range (the loop ends below: m == 0)
		}
		__builtin_amdgcn_wave_barrier();
		const int jfin = n_col > 0 ? end : beg;    // value of the reference's column counter after its loop
		if (jfin == qlen) {
			if (h_last >= gscore) best_ie = i;
			if (h_last > gscore) gscore = h_last;
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
		// (the row maximum itself is one of the stored cells: B >= rowmax + (qlen - 1 - rowmax_j) * max_sc, a cheap test that (2) can hold)
		if (EARLY && (i & 3) == 3 && (gscore >= 0 || rowmax + (qlen - 1 - rowmax_j) * max_sc <= best - clip)) {
			int b = WX_NEG;
			for (int j = lane; j <= qlen; j += 64) {
				const int2 he = HE[j];
				const int hv = j < qlen ? he.x + (qlen - j) * max_sc : WX_NEG;
				const int ev = he.y + (qlen - 1 - j) * max_sc;
				b = max(b, max(hv, ev));
			}
			b = __builtin_amdgcn_readlane(wx_scan_max(b), 63);
			if (gscore >= 0 && b < gscore && b <= best) break;         // (1)
			if (b <= best - clip && gscore <= best - clip) break;      // (2)
		}
	}
	WxResult r;
	r.score = best; r.qle = best_j + 1; r.tle = best_i + 1; r.gtle = best_ie + 1; r.gscore = gscore; r.max_off = max_off;
	return r;
}

} // namespace mbw
#endif
