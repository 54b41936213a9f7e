/* orc_ksw.c — TEST INFRASTRUCTURE (see oracle.h).  CPU restatement of the
 * banded extension (src/ksw.c:380-479) and banded global alignment with
 * traceback (src/ksw.c:504-606).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

#define NEG_INF (-0x40000000)

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* Row-by-row affine-gap extension starting from score h0 at (-1,-1).
 * H[j]/E[j] hold, while row i is processed, H(i-1,j-1) and E(i,j); the live
 * column range [beg,end) follows the non-zero cells of the previous row and
 * the band |i-j| <= w.  Ties for the row maximum go to the LARGEST j, ties
 * between rows to the FIRST row.  Returns the number of cells computed. */
int64_t orc_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t mat[25],
                    int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0, int out6[6])
{
	int *H = (int *)calloc(qlen + 1, sizeof(int)), *E = (int *)calloc(qlen + 1, sizeof(int));
	int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	int64_t cells = 0;
	/* first row: only insertions from h0 */
	H[0] = h0;
	if (qlen >= 1) H[1] = h0 > oe_ins ? h0 - oe_ins : 0;
	for (int j = 2; j <= qlen && H[j - 1] > e_ins; ++j) H[j] = H[j - 1] - e_ins;
	/* shrink the band to what the scores can pay for (src/ksw.c:395-407) */
	int mx = 0;
	for (int i = 0; i < 25; ++i) mx = imax(mx, mat[i]);
	int max_ins = (int)((double)(qlen * mx + end_bonus - o_ins) / e_ins + 1.);
	int max_del = (int)((double)(qlen * mx + end_bonus - o_del) / e_del + 1.);
	w = imin(w, imax(max_ins, 1));
	w = imin(w, imax(max_del, 1));

	int best = h0, best_i = -1, best_j = -1, best_ie = -1, gscore = -1, max_off = 0;
	int beg = 0, end = qlen;
	for (int i = 0; i < tlen; ++i) {
		const int8_t *srow = mat + target[i] * 5;
		int f = 0, rowmax = 0, rowmax_j = -1, hleft;
		beg = imax(beg, i - w);
		end = imin(imin(end, i + w + 1), qlen);
		hleft = beg == 0 ? imax(h0 - (o_del + e_del * (i + 1)), 0) : 0;
		int j;
		for (j = beg; j < end; ++j) {
			int diag = H[j], e = E[j];
			H[j] = hleft;                       /* becomes H(i,j-1) for the next row */
			int M = diag ? diag + srow[query[j]] : 0;
			int h = imax(imax(M, e), f);
			hleft = h;
			if (h >= rowmax) rowmax_j = j;      /* >= : last column wins ties */
			rowmax = imax(rowmax, h);
			E[j] = imax(e - e_del, imax(M - oe_del, 0));
			f = imax(f - e_ins, imax(M - oe_ins, 0));
			++cells;
		}
		H[end] = hleft; E[end] = 0;
		if (j == qlen) {                        /* reached the end of the query */
			if (hleft >= gscore) best_ie = i;   /* gscore > h1 ? keep : take (ties -> later row) */
			gscore = imax(gscore, hleft);
		}
		if (rowmax == 0) break;
		if (rowmax > best) {
			best = rowmax; best_i = i; best_j = rowmax_j;
			max_off = imax(max_off, abs(rowmax_j - i));
		} else if (zdrop > 0) {
			int di = i - best_i, dj = rowmax_j - best_j;
			if (di > dj) { if (best - rowmax - (di - dj) * e_del > zdrop) break; }
			else         { if (best - rowmax - (dj - di) * e_ins > zdrop) break; }
		}
		/* next row's live range: drop leading / trailing all-zero cells */
		for (j = beg; j < end && H[j] == 0 && E[j] == 0; ++j) {}
		beg = j;
		for (j = end; j >= beg && H[j] == 0 && E[j] == 0; --j) {}
		end = imin(j + 2, qlen);
	}
	free(H); free(E);
	out6[0] = best; out6[1] = best_j + 1; out6[2] = best_i + 1; out6[3] = best_ie + 1; out6[4] = gscore; out6[5] = max_off;
	return cells;
}

/* Banded Needleman-Wunsch; per cell a byte d = (F-continues)<<4 | (E-continues)<<2 | (H source)
 * steers the traceback.  src/ksw.c:504-606 */
int orc_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t mat[25],
                int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar_, uint32_t *cigar)
{
	int n_col = imin(qlen, 2 * w + 1);
	int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	int *H = (int *)malloc((qlen + 1) * sizeof(int)), *E = (int *)malloc((qlen + 1) * sizeof(int));
	uint8_t *z = cigar ? (uint8_t *)malloc((size_t)n_col * tlen + 1) : 0;
	H[0] = 0; E[0] = NEG_INF;
	for (int j = 1; j <= qlen; ++j) {
		H[j] = j <= w ? -(o_ins + e_ins * j) : NEG_INF;
		E[j] = NEG_INF;
	}
	for (int i = 0; i < tlen; ++i) {
		const int8_t *srow = mat + target[i] * 5;
		int beg = i > w ? i - w : 0, end = imin(i + w + 1, qlen);
		int f = NEG_INF, hleft = beg == 0 ? -(o_del + e_del * (i + 1)) : NEG_INF;
		for (int j = beg; j < end; ++j) {
			int m = H[j] + srow[query[j]], e = E[j], h, t;
			uint8_t d;
			H[j] = hleft;
			d = m >= e ? 0 : 1; h = m >= e ? m : e;
			if (h < f) { d = 2; h = f; }
			hleft = h;
			t = m - oe_del; e -= e_del;
			if (e > t) d |= 1 << 2; else e = t;
			E[j] = e;
			t = m - oe_ins; f -= e_ins;
			if (f > t) d |= 2 << 4; else f = t;
			if (z) z[(size_t)i * n_col + (j - beg)] = d;
		}
		H[end] = hleft; E[end] = NEG_INF;
	}
	int score = H[qlen];
	if (z) {
		int n = 0, which = 0, i = tlen - 1, k = imin(i + w + 1, qlen) - 1;
		/* collect ops backwards, run-length encoded as len<<4|op (0=M 1=I 2=D) */
		#define PUSH(op, len) do { if (n && (cigar[n-1] & 0xf) == (uint32_t)(op)) cigar[n-1] += (uint32_t)(len) << 4; \
		                           else cigar[n++] = (uint32_t)(len) << 4 | (op); } while (0)
		while (i >= 0 && k >= 0) {
			which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
			if (which == 0) { PUSH(0, 1); --i; --k; }
			else if (which == 1) { PUSH(2, 1); --i; }
			else { PUSH(1, 1); --k; }
		}
		if (i >= 0) PUSH(2, i + 1);
		if (k >= 0) PUSH(1, k + 1);
		#undef PUSH
		for (int a = 0, b = n - 1; a < b; ++a, --b) { uint32_t t = cigar[a]; cigar[a] = cigar[b]; cigar[b] = t; }
		*n_cigar_ = n;
		free(z);
	}
	free(H); free(E);
	return score;
}
