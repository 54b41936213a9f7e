/* ungapped_extend_check.c — TEST ONLY.  c2a_kernel (mpibwa_amd/csrc/c2a_kernel.hip) answers a seed extension WITHOUT the DP
 * when the flank matches the reference along the diagonal with at most one mismatch: then every off-diagonal cell of
 * ksw_extend2 (src/ksw.c:380-479) stays strictly below the diagonal cell of its row (a cell at distance d from the diagonal has
 * paid for a gap of d and has at most as many match columns), and all six outputs follow from the positions of the mismatch.
 * This program restates that closed form on the CPU and compares it with the oracle's ksw_extend2 restatement
 * (oracle/orc_ksw.c: orc_extend2, pinned against the reference by tests/test_oracle_vs_ref.py) on seeded random flanks.
 *
 *   usage: ungapped_extend_check <cases> <seed>      prints "cases closed mismatches"
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../../oracle/oracle.h"

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* 1 and out6 filled when the closed form applies (the rule of c2a_kernel.hip: ungapped_extend) */
static int ungapped_extend(int qlen, const uint8_t *q, int tlen, const uint8_t *t, const int8_t mat[25], int o_del, int e_del, int o_ins,
                           int e_ins, int zdrop, int h0, int out6[6])
{
	if (tlen < qlen || qlen <= 0) return 0;
	/* the scheme must be the plain one: one match score a > 0 on the diagonal, one mismatch score -b < 0 elsewhere */
	const int a = mat[0], b = -mat[1];
	if (a <= 0 || b <= 0) return 0;
	for (int i = 0; i < 4; ++i)
		for (int j = 0; j < 4; ++j)
			if (mat[i * 5 + j] != (i == j ? a : -b)) return 0;
	int mm = 0, p = -1;
	for (int k = 0; k < qlen; ++k) {
		if (q[k] > 3 || t[k] > 3) return 0;
		if (q[k] != t[k]) { if (++mm > 1) return 0; p = k; }
	}
	const int g1 = imin(o_del, o_ins) + imin(e_del, e_ins);
	if (mm == 1 && !((a + b) < g1 && h0 > b && (zdrop <= 0 || a + b <= zdrop))) return 0;
	const int end = h0 + a * qlen - (mm ? a + b : 0);   /* the diagonal's score at the end of the query */
	int best = h0, bl = 0;                                /* best cell: score and its (qle = tle) */
	if (mm == 0) { best = end; bl = qlen; }
	else {
		const int pre = h0 + a * p;                       /* the peak in front of the mismatch (p matched columns) */
		if (p > 0) { best = pre; bl = p; }
		if (end > best) { best = end; bl = qlen; }
	}
	out6[0] = best; out6[1] = bl; out6[2] = bl; out6[3] = qlen; out6[4] = end; out6[5] = 0;
	return 1;
}

static uint64_t rs;
static inline uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }

int main(int argc, char **argv)
{
	long n_cases = argc > 1 ? atol(argv[1]) : 100000;
	rs = argc > 2 ? strtoull(argv[2], 0, 10) * 0x9E3779B97F4A7C15ull + 1 : 88172645463325252ull;
	long closed = 0, bad = 0;
	uint8_t q[512], t[1024];
	for (long c = 0; c < n_cases; ++c) {
		int a = 1, b = 4, o_del = 6, e_del = 1, o_ins = 6, e_ins = 1, zdrop = 100, w_full = 100;
		const int scheme = rnd() % 8;
		if (scheme == 0) { a = 2; b = 3; o_del = 4; e_del = 2; o_ins = 5; e_ins = 1; }
		else if (scheme == 1) { a = 1; b = 1; o_del = 1; e_del = 1; o_ins = 1; e_ins = 1; zdrop = 20; }
		else if (scheme == 2) { a = 1; b = 9; o_del = 2; e_del = 1; o_ins = 3; e_ins = 1; w_full = 40; }
		else if (scheme == 3) { a = 1; b = 4; zdrop = 4; }
		else if (scheme == 4) { a = 3; b = 2; o_del = 5; e_del = 1; o_ins = 6; e_ins = 2; zdrop = 0; }
		int8_t mat[25];
		int k = 0;
		for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) mat[k++] = i == j ? a : -b; mat[k++] = -1; }
		for (int j = 0; j < 5; ++j) mat[k++] = -1;
		const int qlen = 1 + rnd() % (rnd() % 4 == 0 ? 280 : 140);
		const int h0 = (rnd() % 8 == 0 ? 1 + rnd() % 12 : 19 + rnd() % 120) * a;
		for (int j = 0; j < qlen; ++j) q[j] = rnd() & 3;
		/* the low-complexity flanks are the dangerous ones: a shifted diagonal matches as well as the main one */
		if (rnd() % 3 == 0) { const int per = 1 + rnd() % 4; for (int j = per; j < qlen; ++j) q[j] = q[j - per]; }
		int tl = qlen + (int)(rnd() % 3 == 0 ? 0 : rnd() % 150);
		if (rnd() % 12 == 0 && qlen > 3) tl = qlen - 1 - (int)(rnd() % (qlen / 2));
		for (int j = 0; j < tl; ++j) t[j] = j < qlen ? q[j] : (uint8_t)(rnd() % 3 == 0 ? q[j % qlen] : (rnd() & 3));
		const int nmm = rnd() % 4;   /* 0, 1, 2 or 3 substitutions */
		for (int x = 0; x < nmm && x < 3; ++x) { const int at = rnd() % 5 == 0 ? 0 : (int)(rnd() % imin(qlen, tl > 0 ? tl : 1)); if (at < tl) t[at] = (t[at] + 1 + rnd() % 3) & 3; }
		if (rnd() % 40 == 0) q[rnd() % qlen] = 4;
		if (rnd() % 40 == 0 && tl > 0) t[rnd() % tl] = 4;
		int max_ins = (int)((double)(qlen * a + 5 - o_ins) / e_ins + 1.), max_del = (int)((double)(qlen * a + 5 - o_del) / e_del + 1.);
		int w = imin(w_full << (rnd() & 1), imin(imax(max_ins, 1), imax(max_del, 1)));
		int full[6], cf[6];
		orc_extend2(qlen, q, tl, t, mat, o_del, e_del, o_ins, e_ins, w, 5, zdrop, h0, full);
		if (ungapped_extend(qlen, q, tl, t, mat, o_del, e_del, o_ins, e_ins, zdrop, h0, cf)) {
			++closed;
			if (memcmp(full, cf, sizeof full)) {
				if (bad < 8) fprintf(stderr, "MISMATCH case %ld qlen %d tlen %d h0 %d w %d scheme %d: dp %d %d %d %d %d %d closed %d %d %d %d %d %d\n", c, qlen, tl, h0, w, scheme,
				                     full[0], full[1], full[2], full[3], full[4], full[5], cf[0], cf[1], cf[2], cf[3], cf[4], cf[5]);
				++bad;
			}
		}
	}
	printf("%ld %ld %ld\n", n_cases, closed, bad);
	return bad != 0;
}
