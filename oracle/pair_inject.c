/* pair_inject.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Lets the reference's OWN mem_sam_pe() (src/bwamem_pair.c:250-393) — with its mem_matesw (:111-180), mem_pair (:182-243) and,
 * from libbwaref.so, mem_sort_dedup_patch, mem_mark_primary_se, mem_approx_mapq_se, mem_reorder_primary5 (src/bwamem.c) — run on
 * region lists chosen by a test, and reports what it DECIDED instead of the SAM text: nothing of the reference is restated here.
 * Its bwamem_pair.c is compiled where it lies (found through -I$(REF), see oracle/Makefile) with the functions that turn a
 * decision into text renamed to the recorders below:
 *   mem_reg2aln  (src/bwamem.c:1089)    -> which region mem_sam_pe reports for an end (no CIGAR is computed: the test has no reads)
 *   mem_aln2sam  (src/bwamem.c:825)     -> flag and MAPQ of every line it writes
 *   mem_reg2sam  (src/bwamem.c:1003)    -> "the ends are reported independently" (the branch behind no_pairing)
 *   mem_gen_alt  (src/bwamem_extra.c:98)-> "a hit of this end would get an XA tag" (its test :91-110 is evaluated by the reference's
 *                                          own mem_gen_alt on a copy whose CIGARs are not needed: see inj_gen_alt)
 *   ksw_align2   (src/ksw.c:321)        -> "mem_matesw would align here" (it gets a result of score 0: no hit is added)
 * Built into oracle/_ref/libpairinj.so (the reference's other objects come from libbwaref.so).  Used by
 * tests/test_gpu_kernels.py to check pair_simple_kernel against the reference on adversarial region lists.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "bwamem.h"
#include "kstring.h"
#include "ksw.h"
#include "kvec.h"

typedef struct {
	int n_reg2aln, n_lines, n_align, reg2sam, n_xa[2];
	int64_t rb[4], re[4];
	int qb[4], qe[4], score[4], sub[4], secondary[4], truesc[4], w[4];
	int flag[4], mapq[4];
} pair_record_t;

static pair_record_t g_rec;
static const mem_alnreg_v *g_a;   /* the two ends while mem_sam_pe runs */

mem_aln_t inj_reg2aln(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, int l_seq, const char *seq, const mem_alnreg_t *ar);
void inj_aln2sam(const mem_opt_t *opt, const bntseq_t *bns, kstring_t *str, bseq1_t *s, int n, const mem_aln_t *list, int which, const mem_aln_t *m);
void inj_reg2sam(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, bseq1_t *s, mem_alnreg_v *a, int extra_flag, const mem_aln_t *m);
char **inj_gen_alt(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_alnreg_v *a, int l_query, const char *query);
kswr_t inj_ksw_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int xtra, kswq_t **qry);

#define mem_reg2aln inj_reg2aln
#define mem_aln2sam inj_aln2sam
#define mem_reg2sam inj_reg2sam
#define mem_gen_alt inj_gen_alt
#define ksw_align2 inj_ksw_align2
#include "bwamem_pair.c"          /* the reference's file, in place */
#undef mem_reg2aln
#undef mem_aln2sam
#undef mem_reg2sam
#undef mem_gen_alt
#undef ksw_align2

mem_aln_t inj_reg2aln(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, int l_seq, const char *seq, const mem_alnreg_t *ar)
{
	mem_aln_t a;
	memset(&a, 0, sizeof a);
	if (ar == 0) { a.rid = -1; a.pos = -1; a.flag |= 0x4; return a; }
	if (g_rec.n_reg2aln < 4) {
		const int k = g_rec.n_reg2aln;
		g_rec.rb[k] = ar->rb; g_rec.re[k] = ar->re; g_rec.qb[k] = ar->qb; g_rec.qe[k] = ar->qe; g_rec.score[k] = ar->score;
		g_rec.sub[k] = ar->sub; g_rec.secondary[k] = ar->secondary; g_rec.truesc[k] = ar->truesc; g_rec.w[k] = ar->w;
	}
	++g_rec.n_reg2aln;
	a.rid = ar->rid;
	if (ar->secondary >= 0) a.flag |= 0x100;   /* src/bwamem.c:1103 */
	a.score = ar->score; a.sub = ar->sub > ar->csub ? ar->sub : ar->csub;
	return a;
}

void inj_aln2sam(const mem_opt_t *opt, const bntseq_t *bns, kstring_t *str, bseq1_t *s, int n, const mem_aln_t *list, int which, const mem_aln_t *m)
{
	if (g_rec.n_lines < 4) { g_rec.flag[g_rec.n_lines] = list[which].flag; g_rec.mapq[g_rec.n_lines] = list[which].mapq; }
	++g_rec.n_lines;
	kputc('x', str);              /* mem_sam_pe duplicates the string: it must exist */
}

void inj_reg2sam(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, bseq1_t *s, mem_alnreg_v *a, int extra_flag, const mem_aln_t *m)
{
	++g_rec.reg2sam;
	s->sam = 0;
}

/* mem_gen_alt decides per hit whether it goes into the XA list of its primary (src/bwamem_extra.c:98-115) and only then builds
 * CIGARs.  The decision is what the pairing kernel has to predict; the reference's own function is run on the end (it is a
 * different translation unit: the real mem_gen_alt of libbwaref.so) with a reference of one strand's worth of 'A' so that its
 * CIGAR step finds something to chew on, and the number of hits it lists is recorded. */
extern char **mem_gen_alt(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_alnreg_v *a, int l_query, const char *query);
char **inj_gen_alt(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_alnreg_v *a, int l_query, const char *query)
{
	char **xa = mem_gen_alt(opt, bns, pac, a, l_query, query);
	int i, n = 0;
	if (xa) {
		for (i = 0; i < (int)a->n; ++i) { if (xa[i]) ++n; free(xa[i]); }
		free(xa);
	}
	g_rec.n_xa[a == &g_a[1]] = n;
	return 0;
}

kswr_t inj_ksw_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int xtra, kswq_t **qry)
{
	kswr_t r;
	memset(&r, 0, sizeof r);
	r.qb = r.tb = -1; r.te = r.qe = -1; r.score2 = -1; r.te2 = -1;
	++g_rec.n_align;
	return r;
}

/* One pair: regs[0 .. n0) are the regions of end 0, regs[n0 .. n0 + n1) those of end 1 (as mem_align1_core leaves them, after
 * its mem_sort_dedup_patch is NOT assumed: this function runs it like worker1 does, src/bwamem.c:1185, with bns = 0: no
 * patching, since the test has no sequences).  pac: l_pac / 4 + 1 bytes of anything; l_seq: length of both reads.
 * out: the record above as 64 int64 (see pyoracle.ref_pair). */
extern int mem_sort_dedup_patch(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, uint8_t *query, int n, mem_alnreg_t *a);
int inj_sam_pe(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4], uint64_t id, int l_seq, int n0, int n1,
               const mem_alnreg_t *regs, int do_dedup, int64_t *out)
{
	mem_alnreg_v a[2];
	bseq1_t s[2];
	char name[] = "p", *seq = calloc(l_seq + 1, 1);
	int i, k, at = 0;
	for (i = 0; i < 2; ++i) {
		const int n = i ? n1 : n0;
		a[i].n = a[i].m = n;
		a[i].a = malloc((n + 64) * sizeof(mem_alnreg_t));
		a[i].m = n + 64;
		memcpy(a[i].a, regs + (i ? n0 : 0), n * sizeof(mem_alnreg_t));
		if (do_dedup) a[i].n = mem_sort_dedup_patch(opt, 0, 0, 0, a[i].n, a[i].a);
		for (k = 0; k < (int)a[i].n; ++k)
			if (a[i].a[k].rid >= 0 && bns->anns[a[i].a[k].rid].is_alt) a[i].a[k].is_alt = 1;   /* src/bwamem.c:1183-1186 */
		memset(&s[i], 0, sizeof s[i]);
		s[i].name = name; s[i].seq = seq; s[i].l_seq = l_seq;
	}
	memset(&g_rec, 0, sizeof g_rec);
	g_a = a;
	mem_sam_pe(opt, bns, pac, pes, id, s, a);
	free(s[0].sam); if (s[1].sam != s[0].sam) free(s[1].sam);
	free(a[0].a); free(a[1].a); free(seq);
	out[at++] = g_rec.n_reg2aln; out[at++] = g_rec.n_lines; out[at++] = g_rec.n_align; out[at++] = g_rec.reg2sam;
	out[at++] = g_rec.n_xa[0]; out[at++] = g_rec.n_xa[1];
	for (k = 0; k < 4; ++k) {
		out[at++] = g_rec.rb[k]; out[at++] = g_rec.re[k]; out[at++] = g_rec.qb[k]; out[at++] = g_rec.qe[k]; out[at++] = g_rec.score[k];
		out[at++] = g_rec.sub[k]; out[at++] = g_rec.secondary[k]; out[at++] = g_rec.truesc[k]; out[at++] = g_rec.w[k];
		out[at++] = g_rec.flag[k]; out[at++] = g_rec.mapq[k];
	}
	return at;
}
