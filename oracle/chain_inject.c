/* chain_inject.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Lets the reference's OWN mem_chain() and mem_chain_flt() (src/bwamem.c:251-315, 327-385) run on seed sets chosen by a
 * test instead of on what a read happens to produce.  Nothing of the reference is restated here: its bwamem.c is
 * compiled where it lies (found through -I$(REF), see oracle/Makefile) with the three FM-index functions that
 * mem_collect_intv / mem_chain call — bwt_smem1, bwt_seed_strategy1 (src/bwt.c:353, 358) and bwt_sa (src/bwt.c:86) —
 * renamed to the functions below, which hand out the test's intervals and suffix-array values.  Everything downstream
 * (the length filter and the sort by info of mem_collect_intv, l_rep, the occurrence stepping, bns_intv2rid, the B-tree,
 * test_and_merge, the traversal order, mem_chain_weight, the unstable sort and the masking of mem_chain_flt) is the
 * reference's code, unmodified.
 *
 * Built into oracle/_ref/libchaininj.so (the reference's other objects come from libbwaref.so).  Used by
 * tests/test_gpu_kernels.py to check chain_kernel against the reference on adversarial seed sets.
 */
#include <stdint.h>
#include <string.h>

#include "bwt.h"

static int g_n;                  /* injected intervals: x[0] = index of the first SA value in g_sa, x[2] = occurrences, info */
static const bwtintv_t *g_intv;
static const int64_t *g_sa;
static int g_served;             /* the first bwt_smem1 call of a read hands all of them out */

int inj_smem1(const bwt_t *bwt, int len, const uint8_t *q, int x, int min_intv, bwtintv_v *mem, bwtintv_v *tmpvec[2]);
int inj_seed_strategy1(const bwt_t *bwt, int len, const uint8_t *q, int x, int min_len, int max_intv, bwtintv_t *mem);
bwtint_t inj_sa(const bwt_t *bwt, bwtint_t k);

#define bwt_smem1 inj_smem1
#define bwt_seed_strategy1 inj_seed_strategy1
#define bwt_sa inj_sa
#include "bwamem.c"              /* the reference's file, in place */
#undef bwt_smem1
#undef bwt_seed_strategy1
#undef bwt_sa

int inj_smem1(const bwt_t *bwt, int len, const uint8_t *q, int x, int min_intv, bwtintv_v *mem, bwtintv_v *tmpvec[2])
{
	int i;
	mem->n = 0;
	if (!g_served) {
		g_served = 1;
		for (i = 0; i < g_n; ++i) kv_push(bwtintv_t, *mem, g_intv[i]);
	}
	return len;                  /* "the SMEM search reached the end of the read": pass 1 stops, pass 2 finds nothing new */
}

int inj_seed_strategy1(const bwt_t *bwt, int len, const uint8_t *q, int x, int min_len, int max_intv, bwtintv_t *mem)
{
	mem->x[0] = mem->x[1] = mem->x[2] = 0; mem->info = 0;
	return len;                  /* pass 3 finds nothing */
}

bwtint_t inj_sa(const bwt_t *bwt, bwtint_t k) { return (bwtint_t)g_sa[k]; }

/* Runs mem_chain (+ mem_chain_flt when do_flt) of the reference on the injected intervals for a read of `len` bases.
 * out: n_chains, then per chain  rid, n_seeds, w, kept, is_alt, frac_rep (float bits), n_seeds x (rbeg, qbeg, len)
 * in the order the chain array holds them.  Returns the number of int64 written or -1 if cap is too small. */
int64_t inj_chain(const mem_opt_t *opt, const bntseq_t *bns, int len, int n_intv, const bwtintv_t *intv, const int64_t *sa,
                  int do_flt, int64_t *out, int64_t cap)
{
	static bwt_t dummy;
	uint8_t *seq = calloc(len > 0 ? len : 1, 1);
	mem_chain_v chn;
	int64_t at = 0;
	size_t i;
	int j;
	g_n = n_intv; g_intv = intv; g_sa = sa; g_served = 0;
	chn = mem_chain(opt, &dummy, bns, len, seq, 0);
	if (do_flt) chn.n = mem_chain_flt(opt, chn.n, chn.a);
	free(seq);
	if (at + 1 > cap) return -1;
	out[at++] = (int64_t)chn.n;
	for (i = 0; i < chn.n; ++i) {
		const mem_chain_t *c = &chn.a[i];
		uint32_t fb;
		if (at + 6 + 3 * (int64_t)c->n > cap) return -1;
		memcpy(&fb, &c->frac_rep, 4);
		out[at++] = c->rid; out[at++] = c->n; out[at++] = c->w; out[at++] = c->kept; out[at++] = c->is_alt; out[at++] = fb;
		for (j = 0; j < c->n; ++j) { out[at++] = c->seeds[j].rbeg; out[at++] = c->seeds[j].qbeg; out[at++] = c->seeds[j].len; }
		free(c->seeds);
	}
	free(chn.a);
	return at;
}
