/* orc_fmindex.c — TEST INFRASTRUCTURE (see oracle.h).  CPU restatement of the
 * FM-index operations and SMEM seeding of the reference, counting the
 * algorithmic memory work as it goes.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

/* occurrences of base c among the first m (1..16) bases of a packed word,
 * bases stored MSB-first, 2 bits each (layout: src/bwt.h:72-78) */
static inline int cnt_word(uint32_t w, int c, int m)
{
	uint32_t y = ~(w ^ (0x55555555u * (uint32_t)c));
	uint32_t t = y & (y >> 1) & 0x55555555u;
	if (m < 16) t &= ~((1u << (32 - 2 * m)) - 1u);
	return __builtin_popcount(t);
}

/* Occ(c, k) for all four c: number of c in B[0..k] where B is the BWT without
 * '$'; rows at or after `primary` shift down by one.  src/bwt.c:169-186 */
void orc_occ4(orc_fm_t *fm, uint64_t k, uint64_t cnt[4])
{
	if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
	if (k >= fm->primary) --k;
	const uint32_t *blk = fm->bwt + ((k >> 7) << 4);
	memcpy(cnt, blk, 32);
	const uint32_t *w = blk + 8;
	int full = (int)((k & 127) >> 4), rem = (int)(k & 15) + 1;
	for (int c = 0; c < 4; ++c) {
		int n = 0;
		for (int i = 0; i < full; ++i) n += cnt_word(w[i], c, 16);
		n += cnt_word(w[full], c, rem);
		cnt[c] += n;
	}
}

uint64_t orc_occ(orc_fm_t *fm, uint64_t k, int c)
{
	uint64_t cnt[4];
	if (k == fm->seq_len) return fm->L2[c + 1] - fm->L2[c]; /* src/bwt.c:112 */
	orc_occ4(fm, k, cnt);
	return cnt[c];
}

/* src/bwt.c:189-220: two Occ4 queries, k <= l; counted as 1 block when both
 * fall into the same 128-base block, else one block per real query */
void orc_2occ4(orc_fm_t *fm, uint64_t k, uint64_t l, uint64_t ck[4], uint64_t cl[4])
{
	uint64_t kk = k - (k >= fm->primary), ll = l - (l >= fm->primary);
	if (k == (uint64_t)-1 || l == (uint64_t)-1) fm->n_blocks += (k != (uint64_t)-1) + (l != (uint64_t)-1);
	else fm->n_blocks += (kk >> 7) == (ll >> 7) ? 1 : 2;
	orc_occ4(fm, k, ck);
	orc_occ4(fm, l, cl);
}

/* src/bwt.c:262-275 */
void orc_extend(orc_fm_t *fm, const orc_intv_t *ik, orc_intv_t ok[4], int is_back)
{
	uint64_t tk[4], tl[4];
	int a = !is_back, b = is_back; /* a: the side searched in the BWT; b: the mirrored side */
	++fm->n_extend;
	orc_2occ4(fm, ik->x[a] - 1, ik->x[a] - 1 + ik->x[2], tk, tl);
	for (int c = 0; c < 4; ++c) {
		ok[c].x[a] = fm->L2[c] + 1 + tk[c];
		ok[c].x[2] = tl[c] - tk[c];
	}
	/* the mirrored side is laid out T,G,C,A after the (optional) sentinel */
	uint64_t acc = ik->x[b] + (ik->x[a] <= fm->primary && ik->x[a] + ik->x[2] - 1 >= fm->primary);
	for (int c = 3; c >= 0; --c) { ok[c].x[b] = acc; acc += ok[c].x[2]; }
}

static void push(orc_intv_v *v, const orc_intv_t *e)
{
	if (v->n == v->m) { v->m = v->m ? v->m * 2 : 16; v->a = (orc_intv_t *)realloc(v->a, sizeof(orc_intv_t) * v->m); }
	v->a[v->n++] = *e;
}
static void reverse(orc_intv_v *v)
{
	for (int i = 0, j = v->n - 1; i < j; ++i, --j) { orc_intv_t t = v->a[i]; v->a[i] = v->a[j]; v->a[j] = t; }
}
static void set_intv(const orc_fm_t *fm, int c, orc_intv_t *ik) /* src/bwt.h:80 */
{
	ik->x[0] = fm->L2[c] + 1; ik->x[2] = fm->L2[c + 1] - fm->L2[c]; ik->x[1] = fm->L2[3 - c] + 1; ik->info = 0;
}

/* src/bwt.c:289-351 with max_intv == 0.  Forward phase: extend q[x..] to the
 * right, remembering the interval each time its size is about to change.
 * Backward phase: extend all remembered intervals to the left base by base;
 * an interval that cannot be extended (or would drop below min_intv) is a
 * super-maximal match iff no longer one is still alive and it is not contained
 * in the previously emitted one. */
int orc_smem1(orc_fm_t *fm, int len, const uint8_t *q, int x, int min_intv, orc_intv_v *mem)
{
	orc_intv_v cur = {0, 0, 0}, nxt = {0, 0, 0};
	orc_intv_t ik, ok[4];
	int i, ret;
	mem->n = 0;
	if (q[x] > 3) return x + 1;
	if (min_intv < 1) min_intv = 1;
	set_intv(fm, q[x], &ik);
	ik.info = x + 1;
	for (i = x + 1; i < len; ++i) {
		if (q[i] > 3) { push(&cur, &ik); break; }
		int c = 3 - q[i];
		orc_extend(fm, &ik, ok, 0);
		if (ok[c].x[2] != ik.x[2]) {
			push(&cur, &ik);
			if (ok[c].x[2] < (uint64_t)min_intv) break;
		}
		ik = ok[c]; ik.info = i + 1;
	}
	if (i == len) push(&cur, &ik);
	reverse(&cur); /* longest match first */
	ret = (int)cur.a[0].info;
	for (i = x - 1; i >= -1; --i) {
		int c = (i < 0 || q[i] > 3) ? -1 : q[i];
		nxt.n = 0;
		for (int j = 0; j < cur.n; ++j) {
			orc_intv_t *p = &cur.a[j];
			if (c >= 0) orc_extend(fm, p, ok, 1);
			if (c < 0 || ok[c].x[2] < (uint64_t)min_intv) {
				if (nxt.n == 0 && (mem->n == 0 || (uint64_t)(i + 1) < mem->a[mem->n - 1].info >> 32)) {
					orc_intv_t m = *p;
					m.info |= (uint64_t)(i + 1) << 32;
					push(mem, &m);
				}
			} else if (nxt.n == 0 || ok[c].x[2] != nxt.a[nxt.n - 1].x[2]) {
				ok[c].info = p->info;
				push(&nxt, &ok[c]);
			}
		}
		if (nxt.n == 0) break;
		orc_intv_v t = cur; cur = nxt; nxt = t;
	}
	reverse(mem); /* by start coordinate */
	free(cur.a); free(nxt.a);
	return ret;
}

/* src/bwt.c:358-379 */
int orc_seed_strategy1(orc_fm_t *fm, int len, const uint8_t *q, int x, int min_len, int max_intv, orc_intv_t *mem)
{
	orc_intv_t ik, ok[4];
	memset(mem, 0, sizeof *mem);
	if (q[x] > 3) return x + 1;
	set_intv(fm, q[x], &ik);
	for (int i = x + 1; i < len; ++i) {
		if (q[i] > 3) return i + 1;
		int c = 3 - q[i];
		orc_extend(fm, &ik, ok, 0);
		if (ok[c].x[2] < (uint64_t)max_intv && i - x >= min_len) {
			*mem = ok[c];
			mem->info = (uint64_t)x << 32 | (uint32_t)(i + 1);
			return i + 1;
		}
		ik = ok[c];
	}
	return len;
}

static int cmp_info(const void *a, const void *b)
{
	const orc_intv_t *p = (const orc_intv_t *)a, *q = (const orc_intv_t *)b;
	if (p->info != q->info) return p->info < q->info ? -1 : 1;
	/* equal info => same substring of the read => same bi-interval; order is unobservable.
	 * Keep qsort deterministic anyway. */
	if (p->x[0] != q->x[0]) return p->x[0] < q->x[0] ? -1 : 1;
	return 0;
}

/* src/bwamem.c:114-162 */
int orc_collect_intv(orc_fm_t *fm, int len, const uint8_t *seq, int min_seed_len, float split_factor,
                     int split_width, uint64_t max_mem_intv, orc_intv_t **out)
{
	orc_intv_v all = {0, 0, 0}, m1 = {0, 0, 0};
	int split_len = (int)(min_seed_len * split_factor + .499);
	int x = 0;
	while (x < len) { /* pass 1: all SMEMs */
		if (seq[x] > 3) { ++x; continue; }
		x = orc_smem1(fm, len, seq, x, 1, &m1);
		for (int i = 0; i < m1.n; ++i)
			if ((int)((uint32_t)m1.a[i].info - (m1.a[i].info >> 32)) >= min_seed_len) push(&all, &m1.a[i]);
	}
	int old_n = all.n;
	for (int k = 0; k < old_n; ++k) { /* pass 2: re-seed long, rare SMEMs from their middle */
		orc_intv_t p = all.a[k];
		int start = (int)(p.info >> 32), end = (int32_t)p.info;
		if (end - start < split_len || p.x[2] > (uint64_t)split_width) continue;
		orc_smem1(fm, len, seq, (start + end) >> 1, (int)p.x[2] + 1, &m1);
		for (int i = 0; i < m1.n; ++i)
			if ((int)((uint32_t)m1.a[i].info - (m1.a[i].info >> 32)) >= min_seed_len) push(&all, &m1.a[i]);
	}
	if (max_mem_intv > 0) { /* pass 3: LAST-like forward seeds */
		x = 0;
		while (x < len) {
			if (seq[x] > 3) { ++x; continue; }
			orc_intv_t m;
			x = orc_seed_strategy1(fm, len, seq, x, min_seed_len, (int)max_mem_intv, &m);
			if (m.x[2] > 0) push(&all, &m);
		}
	}
	qsort(all.a, all.n, sizeof(orc_intv_t), cmp_info);
	free(m1.a);
	*out = all.a;
	return all.n;
}

/* src/bwt.c:53-59 and 86-96: walk LF until a sampled row */
uint64_t orc_sa(orc_fm_t *fm, uint64_t k)
{
	uint64_t steps = 0, mask = (uint64_t)fm->sa_intv - 1;
	++fm->n_sa_calls;
	while (k & mask) {
		if (k == fm->primary) k = 0;
		else {
			uint64_t x = k - (k > fm->primary);
			uint32_t w = fm->bwt[((x >> 7) << 4) + 8 + ((x & 127) >> 4)];
			int c = (w >> ((~x & 15) << 1)) & 3;
			k = fm->L2[c] + orc_occ(fm, k, c);
		}
		++steps; ++fm->n_sa_steps;
	}
	return steps + fm->sa[k / fm->sa_intv];
}
