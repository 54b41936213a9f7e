/* oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the kernels on the mem_process_seqs() hot path,
 * written from the algorithm as documented in the reference (file:line cited
 * at each function).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libmpibwa_amd.so) never
 * links or calls it.
 *
 * Pinning: the reference ships no golden vectors for this path,
 * so the oracle is pinned against the reference ITSELF, compiled from its own
 * sources into oracle/_ref/libbwaref.so (see oracle/Makefile), on seeded
 * inputs (tests/test_oracle_vs_ref.py) and through the committed fixtures in
 * tests/golden/ that were generated from that build.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>

typedef struct {
	uint64_t primary, L2[5], seq_len;
	const uint32_t *bwt;      /* occ-interleaved words, bwa layout */
	const uint64_t *sa;       /* sampled SA, sa[0] = -1 */
	int sa_intv;
	/* instrumentation (algorithmic work) */
	uint64_t n_extend, n_blocks, n_sa_steps, n_sa_calls;
} orc_fm_t;

typedef struct { uint64_t x[3], info; } orc_intv_t;
typedef struct { int n, m; orc_intv_t *a; } orc_intv_v;

/* src/bwt.c:169-186 / 189-220 */
void orc_occ4(orc_fm_t *fm, uint64_t k, uint64_t cnt[4]);
void orc_2occ4(orc_fm_t *fm, uint64_t k, uint64_t l, uint64_t ck[4], uint64_t cl[4]);
/* src/bwt.c:107-129 */
uint64_t orc_occ(orc_fm_t *fm, uint64_t k, int c);
/* src/bwt.c:262-275 */
void orc_extend(orc_fm_t *fm, const orc_intv_t *ik, orc_intv_t ok[4], int is_back);
/* src/bwt.c:289-351 (max_intv = 0 as used by mem_collect_intv) */
int orc_smem1(orc_fm_t *fm, int len, const uint8_t *q, int x, int min_intv, orc_intv_v *mem);
/* src/bwt.c:358-379 */
int orc_seed_strategy1(orc_fm_t *fm, int len, const uint8_t *q, int x, int min_len, int max_intv, orc_intv_t *mem);
/* src/bwamem.c:114-162; result sorted by info; returns count (array malloc'ed into *out) */
int orc_collect_intv(orc_fm_t *fm, int len, const uint8_t *seq, int min_seed_len, float split_factor,
                     int split_width, uint64_t max_mem_intv, orc_intv_t **out);
/* src/bwt.c:86-96 */
uint64_t orc_sa(orc_fm_t *fm, uint64_t k);

/* src/ksw.c:380-479; out6 = {score,qle,tle,gtle,gscore,max_off}; returns #cells computed */
int64_t orc_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t mat[25],
                    int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0, int out6[6]);
/* src/ksw.c:504-606; cigar buffer must hold qlen+tlen entries; returns score */
int orc_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t mat[25],
                int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar, uint32_t *cigar);

#endif
