// common.cpp — tables and option defaults shared by the host stages.
#include "internal.h"
#include <malloc.h>
#include "hprof.h"
#include <time.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace mbw {

// ASCII -> nt4 code: A/a=0 C/c=1 G/g=2 T/t=3, '-'=5, everything else 4
// (same mapping as nst_nt4_table, src/bntseq.c:47-64).
static struct Nt4Init {
	uint8_t t[256];
	Nt4Init() {
		memset(t, 4, sizeof t);
		t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3;
		t['-'] = 5;
	}
} nt4_init;
const uint8_t *const nt4_table_ptr = nt4_init.t;

// Every chunk hands several hundred MB of malloc()ed SAM records to the caller, who frees them one by one.  With glibc's
// default trim threshold that memory goes back to the kernel and the next chunk pays ~35 000 page faults (0.3-0.5 s of
// system time) to get it again; keeping freed memory in the allocator makes the records of the next chunk land on warm pages.
// MPIBWA_MALLOC_DEFAULTS=1 leaves the allocator alone.
static bool tune_allocator()
{
	if (getenv("MPIBWA_MALLOC_DEFAULTS")) return false;
	mallopt(M_TRIM_THRESHOLD, 1 << 30);
	mallopt(M_TOP_PAD, 64 << 20);
	mallopt(M_MMAP_THRESHOLD, 32 << 20);   // the per-chunk work arrays (tens of MB each) come from the heap too instead of mmap/munmap per chunk
	return true;
}
static bool g_alloc_tuned = tune_allocator();

std::atomic<long long> g_hprof[HP_N];
std::atomic<long long> g_hcount[HP_N];
bool g_hprof_on = getenv("MPIBWA_PROF") != nullptr;
thread_local HProfLocal t_hprof;
static double tsc_per_ms()
{
	static double r = 0;
	if (r == 0) {   // calibrate once against the monotonic clock
		timespec a, b;
		clock_gettime(CLOCK_MONOTONIC, &a);
		const unsigned long long t0 = __rdtsc();
		do clock_gettime(CLOCK_MONOTONIC, &b); while ((b.tv_sec - a.tv_sec) * 1e9 + (b.tv_nsec - a.tv_nsec) < 2e6);
		r = (double)(__rdtsc() - t0) / (((b.tv_sec - a.tv_sec) * 1e9 + (b.tv_nsec - a.tv_nsec)) * 1e-6);
	}
	return r;
}
void hprof_report(const char *tag)
{
	if (!g_hprof_on) return;
	t_hprof.flush();
	static const char *nm[HP_N] = {"matesw", "ksw_align2", "reg2aln", "ksw_global2", "gen_alt", "aln2sam", "mark_primary", "pair", "dedup_patch"};
	fprintf(stderr, "[hprof %s] (thread-ms / calls; nested sections are counted in their parents too)", tag);
	for (int i = 0; i < HP_N; ++i) { fprintf(stderr, " %s=%.1f/%lld", nm[i], g_hprof[i].load() / tsc_per_ms(), g_hcount[i].load()); g_hprof[i] = 0; g_hcount[i] = 0; }
	fprintf(stderr, "\n");
}

} // namespace mbw

extern "C" {

int  bwa_verbose = 3;
char bwa_rg_id[256];

// score matrix: 5x5, rows/cols A C G T N (src/bwa.c:109-119)
void bwa_fill_scmat(int a, int b, int8_t mat[25])
{
	for (int i = 0; i < 5; ++i)
		for (int j = 0; j < 5; ++j)
			mat[i * 5 + j] = (i == 4 || j == 4) ? -1 : (i == j ? a : -b);
}

// defaults of src/bwamem.c:48-84
mem_opt_t *mem_opt_init(void)
{
	mem_opt_t *o = (mem_opt_t *)calloc(1, sizeof(mem_opt_t));
	o->a = 1; o->b = 4;
	o->o_del = o->o_ins = 6;
	o->e_del = o->e_ins = 1;
	o->w = 100;
	o->T = 30;
	o->zdrop = 100;
	o->pen_unpaired = 17;
	o->pen_clip5 = o->pen_clip3 = 5;
	o->max_mem_intv = 20;
	o->min_seed_len = 19;
	o->split_width = 10;
	o->max_occ = 500;
	o->max_chain_gap = 10000;
	o->max_ins = 10000;
	o->mask_level = 0.50f;
	o->drop_ratio = 0.50f;
	o->XA_drop_ratio = 0.80f;
	o->split_factor = 1.5f;
	o->chunk_size = 10000000;
	o->n_threads = 1;
	o->max_XA_hits = 5;
	o->max_XA_hits_alt = 200;
	o->max_matesw = 50;
	o->mask_level_redun = 0.95f;
	o->min_chain_weight = 0;
	o->max_chain_extend = 1 << 30;
	o->mapQ_coef_len = 50;
	o->mapQ_coef_fac = log(o->mapQ_coef_len);
	bwa_fill_scmat(o->a, o->b, o->mat);
	return o;
}

} // extern "C"
