/* driver_shim.c — TEST INFRASTRUCTURE ONLY (tests/test_driver.py; never built into or shipped with the product).
 *
 * LD_PRELOADed under mpibwa_amd/mpibwa_gpu on a box without a GPU, it answers the driver's four device entry points and hands
 * mem_process_seqs() to the REFERENCE's own implementation (oracle/_ref/libbwaref.so, path in MPIBWA_TEST_REFLIB), so that the
 * driver's host side — FASTQ chunks, -f fixmate, -g / -b BGZF blocks, --by-chr routing, MPI-IO — runs end to end in the CPU suite.
 * What it checks is the caller's code around the hot path, never the hot path: the aligned records of these runs are the
 * reference's.  The product itself has no such path: without this preload mi355x_init() ends the process when no MI355X is there. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef void (*process_t)(const void *, const void *, const void *, const uint8_t *, int64_t, int, void *, const void *);

void mem_process_seqs(const void *opt, const void *bwt, const void *bns, const uint8_t *pac, int64_t n_processed, int n, void *seqs, const void *pes0)
{
	static process_t ref;
	if (!ref) {
		const char *path = getenv("MPIBWA_TEST_REFLIB");
		void *h = path ? dlopen(path, RTLD_NOW | RTLD_LOCAL) : 0;
		if (!h || !(ref = (process_t)dlsym(h, "mem_process_seqs"))) { fprintf(stderr, "driver_shim: no reference library (%s)\n", path ? path : "MPIBWA_TEST_REFLIB unset"); exit(3); }
	}
	ref(opt, bwt, bns, pac, n_processed, n, seqs, pes0);
}
int mi355x_device_count(void) { return 1; }
int mi355x_init(int local_rank, const void *idx, const void *comm) { (void)local_rank; (void)idx; (void)comm; return 0; }
void mi355x_finalize(void) {}
double mi355x_prewarm(const void *opt, const void *bwt, const void *bns, const void *pac, int n_reads, int len, int n_calls)
{
	(void)opt; (void)bwt; (void)bns; (void)pac; (void)n_reads; (void)len; (void)n_calls;
	return 0;
}
