/* mpibwa_amd.h — C ABI of the MI355X-native BWA-MEM hot path.
 *
 * This header is the drop-in boundary for mpiBWA's alignment call
 *     mem_process_seqs(opt, bwt, bns, pac, n_processed, n, seqs, pes0)
 * (reference: src/bwamem.h:134, called from src/mainParallel.c:1314, :2355,
 * :3093 and src/mainParallelByChromosome.c:1249, :2502, :3407).
 *
 * Everything here is plain C: pointers, sizes, PODs.  No torch / HIP types
 * cross the boundary.  The struct layouts below are re-declared from the
 * reference's documented x86-64 layouts so that a host program compiled
 * against the reference headers can pass its own objects unchanged:
 *     mem_opt_t     src/bwamem.h:25-57   (168 bytes)
 *     mem_pestat_t  src/bwamem.h:81-85   (32 bytes)
 *     bseq1_t       src/bwa.h:30-33      (48 bytes)
 *     bwt_t         src/bwt.h:46-58      (1120 bytes incl. cnt_table)
 *     bntann1_t     src/bntseq.h:41-48   (40 bytes)
 *     bntamb1_t     src/bntseq.h:50-54   (16 bytes)
 *     bntseq_t      src/bntseq.h:56-64   (48 bytes)
 * Sizes and field offsets are checked at build time by static asserts in
 * mpibwa_amd/csrc/abi_check.cpp (C++) and tests/abi_check.c (plain C).
 */
#ifndef MPIBWA_AMD_H
#define MPIBWA_AMD_H

#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- flag bits of mem_opt_t::flag (src/bwamem.h:14-23) ---- */
#define MEM_F_PE             0x2
#define MEM_F_NOPAIRING      0x4
#define MEM_F_ALL            0x8
#define MEM_F_NO_MULTI       0x10
#define MEM_F_NO_RESCUE      0x20
#define MEM_F_REF_HDR        0x100
#define MEM_F_SOFTCLIP       0x200
#define MEM_F_SMARTPE        0x400
#define MEM_F_PRIMARY5       0x800
#define MEM_F_KEEP_SUPP_MAPQ 0x1000

typedef uint64_t bwtint_t;

typedef struct {
	int a, b;
	int o_del, e_del;
	int o_ins, e_ins;
	int pen_unpaired;
	int pen_clip5, pen_clip3;
	int w;
	int zdrop;
	uint64_t max_mem_intv;
	int T;
	int flag;
	int min_seed_len;
	int min_chain_weight;
	int max_chain_extend;
	float split_factor;
	int split_width;
	int max_occ;
	int max_chain_gap;
	int n_threads;
	int chunk_size;
	float mask_level;
	float drop_ratio;
	float XA_drop_ratio;
	float mask_level_redun;
	float mapQ_coef_len;
	int mapQ_coef_fac;
	int max_ins;
	int max_matesw;
	int max_XA_hits, max_XA_hits_alt;
	int8_t mat[25];
} mem_opt_t;

typedef struct {
	int low, high;
	int failed;
	double avg, std;
} mem_pestat_t;

typedef struct {
	int l_seq, id;
	char *name, *comment, *seq, *qual, *sam;
} bseq1_t;

typedef struct {
	bwtint_t primary;
	bwtint_t L2[5];
	bwtint_t seq_len;
	bwtint_t bwt_size;      /* in 32-bit words */
	uint32_t *bwt;          /* occ-interleaved BWT, 64 B per 128 bases */
	uint32_t cnt_table[256];
	int sa_intv;
	bwtint_t n_sa;
	bwtint_t *sa;
} bwt_t;

typedef struct {
	int64_t offset;
	int32_t len;
	int32_t n_ambs;
	uint32_t gi;
	int32_t is_alt;
	char *name, *anno;
} bntann1_t;

typedef struct {
	int64_t offset;
	int32_t len;
	char amb;
} bntamb1_t;

typedef struct {
	int64_t l_pac;
	int32_t n_seqs;
	uint32_t seed;
	bntann1_t *anns;
	int32_t n_holes;
	bntamb1_t *ambs;
	FILE *fp_pac;
} bntseq_t;

/* in-memory index handle, same fields as the reference's bwaidx_t
 * (src/bwa.h:20-28) so `.map` images can be attached in place. */
typedef struct {
	bwt_t    *bwt;
	bntseq_t *bns;
	uint8_t  *pac;
	int       is_shm;
	int64_t   l_mem;
	uint8_t  *mem;
} bwaidx_t;

/* ------------------------------------------------------------------ */
/* The drop-in entry point (replaces src/bwamem.c:1205-1234).          */
/*                                                                     */
/* Same argument meaning, same in/out contract on seqs[] (seq[] is     */
/* overwritten with nt4 codes, seqs[i].sam is malloc()ed and owned by  */
/* the caller), same stderr messages, failures abort.  The index       */
/* (bwt,bns,pac) must have been uploaded with mi355x_index_upload()    */
/* first; calling without a usable MI355X device aborts loudly — there */
/* is no CPU fallback.                                                 */
/*                                                                     */
/* Threads: the function may be called from several threads at once    */
/* (same index, disjoint seqs[]).  Up to eight calls run side by side  */
/* — the first half of a call is bound by the GPU, the second by the   */
/* host, so chunk i+1 overlaps chunk i — and further callers wait for  */
/* a free slot (calls in flight on overlapping seqs[] abort).          */
/* The reference's own function is re-entrant in the same way (it     */
/* only reads opt and the index).                                      */
/* ------------------------------------------------------------------ */
void mem_process_seqs(const mem_opt_t *opt, const bwt_t *bwt, const bntseq_t *bns, const uint8_t *pac,
                      int64_t n_processed, int n, bseq1_t *seqs, const mem_pestat_t *pes0);

/* option / scoring helpers used by the caller's CLI parsing
 * (src/bwamem.c:48-84, src/bwa.c:109-119, src/bwa.c:16-17). */
mem_opt_t *mem_opt_init(void);
void bwa_fill_scmat(int a, int b, int8_t mat[25]);
extern int  bwa_verbose;
extern char bwa_rg_id[256];
/* read-group / extra header lines of the caller's CLI (src/bwa.c:431-476; used by src/mainParallel.c:356, :368, :373):
 * bwa_set_rg() returns the unescaped "@RG..." line (malloc()ed) and fills bwa_rg_id with its ID, or 0 if malformed;
 * bwa_insert_header() appends an unescaped '@' line to the header text (realloc()ed) and returns it. */
char *bwa_set_rg(const char *s);
char *bwa_insert_header(const char *s, char *hdr);

/* index attach / load (src/bwa.c:262-345: bwa_idx_load_from_disk, bwa_mem2idx) */
bwaidx_t *bwa_idx_load_from_disk(const char *prefix, int which);
int       bwa_mem2idx(int64_t l_mem, uint8_t *mem, bwaidx_t *idx);
void      bwa_idx_destroy(bwaidx_t *idx);
/* `.map` packer (src/bwa.c:347-386, used by mpiBWAIdx src/pidx.c:52-63): turns a disk-loaded index into one contiguous
 * malloc()ed image [bwt_t][bwt words][sa][bntseq_t][ambs][anns][name\0anno\0...][pac] and re-attaches idx to it. */
int       bwa_idx2mem(bwaidx_t *idx);
/* mpiBWAIdx in one call: load <prefix>.{bwt,sa,ann,amb,pac}, pack, write the image to map_path (pointer fields zeroed). */
int       mi355x_write_map(const char *prefix, const char *map_path);

/* ---- MI355X-specific additions (no reference equivalent) ---- */

/* Select device `local_rank`, re-home the FM-index (64-B aligned occ blocks),
 * the sampled SA and the 2-bit pac into HBM.  Returns 0 on success; aborts if
 * no gfx950 device is usable. */
int  mi355x_index_upload(int local_rank, const bwt_t *bwt, const bntseq_t *bns, const uint8_t *pac);
/* Device pointers + sizes of the three index arrays, for the RCCL broadcast
 * done by the host program (rank 0 uploads, others call mi355x_index_alloc
 * then receive into these buffers). */
int  mi355x_index_alloc(int local_rank, const bwt_t *bwt_meta, const bntseq_t *bns);
int  mi355x_index_buffers(void **d_bwt, size_t *bwt_bytes, void **d_sa, size_t *sa_bytes, void **d_pac, size_t *pac_bytes);
/* broadcast path: copy between a caller-owned device buffer and index buffer `which` (0 occ blocks, 1 SA, 2 pac);
 * once all three are filled, mi355x_index_commit() expands the dense SA and makes the index usable */
int  mi355x_index_d2d(int which, void *ext_device_ptr, size_t bytes, int to_index);
int  mi355x_index_commit(void);
void mi355x_finalize(void);

/* One call per rank after the index has been attached (replaces the per-rank copy of src/parallel_aux.c:1779-1830):
 * selects GPU `local_rank`; with comm == NULL or comm->size == 1 the index is uploaded from idx; otherwise rank 0 of
 * the communicator uploads its copy and the occ blocks, the sampled SA and pac reach the other ranks' GPUs by
 * ncclBroadcast (RCCL over xGMI, in pieces), after which every rank expands its dense SA and jump table.
 * The library does not link MPI: the caller lends its own transport for the 128-byte RCCL bootstrap id. */
typedef struct {
	int rank, size;                                                  /* among the ranks that share the broadcast (one per GPU) */
	void (*bcast)(void *buf, size_t bytes, int root, void *user);    /* host-memory broadcast, e.g. a wrapper of MPI_Bcast */
	void *user;
} mi355x_comm_t;
int  mi355x_init(int local_rank, const bwaidx_t *idx, const mi355x_comm_t *comm);
/* GPUs visible to this process (a host program that starts more ranks than that lets ranks share a device) */
int  mi355x_device_count(void);
/* free / total bytes of HBM on the device the index lives on; 0 on success */
int  mi355x_device_memory(size_t *free_bytes, size_t *total_bytes);
/* (re)allocations of device / page-locked work buffers since the library was loaded: they stall every stream of the device, so
 * the number should stand still once every call context has seen its largest chunk */
unsigned long long mi355x_buffer_growths(void);
/* host threads the library uses for this rank's calls (its share of the node's usable CPUs; *ranks_on_node: the ranks it believes
 * share the node, from the launcher's environment).  mi355x_init prints it and refuses to start with fewer than 2. */
int  mi355x_rank_host_threads(int *ranks_on_node);
/* device-computed checksums of the three resident index arrays (occ blocks, sampled SA, pac): mi355x_init compares every rank's
 * with rank 0's after its broadcast and ends the run on a difference; a host that broadcasts by other means does the same with this */
int  mi355x_index_checksums(uint64_t out[3]);
/* mem_process_seqs calls the library runs side by side at most (further callers wait; fewer are admitted while their work buffers
 * would not fit in HBM): the most worker threads a chunk loop has use for */
int  mi355x_max_calls(void);
/* The first-use cost of `n_calls` call contexts (work buffers on the device and page-locked on the host, streams, the host thread
 * pool, the kernels' code objects: 1-2 s per context otherwise paid by the first chunks of the loop the reference brackets with
 * MPI_Wtime, src/mainParallel.c:1238-1319) paid now: n_calls mem_process_seqs calls side by side on n_reads reads of read_len
 * bases sampled from the reference itself, text thrown away.  opt / bwt / bns / pac: what the chunk loop will pass.  A caller
 * with other work left before its loop runs this on a thread of its own meanwhile.  n_calls is also taken as what the caller
 * will keep in flight: with three or more, its calls run in their many-callers mode from the first one on (otherwise the library
 * finds out by itself after two of them).  Returns the seconds it took. */
double mi355x_prewarm(const mem_opt_t *opt, const bwt_t *bwt, const bntseq_t *bns, const uint8_t *pac, int n_reads, int read_len, int n_calls);
/* seconds spent in the RCCL broadcast of the last mi355x_init (0 when none took place) */
double mi355x_init_bcast_seconds(void);

/* own bwa-compatible index builder (formats of src/bwt.c:385-462,
 * src/bntseq.c:66-96,275-328). Writes prefix.{pac,ann,amb,bwt,sa}. */
int  mi355x_index_build(const char *fasta, const char *prefix);
/* same .bwt/.sa, built in HBM (radix sort + prefix doubling on the device) from an N-free forward pac
 * of l_pac bases; used for GRCh38-sized synthetic references.  *seconds = device build time. */
int  mi355x_index_build_gpu(int device, const uint8_t *pac, int64_t l_pac, const char *prefix, double *seconds);

/* ---- stage-level entry points (used by tests/bench for kernel parity and
 *      roofline measurement; each runs ONLY the named HIP kernel) ---- */

/* SMEM seeding (mem_collect_intv, src/bwamem.c:114-162) for n reads.
 * seqs: concatenated nt4 bytes, off[n+1] offsets.  Output: per read up to
 * `cap` intervals (x0,x1,x2,info) = 4 x uint64 each, count in n_out[i],
 * sorted by info.  Returns 0, or -1 if some read overflowed `cap`. */
int mi355x_smem_batch(const mem_opt_t *opt, int n, const uint8_t *seqs, const int64_t *off,
                      int cap, uint64_t *intv_out, int *n_out, double *kernel_ms, uint64_t *algo_bytes);
/* Suffix-array lookup (bwt_sa, src/bwt.c:86-96) for n BWT rows. */
int mi355x_sa_batch(int n, const uint64_t *k, uint64_t *sa_out, double *kernel_ms, uint64_t *algo_bytes);
/* Same lookups answered from the dense SA that mi355x_index_upload() expands in HBM when memory allows
 * (dense != 0; returns -1 if the table is absent), or by the LF walk (dense == 0). */
int mi355x_sa_batch2(int n, const uint64_t *k, uint64_t *sa_out, double *kernel_ms, int dense);
/* expansion time in ms (0 if not expanded) and, through *bytes, the size of the dense table */
double mi355x_sa_dense_info(size_t *bytes);
/* Banded extension (ksw_extend2, src/ksw.c:380-479) for n independent jobs.
 * q/t: concatenated nt4 bytes; per job 5 ints in out: score,qle,tle,gtle,gscore,max_off (6). */
int mi355x_extend_batch(const mem_opt_t *opt, int n, const uint8_t *q, const int64_t *qoff,
                        const uint8_t *t, const int64_t *toff, const int *w, const int *h0,
                        const int *end_bonus, int *out6, double *kernel_ms, uint64_t *cells);

/* Chaining stage (mem_chain + mem_chain_flt, src/bwamem.c:251-385) for n_reads reads given by their seeds
 * (read r owns seeds seed_off[r] .. seed_off[r+1]: rbeg[], and qbeg/len interleaved in qbeg_len[]), computed by
 * chain_kernel (which = 0) or by the library's host path (which = 1).  Output per read from out[out_off[r]]:
 * n_chains (-1 = the kernel leaves the read to the host), then per kept chain rid, n_seeds, far_beg, far_end,
 * rmax0, rmax1, frac_rep (float bits) and n_seeds x (rbeg, qbeg, len) in the order mem_chain2aln visits them.
 * Returns the number of int64 written, or -1 if out_cap is too small. */
int64_t mi355x_chain_batch(const mem_opt_t *opt, const bntseq_t *bns, int n_reads, const int *lens, const int *l_rep,
                           const int64_t *seed_off, const uint64_t *rbeg, const int32_t *qbeg_len, int which,
                           int64_t *out, int64_t out_cap, int64_t *out_off);

/* Final global re-alignment (mem_reg2aln's loop src/bwamem.c:1106-1122 around bwa_gen_cigar2 src/bwa.c:121-207 and
 * ksw_global2 src/ksw.c:504-606) for n_req regions [rb,re) x [qb,qe) of read `read` (nt4 codes, read r =
 * reads[off[r]..off[r+1])) against the 2-bit packed reference `pac` of l_pac bases, computed by aln_kernel.
 * which: 0 = the product's dispatch (no-DP shortcut / narrow-band instantiation with hand-off to the full-size one),
 * 1 = DP requests straight to the full-size instantiation.
 * Per request out_hdr gets score, NM, n_cigar, md_len, flags (1 = the device declined: band matrix beyond its budget);
 * cigar (BAM-encoded u32) goes to cigar_out[cigar_cap * i ..], MD text to md_out[md_cap * i ..]. */
int mi355x_global_batch(const mem_opt_t *opt, int64_t l_pac, const uint8_t *pac, int n_reads, const uint8_t *reads,
                        const int64_t *off, int n_req, const int64_t *rb, const int64_t *re, const int *read,
                        const int *qb, const int *qe, const int *w, const int *truesc, int which,
                        int *out_hdr5, uint32_t *cigar_out, int cigar_cap, char *md_out, int md_cap, double *kernel_ms);

/* Pairing decisions of mem_sam_pe (src/bwamem_pair.c:250-393 with mem_sort_dedup_patch src/bwamem.c:437-489, the test of
 * mem_matesw :118-128, mem_mark_primary_se src/bwamem.c:493-569, mem_pair :182-243, mem_approx_mapq_se src/bwamem.c:952-976,
 * the XA test of src/bwamem_extra.c:91-110) for n_pairs pairs given by the regions of their ends as phase 1 leaves them
 * (regs: mi355x_pair_maxreg() records of 64 bytes per read — rb, re (int64), qb, qe, rid, score, truesc, w, seedcov, seedlen0 (int32), frac_rep
 * (float), pad; n_regs per read), computed by pair_simple_kernel.  status[k] = 1: the pair is decided — desc (56 bytes per read:
 * rb, re, qb, qe, req, rid, flag, mapq, score, sub) and req (40 bytes per read: rb, re, read, qb, qe, w2, truesc, pad) describe the
 * two records of its paired branch; any other value: the kernel leaves the pair to the library's host path (the value names
 * the test that said so).  Returns 0, or -1 when the kernel cannot use these insert-size statistics. */
int mi355x_pair_maxreg(void);
int mi355x_pair_batch(const mem_opt_t *opt, const bntseq_t *bns, const mem_pestat_t pes[4], int64_t n_processed, int n_pairs,
                      const void *regs, const int *n_regs, int max_len, uint8_t *status, void *desc, void *req);

/* Mate-rescue local alignment: ksw_align2() exactly as mem_matesw() calls it (src/bwamem_pair.c:150-177,
 * src/ksw.c:321-356), for n_req windows [rb,re) of a 2-bit packed reference (doubled coordinate) against reads
 * given as nt4 codes (read r = reads[off[r]..off[r+1]) ).  out8 per request: score, te, qe, score2, te2, tb, qb
 * (kswr_t, src/ksw.h:14-20) and flags (1 = the device declined, recompute on the host). */
int mi355x_matesw_batch(const mem_opt_t *opt, int64_t l_pac, const uint8_t *pac, int n_reads, const uint8_t *reads,
                        const int64_t *off, int n_req, const int64_t *rb, const int64_t *re, const int *read,
                        const int *is_rev, int *out8, double *kernel_ms);

/* ---- the caller's side of the boundary: FASTQ text -> mpiBWA's chunks -> bseq1_t[] ----
 * Record scan (find_reads_size_and_offsets, src/parallel_aux.c:682-832): offsets of the record starts
 * (rec_off[n] = end of data) and bases per record; returns the number of records or -(position+1) of a malformed one. */
int64_t mi355x_fastq_scan(const char *buf, int64_t len, int64_t cap, int64_t *rec_off, int32_t *rec_bases);
/* Chunk rule (find_chunks_info, src/parallel_aux.c:1510-1546): a chunk takes records until its base count EXCEEDS
 * maxsiz (= K/2 per file for equal-size pairs, src/mainParallel.c:947; K over both files for trimmed pairs, :1874,
 * pass bases2; K for single-end, :2773).  chunk_first[c] = first record of chunk c, chunk_first[n_chunks] = n. */
int64_t mi355x_fastq_chunks(const int32_t *bases1, const int32_t *bases2, int64_t n, int64_t maxsiz, int64_t cap,
                            int64_t *chunk_first);
/* bseq1_t[] of records [first, first+count) as mpiBWA's main builds them (src/mainParallel.c:1257-1301, :2271-2345):
 * strings NUL-terminated in place inside buf1 / buf2, mates interleaved, name cut at the first white space and a
 * trailing "/digit" dropped.  lockstep: equal-size mode (R2 cut at R1's offsets).  Returns the bases of the chunk. */
int64_t mi355x_fastq_fill(char *buf1, const int64_t *off1, char *buf2, const int64_t *off2, int64_t first, int64_t count,
                          int copy_comment, int lockstep, bseq1_t *seqs);

/* timing of the calling thread's last mem_process_seqs() call (of the last call of any thread when this thread
 * has made none), per stage (ms) */
typedef struct {
	double total_ms, h2d_ms, smem_ms, sa_ms, chain_ms, ext_ms, regs_ms, pestat_ms, sam_ms;
	double k_smem_ms, k_sa_ms, k_ext_ms;        /* HIP-event kernel times */
	uint64_t smem_bytes, sa_bytes, ext_cells;    /* algorithmic work counted on device */
	uint64_t n_reads, n_intv, n_seeds, n_chains, n_ext;
	double plan_ms, aln_ms, k_aln_ms;            /* SAM stage: decisions+collect, CIGAR kernel round trip, its HIP-event time */
	uint64_t n_aln;
	double phase1_ms;                            /* wall time of stages 2-6 (sub-batches overlap, so it is less than their sum) */
	double msw_ms, k_msw_ms;                     /* mate rescue: listing the alignments + waiting for them, HIP-event kernel time */
	uint64_t n_msw;                              /* local alignments computed by the mate-rescue kernel */
	double emit_ms;                              /* SAM stage: the pass that formats the records */
	uint64_t n_sub;                              /* sub-batches of the chunk = launches of each phase-1 kernel */
	uint64_t smem_tab_bytes;                     /* the part of smem_bytes (64 B per occ block) that the third pass took from its jump table instead of fetching */
	uint64_t n_sam_dev;                          /* SAM records written by sam_kernel (the rest are formatted by the host) */
	uint64_t n_pair_dev;                         /* pairs whose pairing decisions (mem_sam_pe) were taken on the device (pair_kernel.hip) */
} mi355x_stats_t;
void mi355x_last_stats(mi355x_stats_t *st);

/* host-logic test hook (no GPU involved): ksw_align2 (src/ksw.c:343-365) as used by mate rescue;
 * out7 = score,te,qe,score2,te2,tb,qb; portable != 0 runs the lane-by-lane statement instead of SSE2 */
void mi355x_host_ksw_align2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del,
                            int e_del, int o_ins, int e_ins, int xtra, int portable, int out7[7]);

/* host-logic test hook (no GPU involved): mem_sam_pe (src/bwamem_pair.c:250-393) as the library's host path runs it — the path of the
 * pairs the pairing kernel leaves to the host: rescue loop, more than eight hits per end, XA, supplementary lines — on the regions of the
 * two ends given as the reference's own mem_alnreg_t records (88 bytes each, src/bwamem.h:59-77; what mem_align1_core returns).
 * s[k].seq: nt4 codes; s[k].sam is set.  Returns the number of rescued hits. */
int   mi355x_host_sam_pe(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4], uint64_t id, bseq1_t s[2],
                         const void *regs0, int n0, const void *regs1, int n1);
/* the same kind of hook for mem_sort_dedup_patch (src/bwamem.c:437-489; the regions of one read as mem_chain2aln leaves them, kept ones
 * written back in their new order, their number returned; query: nt4 codes) and for the single-end half of worker2 (src/bwamem.c:
 * 1187-1203: mem_mark_primary_se with id, mem_reorder_primary5 under -5, mem_reg2sam; s->sam is set) */
int   mi355x_host_sort_dedup_patch(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, uint8_t *query, void *regs, int n);
void  mi355x_host_reg2sam_se(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, bseq1_t *s, const void *regs, int n, int64_t id);
/* ... and for mem_pestat (src/bwamem_pair.c:46-109): regs = the regions of n reads (mates interleaved) one after the other, n_regs[i] of read i */
void  mi355x_host_pestat(const mem_opt_t *opt, int64_t l_pac, int n, const void *regs, const int *n_regs, mem_pestat_t pes[4], int n_threads);
/* ... and for mem_flt_chained_seeds / mem_seed_sw (src/bwamem.c:571-617; reads of ~700 bp and more): the chains of a read by their seeds
 * (mem_seed_t records of 24 bytes, n_seeds[c] per chain, chain after chain); the kept seeds are written back at the start of every chain's
 * slot with their scores, n_seeds[c] updated */
void  mi355x_host_flt_chained_seeds(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, int l_query, const uint8_t *query, int n_chains,
                                    void *seeds, int *n_seeds);

/* CPUs usable by this process (cgroup quota aware) — what the host stages are sized to. */
int   mi355x_host_cpus(void);
/* Caller-side convenience equal to mpiBWA's copy_buffer_thr (src/mainParallel.c:103-127): concatenates
 * every seqs[i].sam into one malloc()ed buffer (returned, NUL-terminated, *total_len bytes) and free()s
 * the per-read strings. */
char *mi355x_collect_sam(bseq1_t *seqs, int n, size_t *total_len);
/* the same into a buffer the caller keeps from chunk to chunk (*buf / *cap: replaced when a chunk needs more); returns the length */
size_t mi355x_collect_sam_into(bseq1_t *seqs, int n, char **buf, size_t *cap);

/* ---- the caller's side behind the call: what mpiBWA does with a chunk's SAM text before it reaches the file ----
 * -f fixmate (fixmate(), src/fixmate.c:601-827): the lines of a pair get the mate's contig / position / strand, MQ, MC and ms tags;
 * seqs[2p].sam / seqs[2p+1].sam are replaced (malloc family).  mi355x_fixmate_pair: one pair, returns its number of SAM lines or -1
 * when the text is not a pair's (left untouched); mi355x_fixmate: the pairs of a chunk on the rank's host threads (call_fixmate,
 * src/parallel_aux.c:2164-2206), returns the lines or -(first read of a failing pair) - 1. */
int     mi355x_fixmate_pair(bseq1_t *s1, bseq1_t *s2, const bntseq_t *bns);
int64_t mi355x_fixmate(bseq1_t *seqs, int n, const bntseq_t *bns);
/* -g / -b output (deflate_block, src/bgzf.c:245-330; compress_and_write_bgzf_thread / _bam_thread, src/parallel_aux.c:2941-3176):
 * SAM text as BGZF blocks of whole records, compressed side by side, byte-identical for any thread count.  out needs
 * mi355x_bgzf_bound(len) bytes; returns the compressed size (0: cap too small).  mi355x_bgzf_eof: the 28-byte empty block the
 * reference appends to its .bam (src/mainParallel.c:1509-1516). */
size_t  mi355x_bgzf_bound(size_t len);
size_t  mi355x_bgzf_compress(const char *text, size_t len, int level, uint8_t *out, size_t cap);
size_t  mi355x_bgzf_eof(uint8_t out[28]);
/* mpiBWAByChr's routing (src/mainParallelByChromosome.c:1340-1455, :3437-3486): the records of `sam` by destination — contig
 * 0 .. n_seqs-1 by RNAME, then "discordant" (only when discordant != 0; a record whose RNAME and RNEXT are two different contigs
 * goes to its contig AND there), last "unmapped" (RNAME '*').  out_text[d] (malloc, NULL when empty) / out_len[d] for
 * d < n_seqs + 1 + (discordant != 0).  Returns the number of records or -(offset of a malformed line) - 1. */
int64_t mi355x_route_by_chr(const char *sam, size_t len, const bntseq_t *bns, int discordant, char **out_text, size_t *out_len);

#ifdef __cplusplus
}
#endif
#endif
