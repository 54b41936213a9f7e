/* mpibwa_gpu.c — a thin MPI host program around the C ABI of libmpibwa_amd.so: one rank per GPU,
 *
 *     mpiexec -n N mpibwa_gpu mem [bwa mem options] [-f] [-g | -b] [--by-chr] [--ordered] [-K bases] [--in-flight chunks] [--no-prewarm] [--dry-run] -o OUT PREFIX R1.fastq [R2.fastq]
 *
 * The `mem` options are the reference's (src/mainParallel.c:291-398: -k -w -A -B -O -E -L -U -T -c -d -r -D -m -s -G -N -W -y -X -h -Q -I -R -H
 * -P -a -M -S -Y -V -5 -q -j -C -v -t -K -x -o, and its output options -f (fixmate, :395), -g (BGZF, :299), -b (BGZF + EOF block under the
 * name the reference calls BAM, :298)); -p and -z are not offered.  --by-chr is mpiBWAByChr (src/mainParallelByChromosome.c): one file
 * per contig plus discordant (pairs without -f) and unmapped in the directory of OUT.
 *
 * It does, with this repo's own code, what mpiBWA's main does around mem_process_seqs():
 *   - every rank takes a byte slice of each FASTQ file, finds the first record boundary in it and scans its records
 *     (offset discovery: src/parallel_aux.c:262-476, record scan :682-832);
 *   - the chunk rule — a chunk takes reads until its base count EXCEEDS maxsiz, src/parallel_aux.c:1510-1546 — is a
 *     running count over the whole file, so it is carried from rank to rank along a pipeline (:1553-1561, 1373-1737):
 *     chunk boundaries do not depend on the number of ranks;
 *   - the chunk table is replicated, chunks are handed out by an RMA fetch-and-add counter (src/mainParallel.c:1112-1119),
 *     each chunk's bytes are read with MPI-IO, turned into bseq1_t (mi355x_fastq_fill), aligned (mem_process_seqs) and
 *     its SAM text appended through the shared file pointer (MPI_File_write_shared, src/mainParallel.c:1390 ff.).
 * Modes as in the reference: single end; pairs in two files of equal byte size (maxsiz = K/2 per file, R2 cut at R1's
 * header offsets); pairs in files of different size (trimmed reads: maxsiz = K over both files, n_processed = reads the
 * rank has already aligned).  The SAM body (all lines not starting with '@') is the reference's for the same -K.
 * The index reaches the GPUs through mi355x_init(): the ranks of a node that have a GPU each get it by RCCL broadcast
 * from the node's first rank; where ranks outnumber GPUs (tests on a one-GPU box) every rank uploads its own copy.
 *
 * Reading, aligning and writing overlap (the reference's docs/TODO:4, "overlaping reading, aligning and writing"): the
 * chunk loop runs on --in-flight worker threads per rank (default 6), each of them fetching the next chunk number, reading its bytes
 * with MPI-IO, building the bseq1_t array, calling mem_process_seqs() — the library runs up to mi355x_max_calls() = twelve calls side by side,
 * the GPU half of one chunk under the host half of another — and appending the SAM text through the shared file pointer.
 * With MPI_THREAD_MULTIPLE the workers call MPI concurrently; with MPI_THREAD_SERIALIZED a mutex takes turns; below that
 * one worker runs (the reference's blocking loop, src/mainParallel.c:1112-1391).
 *
 * Behind the call (the caller's side; the passes are the library's, csrc/sampost.cpp): -f rewrites the lines of every pair with
 * the mate's fields and the MQ / MC / ms tags (src/fixmate.c:601-827), -g / -b compress a chunk's text into BGZF blocks of whole records
 * (src/parallel_aux.c:2941-3176; every record is written — the reference drops the last read of each thread's slice), --by-chr routes the
 * records by RNAME (src/mainParallelByChromosome.c:1340-1455, :3437-3486).
 *
 * Plain C99 + MPI + pthreads; built by mpibwa_amd/build.py when an MPI installation is found (mpibwa_amd/mpibwa_gpu).
 */
#define _GNU_SOURCE
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <pthread.h>
#include <ctype.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <math.h>
#include "mpibwa_amd.h"

#define DIE(...) do { fprintf(stderr, "[mpibwa_gpu] " __VA_ARGS__); fputc('\n', stderr); MPI_Abort(MPI_COMM_WORLD, 1); } while (0)
#define MPI_OK(call) do { int e_ = (call); if (e_ != MPI_SUCCESS) DIE("%s failed (%s:%d)", #call, __FILE__, __LINE__); } while (0)

#define MAX_WORKERS 32
static int g_rank, g_size;

/* ---- one FASTQ file as this rank sees it ---- */
typedef struct {
	MPI_File fh;
	MPI_Offset size;
	int64_t n_local;        /* records that start in this rank's slice */
	int64_t first_index;    /* global index of the first of them */
	int64_t n_total;
	int64_t *off;           /* absolute offset of every local record, + the end of the last one */
	int32_t *bases;
} fq_t;

static void read_at(MPI_File fh, MPI_Offset at, char *buf, int64_t len)
{
	while (len > 0) {   /* MPI counts are ints */
		int piece = len > (1 << 30) ? (1 << 30) : (int)len;
		MPI_Status st;
		MPI_OK(MPI_File_read_at(fh, at, buf, piece, MPI_BYTE, &st));
		at += piece; buf += piece; len -= piece;
	}
}

/* is there a record at p?  "@..\n<bases>\n+..\n<quals of the same length>" (a quality line may itself start with '@') */
static int record_at(const char *p, const char *end)
{
	const char *l[5];
	int k;
	if (p >= end || *p != '@') return 0;
	l[0] = p;
	for (k = 1; k < 5; ++k) {
		const char *q = memchr(l[k - 1], '\n', (size_t)(end - l[k - 1]));
		if (!q) { if (k == 4) { l[4] = end + 1; break; } return -1; }   /* ran out of bytes (k == 4: a last line without newline) */
		l[k] = q + 1;
	}
	if (k < 4) return -1;
	if (*l[2] != '+') return 0;
	return (l[2] - l[1]) == (l[4] - l[3]) ? 1 : 0;
}

/* first record starting at or after `lo` (absolute offset); `size` if there is none */
static MPI_Offset first_record_from(MPI_File fh, MPI_Offset size, MPI_Offset lo)
{
	int64_t window = 1 << 16;
	if (lo <= 0) return 0;
	if (lo >= size) return size;
	for (;;) {
		/* the window starts one byte early: a record starts right behind a newline */
		const MPI_Offset w0 = lo - 1;
		const int64_t len = size - w0 < window ? (int64_t)(size - w0) : window;
		const int at_eof = w0 + len >= size;
		char *buf = malloc((size_t)len + 1);
		read_at(fh, w0, buf, len);
		int undecided = 0;
		MPI_Offset found = -1;
		for (int64_t i = 0; i + 1 < len; ++i) {
			if (buf[i] != '\n') continue;
			const int r = record_at(buf + i + 1, buf + len);
			if (r == 1) { found = w0 + i + 1; break; }
			if (r < 0 && !at_eof) { undecided = 1; break; }   /* the candidate runs out of the window: look at more bytes */
		}
		free(buf);
		if (found >= 0) return found;
		if (!undecided && at_eof) return size;
		if (at_eof) return size;
		window *= 4;
	}
}

static void fq_open(fq_t *f, const char *path)
{
	memset(f, 0, sizeof *f);
	MPI_OK(MPI_File_open(MPI_COMM_WORLD, path, MPI_MODE_RDONLY, MPI_INFO_NULL, &f->fh));
	MPI_OK(MPI_File_get_size(f->fh, &f->size));
	/* byte slices -> record slices: every rank finds the first record of its slice, its slice ends where the next one starts */
	MPI_Offset lo = f->size / g_size * g_rank, start = first_record_from(f->fh, f->size, lo), next;
	long long s = (long long)start, *all = malloc(sizeof(long long) * (size_t)(g_size + 1));
	MPI_OK(MPI_Allgather(&s, 1, MPI_LONG_LONG, all, 1, MPI_LONG_LONG, MPI_COMM_WORLD));
	all[g_size] = (long long)f->size;
	next = (MPI_Offset)all[g_rank + 1];   /* (starts are non-decreasing: a slice without a record start is empty) */
	if (next < start) next = start;
	free(all);
	/* scan the slice */
	int64_t len = next - start, cap = len / 8 + 16;
	char *buf = malloc((size_t)len + 1);
	read_at(f->fh, start, buf, len);
	f->off = malloc(sizeof(int64_t) * (size_t)(cap + 1));
	f->bases = malloc(sizeof(int32_t) * (size_t)cap);
	int64_t n = mi355x_fastq_scan(buf, len, cap, f->off, f->bases);
	if (n < 0) DIE("%s: malformed FASTQ record at byte %lld", path, (long long)(start - n - 1));
	free(buf);
	f->n_local = n;
	for (int64_t i = 0; i <= n; ++i) f->off[i] += start;
	long long nl = n, before = 0, total = 0;
	MPI_OK(MPI_Exscan(&nl, &before, 1, MPI_LONG_LONG, MPI_SUM, MPI_COMM_WORLD));
	if (g_rank == 0) before = 0;
	MPI_OK(MPI_Allreduce(&nl, &total, 1, MPI_LONG_LONG, MPI_SUM, MPI_COMM_WORLD));
	f->first_index = before; f->n_total = total;
}

/* bases of records [i0, i1) of `f` (global indices) as this rank needs them for the chunk rule over both files: the two
 * files are sliced by bytes, so the mate of a local R1 record may have been scanned by another rank.  Every rank sends each
 * other rank exactly the part of its own records that rank asked for (MPI_Alltoallv): nothing is replicated. */
static int32_t *bases_for_range(const fq_t *f, int64_t i0, int64_t i1)
{
	long long mine[4] = {f->first_index, f->first_index + f->n_local, i0, i1};
	long long *all = malloc(sizeof(long long) * 4 * (size_t)g_size);
	MPI_OK(MPI_Allgather(mine, 4, MPI_LONG_LONG, all, 4, MPI_LONG_LONG, MPI_COMM_WORLD));
	int *scnt = calloc((size_t)g_size, sizeof(int)), *sdsp = calloc((size_t)g_size, sizeof(int));
	int *rcnt = calloc((size_t)g_size, sizeof(int)), *rdsp = calloc((size_t)g_size, sizeof(int));
	for (int r = 0; r < g_size; ++r) {
		/* what rank r wants of my records */
		long long lo = all[4 * r + 2] > mine[0] ? all[4 * r + 2] : mine[0], hi = all[4 * r + 3] < mine[1] ? all[4 * r + 3] : mine[1];
		if (hi > lo) { if (hi - lo > 0x7fffffff) DIE("too many records in one exchange"); scnt[r] = (int)(hi - lo); sdsp[r] = (int)(lo - mine[0]); }
		/* what I want of rank r's records */
		lo = i0 > all[4 * r] ? i0 : all[4 * r]; hi = i1 < all[4 * r + 1] ? i1 : all[4 * r + 1];
		if (hi > lo) { if (hi - lo > 0x7fffffff) DIE("too many records in one exchange"); rcnt[r] = (int)(hi - lo); rdsp[r] = (int)(lo - i0); }
	}
	int32_t *out = malloc(sizeof(int32_t) * (size_t)(i1 - i0 + 1));
	MPI_OK(MPI_Alltoallv(f->bases, scnt, sdsp, MPI_INT, out, rcnt, rdsp, MPI_INT, MPI_COMM_WORLD));
	free(all); free(scnt); free(sdsp); free(rcnt); free(rdsp);
	return out;
}

typedef struct { int64_t first; long long off1, off2; } chunk_t;   /* first record, byte offsets of it in R1 / R2 */

static void bcast_cb(void *buf, size_t bytes, int root, void *user) { MPI_Bcast(buf, (int)bytes, MPI_BYTE, root, *(MPI_Comm *)user); }

/* ---- the chunk loop of one rank, shared by its worker threads ---- */
typedef struct {
	const mem_opt_t *opt;
	bwaidx_t *idx;
	MPI_Win win;
	MPI_File out, f1, f2;
	const long long *tab;        /* per chunk: first record, byte offset in R1, byte offset in R2 (+ one closing row) */
	long long n_chunks;
	int paired, lockstep, trimmed, copy_comment;
	const mem_pestat_t *pes0;    /* -I */
	int fixmate;                 /* -f */
	int format, level;           /* 2 SAM text, 1 / 0 BGZF blocks (-b / -g: src/mainParallel.c:226, 298-299); zlib level (3: src/mainParallel.c:227) */
	int by_chr, n_dest;          /* --by-chr: records go to dest[0 .. n_dest) (contigs, [discordant], unmapped) instead of `out` */
	MPI_File *dest;
	int serialize;               /* MPI_THREAD_SERIALIZED: one thread inside MPI at a time */
	pthread_mutex_t mpi_mu, fetch_mu, write_mu;
	int64_t n_fetched;           /* reads of the chunks this rank has taken so far (trimmed pairs: n_processed) */
	int ordered;                 /* --ordered: a rank's chunks reach the file(s) in the order the rank took them */
	int64_t tickets, now_writing;   /* chunks this rank has taken / the ticket whose turn it is to write */
	pthread_mutex_t turn_mu;
	pthread_cond_t turn_cv;
	double t_start;
} loop_t;

#define MPI_ENTER(L) do { if ((L)->serialize) pthread_mutex_lock(&(L)->mpi_mu); } while (0)
#define MPI_LEAVE(L) do { if ((L)->serialize) pthread_mutex_unlock(&(L)->mpi_mu); } while (0)

/* mi355x_prewarm on a thread of its own (no MPI inside) */
typedef struct { const mem_opt_t *opt; bwaidx_t *idx; int n_reads, len, n_calls, started; double secs; } warm_t;
static void *warm_main(void *arg)
{
	warm_t *w = arg;
	w->secs = mi355x_prewarm(w->opt, w->idx->bwt, w->idx->bns, w->idx->pac, w->n_reads, w->len, w->n_calls);
	return 0;
}
/* bases of the first record of a FASTQ file (0: not one) */
static int first_read_len(const char *path)
{
	FILE *fp = fopen(path, "r");
	if (!fp) return 0;
	char *line = malloc(1 << 20);
	int len = 0;
	if (fgets(line, 1 << 20, fp) && line[0] == '@' && fgets(line, 1 << 20, fp)) {
		len = (int)strcspn(line, "\r\n");
	}
	free(line);
	fclose(fp);
	return len;
}

/* a buffer a worker keeps from chunk to chunk (no allocation, no page faults per chunk) */
static void *grown(void **p, size_t *cap, size_t need)
{
	if (need > *cap) { free(*p); *cap = need + need / 8 + 4096; *p = malloc(*cap); if (!*p) DIE("out of memory"); }
	return *p;
}

/* a piece of a chunk's text to a file through its shared pointer: as it is, or as BGZF blocks (zbuf / czbuf: the worker's buffer).
 * A chunk's text goes out in one piece per GiB: the pieces of one chunk must not be separated by another worker's text (the shared
 * file pointer orders the ranks' writes, write_mu the workers' of one rank). */
static void write_text(loop_t *L, MPI_File fh, const char *text, size_t len, void **zbuf, size_t *czbuf)
{
	const char *out = text;
	size_t n = len;
	if (L->format != 2) {
		const size_t cap = mi355x_bgzf_bound(len);
		uint8_t *z = grown(zbuf, czbuf, cap);
		n = mi355x_bgzf_compress(text, len, L->level, z, cap);
		if (len && !n) DIE("BGZF: %zu bytes of text do not fit their buffer", len);
		out = (const char *)z;
	}
	const int many = n > (1u << 30);
	if (many) pthread_mutex_lock(&L->write_mu);
	MPI_ENTER(L);
	for (size_t w = 0; w < n; ) {
		int piece = n - w > (1u << 30) ? (1 << 30) : (int)(n - w);
		MPI_Status st;
		MPI_OK(MPI_File_write_shared(fh, out + w, piece, MPI_BYTE, &st));
		w += (size_t)piece;
	}
	MPI_LEAVE(L);
	if (many) pthread_mutex_unlock(&L->write_mu);
}

/* the header as the file's first bytes (rank 0, before the file is opened by everybody): text, or BGZF blocks of it
 * (create_sam_header / create_bam_header, src/parallel_aux.c:1846-2026); `bytes` is what goes into the file, made once for all files */
static void create_with_header(const char *path, const void *bytes, size_t n)
{
	FILE *fp = fopen(path, "w");
	if (!fp) DIE("cannot create %s", path);
	if (n && fwrite(bytes, 1, n, fp) != n) DIE("cannot write %s", path);
	if (fclose(fp) != 0) DIE("cannot write %s", path);
}

static void *chunk_worker(void *arg)
{
	loop_t *L = arg;
	const long long *tab = L->tab;
	const int paired = L->paired;
	void *pb1 = 0, *pb2 = 0, *po1 = 0, *po2 = 0, *pbb = 0, *pseqs = 0, *zbuf = 0;
	size_t cb1 = 0, cb2 = 0, co1 = 0, co2 = 0, cbb = 0, cseqs = 0, csam = 0, czbuf = 0;
	char *sam = 0;
	char **dtext = L->by_chr ? calloc((size_t)L->n_dest, sizeof(char *)) : 0;
	size_t *dlen = L->by_chr ? calloc((size_t)L->n_dest, sizeof(size_t)) : 0;
	const int prof = getenv("MPIBWA_DRV_PROF") != 0;
	for (;;) {
		long long one = 1, c = 0;
		int64_t n_before;
		/* next chunk by fetch-and-add on rank 0's counter (src/mainParallel.c:1112-1119); the rank's own chunks keep their
		 * order, so that n_processed of a trimmed-pair chunk is what the reference's sequential loop would pass */
		pthread_mutex_lock(&L->fetch_mu);
		MPI_ENTER(L);
		MPI_OK(MPI_Win_lock(MPI_LOCK_SHARED, 0, 0, L->win));
		MPI_OK(MPI_Fetch_and_op(&one, &c, MPI_LONG_LONG, 0, 0, MPI_SUM, L->win));
		MPI_OK(MPI_Win_unlock(0, L->win));
		MPI_LEAVE(L);
		if (c >= L->n_chunks) { pthread_mutex_unlock(&L->fetch_mu); break; }
		const int64_t count = tab[3 * (c + 1)] - tab[3 * c];
		const int n = (int)(count * (paired ? 2 : 1));
		n_before = L->n_fetched;
		L->n_fetched += n;
		const int64_t ticket = L->tickets++;
		pthread_mutex_unlock(&L->fetch_mu);
		const double t0 = MPI_Wtime();
		int64_t len1 = tab[3 * (c + 1) + 1] - tab[3 * c + 1], len2 = paired ? tab[3 * (c + 1) + 2] - tab[3 * c + 2] : 0;
		char *buf1 = grown(&pb1, &cb1, (size_t)len1 + 1), *buf2 = paired ? grown(&pb2, &cb2, (size_t)len2 + 1) : 0;
		MPI_ENTER(L);
		read_at(L->f1, tab[3 * c + 1], buf1, len1);
		if (paired) read_at(L->f2, tab[3 * c + 2], buf2, len2);
		MPI_LEAVE(L);
		buf1[len1] = 0;
		if (paired) buf2[len2] = 0;
		const double t1 = MPI_Wtime();
		int64_t *o1 = grown(&po1, &co1, sizeof(int64_t) * (size_t)(count + 1)), *o2 = paired ? grown(&po2, &co2, sizeof(int64_t) * (size_t)(count + 1)) : 0;
		int32_t *bb = grown(&pbb, &cbb, sizeof(int32_t) * (size_t)(count + 1));
		if (mi355x_fastq_scan(buf1, len1, count, o1, bb) != count) DIE("chunk %lld of R1 does not hold %lld records", c, (long long)count);
		if (paired && mi355x_fastq_scan(buf2, len2, count, o2, bb) != count) DIE("chunk %lld of R2 does not hold %lld records", c, (long long)count);
		bseq1_t *seqs = grown(&pseqs, &cseqs, sizeof(bseq1_t) * (size_t)n);
		memset(seqs, 0, sizeof(bseq1_t) * (size_t)n);
		if (mi355x_fastq_fill(buf1, o1, buf2, o2, 0, count, L->copy_comment, L->lockstep, seqs) < 0) DIE("malformed record in chunk %lld", c);
		const double t2 = MPI_Wtime();
		/* n_processed: 0 for single end and equal-size pairs, the reads this rank has done for trimmed pairs (src/mainParallel.c:1314, 2355-2357, 3093);
		 * mpiBWAByChr passes 0 in all three modes (src/mainParallelByChromosome.c:1249, 2502, 3407) */
		mem_process_seqs(L->opt, L->idx->bwt, L->idx->bns, L->idx->pac, L->trimmed && !L->by_chr ? n_before : 0, n, seqs, L->pes0);
		const double t3 = MPI_Wtime();
		/* -f: the pairs' lines get their mates' fields and tags (call_fixmate, src/mainParallel.c:1321-1355) */
		if (L->fixmate && paired) {
			const int64_t r = mi355x_fixmate(seqs, n, L->idx->bns);
			if (r < 0) DIE("fixmate: the SAM text of read %lld of chunk %lld is not a pair's", (long long)(-r - 1), c);
		}
		const double t3b = MPI_Wtime();
		const size_t sam_len = mi355x_collect_sam_into(seqs, n, &sam, &csam);
		const double t4 = MPI_Wtime();
		/* Chunks are written as their workers finish them: with --in-flight > 1 the records of a file are not in input order (the
		 * reference's are not across ranks either).  --ordered: a worker waits until the chunks this rank took before its own are
		 * written — one rank then writes the file the reference's blocking loop writes, whatever the number of chunks in flight. */
		if (L->ordered) {
			pthread_mutex_lock(&L->turn_mu);
			while (L->now_writing != ticket) pthread_cond_wait(&L->turn_cv, &L->turn_mu);
			pthread_mutex_unlock(&L->turn_mu);
		}
		if (!L->by_chr) write_text(L, L->out, sam, sam_len, &zbuf, &czbuf);
		else {   /* every record to the file of its contig, pairs on two contigs to "discordant" as well, RNAME '*' to "unmapped" */
			const int64_t r = mi355x_route_by_chr(sam, sam_len, L->idx->bns, L->n_dest == L->idx->bns->n_seqs + 2, dtext, dlen);
			if (r < 0) DIE("chunk %lld: a SAM line without RNAME at byte %lld", c, (long long)(-r - 1));
			for (int d = 0; d < L->n_dest; ++d)
				if (dtext[d]) { write_text(L, L->dest[d], dtext[d], dlen[d], &zbuf, &czbuf); free(dtext[d]); dtext[d] = 0; }
		}
		if (L->ordered) {
			pthread_mutex_lock(&L->turn_mu);
			++L->now_writing;
			pthread_cond_broadcast(&L->turn_cv);
			pthread_mutex_unlock(&L->turn_mu);
		}
		if (prof)
			fprintf(stderr, "[mpibwa_gpu] chunk %lld done at %.3f: read %.0f  scan+fill %.0f  align %.0f  fixmate %.0f  collect %.0f  write %.0f ms\n", c, MPI_Wtime() - L->t_start,
			        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3b - t3) * 1e3, (t4 - t3b) * 1e3, (MPI_Wtime() - t4) * 1e3);
	}
	free(sam); free(pseqs); free(po1); free(po2); free(pbb); free(pb1); free(pb2); free(zbuf); free(dtext); free(dlen);
	return 0;
}

int main(int argc, char **argv)
{
	int provided = MPI_THREAD_SINGLE;
	MPI_Init_thread(&argc, &argv, MPI_THREAD_MULTIPLE, &provided);
	MPI_Comm_rank(MPI_COMM_WORLD, &g_rank);
	MPI_Comm_size(MPI_COMM_WORLD, &g_size);
	int n_threads = 0, copy_comment = 0, dry = 0, n_workers = 6, prewarm = 1;
	int dofixmate = 0, write_format = 2, compression_level = 3, by_chr = 0;   /* src/mainParallel.c:223-227 */
	int ordered = 0;
	int scale_a = 0, set_b = 0, set_T = 0, set_U = 0, set_d = 0, set_O = 0, set_E = 0, set_L = 0, set_k = 0, set_r = 0, set_W = 0;
	const char *mode = 0;   /* -x: a preset of the options the user did not set (src/mainParallel.c:294, 398-428) */
	int64_t K = 0;
	const char *out_path = 0, *pos[4];
	int n_pos = 0;
	if (argc < 2 || strcmp(argv[1], "mem") != 0) {
		if (g_rank == 0) fprintf(stderr, "usage: mpiexec -n N %s mem [bwa mem options] [-f] [-g | -b] [--by-chr] [--ordered] [-K bases] [--in-flight chunks] [--no-prewarm] [--dry-run] -o OUT PREFIX R1.fastq [R2.fastq]\n", argv[0]);
		MPI_Finalize();
		return 1;
	}
	/* the reference's `mem` options (src/mainParallel.c:311-398), same letters and meanings; plus --in-flight and --dry-run */
	mem_opt_t *opt = mem_opt_init();
	int ignore_alt = 0;
	char *rg_line = 0, *hdr_line = 0;
	mem_pestat_t pes[4], *pes0 = 0;
	memset(pes, 0, sizeof pes);
	for (int i = 2; i < argc; ++i) {
		const char *a = argv[i];
		if (!strcmp(a, "--dry-run")) { dry = 1; continue; }
		if (!strcmp(a, "--no-prewarm")) { prewarm = 0; continue; }
		if (!strcmp(a, "--by-chr")) { by_chr = 1; continue; }
		if (!strcmp(a, "--ordered")) { ordered = 1; continue; }
		if (!strcmp(a, "--level") && i + 1 < argc) { compression_level = atoi(argv[++i]); continue; }
		if (!strcmp(a, "--in-flight") && i + 1 < argc) { n_workers = atoi(argv[++i]); continue; }
		if (a[0] != '-' || !a[1]) { if (n_pos < 3) pos[n_pos++] = a; continue; }
		if (a[2]) DIE("unknown option %s (options take their value as the next argument)", a);
		const char c = a[1];
		if (strchr("PapMSYV5qjCfbg", c)) {   /* flags */
			if (c == 'P') opt->flag |= MEM_F_NOPAIRING;
			else if (c == 'a') opt->flag |= MEM_F_ALL;
			else if (c == 'p') DIE("-p (interleaved pairs in one file) is not supported: give R1 and R2");
			else if (c == 'M') opt->flag |= MEM_F_NO_MULTI;
			else if (c == 'S') opt->flag |= MEM_F_NO_RESCUE;
			else if (c == 'Y') opt->flag |= MEM_F_SOFTCLIP;
			else if (c == 'V') opt->flag |= MEM_F_REF_HDR;
			else if (c == '5') opt->flag |= MEM_F_PRIMARY5 | MEM_F_KEEP_SUPP_MAPQ;
			else if (c == 'q') opt->flag |= MEM_F_KEEP_SUPP_MAPQ;
			else if (c == 'j') ignore_alt = 1;
			else if (c == 'C') copy_comment = 1;
			else if (c == 'f') dofixmate = 1;          /* src/mainParallel.c:395 */
			else if (c == 'b') write_format = 1;       /* :298 */
			else if (c == 'g') write_format = 0;       /* :299 */
			continue;
		}
		if (!strchr("kwABTUtcdvrDmsGNWyKXhQOELRHIox", c)) DIE("unknown or unsupported option %s", a);
		if (i + 1 >= argc) DIE("option %s needs a value", a);
		char *v = argv[++i], *e = 0;
		if (c == 'k') { opt->min_seed_len = atoi(v); set_k = 1; }
		else if (c == 'x') mode = v;
		else if (c == 'w') opt->w = atoi(v);
		else if (c == 'A') { opt->a = atoi(v); scale_a = 1; }
		else if (c == 'B') { opt->b = atoi(v); set_b = 1; }
		else if (c == 'T') { opt->T = atoi(v); set_T = 1; }
		else if (c == 'U') { opt->pen_unpaired = atoi(v); set_U = 1; }
		else if (c == 't') n_threads = atoi(v);
		else if (c == 'c') opt->max_occ = atoi(v);
		else if (c == 'd') { opt->zdrop = atoi(v); set_d = 1; }
		else if (c == 'v') bwa_verbose = atoi(v);
		else if (c == 'r') { opt->split_factor = (float)atof(v); set_r = 1; }
		else if (c == 'D') opt->drop_ratio = (float)atof(v);
		else if (c == 'm') opt->max_matesw = atoi(v);
		else if (c == 's') opt->split_width = atoi(v);
		else if (c == 'G') opt->max_chain_gap = atoi(v);
		else if (c == 'N') opt->max_chain_extend = atoi(v);
		else if (c == 'W') { opt->min_chain_weight = atoi(v); set_W = 1; }
		else if (c == 'y') opt->max_mem_intv = (uint64_t)atol(v);
		else if (c == 'K') K = atoll(v);
		else if (c == 'X') opt->mask_level = (float)atof(v);
		else if (c == 'h') {
			opt->max_XA_hits = opt->max_XA_hits_alt = (int)strtol(v, &e, 10);
			if (*e != 0 && ispunct((unsigned char)*e) && isdigit((unsigned char)e[1])) opt->max_XA_hits_alt = (int)strtol(e + 1, &e, 10);
		} else if (c == 'Q') {
			opt->mapQ_coef_len = (float)atoi(v);
			opt->mapQ_coef_fac = opt->mapQ_coef_len > 0 ? (int)log(opt->mapQ_coef_len) : 0;
		} else if (c == 'O') {
			opt->o_del = opt->o_ins = (int)strtol(v, &e, 10); set_O = 1;
			if (*e != 0 && ispunct((unsigned char)*e) && isdigit((unsigned char)e[1])) opt->o_ins = (int)strtol(e + 1, &e, 10);
		} else if (c == 'E') {
			opt->e_del = opt->e_ins = (int)strtol(v, &e, 10); set_E = 1;
			if (*e != 0 && ispunct((unsigned char)*e) && isdigit((unsigned char)e[1])) opt->e_ins = (int)strtol(e + 1, &e, 10);
		} else if (c == 'L') {
			opt->pen_clip5 = opt->pen_clip3 = (int)strtol(v, &e, 10); set_L = 1;
			if (*e != 0 && ispunct((unsigned char)*e) && isdigit((unsigned char)e[1])) opt->pen_clip3 = (int)strtol(e + 1, &e, 10);
		} else if (c == 'R') {
			if ((rg_line = bwa_set_rg(v)) == 0) DIE("malformed read group line %s", v);
		} else if (c == 'H') {
			if (v[0] != '@') {   /* a file of header lines */
				FILE *fp = fopen(v, "r");
				if (!fp) DIE("cannot read the header file %s", v);
				char *buf = calloc(1, 0x10000);
				while (fgets(buf, 0xffff, fp)) {
					size_t l = strlen(buf);
					if (l && buf[l - 1] == '\n') buf[l - 1] = 0;
					hdr_line = bwa_insert_header(buf, hdr_line);
				}
				free(buf);
				fclose(fp);
			} else hdr_line = bwa_insert_header(v, hdr_line);
		} else if (c == 'I') {   /* the insert size distribution given by the user (src/mainParallel.c:377-396) */
			pes0 = pes;
			pes[0].failed = pes[2].failed = pes[3].failed = 1;
			pes[1].failed = 0;
			pes[1].avg = strtod(v, &e);
			pes[1].std = pes[1].avg * .1;
			if (*e != 0 && ispunct((unsigned char)*e) && isdigit((unsigned char)e[1])) pes[1].std = strtod(e + 1, &e);
			pes[1].high = (int)(pes[1].avg + 4. * pes[1].std + .499);
			pes[1].low = (int)(pes[1].avg - 4. * pes[1].std + .499);
			if (pes[1].low < 1) pes[1].low = 1;
			if (*e != 0 && ispunct((unsigned char)*e) && isdigit((unsigned char)e[1])) pes[1].high = (int)(strtod(e + 1, &e) + .499);
			if (*e != 0 && ispunct((unsigned char)*e) && isdigit((unsigned char)e[1])) pes[1].low = (int)(strtod(e + 1, &e) + .499);
		} else if (c == 'o') out_path = v;
	}
	if (mode) {   /* the presets of -x for what the user left alone (src/mainParallel.c:398-428); -A does not scale anything then */
		if (!strcmp(mode, "intractg")) {
			if (!set_O) opt->o_del = opt->o_ins = 16;
			if (!set_b) opt->b = 9;
			if (!set_L) opt->pen_clip5 = opt->pen_clip3 = 5;
		} else if (!strcmp(mode, "pacbio") || !strcmp(mode, "pbref") || !strcmp(mode, "ont2d")) {   /* (reads beyond 8 998 bp are refused by the library) */
			const int ont = !strcmp(mode, "ont2d");
			if (!set_O) opt->o_del = opt->o_ins = 1;
			if (!set_E) opt->e_del = opt->e_ins = 1;
			if (!set_b) opt->b = 1;
			if (!set_r) opt->split_factor = 10.f;
			if (!set_W) opt->min_chain_weight = ont ? 20 : 40;
			if (!set_k) opt->min_seed_len = ont ? 14 : 17;
			if (!set_L) opt->pen_clip5 = opt->pen_clip3 = 0;
		} else DIE("unknown read type '%s' (-x intractg | pacbio | pbref | ont2d)", mode);
	} else if (scale_a && opt->a != 1) {   /* -A scales the penalties the user did not set (src/mainParallel.c:430-440) */
		if (!set_b) opt->b *= opt->a;
		if (!set_T) opt->T *= opt->a;
		if (!set_O) { opt->o_del *= opt->a; opt->o_ins *= opt->a; }
		if (!set_E) { opt->e_del *= opt->a; opt->e_ins *= opt->a; }
		if (!set_d) opt->zdrop *= opt->a;
		if (!set_L) { opt->pen_clip5 *= opt->a; opt->pen_clip3 *= opt->a; }
		if (!set_U) opt->pen_unpaired *= opt->a;
	}
	bwa_fill_scmat(opt->a, opt->b, opt->mat);
	if (n_pos < 2 || (!out_path && !dry)) DIE("need -o OUT PREFIX R1 [R2]");
	const char *prefix = pos[0];
	const int paired = n_pos == 3;
	if (n_threads > 0) opt->n_threads = n_threads;
	if (paired) opt->flag |= MEM_F_PE;
	if (K <= 0) K = (int64_t)opt->chunk_size * opt->n_threads;   /* src/mainParallel.c:635 */
	if (provided < MPI_THREAD_SERIALIZED) n_workers = 1;
	if (n_workers < 1) n_workers = 1;
	if (n_workers > mi355x_max_calls()) n_workers = mi355x_max_calls();   /* what the library runs side by side (fewer when their work buffers do not fit) */
	if (n_workers > MAX_WORKERS) n_workers = MAX_WORKERS;

	/* The index goes first: while this thread reads the FASTQ offsets below, another one pays the first-use cost of the call
	 * contexts (mi355x_prewarm: work buffers, streams, thread pool, code objects) on reads sampled from the reference, so that the
	 * chunk loop starts at its steady rate.  --no-prewarm: the first chunk of every worker pays it, as before. */
	bwaidx_t *idx = 0, idx_map;
	pthread_t warm_th;
	warm_t warm;
	memset(&warm, 0, sizeof warm);
	if (!dry) {
		/* ---- index: host copy from the bwa files, device copy through mi355x_init ---- */
		/* PREFIX.map present (mpiBWAIdx / mi355x_write_map): the ranks of a node share ONE host copy of the index — a private
		 * file mapping whose bulk (BWT, SA, pac: read-only) stays in the page cache; only the pages bwa_mem2idx writes pointers
		 * into become private (the reference maps the image into an MPI shared window, src/parallel_aux.c:1745-1838).
		 * Otherwise every rank reads the five bwa files. */
		{
			char *mp = malloc(strlen(prefix) + 8);
			sprintf(mp, "%s.map", prefix);
			int fd = open(mp, O_RDONLY);
			if (fd >= 0) {
				struct stat sb;
				if (fstat(fd, &sb) != 0) DIE("cannot stat %s", mp);
				void *img = mmap(0, (size_t)sb.st_size, PROT_READ | PROT_WRITE, MAP_PRIVATE, fd, 0);
				if (img == MAP_FAILED) DIE("cannot map %s", mp);
				close(fd);
				memset(&idx_map, 0, sizeof idx_map);
				if (bwa_mem2idx((int64_t)sb.st_size, (uint8_t *)img, &idx_map) != 0) DIE("%s is not an index image", mp);
				idx = &idx_map;
				if (g_rank == 0) fprintf(stderr, "[mpibwa_gpu] index attached from %s (%.2f GB, shared by the ranks of a node)\n", mp, sb.st_size / 1e9);
			} else idx = bwa_idx_load_from_disk(prefix, 7);
			free(mp);
		}
		if (!idx) DIE("cannot load the index %s", prefix);
		MPI_Comm node;
		MPI_OK(MPI_Comm_split_type(MPI_COMM_WORLD, MPI_COMM_TYPE_SHARED, g_rank, MPI_INFO_NULL, &node));
		int local_rank, local_size;
		MPI_Comm_rank(node, &local_rank); MPI_Comm_size(node, &local_size);
		const int n_dev = mi355x_device_count();
		if (n_dev <= 0) DIE("no MI355X visible to rank %d", g_rank);
		if (local_size <= n_dev && local_size > 1) {   /* one rank per GPU: one H2D on the node, RCCL broadcast to the other GPUs */
			mi355x_comm_t comm = {local_rank, local_size, bcast_cb, &node};
			mi355x_init(local_rank, idx, &comm);
		} else mi355x_init(local_rank % n_dev, idx, 0);
		if (ignore_alt)   /* -j (src/parallel_aux.c:1831-1833) */
			for (int i = 0; i < idx->bns->n_seqs; ++i) idx->bns->anns[i].is_alt = 0;
		if (g_rank != 0 && bwa_verbose > 1) bwa_verbose = 1;
		if (prewarm) {
			struct stat sb;
			const int len = first_read_len(pos[1]);
			if (len > 0 && stat(pos[1], &sb) == 0) {
				/* reads of one chunk (both ends), and of this rank's share of the input: a record is at least 2 * len + 6 bytes */
				const int64_t per_chunk = K / len + (paired ? 2 : 1), in_file = (int64_t)sb.st_size / (2 * len + 6) * (paired ? 2 : 1);
				const int64_t mine = in_file / g_size;
				warm.n_reads = (int)(per_chunk < mine ? per_chunk : mine);
				warm.n_calls = (int)((mine + per_chunk - 1) / per_chunk);
				if (warm.n_calls > n_workers) warm.n_calls = n_workers;
				warm.len = len; warm.opt = opt; warm.idx = idx;
				if (warm.n_reads >= 20000 && warm.n_calls >= 1 && pthread_create(&warm_th, 0, warm_main, &warm) == 0) warm.started = 1;
			}
		}
	}

	/* ---- partition ---- */
	fq_t f1, f2;
	fq_open(&f1, pos[1]);
	if (paired) {
		fq_open(&f2, pos[2]);
		if (f1.n_total != f2.n_total) DIE("the two FASTQ files hold %lld and %lld reads", (long long)f1.n_total, (long long)f2.n_total);
	}
	/* the reference takes the equal-size branch when both files have the same byte size (src/mainParallel.c:703-728) */
	const int lockstep = paired && f1.size == f2.size, trimmed = paired && !lockstep;
	const int64_t maxsiz = paired && lockstep ? K / 2 : K;
	int32_t *b2 = trimmed ? bases_for_range(&f2, f1.first_index, f1.first_index + f1.n_local) : 0;
	/* the running count travels down the ranks; every rank closes chunks inside its record range */
	long long carry[2] = {0, 0};   /* bases in the open chunk, 1 if a chunk is open */
	if (g_rank > 0) MPI_OK(MPI_Recv(carry, 2, MPI_LONG_LONG, g_rank - 1, 7, MPI_COMM_WORLD, MPI_STATUS_IGNORE));
	int64_t cap = f1.n_local + 2, n_mine = 0;
	int64_t *mine = malloc(sizeof(int64_t) * (size_t)cap);   /* global index of the first record of the chunks that START here */
	{
		long long counter = carry[0];
		int open = (int)carry[1];
		for (int64_t i = 0; i < f1.n_local; ++i) {
			if (!open) { mine[n_mine++] = f1.first_index + i; open = 1; }
			counter += f1.bases[i];
			if (b2) counter += b2[i];
			if (counter > maxsiz) { counter = 0; open = 0; }
		}
		carry[0] = counter; carry[1] = open;
	}
	if (g_rank + 1 < g_size) MPI_OK(MPI_Send(carry, 2, MPI_LONG_LONG, g_rank + 1, 7, MPI_COMM_WORLD));
	/* replicate the chunk table: first record and its byte offset in R1; R2's offset comes from whoever scanned that record */
	long long nm = n_mine, n_chunks = 0, before = 0;
	MPI_OK(MPI_Allreduce(&nm, &n_chunks, 1, MPI_LONG_LONG, MPI_SUM, MPI_COMM_WORLD));
	MPI_OK(MPI_Exscan(&nm, &before, 1, MPI_LONG_LONG, MPI_SUM, MPI_COMM_WORLD));
	if (g_rank == 0) before = 0;
	long long *tab = calloc((size_t)(n_chunks + 1) * 3, sizeof(long long)), *tab_all = calloc((size_t)(n_chunks + 1) * 3, sizeof(long long));
	for (int64_t k = 0; k < n_mine; ++k) {
		tab[3 * (before + k)] = mine[k];
		tab[3 * (before + k) + 1] = f1.off[mine[k] - f1.first_index];
	}
	MPI_OK(MPI_Allreduce(tab, tab_all, (int)((n_chunks + 1) * 3), MPI_LONG_LONG, MPI_SUM, MPI_COMM_WORLD));
	tab_all[3 * n_chunks] = f1.n_total; tab_all[3 * n_chunks + 1] = (long long)f1.size; tab_all[3 * n_chunks + 2] = paired ? (long long)f2.size : 0;
	if (paired) {   /* offsets of the chunks' first records in R2 */
		memset(tab, 0, sizeof(long long) * (size_t)(n_chunks + 1) * 3);
		for (long long c = 0; c < n_chunks; ++c) {
			long long idx = tab_all[3 * c];
			if (idx >= f2.first_index && idx < f2.first_index + f2.n_local) tab[3 * c + 2] = f2.off[idx - f2.first_index];
		}
		long long *t2 = calloc((size_t)(n_chunks + 1) * 3, sizeof(long long));
		MPI_OK(MPI_Allreduce(tab, t2, (int)((n_chunks + 1) * 3), MPI_LONG_LONG, MPI_SUM, MPI_COMM_WORLD));
		for (long long c = 0; c < n_chunks; ++c) tab_all[3 * c + 2] = t2[3 * c + 2];
		free(t2);
	}
	free(tab);
	if (dry) {   /* the chunk table, for the partition tests */
		if (g_rank == 0) {
			printf("mode %s ranks %d chunks %lld reads %lld maxsiz %lld\n", !paired ? "se" : lockstep ? "pe" : "pe_trim", g_size, n_chunks, (long long)f1.n_total, (long long)maxsiz);
			for (long long c = 0; c <= n_chunks; ++c) printf("chunk %lld first %lld off1 %lld off2 %lld\n", c, tab_all[3 * c], tab_all[3 * c + 1], tab_all[3 * c + 2]);
		}
		MPI_Finalize();
		return 0;
	}


	if (warm.started) {
		pthread_join(warm_th, 0);
		if (bwa_verbose >= 3) fprintf(stderr, "[mpibwa_gpu] rank %d: %d call contexts warmed on %d sampled reads each in %.2f s, beside the FASTQ scan\n", g_rank, warm.n_calls, warm.n_reads, warm.secs);
	}
	/* ---- output: rank 0 writes the header, then everybody appends through the shared file pointer ---- */
	/* the reference's header (src/parallel_aux.c:1846-1915): @SQ lines, the -H lines, the read group, the program line */
	char *hdr = 0;
	size_t hdr_len = 0;
	{
		FILE *hp = open_memstream(&hdr, &hdr_len);
		if (!hp) DIE("out of memory");
		for (int i = 0; i < idx->bns->n_seqs; ++i) fprintf(hp, "@SQ\tSN:%s\tLN:%d\n", idx->bns->anns[i].name, idx->bns->anns[i].len);
		if (hdr_line) fprintf(hp, "%s\n", hdr_line);
		if (rg_line) fprintf(hp, "%s\n", rg_line);
		fprintf(hp, "@PG\tID:mpibwa_gpu\tPN:mpibwa_gpu\tVN:r4\tCL:%s", argv[0]);
		for (int i = 1; i < argc; ++i) fprintf(hp, " %s", argv[i]);
		fputc('\n', hp);
		fclose(hp);
	}
	MPI_File out = MPI_FILE_NULL, *dest = 0;
	int n_dest = 0;
	/* what a file starts with: the header text, or its BGZF blocks (made once, however many files there are) */
	const void *hdr_bytes = hdr;
	size_t hdr_n = hdr_len;
	uint8_t *hdr_z = 0;
	if (write_format != 2 && g_rank == 0) {
		const size_t cap = mi355x_bgzf_bound(hdr_len);
		hdr_z = malloc(cap);
		hdr_n = mi355x_bgzf_compress(hdr, hdr_len, compression_level, hdr_z, cap);
		hdr_bytes = hdr_z;
	}
	if (!by_chr) {
		if (g_rank == 0) create_with_header(out_path, hdr_bytes, hdr_n);
		MPI_Barrier(MPI_COMM_WORLD);
		MPI_OK(MPI_File_open(MPI_COMM_WORLD, (char *)out_path, MPI_MODE_WRONLY | MPI_MODE_APPEND, MPI_INFO_NULL, &out));
	} else {
		/* mpiBWAByChr (src/mainParallelByChromosome.c:983-1047): <directory of OUT>/<contig>.sam for every contig, discordant.sam for
		 * pairs without -f, unmapped.sam; .bam / .gz with -b / -g.  The contig files start with the header; as text the other two do
		 * not (create_sam_header_by_chr_file covers the contigs only, src/parallel_aux.c:2650-2723), compressed they do. */
		struct stat sb;
		char *dir = strdup(out_path);
		if (!(stat(out_path, &sb) == 0 && S_ISDIR(sb.st_mode))) {
			char *slash = strrchr(dir, '/');
			if (slash) *slash = 0; else strcpy(dir, ".");
		}
		const int disc = paired && !dofixmate;
		n_dest = idx->bns->n_seqs + 1 + disc;
		dest = malloc(sizeof(MPI_File) * (size_t)n_dest);
		const char *ext = write_format == 2 ? "sam" : write_format == 1 ? "bam" : "gz";
		char **paths = malloc(sizeof(char *) * (size_t)n_dest);
		for (int d = 0; d < n_dest; ++d) {
			const char *nm = d < idx->bns->n_seqs ? idx->bns->anns[d].name : (disc && d == idx->bns->n_seqs) ? "discordant" : "unmapped";
			paths[d] = malloc(strlen(dir) + strlen(nm) + 8);
			sprintf(paths[d], "%s/%s.%s", dir, nm, ext);
			if (g_rank == 0) create_with_header(paths[d], hdr_bytes, d < idx->bns->n_seqs || write_format != 2 ? hdr_n : 0);
		}
		MPI_Barrier(MPI_COMM_WORLD);
		for (int d = 0; d < n_dest; ++d) {
			MPI_OK(MPI_File_open(MPI_COMM_WORLD, paths[d], MPI_MODE_WRONLY | MPI_MODE_APPEND, MPI_INFO_NULL, &dest[d]));
			free(paths[d]);
		}
		free(paths);
		free(dir);
	}
	free(hdr_z);

	/* ---- the chunk loop: next chunk by fetch-and-add on rank 0's counter ---- */
	long long *counter_mem = 0;
	MPI_Win win;
	MPI_OK(MPI_Win_allocate(g_rank == 0 ? sizeof(long long) : 0, sizeof(long long), MPI_INFO_NULL, MPI_COMM_WORLD, &counter_mem, &win));
	if (g_rank == 0) *counter_mem = 0;
	MPI_Barrier(MPI_COMM_WORLD);
	const double t_loop = MPI_Wtime();
	loop_t L;
	memset(&L, 0, sizeof L);
	L.opt = opt; L.idx = idx; L.win = win; L.out = out; L.f1 = f1.fh; L.f2 = paired ? f2.fh : MPI_FILE_NULL;
	L.tab = tab_all; L.n_chunks = n_chunks; L.paired = paired; L.lockstep = lockstep; L.trimmed = trimmed; L.copy_comment = copy_comment; L.pes0 = pes0;
	L.fixmate = dofixmate; L.format = write_format; L.level = compression_level; L.by_chr = by_chr; L.n_dest = n_dest; L.dest = dest;
	L.serialize = provided < MPI_THREAD_MULTIPLE;
	L.t_start = t_loop;
	pthread_mutex_init(&L.mpi_mu, 0);
	pthread_mutex_init(&L.fetch_mu, 0);
	L.ordered = ordered;
	pthread_mutex_init(&L.turn_mu, 0);
	pthread_cond_init(&L.turn_cv, 0);
	pthread_mutex_init(&L.write_mu, 0);
	pthread_t th[MAX_WORKERS];
	for (int w = 1; w < n_workers; ++w)
		if (pthread_create(&th[w], 0, chunk_worker, &L) != 0) DIE("cannot start worker thread %d", w);
	chunk_worker(&L);
	for (int w = 1; w < n_workers; ++w) pthread_join(th[w], 0);
	MPI_Barrier(MPI_COMM_WORLD);
	if (g_rank == 0) {   /* the phase the reference brackets with MPI_Wtime (src/mainParallel.c:1238-1240, 1316-1319), over all ranks */
		const double dt = MPI_Wtime() - t_loop;
		const long long n_reads = (long long)f1.n_total * (paired ? 2 : 1);
		fprintf(stderr, "[mpibwa_gpu] chunk loop: %lld reads in %lld chunks, %d rank(s) x %d chunks in flight, %.3f s = %.3f Mreads/s\n", n_reads, n_chunks, g_size,
		        n_workers, dt, dt > 0 ? n_reads / dt * 1e-6 : 0.);
	}
	if (write_format == 1 && g_rank == 0) {   /* the empty block that ends the file the reference calls BAM (src/mainParallel.c:1508-1516) */
		uint8_t eof[28];
		MPI_Status st;
		mi355x_bgzf_eof(eof);
		if (!by_chr) MPI_OK(MPI_File_write_shared(out, eof, 28, MPI_BYTE, &st));
		for (int d = 0; d < n_dest; ++d) MPI_OK(MPI_File_write_shared(dest[d], eof, 28, MPI_BYTE, &st));
	}
	MPI_Win_free(&win);
	if (!by_chr) MPI_File_close(&out);
	for (int d = 0; d < n_dest; ++d) MPI_File_close(&dest[d]);
	free(dest); free(hdr);
	MPI_File_close(&f1.fh);
	if (paired) MPI_File_close(&f2.fh);
	mi355x_finalize();
	MPI_Finalize();
	return 0;
}
