// fastq.cpp — the caller's side of the path: how mpiBWA turns FASTQ text into the chunks and the
// bseq1_t arrays it hands to mem_process_seqs().  No MPI here: the rules are restated for one process, and because the
// reference's chunking is a running count that is carried from rank to rank (src/parallel_aux.c:1553-1561) the chunk
// boundaries are the same for any number of ranks.
//
//   record scan     find_reads_size_and_offsets           src/parallel_aux.c:682-832   (offset, bytes, bases of every record)
//   chunk rule      find_chunks_info / _trim              src/parallel_aux.c:1510-1546, 1082, 1203
//                   a chunk takes reads until its base count EXCEEDS maxsiz (strict >; the read that crosses closes it);
//                   maxsiz = K/2 per file for equal-size pairs (src/mainParallel.c:947), K over R1+R2 for trimmed pairs
//                   (:1874), K for single-end (:2773); K = -K or chunk_size * n_threads (:635)
//   record parsing  src/mainParallel.c:1257-1301 (equal-size pairs: R2 is cut at R1's positions), :2271-2345 (each file
//                   on its own): name up to the first white space, a trailing "/digit" dropped, comment = rest of the line
#include "internal.h"

#include <cctype>
#include <cstring>
#include <thread>
#include <vector>

using namespace mbw;

// Offsets of the record starts (rec_off[n] = end of the last record) and bases per record.
// Returns the number of records, or -(byte position + 1) of the first record that does not start with '@' / has no '+' line.
extern "C" int64_t mi355x_fastq_scan(const char *buf, int64_t len, int64_t cap, int64_t *rec_off, int32_t *rec_bases)
{
	int64_t n = 0, g = 0;
	while (g < len) {
		if (buf[g] != '@') return -(g + 1);
		const int64_t start = g;
		while (g < len && buf[g] != '\n') ++g;          // name
		++g;
		int64_t s0 = g;
		while (g < len && buf[g] != '\n') ++g;          // bases
		const int64_t bases = g - s0;
		++g;
		if (g >= len || buf[g] != '+') return -(start + 1);
		while (g < len && buf[g] != '\n') ++g;          // +
		++g;
		while (g < len && buf[g] != '\n') ++g;          // qualities
		++g;                                            // the file may end without a final newline
		if (n < cap) { rec_off[n] = start; rec_bases[n] = (int32_t)bases; }
		++n;
	}
	if (n <= cap) rec_off[n < cap ? n : cap] = len < g ? len : g;
	return n;
}

// First record of every chunk (chunk_first[c]; chunk_first[n_chunks] = n).  bases2 != NULL: the count runs over both
// files (trimmed pairs).  Returns the number of chunks; the last one is whatever is left (src/parallel_aux.c:1640-1660).
extern "C" int64_t mi355x_fastq_chunks(const int32_t *bases1, const int32_t *bases2, int64_t n, int64_t maxsiz, int64_t cap, int64_t *chunk_first)
{
	int64_t n_chunks = 0, counter = 0;
	bool open = false;
	for (int64_t i = 0; i < n; ++i) {
		if (!open) { if (n_chunks < cap) chunk_first[n_chunks] = i; open = true; }
		counter += bases1[i];
		if (bases2) counter += bases2[i];
		if (counter > maxsiz) { ++n_chunks; counter = 0; open = false; }
	}
	if (open) ++n_chunks;
	if (n_chunks <= cap) chunk_first[n_chunks < cap ? n_chunks : cap] = n;
	return n_chunks;
}

struct RecLines { char *line[4]; int64_t len[4]; };

// NUL-terminate the four lines of the record [p, e) in place
static bool split_record(char *p, char *e, RecLines &r)
{
	if (p >= e || *p != '@') return false;
	int ln = 0;
	r.line[0] = p;
	for (char *q = p; q < e && ln < 4; ++q)
		if (*q == '\n') {
			*q = '\0';
			r.len[ln] = q - r.line[ln];
			if (++ln < 4) r.line[ln] = q + 1;
		}
	if (ln == 3) { r.len[3] = e - r.line[3]; ln = 4; }   // last record of a file that does not end with a newline
	return ln == 4 && r.line[2][0] == '+';
}

// bseq1_t for records [first, first + count) of one file (buf2 == NULL: single-end, seqs[k]) or two (mates interleaved,
// seqs[2k] / seqs[2k+1]).  Strings are NUL-terminated in place.  lockstep != 0 reproduces the equal-size mode, where
// the reference walks both header lines with one loop driven by R1 (src/mainParallel.c:1274-1276): R2's name ends, and its
// comment starts, at the offsets found in R1.  Returns the number of bases, or -(k + 1) for a malformed record k.
extern "C" int64_t mi355x_fastq_fill(char *buf1, const int64_t *off1, char *buf2, const int64_t *off2, int64_t first, int64_t count,
                                     int copy_comment, int lockstep, bseq1_t *seqs)
{
	const int files = buf2 ? 2 : 1;
	// records are independent of each other: a few threads take contiguous ranges (a chunk is ~700 000 records)
	const int n_thr = count >= 65536 ? 8 : 1;
	std::vector<int64_t> t_bases(n_thr, 0), t_bad(n_thr, -1);
	auto range = [&](int tid) {
	const int64_t k_lo = count * tid / n_thr, k_hi = count * (tid + 1) / n_thr;
	int64_t bases = 0;
	for (int64_t k = k_lo; k < k_hi; ++k) {
		int64_t name_end1 = 0, ws1 = -1, comment1 = 0;   // offsets inside R1's header line (ws1 < 0: no white space in it)
		for (int f = 0; f < files; ++f) {
			char *buf = f ? buf2 : buf1;
			const int64_t *off = f ? off2 : off1;
			bseq1_t *s = &seqs[files * k + f];
			RecLines r;
			if (!split_record(buf + off[first + k], buf + off[first + k + 1], r)) { t_bad[tid] = k; return; }
			char *h = r.line[0], *hend = h + r.len[0];
			char *p;
			if (f == 1 && lockstep) {
				char *t = h + name_end1, *c = h + comment1;
				if (t < hend) *t = '\0';
				if (ws1 >= 0 && h + ws1 < hend) h[ws1] = '\0';
				p = c < hend ? c : hend;
			} else {
				p = h;
				while (*p && !isspace((unsigned char)*p)) ++p;
				char *t = p;
				if (p - 2 > h && *(p - 2) == '/' && isdigit((unsigned char)*(p - 1))) { *(p - 2) = '\0'; t = p - 2; }
				ws1 = -1;
				if (*p) { ws1 = p - h; *p++ = '\0'; }
				name_end1 = t - h; comment1 = p - h;
			}
			s->name = h + 1;
			s->comment = copy_comment ? p : 0;
			s->seq = r.line[1];
			s->l_seq = (int)r.len[1];
			s->qual = r.line[3];
			s->sam = 0;
			s->id = 0;
			bases += r.len[1];
		}
	}
	t_bases[tid] = bases;
	};
	std::vector<std::thread> th;
	for (int t = 1; t < n_thr; ++t) th.emplace_back(range, t);
	range(0);
	for (auto &t : th) t.join();
	int64_t bases = 0;
	for (int t = 0; t < n_thr; ++t) {
		if (t_bad[t] >= 0) return -(t_bad[t] + 1);   // the first malformed record (ranges are in record order)
		bases += t_bases[t];
	}
	return bases;
}
