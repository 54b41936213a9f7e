// sampost.cpp — what mpiBWA does with a chunk's SAM text between mem_process_seqs() and the file (the caller's
// side): `-f` fixmate, `-g` / `-b` BGZF output, and mpiBWAByChr's routing of records to per-contig files.  Host code, no MPI, no GPU:
// the text is the product of the hot path, these passes only re-arrange it.
//
//   fixmate        fixmate()                         src/fixmate.c:601-827   (per pair; parser :160-298, writers :374-599)
//                  call_fixmate                      src/parallel_aux.c:2164-2206   (the loop over a chunk's pairs)
//   BGZF           deflate_block                     src/bgzf.c:245-330      (one gzip member per block, 'BC' extra field)
//                  compress_and_write_bgzf_thread    src/parallel_aux.c:2941-3073   (-g: whole records per block)
//                  compress_and_write_bam_thread     src/parallel_aux.c:3075-3176   (-b: the same text; EOF block src/mainParallel.c:1509-1516)
//   by chromosome  the routing loops                 src/mainParallelByChromosome.c:1340-1455 (pairs), :3437-3486 (single end)
//                  getChr                            src/parallel_aux.c:2625-2648
//
// The reference's fixmate is restated record for record: the same three passes over the lines of a pair, the same fields and tags
// (MQ, MC, ms), the same order of the lines in each mate's text.  Where the reference's behaviour is undefined (it indexes the
// contig table with -1 when a supplementary line has RNEXT '*', and uses the tag text as a printf format) this file prints '*'
// and copies the tags literally.  The reference's compressed writers drop the last read of every thread's slice
// (`end_index = ... - 1`, src/parallel_aux.c:2953-2956) — here every record is written: the decompressed stream is the SAM text.
#include "internal.h"

#include <zlib.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

using namespace mbw;

namespace {

enum : unsigned { F_PAIRED = 1, F_UNMAP = 4, F_MUNMAP = 8, F_REVERSE = 16, F_MREVERSE = 32, F_READ1 = 64, F_READ2 = 128, F_SECONDARY = 256, F_SUPP = 2048 };
constexpr int MIN_QUALITY = 15;   // MD_MIN_QUALITY (src/fixmate.h)

template <class F> void run_parallel(int64_t n, int64_t grain, F f)
{
	const int64_t n_blocks = (n + grain - 1) / grain;
	int n_thr = std::min(mi355x_host_cpus(), 32);
	if (const char *e = getenv("MPIBWA_SAMPOST_THREADS")) n_thr = std::max(1, std::min(atoi(e), 64));   // (the tests run the passes on 1, 3, ... threads)
	n_thr = (int)std::min<int64_t>(n_thr, n_blocks);
	if (n_thr <= 1) { for (int64_t b = 0; b < n_blocks; ++b) f(b * grain, std::min(n, (b + 1) * grain)); return; }
	std::atomic<int64_t> next(0);
	std::vector<std::thread> th;
	auto body = [&] { for (int64_t b; (b = next.fetch_add(1)) < n_blocks;) f(b * grain, std::min(n, (b + 1) * grain)); };
	for (int t = 1; t < n_thr; ++t) th.emplace_back(body);
	body();
	for (auto &t : th) t.join();
}

// contig name -> index, the first of equal names (the reference's linear search stops at the first match, src/fixmate.c:189-196)
struct ContigIndex {
	std::unordered_map<std::string_view, int> by_name;
	explicit ContigIndex(const bntseq_t *bns)
	{
		by_name.reserve((size_t)bns->n_seqs * 2);
		for (int i = 0; i < bns->n_seqs; ++i) by_name.emplace(std::string_view(bns->anns[i].name), i);
	}
	int find(std::string_view name) const
	{
		auto it = by_name.find(name);
		return it == by_name.end() ? -1 : it->second;
	}
};

// one SAM line of a pair, fields as views into the mate's text (readInfo, src/fixmate.h)
struct Rec {
	unsigned flag = 0;
	int tid = -1, mtid = -1;
	uint64_t pos = 0, mpos = 0;
	uint32_t mapq = 0, mmapq = 0, tlen = 0, score = 0, mscore = 0;
	std::string_view cigar, mcigar, seq, qual, aux;   // aux: everything behind the quality field, with the line's '\n'
	const char *name = nullptr;
	bool done = false;
};

std::string_view next_field(const char *&p, const char *end)
{
	const char *t = (const char *)memchr(p, '\t', (size_t)(end - p));
	std::string_view f(p, (size_t)((t ? t : end) - p));
	p = t ? t + 1 : end;
	return f;
}

int64_t to_int(std::string_view f)   // atoi / atol on a field (digits with an optional sign; anything else ends the number)
{
	size_t i = 0;
	bool neg = false;
	if (i < f.size() && (f[i] == '-' || f[i] == '+')) neg = f[i++] == '-';
	uint64_t v = 0;   // (unsigned: a field of twenty digits wraps around instead of overflowing a signed number)
	for (; i < f.size() && f[i] >= '0' && f[i] <= '9'; ++i) v = v * 10 + (uint64_t)(f[i] - '0');
	return (int64_t)(neg ? 0 - v : v);
}

// readParsing (src/fixmate.c:160-298): [line, line_end) holds one record without its '\n'; `tail_end` is behind the '\n'
bool parse_line(const char *line, const char *line_end, const char *tail_end, const ContigIndex &ci, Rec &r)
{
	const char *p = line;
	next_field(p, line_end);                                   // QNAME: the caller's bseq1_t::name is printed instead
	r.flag = (unsigned)to_int(next_field(p, line_end));
	if (r.flag == 0) return false;                             // (the reference asserts a non-zero flag)
	r.tid = ci.find(next_field(p, line_end));
	const std::string_view pos = next_field(p, line_end);
	r.pos = pos == "*" ? (uint64_t)-1 : (uint64_t)to_int(pos);
	r.mapq = (uint32_t)to_int(next_field(p, line_end));
	r.cigar = next_field(p, line_end);
	const std::string_view rnext = next_field(p, line_end);
	r.mtid = rnext == "=" ? r.tid : ci.find(rnext);
	const std::string_view mpos = next_field(p, line_end);
	r.mpos = mpos == "*" ? (uint64_t)-1 : (uint64_t)to_int(mpos);
	r.tlen = (uint32_t)to_int(next_field(p, line_end));
	r.seq = next_field(p, line_end);
	r.qual = next_field(p, line_end);
	for (const char c : r.qual)
		if (c - 33 >= MIN_QUALITY) r.score += (uint32_t)(c - 33);
	r.aux = std::string_view(p, (size_t)(tail_end - p));       // the tags and the newline ("\n" alone when there are none)
	return true;
}

struct Out {
	std::string s;
	void str(std::string_view v) { s.append(v.data(), v.size()); }
	void tab() { s.push_back('\t'); }
	void u64(uint64_t v) { char b[24]; s.append(b, (size_t)snprintf(b, sizeof b, "%llu", (unsigned long long)v)); }
	void i32(int v) { char b[16]; s.append(b, (size_t)snprintf(b, sizeof b, "%d", v)); }
	void u32(uint32_t v) { char b[16]; s.append(b, (size_t)snprintf(b, sizeof b, "%u", v)); }
};

const char *contig(const bntseq_t *bns, int tid) { return tid >= 0 && tid < bns->n_seqs ? bns->anns[tid].name : "*"; }

// the eleven mandatory fields, as every writer of src/fixmate.c prints them (name, flag, RNAME, POS, MAPQ, CIGAR, RNEXT, PNEXT, TLEN, SEQ, QUAL)
void put_core(Out &o, const Rec &r, const char *chr, const char *mchr)
{
	o.str(r.name); o.tab(); o.i32((int)r.flag); o.tab(); o.str(chr); o.tab(); o.u64(r.pos); o.tab(); o.i32((int)r.mapq); o.tab();
	o.str(r.cigar); o.tab(); o.str(mchr); o.tab(); o.u64(r.mpos); o.tab(); o.i32((int)r.tlen); o.tab(); o.str(r.seq); o.tab(); o.str(r.qual); o.tab();
}

// sam_write_unmapped_and_munmapped (src/fixmate.c:447-495)
void write_both_unmapped(Out &o, const Rec &r, const bntseq_t *bns)
{
	const char *chr, *mchr = nullptr;
	if (r.tid == -1 && r.mtid == -1) { chr = "*"; mchr = "*"; }
	else chr = contig(bns, r.tid);
	if (r.tid != -1 && r.tid == r.mtid) mchr = "=";
	if (r.mtid != -1) mchr = contig(bns, r.mtid);
	if (!mchr) mchr = "*";
	put_core(o, r, chr, mchr);
	o.str("ms:i:"); o.u32(r.mscore); o.tab(); o.str(r.aux);
}
// sam_write (src/fixmate.c:549-597): both ends mapped, same contig or not decided by the caller
void write_pair(Out &o, const Rec &r, const bntseq_t *bns)
{
	put_core(o, r, contig(bns, r.tid), r.tid == r.mtid ? "=" : contig(bns, r.mtid));
	o.str("MQ:i:"); o.i32((int)r.mmapq); o.tab(); o.str("ms:i:"); o.u32(r.mscore); o.tab(); o.str(r.aux);
}
// sam_write_discordant (src/fixmate.c:406-443): the mate's contig by name, MQ + MC + ms
void write_discordant(Out &o, const Rec &r, const bntseq_t *bns)
{
	put_core(o, r, contig(bns, r.tid), contig(bns, r.mtid));
	o.str("MQ:i:"); o.i32((int)r.mmapq); o.tab(); o.str("MC:Z:"); o.str(r.mcigar); o.tab(); o.str("ms:i:"); o.u32(r.mscore); o.tab(); o.str(r.aux);
}
// sam_write_mate_unmapped (src/fixmate.c:497-547): one end of the pair is unmapped
void write_one_unmapped(Out &o, const Rec &r, const bntseq_t *bns)
{
	put_core(o, r, contig(bns, r.tid), r.tid == r.mtid ? "=" : contig(bns, r.mtid));
	if (r.flag & F_UNMAP) { o.str("MQ:i:"); o.i32((int)r.mmapq); o.tab(); o.str("MC:Z:"); o.str(r.mcigar); o.tab(); }
	else o.str("MC:Z:*\t");
	o.str("ms:i:"); o.u32(r.mscore); o.tab(); o.str(r.aux);
}
// sam_write_supp_and_secondary (src/fixmate.c:374-404): the mate's contig by name, no tags added
void write_supp(Out &o, const Rec &r, const bntseq_t *bns)
{
	put_core(o, r, contig(bns, r.tid), contig(bns, r.mtid));
	o.str(r.aux);
}

// sync_mate (src/fixmate.c:317-366)
void sync_one_way(const Rec &src, Rec &dst)
{
	dst.mtid = src.tid; dst.mpos = src.pos;
	if (src.flag & F_REVERSE) dst.flag |= F_MREVERSE; else dst.flag &= ~(unsigned)F_MREVERSE;
	if (src.flag & F_UNMAP) dst.flag |= F_MUNMAP;
}
void sync_mate(Rec &a, Rec &b)
{
	if ((b.flag & F_UNMAP) && !(a.flag & F_UNMAP)) { b.tid = a.tid; b.pos = a.pos; }
	if ((a.flag & F_UNMAP) && !(b.flag & F_UNMAP)) { a.tid = b.tid; a.pos = b.pos; }
	sync_one_way(a, b);
	sync_one_way(b, a);
	if (!(a.flag & F_UNMAP)) { b.mmapq = a.mapq; b.mcigar = a.cigar; }
	if (!(b.flag & F_UNMAP)) { a.mmapq = b.mapq; a.mcigar = b.cigar; }
}

// the lines of one mate's text -> recs; false when a line does not parse
bool split_lines(const bseq1_t &s, const ContigIndex &ci, std::vector<Rec> &recs, int &n_lines)
{
	n_lines = 0;
	if (!s.sam) return true;
	const char *p = s.sam, *end = p + strlen(p);
	while (p < end) {
		const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
		if (!nl) break;   // (the reference's line reader stops at text without a newline too: src/fixmate.c:301-315)
		Rec r;
		r.name = s.name ? s.name : "";
		if (!parse_line(p, nl, nl + 1, ci, r)) return false;
		recs.push_back(r);
		++n_lines;
		p = nl + 1;
	}
	return true;
}

int fixmate_pair(bseq1_t *s1, bseq1_t *s2, const bntseq_t *bns, const ContigIndex &ci, std::vector<Rec> &recs, Out o[2])
{
	recs.clear(); o[0].s.clear(); o[1].s.clear();
	int n1 = 0, n2 = 0;
	if (!split_lines(*s1, ci, recs, n1) || !split_lines(*s2, ci, recs, n2)) return -1;
	const int n = n1 + n2;
	Rec *r1 = nullptr, *r2 = nullptr;
	int have = 0;
	auto take = [&](Rec &r) {
		if (r.flag & F_READ1) { r1 = &r; ++have; }
		if (r.flag & F_READ2) { r2 = &r; ++have; }
		return have == 2 && r1 && r2;
	};
	// pass 1: both ends unmapped (src/fixmate.c:693-718)
	for (int i = 0; i < n; ++i) {
		Rec &r = recs[i];
		if (r.done || !(r.flag & F_UNMAP) || !(r.flag & F_MUNMAP)) continue;
		if (!take(r)) continue;
		r1->flag |= F_PAIRED; r2->flag |= F_PAIRED;
		r2->mscore = r1->score; r1->mscore = r2->score;
		write_both_unmapped(o[0], *r1, bns); write_both_unmapped(o[1], *r2, bns);
		r1->done = r2->done = true; have = 0;
	}
	// pass 2: the primary lines of a pair with both ends mapped (src/fixmate.c:721-773)
	have = 0;
	for (int i = 0; i < n; ++i) {
		Rec &r = recs[i];
		if (r.done || !(r.flag & F_PAIRED) || (r.flag & (F_SECONDARY | F_SUPP | F_UNMAP | F_MUNMAP))) continue;
		if (!take(r)) continue;
		const bool other_contigs = r1->tid != r1->mtid && r2->tid != r2->mtid;   // (as the lines came in, before the mates are synchronised)
		sync_mate(*r1, *r2);
		r2->mscore = r1->score; r1->mscore = r2->score;
		if (other_contigs) { write_discordant(o[0], *r1, bns); write_discordant(o[1], *r2, bns); }
		else { write_pair(o[0], *r1, bns); write_pair(o[1], *r2, bns); }
		r1->done = r2->done = true; have = 0;
	}
	// pass 3: secondary and supplementary lines as they come, and the pair with one end unmapped (src/fixmate.c:775-808)
	have = 0;
	for (int i = 0; i < n; ++i) {
		Rec &r = recs[i];
		if (r.done) continue;
		if (r.flag & (F_SECONDARY | F_SUPP)) { write_supp(o[(r.flag & F_READ1) ? 0 : 1], r, bns); r.done = true; continue; }
		if (!(r.flag & F_PAIRED)) continue;
		if (!take(r)) continue;
		sync_mate(*r1, *r2);
		r2->mscore = r1->score; r1->mscore = r2->score;
		write_one_unmapped(o[0], *r1, bns); write_one_unmapped(o[1], *r2, bns);
		r1->done = r2->done = true; have = 0;
	}
	for (int i = 0; i < n; ++i)
		if (!recs[i].done) return -1;   // (the reference asserts that no line is left over, src/fixmate.c:810)
	if (o[0].s.empty() || o[1].s.empty()) return -1;
	char *t1 = (char *)malloc(o[0].s.size() + 1), *t2 = (char *)malloc(o[1].s.size() + 1);
	if (!t1 || !t2) die("out of memory in fixmate");
	memcpy(t1, o[0].s.c_str(), o[0].s.size() + 1);
	memcpy(t2, o[1].s.c_str(), o[1].s.size() + 1);
	free(s1->sam); free(s2->sam);   // (recs' views die here)
	s1->sam = t1; s2->sam = t2;
	return n;
}

// ---- BGZF ----
constexpr size_t BGZF_INPUT = 0xff00;     // text per block: what always fits a 64-KiB block, stored (htslib's choice; src/bgzf.c: 64 KiB, shrinking on overflow)
constexpr size_t BGZF_HEAD = 18, BGZF_TAIL = 8, BGZF_MAX = 0x10000;

// one block (src/bgzf.c:245-330): gzip header with the 'BC' extra field (total block size - 1), raw deflate, CRC32, input length
size_t bgzf_block(const uint8_t *in, size_t n, int level, uint8_t *out)
{
	static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
	size_t clen = 0;
	for (int attempt = 0; attempt < 2; ++attempt) {
		z_stream zs;
		memset(&zs, 0, sizeof zs);
		if (deflateInit2(&zs, attempt ? 0 : level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) die("BGZF: deflateInit2 failed");
		zs.next_in = (Bytef *)in; zs.avail_in = (uInt)n;
		zs.next_out = out + BGZF_HEAD; zs.avail_out = (uInt)(BGZF_MAX - BGZF_HEAD - BGZF_TAIL);
		const int rc = deflate(&zs, Z_FINISH);
		clen = zs.total_out;
		deflateEnd(&zs);
		if (rc == Z_STREAM_END) break;
		if (attempt) die("BGZF: a block of %zu bytes does not fit 64 KiB", n);   // (level 0 always fits BGZF_INPUT bytes)
	}
	memcpy(out, head, 16);
	const size_t total = BGZF_HEAD + clen + BGZF_TAIL;
	out[16] = (uint8_t)((total - 1) & 0xff); out[17] = (uint8_t)((total - 1) >> 8);
	const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in, (uInt)n), isize = (uint32_t)n;
	uint8_t *t = out + BGZF_HEAD + clen;
	for (int k = 0; k < 4; ++k) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)(isize >> (8 * k)); }
	return total;
}

} // namespace

// ---- fixmate ----
// One pair (src/fixmate.c:601): s1->sam / s2->sam are replaced (malloc family, the caller frees them as before).  Returns the
// number of SAM lines of the pair, or -1 when the text is not what mem_sam_pe writes for a pair (a line without a flag, lines left
// over after the three passes): the texts are then left as they were.
extern "C" int mi355x_fixmate_pair(bseq1_t *s1, bseq1_t *s2, const bntseq_t *bns)
{
	const ContigIndex ci(bns);
	std::vector<Rec> recs;
	Out o[2];
	return fixmate_pair(s1, s2, bns, ci, recs, o);
}

// All pairs of a chunk (call_fixmate, src/parallel_aux.c:2164-2206), on the host threads of this rank.  Returns the number of SAM
// lines, or -(index of the first read of the pair that failed) - 1.
extern "C" int64_t mi355x_fixmate(bseq1_t *seqs, int n, const bntseq_t *bns)
{
	if (n & 1) die("mi355x_fixmate: %d reads are not pairs", n);
	const ContigIndex ci(bns);
	std::atomic<int64_t> lines(0), bad(-1);
	run_parallel(n / 2, 2048, [&](int64_t lo, int64_t hi) {
		std::vector<Rec> recs;
		Out o[2];
		int64_t mine = 0;
		for (int64_t p = lo; p < hi; ++p) {
			const int k = fixmate_pair(&seqs[2 * p], &seqs[2 * p + 1], bns, ci, recs, o);
			if (k < 0) { int64_t none = -1; bad.compare_exchange_strong(none, 2 * p); continue; }
			mine += k;
		}
		lines += mine;
	});
	return bad.load() >= 0 ? -bad.load() - 1 : lines.load();
}

// ---- BGZF ----
// room for the compressed form of `len` bytes of text
extern "C" size_t mi355x_bgzf_bound(size_t len) { return (len / (BGZF_INPUT / 2) + 2) * BGZF_MAX; }

// `len` bytes of SAM text as BGZF blocks into out[cap]: blocks end at record ends where a record fits a block (the reference's -g
// writer packs whole reads' records, src/parallel_aux.c:3003-3017), the blocks are compressed side by side and the result does
// not depend on the number of threads.  level: zlib's (-1 = default, as src/bgzf.c:95).  Returns the compressed size, 0 when cap is
// too small.
extern "C" size_t mi355x_bgzf_compress(const char *text, size_t len, int level, uint8_t *out, size_t cap)
{
	if (level > 9 || level < 0) level = Z_DEFAULT_COMPRESSION;
	std::vector<size_t> cut(1, 0);
	for (size_t at = 0; at < len;) {
		size_t n = std::min(BGZF_INPUT, len - at);
		if (at + n < len) {
			const void *nl = memrchr(text + at, '\n', n);
			if (nl) n = (size_t)((const char *)nl - (text + at)) + 1;
		}
		at += n;
		cut.push_back(at);
	}
	const size_t n_blocks = cut.size() - 1;
	if (n_blocks * BGZF_MAX > cap) return 0;
	std::vector<uint32_t> clen(n_blocks);
	// every block into its own 64-KiB slot, then the slots are closed up
	run_parallel((int64_t)n_blocks, 16, [&](int64_t lo, int64_t hi) {
		for (int64_t b = lo; b < hi; ++b) clen[b] = (uint32_t)bgzf_block((const uint8_t *)text + cut[b], cut[b + 1] - cut[b], level, out + (size_t)b * BGZF_MAX);
	});
	size_t w = 0;
	for (size_t b = 0; b < n_blocks; ++b) { memmove(out + w, out + b * BGZF_MAX, clen[b]); w += clen[b]; }
	return w;
}

// the empty block that ends a BGZF file (src/mainParallel.c:1510: the 28 bytes the reference appends to its .bam)
extern "C" size_t mi355x_bgzf_eof(uint8_t out[28])
{
	static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	memcpy(out, eof, 28);
	return 28;
}

// ---- mpiBWAByChr's routing ----
// The records of `sam` (len bytes, whole lines) by destination: 0 .. n_seqs-1 = the contig named in RNAME, then "discordant" (only
// when discordant != 0: pairs without -f, src/mainParallelByChromosome.c:986-1004), last "unmapped" (RNAME '*').  A record whose
// RNAME and RNEXT are two different contigs goes to its contig AND to "discordant" (:1437-1441).  out_text[d] (malloc, or NULL when
// nothing goes there) / out_len[d] for d < n_seqs + 1 + (discordant != 0); record order is kept inside every destination.
// Returns the number of records, or -(byte offset) - 1 of a line that has no RNAME field.
extern "C" int64_t mi355x_route_by_chr(const char *sam, size_t len, const bntseq_t *bns, int discordant, char **out_text, size_t *out_len)
{
	const ContigIndex ci(bns);
	const int n_dest = bns->n_seqs + 1 + (discordant ? 1 : 0), d_unmapped = n_dest - 1, d_disc = discordant ? bns->n_seqs : -1;
	struct Line { size_t at; uint32_t n; int dest; bool disc; };
	// the text in segments of whole lines, parsed side by side; a destination's records keep the order of the text (segment by segment)
	struct Seg { size_t lo = 0, hi = 0; std::vector<Line> lines; std::vector<size_t> bytes; int64_t bad = -1; };
	constexpr size_t SEG = 4u << 20;
	std::vector<Seg> segs;
	for (size_t at = 0; at < len;) {
		size_t hi = std::min(len, at + SEG);
		if (hi < len) {
			const void *nl = memchr(sam + hi, '\n', len - hi);
			hi = nl ? (size_t)((const char *)nl - sam) + 1 : len;
		}
		Seg g;
		g.lo = at; g.hi = hi;
		segs.push_back(std::move(g));
		at = hi;
	}
	run_parallel((int64_t)segs.size(), 1, [&](int64_t s0, int64_t s1) {
		for (int64_t si = s0; si < s1; ++si) {
			Seg &g = segs[(size_t)si];
			g.bytes.assign((size_t)n_dest, 0);
			g.lines.reserve((g.hi - g.lo) / 200 + 16);
			for (size_t at = g.lo; at < g.hi;) {
				const char *p = sam + at, *end = sam + g.hi;
				const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
				const char *le = nl ? nl : end;
				const size_t n = (size_t)(le - p) + (nl ? 1 : 0);
				const char *q = p;
				next_field(q, le); next_field(q, le);
				if (q >= le) { g.bad = (int64_t)at; return; }
				const std::string_view rname = next_field(q, le);
				Line L{at, (uint32_t)n, d_unmapped, false};
				if (rname != "*") {
					const int chr = ci.find(rname);
					if (chr >= 0) {
						L.dest = chr;
						if (discordant) {
							next_field(q, le); next_field(q, le); next_field(q, le);   // POS, MAPQ, CIGAR
							const std::string_view rnext = next_field(q, le);
							const int mchr = rnext == "=" ? chr : rnext == "*" ? -1 : ci.find(rnext);
							L.disc = mchr >= 0 && mchr != chr;
						}
					}
				}
				g.bytes[(size_t)L.dest] += n;
				if (L.disc) g.bytes[(size_t)d_disc] += n;
				g.lines.push_back(L);
				at += n;
			}
		}
	});
	int64_t n_lines = 0;
	std::vector<size_t> total((size_t)n_dest, 0);
	for (Seg &g : segs) {
		if (g.bad >= 0) return -g.bad - 1;
		n_lines += (int64_t)g.lines.size();
		for (int d = 0; d < n_dest; ++d) { const size_t b = g.bytes[(size_t)d]; g.bytes[(size_t)d] = total[(size_t)d]; total[(size_t)d] += b; }   // -> where the segment's records start
	}
	for (int d = 0; d < n_dest; ++d) {
		out_len[d] = total[(size_t)d];
		out_text[d] = nullptr;
		if (total[(size_t)d]) {
			out_text[d] = (char *)malloc(total[(size_t)d] + 1);
			if (!out_text[d]) die("out of memory routing SAM records");
			out_text[d][total[(size_t)d]] = 0;
		}
	}
	run_parallel((int64_t)segs.size(), 1, [&](int64_t s0, int64_t s1) {
		for (int64_t si = s0; si < s1; ++si) {
			Seg &g = segs[(size_t)si];
			for (const Line &L : g.lines) {
				memcpy(out_text[L.dest] + g.bytes[(size_t)L.dest], sam + L.at, L.n); g.bytes[(size_t)L.dest] += L.n;
				if (L.disc) { memcpy(out_text[d_disc] + g.bytes[(size_t)d_disc], sam + L.at, L.n); g.bytes[(size_t)d_disc] += L.n; }
			}
		}
	});
	return n_lines;
}
