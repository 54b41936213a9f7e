// index.cpp — bwa-format index I/O and an own bwa-compatible index builder.
//
// Formats restated from the reference (no code shared):
//   .bwt  : primary, L2[1..4], then occ-interleaved words      src/bwt.c:385-394,443-462
//   .sa   : primary, L2[1..4], sa_intv, seq_len, sa[1..n_sa)    src/bwt.c:396-441
//   .pac  : 2 bit/base MSB-first + tail byte(s)                 src/bntseq.c:224-225,300-315
//   .ann/.amb text                                              src/bntseq.c:66-96,101-158
//   .map  : [bwt_t][bwt words][sa][bntseq_t][ambs][anns][names][pac]   src/bwa.c:310-386
// N bases are replaced by lrand48()&3 after srand48(11)         src/bntseq.c:261,290-291
#include "internal.h"

#include <algorithm>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace mbw {

void die(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	fprintf(stderr, "[mpibwa_amd] FATAL: ");
	vfprintf(stderr, fmt, ap);
	fprintf(stderr, "\n");
	va_end(ap);
	abort();
}

static std::vector<uint8_t> slurp(const std::string &fn)
{
	FILE *fp = fopen(fn.c_str(), "rb");
	if (!fp) die("cannot open '%s': %s", fn.c_str(), strerror(errno));
	fseek(fp, 0, SEEK_END);
	long sz = ftell(fp);
	fseek(fp, 0, SEEK_SET);
	std::vector<uint8_t> buf(sz);
	if (sz && fread(buf.data(), 1, sz, fp) != (size_t)sz) die("short read on '%s'", fn.c_str());
	fclose(fp);
	return buf;
}

void fill_cnt_table(uint32_t tab[256])
{
	// tab[b] packs, per base j, how many of the four 2-bit fields of byte b equal j (byte j of the word)
	for (int b = 0; b < 256; ++b) {
		uint32_t x = 0;
		for (int f = 0; f < 4; ++f) x += 1u << (((b >> (2 * f)) & 3) * 8);
		tab[b] = x;
	}
}

// ---------------------------------------------------------------------------
// loaders
// ---------------------------------------------------------------------------
static bwt_t *load_bwt(const std::string &prefix)
{
	std::vector<uint8_t> raw = slurp(prefix + ".bwt");
	if (raw.size() < 40) die("%s.bwt too short", prefix.c_str());
	bwt_t *bwt = (bwt_t *)calloc(1, sizeof(bwt_t));
	const uint64_t *h = (const uint64_t *)raw.data();
	bwt->primary = h[0];
	for (int i = 0; i < 4; ++i) bwt->L2[i + 1] = h[1 + i];
	bwt->seq_len = bwt->L2[4];
	bwt->bwt_size = (raw.size() - 40) >> 2;
	bwt->bwt = (uint32_t *)malloc(bwt->bwt_size * 4 + 64);
	memcpy(bwt->bwt, raw.data() + 40, bwt->bwt_size * 4);
	fill_cnt_table(bwt->cnt_table);
	return bwt;
}

static void load_sa(const std::string &prefix, bwt_t *bwt)
{
	std::vector<uint8_t> raw = slurp(prefix + ".sa");
	const uint64_t *h = (const uint64_t *)raw.data();
	if (raw.size() < 56 || h[0] != bwt->primary) die("%s.sa does not match .bwt (primary)", prefix.c_str());
	bwt->sa_intv = (int)h[5];
	if (h[6] != bwt->seq_len) die("%s.sa does not match .bwt (seq_len)", prefix.c_str());
	bwt->n_sa = (bwt->seq_len + bwt->sa_intv) / bwt->sa_intv;
	if (raw.size() != 56 + 8 * (bwt->n_sa - 1)) die("%s.sa has unexpected size", prefix.c_str());
	bwt->sa = (bwtint_t *)malloc(8 * bwt->n_sa);
	bwt->sa[0] = (bwtint_t)-1;
	memcpy(bwt->sa + 1, raw.data() + 56, 8 * (bwt->n_sa - 1));
}

static bntseq_t *load_bns(const std::string &prefix)
{
	bntseq_t *bns = (bntseq_t *)calloc(1, sizeof(bntseq_t));
	char line[8192];
	{
		FILE *fp = fopen((prefix + ".ann").c_str(), "r");
		if (!fp) die("cannot open %s.ann", prefix.c_str());
		long long lp;
		if (fscanf(fp, "%lld%d%u", &lp, &bns->n_seqs, &bns->seed) != 3) die("bad .ann header");
		bns->l_pac = lp;
		bns->anns = (bntann1_t *)calloc(bns->n_seqs, sizeof(bntann1_t));
		for (int i = 0; i < bns->n_seqs; ++i) {
			bntann1_t *p = &bns->anns[i];
			if (fscanf(fp, "%u%8191s", &p->gi, line) != 2) die("bad .ann record %d", i);
			p->name = strdup(line);
			// rest of the line = " comment" (or " (null)")
			if (!fgets(line, sizeof line, fp)) die("bad .ann record %d", i);
			size_t l = strlen(line);
			while (l && (line[l - 1] == '\n')) line[--l] = 0;
			if (l > 1 && strcmp(line, " (null)") != 0) p->anno = strdup(line + 1);
			else p->anno = strdup("");
			long long off;
			if (fscanf(fp, "%lld%d%d", &off, &p->len, &p->n_ambs) != 3) die("bad .ann record %d", i);
			p->offset = off;
		}
		fclose(fp);
	}
	{
		FILE *fp = fopen((prefix + ".amb").c_str(), "r");
		if (!fp) die("cannot open %s.amb", prefix.c_str());
		long long lp; int ns;
		if (fscanf(fp, "%lld%d%d", &lp, &ns, &bns->n_holes) != 3) die("bad .amb header");
		if (lp != bns->l_pac || ns != bns->n_seqs) die(".ann and .amb disagree");
		bns->ambs = bns->n_holes ? (bntamb1_t *)calloc(bns->n_holes, sizeof(bntamb1_t)) : 0;
		for (int i = 0; i < bns->n_holes; ++i) {
			long long off;
			if (fscanf(fp, "%lld%d%8191s", &off, &bns->ambs[i].len, line) != 3) die("bad .amb record");
			bns->ambs[i].offset = off;
			bns->ambs[i].amb = line[0];
		}
		fclose(fp);
	}
	// optional .alt: first column = contig name to be flagged ALT (src/bntseq.c:179-204)
	if (FILE *fp = fopen((prefix + ".alt").c_str(), "r")) {
		while (fgets(line, sizeof line, fp)) {
			if (line[0] == '@') continue;
			size_t l = strcspn(line, "\t\r\n");
			line[l] = 0;
			for (int i = 0; i < bns->n_seqs; ++i)
				if (strcmp(bns->anns[i].name, line) == 0) bns->anns[i].is_alt = 1;
		}
		fclose(fp);
	}
	return bns;
}

} // namespace mbw

using namespace mbw;

extern "C" bwaidx_t *bwa_idx_load_from_disk(const char *prefix_, int which)
{
	(void)which; // the reference's BWA_IDX_* selector; we always load everything
	std::string prefix(prefix_);
	bwaidx_t *idx = (bwaidx_t *)calloc(1, sizeof(bwaidx_t));
	idx->bwt = load_bwt(prefix);
	load_sa(prefix, idx->bwt);
	idx->bns = load_bns(prefix);
	std::vector<uint8_t> pac = slurp(prefix + ".pac");
	size_t need = idx->bns->l_pac / 4 + 1;
	if (pac.size() < need) die("%s.pac too short", prefix.c_str());
	idx->pac = (uint8_t *)calloc(need + 64, 1);
	memcpy(idx->pac, pac.data(), need);
	return idx;
}

// Attach to a `.map` image in place (pointer fix-up only).
extern "C" int bwa_mem2idx(int64_t l_mem, uint8_t *mem, bwaidx_t *idx)
{
	int64_t k = 0;
	idx->bwt = (bwt_t *)mem; k += sizeof(bwt_t);
	idx->bwt->bwt = (uint32_t *)(mem + k); k += idx->bwt->bwt_size * 4;
	idx->bwt->sa = (bwtint_t *)(mem + k); k += idx->bwt->n_sa * 8;
	idx->bns = (bntseq_t *)(mem + k); k += sizeof(bntseq_t);
	idx->bns->ambs = (bntamb1_t *)(mem + k); k += (int64_t)idx->bns->n_holes * sizeof(bntamb1_t);
	idx->bns->anns = (bntann1_t *)(mem + k); k += (int64_t)idx->bns->n_seqs * sizeof(bntann1_t);
	for (int i = 0; i < idx->bns->n_seqs; ++i) {
		idx->bns->anns[i].name = (char *)(mem + k); k += strlen((char *)(mem + k)) + 1;
		idx->bns->anns[i].anno = (char *)(mem + k); k += strlen((char *)(mem + k)) + 1;
	}
	idx->pac = mem + k; k += idx->bns->l_pac / 4 + 1;
	if (k != l_mem) die("bwa_mem2idx: image size mismatch (%lld vs %lld)", (long long)k, (long long)l_mem);
	idx->l_mem = k; idx->mem = mem;
	return 0;
}

extern "C" void bwa_idx_destroy(bwaidx_t *idx)
{
	if (!idx) return;
	if (idx->mem == 0) {
		if (idx->bwt) { free(idx->bwt->bwt); free(idx->bwt->sa); free(idx->bwt); }
		if (idx->bns) {
			for (int i = 0; i < idx->bns->n_seqs; ++i) { free(idx->bns->anns[i].name); free(idx->bns->anns[i].anno); }
			free(idx->bns->anns); free(idx->bns->ambs); free(idx->bns);
		}
		free(idx->pac);
	}
	free(idx);
}

// ---------------------------------------------------------------------------
// index builder
// ---------------------------------------------------------------------------
namespace mbw {

// Suffix array by induced sorting (Nong, Zhang & Chan 2009).  s[n-1] must be a
// unique smallest sentinel.  I = index type (int32_t or int64_t).
template <typename I, typename S>
static void sais(const S *s, I *sa, I n, I K)
{
	std::vector<bool> stype(n);
	stype[n - 1] = true;
	for (I i = n - 2; i >= 0; --i)
		stype[i] = s[i] < s[i + 1] || (s[i] == s[i + 1] && stype[i + 1]);
	auto is_lms = [&](I i) { return i > 0 && stype[i] && !stype[i - 1]; };
	std::vector<I> bkt(K);
	auto buckets = [&](bool ends) {
		std::fill(bkt.begin(), bkt.end(), (I)0);
		for (I i = 0; i < n; ++i) ++bkt[s[i]];
		I sum = 0;
		for (I c = 0; c < K; ++c) { sum += bkt[c]; bkt[c] = ends ? sum : sum - bkt[c]; }
	};
	auto induce = [&]() {
		buckets(false);
		for (I i = 0; i < n; ++i) {
			I j = sa[i] - 1;
			if (sa[i] > 0 && !stype[j]) sa[bkt[s[j]]++] = j;
		}
		buckets(true);
		for (I i = n - 1; i >= 0; --i) {
			I j = sa[i] - 1;
			if (sa[i] > 0 && stype[j]) sa[--bkt[s[j]]] = j;
		}
	};
	// 1. sort LMS substrings
	buckets(true);
	std::fill(sa, sa + n, (I)-1);
	for (I i = 1; i < n; ++i)
		if (is_lms(i)) sa[--bkt[s[i]]] = i;
	induce();
	I n1 = 0;
	for (I i = 0; i < n; ++i)
		if (is_lms(sa[i])) sa[n1++] = sa[i];
	std::fill(sa + n1, sa + n, (I)-1);
	I name = 0, prev = -1;
	for (I i = 0; i < n1; ++i) {
		I pos = sa[i];
		bool diff = prev < 0;
		for (I d = 0; !diff; ++d) {
			if (s[pos + d] != s[prev + d] || stype[pos + d] != stype[prev + d]) diff = true;
			else if (d > 0 && (is_lms(pos + d) || is_lms(prev + d))) break;
		}
		if (diff) { ++name; prev = pos; }
		sa[n1 + (pos >> 1)] = name - 1;
	}
	for (I i = n - 1, j = n - 1; i >= n1; --i)
		if (sa[i] >= 0) sa[j--] = sa[i];
	// 2. order of the LMS suffixes
	I *s1 = sa + n - n1;
	if (name < n1) sais<I, I>(s1, sa, n1, name);
	else for (I i = 0; i < n1; ++i) sa[s1[i]] = i;
	// 3. induce the final order
	buckets(true);
	for (I i = 1, j = 0; i < n; ++i)
		if (is_lms(i)) s1[j++] = i;
	for (I i = 0; i < n1; ++i) sa[i] = s1[sa[i]];
	std::fill(sa + n1, sa + n, (I)-1);
	for (I i = n1 - 1; i >= 0; --i) {
		I j = sa[i];
		sa[i] = -1;
		sa[--bkt[s[j]]] = j;
	}
	induce();
}

struct FastaRec { std::string name, comment, seq; };

static std::vector<FastaRec> read_fasta(const char *fn)
{
	FILE *fp = fopen(fn, "r");
	if (!fp) die("cannot open FASTA '%s'", fn);
	std::vector<FastaRec> recs;
	char *line = 0; size_t cap = 0; ssize_t l;
	while ((l = getline(&line, &cap, fp)) > 0) {
		while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
		if (line[0] == '>') {
			recs.emplace_back();
			char *p = line + 1;
			size_t k = strcspn(p, " \t");
			recs.back().name.assign(p, k);
			p += k;
			while (*p == ' ' || *p == '\t') ++p;
			recs.back().comment = p;
		} else if (!recs.empty()) {
			for (ssize_t i = 0; i < l; ++i)
				if (line[i] > ' ') recs.back().seq.push_back(line[i]);
		}
	}
	free(line);
	fclose(fp);
	return recs;
}

static void write_or_die(const std::string &fn, const void *p, size_t n, const char *mode = "wb")
{
	FILE *fp = fopen(fn.c_str(), mode);
	if (!fp || (n && fwrite(p, 1, n, fp) != n)) die("cannot write '%s'", fn.c_str());
	fclose(fp);
}

} // namespace mbw

extern "C" int mi355x_index_build(const char *fasta, const char *prefix_)
{
	std::string prefix(prefix_);
	std::vector<FastaRec> recs = read_fasta(fasta);
	if (recs.empty()) die("no sequences in '%s'", fasta);

	// ---- pac / ann / amb ----
	int64_t l_pac = 0;
	for (auto &r : recs) l_pac += r.seq.size();
	std::vector<uint8_t> fwd(l_pac);
	struct Hole { int64_t off; int32_t len; char amb; };
	std::vector<Hole> holes;
	std::vector<int32_t> n_ambs(recs.size(), 0);
	srand48(11);
	int64_t pos = 0;
	for (size_t r = 0; r < recs.size(); ++r) {
		int last = 0;
		for (char ch : recs[r].seq) {
			int c = nt4_table[(uint8_t)ch];
			if (c >= 4) {
				if (last == ch) ++holes.back().len;
				else { holes.push_back({pos, 1, ch}); ++n_ambs[r]; }
				c = lrand48() & 3;
			}
			last = ch;
			fwd[pos++] = (uint8_t)c;
		}
	}
	{
		std::vector<uint8_t> pac(l_pac / 4 + 1, 0);
		for (int64_t i = 0; i < l_pac; ++i) pac[i >> 2] |= fwd[i] << ((~i & 3) << 1);
		// file size is always l_pac/4 + 2 bytes; the last byte is l_pac % 4
		std::vector<uint8_t> out(pac.begin(), pac.begin() + (l_pac >> 2) + ((l_pac & 3) ? 1 : 0));
		if ((l_pac & 3) == 0) out.push_back(0);
		out.push_back((uint8_t)(l_pac & 3));
		write_or_die(prefix + ".pac", out.data(), out.size());
	}
	{
		std::string ann, amb;
		char buf[256];
		snprintf(buf, sizeof buf, "%lld %d %u\n", (long long)l_pac, (int)recs.size(), 11u);
		ann += buf;
		int64_t off = 0;
		for (size_t r = 0; r < recs.size(); ++r) {
			ann += "0 " + recs[r].name + " " + (recs[r].comment.empty() ? std::string("(null)") : recs[r].comment) + "\n";
			snprintf(buf, sizeof buf, "%lld %d %d\n", (long long)off, (int)recs[r].seq.size(), n_ambs[r]);
			ann += buf;
			off += recs[r].seq.size();
		}
		snprintf(buf, sizeof buf, "%lld %d %u\n", (long long)l_pac, (int)recs.size(), (unsigned)holes.size());
		amb += buf;
		for (auto &h : holes) {
			snprintf(buf, sizeof buf, "%lld %d %c\n", (long long)h.off, h.len, h.amb);
			amb += buf;
		}
		write_or_die(prefix + ".ann", ann.data(), ann.size(), "w");
		write_or_die(prefix + ".amb", amb.data(), amb.size(), "w");
	}

	// ---- text = forward + reverse complement, then suffix array ----
	const int64_t n = 2 * l_pac;
	std::vector<uint8_t> text(n + 1);
	for (int64_t i = 0; i < l_pac; ++i) {
		text[i] = fwd[i] + 1;
		text[n - 1 - i] = (3 - fwd[i]) + 1;
	}
	text[n] = 0;
	bwtint_t L2[5] = {0, 0, 0, 0, 0};
	for (int64_t i = 0; i < n; ++i) ++L2[text[i]];
	for (int i = 2; i <= 4; ++i) L2[i] += L2[i - 1];

	std::vector<uint8_t> B(n);   // BWT with the '$' row removed
	bwtint_t primary = 0;
	const int sa_intv = 32;
	bwtint_t n_sa = (n + sa_intv) / sa_intv;
	std::vector<bwtint_t> sa_smp(n_sa);
	auto finish = [&](auto *sa) {
		// rows 0..n of the (n+1)-row matrix; row `primary` holds suffix 0 (its BWT char is '$')
		for (int64_t i = 0; i <= n; ++i)
			if (sa[i] == 0) primary = i;
		for (int64_t i = 0; i <= n; ++i) {
			if ((bwtint_t)i == primary) continue;
			B[i - ((bwtint_t)i > primary)] = text[sa[i] - 1] - 1;
		}
		for (int64_t i = 0; i <= n; i += sa_intv) sa_smp[i / sa_intv] = sa[i];
	};
	if (n + 1 < (int64_t)0x7fffffff) {
		std::vector<int32_t> sa(n + 1);
		sais<int32_t, uint8_t>(text.data(), sa.data(), (int32_t)(n + 1), 5);
		finish(sa.data());
	} else {
		std::vector<int64_t> sa(n + 1);
		sais<int64_t, uint8_t>(text.data(), sa.data(), n + 1, 5);
		finish(sa.data());
	}
	std::vector<uint8_t>().swap(text);

	// ---- occ-interleaved .bwt: per 128 bases 4 x u64 running counts + 8 x u32 packed bases, then the final counts ----
	{
		const bwtint_t n_occ = (n + 127) / 128 + 1;
		const bwtint_t n_words = ((n + 15) >> 4) + n_occ * 8;
		std::vector<uint32_t> w(n_words, 0);
		bwtint_t c[4] = {0, 0, 0, 0}, k = 0;
		for (int64_t i = 0; i < n; ++i) {
			if ((i & 127) == 0) { memcpy(&w[k], c, 32); k += 8; }
			if ((i & 15) == 0) ++k;
			w[k - 1] |= (uint32_t)B[i] << ((~i & 15) << 1);
			++c[B[i]];
		}
		memcpy(&w[k], c, 32); k += 8;
		if (k != n_words) die("index builder: inconsistent bwt size");
		FILE *fp = fopen((prefix + ".bwt").c_str(), "wb");
		if (!fp) die("cannot write %s.bwt", prefix.c_str());
		fwrite(&primary, 8, 1, fp);
		fwrite(L2 + 1, 8, 4, fp);
		fwrite(w.data(), 4, w.size(), fp);
		fclose(fp);
	}
	{
		FILE *fp = fopen((prefix + ".sa").c_str(), "wb");
		if (!fp) die("cannot write %s.sa", prefix.c_str());
		bwtint_t intv = sa_intv, sl = n;
		fwrite(&primary, 8, 1, fp);
		fwrite(L2 + 1, 8, 4, fp);
		fwrite(&intv, 8, 1, fp);
		fwrite(&sl, 8, 1, fp);
		fwrite(sa_smp.data() + 1, 8, n_sa - 1, fp);
		fclose(fp);
	}
	return 0;
}
