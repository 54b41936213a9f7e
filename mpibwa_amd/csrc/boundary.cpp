// boundary.cpp — the symbols mpiBWA's main() needs besides mem_process_seqs(): read-group / header helpers and the
// `.map` image packer.  Own statements of
//   bwa_set_rg         src/bwa.c:431-463   (called from src/mainParallel.c:356)
//   bwa_insert_header  src/bwa.c:465-476   (called from src/mainParallel.c:368, :373)
//   bwa_idx2mem        src/bwa.c:347-386   (called from src/pidx.c:56 to write <prefix>.map)
// With these exported, the reference's driver objects (mainParallel.o, parallel_aux.o, fixmate.o, tokenizer.o, ...)
// link against libmpibwa_amd.so with no undefined symbol left (tests/test_link.py).
#include "internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace mbw;

namespace {

// "\t" "\n" "\r" "\\" written as two characters become the character; any other escaped character is dropped together
// with its backslash (src/bwa.c:413-429).  In place; the string can only shrink.
void unescape(char *s)
{
	char *w = s;
	for (const char *r = s; *r; ++r) {
		if (*r != '\\') { *w++ = *r; continue; }
		++r;
		switch (*r) {
		case 't': *w++ = '\t'; break;
		case 'n': *w++ = '\n'; break;
		case 'r': *w++ = '\r'; break;
		case '\\': *w++ = '\\'; break;
		default: break;
		}
		if (!*r) break;   // a lone backslash at the end of the string
	}
	*w = 0;
}

void rg_error(const char *what)
{
	if (bwa_verbose >= 1) fprintf(stderr, "[E::bwa_set_rg] %s\n", what);
}

}   // namespace

// Returns the unescaped @RG line (malloc()ed, the caller keeps it as the header line) and fills bwa_rg_id with the value
// of its ID field; 0 on a malformed line.
extern "C" char *bwa_set_rg(const char *s)
{
	memset(bwa_rg_id, 0, sizeof bwa_rg_id);
	if (strncmp(s, "@RG", 3) != 0) { rg_error("the read group line is not started with @RG"); return 0; }
	if (strchr(s, '\t')) { rg_error("the read group line contained literal <tab> characters -- replace with escaped tabs: \\t"); return 0; }
	char *line = strdup(s);
	if (!line) die("out of memory");
	unescape(line);
	const char *id = strstr(line, "\tID:");
	if (!id) { rg_error("no ID within the read group line"); free(line); return 0; }
	id += 4;
	const size_t n = strcspn(id, "\t\n");
	if (n + 1 > sizeof bwa_rg_id) { rg_error("@RG:ID is longer than 255 characters"); free(line); return 0; }
	memcpy(bwa_rg_id, id, n);
	return line;
}

// Appends header line `s` (must start with '@', else hdr is returned unchanged) to the header text `hdr` (may be 0),
// separated by a newline; only the appended part is unescaped.
extern "C" char *bwa_insert_header(const char *s, char *hdr)
{
	if (!s || s[0] != '@') return hdr;
	size_t at = 0;
	if (hdr) {
		const size_t old = strlen(hdr);
		hdr = (char *)realloc(hdr, old + strlen(s) + 2);
		if (!hdr) die("out of memory");
		hdr[old] = '\n';
		at = old + 1;
		strcpy(hdr + at, s);
	} else {
		hdr = strdup(s);
		if (!hdr) die("out of memory");
	}
	unescape(hdr + at);
	return hdr;
}

// Size of the `.map` image of an index: [bwt_t][bwt words][sa][bntseq_t][ambs][anns][name\0anno\0 ...][pac]
static int64_t map_image_bytes(const bwaidx_t *idx)
{
	int64_t k = sizeof(bwt_t) + (int64_t)idx->bwt->bwt_size * 4 + (int64_t)idx->bwt->n_sa * sizeof(bwtint_t);
	k += sizeof(bntseq_t) + (int64_t)idx->bns->n_holes * sizeof(bntamb1_t) + (int64_t)idx->bns->n_seqs * sizeof(bntann1_t);
	for (int i = 0; i < idx->bns->n_seqs; ++i) k += strlen(idx->bns->anns[i].name) + strlen(idx->bns->anns[i].anno) + 2;
	return k + idx->bns->l_pac / 4 + 1;
}

// Lay the index out as one contiguous image; the structs inside keep whatever pointer values they had (bwa_mem2idx
// rebuilds every pointer from the sizes, so a consumer never reads them).
static void map_image_fill(const bwaidx_t *idx, uint8_t *mem)
{
	uint8_t *p = mem;
	auto put = [&p](const void *src, size_t n) { memcpy(p, src, n); p += n; };
	put(idx->bwt, sizeof(bwt_t));
	put(idx->bwt->bwt, (size_t)idx->bwt->bwt_size * 4);
	put(idx->bwt->sa, (size_t)idx->bwt->n_sa * sizeof(bwtint_t));
	put(idx->bns, sizeof(bntseq_t));
	put(idx->bns->ambs, (size_t)idx->bns->n_holes * sizeof(bntamb1_t));
	put(idx->bns->anns, (size_t)idx->bns->n_seqs * sizeof(bntann1_t));
	for (int i = 0; i < idx->bns->n_seqs; ++i) {
		put(idx->bns->anns[i].name, strlen(idx->bns->anns[i].name) + 1);
		put(idx->bns->anns[i].anno, strlen(idx->bns->anns[i].anno) + 1);
	}
	put(idx->pac, (size_t)idx->bns->l_pac / 4 + 1);
}

// Same contract as the reference's: the separately allocated parts of a disk-loaded index are replaced by one malloc()ed
// image and `idx` is re-attached to it (idx->mem / idx->l_mem describe the image mpiBWAIdx writes to <prefix>.map).
extern "C" int bwa_idx2mem(bwaidx_t *idx)
{
	if (idx->mem) return 0;   // already an image
	const int64_t bytes = map_image_bytes(idx);
	uint8_t *mem = (uint8_t *)malloc((size_t)bytes);
	if (!mem) die("out of memory packing the index image (%lld bytes)", (long long)bytes);
	map_image_fill(idx, mem);
	// release the parts (what bwa_idx_destroy does for a disk-loaded index), keep the handle
	free(idx->bwt->bwt); free(idx->bwt->sa); free(idx->bwt);
	for (int i = 0; i < idx->bns->n_seqs; ++i) { free(idx->bns->anns[i].name); free(idx->bns->anns[i].anno); }
	free(idx->bns->anns); free(idx->bns->ambs);
	free(idx->bns);
	free(idx->pac);
	idx->bwt = 0; idx->bns = 0; idx->pac = 0;
	return bwa_mem2idx(bytes, mem, idx);
}

// mpiBWAIdx in one call (src/pidx.c:52-63): load <prefix>.{bwt,sa,ann,amb,pac}, pack, write <prefix>.map.
extern "C" int mi355x_write_map(const char *prefix, const char *map_path)
{
	bwaidx_t *idx = bwa_idx_load_from_disk(prefix, 7);
	if (!idx) return -1;
	const int64_t bytes = map_image_bytes(idx);
	uint8_t *mem = (uint8_t *)malloc((size_t)bytes);
	if (!mem) die("out of memory packing the index image (%lld bytes)", (long long)bytes);
	map_image_fill(idx, mem);
	// the image is meant to be attached by another process: this one's heap addresses stay out of the file
	// (bwa_mem2idx rebuilds every pointer from the sizes)
	{
		uint8_t *p = mem;
		bwt_t *b = (bwt_t *)p;
		p += sizeof(bwt_t) + (size_t)b->bwt_size * 4 + (size_t)b->n_sa * sizeof(bwtint_t);
		b->bwt = 0; b->sa = 0;
		bntseq_t *n = (bntseq_t *)p;
		p += sizeof(bntseq_t) + (size_t)n->n_holes * sizeof(bntamb1_t);
		n->anns = 0; n->ambs = 0; n->fp_pac = 0;
		bntann1_t *a = (bntann1_t *)p;
		for (int i = 0; i < n->n_seqs; ++i) a[i].name = a[i].anno = 0;
	}
	bwa_idx_destroy(idx);
	FILE *fp = fopen(map_path, "wb");
	if (!fp) { free(mem); return -2; }
	const size_t w = fwrite(mem, 1, (size_t)bytes, fp);
	const int rc = fclose(fp);
	free(mem);
	return (w == (size_t)bytes && rc == 0) ? 0 : -3;
}
