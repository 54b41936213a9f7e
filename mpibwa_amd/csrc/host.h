// host.h — host-side stages of the hot path (everything that is not one of the HIP kernels yet).
#ifndef MBW_HOST_H
#define MBW_HOST_H
#include "internal.h"
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace mbw {

struct HSeed {           // mem_seed_t, src/bwamem.c:168-172
	int64_t rbeg;
	int32_t qbeg, len;
	int32_t score;
};

struct HChain {          // mem_chain_t, src/bwamem.c:174-181
	int64_t pos = 0;
	int rid = 0, first = -1;
	uint32_t w = 0;
	int kept = 0, is_alt = 0;
	float frac_rep = 0;
	std::vector<HSeed> seeds;
};

struct HReg {            // mem_alnreg_t, src/bwamem.h:59-77
	int64_t rb = 0, re = 0;
	int qb = 0, qe = 0;
	int rid = 0;
	int score = 0, truesc = 0, sub = 0, alt_sc = 0, csub = 0, sub_n = 0;
	int w = 0, seedcov = 0, secondary = 0, secondary_all = 0, seedlen0 = 0;
	int n_comp = 0, is_alt = 0;
	float frac_rep = 0;
	uint64_t hash = 0;
};
// The regions of one read.  A plain growable array, but able to live in a slice of a batch-wide arena (attach) so
// that a chunk of 667 k reads does not cost 667 k heap blocks that are later freed by other threads; it only falls back
// to its own heap block when a read outgrows its slice (mate rescue adds a few regions at most).
class HRegV {
	HReg *p_ = nullptr;
	uint32_t n_ = 0, cap_ = 0;
	bool own_ = false;
public:
	// true while the list is a fixed point of sort_dedup_patch without patching (set by it; cleared by whoever adds a hit):
	// another such pass would return the list as it is
	bool settled = false;
private:
	void grow(uint32_t want)
	{
		uint32_t nc = cap_ ? cap_ * 2 : 4;
		if (nc < want) nc = want;
		HReg *q = (HReg *)malloc((size_t)nc * sizeof(HReg));
		if (n_) memcpy((void *)q, (const void *)p_, (size_t)n_ * sizeof(HReg));
		if (own_) free(p_);
		p_ = q; cap_ = nc; own_ = true;
	}
public:
	HRegV() {}
	~HRegV() { if (own_) free(p_); }
	HRegV(const HRegV &) = delete;
	HRegV &operator=(const HRegV &) = delete;
	HRegV(HRegV &&o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_), own_(o.own_), settled(o.settled) { o.p_ = nullptr; o.n_ = o.cap_ = 0; o.own_ = false; o.settled = false; }
	HRegV &operator=(HRegV &&o) noexcept { swap(o); return *this; }
	void attach(HReg *ext, uint32_t cap) { if (own_) free(p_); p_ = ext; n_ = 0; cap_ = cap; own_ = false; settled = false; }
	void swap(HRegV &o) { std::swap(p_, o.p_); std::swap(n_, o.n_); std::swap(cap_, o.cap_); std::swap(own_, o.own_); std::swap(settled, o.settled); }
	size_t size() const { return n_; }
	bool empty() const { return n_ == 0; }
	HReg *data() { return p_; }
	const HReg *data() const { return p_; }
	HReg *begin() { return p_; }
	HReg *end() { return p_ + n_; }
	const HReg *begin() const { return p_; }
	const HReg *end() const { return p_ + n_; }
	HReg &operator[](size_t i) { return p_[i]; }
	const HReg &operator[](size_t i) const { return p_[i]; }
	void push_back(const HReg &r) { if (n_ == cap_) grow(n_ + 1); p_[n_++] = r; }
	void insert(HReg *pos, const HReg &r)
	{
		size_t at = pos - p_;
		if (n_ == cap_) grow(n_ + 1);
		memmove((void *)(p_ + at + 1), (const void *)(p_ + at), (n_ - at) * sizeof(HReg));
		p_[at] = r; ++n_;
	}
	void resize(size_t m)
	{
		if (m > cap_) grow((uint32_t)m);
		for (size_t i = n_; i < m; ++i) p_[i] = HReg();
		n_ = (uint32_t)m;
	}
};

// CIGAR of one record: a few operations almost always, so they live inside the object (no heap block per record)
class CigarV {
	static const uint32_t INL = 8;
	uint32_t *p_;
	uint32_t n_ = 0, cap_ = INL;
	uint32_t inl_[INL];
	void grow(uint32_t want)
	{
		uint32_t nc = cap_ * 2 > want ? cap_ * 2 : want;
		uint32_t *q = (uint32_t *)malloc((size_t)nc * 4);
		memcpy(q, p_, (size_t)n_ * 4);
		if (p_ != inl_) free(p_);
		p_ = q; cap_ = nc;
	}
public:
	CigarV() : p_(inl_) {}
	~CigarV() { if (p_ != inl_) free(p_); }
	CigarV(const CigarV &o) : p_(inl_) { assign(o.begin(), o.end()); }
	CigarV &operator=(const CigarV &o) { if (this != &o) assign(o.begin(), o.end()); return *this; }
	void assign(const uint32_t *b, const uint32_t *e)
	{
		uint32_t m = (uint32_t)(e - b);
		if (m > cap_) { n_ = 0; grow(m); }
		if (m) memmove(p_, b, (size_t)m * 4);
		n_ = m;
	}
	size_t size() const { return n_; }
	bool empty() const { return n_ == 0; }
	void clear() { n_ = 0; }
	uint32_t &operator[](size_t i) { return p_[i]; }
	uint32_t operator[](size_t i) const { return p_[i]; }
	uint32_t back() const { return p_[n_ - 1]; }
	const uint32_t *begin() const { return p_; }
	const uint32_t *end() const { return p_ + n_; }
	void push_back(uint32_t v) { if (n_ == cap_) grow(n_ + 1); p_[n_++] = v; }
	void pop_back() { --n_; }
	void push_front(uint32_t v) { if (n_ == cap_) grow(n_ + 1); memmove(p_ + 1, p_, (size_t)n_ * 4); p_[0] = v; ++n_; }
	void pop_front() { memmove(p_, p_ + 1, (size_t)(n_ - 1) * 4); --n_; }
};

struct HAln {            // mem_aln_t, src/bwamem.h:87-98 (cigar + MD kept as separate members)
	int64_t pos = 0;
	int rid = 0, flag = 0;
	uint32_t is_rev = 0, is_alt = 0, mapq = 0, NM = 0;
	CigarV cigar;
	std::string md;
	bool has_xa = false;
	std::string xa;
	int score = 0, sub = 0, alt_sc = 0;
	int n_cigar() const { return (int)cigar.size(); }
};

struct KswResult { int score, te, qe, score2, te2, tb, qb; };   // kswr_t, src/ksw.h:14-20

// ---- where mem_reg2aln's global re-alignment is computed ----
// The SAM stage runs twice around one GPU launch: a COLLECT pass that takes every pairing / marking decision and only
// records which regions need a CIGAR (no text is produced), the device kernel (aln_kernel.hip), and a REPLAY pass that
// repeats the same (deterministic) emission with the device results plugged in.  Without a context the host computes
// the alignment itself (mode HOST) — used for regions the device flags, and by the unit tests of the host logic.
struct AlnReqH { int64_t rb, re; int32_t read, qb, qe, w2, truesc, pad; };   // same layout as the device's AlnReq
struct AlnHdrH { int32_t score, NM, n_cigar, md_len; uint32_t pool_off; int32_t flags; };
struct SamDescH { int64_t rb, re; int32_t qb, qe, req, rid, flag, mapq, score, sub; };   // same layout as the device's SamDesc
struct AlnCtx {
	enum { HOST = 0, COLLECT = 1, REPLAY = 2 };
	int mode = HOST;
	std::vector<AlnReqH> *reqs = nullptr;   // COLLECT: appended in call order
	const AlnHdrH *hdr = nullptr;           // REPLAY: results in the same order, starting at `cursor`
	const uint8_t *pool = nullptr;
	size_t cursor = 0;
	// COLLECT, optional: where sam_pe_emit describes the two lines of a pair that the device can format (sam_kernel.hip);
	// desc[e].req stays < 0 when the pair is not of that kind
	SamDescH *desc = nullptr;
	bool text() const { return mode != COLLECT; }
};

// ---- where mem_matesw's local alignment is computed ----
// Same idea as AlnCtx: the alignments a pair will need are listed up front, computed in one launch of msw_kernel.hip, and
// looked up (by mate, window and strand — the alignment is a pure function of those) when the rescue logic runs.
struct MswReqH { int64_t rb, re; int32_t read, is_rev; };                 // same layout as the device's MswReq
struct MswResH { int32_t score, te, qe, score2, te2, tb, qb, flags; };    // ... MswRes
struct MswCtx { const MswReqH *req = nullptr; const MswResH *res = nullptr; int n = 0; };

struct PairPlan {        // every decision mem_sam_pe takes before it formats anything (src/bwamem_pair.c:264-345)
	int n_pri[2] = {0, 0}, z[2] = {0, 0}, q_se[2] = {0, 0};
	int extra_flag = 1;
	bool paired = false;
	int n_rescue = 0;
};

// ---- reference geometry helpers (src/bntseq.c) ----
int  bns_pos2rid(const bntseq_t *bns, int64_t pos_f);
int  bns_intv2rid(const bntseq_t *bns, int64_t rb, int64_t re);
inline int64_t bns_depos(const bntseq_t *bns, int64_t pos, int *is_rev)
{
	return (*is_rev = (pos >= bns->l_pac)) ? (bns->l_pac << 1) - 1 - pos : pos;
}
std::vector<uint8_t> bns_get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, bool *ok);
std::vector<uint8_t> bns_fetch_seq(const bntseq_t *bns, const uint8_t *pac, int64_t *beg, int64_t mid, int64_t *end, int *rid);

// ---- DP on the host (src/ksw.c) ----
int ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del, int e_del,
                int o_ins, int e_ins, int w, std::vector<uint32_t> *cigar);
KswResult ksw_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, const int8_t *mat, int o_del, int e_del,
                     int o_ins, int e_ins, int xtra);
#define KSW_XBYTE  0x10000
#define KSW_XSTOP  0x20000
#define KSW_XSUBO  0x40000
#define KSW_XSTART 0x80000

// ---- per-read stages ----
int  cal_max_gap(const mem_opt_t *opt, int qlen);
// per-thread scratch of the chaining stage, recycled from read to read (no heap traffic once warm); the HChain
// pointers handed out stay valid until the next chains_from_seeds() on the same scratch
struct ChainScratch {
	struct Impl;
	Impl *p;
	ChainScratch();
	~ChainScratch();
	ChainScratch(const ChainScratch &) = delete;
	ChainScratch &operator=(const ChainScratch &) = delete;
};
void chains_from_seeds(const mem_opt_t *opt, const bntseq_t *bns, int l_query, const HSeed *seeds, int n_seeds, int l_rep,
                       ChainScratch &S, std::vector<HChain *> &chains);
void chain_filter(const mem_opt_t *opt, ChainScratch &S, std::vector<HChain *> &chains);
void filter_chained_seeds(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, int l_query, const uint8_t *query,
                          std::vector<HChain *> &chains);
struct DevChain;
struct DevSeed;
void pack_chain_for_device(const bntseq_t *bns, const HChain &ch, int l_query, const int *gap_h, std::vector<uint64_t> &key, DevChain &d, DevSeed *osd);
int  sort_dedup_patch(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, uint8_t *query, HRegV &regs);
int  mark_primary_se(const mem_opt_t *opt, HRegV &a, int64_t id);
void reorder_primary5(int T, HRegV &a);
int  approx_mapq_se(const mem_opt_t *opt, const HReg *a);
bool gen_cigar2(const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, int w_, int64_t l_pac, const uint8_t *pac,
                int l_query, uint8_t *query, int64_t rb, int64_t re, int *score, std::vector<uint32_t> *cigar, std::string *md, int *NM);
HAln reg2aln(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, int l_query, const char *query, const HReg *ar,
             AlnCtx *ctx = nullptr, int read_idx = 0, bool need_mapq = true);
void reg2sam(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, bseq1_t *s, HRegV &a, int extra_flag, const HAln *m,
             AlnCtx *ctx = nullptr, int read_idx = 0);

// ---- per-batch / per-pair stages (src/bwamem_pair.c) ----
void pestat(const mem_opt_t *opt, int64_t l_pac, int n, const HRegV *regs, mem_pestat_t pes[4], int n_threads = 1);
// the same in pieces: votes of pairs [lo, hi) into 4 x (max_ins + 1) counters, then the statistics from the counters
bool pestat_can_count(const mem_opt_t *opt);
void pestat_gather(const mem_opt_t *opt, int64_t l_pac, int lo, int hi, const HRegV *regs, uint64_t *hist);
void pestat_from_hist(const mem_opt_t *opt, const uint64_t *hist, mem_pestat_t pes[4]);
int  sam_pe(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4], uint64_t id, bseq1_t s[2],
            HRegV a[2]);
// the same in two halves: decisions (mutates a[]), then emission (pure; honours ctx, read0 = index of s[0] in the batch)
void sam_pe_plan(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4], uint64_t id, bseq1_t s[2],
                 HRegV a[2], PairPlan &plan, const MswCtx *mctx = nullptr, int read0 = 0);
void sam_pe_msw_collect(const mem_opt_t *opt, const bntseq_t *bns, const mem_pestat_t pes[4], const bseq1_t s[2], const HRegV a[2], int read0,
                        int max_tlen, std::vector<MswReqH> &out);
void sam_pe_emit(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4], bseq1_t s[2], HRegV a[2],
                 const PairPlan &plan, AlnCtx *ctx, int read0);

inline uint64_t hash_64(uint64_t key)   // Thomas Wang's 64-bit mix, as in src/utils.h:98-109
{
	key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
	key += (key << 3);   key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
	return key;
}

} // namespace mbw
#endif
