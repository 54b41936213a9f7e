// device.hip — device management, index residency in HBM and the stage-level C entry points.
#include <hip/hip_runtime.h>
#include <atomic>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "device.h"
#include "host.h"
#include <cmath>

namespace mbw {

#define HIP_OK(call)                                                                                             \
	do {                                                                                                         \
		hipError_t e_ = (call);                                                                                  \
		if (e_ != hipSuccess) die("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);    \
	} while (0)

static DevIndex g_idx;
DevIndex &dev_index() { return g_idx; }
static std::atomic<unsigned long long> g_buf_growths(0);
void note_buffer_growth(size_t from, size_t to, const char *kind)
{
	++g_buf_growths;
	static const bool log = getenv("MPIBWA_GROWTH_LOG") != nullptr;
	if (log) fprintf(stderr, "[growth] %s buffer %zu -> %zu bytes (reallocation %llu)\n", kind, from, to, g_buf_growths.load());
}
// (re)allocations of device and page-locked work buffers so far: hipFree / hipMalloc / hipHostMalloc stall every stream of the
// device, so a caller that sees this number move in steady state knows where a slow chunk came from
extern "C" unsigned long long mi355x_buffer_growths(void) { return g_buf_growths.load(); }
std::vector<DevBuf *> *g_devbuf_owner = nullptr;
void *DevBuf::ensure(size_t bytes)
{
	if (bytes > cap) {
		size_t want = bytes + bytes / 4 + 256;
		note_buffer_growth(cap, want, "device");
		if (p) HIP_OK(hipFree(p));
		p = nullptr; cap = 0;
		// (what the runtime itself needs later — kernel scratch, queues, events — comes out of the same HBM, and it aborts the process
		// when it finds none: an allocation that leaves less than a reserve free counts as one that did not fit)
		static const size_t reserve = (size_t)(getenv("MPIBWA_HBM_RESERVE_GB") ? atof(getenv("MPIBWA_HBM_RESERVE_GB")) : 6.0) << 30;
		for (int attempt = 0;; ++attempt) {
			if (hipMalloc(&p, want) == hipSuccess) {
				size_t fr = 0, tot = 0;
				if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr >= reserve) break;
				(void)hipFree(p);
			}
			(void)hipGetLastError();
			p = nullptr;
			if (attempt == 0 && want > bytes + 256) { want = bytes + 256; continue; }   // without the head room first
			if (device_memory_pressure(want)) continue;
			size_t fr = 0, tot = 0;
			(void)hipMemGetInfo(&fr, &tot);
			die("device work buffer of %.2f GB does not fit: %.1f of %.1f GB free (index, dense SA and the work buffers of this call share the HBM; no other "
			    "call is in flight and no idle buffer is left to give back)", want / 1e9, fr / 1e9, tot / 1e9);
		}
		cap = want;
	}
	return p;
}
void DevBuf::release()
{
	if (p) (void)hipFree(p);
	p = nullptr; cap = 0;
}

static void require_device(int local_rank)
{
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n == 0)
		die("no HIP device visible: the alignment hot path runs on MI355X only, there is no CPU fallback");
	HIP_OK(hipSetDevice(local_rank % n));
	hipDeviceProp_t prop;
	HIP_OK(hipGetDeviceProperties(&prop, local_rank % n));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		die("device %d is %s; this library carries gfx950 (MI355X) code objects only", local_rank % n, prop.gcnArchName);
	g_idx.device = local_rank % n;
}

static uint64_t index_hash(const bwt_t *bwt, const bntseq_t *bns)
{
	uint64_t h = 1469598103934665603ull;   // FNV-1a
	auto mix = [&](const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
	for (int i = 0; i < bns->n_seqs; ++i) {
		mix(&bns->anns[i].offset, sizeof bns->anns[i].offset);
		mix(&bns->anns[i].len, sizeof bns->anns[i].len);
		if (bns->anns[i].name) mix(bns->anns[i].name, strlen(bns->anns[i].name));
	}
	if (bwt->bwt && bwt->bwt_size >= 16) mix(bwt->bwt, 64);
	return h;
}

// (called under index_mutex() on every mem_process_seqs call: the contig table — 10^5..10^6 names on some references — is only
// hashed again when the caller passes other pointers than the ones that matched last time)
static const void *g_last_ok[4] = {nullptr, nullptr, nullptr, nullptr};
static int g_last_ok_nseqs = -1;
bool index_matches(const bwt_t *bwt, const bntseq_t *bns, const char **what)
{
	const char *w = nullptr;
	const bool same_ptrs = g_last_ok[0] == bwt && g_last_ok[1] == bns && g_last_ok[2] == bns->anns && g_last_ok[3] == bwt->bwt && g_last_ok_nseqs == bns->n_seqs;
	if (bwt->seq_len != g_idx.id_seq_len) w = "seq_len";
	else if (bwt->primary != g_idx.id_primary) w = "primary";
	else if (memcmp(bwt->L2, g_idx.id_L2, sizeof g_idx.id_L2) != 0) w = "L2";
	else if (bns->l_pac != g_idx.l_pac) w = "l_pac";
	else if (bns->n_seqs != g_idx.id_n_seqs) w = "n_seqs";
	else if (!same_ptrs && index_hash(bwt, bns) != g_idx.id_hash) w = "contig table or BWT";
	if (!w) { g_last_ok[0] = bwt; g_last_ok[1] = bns; g_last_ok[2] = bns->anns; g_last_ok[3] = bwt->bwt; g_last_ok_nseqs = bns->n_seqs; }
	else g_last_ok[0] = nullptr;
	if (what) *what = w;
	return w == nullptr;
}

static void no_calls_in_flight(const char *who)
{
	if (calls_in_flight() > 0) die("%s while mem_process_seqs calls are in flight on the resident index", who);
}

static void alloc_index(const bwt_t *bwt, const bntseq_t *bns)
{
	std::lock_guard<std::recursive_mutex> lk(index_mutex());   // no call can pass its residency check while the buffers are replaced
	no_calls_in_flight("index upload");
	g_idx.ready = false;
	if (g_idx.d_blk) { (void)hipFree(g_idx.d_blk); (void)hipFree(g_idx.d_sa); (void)hipFree(g_idx.d_pac); }
	g_idx.blk_bytes = ((size_t)bwt->bwt_size * 4 + 63) / 64 * 64 + 64;   // whole 64-B blocks + one pad block
	g_idx.sa_bytes = (size_t)bwt->n_sa * 8;
	g_idx.pac_bytes = (size_t)bns->l_pac / 4 + 1 + 16;
	HIP_OK(hipMalloc(&g_idx.d_blk, g_idx.blk_bytes));
	HIP_OK(hipMalloc(&g_idx.d_sa, g_idx.sa_bytes));
	HIP_OK(hipMalloc(&g_idx.d_pac, g_idx.pac_bytes));
	HIP_OK(hipMemset(g_idx.d_blk, 0, g_idx.blk_bytes));
	HIP_OK(hipMemset(g_idx.d_pac, 0, g_idx.pac_bytes));
	FmDev &fm = g_idx.fm;
	fm.blk = g_idx.d_blk; fm.sa = (const uint64_t *)g_idx.d_sa; fm.sa_full = nullptr;
	fm.p3tab = nullptr; fm.p3_k = 0; fm.occ32 = nullptr; fm.occ_sb = nullptr; fm.kmt = nullptr; fm.kmt_k = 0;
	if (g_idx.d_occ32) { (void)hipFree(g_idx.d_occ32); g_idx.d_occ32 = nullptr; }
	if (g_idx.d_kmt) { (void)hipFree(g_idx.d_kmt); g_idx.d_kmt = nullptr; g_idx.kmt_bytes = 0; }
	if (g_idx.d_p3tab) { (void)hipFree(g_idx.d_p3tab); g_idx.d_p3tab = nullptr; }
	if (g_idx.d_sa_full) { (void)hipFree(g_idx.d_sa_full); g_idx.d_sa_full = nullptr; g_idx.sa_full_bytes = 0; }
	fm.primary = bwt->primary; fm.seq_len = bwt->seq_len;
	for (int i = 0; i < 5; ++i) fm.L2[i] = bwt->L2[i];
	int sh = 0;
	while ((1 << sh) < bwt->sa_intv) ++sh;
	if ((1 << sh) != bwt->sa_intv) die("SA sampling interval %d is not a power of two", bwt->sa_intv);
	fm.sa_shift = sh;
	g_idx.l_pac = bns->l_pac;
	g_idx.id_primary = bwt->primary; g_idx.id_seq_len = bwt->seq_len; g_idx.id_n_seqs = bns->n_seqs;
	g_idx.id_hash = index_hash(bwt, bns);
	g_last_ok[0] = nullptr;   // a new resident index: the next call hashes its contig table again
	memcpy(g_idx.id_L2, bwt->L2, sizeof g_idx.id_L2);
	if (bwt->seq_len >= (1ull << 34))
		die("reference of %llu symbols: this build packs SA-interval bounds into 34 bits (references up to 8.5 Gbp)", (unsigned long long)bwt->seq_len);
	// the buffers are handed to collectives on other streams next (RCCL, torch): the memsets above must have landed
	HIP_OK(hipDeviceSynchronize());
}

} // namespace mbw

using namespace mbw;

// Expand the sampled SA into a dense one when HBM allows it (MPIBWA_SA_DENSE=0 disables it).
// Jump table of the third seeding pass (fm_kernels.hip): 4^13 entries x 32 B = 2.1 GB, built in a few tens of ms.
// MPIBWA_P3TAB=0 disables it, MPIBWA_P3TAB=<k> chooses the number of extensions folded into it (default 12).
static void maybe_build_p3()
{
	int k = 12;
	if (const char *e = getenv("MPIBWA_P3TAB")) k = atoi(e);
	if (g_idx.d_p3tab) { (void)hipFree(g_idx.d_p3tab); g_idx.d_p3tab = nullptr; g_idx.fm.p3tab = nullptr; }
	if (k < 4 || k > 14) return;
	const size_t bytes = ((size_t)1 << (2 * (k + 1))) * 32;
	size_t free_b = 0, total_b = 0;
	if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + ((size_t)16 << 30)) return;
	HIP_OK(hipMalloc(&g_idx.d_p3tab, bytes));
	launch_p3_build(0, g_idx.fm, k, g_idx.d_p3tab);
	HIP_OK(hipDeviceSynchronize());
	HIP_OK(hipGetLastError());
	g_idx.fm.p3tab = g_idx.d_p3tab;
	g_idx.fm.p3_k = k;
}

// The seeding kernel's k-mer tables (fm_kernels.hip: kmt_build_kernel): the bi-interval of every string of up to K bases, K chosen
// so that the longest table has about as many entries as the index has rows (beyond that a table entry is as cold as an occ block)
// and at most 14 (5.7 GB).  MPIBWA_KMT=<K> chooses K (0 = no tables: every extension through the occ table).
static void maybe_build_kmt()
{
	if (g_idx.d_kmt) { (void)hipFree(g_idx.d_kmt); g_idx.d_kmt = nullptr; g_idx.kmt_bytes = 0; }
	g_idx.fm.kmt = nullptr; g_idx.fm.kmt_k = 0;
	int k = 1;
	while (k < 14 && ((uint64_t)1 << (2 * k)) < g_idx.fm.seq_len) ++k;
	if (const char *e = getenv("MPIBWA_KMT")) k = atoi(e);
	if (k < 1) return;
	if (k > 15) k = 15;
	const size_t bytes = kmt_bytes(k);
	size_t free_b = 0, total_b = 0;
	if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + ((size_t)16 << 30)) return;
	HIP_OK(hipMalloc(&g_idx.d_kmt, bytes));
	launch_kmt_build(0, g_idx.fm, k, g_idx.d_kmt);
	HIP_OK(hipDeviceSynchronize());
	HIP_OK(hipGetLastError());
	g_idx.kmt_bytes = bytes;
	g_idx.fm.kmt = g_idx.d_kmt;
	g_idx.fm.kmt_k = k;
}

// The seeding kernel's own occ table (fm_kernels.hip: occ32_build_kernel), derived on the device from the bwa-format blocks.
static void build_occ32()
{
	if (g_idx.d_occ32) { (void)hipFree(g_idx.d_occ32); g_idx.d_occ32 = nullptr; }
	g_idx.occ32_bytes = occ32_bytes(g_idx.fm.seq_len);
	HIP_OK(hipMalloc(&g_idx.d_occ32, g_idx.occ32_bytes));
	launch_occ32_build(0, g_idx.fm, g_idx.d_occ32);
	HIP_OK(hipDeviceSynchronize());
	HIP_OK(hipGetLastError());
}

static void maybe_expand_sa()
{
	const char *e = getenv("MPIBWA_SA_DENSE");
	if (e && atoi(e) == 0) return;
	size_t need = (size_t)(g_idx.fm.seq_len + 1) * 8, free_b = 0, total_b = 0;
	if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need + (need >> 1) + ((size_t)8 << 30)) return;   // keep room for the batches
	HIP_OK(hipMalloc(&g_idx.d_sa_full, need));
	unsigned long long *d_cnt;
	HIP_OK(hipMalloc(&d_cnt, 64));
	HIP_OK(hipMemset(d_cnt, 0, 64));
	hipEvent_t a, b;
	HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b));
	HIP_OK(hipEventRecord(a, 0));
	launch_sa_expand(0, g_idx.fm, (uint64_t *)g_idx.d_sa_full, d_cnt);
	HIP_OK(hipEventRecord(b, 0));
	HIP_OK(hipEventSynchronize(b));
	HIP_OK(hipGetLastError());
	float ms = 0;
	HIP_OK(hipEventElapsedTime(&ms, a, b));
	g_idx.sa_expand_ms = ms;
	g_idx.sa_full_bytes = need;
	g_idx.fm.sa_full = (const uint64_t *)g_idx.d_sa_full;
	(void)hipFree(d_cnt); (void)hipEventDestroy(a); (void)hipEventDestroy(b);
}

// GPUs this process can see (0 when there is none: callers decide how many ranks share a device)
extern "C" int mi355x_device_count(void)
{
	int n = 0;
	return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

// free / total bytes of the device the index lives on (0 on success), through this library's own HIP runtime
extern "C" int mi355x_device_memory(size_t *free_bytes, size_t *total_bytes)
{
	size_t fr = 0, tot = 0;
	if (g_idx.device >= 0 && hipSetDevice(g_idx.device) != hipSuccess) return -1;
	if (hipMemGetInfo(&fr, &tot) != hipSuccess) return -1;
	if (free_bytes) *free_bytes = fr;
	if (total_bytes) *total_bytes = tot;
	return 0;
}

extern "C" int mi355x_index_alloc(int local_rank, const bwt_t *bwt, const bntseq_t *bns)
{
	require_device(local_rank);
	alloc_index(bwt, bns);   // buffers only: fill them with mi355x_index_d2d / ncclBroadcast, then mi355x_index_commit()
	return 0;
}

extern "C" int mi355x_index_upload(int local_rank, const bwt_t *bwt, const bntseq_t *bns, const uint8_t *pac)
{
	require_device(local_rank);
	alloc_index(bwt, bns);
	HIP_OK(hipMemcpy(g_idx.d_blk, bwt->bwt, (size_t)bwt->bwt_size * 4, hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(g_idx.d_sa, bwt->sa, g_idx.sa_bytes, hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(g_idx.d_pac, pac, (size_t)bns->l_pac / 4 + 1, hipMemcpyHostToDevice));
	const bool dbg = getenv("MPIBWA_DEBUG") != nullptr;
	if (dbg) fprintf(stderr, "[upload] copied\n");
	build_occ32();
	if (dbg) fprintf(stderr, "[upload] occ32 built\n");
	maybe_expand_sa();
	if (dbg) fprintf(stderr, "[upload] SA expanded\n");
	maybe_build_p3();
	maybe_build_kmt();
	if (dbg) fprintf(stderr, "[upload] jump table and k-mer tables built\n");
	g_idx.ready = true;
	return 0;
}

extern "C" int mi355x_sa_batch(int n, const uint64_t *k, uint64_t *sa_out, double *kernel_ms, uint64_t *algo_bytes);
extern "C" int mi355x_index_buffers(void **d_bwt, size_t *bwt_bytes, void **d_sa, size_t *sa_bytes, void **d_pac,
                                    size_t *pac_bytes)
{
	if (!g_idx.d_blk) return -1;
	*d_bwt = g_idx.d_blk; *bwt_bytes = g_idx.blk_bytes;
	*d_sa = g_idx.d_sa; *sa_bytes = g_idx.sa_bytes;
	*d_pac = g_idx.d_pac; *pac_bytes = g_idx.pac_bytes;
	return 0;
}

// Copy `bytes` from a device pointer owned by the caller (e.g. a torch tensor that has just received an RCCL broadcast)
// into index buffer `which` (0 = occ blocks, 1 = sampled SA, 2 = pac), or out of it when to_index == 0.
extern "C" int mi355x_index_d2d(int which, void *ext, size_t bytes, int to_index)
{
	if (!g_idx.d_blk) return -1;
	void *buf = which == 0 ? g_idx.d_blk : which == 1 ? g_idx.d_sa : g_idx.d_pac;
	size_t cap = which == 0 ? g_idx.blk_bytes : which == 1 ? g_idx.sa_bytes : g_idx.pac_bytes;
	if (bytes > cap) return -2;
	HIP_OK(hipMemcpy(to_index ? buf : ext, to_index ? ext : buf, bytes, hipMemcpyDeviceToDevice));
	return 0;
}
// after the three buffers have been filled by broadcast: expand the dense SA and mark the index usable
extern "C" int mi355x_index_commit(void)
{
	if (!g_idx.d_blk) return -1;
	build_occ32();
	maybe_expand_sa();
	maybe_build_p3();
	maybe_build_kmt();
	g_idx.ready = true;
	return 0;
}

extern "C" void mi355x_finalize(void)
{
	std::lock_guard<std::recursive_mutex> lk(index_mutex());
	no_calls_in_flight("mi355x_finalize");
	if (g_idx.d_blk) { (void)hipFree(g_idx.d_blk); (void)hipFree(g_idx.d_sa); (void)hipFree(g_idx.d_pac); }
	if (g_idx.d_sa_full) (void)hipFree(g_idx.d_sa_full);
	if (g_idx.d_p3tab) (void)hipFree(g_idx.d_p3tab);
	if (g_idx.d_occ32) (void)hipFree(g_idx.d_occ32);
	if (g_idx.d_kmt) (void)hipFree(g_idx.d_kmt);
	g_idx = DevIndex();
	release_idle_work_buffers();   // a process that is done with this index gives the HBM of its call contexts back too
}

namespace mbw {

static void need_index()
{
	if (!g_idx.ready) die("index not resident on the device: call mi355x_index_upload() first");
}

struct Timer {
	hipEvent_t a, b;
	Timer() { HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b)); }
	~Timer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
	void start(hipStream_t s) { HIP_OK(hipEventRecord(a, s)); }
	double stop(hipStream_t s)
	{
		HIP_OK(hipEventRecord(b, s));
		HIP_OK(hipEventSynchronize(b));
		float ms = 0;
		HIP_OK(hipEventElapsedTime(&ms, a, b));
		return ms;
	}
};

SmemParams smem_params(const mem_opt_t *opt)
{
	SmemParams sp;
	sp.min_seed_len = opt->min_seed_len;
	sp.split_len = (int)(opt->min_seed_len * opt->split_factor + .499);   // src/bwamem.c:118
	sp.split_width = opt->split_width;
	if (opt->split_width < 0 || opt->split_width >= 65535) die("split_width %d: the seeding kernel packs re-seed requests into 16 bits", opt->split_width);
	sp.max_mem_intv = (int)opt->max_mem_intv;
	return sp;
}

// band clamp of src/ksw.c:395-407 (host side, double arithmetic as in the reference)
int clamp_band(const mem_opt_t *opt, int qlen, int w, int end_bonus)
{
	int mx = 0;
	for (int i = 0; i < 25; ++i) mx = std::max(mx, (int)opt->mat[i]);
	int max_ins = (int)((double)(qlen * mx + end_bonus - opt->o_ins) / opt->e_ins + 1.);
	int max_del = (int)((double)(qlen * mx + end_bonus - opt->o_del) / opt->e_del + 1.);
	w = std::min(w, std::max(max_ins, 1));
	w = std::min(w, std::max(max_del, 1));
	return w;
}

} // namespace mbw

namespace mbw {
bool pair_params(const mem_opt_t *opt, int64_t l_pac, const mem_pestat_t pes[4], int64_t n_processed, int max_len, PairParams &pp, size_t *n_tab_)
{
	memset(&pp, 0, sizeof pp);
	pp.l_pac = l_pac; pp.a = opt->a; pp.b = opt->b; pp.pen_unpaired = opt->pen_unpaired; pp.min_seed_len = opt->min_seed_len; pp.w = opt->w;
	pp.o_del = opt->o_del; pp.e_del = opt->e_del; pp.o_ins = opt->o_ins; pp.e_ins = opt->e_ins;
	pp.max_chain_gap = opt->max_chain_gap; pp.mask_level_redun = opt->mask_level_redun; pp.mask_level = opt->mask_level;
	pp.XA_drop_ratio = opt->XA_drop_ratio; pp.T = opt->T; pp.max_matesw = opt->max_matesw; pp.id0 = (uint64_t)(n_processed >> 1);
	for (int v = 0; v < 40; ++v) pp.lnq[v] = (int)(4.343 * log(v + 1) + .499);
	pp.no_rescue = ((opt->flag & MEM_F_NO_RESCUE) || opt->max_matesw <= 0) ? 1 : 0;
	bool usable = true;
	size_t n_tab = 0;
	for (int d = 0; d < 4; ++d) {
		pp.low[d] = pes[d].low; pp.high[d] = pes[d].high; pp.failed[d] = pes[d].failed ? 1 : 0;
		pp.tab_off[d] = (int)n_tab;
		if (!pes[d].failed) {
			// (a degenerate distribution — std 0, as a user's -I can give — makes the pair score NaN / infinite for some distances,
			// and the conversion of those to int is the one place where the host's and the device's arithmetic differ: host path)
			if (!(pes[d].std > 0) || !std::isfinite(pes[d].avg) || !std::isfinite(pes[d].std)) usable = false;
			if (pes[d].high < pes[d].low || (int64_t)pes[d].high - pes[d].low > (1 << 20)) usable = false;
			else n_tab += (size_t)(pes[d].high - pes[d].low + 1);
		}
	}
	pp.ltab_n = 4 * max_len + 256;
	*n_tab_ = n_tab;
	return usable;
}
void pair_tables(const mem_opt_t *opt, const mem_pestat_t pes[4], const PairParams &pp, size_t n_tab, double *tab)
{
	for (int d = 0; d < 4; ++d)
		if (!pes[d].failed)
			for (int64_t dist = pes[d].low; dist <= pes[d].high; ++dist) {   // src/bwamem_pair.c:218-219, the double part of q
				const double ns = (dist - pes[d].avg) / pes[d].std;
				tab[pp.tab_off[d] + (dist - pes[d].low)] = .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)) * opt->a;
			}
	double *ltab = tab + n_tab;
	ltab[0] = 1.;
	for (int l = 1; l < pp.ltab_n; ++l) ltab[l] = l < opt->mapQ_coef_len ? 1. : opt->mapQ_coef_fac / log(l);   // src/bwamem.c:964
}
} // namespace mbw

// Stage entry of pair_simple_kernel (pair_kernel.hip) for parity tests: n_pairs pairs given by the regions of their two ends
// (regs: PR_MAXREG DevReg records per read, n_regs per read) as they stand after phase 1; status[k] = 1: decided — desc[2k], desc[2k+1]
// (SamDesc) and req[2k], req[2k+1] (AlnReq) are what mem_sam_pe's paired branch reports; else the code of the test that sent the pair to
// the host.  Returns 0, or -1 when the insert-size statistics are not usable by the kernel.
extern "C" int mi355x_pair_maxreg(void) { return PR_MAXREG; }
extern "C" int mi355x_pair_batch(const mem_opt_t *opt, const bntseq_t *bns, const mem_pestat_t pes[4], int64_t n_processed, int n_pairs,
                                 const void *regs, const int *n_regs, int max_len, uint8_t *status, void *desc, void *req)
{
	int nd = 0;
	if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) die("no HIP device visible (no CPU fallback)");
	if (n_pairs <= 0) return 0;
	PairParams pp;
	size_t n_tab = 0;
	if (!pair_params(opt, bns->l_pac, pes, n_processed, max_len, pp, &n_tab)) return -1;
	std::vector<double> tab(n_tab + (size_t)pp.ltab_n);
	pair_tables(opt, pes, pp, n_tab, tab.data());
	std::vector<int64_t> ann_off(bns->n_seqs + 1);
	std::vector<uint8_t> ann_alt(bns->n_seqs + 1, 0), ok((size_t)n_pairs, 1);
	for (int k = 0; k < bns->n_seqs; ++k) { ann_off[k] = bns->anns[k].offset; ann_alt[k] = bns->anns[k].is_alt ? 1 : 0; }
	ann_off[bns->n_seqs] = bns->l_pac;
	const size_t n = (size_t)2 * n_pairs;
	DevReg *d_first; int *d_nf; uint8_t *d_ok, *d_aa, *d_st; int64_t *d_ao; double *d_tab; AlnReq *d_rq; SamDesc *d_ds;
	HIP_OK(hipMalloc(&d_first, n * PR_MAXREG * sizeof(DevReg))); HIP_OK(hipMalloc(&d_nf, n * 4)); HIP_OK(hipMalloc(&d_ok, n_pairs));
	HIP_OK(hipMalloc(&d_aa, ann_alt.size())); HIP_OK(hipMalloc(&d_st, n_pairs)); HIP_OK(hipMalloc(&d_ao, ann_off.size() * 8));
	HIP_OK(hipMalloc(&d_tab, tab.size() * 8)); HIP_OK(hipMalloc(&d_rq, n * sizeof(AlnReq))); HIP_OK(hipMalloc(&d_ds, n * sizeof(SamDesc)));
	HIP_OK(hipMemcpy(d_first, regs, n * PR_MAXREG * sizeof(DevReg), hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_nf, n_regs, n * 4, hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_ok, ok.data(), n_pairs, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_aa, ann_alt.data(), ann_alt.size(), hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_ao, ann_off.data(), ann_off.size() * 8, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
	launch_pair_simple(0, pp, n_pairs, d_first, d_nf, d_ok, d_ao, d_aa, d_tab, d_tab + n_tab, d_st, d_rq, d_ds);
	HIP_OK(hipDeviceSynchronize());
	HIP_OK(hipGetLastError());
	HIP_OK(hipMemcpy(status, d_st, n_pairs, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(desc, d_ds, n * sizeof(SamDesc), hipMemcpyDeviceToHost));
	HIP_OK(hipMemcpy(req, d_rq, n * sizeof(AlnReq), hipMemcpyDeviceToHost));
	(void)hipFree(d_first); (void)hipFree(d_nf); (void)hipFree(d_ok); (void)hipFree(d_aa); (void)hipFree(d_st); (void)hipFree(d_ao); (void)hipFree(d_tab);
	(void)hipFree(d_rq); (void)hipFree(d_ds);
	return 0;
}

extern "C" int mi355x_smem_batch(const mem_opt_t *opt, int n, const uint8_t *seqs, const int64_t *off, int cap,
                                 uint64_t *intv_out, int *n_out, double *kernel_ms, uint64_t *algo_bytes)
{
	need_index();
	if (n <= 0) return 0;
	hipStream_t st = 0;
	int max_len = 0;
	std::vector<int64_t> poff(n + 1);   // 16-byte aligned slots, as the kernel expects
	std::vector<int> lens(n);
	poff[0] = 0;
	for (int i = 0; i < n; ++i) {
		lens[i] = (int)(off[i + 1] - off[i]);
		max_len = std::max(max_len, lens[i]);
		poff[i + 1] = poff[i] + ((lens[i] + 15) & ~15);
	}
	size_t total = off[n];
	std::vector<uint8_t> packed(poff[n] + 16, 0);
	for (int i = 0; i < n; ++i) memcpy(packed.data() + poff[i], seqs + off[i], lens[i]);
	uint8_t *d_seq; int64_t *d_off; uint64_t *d_out; int *d_nout, *d_len; unsigned long long *d_cnt; void *d_scr;
	HIP_OK(hipMalloc(&d_seq, packed.size()));
	HIP_OK(hipMalloc(&d_len, (size_t)n * 4));
	HIP_OK(hipMalloc(&d_off, (size_t)(n + 1) * 8));
	HIP_OK(hipMalloc(&d_out, (size_t)n * cap * 32));
	HIP_OK(hipMalloc(&d_nout, (size_t)n * 4));
	HIP_OK(hipMalloc(&d_cnt, 256));
	size_t per_quad = 0;
	int n_quads = smem_grid_quads(max_len, &per_quad);
	HIP_OK(hipMalloc(&d_scr, per_quad * n_quads));
	HIP_OK(hipMemcpy(d_seq, packed.data(), packed.size(), hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_off, poff.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_len, lens.data(), (size_t)n * 4, hipMemcpyHostToDevice));
	HIP_OK(hipMemset(d_cnt, 0, 256));
	HIP_OK(hipMemset(d_nout, 0, (size_t)n * 4));
	Timer tm;
	tm.start(st);
	const char *ce = getenv("MPIBWA_SMEM_COUNT");   // "0": the production variant of passes 1-2 (no block counting; *algo_bytes = 0)
	const bool count_blocks = !(ce && atoi(ce) == 0);
	launch_smem(st, g_idx.fm, smem_params(opt), n, d_seq, d_off, d_len, cap, d_out, d_nout, max_len, d_cnt, d_scr, per_quad, n_quads, count_blocks);
	double ms = tm.stop(st);
	HIP_OK(hipGetLastError());
	unsigned long long cnt[32];
	HIP_OK(hipMemcpy(cnt, d_cnt, 256, hipMemcpyDeviceToHost));
	HIP_OK(hipMemcpy(n_out, d_nout, (size_t)n * 4, hipMemcpyDeviceToHost));
	HIP_OK(hipMemcpy(intv_out, d_out, (size_t)n * cap * 32, hipMemcpyDeviceToHost));
	(void)hipFree(d_seq); (void)hipFree(d_off); (void)hipFree(d_out); (void)hipFree(d_nout); (void)hipFree(d_cnt);
	(void)hipFree(d_scr); (void)hipFree(d_len);
	// order by info (the reference sorts with an unstable introsort keyed on info only, src/bwamem.c:161;
	// equal keys are identical records, so any order of ties is the same byte sequence)
	uint64_t n_intv = 0;
	for (int i = 0; i < n; ++i) {
		int m = std::min(n_out[i], cap);
		Intv *a = (Intv *)(intv_out + (size_t)i * cap * 4);
		std::sort(a, a + m, [](const Intv &x, const Intv &y) { return x.info < y.info; });
		n_intv += m;
	}
	if (kernel_ms) *kernel_ms = ms;
	if (algo_bytes) *algo_bytes = count_blocks ? cnt[1] * 64 + total + n_intv * 32 : 0;   // 64 B per occ block + read + output
	return cnt[2] ? -1 : 0;
}

// dense != 0: answer from the expanded table (fails if it is absent); dense == 0: LF walk on the sampled SA
extern "C" int mi355x_sa_batch2(int n, const uint64_t *k, uint64_t *sa_out, double *kernel_ms, int dense)
{
	need_index();
	if (!dense) return mi355x_sa_batch(n, k, sa_out, kernel_ms, nullptr);
	if (!g_idx.fm.sa_full) return -1;
	if (n <= 0) return 0;
	uint64_t *d_k, *d_o;
	HIP_OK(hipMalloc(&d_k, (size_t)n * 8)); HIP_OK(hipMalloc(&d_o, (size_t)n * 8));
	HIP_OK(hipMemcpy(d_k, k, (size_t)n * 8, hipMemcpyHostToDevice));
	Timer tm;
	tm.start(0);
	launch_sa_dense(0, g_idx.fm, n, d_k, d_o);
	double ms = tm.stop(0);
	HIP_OK(hipGetLastError());
	HIP_OK(hipMemcpy(sa_out, d_o, (size_t)n * 8, hipMemcpyDeviceToHost));
	(void)hipFree(d_k); (void)hipFree(d_o);
	if (kernel_ms) *kernel_ms = ms;
	return 0;
}
extern "C" double mi355x_sa_dense_info(size_t *bytes) { if (bytes) *bytes = g_idx.sa_full_bytes; return g_idx.sa_expand_ms; }

extern "C" int mi355x_sa_batch(int n, const uint64_t *k, uint64_t *sa_out, double *kernel_ms, uint64_t *algo_bytes)
{
	need_index();
	if (n <= 0) return 0;
	hipStream_t st = 0;
	uint64_t *d_k, *d_o; unsigned long long *d_cnt;
	HIP_OK(hipMalloc(&d_k, (size_t)n * 8));
	HIP_OK(hipMalloc(&d_o, (size_t)n * 8));
	HIP_OK(hipMalloc(&d_cnt, 64));
	HIP_OK(hipMemcpy(d_k, k, (size_t)n * 8, hipMemcpyHostToDevice));
	HIP_OK(hipMemset(d_cnt, 0, 64));
	Timer tm;
	tm.start(st);
	launch_sa(st, g_idx.fm, n, d_k, d_o, d_cnt);
	double ms = tm.stop(st);
	HIP_OK(hipGetLastError());
	unsigned long long cnt[8];
	HIP_OK(hipMemcpy(cnt, d_cnt, 64, hipMemcpyDeviceToHost));
	HIP_OK(hipMemcpy(sa_out, d_o, (size_t)n * 8, hipMemcpyDeviceToHost));
	(void)hipFree(d_k); (void)hipFree(d_o); (void)hipFree(d_cnt);
	if (kernel_ms) *kernel_ms = ms;
	if (algo_bytes) *algo_bytes = cnt[1] * 64 + (uint64_t)n * 8;   // 64 B per LF step + the sampled SA word
	return 0;
}

extern "C" int mi355x_extend_batch(const mem_opt_t *opt, int n, const uint8_t *q, const int64_t *qoff, const uint8_t *t,
                                   const int64_t *toff, const int *w, const int *h0, const int *end_bonus, int *out6,
                                   double *kernel_ms, uint64_t *cells)
{
	int nd = 0;
	if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) die("no HIP device visible (no CPU fallback)");
	if (n <= 0) return 0;
	hipStream_t st = 0;
	std::vector<int> wc(n);
	int max_qlen = 0;
	for (int i = 0; i < n; ++i) {
		int ql = (int)(qoff[i + 1] - qoff[i]);
		max_qlen = std::max(max_qlen, ql);
		wc[i] = clamp_band(opt, ql, w[i], end_bonus[i]);
	}
	uint8_t *d_q, *d_t; int64_t *d_qo, *d_to; int *d_w, *d_h0, *d_out; unsigned long long *d_cells;
	HIP_OK(hipMalloc(&d_q, qoff[n] + 16)); HIP_OK(hipMalloc(&d_t, toff[n] + 16));
	HIP_OK(hipMalloc(&d_qo, (size_t)(n + 1) * 8)); HIP_OK(hipMalloc(&d_to, (size_t)(n + 1) * 8));
	HIP_OK(hipMalloc(&d_w, (size_t)n * 4)); HIP_OK(hipMalloc(&d_h0, (size_t)n * 4));
	HIP_OK(hipMalloc(&d_out, (size_t)n * 24)); HIP_OK(hipMalloc(&d_cells, 8));
	HIP_OK(hipMemcpy(d_q, q, qoff[n], hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_t, t, toff[n], hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_qo, qoff, (size_t)(n + 1) * 8, hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_to, toff, (size_t)(n + 1) * 8, hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_w, wc.data(), (size_t)n * 4, hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_h0, h0, (size_t)n * 4, hipMemcpyHostToDevice));
	HIP_OK(hipMemset(d_cells, 0, 8));
	ExtParams ep;
	memcpy(ep.mat, opt->mat, 25);
	ep.o_del = opt->o_del; ep.e_del = opt->e_del; ep.o_ins = opt->o_ins; ep.e_ins = opt->e_ins; ep.zdrop = opt->zdrop;
	Timer tm;
	tm.start(st);
	launch_extend(st, ep, n, d_q, d_qo, d_t, d_to, d_w, d_h0, nullptr, d_out, d_cells, max_qlen);
	double ms = tm.stop(st);
	HIP_OK(hipGetLastError());
	unsigned long long c = 0;
	HIP_OK(hipMemcpy(&c, d_cells, 8, hipMemcpyDeviceToHost));
	HIP_OK(hipMemcpy(out6, d_out, (size_t)n * 24, hipMemcpyDeviceToHost));
	(void)hipFree(d_q); (void)hipFree(d_t); (void)hipFree(d_qo); (void)hipFree(d_to); (void)hipFree(d_w);
	(void)hipFree(d_h0); (void)hipFree(d_out); (void)hipFree(d_cells);
	if (kernel_ms) *kernel_ms = ms;
	if (cells) *cells = c;
	return 0;
}

namespace mbw {
MswParams msw_params(const mem_opt_t *opt, int64_t l_pac)
{
	MswParams P;
	P.l_pac = l_pac;
	int8_t mn = 127, mx = 0;   // initial values of the reference's scan (src/ksw.c:83-87)
	for (int i = 0; i < 25; ++i) {
		if (opt->mat[i] < mn) mn = opt->mat[i];
		if (opt->mat[i] > mx) mx = opt->mat[i];
	}
	for (int t = 0; t < 4; ++t) {
		P.slo[t] = 0;
		for (int q = 0; q < 4; ++q) P.slo[t] |= (uint32_t)(uint8_t)opt->mat[t * 5 + q] << (8 * q);
		P.s4[t] = opt->mat[t * 5 + 4];
	}
	P.o_del = opt->o_del; P.e_del = opt->e_del; P.o_ins = opt->o_ins; P.e_ins = opt->e_ins;
	P.a = opt->a; P.min_seed_len = opt->min_seed_len;
	P.max_sc = mx > 0 ? mx : 1;
	P.shift = (uint8_t)(256 - (uint8_t)mn);
	return P;
}
}

// Stage-level entry point of the mate-rescue alignment (tests, micro-benchmarks): n_req windows [rb,re) of the packed
// reference `pac` (doubled coordinate, 2 bits per base, l_pac bases) against reads of a batch given as nt4 codes.
// out8 per request: score, te, qe, score2, te2, tb, qb, flags — kswr_t of ksw_align2 with mem_matesw's flags.
extern "C" int mi355x_matesw_batch(const mem_opt_t *opt, int64_t l_pac, const uint8_t *pac, int n_reads, const uint8_t *reads, const int64_t *off,
                                   int n_req, const int64_t *rb, const int64_t *re, const int *read, const int *is_rev, int *out8,
                                   double *kernel_ms)
{
	int nd = 0;
	if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) die("no HIP device visible (no CPU fallback)");
	if (n_req <= 0) return 0;
	hipStream_t st = 0;
	// reads go into 16-byte slots as in the pipeline
	std::vector<int64_t> slot(n_reads + 1);
	std::vector<int> lens(n_reads);
	int max_len = 0;
	slot[0] = 0;
	for (int i = 0; i < n_reads; ++i) {
		lens[i] = (int)(off[i + 1] - off[i]);
		slot[i + 1] = slot[i] + ((lens[i] + 15) & ~15);
		max_len = std::max(max_len, lens[i]);
	}
	if ((int64_t)max_len * opt->a >= 8192 || msw_lds_bytes(max_len) > 160 * 1024) die("mate-rescue kernel: reads too long for the device path");
	std::vector<uint8_t> flat(slot[n_reads] + 16, 4);
	for (int i = 0; i < n_reads; ++i) memcpy(flat.data() + slot[i], reads + off[i], lens[i]);
	std::vector<MswReq> rq(n_req);
	int max_t = 1;
	for (int i = 0; i < n_req; ++i) {
		rq[i].rb = rb[i]; rq[i].re = re[i]; rq[i].read = read[i]; rq[i].is_rev = is_rev[i];
		if (re[i] < rb[i] || re[i] > 2 * l_pac || rb[i] < 0 || read[i] < 0 || read[i] >= n_reads) die("mi355x_matesw_batch: bad request %d", i);
		max_t = std::max(max_t, (int)(re[i] - rb[i]));
	}
	uint8_t *d_seq, *d_pac; int64_t *d_off; int *d_len; MswReq *d_req; MswRes *d_res; uint16_t *d_rows;
	HIP_OK(hipMalloc(&d_seq, flat.size())); HIP_OK(hipMalloc(&d_pac, l_pac / 4 + 16));
	HIP_OK(hipMalloc(&d_off, (size_t)(n_reads + 1) * 8)); HIP_OK(hipMalloc(&d_len, (size_t)n_reads * 4));
	HIP_OK(hipMalloc(&d_req, (size_t)n_req * sizeof(MswReq))); HIP_OK(hipMalloc(&d_res, (size_t)n_req * sizeof(MswRes)));
	HIP_OK(hipMalloc(&d_rows, (size_t)n_req * max_t * 2));
	HIP_OK(hipMemcpy(d_seq, flat.data(), flat.size(), hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_pac, pac, l_pac / 4 + 1, hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_off, slot.data(), (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_len, lens.data(), (size_t)n_reads * 4, hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_req, rq.data(), (size_t)n_req * sizeof(MswReq), hipMemcpyHostToDevice));
	Timer tm;
	tm.start(st);
	std::vector<int> h_list(2 * (size_t)n_req + 16);
	int *d_list;
	HIP_OK(hipMalloc(&d_list, (2 * (size_t)n_req + 16) * sizeof(int)));
	launch_msw(st, msw_params(opt, l_pac), n_req, d_req, d_seq, d_off, d_len, d_pac, d_res, d_rows, max_len, rq.data(), lens.data(), h_list.data(), d_list);
	double ms = tm.stop(st);
	HIP_OK(hipGetLastError());
	static_assert(sizeof(MswRes) == 32, "MswRes layout");
	HIP_OK(hipMemcpy(out8, d_res, (size_t)n_req * sizeof(MswRes), hipMemcpyDeviceToHost));
	(void)hipFree(d_seq); (void)hipFree(d_pac); (void)hipFree(d_off); (void)hipFree(d_len); (void)hipFree(d_req); (void)hipFree(d_res);
	(void)hipFree(d_rows); (void)hipFree(d_list);
	if (kernel_ms) *kernel_ms = ms;
	return 0;
}

// Stage-level entry point of the chaining stage (tests): seeds of n_reads reads -> filtered chains, computed by
// chain_kernel (which = 0) or by the host path (which = 1).  Per read r the output is a run of int64 starting at
// out[out_off[r]]: n_chains (-1 = the device declines the read), then per chain
//   rid, n_seeds, far_beg, far_end, rmax0, rmax1, frac_rep (float bits), and n_seeds x (rbeg, qbeg, len) in visiting order.
// out must hold 1 + 7 * 9... entries per read in the worst case; the caller sizes it as n_reads + 8 * total_seeds + ...
extern "C" int64_t mi355x_chain_batch(const mem_opt_t *opt, const bntseq_t *bns, int n_reads, const int *lens, const int *l_rep,
                                      const int64_t *seed_off, const uint64_t *rbeg, const int32_t *qbeg_len, int which, int64_t *out,
                                      int64_t out_cap, int64_t *out_off)
{
	using namespace mbw;
	const int64_t S = seed_off[n_reads];
	int max_len = 0;
	for (int i = 0; i < n_reads; ++i) max_len = std::max(max_len, lens[i]);
	const int TS = max_len + 2;
	std::vector<int> tab(6 * TS);
	for (int l = 0; l < TS; ++l) {
		tab[l] = cal_max_gap(opt, l);
		tab[TS + l] = tab[2 * TS + l] = tab[3 * TS + l] = tab[4 * TS + l] = 0;
		const double min_l = opt->min_chain_weight ? 1.1f * opt->min_chain_weight : 5.5f * log(l > 0 ? l : 1);
		tab[5 * TS + l] = (l > 0 && min_l > 0.05f * l) ? 1 : 0;
	}
	std::vector<int> nseeds(n_reads);
	for (int i = 0; i < n_reads; ++i) nseeds[i] = (int)(seed_off[i + 1] - seed_off[i]);
	std::vector<int> nch(n_reads, 0);
	std::vector<DevChain> chains(std::max<int64_t>(S, 1));
	std::vector<DevSeed> seeds(std::max<int64_t>(S, 1));
	if (which == 0) {
		int nd = 0;
		if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) die("no HIP device visible (no CPU fallback)");
		std::vector<int64_t> ann_off(bns->n_seqs + 1);
		std::vector<uint8_t> ann_alt(bns->n_seqs + 1, 0);
		for (int k = 0; k < bns->n_seqs; ++k) { ann_off[k] = bns->anns[k].offset; ann_alt[k] = bns->anns[k].is_alt ? 1 : 0; }
		ann_off[bns->n_seqs] = bns->l_pac;
		int *d_len, *d_ns, *d_lrep, *d_tab, *d_nch; int64_t *d_so, *d_ao; uint64_t *d_sa; int32_t *d_qbl; uint8_t *d_aa;
		DevChain *d_ch; DevSeed *d_sd; unsigned int *d_srt;
		HIP_OK(hipMalloc(&d_len, n_reads * 4 + 4)); HIP_OK(hipMalloc(&d_ns, n_reads * 4 + 4)); HIP_OK(hipMalloc(&d_lrep, n_reads * 4 + 4));
		HIP_OK(hipMalloc(&d_tab, tab.size() * 4)); HIP_OK(hipMalloc(&d_nch, n_reads * 4 + 4)); HIP_OK(hipMalloc(&d_so, (n_reads + 1) * 8));
		HIP_OK(hipMalloc(&d_ao, ann_off.size() * 8)); HIP_OK(hipMalloc(&d_aa, ann_alt.size())); HIP_OK(hipMalloc(&d_sa, S * 8 + 8));
		HIP_OK(hipMalloc(&d_qbl, S * 8 + 8)); HIP_OK(hipMalloc(&d_ch, (S + 1) * sizeof(DevChain))); HIP_OK(hipMalloc(&d_sd, (S + 1) * sizeof(DevSeed)));
		HIP_OK(hipMalloc(&d_srt, (S + 1) * 4));
		HIP_OK(hipMemcpy(d_len, lens, n_reads * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_ns, nseeds.data(), n_reads * 4, hipMemcpyHostToDevice));
		HIP_OK(hipMemcpy(d_lrep, l_rep, n_reads * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
		HIP_OK(hipMemcpy(d_so, seed_off, (n_reads + 1) * 8, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_ao, ann_off.data(), ann_off.size() * 8, hipMemcpyHostToDevice));
		HIP_OK(hipMemcpy(d_aa, ann_alt.data(), ann_alt.size(), hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_sa, rbeg, S * 8, hipMemcpyHostToDevice));
		HIP_OK(hipMemcpy(d_qbl, qbeg_len, S * 8, hipMemcpyHostToDevice));
		ChainParams kp;
		kp.l_pac = bns->l_pac; kp.w = opt->w; kp.max_chain_gap = opt->max_chain_gap; kp.min_chain_weight = opt->min_chain_weight;
		kp.min_seed_len = opt->min_seed_len; kp.max_chain_extend = opt->max_chain_extend; kp.mask_level = opt->mask_level; kp.drop_ratio = opt->drop_ratio;
		void *d_gen = nullptr;
		const int gen_cap = std::min(n_reads, 4096);
		HIP_OK(hipMalloc(&d_gen, chain_general_bytes(gen_cap, n_reads)));
		launch_chain(0, kp, n_reads, d_len, d_ns, d_lrep, d_so, d_sa, d_qbl, d_ao, d_aa, bns->n_seqs, d_tab, TS, d_ch, d_sd, d_srt, d_nch, d_gen, gen_cap);
		HIP_OK(hipDeviceSynchronize());
		(void)hipFree(d_gen);
		HIP_OK(hipMemcpy(nch.data(), d_nch, n_reads * 4, hipMemcpyDeviceToHost));
		HIP_OK(hipMemcpy((void *)chains.data(), d_ch, S * sizeof(DevChain), hipMemcpyDeviceToHost));
		HIP_OK(hipMemcpy((void *)seeds.data(), d_sd, S * sizeof(DevSeed), hipMemcpyDeviceToHost));
		(void)hipFree(d_len); (void)hipFree(d_ns); (void)hipFree(d_lrep); (void)hipFree(d_tab); (void)hipFree(d_nch); (void)hipFree(d_so); (void)hipFree(d_ao);
		(void)hipFree(d_aa); (void)hipFree(d_sa); (void)hipFree(d_qbl); (void)hipFree(d_ch); (void)hipFree(d_sd); (void)hipFree(d_srt);
	} else {
		ChainScratch scr;
		std::vector<HSeed> hs;
		std::vector<HChain *> ch;
		std::vector<uint64_t> key;
		for (int i = 0; i < n_reads; ++i) {
			const int ns = nseeds[i];
			if (ns == 0) continue;
			hs.resize(ns);
			for (int k = 0; k < ns; ++k) {
				const int64_t so = seed_off[i] + k;
				hs[k].rbeg = (int64_t)rbeg[so]; hs[k].qbeg = qbeg_len[2 * so]; hs[k].len = hs[k].score = qbeg_len[2 * so + 1];
			}
			chains_from_seeds(opt, bns, lens[i], hs.data(), ns, l_rep[i], scr, ch);
			chain_filter(opt, scr, ch);
			int64_t cur = seed_off[i];
			int c = 0;
			for (const HChain *cp : ch) {
				DevChain &d = chains[seed_off[i] + c];
				pack_chain_for_device(bns, *cp, lens[i], tab.data(), key, d, seeds.data() + cur);
				d.seed_beg = (int)cur;
				cur += d.n_seeds;
				++c;
			}
			nch[i] = c;
		}
	}
	int64_t at = 0;
	for (int i = 0; i < n_reads; ++i) {
		out_off[i] = at;
		if (at + 1 > out_cap) return -1;
		out[at++] = nch[i];
		for (int c = 0; c < nch[i]; ++c) {
			const DevChain &d = chains[seed_off[i] + c];
			if (at + 7 + 3 * (int64_t)d.n_seeds > out_cap) return -1;
			uint32_t fb;
			memcpy(&fb, &d.frac_rep, 4);
			out[at++] = d.rid; out[at++] = d.n_seeds; out[at++] = d.far_beg; out[at++] = d.far_end; out[at++] = d.rmax0; out[at++] = d.rmax1; out[at++] = fb;
			for (int k = 0; k < d.n_seeds; ++k) {
				const DevSeed &s = seeds[d.seed_beg + k];
				out[at++] = s.rbeg; out[at++] = s.qbeg; out[at++] = s.len;
			}
		}
	}
	out_off[n_reads] = at;
	return at;
}

// Stage-level entry point of the CIGAR / MD / NM kernel (tests): n_req regions of reads of a batch against windows of the
// packed reference `pac`, through mem_reg2aln's band-doubling loop (src/bwamem.c:1106-1122) exactly as the SAM stage asks
// for them.  which = 0: the product's dispatch (no-DP / narrow band / full size); 1: DP requests straight to the full-size
// instantiation.  out_hdr5 per request: score, NM, n_cigar, md_len, flags.
extern "C" int mi355x_global_batch(const mem_opt_t *opt, int64_t l_pac, const uint8_t *pac, int n_reads, const uint8_t *reads,
                                   const int64_t *off, int n_req, const int64_t *rb, const int64_t *re, const int *read,
                                   const int *qb, const int *qe, const int *w, const int *truesc, int which,
                                   int *out_hdr5, uint32_t *cigar_out, int cigar_cap, char *md_out, int md_cap, double *kernel_ms)
{
	int nd = 0;
	if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) die("no HIP device visible (no CPU fallback)");
	if (n_req <= 0) return 0;
	hipStream_t st = 0;
	std::vector<int64_t> slot(n_reads + 1);
	int max_len = 0;
	slot[0] = 0;
	for (int i = 0; i < n_reads; ++i) {
		const int len = (int)(off[i + 1] - off[i]);
		slot[i + 1] = slot[i] + ((len + 15) & ~15);
		max_len = std::max(max_len, len);
	}
	std::vector<uint8_t> flat(slot[n_reads] + 16, 4);
	for (int i = 0; i < n_reads; ++i) memcpy(flat.data() + slot[i], reads + off[i], (size_t)(off[i + 1] - off[i]));
	std::vector<AlnReq> rq(n_req);
	for (int i = 0; i < n_req; ++i) {
		if (read[i] < 0 || read[i] >= n_reads || rb[i] < 0 || re[i] > 2 * l_pac) die("mi355x_global_batch: bad request %d", i);
		rq[i].rb = rb[i]; rq[i].re = re[i]; rq[i].read = read[i]; rq[i].qb = qb[i]; rq[i].qe = qe[i]; rq[i].w2 = w[i]; rq[i].truesc = truesc[i];
		rq[i].pad = 0;
	}
	std::vector<int> gaptab(max_len + 2);
	for (int l = 0; l <= max_len + 1; ++l) {   // max_gap of bwa_gen_cigar2 (src/bwa.c:155-158), as pipeline.hip tabulates it
		int max_ins = (int)((double)(((l + 1) >> 1) * opt->mat[0] - opt->o_ins) / opt->e_ins + 1.);
		int max_del = (int)((double)(((l + 1) >> 1) * opt->mat[0] - opt->o_del) / opt->e_del + 1.);
		int g = max_ins > max_del ? max_ins : max_del;
		gaptab[l] = g > 1 ? g : 1;
	}
	const size_t pool_bytes = (size_t)n_req * (4 * 96 + 768) + ((size_t)48 << 20);
	uint8_t *d_seq, *d_pac, *d_pool; int64_t *d_off; AlnReq *d_req; AlnHdr *d_hdr; int *d_gap, *d_lists; unsigned long long *d_cnt;
	HIP_OK(hipMalloc(&d_seq, flat.size())); HIP_OK(hipMalloc(&d_pac, l_pac / 4 + 16)); HIP_OK(hipMalloc(&d_pool, pool_bytes));
	HIP_OK(hipMalloc(&d_off, (size_t)(n_reads + 1) * 8)); HIP_OK(hipMalloc(&d_req, (size_t)n_req * sizeof(AlnReq)));
	HIP_OK(hipMalloc(&d_hdr, (size_t)n_req * sizeof(AlnHdr))); HIP_OK(hipMalloc(&d_gap, gaptab.size() * 4));
	HIP_OK(hipMalloc(&d_lists, (size_t)n_req * 3 * 4)); HIP_OK(hipMalloc(&d_cnt, 256));
	HIP_OK(hipMemcpy(d_seq, flat.data(), flat.size(), hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_pac, pac, l_pac / 4 + 1, hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_off, slot.data(), (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_req, rq.data(), (size_t)n_req * sizeof(AlnReq), hipMemcpyHostToDevice));
	HIP_OK(hipMemcpy(d_gap, gaptab.data(), gaptab.size() * 4, hipMemcpyHostToDevice));
	HIP_OK(hipMemset(d_cnt, 0, 256));
	AlnParams ap;
	ap.l_pac = l_pac; ap.a = opt->a; ap.w = opt->w;
	ExtParams ep;
	memcpy(ep.mat, opt->mat, 25);
	ep.o_del = opt->o_del; ep.e_del = opt->e_del; ep.o_ins = opt->o_ins; ep.e_ins = opt->e_ins; ep.zdrop = opt->zdrop;
	Timer tm;
	tm.start(st);
	launch_aln(st, ap, ep, n_req, d_req, d_seq, d_off, d_pac, d_gap, d_hdr, d_pool, d_cnt, pool_bytes, max_len, max_len + 256, d_lists, which != 0);
	double ms = tm.stop(st);
	HIP_OK(hipGetLastError());
	std::vector<AlnHdr> hdr(n_req);
	std::vector<uint8_t> pool(pool_bytes);
	HIP_OK(hipMemcpy((void *)hdr.data(), d_hdr, (size_t)n_req * sizeof(AlnHdr), hipMemcpyDeviceToHost));
	HIP_OK(hipMemcpy(pool.data(), d_pool, pool_bytes, hipMemcpyDeviceToHost));
	(void)hipFree(d_seq); (void)hipFree(d_pac); (void)hipFree(d_pool); (void)hipFree(d_off); (void)hipFree(d_req); (void)hipFree(d_hdr);
	(void)hipFree(d_gap); (void)hipFree(d_lists); (void)hipFree(d_cnt);
	int rc = 0;
	for (int i = 0; i < n_req; ++i) {
		const AlnHdr &h = hdr[i];
		int *o = out_hdr5 + 5 * (size_t)i;
		o[0] = h.score; o[1] = h.NM; o[2] = h.n_cigar; o[3] = h.md_len; o[4] = h.flags;
		if (h.flags) continue;
		if (h.n_cigar > cigar_cap || h.md_len > md_cap) { rc = -1; continue; }
		memcpy(cigar_out + (size_t)cigar_cap * i, pool.data() + (size_t)h.pool_off * 4, (size_t)h.n_cigar * 4);
		memcpy(md_out + (size_t)md_cap * i, pool.data() + (size_t)h.pool_off * 4 + (size_t)h.n_cigar * 4, (size_t)h.md_len);
	}
	if (kernel_ms) *kernel_ms = ms;
	return rc;
}
