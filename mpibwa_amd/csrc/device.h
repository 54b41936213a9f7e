// device.h — host-visible interface of the HIP side (device.hip, fm_kernels.hip, ext_kernels.hip)
#ifndef MBW_DEVICE_H
#define MBW_DEVICE_H
#include "internal.h"
#include <mutex>
#include <vector>

namespace mbw {

// FM-index constants passed by value to kernels.  The occ blocks keep the
// reference's 64-byte geometry (4 x u64 running counts + 8 x u32 packed
// bases per 128 BWT symbols, src/bwt.h:72-73) but live in a 256-B aligned
// HBM buffer so that one quad (4 lanes x 16 B) fetches a block in one
// coalesced 64-B request.
struct FmDev {
	const void *blk;        // n_blk x 64 B
	const void *occ32;      // device-only occ table of the seeding kernel (fm_kernels.hip): per 32 rows of the BWT with sentinel and per base 16 B {count, count of greater symbols, plane, plane of greater symbols}
	const void *occ_sb;     // its superblock records: absolute counts before every 2^31 rows
	const uint64_t *sa;     // sampled SA, sa[0] = -1
	const uint64_t *sa_full; // optional: SA value of EVERY row (seq_len+1 entries), expanded in HBM at upload; null if absent
	uint64_t primary, seq_len;
	uint64_t L2[5];
	int sa_shift;           // log2(sa_intv)
	// optional jump table of the third seeding pass: bi-interval after the first p3_k forward extensions of every
	// (p3_k + 1)-mer, 32 B per entry {x0, x1, x2, blocks touched}; null if absent
	const void *p3tab;
	int p3_k;
	// optional k-mer tables of the seeding kernel (fm_kernels.hip: kmt_build_kernel): the bi-interval of EVERY string of
	// 1 .. kmt_k bases, 16 B per entry in the interval list's packing, the table of length L at entry (4^L - 4) / 3;
	// an extension whose result is that short is one independent 16-byte load instead of two dependent occ fetches
	const void *kmt;
	int kmt_k;
};

struct DevIndex {
	int device = -1;
	bool ready = false;
	FmDev fm{};
	void *d_blk = nullptr; size_t blk_bytes = 0;
	void *d_occ32 = nullptr; size_t occ32_bytes = 0;
	void *d_sa = nullptr;  size_t sa_bytes = 0;
	void *d_pac = nullptr; size_t pac_bytes = 0;
	void *d_sa_full = nullptr; size_t sa_full_bytes = 0; double sa_expand_ms = 0;
	void *d_p3tab = nullptr;
	void *d_kmt = nullptr; size_t kmt_bytes = 0;
	int64_t l_pac = 0;
	// what is resident, as the caller's host-side index describes itself (compared on every mem_process_seqs call)
	uint64_t id_primary = 0, id_seq_len = 0, id_L2[5] = {0, 0, 0, 0, 0};
	int id_n_seqs = 0;
	uint64_t id_hash = 0;   // of the contig table (offsets, lengths, names) and the first occ block
};
DevIndex &dev_index();
// true when (bwt, bns) describe the index that is resident; message = what differs
bool index_matches(const bwt_t *bwt, const bntseq_t *bns, const char **what);
// calls of mem_process_seqs currently inside the library (pipeline.hip); index upload / release need it to be 0
int calls_in_flight();
void expect_calls_in_flight(int n);   // pipeline.hip: how many calls the caller says it keeps in flight (mi355x_prewarm)
// serialises "is the index resident / the right one, then count the call in" (mem_process_seqs) against upload and release
std::recursive_mutex &index_mutex();
void note_buffer_growth(size_t from, size_t to, const char *kind);   // counted by mi355x_buffer_growths()

// SMEM seeding parameters (subset of mem_opt_t used by mem_collect_intv)
struct SmemParams {
	int min_seed_len, split_len, split_width, max_mem_intv;
};

// ---- device buffers reused across calls ----
struct DevBuf;
// A DevBuf that is constructed while an owner list is open (pipeline.hip: the work buffers of a call context) enters it, so that
// the context's buffers can be given back as a whole: when the index is released, or when another call's buffer does not fit.
extern std::vector<DevBuf *> *g_devbuf_owner;
struct DevBuf {
	void *p = nullptr; size_t cap = 0;
	DevBuf() { if (g_devbuf_owner) g_devbuf_owner->push_back(this); }
	DevBuf(const DevBuf &) = delete;
	DevBuf &operator=(const DevBuf &) = delete;
	// grow-only hipMalloc.  A buffer that does not fit asks the pipeline to make room (device_memory_pressure: the buffers of idle
	// call contexts are given back; with other calls in flight the caller waits for one of them to end and fewer calls are admitted
	// from then on) and only a lone call whose buffer still does not fit ends the process.
	void *ensure(size_t bytes);
	void release();
};
// pipeline.hip: called when a device allocation has failed; true = something was freed or a call has ended: try again
bool device_memory_pressure(size_t wanted);
void release_idle_work_buffers();   // all call contexts that are not inside a call (mi355x_finalize)

// Launchers (all asynchronous on `stream`; kernel time measured by the caller with HIP events)
// SMEM seeding: two launches on the stream — the third pass (smem_kernels.hip, a read per lane) first, then passes 1-2
// (fm_kernels.hip, a read per quad), which append to its output.
// d_off[r] must be a multiple of 16 (reads padded to 16-byte slots); d_len[r] is the true length
void launch_smem(void *stream, const FmDev &fm, const SmemParams &sp, int n_reads, const uint8_t *d_seq,
                 const int64_t *d_off, const int *d_len, int cap, uint64_t *d_out, int *d_nout, int max_len,
                 unsigned long long *d_counters /* zeroed; [1]=blocks, [2]=overflow on return */,
                 void *d_scratch, size_t scratch_bytes_per_quad, int n_quads,
                 bool count_blocks /* counters[1] += the reference's occ blocks of passes 1-2 (the third pass always counts) */);
size_t occ32_bytes(uint64_t seq_len);
size_t kmt_bytes(int k);                                         // all tables of lengths 1 .. k
void launch_kmt_build(void *stream, const FmDev &fm, int k, void *d_tab);   // needs fm.occ32; level by level on the stream
void launch_occ32_build(void *stream, FmDev &fm, void *d_buf);   // from fm.blk (bwa format), after upload / broadcast; sets fm.occ32 / fm.occ_sb
int  smem_grid_quads(int max_len, size_t *scratch_per_quad);
void launch_smem_p3(void *stream, const FmDev &fm, const SmemParams &sp, int n_reads, const uint8_t *d_seq, const int64_t *d_off,
                    const int *d_len, int cap, uint64_t *d_out, int *d_nout, unsigned long long *d_counters);
// fills tab (4^(k+1) entries of 32 B) with the state of bwt_seed_strategy1 after k forward extensions of every (k+1)-mer
void launch_p3_build(void *stream, const FmDev &fm, int k, void *d_tab);

void launch_sa(void *stream, const FmDev &fm, int n, const uint64_t *d_k, uint64_t *d_out,
               unsigned long long *d_counters /* [0]=next task, [1]=steps */);
// dense-SA path: one 8-byte load per lookup from FmDev::sa_full
void launch_sa_dense(void *stream, const FmDev &fm, int n, const uint64_t *d_k, uint64_t *d_out);
// fill `full` (seq_len+1 entries) from the sampled SA by walking LF once over the whole text (seq_len steps in total)
void launch_sa_expand(void *stream, const FmDev &fm, uint64_t *full, unsigned long long *d_counters);

struct ExtParams {
	int8_t mat[25];
	int o_del, e_del, o_ins, e_ins, zdrop;
};
void launch_extend(void *stream, const ExtParams &ep, int n, const uint8_t *d_q, const int64_t *d_qoff,
                   const uint8_t *d_t, const int64_t *d_toff, const int *d_w, const int *d_h0, const int *d_eb,
                   int *d_out6, unsigned long long *d_cells, int max_qlen);

// ---- chain -> region kernel (c2a_kernel.hip) ----
struct DevSeed { int64_t rbeg; int32_t qbeg, len; };           // 16 B
struct DevChain {
	int64_t far_beg, far_end;    // bounds of the chain's contig on the chain's strand (doubled coordinate)
	int32_t seed_beg, n_seeds;   // into the flat seed / order arrays
	int32_t rid;
	float frac_rep;
	int64_t rmax0, rmax1;        // the reference window of mem_chain2aln (src/bwamem.c:642-661), already clamped
};
struct DevReg {                  // the fields of mem_alnreg_t that mem_chain2aln fills (src/bwamem.c:708-783)
	int64_t rb, re;
	int32_t qb, qe, rid, score, truesc, w, seedcov, seedlen0;
	float frac_rep;
	int32_t pad;
};
struct C2aParams {
	int64_t l_pac;
	int a, w, pen_clip5, pen_clip3;
	int early;    // 1: row loops end as soon as no output the kernel reads can change (wave_ext.cuh); 0: the reference's rows; 2: both, differences counted
};
// Reads with more than `heavy_t` chains are not walked by one wavefront: their chains are split into independent groups
// (c2a_groups.hip) and every group is a unit of its own.  max_units = 0: no such reads in the launch.
struct C2aUnits {
	int max_units = 0, heavy_t = 0;
	const unsigned int *n_units = nullptr;   // units that exist (<= max_units)
	const int *ustart = nullptr;             // unit u: chains clist[ustart[u] .. ustart[u + 1])
	const int *unit_rd = nullptr, *unit_av = nullptr;   // its read; first region slot of the unit
	const int *clist = nullptr;
	int *c_rabs = nullptr, *c_rcnt = nullptr;           // per chain (same numbering as the chain array): where its regions are, how many
};
size_t c2a_groups_scratch_bytes(int n_el);
void launch_c2a_groups(void *stream, int n_el, int n_heavy, const int *d_hoff, const int *d_heavy, const int *d_chain_beg, const int *d_reg_beg,
                       const DevChain *d_chains, void *d_scratch, int *d_clist, int *d_ustart, int *d_unit_rd, int *d_unit_av, unsigned int *d_n_units);
// counters of c2a_kernel: C2A_STAT_SLOTS lines of 8 u64 (zeroed by the caller, summed by the caller): [0] DP cells computed, [1] extensions,
// [2] extensions answered without DP, [3] (early = 2) extensions whose used outputs differ
#define C2A_STAT_SLOTS 256
void launch_c2a(void *stream, const C2aParams &P, const ExtParams &ep, int n_reads, const uint8_t *d_seq, const int64_t *d_off,
                const int *d_len, const int *d_chain_beg, const int *d_chain_cnt, const DevChain *d_chains, const DevSeed *d_seeds, unsigned int *d_srt,
                const int *d_reg_beg, DevReg *d_regs, int *d_nregs, const int *d_tab, int tab_stride, const uint8_t *d_pac, unsigned long long *d_counters,
                int max_len, const int *d_order = nullptr, const C2aUnits *units = nullptr /* then d_nregs must be zeroed */);

// ---- seeds -> chains -> filtered chains on the device (chain_kernel.hip) ----
struct ChainParams {
	int64_t l_pac;
	int w, max_chain_gap, min_chain_weight, min_seed_len, max_chain_extend;
	float mask_level, drop_ratio;
};
// Per read r: chains / seeds / order are written at index seed_off[r] onwards (a read never keeps more seeds or chains
// than it had seeds); n_chains[r] = number of kept chains, or -1 when the read needs the host path (more than 9 chains,
// more than 64 seeds, or long enough for mem_flt_chained_seeds).  d_tab: the length tables of c2a + row 5 = "flt is a no-op".
void launch_chain(void *stream, const ChainParams &P, int n_reads, const int *d_len, const int *d_nseeds, const int *d_lrep,
                  const int64_t *d_seed_off, const uint64_t *d_sa, const int32_t *d_qbl, const int64_t *d_ann_off, const uint8_t *d_ann_alt,
                  int n_seqs, const int *d_tab, int tab_stride, DevChain *d_chains, DevSeed *d_seeds, unsigned int *d_srt, int *d_nchains,
                  void *d_gen = nullptr, int gen_cap = 0);
size_t chain_general_bytes(int cap, int n_reads);   // scratch of launch_chain: the B-tree kernel's slices for `cap` reads + the retry lists

size_t reg_pack_tmp_bytes(int n_reads);
void launch_reg_pack(void *stream, int n_reads, const int *d_reg_beg, const int *d_nregs, int *d_reg_pos, const DevReg *d_regs, DevReg *d_packed,
                     void *d_tmp, size_t tmp_bytes, const C2aUnits *units = nullptr, const int *d_chain_beg = nullptr, const int *d_chain_cnt = nullptr);

// ---- final global re-alignment on the device (aln_kernel.hip) ----
struct AlnReq {                  // one call of mem_reg2aln's DP loop (src/bwamem.c:1106-1122)
	int64_t rb, re;
	int32_t read, qb, qe, w2, truesc, pad;
};
struct AlnHdr {                  // result header; cigar (n_cigar x u32) and MD (md_len bytes) sit at pool[4 * pool_off]
	int32_t score, NM, n_cigar, md_len;
	uint32_t pool_off;
	int32_t flags;               // 1 = not done on the device (band matrix too large / caps): the host recomputes it
};
struct AlnParams { int64_t l_pac; int a, w; };
size_t aln_lds_per_block(int max_len, int tcap);   // LDS of the full-size CIGAR kernel for reads of up to max_len bases
void launch_aln(void *stream, const AlnParams &P, const ExtParams &ep, int n_req, const AlnReq *d_req, const uint8_t *d_seq,
                const int64_t *d_off, const uint8_t *d_pac, const int *d_gaptab, AlnHdr *d_hdr, uint8_t *d_pool,
                unsigned long long *d_counters, size_t pool_bytes, int max_len, int tcap, int *d_lists /* 3 * n_req ints of scratch */,
                bool wide_only = false /* DP requests skip the narrow-band instantiation (stage tests) */);

// ---- SAM text of confidently paired reads on the device (sam_kernel.hip) ----
struct SamDesc {                 // one output line: the chosen hit of a read as mem_sam_pe's paired branch reports it
	int64_t rb, re;              // region in the doubled coordinate
	int32_t qb, qe;
	int32_t req;                 // its CIGAR request, relative to the first request of the pair; < 0: the line is not the device's
	int32_t rid, flag, mapq, score, sub;
};
struct SamParams {
	int64_t l_pac;
	int has_qual, rg_len;
	char rg[256];                // bwa_rg_id
};
// d_req_base[pair] = first CIGAR request of the pair in d_hdr; out_len[r] = bytes of the record at arena + out_off[r],
// -1 = the host must format the pair, -2 = not a line of the device
void launch_sam_emit(void *stream, const SamParams &P, int n_reads, const SamDesc *d_desc, const int *d_req_base, const AlnHdr *d_hdr,
                     const uint8_t *d_pool, const uint8_t *d_seq, const int64_t *d_off, const int *d_len, const uint8_t *d_qual,
                     const uint8_t *d_names, const int *d_name_off, const int64_t *d_ann_off, const char *d_ann_names, const int *d_ann_name_off,
                     uint8_t *d_arena, size_t arena_bytes, unsigned long long *d_arena_used, unsigned long long *d_out_off, int *d_out_len);

// ---- pairing decisions of the pairs with one plain hit per end (pair_kernel.hip) ----
#define PR_MAXREG 8               // regions per read the kernel looks at (a read with more is the host's); 4 until round 4
extern "C" int mi355x_pair_maxreg(void);
struct PairParams {
	int64_t l_pac;
	int a, b, pen_unpaired, min_seed_len, w, o_del, e_del, o_ins, e_ins, max_chain_gap, T, max_matesw;
	float mask_level_redun, mask_level, XA_drop_ratio;
	uint64_t id0;             // number of the chunk's first pair (n_processed >> 1): the hash tie-breaks of src/bwamem.c:527, src/bwamem_pair.c:222
	int lnq[40];              // (int)(4.343 * log(n + 1) + .499), src/bwamem.c:972, src/bwamem_pair.c:313
	int no_rescue;            // MEM_F_NO_RESCUE or max_matesw <= 0: mem_sam_pe's rescue loop does not run
	int low[4], high[4], failed[4];   // mem_pestat_t per orientation
	int tab_off[4];           // start of each orientation's run in the pair-score table: entry [dist - low]
	int ltab_n;               // entries of the per-length table
};
// PairParams from the options and the insert-size statistics; false when the kernel cannot take this chunk (a degenerate
// distribution); *n_tab = entries of the pair-score table.  pair_tables fills tab[n_tab + P.ltab_n] (scores, then the per-length table).
bool pair_params(const mem_opt_t *opt, int64_t l_pac, const mem_pestat_t pes[4], int64_t n_processed, int max_len, PairParams &P, size_t *n_tab);
void pair_tables(const mem_opt_t *opt, const mem_pestat_t pes[4], const PairParams &P, size_t n_tab, double *tab);
// per read of a sub-batch: its first PR_MAXREG regions and its number of regions, into chunk-wide arrays (d_first: PR_MAXREG records per read)
void launch_first_reg(void *stream, int n, const int *d_reg_pos, const int *d_nregs, const DevReg *d_packed, DevReg *d_first, int *d_nfirst);
// status[k] = 1: pair k is decided; reqs[2k .. 2k+1] and desc[2k .. 2k+1] are what the host's COLLECT pass would have listed
// (desc.req = 0 / 1, relative to the pair's first request); status 0: the host's pair (reqs marked read = -1)
void launch_pair_simple(void *stream, const PairParams &P, int n_pairs, const DevReg *d_first, const int *d_nfirst, const uint8_t *d_ok,
                        const int64_t *d_ann_off, const uint8_t *d_ann_alt, const double *d_ptab, const double *d_ltab, uint8_t *d_status,
                        AlnReq *d_reqs, SamDesc *d_desc);
void launch_desc_overlay(void *stream, int n_pairs, const uint8_t *d_status, const SamDesc *d_from, SamDesc *d_to);

// ---- mate-rescue local alignment on the device (msw_kernel.hip) ----
struct MswReq {                  // one ksw_align2() call of mem_matesw (src/bwamem_pair.c:150-177)
	int64_t rb, re;              // target window in the doubled coordinate, already clipped to the contig
	int32_t read;                // the mate to align (index into the batch)
	int32_t is_rev;              // align its reverse complement
};
struct MswRes { int32_t score, te, qe, score2, te2, tb, qb, flags; };   // kswr_t + flags (1 = recompute on the host)
struct MswParams {
	int64_t l_pac;
	uint32_t slo[4];             // scores of target base t against query codes 0..3, one byte each
	int s4[4];                   // ... against query code 4 (N)
	int o_del, e_del, o_ins, e_ins;
	int a, min_seed_len;
	int max_sc, shift;           // max(mat) and -min(mat) as the striped kernel derives them (src/ksw.c:83-88)
};
size_t msw_lds_bytes(int max_len);
// d_rows: scratch of n_req * (longest window) u16
// h_req / h_len (read lengths) / h_list / d_list (2 n_req ints each) given: requests next to each other for the same mate and orientation
// are aligned two per quad in packed 16-bit arithmetic (msw2_kernel); otherwise every request on its own
void launch_msw(void *stream, const MswParams &P, int n_req, const MswReq *d_req, const uint8_t *d_seq, const int64_t *d_off, const int *d_len,
                const uint8_t *d_pac, MswRes *d_res, uint16_t *d_rows, int max_len, const MswReq *h_req = nullptr, const int *h_len = nullptr,
                int *h_list = nullptr, int *d_list = nullptr);
MswParams msw_params(const mem_opt_t *opt, int64_t l_pac);

// ---- seed enumeration between SMEM and SA lookup (fm_kernels.hip) ----
// per read: sort intervals by info, l_rep (src/bwamem.c:265-272) and the number of SA rows to look up
void launch_seed_prep(void *stream, int n_reads, int cap, uint64_t *d_intv, const int *d_nintv, int max_occ, int *d_nseeds, int *d_lrep);
// per read: write the SA rows and (qbeg,len) of every seed in mem_chain's order (src/bwamem.c:273-283)
void launch_seed_enum(void *stream, int n_reads, int cap, const uint64_t *d_intv, const int *d_nintv, int max_occ,
                      const int64_t *d_seed_off, uint64_t *d_rows, int32_t *d_qbeg_len);

SmemParams smem_params(const mem_opt_t *opt);
int clamp_band(const mem_opt_t *opt, int qlen, int w, int end_bonus);

} // namespace mbw
#endif
