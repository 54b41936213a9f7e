// device.h — host-visible interface of the HIP side (device.hip, fm_kernels.hip, ext_kernels.hip)
#ifndef MBW_DEVICE_H
#define MBW_DEVICE_H
#include "internal.h"

namespace mbw {

// FM-index constants passed by value to kernels.  The occ blocks keep the
// reference's 64-byte geometry (4 x u64 running counts + 8 x u32 packed
// bases per 128 BWT symbols, src/bwt.h:72-73) but live in a 256-B aligned
// HBM buffer so that one quad (4 lanes x 16 B) fetches a block in one
// coalesced 64-B request.
struct FmDev {
	const void *blk;        // n_blk x 64 B
	const uint64_t *sa;     // sampled SA, sa[0] = -1
	uint64_t primary, seq_len;
	uint64_t L2[5];
	int sa_shift;           // log2(sa_intv)
};

struct DevIndex {
	int device = -1;
	bool ready = false;
	FmDev fm{};
	void *d_blk = nullptr; size_t blk_bytes = 0;
	void *d_sa = nullptr;  size_t sa_bytes = 0;
	void *d_pac = nullptr; size_t pac_bytes = 0;
	int64_t l_pac = 0;
};
DevIndex &dev_index();

// SMEM seeding parameters (subset of mem_opt_t used by mem_collect_intv)
struct SmemParams {
	int min_seed_len, split_len, split_width, max_mem_intv;
};

// ---- device buffers reused across calls ----
struct DevBuf {
	void *p = nullptr; size_t cap = 0;
	void *ensure(size_t bytes);   // grow-only hipMalloc
	void release();
};

// Launchers (all asynchronous on `stream`; kernel time measured by the caller with HIP events)
void launch_smem(void *stream, const FmDev &fm, const SmemParams &sp, int n_reads, const uint8_t *d_seq,
                 const int64_t *d_off, int cap, uint64_t *d_out, int *d_nout, int max_len,
                 unsigned long long *d_counters /* [0]=next read, [1]=blocks, [2]=overflow */,
                 void *d_scratch, size_t scratch_bytes_per_quad, int n_quads);
int  smem_grid_quads(int max_len, size_t *scratch_per_quad);

void launch_sa(void *stream, const FmDev &fm, int n, const uint64_t *d_k, uint64_t *d_out,
               unsigned long long *d_counters /* [0]=next task, [1]=steps */);

struct ExtParams {
	int8_t mat[25];
	int o_del, e_del, o_ins, e_ins, zdrop;
};
void launch_extend(void *stream, const ExtParams &ep, int n, const uint8_t *d_q, const int64_t *d_qoff,
                   const uint8_t *d_t, const int64_t *d_toff, const int *d_w, const int *d_h0, const int *d_eb,
                   int *d_out6, unsigned long long *d_cells, int max_qlen);

} // namespace mbw
#endif
