// smem_kernels.hip — the third seeding pass on gfx950, one read per LANE (64 reads per wavefront).
//
// Replaces, on the device, the reference's
//   bwt_seed_strategy1        src/bwt.c:358-379      as mem_collect_intv calls it (src/bwamem.c:148-158)
//   bwt_extend / bwt_2occ4    src/bwt.c:262-275, 189-220
//
// The third pass depends on the read only, not on passes 1-2 (fm_kernels.hip), so it is its own launch: a forward-only
// walk without any interval list, started from the k-mer jump table (fm.p3tab).  A read per lane pays the control flow
// once per 64 extensions (the quad kernel pays it per 16): half the vector instructions per extension.  What a lane
// needs for that:
//  * it owns both 64-byte occ blocks of its extension (4 + 4 x 16-byte loads; the second set is skipped when k and l
//    fall into the same block) and counts all four bases itself: three popcounts per packed word (lo plane, hi plane,
//    both), with the position masks from a table in LDS — no cross-lane traffic at all;
//  * the read is not staged: a lane keeps two windows of 8 bases around its cursor in registers;
//  * every global load of a loop iteration (occ blocks, read window, jump-table entry, metadata of the lane's next
//    read) is issued in one place and consumed an iteration later at the earliest, so an iteration waits for memory once.
// (Round 2 also built passes 1-2 in this mapping — list[entry][lane] in LDS, circular, spilling to HBM: half the
// vector instructions of the quad kernel, but no faster: 12 waves per CU of lane-private 16-byte loads run into the
// CU's limit of outstanding L1 misses (TCP_PENDING_STALL 66 % of the cycles, 9.6 GB/s of 64-byte lines per CU against
// 12 GB/s for a pure gather), 22.2 ms per chunk against 23 for the quad kernel, and worse on sub-batches because
// 196 608 reads in flight leave a long tail.  DESIGN.md §4.1 has the numbers.)
#include <hip/hip_runtime.h>
#include "device.h"

namespace mbw {

typedef unsigned long long u64;
typedef unsigned int u32;

#define SMEM2_BLOCK 256

// ---- occ counts of one 64-byte block for all four bases -------------------------------------------------------------
// cA,cB: the four 64-bit running counts; wA,wB: 8 packed words (16 symbols each, first symbol in the top bits).
// k is already shifted for the '$' row.  Counts symbols 0 .. k&127 of the block.
struct Occ4 { u64 c0, c1, c2, c3; };

// The position masks come from a table in LDS (the occurrence-count table of the reference, src/bwt.c:42-51, is a byte
// lookup; here counting is popcount and what is tabulated is, per number of counted symbols 1..128, the lo-plane mask of
// each of the 8 words): row kk = 8 x u32, two ds_read_b128 per block instead of five VALU per word.
#define OCC_TAB_WORDS (129 * 8)
__device__ __forceinline__ void occ_tab_init(u32 *tab)
{
	for (int e = threadIdx.x; e < OCC_TAB_WORDS; e += blockDim.x) {
		const int kk = e >> 3, i = e & 7;
		int n = kk - 16 * i;
		n = n < 0 ? 0 : (n > 16 ? 16 : n);
		tab[e] = (u32)(0xFFFFFFFF00000000ull >> (2 * n)) & 0x55555555u;   // lo-plane bits of the first n symbols of word i
	}
}

__device__ __forceinline__ Occ4 lane_occ4(const uint4 &cA, const uint4 &cB, const uint4 &wA, const uint4 &wB, u64 k, const u32 *tab)
{
	const int kk = (int)(k & 127) + 1;            // symbols counted: 1..128
	const uint4 *row = (const uint4 *)(tab + kk * 8);
	const uint4 mA = row[0], mB = row[1];
	const u32 w[8] = {wA.x, wA.y, wA.z, wA.w, wB.x, wB.y, wB.z, wB.w};
	const u32 m[8] = {mA.x, mA.y, mA.z, mA.w, mB.x, mB.y, mB.z, mB.w};
	u32 p1 = 0, p2 = 0, p12 = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		const u32 lo = w[i] & m[i], hi = (w[i] >> 1) & m[i];
		p1 += __popc(lo); p2 += __popc(hi); p12 += __popc(lo & hi);
	}
	Occ4 r;
	r.c3 = (((u64)cB.w << 32) | cB.z) + p12;
	r.c2 = (((u64)cB.y << 32) | cB.x) + (p2 - p12);
	r.c1 = (((u64)cA.w << 32) | cA.z) + (p1 - p12);
	r.c0 = (((u64)cA.y << 32) | cA.x) + ((u32)kk - p1 - p2 + p12);
	return r;
}

// One bwt_extend (src/bwt.c:262-275) for the child `csel` only — the sweep never looks at the other three — split into
// the part that issues the loads and the part that consumes them, so that every global load of a loop iteration is in
// flight before anything waits.
struct BlkRegs { uint4 k0, k1, k2, k3, l0, l1, l2, l3; };
struct Child { u64 s_search, s_other, s2; int nblk; };

// (p, x2): searched side and size of the parent interval
__device__ __forceinline__ void extend_issue(const FmDev &fm, u64 p, u64 x2, BlkRegs &B)
{
	const uint4 *blk = (const uint4 *)fm.blk;
	const u64 k = p - 1, l = k + x2;
	const u64 ka = k - (k >= fm.primary), la = l - (l >= fm.primary);
	const uint4 *bk = blk + (ka >> 7) * 4, *bl = blk + (la >> 7) * 4;
	B.k0 = bk[0]; B.k1 = bk[1]; B.k2 = bk[2]; B.k3 = bk[3];
	if ((ka >> 7) != (la >> 7)) { B.l0 = bl[0]; B.l1 = bl[1]; B.l2 = bl[2]; B.l3 = bl[3]; }
}
// returns the child (searched', mirrored', size) and the number of distinct occ blocks touched (1 or 2: the
// algorithmic-work counter of SURVEY §8d)
__device__ __forceinline__ Child extend_finish(const FmDev &fm, const BlkRegs &B, u64 p, u64 other, u64 x2, int csel, const u32 *tab)
{
	const u64 k = p - 1, l = k + x2;
	const u64 ka = k - (k >= fm.primary), la = l - (l >= fm.primary);
	const bool two = (ka >> 7) != (la >> 7);
	const Occ4 tk = lane_occ4(B.k0, B.k1, B.k2, B.k3, ka, tab);
	const Occ4 tl = lane_occ4(two ? B.l0 : B.k0, two ? B.l1 : B.k1, two ? B.l2 : B.k2, two ? B.l3 : B.k3, la, tab);
	const u64 d0 = tl.c0 - tk.c0, d1 = tl.c1 - tk.c1, d2 = tl.c2 - tk.c2, d3 = tl.c3 - tk.c3;
	const u64 tks = csel == 0 ? tk.c0 : csel == 1 ? tk.c1 : csel == 2 ? tk.c2 : tk.c3;
	const u64 l2s = csel == 0 ? fm.L2[0] : csel == 1 ? fm.L2[1] : csel == 2 ? fm.L2[2] : fm.L2[3];
	Child c;
	c.s2 = csel == 0 ? d0 : csel == 1 ? d1 : csel == 2 ? d2 : d3;
	// mirrored side: the children are laid out T,G,C,A behind the (possible) sentinel
	const u64 above = csel == 0 ? d1 + d2 + d3 : csel == 1 ? d2 + d3 : csel == 2 ? d3 : 0;
	c.s_search = l2s + 1 + tks;
	c.s_other = other + ((p <= fm.primary && p + x2 - 1 >= fm.primary) ? 1 : 0) + above;
	c.nblk = two ? 2 : 1;
	return c;
}

// ---- the bases around a cursor: two windows of 8 bases in registers ---------------------------------------------------
// The read is never staged in LDS.  A window that is not there is REQUESTED (the load goes out with the occ-block
// loads of the iteration) and the lane comes back to the same place one iteration later; sweeps ask for the window
// ahead of their cursor early enough that they never wait.
struct ReadWin {
	u64 w0, w1;
	int a0, a1;            // first position of each window (multiple of 8), -8 = empty
	__device__ __forceinline__ void reset() { a0 = a1 = -8; w0 = w1 = 0; }
	__device__ __forceinline__ bool have(int p) const { const int q = p & ~7; return q == a0 || q == a1; }
	__device__ __forceinline__ int get(int p) const
	{
		const u64 w = (p & ~7) == a0 ? w0 : w1;
		return (int)((w >> ((p & 7) << 3)) & 0xff);
	}
	// install window q, keeping the one that holds position `keep`
	__device__ __forceinline__ void put(int q, u64 v, int keep)
	{
		if ((keep & ~7) == a0) { w1 = v; a1 = q; } else { w0 = v; a0 = q; }
	}
};

// ---- third pass: bwt_seed_strategy1 (src/bwt.c:358-379) from every position, forward only ---------------------------
enum { P_PICK = 0, P_SKIP = 1, P_KMER = 2, P_TAB = 3, P_EXT = 4, P_DONE = 5 };

__global__ void __launch_bounds__(SMEM2_BLOCK)
smem_p3_kernel(FmDev fm, SmemParams sp, int n_reads, const uint8_t *__restrict__ seq, const int64_t *__restrict__ off,
               const int *__restrict__ lens, int cap, u64 *__restrict__ out, int *__restrict__ nout_arr, u64 *counters)
{
	__shared__ u32 occ_tab[OCC_TAB_WORDS];
	occ_tab_init(occ_tab);
	__syncthreads();
	const size_t gtid = (size_t)blockIdx.x * SMEM2_BLOCK + threadIdx.x;
	const int n_lanes = (int)(gridDim.x * SMEM2_BLOCK);
	int st = P_PICK;
	int rd = 0, len = 0, x = 0, i = 0, qi = 0, nout = 0, t = 0;
	u32 idx = 0;
	u64 ik0 = 0, ik1 = 0, ik2 = 0;
	u64 *myout = out;
	const uint8_t *rbase = seq;
	u32 nblk = 0, nblk_tab = 0;   // occ blocks of the extensions of all the lane's reads (added to the launch's counters once, at the end); the part of them that the jump table stood in for
	bool overflow = false, need = false;
	ReadWin W;
	W.reset();
	int wq = -1, wkeep = 0;
	bool tq = false, mq = false;
	int r_next = (int)gtid, nx_len = 0;
	int64_t nx_off = 0;
	if (r_next < n_reads) mq = true;
	BlkRegs B;
	B.k0 = B.k1 = B.k2 = B.k3 = B.l0 = B.l1 = B.l2 = B.l3 = make_uint4(0, 0, 0, 0);
	// the table can only stand in for steps that cannot emit (src/bwt.c:369: i - x >= min_len)
	const bool use_tab = fm.p3tab != nullptr && fm.p3_k < sp.min_seed_len;

	auto start_plain = [&]() {   // the walk from x without the table; the window of x is there
		const int b = W.get(x);
		ik0 = fm.L2[b] + 1; ik2 = fm.L2[b + 1] - fm.L2[b]; ik1 = fm.L2[3 - b] + 1;
		i = x + 1; st = P_EXT;
	};
	for (;;) {
		// ======== issue phase ========
		if (need) extend_issue(fm, ik1, ik2, B);   // forward extension = backward extension on the complement strand
		u64 wv = 0;
		if (wq >= 0) wv = *(const u64 *)(rbase + wq);
		ulonglong2 ta = make_ulonglong2(0, 0), tb = ta;
		if (tq) { const ulonglong2 *e = (const ulonglong2 *)fm.p3tab + (size_t)idx * 2; ta = e[0]; tb = e[1]; }
		int m_len = 0;
		int64_t m_off = 0;
		if (mq) { m_len = lens[r_next]; m_off = off[r_next]; }
		// ======== consume phase ========
		if (wq >= 0) { W.put(wq, wv, wkeep); wq = -1; }
		if (mq) { nx_len = m_len; nx_off = m_off; mq = false; }
		if (tq) {   // state after the first p3_k extensions of the k-mer at x, blocks-touched count included
			ik0 = ta.x; ik1 = ta.y; ik2 = tb.x; nblk += (u32)tb.y; nblk_tab += (u32)tb.y;
			i = x + fm.p3_k + 1; st = P_EXT; tq = false;
		}
		if (need) {
			const Child ch = extend_finish(fm, B, ik1, ik0, ik2, 3 - qi, occ_tab);
			nblk += (u32)ch.nblk;
			if (ch.s2 < (u64)sp.max_mem_intv && i - x >= sp.min_seed_len) {
				if (ch.s2 > 0) {
					if (nout < cap) {
						ulonglong2 *o = (ulonglong2 *)(myout + (size_t)nout * 4);
						o[0] = make_ulonglong2(ch.s_other, ch.s_search);
						o[1] = make_ulonglong2(ch.s2, (u64)x << 32 | (u32)(i + 1));
					} else overflow = true;
					++nout;
				}
				x = i + 1; st = P_SKIP;
			} else { ik0 = ch.s_other; ik1 = ch.s_search; ik2 = ch.s2; ++i; }
		}
		need = false;
		if (st == P_PICK && !mq) {
			if (r_next >= n_reads) st = P_DONE;
			else {
				rd = r_next; len = nx_len; rbase = seq + nx_off;
				myout = out + (size_t)rd * cap * 4;
				nout = 0; x = 0; overflow = false;
				W.reset();
				if (len < sp.min_seed_len || sp.max_mem_intv <= 0) x = len;   // src/bwamem.c:260, :148
				st = P_SKIP;
				r_next += n_lanes;
				if (r_next < n_reads) mq = true;
			}
		}
		if (st == P_SKIP) {
			bool blocked = false;
			while (x < len) {
				if (!W.have(x)) { wq = x & ~7; wkeep = x; blocked = true; break; }
				if (W.get(x) > 3) ++x; else break;
			}
			if (!blocked) {
				if (x >= len) {   // the read is finished
					nout_arr[rd] = nout;
					if (overflow) atomicAdd(&counters[2], 1ull);
					st = P_PICK;
				} else if (use_tab && x + fm.p3_k < len) { st = P_KMER; t = 0; idx = 0; }
				else start_plain();
			}
		}
		if (st == P_KMER) {   // the p3_k + 1 bases from x, as far as the windows in hand go
			while (t <= fm.p3_k) {
				const int pos = x + t;
				if (!W.have(pos)) { wq = pos & ~7; wkeep = x; break; }
				const int bt = W.get(pos);
				if (bt > 3) break;
				idx = idx << 2 | (u32)bt; ++t;
			}
			if (t > fm.p3_k) { tq = true; st = P_TAB; }
			else if (wq < 0) {   // an ambiguous base inside the k-mer: walk up to it the plain way
				if (W.have(x)) start_plain();
				else { wq = x & ~7; wkeep = x + t; }
			}
		}
		if (st == P_EXT) {
			if (i >= len) { x = len; st = P_SKIP; }
			else if (!W.have(i)) { if (wq < 0) { wq = i & ~7; wkeep = i - 1; } }
			else {
				qi = W.get(i);
				if (qi > 3) { x = i + 1; st = P_SKIP; }
				else need = true;
			}
		}
		if (__ballot(st != P_DONE) == 0) break;
		if (wq < 0 && st == P_EXT) {
			const int ahead = i + 4;
			if (ahead < len && !W.have(ahead)) { wq = ahead & ~7; wkeep = i; }
		}
	}
	if (nblk) atomicAdd(&counters[1], (u64)nblk);
	if (nblk_tab) atomicAdd(&counters[4], (u64)nblk_tab);
}

// ---- launch ----------------------------------------------------------------------------------------------------------
// adds the occ blocks of the pass to counters[1] (those the jump table stood in for also to counters[4]) and overflowing reads to counters[2]
void launch_smem_p3(void *stream, const FmDev &fm, const SmemParams &sp, int n_reads, const uint8_t *d_seq, const int64_t *d_off,
                    const int *d_len, int cap, uint64_t *d_out, int *d_nout, unsigned long long *d_counters)
{
	const int want = (n_reads + SMEM2_BLOCK - 1) / SMEM2_BLOCK;
	int blocks = 256 * 4;   // 16 waves per CU (99 VGPRs)
	if (blocks > want) blocks = want < 1 ? 1 : want;
	hipLaunchKernelGGL(smem_p3_kernel, dim3(blocks), dim3(SMEM2_BLOCK), 0, (hipStream_t)stream, fm, sp, n_reads, d_seq, d_off, d_len, cap,
	                   (u64 *)d_out, d_nout, (u64 *)d_counters);
}

} // namespace mbw
