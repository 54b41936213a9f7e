// fm_kernels.hip — FM-index kernels for gfx950 with a QUAD of lanes per task: SMEM seeding (passes 1 and 2 of
// mem_collect_intv; the third pass is smem_p3_kernel in smem_kernels.hip), suffix-array lookup, dense-SA expansion,
// the jump table of the third pass, seed enumeration.
//
// Replaces, on the device, the reference's
//   mem_collect_intv          src/bwamem.c:114-162   (3-pass SMEM seeding)
//   bwt_smem1a                src/bwt.c:289-351
//   bwt_seed_strategy1        src/bwt.c:358-379
//   bwt_extend / bwt_2occ4    src/bwt.c:262-275, 189-220
//   bwt_sa / bwt_invPsi       src/bwt.c:86-96, 53-59
//
// Design (MI355X-first, not a translation):
//  * One read (or one SA task) per QUAD of lanes, 16 per wavefront.  An occ
//    block is 64 B = 4 lanes x 16 B, so every FM-index access is one fully
//    coalesced 64-B request; lane c of the quad then counts base c, so the
//    four child intervals of bwt_extend come out one per lane with no
//    reduction, and the data exchange inside the quad is DPP quad_perm
//    (register-to-register, no LDS).
//  * SMEM seeding reads a device-only occ table (round 3): 64 rows of the BWT
//    with the sentinel in place per 64-byte block, per base a running count
//    and a one-hot bit plane, so Occ(c, k) is one masked 64-bit popcount in
//    lane c (the bwa-format blocks cost ~30 vector instructions per Occ for
//    bit-plane arithmetic on 2-bit words, and a shift of k around the
//    sentinel); the bwa-format blocks stay for the SA walk and the third pass.
//  * Every quad runs a small state machine whose loop body contains exactly
//    one bwt_extend, so quads that are in different phases (forward sweep,
//    backward sweep, re-seeding) or on reads of different
//    length never serialise each other; a quad that finishes a read pulls the
//    next one from a global counter (persistent grid, every wave exits when
//    the counter runs past n_reads).
//  * The per-read interval list of bwt_smem1a lives in LDS (one list, compacted
//    in place: the reference's prev/curr ping-pong never grows), spilling to a
//    per-quad HBM scratch only beyond LCAP entries.
#include <hip/hip_runtime.h>
#include "device.h"

namespace mbw {

typedef unsigned long long u64;
typedef unsigned int u32;

// Two LDS footprints per quad: {interval-list entries of 16 B kept in LDS, bytes for the read itself}.
//   short reads (<= 160 bp): 20 entries + 160 B = 480 B per quad -> 30 KB per workgroup, 5 workgroups per CU (the VGPR limit)
//   otherwise:               23 entries + 256 B = 624 B per quad -> 39.5 KB per workgroup (with the superblock records), 4 workgroups per CU
// (measured on 2x150 bp: 24.9 ms -> 22.7-23.5 ms for the whole chunk; 16 entries spill too often, 32 cost a workgroup)
#ifndef LCAP_S
#define LCAP_S 20
#endif
#define QSLOT_S 160
#ifndef LCAP
#define LCAP 23
#endif
#define QSLOT 256
#ifndef SMEM_WG_S
#define SMEM_WG_S 5       // workgroups per CU with the short-read footprint
#endif
#define SMEM_BLOCK 256     // 4 waves = 64 quads per workgroup
#define SMEM_FETCH 16      // reads a wave takes from the work counter at a time (>= the 16 quads of a wave)

template <int CTRL>
__device__ __forceinline__ u32 dpp(u32 v)
{
	return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ u64 dpp64(u64 v)
{
	return (u64)dpp<CTRL>((u32)(v >> 32)) << 32 | dpp<CTRL>((u32)v);
}
// quad_perm selectors
#define QP(a, b, c, d) ((a) | (b) << 2 | (c) << 4 | (d) << 6)

// ---------------------------------------------------------------------------------------------------------------
// The device-only occ table of the seeding kernel (FmDev::occ32, built at upload by occ32_build_kernel).
//
// bwt_extend (src/bwt.c:262-275) computes four child intervals, but a sweep of bwt_smem1a only ever follows ONE of
// them, base c: it needs  Occ(c, k), Occ(c, l)  for the searched side and the size, and for the mirrored side the
// number of symbols GREATER than c in rows (k, l]  (the children lie T,G,C,A behind the sentinel: src/bwt.c:270-273).
// So the table stores, per block of 32 rows of the BWT *with the sentinel in place* (row fm.primary has no base: no
// shift of k around it, src/bwt.c:173, 193-194) and per base c, one 16-byte record
//        { count of c before the block, count of symbols > c before the block, plane of c, plane of symbols > c }
// with the counts relative to the block's superblock of 2^31 rows (absolute superblock counts: FmDev::occ_sb, 32 records
// staged in LDS with L2[c] + 1 folded in).  One extension = two 16-byte loads and four masked popcounts IN ONE LANE:
// no cross-lane traffic, and the four lanes of a quad can extend four different list entries of a backward row.
// 64 B per 32 rows = 12.4 GB for GRCh38 (of 288 GB); the bwa-format blocks stay for the SA walk and the third pass.
// ---------------------------------------------------------------------------------------------------------------
#define SB_SHIFT 31
#define SB_MAX 8          // 2^34 rows (the 34-bit interval bounds) / 2^31

// One bwt_extend for child `c` only: (p, q, s) = searched side, mirrored side, size of the parent; cbase = table + 16 c,
// sbc = &lds_sb[c] (records of superblock s at sbc[4 s]).  COUNT: returns the number of distinct 128-symbol occ blocks
// the REFERENCE touches for this extension (1 or 2, src/bwt.c:193-194), the algorithmic-work counter of SURVEY §8d.
template <bool COUNT>
__device__ __forceinline__ int lane_extend(const FmDev &fm, const char *cbase, const ulonglong2 *sbc, u64 p, u64 q, u64 s, u64 &oa, u64 &omir, u64 &os)
{
	const u64 k = p - 1, l = k + s;
	const uint4 vk = *(const uint4 *)(cbase + ((k & ~31ull) << 1));
	const uint4 vl = *(const uint4 *)(cbase + ((l & ~31ull) << 1));   // (skipping it when k and l share a block was measured: 14.2 -> 16.8 ms)
	const ulonglong2 bk = sbc[(u32)(k >> SB_SHIFT) * 4], bl = sbc[(u32)(l >> SB_SHIFT) * 4];
	const u32 mk = 0xFFFFFFFFu >> (~(u32)k & 31), ml = 0xFFFFFFFFu >> (~(u32)l & 31);
	const u64 ck = bk.x + (u32)(__popc(vk.z & mk) + vk.x), gk = bk.y + (u32)(__popc(vk.w & mk) + vk.y);
	const u64 cl = bl.x + (u32)(__popc(vl.z & ml) + vl.x), gl = bl.y + (u32)(__popc(vl.w & ml) + vl.y);
	oa = ck;                                                              // L2[c] + 1 + Occ(c, k): the superblock record carries L2[c] + 1
	os = cl - ck;
	omir = q + ((p <= fm.primary && l >= fm.primary) ? 1 : 0) + (gl - gk);  // behind the sentinel and the children of the greater bases
	if (!COUNT) return 0;
	const u64 ka = k - (k >= fm.primary), la = l - (l >= fm.primary);
	return (ka >> 7) == (la >> 7) ? 1 : 2;
}

// ---------------------------------------------------------------------------------------------------------------
// The k-mer tables (FmDev::kmt, round 4).  The bi-interval of a string W — (first row of W, first row of its reverse
// complement, number of occurrences) — is a function of W alone, whichever way bwt_smem1a arrives at it.  So for every
// string of 1 .. K bases it is tabulated once per index (16 bytes: the list entry's packing; the table of length L starts
// at entry (4^L - 4) / 3, the string's bases are the entry number, first base in the top bits), built level by level with
// the very lane_extend the kernel uses.  An extension whose RESULT has at most K bases is then one 16-byte load whose
// address depends on the read only: no dependent pair of occ fetches (at that depth k and l lie in different blocks: two
// misses, and the shallow tables live in L2 / the Infinity Cache), which is where the kernel's L1 misses came from.
// ---------------------------------------------------------------------------------------------------------------
__device__ __host__ __forceinline__ u32 kmt_first(int L) { return 0x55555554u & ((1u << (2 * L)) - 1u); }   // (4^L - 4) / 3, L <= 15
__device__ __forceinline__ uint4 kmt_pack(u64 x0, u64 x1, u64 x2)
{
	return make_uint4((u32)x0, (u32)x1, (u32)x2, (u32)(x0 >> 32) | (u32)(x1 >> 32) << 2 | (u32)(x2 >> 32) << 4);
}
__device__ __forceinline__ void kmt_unpack(const uint4 &v, u64 &x0, u64 &x1, u64 &x2)
{
	x0 = (u64)(v.w & 3) << 32 | v.x;
	x1 = (u64)(v.w >> 2 & 3) << 32 | v.y;
	x2 = (u64)(v.w >> 4 & 3) << 32 | v.z;
}
// level L from level L - 1: one lane per string, the forward extension of its prefix by its last base (src/bwt.c:299-311)
__global__ void __launch_bounds__(256) kmt_build_kernel(FmDev fm, int L, uint4 *__restrict__ tab)
{
	__shared__ ulonglong2 lds_sb[SB_MAX * 4];
	if (threadIdx.x < SB_MAX * 4) {
		ulonglong2 v = ((const ulonglong2 *)fm.occ_sb)[threadIdx.x];
		v.x += fm.L2[threadIdx.x & 3] + 1;
		lds_sb[threadIdx.x] = v;
	}
	__syncthreads();
	const u64 code = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (code >= (1ull << (2 * L))) return;
	const int b = (int)(code & 3);
	u64 x0, x1, x2;
	if (L == 1) { x0 = fm.L2[b] + 1; x2 = fm.L2[b + 1] - fm.L2[b]; x1 = fm.L2[3 - b] + 1; }
	else {
		kmt_unpack(tab[kmt_first(L - 1) + (u32)(code >> 2)], x0, x1, x2);
		if (x2 == 0) x0 = x1 = 0;   // (no sweep ever asks for a string whose prefix does not occur)
		else {
			const int csel = 3 - b;
			u64 oa, omir, os;
			lane_extend<false>(fm, (const char *)fm.occ32 + 16 * csel, lds_sb + csel, x1, x0, x2, oa, omir, os);
			x1 = oa; x0 = omir; x2 = os;
		}
	}
	tab[kmt_first(L) + (u32)code] = kmt_pack(x0, x1, x2);
}
size_t kmt_bytes(int k) { return ((size_t)kmt_first(k) + ((size_t)1 << (2 * k))) * 16; }
void launch_kmt_build(void *stream, const FmDev &fm, int k, void *d_tab)
{
	for (int L = 1; L <= k; ++L) {
		const u64 n = 1ull << (2 * L);
		hipLaunchKernelGGL(kmt_build_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fm, L, (uint4 *)d_tab);
	}
}

// One interval-list entry = 16 bytes: x0,x1,x2 (34 bits each: references up to 2^34 symbols = 8.5 Gbp) and the end
// coordinate (16 bits); a lane reads or writes a whole entry (ds_read_b128 / ds_write_b128).
struct QuadList {
	uint4 *lds;      // `cap` entries of this quad
	uint4 *spill;    // per-quad HBM scratch for entries >= cap
	int cap;
};

__device__ __forceinline__ void list_store(const QuadList &L, int e, u64 x0, u64 x1, u64 x2, u64 end)
{
	uint4 v;
	v.x = (u32)x0; v.y = (u32)x1; v.z = (u32)x2;
	v.w = (u32)(x0 >> 32) | (u32)(x1 >> 32) << 2 | (u32)(x2 >> 32) << 4 | (u32)end << 16;
	if (e < L.cap) L.lds[e] = v;
	else L.spill[e - L.cap] = v;
}
__device__ __forceinline__ void list_load(const QuadList &L, int e, u64 &x0, u64 &x1, u64 &x2, u64 &end)
{
	uint4 v;
	if (e < L.cap) v = L.lds[e];
	else v = L.spill[e - L.cap];
	x0 = (u64)(v.w & 3) << 32 | v.x;
	x1 = (u64)(v.w >> 2 & 3) << 32 | v.y;
	x2 = (u64)(v.w >> 4 & 3) << 32 | v.z;
	end = v.w >> 16;
}

enum { ST_PICK = 0, ST_FWD = 1, ST_BWD = 2, ST_DONE = 4 };

// smem_kernel: passes 1 and 2 of mem_collect_intv (src/bwamem.c:114-147), one read per quad of lanes.
//
// A quad runs the state machine of its read (which call, which sweep, which row) in quad-uniform registers.  Going
// forward (src/bwt.c:299-311) the chain of extensions is serial: the four lanes compute the same extension (their loads
// coalesce into one request).  Going backward (src/bwt.c:315-343) the entries of a row are independent: lane t extends
// entry j + t, four entries per iteration, and the row's bookkeeping — which entries die, which survivors are kept
// (a survivor whose size equals the previous survivor's is dropped, :337), where the kept ones go in the compacted list —
// is a few quad-wide bit operations on two ballots.  Dead entries are a prefix of a row (a longer match cannot occur more
// often than its prefix), so at most the first entry of a row is reported (:330-334).
//
// QLDS: every read of the launch fits its quad's LDS slot, so a base is always a plain LDS byte (otherwise the accessor
// needs a generic pointer and every base costs a flat load).  COUNT: count the reference's occ blocks (counters[1]) —
// tests and the bench's counting pass; the production launch leaves that arithmetic out.
template <bool QLDS, int LC, int QS, bool COUNT, bool KMT>
__global__ void __launch_bounds__(SMEM_BLOCK)
smem_kernel(FmDev fm, SmemParams sp, int n_reads, const uint8_t *__restrict__ seq, const int64_t *__restrict__ off,
            const int *__restrict__ lens, int cap, u64 *__restrict__ out, int *__restrict__ nout_arr, u64 *counters, uint4 *scratch,
            size_t scratch_ent_per_quad)
{
	__shared__ uint4 lds_list[(SMEM_BLOCK / 4) * LC];
	__shared__ uint4 lds_read[(SMEM_BLOCK / 4) * (QS / 16)];
	__shared__ ulonglong2 lds_sb[SB_MAX * 4];
	if (threadIdx.x < SB_MAX * 4) {
		ulonglong2 v = ((const ulonglong2 *)fm.occ_sb)[threadIdx.x];
		v.x += fm.L2[threadIdx.x & 3] + 1;
		lds_sb[threadIdx.x] = v;
	}
	__syncthreads();
	const int lane = threadIdx.x & 63, t = lane & 3, qlead = lane & ~3;
	const int quad_in_blk = threadIdx.x >> 2;
	const size_t quad_gid = (size_t)blockIdx.x * (SMEM_BLOCK / 4) + quad_in_blk;
	QuadList L;
	L.lds = lds_list + quad_in_blk * LC;
	L.cap = LC;
	L.spill = scratch + quad_gid * scratch_ent_per_quad;
	uint4 *myread = lds_read + quad_in_blk * (QS / 16);
	const uint8_t *lq = (const uint8_t *)myread;
	const u32 below = (1u << t) - 1;      // the lower lanes of the quad, as a mask

	int st = ST_PICK, pass = 0;
	int rd = 0, len = 0, x = 0, i = 0, j = 0, np = 0, nc = 0, top = 0, min_intv = 1, ret = 0, last_start = -1;
	int nout = 0, k2 = 0, old_n = 0, last_push_end = 0, p3_first = 0;
	int csel = 0;                 // the child the sweep follows: 3 - base going forward, the base going backward; < 0: no base (backward only)
	const uint8_t *gq = seq;
	bool q_lds = false;
	// read base i: from the quad's LDS copy when the read fits, else from HBM
	auto Q = [&](int i_) -> int { return (QLDS || q_lds) ? lq[i_] : gq[i_]; };
	// the bi-interval to extend next as (searched side, mirrored side, size, end): going forward bwt_smem1a's `ik` with
	// x[1] searched (the same in all four lanes), going backward the lane's list entry with x[0] searched
	u64 cp = 1, cq = 0, cs = 0, c_end = 0;
	// the bases from the sweep's current left end onward, first base in the top bits (going forward: from x, filled as the sweep
	// advances; going backward: from i): the entry number of a short result in the k-mer tables is its top bits
	u64 win = 0;
	const uint4 *kmt = (const uint4 *)fm.kmt;
	const int kmt_k = KMT ? fm.kmt_k : 0;
	u64 carry_s = 0;              // backward: size of the child of the previous entry of the row, if it survived
	bool carry_surv = false;
	u64 *myout = out;
	u32 nblk = 0;                 // occ blocks of the lane's extensions: added to the launch's counter once, at the end
	int w_next = 0, w_end = 0;    // the wave's stock of reads (wave-uniform)
	bool overflow = false;

	// start the forward sweep of bwt_smem1a at position x
	auto begin_smem = [&](int xs, int mi) {
		x = xs; min_intv = mi < 1 ? 1 : mi;
		int b = Q(xs);
		cq = fm.L2[b] + 1; cs = fm.L2[b + 1] - fm.L2[b]; cp = fm.L2[3 - b] + 1; c_end = xs + 1;
		i = xs + 1; top = 0; st = ST_FWD;
		win = (u64)b << 62;
#ifdef SMEM_DEBUG
		printf("begin t=%d xs=%d mi=%d pass=%d k2=%d\n", t, xs, mi, pass, k2);
#endif
	};
	auto push_fwd = [&]() {
		if (t == 0) list_store(L, top, cq, cp, cs, c_end);
		last_push_end = (int)c_end;
		++top;
	};
	auto set_cb = [&]() {   // the sweep's left end has moved to i
		csel = (i < 0 || Q(i) > 3) ? -1 : (int)Q(i);
		if (KMT) win = win >> 2 | (u64)(csel & 3) << 62;
	};
	auto fwd_done = [&]() {   // the list holds `top` entries, longest match last pushed
		ret = last_push_end; np = top; i = x - 1; j = 0; nc = 0; last_start = -1; st = ST_BWD; carry_surv = false;
		set_cb();
#ifdef SMEM_DEBUG
		printf("fwd_done t=%d np=%d x=%d ret=%d\n", t, np, x, ret);
#endif
	};
	auto call_done = [&]() {
		if (pass == 1) x = ret;
		st = ST_PICK;
	};
	// report entry (e0, e1, e2) as the match [start, end) (one lane writes the 32-byte record)
	auto emit = [&](bool writer, u64 e0, u64 e1, u64 e2, int start, int end) {
		if (end - start < sp.min_seed_len) return;
		if (nout < cap) {
			if (writer) {
				ulonglong2 *o = (ulonglong2 *)(myout + (size_t)nout * 4);
				o[0] = make_ulonglong2(e0, e1);
				o[1] = make_ulonglong2(e2, (u64)start << 32 | (u32)end);
			}
		} else overflow = true;
		++nout;
	};

	// One pass per iteration, no inner re-dispatch: [backward bookkeeping] -> [pick the next call / read] ->
	// [forward bookkeeping] -> one extension per lane that has a request -> [consume the results].
	u64 oa = 0, omir = 0, os = 0;
	for (;;) {
		bool need = false, valid = false;
		// ---- backward sweep: end of a row, end of the call, or the next four list entries ----
		if (st == ST_BWD) {
			if (csel >= 0 && j >= np) {
				if (nc == 0) call_done();
				else { np = nc; --i; j = 0; nc = 0; carry_surv = false; set_cb(); }
			}
			if (st == ST_BWD) {
				if (csel < 0) {   // start of the read or an ambiguous base: only the longest live match can be maximal
					list_load(L, top - 1, cp, cq, cs, c_end);
					if (last_start < 0 || i + 1 < last_start) emit(t == 0, cp, cq, cs, i + 1, (int)c_end);
					call_done();
				} else {
					valid = j + t < np;
					cp = 1; cs = 0;                 // a lane without an entry extends a harmless dummy (rows 0..0)
					if (valid) list_load(L, top - 1 - j - t, cp, cq, cs, c_end);
					need = true;
				}
			}
		}
		// ---- the quads that need a read are served from the wave's own small stock: one atomic on the work counter per
		// SMEM_FETCH reads of the wave instead of one per read (an atomic on one address costs ~10 ns of a queue the whole chip
		// shares; the stock is the wave's, so no quad idles while another one hoards reads) ----
		int r_new = -1;
		{
			const bool want = st == ST_PICK && pass == 0;
			const unsigned long long wantm = __ballot(want && t == 0);
			if (wantm) {
				const int n_want = __popcll(wantm), avail = w_end - w_next;
				int base2 = 0;
				if (n_want > avail) {   // SMEM_FETCH >= 16 quads: one refill always covers the rest
					if (lane == __ffsll((long long)wantm) - 1) base2 = (int)atomicAdd(&counters[0], (u64)SMEM_FETCH);
					base2 = __shfl(base2, __ffsll((long long)wantm) - 1);
				}
				const int k = __popcll(wantm & ((1ull << qlead) - 1));   // this quad's rank among the wanting ones
				if (want) r_new = k < avail ? w_next + k : base2 + (k - avail);
				if (n_want > avail) { w_next = base2 + (n_want - avail); w_end = base2 + SMEM_FETCH; }
				else w_next += n_want;
			}
		}
		// ---- between calls: next call of this pass, next pass, next read ----
		while (st == ST_PICK) {
			if (pass == 0) {
				if (r_new < 0) break;   // (a quad that has just finished a read gets its next one at the top of the next iteration)
				const int r = r_new;
				r_new = -1;
				if (r >= n_reads) { st = ST_DONE; break; }
				rd = r; gq = seq + off[r]; len = lens[r];
				q_lds = QLDS || len <= QS;
				if (q_lds) {   // off[] is 16-byte aligned: the quad copies the read with 16-B loads
					const uint4 *src = (const uint4 *)gq;
					for (int k = t; k * 16 < len; k += 4) myread[k] = src[k];
				}
				myout = out + (size_t)r * cap * 4;
				nout = p3_first = nout_arr[r];          // the third pass (smem_p3_kernel, launched before) has written its intervals
				overflow = nout > cap;
				if (overflow) nout = p3_first = cap;
				x = 0;
				pass = len < sp.min_seed_len ? 4 : 1;   // src/bwamem.c:260: shorter than a seed => no intervals
			}
			if (pass == 1) {
				while (x < len && Q(x) > 3) ++x;
				if (x >= len) { pass = 2; k2 = p3_first; old_n = nout < cap ? nout : cap; }   // re-seeding looks at the SMEMs of pass 1 only
				else begin_smem(x, 1);
			} else if (pass == 2) {
				bool found = false;
				while (k2 < old_n) {
					u64 info = myout[(size_t)k2 * 4 + 3], xx2 = myout[(size_t)k2 * 4 + 2];
					++k2;
#ifdef SMEM_DEBUG
					printf("scan t=%d k2=%d old_n=%d nout=%d info=%llx xx2=%llu p3_first=%d\n", t, k2 - 1, old_n, nout, info, xx2, p3_first);
#endif
					int s = (int)(info >> 32), e = (int)(u32)info;
					if (e - s < sp.split_len || xx2 > (u64)sp.split_width) continue;
					begin_smem((s + e) >> 1, (int)xx2 + 1);
					found = true;
					break;
				}
				if (!found) pass = 4;
			} else if (pass == 4) {   // read finished
				if (t == 0) {
					nout_arr[rd] = nout;
					if (overflow) atomicAdd(&counters[2], 1ull);
				}
				pass = 0;
			}
		}
		// ---- forward sweeps ----
		if (st == ST_FWD) {
			int qi = 4;
			if (i < len) qi = Q(i);
			if (qi > 3) { push_fwd(); fwd_done(); }   // end of the read or an ambiguous base: the backward sweep starts in the next iteration
			else { need = true; csel = 3 - qi; }
		}
		if (__ballot(st != ST_DONE) == 0) break;
		// going forward lane 0 extends for the quad, going backward every lane with an entry: the other lanes issue no loads
		{
			const bool mine = st == ST_FWD ? t == 0 : valid;
			if (need && mine) {
				const int tl = st == ST_FWD ? i - x + 1 : (int)c_end - i;   // bases of the result
				if (KMT && tl <= kmt_k) {
					u64 w = win;
					if (st == ST_FWD) w |= (u64)(3 - csel) << (64 - 2 * tl);
					u64 x0, x1;
					kmt_unpack(kmt[kmt_first(tl) + (u32)(w >> (64 - 2 * tl))], x0, x1, os);
					oa = st == ST_FWD ? x1 : x0; omir = st == ST_FWD ? x0 : x1;
				} else nblk += lane_extend<COUNT>(fm, (const char *)fm.occ32 + 16 * csel, lds_sb + csel, cp, cq, cs, oa, omir, os);
			}
		}
		// ---- consume ----
		const bool bwd = need && st == ST_BWD;   // (before the forward branch below may turn the quad around)
		{
			const u64 f_a = dpp64<QP(0, 0, 0, 0)>(oa), f_mir = dpp64<QP(0, 0, 0, 0)>(omir), f_s = dpp64<QP(0, 0, 0, 0)>(os);   // lane 0's child
			if (need && st == ST_FWD) {
				bool stop = false;
				if (f_s != cs) {
					push_fwd();
					if (f_s < (u64)min_intv) { fwd_done(); stop = true; }
				}
				if (!stop) {
					if (KMT && i - x < 32) win |= (u64)(3 - csel) << (62 - 2 * (i - x));
					cp = f_a; cq = f_mir; cs = f_s; c_end = i + 1; ++i;
				}
			}
		}
		// (the ballots and the DPP reads of the backward bookkeeping stay outside lane-divergent branches)
		{
			const bool dead = bwd && valid && os < (u64)min_intv, surv = bwd && valid && !dead;
			const u64 prev_s = dpp64<QP(0, 0, 1, 2)>(os);                 // the child size of the entry before the lane's
			const u32 survq = (u32)(__ballot(surv) >> qlead) & 15u;
			const bool prev_surv = t == 0 ? carry_surv : ((survq << 1) >> t) & 1;
			const bool keep = surv && !(prev_surv && os == (t == 0 ? carry_s : prev_s));
			const u32 keepq = (u32)(__ballot(keep) >> qlead) & 15u;
			const u32 deadq = (u32)(__ballot(dead) >> qlead) & 15u;
			const u64 last_s = dpp64<QP(3, 3, 3, 3)>(os);
			const int end0 = (int)dpp<QP(0, 0, 0, 0)>((u32)c_end);          // the end of entry j (lane 0's): what the row's report is decided on
#ifdef SMEM_DEBUG
			if (bwd && t == 0) printf("row i=%d j=%d np=%d nc=%d min=%d last_start=%d surv=%x dead=%x keep=%x os=%llu cs=%llu end=%d\n", i, j, np, nc, min_intv, last_start, survq, deadq, keepq, os, cs, (int)c_end);
#endif
			if (bwd) {
				// the first entry of a row is reported when it dies and no longer match is alive (src/bwt.c:330-334)
				if (j == 0 && (deadq & 1) && (last_start < 0 || i + 1 < last_start)) {
					emit(t == 0, cp, cq, cs, i + 1, end0);           // (lane 0 holds entry 0 of the row; the other lanes only keep count)
					last_start = i + 1;
				}
				if (keep) list_store(L, top - 1 - nc - __popc(keepq & below), oa, omir, os, c_end);
				nc += __popc(keepq);
				carry_surv = (survq >> 3) & 1; carry_s = last_s;
				j += 4;
			}
		}
	}
	if (COUNT) {
		for (int o = 32; o; o >>= 1) nblk += __shfl_xor(nblk, o);
		if (lane == 0 && nblk) atomicAdd(&counters[1], (u64)nblk);
	}
}

// The device-only occ table (layout above): one thread per block of 32 rows, from the bwa-format blocks (src/bwt.h:72-73:
// per 128 symbols 4 x u64 running counts + 8 x u32 packed symbols, first symbol in the top bits, the sentinel left out).
__device__ __forceinline__ void occ_before_row(const u32 *__restrict__ blk, u64 primary, u64 seq_len, u64 R0, u64 cnt[4])
{
	// number of each base in rows 0 .. R0-1 of the BWT with sentinel
	cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
	if (R0 == 0) return;
	u64 k = R0 - 1;
	if (k > seq_len) k = seq_len;
	if (k == 0 && primary == 0) return;
	const u64 kk = k - (k >= primary);                  // rows 0..k hold the symbols 0..kk (src/bwt.c:173)
	const u64 *cb = (const u64 *)(blk + (kk >> 7) * 16);
	for (int c = 0; c < 4; ++c) cnt[c] = cb[c];
	for (u64 jdx = kk & ~127ull; jdx <= kk; ++jdx) {
		const u32 w = blk[(jdx >> 7) * 16 + 8 + ((jdx & 127) >> 4)];
		++cnt[(w >> ((~(u32)jdx & 15) << 1)) & 3];
	}
}
__global__ void __launch_bounds__(256) occ32_build_kernel(const u32 *__restrict__ blk, u64 primary, u64 seq_len, u64 n_new, uint4 *__restrict__ out,
                                                          ulonglong2 *__restrict__ sb)
{
	const u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (b < SB_MAX) {   // the superblock records (absolute counts before row b << SB_SHIFT)
		u64 cnt[4];
		occ_before_row(blk, primary, seq_len, b << SB_SHIFT, cnt);
		sb[b * 4 + 0] = make_ulonglong2(cnt[0], cnt[1] + cnt[2] + cnt[3]);
		sb[b * 4 + 1] = make_ulonglong2(cnt[1], cnt[2] + cnt[3]);
		sb[b * 4 + 2] = make_ulonglong2(cnt[2], cnt[3]);
		sb[b * 4 + 3] = make_ulonglong2(cnt[3], 0);
	}
	if (b >= n_new) return;
	const u64 R0 = b << 5;
	u64 cnt[4], base[4];
	occ_before_row(blk, primary, seq_len, R0, cnt);
	occ_before_row(blk, primary, seq_len, R0 >> SB_SHIFT << SB_SHIFT, base);
	u32 plane[4] = {0, 0, 0, 0};
	for (int r = 0; r < 32; ++r) {
		const u64 R = R0 + r;
		if (R > seq_len) break;
		if (R == primary) continue;
		const u64 jdx = R - (R > primary);
		const u32 w = blk[(jdx >> 7) * 16 + 8 + ((jdx & 127) >> 4)];
		plane[(w >> ((~(u32)jdx & 15) << 1)) & 3] |= 1u << r;
	}
	u32 rel[4];
	for (int c = 0; c < 4; ++c) rel[c] = (u32)(cnt[c] - base[c]);
	out[b * 4 + 0] = make_uint4(rel[0], rel[1] + rel[2] + rel[3], plane[0], plane[1] | plane[2] | plane[3]);
	out[b * 4 + 1] = make_uint4(rel[1], rel[2] + rel[3], plane[1], plane[2] | plane[3]);
	out[b * 4 + 2] = make_uint4(rel[2], rel[3], plane[2], plane[3]);
	out[b * 4 + 3] = make_uint4(rel[3], 0, plane[3], 0);
}

#define OCC_SB_BYTES (SB_MAX * 4 * 16)
size_t occ32_bytes(uint64_t seq_len) { return OCC_SB_BYTES + ((seq_len + 1 + 31) / 32 + 2) * 64; }
// d_buf: occ32_bytes(seq_len) bytes; the superblock records come first, the blocks behind them
void launch_occ32_build(void *stream, FmDev &fm, void *d_buf)
{
	const u64 n_new = (occ32_bytes(fm.seq_len) - OCC_SB_BYTES) / 64;
	fm.occ_sb = d_buf;
	fm.occ32 = (const char *)d_buf + OCC_SB_BYTES;
	hipLaunchKernelGGL(occ32_build_kernel, dim3((unsigned)((n_new + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u32 *)fm.blk, fm.primary,
	                   fm.seq_len, n_new, (uint4 *)fm.occ32, (ulonglong2 *)d_buf);
}

// One lane per (k+1)-mer: the forward extensions of bwt_seed_strategy1 (src/bwt.c:358-379) with the very function the
// seeding kernel uses, so a table hit is indistinguishable from doing the steps.
__global__ void __launch_bounds__(256) p3_build_kernel(FmDev fm, int k, u64 n_kmers, u64 *__restrict__ tab)
{
	__shared__ ulonglong2 lds_sb[SB_MAX * 4];
	if (threadIdx.x < SB_MAX * 4) {
		ulonglong2 v = ((const ulonglong2 *)fm.occ_sb)[threadIdx.x];
		v.x += fm.L2[threadIdx.x & 3] + 1;
		lds_sb[threadIdx.x] = v;
	}
	__syncthreads();
	const u64 kmer = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (kmer >= n_kmers) return;
	int b = (int)(kmer >> (2 * k)) & 3;
	u64 ik0 = fm.L2[b] + 1, ik2 = fm.L2[b + 1] - fm.L2[b], ik1 = fm.L2[3 - b] + 1;
	u32 nblk = 0;
	for (int s = 1; s <= k; ++s) {
		const int csel = 3 - ((int)(kmer >> (2 * (k - s))) & 3);
		u64 oa, omir, os;
		nblk += lane_extend<true>(fm, (const char *)fm.occ32 + 16 * csel, lds_sb + csel, ik1, ik0, ik2, oa, omir, os);
		ik1 = oa; ik0 = omir; ik2 = os;
	}
	ulonglong2 *o = (ulonglong2 *)(tab + kmer * 4);
	o[0] = make_ulonglong2(ik0, ik1);
	o[1] = make_ulonglong2(ik2, (u64)nblk);
}

void launch_p3_build(void *stream, const FmDev &fm, int k, void *d_tab)
{
	const u64 n_kmers = 1ull << (2 * (k + 1));
	const u64 threads = n_kmers;
	hipLaunchKernelGGL(p3_build_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fm, k, n_kmers, (u64 *)d_tab);
}

int smem_grid_quads(int max_len, size_t *scratch_per_quad)
{
	// As many workgroups per CU as LDS and registers allow (persistent grid): 5 with the short-read footprint, else 4.
	// They fill the LDS of the CU: nothing else that needs LDS starts next to a running SMEM launch (leaving room was
	// measured to cost more than it gives back).  MPIBWA_SMEM_WG_PER_CU overrides.
	const bool small = max_len <= QSLOT_S;
	int per_cu = small ? SMEM_WG_S : 4;
	if (const char *e = getenv("MPIBWA_SMEM_WG_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= per_cu) per_cu = v; }
	int n_blocks = 256 * per_cu;
	const int lc = small ? LCAP_S : LCAP;
	size_t ent = max_len + 1 > lc ? (size_t)(max_len + 1 - lc) : 0;
	*scratch_per_quad = (ent + 1) * sizeof(uint4);
	return n_blocks * (SMEM_BLOCK / 4);
}

void launch_smem(void *stream, const FmDev &fm, const SmemParams &sp, int n_reads, const uint8_t *d_seq,
                 const int64_t *d_off, const int *d_len, int cap, uint64_t *d_out, int *d_nout, int max_len,
                 unsigned long long *d_counters, void *d_scratch, size_t scratch_bytes_per_quad, int n_quads, bool count_blocks)
{
	// the third pass depends on the read only: its own launch, first (it writes the interval counts the kernel below appends to)
	launch_smem_p3(stream, fm, sp, n_reads, d_seq, d_off, d_len, cap, d_out, d_nout, d_counters);
	int n_blocks = n_quads / (SMEM_BLOCK / 4);
	int want = (n_reads + SMEM_BLOCK / 4 - 1) / (SMEM_BLOCK / 4);
	if (want < 1) want = 1;
	if (n_blocks > want) n_blocks = want;
#define SMEM_LAUNCH(...) hipLaunchKernelGGL((smem_kernel<__VA_ARGS__>), dim3(n_blocks), dim3(SMEM_BLOCK), 0, (hipStream_t)stream, fm, sp, n_reads, d_seq, \
	                                    d_off, d_len, cap, (u64 *)d_out, d_nout, (u64 *)d_counters, (uint4 *)d_scratch,                          \
	                                    scratch_bytes_per_quad / sizeof(uint4))
	// counting the reference's occ blocks means doing its extensions: the counting variant never takes a result from the k-mer tables
	if (count_blocks) {
		if (max_len <= QSLOT_S) SMEM_LAUNCH(true, LCAP_S, QSLOT_S, true, false);
		else if (max_len <= QSLOT) SMEM_LAUNCH(true, LCAP, QSLOT, true, false);
		else SMEM_LAUNCH(false, LCAP, QSLOT, true, false);
	} else if (fm.kmt && fm.kmt_k > 0) {
		if (max_len <= QSLOT_S) SMEM_LAUNCH(true, LCAP_S, QSLOT_S, false, true);
		else if (max_len <= QSLOT) SMEM_LAUNCH(true, LCAP, QSLOT, false, true);
		else SMEM_LAUNCH(false, LCAP, QSLOT, false, true);
	} else {
		if (max_len <= QSLOT_S) SMEM_LAUNCH(true, LCAP_S, QSLOT_S, false, false);
		else if (max_len <= QSLOT) SMEM_LAUNCH(true, LCAP, QSLOT, false, false);
		else SMEM_LAUNCH(false, LCAP, QSLOT, false, false);
	}
#undef SMEM_LAUNCH
}

// ---------------------------------------------------------------------------
// Suffix-array lookup: LF-walk to the next sampled row, one task per quad.
// Each step reads exactly one occ block (the BWT symbol at the row and the
// rank of that symbol live in the same 64 bytes).
// ---------------------------------------------------------------------------
#define SA_BLOCK 256

__global__ void __launch_bounds__(SA_BLOCK)
sa_kernel(FmDev fm, int n, const u64 *__restrict__ ks, u64 *__restrict__ out, u64 *counters)
{
	const int lane = threadIdx.x & 63, c = lane & 3, qlead = lane & ~3;
	const uint4 *blk = (const uint4 *)fm.blk;
	const u64 mask = ((u64)1 << fm.sa_shift) - 1;
	bool done = false, have = false;
	u64 k = 0, steps = 0;
	int task = 0;
	u32 nsteps = 0;
	for (;;) {
		// fetch / retire
		while (!done && (!have || (k & mask) == 0)) {
			if (have) {
				if (c == 0) out[task] = steps + fm.sa[k >> fm.sa_shift];
				have = false;
			}
			int t = 0;
			if (c == 0) t = (int)atomicAdd(&counters[0], 1ull);
			t = __shfl(t, qlead);
			if (t >= n) { done = true; break; }
			task = t; k = ks[t]; steps = 0; have = true;
		}
		if (__ballot(!done) == 0) break;
		if (!done) {
			++steps; ++nsteps;
			if (k == fm.primary) k = 0;   // src/bwt.c:58
			else {
				u64 x = k - (k > fm.primary);
				uint4 v = blk[(x >> 7) * 4 + c];
				// the symbol at x: word (x&127)>>4 sits in lane 2 + (word>>2), component word&3
				int wi = (int)(x & 127) >> 4;
				u32 mine = (wi & 3) == 0 ? v.x : (wi & 3) == 1 ? v.y : (wi & 3) == 2 ? v.z : v.w;
				u32 word = __shfl(mine, qlead | (2 + (wi >> 2)));
				int sym = (word >> ((~(u32)x & 15) << 1)) & 3;
				// rank of sym up to and including x: lanes 2,3 count their four words, lane sym>>1 adds the running count
				u64 part = 0;
				if (c >= 2) {
					const u32 pat = 0x55555555u * (u32)sym;
					const int kk = (int)(x & 127) + 1 - (c == 3 ? 64 : 0);
					u32 w4[4] = {v.x, v.y, v.z, v.w};
					u32 cnt = 0;
#pragma unroll
					for (int i2 = 0; i2 < 4; ++i2) {
						int m = kk - 16 * i2;
						m = m < 0 ? 0 : (m > 16 ? 16 : m);
						u32 msk = (u32)(0xFFFFFFFF00000000ull >> (2 * m));
						u32 y = ~(w4[i2] ^ pat);
						cnt += __popc(y & (y >> 1) & 0x55555555u & msk);
					}
					part = cnt;
				} else if (c == (sym >> 1)) {
					part = (sym & 1) ? ((u64)v.w << 32 | v.z) : ((u64)v.y << 32 | v.x);
				}
				part += dpp64<QP(1, 0, 3, 2)>(part);
				part += dpp64<QP(2, 3, 0, 1)>(part);
				k = fm.L2[sym] + part;
			}
		}
	}
	if (c == 0 && nsteps) atomicAdd(&counters[1], (u64)nsteps);
}

void launch_sa(void *stream, const FmDev &fm, int n, const uint64_t *d_k, uint64_t *d_out, unsigned long long *d_counters)
{
	int want = (n + SA_BLOCK / 4 - 1) / (SA_BLOCK / 4);
	int n_blocks = 256 * 8;
	if (want < 1) want = 1;
	if (n_blocks > want) n_blocks = want;
	hipLaunchKernelGGL(sa_kernel, dim3(n_blocks), dim3(SA_BLOCK), 0, (hipStream_t)stream, fm, n, (const u64 *)d_k,
	                   (u64 *)d_out, (u64 *)d_counters);
}

// ---------------------------------------------------------------------------
// Dense suffix array.  288 GB of HBM hold the SA value of every row (50 GB for GRCh38), so a lookup becomes one
// 8-byte load instead of a ~31-step LF walk.  The table is expanded on the device from the sampled SA: LF maps the
// row of suffix i to the row of suffix i-1, so walking LF from every sampled row until the next sampled row visits
// each row exactly once (seq_len steps in total) and assigns SA = start value - steps.  Values are what bwt_sa()
// returns (src/bwt.c:86-96), including its sa[0] = -1 convention, so results are bit-identical to the walk.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(SA_BLOCK)
sa_expand_kernel(FmDev fm, u64 n_samples, u64 *__restrict__ full, u64 *counters)
{
	const int lane = threadIdx.x & 63, c = lane & 3, qlead = lane & ~3;
	const uint4 *blk = (const uint4 *)fm.blk;
	const u64 mask = ((u64)1 << fm.sa_shift) - 1;
	bool done = false, have = false;
	u64 k = 0, v = 0;
	for (;;) {
		while (!done && !have) {
			u64 t = 0;
			if (c == 0) t = atomicAdd(&counters[0], 1ull);
			t = __shfl(t, qlead);
			if (t >= n_samples) { done = true; break; }
			k = t << fm.sa_shift;
			v = t == 0 ? fm.seq_len : fm.sa[t];          // row 0 is the '$' suffix: true value seq_len (stored as -1)
			if (c == 0) full[k] = t == 0 ? ~0ull : v;
			have = true;
		}
		if (__ballot(!done) == 0) break;
		if (!done) {
			// one LF step (same arithmetic as sa_kernel)
			if (k == fm.primary) k = 0;
			else {
				u64 x = k - (k > fm.primary);
				uint4 vv = blk[(x >> 7) * 4 + c];
				int wi = (int)(x & 127) >> 4;
				u32 mine = (wi & 3) == 0 ? vv.x : (wi & 3) == 1 ? vv.y : (wi & 3) == 2 ? vv.z : vv.w;
				u32 word = __shfl(mine, qlead | (2 + (wi >> 2)));
				int sym = (word >> ((~(u32)x & 15) << 1)) & 3;
				u64 part = 0;
				if (c >= 2) {
					const u32 pat = 0x55555555u * (u32)sym;
					const int kk = (int)(x & 127) + 1 - (c == 3 ? 64 : 0);
					u32 w4[4] = {vv.x, vv.y, vv.z, vv.w};
					u32 cnt = 0;
#pragma unroll
					for (int i2 = 0; i2 < 4; ++i2) {
						int m = kk - 16 * i2;
						m = m < 0 ? 0 : (m > 16 ? 16 : m);
						u32 msk = (u32)(0xFFFFFFFF00000000ull >> (2 * m));
						u32 y = ~(w4[i2] ^ pat);
						cnt += __popc(y & (y >> 1) & 0x55555555u & msk);
					}
					part = cnt;
				} else if (c == (sym >> 1)) {
					part = (sym & 1) ? ((u64)vv.w << 32 | vv.z) : ((u64)vv.y << 32 | vv.x);
				}
				part += dpp64<QP(1, 0, 3, 2)>(part);
				part += dpp64<QP(2, 3, 0, 1)>(part);
				k = fm.L2[sym] + part;
			}
			--v;
			if ((k & mask) == 0) have = false;           // reached the next sampled row: its own walk covers the rest
			else if (c == 0) full[k] = v;
		}
	}
}

__global__ void sa_dense_kernel(const u64 *__restrict__ full, int n, const u64 *__restrict__ ks, u64 *__restrict__ out)
{
	int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < n) out[t] = full[ks[t]];
}

void launch_sa_expand(void *stream, const FmDev &fm, uint64_t *full, unsigned long long *d_counters)
{
	u64 n_samples = (fm.seq_len + ((u64)1 << fm.sa_shift)) >> fm.sa_shift;
	hipLaunchKernelGGL(sa_expand_kernel, dim3(256 * 8), dim3(SA_BLOCK), 0, (hipStream_t)stream, fm, n_samples, (u64 *)full, (u64 *)d_counters);
}
void launch_sa_dense(void *stream, const FmDev &fm, int n, const uint64_t *d_k, uint64_t *d_out)
{
	hipLaunchKernelGGL(sa_dense_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const u64 *)fm.sa_full, n, (const u64 *)d_k,
	                   (u64 *)d_out);
}

// ---------------------------------------------------------------------------
// Seed enumeration: one thread per read (tiny integer work, L2-resident data).
// ---------------------------------------------------------------------------
__global__ void seed_prep_kernel(int n_reads, int cap, u64 *intv, const int *__restrict__ nintv, int max_occ,
                                 int *__restrict__ nseeds, int *__restrict__ lrep)
{
	int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	int n = nintv[r] < cap ? nintv[r] : cap;
	ulonglong4 *a = (ulonglong4 *)(intv + (size_t)r * cap * 4);
	// insertion sort by info (.w); equal keys are identical records (same substring => same bi-interval)
	for (int i = 1; i < n; ++i) {
		ulonglong4 v = a[i];
		int j = i - 1;
		while (j >= 0 && a[j].w > v.w) { a[j + 1] = a[j]; --j; }
		a[j + 1] = v;
	}
	int b = 0, e = 0, l_rep = 0, total = 0;
	for (int i = 0; i < n; ++i) {
		ulonglong4 v = a[i];
		int sb = (int)(v.w >> 32), se = (int)(u32)v.w;
		if (v.z > (u64)max_occ) {
			if (sb > e) { l_rep += e - b; b = sb; e = se; }
			else e = e > se ? e : se;
		}
		u64 step = v.z > (u64)max_occ ? v.z / max_occ : 1;
		u64 cnt = (v.z + step - 1) / step;
		total += (int)(cnt < (u64)max_occ ? cnt : (u64)max_occ);
	}
	l_rep += e - b;
	nseeds[r] = total;
	lrep[r] = l_rep;
}

__global__ void seed_enum_kernel(int n_reads, int cap, const u64 *__restrict__ intv, const int *__restrict__ nintv, int max_occ,
                                 const int64_t *__restrict__ seed_off, u64 *__restrict__ rows, int32_t *__restrict__ qbeg_len)
{
	int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	int n = nintv[r] < cap ? nintv[r] : cap;
	const ulonglong4 *a = (const ulonglong4 *)(intv + (size_t)r * cap * 4);
	int64_t o = seed_off[r];
	for (int i = 0; i < n; ++i) {
		ulonglong4 v = a[i];
		int qbeg = (int)(v.w >> 32), slen = (int)(u32)v.w - qbeg;
		u64 step = v.z > (u64)max_occ ? v.z / max_occ : 1;
		int count = 0;
		for (u64 k = 0; k < v.z && count < max_occ; k += step, ++count) {
			rows[o] = v.x + k;
			qbeg_len[2 * o] = qbeg; qbeg_len[2 * o + 1] = slen;
			++o;
		}
	}
}

void launch_seed_prep(void *stream, int n_reads, int cap, uint64_t *d_intv, const int *d_nintv, int max_occ, int *d_nseeds, int *d_lrep)
{
	hipLaunchKernelGGL(seed_prep_kernel, dim3((n_reads + 127) / 128), dim3(128), 0, (hipStream_t)stream, n_reads, cap, (u64 *)d_intv,
	                   d_nintv, max_occ, d_nseeds, d_lrep);
}
void launch_seed_enum(void *stream, int n_reads, int cap, const uint64_t *d_intv, const int *d_nintv, int max_occ,
                      const int64_t *d_seed_off, uint64_t *d_rows, int32_t *d_qbeg_len)
{
	hipLaunchKernelGGL(seed_enum_kernel, dim3((n_reads + 127) / 128), dim3(128), 0, (hipStream_t)stream, n_reads, cap,
	                   (const u64 *)d_intv, d_nintv, max_occ, d_seed_off, (u64 *)d_rows, d_qbeg_len);
}

} // namespace mbw
