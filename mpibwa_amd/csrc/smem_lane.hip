// smem_lane.hip — SMEM seeding, ONE READ PER LANE (64 reads per wavefront).
//
// Same algorithm and state machine as the quad kernel in fm_kernels.hip (mem_collect_intv, src/bwamem.c:114-162;
// bwt_smem1a src/bwt.c:289-351; bwt_seed_strategy1 :358-379; bwt_extend/bwt_2occ4 :189-275), different mapping:
// the gather probe (probe_kernels.hip) shows that lane-private 64-byte blocks stream at the same ~3 TB/s as
// quad-coalesced ones, and the quad kernel is bound by VALU issue because its bookkeeping is replicated in 4 lanes.
// Here every lane owns a read: the bookkeeping is paid once per read, each lane counts all four bases of its two
// occ blocks (three popcounts per packed word: P(lo), P(hi), P(lo&hi)), the interval list lives in a per-lane HBM
// scratch (L2-resident, 16-byte packed entries) and no LDS is used, so occupancy is set by registers only.
#include <hip/hip_runtime.h>
#include "device.h"

namespace mbw {

typedef unsigned long long u64;
typedef unsigned int u32;

#define LANE_BLOCK 256
#define QL_DW 33            // dwords of LDS per lane for the read, 4 bits per base (odd stride: conflict-free), 256 bases
#define QL_MAXLEN 256

struct Occ4 { u64 c0, c1, c2, c3; };

// Occ(., k) for all four bases from one 64-byte block held in registers (k already shifted for the '$' row)
__device__ __forceinline__ Occ4 lane_occ4(const uint4 &cA, const uint4 &cB, const uint4 &wA, const uint4 &wB, u64 k)
{
	const int kk = (int)(k & 127) + 1;            // symbols of this block that are counted (1..128)
	const int full = kk >> 4, rem = kk & 15;       // full words, symbols of the partial word
	const u32 pm = rem ? ((u32)(0xFFFFFFFF00000000ull >> (2 * rem)) & 0x55555555u) : 0u;
	const u32 w[8] = {wA.x, wA.y, wA.z, wA.w, wB.x, wB.y, wB.z, wB.w};
	u32 p1 = 0, p2 = 0, p12 = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		const u32 msk = i < full ? 0x55555555u : (i == full ? pm : 0u);
		const u32 lo = w[i] & msk, hi = (w[i] >> 1) & msk;
		p1 += __popc(lo); p2 += __popc(hi); p12 += __popc(lo & hi);
	}
	Occ4 r;
	r.c3 = (((u64)cB.w << 32) | cB.z) + p12;
	r.c2 = (((u64)cB.y << 32) | cB.x) + (p2 - p12);
	r.c1 = (((u64)cA.w << 32) | cA.z) + (p1 - p12);
	r.c0 = (((u64)cA.y << 32) | cA.x) + ((u32)kk - p1 - p2 + p12);
	return r;
}

enum { LS_PICK = 0, LS_FWD = 1, LS_BWD = 2, LS_P3 = 3, LS_DONE = 4 };

__device__ __forceinline__ void ent_store(uint4 *L, int e, u64 x0, u64 x1, u64 x2, u32 end)
{
	uint4 v;
	v.x = (u32)x0; v.y = (u32)x1; v.z = (u32)x2;
	v.w = (u32)(x0 >> 32) | (u32)(x1 >> 32) << 2 | (u32)(x2 >> 32) << 4 | end << 16;
	L[e] = v;
}
__device__ __forceinline__ void ent_load(const uint4 *L, int e, u64 &x0, u64 &x1, u64 &x2, int &end)
{
	uint4 v = L[e];
	x0 = (u64)(v.w & 3) << 32 | v.x;
	x1 = (u64)(v.w >> 2 & 3) << 32 | v.y;
	x2 = (u64)(v.w >> 4 & 3) << 32 | v.z;
	end = (int)(v.w >> 16);
}

__global__ void __launch_bounds__(LANE_BLOCK)
smem_lane_kernel(FmDev fm, SmemParams sp, int n_reads, const uint8_t *__restrict__ seq, const int64_t *__restrict__ off,
                 const int *__restrict__ lens, int cap, u64 *__restrict__ out, int *__restrict__ nout_arr, u64 *counters,
                 uint4 *scratch, size_t ent_per_lane)
{
	__shared__ u32 lds_q[LANE_BLOCK * QL_DW];
	u32 *myq = lds_q + threadIdx.x * QL_DW;
	const int lane = threadIdx.x & 63;
	const size_t gtid = (size_t)blockIdx.x * LANE_BLOCK + threadIdx.x;
	uint4 *L = scratch + gtid * ent_per_lane;
	const uint4 *blk = (const uint4 *)fm.blk;

	int st = LS_PICK, pass = 0;
	int rd = 0, len = 0, x = 0, i = 0, j = 0, np = 0, nc = 0, top = 0, min_intv = 1, ret = 0, last_start = -1;
	int nout = 0, k2 = 0, old_n = 0, cb = -1, last_push_end = 0, p_end = 0, ik_end = 0, qi = 0;
	const uint8_t *gq = seq;
	bool q_lds = false;
	// base at position p: from the lane's nibble-packed LDS copy when the read fits, else from HBM
	auto Q = [&](int p_) -> int { return q_lds ? (int)((myq[p_ >> 3] >> ((p_ & 7) << 2)) & 15u) : (int)gq[p_]; };
	u64 ik0 = 0, ik1 = 0, ik2 = 0, lastc_x2 = 0, p0 = 0, p1 = 0, p2 = 0;
	u64 *myout = out;
	u32 nblk = 0;
	bool overflow = false, need = false, back = false;
	Occ4 tk{}, tl{};
	u64 req_p = 0, req_x2 = 0, req_other = 0;
	u64 pf0 = 0, pf1 = 0, pf2 = 0;   // next list entry of the backward sweep, fetched together with the occ blocks
	int pf_end = 0, pf_idx = -1;

	auto begin_smem = [&](int xs, int mi) {
		x = xs; min_intv = mi < 1 ? 1 : mi;
		int b = Q(xs);
		ik0 = fm.L2[b] + 1; ik2 = fm.L2[b + 1] - fm.L2[b]; ik1 = fm.L2[3 - b] + 1; ik_end = xs + 1;
		i = xs + 1; top = 0; st = LS_FWD;
	};
	auto push_fwd = [&]() { ent_store(L, top, ik0, ik1, ik2, (u32)ik_end); last_push_end = ik_end; ++top; };
	auto set_cb = [&]() { int b = i < 0 ? 4 : (int)Q(i); cb = b > 3 ? -1 : b; };
	auto fwd_done = [&]() {
		ret = last_push_end; np = top; i = x - 1; j = 0; nc = 0; last_start = -1; st = LS_BWD;
		set_cb();
	};
	auto call_done = [&]() { if (pass == 1) x = ret; st = LS_PICK; };
	auto emit = [&](u64 e0, u64 e1, u64 e2, int start, int end) {
		if (end - start < sp.min_seed_len) return;
		if (nout < cap) {
			ulonglong2 *o = (ulonglong2 *)(myout + (size_t)nout * 4);
			o[0] = make_ulonglong2(e0, e1);
			o[1] = make_ulonglong2(e2, (u64)start << 32 | (u32)end);
		} else overflow = true;
		++nout;
	};

	for (;;) {
		// ---- consume the previous extend ----
		if (need) {
			const int csel = back ? cb : 3 - qi;
			// child csel: searched side a = L2[c]+1+tk[c], size s = tl[c]-tk[c], mirrored side = base + sum of sizes of larger c
			const u64 d0 = tl.c0 - tk.c0, d1 = tl.c1 - tk.c1, d2 = tl.c2 - tk.c2, d3 = tl.c3 - tk.c3;
			const u64 tks = csel == 0 ? tk.c0 : csel == 1 ? tk.c1 : csel == 2 ? tk.c2 : tk.c3;
			const u64 s2 = csel == 0 ? d0 : csel == 1 ? d1 : csel == 2 ? d2 : d3;
			const u64 above = csel == 0 ? d1 + d2 + d3 : csel == 1 ? d2 + d3 : csel == 2 ? d3 : 0;
			const u64 l2s = csel == 0 ? fm.L2[0] : csel == 1 ? fm.L2[1] : csel == 2 ? fm.L2[2] : fm.L2[3];
			const u64 a = l2s + 1 + tks;
			const u64 mir = req_other + ((req_p <= fm.primary && req_p + req_x2 - 1 >= fm.primary) ? 1 : 0) + above;
			const u64 s0 = back ? a : mir, s1 = back ? mir : a;
			if (st == LS_FWD) {
				bool stop = false;
				if (s2 != ik2) {
					push_fwd();
					if (s2 < (u64)min_intv) { fwd_done(); stop = true; }
				}
				if (!stop) { ik0 = s0; ik1 = s1; ik2 = s2; ik_end = i + 1; ++i; }
			} else if (st == LS_BWD) {
				if (s2 < (u64)min_intv) {
					if (nc == 0 && (last_start < 0 || i + 1 < last_start)) { emit(p0, p1, p2, i + 1, p_end); last_start = i + 1; }
				} else if (nc == 0 || s2 != lastc_x2) {
					ent_store(L, top - 1 - nc, s0, s1, s2, (u32)p_end);
					++nc; lastc_x2 = s2;
				}
				++j;
			} else { // LS_P3
				if (s2 < (u64)sp.max_mem_intv && i - x >= sp.min_seed_len) {
					if (s2 > 0) emit(s0, s1, s2, x, i + 1);
					x = i + 1; st = LS_PICK;
				} else { ik0 = s0; ik1 = s1; ik2 = s2; ++i; }
			}
		}
		need = false;
		// ---- backward sweep bookkeeping ----
		if (st == LS_BWD) {
			if (cb >= 0 && j == np) {
				if (nc == 0) call_done();
				else { np = nc; --i; j = 0; nc = 0; set_cb(); }
			}
			if (st == LS_BWD) {
				if (cb < 0) {
					ent_load(L, top - 1, p0, p1, p2, p_end);
					if (last_start < 0 || i + 1 < last_start) emit(p0, p1, p2, i + 1, p_end);
					call_done();
				} else {
					if (pf_idx == j) { p0 = pf0; p1 = pf1; p2 = pf2; p_end = pf_end; }
					else ent_load(L, top - 1 - j, p0, p1, p2, p_end);
					pf_idx = -1;
					need = true; back = true;
				}
			}
		}
		// ---- new reads are handed out to all idle lanes of the wave with one atomic ----
		{
			const bool want = st == LS_PICK && pass == 0;
			const u64 bal = __ballot(want);
			if (bal) {
				const int leader = __ffsll((long long)bal) - 1;
				u64 base = 0;
				if (lane == leader) base = atomicAdd(&counters[0], (u64)__popcll(bal));
				base = __shfl(base, leader);
				if (want) {
					u64 r = base + __popcll(bal & ((1ull << lane) - 1));
					if (r >= (u64)n_reads) st = LS_DONE;
					else {
						rd = (int)r; gq = seq + off[r]; len = lens[r];
						q_lds = len <= QL_MAXLEN;
						if (q_lds) {   // off[] is 16-byte aligned: 16 bases per load, packed to 4 bits each
							const uint4 *src = (const uint4 *)gq;
							for (int k = 0; k * 16 < len; ++k) {
								uint4 v = src[k];
								u32 w4[4] = {v.x, v.y, v.z, v.w};
								u32 lo = 0, hi = 0;
#pragma unroll
								for (int t = 0; t < 4; ++t) {
									u32 n = (w4[t] & 0xf) | (w4[t] >> 4 & 0xf0) | (w4[t] >> 8 & 0xf00) | (w4[t] >> 12 & 0xf000);
									if (t < 2) lo |= n << (16 * t); else hi |= n << (16 * (t - 2));
								}
								myq[2 * k] = lo; myq[2 * k + 1] = hi;
							}
						}
						myout = out + (size_t)r * cap * 4;
						nout = 0; x = 0; overflow = false; nblk = 0;
						pass = len < sp.min_seed_len ? 4 : 1;
					}
				}
			}
		}
		// ---- between calls: next call of this pass, next pass, end of the read ----
		while (st == LS_PICK && pass != 0) {
			if (pass == 1) {
				while (x < len && Q(x) > 3) ++x;
				if (x >= len) { pass = 2; k2 = 0; old_n = nout < cap ? nout : cap; }
				else begin_smem(x, 1);
			} else if (pass == 2) {
				bool found = false;
				while (k2 < old_n) {
					u64 info = myout[(size_t)k2 * 4 + 3], xx2 = myout[(size_t)k2 * 4 + 2];
					++k2;
					int s = (int)(info >> 32), e = (int)(u32)info;
					if (e - s < sp.split_len || xx2 > (u64)sp.split_width) continue;
					begin_smem((s + e) >> 1, (int)xx2 + 1);
					found = true;
					break;
				}
				if (!found) { pass = 3; x = 0; }
			} else if (pass == 3) {
				if (sp.max_mem_intv <= 0) pass = 4;
				else {
					while (x < len && Q(x) > 3) ++x;
					if (x >= len) pass = 4;
					else {
						int b = Q(x);
						ik0 = fm.L2[b] + 1; ik2 = fm.L2[b + 1] - fm.L2[b]; ik1 = fm.L2[3 - b] + 1;
						i = x + 1; st = LS_P3;
					}
				}
			} else {   // pass 4: the read is finished; a new one is picked up in the next iteration
				nout_arr[rd] = nout;
				atomicAdd(&counters[1], (u64)nblk);
				if (overflow) atomicAdd(&counters[2], 1ull);
				pass = 0;
			}
		}
		// ---- forward sweeps ----
		if (st == LS_FWD) {
			if (i < len) qi = Q(i);
			if (i == len || qi > 3) { push_fwd(); fwd_done(); }
			else { need = true; back = false; }
		} else if (st == LS_P3) {
			if (i < len) qi = Q(i);
			if (i == len) { x = len; st = LS_PICK; }
			else if (qi > 3) { x = i + 1; st = LS_PICK; }
			else { need = true; back = false; }
		}
		if (__ballot(st != LS_DONE) == 0) break;
		// ---- one bwt_extend per lane: two 64-byte blocks, all four bases counted by the lane ----
		if (need) {
			const u64 e0 = back ? p0 : ik0, e1 = back ? p1 : ik1, e2 = back ? p2 : ik2;
			req_p = back ? e0 : e1; req_other = back ? e1 : e0; req_x2 = e2;
			const u64 k = req_p - 1, l = k + e2;
			const u64 ka = k - (k >= fm.primary), la = l - (l >= fm.primary);
			const uint4 *bk = blk + (ka >> 7) * 4, *bl = blk + (la >> 7) * 4;
			const uint4 k0 = bk[0], k1 = bk[1], k2v = bk[2], k3 = bk[3];
			const uint4 l0 = bl[0], l1 = bl[1], l2 = bl[2], l3 = bl[3];
			// the entry after this one is never overwritten in this row (writes go to indices >= top-1-j): fetch it now
			if (back && j + 1 < np) { ent_load(L, top - 2 - j, pf0, pf1, pf2, pf_end); pf_idx = j + 1; }
			tk = lane_occ4(k0, k1, k2v, k3, ka);
			tl = lane_occ4(l0, l1, l2, l3, la);
			nblk += (ka >> 7) == (la >> 7) ? 1 : 2;
		}
	}
}

int smem_lane_grid(int max_len, size_t *scratch_per_lane)
{
	*scratch_per_lane = (size_t)(max_len + 2) * sizeof(uint4);
	return 256 * 16 * 64;   // 16 waves per CU
}

void launch_smem_lane(void *stream, const FmDev &fm, const SmemParams &sp, int n_reads, const uint8_t *d_seq,
                      const int64_t *d_off, const int *d_len, int cap, uint64_t *d_out, int *d_nout,
                      unsigned long long *d_counters, void *d_scratch, size_t scratch_bytes_per_lane, int n_lanes)
{
	int n_blocks = n_lanes / LANE_BLOCK;
	int want = (n_reads + LANE_BLOCK - 1) / LANE_BLOCK;
	if (want < 1) want = 1;
	if (n_blocks > want) n_blocks = want;
	hipLaunchKernelGGL(smem_lane_kernel, dim3(n_blocks), dim3(LANE_BLOCK), 0, (hipStream_t)stream, fm, sp, n_reads, d_seq, d_off,
	                   d_len, cap, (u64 *)d_out, d_nout, (u64 *)d_counters, (uint4 *)d_scratch,
	                   scratch_bytes_per_lane / sizeof(uint4));
}

} // namespace mbw
