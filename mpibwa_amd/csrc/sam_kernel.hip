// sam_kernel.hip — SAM text on the device for the records of confidently paired reads.
//
// Device counterpart of the tail of mem_reg2aln (src/bwamem.c:1123-1157: position, strand, squeeze of a leading /
// trailing deletion, soft clips) and of mem_aln2sam (src/bwamem.c:825-946) for the case that makes up the bulk of a
// chunk: a pair that mem_sam_pe reports through its "paired" branch (src/bwamem_pair.c:315-345) with ONE line per read —
// no supplementary / ALT line, no XA, no pa tag, no comment, no XR — and (round 3) the two records of a pair without any hit.  Everything else (unpaired ends, supplementary
// lines, XA, -a, -C, -V, single-end input) stays with the host's formatter (host_regs.cpp: aln2sam), and the host takes
// a pair back whenever the device flags one of its reads (CIGAR computed by the host, record longer than the staging
// buffer).  The host decides WHICH records are written here and all their numbers that involve floating point
// (MAPQ); the kernel turns numbers into bytes.
//
// One wavefront per 64 reads, in two phases.
//   1. a lane per read: the lane works out its record's alignment fields (sam_aln) and prints the short fields — FLAG, POS,
//      MAPQ, CIGAR, PNEXT, TLEN and the tags' numbers — into its own scratch row in LDS, as four "pieces" of text, and adds
//      up the record's length.  A wave prefix sum of the lengths and ONE atomic on the arena cursor place the 64 records
//      back to back (an atomic per record, 1.3 M on one address, cost more than the rest of the kernel).
//   2. the wave walks its records and copies each out across the lanes: QNAME, piece, RNAME, piece, SEQ, QUAL, piece, MD,
//      piece, RG — 64 consecutive bytes per store.
#include <hip/hip_runtime.h>
#include "device.h"

namespace mbw {

#define SAM_ROW 260      // bytes of LDS scratch per lane (65 dwords: rows start in different banks); a record whose short
                         // fields outgrow it (a CIGAR of dozens of operations) goes back to the host

struct SamAln {   // mem_aln_t as mem_reg2aln leaves it, for the fields mem_aln2sam reads
	long long pos;       // 0-based on its contig
	int rid, is_rev, n_cigar, rlen, NM, md_len;
	int clip5, clip3;    // soft clips added in front / behind the device's CIGAR
	int skip_front, skip_back;   // 1 when the first / last operation of the device's CIGAR (a deletion) is dropped
	const uint32_t *cig;
	const uint8_t *md;
	bool ok;
};

__device__ __forceinline__ SamAln sam_aln(const SamDesc &D, const AlnHdr *__restrict__ hdr, const uint8_t *__restrict__ pool, int req_base,
                                          long long l_pac, const long long *__restrict__ ann_off, int l_query)
{
	SamAln a;
	const AlnHdr h = hdr[req_base + D.req];
	a.ok = h.flags == 0;
	a.cig = (const uint32_t *)(pool + (size_t)h.pool_off * 4);
	a.md = (const uint8_t *)(a.cig + h.n_cigar);
	a.n_cigar = h.n_cigar; a.NM = h.NM & 0x3fffff; a.md_len = h.md_len;
	a.is_rev = D.rb >= l_pac;
	const long long p = D.rb < l_pac ? D.rb : D.re - 1;                 // bns_depos
	long long pos = a.is_rev ? (l_pac << 1) - 1 - p : p;
	a.skip_front = a.skip_back = 0;
	if (a.ok && a.n_cigar > 0) {   // squeeze out a leading or trailing deletion
		const uint32_t c0 = a.cig[0], c1 = a.cig[a.n_cigar - 1];
		if ((c0 & 0xf) == 2) { pos += c0 >> 4; a.skip_front = 1; }
		else if ((c1 & 0xf) == 2) a.skip_back = 1;
	}
	a.clip5 = a.clip3 = 0;
	if (D.qb != 0 || D.qe != l_query) {
		a.clip5 = a.is_rev ? l_query - D.qe : D.qb;
		a.clip3 = a.is_rev ? D.qb : l_query - D.qe;
	}
	a.rid = D.rid;
	a.pos = pos - ann_off[D.rid];
	int rl = 0;
	if (a.ok)
		for (int k = a.skip_front; k < a.n_cigar - a.skip_back; ++k) {
			const uint32_t c = a.cig[k];
			if ((c & 0xf) == 0 || (c & 0xf) == 2) rl += (int)(c >> 4);
		}
	a.rlen = rl;
	return a;
}

// ---- the lane's byte sink: its scratch row ----
struct Sink {
	uint8_t *row;
	int len;
	__device__ __forceinline__ void ch(char c) { if (len < SAM_ROW) row[len] = (uint8_t)c; ++len; }
	__device__ __forceinline__ void lit(const char *s) { for (; *s; ++s) ch(*s); }
	__device__ __forceinline__ void num32(uint32_t u)
	{
		char tmp[12];
		int n = 0;
		do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
		while (n) ch(tmp[--n]);
	}
	__device__ __forceinline__ void num(long long v)
	{
		unsigned long long u = v < 0 ? (unsigned long long)(-v) : (unsigned long long)v;
		if (v < 0) ch('-');
		if (u >> 32) {   // never on a real genome (contigs are shorter than 2^31), kept for the general case
			char tmp[24];
			int n = 0;
			do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
			while (n) ch(tmp[--n]);
		} else num32((uint32_t)u);
	}
	// clip5 S + the device's operations (without a dropped deletion) + clip3 S, as text
	__device__ __forceinline__ void cigar(const SamAln &a)
	{
		if (a.clip5) { num32(a.clip5); ch('S'); }
		for (int k = a.skip_front; k < a.n_cigar - a.skip_back; ++k) {
			const uint32_t c = a.cig[k];
			num32(c >> 4);
			ch("MIDSH"[c & 0xf]);
		}
		if (a.clip3) { num32(a.clip3); ch('S'); }
	}
};

template <class T>
__device__ __forceinline__ T bcast(T v, int src)   // the value lane `src` (wave-uniform) holds
{
	static_assert(sizeof(T) == 4 || sizeof(T) == 8, "");
	if (sizeof(T) == 4) {
		int x = __builtin_amdgcn_readlane(*(int *)&v, src);
		return *(T *)&x;
	}
	int lo = __builtin_amdgcn_readlane(((int *)&v)[0], src), hi = __builtin_amdgcn_readlane(((int *)&v)[1], src);
	unsigned long long x = (unsigned)lo | ((unsigned long long)(unsigned)hi << 32);
	return *(T *)&x;
}

__global__ void __launch_bounds__(64)
sam_emit_kernel(SamParams P, int n_reads, const SamDesc *__restrict__ desc, const int *__restrict__ req_base, const AlnHdr *__restrict__ hdr,
                const uint8_t *__restrict__ pool, const uint8_t *__restrict__ seq, const int64_t *__restrict__ off, const int *__restrict__ lens,
                const uint8_t *__restrict__ qual, const uint8_t *__restrict__ names, const int *__restrict__ name_off,
                const long long *__restrict__ ann_off, const char *__restrict__ ann_names, const int *__restrict__ ann_name_off,
                uint8_t *__restrict__ arena, unsigned long long arena_bytes, unsigned long long *arena_used, unsigned long long *out_off, int *out_len)
{
	__shared__ uint8_t rows[64 * SAM_ROW];
	const int lane = threadIdx.x;
	const int n_batch = (n_reads + 63) >> 6;
	for (int bt = blockIdx.x; bt < n_batch; bt += gridDim.x) {
		const int r = (bt << 6) + lane;
		// ---- phase 1: a lane per read ----
		int status = -2;                 // out_len of a read without a device record
		int total = 0;                   // bytes of the record
		int e_a = 0, e_b = 0, e_c = 0, e_d = 0;   // ends of the four pieces in the row
		int name_at = 0, name_len = 0, rn_at = 0, rn_len = 0, mn_at = 0, mn_len = 0, lq = 0, is_rev = 0, md_len = 0;
		long long sq_at = 0;
		const uint8_t *md = nullptr;
		if (r < n_reads) {
			const SamDesc D = desc[r];
			if (D.req == -3) {   // a read of a pair without any hit: "QNAME FLAG * 0 0 * * 0 0 SEQ QUAL AS:i:0 XS:i:0 [RG]" (src/bwamem.c:853-858, 928-929)
				lq = lens[r];
				Sink S;
				S.row = rows + lane * SAM_ROW; S.len = 0;
				name_at = name_off[r]; name_len = name_off[r + 1] - name_at;
				S.ch('\t'); S.num32((uint32_t)(D.flag & 0xffff)); S.ch('\t');
				e_a = S.len;
				S.lit("*\t0\t0\t*\t");
				e_b = S.len;
				S.lit("*\t0\t0\t");
				e_c = S.len;
				if (!P.has_qual) S.ch('*');
				e_d = S.len;
				S.lit("\tAS:i:0\tXS:i:0");
				if (P.rg_len) S.lit("\tRG:Z:");
				is_rev = 0; sq_at = off[r];
				total = name_len + S.len + lq + 1 + (P.has_qual ? lq : 0) + P.rg_len + 1;
				status = total;
			} else if (D.req >= 0) {
				const SamDesc M = desc[r ^ 1];
				const int unit = r >> 1, lm = lens[r ^ 1];
				lq = lens[r];
				const SamAln p = sam_aln(D, hdr, pool, req_base[unit], P.l_pac, ann_off, lq);
				const SamAln m = sam_aln(M, hdr, pool, req_base[unit], P.l_pac, ann_off, lm);
				status = -1;             // a CIGAR the device declined, or a row that overflows: the host formats the pair
				if (p.ok && m.ok) {
					Sink S;
					S.row = rows + lane * SAM_ROW; S.len = 0;
					const int flag = (D.flag & 0xffff) | 0x1 | (p.is_rev ? 0x10 : 0) | (m.is_rev ? 0x20 : 0);
					const int n_p = p.n_cigar - p.skip_front - p.skip_back + (p.clip5 ? 1 : 0) + (p.clip3 ? 1 : 0);
					const int n_m = m.n_cigar - m.skip_front - m.skip_back + (m.clip5 ? 1 : 0) + (m.clip3 ? 1 : 0);
					name_at = name_off[r]; name_len = name_off[r + 1] - name_at;
					rn_at = ann_name_off[p.rid]; rn_len = ann_name_off[p.rid + 1] - rn_at;
					S.ch('\t'); S.num32(flag); S.ch('\t');
					e_a = S.len;
					S.ch('\t'); S.num(p.pos + 1); S.ch('\t'); S.num32(D.mapq); S.ch('\t');
					if (n_p) S.cigar(p); else S.ch('*');
					S.ch('\t');
					if (p.rid == m.rid) S.ch('=');
					else { mn_at = ann_name_off[m.rid]; mn_len = ann_name_off[m.rid + 1] - mn_at; }
					e_b = S.len;
					S.ch('\t'); S.num(m.pos + 1); S.ch('\t');
					if (p.rid == m.rid) {
						const long long p0 = p.pos + (p.is_rev ? p.rlen - 1 : 0), p1 = m.pos + (m.is_rev ? m.rlen - 1 : 0);
						if (n_m == 0 || n_p == 0) S.ch('0');
						else S.num(-(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
					} else S.ch('0');
					S.ch('\t');
					e_c = S.len;
					if (!P.has_qual) S.ch('*');   // (which = 0: SEQ and QUAL are never trimmed)
					if (n_p) {
						S.lit("\tNM:i:"); S.num32(p.NM);
						S.lit("\tMD:Z:");
						md = p.md; md_len = p.md_len;
					}
					e_d = S.len;
					if (n_m) { S.lit("\tMC:Z:"); S.cigar(m); }
					if (D.score >= 0) { S.lit("\tAS:i:"); S.num32(D.score); }
					if (D.sub >= 0) { S.lit("\tXS:i:"); S.num32(D.sub); }
					if (P.rg_len) S.lit("\tRG:Z:");
					is_rev = p.is_rev; sq_at = off[r];
					if (S.len <= SAM_ROW) {
						// QNAME a RNAME b [mate RNAME] c SEQ \t [QUAL] d MD rest RG \n
						total = name_len + S.len + rn_len + mn_len + lq + 1 + (P.has_qual ? lq : 0) + md_len + P.rg_len + 1;
						status = total;
					}
				}
			}
		}
		// place the wave's records back to back: exclusive prefix sum of the lengths, one atomic
		int incl = total;
		for (int d = 1; d < 64; d <<= 1) {
			const int up = __shfl_up(incl, d);
			if (lane >= d) incl += up;
		}
		const int wave_total = __shfl(incl, 63);
		unsigned long long base = 0;
		if (lane == 0 && wave_total) base = atomicAdd(arena_used, (unsigned long long)wave_total);
		base = bcast(base, 0);
		if (base + (unsigned long long)wave_total > arena_bytes) { if (total) { status = -1; total = 0; } }   // (the host sizes the arena for every record)
		const unsigned long long at_mine = base + (unsigned long long)(incl - total);
		if (r < n_reads) { out_len[r] = status; if (total) out_off[r] = at_mine; }
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the rows are read across the lanes from here on
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		// ---- phase 2: the wave copies the records out, one after the other ----
		unsigned long long todo = __ballot(total > 0);
		while (todo) {
			const int i = __builtin_ctzll(todo);
			todo &= todo - 1;
			uint8_t *o = arena + bcast(at_mine, i);
			const uint8_t *row = rows + i * SAM_ROW;
			const int ea = bcast(e_a, i), eb = bcast(e_b, i), ec = bcast(e_c, i), ed = bcast(e_d, i);
			const int l = bcast(lq, i), rev = bcast(is_rev, i);
			auto spread = [&](int n, auto f) {
				for (int k = lane; k < n; k += 64) o[k] = (uint8_t)f(k);
				o += n;
			};
			{ const uint8_t *nm = names + bcast(name_at, i); spread(bcast(name_len, i), [&](int k) { return nm[k]; }); }
			spread(ea, [&](int k) { return row[k]; });
			{ const char *cn = ann_names + bcast(rn_at, i); spread(bcast(rn_len, i), [&](int k) { return cn[k]; }); }
			spread(eb - ea, [&](int k) { return row[ea + k]; });
			{ const char *cn = ann_names + bcast(mn_at, i); spread(bcast(mn_len, i), [&](int k) { return cn[k]; }); }
			spread(ec - eb, [&](int k) { return row[eb + k]; });
			const long long so = bcast(sq_at, i);
			{
				const uint8_t *sq = seq + so;
				if (!rev) spread(l, [&](int k) { return "ACGTN"[sq[k] > 4 ? 4 : sq[k]]; });
				else spread(l, [&](int k) { const int c = sq[l - 1 - k]; return "TGCAN"[c > 4 ? 4 : c]; });
			}
			spread(1, [&](int) { return '\t'; });
			if (P.has_qual) {
				const uint8_t *ql = qual + so;
				if (!rev) spread(l, [&](int k) { return ql[k]; });
				else spread(l, [&](int k) { return ql[l - 1 - k]; });
			}
			spread(ed - ec, [&](int k) { return row[ec + k]; });
			{ const uint8_t *mdp = bcast(md, i); spread(bcast(md_len, i), [&](int k) { return mdp[k]; }); }
			const int rest = bcast(status, i) - (int)(o - (arena + bcast(at_mine, i))) - P.rg_len - 1;   // what is left of the row
			spread(rest, [&](int k) { return row[ed + k]; });
			spread(P.rg_len + 1, [&](int k) { return k < P.rg_len ? P.rg[k] : '\n'; });
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the rows are rewritten by the wave's next batch
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
}

void launch_sam_emit(void *stream, const SamParams &P, int n_reads, const SamDesc *d_desc, const int *d_req_base, const AlnHdr *d_hdr,
                     const uint8_t *d_pool, const uint8_t *d_seq, const int64_t *d_off, const int *d_len, const uint8_t *d_qual,
                     const uint8_t *d_names, const int *d_name_off, const int64_t *d_ann_off, const char *d_ann_names, const int *d_ann_name_off,
                     uint8_t *d_arena, size_t arena_bytes, unsigned long long *d_arena_used, unsigned long long *d_out_off, int *d_out_len)
{
	if (n_reads <= 0) return;
	const int n_batch = (n_reads + 63) >> 6;
	int blocks = n_batch < 256 * 8 ? n_batch : 256 * 8;   // 16.6 KB of LDS per wave: nine waves per CU
	hipLaunchKernelGGL(sam_emit_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, P, n_reads, d_desc, d_req_base, d_hdr, d_pool, d_seq, d_off,
	                   d_len, d_qual, d_names, d_name_off, (const long long *)d_ann_off, d_ann_names, d_ann_name_off, d_arena,
	                   (unsigned long long)arena_bytes, d_arena_used, d_out_off, d_out_len);
}

} // namespace mbw
