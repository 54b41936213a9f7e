// aln_kernel.hip — final global re-alignment (CIGAR + NM + MD) of alignment regions on the device.
//
// Device counterpart of the DP part of mem_reg2aln (src/bwamem.c:1106-1122): bwa_gen_cigar2 (src/bwa.c:121-207)
// = fetch of the reference window (src/bntseq.c:398-419), banded global alignment with traceback
// (ksw_global2, src/ksw.c:504-606) under mem_reg2aln's band-doubling loop, and the NM / MD walk.
// One wavefront per region.  Rows are sequential, lanes run across the band columns; the F recurrence
// F(i,j+1) = max(F(i,j), M(i,j) - o_ins) - e_ins depends on M only, so it is the same max-plus prefix scan as in
// wave_ext.cuh.  The direction bytes live in LDS; the traceback and the MD walk are wave-uniform scalar loops over
// LDS (cheap because thousands of wavefronts do theirs concurrently).  Regions whose band matrix does not fit the
// LDS budget are flagged and re-done by the host (rare: long gaps).
#include <hip/hip_runtime.h>
#include <algorithm>
#include "device.h"
#include "wave_ext.cuh"

namespace mbw {

#define ALN_WAVES 1
// direction bytes per wavefront: grows with the read length (a band of ~40 columns at every row), at least 12 KB
__host__ __device__ inline int aln_zcap(int max_len) { int z = 80 * (max_len + 32); z = (z + 255) & ~255; return z < 12288 ? 12288 : z; }
// ... and of the narrow-band variant: 32 columns at every row
__host__ __device__ inline int aln_zcap_small(int max_len) { int z = 32 * (max_len + 32); return (z + 255) & ~255; }
#define ALN_MDCAP 768
#define ALN_CIGCAP 96
#define ALN_NEG (-0x40000000)

__device__ __forceinline__ int wave_sum_int(int v)
{
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	return v;
}

struct AlnLds {
	int *H, *E, *dummy;
	uint8_t *q, *t, *z, *md;
	uint32_t *cig;
};
#define ALN_PAD 64   // cells behind H / E (and bytes behind q) that the lanes of a strip may read past the row

// decimal digits of v (v >= 0) appended at md[len...]; every lane runs it, lane 0 stores
__device__ __forceinline__ int put_num(uint8_t *md, int len, int v, int lane)
{
	char buf[12];
	int n = 0;
	if (v == 0) buf[n++] = '0';
	while (v > 0) { buf[n++] = '0' + v % 10; v /= 10; }
	if (lane == 0)
		for (int k = 0; k < n; ++k)
			if (len + k < ALN_MDCAP) md[len + k] = buf[n - 1 - k];
	return len + n;
}

// Three instantiations, launched one after the other.  A classification pass (aln_classify_kernel) sorts the request numbers
// into one list per kind first, and the waves of a kind walk their list with a grid-sized stride — nobody launches a
// workgroup per request only to find that the request belongs to another variant.
//   KIND 0  the requests whose query and reference window have the same length.  With band 0 the reference takes its no-DP
//           shortcut (src/bwa.c:143-151); with a band it runs ksw_global2, but when the ungapped alignment loses no more than
//           a + (o_del + e_del) + (o_ins + e_ins) against a perfect match — three mismatches under the default scores —
//           the DP can only return that same alignment, in every round of mem_reg2aln's loop: a path with a gap has an
//           insertion AND a deletion, so at most lq - 1 aligned columns of at most `a` each, and ksw_global2's traceback
//           prefers the diagonal move on ties (src/ksw.c:551-554), which keeps it on the main diagonal whenever that is
//           co-optimal.  Score, CIGAR (lqM), NM and MD are then those of the ungapped alignment, and the wave writes them
//           without a direction matrix: 1.1 KB of LDS instead of 17 KB, as many waves as the CU has slots for.  The
//           same-length requests that lose more are appended to the DP list.
//   KIND 1  DP with a direction matrix of at most 32 columns per row (8.7 KB of LDS per wave: 16 waves per CU instead of 9 —
//           a row is a chain of dependent LDS and cross-lane steps, so the waves per SIMD set the rate).  A request whose
//           band outgrows that in one of mem_reg2aln's rounds is marked (flags = 2) and left to
//   KIND 2  the full-size variant, which starts those requests again from the first round.
#define ALN_PENDING 2
// counters[]: [0] bytes of the result pool handed out, [ALN_CNT + k] length of the request list of kind k
#define ALN_CNT 8
#define ALN_SLAB 1024u

__device__ __forceinline__ bool aln_invalid(const AlnParams &P, const AlnReq &R, int max_len, int tcap)
{
	const int lq = R.qe - R.qb;
	const long long rlen64 = R.re - R.rb;
	const bool bridging = R.rb < P.l_pac && R.re > P.l_pac;
	return lq <= 0 || rlen64 <= 0 || bridging || lq > max_len || rlen64 > tcap;
}
__device__ __forceinline__ bool aln_same_len(const AlnParams &P, const AlnReq &R, int max_len, int tcap)
{
	return !aln_invalid(P, R, max_len, tcap) && R.re - R.rb == R.qe - R.qb;
}

// request numbers -> lists[0 .. n_req) (same length: no DP, or not yet known) and the DP list; one atomic per wave and list
// dp_kind: the list DP requests start in (1 = narrow variant first, the product's dispatch; 2 = straight to the full-size one)
__global__ void __launch_bounds__(256) aln_classify_kernel(AlnParams P, int n_req, const AlnReq *__restrict__ reqs, int max_len, int tcap,
                                                           int *__restrict__ lists, unsigned long long *counters, int dp_kind)
{
	const int rq = blockIdx.x * 256 + threadIdx.x;
	const bool live = rq < n_req && reqs[rq < n_req ? rq : 0].read >= 0;   // (read < 0: an unused slot of the request array, pair_kernel.hip)
	bool nodp = false;
	if (live) nodp = aln_same_len(P, reqs[rq], max_len, tcap);
	const unsigned long long below = (1ull << (threadIdx.x & 63)) - 1;
	for (int pass = 0; pass < 2; ++pass) {
		const bool mine = live && (nodp == (pass == 0));
		const int kind = pass == 0 ? 0 : dp_kind;
		const unsigned long long m = __ballot(mine);
		if (!m) continue;
		unsigned long long base = 0;
		if ((threadIdx.x & 63) == __ffsll((long long)m) - 1) base = atomicAdd(&counters[ALN_CNT + kind], (unsigned long long)__popcll(m));
		base = __shfl(base, __ffsll((long long)m) - 1);
		if (mine) lists[(size_t)kind * n_req + base + __popcll(m & below)] = rq;
	}
}

template <int KIND>
__device__ __forceinline__ void aln_one(int rq, const AlnParams &P, const WxParams &X, int n_req, const AlnReq *__restrict__ reqs,
                                        const uint8_t *__restrict__ seq, const int64_t *__restrict__ off, const uint8_t *__restrict__ pac,
                                        const int *__restrict__ gaptab, AlnHdr *__restrict__ hdr, uint8_t *__restrict__ pool,
                                        unsigned long long *counters, unsigned long long pool_bytes, int max_len, int tcap, int *lists, int dp_kind,
                                        unsigned long long &slab_at, unsigned &slab_left)
{
	extern __shared__ int lds_raw[];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	// carve this wavefront's LDS
	constexpr bool FAST = KIND == 0;
	const int ALN_ZCAP = KIND == 1 ? aln_zcap_small(max_len) : aln_zcap(max_len);
	AlnLds L;
	if (FAST) {
		const size_t per_wave = (size_t)2 * ((max_len + 3) & ~3) + ALN_MDCAP + 16;
		uint8_t *base = (uint8_t *)lds_raw + (size_t)wave * per_wave;
		L.H = L.E = L.dummy = nullptr; L.z = nullptr;
		L.cig = (uint32_t *)base;
		L.q = base + 16;
		L.t = L.q + ((max_len + 3) & ~3);
		L.md = L.t + ((max_len + 3) & ~3);
	} else {
		const size_t per_wave = (size_t)2 * (max_len + 2 + ALN_PAD) * 4 + 4 + ((max_len + 3) & ~3) + ((tcap + 3) & ~3) + ALN_ZCAP + ALN_MDCAP + ALN_CIGCAP * 4;
		uint8_t *base = (uint8_t *)lds_raw + (size_t)wave * per_wave;
		L.H = (int *)base; L.E = L.H + (max_len + 2 + ALN_PAD);
		L.dummy = L.E + (max_len + 2 + ALN_PAD);
		L.cig = (uint32_t *)(L.dummy + 1);
		L.q = (uint8_t *)(L.cig + ALN_CIGCAP);
		L.t = L.q + ((max_len + 3) & ~3);
		L.z = L.t + ((tcap + 3) & ~3);
		L.md = L.z + ALN_ZCAP;
	}

	// the request is the same in every lane but arrives through vector loads: pin it to scalar registers, or the DP
	// loop's bookkeeping (band limits, strip counts, loop tests) is compiled into vector instructions
	AlnReq R = reqs[rq];
	if (KIND != 0) {   // (the no-DP variant has no such loop, and pinned it only unrolls its staging loops into 126 VGPRs)
		const unsigned long long rb = (unsigned long long)R.rb, re = (unsigned long long)R.re;
		R.rb = (int64_t)((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(rb >> 32)) << 32 |
		                 (unsigned)__builtin_amdgcn_readfirstlane((int)rb));
		R.re = (int64_t)((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(re >> 32)) << 32 |
		                 (unsigned)__builtin_amdgcn_readfirstlane((int)re));
		R.read = __builtin_amdgcn_readfirstlane(R.read); R.qb = __builtin_amdgcn_readfirstlane(R.qb);
		R.qe = __builtin_amdgcn_readfirstlane(R.qe); R.w2 = __builtin_amdgcn_readfirstlane(R.w2);
		R.truesc = __builtin_amdgcn_readfirstlane(R.truesc);
	}
	const int lq = R.qe - R.qb;
	const long long rlen64 = R.re - R.rb;
	AlnHdr out;
	out.score = 0; out.NM = -1; out.n_cigar = 0; out.md_len = 0; out.pool_off = 0; out.flags = 0;
	const bool invalid = aln_invalid(P, R, max_len, tcap);
	if (invalid) {
		out.flags = 1;   // host fallback (also reproduces the reference's rejection cases)
		if (lane == 0) hdr[rq] = out;
		return;
	}
	const int rlen = (int)rlen64;
	const bool rev = R.rb >= P.l_pac;
	const uint8_t *rd = seq + off[R.read];
	// stage the (possibly reversed) query and the reference window: on the reverse strand both are flipped so that
	// gaps end up left-aligned on the forward strand (src/bwa.c:136-141)
	for (int j = lane; j < lq; j += 64) L.q[j] = rev ? rd[R.qe - 1 - j] : rd[R.qb + j];
	{
		const long long f0 = rev ? (P.l_pac << 1) - R.re : R.rb;   // forward-strand start of the window
		for (int i = lane; i < rlen; i += 64) {
			long long f = f0 + i;
			int b = (pac[f >> 2] >> ((~f & 3) << 1)) & 3;
			L.t[i] = rev ? 3 - b : b;
		}
	}
	__builtin_amdgcn_wave_barrier();

	const int oe_del = X.o_del + X.e_del, oe_ins = X.o_ins + X.e_ins, e_del = X.e_del, e_ins = X.e_ins;
	// scoring matrix rows as packed bytes (see wave_ext.cuh): byte q of {phi[t], plo[t]} = mat[t][q]
	uint32_t plo[5], phi[5];
#pragma unroll
	for (int t = 0; t < 5; ++t) {
		plo[t] = (uint32_t)(uint8_t)X.mat[t * 5] | (uint32_t)(uint8_t)X.mat[t * 5 + 1] << 8 | (uint32_t)(uint8_t)X.mat[t * 5 + 2] << 16 |
		         (uint32_t)(uint8_t)X.mat[t * 5 + 3] << 24;
		phi[t] = (uint32_t)(uint8_t)X.mat[t * 5 + 4];
	}
	const int wmax4 = P.w << 2;
	int w2 = R.w2, last_sc = -(1 << 30), score = 0, n_cig = 0;
	bool fallback = false;
	for (int it = 0;; ) {
		w2 = w2 < wmax4 ? w2 : wmax4;
		if (FAST) {   // ungapped: lq == rlen, and either w2 == 0 (src/bwa.c:143-151) or the DP cannot do better (see above)
			int s = 0;
			for (int i = lane; i < lq; i += 64) s += X.mat[L.t[i] * 5 + L.q[i]];
			score = wave_sum_int(s);
			if (w2 != 0) {
				int amax = X.mat[0];
				for (int k = 1; k < 25; ++k) amax = X.mat[k] > amax ? X.mat[k] : amax;
				if (amax <= 0 || (long long)lq * amax - score > (long long)amax + oe_del + oe_ins) {   // the DP decides: to its list
					if (lane == 0) lists[(size_t)dp_kind * n_req + atomicAdd(&counters[ALN_CNT + dp_kind], 1ull)] = rq;
					return;
				}
			}
			n_cig = 1;
			if (lane == 0) L.cig[0] = (uint32_t)lq << 4;
			// the other lanes read it below: without this the compiler sinks their load into the else-side of the
			// lane-0 branch, which the hardware runs first
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			break;   // a second round of mem_reg2aln's loop would take the same shortcut and stop on score == last_sc
		} else {
			int d_len = rlen - lq; d_len = d_len < 0 ? -d_len : d_len;
			int w = (gaptab[lq] + d_len + 1) >> 1;
			w = w < w2 ? w : w2;
			const int min_w = d_len + 3;
			w = w > min_w ? w : min_w;
			const int n_col = lq < 2 * w + 1 ? lq : 2 * w + 1;
			if ((long long)n_col * rlen > ALN_ZCAP) { fallback = true; break; }   // KIND 1: to the full-size variant; KIND 2: to the host
			// ---- banded global DP (src/ksw.c:523-589) ----
			for (int j = lane; j <= lq; j += 64) {
				L.H[j] = j == 0 ? 0 : (j <= w ? -(X.o_ins + e_ins * j) : ALN_NEG);
				L.E[j] = ALN_NEG;
			}
			__builtin_amdgcn_wave_barrier();
			int vslo = 0, vshi = 0;   // the scoring-matrix rows of the next 64 target bases, one per lane (as in wave_ext.cuh)
			for (int i = 0; i < rlen; ++i) {
				if ((i & 63) == 0) {
					const int tb = i + lane < rlen ? (int)L.t[i + lane] : 4;
					vslo = (int)(tb == 0 ? plo[0] : tb == 1 ? plo[1] : tb == 2 ? plo[2] : tb == 3 ? plo[3] : plo[4]);
					vshi = (int)(tb == 0 ? phi[0] : tb == 1 ? phi[1] : tb == 2 ? phi[2] : tb == 3 ? phi[3] : phi[4]);
				}
				const uint32_t slo = (uint32_t)__builtin_amdgcn_readlane(vslo, i & 63), shi = (uint32_t)__builtin_amdgcn_readlane(vshi, i & 63);
				const int beg = i > w ? i - w : 0, end = i + w + 1 < lq ? i + w + 1 : lq;
				const int hleft0 = beg == 0 ? -(X.o_del + e_del * (i + 1)) : ALN_NEG;
				const int n_row = end - beg;   // columns of the row (negative: the band has left the query)
				int A = ALN_NEG;   // running max of g_k + e_ins over the columns already done (F(i,beg) = -inf)
				int h_carry = hleft0;
				uint8_t *zi = L.z + (size_t)i * n_col;
				// strips of 64 lanes counted from the row's first column, as in wave_ext.cuh: lane k owns column beg + k, reads
				// eh[j] and writes eh[j].h = h of column j - 1 (lane 0: the first-column value; one past the last column: eh[end].h)
				// and eh[j].e — no bounds tests, what lies past the row goes to a write-only cell
				for (int s0 = 0; s0 <= n_row; s0 += 64) {
					const int j = beg + s0 + lane;
					const bool act = s0 + lane < n_row, wr = s0 + lane <= n_row;
					const int diag = L.H[j], e = L.E[j];
					const int qraw = (int)L.q[j];
					const int qb = act ? qraw : 4;
					const int sc = (int)(int8_t)__builtin_amdgcn_perm(shi, slo, (uint32_t)qb | 0x0c0c0c00u);
					const int m = diag + sc;
					const int g = act ? m - oe_ins + j * e_ins : ALN_NEG - (1 << 28);
					const int incl = wx_scan_max(g);
					const int excl = wx_dpp<0x138, 0xf>(ALN_NEG - (1 << 28), incl);
					const int f = max(A, excl + e_ins) - j * e_ins;     // F(i,j)
					uint8_t d = m >= e ? 0 : 1;
					int h = m >= e ? m : e;
					if (h < f) { d = 2; h = f; }
					int t = m - oe_del;
					int e2 = e - e_del;
					if (e2 > t) d |= 1 << 2; else e2 = t;
					t = m - oe_ins;
					if (f - e_ins > t) d |= 2 << 4;
					const int h_prev = wx_dpp<0x138, 0xf>(h_carry, h);
					(wr ? L.H + j : L.dummy)[0] = h_prev;
					(wr ? L.E + j : L.dummy)[0] = act ? e2 : ALN_NEG;
					(act ? zi + (j - beg) : (uint8_t *)L.dummy)[0] = d;
					A = max(A, __builtin_amdgcn_readlane(incl, 63) + e_ins);
					h_carry = __builtin_amdgcn_readlane(h, 63);
				}
				if (n_row < 0 && lane == 0) { L.H[end] = hleft0; L.E[end] = ALN_NEG; }
				__builtin_amdgcn_wave_barrier();
			}
			score = L.H[lq];
			// ---- traceback (src/ksw.c:590-603): wave-uniform walk, lane 0 stores ----
			{
				int which = 0, i = rlen - 1, k = (i + w + 1 < lq ? i + w + 1 : lq) - 1, n = 0;
				uint32_t cur = 0;   // current run: len << 4 | op, built from the end of the alignment
				bool ovf = false;
				auto push = [&](uint32_t op, uint32_t len) {
					if (cur && (cur & 0xf) == op) cur += len << 4;
					else {
						if (cur) { if (n < ALN_CIGCAP) { if (lane == 0) L.cig[n] = cur; } else ovf = true; ++n; }
						cur = len << 4 | op;
					}
				};
				while (i >= 0 && k >= 0) {
					// the walk is the same in every lane: keep it in scalar registers (readfirstlane), off the vector ALU
					which = __builtin_amdgcn_readfirstlane((int)L.z[(size_t)i * n_col + (k - (i > w ? i - w : 0))]) >> (which << 1) & 3;
					if (which == 0) { push(0, 1); --i; --k; }
					else if (which == 1) { push(2, 1); --i; }
					else { push(1, 1); --k; }
				}
				if (i >= 0) push(2, i + 1);
				if (k >= 0) push(1, k + 1);
				if (cur) { if (n < ALN_CIGCAP) { if (lane == 0) L.cig[n] = cur; } else ovf = true; ++n; }
				if (ovf) { fallback = true; break; }
				__builtin_amdgcn_wave_barrier();
				// reverse into alignment order
				for (int a = lane; a < n / 2; a += 64) { uint32_t x = L.cig[a]; L.cig[a] = L.cig[n - 1 - a]; L.cig[n - 1 - a] = x; }
				__builtin_amdgcn_wave_barrier();
				n_cig = n;
			}
		}
		if (score == last_sc || w2 == wmax4) break;   // src/bwamem.c:1118
		last_sc = score;
		w2 <<= 1;
		if (!(++it < 3 && score < R.truesc - P.a)) break;
	}
	if (fallback) {
		if (KIND == 1) {   // to the full-size variant: append to its list
			if (lane == 0) lists[(size_t)2 * n_req + atomicAdd(&counters[ALN_CNT + 2], 1ull)] = rq;
			return;
		}
		out.flags = 1;
		if (lane == 0) hdr[rq] = out;
		return;
	}
	// ---- NM and MD (src/bwa.c:168-199) ----
	int md_len = 0, n_mm = 0, n_gap = 0;
	{
		const char *int2base = rev ? "TGCAN" : "ACGTN";
		int x = 0, y = 0, u = 0;
		for (int k = 0; k < n_cig; ++k) {
			const uint32_t c = L.cig[k];
			const int op = c & 0xf, len = (int)(c >> 4);
			if (op == 0) {
				for (int i0 = 0; i0 < len; i0 += 64) {
					const int i = i0 + lane;
					unsigned long long mm = __ballot(i < len && L.q[x + i] != L.t[y + i]);
					int cur = 0;                      // positions of this chunk already accounted for
					const int chunk = len - i0 < 64 ? len - i0 : 64;
					while (mm) {
						const int p = __ffsll((long long)mm) - 1;
						mm &= mm - 1;
						u += p - cur;
						md_len = put_num(L.md, md_len, u, lane);
						if (lane == 0 && md_len < ALN_MDCAP) L.md[md_len] = int2base[L.t[y + i0 + p]];
						++md_len; ++n_mm; u = 0; cur = p + 1;
					}
					u += chunk - cur;
				}
				x += len; y += len;
			} else if (op == 2) {
				if (k > 0 && k < n_cig - 1) {
					md_len = put_num(L.md, md_len, u, lane);
					if (lane == 0 && md_len < ALN_MDCAP) L.md[md_len] = '^';
					++md_len;
					for (int i = lane; i < len; i += 64)
						if (md_len + i < ALN_MDCAP) L.md[md_len + i] = int2base[L.t[y + i]];
					md_len += len;
					u = 0; n_gap += len;
				}
				y += len;
			} else { x += len; n_gap += len; }
		}
		md_len = put_num(L.md, md_len, u, lane);
	}
	if (md_len > ALN_MDCAP) {
		out.flags = 1;
		if (lane == 0) hdr[rq] = out;
		return;
	}
	__builtin_amdgcn_wave_barrier();
	// ---- hand the record over: [cigar u32 x n][md bytes] in the pool ----
	const unsigned need = (unsigned)n_cig * 4 + (((unsigned)md_len + 3) & ~3u);
	unsigned long long at = 0;
	// the pool is handed out in slabs of ALN_SLAB bytes per wave: one atomic on the pool cursor per ~25 requests instead of one per
	// request (an atomic on one address costs ~10 ns of a queue the whole chip shares: 680 000 of them were half of this stage)
	if (need <= slab_left) { at = slab_at; slab_at += need; slab_left -= need; }
	else {
		const unsigned take = need > ALN_SLAB ? need : ALN_SLAB;
		if (lane == 0) at = atomicAdd(&counters[0], (unsigned long long)take);
		at = __shfl(at, 0);
		slab_at = at + need; slab_left = take - need;
	}
	if (at + need > pool_bytes) {
		out.flags = 1;
		if (lane == 0) hdr[rq] = out;
		return;
	}
	uint32_t *pc = (uint32_t *)(pool + at);
	for (int k = lane; k < n_cig; k += 64) pc[k] = L.cig[k];
	uint8_t *pm = pool + at + (size_t)n_cig * 4;
	for (int k = lane; k < md_len; k += 64) pm[k] = L.md[k];
	out.score = score; out.NM = n_mm + n_gap; out.n_cigar = n_cig; out.md_len = md_len; out.pool_off = (uint32_t)(at >> 2); out.flags = 0;
	if (lane == 0) hdr[rq] = out;
}

template <int KIND>
__global__ void __launch_bounds__(64 * ALN_WAVES)
aln_kernel(AlnParams P, WxParams X, int n_req, const AlnReq *__restrict__ reqs, const uint8_t *__restrict__ seq,
           const int64_t *__restrict__ off, const uint8_t *__restrict__ pac, const int *__restrict__ gaptab, AlnHdr *__restrict__ hdr,
           uint8_t *__restrict__ pool, unsigned long long *counters, unsigned long long pool_bytes, int max_len, int tcap, int *lists, int dp_kind)
{
	const int wave = threadIdx.x >> 6;
	// the lists of KIND 1 and 2 are added to by the launches before them on the stream
	const int n = (int)counters[ALN_CNT + KIND];
	const int *mine = lists + (size_t)KIND * n_req;
	unsigned long long slab_at = 0;
	unsigned slab_left = 0;
	for (int k = blockIdx.x * ALN_WAVES + wave; k < n; k += gridDim.x * ALN_WAVES) {
		aln_one<KIND>(KIND != 0 ? __builtin_amdgcn_readfirstlane(mine[k]) : mine[k], P, X, n_req, reqs, seq, off, pac, gaptab, hdr, pool, counters, pool_bytes, max_len, tcap, lists, dp_kind,
		              slab_at, slab_left);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the wave's LDS slice is reused by its next request
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
}

size_t aln_lds_per_block(int max_len, int tcap)
{
	const int ALN_ZCAP = aln_zcap(max_len);
	size_t per_wave = (size_t)2 * (max_len + 2 + ALN_PAD) * 4 + 4 + ((max_len + 3) & ~3) + ((tcap + 3) & ~3) + ALN_ZCAP + ALN_MDCAP + ALN_CIGCAP * 4;
	return per_wave * ALN_WAVES;
}

void launch_aln(void *stream, const AlnParams &P, const ExtParams &ep, int n_req, const AlnReq *d_req, const uint8_t *d_seq,
                const int64_t *d_off, const uint8_t *d_pac, const int *d_gaptab, AlnHdr *d_hdr, uint8_t *d_pool,
                unsigned long long *d_counters, size_t pool_bytes, int max_len, int tcap, int *d_lists, bool wide_only)
{
	if (n_req <= 0) return;
	WxParams X;
	for (int i = 0; i < 25; ++i) X.mat[i] = ep.mat[i];
	X.o_del = ep.o_del; X.e_del = ep.e_del; X.o_ins = ep.o_ins; X.e_ins = ep.e_ins; X.zdrop = ep.zdrop;
	const size_t shmem = aln_lds_per_block(max_len, tcap);
	const size_t shmem_small = shmem - (size_t)(aln_zcap(max_len) - aln_zcap_small(max_len)) * ALN_WAVES;
	const size_t shmem_fast = ((size_t)2 * ((max_len + 3) & ~3) + ALN_MDCAP + 16) * ALN_WAVES;
	if (shmem > 64 * 1024 && hipFuncSetAttribute((const void *)aln_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess)
		die("aln_kernel: cannot reserve %zu bytes of LDS", shmem);
	if (shmem_small > 64 * 1024 && hipFuncSetAttribute((const void *)aln_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem_small) != hipSuccess)
		die("aln_kernel: cannot reserve %zu bytes of LDS", shmem_small);
	if (shmem_fast > 64 * 1024) die("aln_kernel: reads of %d bp do not fit the LDS staging buffers", max_len);
	static int s_cus = 0;
	if (!s_cus) {
		int dev = 0;
		hipDeviceProp_t prop;
		if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) die("aln_kernel: cannot query the device");
		s_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	}
	// enough workgroups to fill every wave slot the variant's LDS footprint leaves, twice over (the requests of a list
	// differ in length, so a wave takes the next one as soon as it is done)
	auto grid_for = [&](size_t lds) {
		long long per_cu = (long long)(160 * 1024 / std::max<size_t>(lds, 1));
		per_cu = std::max(1ll, std::min(per_cu, 32ll / ALN_WAVES));
		return dim3((unsigned)std::min<long long>((long long)s_cus * per_cu * 2, ((long long)n_req + ALN_WAVES - 1) / ALN_WAVES));
	};
	const dim3 block(64 * ALN_WAVES);
	hipStream_t st = (hipStream_t)stream;
	hipLaunchKernelGGL(aln_classify_kernel, dim3((n_req + 255) / 256), dim3(256), 0, st, P, n_req, d_req, max_len, tcap, d_lists, d_counters,
	                   wide_only ? 2 : 1);
	hipLaunchKernelGGL(aln_kernel<0>, grid_for(shmem_fast), block, shmem_fast, st, P, X, n_req, d_req, d_seq, d_off, d_pac, d_gaptab, d_hdr, d_pool,
	                   d_counters, (unsigned long long)pool_bytes, max_len, tcap, d_lists, wide_only ? 2 : 1);
	hipLaunchKernelGGL(aln_kernel<1>, grid_for(shmem_small), block, shmem_small, st, P, X, n_req, d_req, d_seq, d_off, d_pac, d_gaptab, d_hdr, d_pool,
	                   d_counters, (unsigned long long)pool_bytes, max_len, tcap, d_lists, wide_only ? 2 : 1);
	hipLaunchKernelGGL(aln_kernel<2>, grid_for(shmem), block, shmem, st, P, X, n_req, d_req, d_seq, d_off, d_pac, d_gaptab, d_hdr, d_pool,
	                   d_counters, (unsigned long long)pool_bytes, max_len, tcap, d_lists, wide_only ? 2 : 1);
	if (hipGetLastError() != hipSuccess) die("aln_kernel: launch failed");
}

} // namespace mbw
