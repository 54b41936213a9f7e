// msw_kernel.hip — mate-rescue local alignment on the device: ksw_align2() as mem_matesw() calls it
// (src/bwamem_pair.c:150-177 -> src/ksw.c:321-356, kernels ksw_u8 src/ksw.c:111-237 and ksw_i16 src/ksw.c:239-319).
//
// What has to be reproduced is not textbook Smith-Waterman.  The reference runs Farrar's striped SSE2 kernel, in which a
// query of qlen bases is cut into P segments of slen = ceil(qlen / P) positions (P = 16 byte lanes when qlen * a < 250,
// else 8 word lanes) and
//   * inside the main loop F only propagates within a segment (every lane starts a row with F = 0);
//   * E(i+1,j) is computed there, from the H that main loop sees ("we disallow adjacent insertion and then
//     deletion", src/ksw.c:176) — i.e. from Hpre = max(Hdiag + s, E, Fseg), not from the final H;
//   * the lazy-F loop afterwards carries F across segment borders and only raises H (exactly: its early exit
//     never drops a carry that could still matter);
//   * positions qlen .. slen*P-1 are padding that scores 0 against everything, and they do take part in the row maximum;
//   * the row maximum feeds a run-length list b[] whose "consecutive row" test looks at the row stored in the last
//     entry, not at the previous row.
// All of this is observable in (score, te, qe, score2, te2), so the kernel keeps two F values per cell — Fseg (reset
// at every segment start) and Ffull (never reset) — and computes
//     Hpre = max(Hdiag + s, E, Fseg)      H = max(Hpre, Ffull)
//     E'   = max(E - e_del, Hpre - oe_del, 0)
//     F*'  = max(F* - e_ins, Hpre - oe_ins, 0)
// which is value-for-value what the striped kernel leaves in its H / E arrays once a row is finished.  No value can
// saturate in the byte flavour (qlen * a < 250 and shift = -min(mat)); a request whose scores could reach the 8-bit
// ceiling is flagged and recomputed on the host.
//
// Mapping (round 2): one request per QUAD of lanes, 16 requests per wave, one wave per workgroup.  The query positions of a
// request are dealt to its four lanes in contiguous quarters, and the quad works as a skewed pipeline: lane j computes its
// quarter of target row i at step i + j.  What a quarter needs from its left neighbour is what the neighbour held after the
// last cell of the same row one step earlier — H(i-1) of that cell (the diagonal), both running F values and the running
// row maximum — four registers handed over with DPP quad_perm, no LDS.  Lane 3 sees the finished row: row maximum into the
// b[] scratch, best score / te / qe, the stop conditions (broadcast back through the quad).  Compared with the lane-per-
// request mapping of round 1 (445 waves of 40 KB LDS for a launch of 28 000 requests: fewer waves than SIMDs, one wave per
// SIMD, launch time = one lane's serial latency) a launch is four times as many waves of a quarter of the serial length
// and 10 KB of LDS each, so every SIMD has work and several waves to hide LDS latency behind.
// The row state of a position — H(i-1,k), E(i,k), the query base of position k and a segment-end flag packed in one dword
// (13 + 13 + 3 bits, bit 31) — lives in LDS as cell[kk][lane] (kk = position inside the lane's quarter), so the 64 lanes of a
// wave touch 64 consecutive dwords (no bank conflicts) and the whole DP runs out of LDS and registers.  HBM traffic is the
// target window (2 bits per row) and one u16 per row for the b[] pass.  The work is VALU-bound: ~30 integer ops per cell.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "device.h"

namespace mbw {

#define HIP_OK(call)                                                                                             \
	do {                                                                                                         \
		hipError_t e_ = (call);                                                                                  \
		if (e_ != hipSuccess) die("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);    \
	} while (0)

namespace {

#define MSW_PAD 7u

__device__ __forceinline__ int msw_base(const uint8_t *__restrict__ pac, int64_t l_pac, int64_t p)
{
	if (p >= l_pac) {
		int64_t f = (l_pac << 1) - 1 - p;
		return 3 - ((pac[f >> 2] >> ((~f & 3) << 1)) & 3);
	}
	return (pac[p >> 2] >> ((~p & 3) << 1)) & 3;
}

struct MswPassOut { int score, te, qe; };

#define MSW_QP(a, b, c, d) ((a) | (b) << 2 | (c) << 4 | (d) << 6)
template <int CTRL>
__device__ __forceinline__ int msw_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }

// query code of position k of a request: the mate as it is or reverse-complemented, N = 4, beyond the read = padding
__device__ __forceinline__ uint32_t msw_code(const uint8_t *__restrict__ ms, int qlen, int is_rev, int k)
{
	if (k >= qlen) return MSW_PAD;
	if (is_rev) { const uint32_t b = ms[qlen - 1 - k]; return b < 4 ? 3 - b : 4; }
	const uint32_t c = ms[k];
	return c > 4 ? 4 : c;
}

// One striped-SW pass for the 16 requests of the wave (a quad of lanes each).  `S` positions per lane already laid out in LDS
// (query codes, H = E = 0; position k = j * S + kk sits at cell[kk * 64 + lane]), `tn` target rows, row i reads
// doubled-coordinate position t0 + i * tdir.  rows != nullptr: lane 3 writes the row maxima to rows[i * row_stride] (the b[]
// list is rebuilt from them afterwards).  The result is valid in every lane of the quad.
__device__ __forceinline__ MswPassOut msw_pass(uint32_t *cell, const MswParams &P, const uint8_t *__restrict__ pac, bool live, int S, int tn,
                                               int64_t t0, int tdir, int endsc, int sat_limit, uint16_t *rows, size_t row_stride, int *sat_hit)
{
	const int lane = threadIdx.x, j = lane & 3;
	int gmax = 0, te = -1, qe = -1;
	bool run = live && tn > 0 && S > 0;
	const int oe_del = P.o_del + P.e_del, oe_ins = P.o_ins + P.e_ins, e_del = P.e_del, e_ins = P.e_ins;
	// uniform trip count of the cell loop: the longest quarter in the wave
	int Swave = run ? S : 0;
	for (int o = 32; o; o >>= 1) Swave = max(Swave, __shfl_xor(Swave, o));
	int co_diag = 0, co_fseg = 0, co_ffull = 0, co_key = 0;   // what this lane held after its last cell of the previous step
	int tb_next = (run && j == 0) ? msw_base(pac, P.l_pac, t0) : 0;
	for (int t = 0; __any(run); ++t) {
		const int i = t - j;                                   // the row this lane works on in this step
		const bool act = run && i >= 0 && i < tn;
		const int tb = tb_next;
		if (run && i + 1 >= 0 && i + 1 < tn) tb_next = msw_base(pac, P.l_pac, t0 + (int64_t)(i + 1) * tdir);   // in flight during this step
		// scores of this row's target base against query codes 0..3 (one byte each) and 4
		const uint32_t slo = tb == 0 ? P.slo[0] : tb == 1 ? P.slo[1] : tb == 2 ? P.slo[2] : P.slo[3];
		// ... and against code 4 in byte 0 of the high half; bytes 5..7 (codes 5, 6 and the padding code 7) score 0
		const uint32_t shi = (uint32_t)(uint8_t)(tb == 0 ? P.s4[0] : tb == 1 ? P.s4[1] : tb == 2 ? P.s4[2] : P.s4[3]);
		// hand-over from the left neighbour (its state after the same row, one step ago); lane 0 starts the row
		const int ci_diag = msw_dpp<MSW_QP(0, 0, 1, 2)>(co_diag), ci_fseg = msw_dpp<MSW_QP(0, 0, 1, 2)>(co_fseg);
		const int ci_ffull = msw_dpp<MSW_QP(0, 0, 1, 2)>(co_ffull), ci_key = msw_dpp<MSW_QP(0, 0, 1, 2)>(co_key);
		int diag = j ? ci_diag : 0, fseg = j ? ci_fseg : 0, ffull = j ? ci_ffull : 0;
		uint32_t key = j ? (uint32_t)ci_key : 0u;   // (row maximum << 16) | (0xffff - first position reaching it)
		const int kbase = j * S;
		const int kmax = act ? S : 0;
		auto step = [&](uint32_t &w, int kk) {
			const int hk = w & 0x1fff;
			int e = (int)__builtin_amdgcn_ubfe(w, 13, 13);
			const uint32_t q = __builtin_amdgcn_ubfe(w, 26, 3);
			// byte q of {shi, slo}: one v_perm_b32 instead of a data-dependent branch per cell
			const int s = (int)(int8_t)__builtin_amdgcn_perm(shi, slo, q | 0x0c0c0c00u);
			const int h = max(max(diag + s, e), fseg);               // Hpre(i,k)
			key = max(key, (uint32_t)h << 16 | (uint32_t)(0xffff - (kbase + kk)));
			const int hfin = max(h, ffull);                          // H(i,k) after the lazy-F pass
			e = max(max(e - e_del, h - oe_del), 0);
			const int t2 = h - oe_ins;
			fseg = max(max(fseg - e_ins, t2), 0);
			ffull = max(max(ffull - e_ins, t2), 0);
			fseg &= ~((int)w >> 31);                                 // bit 31: last position of a segment, F restarts at 0
			diag = hk;
			w = (w & 0xfc000000u) | ((uint32_t)e << 13 | (uint32_t)hfin);
		};
		for (int k = 0; k < Swave; k += 2) {   // quarters are even (the padded query length is a multiple of 8)
			if (k < kmax) {
				uint32_t w0 = cell[k * 64 + lane], w1 = cell[(k + 1) * 64 + lane];
				step(w0, k); step(w1, k + 1);
				cell[k * 64 + lane] = w0; cell[(k + 1) * 64 + lane] = w1;
			}
		}
		if (act) { co_diag = diag; co_fseg = fseg; co_ffull = ffull; co_key = (int)key; }
		// lane 3 has just finished row i: bookkeeping of the whole row, then the verdict goes back to the quad
		int stop = 0;
		if (act && j == 3) {
			const int imax = (int)(key >> 16);
			if (rows) rows[(size_t)i * row_stride] = (uint16_t)imax;
			if (imax > gmax) {
				gmax = imax; te = i; qe = 0xffff - (int)(key & 0xffff);
				if (gmax >= sat_limit) { *sat_hit = 1; stop = 1; }
				if (gmax >= endsc) stop = 1;
			}
			if (i + 1 >= tn) stop = 1;
		}
		stop = msw_dpp<MSW_QP(3, 3, 3, 3)>(stop);
		if (stop) run = false;
	}
	MswPassOut o;
	o.score = msw_dpp<MSW_QP(3, 3, 3, 3)>(gmax); o.te = msw_dpp<MSW_QP(3, 3, 3, 3)>(te); o.qe = msw_dpp<MSW_QP(3, 3, 3, 3)>(qe);
	*sat_hit = msw_dpp<MSW_QP(3, 3, 3, 3)>(*sat_hit);
	return o;
}

// What follows the forward pass of one request: score2 / te2 from the row maxima (the b[] list of src/ksw.c:196-226 rebuilt), the
// reverse pass that finds where the best local alignment starts (KSW_XSTART, src/ksw.c:344-352), the result record.  `cell` is
// re-initialised for the reverse pass (layout of msw_pass).  All four lanes of the quad call it with the same arguments.
__device__ __forceinline__ void msw_tail(uint32_t *cell, const MswParams &P, const uint8_t *__restrict__ pac, bool live, const MswReq &rq, int qlen,
                                         const uint8_t *__restrict__ ms, int r, int n_req, MswPassOut f, int sat, uint16_t *__restrict__ rows,
                                         MswRes *__restrict__ res)
{
	const int lane = threadIdx.x, j = lane & 3;
	const int tlen = (int)(rq.re - rq.rb);
	const bool byte_flavour = qlen * P.a < 250;
	const int PP = byte_flavour ? 16 : 8;
	const int minsc = P.min_seed_len * P.a;              // KSW_XSUBO | min_seed_len * a
	const int sat_limit = byte_flavour ? 255 - P.shift : 0x10000;
	MswRes out;
	out.score = f.score; out.te = f.te; out.qe = f.qe; out.score2 = -1; out.te2 = -1; out.tb = -1; out.qb = -1; out.flags = sat;
	// b[]: runs of rows whose maximum reaches minsc; second best = best run outside te +- ceil(score / max)
	// (the quad's own stores to rows[]: same wave, same addresses, program order)
	// (a window whose best row stays below minsc has no such row at all — nine rescue windows in ten of a repeat read's anchors: the
	// scan of its 600 row maxima, one dependent global load each, was as long as the forward pass itself)
	if (live && j == 0 && !sat && f.te >= 0 && f.score >= minsc) {
		const int d = (f.score + P.max_sc - 1) / P.max_sc;
		const int low = f.te - d, high = f.te + d;
		int last_sc = -1, last_i = -1;
		const int rows_done = f.te >= 0 ? tlen : 0;   // the forward pass never stops early (no KSW_XSTOP, no saturation)
		for (int i = 0; i < rows_done; ++i) {
			const int im = rows[(size_t)i * n_req + r];
			if (im < minsc) continue;
			if (last_i < 0 || last_i + 1 != i) {
				if (last_i >= 0 && (last_i < low || last_i > high) && last_sc > out.score2) { out.score2 = last_sc; out.te2 = last_i; }
				last_sc = im; last_i = i;
			} else if (last_sc < im) { last_sc = im; last_i = i; }
		}
		if (last_i >= 0 && (last_i < low || last_i > high) && last_sc > out.score2) { out.score2 = last_sc; out.te2 = last_i; }
	}
	// second pass over the reversed prefixes: where does the best local alignment start (KSW_XSTART)
	const bool second = live && !sat && f.score >= minsc && f.qe >= 0 && f.te >= 0;
	int qlen2 = second ? f.qe + 1 : 0;
	if (second && f.qe >= qlen) qlen2 = 0;   // cannot happen (the maximum of a row is reached on a real base first); stay in bounds
	const int slen2 = (qlen2 + PP - 1) / PP, npos2 = slen2 * PP, S2 = npos2 >> 2;
	if (qlen2 > 0)   // the codes of positions qe .. 0, padded, dealt to the quad again; H and E cleared
		for (int kk = 0; kk < S2; ++kk) {
			const int k = j * S2 + kk;
			const uint32_t c = k < qlen2 ? msw_code(ms, qlen, rq.is_rev, qlen2 - 1 - k) : MSW_PAD;
			cell[kk * 64 + lane] = c << 26 | ((k + 1) % slen2 == 0 ? 0x80000000u : 0u);
		}
	int sat2 = 0;
	MswPassOut g = msw_pass(cell, P, pac, qlen2 > 0, S2, f.te + 1, rq.rb + f.te, -1, f.score, sat_limit, nullptr, 0, &sat2);
	if (qlen2 > 0 && g.score == f.score) { out.tb = f.te - g.te; out.qb = f.qe - g.qe; }
	if (second && qlen2 == 0) out.flags = 1;
	if (sat2) out.flags = 1;
	if (live && j == 0) res[r] = out;
}

// list: the requests this launch works on (null: all of them, in order)
__global__ __launch_bounds__(64) void msw_kernel(MswParams P, int n_work, int n_req, const int *__restrict__ list, const MswReq *__restrict__ req,
                                                 const uint8_t *__restrict__ seq, const int64_t *__restrict__ off, const int *__restrict__ lens,
                                                 const uint8_t *__restrict__ pac, MswRes *__restrict__ res, uint16_t *__restrict__ rows)
{
	extern __shared__ uint32_t cell[];   // [positions of a quarter][64 lanes]
	const int lane = threadIdx.x, j = lane & 3;
	const int slot = blockIdx.x * 16 + (lane >> 2);
	const bool live = slot < n_work;
	const int r = live ? (list ? list[slot] : slot) : 0;
	MswReq rq;
	rq.rb = rq.re = 0; rq.read = 0; rq.is_rev = 0;
	if (live) rq = req[r];
	const int qlen = live ? lens[rq.read] : 0;
	const int tlen = (int)(rq.re - rq.rb);
	const bool byte_flavour = qlen * P.a < 250;          // KSW_XBYTE as mem_matesw sets it
	const int PP = byte_flavour ? 16 : 8;
	const int slen = (qlen + PP - 1) / PP;
	const int npos = slen * PP, S = npos >> 2;           // positions per lane (npos is a multiple of 8)
	const uint8_t *ms = seq + (live ? off[rq.read] : 0);
	// query codes -> LDS (reverse-complemented for the orientations that need it), H = E = 0
	if (live)
		for (int kk = 0; kk < S; ++kk) {
			const int k = j * S + kk;
			cell[kk * 64 + lane] = msw_code(ms, qlen, rq.is_rev, k) << 26 | ((k + 1) % slen == 0 ? 0x80000000u : 0u);
		}
	const int sat_limit = byte_flavour ? 255 - P.shift : 0x10000;
	int sat = 0;
	MswPassOut f = msw_pass(cell, P, pac, live, S, tlen, rq.rb, 1, 0x10000, sat_limit, live ? rows + r : nullptr, (size_t)n_req, &sat);
	msw_tail(cell, P, pac, live, rq, qlen, ms, r, n_req, f, sat, rows, res);
}

// ---------------------------------------------------------------------------------------------------------------------
// msw2_kernel (round 4): TWO requests of the same mate in the same orientation per quad — the rescue windows of a repeat read's
// anchors: 50 windows for one mate — in the byte flavour (qlen * a < 250: every H, E, F below 256).  The forward pass (the whole
// window; four fifths of the work) runs both alignments in the two 16-bit halves of every register: v_pk_add_i16 / v_pk_sub_i16 /
// v_pk_max_i16 on {H, E, Fseg, Ffull}, 13 vector instructions per cell instead of 25.  What makes that cheap is what the two share:
// the query (one set of codes and segment flags: a dword of eight 4-bit codes per eight positions, read once per eight cells) and
// the row loop; what differs per row — the two target bases — selects one of 16 rows of a 128-entry table in LDS that holds, per
// query code, the two substitution scores as a packed pair.  The row state of a position is one dword: H and E of both alignments
// as four bytes, unpacked and repacked with one v_perm_b32 each.  Row maxima, best cell, saturation are kept per alignment as in
// msw_pass (32-bit keys); the reverse pass and the b[] scan run per alignment through msw_tail, as for single requests.
// ---------------------------------------------------------------------------------------------------------------------
typedef short msw_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ msw_s2 s2_of(uint32_t v) { return __builtin_bit_cast(msw_s2, v); }
__device__ __forceinline__ uint32_t u_of(msw_s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ msw_s2 s2_max(msw_s2 a, msw_s2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ msw_s2 s2_splat(int v) { msw_s2 r; r.x = (short)v; r.y = (short)v; return r; }
typedef unsigned short msw_u2 __attribute__((ext_vector_type(2)));
// max(a - b, 0) per half for values that are never negative: v_pk_sub_u16 with the clamp bit, no separate maximum with 0
__device__ __forceinline__ msw_s2 s2_sub0(msw_s2 a, msw_s2 b)
{
	return __builtin_bit_cast(msw_s2, __builtin_elementwise_sub_sat(__builtin_bit_cast(msw_u2, a), __builtin_bit_cast(msw_u2, b)));
}

__global__ __launch_bounds__(64) void msw2_kernel(MswParams P, int n_pairs, int n_req, const int *__restrict__ pairs, const MswReq *__restrict__ req,
                                                  const uint8_t *__restrict__ seq, const int64_t *__restrict__ off, const int *__restrict__ lens,
                                                  const uint8_t *__restrict__ pac, MswRes *__restrict__ res, uint16_t *__restrict__ rows, int s_max)
{
	extern __shared__ uint32_t lds2[];
	uint32_t *cell = lds2;                                   // [s_max][64]: H_A, E_A, H_B, E_B as bytes (reverse passes: msw_pass's layout)
	uint32_t *codes = lds2 + (size_t)s_max * 64;             // [(s_max + 7) / 8][64]: eight 4-bit entries per dword: query code, bit 3 = end of a segment
	uint32_t *stab = codes + (size_t)((s_max + 7) >> 3) * 64;   // [16][8]: (target base A, target base B) x query code -> the two scores, packed
	const int lane = threadIdx.x, j = lane & 3;
	for (int e = lane; e < 128; e += 64) {
		const int ta = e >> 5, tb = (e >> 3) & 3, q = e & 7;
		const int sa = q < 4 ? (int)(int8_t)(P.slo[ta] >> (8 * q)) : q == 4 ? P.s4[ta] : 0;
		const int sb = q < 4 ? (int)(int8_t)(P.slo[tb] >> (8 * q)) : q == 4 ? P.s4[tb] : 0;
		stab[e] = (uint32_t)(uint16_t)(int16_t)sa | (uint32_t)(uint16_t)(int16_t)sb << 16;
	}
	const int slot = blockIdx.x * 16 + (lane >> 2);
	const bool live = slot < n_pairs;
	// (rB < 0: a request without a partner rides alone — its own launch of a few hundred waves would be as long as one wave's walk
	// through a window, behind this one)
	const int rA = live ? pairs[2 * slot] : 0, rB = live ? pairs[2 * slot + 1] : -1;
	const bool hasB = rB >= 0;
	MswReq qa, qb;
	qa.rb = qa.re = 0; qa.read = 0; qa.is_rev = 0; qb = qa;
	if (live) { qa = req[rA]; if (hasB) qb = req[rB]; else { qb = qa; qb.re = qb.rb; } }
	const int qlen = live ? lens[qa.read] : 0;
	const int tnA = (int)(qa.re - qa.rb), tnB = (int)(qb.re - qb.rb);
	const int slen = (qlen + 15) / 16, npos = slen * 16, S = npos >> 2;   // byte flavour: 16 segments
	const uint8_t *ms = seq + (live ? off[qa.read] : 0);
	if (live) {
		for (int kk = 0; kk < S; ++kk) cell[kk * 64 + lane] = 0;
		for (int c = 0; c * 8 < S; ++c) {
			uint32_t cd = 0;
			for (int u = 0; u < 8 && c * 8 + u < S; ++u) {
				const int k = j * S + c * 8 + u;
				cd |= (msw_code(ms, qlen, qa.is_rev, k) | ((k + 1) % slen == 0 ? 8u : 0u)) << (4 * u);
			}
			codes[c * 64 + lane] = cd;
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	const int sat_limit = 255 - P.shift;
	// the segment length when every live quad of the wave has the same one (else 0: the flags in the codes decide)
	int slen_u = 0;
	{
		const unsigned long long lv = __builtin_amdgcn_ballot_w64(live);
		if (lv) {
			const int first = __builtin_amdgcn_readlane(slen, __ffsll((long long)lv) - 1);
			slen_u = __builtin_amdgcn_ballot_w64(live && slen != first) ? 0 : first;
		}
	}
	// ---- the forward pass of both ----
	int gmaxA = 0, teA = -1, qeA = -1, gmaxB = 0, teB = -1, qeB = -1, satA = 0, satB = 0;
	// (two copies of the pass — segment ends known to the scalar unit, or read from the codes — so that neither has a branch per cell)
	auto forward = [&](auto uniform_segments) {
		constexpr bool UNI = decltype(uniform_segments)::value;
		const int tn = tnA > tnB ? tnA : tnB;
		bool run = live && tn > 0 && S > 0;
		const msw_s2 oe_del = s2_splat(P.o_del + P.e_del), oe_ins = s2_splat(P.o_ins + P.e_ins), e_del = s2_splat(P.e_del), e_ins = s2_splat(P.e_ins);
		const msw_s2 zero = s2_splat(0);
		int Swave = run ? S : 0;
		for (int o = 32; o; o >>= 1) Swave = max(Swave, __shfl_xor(Swave, o));
		uint32_t co_diag = 0, co_fseg = 0, co_ffull = 0, co_key = 0;
		auto base_at = [&](const MswReq &rq, int tnx, int i) -> int {   // (rows past the end of the shorter window: its last row again, results unused)
			const int ii = i < tnx ? i : tnx - 1;
			return ii >= 0 ? msw_base(pac, P.l_pac, rq.rb + ii) : 0;
		};
		int ta_next = (run && j == 0) ? base_at(qa, tnA, 0) : 0, tb_next = (run && j == 0) ? base_at(qb, tnB, 0) : 0;
		uint16_t *rowsA = rows + rA, *rowsB = rows + (hasB ? rB : rA);
		for (int t = 0; __any(run); ++t) {
			const int i = t - j;
			const bool act = run && i >= 0 && i < tn;
			const int ta = ta_next, tb = tb_next;
			if (run && i + 1 >= 0 && i + 1 < tn) { ta_next = base_at(qa, tnA, i + 1); tb_next = base_at(qb, tnB, i + 1); }
			const uint32_t *srow = stab + ((ta << 2 | tb) << 3);
			const uint32_t ci_diag = (uint32_t)msw_dpp<MSW_QP(0, 0, 1, 2)>((int)co_diag), ci_fseg = (uint32_t)msw_dpp<MSW_QP(0, 0, 1, 2)>((int)co_fseg);
			const uint32_t ci_ffull = (uint32_t)msw_dpp<MSW_QP(0, 0, 1, 2)>((int)co_ffull);
			const uint32_t ci_key = (uint32_t)msw_dpp<MSW_QP(0, 0, 1, 2)>((int)co_key);
			msw_s2 diag = s2_of(j ? ci_diag : 0u), fseg = s2_of(j ? ci_fseg : 0u), ffull = s2_of(j ? ci_ffull : 0u);
			// the row maximum of each alignment and the first position that reaches it as ONE 16-bit key per alignment: h * 256 + (255 -
			// position) — the byte flavour has h < 250 and at most 256 positions.  Inside a block of eight cells the key is taken relative
			// to the block (h * 256 + 7 - u: one packed multiply-add with constants and one packed maximum per cell), the block's offset is
			// added once per block.
			msw_u2 key = __builtin_bit_cast(msw_u2, j ? ci_key : 0u);
			msw_u2 c256 = {256, 256};
			asm volatile("" : "+v"(c256));   // (opaque: h * 256 + rel is to be one v_pk_mad_u16, not a shift and an or)
			msw_u2 poscv = {(unsigned short)(248 - j * S), (unsigned short)(248 - j * S)};   // 255 - 7 - (first position of the block)
			const int kmax = act ? S : 0;
			// (H, E and both F are never negative and h = Hpre is not either: "max(x - c, 0)" is one saturating subtraction.  seg_end: the
			// end of a segment as a wave-uniform fact — slen_u > 0: every live quad of the wave has that segment length, which is the rule
			// (the reads of a chunk are of one length) — or bit 3 of the position's code)
			auto step = [&](uint32_t &w, uint32_t cd, int u, bool seg_end, msw_u2 &mblk) {
				const msw_s2 hd = s2_of(__builtin_amdgcn_perm(0u, w, 0x0c020c00u));   // H_A | H_B << 16   (row i - 1)
				msw_s2 e = s2_of(__builtin_amdgcn_perm(0u, w, 0x0c030c01u));          // E_A | E_B << 16
				uint32_t q = __builtin_amdgcn_ubfe(cd, 4 * u, 3);
				asm("" : "+v"(q));   // (opaque: v_bfe_u32 + v_lshl_add_u32 for the table address; folded, it is shift + and + add)
				const msw_s2 s = s2_of(srow[q]);
				const msw_s2 h = s2_max(s2_max(diag + s, e), fseg);                    // Hpre(i, k) of both
				const msw_u2 rel = {(unsigned short)(7 - u), (unsigned short)(7 - u)};
				mblk = __builtin_elementwise_max(mblk, __builtin_bit_cast(msw_u2, h) * c256 + rel);
				const msw_s2 hfin = s2_max(h, ffull);
				e = s2_max(s2_sub0(e, e_del), s2_sub0(h, oe_del));
				const msw_s2 t2 = s2_sub0(h, oe_ins);
				fseg = s2_max(s2_sub0(fseg, e_ins), t2);
				if constexpr (UNI) fseg = seg_end ? zero : fseg;
				else fseg = s2_of(u_of(fseg) & ~(uint32_t)__builtin_amdgcn_sbfe((int)cd, 4 * u + 3, 1));   // -1 at the last position of a segment
				ffull = s2_max(s2_sub0(ffull, e_ins), t2);
				diag = hd;
				w = __builtin_amdgcn_perm(u_of(e), u_of(hfin), 0x06020400u);            // bytes: H_A, E_A, H_B, E_B
			};
			int seg_pos = 0;   // position inside the segment (a lane's quarter is four whole segments: the same in every lane)
			const msw_u2 eight = {8, 8};
			for (int c = 0; c * 8 < Swave; ++c, poscv -= eight) {
				const uint32_t cd = c * 8 < kmax ? codes[c * 64 + lane] : 0u;
				msw_u2 mblk = {0, 0};
#pragma unroll
				for (int u = 0; u < 8; u += 2) {
					const int k = c * 8 + u;
					bool end0 = false, end1 = false;
					if constexpr (UNI) {
						end0 = ++seg_pos == slen_u; if (end0) seg_pos = 0;
						end1 = ++seg_pos == slen_u; if (end1) seg_pos = 0;
					}
					if (k < kmax) {
						uint32_t w0 = cell[k * 64 + lane], w1 = cell[(k + 1) * 64 + lane];
						step(w0, cd, u, end0, mblk); step(w1, cd, u + 1, end1, mblk);
						cell[k * 64 + lane] = w0; cell[(k + 1) * 64 + lane] = w1;
					}
				}
				if (c * 8 < kmax) key = __builtin_elementwise_max(key, mblk + poscv);
			}
			const uint32_t keyAB = __builtin_bit_cast(uint32_t, key);
			if (act) { co_diag = u_of(diag); co_fseg = u_of(fseg); co_ffull = u_of(ffull); co_key = keyAB; }
			int stop = 0;
			if (act && j == 3) {
				if (i < tnA && !satA) {
					const int imax = (int)(keyAB >> 8 & 0xff);
					rowsA[(size_t)i * n_req] = (uint16_t)imax;
					if (imax > gmaxA) { gmaxA = imax; teA = i; qeA = 255 - (int)(keyAB & 0xff); if (gmaxA >= sat_limit) satA = 1; }
				}
				if (i < tnB && !satB) {
					const int imax = (int)(keyAB >> 24);
					rowsB[(size_t)i * n_req] = (uint16_t)imax;
					if (imax > gmaxB) { gmaxB = imax; teB = i; qeB = 255 - (int)(keyAB >> 16 & 0xff); if (gmaxB >= sat_limit) satB = 1; }
				}
				if ((i + 1 >= tnA || satA) && (i + 1 >= tnB || satB)) stop = 1;
			}
			stop = msw_dpp<MSW_QP(3, 3, 3, 3)>(stop);
			if (stop) run = false;
		}
	};
	if (slen_u > 0) forward(std::true_type{}); else forward(std::false_type{});
	MswPassOut fA, fB;
	fA.score = msw_dpp<MSW_QP(3, 3, 3, 3)>(gmaxA); fA.te = msw_dpp<MSW_QP(3, 3, 3, 3)>(teA); fA.qe = msw_dpp<MSW_QP(3, 3, 3, 3)>(qeA);
	fB.score = msw_dpp<MSW_QP(3, 3, 3, 3)>(gmaxB); fB.te = msw_dpp<MSW_QP(3, 3, 3, 3)>(teB); fB.qe = msw_dpp<MSW_QP(3, 3, 3, 3)>(qeB);
	satA = msw_dpp<MSW_QP(3, 3, 3, 3)>(satA); satB = msw_dpp<MSW_QP(3, 3, 3, 3)>(satB);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	msw_tail(cell, P, pac, live, qa, qlen, ms, rA, n_req, fA, satA, rows, res);
	if (__any(live && hasB)) msw_tail(cell, P, pac, live && hasB, qb, qlen, ms, hasB ? rB : 0, n_req, fB, satB, rows, res);
}

} // namespace

size_t msw_lds_bytes(int max_len)
{
	return (size_t)((max_len + 15) / 16 * 16 / 4) * 64 * 4;   // a quarter of the padded query per lane (padding to 16 covers both lane widths)
}

size_t msw2_lds_bytes(int max_len)
{
	const int s_max = (max_len + 15) / 16 * 16 / 4;
	return ((size_t)s_max * 64 + (size_t)((s_max + 7) / 8) * 64 + 128) * 4;
}

// h_req: the requests as the host holds them (for the pairing), or null: every request on its own.  h_list / d_list: room for
// 2 n_req ints each (host side page-locked in the pipeline): the pairs first (2 ints each), then the single requests.
void launch_msw(void *stream, const MswParams &P, int n_req, const MswReq *d_req, const uint8_t *d_seq, const int64_t *d_off, const int *d_len,
                const uint8_t *d_pac, MswRes *d_res, uint16_t *d_rows, int max_len, const MswReq *h_req, const int *h_len, int *h_list, int *d_list)
{
	if (n_req <= 0) return;
	const size_t lds = msw_lds_bytes(max_len), lds2 = msw2_lds_bytes(max_len);
	if (lds > 160 * 1024) die("mate-rescue kernel: reads of %d bp do not fit the LDS row buffers", max_len);
	static size_t s_attr = 0, s_attr2 = 0;
	if (lds > s_attr) {
		HIP_OK(hipFuncSetAttribute((const void *)msw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		s_attr = lds;
	}
	static const bool pairing = !(getenv("MPIBWA_MSW_PAIRS") && atoi(getenv("MPIBWA_MSW_PAIRS")) == 0);
	int n_pairs = 0, n_single = n_req;
	const int *d_single = nullptr;
	if (pairing && h_req && h_len && h_list && d_list && lds2 <= 160 * 1024) {
		// two requests next to each other for the same mate in the same orientation (mem_sam_pe lists an end's anchors one after the
		// other: the windows of a repeat read's anchors are such runs), both in the byte flavour
		// work items of msw2_kernel from the front of the list (two ints each; a byte-flavour request without a partner: (k, -1)), the
		// requests of the word flavour (reads of 250 bp and more) for msw_kernel from its back
		int *pl = h_list, np = 0, ns = 0, n_alone = 0;
		int *sl_end = h_list + 2 * (size_t)n_req;
		for (int k = 0; k < n_req;) {
			const bool byte_k = h_len[h_req[k].read] * P.a < 250;
			if (!byte_k) { *--sl_end = k; ++ns; ++k; continue; }
			if (k + 1 < n_req && h_req[k].read == h_req[k + 1].read && h_req[k].is_rev == h_req[k + 1].is_rev && h_req[k].re > h_req[k].rb &&
			    h_req[k + 1].re > h_req[k + 1].rb) {
				pl[2 * np] = k; pl[2 * np + 1] = k + 1; ++np; k += 2;
			} else { pl[2 * np] = k; pl[2 * np + 1] = -1; ++np; ++n_alone; ++k; }
		}
		n_pairs = np; n_single = ns;
		static const bool say = getenv("MPIBWA_CPUSEC") != nullptr;
		if (say) fprintf(stderr, "[msw] %d requests: %d quads of two windows for one mate, %d of one, %d in the word flavour\n", n_req, np - n_alone, n_alone, ns);
		HIP_OK(hipMemcpyAsync(d_list, h_list, (size_t)2 * n_req * sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream));
		d_single = d_list + (sl_end - h_list);
		if (n_pairs) {
			if (lds2 > s_attr2) {
				HIP_OK(hipFuncSetAttribute((const void *)msw2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
				s_attr2 = lds2;
			}
			const int s_max = (max_len + 15) / 16 * 16 / 4;
			hipLaunchKernelGGL(msw2_kernel, dim3((n_pairs + 15) / 16), dim3(64), lds2, (hipStream_t)stream, P, n_pairs, n_req, (const int *)d_list, d_req, d_seq, d_off,
			                   d_len, d_pac, d_res, d_rows, s_max);
		}
	}
	if (n_single)
		hipLaunchKernelGGL(msw_kernel, dim3((n_single + 15) / 16), dim3(64), lds, (hipStream_t)stream, P, n_single, n_req, d_single, d_req, d_seq, d_off, d_len, d_pac,
		                   d_res, d_rows);
	HIP_OK(hipGetLastError());
}

} // namespace mbw
