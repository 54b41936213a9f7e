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

__global__ __launch_bounds__(64) void msw_kernel(MswParams P, int n_req, const MswReq *__restrict__ req, const uint8_t *__restrict__ seq,
                                                 const int64_t *__restrict__ off, const int *__restrict__ lens, const uint8_t *__restrict__ pac,
                                                 MswRes *__restrict__ res, uint16_t *__restrict__ rows)
{
	extern __shared__ uint32_t cell[];   // [positions of a quarter][64 lanes]
	const int lane = threadIdx.x, j = lane & 3;
	const int r = blockIdx.x * 16 + (lane >> 2);
	const bool live = r < n_req;
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
	const int minsc = P.min_seed_len * P.a;              // KSW_XSUBO | min_seed_len * a
	const int sat_limit = byte_flavour ? 255 - P.shift : 0x10000;
	int sat = 0;
	MswPassOut f = msw_pass(cell, P, pac, live, S, tlen, rq.rb, 1, 0x10000, sat_limit, live ? rows + r : nullptr, (size_t)n_req, &sat);
	MswRes out;
	out.score = f.score; out.te = f.te; out.qe = f.qe; out.score2 = -1; out.te2 = -1; out.tb = -1; out.qb = -1; out.flags = sat;
	// b[]: runs of rows whose maximum reaches minsc; second best = best run outside te +- ceil(score / max)
	// (the quad's own stores to rows[] a few lines up: same wave, same addresses, program order)
	if (live && j == 0 && !sat && f.te >= 0) {
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

} // namespace

size_t msw_lds_bytes(int max_len)
{
	return (size_t)((max_len + 15) / 16 * 16 / 4) * 64 * 4;   // a quarter of the padded query per lane (padding to 16 covers both lane widths)
}

void launch_msw(void *stream, const MswParams &P, int n_req, const MswReq *d_req, const uint8_t *d_seq, const int64_t *d_off, const int *d_len,
                const uint8_t *d_pac, MswRes *d_res, uint16_t *d_rows, int max_len)
{
	if (n_req <= 0) return;
	const size_t lds = msw_lds_bytes(max_len);
	if (lds > 160 * 1024) die("mate-rescue kernel: reads of %d bp do not fit the LDS row buffers", max_len);
	static size_t s_attr = 0;
	if (lds > s_attr) {
		HIP_OK(hipFuncSetAttribute((const void *)msw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		s_attr = lds;
	}
	const int blocks = (n_req + 15) / 16;
	hipLaunchKernelGGL(msw_kernel, dim3(blocks), dim3(64), lds, (hipStream_t)stream, P, n_req, d_req, d_seq, d_off, d_len, d_pac, d_res, d_rows);
	HIP_OK(hipGetLastError());
}

} // namespace mbw
