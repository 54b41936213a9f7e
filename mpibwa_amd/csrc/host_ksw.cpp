// host_ksw.cpp — dynamic programming that still runs on the host:
//   ksw_global2  banded global alignment with traceback          src/ksw.c:504-606
//   ksw_align2   local alignment used by mate rescue             src/ksw.c:63-365
//
// The reference's local alignment is Farrar's striped SSE2 kernel with a
// bounded "lazy F" correction; its score (and the second-best score / end
// positions derived from per-row maxima) depend on that exact evaluation
// order and on 8-/16-bit saturation, so the striped layout is restated lane by
// lane here (plain arrays of P lanes; the compiler is free to vectorise).
#include "host.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <emmintrin.h>

namespace mbw {

#define NEG_INF (-0x40000000)

int ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del, int e_del,
                int o_ins, int e_ins, int w, std::vector<uint32_t> *cigar)
{
	const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	std::vector<int32_t> H(qlen + 1), E(qlen + 1);
	std::vector<uint8_t> z;
	if (cigar) { cigar->clear(); z.resize((size_t)n_col * tlen); }
	H[0] = 0; E[0] = NEG_INF;
	for (int j = 1; j <= qlen; ++j) {
		H[j] = j <= w ? -(o_ins + e_ins * j) : NEG_INF;
		E[j] = NEG_INF;
	}
	for (int i = 0; i < tlen; ++i) {
		const int8_t *srow = mat + target[i] * 5;
		const int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
		int32_t f = NEG_INF, left = beg == 0 ? -(o_del + e_del * (i + 1)) : NEG_INF;
		uint8_t *zi = cigar ? &z[(size_t)i * n_col] : nullptr;
		for (int j = beg; j < end; ++j) {
			int32_t m = H[j] + srow[query[j]], e = E[j], h, t;
			H[j] = left;
			uint8_t d = m >= e ? 0 : 1;
			h = m >= e ? m : e;
			if (h < f) { d = 2; h = f; }
			left = h;
			t = m - oe_del; e -= e_del;
			if (e > t) d |= 1 << 2; else e = t;
			E[j] = e;
			t = m - oe_ins; f -= e_ins;
			if (f > t) d |= 2 << 4; else f = t;
			if (zi) zi[j - beg] = d;
		}
		H[end] = left; E[end] = NEG_INF;
	}
	int score = H[qlen];
	if (cigar) {
		std::vector<uint32_t> &c = *cigar;
		auto push = [&](uint32_t op, uint32_t len) {
			if (!c.empty() && (c.back() & 0xf) == op) c.back() += len << 4;
			else c.push_back(len << 4 | op);
		};
		int which = 0, i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
		while (i >= 0 && k >= 0) {
			which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
			if (which == 0) { push(0, 1); --i; --k; }
			else if (which == 1) { push(2, 1); --i; }
			else { push(1, 1); --k; }
		}
		if (i >= 0) push(2, i + 1);
		if (k >= 0) push(1, k + 1);
		std::reverse(c.begin(), c.end());
	}
	return score;
}

// ---------------------------------------------------------------------------
// striped local alignment
// ---------------------------------------------------------------------------
namespace {

struct Prof {
	int qlen, slen, size;      // size: 1 = 8-bit lanes (16 per vector), 2 = 16-bit lanes (8 per vector)
	uint8_t shift, mdiff, max;
	std::vector<int16_t> qp;   // [5][slen][P] (bytes widened for the 8-bit flavour)
};

Prof make_profile(int size, int qlen, const uint8_t *query, const int8_t *mat)
{
	Prof q;
	q.size = size > 1 ? 2 : 1;
	const int P = 8 * (3 - q.size);
	q.slen = (qlen + P - 1) / P;
	q.qlen = qlen;
	int8_t mn = 127, mx = 0;   // the reference scans with these initial values (src/ksw.c:83-87)
	for (int a = 0; a < 25; ++a) {
		if (mat[a] < mn) mn = mat[a];
		if (mat[a] > mx) mx = mat[a];
	}
	q.max = (uint8_t)mx;
	q.shift = (uint8_t)(256 - (uint8_t)mn);
	q.mdiff = (uint8_t)(mx + q.shift);
	q.qp.assign((size_t)5 * q.slen * P, 0);
	const int nlen = q.slen * P;
	size_t t = 0;
	for (int a = 0; a < 5; ++a) {
		const int8_t *ma = mat + a * 5;
		for (int i = 0; i < q.slen; ++i)
			for (int k = i; k < nlen; k += q.slen) {
				int v = k >= qlen ? 0 : ma[query[k]];
				q.qp[t++] = q.size == 1 ? (int16_t)(uint8_t)(v + q.shift) : (int16_t)v;
			}
	}
	return q;
}

inline int sat_u8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
inline int sat_i16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
inline int subs_u16(int a, int b) { int x = (a & 0xffff) - (b & 0xffff); return x < 0 ? 0 : x; }   // unsigned saturating, result as stored bits

// BYTE = true: 16 unsigned 8-bit lanes; false: 8 signed 16-bit lanes
template <bool BYTE>
KswResult striped_sw(const Prof &q, int tlen, const uint8_t *target, int o_del, int e_del_, int o_ins, int e_ins_, int xtra)
{
	const int P = BYTE ? 16 : 8;
	const int slen = q.slen;
	KswResult r = {0, -1, -1, -1, -1, -1, -1};
	const int minsc = (xtra & KSW_XSUBO) ? xtra & 0xffff : 0x10000;
	const int endsc = (xtra & KSW_XSTOP) ? xtra & 0xffff : 0x10000;
	const int oe_del = BYTE ? ((o_del + e_del_) & 0xff) : ((o_del + e_del_) & 0xffff);
	const int e_del = BYTE ? (e_del_ & 0xff) : (e_del_ & 0xffff);
	const int oe_ins = BYTE ? ((o_ins + e_ins_) & 0xff) : ((o_ins + e_ins_) & 0xffff);
	const int e_ins = BYTE ? (e_ins_ & 0xff) : (e_ins_ & 0xffff);
	const int shift = q.shift;
	std::vector<int> bufH0((size_t)slen * P, 0), bufH1((size_t)slen * P, 0), E((size_t)slen * P, 0), Hmax((size_t)slen * P, 0);
	int *H0 = bufH0.data(), *H1 = bufH1.data();
	std::vector<uint64_t> b;
	int te = -1, gmax = 0;
	auto SUB = [&](int a, int c) { return BYTE ? sat_u8(a - c) : subs_u16(a, c); };
	auto s16 = [](int v) { return (int)(int16_t)v; };   // view stored bits as a signed 16-bit lane
	for (int i = 0; i < tlen; ++i) {
		int h[16], e[16], f[16], t[16], mx[16];
		const int16_t *S = &q.qp[(size_t)target[i] * slen * P];
		for (int l = 0; l < P; ++l) { f[l] = 0; mx[l] = 0; }
		// h = last stripe of the previous row, moved up by one lane (lane 0 <- 0)
		if (slen > 0) {
			const int *last = H0 + (size_t)(slen - 1) * P;
			h[0] = 0;
			for (int l = 1; l < P; ++l) h[l] = last[l - 1];
		} else for (int l = 0; l < P; ++l) h[l] = 0;
		for (int j = 0; j < slen; ++j) {
			const int16_t *Sj = S + (size_t)j * P;
			int *Ej = &E[(size_t)j * P], *H1j = H1 + (size_t)j * P;
			for (int l = 0; l < P; ++l) {
				int hv;
				if (BYTE) { hv = sat_u8(h[l] + Sj[l]); hv = sat_u8(hv - shift); }
				else hv = sat_i16(s16(h[l]) + Sj[l]) & 0xffff;
				int ev = Ej[l];
				if (BYTE) { hv = std::max(hv, ev); hv = std::max(hv, f[l]); mx[l] = std::max(mx[l], hv); }
				else {
					hv = s16(hv) > s16(ev) ? hv : ev;
					hv = s16(hv) > s16(f[l]) ? hv : f[l];
					mx[l] = s16(mx[l]) > s16(hv) ? mx[l] : hv;
				}
				H1j[l] = hv;
				ev = SUB(ev, e_del);
				int tv = SUB(hv, oe_del);
				if (BYTE) ev = std::max(ev, tv); else ev = s16(ev) > s16(tv) ? ev : tv;
				Ej[l] = ev;
				int fv = SUB(f[l], e_ins);
				tv = SUB(hv, oe_ins);
				if (BYTE) fv = std::max(fv, tv); else fv = s16(fv) > s16(tv) ? fv : tv;
				f[l] = fv;
				h[l] = H0[(size_t)j * P + l];
			}
		}
		// lazy F: at most 16 lane shifts, leaving as soon as F cannot raise any H of a stripe
		bool done = false;
		for (int k = 0; k < 16 && !done; ++k) {
			for (int l = P - 1; l > 0; --l) f[l] = f[l - 1];
			f[0] = 0;
			for (int j = 0; j < slen && !done; ++j) {
				int *H1j = H1 + (size_t)j * P;
				bool any = false;
				for (int l = 0; l < P; ++l) {
					int hv = H1j[l];
					if (BYTE) hv = std::max(hv, f[l]); else hv = s16(hv) > s16(f[l]) ? hv : f[l];
					H1j[l] = hv;
					hv = SUB(hv, oe_ins);
					f[l] = SUB(f[l], e_ins);
					if (BYTE) { if (sat_u8(f[l] - hv) != 0) any = true; }
					else { if (s16(f[l]) > s16(hv)) any = true; }
				}
				if (!any) done = true;
			}
		}
		int imax = 0;
		if (BYTE) { for (int l = 0; l < P; ++l) imax = std::max(imax, mx[l]); }
		else {
			int m = s16(mx[0]);
			for (int l = 1; l < P; ++l) m = std::max(m, s16(mx[l]));
			imax = m & 0xffff;   // _mm_extract_epi16 zero-extends
		}
		if (imax >= minsc) {
			if (b.empty() || (int32_t)b.back() + 1 != i) b.push_back((uint64_t)imax << 32 | (uint32_t)i);
			else if ((int)(b.back() >> 32) < imax) b.back() = (uint64_t)imax << 32 | (uint32_t)i;
		}
		if (imax > gmax) {
			gmax = imax; te = i;
			memcpy(Hmax.data(), H1, sizeof(int) * (size_t)slen * P);
			if (BYTE) { if (gmax + shift >= 255 || gmax >= endsc) break; }
			else if (gmax >= endsc) break;
		}
		std::swap(H0, H1);
	}
	if (BYTE) r.score = gmax + shift < 255 ? gmax : 255;
	else r.score = gmax;
	r.te = te;
	if (!BYTE || r.score != 255) {
		int mxv = -1, qlen = slen * P;
		if (!BYTE) r.qe = -1;
		for (int i = 0; i < qlen; ++i) {
			int v = BYTE ? Hmax[i] : (Hmax[i] & 0xffff);
			int pos = i / P + i % P * slen;
			if (v > mxv) { mxv = v; r.qe = pos; }
			else if (v == mxv && pos < r.qe) r.qe = pos;
		}
		if (!b.empty()) {
			int i = (r.score + q.max - 1) / q.max;
			int low = te - i, high = te + i;
			for (uint64_t x : b) {
				int e = (int32_t)x;
				if ((e < low || e > high) && (int)(x >> 32) > r.score2) { r.score2 = (int)(x >> 32); r.te2 = e; }
			}
		}
	}
	return r;
}


// ---- the same striped kernel on 128-bit SSE2 registers (x86-64 hosts) ----
// One template for both lane widths; V16 = 16 unsigned bytes, V8 = 8 signed words.  Operation order follows the
// lane-by-lane version above exactly (that version is the readable statement of the semantics, this one is the fast one;
// tests/test_host_ksw.py checks that they agree).
struct LaneU8 {
	static const int P = 16;
	static __m128i set1(int v) { return _mm_set1_epi8((char)v); }
	static __m128i add_score(__m128i h, __m128i s, __m128i shift) { return _mm_subs_epu8(_mm_adds_epu8(h, s), shift); }
	static __m128i vmax(__m128i a, __m128i b) { return _mm_max_epu8(a, b); }
	static __m128i subs(__m128i a, __m128i b) { return _mm_subs_epu8(a, b); }
	static __m128i shl_lane(__m128i a) { return _mm_slli_si128(a, 1); }
	static bool f_dead(__m128i f, __m128i h) { return _mm_movemask_epi8(_mm_cmpeq_epi8(_mm_subs_epu8(f, h), _mm_setzero_si128())) == 0xffff; }
	static int hmax(__m128i x)
	{
		x = _mm_max_epu8(x, _mm_srli_si128(x, 8)); x = _mm_max_epu8(x, _mm_srli_si128(x, 4));
		x = _mm_max_epu8(x, _mm_srli_si128(x, 2)); x = _mm_max_epu8(x, _mm_srli_si128(x, 1));
		return _mm_extract_epi16(x, 0) & 0xff;
	}
};
struct LaneI16 {
	static const int P = 8;
	static __m128i set1(int v) { return _mm_set1_epi16((short)v); }
	static __m128i add_score(__m128i h, __m128i s, __m128i) { return _mm_adds_epi16(h, s); }
	static __m128i vmax(__m128i a, __m128i b) { return _mm_max_epi16(a, b); }
	static __m128i subs(__m128i a, __m128i b) { return _mm_subs_epu16(a, b); }
	static __m128i shl_lane(__m128i a) { return _mm_slli_si128(a, 2); }
	static bool f_dead(__m128i f, __m128i h) { return _mm_movemask_epi8(_mm_cmpgt_epi16(f, h)) == 0; }
	static int hmax(__m128i x)
	{
		x = _mm_max_epi16(x, _mm_srli_si128(x, 8)); x = _mm_max_epi16(x, _mm_srli_si128(x, 4));
		x = _mm_max_epi16(x, _mm_srli_si128(x, 2));
		return _mm_extract_epi16(x, 0);
	}
};

template <class L>
KswResult striped_sw_sse2(const Prof &q, int tlen, const uint8_t *target, int o_del, int e_del_, int o_ins, int e_ins_, int xtra)
{
	const bool BYTE = L::P == 16;
	const int P = L::P, slen = q.slen;
	KswResult r = {0, -1, -1, -1, -1, -1, -1};
	const int minsc = (xtra & KSW_XSUBO) ? xtra & 0xffff : 0x10000;
	const int endsc = (xtra & KSW_XSTOP) ? xtra & 0xffff : 0x10000;
	const __m128i zero = _mm_setzero_si128(), oe_del = L::set1(o_del + e_del_), e_del = L::set1(e_del_), oe_ins = L::set1(o_ins + e_ins_),
	              e_ins = L::set1(e_ins_), shift = L::set1(q.shift);
	// profile re-packed into vectors: [5][slen]
	std::vector<__m128i> mem((size_t)slen * 9 + 1);
	__m128i *qp = mem.data(), *H0 = qp + (size_t)slen * 5, *H1 = H0 + slen, *E = H1 + slen, *Hmax = E + slen;
	for (size_t v = 0; v < (size_t)slen * 5; ++v) {
		const int16_t *src = &q.qp[v * P];
		if (BYTE) { alignas(16) uint8_t t[16]; for (int l = 0; l < 16; ++l) t[l] = (uint8_t)src[l]; qp[v] = _mm_load_si128((const __m128i *)t); }
		else qp[v] = _mm_loadu_si128((const __m128i *)src);
	}
	for (int j = 0; j < slen; ++j) { H0[j] = zero; E[j] = zero; Hmax[j] = zero; H1[j] = zero; }
	std::vector<uint64_t> b;
	int te = -1, gmax = 0;
	for (int i = 0; i < tlen; ++i) {
		__m128i f = zero, mx = zero, h, e, t;
		const __m128i *S = qp + (size_t)target[i] * slen;
		h = slen > 0 ? L::shl_lane(H0[slen - 1]) : zero;
		for (int j = 0; j < slen; ++j) {
			h = L::add_score(h, S[j], shift);
			e = E[j];
			h = L::vmax(h, e); h = L::vmax(h, f);
			mx = L::vmax(mx, h);
			H1[j] = h;
			e = L::subs(e, e_del); t = L::subs(h, oe_del); e = L::vmax(e, t);
			E[j] = e;
			f = L::subs(f, e_ins); t = L::subs(h, oe_ins); f = L::vmax(f, t);
			h = H0[j];
		}
		bool done = false;
		for (int k = 0; k < 16 && !done; ++k) {
			f = L::shl_lane(f);
			for (int j = 0; j < slen; ++j) {
				h = L::vmax(H1[j], f);
				H1[j] = h;
				h = L::subs(h, oe_ins);
				f = L::subs(f, e_ins);
				if (L::f_dead(f, h)) { done = true; break; }
			}
		}
		int imax = L::hmax(mx);
		if (imax >= minsc) {
			if (b.empty() || (int32_t)b.back() + 1 != i) b.push_back((uint64_t)imax << 32 | (uint32_t)i);
			else if ((int)(b.back() >> 32) < imax) b.back() = (uint64_t)imax << 32 | (uint32_t)i;
		}
		if (imax > gmax) {
			gmax = imax; te = i;
			for (int j = 0; j < slen; ++j) Hmax[j] = H1[j];
			if (BYTE) { if (gmax + q.shift >= 255 || gmax >= endsc) break; }
			else if (gmax >= endsc) break;
		}
		std::swap(H0, H1);
	}
	if (BYTE) r.score = gmax + q.shift < 255 ? gmax : 255;
	else r.score = gmax;
	r.te = te;
	if (!BYTE || r.score != 255) {
		int mxv = -1, qlen = slen * P;
		if (!BYTE) r.qe = -1;
		const uint8_t *t8 = (const uint8_t *)Hmax;
		const uint16_t *t16 = (const uint16_t *)Hmax;
		for (int i = 0; i < qlen; ++i) {
			int v = BYTE ? t8[i] : t16[i];
			int pos = i / P + i % P * slen;
			if (v > mxv) { mxv = v; r.qe = pos; }
			else if (v == mxv && pos < r.qe) r.qe = pos;
		}
		if (!b.empty()) {
			int i = (r.score + q.max - 1) / q.max;
			int low = te - i, high = te + i;
			for (uint64_t x : b) {
				int e2 = (int32_t)x;
				if ((e2 < low || e2 > high) && (int)(x >> 32) > r.score2) { r.score2 = (int)(x >> 32); r.te2 = e2; }
			}
		}
	}
	return r;
}

} // namespace

static bool g_ksw_portable = getenv("MPIBWA_KSW_PORTABLE") != nullptr;

KswResult ksw_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, const int8_t *mat, int o_del, int e_del, int o_ins,
                     int e_ins, int xtra)
{
	const int size = (xtra & KSW_XBYTE) ? 1 : 2;
	Prof q = make_profile(size, qlen, query, mat);
	const bool portable = g_ksw_portable;   // lane-by-lane version, for cross-checking
	auto run = [&](const Prof &p, int x) {
		if (portable)
			return size == 1 ? striped_sw<true>(p, tlen, target, o_del, e_del, o_ins, e_ins, x)
			                 : striped_sw<false>(p, tlen, target, o_del, e_del, o_ins, e_ins, x);
		return size == 1 ? striped_sw_sse2<LaneU8>(p, tlen, target, o_del, e_del, o_ins, e_ins, x)
		                 : striped_sw_sse2<LaneI16>(p, tlen, target, o_del, e_del, o_ins, e_ins, x);
	};
	KswResult r = run(q, xtra);
	if ((xtra & KSW_XSTART) == 0 || ((xtra & KSW_XSUBO) && r.score < (xtra & 0xffff))) return r;
	if (r.qe < 0 || r.te < 0) return r;   // saturated 8-bit score: the reference's second pass cannot confirm it either
	// second pass on the reversed prefixes to find where the best local alignment starts
	std::reverse(query, query + r.qe + 1);
	std::reverse(target, target + r.te + 1);
	Prof q2 = make_profile(size, r.qe + 1, query, mat);
	KswResult rr = run(q2, KSW_XSTOP | r.score);
	std::reverse(query, query + r.qe + 1);
	std::reverse(target, target + r.te + 1);
	if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
	return r;
}

} // namespace mbw

// test hook (host logic, no GPU): the local alignment used by mate rescue; portable != 0 selects the lane-by-lane version
extern "C" void mi355x_host_ksw_align2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del,
                                       int e_del, int o_ins, int e_ins, int xtra, int portable, int out7[7])
{
	std::vector<uint8_t> q(query, query + qlen), t(target, target + tlen);
	bool saved = mbw::g_ksw_portable;
	mbw::g_ksw_portable = portable != 0;
	mbw::KswResult r = mbw::ksw_align2(qlen, q.data(), tlen, t.data(), mat, o_del, e_del, o_ins, e_ins, xtra);
	mbw::g_ksw_portable = saved;
	out7[0] = r.score; out7[1] = r.te; out7[2] = r.qe; out7[3] = r.score2; out7[4] = r.te2; out7[5] = r.tb; out7[6] = r.qb;
}
