// index_gpu.hip — bwa-compatible FM-index construction ON the MI355X.
//
// The reference cannot build an index at all (it relies on an external `bwa index`, docs/README.md:67) and the
// GPU box has neither GRCh38 nor network, so GRCh38-sized synthetic references are indexed here, in HBM, in
// seconds instead of the CPU's hour.  Output files are byte-compatible with src/bwt.c:385-462 (.bwt/.sa); the
// CPU builder in index.cpp (byte-identical to the reference's own hg19.small index) is the known answer this one
// is tested against.
//
// Method (sized for 288 GB of HBM: N = 2 x 3.1 G symbols):
//   text      forward + reverse complement, 2 bit/symbol in 64-bit words
//   round 0   all suffixes are radix-sorted (rocPRIM via hipCUB) on the key
//             [29 symbols | number of valid symbols], bucket by bucket on the leading symbols so that
//             the sort buffers stay at a few GB; rank[pos] = first row of the group of equal keys
//   doubling  only the suffixes that are still tied (repeats: a few % of the genome) go through
//             Larsson-Sadakane style rounds h = 29, 58, 116, ...: key = [dense group id | rank[pos+h]]
//   emit      a suffix whose row is final writes its BWT symbol T[pos-1] and, on rows = 0 mod 32, its SA sample;
//             the full suffix array is never materialised
//   pack      occ-interleaved 64-byte blocks with running counts (device scan)
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "internal.h"

namespace mbw {

typedef unsigned long long u64;
typedef unsigned int u32;

#define HIP_OK(call)                                                                                             \
	do {                                                                                                         \
		hipError_t e_ = (call);                                                                                  \
		if (e_ != hipSuccess) die("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);    \
	} while (0)

#define KMER 29

__device__ __forceinline__ int text_sym(const u64 *T, u64 p) { return (int)(T[p >> 5] >> (62 - 2 * (p & 31))) & 3; }

// forward + reverse complement into 64-bit words, first symbol in the top bits; words past the end are zero
__global__ void build_text_kernel(const uint8_t *__restrict__ pac, u64 l_pac, u64 *__restrict__ T, u64 n_words)
{
	u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= n_words) return;
	const u64 N = l_pac << 1;
	u64 v = 0;
	for (int k = 0; k < 32; ++k) {
		u64 p = (w << 5) + k;
		u64 c = 0;
		if (p < N) {
			u64 f = p < l_pac ? p : N - 1 - p;
			c = (pac[f >> 2] >> ((~f & 3) << 1)) & 3;
			if (p >= l_pac) c = 3 - c;
		}
		v = v << 2 | c;
	}
	T[w] = v;
}

__device__ __forceinline__ u64 kmer_key(const u64 *T, u64 p, u64 N)
{
	u64 w0 = T[p >> 5], w1 = T[(p >> 5) + 1];
	int sh = 2 * (int)(p & 31);
	u64 x = sh ? (w0 << sh) | (w1 >> (64 - sh)) : w0;
	u64 valid = N - p < KMER ? N - p : KMER;          // symbols past the end are zero in T
	return (x >> (64 - 2 * KMER)) << 6 | valid;
}

// positions whose first `kb` symbols equal `bucket`
__global__ void bucket_count_kernel(const u64 *__restrict__ T, u64 N, int kb, u64 *counts)
{
	__shared__ u32 loc[256];
	if (threadIdx.x < 256) loc[threadIdx.x] = 0;
	__syncthreads();
	for (u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x; p < N; p += (u64)gridDim.x * blockDim.x) {
		u64 key = kmer_key(T, p, N);
		u32 b = (u32)(key >> (64 - 2 * kb));
		atomicAdd(&loc[b], 1u);
	}
	__syncthreads();
	if (threadIdx.x < (1u << (2 * kb)) && loc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (u64)loc[threadIdx.x]);
}

// Appends (key, position) of every suffix of one 2-symbol bucket.  The order inside the bucket is irrelevant (it is sorted
// next), so the append is aggregated per workgroup: one atomic on the shared cursor per 1024 suffixes instead of one per
// wavefront — with 6.2 G suffixes and 16 passes the single-address atomic was the whole cost of this kernel (18 s -> see
// DESIGN.md §4.6).
#define FILL_BS 1024
__global__ void __launch_bounds__(FILL_BS)
bucket_fill_kernel(const u64 *__restrict__ T, u64 N, int kb, u32 bucket, u64 *__restrict__ keys, u64 *__restrict__ pos, u64 *cursor)
{
	__shared__ u32 wcnt[FILL_BS / 64];
	__shared__ u64 bbase;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	// grid-stride with a block-uniform trip count (HIP caps one launch at 2^32 work-items)
	for (u64 p0 = (u64)blockIdx.x * blockDim.x; p0 < N; p0 += (u64)gridDim.x * blockDim.x) {
		u64 p = p0 + threadIdx.x;
		bool mine = false;
		u64 key = 0;
		if (p < N) {
			key = kmer_key(T, p, N);
			mine = (u32)(key >> (64 - 2 * kb)) == bucket;
		}
		const u64 bal = __ballot(mine);
		if (lane == 0) wcnt[wave] = (u32)__popcll(bal);
		__syncthreads();
		if (threadIdx.x == 0) {
			u32 tot = 0;
			for (int w = 0; w < FILL_BS / 64; ++w) { u32 c = wcnt[w]; wcnt[w] = tot; tot += c; }   // exclusive prefix in place
			bbase = tot ? atomicAdd(cursor, (u64)tot) : 0;
		}
		__syncthreads();
		if (mine) {
			u64 o = bbase + wcnt[wave] + __popcll(bal & ((1ull << lane) - 1));
			keys[o] = key; pos[o] = p;
		}
		__syncthreads();
	}
}

__global__ void flag_kernel(const u64 *__restrict__ keys, u64 n, u32 *__restrict__ head)
{
	u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n) head[j] = (j == 0 || keys[j] != keys[j - 1]) ? (u32)j : 0u;
}

struct MaxOp { __device__ __forceinline__ u32 operator()(u32 a, u32 b) const { return a > b ? a : b; } };

struct EmitCtx {
	const u64 *T; u64 N;
	uint8_t *B;          // N+1 entries, indexed by full row (row 0 = the '$' suffix)
	u64 *sa_smp;         // (N+32)/32 entries
	u64 *primary;
};

__device__ __forceinline__ void emit_row(const EmitCtx &E, u64 row, u64 p)
{
	if (p == 0) *E.primary = row;
	else E.B[row] = (uint8_t)text_sym(E.T, p - 1);
	if ((row & 31) == 0) E.sa_smp[row >> 5] = p;
}

// after the round-0 sort of one bucket: ranks, final rows, and the list of tied suffixes
__global__ void round0_post_kernel(EmitCtx E, const u64 *__restrict__ pos, const u32 *__restrict__ gstart, u64 n, u64 row_base,
                                   u64 *__restrict__ rank, u64 *__restrict__ u_pos, u64 *__restrict__ u_row, u64 *u_count)
{
	u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	bool tied = false;
	u64 p = 0, grow = 0;
	if (j < n) {
		p = pos[j];
		u32 g = gstart[j];
		grow = row_base + g;
		rank[p] = grow;
		bool single = g == (u32)j && (j + 1 == n || gstart[j + 1] == (u32)(j + 1));
		if (single) emit_row(E, row_base + j, p);
		else tied = true;
	}
	u64 bal = __ballot(tied);
	if (bal) {
		int lane = threadIdx.x & 63, leader = __ffsll((long long)bal) - 1;
		u64 base = 0;
		if (lane == leader) base = atomicAdd(u_count, (u64)__popcll(bal));
		base = __shfl(base, leader);
		if (tied) {
			u64 o = base + __popcll(bal & ((1ull << lane) - 1));
			u_pos[o] = p; u_row[o] = grow;
		}
	}
}

// ---- doubling rounds on the tied suffixes ----
// items are kept sorted by row-group; gid = dense index of the group, g_row/g_idx = its first row / first item
__global__ void dense_gid_kernel(const u64 *__restrict__ u_row, u64 m, u32 *__restrict__ ghead)
{
	u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j < m) ghead[j] = (j == 0 || u_row[j] != u_row[j - 1]) ? 1u : 0u;
}
__global__ void group_table_kernel(const u64 *__restrict__ u_row, const u32 *__restrict__ gid_incl, u64 m, u64 *__restrict__ g_row,
                                   u64 *__restrict__ g_idx)
{
	u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j < m && (j == 0 || u_row[j] != u_row[j - 1])) { u32 g = gid_incl[j] - 1; g_row[g] = u_row[j]; g_idx[g] = j; }
}
__global__ void round_key_kernel(const u64 *__restrict__ u_pos, const u32 *__restrict__ gid_incl, u64 m, u64 h, u64 N,
                                 const u64 *__restrict__ rank, u64 *__restrict__ keys)
{
	u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= m) return;
	u64 q = u_pos[j] + h;
	u64 r2 = q < N ? rank[q] : 0;   // rows start at 1, so 0 sorts before every real suffix
	keys[j] = (u64)(gid_incl[j] - 1) << 34 | r2;
}
// ---- more than 2^30 tied suffixes (a genome that is half repeats): the group number no longer fits next to the 34-bit rank in
// one 64-bit key, so the round's sort is two stable passes over a permutation — by rank[pos + h], then by group — and the group
// numbers travel in an array of their own.
__global__ void wide_key_kernel(const u64 *__restrict__ u_pos, u64 m, u64 h, u64 N, const u64 *__restrict__ rank, u64 *__restrict__ r2, u32 *__restrict__ iota)
{
	u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= m) return;
	u64 q = u_pos[j] + h;
	r2[j] = q < N ? rank[q] : 0;
	iota[j] = (u32)j;
}
__global__ void wide_gather_gid_kernel(const u32 *__restrict__ perm, const u32 *__restrict__ gid_incl, u64 m, u32 *__restrict__ out)
{
	u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j < m) out[j] = gid_incl[perm[j]] - 1;
}
__global__ void wide_gather_kernel(const u32 *__restrict__ perm, const u64 *__restrict__ r2, const u64 *__restrict__ pos, u64 m, u64 *__restrict__ r2_out,
                                   u64 *__restrict__ pos_out)
{
	u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j < m) { const u32 s = perm[j]; r2_out[j] = r2[s]; pos_out[j] = pos[s]; }
}
__global__ void wide_flag_kernel(const u32 *__restrict__ gid, const u64 *__restrict__ r2, u64 n, u32 *__restrict__ head)
{
	u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n) head[j] = (j == 0 || gid[j] != gid[j - 1] || r2[j] != r2[j - 1]) ? (u32)j : 0u;
}

template <bool WIDE>
__global__ void round_post_kernel(EmitCtx E, const u64 *__restrict__ keys, const u32 *__restrict__ gid_sorted, const u64 *__restrict__ pos,
                                  const u32 *__restrict__ sstart, u64 m, const u64 *__restrict__ g_row, const u64 *__restrict__ g_idx,
                                  u64 *__restrict__ new_rank, u64 *__restrict__ o_pos, u64 *__restrict__ o_row, u64 *o_count)
{
	u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	bool tied = false;
	u64 p = 0, srow = 0;
	if (j < m) {
		p = pos[j];
		u32 g = WIDE ? gid_sorted[j] : (u32)(keys[j] >> 34);
		u32 s = sstart[j];
		srow = g_row[g] + (s - g_idx[g]);            // first row of the sub-group
		new_rank[j] = srow;
		bool single = s == (u32)j && (j + 1 == m || sstart[j + 1] == (u32)(j + 1));
		if (single) emit_row(E, g_row[g] + (j - g_idx[g]), p);
		else tied = true;
	}
	// stable compaction is required (items must stay sorted by row): block-level ordered append via ballot ranks
	// is not globally ordered, so tied items are written to their own index and compacted by a select afterwards.
	if (j < m) { o_pos[j] = tied ? p : ~0ull; o_row[j] = tied ? srow : ~0ull; }
	(void)o_count;
}
__global__ void scatter_rank_kernel(const u64 *__restrict__ pos, const u64 *__restrict__ new_rank, u64 m, u64 *__restrict__ rank)
{
	u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (j < m) rank[pos[j]] = new_rank[j];
}
struct NotDropped { __device__ __forceinline__ bool operator()(const u64 &v) const { return v != ~0ull; } };

// ---- final packing ----
__global__ void block_count_kernel(const uint8_t *__restrict__ B, u64 N, u64 primary, u64 n_blk, u64 *__restrict__ c0, u64 *__restrict__ c1,
                                   u64 *__restrict__ c2, u64 *__restrict__ c3)
{
	u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_blk) return;
	u32 c[4] = {0, 0, 0, 0};
	for (int k = 0; k < 128; ++k) {
		u64 i = (b << 7) + k;
		if (i >= N) break;
		u64 row = i + (i >= primary ? 1 : 0);    // skip the '$' row
		++c[B[row]];
	}
	c0[b] = c[0]; c1[b] = c[1]; c2[b] = c[2]; c3[b] = c[3];
}
__global__ void pack_kernel(const uint8_t *__restrict__ B, u64 N, u64 primary, u64 n_blk, const u64 *__restrict__ c0, const u64 *__restrict__ c1,
                            const u64 *__restrict__ c2, const u64 *__restrict__ c3, u32 *__restrict__ out)
{
	u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (b > n_blk) return;
	// n_blk = ceil(N/128); block n_blk is the closing count record (or the partial tail merges into it)
	u64 *cnt = (u64 *)(out + b * 16);
	if (b == n_blk) {   // running totals after the last symbol
		// handled on the host (needs the grand totals); leave zero here
		return;
	}
	cnt[0] = c0[b]; cnt[1] = c1[b]; cnt[2] = c2[b]; cnt[3] = c3[b];   // exclusive prefix sums
	for (int wi = 0; wi < 8; ++wi) {
		u32 v = 0;
		bool any = false;
		for (int k = 0; k < 16; ++k) {
			u64 i = (b << 7) + wi * 16 + k;
			if (i >= N) break;
			any = true;
			u64 row = i + (i >= primary ? 1 : 0);
			v |= (u32)B[row] << ((15 - k) << 1);
		}
		if (any) out[b * 16 + 8 + wi] = v;
	}
}

struct Buf {
	void *p = nullptr;
	explicit Buf(size_t bytes) { HIP_OK(hipMalloc(&p, bytes ? bytes : 8)); }
	~Buf() { (void)hipFree(p); }
	template <class T> T *as() { return (T *)p; }
};

static inline unsigned grid_for(u64 n, int bs)
{
	u64 g = (n + bs - 1) / bs;
	if (g * bs >= (1ull << 32)) die("index builder: launch of %llu work-items exceeds the 2^32 limit", (unsigned long long)(g * bs));
	return (unsigned)g;
}
static inline unsigned grid_strided(u64 n, int bs) { u64 g = (n + bs - 1) / bs; return (unsigned)(g < 262144 ? g : 262144); }
static inline void check_launch(const char *what)
{
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) die("index builder: %s: %s", what, hipGetErrorString(e));
}

} // namespace mbw

using namespace mbw;

// pac: forward strand, 2 bit/base (no N: ambiguous bases already replaced), l_pac bases.
// Writes <prefix>.bwt and <prefix>.sa; returns 0.
extern "C" int mi355x_index_build_gpu(int device, const uint8_t *pac, int64_t l_pac_, const char *prefix_, double *seconds)
{
	int nd = 0;
	if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) die("mi355x_index_build_gpu: no HIP device");
	HIP_OK(hipSetDevice(device % nd));
	hipEvent_t e0, e1;
	HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
	HIP_OK(hipEventRecord(e0, 0));
	const u64 l_pac = (u64)l_pac_, N = l_pac << 1;
	const std::string prefix(prefix_);
	const u64 n_words = (N + 31) / 32 + 2;
	const int BS = 256;

	Buf d_pac(l_pac / 4 + 1), d_T(n_words * 8);
	HIP_OK(hipMemcpy(d_pac.p, pac, l_pac / 4 + 1, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(build_text_kernel, dim3(grid_for(n_words, BS)), dim3(BS), 0, 0, d_pac.as<uint8_t>(), l_pac, d_T.as<u64>(), n_words);
	check_launch("build_text_kernel");
	const u64 *T = d_T.as<u64>();

	// bucket on the first kb symbols so that a bucket stays below ~384 M suffixes
	int kb = 1;
	while (kb < 4 && (N >> (2 * kb)) > (384ull << 20)) ++kb;
	const u32 n_buckets = 1u << (2 * kb);
	Buf d_counts(256 * 8);
	HIP_OK(hipMemset(d_counts.p, 0, 256 * 8));
	hipLaunchKernelGGL(bucket_count_kernel, dim3(grid_strided(N, BS)), dim3(BS), 0, 0, T, N, kb, d_counts.as<u64>());
	check_launch("bucket_count_kernel");
	std::vector<u64> counts(256);
	HIP_OK(hipMemcpy(counts.data(), d_counts.p, 256 * 8, hipMemcpyDeviceToHost));
	u64 max_bucket = 0;
	for (u32 b = 0; b < n_buckets; ++b) max_bucket = std::max(max_bucket, counts[b]);
	if (max_bucket >= (1ull << 31)) die("index builder: bucket of %llu suffixes is too large", max_bucket);

	Buf d_rank(N * 8), d_B(N + 1), d_smp(((N + 32) / 32) * 8), d_primary(8), d_cursor(8);
	HIP_OK(hipMemset(d_B.p, 0, N + 1));
	// tied suffixes: sized on demand (grown by re-allocation if a genome is unusually repetitive)
	u64 u_cap = std::max<u64>(N / 8, 1 << 20);
	Buf *u_pos = new Buf(u_cap * 8), *u_row = new Buf(u_cap * 8);
	Buf d_ucount(8);
	HIP_OK(hipMemset(d_ucount.p, 0, 8));
	EmitCtx E{T, N, d_B.as<uint8_t>(), d_smp.as<u64>(), d_primary.as<u64>()};

	{   // ---- round 0, bucket by bucket ----
		Buf k_in(max_bucket * 8), k_out(max_bucket * 8), v_in(max_bucket * 8), v_out(max_bucket * 8), head(max_bucket * 4), gst(max_bucket * 4);
		size_t tmp_sort = 0, tmp_scan = 0;
		hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, k_in.as<u64>(), k_out.as<u64>(), v_in.as<u64>(), v_out.as<u64>(), (int)max_bucket, 0,
		                                   64 - 2 * kb);
		hipcub::DeviceScan::InclusiveScan(nullptr, tmp_scan, head.as<u32>(), gst.as<u32>(), MaxOp(), (int)max_bucket);
		Buf tmp(std::max(tmp_sort, tmp_scan));
		u64 row_base = 1;   // row 0 is the '$' suffix
		for (u32 b = 0; b < n_buckets; ++b) {
			const u64 n = counts[b];
			if (n == 0) continue;
			HIP_OK(hipMemset(d_cursor.p, 0, 8));
			hipLaunchKernelGGL(bucket_fill_kernel, dim3(grid_strided(N, FILL_BS)), dim3(FILL_BS), 0, 0, T, N, kb, b, k_in.as<u64>(), v_in.as<u64>(), d_cursor.as<u64>());
			check_launch("bucket_fill_kernel");
			size_t ts = tmp_sort;
			HIP_OK(hipcub::DeviceRadixSort::SortPairs(tmp.p, ts, k_in.as<u64>(), k_out.as<u64>(), v_in.as<u64>(), v_out.as<u64>(), (int)n, 0, 64 - 2 * kb));
			hipLaunchKernelGGL(flag_kernel, dim3(grid_for(n, BS)), dim3(BS), 0, 0, k_out.as<u64>(), n, head.as<u32>());
			ts = tmp_scan;
			HIP_OK(hipcub::DeviceScan::InclusiveScan(tmp.p, ts, head.as<u32>(), gst.as<u32>(), MaxOp(), (int)n));
			// make sure the tied list can take this bucket in the worst case
			u64 have = 0;
			HIP_OK(hipMemcpy(&have, d_ucount.p, 8, hipMemcpyDeviceToHost));
			if (have + n > u_cap) {
				u64 ncap = std::max(u_cap + u_cap / 4, have + n);
				Buf *np = new Buf(ncap * 8), *nr = new Buf(ncap * 8);
				HIP_OK(hipMemcpy(np->p, u_pos->p, have * 8, hipMemcpyDeviceToDevice));
				HIP_OK(hipMemcpy(nr->p, u_row->p, have * 8, hipMemcpyDeviceToDevice));
				delete u_pos; delete u_row;
				u_pos = np; u_row = nr; u_cap = ncap;
			}
			hipLaunchKernelGGL(round0_post_kernel, dim3(grid_for(n, BS)), dim3(BS), 0, 0, E, v_out.as<u64>(), gst.as<u32>(), n, row_base,
			                   d_rank.as<u64>(), u_pos->as<u64>(), u_row->as<u64>(), d_ucount.as<u64>());
			check_launch("round0_post_kernel");
			row_base += n;
		}
		HIP_OK(hipDeviceSynchronize());
		if (row_base != N + 1) die("index builder: buckets cover %llu of %llu suffixes", (unsigned long long)(row_base - 1), (unsigned long long)N);
	}

	u64 m = 0;
	HIP_OK(hipMemcpy(&m, d_ucount.p, 8, hipMemcpyDeviceToHost));
	if (m >= (1ull << 31) - 1024) die("index builder: %llu tied suffixes exceed the 2^31 budget of the doubling rounds", m);
	// one composite key [group | rank] while the group number fits 30 bits, else two stable sort passes (MPIBWA_IDX_WIDE=1 forces them: tests)
	const bool wide = m >= (1ull << 30) || getenv("MPIBWA_IDX_WIDE") != nullptr;
	if (m) {
		// the round-0 list was appended in arbitrary order: bring it into row order once (groups are contiguous row ranges).
		// Memory (a genome that is half repeats has 1.5 G tied suffixes): 48 bytes per item for the ping-pong arrays and the keys,
		// 8 for the flag / scan arrays, 8 for the group table (a group has at least two items), 16 more for the wide rounds;
		// arrays that are never alive together share storage (new ranks in `keys`, sorted ranks in `keys2`).
		Buf a_row(m * 8), a_pos(m * 8), d_sel(8);
		size_t t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
		hipcub::DeviceRadixSort::SortPairs(nullptr, t1, a_row.as<u64>(), a_row.as<u64>(), a_pos.as<u64>(), a_pos.as<u64>(), (int)m, 0, 34);
		hipcub::DeviceRadixSort::SortPairs(nullptr, t2, a_row.as<u64>(), a_row.as<u64>(), a_pos.as<u64>(), a_pos.as<u64>(), (int)m, 0, 64);
		hipcub::DeviceScan::InclusiveSum(nullptr, t3, (u32 *)nullptr, (u32 *)nullptr, (int)m);
		hipcub::DeviceSelect::If(nullptr, t4, a_pos.as<u64>(), a_pos.as<u64>(), d_sel.as<u64>(), (int)m, NotDropped());
		hipcub::DeviceRadixSort::SortPairs(nullptr, t5, a_row.as<u64>(), a_row.as<u64>(), (u32 *)nullptr, (u32 *)nullptr, (int)m, 0, 34);
		hipcub::DeviceRadixSort::SortPairs(nullptr, t6, (u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, (int)m, 0, 32);
		Buf tmp(std::max(std::max(std::max(t1, t2), std::max(t3, t4)), std::max(t5, t6)) + 256);
		size_t ts = t1;
		HIP_OK(hipcub::DeviceRadixSort::SortPairs(tmp.p, ts, u_row->as<u64>(), a_row.as<u64>(), u_pos->as<u64>(), a_pos.as<u64>(), (int)m, 0, 34));
		HIP_OK(hipDeviceSynchronize());
		delete u_pos; delete u_row;
		u_pos = u_row = nullptr;
		Buf b_row(m * 8), b_pos(m * 8), keys(m * 8), keys2(m * 8), gflag(m * 4), gid(m * 4), g_row((m / 2 + 2) * 8), g_idx((m / 2 + 2) * 8);
		Buf &nrank = keys;        // written by round_post_kernel, after the round's last reader of `keys`
		Buf &sflag = gflag;       // the group flags are dead once the group numbers exist
		Buf &sst = gid;           // ... and the group numbers once the keys are built
		const u64 mw = wide ? m : 1;
		Buf w_p1(mw * 4), w_p2(mw * 4), w_g1(mw * 4), w_g2(mw * 4);
		Buf &w_r2 = keys2;        // the wide rounds read the sorted ranks of their first pass from nowhere: the slot takes the final order's
		u64 *cur_row = a_row.as<u64>(), *cur_pos = a_pos.as<u64>(), *alt_row = b_row.as<u64>(), *alt_pos = b_pos.as<u64>();
		for (u64 h = KMER; m > 0; h <<= 1) {
			hipLaunchKernelGGL(dense_gid_kernel, dim3(grid_for(m, BS)), dim3(BS), 0, 0, cur_row, m, gflag.as<u32>());
			ts = t3;
			HIP_OK(hipcub::DeviceScan::InclusiveSum(tmp.p, ts, gflag.as<u32>(), gid.as<u32>(), (int)m));
			hipLaunchKernelGGL(group_table_kernel, dim3(grid_for(m, BS)), dim3(BS), 0, 0, cur_row, gid.as<u32>(), m, g_row.as<u64>(), g_idx.as<u64>());
			if (!wide) {
				hipLaunchKernelGGL(round_key_kernel, dim3(grid_for(m, BS)), dim3(BS), 0, 0, cur_pos, gid.as<u32>(), m, h, N, d_rank.as<u64>(), keys.as<u64>());
				ts = t2;
				HIP_OK(hipcub::DeviceRadixSort::SortPairs(tmp.p, ts, keys.as<u64>(), keys2.as<u64>(), cur_pos, alt_pos, (int)m, 0, 64));
				hipLaunchKernelGGL(flag_kernel, dim3(grid_for(m, BS)), dim3(BS), 0, 0, keys2.as<u64>(), m, sflag.as<u32>());
			} else {
				// stable by rank[pos + h], then stable by group: the order of the composite key
				hipLaunchKernelGGL(wide_key_kernel, dim3(grid_for(m, BS)), dim3(BS), 0, 0, cur_pos, m, h, N, d_rank.as<u64>(), keys.as<u64>(), w_p1.as<u32>());
				ts = t5;
				HIP_OK(hipcub::DeviceRadixSort::SortPairs(tmp.p, ts, keys.as<u64>(), keys2.as<u64>(), w_p1.as<u32>(), w_p2.as<u32>(), (int)m, 0, 34));
				hipLaunchKernelGGL(wide_gather_gid_kernel, dim3(grid_for(m, BS)), dim3(BS), 0, 0, w_p2.as<u32>(), gid.as<u32>(), m, w_g1.as<u32>());
				ts = t6;
				HIP_OK(hipcub::DeviceRadixSort::SortPairs(tmp.p, ts, w_g1.as<u32>(), w_g2.as<u32>(), w_p2.as<u32>(), w_p1.as<u32>(), (int)m, 0, 32));
				hipLaunchKernelGGL(wide_gather_kernel, dim3(grid_for(m, BS)), dim3(BS), 0, 0, w_p1.as<u32>(), keys.as<u64>(), cur_pos, m, w_r2.as<u64>(), alt_pos);
				hipLaunchKernelGGL(wide_flag_kernel, dim3(grid_for(m, BS)), dim3(BS), 0, 0, w_g2.as<u32>(), w_r2.as<u64>(), m, sflag.as<u32>());
			}
			ts = t3;
			HIP_OK(hipcub::DeviceScan::InclusiveScan(tmp.p, ts, sflag.as<u32>(), sst.as<u32>(), MaxOp(), (int)m));
			// alt_pos = positions in the new order; cur_pos/cur_row are rewritten with the survivors (dropped = ~0)
			if (!wide)
				hipLaunchKernelGGL(round_post_kernel<false>, dim3(grid_for(m, BS)), dim3(BS), 0, 0, E, keys2.as<u64>(), (const u32 *)nullptr, alt_pos, sst.as<u32>(), m,
				                   g_row.as<u64>(), g_idx.as<u64>(), nrank.as<u64>(), cur_pos, cur_row, (u64 *)nullptr);
			else
				hipLaunchKernelGGL(round_post_kernel<true>, dim3(grid_for(m, BS)), dim3(BS), 0, 0, E, (const u64 *)nullptr, w_g2.as<u32>(), alt_pos, sst.as<u32>(), m,
				                   g_row.as<u64>(), g_idx.as<u64>(), nrank.as<u64>(), cur_pos, cur_row, (u64 *)nullptr);
			hipLaunchKernelGGL(scatter_rank_kernel, dim3(grid_for(m, BS)), dim3(BS), 0, 0, alt_pos, nrank.as<u64>(), m, d_rank.as<u64>());
			// stable compaction of the survivors (dropped entries carry ~0 in both arrays)
			ts = t4;
			HIP_OK(hipcub::DeviceSelect::If(tmp.p, ts, cur_pos, alt_pos, d_sel.as<u64>(), (int)m, NotDropped()));
			ts = t4;
			HIP_OK(hipcub::DeviceSelect::If(tmp.p, ts, cur_row, alt_row, d_sel.as<u64>(), (int)m, NotDropped()));
			u64 kept = 0;
			HIP_OK(hipMemcpy(&kept, d_sel.p, 8, hipMemcpyDeviceToHost));
			std::swap(cur_pos, alt_pos);
			std::swap(cur_row, alt_row);
			m = kept;
			check_launch("doubling round");
			if (h > N) die("index builder: doubling did not converge");
		}
	}
	delete u_pos; delete u_row;

	u64 primary = 0;
	HIP_OK(hipMemcpy(&primary, d_primary.p, 8, hipMemcpyDeviceToHost));
	{   // row 0 is the '$' suffix; its BWT symbol is the last text symbol = complement of the first forward base
		uint8_t c = 3 - ((pac[0] >> 6) & 3);
		HIP_OK(hipMemcpy(d_B.p, &c, 1, hipMemcpyHostToDevice));
		u64 n_ = N;
		HIP_OK(hipMemcpy(d_smp.p, &n_, 8, hipMemcpyHostToDevice));
	}

	// ---- occ-interleaved layout ----
	const u64 n_blk = (N + 127) / 128;
	bwtint_t L2[5] = {0, 0, 0, 0, 0};
	{
		Buf c0(n_blk * 8), c1(n_blk * 8), c2(n_blk * 8), c3(n_blk * 8), d_out((n_blk + 1) * 64);
		HIP_OK(hipMemset(d_out.p, 0, (n_blk + 1) * 64));
		hipLaunchKernelGGL(block_count_kernel, dim3(grid_for(n_blk, BS)), dim3(BS), 0, 0, d_B.as<uint8_t>(), N, primary, n_blk, c0.as<u64>(),
		                   c1.as<u64>(), c2.as<u64>(), c3.as<u64>());
		size_t ts = 0;
		hipcub::DeviceScan::ExclusiveSum(nullptr, ts, c0.as<u64>(), c0.as<u64>(), (int)n_blk);
		Buf tmp(ts + 256);
		u64 *cs[4] = {c0.as<u64>(), c1.as<u64>(), c2.as<u64>(), c3.as<u64>()};
		u64 tot[4];
		for (int c = 0; c < 4; ++c) {   // grand totals = last exclusive prefix + last block count
			u64 lastv = 0, lastp = 0;
			HIP_OK(hipMemcpy(&lastv, cs[c] + n_blk - 1, 8, hipMemcpyDeviceToHost));
			size_t t = ts;
			HIP_OK(hipcub::DeviceScan::ExclusiveSum(tmp.p, t, cs[c], cs[c], (int)n_blk));
			HIP_OK(hipMemcpy(&lastp, cs[c] + n_blk - 1, 8, hipMemcpyDeviceToHost));
			tot[c] = lastp + lastv;
		}
		check_launch("block_count_kernel");
		hipLaunchKernelGGL(pack_kernel, dim3(grid_for(n_blk + 1, BS)), dim3(BS), 0, 0, d_B.as<uint8_t>(), N, primary, n_blk, c0.as<u64>(), c1.as<u64>(),
		                   c2.as<u64>(), c3.as<u64>(), d_out.as<u32>());
		// device image: one 64-B record per 128 symbols; the file closes with a 32-B count record right after the last packed word
		std::vector<uint32_t> dev(n_blk * 16);
		HIP_OK(hipMemcpy(dev.data(), d_out.p, n_blk * 64, hipMemcpyDeviceToHost));
		const u64 tail_words = ((N + 15) >> 4) - (n_blk - 1) * 8;   // packed words in the last block (1..8)
		for (int c = 0; c < 4; ++c) L2[c + 1] = L2[c] + tot[c];
		FILE *fp = fopen((prefix + ".bwt").c_str(), "wb");
		if (!fp) die("cannot write %s.bwt", prefix.c_str());
		fwrite(&primary, 8, 1, fp);
		fwrite(L2 + 1, 8, 4, fp);
		fwrite(dev.data(), 4, (n_blk - 1) * 16 + 8 + tail_words, fp);
		fwrite(tot, 8, 4, fp);
		fclose(fp);
	}
	{
		const u64 n_sa = (N + 32) / 32;
		std::vector<u64> smp(n_sa);
		HIP_OK(hipMemcpy(smp.data(), d_smp.p, n_sa * 8, hipMemcpyDeviceToHost));
		FILE *fp = fopen((prefix + ".sa").c_str(), "wb");
		if (!fp) die("cannot write %s.sa", prefix.c_str());
		u64 intv = 32, sl = N;
		fwrite(&primary, 8, 1, fp);
		fwrite(L2 + 1, 8, 4, fp);
		fwrite(&intv, 8, 1, fp);
		fwrite(&sl, 8, 1, fp);
		fwrite(smp.data() + 1, 8, n_sa - 1, fp);
		fclose(fp);
	}
	HIP_OK(hipEventRecord(e1, 0));
	HIP_OK(hipEventSynchronize(e1));
	float ms = 0;
	HIP_OK(hipEventElapsedTime(&ms, e0, e1));
	if (seconds) *seconds = ms * 1e-3;
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	return 0;
}
