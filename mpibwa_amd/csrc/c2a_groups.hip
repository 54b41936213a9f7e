// c2a_groups.hip — the chains of a read with MANY chains, split into independent groups so that c2a_kernel can give every
// group a wavefront of its own.
//
// mem_chain2aln (src/bwamem.c:632-786) walks the chains of a read one after the other because a seed is skipped when an
// EARLIER extension of the read already covers it (:671-706).  That test can only succeed for a region p and a seed s with
// s inside p on the reference (:676), and p never leaves the reference window of the chain it was extended in (:642-661: the
// rmax[] of its chain, DevChain::rmax0 / rmax1), while s lies inside the window of its own chain.  So chains whose windows do not
// intersect cannot see each other's regions: the connected components of "windows intersect" are independent, whatever their
// order.  A read of a high-copy repeat has hundreds of chains at hundreds of loci — hundreds of components of a chain or two —
// and one wavefront walking them serially was the length of the whole launch on SURVEY §8d's genome (80 ms per sub-batch).
//
// Device-wide, over the chains of all such reads of a sub-batch at once (hipCUB sorts and scans, no per-read cap):
//   sort by (read, rmax0)  ->  running maximum of rmax1 inside a read  ->  a component starts where rmax0 >= that maximum
//   ->  sort by (component, chain number): a unit = the chains of one component in the read's chain order
//   ->  per unit: its read, its chains (clist), where its regions go (a slice of the read's own region slots)
// c2a_kernel writes, per chain, where its regions are and how many; reg_pack_kernel puts them back into the read's chain order,
// which is the order the serial walk would have produced them in (the sorts that follow are unstable: the order is visible).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include "device.h"

namespace mbw {

#define HIP_OK(call)                                                                                             \
	do {                                                                                                         \
		hipError_t e_ = (call);                                                                                  \
		if (e_ != hipSuccess) die("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);    \
	} while (0)

namespace {

typedef unsigned long long u64;
typedef long long i64;
#define POS_BITS 36
#define POS_MASK ((1ull << POS_BITS) - 1)

// element e = the t-th chain of heavy read h (hoff[h] <= e < hoff[h + 1]): key = (h, rmax0), value = the chain's number.
// A wavefront per read, its lanes over the read's chains (a thread per element had to find its read by binary search over the
// offsets — 16 dependent loads — and took 6.7 ms per sub-batch for 2.5 M elements).
__global__ void __launch_bounds__(64) grp_keys_kernel(int n_heavy, const int *__restrict__ hoff, const int *__restrict__ heavy, const int *__restrict__ chain_beg,
                                                      const DevChain *__restrict__ chains, u64 *__restrict__ key, int *__restrict__ val)
{
	const int h = blockIdx.x;
	if (h >= n_heavy) return;
	const int e0 = hoff[h], n = hoff[h + 1] - e0, c0 = chain_beg[heavy[h]];
	for (int t = threadIdx.x; t < n; t += 64) {
		key[e0 + t] = (u64)h << POS_BITS | (u64)chains[c0 + t].rmax0;
		val[e0 + t] = c0 + t;
	}
}
// in (read, rmax0) order: (read, rmax1) for the running maximum
__global__ void grp_ends_kernel(int n_el, const u64 *__restrict__ key, const int *__restrict__ val, const DevChain *__restrict__ chains, u64 *__restrict__ ends)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_el) return;
	ends[i] = (key[i] & ~POS_MASK) | (u64)chains[val[i]].rmax1;
}
// a component starts at i when the read changes or no earlier window of the read reaches rmax0[i]
__global__ void grp_flag_kernel(int n_el, const u64 *__restrict__ key, const u64 *__restrict__ run_max, int *__restrict__ flag)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_el) return;
	int f = 1;
	if (i > 0 && (key[i] >> POS_BITS) == (run_max[i - 1] >> POS_BITS)) f = (key[i] & POS_MASK) >= (run_max[i - 1] & POS_MASK) ? 1 : 0;
	flag[i] = f;
}
// key for the second sort: (component, chain number); value: the chain's seeds (its region slots)
__global__ void grp_keys2_kernel(int n_el, const int *__restrict__ comp_incl, const int *__restrict__ val, const DevChain *__restrict__ chains,
                                 u64 *__restrict__ key2, int *__restrict__ nsd)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_el) return;
	key2[i] = (u64)(unsigned)(comp_incl[i] - 1) << 32 | (u64)(unsigned)val[i];
	nsd[i] = chains[val[i]].n_seeds;
}
// in (component, chain number) order: the chain list, and one unit per component
__global__ void grp_units_kernel(int n_el, int n_heavy, const u64 *__restrict__ key2, const int *__restrict__ slot_excl, const int *__restrict__ hoff,
                                 const int *__restrict__ heavy, const int *__restrict__ reg_beg, int *__restrict__ clist, int *__restrict__ ustart,
                                 int *__restrict__ unit_rd, int *__restrict__ unit_av, unsigned int *__restrict__ n_units)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_el) return;
	clist[i] = (int)(unsigned)key2[i];
	const unsigned comp = (unsigned)(key2[i] >> 32);
	if (i == 0 || (unsigned)(key2[i - 1] >> 32) != comp) {
		// components are numbered in (read, rmax0) order from 0: the component number IS the unit number
		int lo = 0, hi = n_heavy - 1;
		while (lo < hi) {
			const int mid = (lo + hi + 1) >> 1;
			if (hoff[mid] <= i) lo = mid; else hi = mid - 1;
		}
		const int rd = heavy[lo];
		ustart[comp] = i;
		unit_rd[comp] = rd;
		unit_av[comp] = reg_beg[rd] + (slot_excl[i] - slot_excl[hoff[lo]]);   // the slots of the read's chains before it, in this order
	}
	if (i == n_el - 1) { ustart[comp + 1] = n_el; *n_units = comp + 1; }
}

} // namespace

static size_t cub_tmp_bytes(int n_el)
{
	size_t a = 0, b = 0, c = 0, d = 0;
	HIP_OK(hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const u64 *)nullptr, (u64 *)nullptr, (const int *)nullptr, (int *)nullptr, n_el));
	HIP_OK(hipcub::DeviceScan::InclusiveScan(nullptr, b, (const u64 *)nullptr, (u64 *)nullptr, hipcub::Max(), n_el));
	HIP_OK(hipcub::DeviceScan::InclusiveSum(nullptr, c, (const int *)nullptr, (int *)nullptr, n_el));
	HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, d, (const int *)nullptr, (int *)nullptr, n_el));
	size_t m = a > b ? a : b;
	m = m > c ? m : c;
	m = m > d ? m : d;
	return (m + 255) & ~(size_t)255;
}
// scratch: 2 x u64[n] keys, u64[n] ends / run_max (in place), 4 x int[n] (+ the primitives' own)
size_t c2a_groups_scratch_bytes(int n_el)
{
	const size_t n = (size_t)n_el + 64;
	return cub_tmp_bytes(n_el) + 3 * n * 8 + 4 * n * 4 + 1024;
}

// d_hoff: n_heavy + 1 prefix offsets of the heavy reads' chain counts (d_hoff[n_heavy] = n_el); outputs: d_clist[n_el],
// d_ustart[n_el + 1], d_unit_rd[n_el], d_unit_av[n_el], *d_n_units
void launch_c2a_groups(void *stream, int n_el, int n_heavy, const int *d_hoff, const int *d_heavy, const int *d_chain_beg, const int *d_reg_beg,
                       const DevChain *d_chains, void *d_scratch, int *d_clist, int *d_ustart, int *d_unit_rd, int *d_unit_av, unsigned int *d_n_units)
{
	if (n_el <= 0) return;
	hipStream_t st = (hipStream_t)stream;
	const size_t n = (size_t)n_el + 64;
	size_t tmp_bytes = cub_tmp_bytes(n_el);
	uint8_t *p = (uint8_t *)d_scratch;
	void *tmp = p; p += tmp_bytes;
	u64 *key_a = (u64 *)p; p += n * 8;
	u64 *key_b = (u64 *)p; p += n * 8;
	u64 *ends = (u64 *)p; p += n * 8;
	int *val_a = (int *)p; p += n * 4;
	int *val_b = (int *)p; p += n * 4;
	int *ia = (int *)p; p += n * 4;
	int *ib = (int *)p; p += n * 4;
	const dim3 grid((n_el + 255) / 256), block(256);
	hipLaunchKernelGGL(grp_keys_kernel, dim3(n_heavy), dim3(64), 0, st, n_heavy, d_hoff, d_heavy, d_chain_beg, d_chains, key_a, val_a);
	int bits = POS_BITS;
	for (int h = n_heavy; h > 0; h >>= 1) ++bits;
	HIP_OK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key_a, key_b, val_a, val_b, n_el, 0, bits, st));
	hipLaunchKernelGGL(grp_ends_kernel, grid, block, 0, st, n_el, key_b, val_b, d_chains, ends);
	HIP_OK(hipcub::DeviceScan::InclusiveScan(tmp, tmp_bytes, ends, key_a /* run_max */, hipcub::Max(), n_el, st));
	hipLaunchKernelGGL(grp_flag_kernel, grid, block, 0, st, n_el, key_b, key_a, ia);
	HIP_OK(hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, ia, ib, n_el, st));
	hipLaunchKernelGGL(grp_keys2_kernel, grid, block, 0, st, n_el, ib, val_b, d_chains, key_a, val_a /* seeds per chain */);
	int bits2 = 32;
	for (int c = n_el; c > 0; c >>= 1) ++bits2;
	HIP_OK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key_a, key_b, val_a, val_b, n_el, 0, bits2 > 64 ? 64 : bits2, st));
	HIP_OK(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, val_b, ia, n_el, st));
	hipLaunchKernelGGL(grp_units_kernel, grid, block, 0, st, n_el, n_heavy, key_b, ia, d_hoff, d_heavy, d_reg_beg, d_clist, d_ustart, d_unit_rd, d_unit_av, d_n_units);
	HIP_OK(hipGetLastError());
}

} // namespace mbw
