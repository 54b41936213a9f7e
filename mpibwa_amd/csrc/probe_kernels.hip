// probe_kernels.hip — calibration micro-benchmarks: random 64-byte block gathers from a large table, in the access shapes
// the FM-index kernels could use.  They give the measured random-64-B ceiling that roofline fractions are quoted next to
//  and calibrate the FETCH_SIZE counter for this access pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "internal.h"

namespace mbw {
typedef unsigned long long u64;
typedef unsigned int u32;

__device__ __forceinline__ u64 mix(u64 x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
	return x;
}

// shape 0: quad per block, 16 B per lane (one dwordx4)        -> one 64-B request per quad
// shape 1: quad per block, 2 x 8 B per lane (count + words)    -> the smem_kernel shape
// shape 2: lane per block, 4 x 16 B per lane                   -> lane-private blocks
// dep != 0: the next block index depends on the loaded data (pointer-chase, like the SMEM forward sweep)
template <int SHAPE, int DEP>
__global__ void __launch_bounds__(256) gather_probe_kernel(const uint4 *__restrict__ tab, u64 n_blk, int iters, u64 *sink)
{
	const int lane = threadIdx.x & 63, c = lane & 3;
	const u64 gtid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	u64 state = mix((SHAPE == 2 ? gtid : gtid >> 2) + 12345);
	u64 acc = 0;
	for (int it = 0; it < iters; ++it) {
		u64 b = state % n_blk;
		if (SHAPE == 0) {
			uint4 v = tab[b * 4 + c];
			acc += v.x + v.w;
		} else if (SHAPE == 1) {
			const char *base = (const char *)tab + b * 64;
			u64 cnt = *(const u64 *)(base + 8 * c);
			uint2 w = *(const uint2 *)(base + 32 + 8 * c);
			acc += cnt + w.x + w.y;
		} else {
			uint4 v0 = tab[b * 4 + 0], v1 = tab[b * 4 + 1], v2 = tab[b * 4 + 2], v3 = tab[b * 4 + 3];
			acc += v0.x + v1.y + v2.z + v3.w;
		}
		if (DEP) {
			u64 a = acc;
			if (SHAPE != 2) { a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); }   // the whole quad follows the same chain
			state = mix(state + (a & 1) + 1);
		} else state = mix(state + 1);
	}
	if (acc == 0x1234567) sink[0] = acc;
}

} // namespace mbw

using namespace mbw;

// Returns achieved GB/s (64 B x requests / time) for the given shape over a table of `bytes` bytes (allocated here).
extern "C" double mi355x_gather_probe(int shape, int dep, size_t bytes, int waves_per_cu, int iters, double *ms_out)
{
	int nd = 0;
	if (hipGetDeviceCount(&nd) != hipSuccess || nd == 0) die("mi355x_gather_probe: no HIP device");
	uint4 *tab; u64 *sink;
	if (hipMalloc(&tab, bytes) != hipSuccess) die("gather probe: cannot allocate %zu bytes", bytes);
	(void)hipMalloc(&sink, 8);
	(void)hipMemset(tab, 1, bytes);
	const u64 n_blk = bytes / 64;
	const int blocks = 256 * waves_per_cu / 4;
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	auto launch = [&](int its) {
#define GP(S, D) hipLaunchKernelGGL((gather_probe_kernel<S, D>), dim3(blocks), dim3(256), 0, 0, tab, n_blk, its, sink)
		if (shape == 0) { if (dep) GP(0, 1); else GP(0, 0); }
		else if (shape == 1) { if (dep) GP(1, 1); else GP(1, 0); }
		else { if (dep) GP(2, 1); else GP(2, 0); }
#undef GP
	};
	launch(8);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0, 0);
	launch(iters);
	(void)hipEventRecord(e1, 0);
	(void)hipEventSynchronize(e1);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, e0, e1);
	const double groups = (double)blocks * 256 / (shape == 2 ? 1 : 4);
	const double req = groups * iters;
	if (ms_out) *ms_out = ms;
	(void)hipFree(tab); (void)hipFree(sink);
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	return req * 64 / (ms * 1e-3) / 1e9;
}
