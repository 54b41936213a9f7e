// ext_kernels.hip — batch of independent banded extensions (stage-level entry
// for parity tests and GCUPS measurement).  One wavefront per job; see
// wave_ext.cuh for the algorithm.  Reference: ksw_extend2, src/ksw.c:380-479.
#include <hip/hip_runtime.h>
#include "device.h"
#include "wave_ext.cuh"

namespace mbw {

#define EXT_WAVES 4

__global__ void __launch_bounds__(64 * EXT_WAVES)
extend_kernel(WxParams P, int n, const uint8_t *__restrict__ q, const int64_t *__restrict__ qoff,
              const uint8_t *__restrict__ t, const int64_t *__restrict__ toff, const int *__restrict__ w,
              const int *__restrict__ h0, int *__restrict__ out6, unsigned long long *cells_total, int max_qlen)
{
	extern __shared__ int lds[];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int job = blockIdx.x * EXT_WAVES + wave;
	if (job >= n) return;
	const WxLds L = wx_lds(lds + (size_t)wave * wx_lds_ints(max_qlen), max_qlen);
	const uint8_t *qs = q + qoff[job];
	const uint8_t *ts = t + toff[job];
	const int qlen = (int)(qoff[job + 1] - qoff[job]), tlen = (int)(toff[job + 1] - toff[job]);
	unsigned long long cells = 0;
	for (int j = lane; j < qlen; j += 64) L.Qs[j] = qs[j];
	__builtin_amdgcn_wave_barrier();
	WxResult r = wave_extend(qlen, tlen, [&](int i) { return ts[i]; }, P, w[job], h0[job], L, cells);
	if (lane == 0) {
		int *o = out6 + (size_t)job * 6;
		o[0] = r.score; o[1] = r.qle; o[2] = r.tle; o[3] = r.gtle; o[4] = r.gscore; o[5] = r.max_off;
		atomicAdd(cells_total, cells);
	}
}

void launch_extend(void *stream, const ExtParams &ep, int n, const uint8_t *d_q, const int64_t *d_qoff,
                   const uint8_t *d_t, const int64_t *d_toff, const int *d_w, const int *d_h0, const int *d_eb,
                   int *d_out6, unsigned long long *d_cells, int max_qlen)
{
	(void)d_eb;
	WxParams P;
	for (int i = 0; i < 25; ++i) P.mat[i] = ep.mat[i];
	P.o_del = ep.o_del; P.e_del = ep.e_del; P.o_ins = ep.o_ins; P.e_ins = ep.e_ins; P.zdrop = ep.zdrop;
	size_t shmem = (size_t)EXT_WAVES * wx_lds_ints(max_qlen) * sizeof(int);
	int n_blocks = (n + EXT_WAVES - 1) / EXT_WAVES;
	hipLaunchKernelGGL(extend_kernel, dim3(n_blocks), dim3(64 * EXT_WAVES), shmem, (hipStream_t)stream, P, n, d_q, d_qoff,
	                   d_t, d_toff, d_w, d_h0, d_out6, d_cells, max_qlen);
}

} // namespace mbw
