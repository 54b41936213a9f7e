// init.hip — mi355x_init(): device selection + index residency for one rank, with the index reaching the GPUs of the
// other ranks by ncclBroadcast (RCCL over xGMI) instead of one H2D copy per rank.
//
// Replaces, for a one-rank-per-GPU run, what mpiBWA does per rank after attaching the `.map` image
// (src/parallel_aux.c:1745-1838: every rank of a shared-memory group reads the mapped index through the CPU caches).
// The reference has no device side, so there is nothing to mirror line by line; the contract is SURVEY.md §8b/§8e:
// exactly one collective, at start-up, on the three index arrays.
//
// RCCL is loaded at run time (dlopen of librccl.so.1: the library ROCm ships and PyTorch-ROCm bundles), so the product
// library carries no link-time dependency on it and single-rank users never load it.  The library does not link MPI
// either: the caller lends a host-memory broadcast (mpiBWA: a two-line wrapper of MPI_Bcast) for the 128-byte RCCL
// bootstrap id.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "device.h"

namespace mbw {

#define HIP_OK(call)                                                                                             \
	do {                                                                                                         \
		hipError_t e_ = (call);                                                                                  \
		if (e_ != hipSuccess) die("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);    \
	} while (0)

// the five RCCL entry points used, with the types of rccl.h (ncclUniqueId = 128 opaque bytes, ncclUint8 = 1)
struct RcclId { char internal[128]; };
typedef void *RcclComm;
struct Rccl {
	void *so = nullptr;
	int (*GetUniqueId)(RcclId *) = nullptr;
	int (*CommInitRank)(RcclComm *, int, RcclId, int) = nullptr;
	int (*Broadcast)(const void *, void *, size_t, int, int, RcclComm, hipStream_t) = nullptr;
	int (*CommDestroy)(RcclComm) = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
};

static Rccl &rccl()
{
	static Rccl r;
	if (r.so) return r;
	const char *names[] = {getenv("MPIBWA_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	for (const char *n : names) {
		if (!n) continue;
		r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
		if (r.so) break;
	}
	if (!r.so) die("mi355x_init: cannot load RCCL (librccl.so.1): %s", dlerror());
	auto sym = [&](const char *s) { void *p = dlsym(r.so, s); if (!p) die("mi355x_init: RCCL lacks %s", s); return p; };
	r.GetUniqueId = (int (*)(RcclId *))sym("ncclGetUniqueId");
	r.CommInitRank = (int (*)(RcclComm *, int, RcclId, int))sym("ncclCommInitRank");
	r.Broadcast = (int (*)(const void *, void *, size_t, int, int, RcclComm, hipStream_t))sym("ncclBroadcast");
	r.CommDestroy = (int (*)(RcclComm))sym("ncclCommDestroy");
	r.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
	return r;
}

#define RCCL_OK(call)                                                                                           \
	do {                                                                                                        \
		int e_ = (call);                                                                                        \
		if (e_ != 0) die("%s failed: %s (%s:%d)", #call, rccl().GetErrorString(e_), __FILE__, __LINE__);        \
	} while (0)

static double g_bcast_seconds = 0;

// position-sensitive sum of the 64-bit words of a device array (order-independent: any grid computes the same number)
__global__ void __launch_bounds__(256) checksum_kernel(const unsigned long long *__restrict__ w, size_t n_words, unsigned long long *out)
{
	unsigned long long h = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x)
		h += (w[i] ^ (i * 0x9E3779B97F4A7C15ull)) * (2 * i + 1);
	for (int o = 32; o; o >>= 1) h += __shfl_xor(h, o);
	if ((threadIdx.x & 63) == 0 && h) atomicAdd(out, h);
}

} // namespace mbw
using namespace mbw;

// checksums of the three resident index arrays (occ blocks, sampled SA, pac), computed on the device: what a rank received by
// broadcast must hash to what rank 0 sent
extern "C" int mi355x_index_checksums(uint64_t out[3])
{
	void *d[3]; size_t cap[3];
	if (mi355x_index_buffers(&d[0], &cap[0], &d[1], &cap[1], &d[2], &cap[2]) != 0) return -1;
	unsigned long long *d_sum;
	HIP_OK(hipMalloc(&d_sum, 3 * 8));
	HIP_OK(hipMemset(d_sum, 0, 3 * 8));
	for (int w = 0; w < 3; ++w)
		hipLaunchKernelGGL(checksum_kernel, dim3(2048), dim3(256), 0, 0, (const unsigned long long *)d[w], cap[w] / 8, d_sum + w);
	HIP_OK(hipDeviceSynchronize());
	HIP_OK(hipGetLastError());
	HIP_OK(hipMemcpy(out, d_sum, 3 * 8, hipMemcpyDeviceToHost));
	(void)hipFree(d_sum);
	return 0;
}

using namespace mbw;

extern "C" double mi355x_init_bcast_seconds(void) { return g_bcast_seconds; }

extern "C" int mi355x_rank_host_threads(int *ranks_on_node);
// One line per rank about what it runs on, and a refusal to start on a share of the host that cannot feed a GPU: the host stages
// of a call (encoding, region clean-up, the pairs the pairing kernel leaves, SAM text of the records the device does not write)
// take 1.3 CPU-s per 667 k reads on SURVEY 8d's workload — about nine busy cores at the rate one MI355X sustains; below two
// threads per rank the calls in flight only queue behind each other (MPIBWA_ALLOW_FEW_CORES=1 starts anyway).
static void report_rank(int local_rank, const mi355x_comm_t *comm)
{
	int ranks = 1;
	const int thr = mi355x_rank_host_threads(&ranks);
	if (bwa_verbose >= 3)
		fprintf(stderr, "[M::mi355x_init] rank %d of %d: GPU %d of %d visible, %d host threads (this rank's share of the node: %d ranks on it)%s\n",
		        comm ? comm->rank : 0, comm ? comm->size : 1, local_rank, mi355x_device_count(), thr, ranks,
		        thr < 8 ? " - fewer than 8: the host stages will bound the throughput of this rank" : "");
	if (thr < 2 && !(getenv("MPIBWA_ALLOW_FEW_CORES") && atoi(getenv("MPIBWA_ALLOW_FEW_CORES"))))
		die("mi355x_init: %d host thread for this rank (%d ranks share the node's usable CPUs): a rank needs at least 2, about 9 to keep its GPU busy; "
		    "start fewer ranks per node, give the job more cores, or set MPIBWA_ALLOW_FEW_CORES=1", thr, ranks);
}

extern "C" int mi355x_init(int local_rank, const bwaidx_t *idx, const mi355x_comm_t *comm)
{
	if (!idx || !idx->bwt || !idx->bns || !idx->pac) die("mi355x_init: the index handle is not attached");
	g_bcast_seconds = 0;
	report_rank(local_rank, comm);
	if (!comm) return mi355x_index_upload(local_rank, idx->bwt, idx->bns, idx->pac);   // (a communicator of one rank still goes through RCCL)
	if (comm->size < 1 || comm->rank < 0 || comm->rank >= comm->size || !comm->bcast) die("mi355x_init: bad communicator description");

	// rank 0 puts its host copy into HBM; everybody else only allocates the three device arrays
	if (comm->rank == 0) {
		if (mi355x_index_alloc(local_rank, idx->bwt, idx->bns) != 0) return -1;
		void *d[3]; size_t cap[3];
		mi355x_index_buffers(&d[0], &cap[0], &d[1], &cap[1], &d[2], &cap[2]);
		HIP_OK(hipMemcpy(d[0], idx->bwt->bwt, (size_t)idx->bwt->bwt_size * 4, hipMemcpyHostToDevice));
		HIP_OK(hipMemcpy(d[1], idx->bwt->sa, (size_t)idx->bwt->n_sa * 8, hipMemcpyHostToDevice));
		HIP_OK(hipMemcpy(d[2], idx->pac, (size_t)idx->bns->l_pac / 4 + 1, hipMemcpyHostToDevice));
	} else if (mi355x_index_alloc(local_rank, idx->bwt, idx->bns) != 0) return -1;

	(void)hipGetLastError();   // RCCL checks the thread's last HIP error after its own calls: start from a clean slate
	Rccl &R = rccl();
	RcclId id;
	memset(&id, 0, sizeof id);
	if (comm->rank == 0) RCCL_OK(R.GetUniqueId(&id));
	comm->bcast(&id, sizeof id, 0, comm->user);
	RcclComm nc = nullptr;
	RCCL_OK(R.CommInitRank(&nc, comm->size, id, comm->rank));

	void *d[3]; size_t cap[3];
	mi355x_index_buffers(&d[0], &cap[0], &d[1], &cap[1], &d[2], &cap[2]);
	hipStream_t st;
	HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	const auto t0 = std::chrono::steady_clock::now();
	// in place on the index arrays, in pieces: a ring broadcast pipelines piece k+1 behind piece k on every xGMI hop, and
	// no staging buffer is needed (the arrays have the same size on every rank: they are sized from the same metadata)
	size_t piece = (size_t)256 << 20;
	if (const char *e = getenv("MPIBWA_BCAST_PIECE_MB")) { long v = atol(e); if (v > 0) piece = (size_t)v << 20; }
	for (int w = 0; w < 3; ++w)
		for (size_t o = 0; o < cap[w]; o += piece) {
			const size_t n = cap[w] - o < piece ? cap[w] - o : piece;
			RCCL_OK(R.Broadcast((const char *)d[w] + o, (char *)d[w] + o, n, /*ncclUint8*/ 1, 0, nc, st));
		}
	HIP_OK(hipStreamSynchronize(st));
	g_bcast_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	HIP_OK(hipStreamDestroy(st));
	RCCL_OK(R.CommDestroy(nc));
	// Did every rank get what rank 0 sent?  Each rank hashes its three arrays on its device, rank 0's numbers travel through the
	// caller's host broadcast, and a rank that differs ends the run here instead of aligning against a damaged index.
	{
		uint64_t mine[3] = {0, 0, 0}, root[3];
		if (mi355x_index_checksums(mine) != 0) die("mi355x_init: no resident index to verify");
		memcpy(root, mine, sizeof root);
		comm->bcast(root, sizeof root, 0, comm->user);
		for (int w = 0; w < 3; ++w)
			if (root[w] != mine[w])
				die("mi355x_init: rank %d of %d received a damaged index: array %d (%s) hashes to %016llx, rank 0 sent %016llx", comm->rank, comm->size, w,
				    w == 0 ? "occ blocks" : w == 1 ? "sampled SA" : "pac", (unsigned long long)mine[w], (unsigned long long)root[w]);
		if (const char *e = getenv("MPIBWA_VERBOSE_INIT"))
			if (atoi(e)) fprintf(stderr, "[mpibwa_amd] rank %d of %d: index verified against rank 0 (%016llx %016llx %016llx), broadcast %.3f s\n", comm->rank, comm->size,
			                     (unsigned long long)mine[0], (unsigned long long)mine[1], (unsigned long long)mine[2], g_bcast_seconds);
	}
	// every rank expands its own dense SA and jump table from what it now holds (device-local, no further traffic)
	return mi355x_index_commit();
}
