// pipeline.hip — mem_process_seqs(): the drop-in batch driver (replaces src/bwamem.c:1161-1234).
//
// The reference runs worker1 (seed -> chain -> extend, per read) and worker2
// (pairing -> SAM, per pair) over pthreads with a batch-wide mem_pestat
// barrier in between.  Here the batch is processed stage by stage:
//
//   host   nt4-encode, pack reads                              (bwamem.c:1057-1058)
//   GPU    SMEM seeding                 smem_kernel            (bwamem.c:114-162)
//   GPU    interval sort / seed enumeration                    (bwamem.c:161, 265-283)
//   GPU    suffix-array lookup          sa_dense_kernel        (bwt.c:86-96)
//   GPU    chaining + chain filters     chain_kernel           (bwamem.c:251-385, 598-617); host for reads beyond one B-tree node
//   GPU    chain -> regions, banded DP  c2a_kernel             (bwamem.c:632-786, ksw.c:380-479)
//   host   dedup / patch, insert-size votes                    (bwamem.c:439-497, bwamem_pair.c:59-109)
//   ---- mem_pestat over the whole chunk ----
//   GPU    mate-rescue local alignment  msw_kernel             (bwamem_pair.c:111-180, ksw.c:111-356)
//   host   pairing decisions                                   (bwamem_pair.c:182-388)
//   GPU    CIGAR / MD / NM              aln_kernel             (bwamem.c:1106-1122, bwa.c:121-207)
//   host   SAM text                                            (bwamem.c:824-1010)
//
// Up to eight calls run side by side (CallCtx below): the GPU-bound first half of one chunk overlaps the host-bound
// second half of another.
//
// There is no CPU fallback for the GPU stages: without a gfx950 device the call aborts.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/resource.h>
#include <sys/time.h>
#include <unistd.h>

#include "device.h"
#include "host.h"
#include "hprof.h"

namespace mbw {

#define HIP_OK(call)                                                                                             \
	do {                                                                                                         \
		hipError_t e_ = (call);                                                                                  \
		if (e_ != hipSuccess) die("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);    \
	} while (0)

static mi355x_stats_t g_stats;

// CPUs this process may actually use: the cgroup CPU quota when there is one, else the online CPU count
static int usable_cpus_now()
{
	int hw = (int)std::thread::hardware_concurrency();
	if (hw <= 0) hw = 1;
	if (FILE *fp = fopen("/sys/fs/cgroup/cpu.max", "r")) {
		char q[64];
		long long period = 0;
		if (fscanf(fp, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
			long long quota = atoll(q);
			int c = (int)((quota + period - 1) / period);
			if (c >= 1 && c < hw) hw = c;
		}
		fclose(fp);
	}
	return hw;
}
// (asked once: every call used to open the cgroup file again, and under eight calls in flight that open alone held a call up
// for milliseconds)
static int usable_cpus() { static const int c = usable_cpus_now(); return c; }

// Which GPU a lazily uploading rank takes, and how many ranks share the node: the launcher's environment
// (torchrun, Open MPI, MVAPICH2, Slurm/PMI, Intel MPI / MPICH hydra).  -1 / 0 when the launcher says nothing.
static int env_int(const char *const *names, int dflt)
{
	for (; *names; ++names)
		if (const char *e = getenv(*names)) return atoi(e);
	return dflt;
}
static int env_local_rank()
{
	static const char *const n[] = {"LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID", "PMI_LOCAL_RANK",
	                                "MPI_LOCALRANKID", nullptr};
	return env_int(n, -1);
}
static int env_local_size()
{
	static const char *const n[] = {"LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MV2_COMM_WORLD_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE", "PMI_LOCAL_SIZE",
	                                "MPI_LOCALNRANKS", nullptr};
	return env_int(n, 0);
}

static int host_threads(const mem_opt_t *opt)
{
	if (const char *e = getenv("MPIBWA_HOST_THREADS")) { int v = atoi(e); if (v > 0) return v; }
	// the result does not depend on the thread count (as in the reference), so use what the box gives us, divided among
	// the ranks that share this node (one rank per GPU) — but never more than the caller's -t when it asked for several
	int ranks = env_local_size();
	if (ranks < 1) ranks = 1;
	int thr = std::max(1, std::min(usable_cpus() / ranks, 128));
	if (opt && opt->n_threads > 1) thr = std::min(thr, opt->n_threads);
	return thr;
}

// ---- the library's helper threads ----
// One persistent pool for all calls in flight (created on first use, as many threads as the rank's share of the node's CPUs).  A parallel region
// queues one ticket per helper it would like; a pool thread that takes a ticket runs the region's work loop until the region's
// items are gone; the caller runs the same loop, then withdraws the tickets nobody has taken and waits for the helpers that
// did start.  (Regions used to create and join their own threads: ~450 thread creations per call, six calls in flight — stack
// mappings, page faults and exits that all serialise on the process's address-space lock — and up to 96 runnable threads on 16 cores.)
// MPIBWA_THREAD_POOL=0: threads per region as before.
class HelperPool {
public:
	struct Job {
		void (*run)(void *, int);   // (region, helper number 1..)
		void *region;
		std::atomic<int> started{0}, finished{0};
	};
	static HelperPool &get() { static HelperPool *p = new HelperPool;   // never destroyed: its threads wait on it until the process ends
		return *p; }
	bool enabled() const { return !th_.empty(); }
	void run(int helpers, Job &job, void (*self)(void *), void *region)
	{
		if (helpers > 0) {
			{
				std::lock_guard<std::mutex> lk(m_);
				for (int t = 0; t < helpers; ++t) q_.push_back(&job);
			}
			for (int t = 0; t < helpers; ++t) cv_.notify_one();   // (not notify_all: the pool may hold many more threads than this region asks for)
		}
		self(region);
		if (helpers > 0) {
			int mine = 0;
			{
				std::lock_guard<std::mutex> lk(m_);
				for (auto it = q_.begin(); it != q_.end();)
					if (*it == &job) { it = q_.erase(it); ++mine; } else ++it;
			}
			const int took = helpers - mine;   // tickets a pool thread has taken (it bumps `finished` when it is done with the region)
			for (int spin = 0; job.finished.load(std::memory_order_acquire) < took; ++spin)
				if (spin < 200) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(20));
		}
	}
private:
	HelperPool()
	{
		const char *e = getenv("MPIBWA_THREAD_POOL");
		if (e && atoi(e) == 0) return;
		const int n = host_threads(nullptr);   // this rank's share of the node's CPUs (MPIBWA_HOST_THREADS overrides)
		for (int t = 0; t < n; ++t) th_.emplace_back([this]() { loop(); });
		for (auto &t : th_) t.detach();   // they live as long as the process
	}
	void loop()
	{
		for (;;) {
			Job *j;
			{
				std::unique_lock<std::mutex> lk(m_);
				cv_.wait(lk, [this]() { return !q_.empty(); });
				j = q_.front();
				q_.pop_front();
			}
			const int tid = j->started.fetch_add(1) + 1;
			j->run(j->region, tid);
			if (g_hprof_on) t_hprof.flush();   // (hprof.h: a helper's record is folded in when it leaves a region)
			j->finished.fetch_add(1, std::memory_order_release);
		}
	}
	std::mutex m_;
	std::condition_variable cv_;
	std::deque<Job *> q_;
	std::vector<std::thread> th_;
};

template <class F>
static void parallel_for(int n_threads, int n, int chunk, F f)
{
	if (n <= 0) return;
	if (n_threads <= 1 || n <= chunk) { for (int i = 0; i < n; ++i) f(i); return; }
	struct Region {
		std::atomic<int> next{0};
		int n, chunk;
		F *f;
		void work()
		{
			for (;;) {
				int b = next.fetch_add(chunk);
				if (b >= n) break;
				int e = std::min(n, b + chunk);
				for (int i = b; i < e; ++i) (*f)(i);
			}
		}
	} R;
	R.n = n; R.chunk = chunk; R.f = &f;
	const int helpers = std::min(n_threads - 1, (n + chunk - 1) / chunk - 1);
	HelperPool &P = HelperPool::get();
	if (P.enabled()) {
		HelperPool::Job job;
		job.run = [](void *r, int) { ((Region *)r)->work(); };
		job.region = &R;
		P.run(helpers, job, [](void *r) { ((Region *)r)->work(); }, &R);
		return;
	}
	std::vector<std::thread> th;
	for (int t = 0; t < helpers; ++t) th.emplace_back([&R]() { R.work(); });
	R.work();
	for (auto &t : th) t.join();
}

// same, handing whole blocks to f(thread, block, lo, hi) so that a stage can keep per-thread scratch and per-block output
// (thread numbers are 0 .. n_threads - 1 and unique among the threads working on the region at the same time)
template <class F>
static void parallel_blocks(int n_threads, int n, int chunk, F f)
{
	if (n <= 0) return;
	const int nb = (n + chunk - 1) / chunk;
	if (n_threads > nb) n_threads = nb;
	struct Region {
		std::atomic<int> next{0};
		int n, nb, chunk;
		F *f;
		void work(int tid)
		{
			for (;;) {
				int b = next.fetch_add(1);
				if (b >= nb) break;
				(*f)(tid, b, b * chunk, std::min(n, (b + 1) * chunk));
			}
		}
	} R;
	R.n = n; R.nb = nb; R.chunk = chunk; R.f = &f;
	HelperPool &P = HelperPool::get();
	if (P.enabled() && n_threads > 1) {
		HelperPool::Job job;
		job.run = [](void *r, int tid) { ((Region *)r)->work(tid); };
		job.region = &R;
		P.run(n_threads - 1, job, [](void *r) { ((Region *)r)->work(0); }, &R);
		return;
	}
	std::vector<std::thread> th;
	for (int t = 1; t < n_threads; ++t) th.emplace_back([&R, t]() { R.work(t); });
	R.work(0);
	for (auto &t : th) t.join();
}

// MPIBWA_C2A_EARLY: 1 (default) the extension row loops stop early, 0 they run the reference's rows, 2 both with a fatal error on any difference
static int c2a_early_mode() { const char *e = getenv("MPIBWA_C2A_EARLY"); return e ? atoi(e) : 1; }
// The turns on the big kernels are taken in the order of arrival.  With a plain mutex a caller whose thread had to be scheduled
// first (more runnable threads than cores) kept losing the turn to callers that were already running: now and then a chunk that
// takes 0.6 s took 3 s with eight callers, the others none the faster for it.
class TurnLock {
public:
	void lock()
	{
		std::unique_lock<std::mutex> lk(m_);
		const unsigned long long mine = next_++;
		cv_.wait(lk, [&]() { return serving_ == mine; });
	}
	void unlock()
	{
		{ std::lock_guard<std::mutex> lk(m_); ++serving_; }
		cv_.notify_all();
	}
private:
	std::mutex m_;
	std::condition_variable cv_;
	unsigned long long next_ = 0, serving_ = 0;
};
static TurnLock g_smem_turn, g_c2a_turn;
static std::mutex g_pes_lock;

static double now_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static double sys_sec()
{
	struct rusage r;
	getrusage(RUSAGE_SELF, &r);
	return r.ru_stime.tv_sec + 1e-6 * r.ru_stime.tv_usec;
}
static long page_faults()
{
	struct rusage r;
	getrusage(RUSAGE_SELF, &r);
	return r.ru_minflt;
}
static double cpu_sec()
{
	struct rusage r;
	getrusage(RUSAGE_SELF, &r);
	return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
}

// Wait for a stream without burning a core: hipStreamSynchronize spins, and a spinning thread counts against the
// container's CPU quota like a working one (the host stages are the other half of the bottleneck).  Polling with a short
// sleep costs at most ~0.1 ms per wait.
// MPIBWA_SAMPLE=1: where the calls in flight are, sampled every 250 us (which stage, waiting for a kernel or working on the
// host), printed when the process exits: which stages the calls sit in while nobody keeps the GPU busy, and the other way round
static const int SAMPLE_SLOTS = 64, SAMPLE_STAGES = 64, STG_WAIT = 0x100;
static std::atomic<int> g_stage[SAMPLE_SLOTS];
static thread_local std::atomic<int> *t_stage = nullptr;
static inline void stage(int id) { if (t_stage) t_stage->store(id, std::memory_order_relaxed); }
struct StageSampler {
	std::thread th;
	std::atomic<bool> stop{false};
	uint64_t n_samples = 0, in_stage[SAMPLE_STAGES][2] = {{0}}, by_wait[SAMPLE_SLOTS + 1] = {0}, by_host[SAMPLE_SLOTS + 1] = {0};
	uint64_t host_when_idle[SAMPLE_STAGES] = {0}, n_gpu_idle = 0, n_host_idle = 0;
	int min_calls = 1;   // MPIBWA_SAMPLE=n: only samples with at least n calls in flight count
	void run()
	{
		while (!stop.load()) {
			int nw = 0, nh = 0, st[SAMPLE_SLOTS], ns = 0;
			for (int i = 0; i < SAMPLE_SLOTS; ++i) {
				const int v = g_stage[i].load(std::memory_order_relaxed);
				if (!v) continue;
				st[ns++] = v;
				const int id = v & 63, w = (v & STG_WAIT) != 0;
				if (w) ++nw; else if (id != 20 && id != 50) ++nh;
			}
			if (ns >= min_calls) {
				++n_samples; ++by_wait[nw]; ++by_host[nh];
				for (int k = 0; k < ns; ++k) ++in_stage[st[k] & 63][(st[k] & STG_WAIT) != 0];
				if (!nw) { ++n_gpu_idle; for (int k = 0; k < ns; ++k) ++host_when_idle[st[k] & 63]; }
				if (!nh) ++n_host_idle;
			}
			usleep(250);
		}
	}
	void report()
	{
		if (!n_samples) return;
		fprintf(stderr, "[sample] %llu samples with a call in flight; no call waiting for the GPU in %.1f %%, no call on the host in %.1f %%\n",
		        (unsigned long long)n_samples, 100.0 * n_gpu_idle / n_samples, 100.0 * n_host_idle / n_samples);
		fprintf(stderr, "[sample] calls waiting for the GPU:");
		for (int k = 0; k <= 12; ++k) fprintf(stderr, " %d:%.1f%%", k, 100.0 * by_wait[k] / n_samples);
		fprintf(stderr, "\n[sample] calls working on the host:");
		for (int k = 0; k <= 12; ++k) fprintf(stderr, " %d:%.1f%%", k, 100.0 * by_host[k] / n_samples);
		fprintf(stderr, "\n[sample] stage: mean calls in it on the host / waiting for the GPU / on the host while no call waits for the GPU\n");
		for (int id = 0; id < SAMPLE_STAGES; ++id)
			if (in_stage[id][0] + in_stage[id][1])
				fprintf(stderr, "[sample]   %2d: %.3f %.3f %.3f\n", id, (double)in_stage[id][0] / n_samples, (double)in_stage[id][1] / n_samples,
				        n_gpu_idle ? (double)host_when_idle[id] / n_gpu_idle : 0.0);
	}
};
static StageSampler *g_sampler = nullptr;
void sampler_report()
{
	if (!g_sampler) return;
	g_sampler->stop.store(true);
	g_sampler->th.join();
	g_sampler->report();
	delete g_sampler;
	g_sampler = nullptr;
}

static void stream_wait(hipStream_t st)
{
	static const bool spin = getenv("MPIBWA_SPIN_WAIT") != nullptr;
	struct Mark {
		Mark() { if (t_stage) t_stage->fetch_or(STG_WAIT, std::memory_order_relaxed); }
		~Mark() { if (t_stage) t_stage->fetch_and(~STG_WAIT, std::memory_order_relaxed); }
	} mark;
	if (!spin) {
		for (;;) {
			hipError_t e = hipStreamQuery(st);
			if (e == hipSuccess) break;
			if (e != hipErrorNotReady) HIP_OK(e);
			(void)hipGetLastError();   // "not ready" is this thread's last error otherwise: the next library that checks it (RCCL) takes it for a failure
			usleep(60);
		}
	}
	HIP_OK(hipStreamSynchronize(st));
}

struct EvTimer {
	hipEvent_t a, b;
	EvTimer() { HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b)); }
	~EvTimer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
	void start(hipStream_t s) { HIP_OK(hipEventRecord(a, s)); }
	void stop(hipStream_t s) { HIP_OK(hipEventRecord(b, s)); }
	double ms() { HIP_OK(hipEventSynchronize(b)); float m = 0; HIP_OK(hipEventElapsedTime(&m, a, b)); return m; }
};

// device work buffers, grown on demand and kept across calls
// grow-only host buffer kept across calls (no page faults / frees per chunk)
struct HostBuf {
	void *p = nullptr; size_t cap = 0;
	void *ensure(size_t bytes) { if (bytes > cap) { free(p); cap = bytes + bytes / 4 + 4096; p = malloc(cap); if (!p) die("out of memory"); } return p; }
};

// grow-only page-locked host buffer: staging for the bulk H2D / D2H copies (full PCIe rate, no per-chunk page faults)
struct PinBuf {
	void *p = nullptr; size_t cap = 0;
	void *ensure(size_t bytes)
	{
		if (bytes > cap) {
			note_buffer_growth(cap, bytes + bytes / 4 + 4096, "page-locked");
			if (p) HIP_OK(hipHostFree(p));
			cap = bytes + bytes / 4 + 4096;
			HIP_OK(hipHostMalloc(&p, cap, hipHostMallocDefault));
		}
		return p;
	}
};

struct Workspace {
	PinBuf h_nch, h_cbeg, h_ccnt, h_rbeg, h_nseeds, h_lrep, h_nintv;
	// (every copy to or from the device uses page-locked host memory: a pageable target makes hipMemcpyAsync wait — spinning —
	// for the kernels queued before it, and a pageable source is pinned page by page at every call)
	PinBuf h_cnt, h_off, h_len, h_seed_off, h_areq[2];
	DevBuf nch, chain_cnt, reg_pos, regs_packed, ann_off, ann_alt, pack_tmp, order, chain_gen, c2a_stat;
	PinBuf h_c2a_stat;
	PinBuf h_order;
	// reads with many chains, extended as independent groups of chains (c2a_groups.hip)
	PinBuf h_heavy, h_hoff, h_nunits;
	DevBuf heavy, hoff, grp_scratch, grp_clist, grp_ustart, grp_unit_rd, grp_unit_av, grp_nunits, c_rabs, c_rcnt;
	PinBuf h_regs2;
	PinBuf h_flat, h_sa, h_qbl, h_chains, h_seeds, h_srt, h_regs, h_nregs, h_mreq[2], h_mres[2], h_ahdr[2], h_apool[2];
	DevBuf mreq[2], mres[2], mrows[2], alist[2], mlist[2];
	PinBuf h_mlist[2];
	DevBuf seq, off, len, intv, nintv, cnt, scratch, nseeds, lrep, seed_off, rows, qbl, sa;
	DevBuf chain_off, chains, seeds, srt, reg_off, regs, nregs, tab, areq, ahdr, apool, agap, acnt, areq2, ahdr2, apool2, acnt2;
	// SAM text on the device (sam_kernel.hip): line descriptors, names / qualities of the chunk, contig names, output arena per part
	DevBuf sdesc, sbase[2], squal, snames, snoff, sann_names, sann_noff, sarena[2], sused[2], sooff[2], solen[2];
	PinBuf h_sdesc, h_names, h_noff, h_qual, h_sarena[2], h_sooff[2], h_solen[2], h_sbase[2];
	// pairs decided on the device (pair_kernel.hip): first region / region count per read, flags, tables, requests, descriptors
	DevBuf pr_first, pr_nfirst, pr_ok, pr_status, pr_ptab, pr_ltab, pr_req, pr_desc;
	PinBuf h_pr_ok, h_pr_status, h_pr_tab;
	// ... and their CIGAR / SAM-text job per part, launched right behind the pairing kernel
	DevBuf dj_hdr[2], dj_pool[2], dj_cnt[2], dj_list[2], dj_base[2], dj_arena[2], dj_used[2], dj_ooff[2], dj_olen[2];
	PinBuf hj_hdr[2], hj_pool[2], hj_arena[2], hj_ooff[2], hj_olen[2], hj_base[2];
	PinBuf h_small[2];   // counters coming back from the SAM stage's jobs (a pageable target would make the copy spin behind queued kernels)
};
static const int MAX_LANES = 4;
// Everything one mem_process_seqs() call owns between its first and last line.  Eight of them: eight caller threads may be inside
// the function at once (chunk i+1 seeding and extending on the GPU while the host pairs and prints chunk i — the stage that
// keeps the GPU busy and the stage that keeps the host busy belong to different halves of a call).  A ninth caller waits.
struct CallCtx {
	// (first and last members: every DevBuf constructed in between enters `bufs`)
	std::vector<DevBuf *> bufs;
	struct Open { Open(std::vector<DevBuf *> *l) { g_devbuf_owner = l; } } open_{&bufs};
	// ws[lane] / reg_arena[k]: a call with neighbours in flight runs its chunk in one piece through ws[0], a lone call its two
	// sub-batches through ws[0] and ws[1].  The buffers only grow, so a context that has served a lone call regrows ws[0] ONCE,
	// at its first whole chunk (65 buffers: hipFree + hipMalloc stall every stream) — a caller that wants that out of its
	// measurements starts its first rounds of calls together, as bench.py's warm-up does.  (Separate buffer sets per mode
	// were tried: no regrowth at all, but twice the footprint in the first two contexts, and the repeat-rich workload of §6.1
	// no longer fitted with four calls in flight.)
	Workspace ws[MAX_LANES + 1];
	HostBuf reg_arena[17];     // the regions live until the SAM stage
	Workspace gws;             // batch-wide buffers (packed reads, CIGAR requests)
	hipStream_t p_streams[MAX_LANES] = {nullptr}, a_streams[2] = {nullptr, nullptr}, d_streams[2] = {nullptr, nullptr};
	bool busy = false;
	const bseq1_t *seq_lo = nullptr, *seq_hi = nullptr;   // the caller's array while the call runs
	int calls_done = 0;   // since its buffers were last given back (under g_ctx_mu)
	struct Close { Close() { g_devbuf_owner = nullptr; } } close_;
	size_t device_bytes() const { size_t b = 0; for (const DevBuf *d : bufs) b += d->cap; return b; }
	void release_device() { for (DevBuf *d : bufs) d->release(); calls_done = 0; }
};
static const int MAX_CALLS = 12;
static CallCtx g_ctx[MAX_CALLS];
static std::mutex g_ctx_mu;
static std::recursive_mutex g_init_mu;
std::recursive_mutex &index_mutex() { return g_init_mu; }
static std::condition_variable g_ctx_cv;
// Calls admitted at once: MAX_CALLS until a work buffer has failed to fit, then what was in flight at that moment minus one (never
// fewer than one): the callers beyond that wait at the door instead of in the middle of a call.
static int g_admit = MAX_CALLS;             // (under g_ctx_mu)
// What a call context holds in HBM once it has seen a chunk (the largest seen so far) — what a call that starts on a fresh context
// will come to hold.  A call is only let in next to others when the HBM that is free, less what the calls in flight may still grow
// by, covers that (+ the runtime's reserve); until a first call has ended nobody knows and a generous guess stands in.  On references
// whose chunks need tens of GB per context (half the genome in repeats) this is what keeps eight callers from filling the HBM
// together and then all waiting for each other.
static size_t g_footprint = 0;              // (under g_ctx_mu)
static bool room_for_another_call(const CallCtx *cand, int n_busy, int n_reads)
{
	if (n_busy == 0) return true;
	// (before any call has ended: 64 KB per read of the chunk — what a chunk needs when half the reference is high-copy repeats, ten
	// times what an ordinary one does: the first calls of a run start side by side as far as that fits, five on a 288-GB device)
	const size_t foot = g_footprint ? g_footprint : (size_t)std::max(n_reads, 1) * 65536;
	size_t fr = 0, tot = 0;
	if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return true; }
	// (a context that has been through a call holds what a chunk needs, give or take; only the fresh ones still have their growth
	// to come.  Counting the difference to the LARGEST context against every one of them kept callers at the door for seconds on a
	// device whose contexts were all in place: eight of them, each a little smaller than the largest, "owed" more than was free)
	size_t promised = 0;
	for (int i = 0; i < MAX_CALLS; ++i)
		if (g_ctx[i].busy && g_ctx[i].calls_done == 0 && g_ctx[i].device_bytes() < foot) promised += foot - g_ctx[i].device_bytes();
	const size_t have = cand->device_bytes(), need = (cand->calls_done == 0 && have < foot ? foot - have + ((size_t)8 << 30) : (size_t)1 << 30);
	return fr >= promised + need;
}
// what the caller has said about itself (mi355x_prewarm's n_calls): with three or more calls kept in flight, the first calls of its
// loop run in the busy mode too — they used to start as lone callers (two sub-batches, each waiting for its turns among the
// others' kernels) and took 2-3 s instead of 0.6
static std::atomic<int> g_expected_calls(0);
void expect_calls_in_flight(int n) { g_expected_calls.store(n, std::memory_order_relaxed); }
static int g_waiting_for_memory = 0;        // calls stuck in device_memory_pressure (under g_ctx_mu)
static std::condition_variable g_mem_cv;    // a call has ended
struct CtxLease {
	CallCtx *c = nullptr;
	int others = 0;   // calls that were in flight when this one started
	// ... or in the last two seconds: a caller that keeps several calls in flight is treated as such from its third call on, also
	// when several of its calls happen to end together
	bool crowded = false;
	CtxLease(const bseq1_t *seqs, int n)
	{
		const double t_door = now_ms();
		std::unique_lock<std::mutex> lk(g_ctx_mu);
		for (;;) {
			int n_busy = 0;
			for (int i = 0; i < MAX_CALLS; ++i) n_busy += g_ctx[i].busy ? 1 : 0;
			if (n_busy < g_admit) {
				// a context that already holds buffers first (an idle one with buffers next to a busy fresh one would be HBM nobody uses)
				CallCtx *cand = nullptr;
				for (int pass = 0; pass < 2 && !cand; ++pass)
					for (int i = 0; i < MAX_CALLS && !cand; ++i)
						if (!g_ctx[i].busy && (pass == 1 || g_ctx[i].device_bytes() > 0)) cand = &g_ctx[i];
				if (cand && room_for_another_call(cand, n_busy, n)) { c = cand; c->busy = true; }
			}
			if (c) break;
			g_ctx_cv.wait_for(lk, std::chrono::milliseconds(20));   // (a call that ends wakes the waiters; so does memory given back without one ending)
		}
		for (int i = 0; i < MAX_CALLS; ++i) {
			if (!g_ctx[i].busy || &g_ctx[i] == c) continue;
			++others;
			// calls in flight must not share reads: each writes the seq[] and sam of its own
			if (seqs < g_ctx[i].seq_hi && g_ctx[i].seq_lo < seqs + n)
				die("mem_process_seqs: called on seqs[] that another call in flight is still working on");
		}
		c->seq_lo = seqs; c->seq_hi = seqs + n;
		const double now = now_ms();
		static const bool door_log = getenv("MPIBWA_DOOR_LOG") != nullptr;   // calls that waited to be let in (admission by HBM head room)
		if (door_log && now - t_door > 50.) fprintf(stderr, "[door] a call of %d reads waited %.0f ms to be admitted next to %d others\n", n, now - t_door, others);
		if (others >= 2) last_crowded_ms() = now;
		crowded = now - last_crowded_ms() < 2000.0 || g_expected_calls.load(std::memory_order_relaxed) >= 3;
	}
	static double &last_crowded_ms() { static double t = -1e30; return t; }   // (under g_ctx_mu)
	~CtxLease()
	{
		{
			std::lock_guard<std::mutex> lk(g_ctx_mu);
			// (also when a call ENDS among two others: a caller whose calls take longer than the two seconds — first use of its contexts,
			// a hard reference — would otherwise start its next round as a lone caller)
			int n_busy = 0;
			for (int i = 0; i < MAX_CALLS; ++i) n_busy += g_ctx[i].busy ? 1 : 0;
			if (n_busy >= 3) last_crowded_ms() = now_ms();
			if (c->device_bytes() > g_footprint) g_footprint = c->device_bytes();
			++c->calls_done;
			c->busy = false; c->seq_lo = c->seq_hi = nullptr;
		}
		g_ctx_cv.notify_all();
		g_mem_cv.notify_all();
	}
};
// give back the device buffers of every context that is not inside a call; returns the bytes freed
static size_t release_idle_locked(std::unique_lock<std::mutex> &lk)
{
	size_t freed = 0;
	for (int i = 0; i < MAX_CALLS; ++i) {
		CallCtx &x = g_ctx[i];
		if (x.busy || x.device_bytes() == 0) continue;
		x.busy = true;              // nobody leases it while its buffers go
		lk.unlock();
		freed += x.device_bytes();
		x.release_device();
		lk.lock();
		x.busy = false;
	}
	return freed;
}
void release_idle_work_buffers()
{
	std::unique_lock<std::mutex> lk(g_ctx_mu);
	release_idle_locked(lk);
}
bool device_memory_pressure(size_t wanted)
{
	std::unique_lock<std::mutex> lk(g_ctx_mu);
	if (const size_t freed = release_idle_locked(lk)) {
		fprintf(stderr, "[mpibwa_amd] a device work buffer of %.2f GB did not fit: %.2f GB of idle call contexts' buffers given back\n", wanted / 1e9, freed / 1e9);
		return true;
	}
	int n_busy = 0;
	for (int i = 0; i < MAX_CALLS; ++i) n_busy += g_ctx[i].busy ? 1 : 0;
	if (n_busy <= 1) return false;   // a lone call: nothing to wait for
	if (g_admit > n_busy - 1) {
		g_admit = std::max(1, n_busy - 1);
		fprintf(stderr, "[mpibwa_amd] a device work buffer of %.2f GB does not fit with %d calls in flight: %d calls are admitted at once from now on\n",
		        wanted / 1e9, n_busy, g_admit);
	}
	// wait for another call to end (its context then is idle: its buffers are given back above on the next attempt) — unless every
	// call in flight is waiting here, in which case nobody will ever end
	++g_waiting_for_memory;
	bool ok = true;
	for (;;) {
		if (g_waiting_for_memory >= n_busy) { ok = false; break; }
		g_mem_cv.wait_for(lk, std::chrono::milliseconds(50));
		int now_busy = 0;
		for (int i = 0; i < MAX_CALLS; ++i) now_busy += g_ctx[i].busy ? 1 : 0;
		if (now_busy < n_busy) break;
		n_busy = now_busy;
	}
	--g_waiting_for_memory;
	if (!ok) return false;   // every call in flight waits here: nobody will end (the caller of ensure() reports and ends the process)
	release_idle_locked(lk);
	return true;
}
static thread_local mi355x_stats_t t_stats;   // of the last call made by this thread
static thread_local bool t_stats_set = false;
static std::atomic<int> g_in_flight(0);
int calls_in_flight() { return g_in_flight.load(); }

} // namespace mbw

using namespace mbw;

extern "C" void mi355x_last_stats(mi355x_stats_t *st)
{
	if (t_stats_set) { *st = t_stats; return; }
	std::lock_guard<std::mutex> lk(g_ctx_mu);
	*st = g_stats;
}
extern "C" int mi355x_host_cpus(void) { return usable_cpus(); }
// calls the library runs side by side at most (further callers wait at the door; fewer are admitted when their work buffers do not fit)
extern "C" int mi355x_max_calls(void) { return MAX_CALLS; }
// host threads the library will use for this rank's calls: its share of the node's usable CPUs (the launcher's local size), and
// how many ranks it believes share the node
extern "C" int mi355x_rank_host_threads(int *ranks_on_node)
{
	int ranks = env_local_size();
	if (ranks < 1) ranks = 1;
	if (ranks_on_node) *ranks_on_node = ranks;
	return host_threads(nullptr);
}

// Caller-side helper mirroring mpiBWA's copy_buffer_thr (src/mainParallel.c:103-127): concatenate all
// seqs[i].sam into one malloc'ed buffer and free the per-read strings.
// the same into a buffer the caller keeps from chunk to chunk (*buf, *cap: grown with realloc when a chunk needs more)
extern "C" size_t mi355x_collect_sam_into(bseq1_t *seqs, int n, char **buf, size_t *cap)
{
	std::vector<size_t> off(n + 1);
	const int n_thr = std::min(usable_cpus(), 32);
	parallel_for(n_thr, n, 8192, [&](int i) { off[i + 1] = seqs[i].sam ? strlen(seqs[i].sam) : 0; });
	off[0] = 0;
	for (int i = 0; i < n; ++i) off[i + 1] += off[i];
	const size_t tot = off[n];
	if (tot + 1 > *cap) {
		free(*buf);
		*cap = tot + tot / 8 + 4096;
		*buf = (char *)malloc(*cap);
		if (!*buf) die("out of memory collecting SAM");
	}
	char *b = *buf;
	parallel_for(n_thr, n, 8192, [&](int i) {
		if (!seqs[i].sam) return;
		memcpy(b + off[i], seqs[i].sam, off[i + 1] - off[i]);
		free(seqs[i].sam);
		seqs[i].sam = 0;
	});
	b[tot] = 0;
	return tot;
}

extern "C" char *mi355x_collect_sam(bseq1_t *seqs, int n, size_t *total_len)
{
	std::vector<size_t> off(n + 1);
	const int n_thr = std::min(usable_cpus(), 32);
	parallel_for(n_thr, n, 8192, [&](int i) { off[i + 1] = seqs[i].sam ? strlen(seqs[i].sam) : 0; });
	off[0] = 0;
	for (int i = 0; i < n; ++i) off[i + 1] += off[i];
	const size_t tot = off[n];
	char *buf = (char *)malloc(tot + 1);
	if (!buf) die("out of memory collecting SAM");
	parallel_for(n_thr, n, 8192, [&](int i) {
		if (!seqs[i].sam) return;
		memcpy(buf + off[i], seqs[i].sam, off[i + 1] - off[i]);
		free(seqs[i].sam);
		seqs[i].sam = 0;
	});
	buf[tot] = 0;
	if (total_len) *total_len = tot;
	return buf;
}

extern "C" void mem_process_seqs(const mem_opt_t *opt, const bwt_t *bwt, const bntseq_t *bns, const uint8_t *pac,
                                 int64_t n_processed, int n, bseq1_t *seqs, const mem_pestat_t *pes0)
{
	const double t_begin = now_ms(), c_begin = cpu_sec(), s_begin = sys_sec();
	DevIndex &ix = dev_index();
	std::unique_lock<std::recursive_mutex> init_lk(g_init_mu);
	if (!ix.ready) {
		// first call and nobody called mi355x_init / mi355x_index_upload: make the index resident (one rank per GPU, the GPU
		// named by the launcher's local rank).  Several ranks on the node and no local rank known = every rank would pile
		// its 60 GB onto GPU 0: refuse.
		int lr = env_local_rank();
		if (lr < 0) {
			if (env_local_size() > 1) die("mem_process_seqs: %d ranks share this node but the launcher exports no local rank: call mi355x_init(local_rank, ...) first", env_local_size());
			lr = 0;
		}
		mi355x_index_upload(lr, bwt, bns, pac);
	}
	// the resident index must be the one the caller passes: contig table, pac fetches and coordinates of the host stages
	// come from the caller's copy, seeds and extensions from the device's
	const char *what = nullptr;
	if (!index_matches(bwt, bns, &what))
		die("mem_process_seqs: the index passed in is not the one resident on the GPU (%s differs): mi355x_finalize() and upload it first", what);
	struct InFlight { InFlight() { ++g_in_flight; } ~InFlight() { --g_in_flight; } } in_flight;
	init_lk.unlock();
	HIP_OK(hipSetDevice(ix.device));
	mi355x_stats_t STAT;
	memset(&STAT, 0, sizeof STAT);
	struct Publish {   // the statistics of this call become visible when it returns, whichever way
		mi355x_stats_t &s;
		~Publish() { t_stats = s; t_stats_set = true; std::lock_guard<std::mutex> lk(g_ctx_mu); g_stats = s; }
	} publish{STAT};
	if (n <= 0) return;
	CtxLease lease(seqs, n);
	CallCtx &C = *lease.c;
	static std::once_flag sampler_once;
	std::call_once(sampler_once, [] {
		if (!getenv("MPIBWA_SAMPLE")) return;
		g_sampler = new StageSampler;
		g_sampler->min_calls = std::max(1, atoi(getenv("MPIBWA_SAMPLE")));
		g_sampler->th = std::thread([] { g_sampler->run(); });
		atexit(sampler_report);
	});
	struct StageOwner {
		StageOwner(int slot) { t_stage = &g_stage[slot]; stage(1); }
		~StageOwner() { t_stage->store(0); t_stage = nullptr; }
	} stage_owner((int)(&C - g_ctx));
	const int n_thr = host_threads(opt);
	const bool pe = (opt->flag & MEM_F_PE) != 0;
	if (!C.a_streams[0]) {
		// the SAM stage's kernels (mate rescue, CIGAR) are short and the host waits for them with all its threads: they go
		// ahead of the seeding / extension kernels of the other calls in flight (MPIBWA_PRIO=n: no priorities, p: reversed)
		int lo_p = 0, hi_p = 0;   // numerically lowest = highest priority
		HIP_OK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
		const char *pe = getenv("MPIBWA_PRIO");
		const int pp = pe && *pe == 'p' ? hi_p : 0, pa = !pe || *pe == 'a' ? hi_p : 0;
		// The runtime spreads the streams of one priority over its few hardware queues in the order they are created: without the
		// rotation from call to call the first stream of every call shares ONE hardware queue, and that queue — 50 ms of phase-1
		// kernels and copies per chunk, strictly one after the other — is the bottleneck of eight calls in flight.  Rotated, the
		// kernels of different calls overlap: 13.0-13.7 vs 12.3-12.7 Mreads/s in alternating runs (the seeding launches stretch
		// from 15-18 to 17-21 ms in that company).  MPIBWA_STREAM_ROT=0: no rotation.
		hipStream_t ps[MAX_LANES], hs[4];
		for (int l = 0; l < MAX_LANES; ++l) HIP_OK(hipStreamCreateWithPriority(&ps[l], hipStreamNonBlocking, pp));
		for (int l = 0; l < 4; ++l) HIP_OK(hipStreamCreateWithPriority(&hs[l], hipStreamNonBlocking, pa));
		const char *re = getenv("MPIBWA_STREAM_ROT");
		const int rot = re && atoi(re) == 0 ? 0 : (int)(&C - g_ctx);
		for (int l = 0; l < MAX_LANES; ++l) C.p_streams[l] = ps[(l + rot) % MAX_LANES];
		for (int l = 0; l < 2; ++l) { C.a_streams[l] = hs[(l + rot) % 4]; C.d_streams[l] = hs[(2 + l + rot) % 4]; }
	}
	hipStream_t st = C.p_streams[0];   // never the null stream: another call may be in flight
	Workspace &W = C.gws;

	// ---- 1. encode + pack ----
	stage(29);
	int64_t *off = (int64_t *)W.h_off.ensure((size_t)(n + 1) * 8 + 64);   // 16-byte aligned slot of every read in the packed buffer
	int *lens = (int *)W.h_len.ensure((size_t)n * 4 + 64);
	int max_len = 0;
	{   // lengths and a prefix sum over 667 000 records the caller has just written: by blocks, on all threads
		const int BLK = 8192, nb = (n + BLK - 1) / BLK;
		std::vector<int64_t> bsum(nb + 1, 0);
		std::vector<int> bmax(nb, 0);
		parallel_for(n_thr, nb, 1, [&](int b) {
			const int lo = b * BLK, hi = std::min(n, lo + BLK);
			int64_t sl = 0;
			int m = 0;
			for (int i = lo; i < hi; ++i) {
				const int l = seqs[i].l_seq;
				lens[i] = l; sl += (l + 15) & ~15; m = std::max(m, l);
			}
			bsum[b + 1] = sl; bmax[b] = m;
		});
		for (int b = 0; b < nb; ++b) { bsum[b + 1] += bsum[b]; max_len = std::max(max_len, bmax[b]); }
		parallel_for(n_thr, nb, 1, [&](int b) {
			const int lo = b * BLK, hi = std::min(n, lo + BLK);
			int64_t o = bsum[b];
			for (int i = lo; i < hi; ++i) { off[i] = o; o += (lens[i] + 15) & ~15; }
		});
		off[n] = bsum[nb];
	}
	stage(30);
	const size_t flat_bytes = (size_t)off[n] + 16;
	uint8_t *flat = (uint8_t *)W.h_flat.ensure(flat_bytes);
	if ((size_t)max_len + 2 > 9000) die("read of %d bp exceeds the on-chip band buffers of this build (max 8998 bp)", max_len);
	double t1 = now_ms();
	const double c1 = cpu_sec();
	stage(22);
	uint8_t *d_seq = (uint8_t *)W.seq.ensure(flat_bytes);
	int64_t *d_off = (int64_t *)W.off.ensure((size_t)(n + 1) * 8);
	int *d_len = (int *)W.len.ensure((size_t)n * 4);
	// the bases themselves are encoded and uploaded per sub-batch, on the sub-batch's own stream (phase1 below)
	HIP_OK(hipMemcpyAsync(d_off, off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
	HIP_OK(hipMemcpyAsync(d_len, lens, (size_t)n * 4, hipMemcpyHostToDevice, st));
	HIP_OK(hipStreamSynchronize(st));
	stage(23);
	// contig table for the chaining kernel: start of every contig (+ l_pac) and its ALT flag
	std::vector<int64_t> ann_off(bns->n_seqs + 1);
	std::vector<uint8_t> ann_alt(bns->n_seqs + 1, 0);
	for (int k = 0; k < bns->n_seqs; ++k) { ann_off[k] = bns->anns[k].offset; ann_alt[k] = bns->anns[k].is_alt ? 1 : 0; }
	ann_off[bns->n_seqs] = bns->l_pac;
	int64_t *d_ann_off = (int64_t *)W.ann_off.ensure(ann_off.size() * 8);
	uint8_t *d_ann_alt = (uint8_t *)W.ann_alt.ensure(ann_alt.size());
	HIP_OK(hipMemcpyAsync(d_ann_off, ann_off.data(), ann_off.size() * 8, hipMemcpyHostToDevice, st));
	HIP_OK(hipMemcpyAsync(d_ann_alt, ann_alt.data(), ann_alt.size(), hipMemcpyHostToDevice, st));
	HIP_OK(hipStreamSynchronize(st));

	// ---- inputs of the SAM stage (names, qualities, contig names, the gap table of the CIGAR kernel): packed and uploaded by a
	// thread of their own on a side stream while phase 1 runs — 100 MB of qualities per chunk that nothing before the SAM stage reads
	// (reads so long that one request's arrays do not fit the LDS of a CU — beyond ~1 700 bp — get their CIGARs from the library's host code)
	const bool gpu_aln = getenv("MPIBWA_HOST_CIGAR") == nullptr && aln_lds_per_block(max_len, max_len + 256) <= (size_t)160 * 1024;
	int *d_gap = nullptr;
	bool gpu_sam = false;
	SamDescH *sdesc = nullptr;
	SamParams sam_par;
	const uint8_t *d_qual = nullptr, *d_names = nullptr;
	const int *d_noff = nullptr, *d_ann_noff = nullptr;
	const char *d_ann_names = nullptr;
	struct Joiner {   // (a call that dies on the way out must not leave the thread running on its buffers)
		std::thread t;
		void join() { if (t.joinable()) t.join(); }
		~Joiner() { join(); }
	} sam_inputs;
	sam_inputs.t = std::thread([&] {
		HIP_OK(hipSetDevice(ix.device));
		hipStream_t sst = C.d_streams[1];
		std::vector<int> gaptab(max_len + 2);
		for (int l = 0; l <= max_len + 1; ++l) {   // max_gap of bwa_gen_cigar2 (src/bwa.c:155-158), a function of l_query only
			int max_ins = (int)((double)(((l + 1) >> 1) * opt->mat[0] - opt->o_ins) / opt->e_ins + 1.);
			int max_del = (int)((double)(((l + 1) >> 1) * opt->mat[0] - opt->o_del) / opt->e_del + 1.);
			int g = max_ins > max_del ? max_ins : max_del;
			gaptab[l] = g > 1 ? g : 1;
		}
		d_gap = (int *)W.agap.ensure(gaptab.size() * 4);
		HIP_OK(hipMemcpyAsync(d_gap, gaptab.data(), gaptab.size() * 4, hipMemcpyHostToDevice, sst));
		// ---- SAM text of confidently paired reads on the device (sam_kernel.hip) ----
		// The COLLECT pass describes the two lines of every pair that qualifies (AlnCtx::desc); the kernel runs right behind the
		// CIGAR kernel of the part; the REPLAY pass only copies those records out of the arena and formats the rest itself.
		static_assert(sizeof(SamDesc) == sizeof(SamDescH), "host/device record layouts differ");
		gpu_sam = pe && gpu_aln && getenv("MPIBWA_HOST_SAM") == nullptr && !(opt->flag & (MEM_F_ALL | MEM_F_REF_HDR));
		if (gpu_sam) {
			bool any_q = false, all_q = true;
			for (int i = 0; i < n; ++i) { if (seqs[i].qual) any_q = true; else all_q = false; }
			if (any_q && !all_q) gpu_sam = false;   // a mix of reads with and without qualities: the host formats the chunk
			sam_par.l_pac = bns->l_pac; sam_par.has_qual = any_q ? 1 : 0;
			sam_par.rg_len = (int)strnlen(bwa_rg_id, sizeof bwa_rg_id);
			memset(sam_par.rg, 0, sizeof sam_par.rg);
			memcpy(sam_par.rg, bwa_rg_id, (size_t)sam_par.rg_len);
		}
		if (gpu_sam) {
			sdesc = (SamDescH *)W.h_sdesc.ensure((size_t)n * sizeof(SamDescH) + 64);
			int *noff = (int *)W.h_noff.ensure((size_t)(n + 1) * 4 + 64);
			std::vector<int> nlen(n);
			parallel_for(n_thr, n, 8192, [&](int i) { sdesc[i].req = -1; nlen[i] = (int)strlen(seqs[i].name); });
			noff[0] = 0;
			for (int i = 0; i < n; ++i) noff[i + 1] = noff[i] + nlen[i];
			uint8_t *names = (uint8_t *)W.h_names.ensure((size_t)noff[n] + 64);
			uint8_t *hq = sam_par.has_qual ? (uint8_t *)W.h_qual.ensure(flat_bytes) : nullptr;
			parallel_for(n_thr, n, 4096, [&](int i) {
				memcpy(names + noff[i], seqs[i].name, (size_t)nlen[i]);
				if (hq) memcpy(hq + off[i], seqs[i].qual, (size_t)seqs[i].l_seq);
			});
			uint8_t *dn = (uint8_t *)W.snames.ensure((size_t)noff[n] + 64);
			int *dno = (int *)W.snoff.ensure((size_t)(n + 1) * 4);
			HIP_OK(hipMemcpyAsync(dn, names, (size_t)noff[n], hipMemcpyHostToDevice, sst));
			HIP_OK(hipMemcpyAsync(dno, noff, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, sst));
			if (hq) {
				uint8_t *dq = (uint8_t *)W.squal.ensure(flat_bytes);
				HIP_OK(hipMemcpyAsync(dq, hq, flat_bytes, hipMemcpyHostToDevice, sst));
				d_qual = dq;
			}
			// contig names
			std::vector<int> cno(bns->n_seqs + 1, 0);
			for (int k = 0; k < bns->n_seqs; ++k) cno[k + 1] = cno[k] + (int)strlen(bns->anns[k].name);
			std::vector<char> cn((size_t)cno[bns->n_seqs] + 1);
			for (int k = 0; k < bns->n_seqs; ++k) memcpy(cn.data() + cno[k], bns->anns[k].name, (size_t)(cno[k + 1] - cno[k]));
			char *dcn = (char *)W.sann_names.ensure(cn.size() + 64);
			int *dcno = (int *)W.sann_noff.ensure(cno.size() * 4);
			HIP_OK(hipMemcpyAsync(dcn, cn.data(), cn.size(), hipMemcpyHostToDevice, sst));
			HIP_OK(hipMemcpyAsync(dcno, cno.data(), cno.size() * 4, hipMemcpyHostToDevice, sst));
			stream_wait(sst);
			d_names = dn; d_noff = dno; d_ann_names = dcn; d_ann_noff = dcno;
		}
		stream_wait(sst);
	});

	// ---- 2-6. seeding -> SA -> chaining -> extension -> region clean-up, in sub-batches ----
	// The stages of one sub-batch are strictly dependent (GPU, host, GPU, host), so several sub-batches run on their own host
	// threads with their own HIP stream and workspace: the GPU work of one overlaps the host work of the other.
	stage(24);
	std::vector<HRegV> regs(n);
	// insert-size votes are gathered sub-batch by sub-batch (when they will be needed and can be counted)
	std::vector<uint64_t> pes_hist_v;
	if (pe && !pes0 && pestat_can_count(opt)) pes_hist_v.assign(4 * ((size_t)opt->max_ins + 1), 0);
	uint64_t *pes_hist = pes_hist_v.empty() ? nullptr : pes_hist_v.data();
	// Pairs with one plain hit per end are decided on the device after the insert-size statistics (pair_kernel.hip): every
	// sub-batch leaves the first region and the region count of its reads in chunk-wide arrays.
	const bool dev_pair = pe && getenv("MPIBWA_HOST_PAIR") == nullptr && !(opt->flag & (MEM_F_NOPAIRING | MEM_F_ALL | MEM_F_REF_HDR | MEM_F_PRIMARY5)) &&
	                      opt->mapQ_coef_len > 0;
	DevReg *d_pr_first = dev_pair ? (DevReg *)W.pr_first.ensure((size_t)n * PR_MAXREG * sizeof(DevReg)) : nullptr;
	int *d_pr_nfirst = dev_pair ? (int *)W.pr_nfirst.ensure((size_t)n * 4) : nullptr;
	struct P1 { double k_smem = 0, k_sa = 0, k_ext = 0, smem = 0, sa = 0, chain = 0, ext = 0, regs = 0; uint64_t smem_bytes = 0, smem_tab_bytes = 0, sa_bytes = 0, cells = 0, n_ext = 0, n_intv = 0, n_seeds = 0, n_chains = 0; };
	const int n_all = n;
	static const bool take_turns = !(getenv("MPIBWA_TURNS") && atoi(getenv("MPIBWA_TURNS")) == 0);
	auto phase1 = [&](int lo, int hi, Workspace &W, HostBuf &reg_arena, hipStream_t st, int n_thr, P1 &ps) {
		const int n = hi - lo;
		bseq1_t *seqs_r = seqs + lo;
		const int64_t *d_off_r = d_off + lo;
		const int *d_len_r = d_len + lo;
		HIP_OK(hipSetDevice(ix.device));
		stage(2);
		// nt4-encode this sub-batch in place (the caller sees the codes, src/bwamem.c:1057-1058) and into the staging buffer
		parallel_for(n_thr, n, 4096, [&](int i) {
			char *s = seqs_r[i].seq;
			uint8_t *d = flat + off[lo + i];
			for (int k = 0; k < seqs_r[i].l_seq; ++k) {
				s[k] = s[k] < 4 ? s[k] : (char)nt4_table[(uint8_t)s[k]];
				d[k] = (uint8_t)s[k];
			}
		});
		HIP_OK(hipMemcpyAsync(d_seq + off[lo], flat + off[lo], (size_t)(off[hi] - off[lo]) + (hi == n_all ? 16 : 0), hipMemcpyHostToDevice, st));
		EvTimer ev_smem, ev_sa, ev_ext;
		unsigned long long *d_cnt = (unsigned long long *)W.cnt.ensure(256);
		unsigned long long *cnt = (unsigned long long *)W.h_cnt.ensure(256);
		std::vector<int> gap_h(max_len + 2);
		for (int l = 0; l < max_len + 2; ++l) gap_h[l] = cal_max_gap(opt, l);
		// length tables for the device (the floating-point decisions of the reference, resolved per length on the host)
		const int TS = max_len + 2;
		std::vector<int> tab(6 * TS);
		for (int l = 0; l < TS; ++l) {
			tab[l] = gap_h[l];
			tab[TS + l] = clamp_band(opt, l, 1 << 28, opt->pen_clip5);
			tab[2 * TS + l] = clamp_band(opt, l, 1 << 28, opt->pen_clip3);
			tab[3 * TS + l] = (int)ceil(l * .95);
			tab[4 * TS + l] = (int)floor(.1 * l);
			// mem_flt_chained_seeds returns at once for this length (src/bwamem.c:600-602)
			const double min_l = opt->min_chain_weight ? 1.1f * opt->min_chain_weight : 5.5f * log(l > 0 ? l : 1);
			tab[5 * TS + l] = (l > 0 && min_l > 0.05f * l) ? 1 : 0;
		}
		int *d_tab = (int *)W.tab.ensure(tab.size() * 4);
		HIP_OK(hipMemcpyAsync(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, st));

		double t1 = now_ms();
		uint64_t range_bases = 0;
		for (int i = 0; i < n; ++i) range_bases += lens[lo + i];
		// SMEM seeding (retry with a larger per-read capacity in the rare overflow case)
		int cap = std::max(64, std::min(max_len, 96));
		uint64_t *d_intv; int *d_nintv;
		size_t per_quad = 0;
		int n_quads = smem_grid_quads(max_len, &per_quad);
		void *d_scr = W.scratch.ensure(per_quad * n_quads);
		int *d_nseeds = (int *)W.nseeds.ensure((size_t)n * 4), *d_lrep = (int *)W.lrep.ensure((size_t)n * 4);
		int *nseeds = (int *)W.h_nseeds.ensure((size_t)n * 4 + 8), *lrep = (int *)W.h_lrep.ensure((size_t)n * 4 + 8);
		int *nintv = (int *)W.h_nintv.ensure((size_t)n * 4 + 8);
		// MPIBWA_SMEM_COUNT=1: the seeding kernel also counts the occ blocks the reference would touch (the algorithmic bytes of
		// SURVEY §8d; a property of the reads, so the bench counts every chunk once, outside its timed region)
		const char *ce = getenv("MPIBWA_SMEM_COUNT");
		const bool count_blocks = ce && atoi(ce) != 0;
		for (;;) {
			d_intv = (uint64_t *)W.intv.ensure((size_t)n * cap * 32);
			d_nintv = (int *)W.nintv.ensure((size_t)n * 4);
			HIP_OK(hipMemsetAsync(d_cnt, 0, 256, st));
			// the sub-batches (and the other calls in flight) take turns on the big kernels: each one fills the chip by itself, and running them one
			// after the other staggers the sub-batches so that the host stages of one fall under the kernels of the other
			stage(20);
			std::unique_lock<TurnLock> turn(g_smem_turn, std::defer_lock);
			if (take_turns) turn.lock();
			stage(21);
			ev_smem.start(st);
			launch_smem(st, ix.fm, smem_params(opt), n, d_seq, d_off_r, d_len_r, cap, d_intv, d_nintv, max_len, d_cnt, d_scr, per_quad, n_quads, count_blocks);
			ev_smem.stop(st);
			// seed bookkeeping queued right behind it (src/bwamem.c:265-283): one host round trip for both
			launch_seed_prep(st, n, cap, d_intv, d_nintv, opt->max_occ, d_nseeds, d_lrep);
			HIP_OK(hipMemcpyAsync(cnt, d_cnt, 64, hipMemcpyDeviceToHost, st));
			HIP_OK(hipMemcpyAsync(nseeds, d_nseeds, (size_t)n * 4, hipMemcpyDeviceToHost, st));
			HIP_OK(hipMemcpyAsync(lrep, d_lrep, (size_t)n * 4, hipMemcpyDeviceToHost, st));
			HIP_OK(hipMemcpyAsync(nintv, d_nintv, (size_t)n * 4, hipMemcpyDeviceToHost, st));
			stream_wait(st);
			HIP_OK(hipGetLastError());
			if (take_turns) turn.unlock();
			ps.k_smem += ev_smem.ms();
			if (cnt[2] == 0) break;
			cap *= 4;
		}
		ps.smem_bytes = count_blocks ? cnt[1] * 64 + range_bases : 0;
		ps.smem_tab_bytes = cnt[4] * 64;
		double t2 = now_ms();
		stage(3);

		// seed enumeration + SA lookup (+ chaining on the device)
		int64_t *seed_off = (int64_t *)W.h_seed_off.ensure((size_t)(n + 1) * 8 + 64);
		seed_off[0] = 0;
		uint64_t n_intv = 0;
		for (int i = 0; i < n; ++i) { seed_off[i + 1] = seed_off[i] + nseeds[i]; n_intv += nintv[i]; }
		const int64_t S = seed_off[n];
		if (count_blocks) ps.smem_bytes += n_intv * 32;
		ps.n_intv = n_intv; ps.n_seeds = S;
		uint64_t *sa = (uint64_t *)W.h_sa.ensure((size_t)S * 8 + 8);
		int32_t *qbl = (int32_t *)W.h_qbl.ensure((size_t)S * 8 + 8);
		// Chaining on the device for the reads whose ordered map stays a single B-tree node (chain_kernel.hip); the
		// others (n_chains = -1: ~2 % on 2x150 bp) and, with MPIBWA_HOST_CHAIN=1, all reads are chained by the host below.
		const bool host_chain_all = getenv("MPIBWA_HOST_CHAIN") != nullptr;
		const bool dev_chain = !host_chain_all && S > 0;
		int *nch = nullptr;               // device mode: chains kept per read (-1 = host)
		DevChain *d_chains = nullptr; DevSeed *d_seeds = nullptr; unsigned int *d_srt = nullptr;
		if (S > 0) {
			int64_t *d_seed_off = (int64_t *)W.seed_off.ensure((size_t)(n + 1) * 8);
			uint64_t *d_rows = (uint64_t *)W.rows.ensure((size_t)S * 8), *d_sa = (uint64_t *)W.sa.ensure((size_t)S * 8);
			int32_t *d_qbl = (int32_t *)W.qbl.ensure((size_t)S * 8);
			HIP_OK(hipMemcpyAsync(d_seed_off, seed_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
			launch_seed_enum(st, n, cap, d_intv, d_nintv, opt->max_occ, d_seed_off, d_rows, d_qbl);
			if (S > 0x7fffffff) die("too many seeds in one batch");
			HIP_OK(hipMemsetAsync(d_cnt, 0, 256, st));
			ev_sa.start(st);
			if (ix.fm.sa_full) launch_sa_dense(st, ix.fm, (int)S, d_rows, d_sa);   // one 8-byte load per row
			else launch_sa(st, ix.fm, (int)S, d_rows, d_sa, d_cnt);                // LF walk on the sampled SA
			ev_sa.stop(st);
			if (dev_chain) {   // queued right behind the SA lookup: one host round trip for both
				// (room for the tail the host appends: its reads cannot keep more seeds than the S they had)
				d_chains = (DevChain *)W.chains.ensure((size_t)2 * S * sizeof(DevChain));
				d_seeds = (DevSeed *)W.seeds.ensure((size_t)2 * S * sizeof(DevSeed));
				d_srt = (unsigned int *)W.srt.ensure((size_t)2 * S * 4);
				int *d_nch = (int *)W.nch.ensure((size_t)n * 4);
				ChainParams kp;
				kp.l_pac = bns->l_pac; kp.w = opt->w; kp.max_chain_gap = opt->max_chain_gap; kp.min_chain_weight = opt->min_chain_weight;
				kp.min_seed_len = opt->min_seed_len; kp.max_chain_extend = opt->max_chain_extend;
				kp.mask_level = opt->mask_level; kp.drop_ratio = opt->drop_ratio;
				// third launch (reads with more than 9 chains): room for a sixth of the reads, the rest of them stays with the host
				// (MPIBWA_CHAIN_GENERAL=1: the reads with more than 9 chains through the lane-per-read B-tree kernel of rounds 2-3, which needs
				// 18 KB of scratch per read: room for a sixth of the reads at a time)
				static const bool use_general = getenv("MPIBWA_CHAIN_GENERAL") && atoi(getenv("MPIBWA_CHAIN_GENERAL")) != 0;
				const int gen_cap = use_general ? std::min(n, std::max(4096, n / 6)) : std::min(n, 64);
				void *d_gen = W.chain_gen.ensure(chain_general_bytes(gen_cap, n));
				launch_chain(st, kp, n, d_len_r, d_nseeds, d_lrep, d_seed_off, d_sa, d_qbl, d_ann_off, d_ann_alt, bns->n_seqs, d_tab, TS, d_chains, d_seeds,
				             d_srt, d_nch, d_gen, gen_cap);
				nch = (int *)W.h_nch.ensure((size_t)n * 4 + 8);
				HIP_OK(hipMemcpyAsync(nch, d_nch, (size_t)n * 4, hipMemcpyDeviceToHost, st));
			}
			HIP_OK(hipMemcpyAsync(cnt, d_cnt, 64, hipMemcpyDeviceToHost, st));
			// The seeds themselves (16 bytes each, 124 MB per chunk of 2x150 bp) only come back for the reads the host chains: all of
			// them in host mode; in device mode the few reads chain_kernel declined, as a handful of spans once their list is known
			// (one short round trip more) — or everything again when those reads are many (repeat-rich references).
			bool seeds_fetched = false;
			if (!dev_chain) {
				HIP_OK(hipMemcpyAsync(sa, d_sa, (size_t)S * 8, hipMemcpyDeviceToHost, st));
				HIP_OK(hipMemcpyAsync(qbl, d_qbl, (size_t)S * 8, hipMemcpyDeviceToHost, st));
				seeds_fetched = true;
			}
			stream_wait(st);
			HIP_OK(hipGetLastError());
			if (!seeds_fetched) {
				const int64_t GAP = 1 << 15;          // spans closer than this many seeds travel as one
				std::vector<std::pair<int64_t, int64_t>> span;
				int64_t covered = 0;
				for (int i = 0; i < n; ++i) {
					if (nch[i] >= 0 || nseeds[i] == 0) continue;
					if (!span.empty() && seed_off[i] - span.back().second <= GAP) span.back().second = seed_off[i + 1];
					else span.emplace_back(seed_off[i], seed_off[i + 1]);
				}
				for (auto &sp : span) covered += sp.second - sp.first;
				if (span.size() > 256 || covered > S / 2) { span.clear(); span.emplace_back(0, S); }
				for (auto &sp : span) {
					HIP_OK(hipMemcpyAsync(sa + sp.first, d_sa + sp.first, (size_t)(sp.second - sp.first) * 8, hipMemcpyDeviceToHost, st));
					HIP_OK(hipMemcpyAsync(qbl + 2 * sp.first, d_qbl + 2 * sp.first, (size_t)(sp.second - sp.first) * 8, hipMemcpyDeviceToHost, st));
				}
				if (!span.empty()) stream_wait(st);
			}
			ps.k_sa = ev_sa.ms();
			ps.sa_bytes = ix.fm.sa_full ? (uint64_t)S * 16 : cnt[1] * 64 + (uint64_t)S * 8;
		}
		double t3 = now_ms();
		stage(4);

		// chaining and chain filters (host).  Each block of reads is chained by one thread with recycled scratch and packed
		// straight into the device layout (block-local offsets); the blocks are then concatenated after a prefix sum.
		std::vector<int> todo;            // reads chained by the host
		if (dev_chain) {
			for (int i = 0; i < n; ++i)
				if (nch[i] < 0) todo.push_back(i);
		} else {
			todo.resize(n);
			for (int i = 0; i < n; ++i) todo[i] = i;
		}
		const int n_todo = (int)todo.size();
		const int CB = 256, n_cb = (n_todo + CB - 1) / CB;
		struct BlockOut { std::vector<DevChain> ch; std::vector<DevSeed> sd; std::vector<unsigned int> srt; };
		std::vector<BlockOut> bo(n_cb);
		std::vector<int> chain_off(n_todo + 1), reg_off(n_todo + 1);   // per entry of `todo`
		{
			const int nt = std::max(1, n_thr);
			std::vector<std::unique_ptr<ChainScratch>> scr(nt);
			std::vector<std::vector<HSeed>> hsv(nt);
			std::vector<std::vector<HChain *>> chv(nt);
			std::vector<std::vector<uint64_t>> keyv(nt);
			static const bool prof_chain = getenv("MPIBWA_CPUSEC") != nullptr;
			std::vector<unsigned long long> tsc((size_t)nt * 8, 0);
			parallel_blocks(nt, n_todo, CB, [&](int tid, int b, int lo, int hi) {
				if (!scr[tid]) scr[tid].reset(new ChainScratch());
				std::vector<HSeed> &hs = hsv[tid];
				std::vector<HChain *> &chains = chv[tid];
				std::vector<uint64_t> &key = keyv[tid];
				BlockOut &o = bo[b];
				int64_t est = 0;
				for (int t = lo; t < hi; ++t) est += nseeds[todo[t]];
				o.sd.reserve(est); o.srt.reserve(est); o.ch.reserve((hi - lo) * 2);
				for (int t = lo; t < hi; ++t) {
					const int i = todo[t];
					int ns = nseeds[i];
					chain_off[t + 1] = reg_off[t + 1] = 0;
					if (ns == 0) continue;
					const unsigned long long c0 = prof_chain ? __builtin_ia32_rdtsc() : 0;
					hs.resize(ns);
					for (int k = 0; k < ns; ++k) {
						int64_t so = seed_off[i] + k;
						hs[k].rbeg = (int64_t)sa[so]; hs[k].qbeg = qbl[2 * so]; hs[k].len = hs[k].score = qbl[2 * so + 1];
					}
					const unsigned long long c1 = prof_chain ? __builtin_ia32_rdtsc() : 0;
					chains_from_seeds(opt, bns, seqs_r[i].l_seq, hs.data(), ns, lrep[i], *scr[tid], chains);
					const unsigned long long c2 = prof_chain ? __builtin_ia32_rdtsc() : 0;
					chain_filter(opt, *scr[tid], chains);
					const unsigned long long c3 = prof_chain ? __builtin_ia32_rdtsc() : 0;
					filter_chained_seeds(opt, bns, pac, seqs_r[i].l_seq, (const uint8_t *)seqs_r[i].seq, chains);
					const unsigned long long c4 = prof_chain ? __builtin_ia32_rdtsc() : 0;
					if (prof_chain) { tsc[tid * 8 + 0] += c1 - c0; tsc[tid * 8 + 1] += c2 - c1; tsc[tid * 8 + 2] += c3 - c2; tsc[tid * 8 + 3] += c4 - c3; tsc[tid * 8 + 5] += ns; tsc[tid * 8 + 6] += ns > 64; }
					int tot = 0;
					for (const HChain *cp_ : chains) {
						const HChain &ch = *cp_;
						const int cs = (int)ch.seeds.size();
						DevChain d;
						const size_t at = o.sd.size();
						o.sd.resize(at + cs); o.srt.resize(at + cs);
						pack_chain_for_device(bns, ch, seqs_r[i].l_seq, gap_h.data(), key, d, o.sd.data() + at);
						d.seed_beg = (int)at;   // block-local for now
						for (int k = 0; k < cs; ++k) o.srt[at + k] = (unsigned int)k;   // the order array only carries "skipped" marks
						o.ch.push_back(d);
						tot += cs;
					}
					chain_off[t + 1] = (int)chains.size();
					reg_off[t + 1] = tot;
					if (prof_chain) tsc[tid * 8 + 4] += __builtin_ia32_rdtsc() - c4;
				}
			});
			if (prof_chain) {
				unsigned long long t[8] = {0};
				for (int a = 0; a < nt; ++a) for (int b = 0; b < 8; ++b) t[b] += tsc[(size_t)a * 8 + b];
				fprintf(stderr, "[chain Mcycles] seeds->HSeed %.0f  chaining %.0f  filter %.0f  flt_seeds %.0f  pack %.0f   (%llu seeds, %llu reads with >64 seeds, %d reads)\n",
				        t[0] * 1e-6, t[1] * 1e-6, t[2] * 1e-6, t[3] * 1e-6, t[4] * 1e-6, t[5], t[6], n);
			}
		}
		if (getenv("MPIBWA_CHAIN_HIST")) {   // which reads the host chained: seeds in, chains out (log2 buckets)
			unsigned long long hs[20] = {0}, hc[20] = {0}, ss[20] = {0};
			for (int t = 0; t < n_todo; ++t) {
				int b = 0, c = 0;
				while ((1 << (b + 1)) <= nseeds[todo[t]] && b < 19) ++b;
				while ((1 << (c + 1)) <= chain_off[t + 1] && c < 19) ++c;
				++hs[b]; ss[b] += nseeds[todo[t]]; ++hc[c];
			}
			fprintf(stderr, "[chain hist] %d host-chained reads; by seeds (2^b..): ", n_todo);
			for (int b = 0; b < 20; ++b) if (hs[b]) fprintf(stderr, " %d:%llu(%llu)", b, hs[b], ss[b]);
			fprintf(stderr, "; by kept chains: ");
			for (int b = 0; b < 20; ++b) if (hc[b]) fprintf(stderr, " %d:%llu", b, hc[b]);
			fprintf(stderr, "\n");
		}
		chain_off[0] = reg_off[0] = 0;
		for (int t = 0; t < n_todo; ++t) { chain_off[t + 1] += chain_off[t]; reg_off[t + 1] += reg_off[t]; }
		const int NC = chain_off[n_todo], NS = reg_off[n_todo];   // chains / kept seeds of the host-chained reads
		// Device layout.  Host mode: dense arrays.  Device mode: read r owns slots seed_off[r].. of all three arrays, and what
		// the host chained is appended behind the S seed slots.
		const int64_t base = dev_chain ? S : 0;
		DevChain *hchains = (DevChain *)W.h_chains.ensure((size_t)NC * sizeof(DevChain) + 8);
		DevSeed *hseeds = (DevSeed *)W.h_seeds.ensure((size_t)NS * sizeof(DevSeed) + 8);
		unsigned int *hsrt = (unsigned int *)W.h_srt.ensure((size_t)NS * 4 + 8);
		parallel_blocks(n_thr, n_todo, CB, [&](int, int b, int lo, int) {
			BlockOut &o = bo[b];
			const int c0 = chain_off[lo], s0 = reg_off[lo];
			for (size_t c = 0; c < o.ch.size(); ++c) { hchains[c0 + c] = o.ch[c]; hchains[c0 + c].seed_beg += (int)(base + s0); }
			if (!o.sd.empty()) {
				memcpy((void *)(hseeds + s0), (const void *)o.sd.data(), o.sd.size() * sizeof(DevSeed));
				memcpy(hsrt + s0, o.srt.data(), o.srt.size() * 4);
			}
			BlockOut().ch.swap(o.ch); std::vector<DevSeed>().swap(o.sd); std::vector<unsigned int>().swap(o.srt);
		});
		if (base + NS > 0x7fffffff || base + NC > 0x7fffffff) die("too many seeds in one batch");
		int *chain_beg = (int *)W.h_cbeg.ensure((size_t)n * 4 + 8), *chain_cnt = (int *)W.h_ccnt.ensure((size_t)n * 4 + 8);
		int *reg_beg = (int *)W.h_rbeg.ensure((size_t)n * 4 + 8);
		uint64_t n_chains_total = NC;
		if (dev_chain) {
			for (int i = 0; i < n; ++i) { chain_beg[i] = reg_beg[i] = (int)seed_off[i]; chain_cnt[i] = nch[i] > 0 ? nch[i] : 0; n_chains_total += chain_cnt[i]; }
		} else memset(chain_cnt, 0, (size_t)n * 4);
		for (int t = 0; t < n_todo; ++t) {
			const int i = todo[t];
			chain_beg[i] = (int)(base + chain_off[t]); chain_cnt[i] = chain_off[t + 1] - chain_off[t]; reg_beg[i] = (int)(base + reg_off[t]);
		}
		ps.n_chains = n_chains_total;
		const int64_t n_slots = base + NS;    // size of the seed / order / region arrays on the device
		double t4 = now_ms();
		stage(5);

		// chain -> regions on the GPU
		int *nregs = (int *)W.h_nregs.ensure((size_t)n * 4 + 8);
		std::vector<int> reg_pos(n + 1, 0);   // where the regions of read i start in hregs
		DevReg *hregs = nullptr;
		if (n_slots == 0) {
			memset(nregs, 0, (size_t)n * 4);
			// no read of the sub-batch has a seed: first_reg_kernel does not run, so the pairing kernel's slice of region counts must
			// be cleared here (it would otherwise read the previous chunk's, or whatever hipMalloc left there)
			if (d_pr_nfirst) { HIP_OK(hipMemsetAsync(d_pr_nfirst + lo, 0, (size_t)n * 4, st)); stream_wait(st); }
		} else {
			int *d_chain_beg = (int *)W.chain_off.ensure((size_t)n * 4), *d_chain_cnt = (int *)W.chain_cnt.ensure((size_t)n * 4);
			int *d_reg_beg = (int *)W.reg_off.ensure((size_t)n * 4);
			// (device mode: already sized 2 S above, so these calls never move what chain_kernel wrote)
			d_chains = (DevChain *)W.chains.ensure((size_t)std::max<int64_t>(base + NC, 1) * sizeof(DevChain));
			d_seeds = (DevSeed *)W.seeds.ensure((size_t)n_slots * sizeof(DevSeed));
			d_srt = (unsigned int *)W.srt.ensure((size_t)n_slots * 4);
			DevReg *d_regs = (DevReg *)W.regs.ensure((size_t)n_slots * sizeof(DevReg));
			int *d_nregs = (int *)W.nregs.ensure((size_t)(n + 1) * 4);
			HIP_OK(hipMemcpyAsync(d_chain_beg, chain_beg, (size_t)n * 4, hipMemcpyHostToDevice, st));
			HIP_OK(hipMemcpyAsync(d_chain_cnt, chain_cnt, (size_t)n * 4, hipMemcpyHostToDevice, st));
			HIP_OK(hipMemcpyAsync(d_reg_beg, reg_beg, (size_t)n * 4, hipMemcpyHostToDevice, st));
			if (NC) HIP_OK(hipMemcpyAsync(d_chains + base, hchains, (size_t)NC * sizeof(DevChain), hipMemcpyHostToDevice, st));
			if (NS) HIP_OK(hipMemcpyAsync(d_seeds + base, hseeds, (size_t)NS * sizeof(DevSeed), hipMemcpyHostToDevice, st));
			if (NS) HIP_OK(hipMemcpyAsync(d_srt + base, hsrt, (size_t)NS * 4, hipMemcpyHostToDevice, st));
			unsigned long long *d_c2a_stat = (unsigned long long *)W.c2a_stat.ensure(C2A_STAT_SLOTS * 64);
			HIP_OK(hipMemsetAsync(d_c2a_stat, 0, C2A_STAT_SLOTS * 64, st));
			// launch order: reads by decreasing number of seeds (counting sort), the long-running ones first
			int *order = (int *)W.h_order.ensure((size_t)n * 4 + 8);
			{
				const int NB = 1024;
				std::vector<int> start(NB + 1, 0);
				for (int i = 0; i < n; ++i) ++start[NB - 1 - std::min(nseeds[i], NB - 1) + 1];
				for (int b = 0; b < NB; ++b) start[b + 1] += start[b];
				for (int i = 0; i < n; ++i) order[start[NB - 1 - std::min(nseeds[i], NB - 1)]++] = i;
			}
			int *d_order = (int *)W.order.ensure((size_t)n * 4);
			HIP_OK(hipMemcpyAsync(d_order, order, (size_t)n * 4, hipMemcpyHostToDevice, st));
			// Reads with more than a handful of chains (high-copy repeats: hundreds of chains at hundreds of loci) are not walked by one
			// wavefront: their chains are split into groups that cannot see each other's regions (c2a_groups.hip), a unit of c2a_kernel each.
			// MPIBWA_C2A_HEAVY=<chains> moves the threshold (0: every read is walked by one wavefront, as before round 4).
			static const int heavy_t = getenv("MPIBWA_C2A_HEAVY") ? atoi(getenv("MPIBWA_C2A_HEAVY")) : 8;
			C2aUnits units;
			if (heavy_t > 0) {
				int n_heavy = 0;
				int64_t n_el = 0;
				for (int i = 0; i < n; ++i)
					if (chain_cnt[i] > heavy_t) { ++n_heavy; n_el += chain_cnt[i]; }
				if (n_heavy > 0 && n_el < 0x7fffffff) {
					int *hv = (int *)W.h_heavy.ensure((size_t)n_heavy * 4 + 64), *ho = (int *)W.h_hoff.ensure((size_t)(n_heavy + 1) * 4 + 64);
					int k = 0, tot = 0;
					for (int i = 0; i < n; ++i)
						if (chain_cnt[i] > heavy_t) { hv[k] = i; ho[k] = tot; tot += chain_cnt[i]; ++k; }
					ho[k] = tot;
					int *d_hv = (int *)W.heavy.ensure((size_t)n_heavy * 4), *d_ho = (int *)W.hoff.ensure((size_t)(n_heavy + 1) * 4);
					void *d_gs = W.grp_scratch.ensure(c2a_groups_scratch_bytes((int)n_el));
					int *d_clist = (int *)W.grp_clist.ensure((size_t)n_el * 4), *d_ustart = (int *)W.grp_ustart.ensure((size_t)(n_el + 1) * 4);
					int *d_urd = (int *)W.grp_unit_rd.ensure((size_t)n_el * 4), *d_uav = (int *)W.grp_unit_av.ensure((size_t)n_el * 4);
					unsigned int *d_nu = (unsigned int *)W.grp_nunits.ensure(64);
					const size_t n_chain_slots = (size_t)std::max<int64_t>(base + NC, 1);
					units.c_rabs = (int *)W.c_rabs.ensure(n_chain_slots * 4);
					units.c_rcnt = (int *)W.c_rcnt.ensure(n_chain_slots * 4);
					HIP_OK(hipMemcpyAsync(d_hv, hv, (size_t)n_heavy * 4, hipMemcpyHostToDevice, st));
					HIP_OK(hipMemcpyAsync(d_ho, ho, (size_t)(n_heavy + 1) * 4, hipMemcpyHostToDevice, st));
					launch_c2a_groups(st, (int)n_el, n_heavy, d_ho, d_hv, d_chain_beg, d_reg_beg, d_chains, d_gs, d_clist, d_ustart, d_urd, d_uav, d_nu);
					// (the number of units sizes the launch: a grid padded to the number of chains would be millions of empty workgroups)
					unsigned int *nu = (unsigned int *)W.h_nunits.ensure(64);
					HIP_OK(hipMemcpyAsync(nu, d_nu, 4, hipMemcpyDeviceToHost, st));
					stream_wait(st);
					HIP_OK(hipGetLastError());
					units.max_units = (int)nu[0]; units.heavy_t = heavy_t; units.n_units = d_nu; units.ustart = d_ustart;
					units.unit_rd = d_urd; units.unit_av = d_uav; units.clist = d_clist;
					HIP_OK(hipMemsetAsync(d_nregs, 0, (size_t)(n + 1) * 4, st));   // the units of a read add their regions up
				}
			}
			C2aParams cp;
			cp.l_pac = bns->l_pac; cp.a = opt->a; cp.w = opt->w; cp.pen_clip5 = opt->pen_clip5; cp.pen_clip3 = opt->pen_clip3;
			cp.early = c2a_early_mode();
			ExtParams ep;
			memcpy(ep.mat, opt->mat, 25);
			ep.o_del = opt->o_del; ep.e_del = opt->e_del; ep.o_ins = opt->o_ins; ep.e_ins = opt->e_ins; ep.zdrop = opt->zdrop;
			stage(50);
			std::unique_lock<TurnLock> turn(g_c2a_turn, std::defer_lock);
			if (take_turns) turn.lock();
			stage(51);
			ev_ext.start(st);
			// one wavefront per read (any read length)
			launch_c2a(st, cp, ep, n, d_seq, d_off_r, d_len_r, d_chain_beg, d_chain_cnt, d_chains, d_seeds, d_srt, d_reg_beg, d_regs, d_nregs,
			           d_tab, TS, (const uint8_t *)ix.d_pac, d_c2a_stat, max_len, d_order, units.max_units > 0 ? &units : nullptr);
			ev_ext.stop(st);
			// the regions sit in sparse per-read slots: prefix-sum + pack on the device, queued behind the kernel, then one
			// copy of what is usually enough (2 regions per read); the rare rest follows once the total is known
			int *d_reg_pos = (int *)W.reg_pos.ensure((size_t)(n + 1) * 4);
			const int64_t guess = std::min<int64_t>(n_slots, (int64_t)2 * n + 1024);
			DevReg *d_packed = (DevReg *)W.regs_packed.ensure((size_t)n_slots * sizeof(DevReg));
			const size_t tmp_bytes = reg_pack_tmp_bytes(n);
			void *d_tmp = W.pack_tmp.ensure(tmp_bytes);
			launch_reg_pack(st, n, d_reg_beg, d_nregs, d_reg_pos, d_regs, d_packed, d_tmp, tmp_bytes, units.max_units > 0 ? &units : nullptr, d_chain_beg, d_chain_cnt);
			if (d_pr_first) launch_first_reg(st, n, d_reg_pos, d_nregs, d_packed, d_pr_first + (size_t)lo * PR_MAXREG, d_pr_nfirst + lo);
			hregs = (DevReg *)W.h_regs.ensure((size_t)guess * sizeof(DevReg) + 8);
			unsigned long long *stat_h = (unsigned long long *)W.h_c2a_stat.ensure(C2A_STAT_SLOTS * 64);
			HIP_OK(hipMemcpyAsync(stat_h, d_c2a_stat, C2A_STAT_SLOTS * 64, hipMemcpyDeviceToHost, st));
			HIP_OK(hipMemcpyAsync(nregs, d_nregs, (size_t)n * 4, hipMemcpyDeviceToHost, st));
			HIP_OK(hipMemcpyAsync(hregs, d_packed, (size_t)guess * sizeof(DevReg), hipMemcpyDeviceToHost, st));
			stream_wait(st);
			HIP_OK(hipGetLastError());
			if (take_turns) turn.unlock();
			ps.k_ext = ev_ext.ms();
			for (int k = 0; k < 4; ++k) { cnt[k] = 0; for (int sl = 0; sl < C2A_STAT_SLOTS; ++sl) cnt[k] += stat_h[sl * 8 + k]; }
			ps.cells = cnt[0]; ps.n_ext = cnt[1];
			if (cp.early == 2 && cnt[3]) die("c2a_kernel: %llu of %llu extensions change when their row loops stop early", cnt[3], cnt[1]);
			if (getenv("MPIBWA_CPUSEC")) fprintf(stderr, "[c2a] %llu extensions, %llu without DP, %llu cells, kernel %.2f ms\n", cnt[1], cnt[2], cnt[0], ev_ext.ms());
			for (int i = 0; i < n; ++i) reg_pos[i + 1] = reg_pos[i] + nregs[i];
			const int64_t NR = reg_pos[n];
			if (NR > guess) {
				DevReg *all = (DevReg *)W.h_regs2.ensure((size_t)NR * sizeof(DevReg) + 8);
				HIP_OK(hipMemcpyAsync(all, d_packed, (size_t)NR * sizeof(DevReg), hipMemcpyDeviceToHost, st));
				stream_wait(st);
				hregs = all;
			}
		}
		double t5 = now_ms();
		stage(6);

		// region post-processing (host); every read gets a slice of the batch-wide arena: its regions + room for rescued mates
		const int SLACK = 4;
		std::vector<int64_t> slice(n + 1);
		slice[0] = 0;
		for (int i = 0; i < n; ++i) slice[i + 1] = slice[i] + nregs[i] + SLACK;
		HReg *arena = (HReg *)reg_arena.ensure((size_t)slice[n] * sizeof(HReg));
		parallel_for(n_thr, n, 256, [&](int i) {
			HRegV &v = regs[lo + i];
			int m = nregs[i];
			v.attach(arena + slice[i], (uint32_t)(m + SLACK));
			v.resize(m);
			for (int k = 0; k < m; ++k) {
				const DevReg &d = hregs[reg_pos[i] + k];
				HReg &r = v[k];
				r.rb = d.rb; r.re = d.re; r.qb = d.qb; r.qe = d.qe; r.rid = d.rid; r.score = d.score; r.truesc = d.truesc;
				r.w = d.w; r.seedcov = d.seedcov; r.seedlen0 = d.seedlen0; r.frac_rep = d.frac_rep;
			}
			sort_dedup_patch(opt, bns, pac, (uint8_t *)seqs_r[i].seq, v);
			for (HReg &r : v)
				if (r.rid >= 0 && bns->anns[r.rid].is_alt) r.is_alt = 1;
		});
		// insert-size votes of this sub-batch (src/bwamem_pair.c:52-63), so that the barrier only has to add histograms up
		if (pes_hist) {
			const int plo = lo >> 1, np_ = n >> 1, nt = std::max(1, std::min(n_thr, np_ / 4096));
			const size_t hsz = 4 * ((size_t)opt->max_ins + 1);
			std::vector<std::vector<uint64_t>> part(nt);
			parallel_blocks(nt, nt, 1, [&](int, int b, int, int) {
				part[b].assign(hsz, 0);
				pestat_gather(opt, bns->l_pac, plo + (int)((int64_t)np_ * b / nt), plo + (int)((int64_t)np_ * (b + 1) / nt), regs.data(), part[b].data());
			});
			std::lock_guard<std::mutex> g(g_pes_lock);
			for (int b = 0; b < nt; ++b)
				for (size_t v = 0; v < hsz; ++v) pes_hist[v] += part[b][v];
		}
		double t6 = now_ms();
		ps.smem = t2 - t1; ps.sa = t3 - t2; ps.chain = t4 - t3; ps.ext = t5 - t4; ps.regs = t6 - t5;
	};

	// K sub-batches are worked off by up to MAX_LANES host threads ("lanes"), each with its own HIP stream and workspace
	// Sub-batches overlap the GPU and host stages of ONE call.  When enough other calls are in flight they provide that
	// overlap, and one launch per kernel over the whole chunk is cheaper than three (one tail instead of three: the SMEM
	// kernel needs 23 ms for the chunk in one launch, 3 x 10 ms in three).
	// (two sub-batches since round 3: with the pairing decisions on the device the host stages of a sub-batch are short, and a
	// third sub-batch only adds a third tail to every big kernel: 93.5 vs 103-105 ms per chunk with one call in flight)
	int n_sub = lease.crowded ? 1 : 2, n_lanes = 2;
	if (const char *e = getenv("MPIBWA_SUBBATCH")) n_sub = atoi(e);
	if (const char *e = getenv("MPIBWA_LANES")) n_lanes = atoi(e);
	n_sub = std::max(1, std::min(n_sub, 16));
	n_lanes = std::max(1, std::min(n_lanes, std::min(n_sub, MAX_LANES)));
	int min_sub = 40000;   // below this a chunk is not worth splitting
	if (const char *e = getenv("MPIBWA_SUBBATCH_MIN")) min_sub = atoi(e);
	if (n < min_sub) n_sub = n_lanes = 1;
	std::vector<P1> ps(n_sub);
	if (n_sub == 1) phase1(0, n, C.ws[0], C.reg_arena[0], st, n_thr, ps[0]);
	else {
		hipStream_t *s_streams = C.p_streams;
		std::vector<int> cut(n_sub + 1);
		for (int k = 0; k <= n_sub; ++k) cut[k] = (int)((int64_t)n * k / n_sub) & ~1;   // keep mates together
		cut[n_sub] = n;
		// every lane may use all host threads: while one lane waits for a kernel the other one gets the whole CPU share
		int thr_each = n_thr;
		if (const char *e = getenv("MPIBWA_P1_THREADS")) thr_each = std::max(1, atoi(e));
		std::atomic<int> next(0);
		auto lane = [&](int l) {
			for (;;) {
				int k = next.fetch_add(1);
				if (k >= n_sub) break;
				phase1(cut[k], cut[k + 1], C.ws[l], C.reg_arena[k], s_streams[l], thr_each, ps[k]);
			}
		};
		std::vector<std::thread> th;
		for (int l = 1; l < n_lanes; ++l) th.emplace_back(lane, l);
		lane(0);
		for (auto &t : th) t.join();
	}
	for (int k = 0; k < n_sub; ++k) {
		STAT.k_smem_ms += ps[k].k_smem; STAT.k_sa_ms += ps[k].k_sa; STAT.k_ext_ms += ps[k].k_ext;
		STAT.smem_bytes += ps[k].smem_bytes; STAT.smem_tab_bytes += ps[k].smem_tab_bytes; STAT.sa_bytes += ps[k].sa_bytes; STAT.ext_cells += ps[k].cells; STAT.n_ext += ps[k].n_ext;
		STAT.n_intv += ps[k].n_intv; STAT.n_seeds += ps[k].n_seeds; STAT.n_chains += ps[k].n_chains;
		// per-stage wall times: the sub-batches of a lane run back to back and the lanes side by side, so sum / lanes
		STAT.smem_ms += ps[k].smem / n_lanes; STAT.sa_ms += ps[k].sa / n_lanes;
		STAT.chain_ms += ps[k].chain / n_lanes; STAT.ext_ms += ps[k].ext / n_lanes;
		STAT.regs_ms += ps[k].regs / n_lanes;
	}
	double t6 = now_ms();
	STAT.phase1_ms = t6 - t1;
	STAT.n_sub = n_sub;
	const double c6 = cpu_sec();

	// ---- 7. insert-size statistics over the whole batch ----
	stage(7);
	mem_pestat_t pes[4];
	if (pe) {
		if (pes0) memcpy(pes, pes0, 4 * sizeof(mem_pestat_t));
		else if (pes_hist) pestat_from_hist(opt, pes_hist, pes);
		else pestat(opt, bns->l_pac, n, regs.data(), pes, n_thr);
	}
	double t7 = now_ms();

	// ---- 8. pairing decisions, then CIGAR/MD/NM on the GPU, then SAM text ----
	stage(8);
	// Per part of the chunk:  A  decisions + a COLLECT pass that records which regions need a global re-alignment
	// (mem_reg2aln's DP);  B  aln_kernel does them all at once;  C  the same emission again (REPLAY) with the results
	// plugged in.  Two parts are software-pipelined so that B of one part runs while the host does A / C of the other.
	const int n_units = pe ? n >> 1 : n;
	std::vector<PairPlan> plans(pe ? n_units : 0);
	int n_parts = (gpu_aln && n_units >= 20000 && n_sub > 1) ? 2 : 1;
	if (const char *e = getenv("MPIBWA_SAM_PARTS")) n_parts = std::max(1, std::min(2, atoi(e)));
	struct Part {
		int lo = 0, hi = 0;
		AlnReqH *req = nullptr;           // CIGAR requests of the part, in a page-locked buffer of the call context
		size_t n_req = 0;
		std::vector<uint32_t> base;       // first request of every unit of the part
		AlnHdrH *hdr = nullptr;           // results, in page-locked staging buffers
		uint8_t *pool = nullptr;
		int slot = 0;
		// the pairs decided on the device: their two CIGAR requests each (pair_kernel's array, in place) and their records
		struct DevJob {
			bool launched = false;
			hipStream_t st = 0;
			size_t n_req = 0, pool_bytes = 0, arena_bytes = 0;
			AlnHdr *d_hdr = nullptr; uint8_t *d_pool = nullptr; unsigned long long *d_cnt = nullptr;
			EvTimer ev;
			const AlnHdrH *hdr = nullptr; const uint8_t *pool = nullptr;   // host copies: only fetched when the device hands a record back
			const uint8_t *sarena = nullptr; const unsigned long long *sooff = nullptr; const int *solen = nullptr;
		} dj;
		unsigned long long cnt[8] = {0};
		// mate-rescue alignments of the part: requests of unit k are mreq[mbase[k] .. mbase[k+1])
		MswReqH *mreq = nullptr; MswResH *mres = nullptr;
		std::vector<uint32_t> mbase;
		size_t n_mreq = 0;
		EvTimer mev;
		bool m_launched = false;
		// CIGAR requests while they are being listed: per block of 256 units, and where each unit's run starts
		std::vector<std::vector<AlnReqH>> blk_req;
		std::vector<uint32_t> u_first, u_cnt;
		AlnHdr *d_hdr = nullptr; uint8_t *d_pool = nullptr; unsigned long long *d_cnt = nullptr;
		size_t pool_bytes = 0;
		hipStream_t st = 0;
		EvTimer ev;
		// records written by sam_kernel: arena / offsets / lengths (host copies), for the reads 2 lo .. 2 hi of the part
		bool sam_launched = false;
		size_t arena_bytes = 0;
		const uint8_t *sarena = nullptr;
		const unsigned long long *sooff = nullptr;
		const int *solen = nullptr;
	};
	Part parts[2];
	Workspace &WS = C.gws;   // the per-part buffers of the SAM stage (slot 0: the whole chunk or its first half)
	hipStream_t *a_streams = C.a_streams;
	double plan_ms = 0, aln_wait_ms = 0;

	stage(25);
	sam_inputs.join();

	// ---- pairs with one plain hit per end: decided on the device (pair_kernel.hip) ----
	// status[k] = 1: the pair's two CIGAR requests and line descriptors exist on the device; the host neither lists rescue
	// alignments nor plans nor formats it (it only copies the two finished records out, or takes the pair back if the device
	// hands a record back).
	const uint8_t *pstat = nullptr;
	const AlnReq *d_pr_req = nullptr;
	const SamDesc *d_pr_desc = nullptr;
	double pair_dev_ms = 0;
	if (dev_pair && gpu_sam && gpu_aln) {
		const double tp0 = now_ms();
		stage(9);
		PairParams pp;
		size_t n_tab = 0;
		const bool usable = pair_params(opt, bns->l_pac, pes, n_processed, max_len, pp, &n_tab);
		if (usable) {
			double *tab = (double *)W.h_pr_tab.ensure((n_tab + (size_t)pp.ltab_n) * 8 + 64);
			pair_tables(opt, pes, pp, n_tab, tab);
			stage(28);
			uint8_t *ok = (uint8_t *)W.h_pr_ok.ensure((size_t)n_units + 64);
			parallel_for(n_thr, n_units, 8192, [&](int k) {
				ok[k] = !seqs[2 * k].comment && !seqs[2 * k + 1].comment && strcmp(seqs[2 * k].name, seqs[2 * k + 1].name) == 0;
			});
			double *d_tab = (double *)W.pr_ptab.ensure((n_tab + (size_t)pp.ltab_n) * 8 + 64);
			uint8_t *d_ok = (uint8_t *)W.pr_ok.ensure((size_t)n_units + 64);
			uint8_t *d_status = (uint8_t *)W.pr_status.ensure((size_t)n_units + 64);
			AlnReq *d_rq = (AlnReq *)W.pr_req.ensure((size_t)n * sizeof(AlnReq));
			SamDesc *d_ds = (SamDesc *)W.pr_desc.ensure((size_t)n * sizeof(SamDesc));
			uint8_t *hs = (uint8_t *)W.h_pr_status.ensure((size_t)n_units + 64);
			HIP_OK(hipMemcpyAsync(d_tab, tab, (n_tab + (size_t)pp.ltab_n) * 8, hipMemcpyHostToDevice, st));
			HIP_OK(hipMemcpyAsync(d_ok, ok, (size_t)n_units, hipMemcpyHostToDevice, st));
			launch_pair_simple(st, pp, n_units, d_pr_first, d_pr_nfirst, d_ok, d_ann_off, d_ann_alt, d_tab, d_tab + n_tab, d_status, d_rq, d_ds);
			HIP_OK(hipMemcpyAsync(hs, d_status, (size_t)n_units, hipMemcpyDeviceToHost, st));
			stream_wait(st);
			HIP_OK(hipGetLastError());
			pstat = hs; d_pr_req = d_rq; d_pr_desc = d_ds;
		}
		pair_dev_ms = now_ms() - tp0;
	}

	// mate rescue on the device: list the local alignments the pairs of a part will ask for, run them in one launch
	static_assert(sizeof(MswReq) == sizeof(MswReqH) && sizeof(MswRes) == sizeof(MswResH), "host/device record layouts differ");
	const int MSW_MAX_T = 4096;
	const bool gpu_msw = pe && !(opt->flag & MEM_F_NO_RESCUE) && getenv("MPIBWA_HOST_MATESW") == nullptr && (int64_t)max_len * opt->a < 8192 &&
	                     msw_lds_bytes(max_len) <= 160 * 1024;
	double msw_ms = 0;
	static const bool s_cpusec = getenv("MPIBWA_CPUSEC") != nullptr;
	std::atomic<unsigned long long> tsc_plan(0), tsc_emitc(0);
	double cpu_msw = 0, cpu_collect = 0, cpu_emit = 0, sys_emit = 0;
	long pf_emit = 0;
	auto mcollect = [&](Part &P, int slot) {
		stage(10);
		if (!gpu_msw) return;
		double ta = now_ms();
		const double ca = cpu_sec();
		const int nu = P.hi - P.lo, n_blk = (nu + 255) / 256;
		std::vector<std::vector<MswReqH>> blk_req(n_blk);
		std::vector<uint32_t> u_first(nu), u_cnt(nu);
		parallel_for(n_thr, n_blk, 1, [&](int blk) {
			std::vector<MswReqH> &rq = blk_req[blk];
			rq.reserve(256);
			const int lo = P.lo + blk * 256, hi = std::min(P.hi, lo + 256);
			for (int i = lo; i < hi; ++i) {
				const size_t before = rq.size();
				if (!(pstat && pstat[i] == 1)) sam_pe_msw_collect(opt, bns, pes, &seqs[i << 1], &regs[i << 1], i << 1, MSW_MAX_T, rq);
				u_first[i - P.lo] = (uint32_t)before; u_cnt[i - P.lo] = (uint32_t)(rq.size() - before);
			}
		});
		P.mbase.assign(nu + 1, 0);
		for (int i = 0; i < nu; ++i) P.mbase[i + 1] = P.mbase[i] + u_cnt[i];
		P.n_mreq = P.mbase[nu];
		P.mreq = (MswReqH *)WS.h_mreq[slot].ensure(P.n_mreq * sizeof(MswReqH) + 64);
		P.mres = (MswResH *)WS.h_mres[slot].ensure(P.n_mreq * sizeof(MswResH) + 64);
		parallel_for(n_thr, nu, 4096, [&](int i) {
			if (u_cnt[i]) memcpy(&P.mreq[P.mbase[i]], &blk_req[i >> 8][u_first[i]], (size_t)u_cnt[i] * sizeof(MswReqH));
		});
		msw_ms += now_ms() - ta;
		cpu_msw += cpu_sec() - ca;
	};
	auto mlaunch = [&](Part &P, int slot) {   // asynchronous
		stage(11);
		if (!gpu_msw || P.n_mreq == 0) return;
		P.st = a_streams[slot];
		int max_t = 1;
		for (size_t k = 0; k < P.n_mreq; ++k) max_t = std::max(max_t, (int)(P.mreq[k].re - P.mreq[k].rb));
		MswReq *d_req = (MswReq *)WS.mreq[slot].ensure(P.n_mreq * sizeof(MswReq));
		MswRes *d_res = (MswRes *)WS.mres[slot].ensure(P.n_mreq * sizeof(MswRes));
		// row-maximum scratch: at most 2 GiB at a time
		size_t per = std::max<size_t>(64, (((size_t)1 << 31) / ((size_t)max_t * 2)) & ~(size_t)63);
		per = std::min(per, (P.n_mreq + 63) & ~(size_t)63);
		uint16_t *d_rows = (uint16_t *)WS.mrows[slot].ensure(per * (size_t)max_t * 2);
		HIP_OK(hipMemcpyAsync(d_req, P.mreq, P.n_mreq * sizeof(MswReq), hipMemcpyHostToDevice, P.st));
		const MswParams mp = msw_params(opt, bns->l_pac);
		P.mev.start(P.st);
		int *h_ml = (int *)WS.h_mlist[slot].ensure(2 * P.n_mreq * sizeof(int) + 64), *d_ml = (int *)WS.mlist[slot].ensure(2 * P.n_mreq * sizeof(int) + 64);
		static_assert(sizeof(MswReq) == sizeof(MswReqH), "host/device record layouts differ");
		for (size_t b = 0; b < P.n_mreq; b += per) {
			const int cnt = (int)std::min(per, P.n_mreq - b);
			launch_msw(P.st, mp, cnt, d_req + b, d_seq, d_off, d_len, (const uint8_t *)ix.d_pac, d_res + b, d_rows, max_len, (const MswReq *)(P.mreq + b), lens,
			           h_ml + 2 * b, d_ml + 2 * b);
		}
		P.mev.stop(P.st);
		HIP_OK(hipMemcpyAsync(P.mres, d_res, P.n_mreq * sizeof(MswRes), hipMemcpyDeviceToHost, P.st));   // pinned: truly asynchronous
		P.m_launched = true;
	};
	auto mfinish = [&](Part &P) {
		stage(12);
		if (!P.m_launched) return;
		double ta = now_ms();
		stream_wait(P.st);
		HIP_OK(hipGetLastError());
		STAT.k_msw_ms += P.mev.ms();
		STAT.n_msw += P.n_mreq;
		msw_ms += now_ms() - ta;
	};

	// A: decisions + the list of CIGARs to compute.  Two rounds, so that the units that asked for no mate-rescue alignment
	// (most of them) are done while msw_kernel is still running: round 0 = those units, round 1 = the rest + the flat list.
	auto collect = [&](Part &P, int round) {
		stage(13);
		double ta = now_ms();
		const double ca = cpu_sec();
		const int nu = P.hi - P.lo, n_blk = (nu + 255) / 256;
		if (round == 0) { P.blk_req.assign(n_blk, std::vector<AlnReqH>()); P.u_first.assign(nu, 0); P.u_cnt.assign(nu, 0); }
		parallel_for(n_thr, n_blk, 1, [&](int blk) {
			std::vector<AlnReqH> &rq = P.blk_req[blk];
			if (round == 0) rq.reserve(256 * 3);
			AlnCtx ctx;
			ctx.mode = AlnCtx::COLLECT; ctx.reqs = &rq;
			const int lo = P.lo + blk * 256, hi = std::min(P.hi, lo + 256);
			unsigned long long tsc_plan_blk = 0, tsc_emitc_blk = 0;
			for (int i = lo; i < hi; ++i) {
				const int k = i - P.lo;
				if (pstat && pstat[i] == 1) continue;   // decided on the device
				const bool waits = P.m_launched && P.mbase[k + 1] != P.mbase[k];   // needs results of the mate-rescue kernel
				if (waits != (round == 1)) continue;
				const size_t before = rq.size();
				if (pe) {
					MswCtx mc;
					if (waits) { mc.req = P.mreq + P.mbase[k]; mc.res = P.mres + P.mbase[k]; mc.n = (int)(P.mbase[k + 1] - P.mbase[k]); }
					const unsigned long long c0 = s_cpusec ? __builtin_ia32_rdtsc() : 0;
					sam_pe_plan(opt, bns, pac, pes, (uint64_t)((n_processed >> 1) + i), &seqs[i << 1], &regs[i << 1], plans[i], waits ? &mc : nullptr,
					            i << 1);
					const unsigned long long c1 = s_cpusec ? __builtin_ia32_rdtsc() : 0;
					ctx.desc = gpu_sam ? &sdesc[i << 1] : nullptr;
					if (gpu_aln) sam_pe_emit(opt, bns, pac, pes, &seqs[i << 1], &regs[i << 1], plans[i], &ctx, i << 1);
					if (s_cpusec) { tsc_plan_blk += c1 - c0; tsc_emitc_blk += __builtin_ia32_rdtsc() - c1; }
				} else {
					mark_primary_se(opt, regs[i], n_processed + i);
					if (opt->flag & MEM_F_PRIMARY5) reorder_primary5(opt->T, regs[i]);
					if (gpu_aln) reg2sam(opt, bns, pac, &seqs[i], regs[i], 0, 0, &ctx, i);
				}
				P.u_first[k] = (uint32_t)before; P.u_cnt[k] = (uint32_t)(rq.size() - before);
			}
			if (s_cpusec) { tsc_plan += tsc_plan_blk; tsc_emitc += tsc_emitc_blk; }
		});
		if (round == 1) {
			P.base.assign(nu + 1, 0);
			for (int i = 0; i < nu; ++i) P.base[i + 1] = P.base[i] + P.u_cnt[i];
			P.n_req = P.base[nu];
			P.req = (AlnReqH *)WS.h_areq[&P - parts].ensure(P.n_req * sizeof(AlnReqH) + 64);
			parallel_for(n_thr, nu, 4096, [&](int i) {
				if (P.u_cnt[i]) memcpy(&P.req[P.base[i]], &P.blk_req[i >> 8][P.u_first[i]], (size_t)P.u_cnt[i] * sizeof(AlnReqH));
			});
		}
		plan_ms += now_ms() - ta;
		cpu_collect += cpu_sec() - ca;
	};
	// The pairs decided on the device need nothing from the host any more: their CIGARs and records are queued right behind the
	// pairing kernel, on a stream of their own, and run under the host's rescue listing, planning and the mate-rescue kernel.
	auto launch_dev = [&](Part &P, int slot) {
		stage(14);
		if (!pstat) return;
		Part::DevJob &J = P.dj;
		const int nu = P.hi - P.lo, r0 = P.lo << 1, nr = nu << 1;
		J.n_req = (size_t)nr;
		if (!J.n_req) return;
		J.st = C.d_streams[slot];
		J.pool_bytes = J.n_req * 96 + ((size_t)48 << 20);
		J.d_hdr = (AlnHdr *)WS.dj_hdr[slot].ensure(J.n_req * sizeof(AlnHdr));
		J.d_pool = (uint8_t *)WS.dj_pool[slot].ensure(J.pool_bytes);
		J.d_cnt = (unsigned long long *)WS.dj_cnt[slot].ensure(256);
		HIP_OK(hipMemsetAsync(J.d_cnt, 0, 256, J.st));
		AlnParams ap;
		ap.l_pac = bns->l_pac; ap.a = opt->a; ap.w = opt->w;
		ExtParams ep;
		memcpy(ep.mat, opt->mat, 25);
		ep.o_del = opt->o_del; ep.e_del = opt->e_del; ep.o_ins = opt->o_ins; ep.e_ins = opt->e_ins; ep.zdrop = opt->zdrop;
		int *d_lists = (int *)WS.dj_list[slot].ensure(J.n_req * 3 * sizeof(int));
		J.ev.start(J.st);
		launch_aln(J.st, ap, ep, (int)J.n_req, d_pr_req + r0, d_seq, d_off, (const uint8_t *)ix.d_pac, d_gap, J.d_hdr, J.d_pool, J.d_cnt, J.pool_bytes, max_len,
		           max_len + 256, d_lists);
		J.ev.stop(J.st);
		int *hb = (int *)WS.hj_base[slot].ensure((size_t)(nu + 1) * 4 + 64);
		for (int k = 0; k <= nu; ++k) hb[k] = 2 * k;
		int *d_base = (int *)WS.dj_base[slot].ensure((size_t)(nu + 1) * 4);
		J.arena_bytes = (size_t)nr * (size_t)(2 * max_len + 320) + (1 << 20);
		uint8_t *d_arena = (uint8_t *)WS.dj_arena[slot].ensure(J.arena_bytes);
		unsigned long long *d_used = (unsigned long long *)WS.dj_used[slot].ensure(64);
		unsigned long long *d_ooff = (unsigned long long *)WS.dj_ooff[slot].ensure((size_t)nr * 8);
		int *d_olen = (int *)WS.dj_olen[slot].ensure((size_t)nr * 4);
		HIP_OK(hipMemcpyAsync(d_base, hb, (size_t)(nu + 1) * 4, hipMemcpyHostToDevice, J.st));
		HIP_OK(hipMemsetAsync(d_used, 0, 64, J.st));
		launch_sam_emit(J.st, sam_par, nr, d_pr_desc + r0, d_base, J.d_hdr, J.d_pool, d_seq, d_off + r0, d_len + r0, d_qual, d_names, d_noff + r0, d_ann_off,
		                d_ann_names, d_ann_noff, d_arena, J.arena_bytes, d_used, d_ooff, d_olen);
		J.launched = true;
	};
	auto finish_dev = [&](Part &P, int slot) {
		stage(15);
		Part::DevJob &J = P.dj;
		if (!J.launched) return;
		double ta = now_ms();
		const int nr = (P.hi - P.lo) << 1;
		stream_wait(J.st);
		HIP_OK(hipGetLastError());
		STAT.k_aln_ms += J.ev.ms();
		STAT.n_aln += J.n_req;
		unsigned long long *small = (unsigned long long *)WS.h_small[slot].ensure(256);
		HIP_OK(hipMemcpyAsync(small, WS.dj_used[slot].p, 8, hipMemcpyDeviceToHost, J.st));
		stream_wait(J.st);
		const unsigned long long used = std::min<unsigned long long>(small[0], J.arena_bytes);
		uint8_t *ha = (uint8_t *)WS.hj_arena[slot].ensure((size_t)used + 64);
		unsigned long long *ho = (unsigned long long *)WS.hj_ooff[slot].ensure((size_t)nr * 8 + 64);
		int *hl = (int *)WS.hj_olen[slot].ensure((size_t)nr * 4 + 64);
		if (used) HIP_OK(hipMemcpyAsync(ha, WS.dj_arena[slot].p, (size_t)used, hipMemcpyDeviceToHost, J.st));
		HIP_OK(hipMemcpyAsync(ho, WS.dj_ooff[slot].p, (size_t)nr * 8, hipMemcpyDeviceToHost, J.st));
		HIP_OK(hipMemcpyAsync(hl, WS.dj_olen[slot].p, (size_t)nr * 4, hipMemcpyDeviceToHost, J.st));
		stream_wait(J.st);
		J.sarena = ha; J.sooff = ho; J.solen = hl;
		// a record handed back (CIGAR declined, row overflow): the host redoes that pair and needs the CIGAR results of the job
		bool any_back = false;
		for (int k = 0; k < (P.hi - P.lo) && !any_back; ++k)
			if (pstat[P.lo + k] == 1 && (hl[2 * k] < 0 || hl[2 * k + 1] < 0)) any_back = true;
		if (any_back) {
			unsigned long long *cnt8 = small + 8;
			HIP_OK(hipMemcpyAsync(cnt8, J.d_cnt, 64, hipMemcpyDeviceToHost, J.st));
			stream_wait(J.st);
			const size_t pu = std::min<size_t>(cnt8[0], J.pool_bytes);
			AlnHdrH *hh = (AlnHdrH *)WS.hj_hdr[slot].ensure(J.n_req * sizeof(AlnHdr) + 64);
			uint8_t *hp = (uint8_t *)WS.hj_pool[slot].ensure(pu + 64);
			HIP_OK(hipMemcpyAsync(hh, J.d_hdr, J.n_req * sizeof(AlnHdr), hipMemcpyDeviceToHost, J.st));
			if (pu) HIP_OK(hipMemcpyAsync(hp, J.d_pool, pu, hipMemcpyDeviceToHost, J.st));
			stream_wait(J.st);
			J.hdr = hh; J.pool = hp;
		}
		aln_wait_ms += now_ms() - ta;
	};
	auto launch = [&](Part &P, int slot) {   // B (asynchronous)
		stage(16);
		const size_t n_req = P.n_req;
		P.slot = slot;
		if (!gpu_aln || n_req == 0) return;
		static_assert(sizeof(AlnReq) == sizeof(AlnReqH) && sizeof(AlnHdr) == sizeof(AlnHdrH), "host/device record layouts differ");
		P.st = a_streams[slot];
		P.pool_bytes = n_req * 96 + ((size_t)48 << 20);   // + room for the partly used last slab of every wave (aln_kernel.hip: ALN_SLAB)
		AlnReq *d_req = (AlnReq *)(slot ? WS.areq2 : WS.areq).ensure(n_req * sizeof(AlnReq));
		P.d_hdr = (AlnHdr *)(slot ? WS.ahdr2 : WS.ahdr).ensure(n_req * sizeof(AlnHdr));
		P.d_pool = (uint8_t *)(slot ? WS.apool2 : WS.apool).ensure(P.pool_bytes);
		P.d_cnt = (unsigned long long *)(slot ? WS.acnt2 : WS.acnt).ensure(256);
		HIP_OK(hipMemcpyAsync(d_req, P.req, n_req * sizeof(AlnReq), hipMemcpyHostToDevice, P.st));
		HIP_OK(hipMemsetAsync(P.d_cnt, 0, 256, P.st));
		AlnParams ap;
		ap.l_pac = bns->l_pac; ap.a = opt->a; ap.w = opt->w;
		ExtParams ep;
		memcpy(ep.mat, opt->mat, 25);
		ep.o_del = opt->o_del; ep.e_del = opt->e_del; ep.o_ins = opt->o_ins; ep.e_ins = opt->e_ins; ep.zdrop = opt->zdrop;
		P.ev.start(P.st);
		int *d_lists = (int *)WS.alist[slot].ensure(n_req * 3 * sizeof(int));
		launch_aln(P.st, ap, ep, (int)n_req, d_req, d_seq, d_off, (const uint8_t *)ix.d_pac, d_gap, P.d_hdr, P.d_pool, P.d_cnt, P.pool_bytes, max_len,
		           max_len + 256, d_lists);
		P.ev.stop(P.st);   // results are fetched in finish(): a D2H copy into pageable memory would block the host here
		if (gpu_sam) {   // the records of the part's qualifying pairs, queued right behind their CIGARs
			const int r0 = P.lo << 1, nr = (P.hi - P.lo) << 1, nu = P.hi - P.lo;
			SamDesc *d_desc = (SamDesc *)W.sdesc.ensure((size_t)n * sizeof(SamDesc));
			int *hb = (int *)WS.h_sbase[slot].ensure((size_t)(nu + 1) * 4 + 64);
			for (int k = 0; k <= nu; ++k) hb[k] = (int)P.base[k];
			int *d_base = (int *)WS.sbase[slot].ensure((size_t)(nu + 1) * 4);
			P.arena_bytes = (size_t)nr * (size_t)(2 * max_len + 320) + (1 << 20);
			uint8_t *d_arena = (uint8_t *)WS.sarena[slot].ensure(P.arena_bytes);
			unsigned long long *d_used = (unsigned long long *)WS.sused[slot].ensure(64);
			unsigned long long *d_ooff = (unsigned long long *)WS.sooff[slot].ensure((size_t)nr * 8);
			int *d_olen = (int *)WS.solen[slot].ensure((size_t)nr * 4);
			HIP_OK(hipMemcpyAsync(d_desc + r0, sdesc + r0, (size_t)nr * sizeof(SamDesc), hipMemcpyHostToDevice, P.st));
			HIP_OK(hipMemcpyAsync(d_base, hb, (size_t)(nu + 1) * 4, hipMemcpyHostToDevice, P.st));
			HIP_OK(hipMemsetAsync(d_used, 0, 64, P.st));
			launch_sam_emit(P.st, sam_par, nr, d_desc + r0, d_base, P.d_hdr, P.d_pool, d_seq, d_off + r0, d_len + r0, d_qual, d_names, d_noff + r0,
			                d_ann_off, d_ann_names, d_ann_noff, d_arena, P.arena_bytes, d_used, d_ooff, d_olen);
			P.sam_launched = true;
		}
	};
	auto finish = [&](Part &P) {   // wait for B, fetch the pool
		stage(17);
		const size_t n_req = P.n_req;
		if (!gpu_aln || n_req == 0) return;
		double ta = now_ms();
		stream_wait(P.st);
		HIP_OK(hipGetLastError());
		unsigned long long *small = (unsigned long long *)WS.h_small[P.slot].ensure(256) + 16;
		HIP_OK(hipMemcpyAsync(small, P.d_cnt, 64, hipMemcpyDeviceToHost, P.st));
		stream_wait(P.st);
		memcpy(P.cnt, small, 64);
		STAT.k_aln_ms += P.ev.ms();
		if (s_cpusec) {
			unsigned long long c[16];
			HIP_OK(hipMemcpy(c, P.d_cnt, sizeof c, hipMemcpyDeviceToHost));
			fprintf(stderr, "[aln lists] %zu requests: same-length %llu, narrow DP %llu, full DP %llu\n", n_req, c[8], c[9], c[10]);
		}
		size_t used = std::min<size_t>(P.cnt[0], P.pool_bytes);
		P.hdr = (AlnHdrH *)WS.h_ahdr[P.slot].ensure(n_req * sizeof(AlnHdr) + 64);
		P.pool = (uint8_t *)WS.h_apool[P.slot].ensure(used + 64);
		HIP_OK(hipMemcpyAsync(P.hdr, P.d_hdr, n_req * sizeof(AlnHdr), hipMemcpyDeviceToHost, P.st));
		if (used) HIP_OK(hipMemcpyAsync(P.pool, P.d_pool, used, hipMemcpyDeviceToHost, P.st));
		stream_wait(P.st);
		STAT.n_aln += n_req;
		if (P.sam_launched) {
			const int nr = (P.hi - P.lo) << 1;
			HIP_OK(hipMemcpyAsync(small + 8, WS.sused[P.slot].p, 8, hipMemcpyDeviceToHost, P.st));
			stream_wait(P.st);
			const unsigned long long used = std::min<unsigned long long>(small[8], P.arena_bytes);
			uint8_t *ha = (uint8_t *)WS.h_sarena[P.slot].ensure((size_t)used + 64);
			unsigned long long *ho = (unsigned long long *)WS.h_sooff[P.slot].ensure((size_t)nr * 8 + 64);
			int *hl = (int *)WS.h_solen[P.slot].ensure((size_t)nr * 4 + 64);
			if (used) HIP_OK(hipMemcpyAsync(ha, WS.sarena[P.slot].p, (size_t)used, hipMemcpyDeviceToHost, P.st));
			HIP_OK(hipMemcpyAsync(ho, WS.sooff[P.slot].p, (size_t)nr * 8, hipMemcpyDeviceToHost, P.st));
			HIP_OK(hipMemcpyAsync(hl, WS.solen[P.slot].p, (size_t)nr * 4, hipMemcpyDeviceToHost, P.st));
			stream_wait(P.st);
			P.sarena = ha; P.sooff = ho; P.solen = hl;
		}
		aln_wait_ms += now_ms() - ta;
	};
	double emit_ms = 0;
	std::atomic<unsigned long long> n_sam_dev(0), tsc_devcopy(0);
	// which: 0 = the records of the pairs decided on the device (as soon as their job is back: the copies run under the kernels
	// of the other pairs), 1 = everything else, 2 = both
	auto replay = [&](Part &P, int which = 2) {   // C
		stage(18);
		const double ta = now_ms(), ca = cpu_sec(), sa_ = sys_sec();
		const long pf = page_faults();
		if (pe) {
			// per-block counters: a shared atomic bumped once per pair costs more than copying the pair's two records
			parallel_blocks(n_thr, P.hi - P.lo, 128, [&](int, int, int k_lo, int k_hi) {
				unsigned long long n_dev = 0, tsc = 0;
				for (int k = k_lo; k < k_hi; ++k) {
					const int i = P.lo + k;
					const bool dev_pair_k = pstat && pstat[i] == 1;
					const int *solen = dev_pair_k ? P.dj.solen : P.solen;
					const unsigned long long *sooff = dev_pair_k ? P.dj.sooff : P.sooff;
					const uint8_t *sarena = dev_pair_k ? P.dj.sarena : P.sarena;
					const bool early = dev_pair_k && solen && solen[2 * k] >= 0 && solen[2 * k + 1] >= 0;   // pass 0's pairs
					if (which != 2 && early != (which == 0)) continue;
					if (solen && solen[2 * k] >= 0 && solen[2 * k + 1] >= 0) {   // both records were written by sam_kernel
						const unsigned long long tq0 = s_cpusec ? __builtin_ia32_rdtsc() : 0;
						for (int e = 0; e < 2; ++e) {
							const int len = solen[2 * k + e];
							char *sam = (char *)malloc((size_t)len + 1);   // ownership passes to the caller, who free()s it
							if (!sam) die("out of memory");
							memcpy(sam, sarena + sooff[2 * k + e], (size_t)len);
							sam[len] = 0;
							seqs[(i << 1) + e].sam = sam;
						}
						n_dev += 2;
						if (s_cpusec) tsc += __builtin_ia32_rdtsc() - tq0;
						continue;
					}
					AlnCtx ctx;
					if (gpu_aln) { ctx.mode = AlnCtx::REPLAY; ctx.hdr = P.hdr; ctx.pool = P.pool; ctx.cursor = P.base[k]; }
					if (dev_pair_k) {   // the device decided the pair but handed a record back: the host decides it again (same two requests, same order)
						sam_pe_plan(opt, bns, pac, pes, (uint64_t)((n_processed >> 1) + i), &seqs[i << 1], &regs[i << 1], plans[i], nullptr, i << 1);
						ctx.hdr = P.dj.hdr; ctx.pool = P.dj.pool; ctx.cursor = 2 * (size_t)k;
					}
					sam_pe_emit(opt, bns, pac, pes, &seqs[i << 1], &regs[i << 1], plans[i], gpu_aln ? &ctx : nullptr, i << 1);
				}
				n_sam_dev += n_dev; tsc_devcopy += tsc;
			});
		} else if (which != 0) {
			parallel_for(n_thr, P.hi - P.lo, 256, [&](int k) {
				const int i = P.lo + k;
				AlnCtx ctx;
				if (gpu_aln) { ctx.mode = AlnCtx::REPLAY; ctx.hdr = P.hdr; ctx.pool = P.pool; ctx.cursor = P.base[k]; }
				reg2sam(opt, bns, pac, &seqs[i], regs[i], 0, 0, gpu_aln ? &ctx : nullptr, i);
			});
		}
		emit_ms += now_ms() - ta;
		cpu_emit += cpu_sec() - ca;
		sys_emit += sys_sec() - sa_; pf_emit += page_faults() - pf;
	};
	// When does the device pairs' job go out?  Alone, right behind the pairing kernel (its kernels and the copies of its records run
	// under the host's work on the other pairs: 94.8-96.8 vs 98.7-102.8 ms per chunk); with other calls in flight, next to the host
	// pairs' job (their kernels fill the gaps anyway and an early launch only delays their seeding: 12.0-12.7 vs 10.9-11.3 Mreads/s).
	// MPIBWA_DEV_JOB_LATE=0/1 forces either.
	const char *dle = getenv("MPIBWA_DEV_JOB_LATE");
	const bool dev_late = dle ? atoi(dle) != 0 : lease.crowded;
	if (n_parts == 1) {
		parts[0].lo = 0; parts[0].hi = n_units;
		if (!dev_late) launch_dev(parts[0], 0);
		mcollect(parts[0], 0); mlaunch(parts[0], 0);
		if (!dev_late) { finish_dev(parts[0], 0); replay(parts[0], 0); }
		collect(parts[0], 0); mfinish(parts[0]); collect(parts[0], 1);
		if (dev_late) launch_dev(parts[0], 0);
		launch(parts[0], 0); finish(parts[0]);
		if (dev_late) { finish_dev(parts[0], 0); replay(parts[0], 0); }
		hprof_report("decisions + request lists");
		replay(parts[0], 1);
	} else {
		parts[0].lo = 0; parts[0].hi = n_units / 2; parts[1].lo = n_units / 2; parts[1].hi = n_units;
		if (!dev_late) { launch_dev(parts[0], 0); launch_dev(parts[1], 1); }
		mcollect(parts[0], 0); mlaunch(parts[0], 0);
		mcollect(parts[1], 1); mlaunch(parts[1], 1);
		if (!dev_late) { finish_dev(parts[0], 0); replay(parts[0], 0); }   // (the mate-rescue kernels of both parts are running)
		collect(parts[0], 0); mfinish(parts[0]); collect(parts[0], 1);
		if (dev_late) launch_dev(parts[0], 0);
		launch(parts[0], 0);
		if (!dev_late) { finish_dev(parts[1], 1); replay(parts[1], 0); }
		collect(parts[1], 0); mfinish(parts[1]); collect(parts[1], 1);
		if (dev_late) launch_dev(parts[1], 1);
		launch(parts[1], 1);
		finish(parts[0]);
		if (dev_late) { finish_dev(parts[0], 0); replay(parts[0], 0); }
		replay(parts[0], 1);
		finish(parts[1]);
		if (dev_late) { finish_dev(parts[1], 1); replay(parts[1], 0); }
		replay(parts[1], 1);
	}
	STAT.plan_ms = plan_ms; STAT.aln_ms = aln_wait_ms; STAT.msw_ms = msw_ms; STAT.emit_ms = emit_ms;
	STAT.n_sam_dev = n_sam_dev.load();
	if (pstat) {
		uint64_t c[16] = {0};
		for (int k = 0; k < n_units; ++k) ++c[pstat[k] & 15];
		STAT.n_pair_dev = c[1];
		if (s_cpusec) fprintf(stderr, "[pair_kernel] %d pairs: decided %llu; host: no/unnamed hit %llu, > %d hits %llu, patch %llu, ALT/length %llu, rescue %llu, no proper pair %llu, score %llu, second primary hit %llu, XA %llu\n",
		                      n_units, (unsigned long long)c[1], (unsigned long long)c[2], PR_MAXREG, (unsigned long long)c[3], (unsigned long long)c[4], (unsigned long long)c[6],
		                      (unsigned long long)c[7], (unsigned long long)c[8], (unsigned long long)c[9], (unsigned long long)c[10], (unsigned long long)c[11]);
	}
	STAT.plan_ms += pair_dev_ms;
	double t8 = now_ms();
	stage(19);
	hprof_report("sam stage");
	if (s_cpusec) fprintf(stderr, "[plan Mcycles] sam_pe_plan %.0f  emit(collect) %.0f  device-record copy %.0f (%llu records)\n", tsc_plan.load() * 1e-6, tsc_emitc.load() * 1e-6, tsc_devcopy.load() * 1e-6, n_sam_dev.load());
	if (g_hprof_on || s_cpusec)
		fprintf(stderr, "[cpu-sec] encode+h2d %.3f  phase1 %.3f  pestat+sam %.3f (msw-collect %.3f, plan+collect %.3f, emit %.3f [sys %.3f, %ld page faults])  total %.3f  sys %.3f  wall %.3f\n",
		        c1 - c_begin, c6 - c1, cpu_sec() - c6, cpu_msw, cpu_collect, cpu_emit, sys_emit, pf_emit, cpu_sec() - c_begin, sys_sec() - s_begin, (t8 - t_begin) * 1e-3);
	// release the per-read containers in parallel (millions of small blocks: serial destruction costs ~0.2 s per chunk)
	parallel_for(n_thr, n, 8192, [&](int i) { HRegV().swap(regs[i]); });   // only reads that outgrew their arena slice own memory
	STAT.n_reads = n;
	STAT.h2d_ms = t1 - t_begin;
	STAT.pestat_ms = t7 - t6; STAT.sam_ms = t8 - t7; STAT.total_ms = now_ms() - t_begin;
	if (bwa_verbose >= 3)
		fprintf(stderr, "[M::%s] Processed %d reads in %.3f CPU sec, %.3f real sec\n", "mem_process_seqs", n, cpu_sec() - c_begin,
		        (t8 - t_begin) * 1e-3);
}
