// prewarm.cpp — mi355x_prewarm(): the first-use cost of the call contexts paid before the caller's chunk loop starts.
//
// The first mem_process_seqs() call of a context allocates its device and page-locked work buffers (a few tens of buffers,
// grown in steps until they hold a chunk), creates its streams and events, and starts the host thread pool; the first launch
// of every kernel loads its code object.  A chunk that normally takes 0.6 s took 1.2-2.3 s that way, once per context, and the
// eight workers of the driver all paid it inside the loop the reference brackets with MPI_Wtime (src/mainParallel.c:1238-1319).
// This entry runs `n_calls` calls side by side on reads SAMPLED FROM THE RESIDENT REFERENCE (about one substitution per hundred
// bases; pairs facing each other about 300 bp apart when opt asks for paired ends) and throws the text away.  It is an ordinary
// caller of the exported mem_process_seqs (all the pipeline learns from it is how many calls to expect in flight) — so whatever a chunk of that shape touches
// has been touched, at the size such a chunk needs; a caller with other work to do before its loop (reading the FASTQ
// offsets, src/mainParallel.c:640-1100) runs it on a thread of its own meanwhile (driver/mpibwa_gpu.c does).
#include "internal.h"
#include "device.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

using namespace mbw;

namespace {

inline int pac_base(const uint8_t *pac, int64_t p) { return (pac[p >> 2] >> ((~p & 3) << 1)) & 3; }

struct Lcg {
	uint64_t s;
	explicit Lcg(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) {}
	uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); }
};

// one caller: its reads, one call, its text dropped
void one_call(const mem_opt_t *opt, const bwt_t *bwt, const bntseq_t *bns, const uint8_t *pac, int n, int len, uint64_t seed)
{
	static const char nt[] = "ACGT";
	const bool pe = (opt->flag & MEM_F_PE) != 0;
	const int64_t l_pac = bns->l_pac;
	std::vector<bseq1_t> seqs((size_t)n);
	// one block for all names, bases and qualities of the call (mem_process_seqs neither frees nor reallocates them)
	const size_t per = (size_t)len + 1;
	std::vector<char> bases((size_t)n * per), quals((size_t)n * per), names((size_t)n * 12);
	Lcg g(seed);
	for (int i = 0; i < n; i += pe ? 2 : 1) {
		const int ins = pe ? std::max(len, 240 + (int)(g.next() % 120)) : len;
		int64_t p = l_pac > ins + 1 ? (int64_t)(((uint64_t)g.next() << 31 | g.next()) % (uint64_t)(l_pac - ins)) : 0;
		const int span = (int)std::min<int64_t>(ins, l_pac - p);
		for (int e = 0; e < (pe ? 2 : 1) && i + e < n; ++e) {
			char *b = &bases[(size_t)(i + e) * per], *q = &quals[(size_t)(i + e) * per], *nm = &names[(size_t)(i + e) * 12];
			for (int j = 0; j < len; ++j) {
				int c;
				if (e == 0) c = j < span ? pac_base(pac, p + j) : 0;
				else { const int64_t at = p + span - 1 - j; c = at >= p ? 3 - pac_base(pac, at) : 0; }   // the mate: reverse complement of the far end
				if (g.next() % 100 == 0) c = (c + 1 + (int)(g.next() % 3)) & 3;
				b[j] = nt[c];
				q[j] = 'I';
			}
			b[len] = 0; q[len] = 0;
			snprintf(nm, 12, "w%d", pe ? i >> 1 : i);
			bseq1_t &s = seqs[(size_t)i + e];
			memset(&s, 0, sizeof s);
			s.l_seq = len; s.name = nm; s.seq = b; s.qual = q;
		}
	}
	mem_process_seqs(opt, bwt, bns, pac, 0, n, seqs.data(), nullptr);
	for (bseq1_t &s : seqs) free(s.sam);
}

}   // namespace

// n_reads: reads per call (both ends counted; rounded down to whole pairs for paired ends), read_len: bases per read,
// n_calls: calls side by side — what the caller will keep in flight, and the library takes it as that.  Returns the seconds it took; 0 when there was nothing to do.
extern "C" double mi355x_prewarm(const mem_opt_t *opt, const bwt_t *bwt, const bntseq_t *bns, const uint8_t *pac, int n_reads, int read_len, int n_calls)
{
	if (!opt || !bwt || !bns || !pac) die("mi355x_prewarm: opt, bwt, bns and pac are the ones the chunk loop will pass to mem_process_seqs");
	if (opt->flag & MEM_F_PE) n_reads &= ~1;
	if (n_reads <= 0 || read_len <= 0 || n_calls <= 0 || bns->l_pac < read_len) return 0.;
	if (n_calls > mi355x_max_calls()) n_calls = mi355x_max_calls();
	expect_calls_in_flight(n_calls);   // (pipeline.hip: three or more = the caller's calls run in their busy mode from the first one on)
	const auto t0 = std::chrono::steady_clock::now();
	std::vector<std::thread> th;
	for (int c = 1; c < n_calls; ++c) th.emplace_back(one_call, opt, bwt, bns, pac, n_reads, read_len, (uint64_t)c);
	one_call(opt, bwt, bns, pac, n_reads, read_len, 0);
	for (std::thread &t : th) t.join();
	return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
