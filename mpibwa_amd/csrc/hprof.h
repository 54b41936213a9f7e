// hprof.h — optional host-stage profiler (MPIBWA_PROF=1): cumulative thread time per section.
// Sections are entered millions of times per chunk from 16+ threads, so a section costs two rdtsc and two adds on a
// thread-local record; a thread's record is folded into the global counters when a pool thread leaves a parallel region
// (pipeline.hip: HelperPool), when a thread ends, or when the report is printed (the reporting thread).
#ifndef MBW_HPROF_H
#define MBW_HPROF_H
#include <atomic>
#include <x86intrin.h>
namespace mbw {
enum { HP_MATESW = 0, HP_ALIGN2, HP_REG2ALN, HP_GLOBAL2, HP_GENALT, HP_ALN2SAM, HP_MARK, HP_PAIR, HP_DEDUP, HP_N };
extern std::atomic<long long> g_hprof[HP_N];   // TSC ticks
extern std::atomic<long long> g_hcount[HP_N];
extern bool g_hprof_on;
struct HProfLocal {
	long long t[HP_N] = {0}, c[HP_N] = {0};
	void flush() { for (int i = 0; i < HP_N; ++i) if (c[i]) { g_hprof[i] += t[i]; g_hcount[i] += c[i]; t[i] = c[i] = 0; } }
	~HProfLocal() { flush(); }
};
extern thread_local HProfLocal t_hprof;
struct HProf {
	int k; unsigned long long t0 = 0;
	explicit HProf(int k_) : k(k_) { if (g_hprof_on) t0 = __rdtsc(); }
	~HProf() { if (g_hprof_on) { t_hprof.t[k] += (long long)(__rdtsc() - t0); ++t_hprof.c[k]; } }
};
void hprof_report(const char *tag);
}
#endif
