// hprof.h — optional host-stage profiler (MPIBWA_PROF=1): cumulative thread-seconds per section
#ifndef MBW_HPROF_H
#define MBW_HPROF_H
#include <atomic>
#include <chrono>
namespace mbw {
enum { HP_MATESW = 0, HP_ALIGN2, HP_REG2ALN, HP_GLOBAL2, HP_GENALT, HP_ALN2SAM, HP_MARK, HP_PAIR, HP_DEDUP, HP_N };
extern std::atomic<long long> g_hprof[HP_N];
extern std::atomic<long long> g_hcount[HP_N];
extern bool g_hprof_on;
struct HProf {
	int k; std::chrono::steady_clock::time_point t0;
	explicit HProf(int k_) : k(k_) { if (g_hprof_on) t0 = std::chrono::steady_clock::now(); }
	~HProf() { if (g_hprof_on) { g_hprof[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); ++g_hcount[k]; } }
};
void hprof_report(const char *tag);
}
#endif
