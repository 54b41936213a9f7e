// temporary: mem_process_seqs / stats until the host pipeline lands
#include "internal.h"
extern "C" void mem_process_seqs(const mem_opt_t *, const bwt_t *, const bntseq_t *, const uint8_t *, int64_t, int, bseq1_t *, const mem_pestat_t *)
{
	mbw::die("mem_process_seqs: pipeline not linked yet");
}
extern "C" void mi355x_last_stats(mi355x_stats_t *st) { *st = mi355x_stats_t(); }
