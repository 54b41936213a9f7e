// internal.h — shared declarations of the host side of libmpibwa_amd.so
#ifndef MBW_INTERNAL_H
#define MBW_INTERNAL_H

#include "../../include/mpibwa_amd.h"

#include <cstdint>
#include <cstddef>
#include <string>
#include <vector>

namespace mbw {

[[noreturn]] void die(const char *fmt, ...);
void fill_cnt_table(uint32_t tab[256]);
extern const uint8_t *const nt4_table_ptr;
#define nt4_table nt4_table_ptr

// ---- plain records exchanged between host stages and HIP kernels ----

struct Intv {            // bwtintv_t (src/bwt.h:60): bi-interval + (start<<32|end)
	uint64_t x0, x1, x2, info;
};

struct Seed {            // mem_seed_t (src/bwamem.c:168)
	int64_t rbeg;
	int32_t qbeg, len;
	int32_t score;
};

struct Chain {           // mem_chain_t (src/bwamem.c:174)
	int64_t pos;
	int rid;
	int first;
	uint32_t w;
	int kept;
	int is_alt;
	float frac_rep;
	std::vector<Seed> seeds;
};

struct AlnReg {          // mem_alnreg_t (src/bwamem.h:59), 88 bytes in the reference
	int64_t rb, re;
	int qb, qe;
	int rid;
	int score;
	int truesc;
	int sub;
	int alt_sc;
	int csub;
	int sub_n;
	int w;
	int seedcov;
	int secondary;
	int secondary_all;
	int seedlen0;
	int n_comp;
	int is_alt;
	float frac_rep;
	uint64_t hash;
};
typedef std::vector<AlnReg> AlnRegV;

struct Aln {             // mem_aln_t (src/bwamem.h:87)
	int64_t pos;
	int rid;
	int flag;
	uint32_t is_rev, is_alt, mapq, NM;
	int n_cigar;
	std::vector<uint32_t> cigar;
	std::string *unused;
	const char *XA;
	std::string md;
	int score, sub, alt_sc;
};

} // namespace mbw
#endif
