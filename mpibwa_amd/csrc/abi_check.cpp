// abi_check.cpp — build-time proof that include/mpibwa_amd.h re-declares the reference's x86-64 struct layouts
// (sizes from compiling the reference's headers; field offsets from the declarations at
// src/bwamem.h:25-57, :81-85, src/bwa.h:20-33, src/bwt.h:46-58, src/bntseq.h:41-64).  A host program compiled against the
// reference headers passes its own objects to this library unchanged, so any drift here must fail the build.
// tests/csrc/abi_check.c holds the same asserts in plain C (the header is a C header first).
#include "../../include/mpibwa_amd.h"
#include <cstddef>

#define SZ(T, n) static_assert(sizeof(T) == (n), "sizeof(" #T ") differs from the reference's")
#define OFF(T, f, n) static_assert(offsetof(T, f) == (n), "offsetof(" #T ", " #f ") differs from the reference's")

SZ(mem_opt_t, 168);
OFF(mem_opt_t, a, 0); OFF(mem_opt_t, o_del, 8); OFF(mem_opt_t, o_ins, 16); OFF(mem_opt_t, pen_unpaired, 24);
OFF(mem_opt_t, pen_clip5, 28); OFF(mem_opt_t, w, 36); OFF(mem_opt_t, zdrop, 40); OFF(mem_opt_t, max_mem_intv, 48);
OFF(mem_opt_t, T, 56); OFF(mem_opt_t, flag, 60); OFF(mem_opt_t, min_seed_len, 64); OFF(mem_opt_t, min_chain_weight, 68);
OFF(mem_opt_t, max_chain_extend, 72); OFF(mem_opt_t, split_factor, 76); OFF(mem_opt_t, split_width, 80);
OFF(mem_opt_t, max_occ, 84); OFF(mem_opt_t, max_chain_gap, 88); OFF(mem_opt_t, n_threads, 92); OFF(mem_opt_t, chunk_size, 96);
OFF(mem_opt_t, mask_level, 100); OFF(mem_opt_t, drop_ratio, 104); OFF(mem_opt_t, XA_drop_ratio, 108);
OFF(mem_opt_t, mask_level_redun, 112); OFF(mem_opt_t, mapQ_coef_len, 116); OFF(mem_opt_t, mapQ_coef_fac, 120);
OFF(mem_opt_t, max_ins, 124); OFF(mem_opt_t, max_matesw, 128); OFF(mem_opt_t, max_XA_hits, 132);
OFF(mem_opt_t, max_XA_hits_alt, 136); OFF(mem_opt_t, mat, 140);

SZ(mem_pestat_t, 32);
OFF(mem_pestat_t, low, 0); OFF(mem_pestat_t, high, 4); OFF(mem_pestat_t, failed, 8); OFF(mem_pestat_t, avg, 16); OFF(mem_pestat_t, std, 24);

SZ(bseq1_t, 48);
OFF(bseq1_t, l_seq, 0); OFF(bseq1_t, id, 4); OFF(bseq1_t, name, 8); OFF(bseq1_t, comment, 16); OFF(bseq1_t, seq, 24);
OFF(bseq1_t, qual, 32); OFF(bseq1_t, sam, 40);

SZ(bwt_t, 1120);
OFF(bwt_t, primary, 0); OFF(bwt_t, L2, 8); OFF(bwt_t, seq_len, 48); OFF(bwt_t, bwt_size, 56); OFF(bwt_t, bwt, 64);
OFF(bwt_t, cnt_table, 72); OFF(bwt_t, sa_intv, 1096); OFF(bwt_t, n_sa, 1104); OFF(bwt_t, sa, 1112);

SZ(bntann1_t, 40);
OFF(bntann1_t, offset, 0); OFF(bntann1_t, len, 8); OFF(bntann1_t, n_ambs, 12); OFF(bntann1_t, gi, 16); OFF(bntann1_t, is_alt, 20);
OFF(bntann1_t, name, 24); OFF(bntann1_t, anno, 32);

SZ(bntamb1_t, 16);
OFF(bntamb1_t, offset, 0); OFF(bntamb1_t, len, 8); OFF(bntamb1_t, amb, 12);

SZ(bntseq_t, 48);
OFF(bntseq_t, l_pac, 0); OFF(bntseq_t, n_seqs, 8); OFF(bntseq_t, seed, 12); OFF(bntseq_t, anns, 16); OFF(bntseq_t, n_holes, 24);
OFF(bntseq_t, ambs, 32); OFF(bntseq_t, fp_pac, 40);

SZ(bwaidx_t, 48);
OFF(bwaidx_t, bwt, 0); OFF(bwaidx_t, bns, 8); OFF(bwaidx_t, pac, 16); OFF(bwaidx_t, is_shm, 24); OFF(bwaidx_t, l_mem, 32); OFF(bwaidx_t, mem, 40);

extern "C" int mi355x_abi_version(void) { return 2; }   // bumped whenever include/mpibwa_amd.h changes incompatibly
