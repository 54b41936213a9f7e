/* mpibwa_idx.c — the product's counterpart of the reference's third program, mpiBWAIdx (src/pidx.c:28-66):
 *
 *     mpibwa_idx [--build] REF.fa
 *
 * loads the bwa index files REF.fa.{bwt,sa,ann,amb,pac[,alt]} and writes REF.fa.map, the one-piece image the ranks of a node attach
 * (bwa_idx2mem, src/bwa.c:347-386; byte-identical to the reference's: tests/test_boundary.py).  --build first makes those files from the
 * FASTA with the library's builder (`bwa index` is a separate program for the reference; byte-identical files: tests/test_index.py).
 * No MPI, no GPU.  Plain C99 on include/mpibwa_amd.h; built by mpibwa_amd/build.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mpibwa_amd.h"

int main(int argc, char **argv)
{
	int build = 0;
	const char *ref = 0;
	for (int i = 1; i < argc; ++i) {
		if (!strcmp(argv[i], "--build")) build = 1;
		else if (argv[i][0] == '-' || ref) ref = "";
		else ref = argv[i];
	}
	if (!ref || !*ref) {
		fprintf(stderr, "usage: %s [--build] REF.fa\n  writes REF.fa.map from REF.fa.{bwt,sa,ann,amb,pac}; --build makes those from the FASTA first\n", argv[0]);
		return 1;
	}
	if (build && mi355x_index_build(ref, ref) != 0) {
		fprintf(stderr, "[mpibwa_idx] cannot build the index of %s\n", ref);
		return 1;
	}
	char *map = malloc(strlen(ref) + 8);
	sprintf(map, "%s.map", ref);
	if (mi355x_write_map(ref, map) != 0) {
		fprintf(stderr, "[mpibwa_idx] cannot write %s (are %s.bwt, .sa, .ann, .amb and .pac there?)\n", map, ref);
		return 1;
	}
	fprintf(stderr, "[mpibwa_idx] wrote %s\n", map);
	free(map);
	return 0;
}
