/* tests/csrc/cpu_e2e.c -- TEST-ONLY driver: the C host (libburst_host.so) on both sides of the ORACLE.
 * Replaces the device call by oracle/liboracle.so's exhaustive orc_search so that parsers, query pipeline,
 * database readers, consolidation and the .b6 writer can be checked against the golden reference outputs on a
 * machine without a GPU.  Never shipped, never measured.
 *   cpu_e2e <db.edx|refs.fa> <queries.fa> <out.b6> <MODE> <id> <fr:0|1> <z:0|1> <shear:-1|len> <report flags>
 *           [<taxonomy file or ""> <suppress:0|1> <strict:0|1> <taxacut>]
 */
#include "burst_host.h"
#include "burst_oracle.h"
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv) {
	if (argc < 10) { fprintf(stderr, "usage\n"); return 1; }
	const char *ref = argv[1], *qf = argv[2], *outp = argv[3], *ms = argv[4];
	float thres = (float)atof(argv[5]); int fr = atoi(argv[6]), z = atoi(argv[7]); long shear = atol(argv[8]);
	BhMode mode = !strcmp(ms, "BEST") ? BH_BEST : !strcmp(ms, "ALLPATHS") ? BH_ALLPATHS : !strcmp(ms, "FORAGE") ? BH_FORAGE : !strcmp(ms, "ANY") ? BH_ANY : BH_CAPITALIST;
	BhQueries Q; BhDb db; int rc;
	if (getenv("BURST_XALPHA")) {      /* -x: the run's own alphabet, as the command line sets it up */
		uint8_t map[256]; int ns = 0;
		if (bh_alphabet_from_files(ref, qf, map, &ns)) { fprintf(stderr, "%s\n", bh_last_error()); return 1; }
		bh_set_alphabet(map);
	}
	if ((rc = bh_queries_load(qf, thres, fr, 0, 0, 12, z, 0, &Q))) { fprintf(stderr, "%s\n", bh_last_error()); return 2; }
	int isdb = bh_is_edx(ref);
	if (isdb > 0) rc = bh_edx_read(ref, &db); else rc = bh_db_from_fasta(ref, Q.maxLen, thres, shear >= 0, shear > 0 ? shear : 500, 0, &db);
	if (rc) { fprintf(stderr, "%s\n", bh_last_error()); return 2; }
	uint8_t lut[256]; bh_score_lut(z, lut);
	uint32_t *E = malloc(Q.numEntries * 4);
	for (uint64_t e = 0; e < Q.numEntries; ++e) E[e] = Q.emac[e];
	uint64_t cap = 1 << 22;
	OrcHit *hits = malloc(cap * sizeof(*hits));
	uint64_t n = orc_search(db.packed, db.clumpLen, db.numRclumps, db.totR, Q.codes, Q.qoff, E, Q.six, Q.rc, (uint32_t)Q.numEntries,
	                        (uint32_t)Q.numUniq, lut, mode == BH_FORAGE || mode == BH_ANY, hits, cap);
	if (n > cap) { fprintf(stderr, "too many hits\n"); return 3; }
	FILE *o = fopen(outp, "wb");
	uint64_t lines = 0;
	BhTax tax; BhTaxOpts txo; memset(&tax, 0, sizeof tax); memset(&txo, 0, sizeof txo); txo.taxacut = 10;
	if (argc >= 14 && argv[10][0]) {
		if ((rc = bh_tax_load(argv[10], &tax))) { fprintf(stderr, "%s\n", bh_last_error()); return 2; }
		txo.tax = &tax; txo.suppress = atoi(argv[11]); txo.strict = atoi(argv[12]); txo.taxacut = (uint32_t)atoi(argv[13]);
	}
	if ((rc = bh_report_tax(o, &db, &Q, (const BhipHit *)hits, n, mode, atoi(argv[9]), txo.tax ? &txo : NULL, &lines))) { fprintf(stderr, "%s\n", bh_last_error()); return 4; }
	fclose(o);
	printf("%lu hits, %lu lines\n", (unsigned long)n, (unsigned long)lines);
	return 0;
}
