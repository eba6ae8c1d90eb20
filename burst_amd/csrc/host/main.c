/* main.c -- `burst_hip`: the reference's command line (burst.c:4902-5164) in front of the MI355X device path.
 *
 *   burst_hip -r DB.edx -a DB.acx -q reads.fa -o out.b6 -m {BEST|ALLPATHS|CAPITALIST|FORAGE|ANY} -i 0.97 [-fr] [-y] [-w]
 *   burst_hip -r refs.fa -q reads.fa -o out.b6 [-s [len]]            direct FASTA (exhaustive, no accelerator)
 *   burst_hip -r refs.fa -d [QUICK|DNA|RNA] [qLen] -o DB.edx [-a DB.acx] [-s [len]] -i 0.97      database construction
 *
 * Flags not on the hot path (-f fingerprints, -p prepass, -x alphabet, -hr) are refused with the
 * reference's exit code 1.  Extra flags: --device N, --batch N (unique queries per device call), -k {12|15}.
 */
#include "burst_host.h"
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <omp.h>
#include <pthread.h>
#include <unistd.h>

#define BH_MAX_GPUS BH_MAX_RANKS
static double wall(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static int code_to_exit(int rc) { return rc == BH_E_USAGE ? 1 : rc == BH_E_IO ? 2 : rc == BH_E_OOM ? 3 : 4; }
#define DIE(rc) do { fprintf(stderr, "%s\n", bh_last_error()); return code_to_exit(rc); } while (0)
#define NEEDARG(name) do { if (++i == argc || argv[i][0] == '-') { printf("ERROR: %s requires an argument\n", name); return 1; } } while (0)

/* make_accelerator (burst.c:3304-3532): on the device when there is one (tuples of every lane, radix sort, fold: bhip_init
 * with K and no tables, then bhip_acx_export), else -- or with --host-acx / -sa -- by the host builder.  Same bytes either way. */
static int build_accelerator(BhDb *db, int K, int z, int device, int on_host, int skip_ambig) {
	if (!on_host && !skip_ambig) {
		void *hh = NULL;
		if (!bh_device_open_ex(db, device, z, K, &hh)) {
			const int rc = bh_acx_from_device(db, hh, K, z);
			bhip_destroy(hh);
			if (!rc) printf(" --> accelerator built on device %d\n", device);
			return rc;
		}
		printf(" --> no device accelerator build (%s); using the host builder\n", bh_last_error());
	}
	return bh_acx_build_ex(db, K, z, skip_ambig);
}

/* the query pipeline (process_queries, burst.c:2980-3223) on a thread of its own: with an .edx database nothing in it depends on
 * the database, so the file is read, indexed, sorted and de-duplicated while the main thread reads the database and brings it
 * onto the devices */
typedef struct { const char *fn; float thres; int do_rc, incl_ws, do_accel, K, z, skip_ambig; BhQueries *Q; int rc; char err[512]; double secs; } Ingest;
static void *ingest_main(void *p) {
	Ingest *a = p;
	const double t0 = wall();
	a->rc = bh_queries_load(a->fn, a->thres, a->do_rc, a->incl_ws, a->do_accel, a->K, a->z, a->skip_ambig, a->Q);
	if (a->rc) snprintf(a->err, sizeof a->err, "%s", bh_last_error());
	a->secs = wall() - t0;
	return NULL;
}

static void usage(void) {
	puts("\nburst_hip: BURST-compatible optimal aligner, MI355X (gfx950) device path");
	puts("--references (-r) <name>: FASTA/edx DB of reference sequences [required]");
	puts("--accelerator (-a) <name>: Creates/uses a helper DB (acx) [optional]");
	puts("--queries (-q) <name>: FASTA file of queries to search [required if aligning]");
	puts("--output (-o) <name>: Blast6/edx file for output alignments/database [required]");
	puts("--forwardreverse (-fr), --whitespace (-w), --nwildcard (-y), --mode (-m) BEST|ALLPATHS|CAPITALIST|FORAGE|ANY");
	puts("--makedb (-d) [name qLen], --id (-i) <decimal>, --threads (-t) <int>, --shear (-s) [len], --noprogress");
	puts("--taxonomy (-b) <name>, --taxacut (-bc) <num>, --taxa_ncbi (-bn), --taxasuppress (-bs) [STRICT]: taxonomy column (interpolated in CAPITALIST)");
	puts("--gpus <int> [--devices a,b,...] [--gather host|rccl]: shard the queries over the GPUs of this node; every rank's records reach host");
	puts("   memory over its own PCIe link and meet there (host, default) or are gathered over RCCL / xGMI to rank 0's device first (rccl)");
	puts("--shard queries|db: with --gpus, cut the queries (database replicated; default) or the database (every rank holds a range of");
	puts("                    clumps and aligns all queries; one all-reduce of the per-query minimum) -- for databases beyond one device");
	puts("--shards <S>: database shards (implies --shard db; default = --gpus): the ranks form gpus / S replica groups of S shards, the queries cut over the groups");
	puts("--device <int>, --batch <int>, -k <12|15>, --make-acx <name> (with -r DB.edx: rebuild the accelerator of a database)");
	puts("--accelerator-device (-ad): no .acx file, the device builds the accelerator from the .edx (word length -k, default 12)");
	puts("--host-acx: build accelerators (-d ... -a, --make-acx) with the host builder instead of the device");
}

int main(int argc, char **argv) {
	BhMode mode = BH_CAPITALIST;                    /* burst.c:81 */
	float thres = 0.97f;                            /* burst.c:93 */
	int xalpha = 0;
	int z = 1, do_rc = 0, incl_ws = 0, makedb = 0, do_shear = 0, do_accel = 0, dedupe = 0, device = 0, K = 0, skip_ambig = 0, threads = 0, rep_flags = 0;
	long shear_amt = 500, db_qlen = 500;            /* burst.c:94 */
	uint32_t latency = 16;                          /* burst.c:83 */
	int n_gpus = 1, n_gpus_given = 0, gather_host = 1, n_dev_list = 0, dev_list[BH_MAX_GPUS], accel_dev = 0, host_acx = 0, shard_db = 0, n_shards = 0;
	uint64_t batch = 1u << 21;      /* unique queries per device batch: the fixed cost of a batch (launches, synchronisation) is about 1 ms of device time */
	const char *ref_FN = 0, *query_FN = 0, *output_FN = 0, *xcel_FN = 0, *mkacx_FN = 0, *tax_FN = 0;
	BhTax taxonomy; memset(&taxonomy, 0, sizeof taxonomy);
	BhTaxOpts txo; memset(&txo, 0, sizeof txo); txo.taxacut = 10;   /* burst.c:92 */
	setenv("GPU_MAX_HW_QUEUES", "8", 0);      /* HIP runtime: hardware queues for the library's four streams (read when the runtime starts) */
	printf("This is burst_hip [MI355X device path; BURST v1.0 semantics]\n");
	if (argc < 2) { usage(); return 1; }
	for (int i = 1; i < argc; ++i) {
		const char *a = argv[i];
		if (!strcmp(a, "--references") || !strcmp(a, "-r")) { NEEDARG("--references"); ref_FN = argv[i]; }
		else if (!strcmp(a, "--queries") || !strcmp(a, "-q")) { NEEDARG("--queries"); query_FN = argv[i]; }
		else if (!strcmp(a, "--output") || !strcmp(a, "-o")) { NEEDARG("--output"); output_FN = argv[i]; }
		else if (!strcmp(a, "--accelerator") || !strcmp(a, "-a")) { NEEDARG("--accelerator"); xcel_FN = argv[i]; do_accel = 1; }
		else if (!strcmp(a, "--accelerator-device") || !strcmp(a, "-ad")) { accel_dev = 1; do_accel = 1; }
		else if (!strcmp(a, "--host-acx")) host_acx = 1;
		else if (!strcmp(a, "--forwardreverse") || !strcmp(a, "-fr")) do_rc = 1;
		else if (!strcmp(a, "--whitespace") || !strcmp(a, "-w")) incl_ws = 1;
		else if (!strcmp(a, "--npenalize") || !strcmp(a, "-n")) z = 1;
		else if (!strcmp(a, "--nwildcard") || !strcmp(a, "-y")) z = 0;
		else if (!strcmp(a, "--mode") || !strcmp(a, "-m")) {
			NEEDARG("--mode");
			if (!strcmp(argv[i], "BEST")) mode = BH_BEST; else if (!strcmp(argv[i], "ALLPATHS")) mode = BH_ALLPATHS;
			else if (!strcmp(argv[i], "CAPITALIST")) mode = BH_CAPITALIST; else if (!strcmp(argv[i], "FORAGE")) mode = BH_FORAGE;
			else if (!strcmp(argv[i], "ANY")) mode = BH_ANY;
			else { printf("Unsupported run mode '%s'\n", argv[i]); return 1; }
		}
		else if (!strcmp(a, "--makedb") || !strcmp(a, "-d")) {
			makedb = 1;
			if (i + 1 != argc && argv[i + 1][0] != '-' && !atol(argv[i + 1])) {
				++i;
				if (strcmp(argv[i], "DNA") && strcmp(argv[i], "RNA") && strcmp(argv[i], "QUICK")) { printf("Unsupported makedb mode '%s'\n", argv[i]); return 1; }
				if (strcmp(argv[i], "QUICK")) printf(" --> NOTE: -d %s: the compressive clustering of the reference's builder is not included; the database is laid out as -d QUICK (same alignments, less compact)\n", argv[i]);
			}
			if (i + 1 != argc && argv[i + 1][0] != '-') { db_qlen = atol(argv[++i]); if (db_qlen <= 0) { fprintf(stderr, "ERROR: bad max query length '%s'\n", argv[i]); return 1; } }
		}
		else if (!strcmp(a, "--id") || !strcmp(a, "-i")) {
			NEEDARG("--id"); thres = (float)atof(argv[i]);
			if (thres > 1.f || thres < 0.f) { puts("Invalid id range [0-1]"); return 1; }
			if (thres < 0.01f) thres = 0.01f;
		}
		else if (!strcmp(a, "--threads") || !strcmp(a, "-t")) { NEEDARG("--threads"); threads = atoi(argv[i]); }
		else if (!strcmp(a, "--shear") || !strcmp(a, "-s")) {
			do_shear = 1;
			if (i + 1 != argc && argv[i + 1][0] != '-') { shear_amt = atol(argv[++i]); if (shear_amt < 0) { printf("ERROR: bad shear length '%s'\n", argv[i]); return 1; } }
			if (!shear_amt) do_shear = 0;
		}
		else if (!strcmp(a, "--unique") || !strcmp(a, "-u")) dedupe = 1;
		else if (!strcmp(a, "--skipambig") || !strcmp(a, "-sa")) skip_ambig = 1;
		else if (!strcmp(a, "--noprogress")) { }
		else if (!strcmp(a, "--no-dupe-hunt")) rep_flags |= BH_REP_NO_DUPE_HUNT;   /* diagnostics: print every (hit, reference) expansion */
		else if (!strcmp(a, "--make-acx")) { NEEDARG("--make-acx"); mkacx_FN = argv[i]; }
		else if (!strcmp(a, "--device")) { NEEDARG("--device"); device = atoi(argv[i]); }
		else if (!strcmp(a, "--gpus")) { NEEDARG("--gpus"); n_gpus = atoi(argv[i]); n_gpus_given = 1; if (n_gpus < 1) { puts("ERROR: --gpus must be >= 1"); return 1; } }
		else if (!strcmp(a, "--devices")) {      /* explicit device of every rank, e.g. 0,1,2,3 (the same device twice only with --gather host) */
			NEEDARG("--devices");
			for (char *t = strtok(argv[i], ","); t && n_dev_list < BH_MAX_GPUS; t = strtok(NULL, ",")) dev_list[n_dev_list++] = atoi(t);
		}
		else if (!strcmp(a, "--gather")) { NEEDARG("--gather"); gather_host = !strcmp(argv[i], "host"); if (!gather_host && strcmp(argv[i], "rccl")) { puts("ERROR: --gather rccl|host"); return 1; } }
		else if (!strcmp(a, "--shard")) { NEEDARG("--shard"); shard_db = !strcmp(argv[i], "db"); if (!shard_db && strcmp(argv[i], "queries")) { puts("ERROR: --shard queries|db"); return 1; } }
		else if (!strcmp(a, "--shards")) { NEEDARG("--shards"); n_shards = atoi(argv[i]); if (n_shards < 1) { puts("ERROR: --shards must be >= 1"); return 1; } shard_db = 1; }
		else if (!strcmp(a, "--batch")) { NEEDARG("--batch"); batch = strtoull(argv[i], 0, 10); }
		else if (!strcmp(a, "-k")) { NEEDARG("-k"); K = atoi(argv[i]); if (K != 12 && K != 15) { puts("ERROR: -k must be 12 or 15"); return 1; } }
		else if (!strcmp(a, "--help") || !strcmp(a, "-h")) { usage(); return 1; }
		else if (!strcmp(a, "--taxonomy") || !strcmp(a, "-b")) {                       /* burst.c:4949-4954 */
			if (++i == argc || argv[i][0] == '-') { puts("ERROR: --taxonomy requires filename argument"); return 1; }
			tax_FN = argv[i];
			printf(" --> Assigning taxonomy based on mapping file: %s\n", tax_FN);
		}
		else if (!strcmp(a, "--taxacut") || !strcmp(a, "-bc")) {                        /* burst.c:5001-5014 */
			if (++i == argc || argv[i][0] == '-') { puts("ERROR: --taxacut requires numeric argument"); return 1; }
			int temp = atoi(argv[i]);
			if (temp < 2) { double fl = 1.0 / (1.0 - atof(argv[i])); temp = (int)(fl + 0.5); printf(" --> Taxacut: converting %s to %d...\n", argv[i], temp); }
			if (temp < 2) { fputs("ERROR: taxacut must be >= 2\n", stderr); return 1; }
			txo.taxacut = (uint32_t)temp;
			printf(" --> Ignoring 1/%d disagreeing taxonomy calls\n", temp);
		}
		else if (!strcmp(a, "--taxa_ncbi") || !strcmp(a, "-bn")) { txo.ncbi = 1; printf(" --> Using NCBI header formatting for taxonomy lookups\n"); }
		else if (!strcmp(a, "--taxasuppress") || !strcmp(a, "-bs")) {                   /* burst.c:5023-5031 */
			txo.suppress = 1;
			if (i + 1 != argc && argv[i + 1][0] != '-') {
				if (!strcmp(argv[++i], "STRICT")) txo.strict = 1;
				else { fprintf(stderr, "ERROR: Unrecognized taxasuppress '%s'\n", argv[i]); return 1; }
			}
			printf(" --> Surpressing taxonomic specificity by alignment identity%s\n", txo.strict ? " [STRICT]" : "");
		}
		else if (!strcmp(a, "--cache") || !strcmp(a, "-c")) {                           /* burst.c:5079-5084 */
			if (++i == argc || argv[i][0] == '-') { puts("ERROR: --cache requires integer argument"); return 1; }
			printf(" --> Cached matrix lines (%d) are a CPU-side optimisation; ignored on the device path\n", atoi(argv[i]));
		}
		else if (!strcmp(a, "--latency") || !strcmp(a, "-l")) {                         /* burst.c:5085-5090 */
			if (++i == argc || argv[i][0] == '-') { puts("ERROR: --latency requires integer argument"); return 1; }
			latency = (uint32_t)atoi(argv[i]);
			printf(" --> Setting clump formation latency to %d bases\n", atoi(argv[i]));
		}
		else if (!strcmp(a, "--clustradius") || !strcmp(a, "-cr") || !strcmp(a, "--dbpartition") || !strcmp(a, "-dp")) {
			printf("ERROR: option %s belongs to the compressive (-d DNA/RNA) database builder, which burst_hip does not include; use -d QUICK\n", a); return 1;
		}
		else if (!strcmp(a, "--xalphabet") || !strcmp(a, "-x")) { xalpha = 1; printf(" --> Allowing any alphabet (unambiguous ID matching)\n"); }
		else if (!strcmp(a, "--fingerprint") || !strcmp(a, "-f") || !strcmp(a, "--prepass") ||
		         !strcmp(a, "-p") || !strcmp(a, "--heuristic") || !strcmp(a, "-hr")) {
			printf("ERROR: option %s is outside the device hot path and not supported by burst_hip\n", a); return 1;
		}
		else { printf("ERROR: Unrecognized command-line option: %s\n", a); puts("See help by running with just '-h'"); return 1; }
	}
	if (n_dev_list && !n_gpus_given) { n_gpus = n_dev_list; n_gpus_given = 1; }
	if (n_dev_list && n_dev_list != n_gpus) { puts("ERROR: --devices must name one device per --gpus rank"); return 1; }
	if (mkacx_FN) {   /* (re)build an accelerator for an existing .edx:  burst_hip -r DB.edx --make-acx DB.acx [-k 12|15] [-y] */
		if (!ref_FN) { puts("ERROR: --make-acx needs -r DB.edx"); return 1; }
		BhDb db; int rc0;
		if ((rc0 = bh_edx_read(ref_FN, &db))) DIE(rc0);
		if ((rc0 = build_accelerator(&db, K ? K : 12, z, device, host_acx, skip_ambig))) DIE(rc0);
		if ((rc0 = bh_acx_write(&db, mkacx_FN))) DIE(rc0);
		printf("Accelerator written: K=%d, %s format, %u ambiguous clumps\n", db.K, db.acxFmt ? "LARGE" : "SMALL", db.badSz);
		bh_db_free(&db);
		return 0;
	}
	if (!ref_FN || !output_FN) { puts("ERROR: --references and --output are required"); return 1; }
	if (threads > 0) omp_set_num_threads(threads);
	FILE *output = fopen(output_FN, "wb");
	if (!output) { fprintf(stderr, "ERROR: Cannot open output: %s\n", output_FN); return 2; }
	const double start = wall();
	int rc;
	if (makedb && xalpha) { puts("ERROR: -x with -d: a database nibble-packs its symbols whatever the alphabet (burst.c:2810-2824): -x works against FASTA references only"); return 1; }
	if (makedb) {
		fclose(output);
		int e = bh_is_edx(ref_FN);
		if (e < 0) DIE(e);
		if (e) { fputs("ERROR: DBs can't make DBs.\n", stderr); return 1; }
		BhDb db;
		if (!do_shear) db_qlen = 0;                                             /* burst.c:5121 */
		if ((rc = bh_db_from_fasta_ex(ref_FN, (uint32_t)db_qlen, thres, do_shear, shear_amt, 1, latency, &db))) DIE(rc);
		puts("Writing database...");
		if ((rc = bh_edx_write(&db, output_FN, db_qlen, thres))) DIE(rc);
		printf("Database written: %u refs [%u orig], %u clumps, %u maxR\n", db.totR, db.origTotR, db.numRclumps, db.maxLenR);
		if (do_accel) {
			if (!K) K = 12;
			if (accel_dev || !xcel_FN) { puts("ERROR: -ad builds the accelerator at search time; give -a <name> to write one"); return 1; }
			printf("Generating accelerator '%s' (K=%d)\n", xcel_FN, K);
			if ((rc = build_accelerator(&db, K, z, device, host_acx, skip_ambig))) DIE(rc);
			if ((rc = bh_acx_write(&db, xcel_FN))) DIE(rc);
		}
		bh_db_free(&db);
		return 0;
	}
	if (!query_FN) { puts("ERROR: --queries is required when aligning"); return 1; }
	BhDb db; memset(&db, 0, sizeof db);
	double tp = wall();
	#define PHASE(name) do { const double t_ = wall(); printf(" [%-28s %8.3f s]\n", name, t_ - tp); tp = t_; } while (0)
	int usedb = bh_is_edx(ref_FN);
	if (usedb < 0) DIE(usedb);
	if (xalpha) {
		/* -x: raw symbols compared for equality (aded_xalpha / reScoreM_xalpha, burst.c:696-697, 894, 1099).  Upstream that only works
		 * against FASTA references (a database nibble-packs every symbol whatever the alphabet, burst.c:2810-2824) and on the forward
		 * strand (its reverse complement indexes a 16-entry table with the raw byte, burst.c:168, 3102): the same here, said aloud. */
		if (usedb) { fputs("ERROR: DB made without Xalpha; queries can't use Xalpha.\n", stderr); return 1; }      /* (the reference's message, burst.c:2859-2862) */
		if (do_rc) { puts("ERROR: -x compares raw symbols: there is no reverse complement of an arbitrary alphabet (drop -fr)"); return 1; }
		uint8_t map[256];
		int n_sym = 0;
		if ((rc = bh_alphabet_from_files(ref_FN, query_FN, map, &n_sym))) { fprintf(stderr, "%s\n", bh_last_error()); return rc == BH_E_IO ? 2 : 1; }
		printf(" --> Alphabet of %d symbols\n", n_sym);
		bh_set_alphabet(map);
	}
	BhQueries Q; memset(&Q, 0, sizeof Q);
	Ingest ing = {query_FN, thres, do_rc, incl_ws, do_accel, K ? K : (accel_dev || !do_accel ? 12 : 0), z, skip_ambig, &Q, 0, "", 0.0};
	pthread_t ing_thread; int ing_running = 0;
	bh_queries_sort_device(n_dev_list ? dev_list[0] : (n_gpus_given ? 0 : device));      /* large query files are sorted on the (first) search device */
	if (usedb && !getenv("BURST_HOST_SERIAL_INGEST")) ing_running = !pthread_create(&ing_thread, NULL, ingest_main, &ing);
	const int ing_started = ing_running;
	#define JOIN_INGEST() do { if (ing_running) { pthread_join(ing_thread, NULL); ing_running = 0; } } while (0)
	#define DIEJ(rc) do { JOIN_INGEST(); DIE(rc); } while (0)
	if (usedb) {
		puts("\nEDB database provided. Parsing...");
		if ((rc = bh_edx_read(ref_FN, &db))) DIEJ(rc);
		if (db.xalpha) { JOIN_INGEST(); fputs("ERROR: DB made with Xalpha; queries can't use Xalpha.\n", stderr); return 1; }
		printf(" --> EDB: %u refs [%u orig], %u clumps, %u maxR\n", db.totR, db.origTotR, db.numRclumps, db.maxLenR);
	}
	if (do_accel && accel_dev) {
		if (!usedb) { fputs("ERROR: an accelerator needs an .edx database\n", stderr); return 1; }
		if (!K) K = 12;
		printf(" --> [Accel] K=%d, built on the device from the database\n", K);
	} else if (do_accel) {
		if (!usedb) { fputs("ERROR: an accelerator needs an .edx database\n", stderr); return 1; }
		if ((rc = bh_acx_read(xcel_FN, K, z, &db))) DIEJ(rc);      /* K = 0: 12 or 15, whichever the file's exact size says */
		K = db.K;
		printf(" --> [Accel] K=%d, %s format, %u ambiguous clumps\n", K, db.acxFmt ? "LARGE" : "SMALL", db.badSz);
	}
	if (tax_FN) {                                                                    /* burst.c:5142-5149 */
		if ((rc = bh_tax_load(tax_FN, &taxonomy))) DIEJ(rc);
		txo.tax = &taxonomy;
	}
	PHASE("database read");
	if (!usedb) {      /* direct FASTA references: their clumps depend on the longest query (burst.c:5151), the queries come first */
		ingest_main(&ing);
		if (ing.rc) { fprintf(stderr, "%s\n", ing.err); return code_to_exit(ing.rc); }
		printf("Parsed %lu queries, %lu unique [min %u, max %u, maxED %u]; clear %lu, ambiguous %lu, bad %lu\n", (unsigned long)Q.totQ,
		       (unsigned long)Q.numUniq, Q.minLen, Q.maxLen, Q.maxED, (unsigned long)Q.nClear, (unsigned long)Q.nAmbig, (unsigned long)Q.nBad);
		PHASE("queries parsed, sorted");
		if ((rc = bh_db_from_fasta_ex(ref_FN, Q.maxLen, thres, do_shear, shear_amt, dedupe, latency, &db))) DIE(rc);
		printf("There are %u references and hence %u clumps\n", db.totR, db.numRclumps);
	}
	/* Multi-GPU (--gpus N): one host thread and one device handle per GPU (bh_search_multi, bh_multi.c).  --shard queries (default):
	 * the database replicated, unique queries [r U / N, (r+1) U / N) on rank r (a forward entry and its reverse complement stay
	 * together); --shard db: every rank holds a range of the database's clumps and aligns all queries, the per-query minimum is
	 * combined over the ranks.  Then the records meet where the reference's consolidation (incl. CAPITALIST's global vote) runs,
	 * in host memory.  The ranks are threads of this process and every rank's records are in host memory already when its last
	 * batch ends (copied behind each batch over the rank's own PCIe link), so the default hand-over is a concatenation there
	 * (--gather host).  --gather rccl takes them over RCCL / xGMI to rank 0's device first (bhip_comm_gather_hits: ncclAllGather of
	 * the counts + grouped ncclSend / ncclRecv; minima by ncclAllReduce MIN) -- N shares through rank 0's one PCIe link: the path
	 * for ranks that cannot see each other's memory.  --gpus 1 --gather rccl takes the RCCL path with one rank. */
	if (n_gpus > BH_MAX_GPUS) { printf("ERROR: --gpus %d (max %d)\n", n_gpus, BH_MAX_GPUS); return 1; }
	void *hhs[BH_MAX_GPUS]; int rcs[BH_MAX_GPUS]; char errs[BH_MAX_GPUS][512];
	BhMultiRank ranks[BH_MAX_GPUS]; BhDb slices[BH_MAX_GPUS]; uint64_t ru0[BH_MAX_GPUS], ru1[BH_MAX_GPUS];
	memset(hhs, 0, sizeof hhs); memset(ranks, 0, sizeof ranks); memset(slices, 0, sizeof slices);
	for (int r = 0; r < BH_MAX_GPUS; ++r) rcs[r] = BH_E_INTERNAL;
	void *comm = NULL;
	const int use_rccl = n_gpus_given && !gather_host;
	if (shard_db && !n_shards) n_shards = n_gpus;
	if (!shard_db || n_shards < 2) { shard_db = 0; n_shards = 1; }
	if (shard_db && n_gpus == 1 && n_shards > 1 && !use_rccl) {
		/* more shards than devices: the shards take turns on the one device (bh_search_serial_shards) -- a database larger than the
		 * device's memory, at the price of one upload per shard */
		if ((uint32_t)n_shards > db.numRclumps) { puts("ERROR: more database shards than clumps"); return 1; }
		if (n_shards > BH_MAX_GPUS) { printf("ERROR: --shards %d (max %d)\n", n_shards, BH_MAX_GPUS); return 1; }
		if (!n_dev_list) dev_list[0] = device;
		JOIN_INGEST();
		if (!ing_started && usedb) ingest_main(&ing);
		if (ing.rc) { fprintf(stderr, "%s\n", ing.err); return code_to_exit(ing.rc); }
		if (usedb && do_accel && !ing.K) bh_queries_bins(&Q, do_accel, K, z);
		printf("Parsed %lu queries, %lu unique [min %u, max %u, maxED %u]; clear %lu, ambiguous %lu, bad %lu\n", (unsigned long)Q.totQ,
		       (unsigned long)Q.numUniq, Q.minLen, Q.maxLen, Q.maxED, (unsigned long)Q.nClear, (unsigned long)Q.nAmbig, (unsigned long)Q.nBad);
		if (db.shear && (uint32_t)(Q.maxLen / thres) > db.shear) { fputs("ERROR: DB incompatible with selected queries/identity.\n", stderr); return 1; }
		bh_queries_pin(&Q);
		PHASE("queries parsed, page-locked");
		BhRun srun; memset(&srun, 0, sizeof srun);
		double secs[BH_MAX_GPUS], ups[BH_MAX_GPUS];
		const double ts0 = wall();
		if ((rc = bh_search_serial_shards(&db, dev_list[0], n_shards, z, accel_dev ? K : 0, &Q, mode, batch, &srun, secs, ups))) { fprintf(stderr, "%s\n", bh_last_error()); return rc == BH_E_USAGE ? 1 : 4; }
		printf("serial shards: %d shard(s) taking turns on device %d; upload (+ accelerator build) per shard [s]:", n_shards, dev_list[0]);
		for (int r = 0; r < n_shards; ++r) printf(" %.3f", ups[r]);
		printf("; align phase per shard [s]:");
		for (int r = 0; r < n_shards; ++r) printf(" %.4f", secs[r]);
		printf("\n");
		printf("Search complete [%f s, %u batches, %lu candidate (query, clump) pairs, %lu hits]. Consolidating results...\n", wall() - ts0, srun.nBatches,
		       (unsigned long)srun.total.n_pairs, (unsigned long)srun.nHits);
		PHASE("search (all shards, uploads included)");
		uint64_t slines = 0;
		setvbuf(output, NULL, _IOFBF, 1 << 22);
		BhRunView sview; memset(&sview, 0, sizeof sview);
		sview.base = srun.hits; sview.n_runs = 1; sview.off[0] = 0; sview.n[0] = srun.nHits; sview.total = srun.nHits;
		if ((rc = bh_report_view(output, &db, &Q, &sview, mode, (do_accel ? 0 : BH_REP_MERGED_LIST) | rep_flags, tax_FN ? &txo : NULL, &slines))) DIE(rc);
		if (fclose(output)) { fprintf(stderr, "ERROR: write failed: %s\n", output_FN); return 2; }
		printf("Wrote %lu alignments\n", (unsigned long)slines);
		PHASE("consolidation, output");
		printf("\nAlignment time: %f seconds\n", wall() - start);
		fflush(NULL);
		_exit(0);
	}
	if (shard_db && n_gpus == 1 && n_shards > 1) { printf("ERROR: --gpus 1 --shards %d takes the shards in turns on one device, which has no exchange: drop --gather rccl\n", n_shards); return 1; }
	if (n_gpus % n_shards) { printf("ERROR: --shards %d does not divide --gpus %d\n", n_shards, n_gpus); return 1; }
	const int n_groups = n_gpus / n_shards;
	if (shard_db && (uint32_t)n_shards > db.numRclumps) { puts("ERROR: more database shards than clumps"); return 1; }
	if (!n_dev_list) for (int r = 0; r < n_gpus; ++r) dev_list[r] = n_gpus_given ? r : device;
	if (use_rccl && bhip_comm_create(n_gpus, dev_list, &comm)) { fprintf(stderr, "libburst_hip: %s\n", bhip_last_error()); return 4; }
	int shared_dev = 0;
	for (int a = 0; a < n_gpus; ++a) for (int b = a + 1; b < n_gpus; ++b) shared_dev |= dev_list[a] == dev_list[b];
	omp_set_dynamic(0);
	/* A replicated database whose accelerator is built on the devices (-ad): the ranks build it TOGETHER, each the lists of 1/N of the words,
	 * and complete each other's tables device to device (bhip_build_accelerator_shared + bhip_team_share; BURST_HIP_SOLO_BUILD=1: every
	 * rank builds the whole thing as before).  The gates of the devices are held here, around the ranks: ranks on one device meet inside. */
	void *team = NULL;
	const int coop = n_gpus > 1 && !shard_db && accel_dev && !db.hasAcx && !getenv("BURST_HIP_SOLO_BUILD");
	if (coop && bhip_team_create(n_gpus, &team)) { fprintf(stderr, "libburst_hip: %s\n", bhip_last_error()); return 4; }
	if (coop) {
		for (int a = 0; a < n_gpus; ++a) { int seen = 0; for (int b = 0; b < a; ++b) seen |= dev_list[a] == dev_list[b]; if (!seen) bh_device_gate(dev_list[a], 1); }
		printf("Accelerator: built by the %d ranks together (word ranges, device-to-device exchange)\n", n_gpus);
	}
	#pragma omp parallel num_threads(n_gpus)
	{
		const int r = omp_get_thread_num();
		const BhDb *part = &db;
		rcs[r] = BH_OK;
		if (coop) {
			if (omp_get_num_threads() != n_gpus) rcs[r] = BH_E_INTERNAL;      /* (no team: nobody enters the exchange) */
			else if ((rcs[r] = bh_device_open_shared(&db, dev_list[r], z, K, r, n_gpus, bhip_team_share, team, &hhs[r]))) snprintf(errs[r], sizeof errs[r], "%s", bh_last_error());
		} else {
		if (shard_db) {      /* this rank's clumps: a view of the clump area + (with an .acx) the lists restricted to it */
			uint32_t c0, c1;
			bh_clump_shard(&db, n_shards, r % n_shards, &c0, &c1);
			ranks[r].c0 = c0;
			if ((rcs[r] = bh_db_slice(&db, c0, c1, &slices[r]))) snprintf(errs[r], sizeof errs[r], "%s", bh_last_error());
			part = &slices[r];
		}
		/* ranks that share a device (--devices 0,0,...: diagnostics on a one-GPU machine) open their handles one after the other:
		 * the accelerator build sizes its scratch from the memory that is free when it starts */
		if (!rcs[r] && shared_dev) {
			#pragma omp critical (bh_device_open)
			if ((rcs[r] = bh_device_open_ex(part, dev_list[r], z, accel_dev ? K : 0, &hhs[r]))) snprintf(errs[r], sizeof errs[r], "%s", bh_last_error());
		} else if (!rcs[r] && (rcs[r] = bh_device_open_ex(part, dev_list[r], z, accel_dev ? K : 0, &hhs[r]))) snprintf(errs[r], sizeof errs[r], "%s", bh_last_error());
		}
	}
	if (coop) for (int a = 0; a < n_gpus; ++a) { int seen = 0; for (int b = 0; b < a; ++b) seen |= dev_list[a] == dev_list[b]; if (!seen) bh_device_gate(dev_list[a], 0); }
	if (team) bhip_team_destroy(team);
	for (int r = 0; r < n_gpus; ++r) if (rcs[r]) { fprintf(stderr, "%s\n", rcs[r] == BH_E_INTERNAL ? "OpenMP did not start one host thread per GPU" : errs[r]); return 4; }
	for (int r = 0; r < n_gpus; ++r) { char nm[256]; int ncu = 0; uint64_t hbm = 0; if (!bhip_device_info(hhs[r], nm, sizeof nm, &ncu, &hbm)) printf("Device %d: %s, %d CUs, %.0f GiB\n", dev_list[r], nm, ncu, hbm / 1073741824.0); }
	if (shard_db) for (int r = 0; r < n_gpus; ++r) printf("Rank %d: replica group %d, clumps [%u, %u)\n", r, r / n_shards, ranks[r].c0, ranks[r].c0 + slices[r].numRclumps);
	PHASE("device database upload");
	if (usedb) {
		JOIN_INGEST();
		if (!ing_started) ingest_main(&ing);      /* (no thread: read them now) */
		if (ing.rc) { fprintf(stderr, "%s\n", ing.err); return code_to_exit(ing.rc); }
		if (do_accel && !ing.K) bh_queries_bins(&Q, do_accel, K, z);        /* K came with the accelerator file */
		printf("Parsed %lu queries, %lu unique [min %u, max %u, maxED %u]; clear %lu, ambiguous %lu, bad %lu\n", (unsigned long)Q.totQ,
		       (unsigned long)Q.numUniq, Q.minLen, Q.maxLen, Q.maxED, (unsigned long)Q.nClear, (unsigned long)Q.nAmbig, (unsigned long)Q.nBad);
		if (db.shear && (uint32_t)(Q.maxLen / thres) > db.shear) {                /* burst.c:5152-5156 */
			fputs("ERROR: DB incompatible with selected queries/identity.\n", stderr); return 1;
		}
		printf(" [%-28s %8.3f s%s; waited for %.3f s]\n", "queries parsed, sorted", ing.secs, ing_started ? ", on a thread beside the database phases" : "", wall() - tp);
		tp = wall();
	}
	bh_queries_pin(&Q);
	PHASE("query arrays page-locked");
	for (int r = 0; r < n_gpus; ++r) {
		ranks[r].rank = r; ranks[r].hh = hhs[r];
		const int grp = r / n_shards;      /* the ranks of a replica group align the same queries, each against its shard */
		ru0[r] = Q.numUniq * (uint64_t)grp / (uint64_t)n_groups; ru1[r] = Q.numUniq * (uint64_t)(grp + 1) / (uint64_t)n_groups;
		ranks[r].r0 = &ru0[r]; ranks[r].r1 = &ru1[r]; ranks[r].n_ranges = 1;
	}
	BhRun block; memset(&block, 0, sizeof block);
	if (n_gpus > 1 && !use_rccl && !shard_db) {
		/* the ranks' record buffers as slices of ONE page-locked block: what they deliver is then read where it lies (a view over the
		 * block, bh_report_view) instead of being concatenated first */
		uint64_t off[BH_MAX_GPUS + 1]; off[0] = 0;
		const uint64_t strands = Q.numEntries > Q.numUniq ? 2 : 1;
		for (int r = 0; r < n_gpus; ++r) { const uint64_t n = ru1[r] - ru0[r]; off[r + 1] = off[r] + n * strands + n / 2 + (1u << 20); }
		if (!bh_run_reserve(&block, off[n_gpus])) for (int r = 0; r < n_gpus; ++r) { ranks[r].run.hits = block.hits + off[r]; ranks[r].run.capHits = off[r + 1] - off[r]; ranks[r].run.hitsPinned = 2; }
	}
	{	/* device and record buffers for the batches to come (allocations synchronise the device: not inside the search), sized from
		 * the batches that will really be staged: their entry count and their symbols (one long read among millions of short ones
		 * must not size every buffer for n_entries x max_len) */
		const uint64_t strands = Q.numEntries > Q.numUniq ? 2 : 1;
		#pragma omp parallel num_threads(n_gpus)
		{
			const int r = omp_get_thread_num();
			const uint64_t n = ru1[r] - ru0[r], B = n < batch ? n : batch;
			uint64_t sym = 0;
			for (uint64_t u = ru0[r]; u < ru1[r]; u += B ? B : 1) {
				const uint64_t e = u + B < ru1[r] ? u + B : ru1[r];
				uint64_t sy = Q.qoff[e] - Q.qoff[u];
				if (strands == 2) sy += Q.qoff[Q.numUniq + e] - Q.qoff[Q.numUniq + u];
				if (sy > sym) sym = sy;
			}
			if (!ranks[r].run.hits) bh_run_reserve(&ranks[r].run, n * strands + n / 2 + (1u << 20));      /* (no slice of the block below) */
			if (B && bhip_reserve_symbols(hhs[r], (uint32_t)(B * strands), Q.maxLen, sym))      /* last: it ends with a warm-up pass, the search follows at once */
				fprintf(stderr, " --> WARNING: batch buffers not reserved on device %d (%s); they are allocated batch by batch\n", dev_list[r], bhip_last_error());
		}
	}
	BhRun run; memset(&run, 0, sizeof run);
	BhRunView view; memset(&view, 0, sizeof view);
	/* several ranks: the buffer their records meet in, made here and not inside the search (page-locked when a device copy lands
	 * in it, i.e. with the RCCL gather; bh_search_multi grows it if the job brings more) */
	if ((n_gpus > 1 || use_rccl) && !block.hits) { const uint64_t cap = Q.numEntries + Q.numEntries / 2 + (1u << 20); if (use_rccl) bh_run_reserve(&run, cap); else bh_run_reserve_plain(&run, cap); }
	PHASE("batch buffers");
	const double t0 = wall();
	uint64_t cnts[BH_MAX_GPUS]; memset(cnts, 0, sizeof cnts);
	if (n_gpus == 1 && !use_rccl) {
		if ((rc = bh_align_ranges_reuse(hhs[0], &Q, &ru0[0], &ru1[0], 1, mode, batch, &ranks[0].run))) { fprintf(stderr, "%s\n", bh_last_error()); return rc == BH_E_USAGE ? 1 : 4; }
		run = ranks[0].run; memset(&ranks[0].run, 0, sizeof ranks[0].run);
		view.base = run.hits; view.n_runs = 1; view.off[0] = 0; view.n[0] = run.nHits; view.total = run.nHits;
	} else {
		if ((rc = bh_search_multi_ex(ranks, n_gpus, n_gpus, comm, NULL, &Q, mode, batch, shard_db ? n_shards : 0, &run, cnts, &view))) { fprintf(stderr, "%s\n", bh_last_error()); return rc == BH_E_USAGE ? 1 : 4; }
		printf("%s: %d rank(s)%s, records per rank:", use_rccl ? "RCCL gather" : "host gather", n_gpus, shard_db ? ", database-sharded" : "");
		for (int r = 0; r < n_gpus; ++r) printf(" %lu", (unsigned long)cnts[r]);
		printf("; align phase per rank [s]:");
		for (int r = 0; r < n_gpus; ++r) printf(" %.4f", ranks[r].secSearch);
		printf("\n");
		for (int r = 0; r < n_gpus; ++r) run.total.n_pairs += ranks[r].run.total.n_pairs;
	}
	const double t1 = wall();
	printf("Search complete [%f s, %u batches, %lu candidate (query, clump) pairs, %lu hits]. Consolidating results...\n", t1 - t0, run.nBatches,
	       (unsigned long)run.total.n_pairs, (unsigned long)run.nHits);
	PHASE("search (all batches)");
	uint64_t lines = 0;
	setvbuf(output, NULL, _IOFBF, 1 << 22);
	if ((rc = bh_report_view(output, &db, &Q, &view, mode, (do_accel ? 0 : BH_REP_MERGED_LIST) | rep_flags, tax_FN ? &txo : NULL, &lines))) DIE(rc);
	if (fclose(output)) { fprintf(stderr, "ERROR: write failed: %s\n", output_FN); return 2; }      /* (the last buffer of the report: a full disk must not end in exit code 0) */
	printf("Wrote %lu alignments\n", (unsigned long)lines);
	PHASE("consolidation, output");
	printf("\nAlignment time: %f seconds\n", wall() - start);
	fflush(NULL);
	/* The job is done and its output is on disk.  Unpinning and unmapping tens of gigabytes one table after the other took about a
	 * second for a 32 M-read job; the operating system releases a finished process's memory (host and device) far faster.
	 * BURST_HOST_TEARDOWN=1 walks through the orderly release instead (leak checks). */
	if (!getenv("BURST_HOST_TEARDOWN")) _exit(0);
	if (comm) bhip_comm_destroy(comm);
	for (int r = 0; r < n_gpus; ++r) { bhip_destroy(hhs[r]); if (slices[r].numRclumps) bh_db_free(&slices[r]); }
	for (int r = 0; r < n_gpus; ++r) bh_run_free(&ranks[r].run);
	bh_run_free(&block); bh_run_free(&run); bh_queries_free(&Q); bh_db_free(&db); bh_tax_free(&taxonomy);
	return 0;
}
