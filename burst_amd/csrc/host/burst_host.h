/* burst_amd/csrc/host/burst_host.h -- C host side of the alignment path (libburst_host.so + the burst_hip CLI).
 *
 * Mirrors, in plain C, the parts of the reference's single translation unit that sit either side of the
 * kernels: query pipeline (process_queries, burst.c:2980-3223), database readers (read_edb 2842-2975,
 * read_accelerator 3535-3594), direct-FASTA clumping (process_references QUICK path 1840-1858, 2109-2190,
 * 2687-2741), the batch scheduler that replaces the OpenMP loops of do_alignments (4018-4488) by calls into
 * libburst_hip.so, and the per-mode consolidation + .b6 writer (4490-4891).
 * Errors are return codes (<0) with bh_last_error(); nothing here calls exit().
 */
#ifndef BURST_HOST_H
#define BURST_HOST_H
#include <stdint.h>
#include <stdio.h>
#include "burst_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

#define BH_OK        0
#define BH_E_USAGE  -1   /* reference exit(1): usage / format */
#define BH_E_IO     -2   /* reference exit(2): cannot open */
#define BH_E_OOM    -3   /* reference exit(3) */
#define BH_E_INTERNAL -4
#define BH_E_DEVICE -5
#define BH_E_CAPACITY -6   /* not an error of the job: the caller has to take the other path (bh_node_collect_view) */

typedef enum { BH_FORAGE = 0, BH_BEST, BH_ALLPATHS, BH_CAPITALIST, BH_ANY } BhMode;   /* burst.c:75-78 */

/* ---- tables (burst.c:164-192, 1237-1329) ---- */
void bh_score_lut(int z, uint8_t lut[256]);        /* SCOREFAST after setScore(): lut[16*q+r] in {0,1,255} */
void bh_char2code(uint8_t map[256]);               /* CHAR2NUM, bytes >= 128 map to 0 */
void bh_set_alphabet(const uint8_t map[256]);      /* -x: byte -> code 1..15 of a caller-made alphabet (NULL: back to CHAR2NUM); bh_score_lut is then the identity table */
int  bh_alphabet_is_set(void);
int  bh_alphabet_from_files(const char *ref_fa, const char *query_fa, uint8_t map[256], int *n_symbols);      /* BH_E_USAGE beyond 15 symbols */
uint8_t bh_rc_code(uint8_t c);                     /* RVT */
uint32_t bh_error_budget(float thres, uint32_t len);   /* burst.c:3069-3076 */

/* ---- database ---- */
typedef struct BhDb {
	/* .edx (burst.c:2842-2975) */
	int rebase, xalpha;
	uint32_t shear, totR, origTotR, numRclumps, maxLenR, numRefHeads;
	char *headDump;          /* NUL-separated unique headers */
	char **refHead;          /* [origTotR] header of each sheared reference (through RefMap) */
	uint32_t *refMap;        /* [origTotR] sheared ref -> unique header index */
	uint32_t *refStart;      /* [origTotR] or NULL */
	uint32_t *refDedupIx;    /* [totR+1] or NULL */
	uint32_t *tmpRIX;        /* [origTotR] */
	uint32_t *refIxSrt;      /* [totR] = TmpRIX[RefDedupIx[i]] or TmpRIX (burst.c:3688-3693) */
	uint32_t *clumpLen;      /* [numRclumps] */
	uint8_t  *packed;        /* clump area, 16-byte words */
	uint64_t packedWords;
	/* .acx (burst.c:3535-3594), optional */
	int hasAcx, K, acxFmt, acxZ;
	uint32_t *acxLens;       /* [4^K] */
	uint8_t  *acxLists; uint64_t acxListBytes;
	uint32_t *badList; uint32_t badSz;
	/* bookkeeping */
	int identityMap;         /* direct-FASTA runs: RefMap is the identity (burst.c:4545-4551) */
	void *owned[32]; int nOwned;
	void *mapBase; uint64_t mapLen;      /* the clump area of a large .edx is a read-only mapping of the file (bh_edx_read): the processes of a node share one copy */
} BhDb;

int  bh_is_edx(const char *path);                       /* burst.c:4894-4901: first byte has bit 7 set; <0 on IO error */
int  bh_edx_read(const char *path, BhDb *db);
int  bh_acx_read(const char *path, int K, int z, BhDb *db);
/* direct -r fasta (no DB): QUICK pipeline; shear_len = 0 disables shearing (burst.c:5046-5053) */
int  bh_db_from_fasta(const char *path, uint32_t maxLenQ, float thres, int do_shear, long shear_len, int dedupe, BhDb *db);
/* DB construction (tooling for tests/bench) */
int  bh_edx_write(const BhDb *db, const char *path, long db_qlen, float thres);
/* latency = clump formation tolerance, the reference's `-l` (LATENCY, burst.c:83): 16 in bh_db_from_fasta, 0 = input order */
int  bh_db_from_fasta_ex(const char *path, uint32_t maxLenQ, float thres, int do_shear, long shear_len, int dedupe, uint32_t latency, BhDb *db);
/* n .edx files laid end to end into one (every part but the last must fill its last clump; no duplicate-fragment tables): a
 * database too large to be built in one piece in the memory at hand, built part by part */
int  bh_edx_merge(const char *const *paths, int n, const char *out_path);
int  bh_acx_build(BhDb *db, int K, int z);
/* skip_ambig = -sa: leave out every word that holds an ambiguous symbol (burst.c:3360-3366) */
int  bh_acx_build_ex(BhDb *db, int K, int z, int skip_ambig);
/* the accelerator tables of `db` from a device handle that holds this database's accelerator (bhip_acx_export), packed as the
 * reference writes them (burst.c:3501-3530): with a handle whose accelerator was built on the device this is make_accelerator
 * without the host pass */
int  bh_acx_from_device(BhDb *db, void *hip_handle, int K, int z);
/* the same tables written to an .acx file without being held on the host (the list area is streamed from the device run by run) */
int  bh_acx_write_from_device(const BhDb *db, void *hip_handle, int K, int z, const char *path);
/* view of the clumps [c0, c1) with the accelerator restricted to them (database sharding); `db` must outlive the view */
int  bh_db_slice(const BhDb *db, uint32_t c0, uint32_t c1, BhDb *out);
int  bh_acx_write(const BhDb *db, const char *path);
void bh_db_free(BhDb *db);

/* ---- queries (burst.c:2980-3223) ---- */
typedef struct BhQueries {
	uint64_t totQ, numUniq, numEntries;   /* entries = unique forward queries, then (with -fr) their reverse complements */
	char *dump;            /* file contents; heads point into it */
	char **heads;          /* [totQ] headers in sorted-sequence order (QHead after burst.c:3055-3060) */
	uint64_t *offset;      /* [numUniq+1] Offset: reads of unique query i are heads[offset[i]..offset[i+1]) */
	uint8_t *codes;        /* concatenated symbol codes of all entries */
	uint8_t *codes4;       /* the same, two symbols per byte (low nibble first): what the device batches are copied from */
	uint8_t *codes2;       /* the same, four symbols per byte (code - 1 in two bits; symbols beyond A/C/G/T read as A): batches without such symbols are copied from here */
	uint16_t *len16;       /* [numUniq] ShrBin.len again, as the 2-byte lengths the device batches carry instead of 8-byte offsets */
	uint32_t *ambBefore;   /* [numUniq+1] unique queries in front of i that hold a symbol beyond A/C/G/T (a batch [u, u+B) is clean iff the counts at its ends agree) */
	uint64_t *qoff;        /* [numEntries+1] */
	uint32_t *six;         /* [numEntries] shared slot = unique query index */
	uint8_t *rc;           /* [numEntries] */
	uint8_t *flags;        /* [numEntries] BHIP_Q_PREFILTER / BHIP_Q_EXHAUSTIVE (query binning, burst.c:3113-3141) */
	uint16_t *emac;        /* [numEntries] budget of the entry's shared slot */
	uint32_t *len;         /* [numUniq] ShrBin.len */
	uint16_t *ed;          /* [numUniq] ShrBin.ed (initial budget) */
	uint32_t maxLen, minLen, maxED;
	uint64_t nClear, nAmbig, nBad;
	int pinned;            /* codes / qoff are page-locked (bh_queries_pin) */
} BhQueries;

/* device that sorts and de-duplicates large query files (default 0; < 0 = always on the host) */
void bh_queries_sort_device(int device);
void bh_device_gate(int device, int take);
int  bh_device_gate_try(int device);      /* bh_align.c: serialises the set-up steps that size themselves from a device's free memory */
int  bh_queries_load(const char *fasta, float thres, int do_rc, int incl_whitespace, int do_accel, int K, int z,
                     int skip_ambig, BhQueries *q);
/* prefilter / exhaustive route per entry and the clear / ambiguous / bad counts; bh_queries_load does it itself unless it was
 * called with do_accel and K = 0 (queries read beside the database, K not known yet) */
void bh_queries_bins(BhQueries *q, int do_accel, int K, int z);
void bh_queries_free(BhQueries *q);
/* page-lock the arrays the device batches are copied from, so that the copies of batch k+1 run beside the kernels of batch k
 * (optional; needs a device) */
int  bh_queries_pin(BhQueries *q);

/* ---- alignment driver: batches of entries through bhip_align_batch ---- */
typedef struct BhRun {
	BhipHit *hits; uint64_t nHits;        /* all records, q = entry index, sorted by (q, refIx) */
	double secAlign;                      /* wall time inside the device calls */
	BhipStats total;                      /* summed over batches (times in ms) */
	uint32_t nBatches;
	int hitsPinned;                       /* hits is page-locked memory of the device library */
	uint64_t capHits;                     /* records the buffer holds */
	/* optional: called after every batch with the device handle (whose records of that batch are still resident) and the batch's place in
	 * `hits` -- the RCCL gather of bh_search_multi_ex stages the records device to device instead of uploading the host copy again */
	void (*onBatch)(void *ctx, void *hip_handle, uint64_t first_record, uint64_t n_records);
	void *onBatchCtx;
} BhRun;
/* entry range [e0, e1) of unique queries [u0, u1): forward entries u0..u1-1 and (if numEntries > numUniq) their RC twins
 * are always sent together because they share the running minimum (burst.c:277-280, 4218). */
int  bh_align(void *hip_handle, const BhQueries *q, uint64_t u0, uint64_t u1, BhMode mode, uint64_t batch_uniq, BhRun *run);
/* the same over several ranges of unique queries, in the order given (a range is cut into batches of at most batch_uniq);
 * bh_align is the one-range case */
int  bh_align_ranges(void *hip_handle, const BhQueries *q, const uint64_t *u0, const uint64_t *u1, uint32_t n_ranges, BhMode mode, uint64_t batch_uniq, BhRun *run);
/* into a BhRun used before (or zeroed): keeps its page-locked record buffer */
int  bh_align_ranges_reuse(void *hip_handle, const BhQueries *q, const uint64_t *u0, const uint64_t *u1, uint32_t n_ranges, BhMode mode, uint64_t batch_uniq, BhRun *run);
int  bh_run_reserve(BhRun *run, uint64_t cap_records);
int  bh_run_reserve_plain(BhRun *run, uint64_t cap_records);      /* pageable memory */
void bh_run_free(BhRun *run);
/* ---- multi-GPU search (bh_multi.c) ---- */
#define BH_MAX_RANKS 16
typedef struct BhMultiRank {
	int rank;                     /* rank of the node-wide job */
	void *hh;                     /* its device handle */
	const uint64_t *r0, *r1; uint32_t n_ranges;   /* ranges of unique queries it aligns (database-sharded: normally one range, everything) */
	uint32_t c0;                  /* database-sharded: first clump of its slice (added to the records' reference numbers) */
	BhRun run;                    /* its own records (the page-locked buffer is reused between calls) */
	double secSearch;             /* out: wall time of this rank's align phase (before the minima / the gather) */
	int gatherPath;               /* out: how this rank's records reached rank 0 -- 0 no collective (host memory), 1 RCCL gather fed from the device
	                               * (bhip_comm_gather_staged), 2 RCCL gather of the host copy uploaded again (bhip_comm_gather_hits) */
	/* Optional back ends (NULL = the device path).  `align`: in place of the device scheduler bh_align_ranges_reuse(hh, ...) -- it
	 * fills `run` (bh_run_put) with the records of the ranges, q = entry index, sorted by (q, refIx); the CPU tests of the multi-rank
	 * search put the oracle here.  `reduce_min`: the element-wise minimum of `buf` over all ranks, in place, for ranks in different
	 * processes that have no library communicator (a launcher's own collective: torch.distributed all_reduce MIN). */
	int (*align)(void *ctx, const BhQueries *q, const uint64_t *u0, const uint64_t *u1, uint32_t n_ranges, int mode, uint64_t batch_uniq, BhRun *run);
	int (*reduce_min)(void *ctx, uint8_t *buf, uint64_t n);
	void *ctx;
} BhMultiRank;
/* n records into a run's buffer (grown when the run owns it; a shared-memory segment must be large enough) */
int  bh_run_put(BhRun *run, const BhipHit *hits, uint64_t n);
/* clump range of rank `rank` of `n_ranks`: contiguous, about the same number of reference columns each */
void bh_clump_shard(const BhDb *db, int n_ranks, int rank, uint32_t *c0, uint32_t *c1);
/* The search of the n_local ranks that live in this process (listed in rank order; one host thread each) as part of a job of
 * n_ranks: align, [database-sharded: combine the per-query minimum, drop what lies above it], gather the records to rank 0.
 * shard_db = number of database shards (0 / 1 = none: query-sharded; n_ranks = every rank its own shard; a divisor S of n_ranks =
 * n_ranks / S replica groups of S shards, rank = group * S + shard, the caller passes each rank its group's ranges and its shard's c0).
 * comm = communicator over all n_ranks (bhip_comm_create / bhip_comm_create_rank), or NULL when all ranks are local: records
 * and minima then meet in host memory.  all = the gathered records where rank 0 lives (sorted by (query entry, reference));
 * counts[n_ranks] (optional, rank 0's process) = records per rank. */
int  bh_search_multi(BhMultiRank *ranks, int n_local, int n_ranks, void *comm, const BhQueries *q, BhMode mode, uint64_t batch_uniq, int shard_db, BhRun *all, uint64_t *counts);
/* pieces of the database-sharded search, exported for the tests: records in any order -> (query entry, reference) order in place
 * ((entry, reference) pairs must be unique); element-wise minimum of n tables of len bytes into best[0] (NULL tables are skipped) */
int  bh_order_records(BhipHit *hits, uint64_t n, uint64_t n_entries);
void bh_minima_merge(uint8_t *const *best, int n, uint64_t len);
/* n_shards database shards taking turns on ONE device (a database larger than the device): every shard is uploaded, searched with all
 * queries and released; minima over the shards, filter, (query entry, reference) order -- the records of a single device holding
 * everything.  secs / up (n_shards each, may be NULL): align phase and upload (+ accelerator build with build_K) of every shard */
int  bh_search_serial_shards(const BhDb *db, int device, int n_shards, int z, int build_K, const BhQueries *Q, BhMode mode, uint64_t batch, BhRun *all, double *secs, double *up);
/* ---- ranks of one node in different processes: the records meet in shared memory (bh_node.c) ---- */
typedef struct BhNode BhNode;
/* job: a name all ranks of the job share and no other job on the machine, before or after, uses again (a peer that opens early would
 * map the segment a dead job of the same name left behind: the launchers draw 48 random bits per job); cap_records: what the rank expects to deliver per
 * search (a search that brings up to four times as many still fits).  Every rank opens; rank 0 first or at the same time. */
int  bh_node_open(const char *job, int rank, int n_ranks, uint64_t cap_records, BhNode **node);
void bh_node_close(BhNode *node);
void bh_node_attach(BhNode *node, BhRun *run);                        /* the rank's record buffer = its segment */
int  bh_node_begin(BhNode *node);                                     /* a new search: rank 0 has read the previous one's records */
int  bh_node_publish(BhNode *node, const BhRun *run, int status);     /* this rank's records of the search are complete */
int  bh_node_collect(BhNode *node, BhRun *all, uint64_t *counts);     /* rank 0: everybody's records, rank order */
/* records as runs of one address range: run r = base[off[r] .. off[r] + n[r]); what lies between the runs is not records */
typedef struct BhRunView { const BhipHit *base; int n_runs; uint64_t off[BH_MAX_RANKS], n[BH_MAX_RANKS]; uint64_t total; } BhRunView;
int  bh_node_collect_view(BhNode *node, BhRunView *view, uint64_t *counts);   /* rank 0, no copy; BH_E_CAPACITY = use bh_node_collect */
/* bh_search_multi with the hand-over through `node` instead of the communicator's gather (one rank per process: n_local = 1;
 * comm is then only needed for the minima of a database-sharded search) */
int  bh_search_multi_ex(BhMultiRank *ranks, int n_local, int n_ranks, void *comm, BhNode *node, const BhQueries *q, BhMode mode, uint64_t batch_uniq, int shard_db, BhRun *all, uint64_t *counts,
                        BhRunView *view);
/* view (optional; rank 0's process): how to read the result -- with a node and a query-sharded search the ranks' records where they
 * lie, no copy (all->nHits is their number, all->hits is NOT their place); otherwise the one run all->hits[0 .. all->nHits).
 * Valid until this rank's next search.  bh_report_view consumes it. */
int  bh_device_open(const BhDb *db, int device, int z, void **hip_handle);
/* build_K > 0 and a database without accelerator tables: the device builds the accelerator itself (no .acx file) */
int  bh_device_open_ex(const BhDb *db, int device, int z, int build_K, void **hip_handle);
/* the same for one of n_parts handles of a replicated database that build the accelerator together (bhip_build_accelerator_shared) */
int  bh_device_open_shared(const BhDb *db, int device, int z, int build_K, int part, int n_parts, bhip_share_fn share, void *ctx, void **hip_handle);

/* ---- consolidation and .b6 output (burst.c:4553-4891); hits must be sorted by (q, refIx) ---- */
int  bh_report(FILE *out, const BhDb *db, const BhQueries *q, const BhipHit *hits, uint64_t nHits, BhMode mode, uint64_t *nLines);
/* flags: BH_REP_MERGED_LIST = rebuild each query's hit list the way the reference's exhaustive path does (one list per
 * unique query shared by both strands, burst.c:4368, instead of forward list + appended reverse list, 4218/4299-4312);
 * BH_REP_NO_DUPE_HUNT = print every (hit, reference) expansion (diagnostics / tests). */
#define BH_REP_MERGED_LIST  1
#define BH_REP_NO_DUPE_HUNT 2
uint64_t bh_report_format_identities(const float *score, uint64_t n, char *out, uint64_t cap);      /* test entry: "%f" of score * 100 as the report prints it */
int  bh_report_ex(FILE *out, const BhDb *db, const BhQueries *q, const BhipHit *hits, uint64_t nHits, BhMode mode, int flags, uint64_t *nLines);

/* ---- taxonomy (column 13; parse_taxonomy burst.c:447-479, taxa_lookup 409-440, CAPITALIST interpolation 4781-4829, -bs 4820-4828) ---- */
typedef struct { uint64_t n; char **pair; char *blob; } BhTax;        /* pair[2i] = header, pair[2i+1] = taxonomy, sorted by header */
typedef struct {
	const BhTax *tax;     /* NULL = no taxonomy column */
	int suppress;         /* -bs: cut the taxonomy at the level the identity supports */
	int strict;           /* -bs STRICT */
	uint32_t taxacut;     /* -bc (default 10): 1/taxacut of the placements may disagree at a level (CAPITALIST) */
	int ncbi;             /* -bn: headers are '>xxx|accession.version...' */
} BhTaxOpts;
int  bh_tax_load(const char *file, BhTax *T);
void bh_tax_free(BhTax *T);
const char *bh_tax_lookup(const BhTax *T, const char *ref_header, int ncbi);
int  bh_report_tax(FILE *out, const BhDb *db, const BhQueries *q, const BhipHit *hits, uint64_t nHits, BhMode mode, int flags, const BhTaxOpts *tx, uint64_t *nLines);
/* the same over records that lie in several runs (the records of an entry inside one run, contiguous) */
int  bh_report_view(FILE *out, const BhDb *db, const BhQueries *q, const BhRunView *view, BhMode mode, int flags, const BhTaxOpts *tx, uint64_t *nLines);

const char *bh_last_error(void);
int bh_set_error(int code, const char *fmt, ...);

/* ---- synthetic data (tests / bench tooling; behaviour modelled on embalmlets/LLsim.c:175-231) ---- */
int bh_synth_refs(const char *fasta_out, uint32_t n_base, uint32_t n_variants, uint32_t length, double variant_rate, uint64_t seed);
int  bh_synth_refs_range(const char *fasta_out, uint32_t first_base, uint32_t n_base, uint32_t n_variants, uint32_t length, double rate, uint64_t seed);
int bh_synth_reads(const char *refs_fasta, const char *fasta_out, uint64_t n_reads, uint32_t read_len, const uint32_t *edit_choices,
                   uint32_t n_choices, int rc, double iupac_rate, uint64_t seed);
int  bh_synth_reads_ex(const char *refs_fasta, const char *fasta_out, uint64_t n_reads, uint32_t read_len, const uint32_t *edit_choices,
                       uint32_t n_choices, int rc, double iupac_rate, uint64_t seed, uint64_t first_read, int append);

#ifdef __cplusplus
}
#endif
#endif
