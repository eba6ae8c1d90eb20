/* include/burst_hip.h -- C ABI of libburst_hip.so, the MI355X (gfx950) device side of the BURST
 * alignment hot path.  Plain pointers and sizes only; no C++ or torch types cross this line.
 *
 * The reference (knights-lab/BURST, burst.c @ 2024_08_07) has no plugin/FFI interface; what this
 * library replaces are the four kernel call sites inside do_alignments() and the k-mer scour block:
 *
 *   aded_mat16L(...)                burst.c:4215   \  banded edit distance of one query vs one 16-lane clump
 *   aded_mat16 / aded_xalpha(...)   burst.c:4423-4426 /  -> bhip_align_batch (kernel "myers"), bhip_align_pairs
 *   reScoreM_mat16 / _xalpha(...)   burst.c:4226, 4439-4442  -> bhip_align_batch (kernel "rescore")
 *   qsort + postScour + selection   burst.c:4096-4133, 3238-3282 -> bhip_align_batch (kernel "prefilter"),
 *                                                                   bhip_prefilter
 *   4-bit clump unpack              burst.c:4141-4150          -> folded into bhip_init (device layout)
 *
 * Conventions mirror those call sites: the caller owns every host buffer, the library owns device
 * memory, failures are return codes (never exit()), a handle is single-threaded, one handle per device.
 * Return values: 0 = ok, negative = BHIP_E_*; bhip_last_error() gives the text.
 */
#ifndef BURST_HIP_H
#define BURST_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* libburst_hip.so is built with -fvisibility=hidden: the entry points below are its whole dynamic symbol table */
#define BHIP_API __attribute__((visibility("default")))

#define BHIP_OK            0
#define BHIP_E_ARG        -1   /* invalid argument */
#define BHIP_E_DEVICE     -2   /* HIP runtime error (no device, OOM, launch failure) */
#define BHIP_E_CAPACITY   -3   /* caller's hit buffer too small: *n_hits holds the required count */
#define BHIP_E_QUERYLEN   -4   /* query longer than BHIP_MAX_QLEN */
#define BHIP_E_INTERNAL   -5
#define BHIP_E_RESCORE    -6   /* the re-scorer cannot reproduce a hit of the search: the state in which the reference prints "CRITICAL ERROR:
                                 Truncation within known good path" and exits (burst.c:812-816) -- a query whose FIRST symbol has code 0
                                 (row 1 of reScoreM takes the substitution cost alone, burst.c:722-739, while the search may gap it) */

#define BHIP_MAX_QLEN   4095   /* up to 1 024 symbols: register bit-vector kernels (2 .. 32 words); beyond: k_myers_long (vector in LDS, single-stage sweep) */

/* One alignment result = ResultPod (burst.c:3998-4004) minus the list pointer, plus the query index. */
typedef struct BhipHit {
	uint32_t q;         /* index of the query entry in the batch (forward or reverse-complement entry) */
	uint32_t refIx;     /* 16*clump + lane, burst.c:4234 */
	uint32_t finalPos;  /* MetaPack.finalPos, burst.c:862-883: 1-based end column in the sheared reference */
	float    score;     /* MetaPack.score, burst.c:844-860: 1 - ed/(len + numGapQ), IEEE f32 division */
	uint8_t  ed;        /* ResultPod.mismatches = MinA[lane], burst.c:4232 */
	uint8_t  gapR;      /* numGapR */
	uint8_t  gapQ;      /* numGapQ */
	uint8_t  rc;        /* strand of the query entry, burst.c:4238 */
} BhipHit;              /* 20 bytes */

/* Query flags (q_flags[i]) */
#define BHIP_Q_PREFILTER   0   /* candidate clumps come from the .acx prefilter + BadList (burst.c:4078-4284) */
#define BHIP_Q_EXHAUSTIVE  1   /* align against every clump (burst.c:4320-4488: no -a, or "bad"-bin queries) */

/* Counters and timings of the most recent bhip_align_batch / bhip_align_pairs call on a handle.
 * Times are HIP-event milliseconds measured on the library's own stream. */
typedef struct BhipStats {
	uint64_t n_queries;        /* query entries in the call */
	uint64_t n_pairs;          /* (query, clump) units of work aligned */
	uint64_t n_columns;        /* sum over clump-level pairs of ClumpLen: DP columns swept (x16 lanes x qlen rows of cells); lane tasks: n_task_columns */
	uint64_t n_raw_hits;       /* (query, ref) lanes with ed <= budget */
	uint64_t n_hits;           /* records returned */
	uint64_t acx_entries_read; /* list entries gathered by the prefilter */
	uint64_t bytes_algorithmic;/* per-launch algorithmic bytes of the myers kernel */
	uint64_t n_windows;        /* reference lanes that passed the prefix stage of the two-stage edit distance */
	uint64_t n_window_columns; /* columns swept by the full-length stage over those windows */
	uint64_t n_lane_tasks;     /* (query, reference lane) tasks emitted by the lane-resolved prefilter (0 = clump-level path) */
	uint64_t n_task_columns;   /* columns swept by the prefix stage over those tasks */
	uint64_t n_seed_words;     /* sampled query words looked up in the accelerator by the lane-resolved prefilter */
	float ms_h2d, ms_prefilter, ms_peq, ms_myers, ms_rescore, ms_d2h, ms_total;
	float ms_myers_prefix;     /* part of ms_myers spent in k_myers_prefix (the dominant kernel when the two-stage path runs) */
	float ms_myers_window;     /* part of ms_myers spent in k_myers_window */
	float ms_prefilter_hash;   /* part of ms_prefilter spent in k_prefilter_mask (HIP events around that kernel alone) */
	float ms_seed;             /* part of ms_prefilter spent in k_seed_ranges */
	float ms_stage_copy;       /* of ms_h2d (the staging of this batch, on the staging stream, two batches ahead): the copies over PCIe with the unpack kernels and offset scans between them */
	float ms_stage_route;      /* of ms_h2d: k_pack_queries, k_route and the radix sort of the entry lists behind the copies (kernels that share the CUs with the batch being aligned) */
	uint32_t myers_launches;   /* launches of the column-sweeping kernel (k_myers_prefix, or k_myers on the one-stage path) */
	uint32_t prefix_words;     /* NWP of the last launch, 0 = one-stage path */
	uint32_t prefilter_launches; /* launches of the lane-resolved prefilter kernel */
	uint32_t prefilter_algo;   /* kernel of the last such launch: 0 = k_prefilter_cf (counting filter), 1 = k_prefilter_mask (exact hash) */
} BhipStats;

/* Upload a database to device `device` and create a handle.
 *   edx_packed : the clump area of an .edx exactly as on disk (burst.c:2810-2824): for clump c,
 *                ceil(clump_len[c]/2) 16-byte words; byte k of a word = lane k; low nibble = even position.
 *   clump_len  : ClumpLen[n_clumps] (burst.c:2918);  tot_refs = totR (lanes >= tot_refs never hit, burst.c:4229)
 *   acx_lens   : Lens[4^K] of the .acx (burst.c:3558), or NULL: with K = 0 no accelerator is used, with K in 4..15 the accelerator
 *                is BUILT ON THE DEVICE from the references alone (what make_accelerator does, burst.c:3304-3532: same entries in
 *                the same order, same BadList; acx_lists / acx_fmt / badlist are then ignored) -- no .acx file, no upload
 *   acx_lists  : the packed list area (burst.c:3569); acx_fmt 0 = SMALL 20-bit pairs (3265-3274), 1 = LARGE 24-bit (3245-3248)
 *   badlist    : BadList[n_bad] (burst.c:3571): clumps every prefiltered query must be aligned against
 *   score_lut  : 16x16 cost table lut[16*q + r] in {0,1,255} = SCOREFAST after setScore() (burst.c:1310-1328)
 *   xalpha     : must be 0.  The alphabet-agnostic mode (-x; aded_xalpha / reScoreM_xalpha, burst.c:696-697, 894, 1099) needs no device
 *                switch: the host maps the run's alphabet (up to 15 symbols) onto the codes 1..15 and passes the IDENTITY cost table
 *                (0 iff equal, 1 otherwise, the padding code 0 included) as score_lut -- burst_hip -x, host/main.c
 */
BHIP_API int bhip_init(int device, const void *edx_packed, const uint32_t *clump_len, uint32_t n_clumps, uint32_t tot_refs,
              const uint32_t *acx_lens, const void *acx_lists, int acx_fmt, int K,
              const uint32_t *badlist, uint32_t n_bad,
              const uint8_t score_lut[256], int xalpha, void **handle);

/* Prefilter + banded edit distance + re-scoring for a batch of query entries (the body of the OpenMP
 * loops burst.c:4077-4289 and 4343-4484).
 *   q_codes : concatenated 1-byte symbol codes (0..15, burst.c:1288-1307), entry i = [q_off[i], q_off[i+1]); code 0 (a character
 *             outside the IUPAC nucleotide alphabet) costs 255 against everything, i.e. it can only face a gap
 *   q_emac  : per-entry error budget (ShrBin.ed, burst.c:3074-3076)
 *   q_six   : per-entry shared slot (UniBin.six, burst.c:3078, 3106): a forward entry and its reverse
 *             complement carry the same value in [0, n_shared) and share the running minimum (burst.c:4218-4220)
 *   q_rc    : per-entry strand flag copied into BhipHit.rc
 *   q_flags : BHIP_Q_PREFILTER / BHIP_Q_EXHAUSTIVE per entry (NULL = all exhaustive when the handle has no
 *             accelerator, all prefiltered otherwise)
 *   all_hits: 0 = BEST/ALLPATHS/CAPITALIST semantics (only lanes with ed == minimum over the shared slot),
 *             1 = FORAGE semantics (every lane with ed <= budget, burst.c:4224),
 *             2 = BHIP_HITS_BEST: the minimum-only semantics with BEST's choice made on the device -- per query ENTRY the one
 *                 record the reference's BEST scan keeps among that entry's records (burst.c:4847-4891: equal ed, then the higher
 *                 f32 score, then the lower RefIxSrt[refIx]; needs bhip_set_ref_order).  One record per entry with a hit crosses
 *                 PCIe instead of every equally good reference; the choice between a query's two strands stays with the caller.
 *   hits/cap: caller's buffer; on BHIP_E_CAPACITY *n_hits is the number required and nothing is returned.
 * Records are returned sorted by (q, refIx). */
BHIP_API int bhip_align_batch(void *handle, const uint8_t *q_codes, const uint64_t *q_off, const uint16_t *q_emac,
                     const uint32_t *q_six, const uint8_t *q_rc, const uint8_t *q_flags,
                     uint32_t n_q, uint32_t n_shared, int all_hits,
                     BhipHit *hits, uint64_t cap, uint64_t *n_hits);

/* The same in two steps.  bhip_stage_queries is synchronous (the caller's arrays are free again at return) and replaces any
 * batch that was waiting; bhip_align_staged may then run any number of times on the resident batch.
 * bhip_align_batch(...) == bhip_stage_queries(...) followed by bhip_align_staged(...). */
BHIP_API int bhip_stage_queries(void *handle, const uint8_t *q_codes, const uint64_t *q_off, const uint16_t *q_emac,
                       const uint32_t *q_six, const uint8_t *q_rc, const uint8_t *q_flags, uint32_t n_q, uint32_t n_shared);
BHIP_API int bhip_align_staged(void *handle, int all_hits, BhipHit *hits, uint64_t cap, uint64_t *n_hits);
#define BHIP_HITS_MIN   0
#define BHIP_HITS_ALL   1
#define BHIP_HITS_BEST  2
/* The tie-break table of all_hits = BHIP_HITS_BEST: order[refIx] = RefIxSrt[refIx] (burst.c:3688-3693; the reference's BEST prefers the
 * lower original reference number among equally good hits, burst.c:4865-4868), n = tot_refs.  Copied to the device.  order = NULL:
 * nothing is changed, the return value says whether the handle holds a table (1) or not (0). */
BHIP_API int bhip_set_ref_order(void *handle, const uint32_t *order, uint32_t n);

/* Pipelined staging for a batch scheduler (the replacement of the OpenMP loops burst.c:4050-4078 / 4326-4344 hands batch k+1
 * to the device while batch k is being aligned).  A batch is given as up to a few SPANS of consecutive entries of the caller's
 * own arrays (process_queries keeps all forward entries and then all reverse complements, burst.c:3087-3109: a batch of
 * unique queries [u, u+B) is two spans), so nothing is gathered on the host.  Entry j of EVERY span shares slot j (a forward
 * entry and its reverse complement: UniBin.six, burst.c:3106) and is reported as BhipHit.q = q_base + j.
 * The call only ENQUEUES the copies and the device-side routing (length classes, prefilter / exhaustive route, seed plans)
 * on the library's staging stream and returns; the arrays must stay valid until the batch has been aligned.  Three staging
 * slots: beside the batch being aligned, two more can be staged; bhip_align_staged always takes the oldest one that has
 * not been aligned yet (and, when none is waiting, runs the last one again).  With a batch staged AHEAD of the one being
 * aligned -- a caller that stages two batches ahead, as bh_align_ranges does -- the library runs that batch's seed lookups
 * and match profiles beside the current batch's sweeps (option "seed_ahead", default 1).  Page-locked arrays (bhip_alloc_host / bhip_host_register) make the copies
 * asynchronous; pageable ones work too.  max_len = an upper bound of the entry lengths (0 = computed from the offsets). */
typedef struct BhipQuerySpan {
	const uint8_t  *codes;   /* symbol codes; entry j = codes[off[j] .. off[j+1]) */
	const uint8_t  *codes4;  /* optional: the same array packed two symbols per byte (symbol i = (codes4[i >> 1] >> 4 * (i & 1)) & 15, i counted
	                            from the start of `codes`); when every span has it, this is what crosses PCIe (half the bytes) */
	const uint64_t *off;     /* n + 1 offsets into codes (off[0] need not be 0) */
	const uint16_t *emac;    /* n budgets */
	const uint8_t  *rc;      /* n strand flags, or NULL = 0 */
	const uint8_t  *flags;   /* n BHIP_Q_*, or NULL (all spans or none) */
	uint32_t n;
	uint32_t q_base;         /* BhipHit.q of the span's first entry */
	const uint8_t  *codes2;  /* optional, for spans whose every symbol is A, C, G or T (codes 1..4): the array packed four symbols per byte
	                            (symbol i = ((codes2[i >> 2] >> 2 * (i & 3)) & 3) + 1); when every span of a batch has it, it is what crosses
	                            PCIe (a quarter of the bytes) */
	const uint16_t *len;     /* optional: the n entry lengths off[j+1] - off[j]; when every span has it, the lengths cross PCIe instead of the
	                            8-byte offsets (prefix sum on the device).  `codes` and `off` are still required in full: the host pass that
	                            handles batches with symbols outside the alphabet reads them */
} BhipQuerySpan;
BHIP_API int bhip_stage_spans(void *handle, const BhipQuerySpan *spans, uint32_t n_spans, uint32_t n_shared, uint32_t max_len);

/* Optional: a function the library calls from inside bhip_align_staged once the whole chain of the batch (and the seed lookups of the
 * next staged one) has been ENQUEUED and before it waits for the device -- the moment a batch scheduler has for its own host work
 * (staging the batch after next: ~20 asynchronous copies and launches) without leaving the device idle between two batches.  The
 * hook may call bhip_stage_spans on the same handle; nothing else.  fn = NULL removes it.  No reference counterpart (the reference's
 * scheduler is the OpenMP loop burst.c:4077). */
BHIP_API int bhip_set_enqueued_hook(void *handle, void (*fn)(void *ctx), void *ctx);

/* Optional: allocate now what batches of up to n_entries entries of up to max_len symbols will need, so that no allocation
 * (each one synchronises the device) falls into the first batches. */
BHIP_API int bhip_reserve(void *handle, uint32_t n_entries, uint32_t max_len);
/* the same with the symbol count of the largest batch to come (sum of its entries' lengths; 0 = n_entries x max_len): what holds
 * symbols and match profiles is sized from it, so that one long read among millions of short ones does not multiply every buffer */
BHIP_API int bhip_reserve_symbols(void *handle, uint32_t n_entries, uint32_t max_len, uint64_t total_symbols);

/* Page-locked host memory (hipHostMalloc / hipHostRegister behind the C ABI, for callers that are plain C): copies from and to
 * it run asynchronously beside the kernels.  bhip_alloc_host returns NULL when no device is present. */
BHIP_API void *bhip_alloc_host(uint64_t bytes);
BHIP_API void  bhip_free_host(void *p);
BHIP_API int   bhip_host_register(void *p, uint64_t bytes);
BHIP_API int   bhip_host_unregister(void *p);

/* Multi-GPU (no reference counterpart).  Query-sharded: the unique queries are cut across the GPUs of a node -- one
 * handle per device, the database replicated.  The hit records are wanted in rank 0's HOST memory (the consolidation runs there):
 * inside one node the host library lets them meet there without a collective (every rank's records reach host memory over its own
 * PCIe link behind its batches; host/bh_multi.c, host/bh_node.c) -- these entry points are the exchange for ranks that cannot see
 * each other's memory, and the option --gather rccl: one variable-length gather over RCCL / xGMI to rank 0's device
 * (ncclAllGather of the counts + grouped ncclSend / ncclRecv of the 20-byte records) and one copy to its host.  Database-sharded (databases beyond one
 * device): every rank holds a range of clumps and aligns all queries; the hits of a query are the references at its GLOBAL minimum
 * edit distance (burst.c:4217-4277), so the ranks combine one byte per unique query with bhip_comm_allreduce_min (ncclAllReduce,
 * MIN) before the same gather.  The ranks are the threads of one process (bhip_comm_create: ncclCommInitAll over `devices`) or one
 * process each (bhip_comm_unique_id on one of them, the 128 bytes carried to the others by the launcher, bhip_comm_create_rank:
 * ncclCommInitRank); every rank's thread calls the collectives with its own rank, all ranks' calls in flight together.
 * bhip_comm_gather_hits: rank 0 gets all records in rank order, counts[n_ranks] the per-rank numbers; BHIP_E_CAPACITY: *n_total
 * needed.  A rank that cannot take part (out of memory ...) makes the call fail on EVERY rank instead of leaving them waiting. */
BHIP_API int  bhip_comm_create(int n_ranks, const int *devices, void **comm);
BHIP_API int  bhip_comm_unique_id(void *id128);
BHIP_API int  bhip_comm_create_rank(int n_ranks, int rank, int device, const void *id128, void **comm);
BHIP_API int  bhip_comm_gather_hits(void *comm, int rank, const BhipHit *hits, uint64_t n, BhipHit *out, uint64_t cap, uint64_t *n_total, uint64_t *counts);
/* The same gather fed from the DEVICE: bhip_comm_stage_device after every batch copies the records of the handle's last alignment call
 * (still resident there) into the rank's send buffer, device to device, behind the `first_record` records staged before;
 * bhip_comm_gather_staged then sends the n staged records without uploading the host copy again.  A rank whose staging failed makes
 * the gather fail on every rank (the caller falls back on bhip_comm_gather_hits); bhip_comm_stage_reset forgets what was staged. */
BHIP_API int  bhip_comm_stage_device(void *comm, int rank, void *handle, uint64_t first_record, uint64_t *n_records);
BHIP_API int  bhip_comm_gather_staged(void *comm, int rank, uint64_t n, BhipHit *out, uint64_t cap, uint64_t *n_total, uint64_t *counts);
BHIP_API void bhip_comm_stage_reset(void *comm, int rank);
BHIP_API int  bhip_comm_fetch_gathered(void *comm, BhipHit *out, uint64_t cap, uint64_t *n_total);      /* rank 0 after BHIP_E_CAPACITY: no collective, the records are still on its device */
BHIP_API int  bhip_comm_allreduce_min(void *comm, int rank, uint8_t *buf, uint64_t n);
BHIP_API void bhip_comm_destroy(void *comm);

/* Cooperative accelerator build (no reference counterpart: make_accelerator, burst.c:3304-3532, is one process).  With the database
 * replicated over n_parts devices every handle needs the whole accelerator, and building it is most of what a job waits for before its
 * first batch; so every handle -- initialised with K = 0, i.e. without one -- builds the lists of 1/n_parts of the WORD space (runs of
 * words with equal numbers of window tuples, cut alike by every rank from a histogram of the database), and the ranks complete each
 * other's tables in two exchanges: Lens (4^K x 4 bytes in all), then the 4-byte records.  The result on every handle is what bhip_init
 * with K would have built alone, bit for bit (tests/test_gpu_acx.py).
 *   bhip_share_fn: called by every rank at the same two points with ITS array (a device pointer) and the same byte offsets
 *     byte_off[0 .. n_parts]: region [byte_off[r], byte_off[r + 1]) is valid on rank r when the call is made and on every rank when it
 *     returns.  status != 0 announces that this rank could not do its part: the function then moves nothing and returns 1 on EVERY rank
 *     (all handles fall back to building alone); < 0 = the exchange itself failed.
 *   Ready-made exchanges: bhip_comm_share (ctx = BhipCommRank: ncclBroadcast of every region from its builder, one group, over xGMI),
 *   bhip_team_share (ctx = a team of the threads of ONE process: peer copies device to device behind a barrier; the ranks may share a
 *   device), or the caller's own (python -m burst_amd.run stages through torch.distributed where its ranks have no RCCL communicator:
 *   bhip_device_copy moves a region between a device array and host memory).
 * bhip_build_accelerator_shared with n_parts = 1 builds alone (share may be NULL).  All ranks must call it together. */
typedef int (*bhip_share_fn)(void *ctx, void *device_base, const uint64_t *byte_off, int part, int n_parts, int status);
typedef struct { void *comm; int rank; } BhipCommRank;
BHIP_API int  bhip_build_accelerator_shared(void *handle, int K, int part, int n_parts, bhip_share_fn share, void *ctx);
BHIP_API int  bhip_comm_share(void *comm_rank, void *device_base, const uint64_t *byte_off, int part, int n_parts, int status);
BHIP_API int  bhip_team_create(int n_ranks, void **team);
BHIP_API void bhip_team_destroy(void *team);
BHIP_API int  bhip_team_share(void *team, void *device_base, const uint64_t *byte_off, int part, int n_parts, int status);
BHIP_API int  bhip_device_copy(void *dst, const void *src, uint64_t bytes, int to_device);      /* (device addresses are unified: whichever device the calling thread is on) */

/* The accelerator of a handle in the file's terms (read_accelerator's tables, burst.c:3535-3594): Lens[4^K], the clump ids of all
 * lists in word order (ascending inside a list, as the reference writes them with one thread), their 16-bit lane masks (device
 * layout only: bit z = lane z of the clump MAY hold the word -- since the 4-byte records of ABI 5 the device keeps a lane-set CODE
 * per entry, and the mask returned here is decode(code): the exact set for one or two lanes, the smallest enclosing quad pattern
 * beyond (csrc/bhip_lanecode.h); a superset never loses a lane), the BadList.  For a handle whose accelerator was built on the device
 * this is what make_accelerator (burst.c:3304-3532) would have written.  Any output pointer may be NULL. */
BHIP_API int bhip_acx_export(void *handle, uint32_t *lens, uint32_t *clumps, uint16_t *masks, uint64_t cap_entries, uint64_t *n_entries,
                    uint32_t *badlist, uint32_t cap_bad, uint32_t *n_bad);
/* The same lists piece by piece: entries [first, first + n_entries) in word order (a RefSeq-scale accelerator has tens of billions of
 * entries: a caller that writes the .acx streams them instead of holding 4 bytes of each); masks may be NULL. */
BHIP_API int bhip_acx_export_entries(void *handle, uint64_t first, uint64_t n_entries, uint32_t *clumps, uint16_t *masks);

/* Kernel-level entry (what one aded_mat16 call returns, burst.c:1078-1094): for explicit (query, clump)
 * pairs, mins[16*p + z] = edit distance of lane z (255 when > budget of the pair's query). */
BHIP_API int bhip_align_pairs(void *handle, const uint8_t *q_codes, const uint64_t *q_off, const uint16_t *q_emac,
                     uint32_t n_q, const uint32_t *pair_q, const uint32_t *pair_clump, uint64_t n_pairs,
                     uint8_t *mins);

/* Kernel-level entry for the prefilter alone: candidate (query, clump, count) triples with
 * count > max(len-(E+1)K, 0) (burst.c:4091-4092, 4126), BadList clumps not included.
 * Output sorted by (q, clump).  On BHIP_E_CAPACITY *n_out is the number required. */
BHIP_API int bhip_prefilter(void *handle, const uint8_t *q_codes, const uint64_t *q_off, const uint16_t *q_emac,
                   uint32_t n_q, uint32_t *out_q, uint32_t *out_clump, uint32_t *out_count,
                   uint64_t cap, uint64_t *n_out);

/* The records of the last bhip_align_staged / bhip_align_batch call stay resident on the device (same order as the host
 * copy; `hits` may be NULL there to skip the host copy altogether).  This copies them into caller-owned DEVICE memory,
 * e.g. the send buffer of an RCCL gather (no reference counterpart: multi-GPU).  Synchronous. */
BHIP_API int bhip_copy_hits_device(void *handle, void *dst_device, uint64_t cap_records, uint64_t *n_records);

/* With option "async_d2h" = 1 the records of bhip_align_staged / bhip_align_batch reach the caller's buffer BEHIND the call:
 * *n_hits is final at return, the bytes are not until bhip_sync_hits() (or bhip_destroy).  The copy of call k then
 * overlaps the kernels of call k+1; callers alternate between two result buffers (the library page-locks them once).
 * No reference counterpart (the reference appends ResultPods in place, burst.c:4230-4238). */
BHIP_API int bhip_sync_hits(void *handle);

/* Query ingest: sort n records of symbol codes (record r = codes[start[r] .. start[r] + len[r]), codes 0..15) in the order
 * of the reference's query sort -- byte-wise over the common length, the shorter record first, equal records in input order
 * (burst.c:363-366, 636-690) -- and mark the first record of every run of identical ones (the uniqueness pass of
 * burst.c:3036-3053).  perm[i] = input number of the i-th record in sorted order, is_new[i] = 1 iff it differs from the
 * (i-1)-th.  A device-side LSD radix sort over 16-symbol keys; needs no handle (it runs before the database is uploaded) and
 * about codes_bytes + 40 n bytes of device memory for the duration of the call.  max_len >= every len[r]. */
BHIP_API int bhip_sort_queries(int device, const uint8_t *codes, uint64_t codes_bytes, const uint64_t *start, const uint32_t *len,
                      uint64_t n, uint32_t max_len, uint32_t *perm, uint8_t *is_new);

/* Tuning knobs.  "prefilter_stride": 0 (default) = automatic sparse seeds -- per query the largest stride s <= K for
 * which an alignment within budget still keeps >= 3 of the words starting at 0, s, 2s, ... (one edit destroys at most
 * ceil(K/s) of them), fewest .acx look-ups with the same no-false-negative guarantee; s >= 1 forces every s-th word,
 * 1 = every word = the reference's own threshold count > len-(E+1)K (burst.c:4091-4092, 4126).
 * "two_stage": 1 (default) = prefix filter + windowed full-length edit distance, 0 = one full-length sweep.
 * "lanes": 1..15 (default 1) sub-pipelines a staged batch is cut into; their prefilter / sweep / window+re-score stages
 * run as a software pipeline on three HIP streams; "lane_min_entries" (default 32768) = fewest entries worth a lane.
 * "sweep_blocks": 1..8 workgroups per CU of the sweep kernels.  "lane_masks": 1 (default) = lane-resolved prefilter.
 * "prefilter_table": 0 (default, chosen from the accelerator's list lengths) or 9/10/11 = log2 slots of the per-query
 * tables; "prefilter_algo": -1 (default) = counting-filter kernel, switching to the exact-hash kernel for workloads where
 * more than 20 % of the list records survive the filter, 0 / 1 = force one of them; "prefilter_waves": 0 (default, as many as fit) .. 16 single-wave prefilter blocks per CU.
 * "prune": 1 (default) = when only the minimum per shared slot is wanted, lanes whose seed count bounds their edit distance
 * above the query's best bound are swept only if the first sweep leaves room for them (exact: the bound is a lower bound).
 * "async_d2h": 0 (default) / 1 = asynchronous hand-over of the records, see bhip_sync_hits.
 * "rescore_reg": 1 (default) = register-band re-scorer for narrow bands, 0 = LDS band only.
 * "band": 1 (default) = windows whose flagged diagonals fit two to four words of the bit-vector column are swept by the banded kernel
 * (only those words are stepped; exact), 0 = every window through the full column.  "oversub": 1..16 (default 2) = blocks launched per
 * resident block slot of the per-item kernels (prefix tasks, windows, re-scoring); "band_blocks": 0 (default, as many as fit) or the
 * 64-thread blocks per CU of the banded kernel.
 * "host_routing": 0 (default) = batches are routed (length classes, seed plans, lists) by a device kernel, 1 = by the host pass
 * that otherwise only handles batches with query symbols outside the alphabet.  "discard_staged": forget batches that were
 * staged and not aligned (after an error).  "seed_ahead": 1 (default) = the seed lookups and match profiles of the next staged
 * batch run while the current one is swept, 0 = in place; "seed_ahead_blocks" (default 2, 0 = unlimited) / "peq_ahead_blocks"
 * (default 16) = 256-thread blocks per CU those kernels get while they share the device with the sweeps.
 * "seed_min_need": -1 (default) / 0 / n -- of a query's sampled words the ones with the LONGEST accelerator lists are left out while the
 * number of words an alignment within budget is guaranteed to keep stays >= n (any subset of the sampled words gives the same
 * no-false-negative guarantee with a correspondingly smaller count); 0 = every list is walked; -1 = n = 3 when the expected record
 * stream is long enough to matter and short enough for the counting filter to stay selective (bhip_align.hip seed_min_need_for).
 * "seed_drop_len" (default 8): lists shorter than this are never left out.  "prefilter_rb": 0 (default, from the expected stream) or
 * 2 / 3 / 4 = blocks of 64 list records per query the counting-filter kernel fetches one quad ahead and keeps in registers.
 * "prefilter_bytes": 1 (default) = a query whose record stream is at most 255 records counts in bytes (twice the counters in the same LDS).
 * None of these changes a result.  BHIP_OPTS="name=value,..." in the environment sets options at bhip_init. */
BHIP_API int bhip_set_option(void *handle, const char *name, long long value);

/* Stats of the last call; device properties (name, CU count) for reports. */
BHIP_API int bhip_get_stats(void *handle, BhipStats *out);
BHIP_API int bhip_device_info(void *handle, char *name, int name_cap, int *n_cu, uint64_t *hbm_bytes);

BHIP_API void bhip_destroy(void *handle);
BHIP_API const char *bhip_last_error(void);
/* ABI version of this header */
BHIP_API int bhip_abi_version(void);
#define BHIP_ABI_VERSION 8

#ifdef __cplusplus
}
#endif
#endif
