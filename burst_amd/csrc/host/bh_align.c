/* bh_align.c -- GPU batch scheduler: the replacement of the two OpenMP loops of do_alignments
 * (burst.c:4050-4289 accelerated, 4326-4488 exhaustive).  Unique queries are cut into contiguous batches in
 * sorted order; a forward entry and its reverse-complement twin always travel in the same batch because they
 * share the running minimum (ShrBin.ed, burst.c:277-280, 4218-4220).  Batches are staged (bhip_stage_spans) one ahead of
 * the batch being aligned (bhip_align_staged).
 */
#include "burst_host.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <omp.h>
#include <sys/mman.h>

static double now_sec(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* one user of a device's free memory at a time among this process's set-up steps: the device build of an accelerator sizes its slices
 * from the memory that is free when it starts (bhip_acx.hip), so the query sort of the ingest thread (bh_queries.c) on the same device
 * waits for an upload in progress, and the upload for a sort in progress */
static pthread_mutex_t g_dev_gate[64] = { [0 ... 63] = PTHREAD_MUTEX_INITIALIZER };
void bh_device_gate(int device, int take) {
	if (device < 0) return;
	if (take) pthread_mutex_lock(&g_dev_gate[device & 63]); else pthread_mutex_unlock(&g_dev_gate[device & 63]);
}
int bh_device_gate_try(int device) {      /* 1 = taken (release with bh_device_gate(device, 0)), 0 = somebody else is setting up on this device */
	return device < 0 ? 1 : pthread_mutex_trylock(&g_dev_gate[device & 63]) == 0;
}

/* BEST's tie-break table (RefIxSrt, burst.c:3688-3693, 4865-4868) goes to the device with the database, so that -m BEST can have the
 * device keep one record per entry (bhip_align_staged with BHIP_HITS_BEST).  A database without the table (a view made for a
 * kernel test) simply keeps the choice on the host; so does a failed upload (436 MB at the metric's size). */
static void give_ref_order(const BhDb *db, void *hh) {
	if (db->refIxSrt && db->totR && !getenv("BURST_HOST_BEST_ON_HOST")) (void)bhip_set_ref_order(hh, db->refIxSrt, db->totR);
}
int bh_device_open_ex(const BhDb *db, int device, int z, int build_K, void **hip_handle) {
	uint8_t lut[256];
	bh_score_lut(z, lut);
	bh_device_gate(device, 1);
	/* a database read with its .acx goes up with the file's tables; without one, build_K > 0 has the device build the accelerator
	 * from the references (make_accelerator, burst.c:3304-3532, as device kernels) */
	int rc = bhip_init(device, db->packed, db->clumpLen, db->numRclumps, db->totR,
	                   db->hasAcx ? db->acxLens : NULL, db->hasAcx ? db->acxLists : NULL, db->acxFmt, db->hasAcx ? db->K : build_K,
	                   db->badList, db->badSz, lut, db->xalpha, hip_handle);
	bh_device_gate(device, 0);
	if (rc) return bh_set_error(rc == BHIP_E_ARG ? BH_E_USAGE : BH_E_DEVICE, "libburst_hip: %s", bhip_last_error());
	give_ref_order(db, *hip_handle);
	return BH_OK;
}
int bh_device_open(const BhDb *db, int device, int z, void **hip_handle) { return bh_device_open_ex(db, device, z, 0, hip_handle); }

/* One of n_parts handles of a REPLICATED database whose accelerator the devices build together (bhip_build_accelerator_shared: every rank
 * the lists of its share of the words, two exchanges through `share`).  The handle goes up without an accelerator first; all ranks must be
 * in this call together.  The caller holds the devices' gates (bh_device_gate) around the ranks' calls: ranks that share a device wait for
 * each other inside. */
int bh_device_open_shared(const BhDb *db, int device, int z, int build_K, int part, int n_parts, bhip_share_fn share, void *ctx, void **hip_handle) {
	uint8_t lut[256];
	bh_score_lut(z, lut);
	if (db->hasAcx || db->xalpha || build_K <= 0) return bh_set_error(BH_E_USAGE, "the cooperative build is for databases without an accelerator file");
	int rc = bhip_init(device, db->packed, db->clumpLen, db->numRclumps, db->totR, NULL, NULL, 0, 0, db->badList, db->badSz, lut, 0, hip_handle);
	/* (a rank that has no handle still answers the first exchange, or the others would wait for it) */
	if (rc) { static const uint64_t none[BH_MAX_RANKS + 1] = {0}; if (n_parts > 1 && n_parts <= BH_MAX_RANKS) (void)share(ctx, NULL, none, part, n_parts, 1); }
	else rc = bhip_build_accelerator_shared(*hip_handle, build_K, part, n_parts, share, ctx);
	if (rc) return bh_set_error(rc == BHIP_E_ARG ? BH_E_USAGE : BH_E_DEVICE, "libburst_hip: %s", bhip_last_error());
	give_ref_order(db, *hip_handle);
	return BH_OK;
}

static void add_stats(BhipStats *t, const BhipStats *s) {
	t->n_queries += s->n_queries; t->n_pairs += s->n_pairs; t->n_columns += s->n_columns; t->n_raw_hits += s->n_raw_hits;
	t->n_hits += s->n_hits; t->acx_entries_read += s->acx_entries_read; t->bytes_algorithmic += s->bytes_algorithmic;
	t->ms_h2d += s->ms_h2d; t->ms_prefilter += s->ms_prefilter; t->ms_peq += s->ms_peq; t->ms_myers += s->ms_myers;
	t->ms_rescore += s->ms_rescore; t->ms_d2h += s->ms_d2h; t->ms_total += s->ms_total; t->myers_launches += s->myers_launches;
	t->n_windows += s->n_windows; t->n_window_columns += s->n_window_columns; t->n_lane_tasks += s->n_lane_tasks; t->n_task_columns += s->n_task_columns; t->ms_myers_prefix += s->ms_myers_prefix;
	t->ms_myers_window += s->ms_myers_window; t->prefix_words = s->prefix_words;
	t->n_seed_words += s->n_seed_words; t->ms_prefilter_hash += s->ms_prefilter_hash; t->ms_seed += s->ms_seed; t->ms_stage_copy += s->ms_stage_copy; t->ms_stage_route += s->ms_stage_route; t->prefilter_launches += s->prefilter_launches; t->prefilter_algo = s->prefilter_algo;
}

/* page-locked result buffer: the records of batch k are copied out behind bhip_align_staged while batch k+1 computes */
static BhipHit *hits_alloc(uint64_t cap, int *pinned) {
	BhipHit *p = bhip_alloc_host(cap * sizeof(BhipHit));
	*pinned = p != NULL;
	if (!p) p = malloc(cap * sizeof(BhipHit));
	return p;
}
/* (pinned == 2: memory the run does not own -- a shared-memory segment of bh_node.c) */
static void hits_free(BhipHit *p, int pinned) { if (pinned == 2) return; if (pinned) bhip_free_host(p); else free(p); }

void bh_run_free(BhRun *run) { if (run) { if (run->hits) hits_free(run->hits, run->hitsPinned); memset(run, 0, sizeof *run); } }

/* page-lock the arrays the batches are copied from (optional: pageable arrays work, the copies then keep the host thread busy) */
int bh_queries_pin(BhQueries *Q) {
	if (Q->pinned || !Q->numEntries) return BH_OK;
	void *cp = Q->codes4 ? (void *)Q->codes4 : (void *)Q->codes;
	if (bhip_host_register(cp, Q->codes4 ? (Q->qoff[Q->numEntries] + 1) / 2 + 16 : Q->qoff[Q->numEntries] + 16)) return BH_OK;          /* no device / no room: stay pageable */
	if (bhip_host_register(Q->qoff, (Q->numEntries + 1) * sizeof(*Q->qoff))) { bhip_host_unregister(cp); return BH_OK; }
	Q->pinned = 1;
	if (!bhip_host_register(Q->emac, Q->numEntries * sizeof(*Q->emac))) Q->pinned |= 2;
	if (!bhip_host_register(Q->rc, Q->numEntries)) Q->pinned |= 4;
	if (!bhip_host_register(Q->flags, Q->numEntries)) Q->pinned |= 8;
	if (Q->codes2 && !bhip_host_register(Q->codes2, (Q->qoff[Q->numEntries] + 3) / 4 + 16)) Q->pinned |= 16;
	if (Q->len16 && !bhip_host_register(Q->len16, (Q->numUniq + 1) * sizeof(*Q->len16))) Q->pinned |= 32;
	return BH_OK;
}

/* Double-buffered batch loop (the reference's `omp for schedule(dynamic,1)` over bunches, burst.c:4050-4078): batch k+1 is
 * handed to the device -- straight from the query arrays, as one span of forward entries and one of reverse complements --
 * before batch k is aligned, so its copies and routing run beside batch k's kernels; the records of batch k travel back
 * behind the call that produced them (option async_d2h).  The batches are the ranges [u0[i], u1[i]) of unique queries, each
 * cut into pieces of at most batch_uniq, in the order given. */
/* page-locked record buffer of a zeroed or used BhRun for at least cap records (bh_align_ranges_reuse then allocates nothing) */
int bh_run_reserve(BhRun *run, uint64_t cap) {
	if (run->hits && run->capHits >= cap) return BH_OK;
	if (run->hits) hits_free(run->hits, run->hitsPinned);
	run->hits = hits_alloc(cap, &run->hitsPinned); run->capHits = run->hits ? cap : 0; run->nHits = 0;
	return run->hits ? BH_OK : bh_set_error(BH_E_OOM, "OOM:hits");
}
/* the same in pageable memory (a buffer no device copy lands in: the gathered records of ranks that live in one process or share
 * a node).  Touched here by a team of threads, huge pages advised: the first write to 0.8 GB of fresh pages -- 200 000 page faults
 * -- took longer than aligning the reads the records belong to when it happened inside the hand-over. */
int bh_run_reserve_plain(BhRun *run, uint64_t cap) {
	if (run->hits && run->capHits >= cap) return BH_OK;
	if (run->hits) hits_free(run->hits, run->hitsPinned);
	const size_t bytes = ((size_t)cap * sizeof(BhipHit) + ((size_t)1 << 21) - 1) & ~(((size_t)1 << 21) - 1);
	run->hits = aligned_alloc((size_t)1 << 21, bytes); run->hitsPinned = 0; run->capHits = run->hits ? cap : 0; run->nHits = 0;
	if (!run->hits) return bh_set_error(BH_E_OOM, "OOM:hits");
	(void)madvise(run->hits, bytes, MADV_HUGEPAGE);
	const int nt = omp_get_max_threads() > 16 ? 16 : omp_get_max_threads();
	#pragma omp parallel for schedule(static) num_threads(nt)
	for (size_t o = 0; o < bytes; o += (size_t)1 << 21) memset((char *)run->hits + o, 0, (size_t)1 << 21);
	return BH_OK;
}
/* records made elsewhere (a rank's align back end: BhMultiRank.align) into a run */
int bh_run_put(BhRun *run, const BhipHit *hits, uint64_t n) {
	if (run->capHits < n || !run->hits) {
		if (run->hits && run->hitsPinned == 2) return bh_set_error(BH_E_CAPACITY, "the rank's shared-memory segment holds %lu records, %lu needed", (unsigned long)run->capHits, (unsigned long)n);
		int rc = bh_run_reserve_plain(run, n + 1);
		if (rc) return rc;
	}
	if (n) memcpy(run->hits, hits, n * sizeof(BhipHit));
	run->nHits = n;
	return BH_OK;
}
static int align_ranges(void *hh, const BhQueries *Q, const uint64_t *r0, const uint64_t *r1, uint32_t n_ranges, BhMode mode, uint64_t batch_uniq, BhRun *run);
int bh_align_ranges(void *hh, const BhQueries *Q, const uint64_t *r0, const uint64_t *r1, uint32_t n_ranges, BhMode mode, uint64_t batch_uniq, BhRun *run) {
	memset(run, 0, sizeof *run);
	return align_ranges(hh, Q, r0, r1, n_ranges, mode, batch_uniq, run);
}
/* the same into a BhRun that has been used before (or zeroed by the caller): its page-locked record buffer is kept and grown
 * only when needed -- page-locking hundreds of megabytes costs more than aligning a batch */
int bh_align_ranges_reuse(void *hh, const BhQueries *Q, const uint64_t *r0, const uint64_t *r1, uint32_t n_ranges, BhMode mode, uint64_t batch_uniq, BhRun *run) {
	BhipHit *keep = run->hits; const uint64_t cap = run->capHits; const int pin = run->hitsPinned;
	void (*const cb)(void *, void *, uint64_t, uint64_t) = run->onBatch; void *const cbctx = run->onBatchCtx;      /* (the caller's per-batch hook survives the reset) */
	memset(run, 0, sizeof *run);
	run->hits = keep; run->capHits = cap; run->hitsPinned = pin; run->onBatch = cb; run->onBatchCtx = cbctx;
	return align_ranges(hh, Q, r0, r1, n_ranges, mode, batch_uniq, run);
}
/* staging of batch k of a job: a batch of unique queries [u, u + B) is two spans of the caller's arrays (forward entries, then their
 * reverse complements, burst.c:3087-3109); only copies and launches are enqueued here */
typedef struct StageCtx {
	void *hh; const BhQueries *Q; const uint64_t *bu; uint64_t nBatches; int twoStrand; BhRun *run;
	uint64_t hook_next;      /* batch the library's "chain enqueued" hook stages (0 = none) */
	int rc; double secHook;
} StageCtx;
static void stage_batch(StageCtx *c, uint64_t k) {
	const BhQueries *Q = c->Q;
	const uint64_t u_ = c->bu[2 * k], B_ = c->bu[2 * k + 1];
	BhipQuerySpan sp_[2];
	memset(sp_, 0, sizeof sp_);
	/* a batch of A/C/G/T-only queries travels four symbols per byte, every batch with 2-byte lengths (same for both strands) */
	const int clean_ = Q->codes2 && Q->ambBefore && Q->ambBefore[u_ + B_] == Q->ambBefore[u_] && !getenv("BURST_HOST_NO_PACK2");
	if (clean_) sp_[0].codes2 = sp_[1].codes2 = Q->codes2;
	if (Q->len16 && !getenv("BURST_HOST_NO_PACK2")) sp_[0].len = sp_[1].len = Q->len16 + u_;
	sp_[0].codes = Q->codes; sp_[0].codes4 = Q->codes4; sp_[0].off = Q->qoff + u_; sp_[0].emac = Q->emac + u_; sp_[0].rc = Q->rc + u_; sp_[0].flags = Q->flags + u_; sp_[0].n = (uint32_t)B_; sp_[0].q_base = (uint32_t)u_;
	if (c->twoStrand) { const uint64_t e_ = Q->numUniq + u_;
		sp_[1].codes = Q->codes; sp_[1].codes4 = Q->codes4; sp_[1].off = Q->qoff + e_; sp_[1].emac = Q->emac + e_; sp_[1].rc = Q->rc + e_; sp_[1].flags = Q->flags + e_; sp_[1].n = (uint32_t)B_; sp_[1].q_base = (uint32_t)e_; }
	const double ts_ = now_sec();
	if (bhip_stage_spans(c->hh, sp_, c->twoStrand ? 2 : 1, (uint32_t)B_, Q->maxLen)) c->rc = bh_set_error(BH_E_DEVICE, "libburst_hip: %s", bhip_last_error());
	c->run->secAlign += now_sec() - ts_;
}
/* called by the library from inside bhip_align_staged, when the chain of the current batch has been enqueued and before it waits for
 * the device: the batch after next is staged NOW, beside the kernels, instead of between two batches with the device idle */
static void stage_hook(void *p) {
	StageCtx *c = (StageCtx *)p;
	if (c->hook_next && c->hook_next < c->nBatches && !c->rc) { const double t0 = now_sec(); stage_batch(c, c->hook_next); c->secHook += now_sec() - t0; }
	c->hook_next = 0;
}

static int align_ranges(void *hh, const BhQueries *Q, const uint64_t *r0, const uint64_t *r1, uint32_t n_ranges, BhMode mode, uint64_t batch_uniq, BhRun *run) {
	if (!batch_uniq) batch_uniq = 1u << 18;
	if (batch_uniq > 0x7FFFFFFFu) batch_uniq = 0x7FFFFFFFu;
	/* batch list */
	uint64_t nBatches = 0, totU = 0;
	for (uint32_t i = 0; i < n_ranges; ++i) {
		const uint64_t b = r1[i] > Q->numUniq ? Q->numUniq : r1[i];
		if (r0[i] < b) { nBatches += (b - r0[i] + batch_uniq - 1) / batch_uniq; totU += b - r0[i]; }
	}
	if (!nBatches) return BH_OK;
	/* A job of one or two batches (a rank's share of a short job: 10 M reads over 8 GPUs are 1.25 M reads each) has nothing to overlap
	 * its staging, seed lookups and match profiles with -- they are paid in full in front of the chain.  Cutting it into pieces so that
	 * piece k + 1 is prepared while piece k is aligned was MEASURED and does not pay: every piece brings the fixed costs of a batch
	 * (about twenty dependent launches, the prefilter's floor) -- 1.25 M reads against the 19 GB database: 4.49 ms whole, 6.11 ms in four
	 * pieces (profiles/r04h_bench.json).  BURST_HOST_PIECES = n > 1 still cuts such a job into n pieces, for experiments. */
	if (nBatches <= 2 && totU >= ((uint64_t)1 << 18) && getenv("BURST_HOST_PIECES") && atoi(getenv("BURST_HOST_PIECES")) > 1) {
		const uint64_t pieces = (uint64_t)atoi(getenv("BURST_HOST_PIECES"));
		const uint64_t per = (totU + pieces - 1) / pieces;
		if (per < batch_uniq) {
			batch_uniq = per < ((uint64_t)1 << 16) ? ((uint64_t)1 << 16) : per;
			nBatches = 0;
			for (uint32_t i = 0; i < n_ranges; ++i) {
				const uint64_t b = r1[i] > Q->numUniq ? Q->numUniq : r1[i];
				if (r0[i] < b) nBatches += (b - r0[i] + batch_uniq - 1) / batch_uniq;
			}
		}
	}
	uint64_t *bu = malloc(nBatches * 2 * sizeof(*bu));
	if (!bu) return bh_set_error(BH_E_OOM, "OOM:batches");
	{
		uint64_t k = 0;
		for (uint32_t i = 0; i < n_ranges; ++i) {
			const uint64_t b = r1[i] > Q->numUniq ? Q->numUniq : r1[i];
			for (uint64_t u = r0[i]; u < b; u += batch_uniq) {
				const uint64_t B = u + batch_uniq <= b ? batch_uniq : b - u;
				bu[2 * k] = u; bu[2 * k + 1] = B; ++k;
			}
		}
	}
	const int twoStrand = Q->numEntries > Q->numUniq;
	/* ANY: any hit within budget will do (burst.c:4224) -- the report picks the one the reference meets first.  BEST: the device keeps one
	 * record per entry when it holds the tie-break table (the report's scan over an entry's records then meets a single one) */
	const int all_hits = (mode == BH_FORAGE || mode == BH_ANY) ? BHIP_HITS_ALL : (mode == BH_BEST && !getenv("BURST_HOST_BEST_ON_HOST") && bhip_set_ref_order(hh, NULL, 0) == 1) ? BHIP_HITS_BEST : BHIP_HITS_MIN;
	uint64_t capHits = totU * (twoStrand ? 2 : 1) + totU / 2 + (1u << 20), nHits = 0;
	int pinned = run->hitsPinned;
	BhipHit *hits = run->hits;
	if (hits && run->capHits >= capHits) capHits = run->capHits;
	else { if (hits) hits_free(hits, pinned); hits = hits_alloc(capHits, &pinned); }
	run->hits = NULL; run->capHits = 0;
	if (!hits) { free(bu); return bh_set_error(BH_E_OOM, "OOM:hits"); }
	bhip_set_option(hh, "async_d2h", 1);
	int rc = BH_OK;
	StageCtx sc = {hh, Q, bu, nBatches, twoStrand, run, 0, BH_OK, 0.0};
	#define STAGE(k) do { stage_batch(&sc, (k)); if (sc.rc) rc = sc.rc; } while (0)
	const int dbg = getenv("BURST_HOST_DEBUG") != NULL;
	const double tb0 = now_sec();
	/* two batches ahead: while batch k is aligned, batch k+1 is staged already (the library runs its seed lookups and profile
	 * builds beside batch k's sweeps) and batch k+2 is being copied and routed.  The third batch is only staged after the first
	 * has been aligned: with three batches' copies queued before the first record copy, that copy's hipMemcpyAsync blocked the
	 * calling thread for 11 ms (measured, ROCm 7.2) */
	STAGE((uint64_t)0);
	if (nBatches > 1 && rc == BH_OK) STAGE((uint64_t)1);
	const int use_hook = !getenv("BURST_HOST_NO_HOOK") && bhip_set_enqueued_hook(hh, stage_hook, &sc) == 0;
	for (uint64_t k = 0; k < nBatches && rc == BH_OK; ++k) {
		const double tk0 = now_sec();
		/* batch k + 2: staged from inside bhip_align_staged(k), once the chain of batch k is enqueued (the slot of batch k - 1 is free by
		 * then); without the hook, here -- between two batches */
		if (k > 0 && k + 2 < nBatches) { if (use_hook) sc.hook_next = k + 2; else { STAGE(k + 2); if (rc) break; } }
		const double tk1 = now_sec();
		for (;;) {
			uint64_t n = 0;
			const double t0 = now_sec();
			int r = bhip_align_staged(hh, all_hits, hits + nHits, capHits - nHits, &n);
			run->secAlign += now_sec() - t0;
			if (r == BHIP_E_CAPACITY) {          /* grow; the records of the batch stay resident on the device and are delivered by the next call */
				const uint64_t ncap = nHits + n + (nHits + n) / 2 + 1024;
				int npin = 0;
				bhip_sync_hits(hh);
				BhipHit *nh = hits_alloc(ncap, &npin);
				if (!nh) { rc = bh_set_error(BH_E_OOM, "OOM:hits"); break; }
				memcpy(nh, hits, nHits * sizeof(*hits));
				hits_free(hits, pinned);
				hits = nh; pinned = npin; capHits = ncap;
				continue;
			}
			if (r == BHIP_E_RESCORE) {   /* the reference's own stop (burst.c:812-816, exit(1)) */
				printf("\nCRITICAL ERROR: Truncation within known good path.\n");
				rc = bh_set_error(BH_E_USAGE, "libburst_hip: %s", bhip_last_error()); break;
			}
			if (r) { rc = bh_set_error(BH_E_DEVICE, "libburst_hip: %s", bhip_last_error()); break; }
			if (sc.rc) { rc = sc.rc; break; }
			if (sc.hook_next) { STAGE(sc.hook_next); sc.hook_next = 0; if (rc) break; }      /* (the hook did not run: a call that only delivered resident records) */
			if (run->onBatch) run->onBatch(run->onBatchCtx, hh, nHits, n);      /* (the batch's records are still resident on the device) */
			nHits += n;
			if (k == 0 && nBatches > 2) STAGE((uint64_t)2);      /* (the third batch only now: see the note at the loop) */
			BhipStats st;
			if (!bhip_get_stats(hh, &st)) add_stats(&run->total, &st);
			++run->nBatches;
			if (dbg) fprintf(stderr, "[bh_align] batch %lu: %lu entries, staging the next one %.3f ms, align %.3f ms (device %.3f ms), %lu records; %.3f ms since the start\n",
			                 (unsigned long)k, (unsigned long)bu[2 * k + 1], 1e3 * (tk1 - tk0), 1e3 * (now_sec() - tk1), st.ms_total, (unsigned long)n, 1e3 * (now_sec() - tb0));
			if (dbg) fprintf(stderr, "[bh_align]   device ms: staging %.2f, profiles %.2f, seeds %.2f, prefilter %.2f, prefix sweep %.2f, window sweep %.2f, re-scoring %.2f, sort + hand-over %.2f\n",
			                 st.ms_h2d, st.ms_peq, st.ms_seed, st.ms_prefilter_hash, st.ms_myers_prefix, st.ms_myers_window, st.ms_rescore, st.ms_d2h);
			break;
		}
	}
	#undef STAGE
	bhip_set_enqueued_hook(hh, NULL, NULL);
	if (dbg) fprintf(stderr, "[bh_align] %lu batches; %.3f ms of staging calls ran inside bhip_align_staged (chain enqueued, device busy)\n", (unsigned long)nBatches, 1e3 * sc.secHook);
	free(bu);
	{ const double t0 = now_sec(); bhip_sync_hits(hh); run->secAlign += now_sec() - t0; }
	bhip_set_option(hh, "async_d2h", 0);
	if (rc) { bhip_set_option(hh, "discard_staged", 1); hits_free(hits, pinned); return rc; }
	run->hits = hits; run->nHits = nHits; run->hitsPinned = pinned; run->capHits = capHits;
	return BH_OK;
}

int bh_align(void *hh, const BhQueries *Q, uint64_t u0, uint64_t u1, BhMode mode, uint64_t batch_uniq, BhRun *run) {
	return bh_align_ranges(hh, Q, &u0, &u1, 1, mode, batch_uniq, run);
}
