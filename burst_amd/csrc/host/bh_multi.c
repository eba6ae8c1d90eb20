/* bh_multi.c -- the multi-GPU search of a node (no reference counterpart: the reference is one process), shared by
 * the burst_hip command line (ranks = threads of one process, one device handle each) and bench.py (one process per GPU).
 *
 *   query-sharded   the database replicated on every device; rank r aligns its ranges of unique queries (a query and its reverse
 *                   complement stay together: they share the running minimum, burst.c:277-280, 4218); ONE exchange: the gather of
 *                   the 20-byte records to rank 0 (bhip_comm_gather_hits: RCCL over xGMI, or -- all ranks in one process --
 *                   concatenation in host memory).
 *   database-sharded  for databases beyond one device: rank r holds the clumps [c0, c1) (bh_clump_shard: about the same number of
 *                   reference columns each) and aligns ALL queries against them.  What the reference keeps per query are the
 *                   references at the query's minimum edit distance over the WHOLE database (Sb->ed, burst.c:4217-4277), so the
 *                   ranks combine one byte per unique query (bhip_comm_allreduce_min = ncclAllReduce MIN, or a host loop), drop
 *                   what lies above it (not in FORAGE, which keeps everything within budget, burst.c:4224), and gather as before;
 *                   rank 0 puts the records in (query, reference) order: the set a single device holding everything produces.
 *   both            S shards x G replica groups (rank = group * S + shard): a database that needs S devices, its queries cut over
 *                   the G groups -- e.g. the metric's 31.5 GB database on 8 GPUs as 2 shards x 4 groups, each group aligning a
 *                   quarter of the reads at the rate one device reaches on half the database.
 */
#include "burst_host.h"
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>
#include <sys/mman.h>

/* host-side passes over the records are memory-bound: a team of 16 saturates the memory system, and asking for the machine's 256
 * hardware threads between regions of n_local threads makes the OpenMP runtime rebuild its pool every time (measured: 0.12 s for
 * a 5 ms merge) */
static int team(void) { const int t = omp_get_max_threads(); return t > 16 ? 16 : t < 1 ? 1 : t; }

void bh_clump_shard(const BhDb *db, int n_ranks, int rank, uint32_t *c0, uint32_t *c1) {
	uint64_t total = 0;
	for (uint32_t c = 0; c < db->numRclumps; ++c) total += db->clumpLen[c];
	/* cut k = first clump at which the running sum reaches total * k / n */
	uint32_t cut[2] = {0, db->numRclumps};
	uint64_t run = 0; uint32_t c = 0;
	for (int k = 0; k < 2; ++k) {
		const int r = rank + k;
		if (r <= 0) { cut[k] = 0; continue; }
		if (r >= n_ranks) { cut[k] = db->numRclumps; continue; }
		const double want = (double)total * (double)r / (double)n_ranks;
		while (c < db->numRclumps && (double)(run + db->clumpLen[c]) < want) run += db->clumpLen[c++];
		cut[k] = c;
	}
	*c0 = cut[0]; *c1 = cut[1] < cut[0] ? cut[0] : cut[1];
}

/* smallest edit distance per unique query among a rank's records (255 = none) */
static void shard_minima(const BhQueries *Q, const BhRun *run, uint8_t *best) {
	memset(best, 255, Q->numUniq);
	for (uint64_t i = 0; i < run->nHits; ++i) {
		const uint32_t q = run->hits[i].q, s = q < Q->numUniq ? q : q - (uint32_t)Q->numUniq;
		if (run->hits[i].ed < best[s]) best[s] = run->hits[i].ed;
	}
}
static void shard_filter(const BhQueries *Q, BhRun *run, const uint8_t *best) {
	uint64_t k = 0;
	for (uint64_t i = 0; i < run->nHits; ++i) {
		const uint32_t q = run->hits[i].q, s = q < Q->numUniq ? q : q - (uint32_t)Q->numUniq;
		if (run->hits[i].ed == best[s]) run->hits[k++] = run->hits[i];
	}
	run->nHits = k;
}
/* Records in any order -> (query entry, reference) order: what a single device holding the whole database produces.
 * (query entry, reference) pairs are unique -- the ranks' clump ranges are disjoint -- so the order is fully determined by the
 * two keys and the sort need not be stable: a counting sort by entry whose counts and slots are taken with relaxed atomics by a
 * team of threads (neighbouring records belong to neighbouring entries, so the threads rarely meet), a two-level scan, and a
 * pass that puts the few records of each entry in reference order.  The sources are the ranks' runs where they lie (all ranks
 * in one process: no concatenation, no scratch copy) or the gathered array (in place through a scratch array). */
static int by_ref(const void *a, const void *b) {
	const uint32_t x = ((const BhipHit *)a)->refIx, y = ((const BhipHit *)b)->refIx;
	return x < y ? -1 : x > y;
}
static int order_into(const BhipHit *const *src, const uint64_t *cnt, int n_src, BhipHit *dst, uint64_t n_entries) {
	uint64_t n = 0;
	for (int r = 0; r < n_src; ++r) n += cnt[r];
	if (!n) return BH_OK;
	uint64_t *c = calloc(n_entries + 1, sizeof(*c));
	int nt = team(); if (n < (1u << 16)) nt = 1;
	uint64_t *bs = calloc((size_t)nt + 1, sizeof(*bs));
	if (!c || !bs) { free(c); free(bs); return bh_set_error(BH_E_OOM, "OOM:order_records"); }
	int bad = 0;
	#pragma omp parallel num_threads(nt)
	{
		const int t = omp_get_thread_num(), T = omp_get_num_threads();
		/* 1. records per entry */
		int mybad = 0;
		for (int r = 0; r < n_src; ++r) {
			const BhipHit *h = src[r];
			#pragma omp for schedule(static) nowait
			for (uint64_t i = 0; i < cnt[r]; ++i) {
				const uint64_t q = h[i].q;
				if (q >= n_entries) mybad = 1; else __atomic_fetch_add(&c[q], 1, __ATOMIC_RELAXED);
			}
		}
		if (mybad) {
			#pragma omp atomic write
			bad = 1;
		}
		#pragma omp barrier
		/* 2. exclusive scan: c[e] = first slot of entry e */
		const uint64_t lo = n_entries * (uint64_t)t / (uint64_t)T, hi = n_entries * (uint64_t)(t + 1) / (uint64_t)T;
		uint64_t sum = 0;
		for (uint64_t e = lo; e < hi; ++e) sum += c[e];
		bs[t + 1] = sum;
		#pragma omp barrier
		#pragma omp single
		for (int k = 0; k < T; ++k) bs[k + 1] += bs[k];
		uint64_t run = bs[t];
		for (uint64_t e = lo; e < hi; ++e) { const uint64_t k = c[e]; c[e] = run; run += k; }
		#pragma omp barrier
		if (!bad) {
			/* 3. scatter: afterwards c[e] = one past the last slot of entry e */
			for (int r = 0; r < n_src; ++r) {
				const BhipHit *h = src[r];
				#pragma omp for schedule(static) nowait
				for (uint64_t i = 0; i < cnt[r]; ++i) dst[__atomic_fetch_add(&c[h[i].q], 1, __ATOMIC_RELAXED)] = h[i];
			}
			#pragma omp barrier
			/* 4. the records of an entry by reference */
			#pragma omp for schedule(static)
			for (uint64_t e = 0; e < n_entries; ++e) {
				const uint64_t a = e ? c[e - 1] : 0, b = c[e];
				if (b - a > 24) qsort(dst + a, b - a, sizeof(*dst), by_ref);
				else for (uint64_t i = a + 1; i < b; ++i) {
					const BhipHit h = dst[i]; uint64_t k = i;
					while (k > a && dst[k - 1].refIx > h.refIx) { dst[k] = dst[k - 1]; --k; }
					dst[k] = h;
				}
			}
		}
	}
	free(c); free(bs);
	return bad ? bh_set_error(BH_E_INTERNAL, "a record refers to an entry beyond %lu", (unsigned long)n_entries) : BH_OK;
}
int bh_order_records(BhipHit *hits, uint64_t n, uint64_t n_entries) {
	if (n < 2) return n && hits[0].q >= n_entries ? bh_set_error(BH_E_INTERNAL, "a record refers to an entry beyond %lu", (unsigned long)n_entries) : BH_OK;
	/* scratch of the array's size: 2 MB-aligned and advised for huge pages (its first touch is the scatter itself) */
	const size_t bytes = ((size_t)n * sizeof(BhipHit) + ((size_t)1 << 21) - 1) & ~(((size_t)1 << 21) - 1);
	BhipHit *tmp = aligned_alloc((size_t)1 << 21, bytes);
	if (!tmp) return bh_set_error(BH_E_OOM, "OOM:order_records");
	(void)madvise(tmp, bytes, MADV_HUGEPAGE);
	const BhipHit *src = hits;
	const int rc = order_into(&src, &n, 1, tmp, n_entries);
	if (!rc) {
		#pragma omp parallel for schedule(static) num_threads(team())
		for (uint64_t blk = 0; blk < (n + 65535) / 65536; ++blk) {
			const uint64_t a = blk * 65536, b = a + 65536 < n ? a + 65536 : n;
			memcpy(hits + a, tmp + a, (b - a) * sizeof(*tmp));
		}
	}
	free(tmp);
	return rc;
}
/* element-wise minimum of the ranks' per-query minima into best[0] (all ranks in one process, no communicator) */
void bh_minima_merge(uint8_t *const *best, int n, uint64_t len) {
	#pragma omp parallel for schedule(static) num_threads(team())
	for (uint64_t blk = 0; blk < (len + 65535) / 65536; ++blk) {
		const uint64_t a = blk * 65536, b = a + 65536 < len ? a + 65536 : len;
		uint8_t *d = best[0];
		for (int i = 1; i < n; ++i) {
			const uint8_t *s = best[i];
			if (!s) continue;
			for (uint64_t k = a; k < b; ++k) d[k] = s[k] < d[k] ? s[k] : d[k];
		}
	}
}

int bh_search_multi(BhMultiRank *R, int n_local, int n_ranks, void *comm, const BhQueries *Q, BhMode mode, uint64_t batch, int shard_db, BhRun *all, uint64_t *counts) {
	return bh_search_multi_ex(R, n_local, n_ranks, comm, NULL, Q, mode, batch, shard_db, all, counts, NULL);
}
/* per-batch hook of a rank's bh_align_ranges (BhRun.onBatch): the batch's records, still resident on the device, into the send buffer of
 * the RCCL gather.  A failure is remembered by the communicator and announced with the counts of the gather. */
typedef struct { void *comm; int rank; int failed; } StageCb;
static void stage_batch_cb(void *ctx, void *hh, uint64_t first, uint64_t n) {
	StageCb *c = (StageCb *)ctx;
	uint64_t got = 0;
	if (bhip_comm_stage_device(c->comm, c->rank, hh, first, &got) || got != n) {
		if (!c->failed && getenv("BURST_HOST_DEBUG")) fprintf(stderr, "[bh_multi] rank %d: staging the records of a batch on the device failed (%s): the gather will upload the host copy\n", c->rank, bhip_last_error());
		c->failed = 1;
	}
}

int bh_search_multi_ex(BhMultiRank *R, int n_local, int n_ranks, void *comm, BhNode *node, const BhQueries *Q, BhMode mode, uint64_t batch, int shard_db, BhRun *all, uint64_t *counts,
                       BhRunView *view) {
	/* shard_db = number of database shards S (0 / 1: none).  S < n_ranks: the ranks form n_ranks / S replica groups of S shards each
	 * (rank = group * S + shard); the caller gives every rank its group's query ranges and its shard's first clump.  The minima
	 * are combined over ALL ranks at once: the groups' queries are disjoint and a rank says 255 ("none") for queries that are not
	 * its group's, so the minimum over everybody is the minimum over the group. */
	if (n_local < 1 || n_local > n_ranks || n_ranks > BH_MAX_RANKS) return bh_set_error(BH_E_USAGE, "bad rank layout (%d local of %d)", n_local, n_ranks);
	if (!comm && !node && n_local != n_ranks) return bh_set_error(BH_E_USAGE, "without a communicator every rank must live in this process");
	if (node && n_local != 1) return bh_set_error(BH_E_USAGE, "the shared-memory hand-over is for one rank per process");
	if (node && !comm && !R[0].reduce_min && shard_db > 1 && mode != BH_FORAGE && mode != BH_ANY && n_ranks > 1) return bh_set_error(BH_E_USAGE, "database-sharded ranks in different processes need a communicator (or the launcher's reduce_min) for the minima");
	const int dbg = getenv("BURST_HOST_DEBUG") != NULL;
	double tp[5]; tp[0] = omp_get_wtime();
	if (node) { int b = bh_node_begin(node); if (b) return b; bh_node_attach(node, &R[0].run); }
	int rcs[BH_MAX_RANKS]; char errs[BH_MAX_RANKS][512];
	uint8_t *best[BH_MAX_RANKS];
	for (int i = 0; i < n_local; ++i) R[i].gatherPath = 0;
	for (int i = 0; i < n_local; ++i) { rcs[i] = BH_E_INTERNAL; snprintf(errs[i], sizeof errs[i], "rank %d never ran (OpenMP gave the team fewer than %d threads)", R[i].rank, n_local); best[i] = NULL; }
	const int reduce = shard_db > 1 && mode != BH_FORAGE && mode != BH_ANY && n_ranks > 1;
	/* the per-query minima tables are taken NOW: a rank that found no memory for its table after the search would stay out of a
	 * collective its peers are in.  Here the call fails before anything has started (peers in other processes give up on this rank
	 * through the hand-over's time-out) */
	if (reduce) for (int i = 0; i < n_local; ++i) {
		best[i] = malloc(Q->numUniq + 1);
		if (!best[i]) { for (int k = 0; k < i; ++k) free(best[k]); if (node) (void)bh_node_publish(node, &R[0].run, BH_E_OOM); return bh_set_error(BH_E_OOM, "OOM:minima"); }
		memset(best[i], 255, Q->numUniq);
	}
	/* one host thread per local rank; the runtime must grant all of them -- a missing rank would leave the others waiting in the
	 * collectives -- so dynamic team sizes are switched off and the team is checked before anything is enqueued */
	const int dyn = omp_get_dynamic();
	omp_set_dynamic(0);
	int team_ok = 1;
	#pragma omp parallel num_threads(n_local)
	{
		#pragma omp single
		team_ok = omp_get_num_threads() == n_local;
	}
	if (!team_ok) {
		omp_set_dynamic(dyn);
		for (int i = 0; i < n_local; ++i) free(best[i]);
		if (node) (void)bh_node_publish(node, &R[0].run, BH_E_INTERNAL);      /* (the peers must not wait for this rank's records) */
		return bh_set_error(BH_E_INTERNAL, "OpenMP does not grant %d threads (OMP_THREAD_LIMIT?): one host thread per GPU is required", n_local);
	}
	/* With the RCCL gather of a query-sharded search the records go from device to device: every batch's records are copied into the
	 * communicator's send buffer while they are still resident (stage_batch_cb), so that the gather does not upload the host copy again
	 * (host -> device -> xGMI used to be the first two legs).  Database-sharded searches filter the records on the host first. */
	int staged_ok[BH_MAX_RANKS]; StageCb scb[BH_MAX_RANKS];
	for (int i = 0; i < n_local; ++i) {
		staged_ok[i] = 0;
		if (comm && !node && !R[i].align && !(shard_db > 1) && !getenv("BURST_HOST_GATHER_UPLOAD")) {
			scb[i].comm = comm; scb[i].rank = R[i].rank; scb[i].failed = 0;
			bhip_comm_stage_reset(comm, R[i].rank);
			R[i].run.onBatch = stage_batch_cb; R[i].run.onBatchCtx = &scb[i];
			staged_ok[i] = 1;
		}
	}
	/* 1. every rank aligns its share */
	#pragma omp parallel num_threads(n_local)
	{
		const int i = omp_get_thread_num();
		BhMultiRank *r = &R[i];
		const double t0 = omp_get_wtime();
		rcs[i] = r->align ? r->align(r->ctx, Q, r->r0, r->r1, r->n_ranges, (int)mode, batch, &r->run)
		                  : bh_align_ranges_reuse(r->hh, Q, r->r0, r->r1, r->n_ranges, mode, batch, &r->run);
		r->run.onBatch = NULL; r->run.onBatchCtx = NULL;
		r->secSearch = omp_get_wtime() - t0;
		if (rcs[i]) snprintf(errs[i], sizeof errs[i], "%s", r->align ? "the rank's align back end failed" : bh_last_error());
		else if (shard_db > 1) {
			for (uint64_t k = 0; k < r->run.nHits; ++k) r->run.hits[k].refIx += 16u * r->c0;
			if (reduce) shard_minima(Q, &r->run, best[i]);
		}
	}
	tp[1] = omp_get_wtime();
	/* (a rank that failed still has to walk through the collectives its peers are in: it takes part with nothing) */
	/* 2. database-sharded: the per-query minimum over all ranks */
	if (reduce) {
		if (comm || R[0].reduce_min) {
			/* (a rank whose search failed takes part with its table of 255s: nobody is left waiting) */
			#pragma omp parallel num_threads(n_local)
			{
				const int i = omp_get_thread_num();
				if (rcs[i]) memset(best[i], 255, Q->numUniq);
				const int rc = comm ? bhip_comm_allreduce_min(comm, R[i].rank, best[i], Q->numUniq) : R[i].reduce_min(R[i].ctx, best[i], Q->numUniq);
				if (rc && !rcs[i]) { rcs[i] = BH_E_DEVICE; snprintf(errs[i], sizeof errs[i], "%s%s", comm ? "libburst_hip: " : "the launcher's reduce_min failed", comm ? bhip_last_error() : ""); }
			}
		} else if (best[0]) {
			bh_minima_merge(best, n_local, Q->numUniq);
			for (int i = 1; i < n_local; ++i) if (best[i]) { free(best[i]); best[i] = NULL; }
		}
		#pragma omp parallel for schedule(dynamic, 1) num_threads(n_local)
		for (int i = 0; i < n_local; ++i) if (!rcs[i] && (best[i] || (!comm && best[0]))) shard_filter(Q, &R[i].run, best[i] ? best[i] : best[0]);
	}
	for (int i = 0; i < n_local; ++i) free(best[i]);
	tp[2] = omp_get_wtime();
	/* 3. the records to rank 0 */
	int i0 = -1;
	for (int i = 0; i < n_local; ++i) if (R[i].rank == 0) i0 = i;
	uint64_t tot_local = 0;
	for (int i = 0; i < n_local; ++i) tot_local += rcs[i] ? 0 : R[i].run.nHits;
	int rc = BH_OK, ordered = 0, viewed = 0;
	if (node) {
		/* one rank per process, the records already lie in this rank's shared-memory segment: say so; rank 0 takes everybody's */
		const int p = bh_node_publish(node, &R[0].run, rcs[0]);
		if (p && !rcs[0]) { rcs[0] = p; snprintf(errs[0], sizeof errs[0], "%s", bh_last_error()); }
		if (i0 >= 0) {
			/* query-sharded and the caller takes a view: the records stay where they are (rank 0 has every rank's segment mapped side
			 * by side) -- the hand-over is the word per rank.  Otherwise (database-sharded: they have to be put in order anyway; a
			 * search that outgrew its segment; no view wanted) one concatenation into `all`. */
			int c = BH_E_CAPACITY;
			if (view && !(shard_db > 1 && n_ranks > 1)) { c = bh_node_collect_view(node, view, counts); if (!c) { viewed = 1; all->nHits = view->total; } }
			if (c == BH_E_CAPACITY) c = bh_node_collect(node, all, counts);
			if (c && !rcs[0]) { rcs[0] = c; snprintf(errs[0], sizeof errs[0], "%s", bh_last_error()); }
		}
	} else if (comm) {
		/* rank 0's buffer is sized from its own share; when the gathered total does not fit (BHIP_E_CAPACITY) the records stay on
		 * rank 0's device and are fetched into a larger buffer without another collective */
		if (i0 >= 0 && bh_run_reserve(all, tot_local * (uint64_t)(n_ranks > n_local ? n_ranks / n_local : 1) + (tot_local >> 2) + (1u << 16))) {
			for (int i = 0; i < n_local; ++i) if (!rcs[i]) { rcs[i] = BH_E_OOM; snprintf(errs[i], sizeof errs[i], "OOM:hits"); }
		}
		int again = 0; uint64_t need = 0;
		#pragma omp parallel num_threads(n_local)
		{
			const int i = omp_get_thread_num();
			uint64_t n_total = 0;
			/* the records were staged on the device batch by batch (stage_batch_cb): sent from there; otherwise the host copy goes up again */
			int g = staged_ok[i] ? bhip_comm_gather_staged(comm, R[i].rank, rcs[i] ? 0 : R[i].run.nHits, i == i0 ? all->hits : NULL, i == i0 ? all->capHits : 0, &n_total, i == i0 ? counts : NULL) : 1;
			/* (a rank that could not stage -- device memory -- says so with its count: the call fails on every rank, and every rank comes here) */
			R[i].gatherPath = (!g || g == BHIP_E_CAPACITY) ? 1 : 2;
			if (staged_ok[i] && R[i].gatherPath == 2 && dbg) fprintf(stderr, "[bh_multi] rank %d: the device-fed gather was refused (%s)\n", R[i].rank, bhip_last_error());
			if (g && g != BHIP_E_CAPACITY) g = bhip_comm_gather_hits(comm, R[i].rank, rcs[i] ? NULL : R[i].run.hits, rcs[i] ? 0 : R[i].run.nHits, i == i0 ? all->hits : NULL,
			                                    i == i0 ? all->capHits : 0, &n_total, i == i0 ? counts : NULL);
			if (i == i0) need = n_total;
			if (g == BHIP_E_CAPACITY && i == i0) again = 1;
			else if (g && g != BHIP_E_CAPACITY && !rcs[i]) { rcs[i] = BH_E_DEVICE; snprintf(errs[i], sizeof errs[i], "libburst_hip: %s", bhip_last_error()); }
		}
		if (i0 >= 0 && again && !rcs[i0]) {
			if (bh_run_reserve(all, need + 1)) { rcs[i0] = BH_E_OOM; snprintf(errs[i0], sizeof errs[i0], "OOM:hits"); }
			else if (bhip_comm_fetch_gathered(comm, all->hits, all->capHits, &need)) { rcs[i0] = BH_E_DEVICE; snprintf(errs[i0], sizeof errs[i0], "libburst_hip: %s", bhip_last_error()); }
		}
		if (i0 >= 0) all->nHits = need;
	} else {
		/* all ranks in this process: their runs are in host memory already.  Query-sharded and the caller takes a view: when the runs
		 * lie a whole number of records apart (the caller gave the ranks slices of one block) they are read where they lie */
		if (view && !(shard_db > 1 && n_ranks > 1)) {
			const char *base = NULL;
			for (int i = 0; i < n_local; ++i) if (!rcs[i] && R[i].run.nHits && (!base || (const char *)R[i].run.hits < base)) base = (const char *)R[i].run.hits;
			int ok = 1;
			for (int i = 0; i < n_local; ++i) if (!rcs[i] && R[i].run.nHits && ((const char *)R[i].run.hits - base) % (ptrdiff_t)sizeof(BhipHit)) ok = 0;
			if (ok) {
				view->base = (const BhipHit *)base; view->n_runs = n_local; view->total = 0;
				for (int i = 0; i < n_local; ++i) {
					const uint64_t n = rcs[i] ? 0 : R[i].run.nHits;
					view->off[i] = n ? (uint64_t)(((const char *)R[i].run.hits - base) / (ptrdiff_t)sizeof(BhipHit)) : 0; view->n[i] = n; view->total += n;
					if (counts) counts[R[i].rank] = n;
				}
				all->nHits = view->total;
				viewed = 1;
			}
		}
		if (viewed) { }
		else if (bh_run_reserve_plain(all, tot_local + 1)) rc = bh_set_error(BH_E_OOM, "OOM:hits");
		else {
			uint64_t o = 0, at[BH_MAX_RANKS], cnt[BH_MAX_RANKS]; const BhipHit *src[BH_MAX_RANKS];
			for (int i = 0; i < n_local; ++i) {      /* local ranks are listed in rank order */
				cnt[i] = rcs[i] ? 0 : R[i].run.nHits; src[i] = R[i].run.hits;
				if (counts) counts[R[i].rank] = cnt[i];
				at[i] = o; o += cnt[i];
			}
			if (shard_db > 1 && n_ranks > 1) {      /* database-sharded: straight into (query entry, reference) order */
				rc = order_into(src, cnt, n_local, all->hits, Q->numEntries);
				ordered = 1;
			} else {
				#pragma omp parallel for schedule(dynamic, 1) num_threads(n_local)
				for (int i = 0; i < n_local; ++i) if (cnt[i]) memcpy(all->hits + at[i], src[i], cnt[i] * sizeof(BhipHit));
			}
			all->nHits = o;
		}
	}
	tp[3] = omp_get_wtime();
	omp_set_dynamic(dyn);
	for (int i = 0; i < n_local; ++i) if (rcs[i]) return bh_set_error(rcs[i], "%s", errs[i]);
	if (rc) return rc;
	if (i0 >= 0) {
		for (int i = 0; i < n_local; ++i) { all->nBatches += R[i].run.nBatches; all->secAlign += i == i0 ? R[i].run.secAlign : 0; }
		if (shard_db > 1 && n_ranks > 1 && !ordered) rc = bh_order_records(all->hits, all->nHits, Q->numEntries);
	}
	if (view && !viewed && !rc) { view->base = all->hits; view->n_runs = 1; view->off[0] = 0; view->n[0] = all->nHits; view->total = all->nHits; }
	if (dbg) fprintf(stderr, "[bh_search_multi] rank %d of %d: align %.4f s, minima + filter %.4f s, hand-over %.4f s, order %.4f s\n", R[0].rank, n_ranks, tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], omp_get_wtime() - tp[3]);
	return rc;
}

/* A database of S shards on FEWER devices than shards -- here: on one.  The shards take turns on the device: shard s goes up (with
 * -ad its accelerator is built there from the slice alone), ALL queries are aligned against it, its records and per-query minima
 * come back to host memory, the device is given to the next shard.  What bh_search_multi_ex does between ranks happens between
 * turns: minimum over the shards (Sb->ed over the whole database, burst.c:4217-4277), what lies above it dropped (not in FORAGE),
 * the rest put in (query entry, reference) order -- the records one device holding everything produces.  A database larger than the
 * device is served this way at the price of S uploads; `secs[s]` = align phase of shard s, `up[s]` = its upload (+ build). */
int bh_search_serial_shards(const BhDb *db, int device, int n_shards, int z, int build_K, const BhQueries *Q, BhMode mode, uint64_t batch, BhRun *all, double *secs, double *up) {
	if (n_shards < 1 || n_shards > BH_MAX_RANKS) return bh_set_error(BH_E_USAGE, "bad number of shards (%d)", n_shards);
	BhRun runs[BH_MAX_RANKS]; memset(runs, 0, sizeof runs);
	uint8_t *best = NULL, *mine = NULL;
	const int reduce = n_shards > 1 && mode != BH_FORAGE && mode != BH_ANY;
	int rc = BH_OK;
	if (reduce) {
		best = malloc(Q->numUniq + 1); mine = malloc(Q->numUniq + 1);
		if (!best || !mine) { free(best); free(mine); return bh_set_error(BH_E_OOM, "OOM:minima"); }
		memset(best, 255, Q->numUniq);
	}
	const uint64_t strands = Q->numEntries > Q->numUniq ? 2 : 1;
	for (int s = 0; s < n_shards && !rc; ++s) {
		uint32_t c0, c1;
		bh_clump_shard(db, n_shards, s, &c0, &c1);
		BhDb slice; memset(&slice, 0, sizeof slice);
		void *hh = NULL;
		double t0 = omp_get_wtime();
		if ((rc = bh_db_slice(db, c0, c1, &slice))) break;
		if ((rc = bh_device_open_ex(&slice, device, z, build_K, &hh))) { bh_db_free(&slice); break; }
		const uint64_t B = Q->numUniq < batch ? Q->numUniq : batch;
		if (B) (void)bhip_reserve_symbols(hh, (uint32_t)(B * strands), Q->maxLen, 0);
		if (up) up[s] = omp_get_wtime() - t0;
		t0 = omp_get_wtime();
		const uint64_t u0 = 0, u1 = Q->numUniq;
		rc = bh_align_ranges(hh, Q, &u0, &u1, 1, mode, batch, &runs[s]);
		if (secs) secs[s] = omp_get_wtime() - t0;
		bhip_destroy(hh);
		bh_db_free(&slice);
		if (rc) break;
		for (uint64_t k = 0; k < runs[s].nHits; ++k) runs[s].hits[k].refIx += 16u * c0;
		if (reduce) {
			shard_minima(Q, &runs[s], mine);
			uint8_t *both[2] = {best, mine};
			bh_minima_merge(both, 2, Q->numUniq);
		}
		all->nBatches += runs[s].nBatches; all->secAlign += runs[s].secAlign;
		all->total.n_pairs += runs[s].total.n_pairs;
	}
	if (!rc) {
		const BhipHit *src[BH_MAX_RANKS]; uint64_t cnt[BH_MAX_RANKS], tot = 0;
		for (int s = 0; s < n_shards; ++s) { if (reduce) shard_filter(Q, &runs[s], best); src[s] = runs[s].hits; cnt[s] = runs[s].nHits; tot += runs[s].nHits; }
		if (bh_run_reserve_plain(all, tot + 1)) rc = bh_set_error(BH_E_OOM, "OOM:hits");
		else if (n_shards == 1) { if (tot) memcpy(all->hits, src[0], tot * sizeof(BhipHit)); all->nHits = tot; }
		else { rc = order_into(src, cnt, n_shards, all->hits, Q->numEntries); all->nHits = tot; }
	}
	for (int s = 0; s < n_shards; ++s) bh_run_free(&runs[s]);
	free(best); free(mine);
	return rc;
}

