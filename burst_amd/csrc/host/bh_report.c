/* bh_report.c -- per-mode consolidation of hit records into .b6 lines (tail of do_alignments, burst.c:4553-4891;
 * host-side spec in SURVEY.md Appendix D).
 *
 * The reference keeps, per unique query, a LIFO list of ResultPods (list prepended at burst.c:4231, 4447).  Its
 * order leaks into the output only through ties (DUPE_HUNT keeps the first of two overlapping placements on one
 * original reference; CAPITALIST keeps the first of several equally voted placements), and it depends on the path:
 *   - accelerated path: one list per strand (burst.c:4218), clumps visited by descending bunch k-mer count (4130), the
 *     reverse list appended to the forward one (4299-4312).  The visit order depends on the thread count (QBUNCH,
 *     4019-4021), so the reference itself is not reproducible there; we use [forward, refIx descending] ++
 *     [reverse, refIx descending].
 *   - exhaustive path: one list per unique query for both strands (ai = six, burst.c:4368), clumps ascending, entries
 *     in sorted order, lanes ascending (4344-4476).  With one thread this is deterministic and BH_REP_MERGED_LIST
 *     reproduces it exactly: clump descending; inside a clump the strand whose sequence sorts later first; lanes descending.
 */
#include "burst_host.h"
#include <stdlib.h>
#include <string.h>

typedef struct { const BhipHit *h; } Pod;

static inline void coords(const BhDb *db, const BhipHit *rp, uint32_t rix, uint32_t qlen, uint32_t *st, uint32_t *ed) {
	uint32_t mOff = db->refStart ? db->refStart[rix] : 0;
	uint32_t a = rp->finalPos - qlen + rp->gapR + mOff, b = rp->finalPos + mOff;          /* burst.c:4869-4871 */
	if (rp->rc) { *st = b; *ed = a; } else { *st = a; *ed = b; }
}

static void print_line(FILE *out, const char *qh, const char *rh, const BhipHit *rp, uint32_t qlen, uint32_t st, uint32_t ed, uint64_t col12) {
	uint32_t numGap = (uint32_t)rp->gapR + rp->gapQ, numMis = rp->ed - numGap, alLen = qlen + numGap;
	float pct = rp->score * 100;                                                           /* f32 product, then %f (burst.c:4555) */
	fprintf(out, "%s\t%s\t%f\t%u\t%u\t%u\t%u\t%u\t%d\t%u\t%u\t%lu\n", qh, rh, pct, alLen, numMis, numGap, 1, qlen, (int)st, ed,
	        (unsigned)rp->ed, (unsigned long)col12);
}

/* DUPE_HUNT (burst.c:4563-4570): reject a (hit, rix) whose original reference and start lie within qlen/2 of an
 * accepted one.  wide = 1 reproduces the 64-bit ql2 of the ALLPATHS/FORAGE blocks, 0 the 32-bit one of CAPITALIST. */
static int g_nodupe;   /* BH_REP_NO_DUPE_HUNT of the current bh_report_ex call (the report is single-threaded) */
static int dupe_hunt(uint32_t *RC, uint32_t *SC, uint64_t *ddix, uint32_t mapped, uint32_t st, uint32_t ql2, int wide) {
	if (g_nodupe) return 0;
	for (uint64_t d = 0; d < *ddix; ++d) {
		if (RC[d] != mapped) continue;
		if (wide) { if ((uint64_t)SC[d] + ql2 > st && (uint64_t)SC[d] < (uint64_t)st + ql2) return 1; }
		else if ((uint32_t)(SC[d] + ql2) > st && SC[d] < (uint32_t)(st + ql2)) return 1;
	}
	RC[*ddix] = mapped; SC[(*ddix)++] = st;
	return 0;
}

int bh_report(FILE *out, const BhDb *db, const BhQueries *Q, const BhipHit *hits, uint64_t nHits, BhMode mode, uint64_t *nLines) {
	return bh_report_ex(out, db, Q, hits, nHits, mode, 0, nLines);
}

/* 1 if the reverse-complement entry of unique query i sorts after its forward entry (strcmp order of the code strings) */
static int rc_sorts_later(const BhQueries *Q, uint64_t i) {
	const uint8_t *f = Q->codes + Q->qoff[i], *r = Q->codes + Q->qoff[Q->numUniq + i];
	return memcmp(r, f, Q->len[i]) > 0;
}

int bh_report_ex(FILE *out, const BhDb *db, const BhQueries *Q, const BhipHit *hits, uint64_t nHits, BhMode mode, int flags, uint64_t *nLines) {
	const uint64_t nU = Q->numUniq, nE = Q->numEntries;
	const int merged = flags & BH_REP_MERGED_LIST, nodupe = flags & BH_REP_NO_DUPE_HUNT;
	uint64_t lines = 0;
	g_nodupe = nodupe;
	/* per-entry ranges (records of one entry are contiguous) */
	uint64_t *start = calloc(nE + 1, sizeof(*start)); uint32_t *count = calloc(nE + 1, sizeof(*count));
	if (!start || !count) { free(start); free(count); return bh_set_error(BH_E_OOM, "OOM:report"); }
	for (uint64_t k = 0; k < nHits; ++k) {
		const uint32_t e = hits[k].q;
		if (e >= nE) { free(start); free(count); return bh_set_error(BH_E_INTERNAL, "hit refers to entry %u of %lu", e, (unsigned long)nE); }
		if (!count[e]) start[e] = k;
		else if (start[e] + count[e] != k) { free(start); free(count); return bh_set_error(BH_E_INTERNAL, "hit records of entry %u are not contiguous", e); }
		++count[e];
	}
	uint32_t maxIX = 0;
	for (uint32_t i = 0; i < db->totR; ++i) if (db->refIxSrt[i] > maxIX) maxIX = db->refIxSrt[i];
	const uint64_t numBins = (uint64_t)maxIX + 1;
	/* the reference sizes its per-query caches by maxIX+1 (burst.c:4583-4586, 4698-4699); size them by the largest
	 * possible number of (hit, reference) expansions of one query as well, so that they cannot overflow */
	uint32_t maxDup = 1, maxList0 = 0;
	if (db->refDedupIx) for (uint32_t i = 0; i < db->totR; ++i) if (db->refDedupIx[i + 1] - db->refDedupIx[i] > maxDup) maxDup = db->refDedupIx[i + 1] - db->refDedupIx[i];
	for (uint64_t i = 0; i < nU; ++i) { uint32_t n = count[i] + (nE > nU ? count[nU + i] : 0); if (n > maxList0) maxList0 = n; }
	uint64_t capX = numBins + 1;
	if ((uint64_t)maxList0 * maxDup + 1 > capX) capX = (uint64_t)maxList0 * maxDup + 1;
	uint32_t *RefCache = malloc(capX * 4 + 4), *StCache = malloc(capX * 4 + 4), *RIXcache = malloc(capX * 4 + 4);
	const BhipHit **RPcache = malloc((capX + 1) * sizeof(*RPcache));
	size_t *RefCounts = mode == BH_CAPITALIST ? calloc(numBins + 1, sizeof(*RefCounts)) : NULL;
	uint32_t maxList = 0;
	for (uint64_t i = 0; i < nU; ++i) {
		uint32_t n = count[i] + (nE > nU ? count[nU + i] : 0);
		if (n > maxList) maxList = n;
	}
	const BhipHit **list = malloc(((size_t)maxList + 1) * sizeof(*list));
	if (!RefCache || !StCache || !RIXcache || !RPcache || !list || (mode == BH_CAPITALIST && !RefCounts)) {
		free(start); free(count); free(RefCache); free(StCache); free(RIXcache); free(RPcache); free(RefCounts); free(list);
		return bh_set_error(BH_E_OOM, "OOM:report");
	}
	#define MAPPED(rix) (db->identityMap ? (rix) : db->refMap[rix])
	#define BUILD_LIST(i, n) do { n = 0; \
		if (merged && nE > nU) { \
			/* merge the two descending runs by clump; inside one clump the later-sorted strand comes first */ \
			const int rcFirst = rc_sorts_later(Q, i); \
			int64_t a_ = (int64_t)count[i] - 1, b_ = (int64_t)count[nU + (i)] - 1; \
			while (a_ >= 0 || b_ >= 0) { \
				const BhipHit *ha = a_ >= 0 ? hits + start[i] + a_ : NULL, *hb = b_ >= 0 ? hits + start[nU + (i)] + b_ : NULL; \
				int takeA; \
				if (!hb) takeA = 1; else if (!ha) takeA = 0; \
				else if ((ha->refIx >> 4) != (hb->refIx >> 4)) takeA = (ha->refIx >> 4) > (hb->refIx >> 4); \
				else takeA = !rcFirst; \
				if (takeA) { list[n++] = ha; --a_; } else { list[n++] = hb; --b_; } \
			} \
		} else { \
			for (uint32_t k_ = count[i]; k_ > 0; --k_) list[n++] = hits + start[i] + k_ - 1; \
			if (nE > nU) for (uint32_t k_ = count[nU + (i)]; k_ > 0; --k_) list[n++] = hits + start[nU + (i)] + k_ - 1; \
		} } while (0)
	/* expansion over exact-duplicate references (RefDedupIx) or the single representative */
	#define FOR_EXPANSIONS(rp, rixvar, ...) do { \
		if (db->refDedupIx) { for (uint32_t k_ = db->refDedupIx[(rp)->refIx]; k_ < db->refDedupIx[(rp)->refIx + 1]; ++k_) { uint32_t rixvar = db->tmpRIX[k_]; __VA_ARGS__ } } \
		else { uint32_t rixvar = db->refIxSrt[(rp)->refIx]; __VA_ARGS__ } } while (0)

	if (mode == BH_CAPITALIST) {   /* pass A: one vote per unique query and accepted (hit, reference) (burst.c:4700-4727) */
		for (uint64_t i = 0; i < nU; ++i) {
			uint32_t n; BUILD_LIST(i, n);
			if (!n) continue;
			uint32_t b = 0;
			for (uint32_t k = 1; k < n; ++k) if (list[k]->ed < list[b]->ed) b = k;
			uint64_t ddix = 0; const uint32_t qlen = Q->len[i], ql2 = qlen >> 1;
			for (uint32_t k = b; k < n; ++k) {
				const BhipHit *rp = list[k];
				if (rp->ed != list[b]->ed) continue;
				FOR_EXPANSIONS(rp, rix, {
					uint32_t st, ed; coords(db, rp, rix, qlen, &st, &ed); (void)ed;
					uint32_t mapped = MAPPED(rix);
					if (!dupe_hunt(RefCache, StCache, &ddix, mapped, rp->rc ? st : st, ql2, 0)) ++RefCounts[mapped];
				});
			}
		}
	}
	for (uint64_t i = 0; i < nU; ++i) {
		uint32_t n; BUILD_LIST(i, n);
		if (!n) continue;
		const uint32_t qlen = Q->len[i], ql2 = qlen >> 1;
		if (mode == BH_BEST) {                                               /* burst.c:4850-4890 */
			const BhipHit *best = list[0];
			for (uint32_t k = 1; k < n; ++k) {
				const BhipHit *rp = list[k];
				if (rp->ed < best->ed || (rp->ed == best->ed && rp->score > best->score) ||
				    (rp->ed == best->ed && rp->score == best->score && db->refIxSrt[rp->refIx] < db->refIxSrt[best->refIx])) best = rp;
			}
			const uint32_t rix = db->refIxSrt[best->refIx];
			uint32_t st, ed; coords(db, best, rix, qlen, &st, &ed);
			for (uint64_t j = Q->offset[i]; j < Q->offset[i + 1]; ++j) { print_line(out, Q->heads[j], db->refHead[rix], best, qlen, st, ed, i); ++lines; }
		} else if (mode == BH_ANY) {                                         /* any valid hit; column 12 = duplicate flag (burst.c:4268-4272) */
			const BhipHit *rp = list[0];
			const uint32_t rix = db->refIxSrt[rp->refIx];
			uint32_t st, ed; coords(db, rp, rix, qlen, &st, &ed);
			for (uint64_t j = Q->offset[i]; j < Q->offset[i + 1]; ++j) { print_line(out, Q->heads[j], db->refHead[rix], rp, qlen, st, ed, j > Q->offset[i]); ++lines; }
		} else if (mode == BH_ALLPATHS || mode == BH_FORAGE) {               /* burst.c:4582-4640, 4642-4692 */
			uint32_t b = 0;
			if (mode == BH_ALLPATHS) {
				for (uint32_t k = 1; k < n; ++k) if (list[k]->ed < list[b]->ed) b = k;
				if (!(list[b]->score != 0.0f)) continue;                        /* `if (rp->score)` burst.c:4598 */
			}
			uint64_t ddix = 0, rix_ix = 0;
			for (uint32_t k = b; k < n; ++k) {
				const BhipHit *rp = list[k];
				if (mode == BH_ALLPATHS && rp->ed != list[b]->ed) continue;
				FOR_EXPANSIONS(rp, rix, {
					uint32_t st, ed; coords(db, rp, rix, qlen, &st, &ed); (void)ed;
					if (!dupe_hunt(RefCache, StCache, &ddix, MAPPED(rix), st, ql2, 1)) { RPcache[rix_ix] = rp; RIXcache[rix_ix++] = rix; }
				});
			}
			for (uint64_t j = Q->offset[i]; j < Q->offset[i + 1]; ++j) for (uint64_t zz = 0; zz < rix_ix; ++zz) {
				const BhipHit *rp = RPcache[zz]; const uint32_t rix = RIXcache[zz];
				uint32_t st, ed; coords(db, rp, rix, qlen, &st, &ed);
				print_line(out, Q->heads[j], db->refHead[rix], rp, qlen, st, ed, i); ++lines;
			}
		} else {                                                             /* CAPITALIST pass B (burst.c:4746-4779, 4831-4843) */
			uint32_t b = 0;
			for (uint32_t k = 1; k < n; ++k) if (list[k]->ed < list[b]->ed) b = k;
			const BhipHit *best = list[b]; uint32_t bestmap = 0, bestrix = 0; int have = 0;
			uint64_t ddix = 0;
			const BhipHit *first = list[b];
			for (uint32_t k = b; k < n; ++k) {
				const BhipHit *rp = list[k];
				if (rp->ed > best->ed) continue;
				FOR_EXPANSIONS(rp, rix, {
					uint32_t st, ed; coords(db, rp, rix, qlen, &st, &ed); (void)ed;
					uint32_t mapped = MAPPED(rix);
					if (!dupe_hunt(RefCache, StCache, &ddix, mapped, st, ql2, 0)) {
						if (best == rp || RefCounts[mapped] > RefCounts[bestmap] || (RefCounts[mapped] == RefCounts[bestmap] && mapped < bestmap)) {
							best = rp; bestmap = mapped; bestrix = rix; have = 1;
						}
					}
				});
			}
			(void)first;
			if (!have) continue;   /* cannot happen: the first expansion of the first pod is always accepted */
			uint32_t st, ed; coords(db, best, bestrix, qlen, &st, &ed);
			for (uint64_t j = Q->offset[i]; j < Q->offset[i + 1]; ++j) { print_line(out, Q->heads[j], db->refHead[bestrix], best, qlen, st, ed, i); ++lines; }
		}
	}
	free(start); free(count); free(RefCache); free(StCache); free(RIXcache); free(RPcache); free(RefCounts); free(list);
	if (nLines) *nLines = lines;
	return BH_OK;
}
