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
#define _GNU_SOURCE            /* open_memstream */
#include "burst_host.h"
#include <stdlib.h>
#include <string.h>
#include <omp.h>
#include <unistd.h>
#include <sys/types.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <fcntl.h>

typedef struct { const BhipHit *h; } Pod;

static inline void coords(const BhDb *db, const BhipHit *rp, uint32_t rix, uint32_t qlen, uint32_t *st, uint32_t *ed) {
	uint32_t mOff = db->refStart ? db->refStart[rix] : 0;
	uint32_t a = rp->finalPos - qlen + rp->gapR + mOff, b = rp->finalPos + mOff;          /* burst.c:4869-4871 */
	if (rp->rc) { *st = b; *ed = a; } else { *st = a; *ed = b; }
}

/* One chunk's text.  The lines are formatted by hand -- a dozen integer conversions and one "%f" per line through fprintf were
 * most of the report's time -- and must be byte-identical to the reference's fprintf (burst.c:4553-4562). */
typedef struct { char *p; size_t len, cap; int oom; } LineBuf;
static inline int lb_room(LineBuf *b, size_t n) {
	if (b->len + n <= b->cap) return 1;
	size_t nc = b->cap ? b->cap * 2 : (1u << 20);
	while (nc < b->len + n) nc *= 2;
	char *np = realloc(b->p, nc);
	if (!np) { b->oom = 1; return 0; }
	b->p = np; b->cap = nc;
	return 1;
}
static inline void lb_str(LineBuf *b, const char *s) { const size_t n = strlen(s); if (lb_room(b, n + 1)) { memcpy(b->p + b->len, s, n); b->len += n; } }
static inline void lb_chr(LineBuf *b, char c) { if (lb_room(b, 1)) b->p[b->len++] = c; }
static inline void lb_u64(LineBuf *b, uint64_t v) {
	char t[24]; int n = 0;
	do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
	if (lb_room(b, (size_t)n)) while (n) b->p[b->len++] = t[--n];
}
static inline void lb_i32(LineBuf *b, int32_t v) { if (v < 0) { lb_chr(b, '-'); lb_u64(b, (uint64_t)(-(int64_t)v)); } else lb_u64(b, (uint64_t)v); }
/* "%f" of a float promoted to double: a float below 2^24 times 10^6 = 15625 * 2^6 is exact in a double (24 + 14 significant
 * bits), so rounding that product to the nearest integer, ties to even, IS the correctly rounded six-decimal expansion printf
 * gives.  Anything else (negative, huge, not finite) goes through snprintf. */
static inline void lb_pct(LineBuf *b, float pct) {
	if (!(pct >= 0.0f && pct < 16777216.0f)) { char t[64]; snprintf(t, sizeof t, "%f", pct); lb_str(b, t); return; }
	const uint64_t n = (uint64_t)__builtin_nearbyint((double)pct * 1e6);
	lb_u64(b, n / 1000000u);
	if (lb_room(b, 7)) {
		uint32_t fr = (uint32_t)(n % 1000000u);
		char *o = b->p + b->len;
		o[0] = '.';
		for (int k = 6; k >= 1; --k) { o[k] = (char)('0' + fr % 10); fr /= 10; }
		b->len += 7;
	}
}
/* (test entry: the identities of n records as the report prints them, one per line; returns the text length or 0) */
uint64_t bh_report_format_identities(const float *score, uint64_t n, char *out, uint64_t cap) {
	LineBuf b = {NULL, 0, 0, 0};
	for (uint64_t i = 0; i < n; ++i) { lb_pct(&b, score[i] * 100); lb_chr(&b, '\n'); }
	uint64_t len = b.oom || b.len > cap ? 0 : b.len;
	if (len) memcpy(out, b.p, len);
	free(b.p);
	return len;
}
static void print_line_tax(LineBuf *out, const char *qh, const char *rh, const BhipHit *rp, uint32_t qlen, uint32_t st, uint32_t ed, uint64_t col12,
                           int with_tax, const char *taxon) {
	uint32_t numGap = (uint32_t)rp->gapR + rp->gapQ, numMis = rp->ed - numGap, alLen = qlen + numGap;
	float pct = rp->score * 100;                                                           /* f32 product, then %f (burst.c:4555) */
	/* "%s\t%s\t%f\t%u\t%u\t%u\t%u\t%u\t%d\t%u\t%u\t%lu[\t%s]\n" (PRINT_MATCH / PRINT_MATCH_TAX, burst.c:4553-4562) */
	lb_str(out, qh); lb_chr(out, '\t'); lb_str(out, rh); lb_chr(out, '\t'); lb_pct(out, pct); lb_chr(out, '\t');
	lb_u64(out, alLen); lb_chr(out, '\t'); lb_u64(out, numMis); lb_chr(out, '\t'); lb_u64(out, numGap); lb_chr(out, '\t'); lb_chr(out, '1'); lb_chr(out, '\t');
	lb_u64(out, qlen); lb_chr(out, '\t'); lb_i32(out, (int32_t)st); lb_chr(out, '\t'); lb_u64(out, ed); lb_chr(out, '\t'); lb_u64(out, (unsigned)rp->ed); lb_chr(out, '\t');
	lb_u64(out, col12);
	if (with_tax) { lb_chr(out, '\t'); lb_str(out, taxon ? taxon : "(null)"); }
	lb_chr(out, '\n');
}
static void print_line(LineBuf *out, const char *qh, const char *rh, const BhipHit *rp, uint32_t qlen, uint32_t st, uint32_t ed, uint64_t col12) {
	print_line_tax(out, qh, rh, rp, qlen, st, ed, col12, 0, NULL);
}

/* identity a placement needs to keep taxonomic level lm + 1 (TAXLEVELS, burst.c:264-266).  The reference scans the table
 * without a bound; for identities above its last entry the scan runs into whatever follows it in memory (the LENIENT table
 * after the STRICT one, then a pointer).  The sentinel reproduces where that scan stops for every identity <= 1. */
static const float LEVELS_STRICT[] = {.65f, .75f, .78f, .82f, .86f, .94f, .98f, .995f, .55f, .70f, .75f, .80f, .84f, .93f, .97f, .985f, 3.0e38f};
static const float LEVELS_LENIENT[] = {.55f, .70f, .75f, .80f, .84f, .93f, .97f, .985f, 3.0e38f};
static int cmp_str(const void *a, const void *b) { return strcmp(*(char *const *)a, *(char *const *)b); }

/* DUPE_HUNT (burst.c:4563-4570): reject a (hit, rix) whose original reference and start lie within qlen/2 of an
 * accepted one.  wide = 1 reproduces the 64-bit ql2 of the ALLPATHS/FORAGE blocks, 0 the 32-bit one of CAPITALIST. */
static int dupe_hunt(int nodupe, uint32_t *RC, uint32_t *SC, uint64_t *ddix, uint32_t mapped, uint32_t st, uint32_t ql2, int wide) {
	if (nodupe) return 0;      /* BH_REP_NO_DUPE_HUNT of this call */
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
	return bh_report_tax(out, db, Q, hits, nHits, mode, flags, NULL, nLines);
}

int bh_report_tax(FILE *out, const BhDb *db, const BhQueries *Q, const BhipHit *hits, uint64_t nHits, BhMode mode, int flags, const BhTaxOpts *tx,
                  uint64_t *nLines) {
	BhRunView v; v.base = hits; v.n_runs = 1; v.off[0] = 0; v.n[0] = nHits; v.total = nHits;
	return bh_report_view(out, db, Q, &v, mode, flags, tx, nLines);
}
/* The records may lie in several runs of one address range (ranks of a node whose record buffers rank 0 maps side by side,
 * bh_node.c): everything below reaches a record as hits + start[entry] + k, so only the pass that finds the entries' ranges
 * walks the runs. */
int bh_report_view(FILE *out, const BhDb *db, const BhQueries *Q, const BhRunView *view, BhMode mode, int flags, const BhTaxOpts *tx,
                   uint64_t *nLines) {
	const BhipHit *hits = view->base;
	const uint64_t nU = Q->numUniq, nE = Q->numEntries;
	FILE *const real_out = out;
	const BhTax *T = tx ? tx->tax : NULL;
	const int wt = T != NULL, ncbi = tx ? tx->ncbi : 0, suppress = tx ? tx->suppress : 0;
	const uint32_t taxacut = tx && tx->taxacut >= 2 ? tx->taxacut : 10;
	const float *LEVELS = tx && tx->strict ? LEVELS_STRICT : LEVELS_LENIENT;
	const int merged = flags & BH_REP_MERGED_LIST, nodupe = flags & BH_REP_NO_DUPE_HUNT;
	uint64_t lines = 0;
	const int dbg = getenv("BURST_HOST_DEBUG") != NULL;
	const double t_begin = omp_get_wtime();
	double t_render = 0, t_write = 0;
	/* per-entry ranges (records of one entry are contiguous) */
	uint64_t *start = calloc(nE + 1, sizeof(*start)); uint32_t *count = calloc(nE + 1, sizeof(*count));
	if (!start || !count) { free(start); free(count); return bh_set_error(BH_E_OOM, "OOM:report"); }
	for (int r = 0; r < view->n_runs; ++r) for (uint64_t k = view->off[r]; k < view->off[r] + view->n[r]; ++k) {
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
	size_t *RefCounts = mode == BH_CAPITALIST ? calloc(numBins + 1, sizeof(*RefCounts)) : NULL;
	uint32_t maxList = 0;
	for (uint64_t i = 0; i < nU; ++i) {
		uint32_t n = count[i] + (nE > nU ? count[nU + i] : 0);
		if (n > maxList) maxList = n;
	}
	if (mode == BH_CAPITALIST && !RefCounts) { free(start); free(count); return bh_set_error(BH_E_OOM, "OOM:report"); }
	/* Queries are independent once the CAPITALIST votes are in: chunks of unique queries are rendered by a team of threads,
	 * each into its own memory stream with its own scratch, and written out in order.  The one sequential case is BEST with
	 * -b -bs, where an empty taxonomy inherits the string of the previous query (burst.c:4852-4885). */
	int nThreads = omp_get_max_threads() > 32 ? 32 : omp_get_max_threads();
	uint64_t chunkQ = 8192;
	{	/* test hook: BURST_HOST_REPORT_THREADS=<n>[:<queries per chunk>] forces the threaded path on small inputs */
		const char *ev = getenv("BURST_HOST_REPORT_THREADS");
		if (ev && atoi(ev) > 0) { nThreads = atoi(ev); const char *c = strchr(ev, ':'); if (c && atoll(c + 1) > 0) chunkQ = (uint64_t)atoll(c + 1); }
		else if (nU < 4096) nThreads = 1;
	}
	if (wt && suppress && mode == BH_BEST) nThreads = 1;
	int oom = 0, wr = 0;
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
		#pragma omp parallel num_threads(nThreads)
		{
			uint32_t *RefCache = malloc(capX * 4 + 4), *StCache = malloc(capX * 4 + 4);
			const BhipHit **list = malloc(((size_t)maxList + 1) * sizeof(*list));
			if (!RefCache || !StCache || !list) {
				#pragma omp atomic write
				oom = 1;
			}
			#pragma omp barrier
			if (!oom) {
				#pragma omp for schedule(static)
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
							if (!dupe_hunt(nodupe, RefCache, StCache, &ddix, mapped, st, ql2, 0)) { _Pragma("omp atomic") ++RefCounts[mapped]; }
						});
					}
				}
			}
			free(RefCache); free(StCache); free(list);
		}
		if (oom) { free(start); free(count); free(RefCounts); return bh_set_error(BH_E_OOM, "OOM:report"); }
	}
	const double t_pre = omp_get_wtime();
	const uint64_t CH = chunkQ, nChunks = (nU + CH - 1) / CH, chunkGroup = (uint64_t)nThreads * 8;
	LineBuf *cbuf = calloc(nChunks + 1, sizeof(*cbuf)); uint64_t *coff = calloc(nChunks + 2, sizeof(*coff));
	if (!cbuf || !coff) { free(cbuf); free(coff); free(start); free(count); free(RefCounts); return bh_set_error(BH_E_OOM, "OOM:report"); }
	/* the rendered chunks of a group are written side by side at their offsets (pwrite) when the output is a seekable file: one
	 * thread copying gigabytes into the page cache was the other half of the report's time */
	fflush(real_out);
	const int out_fd = fileno(real_out);
	off_t file_pos = out_fd >= 0 ? lseek(out_fd, 0, SEEK_CUR) : (off_t)-1;
	int use_pwrite = file_pos != (off_t)-1;      /* 0 = fwrite by one thread, 1 = pwrite side by side, 2 = copies into a shared mapping side by side */
	/* measured on the GPU box (32 M lines, 2.4 GB, 32 threads): one thread's fwrite 0.3-0.5 s, pwrite side by side 0.6 s (the
	 * writers queue on the file's lock), a shared mapping 1.6 s (page faults): one writer it is, beside the next group's rendering */
	use_pwrite = 0;
	if (getenv("BURST_HOST_REPORT_WRITE")) { const int w_ = atoi(getenv("BURST_HOST_REPORT_WRITE")); if (w_ >= 0 && w_ <= 2 && (use_pwrite || !w_)) use_pwrite = w_; }      /* tuning / test hook */
	char *map_base = NULL; size_t map_len = 0, map_skew = 0;
	int map_fd = -1;      /* a shared writable mapping needs a descriptor opened for reading and writing: the caller's stream is write-only */
	if (use_pwrite == 2) {
		char pth[64]; snprintf(pth, sizeof pth, "/proc/self/fd/%d", out_fd);
		map_fd = open(pth, O_RDWR);
		if (map_fd < 0) use_pwrite = 1;
	}
	#pragma omp parallel num_threads(nThreads) reduction(+:lines)
	{
	uint32_t *RefCache = malloc(capX * 4 + 4), *StCache = malloc(capX * 4 + 4), *RIXcache = malloc(capX * 4 + 4);
	const BhipHit **RPcache = malloc((capX + 1) * sizeof(*RPcache));
	const BhipHit **list = malloc(((size_t)maxList + 1) * sizeof(*list));
	char *Taxon = wt ? calloc(1, 1000000) : NULL;          /* scratch of the interpolated / suppressed string (burst.c:4741, 4852) */
	const char *FinalTaxon = NULL;                           /* BEST keeps the last value when a taxonomy is empty (burst.c:4852-4885) */
	const char **Taxa = NULL; uint32_t *Divergence = NULL;
	if (wt && mode == BH_CAPITALIST) { Taxa = malloc((capX + 1) * sizeof(*Taxa)); Divergence = calloc(capX + 1, sizeof(*Divergence)); }
	if (!RefCache || !StCache || !RIXcache || !RPcache || !list || (wt && !Taxon) || (wt && mode == BH_CAPITALIST && (!Taxa || !Divergence))) {
		#pragma omp atomic write
		oom = 1;
	}
	#pragma omp barrier
	/* groups of chunks: rendered by the team, then written and released by one thread, so that the rendered text held in
	 * memory stays bounded however many queries there are */
	for (uint64_t cg0 = 0; cg0 < nChunks; cg0 += chunkGroup) {
	const uint64_t cg1 = cg0 + chunkGroup < nChunks ? cg0 + chunkGroup : nChunks;
	const double tg0 = omp_get_wtime();
	#pragma omp for schedule(dynamic, 1)
	for (uint64_t ch = cg0; ch < cg1; ++ch) {
	if (oom) continue;
	LineBuf *out = &cbuf[ch];                                /* the loop body below prints to `out` */
	const uint64_t i0 = ch * CH, i1 = i0 + CH < nU ? i0 + CH : nU;
	if (mode == BH_BEST && !wt) {
		/* BEST without taxonomy, the bulk case (one line per read): a line touches half a dozen unrelated places -- the read's header
		 * in the query file, the record, the reference's number, its header pointer, its header, its offset -- and formatting it is
		 * a few hundred instructions, too many for the core to overlap the misses of consecutive lines on its own.  Blocks of 64
		 * queries go through short loops instead: pick the record (the misses of a block overlap), ask for what the line needs, print. */
		enum { BLK = 64 };
		const BhipHit *bb[BLK]; uint32_t brix[BLK]; uint64_t bi[BLK];
		for (uint64_t ib = i0; ib < i1; ib += BLK) {
			const uint64_t ie = ib + BLK < i1 ? ib + BLK : i1;
			uint32_t nb = 0;
			for (uint64_t i = ib; i < ie; ++i) {
				uint32_t n; BUILD_LIST(i, n);
				if (!n) continue;
				const BhipHit *best = list[0];
				for (uint32_t k = 1; k < n; ++k) {
					const BhipHit *rp = list[k];
					if (rp->ed < best->ed || (rp->ed == best->ed && rp->score > best->score) ||
					    (rp->ed == best->ed && rp->score == best->score && db->refIxSrt[rp->refIx] < db->refIxSrt[best->refIx])) best = rp;
				}
				bb[nb] = best; bi[nb] = i; ++nb;
				__builtin_prefetch(&db->refIxSrt[best->refIx]);
				__builtin_prefetch(Q->heads[Q->offset[i]]);
			}
			for (uint32_t k = 0; k < nb; ++k) {
				brix[k] = db->refIxSrt[bb[k]->refIx];
				__builtin_prefetch(&db->refHead[brix[k]]);
				if (db->refStart) __builtin_prefetch(&db->refStart[brix[k]]);
			}
			for (uint32_t k = 0; k < nb; ++k) __builtin_prefetch(db->refHead[brix[k]]);
			for (uint32_t k = 0; k < nb; ++k) {
				const uint64_t i = bi[k]; const uint32_t qlen = Q->len[i];
				uint32_t st, ed; coords(db, bb[k], brix[k], qlen, &st, &ed);
				for (uint64_t j = Q->offset[i]; j < Q->offset[i + 1]; ++j) { print_line_tax(out, Q->heads[j], db->refHead[brix[k]], bb[k], qlen, st, ed, i, 0, NULL); ++lines; }
			}
		}
	} else
	for (uint64_t i = i0; i < i1; ++i) {
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
			if (wt) {                                                              /* burst.c:4872-4887 */
				const char *tt = bh_tax_lookup(T, db->refHead[rix], ncbi);
				if (suppress) {
					uint32_t lm = 0, sc = 0;
					strcpy(Taxon, tt);
					while (LEVELS[lm] < best->score) ++lm;
					if (!lm) FinalTaxon = "";
					else for (int x = 0; Taxon[x]; ++x) {
						if (Taxon[x] == ';' && ++sc == lm) { Taxon[x] = 0; break; }
						FinalTaxon = Taxon;          /* only reached when the string has a character before the cut: an empty taxonomy keeps the previous query's */
					}
				} else FinalTaxon = tt;
			}
			for (uint64_t j = Q->offset[i]; j < Q->offset[i + 1]; ++j) { print_line_tax(out, Q->heads[j], db->refHead[rix], best, qlen, st, ed, i, wt, FinalTaxon); ++lines; }
		} else if (mode == BH_ANY) {                                         /* any valid hit; column 12 = duplicate flag (burst.c:4268-4272) */
			/* The reference prints the first hit within budget a thread meets and marks the query spent (burst.c:4239-4275, 4457-4475).
			 * Exhaustive path, one thread: clumps ascending, the entries of a clump in sorted order, lanes ascending -- the LAST
			 * element of the merged list (a LIFO).  Accelerated path: the clumps come by descending k-mer count of a bunch of queries,
			 * which depends on the thread count: not reproducible, the hit with the fewest edits (first in list order) stands in. */
			const BhipHit *rp = list[merged ? n - 1 : 0];
			if (!merged) for (uint32_t k = 1; k < n; ++k) if (list[k]->ed < rp->ed) rp = list[k];
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
					if (!dupe_hunt(nodupe, RefCache, StCache, &ddix, MAPPED(rix), st, ql2, 1)) { RPcache[rix_ix] = rp; RIXcache[rix_ix++] = rix; }
				});
			}
			for (uint64_t j = Q->offset[i]; j < Q->offset[i + 1]; ++j) for (uint64_t zz = 0; zz < rix_ix; ++zz) {
				const BhipHit *rp = RPcache[zz]; const uint32_t rix = RIXcache[zz];
				uint32_t st, ed; coords(db, rp, rix, qlen, &st, &ed);
				print_line_tax(out, Q->heads[j], db->refHead[rix], rp, qlen, st, ed, i, wt, wt ? bh_tax_lookup(T, db->refHead[rix], ncbi) : NULL); ++lines;   /* burst.c:4631-4633, 4685-4687 */
			}
		} else {                                                             /* CAPITALIST pass B (burst.c:4746-4779, 4831-4843) */
			uint32_t b = 0;
			for (uint32_t k = 1; k < n; ++k) if (list[k]->ed < list[b]->ed) b = k;
			const BhipHit *best = list[b]; uint32_t bestmap = 0, bestrix = 0; int have = 0;
			uint64_t ddix = 0;
			uint32_t tix = 0; float best_score = -1.f;
			const BhipHit *first = list[b];
			for (uint32_t k = b; k < n; ++k) {
				const BhipHit *rp = list[k];
				if (rp->ed > best->ed) continue;
				FOR_EXPANSIONS(rp, rix, {
					uint32_t st, ed; coords(db, rp, rix, qlen, &st, &ed); (void)ed;
					uint32_t mapped = MAPPED(rix);
					if (!dupe_hunt(nodupe, RefCache, StCache, &ddix, mapped, st, ql2, 0)) {
						if (wt) { Taxa[tix++] = bh_tax_lookup(T, db->refHead[rix], ncbi); best_score = rp->score > best_score ? rp->score : best_score; }   /* burst.c:4760-4762 */
						if (best == rp || RefCounts[mapped] > RefCounts[bestmap] || (RefCounts[mapped] == RefCounts[bestmap] && mapped < bestmap)) {
							best = rp; bestmap = mapped; bestrix = rix; have = 1;
						}
					}
				});
			}
			(void)first;
			if (!have) continue;   /* cannot happen: the first expansion of the first pod is always accepted */
			const char *Final = NULL;
			if (wt) {              /* taxonomy interpolation over the accepted placements (burst.c:4781-4829) */
				uint32_t lv = UINT32_MAX;
				if (tix == 1) { strcpy(Taxon, Taxa[0]); Final = Taxon; }
				else {
					qsort(Taxa, tix, sizeof(*Taxa), cmp_str);
					uint32_t maxDiv = 0;
					for (uint32_t zq = 1; zq < tix; ++zq) {   /* shared leading levels of neighbours (+1 when the previous one is a prefix) */
						uint32_t x = 0, dv = 0;
						for (; Taxa[zq - 1][x] && Taxa[zq - 1][x] == Taxa[zq][x]; ++x) dv += Taxa[zq][x] == ';';
						dv += !Taxa[zq - 1][x];
						Divergence[zq] = dv;
						if (dv > maxDiv) maxDiv = dv;
					}
					if (!maxDiv) { Taxon[0] = 0; Final = Taxon; }
					else {
						uint32_t cutoff = tix - tix / taxacut, s0 = 0, e0 = tix;   /* deepest level on which all but 1/taxacut agree */
						for (lv = 1; lv <= maxDiv; ++lv) {
							uint32_t accum = 1;
							for (uint32_t zq = s0 + 1; zq < e0; ++zq) {
								if (Divergence[zq] >= lv) ++accum;
								else if (accum >= cutoff) { e0 = zq; break; }
								else { accum = 1; s0 = zq; }
							}
							if (accum < cutoff) break;
							cutoff = accum - accum / taxacut;
						}
						uint32_t sc = 0, x = 0;
						if (e0) --e0;
						--lv;
						for (; Taxa[e0][x] && (sc += Taxa[e0][x] == ';') < lv; ++x) Taxon[x] = Taxa[e0][x];
						Taxon[x] = 0;
						Final = Taxon;
					}
				}
				if (suppress) {                                                  /* burst.c:4820-4828 */
					uint32_t lm = 0, sc = 0;
					while (lm < lv && LEVELS[lm] < best_score) ++lm;
					if (!lm) Final = "";
					else if (lm < lv) for (int x = 0; Taxon[x]; ++x) if (Taxon[x] == ';' && ++sc == lm) { Taxon[x] = 0; break; }
				}
			}
			uint32_t st, ed; coords(db, best, bestrix, qlen, &st, &ed);
			for (uint64_t j = Q->offset[i]; j < Q->offset[i + 1]; ++j) { print_line_tax(out, Q->heads[j], db->refHead[bestrix], best, qlen, st, ed, i, wt, Final); ++lines; }
		}
	}
	if (out->oom) {
		#pragma omp atomic write
		oom = 1;
	}
	}
	if (!use_pwrite) {
		/* one thread writes the group (chunks in query order) while the others go on to render the next one: the writer joins them
		 * when it is done (the loop's schedule is dynamic), and the barrier behind every group keeps at most two groups in memory */
		#pragma omp single nowait
		{
			t_render += omp_get_wtime() - tg0;
			for (uint64_t ch = cg0; ch < cg1; ++ch) {
				if (!oom && cbuf[ch].len && fwrite(cbuf[ch].p, 1, cbuf[ch].len, real_out) != cbuf[ch].len) wr = 1;
				free(cbuf[ch].p); cbuf[ch].p = NULL;
			}
			t_write += omp_get_wtime() - tg0;
		}
		continue;
	}
	#pragma omp single
	{
		t_render += omp_get_wtime() - tg0;
		coff[cg0] = 0;
		for (uint64_t ch = cg0; ch < cg1; ++ch) coff[ch + 1] = coff[ch] + cbuf[ch].len;      /* chunks in query order */
	}
	if (use_pwrite == 2) {      /* the group's range of the file mapped: the threads copy their chunks into the page cache side by side */
		#pragma omp single
		{
			map_base = NULL;
			const off_t lo = file_pos & ~(off_t)4095;
			map_len = (size_t)(file_pos - lo) + (size_t)coff[cg1];
			if (coff[cg1] && !oom) {
				if (posix_fallocate(out_fd, file_pos, (off_t)coff[cg1])) wr = 1;
				else { map_base = mmap(NULL, map_len, PROT_READ | PROT_WRITE, MAP_SHARED, map_fd, lo); if (map_base == MAP_FAILED) { map_base = NULL; wr = 1; } }
			}
			map_skew = (size_t)(file_pos - lo);
		}
		#pragma omp for schedule(dynamic, 1)
		for (uint64_t ch = cg0; ch < cg1; ++ch) {
			if (map_base && cbuf[ch].len) memcpy(map_base + map_skew + coff[ch], cbuf[ch].p, cbuf[ch].len);
			free(cbuf[ch].p); cbuf[ch].p = NULL;
		}
		#pragma omp single
		{ if (map_base) munmap(map_base, map_len); file_pos += (off_t)coff[cg1]; }
	} else if (use_pwrite) {
		#pragma omp for schedule(dynamic, 1)
		for (uint64_t ch = cg0; ch < cg1; ++ch) {
			size_t done = 0;
			while (!oom && done < cbuf[ch].len) {
				const ssize_t w = pwrite(out_fd, cbuf[ch].p + done, cbuf[ch].len - done, file_pos + (off_t)(coff[ch] + done));
				if (w <= 0) {
					#pragma omp atomic write
					wr = 1;
					break;
				}
				done += (size_t)w;
			}
			free(cbuf[ch].p); cbuf[ch].p = NULL;
		}
		#pragma omp single
		{ file_pos += (off_t)coff[cg1]; }
	}
	#pragma omp master
	t_write += omp_get_wtime() - tg0;      /* (pwrite / mapping variants: rendering + writing of the group) */
	}
	free(RefCache); free(StCache); free(RIXcache); free(RPcache); free(list); free(Taxon); free(Taxa); free(Divergence);
	}
	if (dbg) fprintf(stderr, "[bh_report] %lu lines: tables %.3f s, rendering %.3f s, writing %.3f s (%d threads, %s)\n", (unsigned long)lines, t_pre - t_begin, t_render, t_write - t_render, nThreads, use_pwrite == 2 ? "shared mapping" : use_pwrite ? "pwrite" : "fwrite");
	if (map_fd >= 0) close(map_fd);
	if (use_pwrite && lseek(out_fd, file_pos, SEEK_SET) == (off_t)-1) wr = 1;      /* the stream continues behind what was written */
	free(cbuf); free(coff);
	free(start); free(count); free(RefCounts);
	if (oom) return bh_set_error(BH_E_OOM, "OOM:report");
	if (wr) return bh_set_error(BH_E_IO, "short write on the output file");
	if (nLines) *nLines = lines;
	return BH_OK;
}
