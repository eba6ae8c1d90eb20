/* bh_queries.c -- query pipeline (mirror of process_queries, burst.c:2980-3223, and its parser 636-690).
 *
 * parse 2-line FASTA -> symbol codes -> sort -> dedupe (Offset) -> per-query error budget -> reverse-complement
 * entries -> accelerator bins.  The prefix-"div"/cache bookkeeping of the reference (3187-3206) exists only to
 * let its CPU kernel reuse DP rows; the device path aligns every (query, clump) independently, so it is omitted.
 */
#include "burst_host.h"
#include <stdlib.h>
#include <string.h>
#include <omp.h>
#include <zlib.h>
#include <unistd.h>

static int bh_ingest_threads(void);
typedef struct { const uint8_t *s; uint32_t len; uint64_t ix; } QRef;

/* Whole query file into memory.  gzip files (magic 1f 8b) are inflated; FASTQ (first byte '@': four-line records) is rewritten
 * in place into the two-line FASTA layout the reference's parser takes (burst.c:636-690; the reference itself reads neither). */
static int slurp_queries(const char *path, char **out, uint64_t *out_sz) {
	FILE *f = fopen(path, "rb");
	if (!f) return bh_set_error(BH_E_IO, "Cannot open FASTA file: %s.", path);
	unsigned char magic[2] = {0, 0};
	const size_t got = fread(magic, 1, 2, f);
	char *dump = NULL; uint64_t sz = 0;
	if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
		fclose(f);
		gzFile g = gzopen(path, "rb");
		if (!g) return bh_set_error(BH_E_IO, "Cannot open FASTA file: %s.", path);
		gzbuffer(g, 1 << 20);
		uint64_t cap = 1ull << 26;
		dump = malloc(cap + 17);
		if (!dump) { gzclose(g); return bh_set_error(BH_E_OOM, "OOM reading queries"); }
		for (;;) {
			if (cap - sz < (1u << 24)) { cap *= 2; char *nd = realloc(dump, cap + 17); if (!nd) { free(dump); gzclose(g); return bh_set_error(BH_E_OOM, "OOM reading queries"); } dump = nd; }
			const int n = gzread(g, dump + sz, (unsigned)((cap - sz) > (1u << 30) ? (1u << 30) : (cap - sz)));
			if (n < 0) { free(dump); gzclose(g); return bh_set_error(BH_E_IO, "cannot inflate %s", path); }
			if (!n) break;
			sz += (uint64_t)n;
		}
		gzclose(g);
	} else {
		fseeko(f, 0, SEEK_END);
		sz = (uint64_t)ftello(f);
		rewind(f);
		dump = malloc(sz + 17);
		if (!dump) { fclose(f); return bh_set_error(BH_E_OOM, "OOM reading queries"); }
		/* pieces of 32 MB side by side (a single fread of a cached multi-gigabyte file is one thread's memcpy) */
		const int fd = fileno(f);
		const uint64_t piece = 32u << 20, np = (sz + piece - 1) / piece;
		int bad = 0;
		#pragma omp parallel for num_threads(bh_ingest_threads()) schedule(dynamic, 1) if (np > 1)
		for (uint64_t k = 0; k < np; ++k) {
			uint64_t o = k * piece; const uint64_t e = o + piece < sz ? o + piece : sz;
			while (o < e) {
				const ssize_t g = pread(fd, dump + o, (size_t)(e - o), (off_t)o);
				if (g <= 0) {
					#pragma omp atomic write
					bad = 1;
					break;
				}
				o += (uint64_t)g;
			}
		}
		fclose(f);
		if (bad) { free(dump); return bh_set_error(BH_E_IO, "short read on %s", path); }
	}
	if (sz && dump[0] == '@') {      /* FASTQ: keep lines 1 and 2 of every four, '@' -> '>' */
		/* four-line records only (sequence and qualities on one line each); empty lines between records and at the end of the
		 * file are passed over, the separator line must start with '+' -- a wrapped (multi-line) FASTQ file fails that test
		 * and is refused with a message that says so instead of being read as garbage */
		uint64_t r = 0, w = 0, line = 0, rec = 0;
		while (r < sz) {
			char *nl = memchr(dump + r, '\n', sz - r);
			const uint64_t e = nl ? (uint64_t)(nl - dump) + 1 : sz;
			const uint64_t body = e - r - (nl ? 1 : 0) - ((e - r >= 2 && dump[e - 2] == '\r' && nl) ? 1 : 0);
			if ((line & 3) == 0 && body == 0) { r = e; continue; }      /* blank line at a record boundary */
			if ((line & 3) == 0) {
				++rec;
				if (dump[r] != '@') { free(dump); return bh_set_error(BH_E_USAGE, "ERROR: Malformatted FASTQ file (record %lu does not start with '@'; sequences wrapped over several lines are not supported).", (unsigned long)rec); }
				dump[r] = '>';
			} else if ((line & 3) == 2 && dump[r] != '+') {
				free(dump); return bh_set_error(BH_E_USAGE, "ERROR: Malformatted FASTQ file (record %lu: the third line does not start with '+'; sequences wrapped over several lines are not supported).", (unsigned long)rec);
			}
			if ((line & 3) < 2) { memmove(dump + w, dump + r, e - r); w += e - r; }
			r = e; ++line;
		}
		if (line & 3) { free(dump); return bh_set_error(BH_E_USAGE, "ERROR: Malformatted FASTQ file (the last record is incomplete)."); }
		sz = w;
	}
	memset(dump + sz, 0, 17);
	*out = dump; *out_sz = sz;
	return BH_OK;
}

/* the ingest loops are short and memory-bound: beyond a few dozen threads the fork/join of a 256-thread host costs more than
 * the loop (measured on 2 x EPYC 9575F: 0.84 s with all 256 threads) */
static int bh_ingest_threads(void) { const int n = omp_get_max_threads(); return n > 32 ? 32 : n; }

static int g_sort_device = 0;          /* device of the query sort (bhip_sort_queries); < 0 = host sort */
void bh_queries_sort_device(int device) { g_sort_device = device; }

static int qref_cmp(const void *a, const void *b) {
	const QRef *A = a, *B = b;
	uint32_t n = A->len < B->len ? A->len : B->len;
	int c = memcmp(A->s, B->s, n);           /* codes are 0..15, so memcmp == the reference's strcmp order (burst.c:363-366) */
	if (c) return c;
	if (A->len != B->len) return A->len < B->len ? -1 : 1;
	return A->ix < B->ix ? -1 : (A->ix > B->ix);
}

static inline uint32_t prefix_bucket(const QRef *r) {   /* first 5 symbols, 4 bits each (NIB5, burst.c:383) */
	uint32_t v = 0;
	for (uint32_t i = 0; i < 5; ++i) v = (v << 4) | (i < r->len ? r->s[i] : 0);
	return v;
}

static int sort_qrefs(QRef *a, uint64_t n) {
	if (n < 1u << 16) { qsort(a, n, sizeof(*a), qref_cmp); return 0; }
	const uint32_t NB = 1u << 20;
	uint64_t *cnt = calloc((size_t)NB + 1, sizeof(*cnt));
	QRef *tmp = malloc(n * sizeof(*tmp));
	if (!cnt || !tmp) { free(cnt); free(tmp); return bh_set_error(BH_E_OOM, "OOM sorting queries"); }
	for (uint64_t i = 0; i < n; ++i) ++cnt[prefix_bucket(a + i) + 1];
	for (uint32_t b = 0; b < NB; ++b) cnt[b + 1] += cnt[b];
	uint64_t *pos = malloc((size_t)NB * sizeof(*pos));
	if (!pos) { free(cnt); free(tmp); return bh_set_error(BH_E_OOM, "OOM sorting queries"); }
	memcpy(pos, cnt, (size_t)NB * sizeof(*pos));
	for (uint64_t i = 0; i < n; ++i) tmp[pos[prefix_bucket(a + i)]++] = a[i];
	#pragma omp parallel for num_threads(bh_ingest_threads()) schedule(dynamic, 64)
	for (uint32_t b = 0; b < NB; ++b) if (cnt[b + 1] - cnt[b] > 1)
		qsort(tmp + cnt[b], cnt[b + 1] - cnt[b], sizeof(*tmp), qref_cmp);
	memcpy(a, tmp, n * sizeof(*a));
	free(cnt); free(tmp); free(pos);
	return 0;
}

void bh_queries_free(BhQueries *q) {
	if (!q) return;
	if (q->pinned) {
		bhip_host_unregister(q->codes4 ? q->codes4 : q->codes); bhip_host_unregister(q->qoff);
		if (q->pinned & 2) bhip_host_unregister(q->emac);
		if (q->pinned & 4) bhip_host_unregister(q->rc);
		if (q->pinned & 8) bhip_host_unregister(q->flags);
	}
	if (q->pinned & 16) bhip_host_unregister(q->codes2);
	if (q->pinned & 32) bhip_host_unregister(q->len16);
	free(q->codes2); free(q->len16); free(q->ambBefore);
	free(q->dump); free(q->heads); free(q->offset); free(q->codes); free(q->codes4); free(q->qoff); free(q->six); free(q->rc);
	free(q->flags); free(q->emac); free(q->len); free(q->ed);
	memset(q, 0, sizeof *q);
}

/* accelerator bins (burst.c:3124-3141): bad = too short / too many errors for the k-mer guarantee / > 5 strongly
 * ambiguous symbols; ambiguous = any symbol beyond A/C/G/T.  Clear and ambiguous entries use the device prefilter
 * (words with ambiguous symbols simply do not vote there and the guaranteed count shrinks accordingly; the library
 * falls back to the exhaustive route by itself when no word is guaranteed); bad ones are exhaustive (burst.c:4320).
 * Separate from bh_queries_load so that a caller that reads the queries BESIDE the database (the accelerator file says
 * which K it was built with) can bin them afterwards. */
void bh_queries_bins(BhQueries *Q, int do_accel, int K, int z) {
	const uint64_t numEntries = Q->numEntries;
	uint64_t nClear = 0, nAmbig = 0, nBad = 0;
	#pragma omp parallel for num_threads(bh_ingest_threads()) reduction(+:nClear, nAmbig, nBad) schedule(static)
	for (uint64_t e = 0; e < numEntries; ++e) {
		if (!do_accel) { Q->flags[e] = BHIP_Q_EXHAUSTIVE; continue; }
		const uint32_t len = Q->len[Q->six[e]], ed = Q->ed[Q->six[e]];
		const uint8_t *s = Q->codes + Q->qoff[e];
		int stat = 1;
		if (len < (uint32_t)K || ed >= len / (uint32_t)K) stat = 2;
		else {
			uint32_t totN = 0;
			for (uint32_t j = 0; j < len; ++j) {
				if ((totN += s[j] > 4 + z) > 5) { stat = 2; break; }
				else if (s[j] > 4) stat = 0;
			}
		}
		/* the reference demotes every "bad" entry to the exhaustive path because its word expansion explodes (burst.c:3134);
		 * on the device only entries shorter than K must go there -- for the rest libburst_hip works out per entry whether
		 * any k-mer is still guaranteed (and aligns exhaustively by itself if not), with identical results */
		Q->flags[e] = len < (uint32_t)K ? BHIP_Q_EXHAUSTIVE : BHIP_Q_PREFILTER;
		if (stat == 1) ++nClear; else if (stat == 0) ++nAmbig; else ++nBad;
	}
	Q->nClear = nClear; Q->nAmbig = nAmbig; Q->nBad = nBad;
}

int bh_queries_load(const char *fasta, float thres, int do_rc, int incl_whitespace, int do_accel, int K, int z,
                    int skip_ambig, BhQueries *Q) {
	memset(Q, 0, sizeof *Q);
	(void)skip_ambig;
	const int dbg = getenv("BURST_HOST_DEBUG") != NULL;
	double t_ = omp_get_wtime();
	#define QPH(name) do { if (dbg) { const double n_ = omp_get_wtime(); fprintf(stderr, "[bh_queries] %-22s %.3f s\n", name, n_ - t_); t_ = n_; } } while (0)
	char *dump = NULL; uint64_t sz = 0;
	{ const int rcs = slurp_queries(fasta, &dump, &sz); if (rcs) return rcs; }
	Q->dump = dump;
	QPH("file read");
	if (!sz || *dump != '>') { bh_queries_free(Q); return bh_set_error(BH_E_USAGE, "ERROR: Malformatted FASTA file."); }
	/* strict two-line records: (number of newlines, rounded up to even) / 2 must equal the number of '>' (burst.c:648-654) */
	uint64_t numNL = 0, numLT = 0;
	#pragma omp parallel for num_threads(bh_ingest_threads()) reduction(+:numNL, numLT)
	for (uint64_t i = 0; i < sz; ++i) { numNL += dump[i] == '\n'; numLT += dump[i] == '>'; }
	numNL += numNL & 1;
	if (numLT != numNL / 2) { bh_queries_free(Q); return bh_set_error(BH_E_USAGE, "ERROR: line count != '>' * 2"); }
	const uint64_t totQ = numLT;
	QPH("line count");
	char **heads = malloc(totQ * sizeof(*heads));
	QRef *refs = malloc(totQ * sizeof(*refs));
	if (!heads || !refs) { free(heads); free(refs); bh_queries_free(Q); return bh_set_error(BH_E_OOM, "OOM indexing queries"); }
	uint8_t c2n[256];
	bh_char2code(c2n);
	{	/* Lines alternate header / sequence (burst.c:636-690).  The file is cut into byte ranges; a range owns the lines that START
		 * in it.  The number of line ends in front of a range gives the number of its first line, so all ranges index their
		 * records side by side; the line ends are only overwritten (NUL) after every range has found its first line. */
		const int nt = bh_ingest_threads();
		uint64_t nr = sz < (1u << 22) ? 1 : (uint64_t)nt * 8;
		if (getenv("BURST_HOST_INGEST_RANGES") && atoll(getenv("BURST_HOST_INGEST_RANGES")) > 0) nr = (uint64_t)atoll(getenv("BURST_HOST_INGEST_RANGES"));      /* test hook */
		if (nr > sz) nr = sz ? sz : 1;
		uint64_t *nlBefore = calloc(nr + 2, sizeof(*nlBefore)), *first = malloc((nr + 2) * sizeof(*first)), *firstLine = malloc((nr + 2) * sizeof(*firstLine));
		if (!nlBefore || !first || !firstLine) { free(nlBefore); free(first); free(firstLine); free(heads); free(refs); bh_queries_free(Q); return bh_set_error(BH_E_OOM, "OOM indexing queries"); }
		#pragma omp parallel for num_threads(nt) schedule(static, 1)
		for (uint64_t t = 0; t < nr; ++t) {
			const uint64_t a = sz * t / nr, b = sz * (t + 1) / nr;
			uint64_t c = 0;
			for (const char *p = dump + a, *e = dump + b; p < e && (p = memchr(p, '\n', (size_t)(e - p))); ++p) ++c;
			nlBefore[t + 1] = c;
			if (!a) first[t] = 0;
			else if (dump[a - 1] == '\n') first[t] = a;
			else { const char *q = memchr(dump + a, '\n', (size_t)(sz - a)); first[t] = q ? (uint64_t)(q - dump) + 1 : sz; }
		}
		for (uint64_t t = 0; t < nr; ++t) nlBefore[t + 1] += nlBefore[t];
		for (uint64_t t = 0; t < nr; ++t) {      /* number of the first line that starts in range t = line ends in front of it */
			const uint64_t a = sz * t / nr;
			firstLine[t] = nlBefore[t] + ((a && dump[a - 1] != '\n') ? 1 : 0);
		}
		/* a record whose lines are missing (a header at the very end of the file has no sequence line) is an empty query */
		#pragma omp parallel for num_threads(nt) schedule(static)
		for (uint64_t n = 0; n < totQ; ++n) { heads[n] = dump + sz; refs[n].s = (uint8_t *)dump + sz; refs[n].len = 0; refs[n].ix = n; }
		#pragma omp parallel for num_threads(nt) schedule(static, 1)
		for (uint64_t t = 0; t < nr; ++t) {
			const uint64_t b = sz * (t + 1) / nr;
			uint64_t i = first[t], line = firstLine[t];
			for (; i < b && i < sz && line < 2 * totQ; ++line) {
				char *l = dump + i;
				char *nl = memchr(l, '\n', (size_t)(sz - i));
				if (!nl) nl = dump + sz;
				char *le = nl;
				const uint64_t n = line >> 1;
				/* the reference strips a carriage return from every record but the FIRST (burst.c:664-668 vs 677-684): there it
				 * stays in the header and becomes a last query symbol of code 0 */
				if (n && le > l && le[-1] == '\r') --le;
				if (!(line & 1)) {      /* header line */
					heads[n] = l + 1 <= nl ? l + 1 : nl;
				} else { refs[n].s = (uint8_t *)l; refs[n].len = (uint32_t)(le - l); refs[n].ix = n; }
				i = (uint64_t)(nl - dump) + 1;
			}
		}
		/* the line ends (and carriage returns, and the first blank of a header) become terminators: a second sweep, so that no range
		 * searches for a line end its neighbour has already overwritten */
		#pragma omp parallel for num_threads(nt) schedule(static)
		for (uint64_t n = 0; n < totQ; ++n) {
			char *h = heads[n];
			char *he = h + strcspn(h, "\n");
			if (n && he > h && he[-1] == '\r') --he;
			*he = 0;
			if (!incl_whitespace) for (char *p = h; *p; ++p) if (*p == ' ' || *p == '\t') { *p = 0; break; }   /* burst.c:2987-2992 */
		}
		free(nlBefore); free(first); free(firstLine);
	}
	QPH("record index");
	uint32_t maxLen = 0, minLen = UINT32_MAX;
	#pragma omp parallel for num_threads(bh_ingest_threads()) reduction(max:maxLen) reduction(min:minLen)
	for (uint64_t i = 0; i < totQ; ++i) {
		uint8_t *s = (uint8_t *)refs[i].s;
		for (uint32_t k = 0; k < refs[i].len; ++k) s[k] = c2n[s[k]];                     /* translateNV, burst.c:1207 */
		if (refs[i].len > maxLen) maxLen = refs[i].len;
		if (refs[i].len < minLen) minLen = refs[i].len;
	}
	if (maxLen > BHIP_MAX_QLEN) {
		free(heads); free(refs); bh_queries_free(Q);
		return bh_set_error(BH_E_USAGE, "ERROR: query of %u symbols exceeds the device limit of %d", maxLen, BHIP_MAX_QLEN);
	}
	QPH("translate");
	uint64_t numUniq = 0;
	uint8_t *isNew = malloc(totQ + 1);            /* 1 where a sorted record differs from its predecessor */
	if (!isNew) { free(heads); free(refs); bh_queries_free(Q); return bh_set_error(BH_E_OOM, "OOM:dedupe"); }
	int rc = 0, on_device = 0;
	/* Large inputs: sort and duplicate marking on the device (bhip_sort_queries: LSD radix sort over 16-symbol keys; the same
	 * order as qref_cmp).  It runs before the database is uploaded, on the device the search will use.  Small inputs, and
	 * hosts told so (BURST_HOST_SORT=1 / bh_queries_sort_device(-1)), take the host path below. */
	/* (a database going up to that device right now sizes its accelerator build from the free memory it found: the sort does not
	 * squeeze in beside it, and it does not wait tens of seconds for it either -- the host sorts) */
	if (totQ >= (1u << 18) && totQ < 0x7FFFFFFFull && g_sort_device >= 0 && !getenv("BURST_HOST_SORT") && bh_device_gate_try(g_sort_device)) {
		uint64_t *start = malloc(totQ * sizeof(*start));
		uint32_t *lens = malloc(totQ * sizeof(*lens)), *perm = malloc(totQ * sizeof(*perm));
		QRef *tmp = malloc(totQ * sizeof(*tmp));
		if (start && lens && perm && tmp) {
			#pragma omp parallel for num_threads(bh_ingest_threads())
			for (uint64_t i = 0; i < totQ; ++i) { start[i] = (uint64_t)((char *)refs[i].s - dump); lens[i] = refs[i].len; }
			const int sort_rc = bhip_sort_queries(g_sort_device, (const uint8_t *)dump, sz, start, lens, totQ, maxLen, perm, isNew);
			if (!sort_rc) {
				#pragma omp parallel for num_threads(bh_ingest_threads()) reduction(+:numUniq)
				for (uint64_t i = 0; i < totQ; ++i) { tmp[i] = refs[perm[i]]; numUniq += isNew[i]; }
				{ QRef *t = refs; refs = tmp; tmp = t; }          /* the permuted table is the table from here on */
				on_device = 1;
			} else fprintf(stderr, " --> NOTE: query sort on device %d not available (%s): sorting on the host\n", g_sort_device, bhip_last_error());
		}
		bh_device_gate(g_sort_device, 0);
		free(start); free(lens); free(perm); free(tmp);
		QPH("sort + duplicates (device)");
	}
	if (!on_device) {
	rc = sort_qrefs(refs, totQ);
	QPH("sort");
	if (rc) { free(isNew); free(heads); free(refs); bh_queries_free(Q); return rc; }
	/* uniqueness (burst.c:3036-3053) */
	numUniq = 0;
	#pragma omp parallel for num_threads(bh_ingest_threads()) reduction(+:numUniq) schedule(static)
	for (uint64_t i = 0; i < totQ; ++i) {
		isNew[i] = !i || refs[i].len != refs[i - 1].len || memcmp(refs[i].s, refs[i - 1].s, refs[i].len);
		numUniq += isNew[i];
	}
	}
	const uint64_t numEntries = numUniq * (do_rc ? 2 : 1);
	Q->heads = malloc(totQ * sizeof(*Q->heads));
	Q->offset = malloc((numUniq + 1) * sizeof(*Q->offset));
	Q->qoff = malloc((numEntries + 1) * sizeof(*Q->qoff));
	Q->six = malloc(numEntries * sizeof(*Q->six));
	Q->rc = calloc(numEntries, 1);
	Q->flags = calloc(numEntries, 1);
	Q->emac = malloc(numEntries * sizeof(*Q->emac));
	Q->len = malloc(numUniq * sizeof(*Q->len));
	Q->ed = malloc(numUniq * sizeof(*Q->ed));
	if (!Q->heads || !Q->offset || !Q->qoff || !Q->six || !Q->rc || !Q->flags || !Q->emac || !Q->len || !Q->ed) {
		free(isNew); free(heads); free(refs); bh_queries_free(Q); return bh_set_error(BH_E_OOM, "OOM building query tables");
	}
	uint64_t totLen = 0;
	{	/* in slices: the rank of a unique query = the number of marks in front of its slice + inside it */
		const int nt = bh_ingest_threads();
		uint64_t first[65];
		const int ns = nt > 64 ? 64 : (nt < 1 ? 1 : nt);
		#pragma omp parallel for num_threads(ns) schedule(static, 1)
		for (int t = 0; t < ns; ++t) {
			const uint64_t a = totQ * (uint64_t)t / ns, b = totQ * (uint64_t)(t + 1) / ns;
			uint64_t c = 0;
			for (uint64_t i = a; i < b; ++i) c += isNew[i];
			first[t + 1] = c;
		}
		first[0] = 0;
		for (int t = 0; t < ns; ++t) first[t + 1] += first[t];
		#pragma omp parallel for num_threads(ns) schedule(static, 1) reduction(+:totLen)
		for (int t = 0; t < ns; ++t) {
			const uint64_t a = totQ * (uint64_t)t / ns, b = totQ * (uint64_t)(t + 1) / ns;
			uint64_t u = first[t];
			for (uint64_t i = a; i < b; ++i) {
				Q->heads[i] = heads[refs[i].ix];
				if (isNew[i]) {
					Q->offset[u] = i;
					Q->len[u] = refs[i].len;
					Q->ed[u] = (uint16_t)bh_error_budget(thres, refs[i].len);                  /* burst.c:3074-3076 */
					totLen += refs[i].len;
					++u;
				}
			}
		}
	}
	Q->offset[numUniq] = totQ;
	free(isNew);
	QPH("dedupe");
	Q->codes = malloc(totLen * (do_rc ? 2 : 1) + 32);
	if (!Q->codes) { free(heads); free(refs); bh_queries_free(Q); return bh_set_error(BH_E_OOM, "OOM copying queries"); }
	Q->qoff[0] = 0;
	for (uint64_t i = 0; i < numUniq; ++i) Q->qoff[i + 1] = Q->qoff[i] + Q->len[i];
	if (do_rc) for (uint64_t i = 0; i < numUniq; ++i) Q->qoff[numUniq + i + 1] = Q->qoff[numUniq + i] + Q->len[i];
	uint32_t maxED = 0;
	#pragma omp parallel for num_threads(bh_ingest_threads()) reduction(max:maxED)
	for (uint64_t i = 0; i < numUniq; ++i) {
		const uint8_t *s = refs[Q->offset[i]].s;
		const uint32_t len = Q->len[i];
		memcpy(Q->codes + Q->qoff[i], s, len);
		Q->six[i] = (uint32_t)i; Q->emac[i] = Q->ed[i];
		if (Q->ed[i] > maxED) maxED = Q->ed[i];
		if (do_rc) {                                                                      /* burst.c:3095-3107 */
			uint8_t *d = Q->codes + Q->qoff[numUniq + i];
			for (uint32_t j = 0; j < len; ++j) d[j] = bh_rc_code(s[len - j - 1]);
			Q->six[numUniq + i] = (uint32_t)i; Q->rc[numUniq + i] = 1; Q->emac[numUniq + i] = Q->ed[i];
		}
	}
	{	/* nibble-packed copy: half the bytes over PCIe per batch */
		const uint64_t tot = Q->qoff[numEntries], nb4 = (tot + 1) / 2;
		Q->codes4 = malloc(nb4 + 16);
		if (!Q->codes4) { free(heads); free(refs); bh_queries_free(Q); return bh_set_error(BH_E_OOM, "OOM packing queries"); }
		Q->codes[tot] = 0;
		#pragma omp parallel for num_threads(bh_ingest_threads()) schedule(static)
		for (uint64_t b = 0; b < nb4; ++b) Q->codes4[b] = (uint8_t)((Q->codes[2 * b] & 15) | (Q->codes[2 * b + 1] & 15) << 4);
		memset(Q->codes4 + nb4, 0, 16);
		/* four symbols per byte for the batches that hold A/C/G/T only (a quarter of the bytes over PCIe), 2-byte lengths instead
		 * of 8-byte offsets, and per unique query whether it holds anything else (its reverse complement then does too) */
		const uint64_t nb2 = (tot + 3) / 4;
		Q->codes2 = malloc(nb2 + 16); Q->len16 = malloc((numUniq + 1) * sizeof(*Q->len16)); Q->ambBefore = malloc((numUniq + 2) * sizeof(*Q->ambBefore));
		if (!Q->codes2 || !Q->len16 || !Q->ambBefore) { free(heads); free(refs); bh_queries_free(Q); return bh_set_error(BH_E_OOM, "OOM packing queries"); }
		Q->codes[tot + 1] = Q->codes[tot + 2] = 0;
		#pragma omp parallel for num_threads(bh_ingest_threads()) schedule(static)
		for (uint64_t b = 0; b < nb2; ++b)
			Q->codes2[b] = (uint8_t)(((Q->codes[4 * b] - 1u) & 3u) | ((Q->codes[4 * b + 1] - 1u) & 3u) << 2 | ((Q->codes[4 * b + 2] - 1u) & 3u) << 4 | ((Q->codes[4 * b + 3] - 1u) & 3u) << 6);
		memset(Q->codes2 + nb2, 0, 16);
		#pragma omp parallel for num_threads(bh_ingest_threads()) schedule(static)
		for (uint64_t i = 0; i < numUniq; ++i) {
			const uint8_t *s = Q->codes + Q->qoff[i];
			uint32_t amb = 0;
			for (uint32_t j = 0; j < Q->len[i]; ++j) amb |= (uint32_t)(s[j] - 1u) > 3u;
			Q->ambBefore[i + 1] = amb;
			Q->len16[i] = (uint16_t)Q->len[i];
		}
		Q->ambBefore[0] = 0;
		for (uint64_t i = 0; i < numUniq; ++i) Q->ambBefore[i + 1] += Q->ambBefore[i];
	}
	QPH("copy + reverse complement");
	Q->totQ = totQ; Q->numUniq = numUniq; Q->numEntries = numEntries;
	if (!do_accel || K > 0) bh_queries_bins(Q, do_accel, K, z);      /* (K = 0 with an accelerator: the caller does it once K is known) */
	QPH("bins");
	Q->maxLen = maxLen; Q->minLen = minLen; Q->maxED = maxED;
	free(heads); free(refs);
	return BH_OK;
}
