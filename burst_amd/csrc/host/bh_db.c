/* bh_db.c -- database side of the host: .edx / .acx readers (burst.c:2842-2975, 3535-3594), the direct-FASTA
 * clumping used when -r is not a database (process_references QUICK path, burst.c:1840-1858, 2109-2190, 2687-2741),
 * and writers/builders for both formats (dump_edb 2758-2839, make_accelerator 3304-3532) used by the tests and the bench.
 */
#define _GNU_SOURCE
#include "burst_host.h"
#include <stdlib.h>
#include <string.h>
#include <omp.h>
#include <fcntl.h>
#include <pthread.h>
#include <unistd.h>
#include <sys/stat.h>
#include <sys/mman.h>

static void *own(BhDb *db, void *p) {      /* a full table does not lose track of the block: it is given back and the caller sees an allocation failure */
	if (p && db->nOwned >= 32) { free(p); return NULL; }
	if (p) db->owned[db->nOwned++] = p;
	return p;
}

void bh_db_free(BhDb *db) {
	if (!db) return;
	for (int i = 0; i < db->nOwned; ++i) free(db->owned[i]);
	if (db->mapBase) munmap(db->mapBase, (size_t)db->mapLen);
	memset(db, 0, sizeof *db);
}

int bh_is_edx(const char *path) {
	FILE *f = fopen(path, "rb");
	if (!f) return bh_set_error(BH_E_USAGE, "ERROR: invalid input file.");
	int c = fgetc(f);
	fclose(f);
	if (c == EOF) return bh_set_error(BH_E_USAGE, "ERROR: invalid input file.");
	return (c & 0x80) ? 1 : 0;
}

#define RD(ptr, size, n) do { if (fread((ptr), (size), (n), in) != (size_t)(n)) { fclose(in); bh_db_free(db); \
	return bh_set_error(BH_E_USAGE, "ERROR: truncated database %s", path); } } while (0)
#define ALLOC(dst, bytes) do { (dst) = own(db, malloc((size_t)(bytes) + 1)); if (!(dst)) { fclose(in); bh_db_free(db); \
	return bh_set_error(BH_E_OOM, "OOM:read_edb"); } } while (0)

static int read_region(const char *path, uint64_t off, void *dst, uint64_t n);
static int derive_refixsrt(BhDb *db) {
	if (db->refDedupIx) {                                        /* burst.c:3688-3693 */
		db->refIxSrt = own(db, malloc((size_t)db->totR * sizeof(uint32_t) + 4));
		if (!db->refIxSrt) return bh_set_error(BH_E_OOM, "OOM:RefIxSrt");
		for (uint32_t i = 0; i < db->totR; ++i) db->refIxSrt[i] = db->tmpRIX[db->refDedupIx[i]];
	} else db->refIxSrt = db->tmpRIX;
	return BH_OK;
}

int bh_edx_read(const char *path, BhDb *db) {
	memset(db, 0, sizeof *db);
	FILE *in = fopen(path, "rb");
	if (!in) return bh_set_error(BH_E_USAGE, "ERROR: cannot parse EDB");
	int cb = fgetc(in);
	if (cb == EOF) { fclose(in); return bh_set_error(BH_E_USAGE, "ERROR: invalid input file."); }
	int ver = cb & 0xF;
	if (ver == 2) { fclose(in); return bh_set_error(BH_E_IO, "ERROR: Old DB version. Re-make with new version."); }
	if (ver != 3) { fclose(in); return bh_set_error(BH_E_USAGE, "ERROR: invalid database version %d", ver); }
	db->rebase = (cb >> 6) & 1;
	int hasFP = (cb >> 5) & 1;
	db->xalpha = (cb >> 4) & 1;
	uint64_t totRefHeadLen = 0;
	RD(&totRefHeadLen, 8, 1); RD(&db->shear, 4, 1); RD(&db->totR, 4, 1); RD(&db->origTotR, 4, 1);
	RD(&db->numRclumps, 4, 1); RD(&db->maxLenR, 4, 1);
	ALLOC(db->headDump, totRefHeadLen + 1);
	RD(db->headDump, 1, totRefHeadLen);
	db->headDump[totRefHeadLen] = 0;
	RD(&db->numRefHeads, 4, 1);
	char **uniq;
	ALLOC(uniq, (size_t)db->numRefHeads * sizeof(char *));
	{
		char *p = db->headDump, *end = db->headDump + totRefHeadLen;
		for (uint32_t i = 0; i < db->numRefHeads; ++i) {
			uniq[i] = p;
			while (p < end && *p) ++p;
			if (p < end) ++p;
		}
	}
	ALLOC(db->refMap, (size_t)db->origTotR * 4);
	RD(db->refMap, 4, db->origTotR);
	ALLOC(db->refHead, (size_t)db->origTotR * sizeof(char *));
	for (uint32_t i = 0; i < db->origTotR; ++i) {
		if (db->refMap[i] >= db->numRefHeads) { fclose(in); bh_db_free(db); return bh_set_error(BH_E_USAGE, "ERROR: corrupt RefMap in %s", path); }
		db->refHead[i] = uniq[db->refMap[i]];
	}
	if (db->rebase) { ALLOC(db->refStart, (size_t)db->origTotR * 4); RD(db->refStart, 4, db->origTotR); }
	if (db->totR != db->origTotR) { ALLOC(db->refDedupIx, ((size_t)db->totR + 1) * 4); RD(db->refDedupIx, 4, db->totR + 1); }
	ALLOC(db->tmpRIX, (size_t)db->origTotR * 4);
	RD(db->tmpRIX, 4, db->origTotR);
	ALLOC(db->clumpLen, (size_t)db->numRclumps * 4);
	RD(db->clumpLen, 4, db->numRclumps);
	uint64_t words = 0; uint32_t maxL = db->maxLenR;
	for (uint32_t i = 0; i < db->numRclumps; ++i) {
		if (db->clumpLen[i] > maxL) maxL = db->clumpLen[i];
		words += db->clumpLen[i] / 2u + (db->clumpLen[i] & 1);
	}
	db->maxLenR = maxL;
	/* The clump area.  Gigabytes of it are MAPPED, not copied: the file's pages (page cache, or RAM itself when the file lies in /dev/shm)
	 * are this process's copy and every other process's that reads the same database -- the ranks of a node, one process per GPU, used to
	 * hold one 31.5 GB copy each.  The upload reads it once, front to back.  BURST_HOST_EDX_COPY=1: the private copy of rounds 1-4. */
	const uint64_t at = (uint64_t)ftello(in);
	int mapped = 0;
	if (words * 16 >= (64u << 20) && !(getenv("BURST_HOST_EDX_COPY") && atoi(getenv("BURST_HOST_EDX_COPY")))) {
		struct stat st;
		const long pg = sysconf(_SC_PAGESIZE);
		if (!fstat(fileno(in), &st) && (uint64_t)st.st_size >= at + words * 16 && pg > 0) {
			const uint64_t a0 = at & ~((uint64_t)pg - 1);
			void *m = mmap(NULL, (size_t)(at + words * 16 - a0), PROT_READ, MAP_SHARED, fileno(in), (off_t)a0);
			if (m != MAP_FAILED) {
				db->mapBase = m; db->mapLen = at + words * 16 - a0;
				db->packed = (uint8_t *)m + (at - a0);
				(void)madvise(m, (size_t)db->mapLen, MADV_WILLNEED);
				/* the page-table entries now, by a team: the upload's copy thread would otherwise take 8 M page faults one by one */
				const uint64_t np = (db->mapLen + (uint64_t)pg - 1) / (uint64_t)pg;
				uint64_t sum = 0;
				#pragma omp parallel for schedule(static) reduction(+:sum) num_threads(omp_get_max_threads() > 16 ? 16 : omp_get_max_threads())
				for (uint64_t i = 0; i < np; ++i) sum += ((volatile const uint8_t *)m)[i * (uint64_t)pg];
				db->mapLen += sum & 0;      /* (keeps the reads) */
				mapped = 1;
			}
		}
	}
	if (!mapped) {
		ALLOC(db->packed, (words + 1) * 16);
		if (words * 16 < (64u << 20)) RD(db->packed, 16, words);
		else if (read_region(path, at, db->packed, words * 16)) {      /* gigabytes: several threads copy out of the page cache */
			fclose(in); bh_db_free(db); return bh_set_error(BH_E_USAGE, "ERROR: truncated database %s", path);
		}
	}
	db->packedWords = words;
	(void)hasFP;   /* fingerprint tables, if any, follow and are ignored (-f is out of scope) */
	fclose(in);
	/* the tables index each other (and the report indexes refHead / refStart through them): refuse anything out of range */
	if ((uint64_t)db->numRclumps * 16 < db->totR || db->totR > db->origTotR) { bh_db_free(db); return bh_set_error(BH_E_USAGE, "ERROR: corrupt header counts in %s", path); }
	for (uint32_t i = 0; i < db->origTotR; ++i) if (db->tmpRIX[i] >= db->origTotR) { bh_db_free(db); return bh_set_error(BH_E_USAGE, "ERROR: corrupt TmpRIX in %s", path); }
	if (db->refDedupIx) {
		if (db->refDedupIx[db->totR] > db->origTotR) { bh_db_free(db); return bh_set_error(BH_E_USAGE, "ERROR: corrupt RefDedupIx in %s", path); }
		for (uint32_t i = 0; i < db->totR; ++i) if (db->refDedupIx[i] > db->refDedupIx[i + 1] || db->refDedupIx[i] >= db->origTotR) { bh_db_free(db); return bh_set_error(BH_E_USAGE, "ERROR: corrupt RefDedupIx in %s", path); }
	}
	{ int rcx = derive_refixsrt(db); if (rcx) { bh_db_free(db); return rcx; } }
	return BH_OK;
}
#undef RD
#undef ALLOC

/* large sequential read split over a few threads (the copy out of the page cache is what limits a 7 GB accelerator) */
static int read_region(const char *path, uint64_t off, void *dst, uint64_t n) {
	if (!n) return 0;
	const uint64_t chunk = 64ull << 20;
	const uint64_t nchunks = (n + chunk - 1) / chunk;
	int bad = 0;
	int team = omp_get_max_threads(); if (team > 16) team = 16;
	#pragma omp parallel num_threads(team)
	{
		FILE *f = fopen(path, "rb");
		if (!f) {
			#pragma omp atomic write
			bad = 1;
		}
		#pragma omp for schedule(dynamic, 1)
		for (uint64_t c = 0; c < nchunks; ++c) {
			if (!f) continue;
			const uint64_t o = c * chunk, m = n - o < chunk ? n - o : chunk;
			if (fseeko(f, (off_t)(off + o), SEEK_SET) || fread((uint8_t *)dst + o, 1, m, f) != m) {
				#pragma omp atomic write
				bad = 1;
			}
		}
		if (f) fclose(f);
	}
	return bad;
}

/* K = 12 or 15 (the reference's two builds, burst.c:96-99), or 0 = work it out: the file size must be EXACTLY
 * 5 + 4 * 4^K + list bytes + 4 * badSz for the K it was written with (a DB15 file read as DB12 would otherwise pass a
 * "large enough" check and decode part of its length table as list entries). */
int bh_acx_read(const char *path, int K, int z, BhDb *db) {
	FILE *in = fopen(path, "rb");
	if (!in) return bh_set_error(BH_E_USAGE, "Cannot read accelerator '%s'", path);
	int cb = fgetc(in);
	int ver = cb & 0xF, didZ = (cb >> 6) & 1;
	if (cb == EOF || cb < 128 || (ver != 0 && ver != 1)) { fclose(in); return bh_set_error(BH_E_USAGE, "ERROR: invalid accelerator [%d:%d]", cb, ver); }
	if (didZ && !z) { fclose(in); return bh_set_error(BH_E_USAGE, "ERROR: Accelerator built without '-y'; can't use '-y'"); }
	uint32_t szBL = 0;
	struct stat sb;
	if (fread(&szBL, 4, 1, in) != 1 || stat(path, &sb)) { fclose(in); return bh_set_error(BH_E_USAGE, "ERROR: truncated accelerator %s", path); }
	const uint64_t fsize = (uint64_t)sb.st_size;
	const int tryK[2] = {K ? K : 12, K ? 0 : 15};
	uint32_t *lens = NULL; uint64_t bytes = 0, nw = 0; int foundK = 0;
	for (int t = 0; t < 2 && !foundK && tryK[t]; ++t) {
		nw = 1ull << (2 * tryK[t]);
		if (fsize < 5 + nw * 4 + (uint64_t)szBL * 4) continue;
		lens = malloc(nw * 4);
		if (!lens) { fclose(in); return bh_set_error(BH_E_OOM, "OOM:Lens_rd"); }
		if (read_region(path, 5, lens, nw * 4)) { free(lens); lens = NULL; continue; }
		bytes = 0;
		#pragma omp parallel for reduction(+:bytes) schedule(static)
		for (uint64_t i = 0; i < nw; ++i) bytes += ver == 0 ? (uint64_t)(lens[i] / 2u) * 5 + (lens[i] & 1) * 3 : (uint64_t)lens[i] * 3;
		if (fsize == 5 + nw * 4 + bytes + (uint64_t)szBL * 4) foundK = tryK[t];
		else { free(lens); lens = NULL; }
	}
	if (!foundK) {
		fclose(in);
		if (K) return bh_set_error(BH_E_USAGE, "ERROR: accelerator %s does not have the size of a K=%d accelerator (truncated, or built with the other K?)", path, K);
		return bh_set_error(BH_E_USAGE, "ERROR: accelerator %s is neither a complete K=12 nor a complete K=15 accelerator", path);
	}
	K = foundK;
	if (!own(db, lens)) { fclose(in); return bh_set_error(BH_E_OOM, "OOM:WordDump_rd"); }
	uint32_t *bl = own(db, malloc(((size_t)szBL + 1) * 4));
	uint8_t *lists = own(db, malloc(bytes + 16));
	if (!bl || !lists) { fclose(in); return bh_set_error(BH_E_OOM, "OOM:WordDump_rd"); }
	int bad = read_region(path, 5 + nw * 4, lists, bytes);
	if (!bad) bad |= fseeko(in, (off_t)(5 + nw * 4 + bytes), SEEK_SET) != 0 || fread(bl, 4, szBL, in) != szBL;
	fclose(in);
	if (bad) return bh_set_error(BH_E_USAGE, "ERROR: truncated accelerator %s (was it built with K=%d?)", path, K);
	memset(lists + bytes, 0, 16);
	db->hasAcx = 1; db->K = K; db->acxFmt = ver; db->acxZ = didZ;
	db->acxLens = lens; db->acxLists = lists; db->acxListBytes = bytes; db->badList = bl; db->badSz = szBL;
	return BH_OK;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Direct FASTA -> clumps.  Line-based parser as parse_tl_fasta (burst.c:484-533): a record starts at a line whose
 * first byte is '>', sequence lines are concatenated, lines starting with ' ' or empty are skipped. */
typedef struct { char *head; uint8_t *seq; uint32_t len; } RefRec;

static int parse_ref_fasta(const char *path, RefRec **out, uint32_t *n_out, char **dump_out) {
	FILE *f = fopen(path, "rb");
	if (!f) return bh_set_error(BH_E_IO, "Cannot open FASTA file: %s.", path);
	fseeko(f, 0, SEEK_END); uint64_t sz = (uint64_t)ftello(f); rewind(f);
	char *dump = malloc(sz + 2);
	if (!dump) { fclose(f); return bh_set_error(BH_E_OOM, "OOM:parse_tl_fasta"); }
	if (fread(dump, 1, sz, f) != sz) { fclose(f); free(dump); return bh_set_error(BH_E_IO, "short read on %s", path); }
	fclose(f);
	dump[sz] = '\n'; dump[sz + 1] = 0;
	uint8_t c2n[256]; bh_char2code(c2n);
	uint32_t cap = 1024, n = 0; int lastHd = 0;
	RefRec *R = malloc(cap * sizeof(*R));
	uint8_t *wr = NULL;     /* sequences are compacted in place, behind the read cursor */
	for (char *p = dump, *end = dump + sz; p < end;) {
		char *nl = memchr(p, '\n', (size_t)(end + 1 - p));
		char *le = nl;
		if (le > p && le[-1] == '\r') --le;
		if (*p == '>') {
			if (!lastHd) {
				if (n == cap) { cap *= 2; R = realloc(R, cap * sizeof(*R)); }
				if (n) *wr = 0;
				*le = 0;
				R[n].head = p + 1; R[n].seq = (uint8_t *)le + 1; R[n].len = 0;
				wr = R[n].seq; ++n; lastHd = 1;
			}
		} else if (le > p && *p != ' ' && n) {
			lastHd = 0;
			for (char *s = p; s < le; ++s) *wr++ = c2n[(uint8_t)*s];
			R[n - 1].len = (uint32_t)(wr - R[n - 1].seq);
		}
		p = nl + 1;
	}
	if (wr) *wr = 0;
	if (lastHd && n) --n;
	*out = R; *n_out = n; *dump_out = dump;
	return BH_OK;
}

/* Clump formation order (burst.c:2149-2186).  The reference's tie order among identical fragments is whatever its sort
 * calls leave behind, so the same calls are made here -- the same libc qsort on 16-byte records with comparators of
 * the same meaning -- and the .edx comes out byte-identical to a single-threaded reference run also when the
 * database holds duplicate fragments:
 *   1. all fragments by length (cmpPackLen, 1335-1338);
 *   2. pods of lengths within LATENCY: up to 256 members by strcmp of the NUL-terminated rest of the original sequence
 *      from the fragment's start (cmpPackSeq, 1341-1344 -- it reads past the fragment's own length);
 *   3. larger pods and always the last pod: buckets by the first five symbols in input order, inside a bucket by the
 *      symbols from the sixth up to the shorter length, then shorter first, never "equal" (parallel_sort_tuxedo, 390-405). */
typedef struct { const uint8_t *s; uint32_t len, ix; } Tux;
static int tux_len_cmp(const void *a, const void *b) {
	const Tux *A = a, *B = b;
	return A->len < B->len ? -1 : A->len > B->len;
}
static int tux_rest_cmp(const void *a, const void *b) {
	return strcmp((const char *)((const Tux *)a)->s, (const char *)((const Tux *)b)->s);
}
static int tux_bucket_cmp(const void *a, const void *b) {
	const Tux *A = a, *B = b;
	const uint32_t ml = A->len < B->len ? A->len : B->len;
	uint32_t i = 5;
	for (; i < ml; ++i) if (A->s[i] != B->s[i]) break;
	if (i < ml) return (int)A->s[i] - (int)B->s[i];
	return A->len < B->len ? -1 : 1;
}
typedef struct { uint32_t nib, pos; } NibPos;
static int nibpos_cmp(const void *a, const void *b) {
	const NibPos *A = a, *B = b;
	if (A->nib != B->nib) return A->nib < B->nib ? -1 : 1;
	return A->pos < B->pos ? -1 : (A->pos > B->pos);
}
static void tux_bucket_sort(Tux *pack, uint32_t n) {
	NibPos *np = malloc((size_t)n * sizeof(*np));
	Tux *tmp = malloc((size_t)n * sizeof(*tmp));
	for (uint32_t i = 0; i < n; ++i) {
		const uint8_t *s = pack[i].s;
		np[i].nib = (uint32_t)s[0] << 16 | (uint32_t)s[1] << 12 | (uint32_t)s[2] << 8 | (uint32_t)s[3] << 4 | (uint32_t)s[4];
		np[i].nib &= 0xFFFFFu; np[i].pos = i;
	}
	qsort(np, n, sizeof(*np), nibpos_cmp);                 /* buckets ascending, members in input order */
	for (uint32_t i = 0; i < n; ++i) tmp[i] = pack[np[i].pos];
	/* the buckets are independent of each other: sorted side by side (each by the same libc qsort call as a serial pass makes) */
	uint32_t nb = 0, *bstart = malloc(((size_t)(n < (1u << 20) ? n : (1u << 20)) + 2) * 4);
	for (uint32_t a = 0; a < n;) {
		uint32_t b = a + 1;
		while (b < n && np[b].nib == np[a].nib) ++b;
		bstart[nb++] = a;
		a = b;
	}
	bstart[nb] = n;
	#pragma omp parallel for schedule(dynamic, 1) if (n > 65536)
	for (uint32_t k = 0; k < nb; ++k) qsort(tmp + bstart[k], bstart[k + 1] - bstart[k], sizeof(*tmp), tux_bucket_cmp);
	free(bstart);
	memcpy(pack, tmp, (size_t)n * sizeof(*tmp));
	free(np); free(tmp);
}


int bh_db_from_fasta(const char *path, uint32_t maxLenQ, float thres, int do_shear, long shear_len, int dedupe, BhDb *db) {
	return bh_db_from_fasta_ex(path, maxLenQ, thres, do_shear, shear_len, dedupe, 16, db);      /* LATENCY, burst.c:83 */
}
int bh_db_from_fasta_ex(const char *path, uint32_t maxLenQ, float thres, int do_shear, long shear_len, int dedupe, uint32_t latency, BhDb *db) {
	memset(db, 0, sizeof *db);
	RefRec *R = NULL; uint32_t nR = 0; char *dump = NULL;
	int rc = parse_ref_fasta(path, &R, &nR, &dump);
	if (rc) return rc;
	own(db, dump);
	if (!nR) { free(R); bh_db_free(db); return bh_set_error(BH_E_USAGE, "ERROR: no references in %s", path); }
	/* shear (burst.c:1852-1858, 2109-2141) */
	uint32_t totR = nR;
	char **head; const uint8_t **seq; uint32_t *len, *start = NULL;
	uint32_t shear_cap = 0;
	if (do_shear && shear_len > 0) {
		uint32_t minShear = (uint32_t)(maxLenQ / thres), shear = minShear > (uint32_t)shear_len ? minShear : (uint32_t)shear_len, ov = minShear;
		uint64_t cnt = 0;
		for (uint32_t i = 0; i < nR; ++i) {
			long unit = (long)R[i].len - (long)ov; if (unit < 0) unit = 1;
			cnt += (uint64_t)(unit / shear + (unit % shear != 0));
		}
		totR = (uint32_t)cnt;
		head = malloc((size_t)totR * sizeof(*head)); seq = malloc((size_t)totR * sizeof(*seq));
		len = malloc((size_t)totR * 4); start = own(db, malloc((size_t)totR * 4));
		uint32_t maxL = shear + ov, x = 0;
		shear_cap = maxL;
		for (uint32_t i = 0; i < nR; ++i) {
			long unit = (long)R[i].len - (long)ov; if (unit < 0) unit = 1;
			for (long j = 0; j < unit; j += shear) {
				head[x] = R[i].head; seq[x] = R[i].seq + j; start[x] = (uint32_t)j;
				uint32_t l = R[i].len - (uint32_t)j;
				len[x++] = l > maxL ? maxL : l;
			}
		}
		db->rebase = 1; db->shear = minShear;
	} else {
		head = malloc((size_t)totR * sizeof(*head)); seq = malloc((size_t)totR * sizeof(*seq)); len = malloc((size_t)totR * 4);
		for (uint32_t i = 0; i < nR; ++i) head[i] = R[i].head, seq[i] = R[i].seq, len[i] = R[i].len;
	}
	/* order: by length, then lexicographically inside pods whose lengths differ by at most LATENCY (default 16, `-l`)
	 * (burst.c:2149-2186); `-l 0` keeps the input order (2187-2189) */
	uint32_t *srt = own(db, malloc(((size_t)totR + 1) * 4));
	uint32_t maxLenR = 0;
	if (latency) {
		Tux *T = malloc((size_t)totR * sizeof(*T));
		for (uint32_t i = 0; i < totR; ++i) T[i].s = seq[i], T[i].len = len[i], T[i].ix = i;
		qsort(T, totR, sizeof(*T), tux_len_cmp);
		maxLenR = T[totR - 1].len;
		uint32_t prev = 0, tol = T[0].len;
		for (uint32_t i = 1; i < totR; ++i) {
			if (T[i].len > tol + latency) {
				tol = T[i].len;
				if (i - prev > 1) { if (i - prev > 256) tux_bucket_sort(T + prev, i - prev); else qsort(T + prev, i - prev, sizeof(*T), tux_rest_cmp); }
				prev = i;
			}
		}
		if (prev < totR - 1) tux_bucket_sort(T + prev, totR - prev);
		for (uint32_t i = 0; i < totR; ++i) srt[i] = T[i].ix;
		free(T);
	} else {
		if (shear_cap) maxLenR = shear_cap;       /* the reference starts from shear + overlap here ("may actually be less", burst.c:2133) */
		for (uint32_t i = 0; i < totR; ++i) { srt[i] = i; if (len[i] > maxLenR) maxLenR = len[i]; }
	}
	db->origTotR = totR; db->tmpRIX = srt; db->refIxSrt = srt; db->totR = totR;
	if (dedupe) {                                                /* burst.c:2192-2230 */
		uint32_t *dd = own(db, calloc((size_t)totR + 2, 4)), uix = 0;
		for (uint32_t i = 1; i < totR; ++i) {
			uint32_t a = srt[i], b = srt[i - 1];
			if (!(len[a] == len[b] && !memcmp(seq[a], seq[b], len[a]))) dd[++uix] = i;
		}
		dd[++uix] = totR;
		for (uint32_t i = 0; i < uix; ++i) {                     /* lowest original index leads each duplicate set: the same chain of */
			uint32_t bix = srt[dd[i]];                            /* swaps with the running minimum as burst.c:2213-2220 */
			for (uint32_t m = dd[i] + 1; m < dd[i + 1]; ++m) if (srt[m] < bix) { bix = srt[m]; srt[m] = srt[dd[i]]; srt[dd[i]] = bix; }
		}
		if (uix != totR) {
			uint32_t *u = own(db, malloc(((size_t)uix + 1) * 4));
			for (uint32_t i = 0; i < uix; ++i) u[i] = srt[dd[i]];
			db->refIxSrt = u; db->refDedupIx = dd; db->totR = uix;
		}
	}
	/* clumps of 16 (burst.c:2687-2737).  The reference copies symbol j while RefLen >= j, i.e. one symbol past a
	 * shorter lane's end: the terminator (0) or, for a shear, the next base of the parent sequence.  Reproduced. */
	const uint32_t nU = db->totR, nC = (nU + 15) / 16;
	uint32_t *cl = own(db, malloc(((size_t)nC + 1) * 4));
	uint64_t words = 0;
	for (uint32_t c = 0; c < nC; ++c) {
		uint32_t m = 0;
		for (uint32_t k = 16 * c; k < nU && k < 16 * c + 16; ++k) if (len[db->refIxSrt[k]] > m) m = len[db->refIxSrt[k]];
		cl[c] = m; words += m / 2u + (m & 1);
	}
	uint8_t *packed = own(db, calloc(words + 1, 16));
	if (!packed) { free(head); free(seq); free(len); free(R); bh_db_free(db); return bh_set_error(BH_E_OOM, "OOM:RefClump"); }
	uint64_t *cw = malloc(((size_t)nC + 1) * 8);
	if (!cw) { free(head); free(seq); free(len); free(R); bh_db_free(db); return bh_set_error(BH_E_OOM, "OOM:RefClump"); }
	cw[0] = 0;
	for (uint32_t c = 0; c < nC; ++c) cw[c + 1] = cw[c] + cl[c] / 2u + (cl[c] & 1);
	#pragma omp parallel for schedule(static, 256)
	for (uint32_t c = 0; c < nC; ++c) {
		const uint64_t w0 = cw[c];
		for (uint32_t k = 16 * c; k < nU && k < 16 * c + 16; ++k) {
			const uint32_t r = db->refIxSrt[k], L = len[r];
			const uint8_t *s = seq[r];
			for (uint32_t j = 0; j < cl[c] && j <= L; ++j)
				packed[(w0 + j / 2) * 16 + (k & 15)] |= (uint8_t)((s[j] & 15) << (4 * (j & 1)));
		}
	}
	free(cw);
	db->clumpLen = cl; db->numRclumps = nC; db->packed = packed; db->packedWords = words; db->maxLenR = maxLenR;
	/* headers: one per sheared reference; RefMap = unique header index, built as dump_edb does (burst.c:2769-2786) */
	db->refHead = own(db, malloc((size_t)totR * sizeof(char *)));
	db->refMap = own(db, malloc((size_t)totR * 4));
	for (uint32_t i = 0; i < totR; ++i) db->refHead[i] = head[i];
	{
		typedef struct { const char *s; uint32_t ix; } HP;
		HP *H = malloc((size_t)totR * sizeof(*H));
		for (uint32_t i = 0; i < totR; ++i) H[i].s = head[i], H[i].ix = i;
		int hp_cmp(const void *a, const void *b) { return strcmp(((const HP *)a)->s, ((const HP *)b)->s); }
		qsort(H, totR, sizeof(*H), hp_cmp);
		uint32_t nix = 0;
		for (uint32_t i = 0; i < totR; ++i) {
			if (i && strcmp(H[i].s, H[i - 1].s)) ++nix;
			db->refMap[H[i].ix] = nix;
		}
		db->numRefHeads = nix + 1;
		free(H);
	}
	db->refStart = start;
	db->identityMap = 1;
	free(head); free(seq); free(len); free(R);
	return BH_OK;
}

int bh_edx_write(const BhDb *db, const char *path, long db_qlen, float thres) {
	FILE *o = fopen(path, "wb");
	if (!o) return bh_set_error(BH_E_IO, "ERROR: Cannot open output: %s", path);
	/* unique sorted headers (burst.c:2769-2786) */
	uint32_t nH = db->numRefHeads;
	const char **uh = calloc(nH, sizeof(*uh));
	for (uint32_t i = 0; i < db->origTotR; ++i) uh[db->refMap[i]] = db->refHead[i];
	uint64_t hl = 0;
	for (uint32_t i = 0; i < nH; ++i) hl += strlen(uh[i]) + 1;
	uint8_t ctrl = (uint8_t)(1 << 7 | (db->rebase ? 1 : 0) << 6 | 0 << 5 | 0 << 4 | 3);
	uint32_t shear = db->rebase ? (uint32_t)(db_qlen / thres) : 0;
	fwrite(&ctrl, 1, 1, o); fwrite(&hl, 8, 1, o); fwrite(&shear, 4, 1, o);
	fwrite(&db->totR, 4, 1, o); fwrite(&db->origTotR, 4, 1, o); fwrite(&db->numRclumps, 4, 1, o); fwrite(&db->maxLenR, 4, 1, o);
	for (uint32_t i = 0; i < nH; ++i) fwrite(uh[i], 1, strlen(uh[i]) + 1, o);
	fwrite(&nH, 4, 1, o);
	fwrite(db->refMap, 4, db->origTotR, o);
	if (db->rebase) fwrite(db->refStart, 4, db->origTotR, o);
	if (db->totR != db->origTotR) fwrite(db->refDedupIx, 4, (size_t)db->totR + 1, o);
	fwrite(db->tmpRIX, 4, db->origTotR, o);
	fwrite(db->clumpLen, 4, db->numRclumps, o);
	fwrite(db->packed, 16, db->packedWords, o);
	free(uh);
	if (fclose(o)) return bh_set_error(BH_E_IO, "ERROR: write failed: %s", path);
	return BH_OK;
}

/* Several .edx files -> one, section by section (no chunk is ever held in memory): the databases are laid end to end -- headers,
 * reference maps (shifted by the headers in front), fragment starts, sort permutations (shifted by the fragments in front), clump
 * lengths, clump areas.  Reference number = 16 * clump + lane, so a part that does not fill its last clump leaves lanes of padding
 * in the middle of the merged database: those, and parts that carry duplicate-fragment tables (totR != origTotR: strain-level
 * redundancy, bench.py --db-profile strains), are expressed through the merged file's own RefDedupIx -- a padding lane is a unique
 * reference with NO original (an empty range; its symbols are pads, nothing aligns to it), a part without a table contributes the
 * identity.  Without either the merged file has no table, as before.  For databases too large to be BUILT in one piece in the
 * host memory at hand (bench.py: the metric's 31.5 GB stand-in in a 300 GB container); each part is what -d QUICK makes of its
 * share of the references.  No reference counterpart. */
typedef struct EdxHead { uint8_t ctrl; uint64_t hl; uint32_t shear, totR, origTotR, numRclumps, maxLenR, nH; uint64_t off_heads, off_map, off_start, off_dedup, off_rix, off_clen, off_packed, words; } EdxHead;
static int edx_head(const char *path, EdxHead *h) {
	FILE *f = fopen(path, "rb");
	if (!f) return bh_set_error(BH_E_IO, "cannot read %s", path);
	int ok = fread(&h->ctrl, 1, 1, f) == 1 && fread(&h->hl, 8, 1, f) == 1 && fread(&h->shear, 4, 1, f) == 1 && fread(&h->totR, 4, 1, f) == 1 &&
	         fread(&h->origTotR, 4, 1, f) == 1 && fread(&h->numRclumps, 4, 1, f) == 1 && fread(&h->maxLenR, 4, 1, f) == 1;
	h->off_heads = 29;
	if (ok) ok = !fseeko(f, (off_t)(h->off_heads + h->hl), SEEK_SET) && fread(&h->nH, 4, 1, f) == 1;
	h->off_map = h->off_heads + h->hl + 4;
	const int rebase = (h->ctrl >> 6) & 1;
	h->off_start = h->off_map + 4ull * h->origTotR;
	h->off_dedup = h->off_start + (rebase ? 4ull * h->origTotR : 0);
	h->off_rix = h->off_dedup + (h->totR != h->origTotR ? 4ull * ((uint64_t)h->totR + 1) : 0);      /* RefDedupIx[totR + 1] only in databases with duplicates (burst.c:2911-2916) */
	h->off_clen = h->off_rix + 4ull * h->origTotR;
	h->off_packed = h->off_clen + 4ull * h->numRclumps;
	h->words = 0;
	if (ok) {
		ok = !fseeko(f, (off_t)h->off_clen, SEEK_SET);
		uint32_t buf[4096];
		for (uint32_t c = 0; ok && c < h->numRclumps;) {
			const uint32_t n = h->numRclumps - c < 4096 ? h->numRclumps - c : 4096;
			ok = fread(buf, 4, n, f) == n;
			for (uint32_t i = 0; i < n; ++i) h->words += buf[i] / 2u + (buf[i] & 1u);
			c += n;
		}
	}
	fclose(f);
	return ok ? BH_OK : bh_set_error(BH_E_IO, "truncated database %s", path);
}
/* copy n bytes of `in` (from offset off) to `out`; add > 0: the bytes are 32-bit numbers and `add` is added to each */
static int copy_section(FILE *out, const char *path, uint64_t off, uint64_t n, uint32_t add) {
	FILE *f = fopen(path, "rb");
	if (!f || fseeko(f, (off_t)off, SEEK_SET)) { if (f) fclose(f); return bh_set_error(BH_E_IO, "cannot read %s", path); }
	const size_t B = 8u << 20;
	uint8_t *buf = malloc(B);
	if (!buf) { fclose(f); return bh_set_error(BH_E_OOM, "OOM:merge"); }
	int rc = BH_OK;
	while (n && !rc) {
		const size_t k = n < B ? (size_t)n : B;
		if (fread(buf, 1, k, f) != k) { rc = bh_set_error(BH_E_IO, "truncated database %s", path); break; }
		if (add) { uint32_t *w = (uint32_t *)buf; for (size_t i = 0; i < k / 4; ++i) w[i] += add; }
		if (fwrite(buf, 1, k, out) != k) rc = bh_set_error(BH_E_IO, "write failed");
		n -= k;
	}
	free(buf); fclose(f);
	return rc;
}
int bh_edx_merge(const char *const *paths, int n, const char *out_path) {
	if (n < 1 || n > 64) return bh_set_error(BH_E_USAGE, "bad number of databases to merge (%d)", n);
	EdxHead h[64];
	uint64_t hl = 0, totR = 0, orig = 0, clumps = 0, nH = 0; uint32_t maxL = 0;
	int dd = 0;          /* the merged file carries a RefDedupIx: some part has one, or leaves padding lanes in the middle */
	for (int i = 0; i < n; ++i) {
		int rc = edx_head(paths[i], &h[i]);
		if (rc) return rc;
		if (h[i].totR != h[i].origTotR || (i + 1 < n && (h[i].totR & 15u))) dd = 1;
		if (h[i].numRclumps != (h[i].totR + 15u) / 16u) return bh_set_error(BH_E_USAGE, "%s: %u references in %u clumps", paths[i], h[i].totR, h[i].numRclumps);
		if (h[i].ctrl != h[0].ctrl || h[i].shear != h[0].shear) return bh_set_error(BH_E_USAGE, "%s was built with other settings than %s", paths[i], paths[0]);
		hl += h[i].hl; orig += h[i].origTotR; clumps += h[i].numRclumps; nH += h[i].nH;
		totR += i + 1 < n ? 16ull * h[i].numRclumps : h[i].totR;      /* (a part in the middle counts with the padding lanes of its last clump) */
		if (h[i].maxLenR > maxL) maxL = h[i].maxLenR;
	}
	if (totR > 0xFFFFFFFFull || orig > 0xFFFFFFFFull || clumps >= (1ull << 24) || nH > 0xFFFFFFFFull) return bh_set_error(BH_E_USAGE, "the merged database would have %lu references in %lu clumps: beyond the format", (unsigned long)totR, (unsigned long)clumps);
	if (dd && totR == orig) return bh_set_error(BH_E_USAGE, "the merged database would have as many unique references as originals: its duplicate table could not be told from none");
	FILE *o = fopen(out_path, "wb");
	if (!o) return bh_set_error(BH_E_IO, "ERROR: Cannot open output: %s", out_path);
	setvbuf(o, NULL, _IOFBF, 8u << 20);
	const uint32_t totR32 = (uint32_t)totR, orig32 = (uint32_t)orig, cl32 = (uint32_t)clumps, nH32 = (uint32_t)nH;
	int rc = BH_OK;
	if (fwrite(&h[0].ctrl, 1, 1, o) != 1 || fwrite(&hl, 8, 1, o) != 1 || fwrite(&h[0].shear, 4, 1, o) != 1 || fwrite(&totR32, 4, 1, o) != 1 || fwrite(&orig32, 4, 1, o) != 1 ||
	    fwrite(&cl32, 4, 1, o) != 1 || fwrite(&maxL, 4, 1, o) != 1) rc = bh_set_error(BH_E_IO, "ERROR: write failed: %s", out_path);
	for (int i = 0; i < n && !rc; ++i) rc = copy_section(o, paths[i], h[i].off_heads, h[i].hl, 0);
	if (!rc && fwrite(&nH32, 4, 1, o) != 1) rc = bh_set_error(BH_E_IO, "ERROR: write failed: %s", out_path);
	{ uint64_t hb = 0; for (int i = 0; i < n && !rc; ++i) { rc = copy_section(o, paths[i], h[i].off_map, 4ull * h[i].origTotR, (uint32_t)hb); hb += h[i].nH; } }
	if ((h[0].ctrl >> 6) & 1) for (int i = 0; i < n && !rc; ++i) rc = copy_section(o, paths[i], h[i].off_start, 4ull * h[i].origTotR, 0);
	if (dd) {          /* RefDedupIx[totR + 1]: where the originals of unique reference i start in the list of originals below */
		uint64_t ob = 0;
		uint32_t *buf = malloc((size_t)(1u << 20) * 4);
		if (!buf) rc = bh_set_error(BH_E_OOM, "OOM:merge");
		for (int i = 0; i < n && !rc; ++i) {
			const uint64_t lanes = i + 1 < n ? 16ull * h[i].numRclumps : h[i].totR;
			if (h[i].totR != h[i].origTotR) rc = copy_section(o, paths[i], h[i].off_dedup, 4ull * h[i].totR, (uint32_t)ob);
			else for (uint64_t k = 0; k < h[i].totR && !rc;) {
				const uint32_t m = h[i].totR - k < (1u << 20) ? (uint32_t)(h[i].totR - k) : (1u << 20);
				for (uint32_t j = 0; j < m; ++j) buf[j] = (uint32_t)(ob + k + j);
				if (fwrite(buf, 4, m, o) != m) rc = bh_set_error(BH_E_IO, "ERROR: write failed: %s", out_path);
				k += m;
			}
			ob += h[i].origTotR;
			const uint32_t end = (uint32_t)ob;          /* padding lanes: no originals */
			for (uint64_t k = h[i].totR; k < lanes && !rc; ++k) if (fwrite(&end, 4, 1, o) != 1) rc = bh_set_error(BH_E_IO, "ERROR: write failed: %s", out_path);
		}
		const uint32_t end = (uint32_t)ob;
		if (!rc && fwrite(&end, 4, 1, o) != 1) rc = bh_set_error(BH_E_IO, "ERROR: write failed: %s", out_path);
		free(buf);
	}
	{ uint64_t ob = 0; for (int i = 0; i < n && !rc; ++i) { rc = copy_section(o, paths[i], h[i].off_rix, 4ull * h[i].origTotR, (uint32_t)ob); ob += h[i].origTotR; } }
	for (int i = 0; i < n && !rc; ++i) rc = copy_section(o, paths[i], h[i].off_clen, 4ull * h[i].numRclumps, 0);
	for (int i = 0; i < n && !rc; ++i) rc = copy_section(o, paths[i], h[i].off_packed, 16ull * h[i].words, 0);
	if (fclose(o) && !rc) rc = bh_set_error(BH_E_IO, "ERROR: write failed: %s", out_path);
	return rc;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Accelerator: for every clump the set of K-mers (2 bits per base, first base most significant, burst.c:4097-4102)
 * occurring in any of its 16 lanes, expanded over IUPAC codes (AMBIGS, burst.c:1372-1375); words containing N are
 * skipped when N is penalised (burst.c:3368-3374); clumps whose expansion exceeds the budget go to the BadList
 * (burst.c:3341-3354).  Lists are written in ascending clump order (the reference's order with -t 1). */
static const uint8_t AMB_N[16] = {0, 1, 1, 1, 1, 4, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3};
static const uint8_t AMB[16][4] = {{0}, {0}, {1}, {2}, {3}, {0, 1, 2, 3}, {2, 3}, {0, 1}, {0, 2}, {1, 3}, {1, 2}, {0, 3}, {1, 2, 3}, {0, 1, 2}, {0, 1, 3}, {0, 2, 3}};

static void lane_codes(const BhDb *db, uint64_t w0, uint32_t L, uint32_t z, uint8_t *out) {
	for (uint32_t j = 0; j < L; ++j) {
		uint8_t b = db->packed[(w0 + j / 2) * 16 + z];
		out[j] = (j & 1) ? b >> 4 : b & 15;
	}
}

static void expand_word(const uint8_t *s, int K, int ix, uint32_t w, uint8_t *seen, uint32_t *cache, uint32_t *n) {
	if (ix == K) { if (!(seen[w >> 3] & (1 << (w & 7)))) { seen[w >> 3] |= (uint8_t)(1 << (w & 7)); cache[(*n)++] = w; } return; }
	for (int i = 0; i < AMB_N[s[ix]]; ++i) expand_word(s, K, ix + 1, (w << 2) | AMB[s[ix]][i], seen, cache, n);
}


int bh_acx_build(BhDb *db, int K, int z) { return bh_acx_build_ex(db, K, z, 0); }
/* skip_ambig = -sa: words holding any ambiguous symbol are left out, no BadList (burst.c:3341, 3360-3366) */
int bh_acx_build_ex(BhDb *db, int K, int z, int skip_ambig) {
	const uint64_t nw = 1ull << (2 * K);
	const uint32_t nC = db->numRclumps;
	const uint64_t fullSize = K > 14 ? 0x7FFFFFFFull : (1ull << 24);
	uint32_t *lens = own(db, calloc(nw, 4));
	uint64_t *coff = malloc(((size_t)nC + 1) * 8);
	uint8_t *isBad = calloc(nC, 1);
	if (!lens || !coff || !isBad) return bh_set_error(BH_E_OOM, "OOM:AccelerantF");
	coff[0] = 0;
	for (uint32_t c = 0; c < nC; ++c) coff[c + 1] = coff[c] + db->clumpLen[c] / 2u + (db->clumpLen[c] & 1);
	/* per-clump word sets, kept to fill the lists in a second sweep */
	uint32_t **words = calloc(nC, sizeof(*words)); uint32_t *nwords = calloc(nC, 4);
	int oom = 0;
	/* every thread owns a 4^K-bit "seen" map (128 MB at K = 15): bound the team so that the maps stay below ~2 GB */
	int team = omp_get_max_threads();
	if (K > 13 && team > 16) team = 16;
	/* expansion estimate per window with a ambiguous symbols: 3^a (N penalised) or 4^a -- the reference's table holds
	 * 61 for 4^3 (burst.c:3324), kept so that the same clumps land in the BadList */
	static const uint64_t POW3[16] = {1, 3, 9, 27, 81, 243, 729, 2187, 6561, 19683, 59049, 177147, 531441, 1594323, 4782969, 14348907};
	static const uint64_t POW4[16] = {1, 4, 16, 61, 256, 1024, 4096, 16384, 65536, 262144, 1048576, 4194304, 16777216, 67108864, 268435456, 1073741824};
	const uint64_t *powx = z ? POW3 : POW4;
	#pragma omp parallel num_threads(team)
	{
		uint8_t *seen = calloc(nw >> 3 ? nw >> 3 : 1, 1);
		uint32_t capc = 1u << 16; uint32_t *cache = malloc((size_t)capc * 4);
		uint8_t *lane = malloc((size_t)db->maxLenR + 64);
		#pragma omp for schedule(dynamic, 16)
		for (uint32_t c = 0; c < nC; ++c) {
			const uint32_t L = db->clumpLen[c];
			uint32_t n = 0; int bad = 0;
			uint64_t tsum = 0;
			for (uint32_t zz = 0; zz < 16 && !bad; ++zz) {
				if (16ull * c + zz >= db->totR) break;
				lane_codes(db, coff[c], L, zz, lane);
				uint32_t ll = L; while (ll && lane[ll - 1] == 0) --ll;      /* lane's own length (pads are code 0) */
				if (ll < (uint32_t)K) continue;
				/* expansion budget as the reference estimates it: 3^a (N penalised) or 4^a per window (burst.c:3322-3353) */
				uint32_t asum = 0;
				for (uint32_t j = 0; j < ll && !skip_ambig; ++j) {
					if (j >= (uint32_t)K - 1) {
						tsum += powx[asum & 15];
						if (lane[j - (K - 1)] > 4 + z) --asum;
					}
					if (lane[j] > 4 + z) ++asum;
					if (tsum >= fullSize) { bad = 1; break; }
				}
				if (bad) break;
				for (uint32_t j = 0; j + K <= ll; ++j) {
					int skip = 0;
					if (skip_ambig) { for (int k = 0; k < K; ++k) if (lane[j + k] >= 5) { j += k; skip = 1; break; } }
					else if (z) for (int k = 0; k < K; ++k) if (lane[j + k] == 5) { j += k; skip = 1; break; }
					if (skip) continue;
					uint64_t need = 1; for (int k = 0; k < K; ++k) need *= AMB_N[lane[j + k]];
					if (n + need > capc) { while (n + need > capc) capc *= 2; cache = realloc(cache, (size_t)capc * 4); }
					expand_word(lane + j, K, 0, 0, seen, cache, &n);
				}
			}
			for (uint32_t i = 0; i < n; ++i) seen[cache[i] >> 3] = 0;
			if (bad) { isBad[c] = 1; continue; }
			words[c] = malloc((size_t)(n ? n : 1) * 4);
			if (!words[c]) { oom = 1; continue; }
			memcpy(words[c], cache, (size_t)n * 4); nwords[c] = n;
			for (uint32_t i = 0; i < n; ++i) {
				#pragma omp atomic
				++lens[cache[i]];
			}
		}
		free(seen); free(cache); free(lane);
	}
	if (oom) return bh_set_error(BH_E_OOM, "OOM:Accelerant.Refs");
	uint64_t tot = 0;
	uint64_t *offs = malloc((nw + 1) * 8);      /* running fill position of every word; ends as the END of its list */
	if (!offs) return bh_set_error(BH_E_OOM, "OOM:Accelerant.Refs");
	for (uint64_t w = 0; w < nw; ++w) { offs[w] = tot; tot += lens[w]; }
	offs[nw] = tot;
	uint32_t *ent = malloc((tot + 1) * 4);
	if (!ent) return bh_set_error(BH_E_OOM, "OOM:Accelerant.Refs");
	for (uint32_t c = 0; c < nC; ++c) { for (uint32_t i = 0; i < nwords[c]; ++i) ent[offs[words[c][i]]++] = c; free(words[c]); }
	free(words); free(nwords);
	uint32_t nb = 0;
	for (uint32_t c = 0; c < nC; ++c) nb += isBad[c];
	uint32_t *bl = own(db, malloc(((size_t)nb + 1) * 4));
	nb = 0; for (uint32_t c = 0; c < nC; ++c) if (isBad[c]) bl[nb++] = c;
	free(isBad); free(coff);
	/* pack (burst.c:3501-3528) */
	int fmt = nC > 1048574 ? 1 : 0;
	uint64_t bytes = 0;
	for (uint64_t w = 0; w < nw; ++w) bytes += fmt ? (uint64_t)lens[w] * 3 : (uint64_t)(lens[w] / 2u) * 5 + (lens[w] & 1) * 3;
	uint8_t *lists = own(db, malloc(bytes + 16)), *p = lists;
	if (!lists) return bh_set_error(BH_E_OOM, "OOM:WordDump");
	for (uint64_t w = 0; w < nw; ++w) {
		uint32_t n = lens[w];
		if (!n) continue;
		const uint32_t *l = ent + (offs[w] - n);
		if (fmt) for (uint32_t i = 0; i < n; ++i) { p[0] = (uint8_t)l[i]; p[1] = (uint8_t)(l[i] >> 8); p[2] = (uint8_t)(l[i] >> 16); p += 3; }
		else {
			uint32_t i = 0;
			for (; i + 1 < n; i += 2) { uint64_t v = (uint64_t)l[i] | ((uint64_t)l[i + 1] << 20); memcpy(p, &v, 5); p += 5; }
			if (i < n) { uint64_t v = l[i]; memcpy(p, &v, 3); p += 3; }
		}
	}
	memset(p, 0, 16);
	free(ent); free(offs);
	db->hasAcx = 1; db->K = K; db->acxFmt = fmt; db->acxZ = z ? 1 : 0;
	db->acxLens = lens; db->acxLists = lists; db->acxListBytes = bytes; db->badList = bl; db->badSz = nb;
	return BH_OK;
}

int bh_acx_from_device(BhDb *db, void *hh, int K, int z) {
	uint64_t tot = 0; uint32_t nb = 0;
	if (bhip_acx_export(hh, NULL, NULL, NULL, 0, &tot, NULL, 0, &nb)) return bh_set_error(BH_E_DEVICE, "libburst_hip: %s", bhip_last_error());
	const uint64_t nw = 1ull << (2 * K);
	uint32_t *lens = malloc(nw * 4), *ent = malloc((tot + 1) * 4), *bl = malloc(((size_t)nb + 1) * 4);
	if (!lens || !ent || !bl) { free(lens); free(ent); free(bl); return bh_set_error(BH_E_OOM, "OOM:Accelerant.Refs"); }
	if (bhip_acx_export(hh, lens, ent, NULL, tot, &tot, bl, nb, &nb)) { free(lens); free(ent); free(bl); return bh_set_error(BH_E_DEVICE, "libburst_hip: %s", bhip_last_error()); }
	/* pack (burst.c:3501-3528): byte position of every 65536th word first, then the blocks in parallel */
	const int fmt = db->numRclumps > 1048574 ? 1 : 0;
	const uint64_t BW = 65536, nblk = (nw + BW - 1) / BW;
	uint64_t *bbyte = malloc((nblk + 1) * 8), *bent = malloc((nblk + 1) * 8);
	if (!bbyte || !bent) { free(lens); free(ent); free(bl); free(bbyte); free(bent); return bh_set_error(BH_E_OOM, "OOM:WordDump"); }
	#pragma omp parallel for schedule(static)
	for (uint64_t b = 0; b < nblk; ++b) {
		uint64_t by = 0, en = 0;
		const uint64_t w1 = (b + 1) * BW < nw ? (b + 1) * BW : nw;
		for (uint64_t w = b * BW; w < w1; ++w) { en += lens[w]; by += fmt ? (uint64_t)lens[w] * 3 : (uint64_t)(lens[w] / 2u) * 5 + (lens[w] & 1) * 3; }
		bbyte[b + 1] = by; bent[b + 1] = en;
	}
	bbyte[0] = bent[0] = 0;
	for (uint64_t b = 0; b < nblk; ++b) { bbyte[b + 1] += bbyte[b]; bent[b + 1] += bent[b]; }
	const uint64_t bytes = bbyte[nblk];
	uint8_t *lists = malloc(bytes + 16);
	if (!lists) { free(lens); free(ent); free(bl); free(bbyte); free(bent); return bh_set_error(BH_E_OOM, "OOM:WordDump"); }
	#pragma omp parallel for schedule(dynamic, 16)
	for (uint64_t b = 0; b < nblk; ++b) {
		uint8_t *p = lists + bbyte[b];
		const uint32_t *l = ent + bent[b];
		const uint64_t w1 = (b + 1) * BW < nw ? (b + 1) * BW : nw;
		for (uint64_t w = b * BW; w < w1; ++w) {
			const uint32_t n = lens[w];
			if (fmt) for (uint32_t i = 0; i < n; ++i) { p[0] = (uint8_t)l[i]; p[1] = (uint8_t)(l[i] >> 8); p[2] = (uint8_t)(l[i] >> 16); p += 3; }
			else {
				uint32_t i = 0;
				for (; i + 1 < n; i += 2) { const uint64_t v = (uint64_t)l[i] | ((uint64_t)l[i + 1] << 20); memcpy(p, &v, 5); p += 5; }
				if (i < n) { const uint64_t v = l[i]; memcpy(p, &v, 3); p += 3; }
			}
			l += n;
		}
	}
	memset(lists + bytes, 0, 16);
	free(ent); free(bbyte); free(bent);
	if (!own(db, lens) || !own(db, lists) || !own(db, bl)) return bh_set_error(BH_E_OOM, "OOM:Accelerant.Refs");
	db->hasAcx = 1; db->K = K; db->acxFmt = fmt; db->acxZ = z ? 1 : 0;
	db->acxLens = lens; db->acxLists = lists; db->acxListBytes = bytes; db->badList = bl; db->badSz = nb;
	return BH_OK;
}

/* One packed run of the list area on its way to the file while the next one is fetched and packed: the writer thread of
 * bh_acx_write_from_device.  `len` of slot s is set by the producer (0 = nothing queued), cleared by the writer. */
typedef struct {
	FILE *o; pthread_mutex_t mu; pthread_cond_t cv;
	uint8_t *buf[2]; uint64_t len[2]; int stop, failed;
} AcxWriter;

static void *acx_writer_main(void *arg) {
	AcxWriter *w = arg;
	for (int s = 0;; s ^= 1) {
		pthread_mutex_lock(&w->mu);
		while (!w->len[s] && !w->stop) pthread_cond_wait(&w->cv, &w->mu);
		const uint64_t n = w->len[s];
		pthread_mutex_unlock(&w->mu);
		if (!n) break;                                  /* (stop, nothing queued) */
		const int bad = fwrite(w->buf[s], 1, n, w->o) != n;
		pthread_mutex_lock(&w->mu);
		w->len[s] = 0; if (bad) w->failed = 1;
		pthread_cond_broadcast(&w->cv);
		pthread_mutex_unlock(&w->mu);
	}
	return NULL;
}

/* The .acx of a database straight from the device that holds its accelerator, list area streamed: the length table comes over
 * whole (4 bytes per word), the entries in runs of whole words of about 2^27 entries, packed (burst.c:3501-3528) and written as they
 * come -- the host never holds more than two runs (bh_acx_from_device + bh_acx_write hold 4 + 3 bytes of EVERY entry: 230 GB for
 * the 33 G entries of a 19 GB database).  The file is written front to back and never sought, so `path` may be a FIFO: the
 * reference's read_accelerator (burst.c:3535-3594) is fopen + fgetc + four sequential fread calls, and a 167 GB accelerator then
 * never exists as a file (bench.py's cpu_baseline at the metric's size).  A run is written by a thread of its own while the next
 * one is fetched from the device and packed. */
int bh_acx_write_from_device(const BhDb *db, void *hh, int K, int z, const char *path) {
	uint64_t tot = 0; uint32_t nb = 0;
	if (bhip_acx_export(hh, NULL, NULL, NULL, 0, &tot, NULL, 0, &nb)) return bh_set_error(BH_E_DEVICE, "libburst_hip: %s", bhip_last_error());
	const uint64_t nw = 1ull << (2 * K);
	uint32_t *lens = malloc(nw * 4), *bl = malloc(((size_t)nb + 1) * 4);
	if (!lens || !bl) { free(lens); free(bl); return bh_set_error(BH_E_OOM, "OOM:Accelerant.Refs"); }
	if (bhip_acx_export(hh, lens, NULL, NULL, 0, &tot, bl, nb, &nb)) { free(lens); free(bl); return bh_set_error(BH_E_DEVICE, "libburst_hip: %s", bhip_last_error()); }
	const int fmt = db->numRclumps > 1048574 ? 1 : 0;
	FILE *o = fopen(path, "wb");                            /* (a FIFO: returns once the reader has opened its end) */
	if (!o) { free(lens); free(bl); return bh_set_error(BH_E_USAGE, "Cannot write accelerator '%s'", path); }
	struct stat st;
	if (!fstat(fileno(o), &st) && S_ISFIFO(st.st_mode))     /* the largest pipe buffer the kernel grants: fewer sleeps per gigabyte */
		for (int sz = 64 << 20; sz >= (1 << 20) && fcntl(fileno(o), F_SETPIPE_SZ, sz) < 0; sz >>= 1) ;
	setvbuf(o, NULL, _IOFBF, 8u << 20);
	const uint8_t vers = (uint8_t)(1 << 7 | (z ? 1 : 0) << 6 | fmt);
	int rc = BH_OK;
	if (fwrite(&vers, 1, 1, o) != 1 || fwrite(&nb, 4, 1, o) != 1 || fwrite(lens, 4, nw, o) != nw) rc = bh_set_error(BH_E_IO, "ERROR: write failed: %s", path);
	const uint64_t RUN = 1ull << 27;
	uint32_t *ent = malloc((RUN + (1u << 24)) * 4);
	AcxWriter w; memset(&w, 0, sizeof w); w.o = o;
	w.buf[0] = malloc((RUN + (1u << 24)) * 3 + 16); w.buf[1] = malloc((RUN + (1u << 24)) * 3 + 16);
	pthread_t th; int have_th = 0;
	if (!rc && !(ent && w.buf[0] && w.buf[1])) rc = bh_set_error(BH_E_OOM, "OOM:WordDump");
	if (!rc) {
		pthread_mutex_init(&w.mu, NULL); pthread_cond_init(&w.cv, NULL);
		if (pthread_create(&th, NULL, acx_writer_main, &w)) rc = bh_set_error(BH_E_INTERNAL, "no writer thread");
		else have_th = 1;
	}
	uint64_t e0 = 0;
	int slot = 0;
	for (uint64_t w0 = 0; w0 < nw && !rc;) {
		uint64_t w1 = w0, n = 0;
		while (w1 < nw && w1 - w0 < (1u << 24) && n < RUN) n += lens[w1++];      /* (a list has fewer than 2^24 entries: the buffers hold RUN + 2^24) */
		if (n && bhip_acx_export_entries(hh, e0, n, ent, NULL)) { rc = bh_set_error(BH_E_DEVICE, "libburst_hip: %s", bhip_last_error()); break; }
		/* byte position of every word of the run, then the words side by side */
		const uint64_t nwr = w1 - w0;
		uint64_t *pos = malloc((nwr + 1) * 16);
		if (!pos) { rc = bh_set_error(BH_E_OOM, "OOM:WordDump"); break; }
		uint64_t *epos = pos + nwr + 1;
		pos[0] = 0; epos[0] = 0;
		for (uint64_t k = 0; k < nwr; ++k) { const uint32_t l = lens[w0 + k]; pos[k + 1] = pos[k] + (fmt ? (uint64_t)l * 3 : (uint64_t)(l / 2u) * 5 + (l & 1u) * 3); epos[k + 1] = epos[k] + l; }
		pthread_mutex_lock(&w.mu);                          /* the slot's previous run has left */
		while (w.len[slot] && !w.failed) pthread_cond_wait(&w.cv, &w.mu);
		const int failed = w.failed;
		pthread_mutex_unlock(&w.mu);
		if (failed) { free(pos); break; }
		uint8_t *out = w.buf[slot];
		#pragma omp parallel for schedule(static)
		for (uint64_t k = 0; k < nwr; ++k) {
			uint8_t *p = out + pos[k];
			const uint32_t *l = ent + epos[k];
			const uint32_t m = lens[w0 + k];
			if (fmt) for (uint32_t i = 0; i < m; ++i) { p[0] = (uint8_t)l[i]; p[1] = (uint8_t)(l[i] >> 8); p[2] = (uint8_t)(l[i] >> 16); p += 3; }
			else {
				uint32_t i = 0;
				for (; i + 1 < m; i += 2) { const uint64_t v = (uint64_t)l[i] | ((uint64_t)l[i + 1] << 20); memcpy(p, &v, 5); p += 5; }
				if (i < m) { const uint64_t v = l[i]; memcpy(p, &v, 3); p += 3; }
			}
		}
		if (pos[nwr]) {
			pthread_mutex_lock(&w.mu);
			w.len[slot] = pos[nwr];
			pthread_cond_broadcast(&w.cv);
			pthread_mutex_unlock(&w.mu);
			slot ^= 1;
		}
		free(pos);
		e0 += n; w0 = w1;
	}
	if (have_th) {
		pthread_mutex_lock(&w.mu);
		while ((w.len[0] || w.len[1]) && !w.failed) pthread_cond_wait(&w.cv, &w.mu);
		w.stop = 1;
		pthread_cond_broadcast(&w.cv);
		pthread_mutex_unlock(&w.mu);
		pthread_join(th, NULL);
		pthread_mutex_destroy(&w.mu); pthread_cond_destroy(&w.cv);
		if (w.failed && !rc) rc = bh_set_error(BH_E_IO, "ERROR: write failed: %s", path);
	}
	if (!rc && e0 != tot) rc = bh_set_error(BH_E_INTERNAL, "accelerator lists: %lu entries written, %lu expected", (unsigned long)e0, (unsigned long)tot);
	if (!rc && fwrite(bl, 4, nb, o) != nb) rc = bh_set_error(BH_E_IO, "ERROR: write failed: %s", path);
	free(ent); free(w.buf[0]); free(w.buf[1]); free(lens); free(bl);
	if (fclose(o) && !rc) rc = bh_set_error(BH_E_IO, "ERROR: write failed: %s", path);
	return rc;
}

int bh_acx_write(const BhDb *db, const char *path) {
	if (!db->hasAcx) return bh_set_error(BH_E_USAGE, "no accelerator to write");
	FILE *o = fopen(path, "wb");
	if (!o) return bh_set_error(BH_E_USAGE, "Cannot write accelerator '%s'", path);
	uint8_t vers = (uint8_t)(1 << 7 | (db->acxZ ? 1 : 0) << 6 | db->acxFmt);
	fwrite(&vers, 1, 1, o); fwrite(&db->badSz, 4, 1, o);
	fwrite(db->acxLens, 4, 1ull << (2 * db->K), o);
	fwrite(db->acxLists, 1, db->acxListBytes, o);
	fwrite(db->badList, 4, db->badSz, o);
	if (fclose(o)) return bh_set_error(BH_E_IO, "ERROR: write failed: %s", path);
	return BH_OK;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Database sharding (multi-GPU mode for databases that do not fit one device, host/bh_multi.c): a view of the
 * clumps [c0, c1) of `db` -- clump area, lengths and reference count are pointers into `db` (which must outlive the
 * view), the accelerator is the sub-list of every word restricted to those clumps, renumbered from 0, in the .acx
 * packing (burst.c:3501-3528) its clump count asks for.  Reference index of a slice hit + 16*c0 = index in `db`. */
static inline uint32_t acx_entry(const uint8_t *base, int fmt, uint32_t i) {
	if (fmt) { const uint8_t *p = base + (size_t)i * 3; return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16; }
	const uint8_t *p = base + (size_t)(i >> 1) * 5;
	uint64_t v = 0; memcpy(&v, p, (i & 1) ? 5 : 3);
	return (i & 1) ? (uint32_t)(v >> 20) & 0xFFFFFu : (uint32_t)v & 0xFFFFFu;
}

int bh_db_slice(const BhDb *db, uint32_t c0, uint32_t c1, BhDb *out) {
	if (c1 > db->numRclumps) c1 = db->numRclumps;
	if (c0 >= c1) return bh_set_error(BH_E_USAGE, "empty database slice [%u, %u)", c0, c1);
	memset(out, 0, sizeof *out);
	uint64_t w0 = 0, wn = 0;
	for (uint32_t c = 0; c < c1; ++c) { const uint64_t w = db->clumpLen[c] / 2u + (db->clumpLen[c] & 1); if (c < c0) w0 += w; else wn += w; }
	out->rebase = db->rebase; out->xalpha = db->xalpha; out->shear = db->shear; out->maxLenR = db->maxLenR;
	out->numRclumps = c1 - c0;
	out->totR = (uint32_t)((16ull * c1 < db->totR ? 16ull * c1 : db->totR) - 16ull * c0);
	out->origTotR = out->totR;
	out->clumpLen = db->clumpLen + c0;
	out->packed = db->packed + w0 * 16;
	out->packedWords = wn;
	out->identityMap = 1;                 /* header tables stay with the full database */
	if (db->refIxSrt) out->refIxSrt = db->refIxSrt + 16ull * c0;      /* (a view, not owned: BEST's tie-break table for the slice's own device, bh_device_open_ex) */
	if (!db->hasAcx) return BH_OK;
	const int K = db->K, fmtIn = db->acxFmt, fmtOut = out->numRclumps > 1048574 ? 1 : 0;
	const uint64_t nw = 1ull << (2 * K);
	uint32_t *lens = own(out, calloc(nw, 4));
	uint64_t *inOff = malloc((nw + 1) * 8), *outOff = malloc((nw + 1) * 8);
	if (!lens || !inOff || !outOff) { free(inOff); free(outOff); bh_db_free(out); return bh_set_error(BH_E_OOM, "OOM:slice"); }
	uint64_t pos = 0;
	for (uint64_t w = 0; w < nw; ++w) { inOff[w] = pos; const uint32_t n = db->acxLens[w]; pos += fmtIn ? (uint64_t)n * 3 : (uint64_t)(n / 2u) * 5 + (n & 1) * 3; }
	/* a list is in the order its builder's threads filled it (ascending only for a single-threaded build): linear filter */
	#pragma omp parallel for schedule(dynamic, 65536)
	for (uint64_t w = 0; w < nw; ++w) {
		const uint32_t n = db->acxLens[w];
		const uint8_t *base = db->acxLists + inOff[w];
		uint32_t k = 0;
		for (uint32_t i = 0; i < n; ++i) { const uint32_t c = acx_entry(base, fmtIn, i); k += c >= c0 && c < c1; }
		lens[w] = k;
	}
	uint64_t bytes = 0;
	for (uint64_t w = 0; w < nw; ++w) { outOff[w] = bytes; const uint32_t n = lens[w]; bytes += fmtOut ? (uint64_t)n * 3 : (uint64_t)(n / 2u) * 5 + (n & 1) * 3; }
	uint8_t *lists = own(out, malloc(bytes + 16));
	if (!lists) { free(inOff); free(outOff); bh_db_free(out); return bh_set_error(BH_E_OOM, "OOM:slice"); }
	#pragma omp parallel for schedule(dynamic, 65536)
	for (uint64_t w = 0; w < nw; ++w) {
		if (!lens[w]) continue;
		const uint32_t n = db->acxLens[w];
		const uint8_t *base = db->acxLists + inOff[w];
		uint8_t *p = lists + outOff[w];
		uint32_t k = 0, held = 0;
		for (uint32_t i = 0; i < n; ++i) {
			const uint32_t c = acx_entry(base, fmtIn, i);
			if (c < c0 || c >= c1) continue;
			const uint32_t v = c - c0;
			if (fmtOut) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p += 3; }
			else if (k & 1) { const uint64_t pair = (uint64_t)held | (uint64_t)v << 20; memcpy(p, &pair, 5); p += 5; }
			else held = v;
			++k;
		}
		if (!fmtOut && (k & 1)) { const uint64_t last = held; memcpy(p, &last, 3); }
	}
	memset(lists + bytes, 0, 16);
	free(inOff); free(outOff);
	uint32_t nb = 0;
	for (uint32_t i = 0; i < db->badSz; ++i) nb += db->badList[i] >= c0 && db->badList[i] < c1;
	uint32_t *bl = own(out, malloc(((size_t)nb + 1) * 4));
	nb = 0;
	for (uint32_t i = 0; i < db->badSz; ++i) if (db->badList[i] >= c0 && db->badList[i] < c1) bl[nb++] = db->badList[i] - c0;
	out->hasAcx = 1; out->K = K; out->acxFmt = fmtOut; out->acxZ = db->acxZ;
	out->acxLens = lens; out->acxLists = lists; out->acxListBytes = bytes; out->badList = bl; out->badSz = nb;
	return BH_OK;
}
