/* bh_synth.c -- seeded synthetic references and reads for the tests and the bench (no datasets are reachable
 * offline).  Behaviour modelled on the reference's read simulator embalmlets/LLsim.c:175-231: uniform start,
 * fixed window, an exact number of edits per read at distinct positions with substitution : deletion : insertion
 * = 3 : 1 : 1, optional reverse complement of half the reads.  Not part of the alignment path.
 */
#include "burst_host.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/uio.h>
#include <unistd.h>

static double bh_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static inline uint64_t rng_next(uint64_t *s) {            /* xorshift64* */
	uint64_t x = *s; x ^= x >> 12; x ^= x << 25; x ^= x >> 27; *s = x;
	return x * 0x2545F4914F6CDD1DULL;
}
static inline double rng_unit(uint64_t *s) { return (rng_next(s) >> 11) * (1.0 / 9007199254740992.0); }
static const char BASES[4] = {'A', 'C', 'G', 'T'};

static void synth_family(char *w0, size_t *used, char *base, uint32_t b, uint32_t n_variants, uint32_t length, double rate, uint64_t seed) {
	uint64_t s = (seed * 0x9E3779B97F4A7C15ULL + 0x1234567ULL) ^ ((uint64_t)(b + 1) * 0xD6E8FEB86659FD93ULL);
	if (!s) s = 1;
	(void)rng_next(&s); (void)rng_next(&s);
	char *w = w0;
	for (uint32_t i = 0; i < length; ++i) base[i] = BASES[rng_next(&s) & 3];
	for (uint32_t v = 0; v < n_variants; ++v) {
		w += sprintf(w, ">ref_b%u_v%u\n", b, v);
		for (uint32_t i = 0; i < length; ++i) {
			if (v && rng_unit(&s) < rate) {
				uint32_t kk = (uint32_t)(rng_next(&s) % 5);
				if (kk < 3) { char c; do c = BASES[rng_next(&s) & 3]; while (c == base[i]); *w++ = c; }
				else if (kk == 3) { /* deletion */ }
				else { *w++ = BASES[rng_next(&s) & 3]; *w++ = base[i]; }
			} else *w++ = base[i];
		}
		*w++ = '\n';
	}
	*used = (size_t)(w - w0);
}

int bh_synth_refs(const char *fasta_out, uint32_t n_base, uint32_t n_variants, uint32_t length, double rate, uint64_t seed) {
	return bh_synth_refs_range(fasta_out, 0, n_base, n_variants, length, rate, seed);
}
/* base sequences [first_base, first_base + n_base) of the same family of sequences (a sequence depends on the seed and its number only):
 * a large set of references written part by part */
int bh_synth_refs_range(const char *fasta_out, uint32_t first_base, uint32_t n_base, uint32_t n_variants, uint32_t length, double rate, uint64_t seed) {
	FILE *o = fopen(fasta_out, "wb");
	if (!o) return bh_set_error(BH_E_IO, "cannot write %s", fasta_out);
	setvbuf(o, NULL, _IONBF, 0);
	/* every base sequence has a generator of its own (seeded by its number), so the file does not depend on the number of threads:
	 * blocks of families are made side by side, squeezed together and written in order -- one thread writes block k while the team
	 * makes block k + 1 (two buffers) */
	const size_t per_fam = (size_t)n_variants * (2 * (size_t)length + 48);
	/* families per block: 32 768 pairs of 1.4-kb sequences are 190 MB; families of hundreds of variants (the "strains" profile of the
	 * bench) keep the block near that size */
	uint32_t BLK = 32768;
	while (BLK > 64 && (size_t)BLK * per_fam > ((size_t)256 << 20)) BLK >>= 1;
	char *buf[2] = {malloc((size_t)BLK * per_fam), malloc((size_t)BLK * per_fam)};
	size_t *used = malloc((size_t)BLK * sizeof(*used)), *at = malloc(((size_t)BLK + 1) * sizeof(*at));      /* piece lengths of the two buffers */
	if (!buf[0] || !buf[1] || !used || !at) { free(buf[0]); free(buf[1]); free(used); free(at); fclose(o); return bh_set_error(BH_E_OOM, "OOM:synth"); }
	const int dbg = getenv("BURST_HOST_DEBUG") != NULL;
	double t_make = 0, t_all = bh_now();
	/* the pieces of a block (one per family, at a fixed stride) go to the file with writev: no copy to squeeze them together */
	size_t *used2[2] = {used, at};
	uint32_t pend_n = 0; int pend_buf = 0, io_err = 0;
	const int fd = fileno(o);
	for (uint32_t b0 = 0, it = 0; b0 < n_base || pend_n; b0 += BLK, ++it) {
		const uint32_t nb = b0 < n_base ? (n_base - b0 < BLK ? n_base - b0 : BLK) : 0;
		char *mine = buf[it & 1];
		size_t *mused = used2[it & 1];
		const double t0 = bh_now();
		#pragma omp parallel
		{
			char *base = malloc(length + 1);
			/* the block before goes to the file meanwhile (one thread; the others start on the loop at once) */
			#pragma omp single nowait
			{
				struct iovec iov[512];
				const char *pb = buf[pend_buf]; const size_t *pu = used2[pend_buf];
				for (uint32_t k = 0; k < pend_n && !io_err;) {
					int n = 0; size_t want = 0;
					for (; n < 512 && k + (uint32_t)n < pend_n; ++n) { iov[n].iov_base = (void *)(pb + (size_t)(k + (uint32_t)n) * per_fam); iov[n].iov_len = pu[k + (uint32_t)n]; want += pu[k + (uint32_t)n]; }
					ssize_t got = writev(fd, iov, n);
					if (got < 0) { io_err = 1; break; }
					if ((size_t)got < want) {      /* short write: finish this group piece by piece */
						size_t done = (size_t)got;
						for (int j = 0; j < n && !io_err; ++j) {
							if (done >= iov[j].iov_len) { done -= iov[j].iov_len; continue; }
							const char *p = (const char *)iov[j].iov_base + done; size_t left = iov[j].iov_len - done; done = 0;
							while (left) { ssize_t g = write(fd, p, left); if (g <= 0) { io_err = 1; break; } p += g; left -= (size_t)g; }
						}
					}
					k += (uint32_t)n;
				}
			}
			#pragma omp for schedule(dynamic, 64)
			for (uint32_t k = 0; k < nb; ++k) synth_family(mine + (size_t)k * per_fam, &mused[k], base, first_base + b0 + k, n_variants, length, rate, seed);
			free(base);
		}
		pend_n = nb; pend_buf = (int)(it & 1);
		t_make += bh_now() - t0;
		if (!nb) break;
	}
	free(buf[0]); free(buf[1]); free(used); free(at);
	if (dbg) fprintf(stderr, "[host] synthetic references: %.2f s (%.2f s in the generate + write-behind loop)\n", bh_now() - t_all, t_make);
	if (fclose(o) || io_err) return bh_set_error(BH_E_IO, "write failed: %s", fasta_out);
	return BH_OK;
}

static const char RCMAP[128] = {['A'] = 'T', ['C'] = 'G', ['G'] = 'C', ['T'] = 'A'};
/* IUPAC symbols whose base set contains the given base */
static const char *COMPAT[4] = {"MRWVHD", "MYSBVH", "KRSBVD", "KYWBHD"};

int bh_synth_reads(const char *refs_fasta, const char *fasta_out, uint64_t n_reads, uint32_t read_len, const uint32_t *edit_choices,
                   uint32_t n_choices, int rc, double iupac_rate, uint64_t seed) {
	return bh_synth_reads_ex(refs_fasta, fasta_out, n_reads, read_len, edit_choices, n_choices, rc, iupac_rate, seed, 0, 0);
}
/* first_read: number of the first read in the names (reads drawn part by part from the parts of a large set of references keep
 * unique names); append: add to the file instead of replacing it */
int bh_synth_reads_ex(const char *refs_fasta, const char *fasta_out, uint64_t n_reads, uint32_t read_len, const uint32_t *edit_choices,
                      uint32_t n_choices, int rc, double iupac_rate, uint64_t seed, uint64_t first_read, int append) {
	FILE *f = fopen(refs_fasta, "rb");
	if (!f) return bh_set_error(BH_E_IO, "Cannot open FASTA file: %s.", refs_fasta);
	fseeko(f, 0, SEEK_END); uint64_t sz = (uint64_t)ftello(f); rewind(f);
	char *dump = malloc(sz + 2);
	if (!dump || fread(dump, 1, sz, f) != sz) { fclose(f); free(dump); return bh_set_error(BH_E_IO, "cannot read %s", refs_fasta); }
	fclose(f); dump[sz] = '\n';
	/* single-line records only (what bh_synth_refs writes) */
	uint64_t cap = 1024, n = 0; char **seq = malloc(cap * sizeof(*seq)); uint32_t *len = malloc(cap * 4);
	for (char *p = dump, *end = dump + sz; p < end;) {
		char *nl = memchr(p, '\n', (size_t)(end + 1 - p));
		if (*p != '>' && nl > p) {
			if (n == cap) { cap *= 2; seq = realloc(seq, cap * sizeof(*seq)); len = realloc(len, cap * 4); }
			seq[n] = p; len[n] = (uint32_t)(nl - p); if (len[n] && p[len[n] - 1] == '\r') --len[n];
			++n;
		}
		p = nl + 1;
	}
	uint64_t nOk = 0;
	for (uint64_t i = 0; i < n; ++i) nOk += len[i] > read_len;
	if (!nOk) { free(dump); free(seq); free(len); return bh_set_error(BH_E_USAGE, "no reference longer than %u", read_len); }
	FILE *o = fopen(fasta_out, append ? "ab" : "wb");
	if (!o) { free(dump); free(seq); free(len); return bh_set_error(BH_E_IO, "cannot write %s", fasta_out); }
	setvbuf(o, NULL, _IOFBF, 1 << 22);
	uint64_t s = seed * 0xD1B54A32D192ED03ULL + 0x7654321ULL;
	char *rd = malloc(2 * (size_t)read_len + 16);
	uint32_t *pos = malloc(((size_t)read_len + 1) * 4);
	uint8_t *kind = calloc((size_t)read_len + 1, 1);
	for (uint64_t r = 0; r < n_reads; ++r) {
		uint64_t gi; do gi = rng_next(&s) % n; while (len[gi] <= read_len);
		const uint32_t st = (uint32_t)(rng_next(&s) % (len[gi] - read_len));
		const char *w = seq[gi] + st;
		uint32_t ne = n_choices ? edit_choices[rng_next(&s) % n_choices] : 0;
		if (ne > read_len) ne = read_len;
		/* ne distinct positions: partial Fisher-Yates */
		for (uint32_t i = 0; i < read_len; ++i) pos[i] = i;
		for (uint32_t i = 0; i < ne; ++i) { uint32_t j = i + (uint32_t)(rng_next(&s) % (read_len - i)); uint32_t t = pos[i]; pos[i] = pos[j]; pos[j] = t; }
		memset(kind, 0, read_len);
		for (uint32_t i = 0; i < ne; ++i) { uint32_t k = (uint32_t)(rng_next(&s) % 5); kind[pos[i]] = (uint8_t)(k < 3 ? 1 : (k == 3 ? 2 : 3)); }
		uint32_t m = 0;
		for (uint32_t i = 0; i < read_len; ++i) {
			char c = w[i];
			if (kind[i] == 1) { char x; do x = BASES[rng_next(&s) & 3]; while (x == c); rd[m++] = x; }
			else if (kind[i] == 2) { }
			else if (kind[i] == 3) { rd[m++] = BASES[rng_next(&s) & 3]; rd[m++] = c; }
			else rd[m++] = c;
		}
		if (iupac_rate > 0) for (uint32_t i = 0; i < m; ++i) if (rng_unit(&s) < iupac_rate) {
			const char *b = memchr(BASES, rd[i], 4);
			if (b) rd[i] = COMPAT[b - BASES][rng_next(&s) % 6];
		}
		int isrc = rc && (rng_next(&s) & 1);
		if (isrc) {
			for (uint32_t i = 0; i < m / 2; ++i) { char a = rd[i], b = rd[m - 1 - i]; rd[i] = b; rd[m - 1 - i] = a; }
			for (uint32_t i = 0; i < m; ++i) { char c = RCMAP[(int)rd[i] & 127]; if (c) rd[i] = c; else {
				/* complement of an IUPAC symbol */
				static const char *from = "KMRYSWBVHDN", *to = "MKYRSWVBDHN"; const char *q = strchr(from, rd[i]); if (q) rd[i] = to[q - from]; } }
		}
		fprintf(o, ">read%lu_g%lu_p%u_e%u%s\n", (unsigned long)(first_read + r), (unsigned long)gi, st, ne, isrc ? "_rc" : "");
		fwrite(rd, 1, m, o); fputc('\n', o);
	}
	free(rd); free(pos); free(kind); free(dump); free(seq); free(len);
	if (fclose(o)) return bh_set_error(BH_E_IO, "write failed: %s", fasta_out);
	return BH_OK;
}
