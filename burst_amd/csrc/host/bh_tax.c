/* bh_tax.c -- taxonomy map for column 13 of the .b6: tab-separated "<reference header>\t<taxonomy>"
 * lines (parse_taxonomy, burst.c:447-479), sorted by header (burst.c:5146-5148), looked up by the header of the reference
 * a line reports (taxa_lookup_generic / taxa_lookup_ncbi, burst.c:409-440).  Everything here is host-side bookkeeping
 * downstream of the device path; it exists so that CAPITALIST can be run the way users run it (README.md of the
 * reference, "-m CAPITALIST -b taxonomy.txt"). */
#include "burst_host.h"
#include <stdlib.h>
#include <string.h>

static const char BH_NULLTAX[1] = {0};

static int by_head(const void *a, const void *b) {
	const char *const *x = a, *const *y = b;
	return strcmp(x[0], y[0]);                       /* pairs are stored as {head, tax}: compare heads */
}

void bh_tax_free(BhTax *T) {
	if (!T) return;
	free(T->pair); free(T->blob);
	memset(T, 0, sizeof *T);
}

int bh_tax_load(const char *file, BhTax *T) {
	memset(T, 0, sizeof *T);
	FILE *f = fopen(file, "rb");
	if (!f) return bh_set_error(BH_E_IO, "Cannot open TAXONOMY file: %s.", file);
	fseeko(f, 0, SEEK_END);
	const uint64_t sz = (uint64_t)ftello(f);
	rewind(f);
	char *blob = malloc(sz + 2);
	if (!blob) { fclose(f); return bh_set_error(BH_E_OOM, "OOM:taxonomy"); }
	if (fread(blob, 1, sz, f) != sz) { fclose(f); free(blob); return bh_set_error(BH_E_IO, "short read on %s", file); }
	fclose(f);
	blob[sz] = '\n'; blob[sz + 1] = 0;
	uint64_t nl = 0;
	for (uint64_t i = 0; i <= sz; ++i) nl += blob[i] == '\n';
	char **pair = malloc((nl + 1) * 2 * sizeof(*pair));
	if (!pair) { free(blob); return bh_set_error(BH_E_OOM, "OOM:taxonomy"); }
	uint64_t n = 0;
	for (uint64_t i = 0; i < sz;) {                  /* one record per line; the last line may lack its newline */
		char *line = blob + i, *e = memchr(line, '\n', sz + 1 - i);
		const uint64_t next = (uint64_t)(e - blob) + 1;
		char *tab = memchr(line, '\t', (size_t)(e - line));
		if (!tab) { free(pair); free(blob); return bh_set_error(BH_E_IO, "ERROR: invalid taxonomy [%lu]", (unsigned long)n); }   /* burst.c:462, exit(2) */
		*tab = 0;
		char *tx = tab + 1, *te = tx;
		while (te < e && *te != '\r' && *te != '\t') ++te;   /* the taxonomy ends at newline, CR or a further tab (burst.c:467) */
		*te = 0; *e = 0;
		pair[2 * n] = line; pair[2 * n + 1] = tx;
		++n;
		i = next;
	}
	if (!n) { free(pair); free(blob); return bh_set_error(BH_E_USAGE, "ERROR: invalid taxonomy"); }
	qsort(pair, n, 2 * sizeof(*pair), by_head);
	T->n = n; T->pair = pair; T->blob = blob;
	return BH_OK;
}

/* Binary search with the reference's own probe sequence (it decides which of several equal headers answers and what an
 * absent header returns): the table is searched as p[0..sz] with sz = n - 1, probing p[w + 1], w = sz / 2. */
static int head_cmp(const char *ref, const char *key, int ncbi, int *match) {
	/* walks both strings while equal; *match = 1 when ref ended while still equal (ncbi: also when the key continues with a
	 * version suffix ".N" where ref ends); otherwise returns the sign of (ref char - key char) at the first difference */
	const char *r = ref, *k = key;
	for (;; ++r, ++k) {
		if (*r != *k) break;
		if (!*r) { *match = 1; return 0; }
	}
	if (ncbi && *k == '.' && !*r) { *match = 1; return 0; }
	*match = 0;
	const char kc = (ncbi && *k == '.') ? 0 : *k;
	return *r < kc ? -1 : 1;
}

const char *bh_tax_lookup(const BhTax *T, const char *key, int ncbi) {
	if (!T || !T->n) return BH_NULLTAX;
	if (ncbi) {                                      /* '>xxx|accession...': the key starts after the first four characters */
		size_t l = strnlen(key, 4);
		key += l < 4 ? l : 4;
	}
	char **p = T->pair;
	uint64_t sz = T->n - 1;
	int m;
	while (sz) {
		const uint64_t w = sz >> 1;
		const int c = head_cmp(p[2 * (w + 1)], key, ncbi, &m);
		if (m) return p[2 * (w + 1) + 1];
		if (c < 0) { p += 2 * (w + 1); sz -= w + 1; } else sz = w;
	}
	(void)head_cmp(p[0], key, ncbi, &m);
	return m ? p[1] : BH_NULLTAX;
}
