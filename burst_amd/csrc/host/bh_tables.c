/* bh_tables.c -- alphabet, cost table and error budget (burst.c:164-192, 1237-1329, 3069-3076). */
#include "burst_host.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static _Thread_local char g_bh_err[512];
const char *bh_last_error(void) { return g_bh_err; }
int bh_set_error(int code, const char *fmt, ...) {
	va_list ap; va_start(ap, fmt); vsnprintf(g_bh_err, sizeof g_bh_err, fmt, ap); va_end(ap);
	return code;
}

/* Base sets of the 16 codes ". A C G T N K M R Y S W B V H D" (A=1, C=2, G=4, T=8).  The reference's table
 * (SCORENVedN, burst.c:172-190) scores 0 exactly when one set contains the other; code 0 is 255 against
 * everything; setScore() (burst.c:1256-1285, 1313-1328) overwrites row and column N with Z when Z != 0. */
static const uint8_t BASESET[16] = {0, 1, 2, 4, 8, 15, 12, 3, 5, 10, 6, 9, 14, 7, 11, 13};

/* -x / --xalphabet ("any alphabet, unambiguous ID matching", burst.c:86, 4945-4948): the reference then skips the translation to its
 * 16 codes (burst.c:1845-1847, 3005-3010) and scores raw bytes by equality (DIAGSC_XALPHA, burst.c:696-697: cost 0 iff equal, 1
 * otherwise -- also against the padding byte 0 of a shorter lane, which costs 255 with the nucleotide table).  The device layout has four
 * bits per symbol, so an alphabet of up to 15 distinct bytes is MAPPED onto the codes 1..15 in ascending byte order (the sort of the
 * queries and of the reference fragments keeps its order) and scored by the identity table; more symbols are refused.  Process-wide,
 * set by the command line before anything is parsed. */
static int g_xalpha;
static uint8_t g_xmap[256];
void bh_set_alphabet(const uint8_t map[256]) {
	g_xalpha = map != NULL;
	if (map) memcpy(g_xmap, map, 256);
}
int bh_alphabet_is_set(void) { return g_xalpha; }
/* the alphabet of a run: every byte of the sequence lines of the two FASTA files (headers, line ends apart), codes 1..n in ascending
 * byte order */
int bh_alphabet_from_files(const char *ref_fa, const char *query_fa, uint8_t map[256], int *n_symbols) {
	uint8_t seen[256];
	memset(seen, 0, sizeof seen); memset(map, 0, 256);
	const char *files[2] = {ref_fa, query_fa};
	for (int f = 0; f < 2; ++f) {
		FILE *in = fopen(files[f], "rb");
		if (!in) return bh_set_error(BH_E_IO, "ERROR: cannot open %s", files[f]);
		int c, bol = 1, head = 0;
		while ((c = fgetc(in)) != EOF) {
			if (bol) head = c == '>';
			bol = c == '\n';
			if (!head && c != '\n' && c != '\r') seen[c] = 1;
		}
		fclose(in);
	}
	int n = 0;
	for (int c = 0; c < 256; ++c) if (seen[c] && ++n <= 15) map[c] = (uint8_t)n;
	if (n_symbols) *n_symbols = n;
	if (n > 15) return bh_set_error(BH_E_USAGE, "ERROR: -x: %d distinct symbols in the references and queries; the device layout holds 15 (four bits per symbol)", n);
	return BH_OK;
}

void bh_score_lut(int z, uint8_t lut[256]) {
	if (g_xalpha) { for (int q = 0; q < 16; ++q) for (int r = 0; r < 16; ++r) lut[16 * q + r] = q == r ? 0 : 1; return; }
	for (int q = 0; q < 16; ++q) for (int r = 0; r < 16; ++r) {
		uint8_t v = 255;
		if (q && r) {
			uint8_t both = BASESET[q] & BASESET[r];
			v = (both == BASESET[q] || both == BASESET[r]) ? 0 : 1;
			if (z && (q == 5 || r == 5)) v = (uint8_t)z;
		}
		lut[16 * q + r] = v;
	}
}

void bh_char2code(uint8_t map[256]) {
	if (g_xalpha) { memcpy(map, g_xmap, 256); return; }
	memset(map, 0, 256);
	for (int c = 'A'; c <= 'Z'; ++c) map[c] = 5;
	for (int c = 'a'; c < 'z'; ++c) map[c] = 5;      /* the reference's loop stops before 'z' (burst.c:1291) */
	const char *sym = "ACGTKMRYSWBVHD";
	const uint8_t code[] = {1, 2, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
	for (int i = 0; sym[i]; ++i) map[(int)sym[i]] = map[(int)sym[i] + 32] = code[i];
	map['U'] = map['u'] = 4;
}

uint8_t bh_rc_code(uint8_t c) {
	static const uint8_t rv[16] = {0, 4, 3, 2, 1, 5, 7, 6, 9, 8, 10, 11, 13, 12, 15, 14};
	return rv[c & 15];
}

uint32_t bh_error_budget(float thres, uint32_t len) {
	float reqID = 1 / thres - 1;
	uint32_t ed = reqID * len;
	return ed < 254 ? ed : 254;
}
