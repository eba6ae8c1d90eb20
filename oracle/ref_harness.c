/* oracle/ref_harness.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin C-ABI shim around the *real* reference kernels.  The reference translation unit
 * is pulled in by path at build time (`#include "/root/reference/burst.c"`), so no
 * reference source is copied into this repository; the resulting shared object lands in
 * oracle/_ref/ (git-ignored).  It exists to (a) pin oracle/burst_oracle.c against the
 * reference's own aded_mat16 / aded_mat16L / reScoreM_mat16 (burst.c:1097, 1106, 890) and
 * (b) generate the golden kernel vectors under tests/golden/ (tests/golden/make_golden.py).
 *
 * Buffer conventions follow the reference call sites burst.c:4052-4065, 4139-4151, 4215-4227.
 */
#define main burst_reference_main
#include "/root/reference/burst.c"
#undef main

/* force out-of-line definitions of the gnu11 `inline` kernels */
extern inline uint32_t aded_mat16(DualCoil *ref, char *query, uint32_t rwidth, uint32_t qlen, uint32_t width,
	DualCoil *Matrix, DualCoil *profile, uint32_t maxED, uint32_t startQ, uint32_t *LoBound, uint32_t *HiBound, DualCoil *MinA);
extern inline uint32_t aded_mat16L(DualCoil *ref, char *query, uint32_t rwidth, uint32_t qlen, uint32_t width, uint32_t minlen,
	DualCoil *Matrix, DualCoil *profile, uint32_t maxED, uint32_t startQ, uint32_t *LoBound, uint32_t *HiBound, DualCoil *MinA);

static int harness_cache0 = 150;

/* Z = 1 (default, N penalised) or 0 (-y).  burst.c:1237 */
void ref_setscore(int z) { Z = (char)z; setScore(); }

/* copy of the effective 16x16 cost table as the SSSE3 path sees it (SCOREFAST, burst.c:1310-1328) */
void ref_get_scorefast(uint8_t out[256]) {
	for (int q = 0; q < 16; ++q) {
		DualCoil d; d.v = SCOREFAST[q];
		for (int r = 0; r < 16; ++r) out[q*16 + r] = d.u8[r];
	}
}

/* ASCII -> code, burst.c:1207 translateNV */
void ref_translate(char *s, size_t len) { translateNV(s, len); }

/* One (query, clump) unit of work exactly as burst.c:4139-4227 drives it (fresh state, startQ = 1).
 * clump: clumpLen rows of 16 bytes (byte k = lane k, codes 0..15), i.e. the unpacked DualCoil array.
 * variant: 0 = aded_mat16 (exhaustive path 4425), 1 = aded_mat16L (accelerated path 4215, needs minlen).
 * Outputs: mins[16], return value = min over lanes (UINT32_MAX when truncated).
 * If do_rescore != 0 and ret <= maxED, reScoreM_mat16 runs with bound = (bound_override >= 0 ?
 * bound_override : ret) and fills score/finalPos/gapR/gapQ. */
uint32_t ref_align_clump(const uint8_t *clump, uint32_t clumpLen, const char *qcodes, uint32_t qlen,
		uint32_t maxED, int variant, uint32_t minlen, uint8_t mins[16],
		int do_rescore, int bound_override, float score[16], uint32_t finalPos[16],
		uint8_t gapR[16], uint8_t gapQ[16]) {
	cacheSz = harness_cache0 < (int)qlen + 2 ? harness_cache0 : (int)qlen + 2;   /* burst.c:3193 */
	uint32_t rdim = clumpLen + 2, qdim = qlen + 2;
	void *m0, *s0, *s1, *s2, *r0;
	DualCoil *Matrices = calloc_a(16, (size_t)(cacheSz + 2) * rdim * sizeof(DualCoil), &m0),
		*ScoresEX = calloc_a(16, 2 * (size_t)rdim * sizeof(DualCoil), &s0),
		*ShiftsEX = calloc_a(16, 2 * (size_t)rdim * sizeof(DualCoil), &s1),
		*ShiftsBX = calloc_a(16, 2 * (size_t)rdim * sizeof(DualCoil), &s2),
		*rclump = calloc_a(16, (size_t)(2 + rdim) * sizeof(DualCoil), &r0);
	uint32_t *HiBound = calloc(qdim + 2, sizeof(*HiBound)), *LoBound = calloc(qdim + 2, sizeof(*LoBound));
	for (int j = 0; j < cacheSz + 2; ++j) Matrices[(size_t)j * rdim].v = _mm_set1_epi8(MIN(j * GAP, 255));
	*LoBound = -1, LoBound[1] = 1;
	memcpy(rclump, clump, (size_t)clumpLen * 16);
	uint32_t rlen = clumpLen + 1;
	HiBound[1] = rlen;
	char *q = malloc(qlen + 1); memcpy(q, qcodes, qlen); q[qlen] = 0;
	DualCoil mn; mn.v = _mm_set1_epi8(-1);
	uint32_t ret;
	if (variant == 1) ret = aded_mat16L(rclump, q, rlen, qlen, rdim, minlen, Matrices, 0, maxED, 1, LoBound, HiBound, &mn);
	else ret = aded_mat16(rclump, q, rlen, qlen, rdim, Matrices, 0, maxED, 1, LoBound, HiBound, &mn);
	if (ret == (uint32_t)-1) memset(mins, 255, 16); else memcpy(mins, mn.u8, 16);
	if (do_rescore && ret <= maxED) {
		MetaPack MPK;
		uint32_t bound = bound_override >= 0 ? (uint32_t)bound_override : ret;
		reScoreM_mat16(rclump, q, rlen, qlen, rdim, ScoresEX, ShiftsEX, ShiftsBX, bound, 0, &MPK);
		for (int z = 0; z < 16; ++z)
			score[z] = MPK.score[z], finalPos[z] = MPK.finalPos[z], gapR[z] = MPK.numGapR[z], gapQ[z] = MPK.numGapQ[z];
	}
	free(m0); free(s0); free(s1); free(s2); free(r0); free(HiBound); free(LoBound); free(q);
	return ret;
}
