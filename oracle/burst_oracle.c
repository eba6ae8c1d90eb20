/* oracle/burst_oracle.c -- TEST INFRASTRUCTURE ONLY.  See burst_oracle.h for scope and pinning.
 * Plain C restatement of the reference hot path; every function cites the reference lines it follows
 * (/root/reference/burst.c @ 2024_08_07).  Deliberately simple: full-width dynamic programming, no
 * pruning, no bit tricks -- it is the checker, not the thing measured.
 */
#include "burst_oracle.h"
#include <stdlib.h>
#include <string.h>

static inline uint8_t sat8(unsigned v) { return v > 255u ? 255u : (uint8_t)v; }

/* IUPAC base sets (A=1,C=2,G=4,T=8) in the reference's code order . A C G T N K M R Y S W B V H D
 * (burst.c:172-174).  Two codes cost 0 iff one set contains the other; code 0 costs 255 against
 * everything; with z != 0 the N row and column are forced to z (burst.c:1256-1285). */
static const uint8_t ORC_SETS[16] = {0, 1, 2, 4, 8, 15, 12, 3, 5, 10, 6, 9, 14, 7, 11, 13};

void orc_score_lut(int z, uint8_t lut[256]) {
	for (int a = 0; a < 16; ++a) for (int b = 0; b < 16; ++b) {
		uint8_t c;
		if (!a || !b) c = 255;
		else {
			uint8_t sa = ORC_SETS[a], sb = ORC_SETS[b];
			c = ((sa & sb) == sa || (sa & sb) == sb) ? 0 : 1;
			if (z && (a == 5 || b == 5)) c = (uint8_t)z;
		}
		lut[16 * a + b] = c;
	}
}

void orc_char2code(uint8_t map[128]) {   /* burst.c:1288-1307; note the loop bound quirk: 'z' -> 0 */
	for (int i = 0; i < 128; ++i) map[i] = 0;
	for (int i = 65; i < 91; ++i) map[i] = 5;
	for (int i = 97; i < 122; ++i) map[i] = 5;
	static const char *L = "ACGTKMRYSWBVHD";
	static const uint8_t C[] = {1, 2, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
	for (int i = 0; L[i]; ++i) map[(int)L[i]] = map[(int)L[i] + 32] = C[i];
	map['U'] = map['u'] = 4;
}

uint8_t orc_rc_code(uint8_t c) {         /* burst.c:168 */
	static const uint8_t RVT[16] = {0, 4, 3, 2, 1, 5, 7, 6, 9, 8, 10, 11, 13, 12, 15, 14};
	return RVT[c & 15];
}

uint32_t orc_error_budget(float thres, uint32_t len) {   /* burst.c:3069, 3074-3076 */
	float reqID = 1 / thres - 1;
	uint32_t ed = reqID * len;
	return ed < 254 ? ed : 254;
}

void orc_unpack_clump(const uint8_t *packed, uint32_t clumpLen, uint8_t *rows) {   /* burst.c:4141-4150 */
	for (uint32_t w = 0; w < clumpLen; w += 2) {
		const uint8_t *src = packed + (size_t)(w / 2) * 16;
		for (int k = 0; k < 16; ++k) {
			rows[(size_t)w * 16 + k] = src[k] & 15;
			if (w + 1 < clumpLen) rows[(size_t)(w + 1) * 16 + k] = src[k] >> 4;
		}
	}
}

/* burst.c:1003-1095 / 1106-1204: recurrence 1014-1027 (= 1148-1159), column 0 at 1013, row 0 zeros (4052),
 * final min over the last row 1078-1094.  The reference only evaluates a band and forces cells > maxED to
 * 255, which cannot change any value <= maxED; here every cell is evaluated. */
uint32_t orc_aded_clump(const uint8_t *rows, uint32_t n, const uint8_t *q, uint32_t m,
                        uint32_t maxED, const uint8_t lut[256], uint8_t mins[16]) {
	uint8_t *prev = calloc((size_t)(n + 1) * 16, 1), *cur = malloc((size_t)(n + 1) * 16);
	for (uint32_t y = 1; y <= m; ++y) {
		const uint8_t *L = lut + 16 * q[y - 1];
		memset(cur, sat8(y), 16);
		for (uint32_t x = 1; x <= n; ++x) {
			const uint8_t *rc = rows + (size_t)(x - 1) * 16;
			const uint8_t *pd = prev + (size_t)(x - 1) * 16, *pu = prev + (size_t)x * 16,
				*cl = cur + (size_t)(x - 1) * 16;
			uint8_t *c = cur + (size_t)x * 16;
			for (int z = 0; z < 16; ++z) {
				unsigned d = sat8(pd[z] + L[rc[z]]), u = sat8(pu[z] + 1u), l = sat8(cl[z] + 1u);
				unsigned s = d < u ? d : u;
				c[z] = (uint8_t)(s < l ? s : l);
			}
		}
		uint8_t *t = prev; prev = cur; cur = t;
	}
	uint32_t best = UINT32_MAX;
	for (int z = 0; z < 16; ++z) {
		unsigned mn = 255;
		for (uint32_t x = 1; x <= n; ++x) if (prev[(size_t)x * 16 + z] < mn) mn = prev[(size_t)x * 16 + z];
		if (mn > maxED) mn = 255;
		mins[z] = (uint8_t)mn;
		if (mn <= maxED && mn < best) best = mn;
	}
	free(prev); free(cur);
	return best;
}

/* burst.c:713-886.  Planes: D score, H = "shift" (left moves, numGapQ), V = "shiftR" (up moves, numGapR). */
int orc_rescore_lane(const uint8_t *q, uint32_t m, const uint8_t *r, uint32_t n, uint32_t B,
                     const uint8_t lut[256], OrcHit *out) {
	size_t W = (size_t)n + 1;
	uint8_t *buf = calloc(6 * W, 1);
	uint8_t *pD = buf, *pH = buf + W, *pV = buf + 2 * W, *cD = buf + 3 * W, *cH = buf + 4 * W, *cV = buf + 5 * W;
	/* row 1: burst.c:722-739 (score = raw cost, shift = 1 iff cost==1 and left cell==0, shiftR = 0) */
	cD[0] = 1; cH[0] = 0; cV[0] = 1;
	for (uint32_t x = 1; x <= n; ++x) {
		uint8_t s = lut[16 * q[0] + r[x - 1]];
		cD[x] = s; cH[x] = (s == 1 && cD[x - 1] == 0) ? 1 : 0; cV[x] = 0;
	}
	for (uint32_t y = 2; y <= m; ++y) {   /* burst.c:740-821 */
		uint8_t *t;
		t = pD; pD = cD; cD = t; t = pH; pH = cH; cH = t; t = pV; pV = cV; cV = t;
		cD[0] = sat8(y); cH[0] = 0; cV[0] = sat8(y);          /* 747-750 */
		const uint8_t *L = lut + 16 * q[y - 1];
		for (uint32_t x = 1; x <= n; ++x) {
			unsigned sD = sat8(pD[x - 1] + L[r[x - 1]]), hD = pH[x - 1], vD = pV[x - 1];   /* 763-767 */
			unsigned sU = sat8(pD[x] + 1u), hU = pH[x], vU = sat8(pV[x] + 1u);            /* 768-770 */
			unsigned s = sD < sU ? sD : sU, h, v;
			/* 771-779: keep D iff D is the min and not (tie with U while shiftU > shift) */
			int keepD = (sD == s) && !((sU == sD) && (hU > hD));
			if (keepD) h = hD, v = vD; else h = hU, v = vU;
			unsigned sL = sat8(cD[x - 1] + 1u), hL = sat8(cH[x - 1] + 1u), vL = cV[x - 1];  /* 783-788 */
			unsigned s2 = s < sL ? s : sL;
			int keep = (s == s2) && !((sL == s) && (hL > h));                             /* 789-795 */
			if (!keep) h = hL, v = vL;
			s = s2;
			if (s >= B + 1) s = 255;                                                      /* 802-803 */
			cD[x] = (uint8_t)s; cH[x] = (uint8_t)h; cV[x] = (uint8_t)v;
		}
	}
	/* final selection 824-842 and finalPos 862-879 */
	unsigned bs = 255, bh = 0, bv = 0;
	for (uint32_t x = 1; x <= n; ++x) {
		unsigned s = cD[x], h = cH[x];
		if (s < bs || (s == bs && h > bh)) bs = s, bh = h, bv = cV[x];
	}
	uint32_t fin = (uint32_t)-1;
	for (uint32_t x = 1; x <= n; ++x) if (cD[x] == bs && cH[x] == bh) fin = x;
	int hit = bs <= B;
	if (hit && out) {
		out->ed = (uint8_t)bs; out->gapQ = (uint8_t)bh; out->gapR = (uint8_t)bv; out->finalPos = fin;
		out->score = 1.0f - (float)bs / ((float)m + (float)bh);                          /* 844-847 */
	}
	free(buf);
	return hit;
}

static int hit_cmp(const void *a, const void *b) {
	const OrcHit *A = a, *B = b;
	if (A->q != B->q) return A->q < B->q ? -1 : 1;
	if (A->refIx != B->refIx) return A->refIx < B->refIx ? -1 : 1;
	return 0;
}

uint64_t orc_search(const uint8_t *packed, const uint32_t *clumpLen, uint32_t nClumps, uint32_t totR,
                    const uint8_t *qcodes, const uint64_t *qoff, const uint32_t *qE,
                    const uint32_t *qsix, const uint8_t *qrc, uint32_t nq, uint32_t nShared,
                    const uint8_t lut[256], int all_hits, OrcHit *hits, uint64_t cap) {
	uint32_t maxLen = 0;
	uint64_t *coff = malloc(((size_t)nClumps + 1) * sizeof(*coff));
	coff[0] = 0;
	for (uint32_t c = 0; c < nClumps; ++c) {
		coff[c + 1] = coff[c] + clumpLen[c] / 2u + (clumpLen[c] & 1);
		if (clumpLen[c] > maxLen) maxLen = clumpLen[c];
	}
	/* pass 1: every (query entry, clump) -> mins[16] */
	uint8_t *allmins = malloc((size_t)nq * nClumps * 16);
	#pragma omp parallel
	{
		uint8_t *rows = malloc((size_t)(maxLen + 2) * 16);
		#pragma omp for schedule(dynamic, 1)
		for (uint32_t c = 0; c < nClumps; ++c) {
			orc_unpack_clump(packed + coff[c] * 16, clumpLen[c], rows);
			for (uint32_t j = 0; j < nq; ++j)
				orc_aded_clump(rows, clumpLen[c], qcodes + qoff[j], (uint32_t)(qoff[j + 1] - qoff[j]),
				               qE[j], lut, allmins + ((size_t)j * nClumps + c) * 16);
		}
		free(rows);
	}
	/* shared minimum per unique query: the net effect of the Sb->ed tightening (burst.c:4217-4223, 4429-4436) */
	uint32_t *best = malloc((size_t)nShared * sizeof(*best));
	for (uint32_t s = 0; s < nShared; ++s) best[s] = UINT32_MAX;
	for (uint32_t j = 0; j < nq; ++j) for (uint32_t c = 0; c < nClumps; ++c) for (int z = 0; z < 16; ++z) {
		uint8_t e = allmins[((size_t)j * nClumps + c) * 16 + z];
		if (c * 16u + z >= totR) continue;                                   /* burst.c:4229 */
		if (e <= qE[j] && e < best[qsix[j]]) best[qsix[j]] = e;
	}
	/* pass 2: rescoring the kept lanes (burst.c:4224-4237) */
	uint64_t nh = 0;
	uint8_t *rows = malloc((size_t)(maxLen + 2) * 16), *lane = malloc(maxLen + 2);
	for (uint32_t j = 0; j < nq; ++j) for (uint32_t c = 0; c < nClumps; ++c) {
		const uint8_t *mn = allmins + ((size_t)j * nClumps + c) * 16;
		int any = 0;
		for (int z = 0; z < 16; ++z) {
			if (c * 16u + z >= totR || mn[z] > qE[j]) continue;
			if (all_hits || mn[z] == best[qsix[j]]) any = 1;
		}
		if (!any) continue;
		orc_unpack_clump(packed + coff[c] * 16, clumpLen[c], rows);
		for (int z = 0; z < 16; ++z) {
			if (c * 16u + z >= totR || mn[z] > qE[j]) continue;
			if (!all_hits && mn[z] != best[qsix[j]]) continue;
			for (uint32_t x = 0; x < clumpLen[c]; ++x) lane[x] = rows[(size_t)x * 16 + z];
			OrcHit h; memset(&h, 0, sizeof h);
			uint32_t bound = all_hits ? qE[j] : mn[z];                       /* burst.c:4224 */
			if (!orc_rescore_lane(qcodes + qoff[j], (uint32_t)(qoff[j + 1] - qoff[j]), lane, clumpLen[c], bound, lut, &h))
				continue;   /* unreachable for valid input (burst.c:812-816 would abort the reference) */
			h.q = j; h.refIx = c * 16u + z; h.rc = qrc ? qrc[j] : 0;
			h.ed = mn[z];                                                    /* tmp->mismatches = mins.u8[z], burst.c:4232 */
			if (nh < cap) hits[nh] = h;
			++nh;
		}
	}
	qsort(hits, nh < cap ? nh : cap, sizeof(*hits), hit_cmp);
	free(rows); free(lane); free(best); free(allmins); free(coff);
	return nh;
}

uint32_t orc_prefilter_counts(const uint8_t *q, uint32_t m, uint32_t E, int K, uint32_t stride,
                              const uint64_t *offs, const uint32_t *entries,
                              uint32_t nClumps, uint16_t *counts) {
	memset(counts, 0, (size_t)nClumps * sizeof(*counts));
	if (m < (uint32_t)K) return 0;
	if (!stride) stride = 1;
	uint32_t mask = K == 16 ? 0xFFFFFFFFu : ((1u << (2 * K)) - 1);
	for (uint32_t p = 0; p + K <= m; p += stride) {        /* burst.c:4097-4104: w = w<<2 | (code-1), first base most significant */
		uint32_t w = 0; int ok = 1;
		for (int k = 0; k < K; ++k) { uint32_t c = q[p + k]; if (c < 1 || c > 4) ok = 0; w = (w << 2) | ((c - 1) & 3); }
		if (!ok) continue;
		uint32_t t = w & mask;
		for (uint64_t e = offs[t]; e < offs[t + 1]; ++e)    /* burst.c:3245-3249: one count per (position, clump) */
			if (counts[entries[e]] < UINT16_MAX) ++counts[entries[e]];
	}
	/* stride 1: need = len-K+1-E*K = mmatch+1 (burst.c:4091-4092).  stride s: one edit destroys at most ceil(K/s) sampled words */
	int need = (int)((m - K) / stride + 1) - (int)(E * ((K + stride - 1) / stride));
	uint32_t thr = need > 0 ? (uint32_t)need - 1 : 0;
	uint32_t n = 0;
	for (uint32_t c = 0; c < nClumps; ++c) n += counts[c] > thr;
	return n;
}
