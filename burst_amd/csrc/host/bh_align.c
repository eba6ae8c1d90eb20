/* bh_align.c -- GPU batch scheduler: the replacement of the two OpenMP loops of do_alignments
 * (burst.c:4050-4289 accelerated, 4326-4488 exhaustive).  Unique queries are cut into contiguous batches in
 * sorted order; a forward entry and its reverse-complement twin always travel in the same batch because they
 * share the running minimum (ShrBin.ed, burst.c:277-280, 4218-4220).  Each batch is one bhip_align_batch call.
 */
#include "burst_host.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_sec(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int bh_device_open(const BhDb *db, int device, int z, void **hip_handle) {
	uint8_t lut[256];
	bh_score_lut(z, lut);
	int rc = bhip_init(device, db->packed, db->clumpLen, db->numRclumps, db->totR,
	                   db->hasAcx ? db->acxLens : NULL, db->hasAcx ? db->acxLists : NULL, db->acxFmt, db->K,
	                   db->badList, db->badSz, lut, db->xalpha, hip_handle);
	if (rc) return bh_set_error(rc == BHIP_E_ARG ? BH_E_USAGE : BH_E_DEVICE, "libburst_hip: %s", bhip_last_error());
	return BH_OK;
}

void bh_run_free(BhRun *run) { if (run) { free(run->hits); memset(run, 0, sizeof *run); } }

static void add_stats(BhipStats *t, const BhipStats *s) {
	t->n_queries += s->n_queries; t->n_pairs += s->n_pairs; t->n_columns += s->n_columns; t->n_raw_hits += s->n_raw_hits;
	t->n_hits += s->n_hits; t->acx_entries_read += s->acx_entries_read; t->bytes_algorithmic += s->bytes_algorithmic;
	t->ms_h2d += s->ms_h2d; t->ms_prefilter += s->ms_prefilter; t->ms_peq += s->ms_peq; t->ms_myers += s->ms_myers;
	t->ms_rescore += s->ms_rescore; t->ms_d2h += s->ms_d2h; t->ms_total += s->ms_total; t->myers_launches += s->myers_launches;
	t->n_windows += s->n_windows; t->n_window_columns += s->n_window_columns; t->n_lane_tasks += s->n_lane_tasks; t->n_task_columns += s->n_task_columns; t->ms_myers_prefix += s->ms_myers_prefix;
	t->ms_myers_window += s->ms_myers_window; t->prefix_words = s->prefix_words;
	t->n_seed_words += s->n_seed_words; t->ms_prefilter_hash += s->ms_prefilter_hash; t->ms_seed += s->ms_seed; t->prefilter_launches += s->prefilter_launches; t->prefilter_algo = s->prefilter_algo;
}

int bh_align(void *hh, const BhQueries *Q, uint64_t u0, uint64_t u1, BhMode mode, uint64_t batch_uniq, BhRun *run) {
	memset(run, 0, sizeof *run);
	if (u1 > Q->numUniq) u1 = Q->numUniq;
	if (u0 >= u1) return BH_OK;
	if (!batch_uniq) batch_uniq = 1u << 18;
	const int twoStrand = Q->numEntries > Q->numUniq;
	const int all_hits = mode == BH_FORAGE;
	uint64_t capHits = 1u << 20, nHits = 0;
	BhipHit *hits = malloc(capHits * sizeof(*hits));
	/* scratch for one batch */
	const uint64_t maxB = batch_uniq < (u1 - u0) ? batch_uniq : (u1 - u0), maxE = maxB * (twoStrand ? 2 : 1);
	uint64_t *qoff = malloc((maxE + 1) * sizeof(*qoff));
	uint16_t *emac = malloc(maxE * sizeof(*emac));
	uint32_t *six = malloc(maxE * sizeof(*six));
	uint8_t *rcf = malloc(maxE), *flags = malloc(maxE);
	uint8_t *codes = NULL; uint64_t capCodes = 0;
	if (!hits || !qoff || !emac || !six || !rcf || !flags) { free(hits); free(qoff); free(emac); free(six); free(rcf); free(flags); return bh_set_error(BH_E_OOM, "OOM:batch"); }
	int rc = BH_OK;
	for (uint64_t u = u0; u < u1 && rc == BH_OK; u += batch_uniq) {
		const uint64_t B = (u + batch_uniq <= u1 ? batch_uniq : u1 - u), nE = B * (twoStrand ? 2 : 1);
		const uint64_t fb = Q->qoff[u], fe = Q->qoff[u + B];
		const uint64_t rb = twoStrand ? Q->qoff[Q->numUniq + u] : 0, re = twoStrand ? Q->qoff[Q->numUniq + u + B] : 0;
		const uint64_t nb = (fe - fb) + (re - rb);
		if (nb + 16 > capCodes) { free(codes); capCodes = nb + nb / 4 + 64; codes = malloc(capCodes); if (!codes) { rc = bh_set_error(BH_E_OOM, "OOM:batch codes"); break; } }
		memcpy(codes, Q->codes + fb, fe - fb);
		if (twoStrand) memcpy(codes + (fe - fb), Q->codes + rb, re - rb);
		for (uint64_t j = 0; j < B; ++j) {
			const uint64_t e = u + j;
			qoff[j] = Q->qoff[e] - fb; emac[j] = Q->emac[e]; six[j] = (uint32_t)j; rcf[j] = Q->rc[e]; flags[j] = Q->flags[e];
			if (twoStrand) {
				const uint64_t er = Q->numUniq + u + j;
				qoff[B + j] = (fe - fb) + (Q->qoff[er] - rb); emac[B + j] = Q->emac[er]; six[B + j] = (uint32_t)j; rcf[B + j] = Q->rc[er]; flags[B + j] = Q->flags[er];
			}
		}
		qoff[nE] = nb;
		for (;;) {
			uint64_t n = 0;
			const double t0 = now_sec();
			int r = bhip_align_batch(hh, codes, qoff, emac, six, rcf, flags, (uint32_t)nE, (uint32_t)B, all_hits,
			                         hits + nHits, capHits - nHits, &n);
			run->secAlign += now_sec() - t0;
			if (r == BHIP_E_CAPACITY) {
				capHits = nHits + n + (nHits + n) / 2 + 1024;
				BhipHit *nh = realloc(hits, capHits * sizeof(*hits));
				if (!nh) { rc = bh_set_error(BH_E_OOM, "OOM:hits"); break; }
				hits = nh;
				continue;
			}
			if (r == BHIP_E_RESCORE) {   /* the reference's own stop (burst.c:812-816, exit(1)) */
				printf("\nCRITICAL ERROR: Truncation within known good path.\n");
				rc = bh_set_error(BH_E_USAGE, "libburst_hip: %s", bhip_last_error()); break;
			}
			if (r) { rc = bh_set_error(BH_E_DEVICE, "libburst_hip: %s", bhip_last_error()); break; }
			for (uint64_t k = nHits; k < nHits + n; ++k) {        /* local entry index -> global entry index */
				const uint32_t lq = hits[k].q;
				hits[k].q = (uint32_t)(lq < B ? u + lq : Q->numUniq + u + (lq - B));
			}
			nHits += n;
			BhipStats st;
			if (!bhip_get_stats(hh, &st)) add_stats(&run->total, &st);
			++run->nBatches;
			break;
		}
	}
	free(qoff); free(emac); free(six); free(rcf); free(flags); free(codes);
	if (rc) { free(hits); return rc; }
	run->hits = hits; run->nHits = nHits;
	return BH_OK;
}
