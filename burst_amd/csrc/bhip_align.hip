// burst_amd/csrc/bhip_align.hip -- the alignment chain of a staged batch (include/burst_hip.h: bhip_align_staged, bhip_align_batch,
// bhip_reserve) and the kernel-level entry points (bhip_align_pairs, bhip_prefilter): what replaces the bodies of the two OpenMP
// loops of do_alignments (burst.c:4077-4289, 4343-4484).  HIP-event timing of every phase on the stream it runs on.
#include "bhip_handle.h"

// ---- ordering of the output records: (q, refIx) ascending, done on the device.  A query has one or two records, rarely
// more, so a counting sort by query (rank inside the query from the counting atomic, offsets from one exclusive scan)
// followed by a tiny in-place sort of the few multi-record groups replaces a 7-pass radix sort of 64-bit keys ----
// (n_dev: the record count still lives on the device -- the sort is enqueued behind the re-scorer before the host has seen it)
__global__ void k_hit_count(const BhipHit *__restrict__ hits, uint32_t n, const uint32_t *__restrict__ n_dev, uint32_t *__restrict__ cnt, uint32_t *__restrict__ rank) {
	if (n_dev) n = *n_dev < n ? *n_dev : n;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) rank[i] = atomicAdd(&cnt[hits[i].q], 1u);
}
__global__ void k_hit_scatter(const BhipHit *__restrict__ in, uint32_t n, const uint32_t *__restrict__ n_dev, const uint32_t *__restrict__ off, const uint32_t *__restrict__ rank,
                              BhipHit *__restrict__ out, const uint32_t *__restrict__ qmap) {      // qmap: batch entry -> query number reported to the caller
	if (n_dev) n = *n_dev < n ? *n_dev : n;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		BhipHit h = in[i];
		const uint32_t dst = off[h.q] + rank[i];
		if (qmap) h.q = qmap[h.q];
		out[dst] = h;
	}
}
// Batches with query symbols of code 0 (see Handle::qcodes_s): the sweeps ran on the queries without those symbols; every
// such symbol costs exactly one edit more (it can only face a gap), so its count goes onto the raw hits and onto the
// running minima before the re-scorer -- which sees the original queries -- takes over.
__global__ void k_junk_adjust_raw(BhipRawHit *__restrict__ raw, const uint32_t *__restrict__ n_raw_dev, uint32_t raw_cap, const uint8_t *__restrict__ nx) {
	uint32_t n = *n_raw_dev;
	if (n > raw_cap) n = raw_cap;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const uint32_t x = nx[raw[i].q]; raw[i].ed += x; raw[i].m += x; }
}
__global__ void k_junk_adjust_best(uint32_t *__restrict__ best, const uint8_t *__restrict__ nx_six, uint32_t s0, uint32_t s1) {
	for (uint32_t s = s0 + blockIdx.x * blockDim.x + threadIdx.x; s < s1; s += gridDim.x * blockDim.x)
		if (nx_six[s] && best[s] != 0xFFFFFFFFu) best[s] += nx_six[s];
}

// The records of a query, scattered in arrival order, into reference order.  Round 5: one WAVE per group that needs it, a rank sort --
// lane l holds record l, its place is the number of records of the group with a smaller reference number ((query, reference) pairs are
// unique) -- instead of one thread per query shell-sorting 20-byte records in global memory: with strain-level redundancy every read
// has twenty equally good references (and some hundreds), and a wave took as long as its longest query: 330 of a batch's 410 ms.
// Groups of up to 64 records are permuted through registers; larger ones go through `scratch` (a buffer of their own,
// same size, same offsets) 64 records at a time, each ranked against the whole group.
__global__ __launch_bounds__(256) void k_hit_fix(BhipHit *__restrict__ out, const uint32_t *__restrict__ off, const uint32_t *__restrict__ cnt, uint32_t n_q,
                                               BhipHit *__restrict__ scratch, uint32_t scratch_cap) {
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
	struct Rec { uint32_t w[5]; };
	static_assert(sizeof(BhipHit) == 20, "BhipHit is five dwords");
	for (uint32_t q0 = wave * 64u; q0 < n_q; q0 += n_waves * 64u) {
		const uint32_t q = q0 + lane;
		const uint32_t n_mine = q < n_q ? cnt[q] : 0u, o_mine = q < n_q ? off[q] : 0u;
		unsigned long long todo = __ballot(n_mine >= 2u);
		while (todo) {
			const int b = __builtin_ctzll(todo);
			todo &= todo - 1ull;
			const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)n_mine, b), o = (uint32_t)__builtin_amdgcn_readlane((int)o_mine, b);
			Rec *grp = (Rec *)(out + o);
			if (n <= 64u) {
				Rec r = {};
				if (lane < n) r = grp[lane];
				const uint32_t key = lane < n ? r.w[1] : 0xFFFFFFFFu;          // refIx
				uint32_t rank = 0;
				// (stable: equal keys -- which the (query, reference) uniqueness of the records rules out -- would keep their order instead of colliding)
				for (uint32_t j = 0; j < n; ++j) { const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)j); rank += (kj < key || (kj == key && j < lane)) ? 1u : 0u; }
				if (lane < n) grp[rank] = r;                                   // (every lane's load is complete before the first store: the rank depends on all keys)
			} else if ((uint64_t)o + n <= scratch_cap) {
				Rec *tmp = (Rec *)(scratch + o);
				for (uint32_t i = lane; i < n; i += 64u) tmp[i] = grp[i];
				__threadfence();
				for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
					const uint32_t i = i0 + lane;
					Rec r = {};
					if (i < n) r = tmp[i];
					const uint32_t key = i < n ? r.w[1] : 0xFFFFFFFFu;
					uint32_t rank = 0;
					for (uint32_t j0 = 0; j0 < n; j0 += 64u) {
						const uint32_t kj = j0 + lane < n ? tmp[j0 + lane].w[1] : 0xFFFFFFFFu;
						const uint32_t m = n - j0 < 64u ? n - j0 : 64u;
						for (uint32_t j = 0; j < m; ++j) { const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)kj, (int)j); rank += (kk < key || (kk == key && j0 + j < i)) ? 1u : 0u; }
					}
					if (i < n) grp[rank] = r;
				}
			} else if (lane == 0) {                                            // (no scratch of that size: the serial sort)
				BhipHit *a = out + o;
				uint32_t gap = 1;
				while (gap < n / 3) gap = 3 * gap + 1;
				for (; gap >= 1; gap /= 3)
					for (uint32_t i = gap; i < n; ++i) {
						const BhipHit v = a[i];
						uint32_t j = i;
						for (; j >= gap && a[j - gap].refIx > v.refIx; j -= gap) a[j] = a[j - gap];
						a[j] = v;
					}
			}
		}
	}
}
__global__ void k_set_rank_ptrs(SharedCtr *sc, uint32_t *cnt, uint32_t *rank) { sc->cnt = cnt; sc->rank = rank; }

// ---- BEST on the device (all_hits = BHIP_HITS_BEST) ----------------------------------------------------------------
// The reference's BEST keeps, of a query's hits, the one with the fewest edits, then the higher f32 identity, then the lower original
// reference number RefIxSrt[refIx] (burst.c:4847-4891).  With the minimum-only semantics every record of an entry carries the same edit
// distance (k_rescore_classify lets ed == best[six] through and nothing else), so the choice inside an ENTRY is the minimum of the 64-bit
// key (~score bits, order[refIx]) -- scores are non-negative floats, their bit patterns order like their values; (entry, refIx) pairs
// are unique, so no two records of an entry share a key.  With strain-level redundancy a read has twenty equally good references:
// one record per entry leaves the device instead of all of them, and the counting sort of the records (a scatter of 20-byte records
// and a rank sort per group) is replaced by two streaming passes.  The choice between the two strands of a query (one entry each)
// stays with the host's consolidation, which knows the list order the reference would have met them in.
__device__ __forceinline__ unsigned long long best_key_of(const BhipHit &h, const uint32_t *__restrict__ order) {
	return (unsigned long long)(~__float_as_uint(h.score)) << 32 | order[h.refIx];
}
// (an entry with ONE record -- the rule on a low-redundancy database -- has nothing to choose: no table look-up, no atomic, for it)
__global__ void k_best_key(const BhipHit *__restrict__ hits, uint32_t n, const uint32_t *__restrict__ n_dev, const uint32_t *__restrict__ order, const uint32_t *__restrict__ cnt,
                           unsigned long long *__restrict__ key) {
	if (n_dev) n = *n_dev < n ? *n_dev : n;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const BhipHit h = hits[i];
		if (cnt[h.q] > 1u) atomicMin(&key[h.q], best_key_of(h, order));
	}
}
__global__ void k_best_flag(const uint32_t *__restrict__ cnt, uint32_t n_q, uint32_t *__restrict__ flag) {      // records per entry -> 1 where the entry has any (flag[n_q] = 0: the scan's total)
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n_q; i += gridDim.x * blockDim.x) flag[i] = i < n_q && cnt[i] ? 1u : 0u;
}
__global__ void k_best_emit(const BhipHit *__restrict__ hits, uint32_t n, const uint32_t *__restrict__ n_dev, const uint32_t *__restrict__ order, const uint32_t *__restrict__ cnt,
                            const unsigned long long *__restrict__ key, const uint32_t *__restrict__ off, BhipHit *__restrict__ out, const uint32_t *__restrict__ qmap) {
	if (n_dev) n = *n_dev < n ? *n_dev : n;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		BhipHit h = hits[i];
		if (cnt[h.q] > 1u && best_key_of(h, order) != key[h.q]) continue;
		const uint32_t dst = off[h.q];
		if (qmap) h.q = qmap[h.q];
		out[dst] = h;
	}
}
extern "C" int bhip_set_ref_order(void *handle, const uint32_t *order, uint32_t n) {
	Handle *h = (Handle *)handle;
	if (!h) return fail(BHIP_E_ARG, "null handle");
	if (!order) return h->n_order ? 1 : 0;
	if (n < h->tot_refs) return fail(BHIP_E_ARG, "reference order table holds %u entries, the database has %u references", n, h->tot_refs);
	HIPCHK(hipSetDevice(h->device));
	int rc;
	h->n_order = 0;
	if ((rc = h->ref_order.reserve((size_t)h->tot_refs * 4 + 16))) return rc;
	HIPCHK(hipMemcpy(h->ref_order.p, order, (size_t)h->tot_refs * 4, hipMemcpyHostToDevice));
	h->n_order = h->tot_refs;
	return BHIP_OK;
}

// ---- kernel launch helpers (st = stream to launch on) ---------------------------------------------------------------
static void launch_myers(Handle *h, Lane *L, hipStream_t st, int NW, uint32_t grid, const uint2 *pairs, const uint32_t *n_pairs_dev,
		uint64_t n_pairs_host, uint32_t li_base, const uint32_t *qlist, BhipRawHit *raw, uint32_t *n_raw, uint32_t raw_cap, uint32_t *best,
		uint8_t *mins, Counters *dc) {
	if (NW > 32) {      // queries beyond 1 024 symbols: the vector lives in LDS (2 x NW words per thread, one wave per block)
		hipLaunchKernelGGL(k_myers_long, dim3(grid * 4u), dim3(64), (size_t)NW * 512u, st, pairs, n_pairs_dev, n_pairs_host, h->n_clumps, li_base, qlist,
			L->peq.as<uint32_t>(), h->s_off(), h->s_emac(), (best && h->cur->st_has_six) ? h->cur->qsix.as<uint32_t>() : nullptr,
			h->ref_lane.as<uint4>(), h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), h->tot_refs, raw, n_raw, raw_cap, best, mins, &dc->col_sum, &dc->qlen_sum, (uint32_t)NW);
		return;
	}
	#define LM(N) hipLaunchKernelGGL(k_myers<N>, dim3(grid), dim3(256), 0, st, pairs, n_pairs_dev, n_pairs_host, h->n_clumps, li_base, qlist, \
		L->peq.as<uint32_t>(), h->s_off(), h->s_emac(), (best && h->cur->st_has_six) ? h->cur->qsix.as<uint32_t>() : nullptr, \
		h->ref_lane.as<uint4>(), h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), h->tot_refs, raw, n_raw, raw_cap, best, mins, &dc->col_sum, &dc->qlen_sum)
	switch (NW) { case 2: LM(2); break; case 4: LM(4); break; case 6: LM(6); break; case 8: LM(8); break; case 10: LM(10); break;
		case 16: LM(16); break; default: LM(32); break; }
	#undef LM
}
static void launch_prefix(Handle *h, Lane *L, hipStream_t st, int NWP, uint32_t grid, const uint2 *pairs, const uint32_t *n_pairs_dev,
		uint64_t n_pairs_host, uint32_t li_base, const uint32_t *qlist, uint32_t *n_wins, Counters *dc) {
	#define LP(N) hipLaunchKernelGGL(k_myers_prefix<N>, dim3(grid), dim3(256), 0, st, pairs, n_pairs_dev, n_pairs_host, h->n_clumps, li_base, qlist, \
		L->peqp.as<uint32_t>(), h->s_off(), h->s_emac(), h->ref_lane.as<uint4>(), h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), \
		h->tot_refs, L->wins.as<BhipWin>(), n_wins, (uint32_t)L->win_cap, &dc->col_sum, &dc->qlen_sum, dc->win_class_seen, h->cur->st_has_six ? h->cur->qsix.as<uint32_t>() : nullptr)
	if (NWP == 1) LP(1); else if (NWP == 2) LP(2); else if (NWP == 3) LP(3); else if (NWP == 4) LP(4); else LP(6);
	#undef LP
}
static void launch_prefix_task(Handle *h, Lane *L, hipStream_t st, int NWP, uint32_t grid, const uint2 *tasks, const uint32_t *n_tasks_dev, const uint32_t *qlist,
		BhipWin *wins, uint32_t *n_wins, Counters *dc, const uint4 *qmeta) {
	#define LT(N) hipLaunchKernelGGL(k_myers_prefix_task<N>, dim3(grid), dim3(64), 0, st, tasks, n_tasks_dev, (uint32_t)L->task_cap, qlist, \
		L->peqp.as<uint32_t>(), h->s_off(), h->s_emac(), h->ref_lane.as<uint4>(), h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), \
		wins, n_wins, (uint32_t)L->win_cap, &dc->tcol_sum, dc->win_class_seen, qmeta, h->cur->st_has_six ? h->cur->qsix.as<uint32_t>() : nullptr)
	if (NWP == 1) LT(1); else if (NWP == 2) LT(2); else if (NWP == 3) LT(3); else if (NWP == 4) LT(4); else LT(6);
	#undef LT
}
// Resident blocks per CU of a kernel from its static register / LDS use (512 VGPRs per SIMD lane granted in steps of 8, at
// most 8 waves per SIMD; about 148 KB of LDS): the persistent grid-stride kernels are launched with exactly that many
// blocks, a block that has to wait for a free slot would run its whole share after the others.
static uint32_t blocks_per_cu(const void *fn, uint32_t threads, size_t dyn_lds) {
	hipFuncAttributes fa;
	memset(&fa, 0, sizeof fa);
	if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return 4;
	const uint32_t waves_per_block = (threads + 63) / 64;
	const uint32_t by_reg = 4u * std::min(8u, 512u / (uint32_t)std::max(8, (fa.numRegs + 7) & ~7)) / waves_per_block;
	const size_t lds = fa.sharedSizeBytes + dyn_lds;
	const uint32_t by_lds = lds ? (uint32_t)((148u * 1024u) / std::max<size_t>(512, (lds + 511) & ~(size_t)511)) : 64u;
	return std::max(1u, std::min(by_reg, by_lds));
}

static void launch_window(Handle *h, Lane *L, hipStream_t wst, int cls, int NWP, uint32_t grid_cap, const BhipWin *wins, const uint32_t *n_wins, Counters *dc) {
	const uint32_t *six = h->cur->st_has_six ? h->cur->qsix.as<uint32_t>() : nullptr;
	const int NWc = kClasses[cls];
	// queries of three words and more: the windows whose flagged diagonals fit the two-word band go to k_myers_window_band, the
	// full-column kernel takes the rest (flag BHIP_WIN_WIDE, set by the prefix kernels)
	// the windows whose flagged diagonals fit a band of 2, 3 or 4 words (class 0 .. 2 in the record, set by the prefix kernels) go to
	// k_myers_window_band<2 .. 4> where the query has more words than that; the full-column kernel takes the rest
	const int min_class = h->opt_no_band ? 0 : NWc >= 8 ? 3 : NWc >= 6 ? 2 : NWc >= 4 ? 1 : 0;
	#define LB(BW) { uint32_t per_cu = blocks_per_cu((const void *)k_myers_window_band<BW>, 64u, 0); \
		if (h->opt_band_blocks > 0) per_cu = (uint32_t)h->opt_band_blocks; \
		const uint32_t grid = (uint32_t)h->n_cu * per_cu * (uint32_t)h->opt_oversub; \
		hipLaunchKernelGGL(k_myers_window_band<BW>, dim3(grid), dim3(64), 0, wst, wins, n_wins, (uint32_t)L->win_cap, NWP, NWc, L->peq.as<uint32_t>(), six, \
			h->ref_lane.as<uint4>(), L->raw.as<BhipRawHit>(), &dc->n_raw, (uint32_t)L->raw_cap, h->best.as<uint32_t>(), &dc->wcol_sum, dc->win_class_seen); }
	if (min_class >= 1) LB(2);
	if (min_class >= 2) LB(3);
	if (min_class >= 3) LB(4);
	#undef LB
	#define LW(N) { const uint32_t thr = (N) <= 8 ? 64u : 256u;      /* NW <= 8: per-thread A/C/G/T profile rows in LDS, 64-thread blocks */ \
		const uint32_t grid = std::min<uint32_t>(grid_cap * (256u / thr), (uint32_t)h->n_cu * blocks_per_cu((const void *)k_myers_window<N>, thr, 0)); \
		hipLaunchKernelGGL(k_myers_window<N>, dim3(grid), dim3(thr), 0, wst, wins, n_wins, (uint32_t)L->win_cap, NWP, min_class, \
		L->peq.as<uint32_t>(), six, h->ref_lane.as<uint4>(), \
		L->raw.as<BhipRawHit>(), &dc->n_raw, (uint32_t)L->raw_cap, h->best.as<uint32_t>(), &dc->wcol_sum, dc->win_class_seen); }
	switch (NWc) { case 2: LW(2); break; case 4: LW(4); break; case 6: LW(6); break; case 8: LW(8); break; case 10: LW(10); break;
		case 16: LW(16); break; default: LW(32); break; }
	#undef LW
}


static int upload_queries(Handle *h, const uint8_t *q_codes, const uint64_t *q_off, const uint16_t *q_emac,
                          const uint32_t *q_six, const uint8_t *q_rc, uint32_t n_q) {
	const uint64_t nb = q_off[n_q];
	int rc;
	h->cur->st_has_junk = false;      // bhip_stage_queries builds the search view after this upload when the batch needs one
	h->cur->st_maxlen = 0; h->cur->st_maxE = 0;
	for (uint32_t i = 0; i < n_q; ++i) {
		h->cur->st_maxlen = std::max<uint32_t>(h->cur->st_maxlen, (uint32_t)(q_off[i + 1] - q_off[i]));
		h->cur->st_maxE = std::max<uint32_t>(h->cur->st_maxE, q_emac[i]);
	}
	if ((rc = h->cur->qcodes.reserve(nb + 16))) return rc;
	if ((rc = h->cur->qoff.reserve((n_q + 1) * sizeof(uint64_t)))) return rc;
	if ((rc = h->cur->qemac.reserve((n_q + 1) * sizeof(uint16_t)))) return rc;
	if ((rc = h->cur->qsix.reserve((n_q + 1) * sizeof(uint32_t)))) return rc;
	if ((rc = h->cur->qrc.reserve(n_q + 1))) return rc;
	HIPCHK(hipMemcpyAsync(h->cur->qcodes.p, q_codes, nb, hipMemcpyHostToDevice, h->stream));
	HIPCHK(hipMemcpyAsync(h->cur->qoff.p, q_off, (n_q + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
	HIPCHK(hipMemcpyAsync(h->cur->qemac.p, q_emac, n_q * sizeof(uint16_t), hipMemcpyHostToDevice, h->stream));
	if (q_six) HIPCHK(hipMemcpyAsync(h->cur->qsix.p, q_six, n_q * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
	if (q_rc) HIPCHK(hipMemcpyAsync(h->cur->qrc.p, q_rc, n_q, hipMemcpyHostToDevice, h->stream));
	return 0;
}

static int upload_plan(Handle *h, const uint8_t *q_codes, const uint64_t *q_off, const uint16_t *q_emac, uint32_t n_q, std::vector<uint32_t> &plan) {
	int rc;
	if ((rc = h->cur->plan.reserve((size_t)n_q * 4 + 16))) return rc;
	(void)q_codes; (void)q_off; (void)q_emac;
	HIPCHK(hipMemcpyAsync(h->cur->plan.p, plan.data(), (size_t)n_q * 4, hipMemcpyHostToDevice, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	return 0;
}

// prefilter of list positions [0, n_list) of `d_qlist` on the lane's stream
static int launch_prefilter(Handle *h, Lane *L, hipStream_t pf_st, const uint32_t *d_qlist, uint32_t n_list, uint2 *cand, uint32_t *candcnt, uint32_t cand_cap,
                            bool with_bad, uint32_t *n_cand_dev, Counters *dc) {
	const uint32_t *bad = with_bad ? h->bad.as<uint32_t>() : nullptr;
	const uint32_t n_bad = with_bad ? h->n_bad : 0;
	const uint32_t *plan = h->cur->plan.as<uint32_t>();
	hipStream_t st = pf_st;
	int rc;
	// main pass: hashed counters, four queries per wave (any database size)
	if ((rc = L->fb_list.reserve((size_t)n_list * 4 + 16))) return rc;
	HIPCHK(hipMemsetAsync(&dc->n_fb, 0, 4, st));
	L->fb_dirty = true;      // (the overflow list and its counter are shared with launch_prefilter_mask: a later class of this lane must not find them)
	{
		const uint32_t n_quads = (n_list + 3) / 4;
		const uint32_t grid = std::min<uint32_t>(n_quads, (uint32_t)h->n_cu * 6);
		hipLaunchKernelGGL(k_prefilter_hash, dim3(grid), dim3(64), 0, st, h->s_codes(), h->s_off(), h->s_emac(),
			d_qlist, n_list, h->acx_view(), h->K, bad, n_bad, cand, candcnt, n_cand_dev, cand_cap, &dc->ent_read,
			plan, L->fb_list.as<uint32_t>(), &dc->n_fb);
		HIPCHK(hipGetLastError());
	}
	// fallback pass for the (rare) queries whose table overflowed: dense per-clump counters, LDS if they fit, else global memory
	const bool narrow = h->cur->st_maxlen < 255u + (uint32_t)h->K;
	const size_t lds_w = ((size_t)(h->n_clumps + (narrow ? 3 : 1)) / (narrow ? 4 : 2)) * 4 + 1536u * 4 + 512u * 8 + 512u * 4 + 16;
	const uint32_t nw32 = (h->n_clumps + 1) / 2;
	if (lds_w <= 64 * 1024) {
		const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(16, (160 * 1024) / lds_w));
		const uint32_t grid = std::min<uint32_t>(n_list, (uint32_t)h->n_cu * per_cu);
		if (narrow) hipLaunchKernelGGL(k_prefilter_wave<uint8_t>, dim3(grid), dim3(64), lds_w, st, h->s_codes(), h->s_off(),
			h->s_emac(), d_qlist, n_list, h->acx_view(), h->K, h->n_clumps, bad, n_bad, cand, candcnt,
			n_cand_dev, cand_cap, &dc->ent_read, plan, L->fb_list.as<uint32_t>(), &dc->n_fb);
		else hipLaunchKernelGGL(k_prefilter_wave<uint16_t>, dim3(grid), dim3(64), lds_w, st, h->s_codes(), h->s_off(),
			h->s_emac(), d_qlist, n_list, h->acx_view(), h->K, h->n_clumps, bad, n_bad, cand, candcnt,
			n_cand_dev, cand_cap, &dc->ent_read, plan, L->fb_list.as<uint32_t>(), &dc->n_fb);
	} else {
		// dense counters in global memory, one workgroup per query (very large databases only)
		uint32_t grid = std::min<uint32_t>(n_list, (uint32_t)h->n_cu * 2);
		if ((rc = L->gcnt.reserve((size_t)grid * nw32 * 4))) return rc;
		hipLaunchKernelGGL(k_prefilter<false>, dim3(grid), dim3(256), 0, st, h->s_codes(), h->s_off(),
			h->s_emac(), d_qlist, n_list, h->acx_view(), h->K, h->n_clumps,
			L->gcnt.as<uint32_t>(), bad, n_bad, cand, candcnt, n_cand_dev, cand_cap, &dc->ent_read, L->fb_list.as<uint32_t>(), &dc->n_fb, plan);
	}
	HIPCHK(hipGetLastError());
	return 0;
}

// lane-resolved prefilter: tasks (list position, reference lane) into L->tasks; queries whose table overflowed go through the
// words per query row of the range table (8 when no query of the class samples more)
static uint32_t seed_row_words(uint32_t maxwords) { return maxwords <= 8 ? 8u : std::max<uint32_t>(16u, (maxwords + 15u) & ~15u); }
// prefix words of the two-stage sweep for a class (0 = one-stage sweep): about 6 prefix symbols per allowed edit, shorter than the query vector
static int class_prefix_words(const Handle *h, uint32_t maxE, int NW) {
	if (!h->opt_two_stage || NW > 32) return 0;      // (beyond 1 024 symbols: single stage, k_myers_long)
	const uint32_t want = (6 * maxE + 31) / 32;
	int NWP = want <= 1 ? 1 : (want <= 2 ? 2 : (want <= 3 ? 3 : (want <= 4 ? 4 : (want <= 6 ? 6 : 0))));
	if (NWP >= NW) NWP = 0;
	return NWP;
}
// match profiles (k_build_peq) of one (lane, class) list of staged batch S: full-length rows into `peq`, prefix rows into `peqp`
static int launch_peq(Handle *h, hipStream_t st, StageSlot *S, const uint32_t *d_qlist, uint32_t n_list, int NW, int NWP, DBuf &peq, DBuf &peqp, uint32_t blocks_per_cu = 16) {
	int rc;
	if ((rc = peq.reserve((size_t)n_list * 16 * NW * 4))) return rc;
	if ((rc = peqp.reserve((size_t)n_list * 16 * 6 * 4))) return rc;
	const bool junk = S->st_has_junk;
	const uint8_t *codes = junk ? S->qcodes_s.as<uint8_t>() : S->qcodes.as<uint8_t>();
	const uint64_t *off = junk ? S->qoff_s.as<uint64_t>() : S->qoff.as<uint64_t>();
	const uint32_t *pack = junk ? S->qpack_s.as<uint32_t>() : S->qpack.as<uint32_t>();
	{
		const uint32_t qb = 256u / (uint32_t)NW;
		const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)n_list + qb - 1) / qb, (uint64_t)h->n_cu * blocks_per_cu);
		hipLaunchKernelGGL(k_build_peq, dim3(grid), dim3(256), 0, st, codes, off, d_qlist, n_list, NW, 0, h->mm, peq.as<uint32_t>(), pack, (S->st_maxlen + 7) / 8, h->peq_rows);
		HIPCHK(hipGetLastError());
	}
	if (NWP) {
		const uint32_t qb = 256u / (uint32_t)NWP;
		const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)n_list + qb - 1) / qb, (uint64_t)h->n_cu * blocks_per_cu);
		hipLaunchKernelGGL(k_build_peq, dim3(grid), dim3(256), 0, st, codes, off, d_qlist, n_list, NWP, 32 * NWP, h->mm, peqp.as<uint32_t>(), pack, (S->st_maxlen + 7) / 8, h->peq_rows);
		HIPCHK(hipGetLastError());
	}
	return 0;
}
// k_seed_ranges for one (lane, class) list of staged batch S into the lane's per-class buffers
// per-query counters of the counting-filter prefilter for an expected record stream (sampled words x occurrence-weighted mean list length)
static int pf_table_bits(const Handle *h, int algo, double expect) {
	return h->opt_pf_table ? h->opt_pf_table : algo == 0 ? (expect <= 600.0 ? 9 : expect <= 1200.0 ? 10 : 11) : (expect <= 230.0 ? 9 : expect <= 470.0 ? 10 : 11);
}
// Leaving out a query's longest list (k_seed_ranges: its guaranteed count drops from 4 to 3 for a 100-bp read at 98 %) walks ~21 % fewer
// records -- and lets more of them through the counting filter: a record survives when its counter holds need - 1 OTHER records, and a
// survivor costs about eight records' worth of work (exact-table insertion).  Measured at three database sizes:
// it pays while the remaining stream loads the counters below ~0.35 per counter (19 GB database: 152 records on 512 counters, +6 %) and
// costs at the metric's size (246 records: 8 % survivors instead of 2.5 %, -6 %); on small databases the kernel's time does not depend on
// the records at all and the extra candidates only cost sweeps.  -1 = by that rule, 0 = never, n = whenever the count stays >= n.
static uint32_t seed_min_need_for(const Handle *h, double mean_words, uint32_t W16) {
	if (h->opt_seed_min_need >= 0) return (uint32_t)h->opt_seed_min_need;
	const double t_all = mean_words * h->acx_wmean;
	const double t_less = t_all * (mean_words > 1.0 ? (mean_words - 1.0) / mean_words : 1.0) * 0.93;
	// (k_prefilter_cf: a stream of at most 255 records counts in bytes: twice the counters; the streams of a batch scatter around their mean.
	// k_prefilter_cw: 1 024 byte slots of list masks for up to 8 lists whatever the stream's length, 512 halfword slots beyond)
	const double counters = h->opt_pf_cw ? (W16 <= 8 ? 1024.0 : 512.0) : (double)(1u << pf_table_bits(h, 0, t_all)) * (h->opt_pf_bytes && t_less <= 200.0 ? 2.0 : 1.0);
	return (t_all >= 100.0 && t_less / counters <= 0.35) ? 3u : 0u;
}
static int launch_seed(Handle *h, Lane *L, hipStream_t st, StageSlot *S, int cls, const uint32_t *d_qlist, uint32_t n_list, uint32_t maxwords, double mean_words, bool ahead = false) {
	int rc;
	const uint32_t W16 = seed_row_words(maxwords);
	L->seeded_ok[cls] = false;
	if ((rc = L->ranges_c[cls].reserve((size_t)n_list * W16 * 8 + 16))) return rc;
	if ((rc = L->hdr_c[cls].reserve((size_t)n_list * 8 + 16))) return rc;
	DBuf &qm = L->qmeta_c[S->seq & 1][cls];
	if ((rc = qm.reserve((size_t)n_list * 16 + 16))) return rc;
	L->qmeta_seq[S->seq & 1][cls] = 0;
	const uint64_t n_thr = (uint64_t)n_list * W16;
	hipEvent_t *ev = L->ev_seed[S->seq & 1][cls];
	const bool junk = S->st_has_junk;
	HIPCHK(hipEventRecord(ev[0], st));
	// ahead of its batch the kernel shares the device with the sweeps of the batch before: a few blocks per CU leave them their
	// wave slots (option "seed_ahead_blocks"), and it still ends long before it is needed
	const uint64_t full = (n_thr + 255) / 256;
	const uint32_t grid = (uint32_t)(ahead && h->opt_seed_ahead_blocks > 0 ? std::min<uint64_t>(full, (uint64_t)h->n_cu * (uint64_t)h->opt_seed_ahead_blocks) : full);
	hipLaunchKernelGGL(k_seed_ranges, dim3(grid), dim3(256), 0, st,
		junk ? S->qcodes_s.as<uint8_t>() : S->qcodes.as<uint8_t>(), junk ? S->qoff_s.as<uint64_t>() : S->qoff.as<uint64_t>(), d_qlist, n_list,
		h->acx_view(), h->K, S->plan.as<uint32_t>(), W16, L->ranges_c[cls].as<uint2>(), L->hdr_c[cls].as<uint2>(),
		junk ? S->qpack_s.as<uint32_t>() : S->qpack.as<uint32_t>(), (S->st_maxlen + 7) / 8, junk ? S->qemac_s.as<uint16_t>() : S->qemac.as<uint16_t>(), qm.as<uint4>(), S->st_has_six ? S->qsix.as<uint32_t>() : nullptr,
		seed_min_need_for(h, mean_words, W16), (uint32_t)h->opt_seed_drop_len, h->alt);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(ev[1], st));
	L->qmeta_seq[S->seq & 1][cls] = S->seq + 1;
	L->seeded_ok[cls] = true; L->seeded_seq[cls] = S->seq; L->seeded_n[cls] = n_list; L->seeded_W16[cls] = W16;
	return 0;
}

// dense clump-level kernels into L->cand as (list position, clump) pairs
static int launch_prefilter_mask(Handle *h, Lane *L, hipStream_t st, int cls, const uint32_t *d_qlist, uint32_t n_list, uint32_t maxwords, uint32_t *n_tasks_dev,
                                 uint32_t *n_cand_dev, Counters *dc, int prune) {
	int rc;
	if ((rc = L->fb_list.reserve((size_t)n_list * 8 + 64))) return rc;      // two lists: the queries that overflowed the first pass, and the second
	// (n_fb, n_fb2; the lane's first class finds the whole counter block zeroed by enqueue_lane -- unless an earlier class went through the
	// clump-level prefilter, which counts ITS overflowed queries in n_fb: round 6's fuzzer under BHIP_POISON met a second pass that took the
	// list positions of a 2-word class for those of the 4-word class behind it)
	if (L->pf_launches || L->fb_dirty) HIPCHK(hipMemsetAsync(&dc->n_fb, 0, 8, st));
	L->fb_dirty = false;
	uint32_t *fb1 = L->fb_list.as<uint32_t>(), *fb2 = fb1 + n_list + 8, *fb_dense = fb1;
	uint32_t *n_fb_dense = &dc->n_fb;
	const uint32_t W16 = seed_row_words(maxwords);
	// the lookups of this batch may have run ahead (seed_next_batch, during the previous call)
	if (!(L->seeded_ok[cls] && L->seeded_seq[cls] == h->cur->seq && L->seeded_n[cls] == n_list && L->seeded_W16[cls] == W16))
		if ((rc = launch_seed(h, L, st, h->cur, cls, d_qlist, n_list, maxwords, n_list ? (double)L->seed_words[cls] / (double)n_list : (double)maxwords))) return rc;
	const int algo = h->opt_pf_algo >= 0 ? h->opt_pf_algo : L->pf_algo;
	const uint32_t n_quads = (n_list + 3) / 4;
	// hash table size per query from the expected number of distinct clumps (sampled words x occurrence-weighted mean list
	// length): 512 slots keep 12 single-wave blocks on a CU, 1024 -> 7, 2048 -> 4
	const double expect = (n_list ? (double)L->seed_words[cls] / (double)n_list : (double)maxwords) * h->acx_wmean;   // mean, not max: outliers use the fallback
	// (the touched list holds half the slots; a query that exceeds it is re-done by the dense fallback, so the estimate -- an
	// upper bound, every repeated clump counted once per word -- may be cut close)
	// (counting filter: the approximate counters tolerate a load around 1 -- false survivors only cost work)
	const int htb = pf_table_bits(h, algo, expect);
	// resident single-wave blocks per CU from the kernel's static LDS / register use (measured on gfx950: 11 blocks of 13 144 B
	// fit a CU and 12 do not, 10 of 14 168 B fit and 11 do not: about 148 KB of the 160 KB are available to them; 512 VGPRs per SIMD lane in steps of 8).  The kernel is a persistent loop over a static
	// partition of the list: one block too many per CU would run after the others and double the time.
	hipFuncAttributes fa;
	memset(&fa, 0, sizeof fa);
	// record blocks (64 per query) the counting-filter kernel keeps in registers: the expected stream of a query after the longest
	// lists have been left out (expect counts them all: an upper bound), 2 .. 4; the wider tables only come with 2 or 4
	int rb = h->opt_pf_rb ? h->opt_pf_rb : (expect <= 110.0 ? 2 : expect <= 230.0 ? 3 : 4);
	if (htb != 9 && rb == 3) rb = 4;
	const bool cw = algo == 0 && h->opt_pf_cw;      // k_prefilter_cw / k_prefilter_cq: the slot layout follows the number of lists a query can have
	const int cw_mode = W16 <= 8 ? 0 : W16 <= 16 ? 1 : 2;
	const bool cq = cw && h->opt_pf_cw == 2 && cw_mode < 2;      // four queries per wave, streams walked by the whole wave (up to 16 lists per query)
	// (the superseded counting-filter kernels -- k_prefilter_cf, k_prefilter_cw<0 / 1>: options prefilter_cw = 0 / 1 -- are not in this library:
	// libburst_hip_legacy.so, which the tests load in front of it, provides them through two weak symbols)
	const bool legacy = algo == 0 && !cq && !(cw && cw_mode == 2);
	if (legacy && (!bhip_legacy_pf_launch || !bhip_legacy_pf_attrs))
		return fail(BHIP_E_ARG, "option prefilter_cw = %d selects a superseded prefilter kernel that is not part of libburst_hip.so (tests: libburst_hip_legacy.so is loaded first)", h->opt_pf_cw);
	{
		const void *fp = cq ? (cw_mode == 0 ? (const void *)k_prefilter_cq<0, 0> : (const void *)k_prefilter_cq<1, 0>)
			: cw && cw_mode == 2 ? (const void *)k_prefilter_cw<2, 0>
			: algo == 0 ? nullptr
			: (htb == 9 ? (const void *)k_prefilter_mask<9> : htb == 10 ? (const void *)k_prefilter_mask<10> : (const void *)k_prefilter_mask<11>);
		if (legacy) { size_t lds = 0; int regs = 0; if (bhip_legacy_pf_attrs(cw ? 1 : 0, htb, rb, cw_mode, &lds, &regs)) { lds = 48 * 1024; regs = 128; } fa.sharedSizeBytes = lds; fa.numRegs = regs; }
		else if (hipFuncGetAttributes(&fa, fp) != hipSuccess) { fa.sharedSizeBytes = 48 * 1024; fa.numRegs = 128; }
	}
	const uint32_t by_lds = (148u * 1024u) / (uint32_t)std::max<size_t>(512, (fa.sharedSizeBytes + 511) & ~(size_t)511);
	const uint32_t by_reg = 4u * (512u / (uint32_t)std::max(8, (fa.numRegs + 7) & ~7));
	const uint32_t fit = std::max<uint32_t>(1u, std::min<uint32_t>(cw ? 32u : 12u, std::min(by_lds, by_reg)));
	if (getenv("BHIP_DEBUG")) {
		if (cw) fprintf(stderr, "[bhip] prefilter kernel: %s, slot mode %d, %zu B LDS, %d VGPRs -> %u blocks per CU\n", cq ? "four queries per wave, streams by the whole wave" : "one query per wave", cw_mode, fa.sharedSizeBytes, fa.numRegs, fit);
		else fprintf(stderr, "[bhip] prefilter kernel: table 2^%d, %d record blocks in registers, %zu B LDS, %d VGPRs -> %u blocks per CU\n", htb, rb, fa.sharedSizeBytes, fa.numRegs, fit);
	}
	const uint32_t waves = h->opt_pf_waves ? std::min<uint32_t>((uint32_t)h->opt_pf_waves, fit) : fit;
	const uint32_t grid = std::min<uint32_t>(cw && !cq ? n_list : n_quads, (uint32_t)h->n_cu * waves);
	HIPCHK(hipEventRecord(L->ev_pf[cls][1], st));
	if (cw || algo == 0) {
		// first pass, then the queries whose survivors overflowed its exact lane table once more with the largest tables (BIG / <11, 4>);
		// an empty second pass costs ~10 us, the dense per-clump fallback behind it 12 ms per launch at 6.8 M clumps whatever the number of queries
#define PFW_ARGS(FB, NFB, SEL, NSEL) L->ranges_c[cls].as<uint2>(), L->hdr_c[cls].as<uint2>(), W16, n_list, \
		h->acx_view().rec, h->bad.as<uint32_t>(), h->n_bad, \
		h->clump_len.as<uint32_t>(), h->tot_refs, L->tasks.as<uint2>(), n_tasks_dev, (uint32_t)L->task_cap, &dc->ent_read, \
		FB, NFB, &dc->unit_sum, &dc->col_sum, &dc->qlen_sum, &dc->surv_sum, \
		L->tasks2.as<uint2>(), &dc->n_tasks2_cls[cls], prune, SEL, NSEL, 0
		// the second pass of a strain-rich batch is not a handful of queries (round 5 gave it one block per CU): as many blocks as fit
		const uint32_t g2 = (uint32_t)h->n_cu * std::max<uint32_t>(1u, std::min<uint32_t>(4u, blocks_per_cu(cw_mode == 0 ? (const void *)k_prefilter_cq<0, 1> : (const void *)k_prefilter_cq<1, 1>, 64u, 0)));
		const bool two_pass = cw || htb != 11;
		if (legacy) {
			BhipPfLaunch a;
			memset(&a, 0, sizeof a);
			a.kind = cw ? 1 : 0; a.htb = htb; a.rb = rb; a.cw_mode = cw_mode; a.big = 0; a.grid = grid; a.stream = (void *)st;
			a.ranges = L->ranges_c[cls].as<uint2>(); a.hdr = L->hdr_c[cls].as<uint2>(); a.W16 = W16; a.n_list = n_list; a.ent = h->acx_view().rec; a.bad = h->bad.as<uint32_t>(); a.n_bad = h->n_bad;
			a.clump_len = h->clump_len.as<uint32_t>(); a.tot_refs = h->tot_refs; a.tasks = L->tasks.as<uint2>(); a.n_tasks = n_tasks_dev; a.task_cap = (uint32_t)L->task_cap; a.ent_read = &dc->ent_read;
			a.fb = fb1; a.n_fb = &dc->n_fb; a.unit_sum = &dc->unit_sum; a.col_sum = &dc->col_sum; a.qlen_sum = &dc->qlen_sum; a.surv_sum = &dc->surv_sum;
			a.tasks2 = L->tasks2.as<uint2>(); a.n_tasks2 = &dc->n_tasks2_cls[cls]; a.prune = prune; a.sel = nullptr; a.n_sel = nullptr; a.bytes = cw ? 0 : h->opt_pf_bytes;
			if (bhip_legacy_pf_launch(&a)) return fail(BHIP_E_DEVICE, "launch of a legacy prefilter kernel failed");
			if (two_pass) {
				a.big = 1; a.htb = 11; a.rb = 4; a.grid = (uint32_t)h->n_cu; a.fb = fb2; a.n_fb = &dc->n_fb2; a.sel = fb1; a.n_sel = &dc->n_fb;
				if (bhip_legacy_pf_launch(&a)) return fail(BHIP_E_DEVICE, "launch of a legacy prefilter kernel failed");
			}
		} else if (cq && cw_mode == 0) {
			hipLaunchKernelGGL((k_prefilter_cq<0, 0>), dim3(grid), dim3(64), 0, st, PFW_ARGS(fb1, &dc->n_fb, (const uint32_t *)nullptr, (const uint32_t *)nullptr));
			HIPCHK(hipGetLastError());
			hipLaunchKernelGGL((k_prefilter_cq<0, 1>), dim3(g2), dim3(64), 0, st, PFW_ARGS(fb2, &dc->n_fb2, (const uint32_t *)fb1, (const uint32_t *)&dc->n_fb));
		} else if (cq) {
			hipLaunchKernelGGL((k_prefilter_cq<1, 0>), dim3(grid), dim3(64), 0, st, PFW_ARGS(fb1, &dc->n_fb, (const uint32_t *)nullptr, (const uint32_t *)nullptr));
			HIPCHK(hipGetLastError());
			hipLaunchKernelGGL((k_prefilter_cq<1, 1>), dim3(g2), dim3(64), 0, st, PFW_ARGS(fb2, &dc->n_fb2, (const uint32_t *)fb1, (const uint32_t *)&dc->n_fb));
		} else {      // plans beyond 16 lists per query: one query per wave
			hipLaunchKernelGGL((k_prefilter_cw<2, 0>), dim3(grid), dim3(64), 0, st, PFW_ARGS(fb1, &dc->n_fb, (const uint32_t *)nullptr, (const uint32_t *)nullptr));
			HIPCHK(hipGetLastError());
			hipLaunchKernelGGL((k_prefilter_cw<2, 1>), dim3((uint32_t)h->n_cu), dim3(64), 0, st, PFW_ARGS(fb2, &dc->n_fb2, (const uint32_t *)fb1, (const uint32_t *)&dc->n_fb));
		}
#undef PFW_ARGS
		if (two_pass) { fb_dense = fb2; n_fb_dense = &dc->n_fb2; }
	} else {
#define PFM_LAUNCH(B) hipLaunchKernelGGL(k_prefilter_mask<B>, dim3(grid), dim3(64), 0, st, L->ranges_c[cls].as<uint2>(), L->hdr_c[cls].as<uint2>(), W16, n_list, \
		h->acx_view().rec, h->bad.as<uint32_t>(), h->n_bad, \
		h->clump_len.as<uint32_t>(), h->tot_refs, L->tasks.as<uint2>(), n_tasks_dev, (uint32_t)L->task_cap, &dc->ent_read, \
		fb1, &dc->n_fb, &dc->unit_sum, &dc->col_sum, &dc->qlen_sum, L->cand.as<uint2>(), n_cand_dev, (uint32_t)L->cand_cap)
		if (htb == 9) PFM_LAUNCH(9); else if (htb == 10) PFM_LAUNCH(10); else PFM_LAUNCH(11);
#undef PFM_LAUNCH
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(L->ev_pf[cls][2], st));
	++L->pf_launches;
	L->pf_algo_used = cq ? 3 : cw ? 2 : algo;
	// dense fallback for overflowed queries (clump-level pairs)
	const uint32_t *bad = h->bad.as<uint32_t>();
	const bool narrow = h->cur->st_maxlen < 255u + (uint32_t)h->K;
	const size_t lds_w = ((size_t)(h->n_clumps + (narrow ? 3 : 1)) / (narrow ? 4 : 2)) * 4 + 1536u * 4 + 512u * 8 + 512u * 4 + 16;
	const uint32_t nw32 = (h->n_clumps + 1) / 2;
	if (lds_w <= 64 * 1024) {
		const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(16, (160 * 1024) / lds_w));
		const uint32_t g2 = std::min<uint32_t>(n_list, (uint32_t)h->n_cu * per_cu);
		if (narrow) hipLaunchKernelGGL(k_prefilter_wave<uint8_t>, dim3(g2), dim3(64), lds_w, st, h->s_codes(), h->s_off(),
			h->s_emac(), d_qlist, n_list, h->acx_view(), h->K, h->n_clumps, bad, h->n_bad, L->cand.as<uint2>(),
			(uint32_t *)nullptr, n_cand_dev, (uint32_t)L->cand_cap, &dc->ent_read, h->cur->plan.as<uint32_t>(), fb_dense, n_fb_dense);
		else hipLaunchKernelGGL(k_prefilter_wave<uint16_t>, dim3(g2), dim3(64), lds_w, st, h->s_codes(), h->s_off(),
			h->s_emac(), d_qlist, n_list, h->acx_view(), h->K, h->n_clumps, bad, h->n_bad, L->cand.as<uint2>(),
			(uint32_t *)nullptr, n_cand_dev, (uint32_t)L->cand_cap, &dc->ent_read, h->cur->plan.as<uint32_t>(), fb_dense, n_fb_dense);
	} else {
		uint32_t g2 = std::min<uint32_t>(n_list, (uint32_t)h->n_cu * 2);
		if ((rc = L->gcnt.reserve((size_t)g2 * nw32 * 4))) return rc;
		hipLaunchKernelGGL(k_prefilter<false>, dim3(g2), dim3(256), 0, st, h->s_codes(), h->s_off(),
			h->s_emac(), d_qlist, n_list, h->acx_view(), h->K, h->n_clumps,
			L->gcnt.as<uint32_t>(), bad, h->n_bad, L->cand.as<uint2>(), (uint32_t *)nullptr, n_cand_dev, (uint32_t)L->cand_cap, &dc->ent_read,
			fb_dense, n_fb_dense, h->cur->plan.as<uint32_t>());
	}
	HIPCHK(hipGetLastError());
	return 0;
}

// Allocate, ahead of the first batch, what batches of up to n_entries entries of up to max_len symbols need (both staging slots,
// the scratch of the alignment kernels, the record buffers): a batch scheduler calls it once so that no allocation -- each one
// synchronises the device -- falls into its first batches.
extern "C" int bhip_reserve(void *handle, uint32_t n_entries, uint32_t max_len) { return bhip_reserve_symbols(handle, n_entries, max_len, 0); }
// total_symbols > 0: the symbols of the largest batch that will be staged (sum of its entries' lengths).  Buffers that hold the
// symbols themselves and the match profiles are then sized from it -- one 1 000-symbol read among millions of 100-symbol ones
// must not make every buffer n_entries x max_len large -- while the fixed-stride tables keep their max_len stride.
extern "C" int bhip_reserve_symbols(void *handle, uint32_t n_entries, uint32_t max_len, uint64_t total_symbols) {
	Handle *h = (Handle *)handle;
	if (!h) return fail(BHIP_E_ARG, "null handle");
	if (!n_entries) return BHIP_OK;
	if (!max_len || max_len > BHIP_MAX_QLEN) max_len = BHIP_MAX_QLEN;
	HIPCHK(hipSetDevice(h->device));
	int rc;
	if (getenv("BHIP_DEBUG")) { size_t f_ = 0, t_ = 0; if (hipMemGetInfo(&f_, &t_) == hipSuccess) fprintf(stderr, "[bhip] reserve for %u entries of up to %u symbols: %.2f GB of the device's %.2f free\n", n_entries, max_len, f_ / 1e9, t_ / 1e9); }
	const size_t n = n_entries, nb = total_symbols ? (size_t)std::min<uint64_t>(total_symbols + 64, (uint64_t)n * max_len) : n * max_len, qw = (max_len + 7) / 8;
	// profile words: an entry of len symbols has a vector of at most 2 x (len / 32 + 1) words (class rounding), 16 rows of them
	const size_t cw = (size_t)class_words(class_of_len(max_len), max_len);
	const size_t peq_words = total_symbols ? std::min<size_t>(n * cw, 2 * (nb / 32 + n)) : n * cw;
	for (StageSlot &S : h->slots) {
		if ((rc = slot_init(&S))) return rc;
		if ((rc = S.qcodes4.reserve(nb / 2 + 128)) || (rc = S.qcodes.reserve(nb + 128)) || (rc = S.qoff.reserve((n + 1) * 8)) || (rc = S.qemac.reserve((n + 1) * 2)) ||
		    (rc = S.qsix.reserve((n + 1) * 4)) || (rc = S.qrc.reserve(n + 1)) || (rc = S.qflags.reserve(n + 1)) || (rc = S.qmap.reserve((n + 1) * 4)) ||
		    (rc = S.off_raw.reserve((n + 8) * 8)) || (rc = S.plan.reserve(n * 4 + 16)) || (rc = S.qpack.reserve(n * qw * 4 + 64)) || (rc = S.key.reserve(n + 16)) ||
		    (rc = S.key_sorted.reserve(n + 16)) || (rc = S.idx.reserve(n * 4 + 16)) || (rc = S.idx_sorted.reserve(n * 4 + 16))) return rc;
		size_t tb = 0;
		HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, S.key.as<uint8_t>(), S.key_sorted.as<uint8_t>(), S.idx.as<uint32_t>(), S.idx_sorted.as<uint32_t>(), (int)n, 0, 8, h->stage_stream));
		if ((rc = S.sort_tmp.reserve(tb + 16))) return rc;
	}
	if (getenv("BHIP_DEBUG")) { size_t f_ = 0, t_ = 0; if (hipMemGetInfo(&f_, &t_) == hipSuccess) fprintf(stderr, "[bhip] ... staging slots reserved: %.2f GB free\n", f_ / 1e9); }
	if ((rc = ensure_lanes(h, 1))) return rc;
	Lane *L = h->lanes[0];
	lane_capacity_floor(h, L, n);
	const int cls = class_of_len(max_len);
	if ((rc = L->cand.reserve(L->cand_cap * sizeof(uint2))) || (rc = L->raw.reserve(L->raw_cap * sizeof(BhipRawHit))) || (rc = L->wide.reserve(L->raw_cap * sizeof(uint32_t))) ||
	    (rc = L->rs_lists.reserve(L->raw_cap * sizeof(uint32_t) * 10)) || (rc = L->scratch.reserve(L->scratch_cap * sizeof(uint32_t))) || (rc = L->wins.reserve(L->win_cap * sizeof(BhipWin))) ||
	    (rc = L->tasks.reserve(L->task_cap * sizeof(uint2))) || (rc = L->tasks2.reserve(L->task_cap * sizeof(uint2))) || (rc = L->tasks2k.reserve(L->task_cap * sizeof(uint2))) ||
	    (rc = L->wins2.reserve(L->win_cap * sizeof(BhipWin))) || (rc = L->peq.reserve(peq_words * 16 * 4)) || (rc = L->peqp.reserve(n * 16 * 6 * 4)) ||
	    (rc = L->peq_alt.reserve(peq_words * 16 * 4)) || (rc = L->peqp_alt.reserve(n * 16 * 6 * 4)) ||
	    (rc = L->fb_list.reserve(n * 8 + 64)) || (rc = L->ranges_c[cls].reserve(n * 16 * 8 + 16)) || (rc = L->hdr_c[cls].reserve(n * 8 + 16))) return rc;
	// (BEST on the device: its per-entry keys and the read-back word as well -- nothing is allocated inside the first batch)
	if (h->n_order) { if ((rc = h->best_key.reserve((n + 1) * 8)) || (rc = h->sort_idx.reserve(std::max<size_t>((size_t)h->out_cap, n + 1) * 4))) return rc; if (!h->nsel_pinned) HIPCHK(hipHostMalloc((void **)&h->nsel_pinned, 64, hipHostMallocDefault)); }
	if ((rc = h->best.reserve((n + 1) * 4)) || (rc = h->out.reserve(h->out_cap * sizeof(BhipHit))) || (rc = h->shared_ctr.reserve(sizeof(SharedCtr))) ||
	    (rc = h->sort_idx.reserve(h->out_cap * 4)) || (rc = h->sort_keys.reserve((n + 1) * 4)) || (rc = h->sort_keys2.reserve((n + 1) * 4)) ||
	    (rc = h->out_sorted.reserve(h->out_cap * sizeof(BhipHit))) || (rc = h->out_sorted2.reserve(h->out_cap * sizeof(BhipHit)))) return rc;
	{
		size_t tb = 0;
		HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, h->sort_keys.as<uint32_t>(), h->sort_keys2.as<uint32_t>(), (int)(n + 1), h->stream));
		if ((rc = h->sort_tmp.reserve(tb))) return rc;
	}
	if (getenv("BHIP_DEBUG")) { size_t f_ = 0, t_ = 0; if (hipMemGetInfo(&f_, &t_) == hipSuccess) fprintf(stderr, "[bhip] ... lane and record buffers reserved: %.2f GB free\n", f_ / 1e9); }
	if (!h->copy_stream) { HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking)); HIPCHK(hipEventCreateWithFlags(&h->ev_sorted, hipEventDisableTiming));
		for (auto &e : h->ev_copied) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
	{	// A synthetic batch through the whole path, the way a batch scheduler drives it (page-locked arrays, asynchronous copies
		// in and out): the first launch of every kernel, the first use of the copy engines from the staging and hand-over
		// streams and the first touch of the new buffers cost tens of milliseconds that would otherwise land in the caller's
		// first real batch; an idle device also clocks down, and a few milliseconds of work bring it back up.
		const uint32_t nw = std::min<uint32_t>(n_entries, 1u << 17), len = std::min<uint32_t>(max_len, 100u);
		const size_t nb_w = (size_t)nw * len + 16;
		uint8_t *pin = nullptr;
		const size_t bytes = nb_w + nb_w / 2 + 16 + ((size_t)nw + 1) * 8 + (size_t)nw * 2 + 64 + (size_t)nw * 4 * sizeof(BhipHit);
		HIPCHK(hipHostMalloc((void **)&pin, bytes, hipHostMallocPortable));
		uint8_t *codes = pin, *codes4 = pin + nb_w;
		uint64_t *off = (uint64_t *)(pin + ((nb_w + nb_w / 2 + 16 + 7) & ~(size_t)7));
		uint16_t *emac = (uint16_t *)(off + nw + 1);
		BhipHit *hbuf = (BhipHit *)(((uintptr_t)(emac + nw) + 63) & ~(uintptr_t)63);
		uint64_t x = 88172645463325252ull;
		for (size_t i = 0; i < nb_w; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; codes[i] = (uint8_t)(1 + (x & 3)); }
		for (size_t i = 0; i < nb_w / 2; ++i) codes4[i] = (uint8_t)(codes[2 * i] | codes[2 * i + 1] << 4);
		for (uint32_t i = 0; i <= nw; ++i) off[i] = (uint64_t)i * len;
		for (uint32_t i = 0; i < nw; ++i) emac[i] = (uint16_t)(len / 40);
		BhipQuerySpan sp;
		memset(&sp, 0, sizeof sp);
		sp.codes = codes; sp.codes4 = codes4; sp.off = off; sp.emac = emac; sp.n = nw;
		uint64_t n_out = 0;
		const int async = h->opt_async_d2h;
		h->opt_async_d2h = 1;
		for (StageSlot &S : h->slots) { S.state = 0; S.st_valid = false; }
		rc = 0;
		for (int rep = 0; rep < 4 && !rc; ++rep) {
			rc = bhip_stage_spans(h, &sp, 1, nw, len);
			if (!rc) rc = bhip_align_staged(h, 0, hbuf, (uint64_t)nw * 4, &n_out);
		}
		(void)bhip_sync_hits(h);
		{	// the first LARGE asynchronous copy in each direction takes another path through the runtime than the small ones above
			// and blocks its caller for ~19 ms once per process (measured in front of the first 43 MB hand-over copy): make it here
			// ... and so does the first copy that is enqueued while another one is still in flight on the same stream (measured: 16 ms in
			// front of the second hand-over copy of a process): two of each, back to back
			const size_t big = std::min<size_t>(64u << 20, std::min(h->out_sorted.cap, h->out_sorted2.cap));
			void *tmp = nullptr;
			if (big && hipHostMalloc(&tmp, 2 * big, hipHostMallocPortable) == hipSuccess) {
				(void)hipMemcpyAsync(tmp, h->out_sorted.p, big, hipMemcpyDeviceToHost, h->copy_stream);
				(void)hipMemcpyAsync((char *)tmp + big, h->out_sorted2.p, big, hipMemcpyDeviceToHost, h->copy_stream);
				(void)hipMemcpyAsync(tmp, h->out_sorted.p, big, hipMemcpyDeviceToHost, h->copy_stream);
				(void)hipStreamSynchronize(h->copy_stream);
				(void)hipMemcpyAsync(h->out_sorted.p, tmp, big, hipMemcpyHostToDevice, h->stage_stream);
				(void)hipMemcpyAsync(h->out_sorted2.p, (char *)tmp + big, big, hipMemcpyHostToDevice, h->stage_stream);
				(void)hipMemcpyAsync(h->out_sorted.p, tmp, big, hipMemcpyHostToDevice, h->stage_stream);
				(void)hipStreamSynchronize(h->stage_stream);
				// ... and a hand-over copy enqueued while the staging copies of two batches are still queued (15-17 ms once, measured
				// in front of the second batch's hand-over of the first call that stages two batches ahead)
				const size_t piece = big / 16;
				if (piece) {
					for (int i = 0; i < 12; ++i) (void)hipMemcpyAsync((char *)h->out_sorted.p + (size_t)i * piece, (char *)tmp + (size_t)i * piece, piece, hipMemcpyHostToDevice, h->stage_stream);
					(void)hipMemcpyAsync((char *)tmp + big, h->out_sorted2.p, big, hipMemcpyDeviceToHost, h->copy_stream);
					(void)hipStreamSynchronize(h->copy_stream);
					(void)hipStreamSynchronize(h->stage_stream);
				}
				(void)hipHostFree(tmp);
			}
			(void)hipGetLastError();
		}
		h->opt_async_d2h = async;
		for (StageSlot &S : h->slots) { S.state = 0; S.st_valid = false; S.spans.clear(); }
		h->res_valid = false;
		(void)hipHostFree(pin);
		if (rc) return rc;
	}
	return BHIP_OK;
}
// enqueue one lane's whole chain (no host synchronisation).  `start` = event every stream must wait for (buffers reset).
static int enqueue_lane(Handle *h, Lane *L, int all_hits, hipEvent_t start, uint32_t band_rows, uint32_t qw, uint32_t rw) {
	int rc;
	if ((rc = L->cand.reserve(L->cand_cap * sizeof(uint2)))) return rc;
	if ((rc = L->raw.reserve(L->raw_cap * sizeof(BhipRawHit)))) return rc;
	if ((rc = L->wide.reserve(L->raw_cap * sizeof(uint32_t)))) return rc;
	if ((rc = L->rs_lists.reserve(L->raw_cap * sizeof(uint32_t) * 10))) return rc;
	if ((rc = L->scratch.reserve(L->scratch_cap * sizeof(uint32_t)))) return rc;
	if ((rc = L->wins.reserve(L->win_cap * sizeof(BhipWin)))) return rc;
	if ((rc = L->tasks.reserve(L->task_cap * sizeof(uint2)))) return rc;
	if ((rc = L->tasks2.reserve(L->task_cap * sizeof(uint2)))) return rc;
	if ((rc = L->tasks2k.reserve(L->task_cap * sizeof(uint2)))) return rc;
	if ((rc = L->wins2.reserve(L->win_cap * sizeof(BhipWin)))) return rc;
	hipStream_t pf = h->pf_stream, sw = h->sweep_stream, po = h->post_stream;
	HIPCHK(hipMemsetAsync(L->counters.p, 0, sizeof(Counters), pf));
	Counters *dc = L->counters.as<Counters>();
	SharedCtr *sc = h->shared_ctr.as<SharedCtr>();
	L->launches = 0; L->prefix_words = 0; L->n_pairs_ex = 0; L->pf_launches = 0; L->fb_dirty = false;
	for (int c = 0; c < kNumClasses; ++c) { L->pf_masked[c] = false; L->pruned[c] = false; }
	const uint32_t grid_my = (uint32_t)h->n_cu * (uint32_t)h->opt_sweep_blocks;   // < 8 leaves wave slots for the other stages' kernels
	(void)start;
	for (int cls = 0; cls < kNumClasses; ++cls) {
		const uint32_t n_pf = L->npf[cls], n_ex = L->nex[cls], n_list = n_pf + n_ex;
		if (!n_list) continue;
		const int NW = class_words(cls, L->maxlen);
		const uint32_t *qlist = L->qlist[cls];
		hipEvent_t *ce = L->ev_cls[cls];
		// the lane's peq buffers are reused class after class: do not rebuild them before the previous class's window stage is done
		// (the profiles are built on the sweep stream, which is idle while this class's seeds and prefilter run on theirs)
		if (L->launches) { HIPCHK(hipStreamWaitEvent(pf, L->ev_rs[0], 0)); HIPCHK(hipStreamWaitEvent(sw, L->ev_rs[0], 0)); }
		const int NWP = class_prefix_words(h, L->maxE[cls], NW);   // two-stage edit distance when a prefix of 32*NWP symbols is selective for this class's budgets
		HIPCHK(hipEventRecord(ce[0], sw));
		L->peq_ahead[cls] = false;
		if (L->alt_ok && L->alt_seq == h->cur->seq && L->alt_cls == cls && L->alt_n == n_list && L->alt_nwp == NWP && !L->launches) {
			// built ahead during the previous batch (seed_next_batch): that batch is through, its profiles are not needed any more
			std::swap(L->peq, L->peq_alt); std::swap(L->peqp, L->peqp_alt);
			std::swap(L->ev_peq_cur[0], L->ev_peq_alt[0]); std::swap(L->ev_peq_cur[1], L->ev_peq_alt[1]);
			L->alt_ok = false; L->peq_ahead[cls] = true;
		} else if ((rc = launch_peq(h, sw, h->cur, qlist, n_list, NW, NWP, L->peq, L->peqp))) return rc;
		L->prefix_words = (uint32_t)NWP;
		HIPCHK(hipEventRecord(ce[1], sw));
		HIPCHK(hipEventRecord(ce[7], pf));
		const bool masked = NWP && n_pf && h->has_masks && h->opt_lane_masks;
		// lower-bound pruning (second sweep) only when the minimum per shared slot is all that is wanted, with the counting-filter
		// kernel (it sees all lane counts of a query at once) and while a list position fits the 24 bits next to the bound
		const int prune = masked && !all_hits && h->opt_prune && n_list < (1u << 24) && (h->opt_pf_algo >= 0 ? h->opt_pf_algo : L->pf_algo) == 0;
		if (n_pf) {
			if (masked) { if ((rc = launch_prefilter_mask(h, L, pf, cls, qlist, n_pf, L->maxwords[cls], &dc->n_tasks_cls[cls], &dc->n_cand_cls[cls], dc, prune))) return rc; }
			else if ((rc = launch_prefilter(h, L, pf, qlist, n_pf, L->cand.as<uint2>(), nullptr, (uint32_t)L->cand_cap, true, &dc->n_cand_cls[cls], dc))) return rc;
		}
		// (query, length, budget) per list position, written by this batch's k_seed_ranges: what the prefix sweeps of the tasks start from
		const uint4 *qmeta_cls = (masked && L->qmeta_seq[h->cur->seq & 1][cls] == h->cur->seq + 1) ? L->qmeta_c[h->cur->seq & 1][cls].as<uint4>() : nullptr;
		L->masked = masked;
		L->pf_masked[cls] = masked && n_pf;
		HIPCHK(hipEventRecord(ce[2], pf));
		// column sweep on the sweep stream, behind this lane's prefilter
		HIPCHK(hipStreamWaitEvent(sw, ce[2], 0));
		HIPCHK(hipEventRecord(ce[6], sw));
		if (n_pf) {
			if (masked) launch_prefix_task(h, L, sw, NWP, (uint32_t)h->n_cu * 4u * (uint32_t)h->opt_sweep_blocks * (uint32_t)h->opt_oversub, L->tasks.as<uint2>(), &dc->n_tasks_cls[cls], qlist, L->wins.as<BhipWin>(), &dc->n_wins_cls[cls], dc, qmeta_cls);
			// (beside the lane tasks the clump-level pairs are the rare overflow of the prefilter, usually none at all: a small grid --
			// an empty launch of 2 048 workgroups waited ~0.24 ms for slots on a device busy with the next batch's seed lookups and staging)
			if (NWP) launch_prefix(h, L, sw, NWP, masked ? std::min<uint32_t>(grid_my, (uint32_t)h->n_cu) : grid_my, L->cand.as<uint2>(), &dc->n_cand_cls[cls], L->cand_cap, 0, qlist, &dc->n_wins_cls[cls], dc);
			else launch_myers(h, L, sw, NW, grid_my, L->cand.as<uint2>(), &dc->n_cand_cls[cls], L->cand_cap, 0, qlist, L->raw.as<BhipRawHit>(),
				&dc->n_raw, (uint32_t)L->raw_cap, h->best.as<uint32_t>(), nullptr, dc);
			HIPCHK(hipGetLastError());
			++L->launches;
		}
		HIPCHK(hipEventRecord(ce[3], sw));
		if (n_ex) {
			const uint64_t np = (uint64_t)n_ex * h->n_clumps;
			const uint32_t g = (uint32_t)std::min<uint64_t>((np + 15) / 16, grid_my);
			if (NWP) launch_prefix(h, L, sw, NWP, g, nullptr, nullptr, np, n_pf, qlist, &dc->n_wins_cls[cls], dc);
			else launch_myers(h, L, sw, NW, g, nullptr, nullptr, np, n_pf, qlist, L->raw.as<BhipRawHit>(), &dc->n_raw, (uint32_t)L->raw_cap,
				h->best.as<uint32_t>(), nullptr, dc);
			HIPCHK(hipGetLastError());
			++L->launches;
			L->n_pairs_ex += np;
		}
		HIPCHK(hipEventRecord(ce[4], sw));
		HIPCHK(hipStreamWaitEvent(po, ce[4], 0));
		if (NWP) { launch_window(h, L, po, cls, NWP, grid_my, L->wins.as<BhipWin>(), &dc->n_wins_cls[cls], dc); HIPCHK(hipGetLastError()); }
		L->pruned[cls] = masked && prune && n_pf;
		if (masked && prune && n_pf) {
			// second sweep: the deferred lanes whose lower bound is not above the minimum found by the first sweep
			HIPCHK(hipEventRecord(L->ev_ph[cls][0], po));
			HIPCHK(hipStreamWaitEvent(sw, L->ev_ph[cls][0], 0));
			hipLaunchKernelGGL(k_task_filter, dim3((uint32_t)h->n_cu * 8), dim3(256), 0, sw, L->tasks2.as<uint2>(), &dc->n_tasks2_cls[cls], (uint32_t)L->task_cap, qlist,
				h->cur->st_has_six ? h->cur->qsix.as<uint32_t>() : nullptr, h->best.as<uint32_t>(), L->tasks2k.as<uint2>(), &dc->n_tasks2k_cls[cls]);
			launch_prefix_task(h, L, sw, NWP, (uint32_t)h->n_cu * 4u * (uint32_t)h->opt_sweep_blocks * (uint32_t)h->opt_oversub, L->tasks2k.as<uint2>(), &dc->n_tasks2k_cls[cls], qlist, L->wins2.as<BhipWin>(), &dc->n_wins2_cls[cls], dc, qmeta_cls);
			HIPCHK(hipGetLastError());
			HIPCHK(hipEventRecord(L->ev_ph[cls][1], sw));
			HIPCHK(hipStreamWaitEvent(po, L->ev_ph[cls][1], 0));
			launch_window(h, L, po, cls, NWP, grid_my, L->wins2.as<BhipWin>(), &dc->n_wins2_cls[cls], dc);
			HIPCHK(hipGetLastError());
		}
		HIPCHK(hipEventRecord(ce[5], po));
		HIPCHK(hipEventRecord(L->ev_rs[0], po));
	}
	// re-scoring of the kept reference lanes of this lane's shared slots
	HIPCHK(hipEventRecord(L->ev_rs[0], po));
	if (h->cur->st_has_junk) {       // back to the units of the original queries (see Handle::qcodes_s); this lane owns the shared slots [s0, s1)
		const uint32_t nl_ = h->cur->st_lanes, nsh_ = h->cur->st_nshared;
		uint32_t li_ = 0;
		for (uint32_t l = 0; l < nl_; ++l) if (h->lanes[l] == L) li_ = l;
		const uint32_t s0 = (uint32_t)(((uint64_t)li_ * nsh_ + nl_ - 1) / nl_), s1 = (uint32_t)(((uint64_t)(li_ + 1) * nsh_ + nl_ - 1) / nl_);
		hipLaunchKernelGGL(k_junk_adjust_raw, dim3((uint32_t)h->n_cu * 4), dim3(256), 0, po, L->raw.as<BhipRawHit>(), &dc->n_raw, (uint32_t)L->raw_cap, h->cur->nx.as<uint8_t>());
		if (s1 > s0) hipLaunchKernelGGL(k_junk_adjust_best, dim3(std::min<uint32_t>((s1 - s0 + 255) / 256, (uint32_t)h->n_cu * 4)), dim3(256), 0, po, h->best.as<uint32_t>(), h->cur->nx_six.as<uint8_t>(), s0, s1);
		HIPCHK(hipGetLastError());
	}
	// classify (exact matches leave here), register-band variants for the narrow bands, LDS band for the rest
	const uint32_t qw_g = (h->cur->st_maxlen + 7) / 8;
	hipLaunchKernelGGL(k_rescore_classify, dim3((uint32_t)h->n_cu * 8), dim3(256), 0, po, L->raw.as<BhipRawHit>(), &dc->n_raw, (uint32_t)L->raw_cap,
		h->best.as<uint32_t>(), all_hits, h->cur->qoff.as<uint64_t>(), h->cur->st_has_six ? h->cur->qsix.as<uint32_t>() : nullptr, h->cur->st_has_rc ? h->cur->qrc.as<uint8_t>() : nullptr,
		h->clump_len.as<uint32_t>(), h->out.as<BhipHit>(), &sc->n_out, (uint32_t)h->out_cap, L->rs_lists.as<uint32_t>(), dc->n_rs, L->wide.as<uint32_t>(), &dc->n_wide,
		band_rows, h->opt_rescore_reg);
	HIPCHK(hipGetLastError());
	if (h->opt_rescore_reg) {
#define RS_LAUNCH(SET, BLOCKS) hipLaunchKernelGGL(k_rescore_reg<SET>, dim3((uint32_t)h->n_cu * std::min<uint32_t>(32u, blocks_per_cu((const void *)k_rescore_reg<SET>, 64, 0)) * (uint32_t)h->opt_oversub), dim3(64), 0, po, L->raw.as<BhipRawHit>(), L->rs_lists.as<uint32_t>(), dc->n_rs, (uint32_t)L->raw_cap, \
			h->cur->qoff.as<uint64_t>(), h->cur->st_has_rc ? h->cur->qrc.as<uint8_t>() : nullptr, h->cur->qpack.as<uint32_t>(), qw_g, \
			h->ref_lane.as<uint8_t>(), h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), h->lut.as<uint8_t>(), h->out.as<BhipHit>(), &sc->n_out, (uint32_t)h->out_cap, &sc->err)
		RS_LAUNCH(0, 16);
		HIPCHK(hipGetLastError());
		RS_LAUNCH(3, 16);         // 12 diagonals: a kernel of its own, so that the 4 / 6 / 8 variants run at 80 registers (6 waves per SIMD instead of 4)
		HIPCHK(hipGetLastError());
		RS_LAUNCH(1, 12);
		HIPCHK(hipGetLastError());
		RS_LAUNCH(2, 8);          // 32 / 40 / 48 diagonals (usually empty lists: large budgets, or repeats that stretch the end-column range)
		HIPCHK(hipGetLastError());
#undef RS_LAUNCH
	}
	const uint32_t grid_rs = (uint32_t)h->n_cu * (h->opt_rescore_reg ? 4 : 16);
	const size_t lds_rs = (size_t)(band_rows + 1 + qw + rw) * 256;
	hipLaunchKernelGGL(k_rescore<false>, dim3(grid_rs), dim3(64), lds_rs, po, L->raw.as<BhipRawHit>(), &dc->n_raw, (uint32_t)L->raw_cap,
		L->rs_lists.as<uint32_t>() + (size_t)9 * L->raw_cap, &dc->n_rs[9], h->best.as<uint32_t>(), all_hits, h->cur->qcodes.as<uint8_t>(), h->cur->qoff.as<uint64_t>(),
		h->cur->st_has_six ? h->cur->qsix.as<uint32_t>() : nullptr, h->cur->st_has_rc ? h->cur->qrc.as<uint8_t>() : nullptr, h->ref_lane.as<uint8_t>(), h->ref_off.as<uint64_t>(),
		h->clump_len.as<uint32_t>(), h->lut.as<uint8_t>(), h->out.as<BhipHit>(), &sc->n_out, (uint32_t)h->out_cap, L->wide.as<uint32_t>(),
		&dc->n_wide, (uint32_t *)nullptr, &dc->scratch_used, 0ull, &sc->err, qw ? h->cur->qpack.as<uint32_t>() : nullptr, band_rows, qw, rw);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(L->ev_rs[1], po));
	HIPCHK(hipMemcpyAsync(L->hc_pinned, dc, sizeof(Counters), hipMemcpyDeviceToHost, po));
	return 0;
}

// Seed lookups of the NEXT staged batch, enqueued on the prefilter stream behind the current batch's prefilter: they run
// beside the current batch's sweeps and re-scoring (k_seed_ranges waits for HBM 70 % of its time and issues VALU work 6 % of
// it; the sweeps are VALU-bound).  Called with the current batch fully enqueued; waits (host) for the staging of the next batch
// or the end of the current one, whichever comes first.  Failures only mean the lookups run in place later.
static void seed_next_batch(Handle *h, StageSlot *cur, hipEvent_t cur_done) {
	if (!h->opt_seed_ahead || !h->has_acx || !h->has_masks || !h->opt_lane_masks || h->opt_host_routing) return;
	StageSlot *N = nullptr;
	for (StageSlot &S : h->slots) if (&S != cur && S.state == 1 && (!N || S.seq < N->seq)) N = &S;      // the batch that is aligned next
	if (!N || !N->st_nq) return;
	if (!N->resolved) {
		for (;;) {
			const hipError_t e = hipEventQuery(N->ev_done);
			if (e == hipSuccess) break;
			if (e != hipErrorNotReady) { (void)hipGetLastError(); return; }
			if (hipEventQuery(cur_done) != hipErrorNotReady) { (void)hipGetLastError(); return; }      // the current batch is through: nothing left to hide behind
			std::this_thread::yield();
		}
		(void)hipGetLastError();
		const BhipStageInfo &I = *N->info_pinned;
		if (I.err || I.junk) return;             // errors and the host routing pass are bhip_align_staged's business
		if (resolve_slot(h, N)) return;
	}
	if (N->st_has_junk || ensure_lanes(h, N->st_lanes)) return;
	for (uint32_t l = 0; l < N->st_lanes && l < h->lanes.size(); ++l) {
		Lane *L = h->lanes[l];
		for (int cls = 0; cls < kNumClasses; ++cls) {
			const uint32_t n_pf = N->npf[l][cls];
			if (!n_pf || !class_prefix_words(h, N->maxE[l][cls], class_words(cls, N->maxlen_lane[l]))) continue;      // (lane-resolved prefilter only)
			if (L->seeded_ok[cls] && L->seeded_seq[cls] == N->seq) continue;
			if (launch_seed(h, L, h->pf_stream, N, cls, N->idx_sorted.as<uint32_t>() + N->qlist_off[l][cls], n_pf, N->maxwords[l][cls], (double)N->seed_words[l][cls] / (double)n_pf, true)) { (void)hipGetLastError(); return; }
		}
		// the match profiles as well, when the lane has a single class (its two buffer pairs then simply alternate): built in place
		// they would run beside the prefilter -- which no longer has its seed lookups in front -- and slow it down
		int only = -1, n_cls = 0;
		for (int cls = 0; cls < kNumClasses; ++cls) if (N->npf[l][cls] + N->nex[l][cls]) { only = cls; ++n_cls; }
		if (n_cls == 1 && !(L->alt_ok && L->alt_seq == N->seq)) {
			const uint32_t n_list = N->npf[l][only] + N->nex[l][only];
			const int NW = class_words(only, N->maxlen_lane[l]), NWP = class_prefix_words(h, N->maxE[l][only], NW);
			L->alt_ok = false;
			if (hipEventRecord(L->ev_peq_alt[0], h->pf_stream) != hipSuccess ||
			    launch_peq(h, h->pf_stream, N, N->idx_sorted.as<uint32_t>() + N->qlist_off[l][only], n_list, NW, NWP, L->peq_alt, L->peqp_alt, (uint32_t)h->opt_peq_ahead_blocks) ||
			    hipEventRecord(L->ev_peq_alt[1], h->pf_stream) != hipSuccess) { (void)hipGetLastError(); return; }
			L->alt_ok = true; L->alt_seq = N->seq; L->alt_cls = only; L->alt_nwp = NWP; L->alt_n = n_list;
		}
	}
}

extern "C" int bhip_set_enqueued_hook(void *handle, void (*fn)(void *), void *ctx) {
	Handle *h = (Handle *)handle;
	if (!h) return fail(BHIP_E_ARG, "null handle");
	h->enqueued_hook = fn; h->enqueued_ctx = ctx;
	return BHIP_OK;
}

extern "C" int bhip_align_staged(void *handle, int all_hits_arg, BhipHit *hits, uint64_t cap, uint64_t *n_hits) {
	Handle *h = (Handle *)handle;
	if (!h || !n_hits) return fail(BHIP_E_ARG, "null argument");
	*n_hits = 0;
	if (all_hits_arg < 0 || all_hits_arg > BHIP_HITS_BEST) return fail(BHIP_E_ARG, "all_hits must be 0, 1 or 2 (BHIP_HITS_BEST)");
	const int all_hits = all_hits_arg == BHIP_HITS_ALL ? 1 : 0;      // what the kernels are told: every hit within budget, or the minimum per shared slot
	const bool sel_best = all_hits_arg == BHIP_HITS_BEST;            // ... and of those one record per entry, chosen on the device
	if (sel_best && !h->n_order) return fail(BHIP_E_ARG, "BHIP_HITS_BEST needs the reference order table (bhip_set_ref_order)");
	memset(&h->stats, 0, sizeof h->stats);
	HIPCHK(hipSetDevice(h->device));
	// the batch: the oldest one staged and not aligned yet, else the one aligned last (staged once, run any number of times)
	StageSlot *slot = nullptr;
	for (StageSlot &S : h->slots) if (S.state == 1 && (!slot || S.seq < slot->seq)) slot = &S;
	if (!slot) for (StageSlot &S : h->slots) if (S.state == 2 && (!slot || S.seq > slot->seq)) slot = &S;
	if (!slot) return fail(BHIP_E_ARG, "no staged queries (call bhip_stage_queries first)");
	{ int rcs = resolve_slot(h, slot); if (rcs) { slot->state = 0; return rcs; } }
	apply_slot(h, slot);
	const uint32_t n_q = h->cur->st_nq, n_shared = h->cur->st_nshared, nl = h->cur->st_lanes;
	if (!n_q) { slot->state = 2; return BHIP_OK; }
	// LDS plan of the re-scorer: band rows for the widest expected band (2*maxE+1 plus slack), query and reference staging
	const uint32_t band_rows = std::min<uint32_t>(BHIP_RESCORE_WMAX, 2 * h->cur->st_maxE + 1 + 9);
	uint32_t qw = (h->cur->st_maxlen + 7) / 8, rw = (h->cur->st_maxlen + band_rows + 24) / 8 + 2;
	if ((size_t)(band_rows + 1 + qw + rw) * 256 > 40 * 1024) { qw = 0; rw = 0; }      // long queries: per-row global reads instead
	SharedCtr hsc;
	uint32_t n_deliver = 0;                // records the caller gets: all of them, or one per entry (sel_best)
	bool sorted_ahead = false; int o_ahead = 0;
	if (sel_best) { int rcs; if ((rcs = h->best_key.reserve((size_t)(n_q + 1) * 8)) || (rcs = h->sort_idx.reserve(std::max<size_t>((size_t)h->out_cap, (size_t)n_q + 1) * 4))) return rcs;      // (the rank array doubles as the per-entry flags)
		if (!h->nsel_pinned) HIPCHK(hipHostMalloc((void **)&h->nsel_pinned, 64, hipHostMallocDefault)); }
	// the grouping of the records on `st` into `sorted`: the counting sort by entry (scatter + a rank sort inside every group), or -- sel_best --
	// the choice of one record per entry (two streaming passes).  cnt = records per entry (from the re-scoring kernels, or k_hit_count)
	auto enqueue_grouping = [&](hipStream_t st, uint32_t n_host, const uint32_t *n_dev, DBuf &sorted, uint32_t *cnt, uint32_t *off, uint32_t *rank, size_t tmp_bytes) -> int {
		const uint32_t g = (uint32_t)h->n_cu * 8;
		const uint32_t *qmap = h->cur->has_qmap ? h->cur->qmap.as<uint32_t>() : (const uint32_t *)nullptr;
		if (sel_best) {
			// (flags into the rank array: the ranks the re-scoring kernels took are not needed when nothing is sorted)
			hipLaunchKernelGGL(k_best_key, dim3(g), dim3(256), 0, st, h->out.as<BhipHit>(), n_host, n_dev, h->ref_order.as<uint32_t>(), cnt, h->best_key.as<unsigned long long>());
			hipLaunchKernelGGL(k_best_flag, dim3(std::min<uint32_t>((n_q + 256) / 256, g)), dim3(256), 0, st, cnt, n_q, rank);
			HIPCHK(hipcub::DeviceScan::ExclusiveSum(h->sort_tmp.p, tmp_bytes, rank, off, (int)(n_q + 1), st));
			hipLaunchKernelGGL(k_best_emit, dim3(g), dim3(256), 0, st, h->out.as<BhipHit>(), n_host, n_dev, h->ref_order.as<uint32_t>(), cnt, h->best_key.as<unsigned long long>(), off, sorted.as<BhipHit>(), qmap);
			HIPCHK(hipGetLastError());
			HIPCHK(hipMemcpyAsync(h->nsel_pinned, off + n_q, 4, hipMemcpyDeviceToHost, st));      // (the scan runs over n_q + 1 counters, the last one zero: its offset is the total)
			return 0;
		}
		HIPCHK(hipcub::DeviceScan::ExclusiveSum(h->sort_tmp.p, tmp_bytes, cnt, off, (int)(n_q + 1), st));
		hipLaunchKernelGGL(k_hit_scatter, dim3(g), dim3(256), 0, st, h->out.as<BhipHit>(), n_host, n_dev, off, rank, sorted.as<BhipHit>(), qmap);
		hipLaunchKernelGGL(k_hit_fix, dim3(std::min<uint32_t>((n_q + 255) / 256, (uint32_t)h->n_cu * 8)), dim3(256), 0, st, sorted.as<BhipHit>(), off, cnt, n_q, h->sort_scratch.as<BhipHit>(), (uint32_t)(h->sort_scratch.cap / sizeof(BhipHit)));
		HIPCHK(hipGetLastError());
		return 0;
	};
	for (int attempt = 0; attempt < 24; ++attempt) {
		int rc;
		sorted_ahead = false;
		// the records of this batch are still resident when the previous call only failed for the size of the caller's buffer
		if (h->res_valid && h->res_seq == slot->seq && h->res_all_hits == all_hits_arg) { hsc.n_out = h->res_n_raw; hsc.err = 0; h->stats = h->res_stats; n_deliver = h->res_n; *n_hits = n_deliver; }
		else {
		h->res_valid = false;
		if ((rc = h->best.reserve((size_t)(n_shared + 1) * 4))) return rc;
		if ((rc = h->out.reserve(h->out_cap * sizeof(BhipHit)))) return rc;
		if ((rc = h->shared_ctr.reserve(sizeof(SharedCtr)))) return rc;
		HIPCHK(hipEventRecord(h->ev[0], h->stream));
		HIPCHK(hipMemsetAsync(h->best.p, 0xFF, (size_t)n_shared * 4, h->stream));
		HIPCHK(hipMemsetAsync(h->shared_ctr.p, 0, 2 * sizeof(uint32_t), h->stream));
		{	// the counting sort's counters (zeroed here, off the critical path) and ranks, for the re-scoring kernels
			if ((rc = h->sort_idx.reserve((size_t)h->out_cap * 4)) || (rc = h->sort_keys.reserve((size_t)(n_q + 1) * 4))) return rc;
			HIPCHK(hipMemsetAsync(h->sort_keys.p, 0, (size_t)(n_q + 1) * 4, h->stream));
			if (sel_best) HIPCHK(hipMemsetAsync(h->best_key.p, 0xFF, (size_t)(n_q + 1) * 8, h->stream));
			hipLaunchKernelGGL(k_set_rank_ptrs, dim3(1), dim3(1), 0, h->stream, h->shared_ctr.as<SharedCtr>(), h->sort_keys.as<uint32_t>(), h->sort_idx.as<uint32_t>());
		}
		HIPCHK(hipEventRecord(h->ev[1], h->stream));
		HIPCHK(hipStreamWaitEvent(h->sweep_stream, h->ev[1], 0));
		// (the prefilter stream does not wait for these fills: nothing it runs touches `best` or the shared counters -- the sweeps and the
		// re-scorer do, on the stream the fills are on -- and the previous batch has been waited for by the host: the prefilter starts
		// ~75 us earlier)
		HIPCHK(hipStreamWaitEvent(h->post_stream, h->ev[1], 0));
		for (uint32_t l = 0; l < nl; ++l) if (h->lanes[l]->n_entries) if ((rc = enqueue_lane(h, h->lanes[l], all_hits, h->ev[1], band_rows, qw, rw))) return rc;
		HIPCHK(hipEventRecord(h->ev[2], h->pf_stream));          // this batch's share of the prefilter stream ends here
		// the records are grouped by query (counting sort) right behind the re-scorer, with the record count read on the device: no
		// host round trip between the two.  Set aside when a lane needs the wide-band re-scorer afterwards (sorted again then).
		sorted_ahead = false;
		{
			const bool async = h->opt_async_d2h && hits;
			o_ahead = async ? (h->out_idx ^ 1) : 0;
			DBuf &sorted = o_ahead ? h->out_sorted2 : h->out_sorted;
			size_t tmp_bytes = 0;
			uint32_t *cnt = nullptr, *off = nullptr, *rank = nullptr;
			if ((rc = h->sort_idx.reserve((size_t)h->out_cap * 4)) || (rc = h->sort_keys.reserve((size_t)(n_q + 1) * 4)) || (rc = h->sort_keys2.reserve((size_t)(n_q + 1) * 4)) ||
			    (rc = sorted.reserve((size_t)h->out_cap * sizeof(BhipHit))) || (rc = h->sort_scratch.reserve((size_t)h->out_cap * sizeof(BhipHit)))) return rc;
			cnt = h->sort_keys.as<uint32_t>(); off = h->sort_keys2.as<uint32_t>(); rank = h->sort_idx.as<uint32_t>();
			HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, cnt, off, (int)(n_q + 1), h->post_stream));
			if ((rc = h->sort_tmp.reserve(tmp_bytes))) return rc;
			SharedCtr *sc = h->shared_ctr.as<SharedCtr>();
			if (h->copy_pending[o_ahead]) HIPCHK(hipStreamWaitEvent(h->post_stream, h->ev_copied[o_ahead], 0));      // the copy that last read this buffer
			HIPCHK(hipEventRecord(h->ev[4], h->post_stream));
			// (counts and ranks were taken by the re-scoring kernels as they wrote the records)
			if ((rc = enqueue_grouping(h->post_stream, (uint32_t)h->out_cap, &sc->n_out, sorted, cnt, off, rank, tmp_bytes))) return rc;
			HIPCHK(hipEventRecord(h->ev[5], h->post_stream));
			sorted_ahead = true;
		}
		if (!h->hsc_pinned) HIPCHK(hipHostMalloc((void **)&h->hsc_pinned, sizeof(SharedCtr), hipHostMallocDefault));
		HIPCHK(hipMemcpyAsync(h->hsc_pinned, h->shared_ctr.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, h->post_stream));
		HIPCHK(hipEventRecord(h->ev[3], h->post_stream));
		seed_next_batch(h, slot, h->ev[3]);
		if (h->enqueued_hook && attempt == 0) h->enqueued_hook(h->enqueued_ctx);      // the caller's host work for later batches, while the device is busy with this one
		HIPCHK(hipEventSynchronize(h->ev[2]));
		HIPCHK(hipStreamSynchronize(h->sweep_stream));
		HIPCHK(hipStreamSynchronize(h->post_stream));
		for (uint32_t l = 0; l < nl; ++l) if (h->lanes[l]->n_entries) h->lanes[l]->hc = *h->lanes[l]->hc_pinned;
		for (uint32_t l = 0; l < nl; ++l) {      // a lane whose records mostly survive the counting filter does better with the exact table
			Lane *L = h->lanes[l];
			// (with the minimum-only semantics the counting-filter kernel also splits off the lanes that cannot hold a minimum --
			// the second sweep -- which the exact-table kernel does not: it only takes over when most records survive)
			// Round 6: measured again with k_prefilter_cq on a database with strain-level redundancy (57 % of the records survive its filter: a read
			// of a 500-strain family meets the family's clumps in every one of its lists): the exact-table kernel took 61.5 ms per 2 M reads where
			// k_prefilter_cq takes 8.5 (gpurun_out/r06a, r06b) -- with the minimum-only semantics the switch is off while that kernel is in use
			const bool cq_min_only = !all_hits && h->opt_pf_cw == 2 && L->pf_algo_used == 3;
			if (L->n_entries && L->pf_algo == 0 && !cq_min_only && L->hc.ent_read > 100000 && (double)L->hc.surv_sum > (all_hits ? 0.20 : 0.50) * (double)L->hc.ent_read) L->pf_algo = 1;
		}
		if (getenv("BHIP_DEBUG")) for (uint32_t l = 0; l < nl; ++l) {
			const Lane *L = h->lanes[l];
			if (!L->n_entries) continue;
			fprintf(stderr, "[bhip] lane %u: %llu list records, %llu survived the counting filter, next prefilter algorithm %d\n", l, (unsigned long long)L->hc.ent_read, (unsigned long long)L->hc.surv_sum, L->pf_algo);
			for (int cls = 0; cls < kNumClasses; ++cls) if (L->npf[cls] + L->nex[cls])
				fprintf(stderr, "[bhip] lane %u class NW=%d: prefiltered %u exhaustive %u maxE %u maxwords %u | tasks %u + deferred %u (kept %u) clump pairs %u windows %u + %u | fallback queries(last class) %u raw %u\n",
					l, kClasses[cls], L->npf[cls], L->nex[cls], L->maxE[cls], L->maxwords[cls], L->hc.n_tasks_cls[cls], L->hc.n_tasks2_cls[cls], L->hc.n_tasks2k_cls[cls], L->hc.n_cand_cls[cls], L->hc.n_wins_cls[cls], L->hc.n_wins2_cls[cls], L->hc.n_fb, L->hc.n_raw);
			if (L->hc.n_fb) fprintf(stderr, "[bhip] lane %u: %u queries overflowed the first prefilter pass, %u the second (dense fallback)\n", l, L->hc.n_fb, L->hc.n_fb2);
		}
		// capacity checks (first call of a workload: grow and redo)
		bool retry = false;
		for (uint32_t l = 0; l < nl; ++l) {
			Lane *L = h->lanes[l];
			if (!L->n_entries) continue;
			const Counters &c = L->hc;
			for (int cls = 0; cls < kNumClasses; ++cls) {
				if (c.n_cand_cls[cls] > L->cand_cap) { L->cand_cap = (uint64_t)c.n_cand_cls[cls] + c.n_cand_cls[cls] / 8 + 1024; retry = true; }
				if (c.n_tasks_cls[cls] > L->task_cap) { L->task_cap = (uint64_t)c.n_tasks_cls[cls] + c.n_tasks_cls[cls] / 8 + 1024; retry = true; }
				if (c.n_tasks2_cls[cls] > L->task_cap) { L->task_cap = (uint64_t)c.n_tasks2_cls[cls] + c.n_tasks2_cls[cls] / 8 + 1024; retry = true; }
				if (c.n_wins2_cls[cls] > L->win_cap) { L->win_cap = (uint64_t)c.n_wins2_cls[cls] + c.n_wins2_cls[cls] / 8 + 1024; retry = true; }
				if (c.n_wins_cls[cls] > L->win_cap) { L->win_cap = (uint64_t)c.n_wins_cls[cls] + c.n_wins_cls[cls] / 8 + 1024; retry = true; }
			}
			if (c.n_raw > L->raw_cap) { L->raw_cap = (uint64_t)c.n_raw + c.n_raw / 8 + 1024; retry = true; }
		}
		if (retry) continue;
		// rare: bands wider than the LDS plan (repeats inside one shear) -> global-scratch variant, lane by lane
		bool scratch_retry = false;
		for (uint32_t l = 0; l < nl; ++l) {
			Lane *L = h->lanes[l];
			if (!L->n_entries || !L->hc.n_wide) continue;
			sorted_ahead = false;                  // more records are on their way
			Counters *dc = L->counters.as<Counters>();
			SharedCtr *sc = h->shared_ctr.as<SharedCtr>();
			hipLaunchKernelGGL(k_rescore<true>, dim3(std::min<uint32_t>((L->hc.n_wide + 63) / 64, (uint32_t)h->n_cu * 16)), dim3(64), 256, h->post_stream,
				L->raw.as<BhipRawHit>(), &dc->n_raw, (uint32_t)L->raw_cap, L->wide.as<uint32_t>(), &dc->n_wide, h->best.as<uint32_t>(), all_hits,
				h->cur->qcodes.as<uint8_t>(), h->cur->qoff.as<uint64_t>(), h->cur->st_has_six ? h->cur->qsix.as<uint32_t>() : nullptr, h->cur->st_has_rc ? h->cur->qrc.as<uint8_t>() : nullptr,
				h->ref_lane.as<uint8_t>(), h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), h->lut.as<uint8_t>(), h->out.as<BhipHit>(),
				&sc->n_out, (uint32_t)h->out_cap, (uint32_t *)nullptr, (uint32_t *)nullptr, L->scratch.as<uint32_t>(), &dc->scratch_used,
				(unsigned long long)L->scratch_cap, &sc->err, (const uint32_t *)nullptr, 0u, 0u, 0u);
			HIPCHK(hipGetLastError());
			HIPCHK(hipMemcpyAsync(&L->hc, dc, sizeof(Counters), hipMemcpyDeviceToHost, h->post_stream));
			HIPCHK(hipStreamSynchronize(h->post_stream));
			if (L->hc.scratch_used > L->scratch_cap) { L->scratch_cap = (uint64_t)L->hc.scratch_used + 1024; scratch_retry = true; }
		}
		if (sorted_ahead) hsc = *h->hsc_pinned;      // (nothing ran after the chain: the copy behind it is current)
		else HIPCHK(hipMemcpy(&hsc, h->shared_ctr.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
		if (scratch_retry || (hsc.err & 2u)) continue;
		if (hsc.err & 1u) return fail(BHIP_E_RESCORE, "re-scoring could not reproduce a hit found by the edit-distance kernel (a query starting with a symbol outside the alphabet? the reference stops here as well: CRITICAL ERROR: Truncation within known good path, burst.c:812-816)");
		if (hsc.n_out > h->out_cap) { h->out_cap = (uint64_t)hsc.n_out + hsc.n_out / 8 + 1024; continue; }
		n_deliver = sel_best ? (sorted_ahead ? *h->nsel_pinned : std::min<uint32_t>(hsc.n_out, n_q)) : hsc.n_out;      // (not grouped yet: an upper bound; the grouping below gives the number)
		*n_hits = n_deliver;
		h->last_n_out = 0;
		// statistics
		BhipStats &S = h->stats;
		S.n_queries = n_q; S.n_hits = n_deliver;
		uint64_t qlen_sum = 0;
		for (uint32_t l = 0; l < nl; ++l) {
			Lane *L = h->lanes[l];
			if (!L->n_entries) continue;
			const Counters &c = L->hc;
			S.n_pairs += L->n_pairs_ex + c.unit_sum; S.n_columns += c.col_sum; S.n_task_columns += c.tcol_sum; S.n_raw_hits += c.n_raw; S.acx_entries_read += c.ent_read;
			S.myers_launches += L->launches; S.prefilter_launches += L->pf_launches; if (L->pf_launches) S.prefilter_algo = (uint32_t)L->pf_algo_used; S.n_window_columns += c.wcol_sum; qlen_sum += c.qlen_sum;
			if (L->prefix_words) S.prefix_words = L->prefix_words;
			for (int cls = 0; cls < kNumClasses; ++cls) {
				S.n_pairs += c.n_cand_cls[cls]; S.n_windows += c.n_wins_cls[cls] + c.n_wins2_cls[cls]; S.n_lane_tasks += c.n_tasks_cls[cls] + c.n_tasks2k_cls[cls];
				if (!(L->npf[cls] + L->nex[cls])) continue;
				hipEvent_t *ce = L->ev_cls[cls];
				S.ms_peq += L->peq_ahead[cls] ? ev_ms(L->ev_peq_cur[0], L->ev_peq_cur[1]) : ev_ms(ce[0], ce[1]);
				if (L->npf[cls]) S.ms_prefilter += ev_ms(ce[7], ce[2]);
				if (L->npf[cls] && L->pf_masked[cls]) { { hipEvent_t *es = L->ev_seed[h->cur->seq & 1][cls]; S.ms_seed += ev_ms(es[0], es[1]); } S.ms_prefilter_hash += ev_ms(L->ev_pf[cls][1], L->ev_pf[cls][2]); S.n_seed_words += L->seed_words[cls]; }
				float sweep = ev_ms(ce[6], ce[4]), win = ev_ms(ce[4], ce[5]);
				if (L->pruned[cls]) { const float second = ev_ms(L->ev_ph[cls][0], L->ev_ph[cls][1]); sweep += second; win -= second; }   // filter + second task sweep sit between the two window launches
				S.ms_myers += sweep + win;
				if (L->prefix_words) { S.ms_myers_prefix += sweep; S.ms_myers_window += win; }
			}
			S.ms_rescore += ev_ms(L->ev_rs[0], L->ev_rs[1]);
		}
		S.bytes_algorithmic = 8ull * S.n_columns + qlen_sum / 2 + 192ull * S.n_pairs;
		h->res_valid = true; h->res_seq = slot->seq; h->res_all_hits = all_hits_arg; h->res_n = n_deliver; h->res_n_raw = hsc.n_out; h->res_stats = h->stats;
		}
		BhipStats &S = h->stats;
		if (hits && n_deliver > cap) return fail(BHIP_E_CAPACITY, "hit buffer holds %llu records, %u needed", (unsigned long long)cap, n_deliver);
		HIPCHK(hipEventRecord(h->ev[8], h->stream));
		const bool dbg_t = getenv("BHIP_DEBUG_TIMES") != nullptr;
		const auto tq0 = std::chrono::steady_clock::now();
		auto tq = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq0).count(); };
		double tq1 = 0, tq2 = 0, tq3 = 0, tq4 = 0;
		if (hsc.n_out) {
			const uint32_t n = hsc.n_out;
			if ((rc = h->sort_idx.reserve(std::max<size_t>((size_t)n, sel_best ? (size_t)n_q + 1 : 0) * 4)) || (rc = h->sort_keys.reserve((size_t)(n_q + 1) * 4)) || (rc = h->sort_keys2.reserve((size_t)(n_q + 1) * 4)) ||
			    0) return rc;
			const bool async = h->opt_async_d2h && hits;
			const int o = async ? (h->out_idx ^= 1) : 0;
			DBuf &sorted = o ? h->out_sorted2 : h->out_sorted;
			if (sorted_ahead && o == o_ahead) tq1 = tq();          // grouped already, behind the re-scorer
			else {
			if (h->copy_pending[o]) { HIPCHK(hipEventSynchronize(h->ev_copied[o])); h->copy_pending[o] = false; }    // the copy that last read this buffer
			if ((rc = sorted.reserve((size_t)n * sizeof(BhipHit))) || (rc = h->sort_scratch.reserve((size_t)n * sizeof(BhipHit)))) return rc;
			uint32_t *cnt = h->sort_keys.as<uint32_t>(), *off = h->sort_keys2.as<uint32_t>(), *rank = h->sort_idx.as<uint32_t>();
			const uint32_t g = std::min<uint32_t>((n + 255) / 256, (uint32_t)h->n_cu * 8);
			HIPCHK(hipMemsetAsync(cnt, 0, (size_t)(n_q + 1) * 4, h->stream));
			if (sel_best) HIPCHK(hipMemsetAsync(h->best_key.p, 0xFF, (size_t)(n_q + 1) * 8, h->stream));
			tq1 = tq();
			hipLaunchKernelGGL(k_hit_count, dim3(g), dim3(256), 0, h->stream, h->out.as<BhipHit>(), n, (const uint32_t *)nullptr, cnt, rank);
			size_t tmp_bytes = 0;
			HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, cnt, off, (int)(n_q + 1), h->stream));
			if ((rc = h->sort_tmp.reserve(tmp_bytes))) return rc;
			if ((rc = enqueue_grouping(h->stream, n, (const uint32_t *)nullptr, sorted, cnt, off, rank, tmp_bytes))) return rc;
			if (sel_best) {      // (the number of selected records is only known now: rare path -- records that stayed resident, wide bands)
				HIPCHK(hipStreamSynchronize(h->stream));
				n_deliver = *h->nsel_pinned; *n_hits = n_deliver; S.n_hits = n_deliver; h->res_n = n_deliver;
				if (hits && n_deliver > cap) return fail(BHIP_E_CAPACITY, "hit buffer holds %llu records, %u needed", (unsigned long long)cap, n_deliver);
			}
			}
			tq2 = tq();
			const size_t bytes = (size_t)n_deliver * sizeof(BhipHit);
			bool queued = false;
			if (async) {
				// page-lock the caller's buffer (kept registered: callers alternate between two buffers), then copy on the copy stream
				if (!h->copy_stream) { HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking)); HIPCHK(hipEventCreateWithFlags(&h->ev_sorted, hipEventDisableTiming));
					for (auto &e : h->ev_copied) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
				const size_t want = (size_t)cap * sizeof(BhipHit);
				bool reg_ok = h->reg_ptr[o] == (void *)hits && h->reg_bytes[o] >= bytes;
				if (!reg_ok) {      // already page-locked by the caller (bhip_alloc_host / bhip_host_register)?
					hipPointerAttribute_t at;
					memset(&at, 0, sizeof at);
					if (hipPointerGetAttributes(&at, (const void *)hits) == hipSuccess && at.type == hipMemoryTypeHost) reg_ok = true;
					else (void)hipGetLastError();
				}
				if (!reg_ok) {
					if (h->reg_ptr[o]) { (void)hipHostUnregister(h->reg_ptr[o]); h->reg_ptr[o] = nullptr; }
					if (h->reg_ptr[o ^ 1] == (void *)hits) { if (h->copy_pending[o ^ 1]) { HIPCHK(hipEventSynchronize(h->ev_copied[o ^ 1])); h->copy_pending[o ^ 1] = false; }
						(void)hipHostUnregister(h->reg_ptr[o ^ 1]); h->reg_ptr[o ^ 1] = nullptr; }
					if (hipHostRegister((void *)hits, want, hipHostRegisterDefault) == hipSuccess) { h->reg_ptr[o] = (void *)hits; h->reg_bytes[o] = want; reg_ok = true; }
					else (void)hipGetLastError();
				}
				tq3 = tq();
				if (reg_ok) {
					HIPCHK(hipEventRecord(h->ev_sorted, h->stream));
					HIPCHK(hipStreamWaitEvent(h->copy_stream, h->ev_sorted, 0));
					HIPCHK(hipMemcpyAsync(hits, sorted.p, bytes, hipMemcpyDeviceToHost, h->copy_stream));
					HIPCHK(hipEventRecord(h->ev_copied[o], h->copy_stream));
					h->copy_pending[o] = true;
					queued = true;
				}
			}
			if (hits && !queued) HIPCHK(hipMemcpyAsync(hits, sorted.p, bytes, hipMemcpyDeviceToHost, h->stream));
			h->last_n_out = n_deliver; h->last_out = o;
		}
		tq4 = tq();
		HIPCHK(hipEventRecord(h->ev[9], h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		if (dbg_t) fprintf(stderr, "[bhip] delivery host ms: reserve+memset %.3f, sort launches %.3f, pointer check %.3f, copy enqueue %.3f, sync %.3f\n", tq1, tq2 - tq1, tq3 - tq2, tq4 - tq3, tq() - tq4);
		S.ms_h2d = h->cur->st_ms_h2d; S.ms_stage_copy = h->cur->st_ms_copy; S.ms_stage_route = h->cur->st_ms_route; S.ms_d2h = ev_ms(h->ev[8], h->ev[9]) + (sorted_ahead ? ev_ms(h->ev[4], h->ev[5]) : 0.0f); S.ms_total = ev_ms(h->ev[0], h->ev[9]);
		slot->state = 2;
		h->res_valid = false;
		return BHIP_OK;
	}
	return fail(BHIP_E_INTERNAL, "buffers kept overflowing");
}

extern "C" int bhip_align_batch(void *handle, const uint8_t *q_codes, const uint64_t *q_off, const uint16_t *q_emac,
                                const uint32_t *q_six, const uint8_t *q_rc, const uint8_t *q_flags,
                                uint32_t n_q, uint32_t n_shared, int all_hits,
                                BhipHit *hits, uint64_t cap, uint64_t *n_hits) {
	if (!n_hits) return fail(BHIP_E_ARG, "null argument");
	*n_hits = 0;
	int rc = bhip_stage_queries(handle, q_codes, q_off, q_emac, q_six, q_rc, q_flags, n_q, n_shared);
	if (rc) return rc;
	return bhip_align_staged(handle, all_hits, hits, cap, n_hits);
}

extern "C" int bhip_align_pairs(void *handle, const uint8_t *q_codes, const uint64_t *q_off, const uint16_t *q_emac,
                                uint32_t n_q, const uint32_t *pair_q, const uint32_t *pair_clump, uint64_t n_pairs, uint8_t *mins) {
	Handle *h = (Handle *)handle;
	if (!h || !q_codes || !q_off || !q_emac || !pair_q || !pair_clump || !mins) return fail(BHIP_E_ARG, "null argument");
	memset(&h->stats, 0, sizeof h->stats);
	for (StageSlot &S : h->slots) { S.state = 0; S.st_valid = false; }
	h->cur = &h->slots[0];
	if (!n_pairs || !n_q) return BHIP_OK;
	HIPCHK(hipSetDevice(h->device));
	int rc;
	if ((rc = ensure_lanes(h, 1))) return rc;
	Lane *L = h->lanes[0];
	uint32_t maxlen = 0;
	for (uint32_t i = 0; i < n_q; ++i) maxlen = std::max<uint32_t>(maxlen, (uint32_t)(q_off[i + 1] - q_off[i]));
	if (maxlen > BHIP_MAX_QLEN) return fail(BHIP_E_QUERYLEN, "query longer than %d", BHIP_MAX_QLEN);
	const int cls = class_of_len(std::max<uint32_t>(maxlen, 1)), NW = class_words(cls, maxlen);
	std::vector<uint2> pr(n_pairs);
	for (uint64_t p = 0; p < n_pairs; ++p) {
		if (pair_q[p] >= n_q || pair_clump[p] >= h->n_clumps) return fail(BHIP_E_ARG, "pair %llu out of range", (unsigned long long)p);
		pr[p] = make_uint2(pair_q[p], pair_clump[p]);
	}
	h->cur->st_has_six = false; h->cur->st_has_rc = false;
	if ((rc = upload_queries(h, q_codes, q_off, q_emac, nullptr, nullptr, n_q))) return rc;
	if ((rc = L->peq.reserve((size_t)n_q * 16 * NW * 4))) return rc;
	if ((rc = h->pairs.reserve(n_pairs * sizeof(uint2)))) return rc;
	if ((rc = h->mins.reserve(n_pairs * 16))) return rc;
	hipStream_t st = h->stream;
	HIPCHK(hipMemcpyAsync(h->pairs.p, pr.data(), n_pairs * sizeof(uint2), hipMemcpyHostToDevice, st));
	HIPCHK(hipMemsetAsync(L->counters.p, 0, sizeof(Counters), st));
	Counters *dc = L->counters.as<Counters>();
	const uint32_t qb = 256u / (uint32_t)NW;
	hipLaunchKernelGGL(k_build_peq, dim3((uint32_t)std::min<uint64_t>(((uint64_t)n_q + qb - 1) / qb, (uint64_t)h->n_cu * 16)), dim3(256), 0, st,
		h->cur->qcodes.as<uint8_t>(), h->cur->qoff.as<uint64_t>(), (const uint32_t *)nullptr, n_q, NW, 0, h->mm, L->peq.as<uint32_t>(), (const uint32_t *)nullptr, 0u, h->peq_rows);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(h->ev[0], st));
	launch_myers(h, L, st, NW, (uint32_t)std::min<uint64_t>((n_pairs + 15) / 16, (uint64_t)h->n_cu * 8), h->pairs.as<uint2>(), nullptr, n_pairs, 0, nullptr,
		nullptr, nullptr, 0, nullptr, h->mins.as<uint8_t>(), dc);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(h->ev[1], st));
	HIPCHK(hipMemcpyAsync(mins, h->mins.p, n_pairs * 16, hipMemcpyDeviceToHost, st));
	Counters hc;
	HIPCHK(hipMemcpyAsync(&hc, dc, sizeof hc, hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	h->stats.n_queries = n_q; h->stats.n_pairs = n_pairs; h->stats.n_columns = hc.col_sum; h->stats.myers_launches = 1;
	h->stats.bytes_algorithmic = 8ull * hc.col_sum + hc.qlen_sum / 2 + 192ull * n_pairs;
	h->stats.ms_myers = ev_ms(h->ev[0], h->ev[1]); h->stats.ms_total = h->stats.ms_myers;
	return BHIP_OK;
}

extern "C" int bhip_prefilter(void *handle, const uint8_t *q_codes, const uint64_t *q_off, const uint16_t *q_emac, uint32_t n_q,
                              uint32_t *out_q, uint32_t *out_clump, uint32_t *out_count, uint64_t cap, uint64_t *n_out) {
	Handle *h = (Handle *)handle;
	if (!h || !q_codes || !q_off || !q_emac || !n_out) return fail(BHIP_E_ARG, "null argument");
	if (!h->has_acx) return fail(BHIP_E_ARG, "handle has no accelerator");
	*n_out = 0;
	for (StageSlot &S : h->slots) { S.state = 0; S.st_valid = false; }
	h->cur = &h->slots[0];
	if (!n_q) return BHIP_OK;
	HIPCHK(hipSetDevice(h->device));
	int rc;
	if ((rc = ensure_lanes(h, 1))) return rc;
	Lane *L = h->lanes[0];
	for (int attempt = 0; attempt < 4; ++attempt) {
		if ((rc = upload_queries(h, q_codes, q_off, q_emac, nullptr, nullptr, n_q))) return rc;
		{
			std::vector<uint32_t> plan(n_q, 1u);
			for (uint32_t i = 0; i < n_q; ++i) plan[i] = make_seed_plan(q_codes + q_off[i], (uint32_t)(q_off[i + 1] - q_off[i]), q_emac[i], (uint32_t)h->K, h->opt_prefilter_stride, h->alt);
			if ((rc = upload_plan(h, q_codes, q_off, q_emac, n_q, plan))) return rc;
	{	// 4-bit packed copy of the queries at a fixed stride (layout used by the seed, profile and re-scoring kernels)
		const uint32_t qw_g = (h->cur->st_maxlen + 7) / 8;
		if ((rc = h->cur->qpack.reserve((size_t)n_q * qw_g * 4 + 16))) return rc;
		const uint64_t total = (uint64_t)n_q * qw_g;
		if (total) hipLaunchKernelGGL(k_pack_queries, dim3((uint32_t)std::min<uint64_t>((total + 255) / 256, (uint64_t)h->n_cu * 16)), dim3(256), 0, h->stream,
			h->cur->qcodes.as<uint8_t>(), h->cur->qoff.as<uint64_t>(), n_q, qw_g, h->cur->qpack.as<uint32_t>());
		HIPCHK(hipGetLastError());
	}
		}
		if ((rc = L->cand.reserve(L->cand_cap * sizeof(uint2)))) return rc;
		if ((rc = L->candcnt.reserve(L->cand_cap * sizeof(uint32_t)))) return rc;
		HIPCHK(hipStreamSynchronize(h->stream));
		HIPCHK(hipMemsetAsync(L->counters.p, 0, sizeof(Counters), L->stream));
		Counters *dc = L->counters.as<Counters>();
		HIPCHK(hipEventRecord(h->ev[0], L->stream));
		if ((rc = launch_prefilter(h, L, L->stream, nullptr, n_q, L->cand.as<uint2>(), L->candcnt.as<uint32_t>(), (uint32_t)L->cand_cap, false, &dc->n_cand, dc))) return rc;
		HIPCHK(hipEventRecord(h->ev[1], L->stream));
		Counters hc;
		HIPCHK(hipMemcpyAsync(&hc, dc, sizeof hc, hipMemcpyDeviceToHost, L->stream));
		HIPCHK(hipStreamSynchronize(L->stream));
		if (hc.n_cand > L->cand_cap) { L->cand_cap = (uint64_t)hc.n_cand + 1024; continue; }
		*n_out = hc.n_cand;
		memset(&h->stats, 0, sizeof h->stats);
		h->stats.n_queries = n_q; h->stats.n_pairs = hc.n_cand; h->stats.acx_entries_read = hc.ent_read; h->stats.ms_prefilter = ev_ms(h->ev[0], h->ev[1]);
		if (hc.n_cand > cap) return fail(BHIP_E_CAPACITY, "candidate buffer holds %llu, %u needed", (unsigned long long)cap, hc.n_cand);
		std::vector<uint2> c(hc.n_cand); std::vector<uint32_t> cc(hc.n_cand);
		if (hc.n_cand) {
			HIPCHK(hipMemcpy(c.data(), L->cand.p, hc.n_cand * sizeof(uint2), hipMemcpyDeviceToHost));
			HIPCHK(hipMemcpy(cc.data(), L->candcnt.p, hc.n_cand * sizeof(uint32_t), hipMemcpyDeviceToHost));
		}
		std::vector<uint32_t> ord(hc.n_cand);
		for (uint32_t i = 0; i < hc.n_cand; ++i) ord[i] = i;
		std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return c[a].x != c[b].x ? c[a].x < c[b].x : c[a].y < c[b].y; });
		for (uint32_t i = 0; i < hc.n_cand; ++i) {
			if (out_q) out_q[i] = c[ord[i]].x;
			if (out_clump) out_clump[i] = c[ord[i]].y;
			if (out_count) out_count[i] = cc[ord[i]];
		}
		return BHIP_OK;
	}
	return fail(BHIP_E_INTERNAL, "candidate buffer kept overflowing");
}

// Device-resident copy of the last call's records (same order as the host copy): for device-side collectives.
extern "C" int bhip_copy_hits_device(void *handle, void *dst_device, uint64_t cap_records, uint64_t *n_records) {
	Handle *h = (Handle *)handle;
	if (!h || !n_records) return fail(BHIP_E_ARG, "null argument");
	*n_records = h->last_n_out;
	if (!h->last_n_out) return BHIP_OK;
	// (a count query -- no destination, no room -- is answered like any buffer that is too small: BHIP_E_CAPACITY with *n_records set)
	if (!dst_device && cap_records) return fail(BHIP_E_ARG, "null destination");
	if (h->last_n_out > cap_records) return fail(BHIP_E_CAPACITY, "device buffer holds %llu records, %llu needed", (unsigned long long)cap_records, (unsigned long long)h->last_n_out);
	HIPCHK(hipSetDevice(h->device));
	HIPCHK(hipMemcpyAsync(dst_device, (h->last_out ? h->out_sorted2 : h->out_sorted).p, (size_t)h->last_n_out * sizeof(BhipHit), hipMemcpyDeviceToDevice, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	return BHIP_OK;
}

// With option "async_d2h" the records of bhip_align_staged / bhip_align_batch arrive in the caller's buffer behind the call
// (the count is final at return); this waits for every copy still in flight.
extern "C" int bhip_sync_hits(void *handle) {
	Handle *h = (Handle *)handle;
	if (!h) return fail(BHIP_E_ARG, "null handle");
	HIPCHK(hipSetDevice(h->device));
	for (int o = 0; o < 2; ++o) if (h->copy_pending[o]) { HIPCHK(hipEventSynchronize(h->ev_copied[o])); h->copy_pending[o] = false; }
	return BHIP_OK;
}

