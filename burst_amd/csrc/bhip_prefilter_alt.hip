// burst_amd/csrc/bhip_prefilter_alt.hip -- the prefilter kernels beside k_prefilter_cq that a batch of the PRODUCT can reach:
//   k_prefilter_hash / k_prefilter_wave / k_prefilter   clump-level path (handles without lane information, one-stage classes) and the
//                                                       dense fallback of queries that overflow every table (burst.c:3238-3282 as is);
//   k_prefilter_mask<HTB>                               exact clump hash in two passes: FORAGE over dense families (prefilter_algo = 1);
//   k_prefilter_cw<2, BIG>                              one query per wave, any number of lists: plans beyond 16 sampled words per query.
#undef PFM_PROF          // (the phase timers belong to bhip_prefilter.hip)
#include "bhip_pf_common.h"
#include "bhip_prefilter_cw.h"
// ------------------------------------------------------------------------------------------------
// Prefilter (burst.c:4096-4133 + postScour 3238-3282, per query instead of per bunch of 16).
// counter[c] = number of query k-mer positions whose word occurs in clump c.  A clump is a candidate iff
// counter > mmatch, mmatch = max(len - (E+1)K, 0): every alignment with <= E edits keeps at least
// len-K+1-E*K = mmatch+1 intact words (burst.c:4091-4092, 4163-4164), so no valid clump is dropped.
// Words containing a symbol outside A/C/G/T are skipped here; the host routes such queries to the
// exhaustive path.
// ------------------------------------------------------------------------------------------------
template <bool LDS_CNT>
__global__ __launch_bounds__(256) void k_prefilter(
		const uint8_t *__restrict__ qcodes, const uint64_t *__restrict__ qoff, const uint16_t *__restrict__ qemac,
		const uint32_t *__restrict__ qlist, uint32_t n_list,
		BhipAcxView acx, int K, uint32_t n_clumps,
		uint32_t *__restrict__ g_cnt, const uint32_t *__restrict__ bad, uint32_t n_bad,
		uint2 *__restrict__ cand, uint32_t *__restrict__ cand_cnt_out, uint32_t *__restrict__ n_cand, uint32_t cand_cap,
		unsigned long long *__restrict__ ent_read, const uint32_t *__restrict__ sel, const uint32_t *__restrict__ n_sel_dev,
		const uint32_t *__restrict__ plan) {   // plan made with stride 1 for this kernel
	extern __shared__ __attribute__((aligned(16))) uint32_t s_cnt[];
	const uint32_t nw32 = (n_clumps + 1) >> 1;
	uint32_t *cnt = LDS_CNT ? s_cnt : g_cnt + (uint64_t)blockIdx.x * nw32;
	const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t wmask = K == 16 ? 0xFFFFFFFFu : ((1u << (2 * K)) - 1u);
	unsigned long long my_ent = 0;
	const uint32_t n_iter = sel ? (*n_sel_dev < n_list ? *n_sel_dev : n_list) : n_list;
	for (uint32_t it = blockIdx.x; it < n_iter; it += gridDim.x) {
		const uint32_t li = sel ? sel[it] : it;
		const uint32_t q = qlist ? qlist[li] : li;
		const uint64_t b = qoff[q];
		const uint32_t len = (uint32_t)(qoff[q + 1] - b), E = qemac[q];
		for (uint32_t i = tid; i < nw32; i += 256) cnt[i] = 0;
		__syncthreads();
		if (len >= (uint32_t)K) {
			const uint32_t nwords = len - K + 1;
			// 64 word positions per wave pass: lane j builds the word starting at base+j
			for (uint32_t base = wave * 64; base < nwords; base += 256) {
				const uint32_t p = base + lane;
				uint32_t w = 0, ok = p < nwords;
				if (ok) for (int k = 0; k < K; ++k) {
					uint32_t c = qcodes[b + p + k];
					ok &= (c - 1u) < 4u;
					w = (w << 2) | ((c - 1u) & 3u);
				}
				w &= wmask;
				unsigned long long beg = 0; uint32_t n = 0;
				if (ok) bhip_acx_range(acx, w, beg, n);
				my_ent += n;
				// short lists: each lane walks its own; long lists: the wave walks them together
				unsigned long long longm = __ballot(n > 32);
				if (n <= 32) for (uint32_t e = 0; e < n; ++e) {
					uint32_t c = bhip_acx_clump(acx.rec, beg + e);
					atomicAdd(&cnt[c >> 1], 1u << ((c & 1) * 16));
				}
				while (longm) {
					const int src = __builtin_ctzll(longm);
					longm &= longm - 1;
					const unsigned long long lb = __shfl(beg, src); const uint32_t ln = __shfl(n, src);
					for (uint32_t e = lane; e < ln; e += 64) {
						uint32_t c = bhip_acx_clump(acx.rec, lb + e);
						atomicAdd(&cnt[c >> 1], 1u << ((c & 1) * 16));
					}
				}
			}
		}
		__syncthreads();
		// (this kernel counts words of A/C/G/T only: of the plan's need, the x words that vote through expansions are not seen here; when
		// nothing is left of it every clump is a candidate)
		const uint32_t px = plan ? BHIP_PLAN_X(plan[q]) : 0u, pn = plan ? BHIP_PLAN_NEED(plan[q]) : 0u;
		const uint32_t need1 = pn > px ? pn - px : 0u;
		const bool takeall = px && !need1;
		const uint32_t kload = E * K + K, mmatch = plan ? (need1 ? need1 - 1 : 0u) : (kload < len ? len - kload : 0);
		for (uint32_t c = tid; c < n_clumps; c += 256) {
			const uint32_t v = (cnt[c >> 1] >> ((c & 1) * 16)) & 0xFFFFu;
			if (v > mmatch || takeall) {
				const uint32_t pos = atomicAdd(n_cand, 1u);
				if (pos < cand_cap) { cand[pos] = make_uint2(li, c); if (cand_cnt_out) cand_cnt_out[pos] = v; }
			}
		}
		for (uint32_t i = tid; i < n_bad && !takeall; i += 256) {          // burst.c:4136-4138, 4282-4283 (with every clump taken they are in already)
			const uint32_t pos = atomicAdd(n_cand, 1u);
			if (pos < cand_cap) { cand[pos] = make_uint2(li, bad[i]); if (cand_cnt_out) cand_cnt_out[pos] = 0xFFFFFFFFu; }
		}
		__syncthreads();
	}
	if (ent_read && my_ent) atomicAdd(ent_read, my_ent);
}

template __global__ void k_prefilter<false>(const uint8_t *, const uint64_t *, const uint16_t *, const uint32_t *, uint32_t,
	BhipAcxView, int, uint32_t, uint32_t *, const uint32_t *, uint32_t, uint2 *, uint32_t *, uint32_t *, uint32_t, unsigned long long *,
	const uint32_t *, const uint32_t *, const uint32_t *);


// ------------------------------------------------------------------------------------------------
// Prefilter, wave-per-query variant (used whenever the per-clump counters of one query fit a wave's LDS slice).
// Differences to k_prefilter above, all aimed at what the round-1 profile showed to dominate (profiles/r01_*):
//   * one 64-lane wave owns a query (no workgroup barriers), several waves per CU run independent queries;
//   * counters are bytes (CNT = uint8_t, four per dword) while len-K+1 <= 255, else 16-bit;
//   * no dense zero/scan per query: the first increment of a counter (atomic returns 0) appends the clump to a
//     touched list; only touched counters are tested against the threshold and reset.  Dense fallback if the list overflows;
//   * candidates are staged in LDS and flushed with ONE global atomic per flush instead of one returning atomic per
//     candidate (2.2 M same-address atomics per launch saturated the L2 atomic unit at ~90/us).
// ------------------------------------------------------------------------------------------------
// Seed plan of one query (k_route on the device, make_seed_plan on the host: bhip_seed_plan; layout BHIP_PLAN_* in bhip_internal.h:
// stride | need << 8 | x << 24 | used << 28): word starts 0, s, 2s, ... <= len-K are sampled.  A word of A/C/G/T votes; with
// non-overlapping words (s = K) a word holding exactly ONE ambiguous symbol with 2..4 compatible bases votes through its expansions
// (x such words, `used` extra word slots: the reference's storeAmbigWords, burst.c:3232-3236, restricted to one ambiguous symbol per
// word); any other word does not vote.  One edit destroys at most ceil(K/s) sampled words, so an alignment with <= E edits keeps
// need = W_voting - E*ceil(K/s) of them.  s = 1 with no ambiguity is the reference's scheme (need = len-K+1-E*K = mmatch+1,
// burst.c:4091-4092).  Queries with need < 1 never reach these kernels (they are routed to the exhaustive path); the clump-level
// kernels below count strictly (words of A/C/G/T only): need - x, and every clump when nothing is left.
#define PF2_TL 1536u      // touched-list capacity (clump ids, u32)
#define PF2_STAGE 512u    // staged candidates (uint2)
template <typename CNT>
__global__ __launch_bounds__(64) void k_prefilter_wave(
		const uint8_t *__restrict__ qcodes, const uint64_t *__restrict__ qoff, const uint16_t *__restrict__ qemac,
		const uint32_t *__restrict__ qlist, uint32_t n_list,
		BhipAcxView acx, int K, uint32_t n_clumps,
		const uint32_t *__restrict__ bad, uint32_t n_bad,
		uint2 *__restrict__ cand, uint32_t *__restrict__ cand_cnt_out, uint32_t *__restrict__ n_cand, uint32_t cand_cap,
		unsigned long long *__restrict__ ent_read, const uint32_t *__restrict__ plan,
		const uint32_t *__restrict__ sel, const uint32_t *__restrict__ n_sel_dev) {   // optional: only list positions sel[0..*n_sel_dev)
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
	constexpr uint32_t PER = 4 / sizeof(CNT), BITS = 8 * sizeof(CNT), MASK = (1u << BITS) - 1u;
	const uint32_t nw32 = (n_clumps + PER - 1) / PER;
	uint32_t *cnt = smem;                       // [nw32]
	uint32_t *tl = cnt + nw32;                  // [PF2_TL]
	uint2 *stage = (uint2 *)(tl + PF2_TL);      // [PF2_STAGE]
	uint32_t *stage_v = (uint32_t *)(stage + PF2_STAGE);   // [PF2_STAGE] counts (only written when cand_cnt_out)
	uint32_t *ctr = stage_v + PF2_STAGE;        // [0] touched count, [1] staged count
	const uint32_t lane = threadIdx.x;
	const uint32_t wmask = K == 16 ? 0xFFFFFFFFu : ((1u << (2 * K)) - 1u);
	for (uint32_t i = lane; i < nw32; i += 64) cnt[i] = 0;
	if (lane < 2) ctr[lane] = 0;
	__syncthreads();
	unsigned long long my_ent = 0;

	auto push = [&](uint32_t li, uint32_t c, uint32_t v) {
		const uint32_t pos = atomicAdd(&ctr[1], 1u);
		if (pos < PF2_STAGE) { stage[pos] = make_uint2(li, c); if (cand_cnt_out) stage_v[pos] = v; }
		else {   // staging buffer full inside one query (very permissive threshold): direct append
			const uint32_t g = atomicAdd(n_cand, 1u);
			if (g < cand_cap) { cand[g] = make_uint2(li, c); if (cand_cnt_out) cand_cnt_out[g] = v; }
		}
	};
	auto flush = [&]() {
		__syncthreads();
		const uint32_t n = ctr[1] < PF2_STAGE ? ctr[1] : PF2_STAGE;
		uint32_t base = 0;
		if (n) {
			if (lane == 0) base = atomicAdd(n_cand, n);
			base = __shfl(base, 0);
			for (uint32_t i = lane; i < n; i += 64) if (base + i < cand_cap) { cand[base + i] = stage[i]; if (cand_cnt_out) cand_cnt_out[base + i] = stage_v[i]; }
		}
		__syncthreads();
		if (lane == 0) ctr[1] = 0;
		__syncthreads();
	};
	auto bump = [&](uint32_t c) {
		const uint32_t sh = (c % PER) * BITS;
		const uint32_t old = atomicAdd(&cnt[c / PER], 1u << sh);
		if (((old >> sh) & MASK) == 0) {
			const uint32_t pos = atomicAdd(&ctr[0], 1u);
			if (pos < PF2_TL) tl[pos] = c;
		}
	};

	const uint32_t n_iter = sel ? (*n_sel_dev < n_list ? *n_sel_dev : n_list) : n_list;
	for (uint32_t it = blockIdx.x; it < n_iter; it += gridDim.x) {
		const uint32_t li = sel ? sel[it] : it;
		const uint32_t q = qlist ? qlist[li] : li;
		const uint64_t b = qoff[q];
		const uint32_t len = (uint32_t)(qoff[q + 1] - b), E = qemac[q];
		const uint32_t stride = plan[q] & 255u, px_ = BHIP_PLAN_X(plan[q]), pn_ = BHIP_PLAN_NEED(plan[q]);
		const uint32_t need = pn_ > px_ ? pn_ - px_ : 0u;      // (strict counting: without the words that vote through expansions)
		const bool takeall = px_ && !need;
		(void)E;
		if (len >= (uint32_t)K) {
			const uint32_t nwords = (len - K) / stride + 1;
			for (uint32_t base = 0; base < nwords; base += 64) {
				const uint32_t j = base + lane, p = j * stride;
				uint32_t w = 0, ok = j < nwords;
				if (ok) for (int k = 0; k < K; ++k) {
					const uint32_t c = qcodes[b + p + k];
					ok &= (c - 1u) < 4u;
					w = (w << 2) | ((c - 1u) & 3u);
				}
				w &= wmask;
				unsigned long long beg = 0; uint32_t n = 0;
				if (ok) bhip_acx_range(acx, w, beg, n);
				my_ent += n;
				unsigned long long longm = __ballot(n > 32);
				if (n <= 32) {
					uint32_t e = 0;
					for (; e + 4 <= n; e += 4) {   // four independent loads in flight
						const uint32_t c0 = bhip_acx_clump(acx.rec, beg + e), c1 = bhip_acx_clump(acx.rec, beg + e + 1), c2 = bhip_acx_clump(acx.rec, beg + e + 2), c3 = bhip_acx_clump(acx.rec, beg + e + 3);
						bump(c0); bump(c1); bump(c2); bump(c3);
					}
					for (; e < n; ++e) bump(bhip_acx_clump(acx.rec, beg + e));
				}
				while (longm) {
					const int src = __builtin_ctzll(longm);
					longm &= longm - 1;
					const unsigned long long lb = __shfl(beg, src); const uint32_t ln = __shfl(n, src);
					for (uint32_t e = lane; e < ln; e += 64) bump(bhip_acx_clump(acx.rec, lb + e));
				}
			}
		}
		__syncthreads();
		const uint32_t mmatch = need ? need - 1 : 0;      // candidate iff count >= need (count > 0 when no words are guaranteed)
		const uint32_t nt = ctr[0];
		if (takeall) {      // nothing of the guarantee is visible to strict counting: every clump
			for (uint32_t i = lane; i < nw32; i += 64) cnt[i] = 0;
			for (uint32_t c = lane; c < n_clumps; c += 64) push(li, c, 0);
		} else if (nt <= PF2_TL) {
			for (uint32_t i = lane; i < nt; i += 64) {
				const uint32_t c = tl[i], sh = (c % PER) * BITS;
				const uint32_t v = (cnt[c / PER] >> sh) & MASK;
				atomicAnd(&cnt[c / PER], ~(MASK << sh));
				if (v > mmatch) push(li, c, v);
			}
		} else {   // touched list overflowed: dense pass
			for (uint32_t i = lane; i < nw32; i += 64) {
				const uint32_t word = cnt[i];
				if (word) {
					cnt[i] = 0;
					for (uint32_t j = 0; j < PER; ++j) { const uint32_t v = (word >> (j * BITS)) & MASK; if (v > mmatch && i * PER + j < n_clumps) push(li, i * PER + j, v); }
				}
			}
		}
		for (uint32_t i = lane; i < n_bad && !takeall; i += 64) push(li, bad[i], 0xFFFFFFFFu);          // burst.c:4136-4138, 4282-4283 (with every clump taken they are in already)
		__syncthreads();
		if (lane == 0) ctr[0] = 0;
		if (ctr[1] >= PF2_STAGE / 2) flush(); else __syncthreads();
	}
	flush();
	if (ent_read && my_ent) atomicAdd(ent_read, my_ent);
}
template __global__ void k_prefilter_wave<uint8_t>(const uint8_t *, const uint64_t *, const uint16_t *, const uint32_t *, uint32_t,
	BhipAcxView, int, uint32_t, const uint32_t *, uint32_t, uint2 *, uint32_t *, uint32_t *, uint32_t, unsigned long long *, const uint32_t *,
	const uint32_t *, const uint32_t *);
template __global__ void k_prefilter_wave<uint16_t>(const uint8_t *, const uint64_t *, const uint16_t *, const uint32_t *, uint32_t,
	BhipAcxView, int, uint32_t, const uint32_t *, uint32_t, uint2 *, uint32_t *, uint32_t *, uint32_t, unsigned long long *, const uint32_t *,
	const uint32_t *, const uint32_t *);

// ------------------------------------------------------------------------------------------------
// Prefilter, hashed variant: FOUR queries per wave (16 lanes each), per-query open-addressing table in LDS instead of
// dense per-clump counters, so LDS use no longer depends on the database size (RefSeq-scale DBs have millions of
// clumps) and 4-6x more queries are in flight per CU -- the kernel is bound by the latency of the random .acx list
// reads, not by arithmetic.  Slot = (clump+1) << 8 | count (clump ids are < 2^24 by the .acx format, burst.c:3509;
// counts <= 255 is guaranteed by the seed plan).  New keys go to a per-query touched list; the final pass reads and
// clears only touched slots.  A query that overflows its table or list is handed to the dense kernel (sel list).
// ------------------------------------------------------------------------------------------------
#define PFH_HT 1024u
#define PFH_TL 448u
#define PFH_STAGE 512u
__global__ __launch_bounds__(64) void k_prefilter_hash(
		const uint8_t *__restrict__ qcodes, const uint64_t *__restrict__ qoff, const uint16_t *__restrict__ qemac,
		const uint32_t *__restrict__ qlist, uint32_t n_list,
		BhipAcxView acx, int K,
		const uint32_t *__restrict__ bad, uint32_t n_bad,
		uint2 *__restrict__ cand, uint32_t *__restrict__ cand_cnt_out, uint32_t *__restrict__ n_cand, uint32_t cand_cap,
		unsigned long long *__restrict__ ent_read, const uint32_t *__restrict__ plan,
		uint32_t *__restrict__ fb_list, uint32_t *__restrict__ n_fb) {
	__shared__ uint32_t s_tab[4][PFH_HT];
	__shared__ uint16_t s_tl[4][PFH_TL];
	__shared__ uint2 s_stage[PFH_STAGE];
	__shared__ uint32_t s_stage_v[PFH_STAGE];
	__shared__ uint32_t s_ctr[8];           // [g] touched count of group g, [4] staged, [5+..] unused
	__shared__ uint32_t s_ovf[4];
	const uint32_t lane = threadIdx.x, g = lane >> 4, gl = lane & 15;
	const uint32_t wmask = K == 16 ? 0xFFFFFFFFu : ((1u << (2 * K)) - 1u);
	for (uint32_t i = lane; i < 4 * PFH_HT; i += 64) (&s_tab[0][0])[i] = 0;
	if (lane < 8) s_ctr[lane] = 0;
	if (lane < 4) s_ovf[lane] = 0;
	__syncthreads();
	unsigned long long my_ent = 0;

	auto push = [&](uint32_t li, uint32_t c, uint32_t v) {
		const uint32_t pos = atomicAdd(&s_ctr[4], 1u);
		if (pos < PFH_STAGE) { s_stage[pos] = make_uint2(li, c); if (cand_cnt_out) s_stage_v[pos] = v; }
		else {
			const uint32_t gp = atomicAdd(n_cand, 1u);
			if (gp < cand_cap) { cand[gp] = make_uint2(li, c); if (cand_cnt_out) cand_cnt_out[gp] = v; }
		}
	};
	auto flush = [&]() {
		__syncthreads();
		const uint32_t n = s_ctr[4] < PFH_STAGE ? s_ctr[4] : PFH_STAGE;
		uint32_t base = 0;
		if (n) {
			if (lane == 0) base = atomicAdd(n_cand, n);
			base = __shfl(base, 0);
			for (uint32_t i = lane; i < n; i += 64) if (base + i < cand_cap) { cand[base + i] = s_stage[i]; if (cand_cnt_out) cand_cnt_out[base + i] = s_stage_v[i]; }
		}
		__syncthreads();
		if (lane == 0) s_ctr[4] = 0;
		__syncthreads();
	};
	// insert-or-increment clump c in the table of group tg
	auto bump = [&](uint32_t tg, uint32_t c) {
		const uint32_t key = (c + 1u) << 8;
		uint32_t slot = (c * 0x9E3779B1u) >> (32 - 10);
		uint32_t *tab = s_tab[tg];
		for (uint32_t probes = 0; probes < PFH_HT; ++probes, slot = (slot + 1) & (PFH_HT - 1)) {
			uint32_t old = tab[slot];
			if (old == 0) {
				old = atomicCAS(&tab[slot], 0u, key | 1u);
				if (old == 0) {   // new key
					const uint32_t pos = atomicAdd(&s_ctr[tg], 1u);
					if (pos < PFH_TL) s_tl[tg][pos] = (uint16_t)slot; else s_ovf[tg] = 1;
					return;
				}
			}
			if ((old & 0xFFFFFF00u) == key) { atomicAdd(&tab[slot], 1u); return; }
		}
		s_ovf[tg] = 1;
	};

	const uint32_t n_quads = (n_list + 3) >> 2;
	for (uint32_t quad = blockIdx.x; quad < n_quads; quad += gridDim.x) {
		const uint32_t li = quad * 4 + g;
		const bool live = li < n_list;
		uint32_t q = 0, len = 0, stride = 1, need = 0, nwords = 0;
		uint64_t b = 0;
		if (live) {
			q = qlist ? qlist[li] : li;
			b = qoff[q];
			len = (uint32_t)(qoff[q + 1] - b);
			if (len >= (uint32_t)K) {
				stride = plan[q] & 255u;      // the host keeps (len-K)/stride + 1 <= 255 (8-bit counts)
				{ const uint32_t px_ = BHIP_PLAN_X(plan[q]), pn_ = BHIP_PLAN_NEED(plan[q]); need = pn_ > px_ ? pn_ - px_ : 0u; if (px_ && !need) s_ovf[g] = 1; }      // (strict counting; nothing left of the guarantee: the dense kernels take every clump)
				nwords = (len - K) / stride + 1;
			}
		}
		uint32_t maxw = nwords;
		#pragma unroll
		for (int o = 32; o >= 1; o >>= 1) { const uint32_t t = __shfl_xor(maxw, o); maxw = t > maxw ? t : maxw; }
		for (uint32_t base = 0; base < maxw; base += 16) {
			const uint32_t j = base + gl, p = j * stride;
			uint32_t w = 0, ok = live && j < nwords;
			if (ok) for (int k = 0; k < K; ++k) {
				const uint32_t c = qcodes[b + p + k];
				ok &= (c - 1u) < 4u;
				w = (w << 2) | ((c - 1u) & 3u);
			}
			w &= wmask;
			unsigned long long beg = 0; uint32_t n = 0;
			if (ok) bhip_acx_range(acx, w, beg, n);
			my_ent += n;
			unsigned long long longm = __ballot(n > 48);
			if (n <= 48) {
				uint32_t e = 0;
				for (; e + 4 <= n; e += 4) {
					const uint32_t c0 = bhip_acx_clump(acx.rec, beg + e), c1 = bhip_acx_clump(acx.rec, beg + e + 1), c2 = bhip_acx_clump(acx.rec, beg + e + 2), c3 = bhip_acx_clump(acx.rec, beg + e + 3);
					bump(g, c0); bump(g, c1); bump(g, c2); bump(g, c3);
				}
				for (; e < n; ++e) bump(g, bhip_acx_clump(acx.rec, beg + e));
			}
			while (longm) {   // long lists: the whole wave walks them, inserting into the owner's table
				const int src = __builtin_ctzll(longm);
				longm &= longm - 1;
				const unsigned long long lb = __shfl(beg, src); const uint32_t ln = __shfl(n, src), tg = (uint32_t)src >> 4;
				for (uint32_t e = lane; e < ln; e += 64) bump(tg, bhip_acx_clump(acx.rec, lb + e));
			}
		}
		__syncthreads();
		// evaluate and clear the touched slots of the own group
		const uint32_t nt = s_ctr[g] < PFH_TL ? s_ctr[g] : PFH_TL;
		const uint32_t ovf = s_ovf[g];
		const uint32_t thr = need ? need - 1 : 0;
		if (live && !ovf) {
			for (uint32_t i = gl; i < nt; i += 16) {
				const uint32_t slot = s_tl[g][i], v = s_tab[g][slot];
				s_tab[g][slot] = 0;
				if ((v & 255u) > thr) push(li, (v >> 8) - 1u, v & 255u);
			}
			for (uint32_t i = gl; i < n_bad; i += 16) push(li, bad[i], 0xFFFFFFFFu);       // burst.c:4136-4138, 4282-4283
		} else if (ovf) {
			for (uint32_t i = gl; i < PFH_HT; i += 16) s_tab[g][i] = 0;
			if (live && gl == 0) { const uint32_t pos = atomicAdd(n_fb, 1u); fb_list[pos] = li; }
		}
		__syncthreads();
		if (gl == 0) { s_ctr[g] = 0; s_ovf[g] = 0; }
		if (s_ctr[4] >= PFH_STAGE / 2) flush(); else __syncthreads();
	}
	flush();
	if (ent_read && my_ent) atomicAdd(ent_read, my_ent);
}


// Prefilter with per-lane counts, in two passes over the query's .acx lists so that the wide per-lane counters are
// touched only for clumps that can matter:
//   pass 1  clump-level counts exactly as k_prefilter_hash (slot = (clump+1) << 8 | count);
//   select  clumps with count >= need become "candidates" (a clump-level count below need implies every lane is below);
//           the slot's low byte is re-used for the candidate index, all other touched slots keep their key (probe chains
//           stay intact) with a zero byte;
//   pass 2  the lists are walked again (L2-resident by now); entries of candidate clumps add their 16-bit lane mask into
//           sixteen 8-bit lane counters (two 64-bit LDS atomics, ~3 % of the entries);
//   emit    (list position, reference lane) TASKS for lanes with count >= need -> k_myers_prefix_task.
// More candidate clumps in one query than the lane counters hold (24 with the 512-slot table, 80 above): the surplus clumps are emitted as clump-level pairs (16-lane kernel).
template <int HTB>
__global__ __launch_bounds__(64) void k_prefilter_mask(
		const uint2 *__restrict__ ranges, const uint2 *__restrict__ hdr, uint32_t W16, uint32_t n_list,
		const uint32_t *__restrict__ ent,   // 4-byte (clump, lane-set code) records
		const uint32_t *__restrict__ bad, uint32_t n_bad, const uint32_t *__restrict__ clump_len, uint32_t tot_refs,
		uint2 *__restrict__ tasks, uint32_t *__restrict__ n_tasks, uint32_t task_cap,
		unsigned long long *__restrict__ ent_read,
		uint32_t *__restrict__ fb_list, uint32_t *__restrict__ n_fb,
		unsigned long long *__restrict__ unit_sum, unsigned long long *__restrict__ col_sum, unsigned long long *__restrict__ qlen_sum,
		uint2 *__restrict__ pairs, uint32_t *__restrict__ n_pairs, uint32_t pair_cap) {

	__shared__ uint32_t s_tab[4][(1u << HTB)];
	__shared__ uint16_t s_tl[4][(1u << (HTB - 1))];
	constexpr uint32_t CAND = HTB <= 9 ? 24u : 80u;      // candidate clumps per query with lane counters (LDS: 20 B each)
	__shared__ unsigned long long s_cc[4][CAND][2];
	__shared__ uint32_t s_cclump[4][CAND];
	__shared__ uint2 s_stage[PFM_STAGE];
	__shared__ uint32_t s_ctr[12];          // [g] touched count, [4] staged, [5+g] candidates of group g
	__shared__ uint32_t s_ovf[4];
	__shared__ uint32_t s_dummy[64];
	__shared__ uint16_t s_lut[256];         // lane-set code -> lane mask
	const uint32_t lane = threadIdx.x, g = lane >> 4, gl = lane & 15;
	s_dummy[lane] = 0;
	for (uint32_t i = lane; i < 256; i += 64) s_lut[i] = (uint16_t)bhip_lane_code_mask(i);
	for (uint32_t i = lane; i < 4 * (1u << HTB); i += 64) (&s_tab[0][0])[i] = 0;
	for (uint32_t i = lane; i < 4 * CAND * 2; i += 64) (&s_cc[0][0][0])[i] = 0;
	if (lane < 12) s_ctr[lane] = 0;
	if (lane < 4) s_ovf[lane] = 0;
	__syncthreads();
	unsigned long long my_ent = 0, my_units = 0, my_cols = 0, my_qlen = 0;
	uint32_t sink = 0;                      // see bhip_acx_raw_or_pad (bhip_internal.h)
#ifdef PFM_PROF
	unsigned long long my_t[8] = {0,0,0,0,0,0,0,0}, t_last = wall_clock64();
#endif

	auto push = [&](uint32_t li, uint32_t refIx) {
		const uint32_t pos = atomicAdd(&s_ctr[4], 1u);
		if (pos < PFM_STAGE) s_stage[pos] = make_uint2(li, refIx);
		else { const uint32_t gp = atomicAdd(n_tasks, 1u); if (gp < task_cap) tasks[gp] = make_uint2(li, refIx); }
	};
	auto flush = [&]() {
		__syncthreads();
		const uint32_t n = s_ctr[4] < PFM_STAGE ? s_ctr[4] : PFM_STAGE;
		uint32_t base = 0;
		if (n) {
			if (lane == 0) base = atomicAdd(n_tasks, n);
			base = __shfl(base, 0);
			for (uint32_t i = lane; i < n; i += 64) if (base + i < task_cap) tasks[base + i] = s_stage[i];
		}
		__syncthreads();
		if (lane == 0) s_ctr[4] = 0;
		__syncthreads();
	};
	uint32_t tcnt = 0;                      // touched slots of this lane's own group (replicated in its 16 lanes)
	auto lanes_add = [&](uint32_t tg, uint32_t c, uint32_t code) {   // pass 2: only candidate clumps have a non-zero low byte
		const uint32_t key = (c + 1u) << 8;
		uint32_t slot = (c * 0x9E3779B1u) >> (32 - HTB);
		const uint32_t *tab = s_tab[tg];
		for (uint32_t probes = 0; probes < (1u << HTB); ++probes, slot = (slot + 1) & ((1u << HTB) - 1)) {
			const uint32_t v = tab[slot];
			if (v == 0) return;
			if ((v & 0xFFFFFF00u) == key) {
				const uint32_t ci = v & 255u;
				if (ci) {
					const uint32_t mask = s_lut[code & 255u];
					if (mask & 0xFFu) atomicAdd(&s_cc[tg][ci - 1][0], spread8(mask & 0xFFu));
					if (mask >> 8) atomicAdd(&s_cc[tg][ci - 1][1], spread8(mask >> 8));
				}
				return;
			}
		}
	};

	const uint32_t n_quads = (n_list + 3) >> 2;
	// (need, words, length) and the first 16 list ranges of the next quad are fetched one iteration ahead (k_seed_ranges
	// produced them), so the only exposed memory round trip per quad is the list records themselves
	uint2 hd_n = make_uint2(0, 0), rg_n = make_uint2(0, 0);
	if (blockIdx.x * 4 + g < n_list) { hd_n = hdr[blockIdx.x * 4 + g]; if (gl < W16) rg_n = ranges[(size_t)(blockIdx.x * 4 + g) * W16 + gl]; }
	for (uint32_t quad = blockIdx.x; quad < n_quads; quad += gridDim.x) {
		const uint32_t li = quad * 4 + g;
		const bool live = li < n_list;
		tcnt = 0;
		const uint2 hd = hd_n, rg = rg_n;
		{
			const uint32_t li_n = (quad + gridDim.x) * 4 + g;
			hd_n = make_uint2(0, 0); rg_n = make_uint2(0, 0);
			if (quad + gridDim.x < n_quads && li_n < n_list) { hd_n = hdr[li_n]; if (gl < W16) rg_n = ranges[(size_t)li_n * W16 + gl]; }
		}
		const uint32_t need = hd.x & 0xFFFFu, nwords = live ? hd.x >> 16 : 0u, len = hd.y & 0xFFFu;
		uint32_t maxw = nwords;
		#pragma unroll
		for (int o = 32; o >= 1; o >>= 1) { const uint32_t t = __shfl_xor(maxw, o); maxw = t > maxw ? t : maxw; }
		auto word_range = [&](uint32_t j, unsigned long long &beg, uint32_t &n) {
			uint2 r = make_uint2(0, 0);
			if (live && j < nwords) r = ranges[(size_t)li * W16 + j];
			beg = (unsigned long long)r.x | (unsigned long long)(r.y >> 24) << 32; n = r.y & 0xFFFFFFu;
		};
		PFM_T(0);
		// ---- pass 1: clump-level counts.  The (up to 16) lists of a query are walked as ONE flattened record stream by the
		// 16 lanes of its group: record i belongs to the list k with excl[k] <= i < excl[k+1] (4-step search over the group's
		// exclusive prefix sums), so the lanes stay busy whatever the individual list lengths.  Blocks of 4 rounds (64
		// records per query) are loaded together and updated in lock step; the first PFM_RB blocks stay in registers for pass 2.
		const unsigned long long beg = live ? ((unsigned long long)rg.x | (unsigned long long)(rg.y >> 24) << 32) : 0ull;
		const uint32_t n0 = live ? rg.y & 0xFFFFFFu : 0u;
		my_ent += n0;
		auto group_scan = [&](uint32_t n, uint32_t &T, uint32_t &excl) {
			uint32_t ps = n;
			#pragma unroll
			for (uint32_t o = 1; o < 16; o <<= 1) { const uint32_t t = __shfl_up(ps, o, 16); if (gl >= o) ps += t; }
			T = __shfl(ps, 15, 16);
			excl = ps - n;
		};
		auto wave_blocks = [&](uint32_t T) -> uint32_t {
			uint32_t m = T, t;
			t = __shfl_xor(m, 16); m = t > m ? t : m;
			t = __shfl_xor(m, 32); m = t > m ? t : m;
			return (m + 63) >> 6;
		};
		auto load4 = [&](uint32_t ex, unsigned long long dl, uint32_t T, uint32_t b, uint2 (&rec)[4]) {
			#pragma unroll
			for (uint32_t u = 0; u < 4; ++u) {
				const uint32_t i = (b * 4 + u) * 16 + gl;
				uint32_t kk = 0;
				kk += __shfl(ex, 8, 16) <= i ? 8u : 0u;
				kk += __shfl(ex, kk + 4, 16) <= i ? 4u : 0u;
				kk += __shfl(ex, kk + 2, 16) <= i ? 2u : 0u;
				kk += __shfl(ex, kk + 1, 16) <= i ? 1u : 0u;
				const unsigned long long addr = __shfl(dl, kk, 16) + i;
				const uint32_t v = bhip_acx_raw_or_pad(ent, addr, i < T, hdr, sink);
				rec[u] = make_uint2(v == BHIP_REC_PAD ? 0xFFFFFFFFu : v & 0xFFFFFFu, v >> 24);      // .y = lane-set code
			}
		};
		auto bump_block = [&](uint2 (&rec)[4]) {
			const uint32_t c[4] = {rec[0].x, rec[1].x, rec[2].x, rec[3].x};
			const bool valid[4] = {c[0] != 0xFFFFFFFFu, c[1] != 0xFFFFFFFFu, c[2] != 0xFFFFFFFFu, c[3] != 0xFFFFFFFFu};
			uint32_t slot[4]; bool ins[4], fail;
			pfm_bump4<HTB>(s_tab[g], &s_dummy[lane], c, valid, slot, ins, fail);
			if (fail) s_ovf[g] = 1;
			#pragma unroll
			for (int u = 0; u < 4; ++u) {
				rec[u].y |= slot[u] << 16;
				const uint32_t m16 = (uint32_t)(__ballot(ins[u]) >> (lane & 48u)) & 0xFFFFu;
				if (ins[u]) {
					const uint32_t pos = tcnt + __popc(m16 & ((1u << gl) - 1u));
					if (pos < (1u << (HTB - 1))) s_tl[g][pos] = (uint16_t)slot[u]; else s_ovf[g] = 1;
				}
				tcnt += __popc(m16);
			}
		};
		uint32_t T0, ex0;
		group_scan(n0, T0, ex0);
		const unsigned long long dl0 = beg - ex0;
		const uint32_t nblk0 = wave_blocks(T0);
		uint2 rc[PFM_RB][4];           // .x = clump, .y = lane-set code | slot << 16
		#pragma unroll
		for (uint32_t b = 0; b < PFM_RB; ++b) if (b < nblk0) load4(ex0, dl0, T0, b, rc[b]);
		PFM_T(6);
		#pragma unroll
		for (uint32_t b = 0; b < PFM_RB; ++b) if (b < nblk0) bump_block(rc[b]);
		PFM_T(7);
		for (uint32_t b = PFM_RB; b < nblk0; ++b) { uint2 rec[4]; load4(ex0, dl0, T0, b, rec); bump_block(rec); }
		for (uint32_t base = 16; base < maxw; base += 16) {       // queries with more than 16 sampled words: further chunks, not cached
			unsigned long long xb; uint32_t xn, T, ex;
			word_range(base + gl, xb, xn);
			my_ent += xn;
			group_scan(xn, T, ex);
			const uint32_t nb = wave_blocks(T);
			for (uint32_t b = 0; b < nb; ++b) { uint2 rec[4]; load4(ex, xb - ex, T, b, rec); bump_block(rec); }
		}
		__syncthreads();
		PFM_T(2);
		// ---- select candidates
		const uint32_t nt = tcnt < (1u << (HTB - 1)) ? tcnt : (1u << (HTB - 1));
		const uint32_t ovf = s_ovf[g];
		const uint32_t thr = need ? need : 1u;          // a lane (hence its clump) is a candidate iff count >= max(need, 1)
		if (live && !ovf) {
			for (uint32_t i = gl; i < nt; i += 16) {
				const uint32_t slot = s_tl[g][i], v = s_tab[g][slot];
				uint32_t tag = 0;
				if ((v & 255u) >= thr) {
					const uint32_t ci = atomicAdd(&s_ctr[5 + g], 1u);
					const uint32_t c = (v >> 8) - 1u;
					if (ci < CAND) { tag = ci + 1; s_cclump[g][ci] = c; }
					else {   // too many candidate clumps for the lane counters: hand the clump to the 16-lane kernel
						const uint32_t gp = atomicAdd(n_pairs, 1u);
						if (gp < pair_cap) pairs[gp] = make_uint2(li, c);
					}
				}
				s_tab[g][slot] = (v & 0xFFFFFF00u) | tag;
			}
		}
		__syncthreads();
		PFM_T(3);
		// ---- pass 2: lane counters of the candidate clumps
		const uint32_t ncand = s_ctr[5 + g] < CAND ? s_ctr[5 + g] : CAND;
		const bool mine = live && !ovf && ncand > 0;
		if (__any(mine)) {
			#pragma unroll
			for (uint32_t b = 0; b < PFM_RB; ++b) if (b < nblk0) {
				uint32_t ci[4];
				#pragma unroll
				for (int u = 0; u < 4; ++u) ci[u] = s_tab[g][rc[b][u].y >> 16];      // slot 0 for padding records: harmless read
				#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const uint32_t tag = ci[u] & 255u, mask = s_lut[rc[b][u].y & 255u];
					if (mine && rc[b][u].x != 0xFFFFFFFFu && tag) {
						if (mask & 0xFFu) atomicAdd(&s_cc[g][tag - 1][0], spread8(mask & 0xFFu));
						if (mask >> 8) atomicAdd(&s_cc[g][tag - 1][1], spread8(mask >> 8));
					}
				}
			}
			for (uint32_t b = PFM_RB; b < nblk0; ++b) {
				uint2 rec[4];
				load4(ex0, dl0, T0, b, rec);
				#pragma unroll
				for (int u = 0; u < 4; ++u) if (mine && rec[u].x != 0xFFFFFFFFu) lanes_add(g, rec[u].x, rec[u].y);
			}
			for (uint32_t base = 16; base < maxw; base += 16) {
				unsigned long long xb; uint32_t xn, T, ex;
				word_range(base + gl, xb, xn);
				group_scan(xn, T, ex);
				const uint32_t nb = wave_blocks(T);
				for (uint32_t b = 0; b < nb; ++b) {
					uint2 rec[4];
					load4(ex, xb - ex, T, b, rec);
					#pragma unroll
					for (int u = 0; u < 4; ++u) if (mine && rec[u].x != 0xFFFFFFFFu) lanes_add(g, rec[u].x, rec[u].y);
				}
			}
		}
		__syncthreads();
		PFM_T(4);
		// ---- emit tasks, clear
		if (live && !ovf) {
			for (uint32_t i = gl; i < ncand; i += 16) {
				const uint32_t c = s_cclump[g][i];
				const unsigned long long lo = s_cc[g][i][0], hi = s_cc[g][i][1];
				s_cc[g][i][0] = 0; s_cc[g][i][1] = 0;
				uint32_t any = 0;
				#pragma unroll
				for (uint32_t z = 0; z < 16; ++z) {
					const uint32_t v = (uint32_t)(((z < 8 ? lo : hi) >> (8 * (z & 7))) & 255u);
					const uint32_t refIx = c * 16 + z;
					if (v >= thr && refIx < tot_refs) { push(li, refIx); any = 1; }
				}
				if (any) { ++my_units; my_cols += clump_len[c]; my_qlen += len; }
			}
			for (uint32_t i = gl; i < nt; i += 16) s_tab[g][s_tl[g][i]] = 0;
			for (uint32_t i = gl; i < n_bad; i += 16) {        // burst.c:4136-4138, 4282-4283: every lane of the ambiguous clumps
				const uint32_t c = bad[i];
				for (uint32_t z = 0; z < 16; ++z) if (c * 16 + z < tot_refs) push(li, c * 16 + z);
				++my_units; my_cols += clump_len[c]; my_qlen += len;
			}
		} else if (ovf) {
			for (uint32_t i = gl; i < (1u << HTB); i += 16) s_tab[g][i] = 0;
			for (uint32_t i = gl; i < CAND; i += 16) { s_cc[g][i][0] = 0; s_cc[g][i][1] = 0; }
			if (live && gl == 0) { const uint32_t pos = atomicAdd(n_fb, 1u); fb_list[pos] = li; }
		}
		__syncthreads();
		if (gl == 0) { s_ctr[g] = 0; s_ctr[5 + g] = 0; s_ovf[g] = 0; }
		if (s_ctr[4] >= PFM_STAGE / 2) flush(); else __syncthreads();
		PFM_T(5);
	}
	flush();
#ifdef PFM_PROF
	if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_pfm_prof[i], my_t[i]);
#endif
	if (ent_read && my_ent) atomicAdd(ent_read, my_ent);
	if (my_units) { atomicAdd(unit_sum, my_units); atomicAdd(col_sum, my_cols); atomicAdd(qlen_sum, my_qlen); }
	if (n_list == 0xFFFFFFFFu) fb_list[0] = sink;       // never: keeps the record loads unconditional
}

template __global__ void k_prefilter_mask<9>(const uint2 *, const uint2 *, uint32_t, uint32_t, const uint32_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t,
	uint2 *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, unsigned long long *, uint2 *, uint32_t *, uint32_t);
template __global__ void k_prefilter_mask<10>(const uint2 *, const uint2 *, uint32_t, uint32_t, const uint32_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t,
	uint2 *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, unsigned long long *, uint2 *, uint32_t *, uint32_t);
template __global__ void k_prefilter_mask<11>(const uint2 *, const uint2 *, uint32_t, uint32_t, const uint32_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t,
	uint2 *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, unsigned long long *, uint2 *, uint32_t *, uint32_t);

BHIP_INST_PFCW(2, 0) BHIP_INST_PFCW(2, 1)
