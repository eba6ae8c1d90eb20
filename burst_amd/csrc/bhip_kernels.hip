// burst_amd/csrc/bhip_kernels.hip -- gfx950 (MI355X) kernels of the BURST alignment hot path.
//
// What the reference computes with 16-lane SSE rows (burst.c:1003-1204 aded_*, 713-886 reScoreM_*,
// 3238-3282 postScour*) is re-designed here for 64-wide wavefronts:
//
//   k_transpose_refs : database set-up (the lane-major reference layout; the accelerator kernels are in bhip_acx.hip).
//   k_pack_queries, k_build_peq : 4-bit packed queries; per query 16 match bit-vectors (one per reference symbol).
//   k_seed_ranges, k_prefilter_cf / k_prefilter_mask : sampled words -> .acx list ranges -> per-query counts in LDS,
//                      resolved to single reference lanes -> (query, lane) tasks, split by a lower bound on their
//                      edit distance.  k_prefilter_hash / k_prefilter_wave / k_prefilter: clump-level fallbacks.
//   k_myers_prefix_task, k_task_filter, k_myers_window : two-stage Myers/Hyyro bit-parallel semi-global edit distance, one
//                      thread per (query, reference lane) (carry chains via v_add_co / v_addc_co); k_myers_prefix and
//                      k_myers are the 16-lane (one aded_mat16 call per group) variants for clump-level units.
//   k_rescore_classify, k_rescore_reg, k_rescore : one thread per surviving hit; 3-plane (score, gapQ, gapR) DP restricted
//                      to the diagonals that can reach a minimal end cell, with the reference's exact tie-breaks.
//
// No MFMA: the recurrence is integer min-plus / bit logic (VALU + LDS bound).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "burst_hip.h"
#include "bhip_internal.h"

// ------------------------------------------------------------------------------------------------
// DB upload: byte transpose of the .edx clump area
// src: for clump c, rows j = 0..ceil(L/2)-1 of 16 bytes (byte z = lane z, nibbles = positions 2j, 2j+1)
// dst_lane: for clump c, lane z, chunk t: 16 bytes, byte i = src row (16t+i) byte z  (zero beyond the clump)
// ------------------------------------------------------------------------------------------------
__global__ void k_transpose_refs(const uint8_t *__restrict__ src, const uint64_t *__restrict__ src_off,
                                 const uint32_t *__restrict__ clump_len, const uint64_t *__restrict__ dst_off,
                                 uint32_t n_clumps, uint4 *__restrict__ dst_lane) {
	// one 16-thread group per (clump, chunk); grid-stride over clumps.  ONE layout: `dst_lane` keeps each lane's chunks
	// contiguous inside the clump's area (the one-thread-per-lane kernels stream 16 B after 16 B of one cache line instead of
	// touching a new 128-byte line for every 32 columns).  Rounds 1-3 kept a second, chunk-interleaved copy for the clump-level
	// kernels (16 threads = 16 lanes reading 256 contiguous bytes); those kernels execute hundreds of instructions per 16-byte
	// load and read the lane-major words just as well -- the copy was 31 GB of the metric's database for nothing
	const uint32_t z = threadIdx.x & 15, g = threadIdx.x >> 4, gpb = blockDim.x >> 4;
	for (uint32_t c = blockIdx.x; c < n_clumps; c += gridDim.x) {
		const uint32_t L = clump_len[c], nrows = (L + 1) >> 1, nchunks = (L + 31) >> 5;
		const uint8_t *s = src + src_off[c] * 16;
		for (uint32_t t = g; t < nchunks; t += gpb) {
			uint32_t w[4] = {0, 0, 0, 0};
			#pragma unroll
			for (int i = 0; i < 16; ++i) {
				uint32_t row = 16 * t + i;
				uint32_t b = row < nrows ? s[(uint64_t)row * 16 + z] : 0u;
				w[i >> 2] |= b << (8 * (i & 3));
			}
			dst_lane[dst_off[c] * 16 + (uint64_t)z * nchunks + t] = make_uint4(w[0], w[1], w[2], w[3]);
		}
	}
}


// ------------------------------------------------------------------------------------------------
// Match bit-vectors (DIAGSC_MAT16, burst.c:700: SCOREFAST[qLet] shuffled by the reference symbol).
// The query is TOP-aligned in its NW x 32-bit vector: query symbol i lives at bit i + (32*NW - len), so the
// last symbol is always bit 31 of word NW-1 and k_myers needs no per-query bit index.  The low 32*NW - len
// "filler rows" match every symbol and start with vertical delta 0, which keeps them identically 0 = the
// free-start boundary row D[0][x] = 0 of the reference (burst.c:4052 calloc'd row 0).
// peq[(li*16 + c)*NW + w] bit k = 1 iff k is a filler row or cost(query[32w+k-shift], c) == 0.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_build_peq(const uint8_t *__restrict__ qcodes, const uint64_t *__restrict__ qoff,
                            const uint32_t *__restrict__ qlist, uint32_t n_list, int NW, int prefix_len,
                            BhipMatchMask mm, uint32_t *__restrict__ peq, const uint32_t *__restrict__ qpack, uint32_t qw, uint32_t n_rows) {
	// n_rows: 16, or 5 when no reference holds a symbol beyond A/C/G/T (codes 1..4; 0 = pad): the rows 5..15 of every table would be
	// written -- 11/16 of 1.3 GB per 2 M-read batch -- and never read.  The table keeps its 16-row stride.
	// One thread per (query, word) reads its 32 symbols once and produces the 16 symbol rows of that word.  A block owns
	// QB = 256 / NW whole queries; the rows go through LDS so that the block's 16*NW*QB output words leave as one
	// contiguous, fully coalesced stream (the natural per-thread stores hit 16 B pieces of 16 different lines).
	__shared__ uint32_t s_mm[16];
	__shared__ uint32_t s_out[256 * 16 + 256];
	const uint32_t tid = threadIdx.x;
	if (tid < 16) s_mm[tid] = mm.m[tid];
	const uint32_t QB = 256u / (uint32_t)NW, per_q = 16u * (uint32_t)NW;
	const uint32_t lq = tid / (uint32_t)NW, w = tid % (uint32_t)NW;
	for (uint32_t q0 = blockIdx.x * QB; q0 < n_list; q0 += gridDim.x * QB) {
		__syncthreads();
		const uint32_t li = q0 + lq;
		if (lq < QB && li < n_list) {
			const uint32_t q = qlist ? qlist[li] : li;
			const uint64_t b = qoff[q];
			int len = (int)(qoff[q + 1] - b);
			if (prefix_len > 0 && len > prefix_len) len = prefix_len;     // table of the first prefix_len symbols only (k_myers_prefix)
			const int shift = 32 * NW - len;
			// X[j] = 16-bit match mask of symbol j (low half) and of symbol j + 16 (high half); a 16 x 16 bit transpose done on
			// both halves at once turns the 16 words into the 16 symbol rows (bit k of row c = bit c of the mask of symbol k)
			uint32_t row[16];
			if (qpack) {   // 4-bit packed symbols: five dwords cover the 32 positions at any alignment
				const int pb = 32 * (int)w - shift, j0 = pb >> 3;
				const uint32_t *qp = qpack + (uint64_t)q * qw;
				uint32_t D[5], dd[4];
				#pragma unroll
				for (int i = 0; i < 5; ++i) D[i] = (j0 + i >= 0 && (uint32_t)(j0 + i) < qw) ? qp[j0 + i] : 0u;
				#pragma unroll
				for (int i = 0; i < 4; ++i) dd[i] = __builtin_amdgcn_alignbit(D[i + 1], D[i], 4u * ((uint32_t)pb & 7u));
				#pragma unroll
				for (int j = 0; j < 16; ++j) {
					const uint32_t lo = pb + j < 0 ? 0xFFFFu : s_mm[(dd[j >> 3] >> (4 * (j & 7))) & 15u];
					const uint32_t hi = pb + j + 16 < 0 ? 0xFFFFu : s_mm[(dd[2 + (j >> 3)] >> (4 * (j & 7))) & 15u];
					row[j] = lo | (hi << 16);
				}
			} else {
				#pragma unroll
				for (int j = 0; j < 16; ++j) {
					const int p0 = 32 * (int)w + j - shift, p1 = p0 + 16;
					const uint32_t lo = p0 < 0 ? 0xFFFFu : s_mm[qcodes[b + p0] & 15], hi = p1 < 0 ? 0xFFFFu : s_mm[qcodes[b + p1] & 15];
					row[j] = lo | (hi << 16);
				}
			}
			#pragma unroll
			for (int st = 0; st < 4; ++st) {
				const int j = 8 >> st;
				const uint32_t msk = st == 0 ? 0x00FF00FFu : st == 1 ? 0x0F0F0F0Fu : st == 2 ? 0x33333333u : 0x55555555u;
				#pragma unroll
				for (int k = 0; k < 16; ++k) if (!(k & j)) {
					const uint32_t t = ((row[k] >> j) ^ row[k + j]) & msk;
					row[k + j] ^= t;
					row[k] ^= t << j;
				}
			}
			#pragma unroll
			for (int c = 0; c < 16; ++c) s_out[lq * (per_q + 1) + (uint32_t)c * (uint32_t)NW + w] = row[c];     // +1: bank spread
		}
		__syncthreads();
		const uint32_t nq = n_list - q0 < QB ? n_list - q0 : QB;
		const uint32_t per_w = n_rows * (uint32_t)NW, total = nq * per_w;      // dwords written per query
		uint32_t *dst = peq + (uint64_t)q0 * per_q;
		const uint32_t st_q = 256u / per_w, st_r = 256u % per_w;
		uint32_t oq = tid / per_w, orr = tid % per_w;
		for (uint32_t idx = tid; idx < total; idx += 256) {
			dst[oq * per_q + orr] = s_out[oq * (per_q + 1) + orr];
			oq += st_q; orr += st_r;
			if (orr >= per_w) { orr -= per_w; ++oq; }
		}
	}
}

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
#ifdef PFM_PROF
__device__ unsigned long long g_pfm_prof[8];
#if PFM_PROF == 2      // without draining the memory pipeline: issue + stall time of each phase as it really runs
#define PFM_T(i) do { const unsigned long long t_ = wall_clock64(); if (lane == 0) my_t[i] += t_ - t_last; t_last = t_; } while (0)
#else
#define PFM_T(i) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = wall_clock64(); if (lane == 0) my_t[i] += t_ - t_last; t_last = t_; } while (0)
#endif
#else
#define PFM_T(i) do {} while (0)
#endif
#define PFM_STAGE 128u
#define PFM_RB 3u            // blocks of 64 records per query kept in registers between the passes
__device__ __forceinline__ unsigned long long spread8(uint32_t m8) {   // bit i of m8 -> bit 8*i
	unsigned long long x = m8;
	x = (x | (x << 28)) & 0x0000000F0000000Full;
	x = (x | (x << 14)) & 0x0003000300030003ull;
	x = (x | (x << 7)) & 0x0101010101010101ull;
	return x;
}
// Four hash-table updates in lock step (independent LDS round trips overlap).  CAS first: most updates of a
// query are first sightings of a clump, which complete in one round trip; a key hit costs one more (no-return) add.
template <int HTB>
__device__ __forceinline__ void pfm_bump4(uint32_t *tab, uint32_t *dummy, const uint32_t (&c)[4], const bool (&valid)[4], uint32_t (&slot)[4], bool (&ins)[4], bool &fail) {
	uint32_t key[4]; bool act[4];
	#pragma unroll
	for (int k = 0; k < 4; ++k) { key[k] = (c[k] + 1u) << 8; slot[k] = (c[k] * 0x9E3779B1u) >> (32 - HTB); act[k] = valid[k]; ins[k] = false; }
	bool any = valid[0] | valid[1] | valid[2] | valid[3];
	for (uint32_t probes = 0; any && probes < (1u << HTB); ++probes) {
		uint32_t old[4];
		// finished chains compare-and-swap a private dummy word with a value that never matches: no branches between the
		// four LDS round trips, so they are in flight together
		#pragma unroll
		for (int k = 0; k < 4; ++k) old[k] = atomicCAS(act[k] ? &tab[slot[k]] : dummy, act[k] ? 0u : 0xFFFFFFFFu, key[k] | 1u);
		any = false;
		#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const bool hit = act[k] && (old[k] & 0xFFFFFF00u) == key[k];
			const bool fresh = act[k] && old[k] == 0;
			const bool step = act[k] && !hit && !fresh;
			if (hit) atomicAdd(&tab[slot[k]], 1u);
			ins[k] |= fresh;
			slot[k] = step ? (slot[k] + 1) & ((1u << HTB) - 1) : slot[k];
			act[k] = step;
			any |= step;
		}
	}
	fail = any;
}
// Seed lookup for the lane-resolved prefilter: one thread per (query of the list, sampled word) turns the word into its
// .acx list range; the header carries need | words << 16 and the length.  Keeps the dependent chain
// list -> offsets -> symbols -> acx offsets out of the hash kernel (fully parallel here, four round trips there).
__global__ __launch_bounds__(256) void k_seed_ranges(
		const uint8_t *__restrict__ qcodes, const uint64_t *__restrict__ qoff, const uint32_t *__restrict__ qlist, uint32_t n_list,
		BhipAcxView acx, int K, const uint32_t *__restrict__ plan, uint32_t W16,
		uint2 *__restrict__ ranges, uint2 *__restrict__ hdr, const uint32_t *__restrict__ qpack, uint32_t qw, const uint16_t *__restrict__ qemac,
		uint4 *__restrict__ qmeta, const uint32_t *__restrict__ qsix,         // qmeta[list position] = (query entry, length | budget << 16, shared slot): one sector for the prefix sweep instead of three
		uint32_t min_need, uint32_t drop_len,                                 // the longest lists of a query are left out while `need` stays >= min_need (0: never), lists shorter than drop_len stay
		BhipAlt alt) {                                                        // compatible bases per query symbol code: expansions of ambiguous words (plan bits 24..31)
	// (grid-stride: run ahead beside another batch's sweeps, the kernel is launched with a few blocks per CU only)
	for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < (uint64_t)n_list * W16; t += (uint64_t)gridDim.x * 256) {
	const uint32_t li = (uint32_t)(t / W16), j = (uint32_t)(t % W16);
	const uint32_t q = qlist ? qlist[li] : li;
	const uint64_t b = qoff[q];
	const uint32_t len = (uint32_t)(qoff[q + 1] - b);
	uint32_t stride = 1, need = 0, nwords = 0, n_exp = 0, n_pos = 0;
	if (len >= (uint32_t)K) { const uint32_t pl = plan[q]; stride = pl & 255u; need = BHIP_PLAN_NEED(pl); n_pos = (len - K) / stride + 1; n_exp = BHIP_PLAN_X(pl) ? BHIP_PLAN_USED(pl) : 0u; nwords = n_pos + n_exp; }
	if (nwords > W16) nwords = W16;
	uint2 r = make_uint2(0, 0);
	if (n_exp) {
		// A query with expanded words (rare: plan bits 24..31; stride == K, the words do not overlap).  Slot j < n_pos is the word at j K:
		// a word of A/C/G/T as usual, an expandable one -- if the budget walk reaches it -- with the FIRST compatible base in place of its
		// ambiguous symbol; slot n_pos + e is the e-th further alternative, found by the same walk.  Symbol by symbol: this path is
		// off the critical path and taken by a handful of queries per batch.
		const uint32_t *qp = qpack + (uint64_t)q * qw;
		auto sym = [&](uint32_t i) -> uint32_t { return qpack ? (qp[i >> 3] >> (4u * (i & 7u))) & 15u : (uint32_t)qcodes[b + i]; };
		const uint32_t Ku = (uint32_t)K;
		uint32_t wj = 0xFFFFFFFFu, alt_ix = 0, amb_k = 0;      // the word this slot looks up: its number, which alternative, where its ambiguous symbol is
		uint32_t used = 0;
		if ((W16 & (W16 - 1u)) == 0u && W16 <= 64u) {
			// the slots of a query are W16 consecutive lanes of one wave: every lane classifies ITS word once, the classes go round by
			// lane reads, and each lane walks the budget over them (n_pos reads instead of n_pos x K symbol extractions per lane)
			uint32_t ak0 = 0, ex0 = 0;
			const uint32_t c0 = j < n_pos ? bhip_word_class(sym, j * Ku, Ku, alt, ak0, ex0) : 0u;
			const uint32_t mine = c0 | ex0 << 2 | ak0 << 4;
			for (uint32_t t = 0; t < n_pos; ++t) {
				const uint32_t v = (uint32_t)__shfl((int)mine, (int)t, (int)W16);
				const uint32_t c = v & 3u, ex = (v >> 2) & 3u, ak = v >> 4;
				const bool fits = c == 2u && used + ex <= BHIP_EXPAND_SLOTS;
				if (j < n_pos) { if (t == j && (c == 1u || fits)) { wj = t; alt_ix = 0; amb_k = c == 2u ? ak : 0xFFFFFFFFu; } }
				else if (wj == 0xFFFFFFFFu && fits && j - n_pos >= used && j - n_pos < used + ex) { wj = t; alt_ix = 1u + (j - n_pos - used); amb_k = ak; }
				if (fits) used += ex;
			}
			if (j >= nwords) wj = 0xFFFFFFFFu;
		} else if (j < nwords) {
			const uint32_t upto = j < n_pos ? j + 1 : n_pos;
			for (uint32_t t = 0; t < upto && wj == 0xFFFFFFFFu; ++t) {
				uint32_t ak, ex;
				const uint32_t c = bhip_word_class(sym, t * Ku, Ku, alt, ak, ex);
				const bool fits = c == 2u && used + ex <= BHIP_EXPAND_SLOTS;
				if (j < n_pos) { if (t == j && (c == 1u || fits)) { wj = t; alt_ix = 0; amb_k = c == 2u ? ak : 0xFFFFFFFFu; } }
				else if (fits && j - n_pos >= used && j - n_pos < used + ex) { wj = t; alt_ix = 1u + (j - n_pos - used); amb_k = ak; }
				if (fits) used += ex;
			}
		}
		if (wj != 0xFFFFFFFFu) {
			uint32_t w = 0;
			for (uint32_t k = 0; k < Ku; ++k) {
				const uint32_t c = sym(wj * Ku + k);
				const uint32_t base = k == amb_k ? ((uint32_t)alt.base[c] >> (2u * alt_ix)) & 3u : (c - 1u) & 3u;
				w = (w << 2) | base;
			}
			w &= Ku == 16 ? 0xFFFFFFFFu : ((1u << (2 * Ku)) - 1u);
			unsigned long long beg; uint32_t n;
			bhip_acx_range(acx, w, beg, n);
			r.x = (uint32_t)beg; r.y = n | (uint32_t)(beg >> 32) << 24;
		}
	} else if (j < nwords) {
		const uint32_t wmask = K == 16 ? 0xFFFFFFFFu : ((1u << (2 * K)) - 1u);
		const uint32_t p = j * stride;
		uint32_t w = 0, ok = 1;
		if (qpack) {   // K <= 15 symbols = at most three dwords of 4-bit codes
			const uint32_t *qp = qpack + (uint64_t)q * qw;
			const uint32_t j0 = p >> 3, sh = 4u * (p & 7u);
			const uint32_t d0 = qp[j0], d1 = j0 + 1 < qw ? qp[j0 + 1] : 0u, d2 = j0 + 2 < qw ? qp[j0 + 2] : 0u;
			const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh), hi = __builtin_amdgcn_alignbit(d2, d1, sh);
			// eight 4-bit codes -> eight 2-bit symbols, first symbol most significant, all at once (codes 1..4 = A C G T;
			// any other code in the word's first n nibbles clears `good`)
			auto pack8 = [](uint32_t x, uint32_t n, bool &good) -> uint32_t {
				const uint32_t keep = n >= 8 ? 0xFFFFFFFFu : ((1u << (4 * n)) - 1u);
				const uint32_t xm = (x & keep) | (0x11111111u & ~keep);            // unused nibbles read as A
				const uint32_t zero = (xm - 0x11111111u) & ~xm & 0x88888888u;       // a nibble of code 0 (would borrow below)
				const uint32_t t = xm - 0x11111111u;                                // code - 1 per nibble
				good = good && !zero && !(t & 0xCCCCCCCCu);
				uint32_t y = (t | (t >> 2)) & 0x0F0F0F0Fu;
				y = (y | (y >> 4)) & 0x00FF00FFu;
				y = (y | (y >> 8)) & 0xFFFFu;                                       // symbol k at bits 2k, 2k + 1
				const uint32_t r = __brev(y) >> 16;                                 // order reversed, bits of a pair swapped
				return ((r & 0x5555u) << 1) | ((r >> 1) & 0x5555u);               // 16 bits, symbol 0 on top
			};
			bool good = true;
			const uint32_t Ku = (uint32_t)K;
			const uint32_t w_lo = pack8(lo, Ku < 8 ? Ku : 8u, good);
			if (Ku <= 8) w = w_lo >> (16 - 2 * Ku);
			else { const uint32_t w_hi = pack8(hi, Ku - 8, good); w = (w_lo << (2 * (Ku - 8))) | (w_hi >> (16 - 2 * (Ku - 8))); }
			ok = good ? 1u : 0u;
		} else for (int k = 0; k < K; ++k) {
			const uint32_t c = qcodes[b + p + k];
			ok &= (c - 1u) < 4u;
			w = (w << 2) | ((c - 1u) & 3u);
		}
		w &= wmask;
		if (ok) {      // range = first entry (40 bits) and length (24 bits): x = low 32 bits of the entry, y = length | high bits << 24
			unsigned long long beg; uint32_t n;
			bhip_acx_range(acx, w, beg, n);
			r.x = (uint32_t)beg; r.y = n | (uint32_t)(beg >> 32) << 24;
		}
	}
	// Every sampled word is one vote and `need` of the nwords votes survive E edits -- of ANY subset of n' of those words, need - (nwords - n')
	// do.  The lists have very different lengths (and their sum is what the prefilter walks: the whole slope of a batch's time over
	// the database size), so the longest ones are left out as long as the smaller `need` still says something.  The words of a query
	// are W16 (8 or 16) consecutive lanes of one row; a left-out list is an empty range and one vote less in the header.
	if (min_need && W16 <= 16u) {
		const uint32_t n_mine = r.y & 0xFFFFFFu;
		uint32_t rank = 0;
		for (uint32_t k = 0; k < W16; ++k) {
			const uint32_t n_k = (uint32_t)__shfl((int)n_mine, (int)k, (int)W16);
			rank += (n_k > n_mine || (n_k == n_mine && k < j)) ? 1u : 0u;
		}
		const uint32_t allowed = need > min_need ? need - min_need : 0u;
		const bool drop = rank < allowed && n_mine >= drop_len && n_mine > 0u;
		const unsigned long long bal = __ballot(drop);
		const uint32_t row0 = (threadIdx.x & 63u) - j;
		const uint32_t ndrop = (uint32_t)__popcll((bal >> row0) & ((1ull << W16) - 1ull));
		if (drop) r = make_uint2(0, 0);
		need -= ndrop;
	}
	ranges[t] = r;
	// header: need | words << 16 ; length | budget << 12 | (words one edit can destroy = ceil(K / stride)) << 20
	if (j == 0) {
		const uint32_t Eq_ = qemac[q];
		hdr[li] = make_uint2((need > 0xFFFFu ? 0xFFFFu : need) | nwords << 16, len | (uint32_t)(Eq_ > 255 ? 255 : Eq_) << 12 | ((uint32_t)(K + stride - 1) / stride) << 20);
		if (qmeta) qmeta[li] = make_uint4(q, len | Eq_ << 16, qsix ? qsix[q] : q, 0u);
	}
	}
}

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

// ------------------------------------------------------------------------------------------------
// Lane-resolved prefilter, counting-filter variant (same inputs and outputs as k_prefilter_mask).
// Most list records of a query belong to clumps that share only one or two words with it; the exact per-clump hash
// table of k_prefilter_mask pays a returning compare-and-swap for each of them.  Here every record first bumps one of
// 1 << CB approximate 16-bit counters (hash of the clump id, fire-and-forget LDS adds, no key, no probing).  A record
// whose counter stays below `need` cannot belong to a candidate clump (its counter is an upper bound of its clump's
// count), so only the survivors -- about one record in six on the bench workload -- are looked at again: they are
// compacted through a small LDS ring so that 16 lanes work on 16 survivors, inserted by clump id into a small exact
// table that carries the sixteen 8-bit lane counters directly, and the lanes that reach `need` are emitted.  No
// false negatives: a record of a clump with count >= need always survives; false survivors only cost work.
// ------------------------------------------------------------------------------------------------
#ifndef CF_MINWAVES
#define CF_MINWAVES 3
#endif
// The workgroup of this kernel is ONE wave: its LDS operations are issued and completed in program order, so a later read sees an
// earlier update by any lane without a barrier.  __syncthreads() would still cost an s_waitcnt vmcnt(0) lgkmcnt(0) -- a wait for
// every load in flight, i.e. for the record prefetch of the NEXT quad that the software pipeline has just issued.  What the phases
// need between them is only that the compiler keeps their LDS accesses in order.  (-DCF_BARRIERS=1 puts the barriers back.)
#if defined(CF_BARRIERS) && CF_BARRIERS
#define CF_WAVE_ORDER() __syncthreads()
#else
#define CF_WAVE_ORDER() __asm__ volatile("" ::: "memory")
#endif
// inclusive prefix sum over the 64 lanes of a wave (all lanes active): four shifts inside each row of 16 lanes, then lane 15 of
// rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2 and 3 -- data-parallel-primitive moves, no LDS round trip
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x) {
	int v = (int)x;
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);    // row_shr:1
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);    // row_shr:2
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);    // row_shr:4
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);    // row_shr:8
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);    // row_bcast:15 -> rows 1, 3
	v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);    // row_bcast:31 -> rows 2, 3
	return (uint32_t)v;
}
template <int CB, int RBT>
__global__ __launch_bounds__(64, CF_MINWAVES) void k_prefilter_cf(
		const uint2 *__restrict__ ranges, const uint2 *__restrict__ hdr, uint32_t W16, uint32_t n_list,
		const uint32_t *__restrict__ ent,   // 4-byte (clump, lane-set code) records
		const uint32_t *__restrict__ bad, uint32_t n_bad, const uint32_t *__restrict__ clump_len, uint32_t tot_refs,
		uint2 *__restrict__ tasks, uint32_t *__restrict__ n_tasks, uint32_t task_cap,
		unsigned long long *__restrict__ ent_read,
		uint32_t *__restrict__ fb_list, uint32_t *__restrict__ n_fb,
		unsigned long long *__restrict__ unit_sum, unsigned long long *__restrict__ col_sum, unsigned long long *__restrict__ qlen_sum,
		unsigned long long *__restrict__ surv_sum,
		uint2 *__restrict__ tasks2, uint32_t *__restrict__ n_tasks2, int prune,     // prune: lanes that cannot hold a minimum go to tasks2 with their lower bound
		const uint32_t *__restrict__ sel, const uint32_t *__restrict__ n_sel_dev, int byte_counters) { // sel: optional: only the list positions sel[0 .. *n_sel_dev) -- the second pass over the
		                                                                            // queries that overflowed the first pass's tables, with the largest tables
	constexpr uint32_t NCNT = 1u << CB;                                   // approximate counters per query (16 bit each)
	constexpr uint32_t LT = CB <= 9 ? 64u : (CB == 10 ? 128u : 256u);     // exact lane-table slots per query
	constexpr uint32_t CF_STAGE = 64u;                                     // staged tasks per output list
	constexpr uint32_t RING = 32u;                                         // >= 15 pending + 16 new survivors (the ring is drained after every 16 offered records); a power of two
	__shared__ __attribute__((aligned(16))) uint32_t s_cnt[4][NCNT / 2];
	__shared__ uint32_t s_key[4][LT];
	__shared__ unsigned long long s_lc[4][LT][2];
	__shared__ uint32_t s_ring[4][RING];                                   // raw record words
	__shared__ uint16_t s_lut[256];                                        // lane-set code -> lane mask
	__shared__ uint8_t s_used[4][LT];                                      // slots of the lane table in use (LT <= 256)
	__shared__ uint2 s_stage[2][CF_STAGE];
	__shared__ uint32_t s_ovf[4];
	__shared__ uint32_t s_dummy[16];          // compare-and-swap target of idle lanes (never written: the compare value cannot match)
	const uint32_t lane = threadIdx.x, g = lane >> 4, gl = lane & 15;
	if (lane < 16) s_dummy[lane] = 0;
	for (uint32_t i = lane; i < 256; i += 64) s_lut[i] = (uint16_t)bhip_lane_code_mask(i);
	for (uint32_t i = lane; i < 4 * NCNT / 2; i += 64) (&s_cnt[0][0])[i] = 0;
	for (uint32_t i = lane; i < 4 * LT; i += 64) { (&s_key[0][0])[i] = 0; (&s_lc[0][0][0])[2 * i] = 0; (&s_lc[0][0][0])[2 * i + 1] = 0; }
	if (lane < 4) s_ovf[lane] = 0;
	__syncthreads();
	unsigned long long my_ent = 0, my_units = 0, my_cols = 0, my_qlen = 0, my_surv = 0;
	uint32_t sink = 0, sink_h = 0;          // see bhip_acx_raw_or_pad (bhip_internal.h)
#ifdef PFM_PROF
	unsigned long long my_t[8] = {0,0,0,0,0,0,0,0}, t_last = wall_clock64();
#endif

	// Staged tasks: this block is ONE wave, so the fill counts of the two output lists are wave-uniform registers and the
	// positions of a lane's tasks come from a prefix sum over the wave: no LDS atomics, no per-task round trip.
	// which = 0: first sweep, 1: deferred (li_lb = li | bound << 24).
	uint32_t nst[2] = {0u, 0u};
	const unsigned long long lt_mask = (1ull << lane) - 1ull;
	auto flush_one = [&](uint32_t which) {
		const uint32_t n = nst[which];
		if (n) {
			uint32_t base = 0;
			if (lane == 0) base = atomicAdd(which ? n_tasks2 : n_tasks, n);
			base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
			uint2 *dst = which ? tasks2 : tasks;
			if (lane < n && base + lane < task_cap) dst[base + lane] = s_stage[which][lane];
			__syncthreads();
		}
		nst[which] = 0;
	};
	auto put_row = [&](uint32_t which, bool mine, uint32_t li_lb, uint32_t refIx) {     // wave-uniform call; `mine`: this lane has a task for list `which`
		const unsigned long long m = __ballot(mine);
		const uint32_t cnt = (uint32_t)__popcll(m);
		if (!cnt) return;
		if (nst[which] + cnt > CF_STAGE) flush_one(which);
		if (mine) s_stage[which][nst[which] + (uint32_t)__popcll(m & lt_mask)] = make_uint2(li_lb, refIx);
		nst[which] += cnt;
	};
	auto flush = [&]() { flush_one(0); flush_one(1); };

	const uint32_t n_items = sel ? (*n_sel_dev < n_list ? *n_sel_dev : n_list) : n_list;      // queries this launch works on
	const uint32_t n_quads = (n_items + 3) >> 2;
	constexpr uint32_t RB = RBT;             // blocks of 64 records per query that are fetched one quad ahead and stay in registers between the two looks
	                                         // (2, 3 or 4: the launcher takes the smallest that holds the expected record stream of a query -- what lies
	                                         // beyond is loaded where it is consumed, twice, with its latency exposed: 40 % of the kernel at 150 records per read)
	// (cross-lane moves by data-parallel primitives and lane reads where the pattern is fixed: a shuffle is an LDS round trip, and
	// this kernel's time is the sum of its dependent LDS round trips)
#define GROUP_PICK(v, l) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x150 + (l), 0xF, 0xF, false))      /* lane l (0..15, a constant) of the own group: row_newbcast */
	auto wave_max4 = [&](uint32_t v) -> uint32_t {                    // maximum over the four groups of a group-uniform value
		const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16),
			c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
		const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
		return ab > cd ? ab : cd;
	};
	auto group_scan = [&](uint32_t n, uint32_t &T, uint32_t &excl) {
		int ps = (int)n;
		ps += __builtin_amdgcn_update_dpp(0, ps, 0x111, 0xF, 0xF, false);    // row_shr:1 (a row = the 16 lanes of a group; lanes without a source add 0)
		ps += __builtin_amdgcn_update_dpp(0, ps, 0x112, 0xF, 0xF, false);    // row_shr:2
		ps += __builtin_amdgcn_update_dpp(0, ps, 0x114, 0xF, 0xF, false);    // row_shr:4
		ps += __builtin_amdgcn_update_dpp(0, ps, 0x118, 0xF, 0xF, false);    // row_shr:8
		T = GROUP_PICK(ps, 15);
		excl = (uint32_t)ps - n;
	};
	auto wave_blocks = [&](uint32_t T) -> uint32_t { return (wave_max4(T) + 63) >> 6; };
	auto load4 = [&](uint32_t ex, unsigned long long dl, uint32_t T, uint32_t b, uint32_t (&rec)[4]) {      // see k_prefilter_mask
		#pragma unroll
		for (uint32_t u = 0; u < 4; ++u) {
			const uint32_t i = (b * 4 + u) * 16 + gl;
			uint32_t kk = 0;
			if (W16 > 8) kk += __shfl(ex, 8, 16) <= i ? 8u : 0u;      // (uniform) with 8 words per query the upper half is empty
			kk += __shfl(ex, kk + 4, 16) <= i ? 4u : 0u;
			kk += __shfl(ex, kk + 2, 16) <= i ? 2u : 0u;
			kk += __shfl(ex, kk + 1, 16) <= i ? 1u : 0u;
			const unsigned long long addr = __shfl(dl, kk, 16) + i;
			rec[u] = bhip_acx_raw_or_pad(ent, addr, i < T, hdr, sink);
		}
	};
	// Software pipeline over the quads of this block: the header and list ranges (k_seed_ranges made them) are fetched TWO
	// iterations ahead and the first RB blocks of list records ONE iteration ahead, so that the gather of a quad's records --
	// short reads at random addresses, 43 % of the wave cycles when it was waited for in place -- runs while the previous
	// quad is counted.
	// (unconditional loads from clamped, always valid addresses, masked afterwards: a load under a condition is compiled as a
	// branch with an s_waitcnt vmcnt(0) at its join, which would expose the latency this prefetch is there to hide -- and wait
	// for every other load in flight)
	typedef const unsigned long long __attribute__((address_space(1))) *g64_t;
	auto fetch_hdr_issue = [&](uint32_t quad, unsigned long long &h, unsigned long long &r) {      // raw words; nothing here waits for them
		const uint32_t it = quad * 4 + g;
		const bool ok = it < n_items, okw = ok && gl < W16;         // (it < n_items implies quad < n_quads)
		uint32_t lic = ok ? it : 0u;
		if (sel) lic = n_items ? sel[lic] : 0u;                     // (wave-uniform branch; the first pass has no selection)
		h = ((g64_t)(uintptr_t)(hdr + lic))[0]; r = ((g64_t)(uintptr_t)(ranges + ((size_t)lic * W16 + (okw ? gl : 0u))))[0];
	};
	auto fetch_hdr_finish = [&](uint32_t quad, unsigned long long h, unsigned long long r, uint2 &hd, uint2 &rg) {
		const uint32_t it = quad * 4 + g;
		const bool ok = it < n_items, okw = ok && gl < W16;
		sink_h ^= (uint32_t)h + (uint32_t)r;         // (its own chain: folded into `sink`, the compiler consumes the words where that chain is first touched)
		const uint32_t mh = ok ? 0xFFFFFFFFu : 0u, mr = okw ? 0xFFFFFFFFu : 0u;
		hd = make_uint2((uint32_t)h & mh, (uint32_t)(h >> 32) & mh);
		rg = make_uint2((uint32_t)r & mr, (uint32_t)(r >> 32) & mr);
	};
	auto fetch_hdr = [&](uint32_t quad, uint2 &hd, uint2 &rg) { unsigned long long h, r; fetch_hdr_issue(quad, h, r); fetch_hdr_finish(quad, h, r, hd, rg); };
	// issue: the record words of the first RB blocks of a quad's record stream (nothing here waits for them)
	auto start_stream = [&](uint32_t quad, const uint2 &rg, uint32_t &T, uint32_t &ex, unsigned long long &dl, uint32_t &nblk, uint32_t (&raw)[RB][4]) -> uint32_t {
		const bool lv = quad < n_quads && quad * 4 + g < n_items;
		const unsigned long long beg = lv ? ((unsigned long long)rg.x | (unsigned long long)(rg.y >> 24) << 32) : 0ull;
		const uint32_t n0 = lv ? rg.y & 0xFFFFFFu : 0u;
		group_scan(n0, T, ex);
		dl = beg - ex;
		nblk = wave_blocks(T);
		// which list does stream position i belong to: the search over the group's exclusive prefix sums, all RB * 4 positions of
		// this lane stage by stage (their cross-lane reads are in flight together: one LDS round trip per stage, not per position)
		uint32_t kk[RB * 4];
		if (W16 <= 8) {
			// eight lists: the seven inner boundaries are broadcast inside the group (data-parallel moves) and the binary search
			// becomes a selection tree in registers -- no LDS round trip at all
			const uint32_t e1 = GROUP_PICK(ex, 1), e2 = GROUP_PICK(ex, 2), e3 = GROUP_PICK(ex, 3), e4 = GROUP_PICK(ex, 4),
				e5 = GROUP_PICK(ex, 5), e6 = GROUP_PICK(ex, 6), e7 = GROUP_PICK(ex, 7);
			#pragma unroll
			for (uint32_t j = 0; j < RB * 4; ++j) {
				const uint32_t i = j * 16 + gl;
				const bool a = e4 <= i;
				const bool b = (a ? e6 : e2) <= i;
				const uint32_t lo13 = b ? e3 : e1, hi57 = b ? e7 : e5;
				const bool c = (a ? hi57 : lo13) <= i;
				kk[j] = (a ? 4u : 0u) + (b ? 2u : 0u) + (c ? 1u : 0u);
			}
		} else {
			const uint32_t e8 = GROUP_PICK(ex, 8);
			#pragma unroll
			for (uint32_t j = 0; j < RB * 4; ++j) kk[j] = e8 <= j * 16 + gl ? 8u : 0u;
			#pragma unroll
			for (uint32_t step = 4; step >= 1; step >>= 1) {
				uint32_t t[RB * 4];
				#pragma unroll
				for (uint32_t j = 0; j < RB * 4; ++j) t[j] = __shfl(ex, kk[j] + step, 16);
				#pragma unroll
				for (uint32_t j = 0; j < RB * 4; ++j) kk[j] += t[j] <= j * 16 + gl ? step : 0u;
			}
		}
		unsigned long long base[RB * 4];
		#pragma unroll
		for (uint32_t j = 0; j < RB * 4; ++j) base[j] = __shfl(dl, kk[j], 16);
		#pragma unroll
		for (uint32_t b = 0; b < RB; ++b) {
			#pragma unroll
			for (uint32_t u = 0; u < 4; ++u) {
				const uint32_t i = (b * 4 + u) * 16 + gl;
				raw[b][u] = bhip_acx_raw_issue(ent, base[b * 4 + u] + i, i < T, hdr);
			}
		}
		return n0;
	};
	// consume: padding where the stream has ended
	auto finish_stream = [&](uint32_t T, const uint32_t (&raw)[RB][4], uint32_t (&r)[RB][4]) {
		#pragma unroll
		for (uint32_t b = 0; b < RB; ++b) {
			#pragma unroll
			for (uint32_t u = 0; u < 4; ++u) r[b][u] = bhip_acx_raw_finish(raw[b][u], (b * 4 + u) * 16 + gl < T, sink);
		}
	};
	uint2 hd_c, rg_c, hd_n, rg_n;
	fetch_hdr(blockIdx.x, hd_c, rg_c);
	fetch_hdr(blockIdx.x + gridDim.x, hd_n, rg_n);
	uint32_t T0, ex0, nblk0; unsigned long long dl0;
	uint32_t rc[RB][4], raw[RB][4];          // record words: clump | lane-set code << 24, BHIP_REC_PAD beyond the stream
	uint32_t n0 = start_stream(blockIdx.x, rg_c, T0, ex0, dl0, nblk0, raw);
	finish_stream(T0, raw, rc);
	for (uint32_t quad = blockIdx.x; quad < n_quads; quad += gridDim.x) {
		const bool live = quad * 4 + g < n_items;
		const uint32_t li = sel ? (live ? sel[quad * 4 + g] : 0u) : quad * 4 + g;      // list position of this group's query
		const uint2 hd = hd_c;
		unsigned long long h_raw, r_raw;
		fetch_hdr_issue(quad + 2 * gridDim.x, h_raw, r_raw);
		uint32_t T1, ex1, nblk1; unsigned long long dl1;
		const uint32_t n1 = start_stream(quad + gridDim.x, rg_n, T1, ex1, dl1, nblk1, raw);
		const uint32_t need = hd.x & 0xFFFFu, nwords = live ? hd.x >> 16 : 0u, len = hd.y & 0xFFFu;
		const uint32_t budget = (hd.y >> 12) & 255u, dper = (hd.y >> 20) & 15u ? (hd.y >> 20) & 15u : 1u;
		const uint32_t thr = need ? need : 1u;
		const uint32_t maxw = wave_max4(nwords);          // (nwords is the same in the 16 lanes of a group)
		auto word_range = [&](uint32_t j, unsigned long long &beg, uint32_t &n) {
			uint2 r = make_uint2(0, 0);
			if (live && j < nwords) r = ranges[(size_t)li * W16 + j];
			beg = (unsigned long long)r.x | (unsigned long long)(r.y >> 24) << 32; n = r.y & 0xFFFFFFu;
		};
		// A query whose whole record stream is at most 255 records cannot drive a counter beyond 255: its counters are BYTES, twice as
		// many in the same LDS (2 << CB per query) -- half the load per counter, a third to a quarter of the false survivors (a survivor
		// costs about eight records' worth of work).  Longer streams keep the 16-bit counters.  cshift = log2 of the counter's bits.
		const bool nar = byte_counters && nwords <= 16u && T0 <= 255u;
		const uint32_t cshift = nar ? 3u : 4u, cper = nar ? 3u : 1u, cmask = nar ? 0xFFu : 0xFFFFu, hsh = nar ? 0u : 1u;      // (group-uniform)
		auto count4 = [&](const uint32_t (&rec)[4]) {     // phase A: approximate counters, no return values
			#pragma unroll
			for (int u = 0; u < 4; ++u) if (rec[u] != BHIP_REC_PAD) {
				const uint32_t h = (((rec[u] & 0xFFFFFFu) * 0x9E3779B1u) >> (31 - CB)) >> hsh;      // CB + 1 bits (bytes) or CB bits
				atomicAdd(&s_cnt[g][h >> (5u - cshift)], 1u << ((h & cper) << cshift));
			}
		};
		uint32_t pending = 0, head = 0;          // survivors waiting in this group's ring (replicated in its 16 lanes)
		uint32_t nused = 0;                      // slots of this group's lane table in use (replicated)
		auto c_round = [&]() {                    // wave-uniform: every group moves up to 16 survivors into its lane table
			const uint32_t take = pending < 16 ? pending : 16;
			const bool active = gl < take;
			const uint32_t hpos = (head + gl) & (RING - 1);
			const uint32_t rec = active ? s_ring[g][hpos] : 0u;
			const uint32_t clump = rec & 0xFFFFFFu, mask = s_lut[rec >> 24];
			const uint32_t key = clump + 1u;
			uint32_t slot = (clump * 0x85EBCA6Bu) >> (32 - (CB <= 9 ? 6 : (CB == 10 ? 7 : 8)));
			bool act = active, found = false, fresh = false;
			for (uint32_t probes = 0; __any(act) && probes < LT; ++probes) {
				const uint32_t old = atomicCAS(act ? &s_key[g][slot] : &s_dummy[gl], act ? 0u : 0xFFFFFFFFu, key);
				const bool ok = act && (old == 0 || old == key);
				fresh |= act && old == 0;
				found |= ok;
				act = act && !ok;
				slot = act ? (slot + 1) & (LT - 1) : slot;
			}
			if (act) s_ovf[g] = 1;
			{
				const uint32_t m16 = (uint32_t)(__ballot(fresh) >> (lane & 48u)) & 0xFFFFu;
				if (fresh) s_used[g][nused + __popc(m16 & ((1u << gl) - 1u))] = (uint8_t)slot;
				nused += __popc(m16);
			}
			if (found) {
				if (mask & 0xFFu) atomicAdd(&s_lc[g][slot][0], spread8(mask & 0xFFu));
				if (mask >> 8) atomicAdd(&s_lc[g][slot][1], spread8(mask >> 8));
			}
			head = (head + take) & (RING - 1);
			pending -= take;
		};
		auto offer4 = [&](const uint32_t (&rec)[4]) {    // phase B: survivors of the counter test go to the ring
			uint32_t cv[4];
			#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const uint32_t h = rec[u] != BHIP_REC_PAD ? (((rec[u] & 0xFFFFFFu) * 0x9E3779B1u) >> (31 - CB)) >> hsh : 0u;
				cv[u] = (s_cnt[g][h >> (5u - cshift)] >> ((h & cper) << cshift)) & cmask;
			}
			#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const bool surv = rec[u] != BHIP_REC_PAD && cv[u] >= thr;
				const uint32_t m16 = (uint32_t)(__ballot(surv) >> (lane & 48u)) & 0xFFFFu;
				if (surv) {
					uint32_t pos = head + pending + __popc(m16 & ((1u << gl) - 1u));
					pos &= RING - 1;
					s_ring[g][pos] = rec[u];
				}
				pending += __popc(m16);
				if (gl == 0) my_surv += __popc(m16);
				while (__any(pending >= 16)) c_round();       // (at most 15 + 16 pending: the ring holds 32)
			}
		};

		PFM_T(0);
		my_ent += n0;
		// ---- phase A over every record of the query
		#pragma unroll
		for (uint32_t b = 0; b < RB; ++b) if (b < nblk0) count4(rc[b]);
		for (uint32_t b = RB; b < nblk0; b += 2) {      // (two blocks' loads in flight together)
			uint32_t rec[4], rec2[4];
			load4(ex0, dl0, T0, b, rec); load4(ex0, dl0, T0, b + 1, rec2);      // (a block beyond the stream is all padding)
			count4(rec); count4(rec2);
		}
		uint32_t gtot = T0;                        // records of this group's query (16-bit counters: beyond 65 535 the query takes the dense fallback)
		for (uint32_t base = 16; base < maxw; base += 16) {
			unsigned long long xb; uint32_t xn, T, ex;
			word_range(base + gl, xb, xn);
			my_ent += xn;
			group_scan(xn, T, ex);
			gtot = gtot + T < gtot ? 0xFFFFFFFFu : gtot + T;
			const uint32_t nb = wave_blocks(T);
			for (uint32_t b = 0; b < nb; ++b) { uint32_t rec[4]; load4(ex, xb - ex, T, b, rec); count4(rec); }
		}
		if (gtot > 65535u && gl == 0) s_ovf[g] = 1;
		CF_WAVE_ORDER();
		PFM_T(7);
		// ---- phase B: second look at every record (registers for the first blocks, L2 for the rest)
		#pragma unroll
		for (uint32_t b = 0; b < RB; ++b) if (b < nblk0) offer4(rc[b]);
		for (uint32_t b = RB; b < nblk0; b += 2) {
			uint32_t rec[4], rec2[4];
			load4(ex0, dl0, T0, b, rec); load4(ex0, dl0, T0, b + 1, rec2);
			offer4(rec); offer4(rec2);
		}
		for (uint32_t base = 16; base < maxw; base += 16) {
			unsigned long long xb; uint32_t xn, T, ex;
			word_range(base + gl, xb, xn);
			group_scan(xn, T, ex);
			const uint32_t nb = wave_blocks(T);
			for (uint32_t b = 0; b < nb; ++b) { uint32_t rec[4]; load4(ex, xb - ex, T, b, rec); offer4(rec); }
		}
		PFM_T(2);
		while (__any(pending > 0)) c_round();
		CF_WAVE_ORDER();
		PFM_T(3);
		// ---- emit the lanes that reach the threshold, clear the tables
		// Slot-parallel: lane gl of a group owns the group's gl-th used slot.  The positions of its tasks in the two staged lists
		// come from ONE wave-wide prefix sum over the per-lane counts (DPP, no LDS round trip); the stores are fire-and-forget.
		// A lane with c matching words lost (W_valid - c) words, one edit destroys at most `dper` of them: its edit distance
		// is at least budget - (c - need) / dper.  Unless every hit within budget is wanted, only the lanes with the
		// smallest bound are swept at once; the others wait for the minimum those produce (k_task_filter).
		const uint32_t ovf = s_ovf[g];
		const bool em = live && !ovf;
		const uint32_t nu = em ? nused : 0u;
		const uint32_t nu_max = wave_max4(nu);
		const uint32_t inv_dper = 65536u / dper + 1u;        // x / dper == (x * inv_dper) >> 16 for x < 256, dper < 16
		auto lanes_ge = [&](unsigned long long lo, unsigned long long hi, uint32_t t) -> uint32_t {
			uint32_t m16 = 0;
			if (nwords < 128) {      // byte-parallel compare: (b | 0x80) - t keeps its top bit iff b >= t; top bits gathered by a multiply
				const unsigned long long H = 0x8080808080808080ull, L1 = 0x0101010101010101ull, G = 0x0102040810204080ull;
				const unsigned long long tl = ((lo | H) - t * L1) & H, th = ((hi | H) - t * L1) & H;
				m16 = (uint32_t)(((tl >> 7) * G) >> 56) | ((uint32_t)(((th >> 7) * G) >> 56) << 8);
			} else {
				#pragma unroll
				for (uint32_t z = 0; z < 16; ++z) m16 |= ((uint32_t)(((z < 8 ? lo : hi) >> (8 * (z & 7))) & 255u) >= t ? 1u : 0u) << z;
			}
			return m16;
		};
		auto look = [&](uint32_t iu, uint32_t &slot, uint32_t &c, unsigned long long &lo, unsigned long long &hi) -> uint32_t {
			const bool has = iu < nu;
			slot = has ? (uint32_t)s_used[g][iu] : 0u;
			c = s_key[g][slot] - 1u; lo = s_lc[g][slot][0]; hi = s_lc[g][slot][1];
			const uint32_t first = c * 16u, nv = first < tot_refs ? (tot_refs - first < 16u ? tot_refs - first : 16u) : 0u;     // lanes of the clump that exist
			return has ? lanes_ge(lo, hi, thr) & ((1u << nv) - 1u) : 0u;
		};
		auto byte_of = [&](unsigned long long lo, unsigned long long hi, uint32_t z) -> uint32_t { return (uint32_t)((z < 8 ? lo : hi) >> (8u * (z & 7u))) & 255u; };
		auto group_max = [&](uint32_t v) -> uint32_t {      // maximum over the 16 lanes of the group: neighbours, pairs of neighbours, then the two mirror moves
			int t;
			t = __builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); v = (uint32_t)t > v ? (uint32_t)t : v;     // quad_perm:[1,0,3,2]
			t = __builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false); v = (uint32_t)t > v ? (uint32_t)t : v;     // quad_perm:[2,3,0,1]
			t = __builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false); v = (uint32_t)t > v ? (uint32_t)t : v;    // row_half_mirror
			t = __builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false); v = (uint32_t)t > v ? (uint32_t)t : v;    // row_mirror
			return v;
		};
		uint32_t slot0, c0; unsigned long long lo0, hi0;
		const uint32_t m16_0 = look(gl, slot0, c0, lo0, hi0);
		uint32_t cmax_all = 0;
		if (prune) {
			uint32_t cmax = 0;
			for (uint32_t m = m16_0; m; m &= m - 1) { const uint32_t v = byte_of(lo0, hi0, (uint32_t)__builtin_ctz(m)); cmax = v > cmax ? v : cmax; }
			for (uint32_t iu0 = 16; iu0 < nu_max; iu0 += 16) {
				uint32_t sl, c; unsigned long long lo, hi;
				for (uint32_t m = look(iu0 + gl, sl, c, lo, hi); m; m &= m - 1) { const uint32_t v = byte_of(lo, hi, (uint32_t)__builtin_ctz(m)); cmax = v > cmax ? v : cmax; }
			}
			cmax_all = group_max(cmax);
		}
		PFM_T(1);
		auto emit_slots = [&](uint32_t iu, uint32_t slot, uint32_t c, unsigned long long lo, unsigned long long hi, uint32_t m16) {
			if (iu < nu) { s_key[g][slot] = 0; s_lc[g][slot][0] = 0; s_lc[g][slot][1] = 0; }     // (this wave's reads of the slot are done: LDS operations of one wave stay in order)
			const uint32_t m0 = prune ? m16 & lanes_ge(lo, hi, cmax_all > thr ? cmax_all : thr) : m16, m1 = m16 & ~m0;
			const uint32_t cnt = (uint32_t)__popc(m0) | (uint32_t)__popc(m1) << 16;
			const uint32_t incl = wave_incl_scan_u32(cnt), tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63), excl = incl - cnt;
			if (!tot) return;                         // wave-uniform
			const uint32_t tot0 = tot & 0xFFFFu, tot1 = tot >> 16;
			uint32_t p[2]; bool direct[2];
			#pragma unroll
			for (uint32_t w = 0; w < 2; ++w) {
				const uint32_t tw = w ? tot1 : tot0, ew = w ? excl >> 16 : excl & 0xFFFFu;
				direct[w] = false;
				if (tw && nst[w] + tw > CF_STAGE) flush_one(w);
				if (tw > CF_STAGE) {                  // more than the stage holds in one go: straight to the list
					uint32_t base = 0;
					if (lane == 0) base = atomicAdd(w ? n_tasks2 : n_tasks, tw);
					p[w] = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + ew; direct[w] = true;
				} else { p[w] = nst[w] + ew; nst[w] += tw; }
			}
			PFM_T(5);
			for (uint32_t m = m16; m; m &= m - 1) {
				const uint32_t z = (uint32_t)__builtin_ctz(m), w = (m1 >> z) & 1u;
				uint32_t lb = 0;
				if (prune) { const uint32_t gain = ((byte_of(lo, hi, z) - need) * inv_dper) >> 16; lb = gain >= budget ? 0u : budget - gain; }
				const uint2 task = make_uint2(li | lb << 24, c * 16u + z);
				const uint32_t pos = p[w]; p[w] = pos + 1;
				if (direct[w]) { if (pos < task_cap) (w ? tasks2 : tasks)[pos] = task; }
				else s_stage[w][pos] = task;
			}
			if (m16) { ++my_units; my_qlen += len; }       // (the swept columns of lane tasks are counted by the sweep: tcol_sum)
		};
		emit_slots(gl, slot0, c0, lo0, hi0, m16_0);
		for (uint32_t iu0 = 16; iu0 < nu_max; iu0 += 16) {
			uint32_t sl, c; unsigned long long lo, hi;
			const uint32_t m16 = look(iu0 + gl, sl, c, lo, hi);
			emit_slots(iu0 + gl, sl, c, lo, hi, m16);
		}
		for (uint32_t i = 0; i < n_bad; ++i) {         // burst.c:4136-4138, 4282-4283: every lane of the ambiguous clumps
			const uint32_t c = bad[i];
			put_row(0, em && c * 16u + gl < tot_refs, li, c * 16u + gl);
			if (em && gl == 0) { ++my_units; my_qlen += len; }
		}
		if (ovf) {
			for (uint32_t i = gl; i < LT; i += 16) { s_key[g][i] = 0; s_lc[g][i][0] = 0; s_lc[g][i][1] = 0; }
			if (live && gl == 0) { const uint32_t pos = atomicAdd(n_fb, 1u); fb_list[pos] = li; }
		}
		PFM_T(4);
		{
			uint4 *cz = (uint4 *)&s_cnt[g][0];
			for (uint32_t i = gl; i < NCNT / 8; i += 16) cz[i] = make_uint4(0, 0, 0, 0);
		}
		CF_WAVE_ORDER();
		if (gl == 0) s_ovf[g] = 0;
		CF_WAVE_ORDER();
		PFM_T(5);
		// rotate the pipeline
		uint2 hd_nn, rg_nn;
		fetch_hdr_finish(quad + 2 * gridDim.x, h_raw, r_raw, hd_nn, rg_nn);
		hd_c = hd_n; hd_n = hd_nn; rg_n = rg_nn;
		T0 = T1; ex0 = ex1; dl0 = dl1; nblk0 = nblk1; n0 = n1;
		finish_stream(T0, raw, rc);          // the records fetched during this iteration are first looked at here
		PFM_T(6);
	}
	flush();
#ifdef PFM_PROF
	if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_pfm_prof[i], my_t[i]);
#endif
	if (ent_read && my_ent) atomicAdd(ent_read, my_ent);
	if (surv_sum && my_surv) atomicAdd(surv_sum, my_surv);
	if (n_list == 0xFFFFFFFFu) { fb_list[0] = sink; fb_list[1] = sink_h; }       // never: keeps the record loads unconditional
	if (my_units) { atomicAdd(unit_sum, my_units); atomicAdd(col_sum, my_cols); atomicAdd(qlen_sum, my_qlen); }
}
#define BHIP_INST_PFCF(CB, RB) \
	template __global__ void k_prefilter_cf<CB, RB>(const uint2 *, const uint2 *, uint32_t, uint32_t, const uint32_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t, \
		uint2 *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, unsigned long long *, unsigned long long *, \
		uint2 *, uint32_t *, int, const uint32_t *, const uint32_t *, int);
BHIP_INST_PFCF(9, 2) BHIP_INST_PFCF(9, 3) BHIP_INST_PFCF(9, 4) BHIP_INST_PFCF(10, 2) BHIP_INST_PFCF(10, 4) BHIP_INST_PFCF(11, 2) BHIP_INST_PFCF(11, 4)

// ------------------------------------------------------------------------------------------------
// Lane-resolved prefilter, counting filter with ONE QUERY PER WAVE (round 5; same inputs and outputs as k_prefilter_cf).
// k_prefilter_cf gives a query 16 lanes and a wave four queries: right while a query's record stream is a few dozen records
// (databases of a few GB), and register-bound beyond -- at the metric's size a query walks ~310 records, the kernel keeps 16
// record registers per lane plus the next quad's 16 in flight (168 VGPRs: 3 waves per SIMD) and spends a quarter of its time
// finding, per lane and stream position, which list the position belongs to (a selection tree over per-group boundaries).
// Here the 64 lanes walk ONE query's stream, 64 records per row:
//  * the list boundaries are WAVE-UNIFORM: lane l holds list l's end position and biased base address, a scalar cursor walks
//    them (v_readlane with a scalar index), and a row costs one address select per list boundary that falls into it -- about
//    1.4 per row instead of a tree per position;
//  * a row is one VGPR: 6 resident rows (384 records) + their list numbers are 12 registers, the kernel runs at 8 waves per
//    SIMD and hides its gather behind the other waves instead of behind a software pipeline;
//  * the approximate counters are LIST MASKS: a slot holds one bit per list (a list names a clump at most once, burst.c:3385-3386,
//    so "lists with a record in this slot" is the same upper bound of a clump's count as "records in this slot") -- a byte per slot
//    for up to 8 lists, whatever the stream's length: 1 024 slots in the kilobyte that held 512 sixteen-bit counters, OR instead of
//    ADD (idempotent: the lanes beyond the stream's end repeat its last record instead of being masked), no overflow;
//    MODE 1: 16 lists, 512 halfword slots; MODE 2: any number of lists, 512 sixteen-bit counters as before;
//  * the slot of a clump is its low bits: the clump numbers of a list are unrelated, a multiply per record buys nothing;
//  * the exact lane table, the survivor ring and the emit work on 64 survivors / 4 table slots x 16 reference lanes at a time.
// No false negatives, as before: a record of a clump that >= need lists name finds >= need bits in its slot.
// BIG = 1: four times the slots and the lane table, for the second pass over queries that overflowed the first.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {      // maximum over the 64 lanes (all active), in every lane
	int v = (int)x, t;
	t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false); v = (uint32_t)t > (uint32_t)v ? t : v;    // row_shr:1
	t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false); v = (uint32_t)t > (uint32_t)v ? t : v;    // row_shr:2
	t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false); v = (uint32_t)t > (uint32_t)v ? t : v;    // row_shr:4
	t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false); v = (uint32_t)t > (uint32_t)v ? t : v;    // row_shr:8
	const uint32_t a = (uint32_t)__builtin_amdgcn_readlane(v, 15), b = (uint32_t)__builtin_amdgcn_readlane(v, 31),
		c = (uint32_t)__builtin_amdgcn_readlane(v, 47), d = (uint32_t)__builtin_amdgcn_readlane(v, 63);
	const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
	return ab > cd ? ab : cd;
}
template <int MODE, int BIG>
__global__ __launch_bounds__(64) void k_prefilter_cw(
		const uint2 *__restrict__ ranges, const uint2 *__restrict__ hdr, uint32_t W16, uint32_t n_list,
		const uint32_t *__restrict__ ent,   // 4-byte (clump, lane-set code) records
		const uint32_t *__restrict__ bad, uint32_t n_bad, const uint32_t *__restrict__ clump_len, uint32_t tot_refs,
		uint2 *__restrict__ tasks, uint32_t *__restrict__ n_tasks, uint32_t task_cap,
		unsigned long long *__restrict__ ent_read,
		uint32_t *__restrict__ fb_list, uint32_t *__restrict__ n_fb,
		unsigned long long *__restrict__ unit_sum, unsigned long long *__restrict__ col_sum, unsigned long long *__restrict__ qlen_sum,
		unsigned long long *__restrict__ surv_sum,
		uint2 *__restrict__ tasks2, uint32_t *__restrict__ n_tasks2, int prune,
		const uint32_t *__restrict__ sel, const uint32_t *__restrict__ n_sel_dev, int) {
	constexpr uint32_t FB = MODE == 0 ? 8u : 16u;                         // bits per slot
	constexpr uint32_t SB = MODE == 0 ? 2u : 1u;                          // log2 slots per dword
	constexpr uint32_t NDW = BIG ? 1024u : 256u;                          // dwords of slots: 1 KB (4 KB)
	constexpr uint32_t NS = NDW << SB;                                    // slots
	constexpr uint32_t LTB = BIG ? 8u : 6u, LT = 1u << LTB;               // exact lane-table slots
	constexpr uint32_t RING = 128u;                                       // <= 63 pending + 64 new survivors
	constexpr uint32_t CW_STAGE = 64u;
	constexpr uint32_t R = 6u;                                            // rows of 64 records that stay in registers between the two looks
	__shared__ __attribute__((aligned(16))) uint32_t s_cnt[NDW];
	__shared__ uint32_t s_key[LT];
	__shared__ unsigned long long s_lc[LT][2];
	__shared__ uint32_t s_ring[RING];
	__shared__ uint16_t s_lut[256];
	__shared__ uint8_t s_used[LT];
	__shared__ uint2 s_stage[2][CW_STAGE];
	const uint32_t lane = threadIdx.x, z = lane & 15u, sg = lane >> 4;
	for (uint32_t i = lane; i < 256; i += 64) s_lut[i] = (uint16_t)bhip_lane_code_mask(i);
	for (uint32_t i = lane; i < NDW; i += 64) s_cnt[i] = 0;
	for (uint32_t i = lane; i < LT; i += 64) { s_key[i] = 0; s_lc[i][0] = 0; s_lc[i][1] = 0; }
	__syncthreads();
	unsigned long long my_ent = 0, my_units = 0, my_qlen = 0, my_surv = 0;
	const unsigned long long lt_mask = (1ull << lane) - 1ull;
#ifdef PFM_PROF
	unsigned long long my_t[8] = {0,0,0,0,0,0,0,0}, t_last = wall_clock64();      // 0 lists + addresses + load issue, 1 first look (waits for the records), 2 second look, 3 survivor rounds, 4 emit, 5 clear, 6 loop top
#endif
	uint32_t nst[2] = {0u, 0u};
	auto flush_one = [&](uint32_t which) {
		const uint32_t n = nst[which];
		if (n) {
			uint32_t base = 0;
			if (lane == 0) base = atomicAdd(which ? n_tasks2 : n_tasks, n);
			base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
			uint2 *dst = which ? tasks2 : tasks;
			if (lane < n && base + lane < task_cap) dst[base + lane] = s_stage[which][lane];
			CF_WAVE_ORDER();
		}
		nst[which] = 0;
	};
	auto put = [&](uint32_t which, bool mine, uint32_t a, uint32_t b) {      // wave-uniform call; `mine`: this lane has a task for list `which`
		const unsigned long long m = __ballot(mine);
		const uint32_t cnt = (uint32_t)__popcll(m);
		if (!cnt) return;
		if (nst[which] + cnt > CW_STAGE) flush_one(which);
		if (mine) s_stage[which][nst[which] + (uint32_t)__popcll(m & lt_mask)] = make_uint2(a, b);
		nst[which] += cnt;
	};
	const uint32_t n_items = sel ? (*n_sel_dev < n_list ? *n_sel_dev : n_list) : n_list;
	const uint32_t Wc = W16 < 64u ? W16 : 64u;                            // lists of the first chunk (lane l holds list l)
	const uint32_t n_chunks = (W16 + 63u) >> 6;
	typedef const unsigned long long __attribute__((address_space(1))) *g64_t;
	auto fetch = [&](uint32_t q, unsigned long long &h, unsigned long long &r) {      // header and the first 64 list ranges of item q (clamped: always a valid address)
		const uint32_t qc = q < n_items ? q : 0u;
		const uint32_t lic = sel ? (n_items ? sel[qc] : 0u) : qc;
		h = ((g64_t)(uintptr_t)(hdr + lic))[0];
		r = ((g64_t)(uintptr_t)(ranges + ((size_t)lic * W16 + (lane < Wc ? lane : 0u))))[0];
	};
	unsigned long long h_n, r_n;
	fetch(blockIdx.x, h_n, r_n);
	for (uint32_t q = blockIdx.x; q < n_items; q += gridDim.x) {
		const uint32_t li = sel ? sel[q] : q;
		const uint2 hd = make_uint2((uint32_t)h_n, (uint32_t)(h_n >> 32));
		const unsigned long long r_c = r_n;
		fetch(q + gridDim.x, h_n, r_n);                                   // one query ahead: the only exposed latency of a query is its records'
		const uint32_t need = hd.x & 0xFFFFu, len = hd.y & 0xFFFu;
		const uint32_t budget = (hd.y >> 12) & 255u, dper = (hd.y >> 20) & 15u ? (hd.y >> 20) & 15u : 1u;
		const uint32_t thr = need ? need : 1u;
		uint32_t pend = 0, head = 0, nused = 0, ovf = 0;                  // wave-uniform
		PFM_T(6);
		// ---- the lists of one chunk: lane l = list l.  eend = end of the list in the chunk's flattened stream; ab = biased address:
		// the record at stream position i of list l is at ab_l + 4 i
		uint32_t eend; unsigned long long ab; uint32_t T;
		auto chunk_lists = [&](uint32_t c) {
			unsigned long long rr = r_c;
			if (c) { rr = 0; if (c * 64u + lane < W16) rr = ((const unsigned long long *)ranges)[(size_t)li * W16 + c * 64u + lane]; }
			const uint32_t rx = (uint32_t)rr, ry = (uint32_t)(rr >> 32);
			const uint32_t n = (c * 64u + lane < W16) ? ry & 0xFFFFFFu : 0u;
			const unsigned long long beg = (unsigned long long)rx | (unsigned long long)(ry >> 24) << 32;
			eend = wave_incl_scan_u32(n);
			T = (uint32_t)__builtin_amdgcn_readlane((int)eend, 63);
			ab = (unsigned long long)(uintptr_t)ent + 4ull * (beg - (unsigned long long)(eend - n));
			if (__builtin_amdgcn_readfirstlane((int)(eend < n))) T = 0xFFFFFFFFu;      // (never: 64 lists of < 2^24 entries)
		};
		// which list does stream position i belong to: the number of lists that end at or before it.  The ends are WAVE-UNIFORM (lane l
		// holds list l's): up to 8 lists, seven scalar boundaries and a compare + add each; beyond, a binary search over the lanes.  The
		// list's biased base address then comes from its lane (two cross-lane reads) -- no loop, no branch, the same for every row.
		uint32_t eb[7] = {0, 0, 0, 0, 0, 0, 0};
		auto list_ends = [&]() {
			if (MODE == 0) {
				#pragma unroll
				for (uint32_t j = 0; j < 7; ++j) eb[j] = (uint32_t)__builtin_amdgcn_readlane((int)eend, (int)j);
			}
		};
		auto row_addr = [&](uint32_t r, uint32_t &kreg) -> bhip_gptr_t {
			const uint32_t i = r * 64u + lane;
			const uint32_t ic = i < T ? i : T - 1u;                       // beyond the stream: its last record once more (OR is idempotent; the second look tests i < T)
			uint32_t kk = 0;
			if (MODE == 0) {
				#pragma unroll
				for (uint32_t j = 0; j < 7; ++j) kk += eb[j] <= ic ? 1u : 0u;
			} else {
				#pragma unroll
				for (uint32_t step = 32; step >= 1; step >>= 1) kk += (uint32_t)__shfl((int)eend, (int)(kk + step - 1u), 64) <= ic ? step : 0u;      // (lanes without a list end at T > ic)
			}
			const uint32_t a_lo = (uint32_t)__shfl((int)(uint32_t)ab, (int)kk, 64), a_hi = (uint32_t)__shfl((int)(uint32_t)(ab >> 32), (int)kk, 64);
			kreg = kk;
			return (bhip_gptr_t)(uintptr_t)(((unsigned long long)a_hi << 32 | a_lo) + 4ull * ic);
		};
		auto slot_dw = [&](uint32_t rec) -> uint32_t { return (rec & (NS - 1u)) >> SB; };
		auto slot_sh = [&](uint32_t rec) -> uint32_t { return (rec & ((1u << SB) - 1u)) * FB; };
		auto count1 = [&](uint32_t rec, uint32_t kreg, uint32_t i) {     // first look
			if (MODE == 2) atomicAdd(&s_cnt[slot_dw(rec)], (i < T ? 1u : 0u) << slot_sh(rec));
			else atomicOr(&s_cnt[slot_dw(rec)], 1u << (slot_sh(rec) + kreg));
		};
		auto c_round = [&]() {                                            // up to 64 survivors into the exact lane table
			const uint32_t take = pend < 64u ? pend : 64u;
			const bool active = lane < take;
			const uint32_t rec = active ? s_ring[(head + lane) & (RING - 1u)] : 0u;
			const uint32_t clump = rec & 0xFFFFFFu, key = clump + 1u, mask = s_lut[rec >> 24];
			uint32_t slot = (clump * 0x85EBCA6Bu) >> (32u - LTB);
			bool act = active, found = false, fresh = false;
			for (uint32_t probes = 0; __any(act) && probes < LT; ++probes) {
				uint32_t old = 0xFFFFFFFFu;
				if (act) old = atomicCAS(&s_key[slot], 0u, key);
				const bool ok = act && (old == 0u || old == key);
				fresh |= act && old == 0u;
				found |= ok;
				act = act && !ok;
				slot = act ? (slot + 1u) & (LT - 1u) : slot;
			}
			if (__any(act)) ovf = 1u;
			const unsigned long long mf = __ballot(fresh);
			if (fresh) s_used[nused + (uint32_t)__popcll(mf & lt_mask)] = (uint8_t)slot;
			nused += (uint32_t)__popcll(mf);
			if (found) {
				if (mask & 0xFFu) atomicAdd(&s_lc[slot][0], spread8(mask & 0xFFu));
				if (mask >> 8) atomicAdd(&s_lc[slot][1], spread8(mask >> 8));
			}
			head = (head + take) & (RING - 1u);
			pend -= take;
		};
		auto offer1 = [&](uint32_t rec, uint32_t i) {                     // second look: survivors of the slot test go to the ring
			const uint32_t f = (s_cnt[slot_dw(rec)] >> slot_sh(rec)) & ((1u << FB) - 1u);
			const bool surv = (MODE == 2 ? f : (uint32_t)__popc(f)) >= thr && i < T;
			const unsigned long long m = __ballot(surv);
			if (m) {
				if (surv) s_ring[(head + pend + (uint32_t)__popcll(m & lt_mask)) & (RING - 1u)] = rec;
				pend += (uint32_t)__popcll(m);
				my_surv += (uint32_t)__popcll(m);
				if (pend >= 64u) c_round();
			}
		};
		// ---- first look over every record; the first R rows of the first chunk stay in registers
		uint32_t rc[R], kr[R];
		uint32_t T0 = 0, rows0 = 0;
		unsigned long long gtot = 0;
		for (uint32_t c = 0; c < n_chunks; ++c) {
			chunk_lists(c);
			if (T == 0xFFFFFFFFu) { ovf = 1u; break; }
			gtot += T;
			const uint32_t rows = (T + 63u) >> 6;
			list_ends();
			uint32_t r0 = 0;
			if (c == 0) {
				T0 = T; rows0 = rows;
				#pragma unroll
				for (uint32_t r = 0; r < R; ++r) if (r < rows) rc[r] = row_addr(r, kr[r])[0];
				PFM_T(0);
				#pragma unroll
				for (uint32_t r = 0; r < R; ++r) if (r < rows) count1(rc[r], kr[r], r * 64u + lane);
				r0 = R;
			}
			for (uint32_t r = r0; r < rows; ++r) {
				uint32_t k;
				const uint32_t rec = row_addr(r, k)[0];
				count1(rec, k, r * 64u + lane);
			}
		}
		my_ent += gtot;
		if (MODE == 2 && gtot > 65535ull) ovf = 1u;
		CF_WAVE_ORDER();
		PFM_T(1);
		// ---- second look
		if (!ovf) for (uint32_t c = 0; c < n_chunks; ++c) {
			uint32_t rows, r0 = 0;
			if (c == 0) {
				T = T0; rows = rows0;
				#pragma unroll
				for (uint32_t r = 0; r < R; ++r) if (r < rows) offer1(rc[r], r * 64u + lane);
				r0 = R;
				if (rows > R) { chunk_lists(0); list_ends(); }      // (the lists again: later chunks have been through the registers)
			} else { chunk_lists(c); rows = (T + 63u) >> 6; list_ends(); }
			for (uint32_t r = r0; r < rows; ++r) {
				uint32_t k;
				const uint32_t rec = row_addr(r, k)[0];
				offer1(rec, r * 64u + lane);
			}
		}
		PFM_T(2);
		while (pend) c_round();
		CF_WAVE_ORDER();
		PFM_T(3);
		// ---- emit: four table slots x sixteen reference lanes per pass.  A lane with c matching words lost (W_valid - c) words, one edit
		// destroys at most `dper` of them: its edit distance is at least budget - (c - need) / dper.  Unless every hit within budget is
		// wanted, only the lanes with the query's largest count are swept at once; the others wait for the minimum those produce.
		if (!ovf) {
			const uint32_t inv_dper = 65536u / dper + 1u;                 // x / dper == (x * inv_dper) >> 16 for x < 256, dper < 16
			auto look = [&](uint32_t p, uint32_t &slot, uint32_t &c, uint32_t &cz) -> bool {
				const uint32_t iu = p * 4u + sg;
				const bool has = iu < nused;
				slot = has ? (uint32_t)s_used[iu] : 0u;
				c = s_key[slot] - 1u;
				cz = ((const uint8_t *)&s_lc[slot][0])[z];
				return has && c * 16u + z < tot_refs && cz >= thr;
			};
			auto emit_pass = [&](uint32_t p, bool ok, uint32_t slot, uint32_t c, uint32_t cz, uint32_t t0) {
				const bool has = p * 4u + sg < nused;
				CF_WAVE_ORDER();
				if (has && z < 2u) s_lc[slot][z] = 0;                     // (this wave's reads of the slot are done: LDS operations of one wave stay in order)
				if (has && z == 2u) s_key[slot] = 0;
				const bool first = ok && (!prune || cz >= t0);
				uint32_t lb = 0;
				if (prune) { const uint32_t gain = ((cz - need) * inv_dper) >> 16; lb = gain >= budget ? 0u : budget - gain; }
				put(0, first, li, c * 16u + z);
				put(1, ok && !first, li | lb << 24, c * 16u + z);
				const unsigned long long mo = __ballot(ok);
				const uint32_t units = ((mo & 0xFFFFull) ? 1u : 0u) + ((mo >> 16 & 0xFFFFull) ? 1u : 0u) + ((mo >> 32 & 0xFFFFull) ? 1u : 0u) + ((mo >> 48) ? 1u : 0u);
				my_units += units; my_qlen += (unsigned long long)units * len;
			};
			if (nused <= 4u) {                                            // the usual case: every used slot in one pass, looked at once
				uint32_t slot, c, cz;
				const bool ok = look(0, slot, c, cz);
				uint32_t t0 = thr;
				if (prune) { const uint32_t cm = wave_max_u32(ok ? cz : 0u); t0 = cm > thr ? cm : thr; }
				if (nused) emit_pass(0, ok, slot, c, cz, t0);
			} else {
				uint32_t t0 = thr;
				if (prune) {
					uint32_t cmax = 0;
					for (uint32_t p = 0; p * 4u < nused; ++p) { uint32_t sl, c, cz; if (look(p, sl, c, cz)) cmax = cz > cmax ? cz : cmax; }
					const uint32_t cm = wave_max_u32(cmax);
					t0 = cm > thr ? cm : thr;
				}
				for (uint32_t p = 0; p * 4u < nused; ++p) {
					uint32_t slot, c, cz;
					const bool ok = look(p, slot, c, cz);
					emit_pass(p, ok, slot, c, cz, t0);
				}
			}
			for (uint32_t i = 0; i < n_bad; i += 4) {                     // burst.c:4136-4138, 4282-4283: every lane of the ambiguous clumps
				const bool in = i + sg < n_bad;
				const uint32_t c = in ? bad[i + sg] : 0u;
				put(0, in && c * 16u + z < tot_refs, li, c * 16u + z);
				{ const uint32_t nb4 = n_bad - i < 4u ? n_bad - i : 4u; my_units += nb4; my_qlen += (unsigned long long)nb4 * len; }
			}
		} else {
			for (uint32_t i = lane; i < LT; i += 64) { s_key[i] = 0; s_lc[i][0] = 0; s_lc[i][1] = 0; }
			if (lane == 0) { const uint32_t pos = atomicAdd(n_fb, 1u); fb_list[pos] = li; }
		}
		PFM_T(4);
		{
			uint4 *cz4 = (uint4 *)&s_cnt[0];
			for (uint32_t i = lane; i < NDW / 4u; i += 64) cz4[i] = make_uint4(0, 0, 0, 0);
		}
		CF_WAVE_ORDER();
		PFM_T(5);
	}
	flush_one(0); flush_one(1);
#ifdef PFM_PROF
	if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_pfm_prof[i], my_t[i]);
#endif
	if (lane == 0) {
		if (ent_read && my_ent) atomicAdd(ent_read, my_ent);
		if (surv_sum && my_surv) atomicAdd(surv_sum, my_surv);
		if (my_units) { atomicAdd(unit_sum, my_units); atomicAdd(qlen_sum, my_qlen); }
	}
	(void)clump_len; (void)col_sum;
}
// ------------------------------------------------------------------------------------------------
// Lane-resolved prefilter, counting filter, FOUR queries per wave with the record streams walked by the WHOLE wave (round 5).
// k_prefilter_cw (one query per wave) showed two things on the device (PMC, gpurun_out/r05d): walking a query's stream with 64 lanes and
// wave-uniform list boundaries costs ~30 vector instructions per 64 records where k_prefilter_cf spends ~75 -- and everything ELSE a query
// needs (list scan, survivor insertion, emit, clearing: ~230 vector and ~250 scalar instructions) is then paid per query by a wave in which
// a handful of lanes do the work, which is why it loses to k_prefilter_cf on small databases (260 against 187 vector instructions per query
// at 35 records per read) and wins only 20 % at the metric's size.  This kernel keeps both halves where they are cheap:
//  * per QUAD of queries, group-parallel as in k_prefilter_cf (16 lanes per query): list lengths -> stream positions (row DPP scans), the
//    survivor rounds (16 survivors of each query per round), the emit (one exact-table slot of each query per pass, its 16 reference lanes
//    in the group's lanes), the table clears -- a quarter of the per-query cost;
//  * per QUERY of the quad, wave-parallel as in k_prefilter_cw: the list ends of the query become seven scalars (v_readlane from its
//    group), a row of 64 stream positions finds its list with a compare + add per boundary, the two looks at the records are the
//    list-mask slots of k_prefilter_cw (1 024 byte slots per query for up to 8 lists, OR instead of ADD, positions beyond the stream
//    repeat its last record), the records of the first R rows stay in registers between the looks.
// For lists per query <= 16 (MODE 0: <= 8, byte slots; MODE 1: halfword slots); longer plans keep k_prefilter_cw<2>.  BIG = 1: the second
// pass over queries whose survivors overflowed the 32-slot exact table of the first, with four times the slots and the table.
// ------------------------------------------------------------------------------------------------
#ifndef CQ_MINWAVES
#define CQ_MINWAVES 1          // waves per SIMD the register allocation aims at (tools/build_variant.sh: -DCQ_MINWAVES=5 -DCQ_LTB=4 -DCQ_STAGE_N=32 for the occupancy A/B)
#endif
#ifndef CQ_LTB
#define CQ_LTB 5
#endif
#ifndef CQ_STAGE_N
#define CQ_STAGE_N 64
#endif
template <int MODE, int BIG>
__global__ __launch_bounds__(64, CQ_MINWAVES) void k_prefilter_cq(
		const uint2 *__restrict__ ranges, const uint2 *__restrict__ hdr, uint32_t W16, uint32_t n_list,
		const uint32_t *__restrict__ ent,   // 4-byte (clump, lane-set code) records
		const uint32_t *__restrict__ bad, uint32_t n_bad, const uint32_t *__restrict__ clump_len, uint32_t tot_refs,
		uint2 *__restrict__ tasks, uint32_t *__restrict__ n_tasks, uint32_t task_cap,
		unsigned long long *__restrict__ ent_read,
		uint32_t *__restrict__ fb_list, uint32_t *__restrict__ n_fb,
		unsigned long long *__restrict__ unit_sum, unsigned long long *__restrict__ col_sum, unsigned long long *__restrict__ qlen_sum,
		unsigned long long *__restrict__ surv_sum,
		uint2 *__restrict__ tasks2, uint32_t *__restrict__ n_tasks2, int prune,
		const uint32_t *__restrict__ sel, const uint32_t *__restrict__ n_sel_dev, int) {
	constexpr uint32_t FB = MODE == 0 ? 8u : 16u;                         // bits per slot
	constexpr uint32_t SB = MODE == 0 ? 2u : 1u;                          // log2 slots per dword
	constexpr uint32_t NDW = BIG ? 1024u : 256u;                          // dwords of slots per query: 1 KB (4 KB)
	constexpr uint32_t NS = NDW << SB;                                    // slots per query
	constexpr uint32_t LTB = BIG ? 7u : (uint32_t)CQ_LTB, LT = 1u << LTB;               // exact lane-table slots per query
	constexpr uint32_t RING = 64u;                                        // survivors of a query waiting for the rounds at the end of the quad (a power of two, >= one row)
	constexpr uint32_t CQ_STAGE = (uint32_t)CQ_STAGE_N;
	constexpr uint32_t R = 6u;                                            // rows of 64 records of a query that stay in registers between the two looks
	__shared__ __attribute__((aligned(16))) uint32_t s_cnt[4][NDW];
	__shared__ uint32_t s_key[4][LT];
	__shared__ unsigned long long s_lc[4][LT][2];
	__shared__ uint32_t s_ring[4][RING];
	__shared__ uint16_t s_lut[256];
	__shared__ uint8_t s_used[4][LT];
	__shared__ uint2 s_stage[2][CQ_STAGE];
	__shared__ uint32_t s_dummy[16];          // compare-and-swap target of idle lanes (never written: the compare value cannot match)
	const uint32_t lane = threadIdx.x, g = lane >> 4, gl = lane & 15u;
	if (lane < 16) s_dummy[lane] = 0;
	for (uint32_t i = lane; i < 256; i += 64) s_lut[i] = (uint16_t)bhip_lane_code_mask(i);
	for (uint32_t i = lane; i < 4 * NDW; i += 64) (&s_cnt[0][0])[i] = 0;
	for (uint32_t i = lane; i < 4 * LT; i += 64) { (&s_key[0][0])[i] = 0; (&s_lc[0][0][0])[2 * i] = 0; (&s_lc[0][0][0])[2 * i + 1] = 0; }
	__syncthreads();
	uint32_t my_ent = 0, my_units = 0, my_qlen = 0, my_surv = 0;         // (per wave and launch: well inside 32 bits)
	const unsigned long long lt_mask = (1ull << lane) - 1ull;
#ifdef PFM_PROF
	unsigned long long my_t[8] = {0,0,0,0,0,0,0,0}, t_last = wall_clock64();      // 0 addresses + load issue, 1 first look (waits for the records), 2 second look, 3 survivor rounds, 4 emit, 5 clear, 6 quad setup
#endif
	uint32_t nst[2] = {0u, 0u};
	auto flush_one = [&](uint32_t which) {
		const uint32_t n = nst[which];
		if (n) {
			uint32_t base = 0;
			if (lane == 0) base = atomicAdd(which ? n_tasks2 : n_tasks, n);
			base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
			uint2 *dst = which ? tasks2 : tasks;
			if (lane < n && base + lane < task_cap) dst[base + lane] = s_stage[which][lane];
			CF_WAVE_ORDER();
		}
		nst[which] = 0;
	};
	auto put = [&](uint32_t which, bool mine, uint32_t a, uint32_t b) {      // wave-uniform call; `mine`: this lane has a task for list `which`
		const unsigned long long m = __ballot(mine);
		const uint32_t cnt = (uint32_t)__popcll(m);
		if (!cnt) return;
		if (cnt > CQ_STAGE) {                     // more than the stage holds in one go: straight to the list
			uint32_t base = 0;
			if (lane == 0) base = atomicAdd(which ? n_tasks2 : n_tasks, cnt);
			base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + (uint32_t)__popcll(m & lt_mask);
			if (mine && base < task_cap) (which ? tasks2 : tasks)[base] = make_uint2(a, b);
			return;
		}
		if (nst[which] + cnt > CQ_STAGE) flush_one(which);
		if (mine) s_stage[which][nst[which] + (uint32_t)__popcll(m & lt_mask)] = make_uint2(a, b);
		nst[which] += cnt;
	};
	auto group_scan = [&](uint32_t n) -> uint32_t {                       // inclusive prefix sum inside each group of 16 lanes
		int ps = (int)n;
		ps += __builtin_amdgcn_update_dpp(0, ps, 0x111, 0xF, 0xF, false);    // row_shr:1 (a row = the 16 lanes of a group; lanes without a source add 0)
		ps += __builtin_amdgcn_update_dpp(0, ps, 0x112, 0xF, 0xF, false);
		ps += __builtin_amdgcn_update_dpp(0, ps, 0x114, 0xF, 0xF, false);
		ps += __builtin_amdgcn_update_dpp(0, ps, 0x118, 0xF, 0xF, false);
		return (uint32_t)ps;
	};
	auto group_max = [&](uint32_t v) -> uint32_t {                        // maximum over the 16 lanes of the group, in every lane
		int t;
		t = __builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); v = (uint32_t)t > v ? (uint32_t)t : v;     // quad_perm:[1,0,3,2]
		t = __builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false); v = (uint32_t)t > v ? (uint32_t)t : v;     // quad_perm:[2,3,0,1]
		t = __builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false); v = (uint32_t)t > v ? (uint32_t)t : v;    // row_half_mirror
		t = __builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false); v = (uint32_t)t > v ? (uint32_t)t : v;    // row_mirror
		return v;
	};
	auto wave_max4 = [&](uint32_t v) -> uint32_t {                        // maximum over the four groups of a group-uniform value
		const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16),
			c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
		const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
		return ab > cd ? ab : cd;
	};
	auto spread4 = [](uint32_t nib) -> uint32_t { return (nib * 0x00204081u) & 0x01010101u; };      // bit i of a nibble -> bit 8 i
	const uint32_t n_items = sel ? (*n_sel_dev < n_list ? *n_sel_dev : n_list) : n_list;
	const uint32_t n_quads = (n_items + 3) >> 2;
	typedef const unsigned long long __attribute__((address_space(1))) *g64_t;
	auto fetch = [&](uint32_t quad, unsigned long long &h, unsigned long long &r) {      // header of this group's query and range gl of it (clamped: always a valid address)
		const uint32_t it = quad * 4 + g;
		const uint32_t itc = it < n_items ? it : 0u;
		const uint32_t lic = sel ? (n_items ? sel[itc] : 0u) : itc;
		h = ((g64_t)(uintptr_t)(hdr + lic))[0];
		r = ((g64_t)(uintptr_t)(ranges + ((size_t)lic * W16 + (gl < W16 ? gl : 0u))))[0];
	};
	unsigned long long h_n, r_n;
	fetch(blockIdx.x, h_n, r_n);
	for (uint32_t quad = blockIdx.x; quad < n_quads; quad += gridDim.x) {
		const bool live = quad * 4 + g < n_items;
		const uint32_t li = sel ? (live ? sel[quad * 4 + g] : 0u) : quad * 4 + g;      // list position of this group's query
		const uint2 hd = make_uint2((uint32_t)h_n, (uint32_t)(h_n >> 32));
		const unsigned long long r_c = r_n;
		fetch(quad + gridDim.x, h_n, r_n);                                // one quad ahead
		const uint32_t need = hd.x & 0xFFFFu, len = hd.y & 0xFFFu;
		const uint32_t budget = (hd.y >> 12) & 255u, dper = (hd.y >> 20) & 15u ? (hd.y >> 20) & 15u : 1u;
		const uint32_t thr = need ? need : 1u;                            // (group-uniform)
		// ---- the lists of the quad's queries: lane gl of group g = list gl of query g.  eend = end of the list in its query's flattened
		// stream; ab = biased address: the record at stream position i of the list is at ab + 4 i
		const uint32_t rx = (uint32_t)r_c, ry = (uint32_t)(r_c >> 32);
		const uint32_t n0 = (live && gl < W16) ? ry & 0xFFFFFFu : 0u;
		const unsigned long long beg = (unsigned long long)rx | (unsigned long long)(ry >> 24) << 32;
		const uint32_t eend = group_scan(n0);
		const unsigned long long ab = (unsigned long long)(uintptr_t)ent + 4ull * (beg - (unsigned long long)(eend - n0));
		uint32_t pend[4] = {0u, 0u, 0u, 0u};                              // survivors waiting in the queries' rings (wave-uniform)
		uint32_t nused = 0, ovf = 0;                                      // slots of this group's lane table in use / table overflow (replicated in the group)
		// ---- survivor rounds: every group moves up to 16 survivors of its query into its exact lane table
		auto drain = [&]() {
			uint32_t pv = g == 0 ? pend[0] : g == 1 ? pend[1] : g == 2 ? pend[2] : pend[3];
			uint32_t head = 0;
			while (__any(pv > 0)) {
				const uint32_t take = pv < 16u ? pv : 16u;
				const bool active = gl < take;
				const uint32_t rec = active ? s_ring[g][head + gl] : 0u;
				const uint32_t clump = rec & 0xFFFFFFu, key = clump + 1u, mask = s_lut[rec >> 24];
				uint32_t slot = (clump * 0x85EBCA6Bu) >> (32u - LTB);
				bool act = active, found = false, fresh = false;
				for (uint32_t probes = 0; __any(act) && probes < LT; ++probes) {
					const uint32_t old = atomicCAS(act ? &s_key[g][slot] : &s_dummy[gl], act ? 0u : 0xFFFFFFFFu, key);
					const bool ok = act && (old == 0u || old == key);
					fresh |= act && old == 0u;
					found |= ok;
					act = act && !ok;
					slot = act ? (slot + 1u) & (LT - 1u) : slot;
				}
				const uint32_t m_act = (uint32_t)(__ballot(act) >> (lane & 48u)) & 0xFFFFu;
				if (m_act) ovf = 1u;
				const uint32_t m16 = (uint32_t)(__ballot(fresh) >> (lane & 48u)) & 0xFFFFu;
				if (fresh) s_used[g][nused + __popc(m16 & ((1u << gl) - 1u))] = (uint8_t)slot;
				nused += __popc(m16);
				if (found) {
					const unsigned long long lo = (unsigned long long)spread4((mask >> 4) & 15u) << 32 | spread4(mask & 15u);
					const unsigned long long hi = (unsigned long long)spread4(mask >> 12) << 32 | spread4((mask >> 8) & 15u);
					if (lo) atomicAdd(&s_lc[g][slot][0], lo);
					if (hi) atomicAdd(&s_lc[g][slot][1], hi);
				}
				head += take; pv -= take;
			}
			pend[0] = pend[1] = pend[2] = pend[3] = 0;
		};
		PFM_T(6);
		// ---- the record streams, 64 stream positions per row.  The loads of ALL four queries are issued first (their gathers are in flight
		// together: a wave waits for memory once per quad, not once per query), then the first look over the four queries, then the second.
		constexpr uint32_t NB = MODE == 0 ? 7u : 15u, KB = MODE == 0 ? 3u : 4u;      // list ends that matter / bits of a list number
		uint32_t rc[4][R], krp[4];                                        // records of the resident rows, their list numbers (KB bits per row)
		uint32_t Tq[4];
		auto row_rec = [&](uint32_t q, uint32_t T, const uint32_t (&eb)[NB], uint32_t r, uint32_t &kreg) -> uint32_t {
			const uint32_t i = r * 64u + lane;
			const uint32_t ic = i < T ? i : T - 1u;                       // beyond the stream: its last record once more (OR is idempotent; the second look tests i < T)
			uint32_t kk = 0;
			#pragma unroll
			for (uint32_t j = 0; j < NB; ++j) kk += eb[j] <= ic ? 1u : 0u;          // lists that end at or before the position = its list
			const uint32_t src = q * 16u + kk;
			const uint32_t a_lo = (uint32_t)__shfl((int)(uint32_t)ab, (int)src, 64), a_hi = (uint32_t)__shfl((int)(uint32_t)(ab >> 32), (int)src, 64);
			kreg = kk;
			return ((bhip_gptr_t)(uintptr_t)(((unsigned long long)a_hi << 32 | a_lo) + 4ull * ic))[0];
		};
		auto list_ends = [&](uint32_t q, uint32_t (&eb)[NB]) {           // ends of the query's lists but the last: wave-uniform
			#pragma unroll
			for (uint32_t j = 0; j < NB; ++j) eb[j] = (uint32_t)__builtin_amdgcn_readlane((int)eend, (int)(q * 16u + j));
		};
		#pragma unroll
		for (uint32_t q = 0; q < 4; ++q) {
			const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)eend, (int)(q * 16u + 15u));
			Tq[q] = T; krp[q] = 0;
			if (T == 0u) continue;                                        // (wave-uniform)
			my_ent += T;
			uint32_t eb[NB];
			list_ends(q, eb);
			const uint32_t rows = (T + 63u) >> 6;
			#pragma unroll
			for (uint32_t r = 0; r < R; ++r) if (r < rows) { uint32_t kr; rc[q][r] = row_rec(q, T, eb, r, kr); krp[q] |= kr << (KB * r); }
		}
		PFM_T(0);
		auto count1 = [&](uint32_t q, uint32_t rec, uint32_t kreg) {     // first look: the record's list leaves its bit in the record's slot
			atomicOr(&s_cnt[q][(rec & (NS - 1u)) >> SB], 1u << ((rec & ((1u << SB) - 1u)) * FB + kreg));
		};
		#pragma unroll
		for (uint32_t q = 0; q < 4; ++q) {
			const uint32_t T = Tq[q];
			if (T == 0u) continue;
			const uint32_t rows = (T + 63u) >> 6;
			#pragma unroll
			for (uint32_t r = 0; r < R; ++r) if (r < rows) count1(q, rc[q][r], (krp[q] >> (KB * r)) & ((1u << KB) - 1u));
			if (rows > R) {                                               // (streams beyond R rows: loaded where they are looked at, twice)
				uint32_t eb[NB];
				list_ends(q, eb);
				for (uint32_t r = R; r < rows; ++r) { uint32_t kr; const uint32_t rec = row_rec(q, T, eb, r, kr); count1(q, rec, kr); }
			}
		}
		CF_WAVE_ORDER();
		PFM_T(1);
		#pragma unroll
		for (uint32_t q = 0; q < 4; ++q) {
			const uint32_t T = Tq[q];
			if (T == 0u) continue;
			const uint32_t rows = (T + 63u) >> 6;
			const uint32_t thr_q = (uint32_t)__builtin_amdgcn_readlane((int)thr, (int)(q * 16u));
			auto offer1 = [&](uint32_t rec, uint32_t i) {                 // second look: records whose slot names enough lists go to the query's ring
				const uint32_t f = (s_cnt[q][(rec & (NS - 1u)) >> SB] >> ((rec & ((1u << SB) - 1u)) * FB)) & ((1u << FB) - 1u);
				const bool surv = (uint32_t)__popc(f) >= thr_q && i < T;
				const unsigned long long m = __ballot(surv);
				if (m) {
					const uint32_t cnt = (uint32_t)__popcll(m);
					if (pend[q] + cnt > RING) drain();                    // (rare: the rings are drained at the end of every quad)
					if (surv) s_ring[q][pend[q] + (uint32_t)__popcll(m & lt_mask)] = rec;
					pend[q] += cnt;
					my_surv += cnt;
				}
			};
			#pragma unroll
			for (uint32_t r = 0; r < R; ++r) if (r < rows) offer1(rc[q][r], r * 64u + lane);
			if (rows > R) {
				uint32_t eb[NB];
				list_ends(q, eb);
				for (uint32_t r = R; r < rows; ++r) { uint32_t kr; const uint32_t rec = row_rec(q, T, eb, r, kr); offer1(rec, r * 64u + lane); }
			}
		}
		PFM_T(2);
		drain();
		CF_WAVE_ORDER();
		PFM_T(3);
		// ---- emit the lanes that reach the threshold, clear the tables.  Slot-parallel, as in k_prefilter_cf: lane gl of a group owns the
		// group's gl-th used slot (most used slots are false survivors without a single passing lane: a byte-parallel compare says so at
		// once).  The positions of a lane's tasks in the two staged lists come from ONE wave-wide prefix sum over the per-lane counts.
		// A lane with c matching words lost (W_valid - c) words, one edit destroys at most `dper` of them: its edit distance is at least
		// budget - (c - need) / dper.  Unless every hit within budget is wanted, only the lanes with the query's largest count are swept at
		// once; the others wait for the minimum those produce (k_task_filter).
		const bool em = live && !ovf;
		const uint32_t nu = em ? nused : 0u;
		const uint32_t nu_max = wave_max4(nu);
		const uint32_t inv_dper = 65536u / dper + 1u;                     // x / dper == (x * inv_dper) >> 16 for x < 256, dper < 16
		auto lanes_ge = [&](unsigned long long lo, unsigned long long hi, uint32_t t) -> uint32_t {      // 16-bit set of the slot's lane counters >= t (t < 128)
			const unsigned long long H = 0x8080808080808080ull, L1 = 0x0101010101010101ull, G = 0x0102040810204080ull;
			const unsigned long long tl = ((lo | H) - t * L1) & H, th = ((hi | H) - t * L1) & H;
			return (uint32_t)(((tl >> 7) * G) >> 56) | ((uint32_t)(((th >> 7) * G) >> 56) << 8);
		};
		auto lanes_ge_any = [&](unsigned long long lo, unsigned long long hi, uint32_t t) -> uint32_t {
			if (t < 128u) return lanes_ge(lo, hi, t);
			uint32_t m16 = 0;
			#pragma unroll
			for (uint32_t zz = 0; zz < 16; ++zz) m16 |= ((uint32_t)(((zz < 8 ? lo : hi) >> (8 * (zz & 7))) & 255u) >= t ? 1u : 0u) << zz;
			return m16;
		};
		auto look = [&](uint32_t iu, uint32_t &slot, uint32_t &c, unsigned long long &lo, unsigned long long &hi) -> uint32_t {
			const bool has = iu < nu;
			slot = has ? (uint32_t)s_used[g][iu] : 0u;
			c = s_key[g][slot] - 1u; lo = s_lc[g][slot][0]; hi = s_lc[g][slot][1];
			const uint32_t first = c * 16u, nv = first < tot_refs ? (tot_refs - first < 16u ? tot_refs - first : 16u) : 0u;     // lanes of the clump that exist
			return has ? lanes_ge_any(lo, hi, thr) & ((1u << nv) - 1u) : 0u;
		};
		auto byte_of = [&](unsigned long long lo, unsigned long long hi, uint32_t zz) -> uint32_t { return (uint32_t)((zz < 8 ? lo : hi) >> (8u * (zz & 7u))) & 255u; };
		uint32_t slot0, c0; unsigned long long lo0, hi0;
		const uint32_t m16_0 = look(gl, slot0, c0, lo0, hi0);
		uint32_t cmax_all = 0;
		if (prune) {
			uint32_t cmax = 0;
			for (uint32_t m = m16_0; m; m &= m - 1) { const uint32_t v = byte_of(lo0, hi0, (uint32_t)__builtin_ctz(m)); cmax = v > cmax ? v : cmax; }
			for (uint32_t iu0 = 16; iu0 < nu_max; iu0 += 16) {
				uint32_t sl, c; unsigned long long lo, hi;
				for (uint32_t m = look(iu0 + gl, sl, c, lo, hi); m; m &= m - 1) { const uint32_t v = byte_of(lo, hi, (uint32_t)__builtin_ctz(m)); cmax = v > cmax ? v : cmax; }
			}
			cmax_all = group_max(cmax);
		}
		PFM_T(7);
		auto emit_slots = [&](uint32_t iu, uint32_t slot, uint32_t c, unsigned long long lo, unsigned long long hi, uint32_t m16) {
			if (iu < nused) { s_key[g][slot] = 0; s_lc[g][slot][0] = 0; s_lc[g][slot][1] = 0; }     // (this wave's reads of the slot are done: LDS operations of one wave stay in order)
			const uint32_t m0 = prune ? m16 & lanes_ge_any(lo, hi, cmax_all > thr ? cmax_all : thr) : m16, m1 = m16 & ~m0;
			const uint32_t cnt = (uint32_t)__popc(m0) | (uint32_t)__popc(m1) << 16;
			const uint32_t incl = wave_incl_scan_u32(cnt), tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63), excl = incl - cnt;
			if (!tot) return;                         // wave-uniform
			const uint32_t tot0 = tot & 0xFFFFu, tot1 = tot >> 16;
			uint32_t p[2]; bool direct[2];
			#pragma unroll
			for (uint32_t w = 0; w < 2; ++w) {
				const uint32_t tw = w ? tot1 : tot0, ew = w ? excl >> 16 : excl & 0xFFFFu;
				direct[w] = false;
				if (tw && nst[w] + tw > CQ_STAGE) flush_one(w);
				if (tw > CQ_STAGE) {                  // more than the stage holds in one go: straight to the list
					uint32_t base = 0;
					if (lane == 0) base = atomicAdd(w ? n_tasks2 : n_tasks, tw);
					p[w] = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + ew; direct[w] = true;
				} else { p[w] = nst[w] + ew; nst[w] += tw; }
			}
			PFM_T(5);
			for (uint32_t m = m16; m; m &= m - 1) {
				const uint32_t zz = (uint32_t)__builtin_ctz(m), w = (m1 >> zz) & 1u;
				uint32_t lb = 0;
				if (prune) { const uint32_t gain = ((byte_of(lo, hi, zz) - need) * inv_dper) >> 16; lb = gain >= budget ? 0u : budget - gain; }
				const uint2 task = make_uint2(li | lb << 24, c * 16u + zz);
				const uint32_t pos = p[w]; p[w] = pos + 1;
				if (direct[w]) { if (pos < task_cap) (w ? tasks2 : tasks)[pos] = task; }
				else s_stage[w][pos] = task;
			}
			if (m16) { ++my_units; my_qlen += len; }       // (the swept columns of lane tasks are counted by the sweep: tcol_sum)
			PFM_T(4);
		};
		if (nu_max) emit_slots(gl, slot0, c0, lo0, hi0, m16_0);
		for (uint32_t iu0 = 16; iu0 < nu_max; iu0 += 16) {
			uint32_t sl, c; unsigned long long lo, hi;
			const uint32_t m16 = look(iu0 + gl, sl, c, lo, hi);
			emit_slots(iu0 + gl, sl, c, lo, hi, m16);
		}
		for (uint32_t i = 0; i < n_bad; ++i) {                            // burst.c:4136-4138, 4282-4283: every lane of the ambiguous clumps
			const uint32_t c = bad[i];
			put(0, em && c * 16u + gl < tot_refs, li, c * 16u + gl);
			if (em && gl == 0) { ++my_units; my_qlen += len; }
		}
		if (__any(ovf != 0u)) {
			if (ovf) {
				for (uint32_t i = gl; i < LT; i += 16) { s_key[g][i] = 0; s_lc[g][i][0] = 0; s_lc[g][i][1] = 0; }
				if (live && gl == 0) { const uint32_t pos = atomicAdd(n_fb, 1u); fb_list[pos] = li; }
			}
		}
		PFM_T(3);
		{
			uint4 *cz4 = (uint4 *)&s_cnt[0][0];
			for (uint32_t i = lane; i < 4u * NDW / 4u; i += 64) cz4[i] = make_uint4(0, 0, 0, 0);
		}
		CF_WAVE_ORDER();
		PFM_T(5);
	}
	flush_one(0); flush_one(1);
#ifdef PFM_PROF
	if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_pfm_prof[i], my_t[i]);
#endif
	if (lane == 0) {
		if (ent_read && my_ent) atomicAdd(ent_read, (unsigned long long)my_ent);
		if (surv_sum && my_surv) atomicAdd(surv_sum, (unsigned long long)my_surv);
	}
	if (my_units) { atomicAdd(unit_sum, (unsigned long long)my_units); atomicAdd(qlen_sum, (unsigned long long)my_qlen); }
	(void)clump_len; (void)col_sum;
}
#define BHIP_INST_PFCQ(M, B) \
	template __global__ void k_prefilter_cq<M, B>(const uint2 *, const uint2 *, uint32_t, uint32_t, const uint32_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t, \
		uint2 *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, unsigned long long *, unsigned long long *, \
		uint2 *, uint32_t *, int, const uint32_t *, const uint32_t *, int);
BHIP_INST_PFCQ(0, 0) BHIP_INST_PFCQ(1, 0) BHIP_INST_PFCQ(0, 1) BHIP_INST_PFCQ(1, 1)

#define BHIP_INST_PFCW(M, B) \
	template __global__ void k_prefilter_cw<M, B>(const uint2 *, const uint2 *, uint32_t, uint32_t, const uint32_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t, \
		uint2 *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, unsigned long long *, unsigned long long *, \
		uint2 *, uint32_t *, int, const uint32_t *, const uint32_t *, int);
BHIP_INST_PFCW(0, 0) BHIP_INST_PFCW(1, 0) BHIP_INST_PFCW(2, 0) BHIP_INST_PFCW(0, 1) BHIP_INST_PFCW(1, 1) BHIP_INST_PFCW(2, 1)

template __global__ void k_prefilter_mask<9>(const uint2 *, const uint2 *, uint32_t, uint32_t, const uint32_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t,
	uint2 *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, unsigned long long *, uint2 *, uint32_t *, uint32_t);
template __global__ void k_prefilter_mask<10>(const uint2 *, const uint2 *, uint32_t, uint32_t, const uint32_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t,
	uint2 *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, unsigned long long *, uint2 *, uint32_t *, uint32_t);
template __global__ void k_prefilter_mask<11>(const uint2 *, const uint2 *, uint32_t, uint32_t, const uint32_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t,
	uint2 *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, unsigned long long *, uint2 *, uint32_t *, uint32_t);

// ------------------------------------------------------------------------------------------------
// Bit-parallel semi-global edit distance (Myers 1999 / Hyyro 2003), NW x 32-bit words per DP column.
// State per (query, reference lane): vertical deltas Pv/Mv of the current column; the tracked score is
// D[m][x] = min over start positions of the edit distance of the query against ref[..x], i.e. the last-row
// cell of the reference's recurrence (burst.c:1155-1159 with D[0][x]=0, D[y][0]=y).  Its minimum over x is
// MinA[lane] of aded_mat16 (burst.c:1187-1192).  Pad symbols (code 0) match nothing but filler rows;
// trailing pads cannot lower the minimum.
// ------------------------------------------------------------------------------------------------
// Measured issue rates on gfx950 (tools/ubench/valu_rate.hip): v_and/v_or/v_xor/v_bitop3/v_add_u32 issue every ~2 cycles
// per SIMD, while v_lshlrev/v_lshrrev/v_alignbit/v_bfi/v_add3/v_min3/v_addc_co are half rate.  So the one-bit shifts of the
// horizontal deltas are done as X + X (a carry chain across words), whose final carry-out IS the top bit the score
// update needs -- no shift instruction at all in the single-word case.
template <int NW>
__device__ __forceinline__ void myers_step(const uint32_t (&Eq)[NW], uint32_t (&Pv)[NW], uint32_t (&Mv)[NW], int &score) {
	uint32_t Ph[NW], Mh[NW];
	uint32_t carry = 0;
	#pragma unroll
	for (int w = 0; w < NW; ++w) {
		uint32_t co;
		const uint32_t s = __builtin_addc(Eq[w] & Pv[w], Pv[w], carry, &co);
		carry = co;
		const uint32_t Xh = (s ^ Pv[w]) | Eq[w];
		Ph[w] = Mv[w] | ~(Xh | Pv[w]);
		Mh[w] = Pv[w] & Xh;
	}
	// shift the horizontal deltas down one row (delta entering the first row is 0: free start); the carry out of the
	// last word is the delta of the last query row, i.e. the change of D[m][x]
	uint32_t cP = 0, cM = 0;
	#pragma unroll
	for (int w = 0; w < NW; ++w) {
		uint32_t co;
		Ph[w] = __builtin_addc(Ph[w], Ph[w], cP, &co); cP = co;
		Mh[w] = __builtin_addc(Mh[w], Mh[w], cM, &co); cM = co;
	}
	score += (int)cP - (int)cM;
	#pragma unroll
	for (int w = 0; w < NW; ++w) {
		const uint32_t Xv = Eq[w] | Mv[w];
		Pv[w] = Mh[w] | ~(Xv | Ph[w]);
		Mv[w] = Ph[w] & Xv;
	}
}

template <int NW>
__global__ __launch_bounds__(256) void k_myers(
		const uint2 *__restrict__ pairs, const uint32_t *__restrict__ n_pairs_dev, uint64_t n_pairs_host,   // device count is clamped to n_pairs_host (buffer capacity)
		uint32_t n_clumps_implicit,        // pairs == nullptr: p -> (list position li_base + p / n_clumps, clump p % n_clumps)
		uint32_t li_base,
		const uint32_t *__restrict__ qlist, // list position -> batch query index (nullptr: identity)
		const uint32_t *__restrict__ peq, const uint64_t *__restrict__ qoff, const uint16_t *__restrict__ qemac,
		const uint32_t *__restrict__ qsix,
		const uint4 *__restrict__ ref, const uint64_t *__restrict__ ref_off, const uint32_t *__restrict__ clump_len,
		uint32_t tot_refs,
		BhipRawHit *__restrict__ raw, uint32_t *__restrict__ n_raw, uint32_t raw_cap, uint32_t *__restrict__ best,
		uint8_t *__restrict__ mins_out, unsigned long long *__restrict__ col_sum, unsigned long long *__restrict__ qlen_sum) {
	// per 16-lane group: 16 symbol rows of NW words; rows rotated by 4*(group&3) so that the two groups a
	// ds_read_b128 service set can mix never collide on the A/C/G/T rows
	__shared__ __attribute__((aligned(16))) uint32_t s_peq[16][16 * NW];
	const uint32_t tid = threadIdx.x, g = tid >> 4, z = tid & 15, rot = 4 * (g & 3);
	const uint64_t n_pairs = n_pairs_dev ? ((uint64_t)*n_pairs_dev < n_pairs_host ? (uint64_t)*n_pairs_dev : n_pairs_host) : n_pairs_host;
	const uint64_t n_tiles = (n_pairs + 15) >> 4;
	unsigned long long my_cols = 0, my_qlen = 0;
	for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint64_t p = tile * 16 + g;
		const bool live = p < n_pairs;
		uint32_t li = 0, c = 0;
		if (live) {
			if (pairs) { const uint2 pr = pairs[p]; li = pr.x; c = pr.y; }
			else { li = li_base + (uint32_t)(p / n_clumps_implicit); c = (uint32_t)(p % n_clumps_implicit); }
		}
		__syncthreads();   // the previous tile's tables are no longer in use
		if (live) {
			const uint32_t *src = peq + ((uint64_t)li * 16 + z) * NW;
			uint32_t *dstrow = &s_peq[g][((z + rot) & 15) * NW];
			#pragma unroll
			for (int w = 0; w < NW; ++w) dstrow[w] = src[w];
		}
		__syncthreads();
		if (!live) continue;
		const uint32_t q = qlist ? qlist[li] : li;
		const uint32_t m = (uint32_t)(qoff[q + 1] - qoff[q]), E = qemac[q];
		const uint32_t L = clump_len[c], nchunks = (L + 31) >> 5;
		// column 0: D[y][0] = y on the query rows (delta +1), 0 on the filler rows
		uint32_t Pv[NW], Mv[NW];
		#pragma unroll
		for (int w = 0; w < NW; ++w) {
			const int lo = 32 * NW - (int)m - 32 * w;   // first query bit within this word
			Pv[w] = lo <= 0 ? 0xFFFFFFFFu : (lo >= 32 ? 0u : (0xFFFFFFFFu << lo));
			Mv[w] = 0;
		}
		int score = (int)m, bestS = 0x7FFFFFFF;
		uint32_t first = 0, last = 0;
		const uint4 *rp = ref + ref_off[c] * 16 + (uint64_t)z * nchunks;      // lane-major: [clump][lane][chunk]
		const uint32_t *tab = &s_peq[g][0];
		for (uint32_t t = 0; t < nchunks; ++t) {
			const uint4 ch = rp[t];
			const uint32_t dw[4] = {ch.x, ch.y, ch.z, ch.w};
			#pragma unroll
			for (int i = 0; i < 32; ++i) {
				const uint32_t sym = (dw[i >> 3] >> (4 * (i & 7))) & 15u;
				const uint32_t *row = tab + ((sym + rot) & 15u) * NW;
				uint32_t Eq[NW];
				#pragma unroll
				for (int w = 0; w < NW; ++w) Eq[w] = row[w];
				myers_step<NW>(Eq, Pv, Mv, score);
				const uint32_t col = t * 32 + i + 1;
				const bool lt = score < bestS, le = score <= bestS;
				bestS = lt ? score : bestS;
				first = lt ? col : first;
				last = le ? col : last;
			}
		}
		const uint32_t refIx = c * 16 + z;
		const bool hit = (uint32_t)bestS <= E && refIx < tot_refs;
		if (mins_out) mins_out[p * 16 + z] = (uint32_t)bestS <= E ? (uint8_t)bestS : (uint8_t)255;
		if (hit && raw) {
			const uint32_t pos = atomicAdd(n_raw, 1u);
			if (pos < raw_cap) {
				BhipRawHit h; h.q = q; h.refIx = refIx; h.ed = (uint32_t)bestS; h.e_first = first; h.e_last = last;
				h.m = m; h.L = L; h.six = qsix ? qsix[q] : q; h.rbase = ref_off[c] * 16 + (uint64_t)z * nchunks;
				raw[pos] = h;
			}
			if (best) atomicMin(&best[qsix ? qsix[q] : q], (uint32_t)bestS);
		}
		if (z == 0) { my_cols += L; my_qlen += m; }
	}
	if (col_sum && my_cols) { atomicAdd(col_sum, my_cols); atomicAdd(qlen_sum, my_qlen); }
}

// Queries beyond 1 024 symbols (up to BHIP_MAX_QLEN): the same recurrence with the vertical deltas of a (query, reference lane) pair in
// LDS -- NW words each of Pv and Mv per thread, [word][thread] so that a wave's access to one word is one conflict-free row -- and the
// match rows read from the profile table in global memory (a pair's 16 lanes read the same few rows: L1 hits).  One wave = four pairs.
// Single stage, every column of the clump: long queries are rare, the kernel is there so that they are aligned at all.
__global__ __launch_bounds__(64) void k_myers_long(
		const uint2 *__restrict__ pairs, const uint32_t *__restrict__ n_pairs_dev, uint64_t n_pairs_host, uint32_t n_clumps_implicit, uint32_t li_base,
		const uint32_t *__restrict__ qlist, const uint32_t *__restrict__ peq, const uint64_t *__restrict__ qoff, const uint16_t *__restrict__ qemac,
		const uint32_t *__restrict__ qsix, const uint4 *__restrict__ ref, const uint64_t *__restrict__ ref_off, const uint32_t *__restrict__ clump_len,
		uint32_t tot_refs, BhipRawHit *__restrict__ raw, uint32_t *__restrict__ n_raw, uint32_t raw_cap, uint32_t *__restrict__ best,
		uint8_t *__restrict__ mins_out, unsigned long long *__restrict__ col_sum, unsigned long long *__restrict__ qlen_sum, uint32_t NW) {
	extern __shared__ uint32_t s_long[];                      // Pv[NW][64] | Mv[NW][64]
	const uint32_t tid = threadIdx.x, g = tid >> 4, z = tid & 15;
	uint32_t *sP = s_long + tid, *sM = s_long + (size_t)NW * 64 + tid;
	const uint64_t n_pairs = n_pairs_dev ? ((uint64_t)*n_pairs_dev < n_pairs_host ? (uint64_t)*n_pairs_dev : n_pairs_host) : n_pairs_host;
	const uint64_t n_tiles = (n_pairs + 3) >> 2;
	unsigned long long my_cols = 0, my_qlen = 0;
	for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint64_t p = tile * 4 + g;
		if (p >= n_pairs) continue;                          // (no barriers in this kernel: a thread only ever touches its own LDS column)
		uint32_t li, c;
		if (pairs) { const uint2 pr = pairs[p]; li = pr.x; c = pr.y; }
		else { li = li_base + (uint32_t)(p / n_clumps_implicit); c = (uint32_t)(p % n_clumps_implicit); }
		const uint32_t q = qlist ? qlist[li] : li;
		const uint32_t m = (uint32_t)(qoff[q + 1] - qoff[q]), E = qemac[q];
		const uint32_t L = clump_len[c], nchunks = (L + 31) >> 5;
		for (uint32_t w = 0; w < NW; ++w) {                  // column 0: D[y][0] = y on the query rows, 0 on the filler rows below them
			const int lo = 32 * (int)NW - (int)m - 32 * (int)w;
			sP[w * 64] = lo <= 0 ? 0xFFFFFFFFu : (lo >= 32 ? 0u : (0xFFFFFFFFu << lo));
			sM[w * 64] = 0;
		}
		int score = (int)m, bestS = 0x7FFFFFFF;
		uint32_t first = 0, last = 0;
		const uint4 *rp = ref + ref_off[c] * 16 + (uint64_t)z * nchunks;
		const uint32_t *tab = peq + (uint64_t)li * 16 * NW;
		for (uint32_t t = 0; t < nchunks; ++t) {
			const uint4 ch = rp[t];
			#pragma unroll 1
			for (uint32_t i = 0; i < 32; ++i) {
				const uint32_t d = i < 8 ? ch.x : i < 16 ? ch.y : i < 24 ? ch.z : ch.w;
				const uint32_t *row = tab + ((d >> (4 * (i & 7))) & 15u) * NW;
				uint32_t carry = 0, cP = 0, cM = 0;
				for (uint32_t w = 0; w < NW; ++w) {          // one pass: the three carries all run from word 0 upwards
					const uint32_t Eq = row[w], Pv = sP[w * 64], Mv = sM[w * 64];
					uint32_t co;
					const uint32_t s = __builtin_addc(Eq & Pv, Pv, carry, &co); carry = co;
					const uint32_t Xh = (s ^ Pv) | Eq;
					uint32_t Ph = Mv | ~(Xh | Pv), Mh = Pv & Xh;
					Ph = __builtin_addc(Ph, Ph, cP, &co); cP = co;
					Mh = __builtin_addc(Mh, Mh, cM, &co); cM = co;
					const uint32_t Xv = Eq | Mv;
					sP[w * 64] = Mh | ~(Xv | Ph);
					sM[w * 64] = Ph & Xv;
				}
				score += (int)cP - (int)cM;
				const uint32_t col = t * 32 + i + 1;
				const bool lt = score < bestS, le = score <= bestS;
				bestS = lt ? score : bestS;
				first = lt ? col : first;
				last = le ? col : last;
			}
		}
		const uint32_t refIx = c * 16 + z;
		const bool hit = (uint32_t)bestS <= E && refIx < tot_refs;
		if (mins_out) mins_out[p * 16 + z] = (uint32_t)bestS <= E ? (uint8_t)bestS : (uint8_t)255;
		if (hit && raw) {
			const uint32_t pos = atomicAdd(n_raw, 1u);
			if (pos < raw_cap) {
				BhipRawHit h; h.q = q; h.refIx = refIx; h.ed = (uint32_t)bestS; h.e_first = first; h.e_last = last;
				h.m = m; h.L = L; h.six = qsix ? qsix[q] : q; h.rbase = ref_off[c] * 16 + (uint64_t)z * nchunks;
				raw[pos] = h;
			}
			if (best) atomicMin(&best[qsix ? qsix[q] : q], (uint32_t)bestS);
		}
		if (z == 0) { my_cols += L; my_qlen += m; }
	}
	if (col_sum && my_cols) { atomicAdd(col_sum, my_cols); atomicAdd(qlen_sum, my_qlen); }
}

// ------------------------------------------------------------------------------------------------
// Two-stage edit distance.  An alignment of the whole query with <= E edits contains an alignment of its first
// P symbols with <= E edits, ending at some column x.  Stage A (k_myers_prefix<NWP>) therefore sweeps every column of
// every candidate lane with a P = 32*NWP-symbol bit-vector only (~2.5x fewer instructions per column than the
// 4-word kernel) and records, per lane, which 32-column chunks contain a column with prefix score <= E.  Stage B
// (k_myers_window<NW>) runs the full-length recurrence only over the columns an alignment through a flagged chunk can
// touch: [first flagged column - P - E, last flagged column + (m - P) + E].  Every end column with D[m][x] <= E lies
// in that window with its whole alignment, so ed / first / last end column are identical to the full sweep.
// ------------------------------------------------------------------------------------------------
template <int NWP>
__global__ __launch_bounds__(256) void k_myers_prefix(
		const uint2 *__restrict__ pairs, const uint32_t *__restrict__ n_pairs_dev, uint64_t n_pairs_host,
		uint32_t n_clumps_implicit, uint32_t li_base,
		const uint32_t *__restrict__ qlist, const uint32_t *__restrict__ peqp, const uint64_t *__restrict__ qoff,
		const uint16_t *__restrict__ qemac,
		const uint4 *__restrict__ ref, const uint64_t *__restrict__ ref_off, const uint32_t *__restrict__ clump_len,
		uint32_t tot_refs, BhipWin *__restrict__ wins, uint32_t *__restrict__ n_wins, uint32_t win_cap,
		unsigned long long *__restrict__ col_sum, unsigned long long *__restrict__ qlen_sum, uint32_t *__restrict__ cls_seen,
		const uint32_t *__restrict__ qsix) {
	__shared__ __attribute__((aligned(16))) uint32_t s_peq[16][16 * NWP];
	const uint32_t tid = threadIdx.x, g = tid >> 4, z = tid & 15;
	const uint64_t n_pairs = n_pairs_dev ? ((uint64_t)*n_pairs_dev < n_pairs_host ? (uint64_t)*n_pairs_dev : n_pairs_host) : n_pairs_host;
	const uint64_t n_tiles = (n_pairs + 15) >> 4;
	unsigned long long my_cols = 0, my_qlen = 0;
	for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const uint64_t p = tile * 16 + g;
		const bool live = p < n_pairs;
		uint32_t li = 0, c = 0;
		if (live) {
			if (pairs) { const uint2 pr = pairs[p]; li = pr.x; c = pr.y; }
			else { li = li_base + (uint32_t)(p / n_clumps_implicit); c = (uint32_t)(p % n_clumps_implicit); }
		}
		__syncthreads();
		if (live) {
			const uint32_t *src = peqp + ((uint64_t)li * 16 + z) * NWP;
			#pragma unroll
			for (int w = 0; w < NWP; ++w) s_peq[g][z * NWP + w] = src[w];
		}
		__syncthreads();
		if (!live) continue;
		const uint32_t q = qlist ? qlist[li] : li;
		const uint32_t m = (uint32_t)(qoff[q + 1] - qoff[q]), E = qemac[q];
		const uint32_t P = m < 32u * NWP ? m : 32u * NWP;
		const uint32_t L = clump_len[c], nchunks = (L + 31) >> 5;
		uint32_t Pv[NWP], Mv[NWP];
		#pragma unroll
		for (int w = 0; w < NWP; ++w) {
			const int lo = 32 * NWP - (int)P - 32 * w;
			Pv[w] = lo <= 0 ? 0xFFFFFFFFu : (lo >= 32 ? 0u : (0xFFFFFFFFu << lo));
			Mv[w] = 0;
		}
		int score = (int)P;
		uint32_t g_first = 0xFFFFFFFFu, g_last = 0;
		const uint4 *rp = ref + ref_off[c] * 16 + (uint64_t)z * nchunks;      // lane-major: [clump][lane][chunk]
		const uint32_t *tab = &s_peq[g][0];
		for (uint32_t t = 0; t < nchunks; ++t) {
			const uint4 ch = rp[t];
			const uint32_t dw[4] = {ch.x, ch.y, ch.z, ch.w};
			#pragma unroll
			for (int i8 = 0; i8 < 4; ++i8) {
				int cmin = 0x7FFFFFFF;
				#pragma unroll
				for (int i = 8 * i8; i < 8 * i8 + 8; ++i) {
					const uint32_t sym = (dw[i >> 3] >> (4 * (i & 7))) & 15u;
					uint32_t Eq[NWP];
					#pragma unroll
					for (int w = 0; w < NWP; ++w) Eq[w] = tab[sym * NWP + w];
					myers_step<NWP>(Eq, Pv, Mv, score);
					cmin = score < cmin ? score : cmin;
				}
				const bool fl = (uint32_t)cmin <= E;
				const uint32_t gi = t * 4 + (uint32_t)i8;
				g_first = fl && gi < g_first ? gi : g_first;
				g_last = fl ? gi : g_last;
			}
		}
		const uint32_t refIx = c * 16 + z;
		if (g_first != 0xFFFFFFFFu && refIx < tot_refs) {
			const uint32_t pos = atomicAdd(n_wins, 1u);
			if (pos < win_cap) {
				const uint32_t wc = bhip_win_class(g_first, g_last, E);
				if (wc && !__builtin_nontemporal_load(&cls_seen[wc])) cls_seen[wc] = 1;
				BhipWin w; w.li = li; w.refIx = refIx; w.g_first = g_first | wc << 30; w.g_last = g_last;
				w.q = q; w.mE = m | E << 16; w.nchunks = nchunks; w.L = L; w.rbase = ref_off[c] * 16 + (uint64_t)z * nchunks; w.six = qsix ? qsix[q] : q; w.pad2 = 0;
				wins[pos] = w;
			}
		}
		if (z == 0) { my_cols += L; my_qlen += m; }
	}
	if (col_sum && my_cols) { atomicAdd(col_sum, my_cols); atomicAdd(qlen_sum, my_qlen); }
}

// Stage A over lane TASKS (list position, reference lane) from k_prefilter_mask: one thread per task, each with its own
// 16-row prefix table in LDS ([row][thread] layout: conflict-free for any symbol mix).  Same recurrence and flags as
// k_myers_prefix.
template <int NWP>
__global__ __launch_bounds__(64) void k_myers_prefix_task(
		const uint2 *__restrict__ tasks, const uint32_t *__restrict__ n_tasks_dev, uint32_t task_cap,
		const uint32_t *__restrict__ qlist, const uint32_t *__restrict__ peqp, const uint64_t *__restrict__ qoff,
		const uint16_t *__restrict__ qemac,
		const uint4 *__restrict__ ref, const uint64_t *__restrict__ ref_off, const uint32_t *__restrict__ clump_len,
		BhipWin *__restrict__ wins, uint32_t *__restrict__ n_wins, uint32_t win_cap, unsigned long long *__restrict__ tcol_sum,
		uint32_t *__restrict__ cls_seen, const uint4 *__restrict__ qmeta, const uint32_t *__restrict__ qsix) {      // qmeta (from k_seed_ranges) replaces qlist / qoff / qemac / qsix
	// NWP <= 2: the 16-row prefix table of the task sits in a private LDS column; wider prefixes would leave room for only
	// 2-3 waves per SIMD that way, so they read the rows from global memory (L1/L2 hits, one dwordx2/x4 load per column)
	constexpr bool LDS_TAB = NWP <= 2;
	__shared__ uint32_t s_peq[LDS_TAB ? 16 * NWP : 1][64];
	const uint32_t tid = threadIdx.x;
	uint32_t n = *n_tasks_dev;
	if (n > task_cap) n = task_cap;
	unsigned long long my_cols = 0;
	for (uint32_t i0 = blockIdx.x * 64; i0 < n; i0 += gridDim.x * 64) {
		const uint32_t i = i0 + tid;
		const bool live = i < n;
		uint2 tk = make_uint2(0, 0);
		if (live) {
			tk = tasks[i];
			tk.x &= 0xFFFFFFu;                 // the top byte is the lower bound used by k_task_filter
			if (LDS_TAB) {
				const uint32_t *src = peqp + (uint64_t)tk.x * 16 * NWP;
				#pragma unroll
				for (int r = 0; r < (LDS_TAB ? 16 * NWP : 1); ++r) s_peq[r][tid] = src[r];
			}
		}
		if (!live) continue;        // the table column is private to the thread: no barrier needed
		const uint32_t li = tk.x, refIx = tk.y, c = refIx >> 4, z = refIx & 15;
		uint32_t q, m, E, six;
		if (qmeta) { const uint4 qm = qmeta[li]; q = qm.x; m = qm.y & 0xFFFFu; E = qm.y >> 16; six = qm.z; }
		else { q = qlist ? qlist[li] : li; m = (uint32_t)(qoff[q + 1] - qoff[q]); E = qemac[q]; six = qsix ? qsix[q] : q; }
		const uint32_t P = m < 32u * NWP ? m : 32u * NWP;
		const uint32_t L = clump_len[c], nchunks = (L + 31) >> 5;
		uint32_t Pv[NWP], Mv[NWP];
		#pragma unroll
		for (int w = 0; w < NWP; ++w) {
			const int lo = 32 * NWP - (int)P - 32 * w;
			Pv[w] = lo <= 0 ? 0xFFFFFFFFu : (lo >= 32 ? 0u : (0xFFFFFFFFu << lo));
			Mv[w] = 0;
		}
		int score = (int)P;
		uint32_t g_first = 0xFFFFFFFFu, g_last = 0;
		const uint4 *rp = ref + ref_off[c] * 16 + (uint64_t)z * nchunks;        // lane-major copy
		const uint32_t *gtab = peqp + (uint64_t)li * 16 * NWP;
		for (uint32_t t0 = 0; t0 < nchunks; t0 += 4) {      // 4 chunks = 64 contiguous bytes of this lane per round of loads
			uint4 chs[4];
			#pragma unroll
			for (uint32_t u = 0; u < 4; ++u) chs[u] = t0 + u < nchunks ? rp[t0 + u] : make_uint4(0, 0, 0, 0);
			#pragma unroll
			for (uint32_t u = 0; u < 4; ++u) {
				const uint32_t t = t0 + u;
				if (t >= nchunks) break;
				const uint32_t dw[4] = {chs[u].x, chs[u].y, chs[u].z, chs[u].w};
				#pragma unroll
				for (int k8 = 0; k8 < 4; ++k8) {      // per dword of symbols: the flagged range is kept at a granularity of 8 columns
					int cmin = 0x7FFFFFFF;
					#pragma unroll
					for (int k = 8 * k8; k < 8 * k8 + 8; ++k) {
						const uint32_t sym = (dw[k >> 3] >> (4 * (k & 7))) & 15u;
						uint32_t Eq[NWP];
						#pragma unroll
						for (int w = 0; w < NWP; ++w) Eq[w] = LDS_TAB ? s_peq[LDS_TAB ? sym * NWP + w : 0][tid] : gtab[sym * NWP + w];
						myers_step<NWP>(Eq, Pv, Mv, score);
						cmin = score < cmin ? score : cmin;
					}
					const bool fl = (uint32_t)cmin <= E;
					const uint32_t gi = t * 4 + (uint32_t)k8;
					g_first = fl && gi < g_first ? gi : g_first;
					g_last = fl ? gi : g_last;
				}
			}
		}
		if (g_first != 0xFFFFFFFFu) {
			const uint32_t pos = atomicAdd(n_wins, 1u);
			if (pos < win_cap) {
				const uint32_t wc = bhip_win_class(g_first, g_last, E);
				if (wc && !__builtin_nontemporal_load(&cls_seen[wc])) cls_seen[wc] = 1;
				BhipWin w; w.li = li; w.refIx = refIx; w.g_first = g_first | wc << 30; w.g_last = g_last;
				w.q = q; w.mE = m | E << 16; w.nchunks = nchunks; w.L = L; w.rbase = (uint64_t)(rp - ref); w.six = six; w.pad2 = 0;
				wins[pos] = w;
			}
		}
		my_cols += L;
	}
	if (tcol_sum && my_cols) atomicAdd(tcol_sum, my_cols);
}
// Deferred lane tasks (li | bound << 24, refIx): keep those whose lower bound does not exceed the minimum edit distance
// found so far for their shared slot (equal bounds stay: ties are hits too).  One reservation per wave.
__global__ __launch_bounds__(256) void k_task_filter(const uint2 *__restrict__ in, const uint32_t *__restrict__ n_in_dev, uint32_t cap,
		const uint32_t *__restrict__ qlist, const uint32_t *__restrict__ qsix, const uint32_t *__restrict__ best,
		uint2 *__restrict__ out, uint32_t *__restrict__ n_out) {
	__shared__ uint32_t s_n, s_base;
	uint32_t n = *n_in_dev;
	if (n > cap) n = cap;
	const uint32_t tid = threadIdx.x;
	for (uint32_t chunk = blockIdx.x * 2048u; chunk < n; chunk += gridDim.x * 2048u) {      // one global reservation per 2048 tasks
		if (tid == 0) s_n = 0;
		__syncthreads();
		uint2 tk[8]; uint32_t rank[8]; bool keep[8];
		#pragma unroll
		for (int t = 0; t < 8; ++t) {
			const uint32_t i = chunk + (uint32_t)t * 256u + tid;
			keep[t] = false; rank[t] = 0; tk[t] = make_uint2(0, 0);
			if (i < n) {
				tk[t] = in[i];
				const uint32_t li = tk[t].x & 0xFFFFFFu, q = qlist ? qlist[li] : li;
				keep[t] = (tk[t].x >> 24) <= best[qsix ? qsix[q] : q];
			}
		}
		#pragma unroll
		for (int t = 0; t < 8; ++t) if (keep[t]) rank[t] = atomicAdd(&s_n, 1u);
		__syncthreads();
		if (tid == 0 && s_n) s_base = atomicAdd(n_out, s_n);
		__syncthreads();
		#pragma unroll
		for (int t = 0; t < 8; ++t) if (keep[t]) out[s_base + rank[t]] = tk[t];
		__syncthreads();
	}
}

#define BHIP_INST_PREFIX_TASK(NWP) \
	template __global__ void k_myers_prefix_task<NWP>(const uint2 *, const uint32_t *, uint32_t, const uint32_t *, const uint32_t *, const uint64_t *, \
		const uint16_t *, const uint4 *, const uint64_t *, const uint32_t *, BhipWin *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, const uint4 *, const uint32_t *);
BHIP_INST_PREFIX_TASK(1) BHIP_INST_PREFIX_TASK(2) BHIP_INST_PREFIX_TASK(3) BHIP_INST_PREFIX_TASK(4) BHIP_INST_PREFIX_TASK(6)

template <int NW>
__global__ __launch_bounds__(256) void k_myers_window(
		const BhipWin *__restrict__ wins, const uint32_t *__restrict__ n_wins_dev, uint32_t win_cap, int NWP, int min_class,
		const uint32_t *__restrict__ peq, const uint32_t *__restrict__ qsix, const uint4 *__restrict__ ref,
		BhipRawHit *__restrict__ raw, uint32_t *__restrict__ n_raw, uint32_t raw_cap, uint32_t *__restrict__ best,
		unsigned long long *__restrict__ wcol_sum, const uint32_t *__restrict__ cls_seen) {
	if (min_class > 0) {      // no window of a class left to this kernel was flagged in this call: nothing to look for
		uint32_t any = 0;
		for (int c = min_class; c < 4; ++c) any |= cls_seen[c];
		if (!any) return;
	}
	// NW <= 8: the profile rows of A, C, G, T (all a reference without IUPAC codes ever asks for) sit in a private LDS column
	// (64-thread blocks, 16 * NW bytes per thread); the other twelve rows stay in global memory.  Read from global memory
	// alone the tables fight over the 32 KB L1 (the kernel then ran fastest at 2 of 8 possible waves per SIMD).
	// min_class: the windows k_myers_window_band<2 .. 4> take (band class below min_class) are passed over.
	constexpr bool LDS_TAB = NW <= 8;
	__shared__ uint32_t s_tab[LDS_TAB ? 4 * NW : 1][LDS_TAB ? 64 : 1];
	uint32_t n = *n_wins_dev;
	if (n > win_cap) n = win_cap;
	unsigned long long my_cols = 0;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const BhipWin w = wins[i];
		if ((int)(w.g_first >> 30) < min_class) continue;
		const uint32_t li = w.li, q = w.q;
		const uint32_t m = w.mE & 0xFFFFu, E = w.mE >> 16;
		const uint32_t P = m < 32u * (uint32_t)NWP ? m : 32u * (uint32_t)NWP;
		const uint32_t nchunks = w.nchunks, gA8 = w.g_first & BHIP_WIN_GMASK;
		// prefix ends lie in the 1-based columns 8 g_first + 1 .. 8 g_last + 8: an alignment ending its prefix there starts no earlier
		// than P + E - 1 columns before and ends no later than (m - P) + E columns behind
		const int col_lo = (int)(gA8 * 8 + 2) - (int)(P + E), col_hi = (int)((w.g_last + 1) * 8 + (m - P) + E);
		// swept columns [c_lo, c_hi] (0-based), at a granularity of 8 (one dword of reference symbols): a fresh column state is a
		// valid start anywhere (free start of the semi-global alignment), so nothing before the first needed column is swept
		const uint32_t c_lo = col_lo > 1 ? (uint32_t)(col_lo - 1) : 0u;
		uint32_t c_hi = (uint32_t)(col_hi - 1);
		if (c_hi >= nchunks * 32) c_hi = nchunks * 32 - 1;
		const uint32_t tA = c_lo >> 5, tB = c_hi >> 5, gA = (c_lo & 31u) >> 3, gE = (c_hi & 31u) >> 3;
		uint32_t Pv[NW], Mv[NW];
		#pragma unroll
		for (int k = 0; k < NW; ++k) {
			const int lo = 32 * NW - (int)m - 32 * k;
			Pv[k] = lo <= 0 ? 0xFFFFFFFFu : (lo >= 32 ? 0u : (0xFFFFFFFFu << lo));
			Mv[k] = 0;
		}
		int score = (int)m, bestS = 0x7FFFFFFF;
		uint32_t first = 0, last = 0;
		const uint4 *rp = ref + w.rbase;        // lane-major copy
		const uint32_t *tab = peq + (uint64_t)li * 16 * NW;
		if (LDS_TAB) {          // rows of the codes 1..4 = dwords NW .. 5 * NW - 1 of the table
			#pragma unroll
			for (int r = 0; r < (LDS_TAB ? 4 * NW : 1); ++r) s_tab[LDS_TAB ? r : 0][LDS_TAB ? threadIdx.x : 0] = tab[NW + r];
		}
		uint4 ch_next = rp[tA];
		for (uint32_t t = tA; t <= tB; ++t) {
			const uint4 ch = ch_next;
			if (t < tB) ch_next = rp[t + 1];          // the next 16 bytes of this lane while this chunk is swept
			const uint32_t g0 = t == tA ? gA : 0u, g1 = t == tB ? gE : 3u;
			for (uint32_t gq = g0; gq <= g1; ++gq) {
				const uint32_t d = gq == 0 ? ch.x : gq == 1 ? ch.y : gq == 2 ? ch.z : ch.w;
				#pragma unroll
				for (int k8 = 0; k8 < 8; ++k8) {
					const uint32_t sym = (d >> (4 * k8)) & 15u;
					uint32_t Eq[NW];
					if (LDS_TAB && sym - 1u < 4u) {
						#pragma unroll
						for (int x = 0; x < NW; ++x) Eq[x] = s_tab[LDS_TAB ? (sym - 1u) * NW + x : 0][LDS_TAB ? threadIdx.x : 0];
					} else {
						#pragma unroll
						for (int x = 0; x < NW; ++x) Eq[x] = tab[sym * NW + x];
					}
					myers_step<NW>(Eq, Pv, Mv, score);
					const uint32_t col = t * 32 + gq * 8 + (uint32_t)k8 + 1;
					const bool lt = score < bestS, le = score <= bestS;
					bestS = lt ? score : bestS;
					first = lt ? col : first;
					last = le ? col : last;
				}
			}
		}
		my_cols += ((tB * 4 + gE) - (tA * 4 + gA) + 1) * 8;
		if ((uint32_t)bestS <= E) {
			const uint32_t pos = atomicAdd(n_raw, 1u);
			if (pos < raw_cap) {
				BhipRawHit h; h.q = q; h.refIx = w.refIx; h.ed = (uint32_t)bestS; h.e_first = first; h.e_last = last;
				h.m = m; h.L = w.L; h.six = w.six; h.rbase = w.rbase;
				raw[pos] = h;
			}
			if (best) atomicMin(&best[w.six], (uint32_t)bestS);
		}
	}
	if (wcol_sum && my_cols) atomicAdd(wcol_sum, my_cols);
}

// The same sweep for the windows whose flagged diagonals fit a band of BW words (bhip_win_class == BW - 2), whatever the query
// length class NW > BW.
// Band: a cell (P, x) of an alignment within E edits has x in the flagged columns, so it lies on a diagonal x - y in
// [8 g_first + 1 - P, 8 g_last + 8 - P], and every other cell of that alignment within E diagonals of it: dmax - dmin + 1 =
// 8 (g_last - g_first) + 8 + 2 E diagonals.  While at most 32 (BW - 1) - 6 of them, the rows any such alignment touches
// within one dword of reference symbols (8 columns, the band climbs 8 rows) fit in BW consecutive words of the column: only
// those are stepped, and they move up one word when the band has left the lowest one.  The row under the lowest word is
// taken to grow by one per column, and a word not reached yet keeps its column-0 deltas (+1 per row): both are upper bounds
// of the true cells, so every cell outside the band is over-estimated, every cell an alignment within E edits passes through
// is exact, and (minimum, first and last end column) come out as from the full column whenever the minimum is <= E -- which
// is all that is ever reported.  The A/C/G/T rows of the BW words sit in LDS (4 BW dwords per thread); the rows of the word
// above wait in registers, loaded one move ahead, so that a move never waits for memory.
template <int BW>
__global__ __launch_bounds__(64) void k_myers_window_band(
		const BhipWin *__restrict__ wins, const uint32_t *__restrict__ n_wins_dev, uint32_t win_cap, int NWP, int NW,
		const uint32_t *__restrict__ peq, const uint32_t *__restrict__ qsix, const uint4 *__restrict__ ref,
		BhipRawHit *__restrict__ raw, uint32_t *__restrict__ n_raw, uint32_t raw_cap, uint32_t *__restrict__ best,
		unsigned long long *__restrict__ wcol_sum, const uint32_t *__restrict__ cls_seen) {
	if (BW > 2 && !cls_seen[BW - 2]) return;
	__shared__ uint32_t s_tab[4 * BW][64];          // [BW * (code - 1) + word of the band][thread]
	const uint32_t tid = threadIdx.x;
	uint32_t n = *n_wins_dev;
	if (n > win_cap) n = win_cap;
	unsigned long long my_cols = 0;
	for (uint32_t i = blockIdx.x * 64u + tid; i < n; i += gridDim.x * 64u) {
		const BhipWin w = wins[i];
		if (w.g_first >> 30 != (uint32_t)(BW - 2)) continue;
		const uint32_t m = w.mE & 0xFFFFu, E = w.mE >> 16, g_first = w.g_first & BHIP_WIN_GMASK;
		const uint32_t P = m < 32u * (uint32_t)NWP ? m : 32u * (uint32_t)NWP;
		const uint32_t nchunks = w.nchunks;
		const int col_lo = (int)(g_first * 8 + 2) - (int)(P + E), col_hi = (int)((w.g_last + 1) * 8 + (m - P) + E);
		const uint32_t c_lo = col_lo > 1 ? (uint32_t)(col_lo - 1) : 0u;
		uint32_t c_hi = (uint32_t)(col_hi - 1);
		if (c_hi >= nchunks * 32) c_hi = nchunks * 32 - 1;
		const uint32_t jA = c_lo >> 3, jB = c_hi >> 3, tB = jB >> 2;      // dwords of 8 symbols / last chunk of 32
		const int dmax = (int)(w.g_last * 8 + 8) - (int)P + (int)E;
		const int shift = 32 * NW - (int)m;           // filler rows under the query (bit of query row y = shift + y - 1)
		int ylo = (int)(jA * 8 + 1) - dmax; if (ylo < 0) ylo = 0;      // lowest row needed in the first column (0: the free-start row)
		int wb = (shift - 1 + ylo) >> 5;
		wb = wb < 0 ? 0 : (wb > NW - BW ? NW - BW : wb);
		auto init_word = [&](int k) -> uint32_t { const int lo = shift - 32 * k; return lo <= 0 ? 0xFFFFFFFFu : (lo >= 32 ? 0u : (0xFFFFFFFFu << lo)); };
		uint32_t Pb[BW], Mb[BW];
		#pragma unroll
		for (int k = 0; k < BW; ++k) { Pb[k] = init_word(wb + k); Mb[k] = 0; }
		int sc = 32 * (wb + BW) - shift; sc = sc < 0 ? 0 : sc;   // D at the top row of the highest word, one column before the first
		uint32_t hb = 32 * wb > shift ? 1u : 0u;                 // the row under the lowest word is a query row: +1 per column
		bool top = wb == NW - BW;                                // the last query row is in the band's words: its score is compared
		int bestS = 0x7FFFFFFF;
		uint32_t first = 0, last = 0;
		const uint4 *rp = ref + w.rbase;
		const uint32_t *tab = peq + (uint64_t)w.li * 16 * (uint32_t)NW;
		uint32_t nx[4];                              // the rows of the word above, fetched a move ahead
		#pragma unroll
		for (int r = 0; r < 4; ++r) {
			#pragma unroll
			for (int k = 0; k < BW; ++k) s_tab[BW * r + k][tid] = tab[(r + 1) * NW + wb + k];
			nx[r] = tab[(r + 1) * NW + (wb + BW < NW ? wb + BW : NW - 1)];
		}
		// track: the highest word is the query's last, so sc is D[m][col] -- before that there is nothing to compare
		auto step = [&](auto track, const uint32_t (&Eq)[BW], uint32_t col) {
			uint32_t Ph[BW], Mh[BW];
			uint32_t carry = 0;
			#pragma unroll
			for (int k = 0; k < BW; ++k) {
				uint32_t co;
				const uint32_t sum = __builtin_addc(Eq[k] & Pb[k], Pb[k], carry, &co);
				carry = co;
				const uint32_t Xh = (sum ^ Pb[k]) | Eq[k];
				Ph[k] = Mb[k] | ~(Xh | Pb[k]);
				Mh[k] = Pb[k] & Xh;
			}
			uint32_t cP = 0, cM = 0;
			#pragma unroll
			for (int k = 0; k < BW; ++k) {
				uint32_t co;
				Ph[k] = __builtin_addc(Ph[k], Ph[k], cP, &co); cP = co;
				Mh[k] = __builtin_addc(Mh[k], Mh[k], cM, &co); cM = co;
			}
			Ph[0] |= hb;
			sc += (int)cP - (int)cM;
			#pragma unroll
			for (int k = 0; k < BW; ++k) {
				const uint32_t Xv = Eq[k] | Mb[k];
				Pb[k] = Mh[k] | ~(Xv | Ph[k]);
				Mb[k] = Ph[k] & Xv;
			}
			if constexpr (decltype(track)::value) {
				const bool lt = sc < bestS, le = sc <= bestS;
				bestS = lt ? sc : bestS;
				first = lt ? col : first;
				last = le ? col : last;
			}
		};
		auto sweep8 = [&](auto track, uint32_t d, uint32_t col0) {
			const uint32_t dm = d - 0x11111111u;
			if ((dm & 0xCCCCCCCCu) == 0) {       // eight of A, C, G, T: rows from LDS, no test per symbol
				#pragma unroll
				for (int k8 = 0; k8 < 8; ++k8) {
					const uint32_t r = ((dm >> (4 * k8)) & 3u) * (uint32_t)BW;
					uint32_t Eq[BW];
					#pragma unroll
					for (int k = 0; k < BW; ++k) Eq[k] = s_tab[r + k][tid];
					step(track, Eq, col0 + (uint32_t)k8);
				}
			} else {
				#pragma unroll
				for (int k8 = 0; k8 < 8; ++k8) {
					const uint32_t sym = (d >> (4 * k8)) & 15u;
					uint32_t Eq[BW];
					#pragma unroll
					for (int k = 0; k < BW; ++k) Eq[k] = tab[sym * NW + wb + k];
					step(track, Eq, col0 + (uint32_t)k8);
				}
			}
		};
		uint32_t tcur = jA >> 2;
		// the dword in turn is always c0: the chunk is a shift register of four scalars (a component of a uint4 picked by j & 3, or
		// shifted in place, sent both chunks to scratch memory); n0..n3 = the next 16 bytes of this lane, loaded a chunk ahead
		uint32_t c0, c1, c2, c3, n0, n1, n2, n3;
		{ const uint4 v = rp[tcur]; c0 = v.x; c1 = v.y; c2 = v.z; c3 = v.w; }
		{ const uint4 v = tcur < tB ? rp[tcur + 1] : make_uint4(0, 0, 0, 0); n0 = v.x; n1 = v.y; n2 = v.z; n3 = v.w; }
		for (uint32_t r = 0; r < (jA & 3u); ++r) { c0 = c1; c1 = c2; c2 = c3; }
		for (uint32_t j = jA; j <= jB; ++j) {
			const uint32_t d = c0;
			const uint32_t col0 = j * 8 + 1;
			if (wb < NW - BW && (int)col0 - dmax + shift - 1 >= 32 * (wb + 1)) {     // the band has left the lowest word
				++wb;
				#pragma unroll
				for (int k = 0; k + 1 < BW; ++k) { Pb[k] = Pb[k + 1]; Mb[k] = Mb[k + 1]; }
				Pb[BW - 1] = init_word(wb + BW - 1); Mb[BW - 1] = 0;
				sc += __popc(Pb[BW - 1]);
				hb = 32 * wb > shift ? 1u : 0u;
				top = wb == NW - BW;
				#pragma unroll
				for (int r = 0; r < 4; ++r) {
					#pragma unroll
					for (int k = 0; k + 1 < BW; ++k) s_tab[BW * r + k][tid] = s_tab[BW * r + k + 1][tid];
					s_tab[BW * r + BW - 1][tid] = nx[r];
					nx[r] = tab[(r + 1) * NW + (wb + BW < NW ? wb + BW : NW - 1)];
				}
			}
			if (top) sweep8(std::true_type(), d, col0); else sweep8(std::false_type(), d, col0);
			if ((j & 3u) == 3u) {
				c0 = n0; c1 = n1; c2 = n2; c3 = n3; ++tcur;
				if (tcur < tB) { const uint4 v = rp[tcur + 1]; n0 = v.x; n1 = v.y; n2 = v.z; n3 = v.w; }
			} else { c0 = c1; c1 = c2; c2 = c3; }
		}
		my_cols += (jB - jA + 1) * 8;
		if ((uint32_t)bestS <= E) {
			const uint32_t pos = atomicAdd(n_raw, 1u);
			if (pos < raw_cap) {
				BhipRawHit h; h.q = w.q; h.refIx = w.refIx; h.ed = (uint32_t)bestS; h.e_first = first; h.e_last = last;
				h.m = m; h.L = w.L; h.six = w.six; h.rbase = w.rbase;
				raw[pos] = h;
			}
			if (best) atomicMin(&best[w.six], (uint32_t)bestS);
		}
	}
	if (wcol_sum && my_cols) atomicAdd(wcol_sum, my_cols);
}
#define BHIP_INST_BAND(BW) \
	template __global__ void k_myers_window_band<BW>(const BhipWin *, const uint32_t *, uint32_t, int, int, const uint32_t *, const uint32_t *, \
		const uint4 *, BhipRawHit *, uint32_t *, uint32_t, uint32_t *, unsigned long long *, const uint32_t *);
BHIP_INST_BAND(2) BHIP_INST_BAND(3) BHIP_INST_BAND(4)

#define BHIP_INST_PREFIX(NWP) \
	template __global__ void k_myers_prefix<NWP>(const uint2 *, const uint32_t *, uint64_t, uint32_t, uint32_t, const uint32_t *, const uint32_t *, \
		const uint64_t *, const uint16_t *, const uint4 *, const uint64_t *, const uint32_t *, uint32_t, BhipWin *, uint32_t *, uint32_t, \
		unsigned long long *, unsigned long long *, uint32_t *, const uint32_t *);
BHIP_INST_PREFIX(1) BHIP_INST_PREFIX(2) BHIP_INST_PREFIX(3) BHIP_INST_PREFIX(4) BHIP_INST_PREFIX(6)
#define BHIP_INST_WINDOW(NW) \
	template __global__ void k_myers_window<NW>(const BhipWin *, const uint32_t *, uint32_t, int, int, const uint32_t *, const uint32_t *, \
		const uint4 *, BhipRawHit *, uint32_t *, uint32_t, uint32_t *, unsigned long long *, const uint32_t *);
BHIP_INST_WINDOW(2) BHIP_INST_WINDOW(4) BHIP_INST_WINDOW(6) BHIP_INST_WINDOW(8) BHIP_INST_WINDOW(10) BHIP_INST_WINDOW(16) BHIP_INST_WINDOW(32)

#define BHIP_INST_MYERS(NW) \
	template __global__ void k_myers<NW>(const uint2 *, const uint32_t *, uint64_t, uint32_t, uint32_t, const uint32_t *, const uint32_t *, \
		const uint64_t *, const uint16_t *, const uint32_t *, const uint4 *, const uint64_t *, const uint32_t *, uint32_t, \
		BhipRawHit *, uint32_t *, uint32_t, uint32_t *, uint8_t *, unsigned long long *, unsigned long long *);
BHIP_INST_MYERS(2) BHIP_INST_MYERS(4) BHIP_INST_MYERS(6) BHIP_INST_MYERS(8) BHIP_INST_MYERS(10)
BHIP_INST_MYERS(16) BHIP_INST_MYERS(32)

// ------------------------------------------------------------------------------------------------
// Re-scoring (reScoreM_mat16, burst.c:713-886; scalar spec in SURVEY.md Appendix C).  One thread per hit.
// Only the diagonals x - y in [e_first - m - B, e_last - m + B] can hold an ancestor of a final cell with
// score <= B (every gap costs 1), so the three planes (score D, gapQ "shift" H, gapR "shiftR" V) are kept
// for that band only, one packed word (D | H<<8 | V<<16) per diagonal, updated in place row by row:
//   diag pred = band[k] (previous row), up pred = band[k+1] (previous row), left pred = carried register.
// Cells outside the band or the matrix count as 255; cells >= B+1 are forced to 255 (burst.c:802-803).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sat8u(uint32_t v) { return v > 255u ? 255u : v; }

// 32 consecutive reference symbols of one lane (chunk t4 of the lane-major copy); zeros outside the clump
__device__ __forceinline__ uint4 ref_chunk_lane(const uint32_t *__restrict__ refw_lane, uint64_t clump_base, uint32_t z, int t4, uint32_t nchunks) {
	if (t4 < 0 || (uint32_t)t4 >= nchunks) return make_uint4(0, 0, 0, 0);
	return ((const uint4 *)refw_lane)[clump_base * 16 + (uint64_t)z * nchunks + (uint32_t)t4];
}
// the same two reads from the lane's own address (BhipRawHit::rbase, in uint4 units)
__device__ __forceinline__ uint4 ref_chunk_at(const uint32_t *__restrict__ refw_lane, uint64_t rbase, int t4, uint32_t nchunks) {
	if (t4 < 0 || (uint32_t)t4 >= nchunks) return make_uint4(0, 0, 0, 0);
	return ((const uint4 *)refw_lane)[rbase + (uint32_t)t4];
}
__device__ __forceinline__ uint32_t ref_dword_at(const uint32_t *__restrict__ refw_lane, uint64_t rbase, int j8, uint32_t nchunks) {
	if (j8 < 0 || (uint32_t)j8 >= nchunks * 4) return 0u;
	return refw_lane[rbase * 4 + (uint32_t)j8];
}
__device__ __forceinline__ uint32_t pick4(const uint4 v, uint32_t i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ uint32_t ref_dword_lane(const uint32_t *__restrict__ refw_lane, uint64_t clump_base, uint32_t z, int j8, uint32_t nchunks) {
	if (j8 < 0 || (uint32_t)j8 >= nchunks * 4) return 0u;
	return refw_lane[(clump_base * 16 + (uint64_t)z * nchunks) * 4 + (uint32_t)j8];
}

// 4-bit packing of the queries at a fixed stride of qw dwords per query (k_rescore preloads them into LDS)
__global__ void k_pack_queries(const uint8_t *__restrict__ qcodes, const uint64_t *__restrict__ qoff, uint32_t n_q, uint32_t qw,
                               uint32_t *__restrict__ qpack) {
	// one thread per output dword = 8 symbols: their 8 bytes come as three aligned dwords and two byte-funnel shifts (eight byte loads
	// per thread made this kernel the slowest of the staging: 0.40 ms for 2 M reads), the low nibbles are squeezed together with
	// three shift-or-mask steps per half
	const uint64_t total = (uint64_t)n_q * qw;
	const uint32_t head = (uint32_t)((uintptr_t)qcodes & 3u);
	const uint32_t *cw = (const uint32_t *)(qcodes - head);
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
		uint32_t q, j;
		if (total <= 0xFFFFFFFFull) { q = (uint32_t)i / qw; j = (uint32_t)i - q * qw; } else { q = (uint32_t)(i / qw); j = (uint32_t)(i % qw); }
		const uint64_t b = qoff[q];
		const uint32_t len = (uint32_t)(qoff[q + 1] - b);
		uint32_t v = 0;
		if (8 * j < len) {
			const uint32_t nsym = len - 8 * j < 8 ? len - 8 * j : 8u;
			const uint64_t a = b + 8ull * j + head;              // byte address counted from cw
			const uint64_t w = a >> 2;
			const uint32_t sh = (uint32_t)a & 3u, last = (uint32_t)((a + nsym - 1) >> 2) - (uint32_t)w;      // dwords beyond the first that hold a needed byte
			const uint32_t w0 = cw[w], w1 = last >= 1 ? cw[w + 1] : 0u, w2 = last >= 2 ? cw[w + 2] : 0u;
			uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, sh), hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
			lo &= 0x0F0F0F0Fu; lo = (lo | lo >> 4) & 0x00FF00FFu; lo = (lo | lo >> 8) & 0xFFFFu;
			hi &= 0x0F0F0F0Fu; hi = (hi | hi >> 4) & 0x00FF00FFu; hi = (hi | hi >> 8) & 0xFFFFu;
			v = lo | hi << 16;
			if (nsym < 8) v &= (1u << (4 * nsym)) - 1u;
		}
		qpack[i] = v;
	}
}

// ------------------------------------------------------------------------------------------------
// Staging of a batch on the device (the host only enqueues copies): k_span_fill turns the offsets of a caller's span of
// entries into batch offsets, shared slots and reported query numbers; k_route does what the host pass of round 1 did per
// entry -- length class, sub-pipeline, prefilter or exhaustive route, seed plan (stride and guaranteed count, see
// k_prefilter_wave above) -- and leaves per-list counts and maxima in a BhipStageInfo; a stable 8-bit radix sort of the
// entry numbers by key then yields every (lane, class) list in entry order.
// ------------------------------------------------------------------------------------------------
__global__ void k_span_fill(const uint64_t *__restrict__ off_raw, uint32_t n, uint32_t ebase, uint64_t pos_base, uint32_t q_base,
                            uint64_t *__restrict__ qoff, uint32_t *__restrict__ qsix, uint32_t *__restrict__ qmap) {
	const uint64_t o0 = off_raw[0];
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j <= n; j += gridDim.x * blockDim.x) {
		qoff[ebase + j] = off_raw[j] - o0 + pos_base;
		if (j < n) { if (qsix) qsix[ebase + j] = j; qmap[ebase + j] = q_base + j; }
	}
}

// symbols uploaded two per byte -> one per byte (the kernels that walk single symbols read bytes): n_sym symbols starting at
// nibble src0 of `packed` go to dst[0 .. n_sym).  One thread per aligned output dword: its four nibbles sit in two aligned
// input dwords (funnel shift), whatever the alignment of either side.
__global__ void k_unpack4(const uint8_t *__restrict__ packed, uint64_t src0, uint64_t n_sym, uint8_t *__restrict__ dst) {
	const uint32_t head = (uint32_t)((uintptr_t)dst & 3u);
	uint8_t *base = dst - head;
	const uint32_t phead = (uint32_t)((uintptr_t)packed & 3u);
	const uint32_t *pw = (const uint32_t *)(packed - phead);
	const uint64_t n_words = (head + n_sym + 3) >> 2;
	for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < n_words; t += (uint64_t)gridDim.x * blockDim.x) {
		const long long i0 = (long long)(4 * t) - (long long)head;            // symbol index of byte 0 of this output dword
		const long long sn = (long long)src0 + i0 + 8ll * phead;              // its nibble, counted from pw
		uint32_t nib = 0;
		if (sn >= 0) {
			const uint64_t w = (uint64_t)sn >> 3;
			const unsigned long long two = (unsigned long long)pw[w] | (unsigned long long)pw[w + 1] << 32;
			nib = (uint32_t)(two >> (4u * (uint32_t)(sn & 7))) & 0xFFFFu;
		} else {      // only the first dword of a span can start before the data
			const unsigned long long two = (unsigned long long)pw[0] | (unsigned long long)pw[1] << 32;
			nib = (uint32_t)(two << (4u * (uint32_t)(-sn))) & 0xFFFFu;
		}
		const uint32_t v = (nib & 15u) | (nib & 0xF0u) << 4 | (nib & 0xF00u) << 8 | (nib & 0xF000u) << 12;
		if (i0 >= 0 && (uint64_t)i0 + 4 <= n_sym) ((uint32_t *)base)[t] = v;
		else for (uint32_t b2 = 0; b2 < 4; ++b2) { const long long idx = i0 + b2; if (idx >= 0 && (uint64_t)idx < n_sym) base[4 * t + b2] = (uint8_t)(v >> (8 * b2)); }
	}
}

// the same for symbols uploaded FOUR per byte (spans of A/C/G/T only: code - 1 in two bits): n_sym symbols starting at symbol
// src0 of `packed` go to dst[0 .. n_sym) as codes 1..4
__global__ void k_unpack2(const uint8_t *__restrict__ packed, uint64_t src0, uint64_t n_sym, uint8_t *__restrict__ dst) {
	const uint32_t head = (uint32_t)((uintptr_t)dst & 3u);
	uint8_t *base = dst - head;
	const uint32_t phead = (uint32_t)((uintptr_t)packed & 3u);
	const uint32_t *pw = (const uint32_t *)(packed - phead);
	const uint64_t n_words = (head + n_sym + 3) >> 2;
	for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < n_words; t += (uint64_t)gridDim.x * blockDim.x) {
		const long long i0 = (long long)(4 * t) - (long long)head;            // symbol index of byte 0 of this output dword
		const long long sn = (long long)src0 + i0 + 16ll * phead;             // its position (in symbols) counted from pw
		uint32_t two = 0;
		if (sn >= 0) {
			const uint64_t w = (uint64_t)sn >> 4;
			const unsigned long long x = (unsigned long long)pw[w] | (unsigned long long)pw[w + 1] << 32;
			two = (uint32_t)(x >> (2u * (uint32_t)(sn & 15))) & 0xFFu;
		} else {      // only the first dword of a span can start before the data
			const unsigned long long x = (unsigned long long)pw[0] | (unsigned long long)pw[1] << 32;
			two = (uint32_t)(x << (2u * (uint32_t)(-sn))) & 0xFFu;
		}
		const uint32_t v = ((two & 3u) | (two & 0xCu) << 6 | (two & 0x30u) << 12 | (two & 0xC0u) << 18) + 0x01010101u;
		if (i0 >= 0 && (uint64_t)i0 + 4 <= n_sym) ((uint32_t *)base)[t] = v;
		else for (uint32_t b2 = 0; b2 < 4; ++b2) { const long long idx = i0 + b2; if (idx >= 0 && (uint64_t)idx < n_sym) base[4 * t + b2] = (uint8_t)(v >> (8 * b2)); }
	}
}

// seed plan of one entry: same choice as make_seed_plan (bhip_api.hip); vb = bit p set iff the word at p holds only A/C/G/T
// (qp = the entry's 4-bit packed symbols: with non-overlapping words -- stride K -- the words that hold one ambiguous symbol vote through their expansions,
// bhip_internal.h; the same walk as the host's)
__device__ uint32_t bhip_seed_plan(uint32_t len, uint32_t E, uint32_t K, int stride_opt, bool clean, const uint32_t *vb, const uint32_t *qp, const BhipAlt &A) {
	if (len < K) return 1u;
	const uint32_t npos = len - K + 1;
	uint32_t xk = 0, usedk = 0;
	auto sym = [&](uint32_t i) -> uint32_t { return (qp[i >> 3] >> (4u * (i & 7u))) & 15u; };
	auto need_of = [&](uint32_t st) -> int {
		uint32_t W = 0;
		if (clean) W = (len - K) / st + 1;
		else if (st == K) { uint32_t ws; bhip_expand_walk(sym, K, (len - K) / K + 1, A, ws, xk, usedk); W = ws + xk; }
		else if (len <= 1024u) for (uint32_t p = 0; p < npos; p += st) W += (vb[p >> 5] >> (p & 31u)) & 1u;
		else for (uint32_t p = 0; p < npos; p += st) {      // (beyond the bitmap's 1 024 positions: the K symbols of every sampled word)
			uint32_t ok = 1u;
			for (uint32_t k = 0; k < K; ++k) ok &= (sym(p + k) - 1u) < 4u ? 1u : 0u;
			W += ok;
		}
		return (int)W - (int)(E * ((K + st - 1) / st));
	};
	const uint32_t smin = (len - K) / 254 + 1, smax = K > smin ? K : smin;
	uint32_t best_s = 0; int best_n = 0;
	if (stride_opt > 0) { best_s = (uint32_t)stride_opt > smin ? (uint32_t)stride_opt : smin; best_n = need_of(best_s); }
	else {
		for (uint32_t st = smax; st >= smin; --st) { const int n = need_of(st); if (n >= 3) { best_s = st; best_n = n; break; } if (st == smin) break; }
		if (!best_s) for (uint32_t st = smin; st <= smax; ++st) { const int n = need_of(st); if (n > best_n) { best_n = n; best_s = st; } }
		if (!best_s) { best_s = smin; best_n = need_of(smin); }
	}
	if (best_n < 1) best_n = 0;
	if (best_n > 0xFFFF) best_n = 0xFFFF;
	const bool ex = !clean && best_s == K && best_n > 0 && xk > 0;
	return (best_s & 255u) | ((uint32_t)best_n << 8) | (ex ? xk << 24 | usedk << 28 : 0u);
}

__global__ __launch_bounds__(256) void k_route(
		const uint64_t *__restrict__ qoff, const uint32_t *__restrict__ qpack, uint32_t qw, const uint16_t *__restrict__ qemac,
		const uint32_t *__restrict__ qsix, const uint8_t *__restrict__ qflags, uint32_t n_q, uint32_t n_shared, uint32_t n_lanes,
		int has_acx, int K, int stride_opt,
		uint32_t *__restrict__ plan, uint8_t *__restrict__ key_out, uint32_t *__restrict__ idx_out, BhipStageInfo *__restrict__ info, BhipAlt alt) {
	__shared__ uint32_t s_count[256], s_maxE[BHIP_ROUTE_KEYS / 2], s_maxw[BHIP_ROUTE_KEYS / 2], s_seed[BHIP_ROUTE_KEYS / 2], s_maxlen[16], s_nent[16], s_misc[4];
	const uint32_t tid = threadIdx.x;
	s_count[tid] = 0;
	if (tid < BHIP_ROUTE_KEYS / 2) { s_maxE[tid] = 0; s_maxw[tid] = 0; s_seed[tid] = 0; }
	if (tid < 16) { s_maxlen[tid] = 0; s_nent[tid] = 0; }
	if (tid < 4) s_misc[tid] = 0;
	__syncthreads();
	for (uint32_t i = blockIdx.x * 256 + tid; i < n_q; i += gridDim.x * 256) {
		const uint64_t b = qoff[i];
		const uint64_t len64 = qoff[i + 1] - b;
		uint32_t key = BHIP_ROUTE_SKIP, pl = 1u;
		if (len64 > BHIP_MAX_QLEN || len64 > 8ull * qw) { if (!atomicExch(&info->err, 1u)) { info->err_i = i; info->err_len = (uint32_t)(len64 > 0xFFFFFFFFull ? 0xFFFFFFFFull : len64); } }
		else if (qsix && qsix[i] >= n_shared) { if (!atomicExch(&info->err, 2u)) info->err_i = i; }
		else if (len64) {
			const uint32_t len = (uint32_t)len64, E = qemac[i];
			const uint32_t *qp = qpack + (uint64_t)i * qw;
			// one pass over the symbols: anything outside A/C/G/T? any code 0?
			uint32_t n_zero = 0, n_other = 0;
			for (uint32_t j = 0; j < (len + 7) >> 3; ++j) {      // eight symbols per step: bit 3 of a nibble of `zero` / `other` flags that symbol
				const uint32_t d = qp[j], nsym = len - 8 * j < 8 ? len - 8 * j : 8u;
				const uint32_t valid = nsym >= 8 ? 0x88888888u : (0x88888888u & ((1u << (4 * nsym)) - 1u));
				const uint32_t zero = ~(((d & 0x77777777u) + 0x77777777u) | d);                          // code 0
				const uint32_t ge5 = d | ((d << 1) & ((d << 2) | (d << 3)));                             // code >= 5: bit 3, or bit 2 with bit 1 or bit 0
				n_zero += __popc(zero & valid);
				n_other += __popc((zero | ge5) & valid);
			}
			if (n_zero) atomicOr(&s_misc[0], 1u);
			const uint32_t six = qsix ? qsix[i] : i;
			const uint32_t l = (uint32_t)(((unsigned long long)six * n_lanes) / n_shared);
			const uint32_t cls = len <= 64 ? 0u : len <= 128 ? 1u : len <= 192 ? 2u : len <= 256 ? 3u : len <= 320 ? 4u : len <= 512 ? 5u : len <= 1024 ? 6u : 7u;
			uint32_t ex = qflags ? (qflags[i] == BHIP_Q_EXHAUSTIVE) : !has_acx;
			if (!has_acx) ex = 1;
			if (!ex) {
				uint32_t vb[32];
				const bool clean = n_other == 0;
				if (!clean && len >= (uint32_t)K && len <= 1024u) {
					for (uint32_t w = 0; w < 32; ++w) vb[w] = 0;
					uint32_t run = 0;
					for (uint32_t p = 0; p < len; ++p) {
						const uint32_t c = (qp[p >> 3] >> (4 * (p & 7u))) & 15u;
						run = (c - 1u) < 4u ? run + 1 : 0;
						if (p + 1 >= (uint32_t)K && run >= (uint32_t)K) { const uint32_t w0 = p + 1 - K; vb[w0 >> 5] |= 1u << (w0 & 31u); }
					}
				}
				pl = bhip_seed_plan(len, E, (uint32_t)K, stride_opt, clean, vb, qp, alt);
				if (BHIP_PLAN_NEED(pl) == 0) ex = 1;        // no word is guaranteed to survive: exhaustive (burst.c:3130-3131 does the same for "bad" queries)
			}
			const uint32_t lc = l * BHIP_N_CLASSES + cls;
			key = lc * 2 + ex;
			atomicAdd(&s_count[key], 1u);
			atomicMax(&s_maxE[lc], E);
			if (!ex && len >= (uint32_t)K) {
				const uint32_t nwd = (len - K) / (pl & 255u) + 1 + BHIP_PLAN_USED(pl);      // (+ the slots of expanded words)
				atomicMax(&s_maxw[lc], nwd);
				atomicAdd(&s_seed[lc], nwd);
			}
			atomicMax(&s_maxlen[l], len);
			atomicAdd(&s_nent[l], 1u);
			atomicMax(&s_misc[1], E);
		}
		plan[i] = pl;
		key_out[i] = (uint8_t)key;
		idx_out[i] = i;
	}
	__syncthreads();
	if (s_count[tid]) atomicAdd(&info->count[tid], s_count[tid]);
	if (tid < BHIP_ROUTE_KEYS / 2) {
		if (s_maxE[tid]) atomicMax(&info->maxE[tid], s_maxE[tid]);
		if (s_maxw[tid]) atomicMax(&info->maxwords[tid], s_maxw[tid]);
		if (s_seed[tid]) atomicAdd(&info->seed_words[tid], (unsigned long long)s_seed[tid]);
	}
	if (tid < 16) { if (s_maxlen[tid]) { atomicMax(&info->maxlen_lane[tid], s_maxlen[tid]); atomicMax(&info->maxlen_all, s_maxlen[tid]); } if (s_nent[tid]) atomicAdd(&info->n_entries_lane[tid], s_nent[tid]); }
	if (tid == 0) { if (s_misc[0]) atomicOr(&info->junk, 1u); if (s_misc[1]) atomicMax(&info->maxE_all, s_misc[1]); }
}

// Dynamic LDS layout (dwords, all [row][64 threads]): band[band_rows + 1] | qbuf[qw] | rbuf[rw].
// A hit uses the LDS copies when m <= 8*qw and its reference segment fits rw dwords, else it reads global memory per row.
template <bool WIDE>
__global__ __launch_bounds__(64) void k_rescore(
		const BhipRawHit *__restrict__ raw, const uint32_t *__restrict__ n_raw_dev, uint32_t raw_cap,
		const uint32_t *__restrict__ wide_in, const uint32_t *__restrict__ n_wide_in,
		const uint32_t *__restrict__ best, int all_hits,
		const uint8_t *__restrict__ qcodes, const uint64_t *__restrict__ qoff,
		const uint32_t *__restrict__ qsix, const uint8_t *__restrict__ qrc,
		const uint8_t *__restrict__ refb, const uint64_t *__restrict__ ref_off, const uint32_t *__restrict__ clump_len,
		const uint8_t *__restrict__ lut,
		BhipHit *__restrict__ out, uint32_t *__restrict__ n_out, uint32_t out_cap,
		uint32_t *__restrict__ wide_out, uint32_t *__restrict__ n_wide_out,
		uint32_t *__restrict__ g_scratch, unsigned long long *__restrict__ scratch_used, unsigned long long scratch_cap,
		uint32_t *__restrict__ err_flags,
		const uint32_t *__restrict__ qpack, uint32_t band_rows, uint32_t qw, uint32_t rw) {
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
	__shared__ uint32_t s_mm[16];      // match masks: bit r of s_mm[q] = (cost(q, r) == 0), bit 16 + r = (cost(q, r) != 255): a finite cost is 0 or 1
	                                   // (nucleotide table: 255 exactly against the pad code 0; -x: the identity table, a pad costs 1 like any other symbol, burst.c:696-697)
	const uint32_t tid = threadIdx.x;
	uint32_t *s_band = smem + tid;                              // [k * 64]
	uint32_t *s_q = smem + (size_t)(band_rows + 1) * 64 + tid;  // [j * 64]
	uint32_t *s_r = s_q + (size_t)qw * 64;                      // [j * 64]
	if (tid < 16) { uint32_t m = 0; for (int r = 0; r < 16; ++r) m |= (lut[16 * tid + r] == 0 ? 1u : 0u) << r | (lut[16 * tid + r] != 255 ? 1u : 0u) << (16 + r); s_mm[tid] = m; }
	__syncthreads();
	const uint32_t *refw = (const uint32_t *)refb;
	uint32_t n = wide_in ? *n_wide_in : *n_raw_dev;
	if (n > raw_cap) n = raw_cap;
	for (uint32_t ii = blockIdx.x * 64 + tid; ii < n; ii += gridDim.x * 64) {
		const uint32_t i = wide_in ? wide_in[ii] : ii;       // index into raw[]
		const BhipRawHit h = raw[i];
		const uint32_t q = h.q, six = qsix ? qsix[q] : q;
		if (!all_hits && h.ed != best[six]) continue;
		const uint32_t B = h.ed, c = h.refIx >> 4, z = h.refIx & 15, L = clump_len[c], nchunks = (L + 31) >> 5;
		const uint64_t qb = qoff[q];
		const int m = (int)(qoff[q + 1] - qb);
		const int e2 = (int)(h.e_last < L ? h.e_last : L);
		if (B == 0) {
			// exact match: the only final cells with score 0 are gap-free, so gapQ = gapR = 0, the end is the LAST column
			// with score 0 (burst.c:862-879) and the identity is 1 - 0/len
			const uint32_t pos = atomicAdd(n_out, 1u);
			if (pos < out_cap) {
				BhipHit o; o.q = q; o.refIx = h.refIx; o.finalPos = (uint32_t)e2; o.score = 1.0f - 0.0f / (float)m;
				o.ed = 0; o.gapR = 0; o.gapQ = 0; o.rc = qrc ? qrc[q] : 0;
				out[pos] = o;
				bhip_hit_rank(n_out, pos, q);
			}
			continue;
		}
		const int e1 = (int)h.e_first;
		const int dlo = e1 - m - (int)B, dhi = e2 - m + (int)B, Wd = dhi - dlo + 1;
		uint32_t *band; uint32_t stride;
		if (!WIDE) {
			if (Wd > (int)band_rows) {   // rare (repeats inside one shear): defer to the global-scratch variant
				const uint32_t pos = atomicAdd(n_wide_out, 1u);
				wide_out[pos] = i;
				continue;
			}
			band = s_band; stride = 64;
		} else {
			const unsigned long long off = atomicAdd(scratch_used, (unsigned long long)(Wd + 1));
			if (off + Wd + 1 > scratch_cap) { atomicOr(err_flags, 2u); continue; }
			band = g_scratch + off; stride = 1;
		}
		const uint64_t cbase = ref_off[c];
		if (B > 254u) { atomicOr(err_flags, 1u); continue; }     // beyond the reference's 8-bit DP
		// One cell = one word ordered exactly like the reference's tie-breaks (burst.c:771-798): score in the top bits, then
		// 255 - gapQ (larger gapQ wins a score tie), then the predecessor priority diag < up < left, then gapR as payload; the
		// three-way choice is a single unsigned minimum.  Cells above the budget collapse to one INVALID value: they can
		// never be the predecessor of a cell within the budget, so their gap counts are never observed.
		constexpr uint32_t SS = 18, GS = 10;
		const uint32_t INVALID = (512u << SS) | (255u << GS);
		const uint32_t STEP_U = (1u << SS) + 1u + (1u << 8), STEP_L = (1u << SS) - (1u << GS) + (2u << 8);
		// stage the query and the reference segment [dlo-1, dlo+m+Wd] in LDS when they fit
		const int j8_0 = (dlo - 1) >> 3, j8_1 = (dlo + m + Wd) >> 3;
		const bool pre = qpack && (uint32_t)m <= 8 * qw && (uint32_t)(j8_1 - j8_0 + 1) <= rw;
		if (pre) {
			const uint32_t *qp = qpack + (uint64_t)q * qw;
			for (uint32_t j = 0; j < (uint32_t)(m + 7) >> 3; ++j) s_q[j * 64] = qp[j];
			for (int j = j8_0; j <= j8_1; ++j) s_r[(uint32_t)(j - j8_0) * 64] = ref_dword_lane(refw, cbase, z, j, nchunks);
		}
		auto rdw = [&](int j8) -> uint32_t { return pre ? s_r[(uint32_t)(j8 - j8_0) * 64] : ref_dword_lane(refw, cbase, z, j8, nchunks); };
		// row 0: D = 0 wherever the column exists (burst.c:4052), else invalid
		for (int k = 0; k <= Wd; ++k) {
			const int x = dlo + k;
			band[(uint32_t)k * stride] = (k < Wd && x >= 0 && x <= (int)L) ? (255u << GS) : INVALID;
		}
		uint32_t qdw = 0;
		for (int y = 1; y <= m; ++y) {
			uint32_t qc;
			if (pre) { if (((y - 1) & 7) == 0) qdw = s_q[(uint32_t)((y - 1) >> 3) * 64]; qc = (qdw >> (4 * ((y - 1) & 7))) & 15u; }
			else qc = qcodes[qb + y - 1] & 15u;
			const uint32_t mrow = s_mm[qc], m1 = mrow >> 16;
			const uint32_t col0 = (uint32_t)y <= B ? (((uint32_t)y << SS) | (255u << GS) | (uint32_t)y) : INVALID;   // D=y, H=0, V=y (burst.c:747-750)
			const int x0 = y + dlo;
			uint32_t left = (x0 - 1 == 0) ? col0 : INVALID;
			// reference symbols of this row: positions x0-1+k (0-based), fetched 8 at a time
			int pos = x0 - 1;
			uint32_t dw = rdw(pos >> 3);
			uint32_t prev_sym = (y == 1) ? ((rdw((pos - 1) >> 3) >> (4 * ((pos - 1) & 7))) & 15u) : 0u;
			uint32_t dg = band[0];
			for (int k = 0; k < Wd; ++k, ++pos) {
				const int x = x0 + k;
				if ((pos & 7) == 0 && k) dw = rdw(pos >> 3);
				const uint32_t r = (dw >> (4 * (pos & 7))) & 15u;
				const uint32_t up = band[(uint32_t)(k + 1) * stride];
				uint32_t cell;
				if (x < 1) cell = (x == 0) ? col0 : INVALID;
				else if (x > (int)L) cell = INVALID;
				else {
					const uint32_t cst = ((mrow >> r) & 1u) ? 0u : (((m1 >> r) & 1u) ? 1u : 255u);
					if (y == 1) {   // burst.c:722-739
						uint32_t hh = 0;
						if (cst == 1 && x >= 2) hh = (mrow >> prev_sym) & 1u;      // left cell of row 1 is 0 iff its symbol matches
						cell = cst == 255u ? INVALID : ((cst << SS) | ((255u - hh) << GS));
					} else {
						const uint32_t cD = dg + (cst << SS), cU = up + STEP_U, cL = left + STEP_L;
						uint32_t cm = cD < cU ? cD : cU;
						cm = cm < cL ? cm : cL;
						cm &= ~0x300u;
						cell = (cm >> SS) > B ? INVALID : cm;                                // burst.c:802-803
					}
				}
				prev_sym = r;
				band[(uint32_t)k * stride] = cell;
				left = cell;
				dg = up;
			}
		}
		// final selection over the last row (burst.c:824-842) and end position (862-879)
		uint32_t bkey = 0xFFFFFFFFu, bv = 0, fin = 0xFFFFFFFFu;
		for (int k = 0; k < Wd; ++k) {
			const int x = m + dlo + k;
			if (x < 1 || x > (int)L) continue;
			const uint32_t cell = band[(uint32_t)k * stride], key = cell >> GS;
			if (key < bkey) { bkey = key; bv = cell & 255u; }
		}
		for (int k = 0; k < Wd; ++k) {
			const int x = m + dlo + k;
			if (x < 1 || x > (int)L) continue;
			if ((band[(uint32_t)k * stride] >> GS) == bkey) fin = (uint32_t)x;
		}
		const uint32_t bs = bkey >> 8, bh = 255u - (bkey & 255u);
		if (bs != B) { atomicOr(err_flags, 1u); continue; }   // the reference would abort here (burst.c:812-816)
		const uint32_t pos = atomicAdd(n_out, 1u);
		if (pos < out_cap) {
			BhipHit o;
			o.q = q; o.refIx = h.refIx; o.finalPos = fin;
			o.score = 1.0f - (float)bs / ((float)m + (float)bh);                                  // burst.c:844-847
			o.ed = (uint8_t)B; o.gapR = (uint8_t)bv; o.gapQ = (uint8_t)bh; o.rc = qrc ? qrc[q] : 0;
			out[pos] = o;
			bhip_hit_rank(n_out, pos, q);
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Re-scoring, fast path.  k_rescore_classify filters the raw hits (best-of-slot unless all hits are wanted), emits
// the exact matches directly and sorts the rest by band width into index lists; k_rescore_reg<WB> then runs the same
// banded 3-plane recurrence as k_rescore with the band (<= WB diagonals) in REGISTERS, fully unrolled, all 64 lanes of
// a wave busy with hits of similar width.  Bands beyond the widest register variant go to k_rescore (LDS band) and
// beyond that to its global-scratch variant.
// ------------------------------------------------------------------------------------------------
// index lists: 0..8 register variants (4, 6, 8, 12, 16, 24, 32, 40, 48 diagonals), 9 LDS band; 10 = global scratch (`wide`), 11 = exact match (emitted)
__global__ __launch_bounds__(256) void k_rescore_classify(
		const BhipRawHit *__restrict__ raw, const uint32_t *__restrict__ n_raw_dev, uint32_t raw_cap,
		const uint32_t *__restrict__ best, int all_hits, const uint64_t *__restrict__ qoff,
		const uint32_t *__restrict__ qsix, const uint8_t *__restrict__ qrc, const uint32_t *__restrict__ clump_len,
		BhipHit *__restrict__ out, uint32_t *__restrict__ n_out, uint32_t out_cap,
		uint32_t *__restrict__ lists, uint32_t *__restrict__ counts, uint32_t *__restrict__ wide, uint32_t *__restrict__ n_wide,
		uint32_t band_rows, int use_reg) {
	// one global reservation per bucket and 1024-hit chunk (ranks inside the chunk come from LDS counters)
	__shared__ uint32_t s_cnt[12], s_base[12];
	uint32_t n = *n_raw_dev;
	if (n > raw_cap) n = raw_cap;
	const uint32_t tid = threadIdx.x;
	for (uint32_t chunk = blockIdx.x * 1024u; chunk < n; chunk += gridDim.x * 1024u) {
		if (tid < 12) s_cnt[tid] = 0;
		__syncthreads();
		int bucket[4]; uint32_t rank[4], e2v[4], mv[4], qv[4], rv[4];
		#pragma unroll
		for (int t = 0; t < 4; ++t) {
			const uint32_t i = chunk + (uint32_t)t * 256u + tid;
			bucket[t] = -1; rank[t] = 0; e2v[t] = 0; mv[t] = 0; qv[t] = 0; rv[t] = 0;
			if (i < n) {
				const BhipRawHit h = raw[i];
				if (all_hits || h.ed == best[h.six]) {          // (slot, clump length and query length travel in the record)
					const uint32_t L = h.L;
					const uint32_t e2 = h.e_last < L ? h.e_last : L;
					qv[t] = h.q; rv[t] = h.refIx; e2v[t] = e2;
					if (h.ed == 0) { bucket[t] = 11; mv[t] = h.m; }     // exact match
					else {
						const uint32_t Wd = e2 - h.e_first + 2 * h.ed + 1;
						int bk = !use_reg || h.ed > 254u ? 9 : Wd <= 4 ? 0 : Wd <= 6 ? 1 : Wd <= 8 ? 2 : Wd <= 12 ? 3 : Wd <= 16 ? 4 : Wd <= 24 ? 5 : Wd <= 32 ? 6 : Wd <= 40 ? 7 : Wd <= 48 ? 8 : 9;
						if (Wd > band_rows && bk == 9) bk = 10;
						bucket[t] = bk;
					}
					rank[t] = atomicAdd(&s_cnt[bucket[t]], 1u);
				}
			}
		}
		__syncthreads();
		if (tid < 12 && s_cnt[tid]) s_base[tid] = atomicAdd(tid == 11 ? n_out : tid == 10 ? n_wide : &counts[tid], s_cnt[tid]);
		__syncthreads();
		#pragma unroll
		for (int t = 0; t < 4; ++t) {
			const uint32_t i = chunk + (uint32_t)t * 256u + tid;
			const int bk = bucket[t];
			if (bk < 0) continue;
			const uint32_t pos = s_base[bk] + rank[t];
			if (bk == 11) {   // gap-free, end = LAST column with score 0 (burst.c:862-879), identity 1 - 0/len
				if (pos < out_cap) {
					BhipHit o; o.q = qv[t]; o.refIx = rv[t]; o.finalPos = e2v[t]; o.score = 1.0f - 0.0f / (float)mv[t];
					o.ed = 0; o.gapR = 0; o.gapQ = 0; o.rc = qrc ? qrc[qv[t]] : 0;
					out[pos] = o;
					bhip_hit_rank(n_out, pos, qv[t]);
				}
			} else if (bk == 10) wide[pos] = i;
			else lists[(size_t)bk * raw_cap + pos] = i;
		}
		__syncthreads();
	}
}

template <int WB>
__device__ __forceinline__ void rescore_reg_one(
		const BhipRawHit *__restrict__ hp, bool live, uint32_t *s_mm, uint32_t fastq, uint32_t lane,
		const uint64_t *__restrict__ qoff, const uint8_t *__restrict__ qrc, const uint32_t *__restrict__ qpack, uint32_t qw,
		const uint32_t *__restrict__ refw, const uint64_t *__restrict__ ref_off, const uint32_t *__restrict__ clump_len,
		BhipHit *__restrict__ out, uint32_t *__restrict__ n_out, uint32_t out_cap, uint32_t *__restrict__ err_flags) {
	constexpr int NW = (WB + 7) / 8 + 1;
	constexpr uint32_t SS = 18, GS = 10;
	const uint32_t INVALID = (512u << SS) | (255u << GS), Z0 = 255u << GS;
	const uint32_t STEP_U = (1u << SS) + 1u + (1u << 8), STEP_L = (1u << SS) - (1u << GS) + (2u << 8);
	uint32_t bkey = 0xFFFFFFFFu, bv = 0, fin = 0xFFFFFFFFu, m = 0, B = 0, hq = 0, hrefIx = 0;
	if (live) {
		const uint32_t q = hp->q;
		hq = q; hrefIx = hp->refIx;
		B = hp->ed;
		const uint32_t h_first = hp->e_first, h_last = hp->e_last;
		const uint32_t L = hp->L, nchunks = (L + 31) >> 5;      // clump length, query length and the lane's address come with the record
		m = hp->m;
		const int e2 = (int)(h_last < L ? h_last : L), e1 = (int)h_first;
		const int dlo = e1 - (int)m - (int)B, Wd = e2 - e1 + 2 * (int)B + 1;
		const uint64_t rbase = hp->rbase;
		const uint32_t LIM = (B + 1) << SS;
		uint32_t bd[WB + 1];
		#pragma unroll
		for (int k = 0; k < WB; ++k) { const int x = dlo + k; bd[k] = (k < Wd && x >= 0 && x <= (int)L) ? Z0 : INVALID; }   // row 0 (burst.c:4052)
		bd[WB] = INVALID;
		int p = dlo, j8 = dlo >> 3;               // p = 0-based reference position under cell k = 0 of the current row
		uint32_t d[NW];
		#pragma unroll
		for (int i = 0; i < NW; ++i) d[i] = ref_dword_at(refw, rbase, j8 + i, nchunks);
		auto all_acgt = [&]() { uint32_t bad = 0;
			#pragma unroll
			for (int i = 0; i < NW; ++i) bad |= (d[i] - 0x11111111u) & 0xCCCCCCCCu;
			return bad == 0; };
		bool dclean = all_acgt();                 // every symbol of the register window is one of A, C, G, T (codes 1..4)
		// the symbols ahead of the window come 32 at a time (one 16-byte load per 32 rows instead of a 4-byte load per 8 rows)
		int jn = j8 + NW;                                      // dword index of the next refill
		uint4 ahead = ref_chunk_at(refw, rbase, jn >> 2, nchunks);
		const uint32_t *qp = qpack + (uint64_t)q * qw;
		uint32_t qd = qp[0], q_next = qw > 1 ? qp[1] : 0u;
		uint32_t prev_sym = (ref_dword_at(refw, rbase, (p - 1) >> 3, nchunks) >> (4 * ((p - 1) & 7))) & 15u;
		for (int y = 1; y <= (int)m; ++y) {
			const uint32_t qi = (uint32_t)(y - 1);
			if ((qi & 7u) == 0 && qi) { qd = q_next; q_next = (qi >> 3) + 1 < qw ? qp[(qi >> 3) + 1] : 0u; }
			const uint32_t qc = (qd >> (4 * (qi & 7u))) & 15u;
			const uint32_t col0 = (uint32_t)y <= B ? (((uint32_t)y << SS) | Z0 | (uint32_t)y) : INVALID;   // D=y, H=0, V=y (burst.c:747-750)
			const int x0 = y + dlo;
			const bool fast_row = y > 1 && x0 >= 1 && dclean && ((fastq >> qc) & 1u);
			const uint32_t mrow = fast_row ? 0u : s_mm[qc];      // (the usual row needs no table: no LDS round trip per row)
			const uint32_t m1 = mrow >> 16;                      // (costs that are finite: all but the pad column with the nucleotide table)
			uint32_t dd[NW - 1];
			{
				const uint32_t sh = 4u * ((uint32_t)p & 7u);
				#pragma unroll
				for (int i = 0; i < NW - 1; ++i) dd[i] = __builtin_amdgcn_alignbit(d[i + 1], d[i], sh);
			}
			const int kmax = (Wd - 1) < ((int)L - x0) ? (Wd - 1) : ((int)L - x0);      // cells beyond are outside the band or the matrix
			if (y == 1) {   // burst.c:722-739
				#pragma unroll
				for (int k = 0; k < WB; ++k) {
					const int x = x0 + k;
					const uint32_t r = (dd[k >> 3] >> (4 * (k & 7))) & 15u;
					const uint32_t cst = ((mrow >> r) & 1u) ? 0u : (((m1 >> r) & 1u) ? 1u : 255u);
					uint32_t hh = 0;
					if (cst == 1 && x >= 2) hh = (mrow >> prev_sym) & 1u;          // left cell of row 1 is 0 iff its symbol matches
					uint32_t cell = cst == 255u ? INVALID : ((cst << SS) | ((255u - hh) << GS));
					if (x < 1) cell = x == 0 ? col0 : INVALID;
					if (k > kmax) cell = INVALID;
					prev_sym = r;
					bd[k] = cell;
				}
			} else if (fast_row) {
				// the usual row: query symbol and all reference symbols in reach are A/C/G/T, of which only the equal one costs 0 --
				// the eight costs of a dword come from three integer operations (a nibble of x is zero iff the symbols are equal;
				// bit 3 of (x & 7 + 7) | x is set iff the nibble is not), and no cell of the row lies left of column 1
				uint32_t nz[NW - 1];
				#pragma unroll
				for (int i = 0; i < NW - 1; ++i) { const uint32_t x = dd[i] ^ (qc * 0x11111111u); nz[i] = ((x & 0x77777777u) + 0x77777777u) | x; }
				uint32_t left = (x0 - 1 == 0) ? col0 : INVALID;
				uint32_t dg = bd[0];
				#pragma unroll
				for (int k = 0; k < WB; ++k) {
					const uint32_t up = bd[k + 1];
					const uint32_t cD = dg + (((nz[k >> 3] >> (4 * (k & 7) + 3)) & 1u) << SS), cU = up + STEP_U, cL = left + STEP_L;
					uint32_t cm = cD < cU ? cD : cU;
					cm = cm < cL ? cm : cL;
					cm &= ~0x300u;
					uint32_t cell = cm >= LIM ? INVALID : cm;
					if (k > kmax) cell = INVALID;
					bd[k] = cell;
					left = cell;
					dg = up;
				}
			} else {
				uint32_t left = (x0 - 1 == 0) ? col0 : INVALID;
				uint32_t dg = bd[0];
				#pragma unroll
				for (int k = 0; k < WB; ++k) {
					const uint32_t r = (dd[k >> 3] >> (4 * (k & 7))) & 15u;
					const uint32_t up = bd[k + 1];
					const uint32_t cstS = ((mrow >> r) & 1u) ? 0u : (((m1 >> r) & 1u) ? (1u << SS) : (255u << SS));
					const uint32_t cD = dg + cstS, cU = up + STEP_U, cL = left + STEP_L;
					uint32_t cm = cD < cU ? cD : cU;
					cm = cm < cL ? cm : cL;
					cm &= ~0x300u;
					uint32_t cell = cm >= LIM ? INVALID : cm;                          // burst.c:802-803
					if (x0 < 1) { const int x = x0 + k; if (x < 1) cell = x == 0 ? col0 : INVALID; }
					if (k > kmax) cell = INVALID;
					bd[k] = cell;
					left = cell;
					dg = up;
				}
			}
			++p;
			if ((p & 7) == 0) {
				#pragma unroll
				for (int i = 0; i < NW - 1; ++i) d[i] = d[i + 1];
				d[NW - 1] = pick4(ahead, (uint32_t)jn & 3u);
				dclean = all_acgt();
				++j8; ++jn;
				if ((jn & 3) == 0) ahead = ref_chunk_at(refw, rbase, jn >> 2, nchunks);
			}
		}
		// final selection over the last row (burst.c:824-842) and end position (862-879)
		#pragma unroll
		for (int k = 0; k < WB; ++k) {
			const int x = (int)m + dlo + k;
			if (k < Wd && x >= 1 && x <= (int)L) { const uint32_t key = bd[k] >> GS; if (key < bkey) { bkey = key; bv = bd[k] & 255u; } }
		}
		#pragma unroll
		for (int k = 0; k < WB; ++k) {
			const int x = (int)m + dlo + k;
			if (k < Wd && x >= 1 && x <= (int)L && (bd[k] >> GS) == bkey) fin = (uint32_t)x;
		}
	}
	const uint32_t bs = bkey >> 8, bh = 255u - (bkey & 255u);
	const bool ok = live && bs == B;
	if (live && !ok) atomicOr(err_flags, 1u);                     // the reference would abort here (burst.c:812-816)
	const unsigned long long bm = __ballot(ok);
	if (bm) {
		uint32_t base = 0;
		if (lane == (uint32_t)__builtin_ctzll(bm)) base = atomicAdd(n_out, (uint32_t)__popcll(bm));
		base = __shfl(base, __builtin_ctzll(bm));
		if (ok) {
			const uint32_t pos = base + __popcll(bm & ((1ull << lane) - 1ull));
			if (pos < out_cap) {
				BhipHit o;
				o.q = hq; o.refIx = hrefIx; o.finalPos = fin;
				o.score = 1.0f - (float)bs / ((float)m + (float)bh);                                  // burst.c:844-847
				o.ed = (uint8_t)B; o.gapR = (uint8_t)bv; o.gapQ = (uint8_t)bh; o.rc = qrc ? qrc[hq] : 0;
				out[pos] = o;
				bhip_hit_rank(n_out, pos, hq);
			}
		}
	}
}

template <int SET>       // 0: bands of 4 / 6 / 8 diagonals, 3: 12, 1: 16 / 24, 2: 32 / 40 / 48 (separate kernels: the register budget of the wide ones would halve the occupancy of the narrow ones)
__global__ __launch_bounds__(64) void k_rescore_reg(
		const BhipRawHit *__restrict__ raw, const uint32_t *__restrict__ lists, const uint32_t *__restrict__ counts, uint32_t raw_cap,
		const uint64_t *__restrict__ qoff, const uint8_t *__restrict__ qrc, const uint32_t *__restrict__ qpack, uint32_t qw,
		const uint8_t *__restrict__ refb, const uint64_t *__restrict__ ref_off, const uint32_t *__restrict__ clump_len,
		const uint8_t *__restrict__ lut,
		BhipHit *__restrict__ out, uint32_t *__restrict__ n_out, uint32_t out_cap, uint32_t *__restrict__ err_flags) {
	__shared__ uint32_t s_mm[16];      // match masks: bit r of s_mm[q] = (cost(q, r) == 0), bit 16 + r = (cost(q, r) != 255): a finite cost is 0 or 1
	                                   // (nucleotide table: 255 exactly against the pad code 0; -x: the identity table, a pad costs 1 like any other symbol, burst.c:696-697)
	const uint32_t tid = threadIdx.x;
	if (tid < 16) { uint32_t mm = 0; for (int r = 0; r < 16; ++r) mm |= (lut[16 * tid + r] == 0 ? 1u : 0u) << r | (lut[16 * tid + r] != 255 ? 1u : 0u) << (16 + r); s_mm[tid] = mm; }
	__syncthreads();
	// bit q: query symbol q is one of A, C, G, T and among those four matches only itself (the rows whose costs need no table)
	uint32_t fastq = 0;
	for (uint32_t qc = 1; qc <= 4; ++qc) if ((s_mm[qc] & 0x1Eu) == (1u << qc)) fastq |= 1u << qc;
	const uint32_t *refw = (const uint32_t *)refb;
#define BHIP_RS_RUN(b, WB) { \
		uint32_t n = counts[b]; if (n > raw_cap) n = raw_cap; \
		const uint32_t *lst = lists + (size_t)(b) * raw_cap; \
		const uint32_t n_round = (n + 63u) & ~63u; \
		for (uint32_t i = blockIdx.x * 64 + tid; i < n_round; i += gridDim.x * 64) { \
			const bool live = i < n; \
			rescore_reg_one<WB>(raw + (live ? lst[i] : 0u), live, s_mm, fastq, tid, qoff, qrc, qpack, qw, refw, ref_off, clump_len, out, n_out, out_cap, err_flags); \
		} }
	if (SET == 0) { BHIP_RS_RUN(0, 4) BHIP_RS_RUN(1, 6) BHIP_RS_RUN(2, 8) }
	else if (SET == 3) { BHIP_RS_RUN(3, 12) }
	else if (SET == 1) { BHIP_RS_RUN(4, 16) BHIP_RS_RUN(5, 24) }
	else { BHIP_RS_RUN(6, 32) BHIP_RS_RUN(7, 40) BHIP_RS_RUN(8, 48) }
#undef BHIP_RS_RUN
}
template __global__ void k_rescore_reg<0>(const BhipRawHit *, const uint32_t *, const uint32_t *, uint32_t, const uint64_t *, const uint8_t *, const uint32_t *, uint32_t,
	const uint8_t *, const uint64_t *, const uint32_t *, const uint8_t *, BhipHit *, uint32_t *, uint32_t, uint32_t *);
template __global__ void k_rescore_reg<1>(const BhipRawHit *, const uint32_t *, const uint32_t *, uint32_t, const uint64_t *, const uint8_t *, const uint32_t *, uint32_t,
	const uint8_t *, const uint64_t *, const uint32_t *, const uint8_t *, BhipHit *, uint32_t *, uint32_t, uint32_t *);
template __global__ void k_rescore_reg<2>(const BhipRawHit *, const uint32_t *, const uint32_t *, uint32_t, const uint64_t *, const uint8_t *, const uint32_t *, uint32_t,
	const uint8_t *, const uint64_t *, const uint32_t *, const uint8_t *, BhipHit *, uint32_t *, uint32_t, uint32_t *);
template __global__ void k_rescore_reg<3>(const BhipRawHit *, const uint32_t *, const uint32_t *, uint32_t, const uint64_t *, const uint8_t *, const uint32_t *, uint32_t,
	const uint8_t *, const uint64_t *, const uint32_t *, const uint8_t *, BhipHit *, uint32_t *, uint32_t, uint32_t *);

template __global__ void k_rescore<false>(const BhipRawHit *, const uint32_t *, uint32_t, const uint32_t *, const uint32_t *, const uint32_t *, int,
	const uint8_t *, const uint64_t *, const uint32_t *, const uint8_t *, const uint8_t *, const uint64_t *, const uint32_t *, const uint8_t *,
	BhipHit *, uint32_t *, uint32_t, uint32_t *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long, uint32_t *,
	const uint32_t *, uint32_t, uint32_t, uint32_t);
template __global__ void k_rescore<true>(const BhipRawHit *, const uint32_t *, uint32_t, const uint32_t *, const uint32_t *, const uint32_t *, int,
	const uint8_t *, const uint64_t *, const uint32_t *, const uint8_t *, const uint8_t *, const uint64_t *, const uint32_t *, const uint8_t *,
	BhipHit *, uint32_t *, uint32_t, uint32_t *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long, uint32_t *,
	const uint32_t *, uint32_t, uint32_t, uint32_t);

#ifdef PFM_PROF
extern "C" BHIP_API int bhip_debug_prof(unsigned long long *out, int reset) {
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pfm_prof), 64) != hipSuccess) return -1;
	if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pfm_prof), z, 64) != hipSuccess) return -1; }
	return 0;
}
#endif
