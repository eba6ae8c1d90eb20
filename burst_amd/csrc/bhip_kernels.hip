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
		// The rest of the query (m - P symbols) needs at least m - P - E more columns behind the prefix's end: a prefix that ends beyond
		// column L - (m - P) + E belongs to no alignment within budget, and the chunks that only hold such columns are not swept
		// (round 6: 2 of a 612-column lane's 20 chunks for a 100-symbol read)
		const uint32_t x_max = L + E > m - P ? L + E - (m - P) : 0u;
		const uint32_t t_end = x_max / 32u + 1u < nchunks ? x_max / 32u + 1u : nchunks;
		for (uint32_t t0 = 0; t0 < t_end; t0 += 4) {      // 4 chunks = 64 contiguous bytes of this lane per round of loads
			uint4 chs[4];
			#pragma unroll
			for (uint32_t u = 0; u < 4; ++u) chs[u] = t0 + u < t_end ? rp[t0 + u] : make_uint4(0, 0, 0, 0);
			#pragma unroll
			for (uint32_t u = 0; u < 4; ++u) {
				const uint32_t t = t0 + u;
				if (t >= t_end) break;
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
		my_cols += t_end * 32u < L ? t_end * 32u : L;
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
#ifndef WB_PREFETCH
#define WB_PREFETCH 0          // 1: the next window's record is loaded while the current one is swept (tools/build_variant.sh ... -DWB_PREFETCH=1; measured, §3)
#endif
#if WB_PREFETCH
	BhipWin w_next;
	if (blockIdx.x * 64u + tid < n) w_next = wins[blockIdx.x * 64u + tid];
#endif
	for (uint32_t i = blockIdx.x * 64u + tid; i < n; i += gridDim.x * 64u) {
#if WB_PREFETCH
		const BhipWin w = w_next;
		if (i + gridDim.x * 64u < n) w_next = wins[i + gridDim.x * 64u];
#else
		const BhipWin w = wins[i];
#endif
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

