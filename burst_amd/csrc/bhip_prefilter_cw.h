// burst_amd/csrc/bhip_prefilter_cw.h -- k_prefilter_cw<MODE, BIG>, one query per wave (round 5's first list-mask kernel).  MODE 2 (plans beyond
// 16 lists per query: BASELINE configs[4]'s 320-symbol reads at 95 %) is instantiated in the product (bhip_prefilter_alt.hip); MODE 0 / 1
// are superseded by k_prefilter_cq and live in the test-only library (bhip_prefilter_legacy.hip).
#ifndef BHIP_PREFILTER_CW_H
#define BHIP_PREFILTER_CW_H
#include "bhip_pf_common.h"
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
#define BHIP_INST_PFCW(M, B) \
	template __global__ void k_prefilter_cw<M, B>(const uint2 *, const uint2 *, uint32_t, uint32_t, const uint32_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t, \
		uint2 *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, unsigned long long *, unsigned long long *, \
		uint2 *, uint32_t *, int, const uint32_t *, const uint32_t *, int);
#endif
