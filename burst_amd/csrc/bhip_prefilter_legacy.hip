// burst_amd/csrc/bhip_prefilter_legacy.hip -- the counting-filter kernels k_prefilter_cq superseded: k_prefilter_cf<CB, RB> (rounds 2-4: four
// queries per wave, 16 lanes each, 16-bit counters + exact lane table) and k_prefilter_cw<0 / 1> (round 5: one query per wave).  NOT part of
// libburst_hip.so: built into libburst_hip_legacy.so, which the TESTS load in front of the product library (tests/conftest.py ->
// burst_amd.capi: RTLD_GLOBAL) so that the options prefilter_cw = 0 / 1 keep running them as independent implementations of the same
// candidate set (test_tuning_options_do_not_change_results, the fuzzer).  The product library reaches them through two weak symbols
// (bhip_internal.h: BhipPfLaunch) and refuses those options when the library is not loaded.
#undef PFM_PROF
#include "bhip_pf_common.h"
#include "bhip_prefilter_cw.h"
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

BHIP_INST_PFCW(0, 0) BHIP_INST_PFCW(1, 0) BHIP_INST_PFCW(0, 1) BHIP_INST_PFCW(1, 1)

// ---- the two entry points the product library looks for (weak there, defined here) ----
template <typename F> static int pf_attrs(F fn, size_t *lds, int *regs) {
	hipFuncAttributes fa;
	if (hipFuncGetAttributes(&fa, (const void *)fn) != hipSuccess) { (void)hipGetLastError(); return -1; }
	*lds = fa.sharedSizeBytes; *regs = fa.numRegs;
	return 0;
}
extern "C" __attribute__((visibility("default"))) int bhip_legacy_pf_attrs(int kind, int htb, int rb, int cw_mode, size_t *lds, int *regs) {
	if (kind == 0) {
		if (htb == 9) return rb == 2 ? pf_attrs(k_prefilter_cf<9, 2>, lds, regs) : rb == 3 ? pf_attrs(k_prefilter_cf<9, 3>, lds, regs) : pf_attrs(k_prefilter_cf<9, 4>, lds, regs);
		if (htb == 10) return rb == 2 ? pf_attrs(k_prefilter_cf<10, 2>, lds, regs) : pf_attrs(k_prefilter_cf<10, 4>, lds, regs);
		return rb == 2 ? pf_attrs(k_prefilter_cf<11, 2>, lds, regs) : pf_attrs(k_prefilter_cf<11, 4>, lds, regs);
	}
	return cw_mode == 0 ? pf_attrs(k_prefilter_cw<0, 0>, lds, regs) : pf_attrs(k_prefilter_cw<1, 0>, lds, regs);
}
extern "C" __attribute__((visibility("default"))) int bhip_legacy_pf_launch(const BhipPfLaunch *a) {
	hipStream_t st = (hipStream_t)a->stream;
#define PF_ARGS a->ranges, a->hdr, a->W16, a->n_list, a->ent, a->bad, a->n_bad, a->clump_len, a->tot_refs, a->tasks, a->n_tasks, a->task_cap, a->ent_read, a->fb, a->n_fb, \
	a->unit_sum, a->col_sum, a->qlen_sum, a->surv_sum, a->tasks2, a->n_tasks2, a->prune, a->sel, a->n_sel, a->bytes
#define CF(B, R) hipLaunchKernelGGL((k_prefilter_cf<B, R>), dim3(a->grid), dim3(64), 0, st, PF_ARGS)
#define CW(M, G) hipLaunchKernelGGL((k_prefilter_cw<M, G>), dim3(a->grid), dim3(64), 0, st, PF_ARGS)
	if (a->kind == 0) {
		if (a->htb == 9) { if (a->rb == 2) CF(9, 2); else if (a->rb == 3) CF(9, 3); else CF(9, 4); }
		else if (a->htb == 10) { if (a->rb == 2) CF(10, 2); else CF(10, 4); }
		else { if (a->rb == 2) CF(11, 2); else CF(11, 4); }
	} else if (a->cw_mode == 0) { if (a->big) CW(0, 1); else CW(0, 0); }
	else { if (a->big) CW(1, 1); else CW(1, 0); }
#undef CF
#undef CW
#undef PF_ARGS
	return (int)hipGetLastError();
}
