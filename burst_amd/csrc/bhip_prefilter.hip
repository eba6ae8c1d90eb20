// burst_amd/csrc/bhip_prefilter.hip -- the product's lane-resolved prefilter: k_seed_ranges (sampled words -> accelerator list ranges) and
// k_prefilter_cq (four queries per wave, their record streams walked by the whole wave).  Replaces word extraction + qsort + postScour20/24 +
// selection (burst.c:4096-4133, 3238-3285).  The fallbacks a batch can reach (plans beyond 16 lists, overflowed queries, the clump-level
// path, the exact-table kernel for FORAGE over dense families) are in bhip_prefilter_alt.hip; the superseded counting-filter kernels
// (k_prefilter_cf, k_prefilter_cw<0 / 1>) in bhip_prefilter_legacy.hip, outside the product library.
#include "bhip_pf_common.h"
#ifdef PFM_PROF
__device__ unsigned long long g_pfm_prof[8];
#endif
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
#ifndef CQ_R
#define CQ_R 6                 // rows of 64 records of a query that stay in registers between the two looks
#endif
#ifndef CQ_RING
#define CQ_RING 64             // survivors of a query waiting for the rounds at the end of the quad (a power of two, >= one row... or half a row with CQ_RING=32: drained more often)
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
	constexpr uint32_t RING = (uint32_t)CQ_RING;                          // survivors of a query waiting for the rounds at the end of the quad
	constexpr uint32_t CQ_STAGE = (uint32_t)CQ_STAGE_N;
	constexpr uint32_t R = (uint32_t)CQ_R;                                // rows of 64 records of a query that stay in registers between the two looks
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
	// ---- Software pipeline over the quads (round 6).  Until then a quad was: set up, issue the record loads of its four queries, WAIT, two
	// looks, survivor rounds, emit, clear -- a third of the kernel's time was that wait (phase timers with the memory pipeline drained at
	// every boundary, gpurun_out/r06g_prof: 34 %).  The record registers are dead once the second look is over, so the NEXT quad is set up and
	// its loads are issued right there (`prep`), and they are in flight during this quad's survivor rounds, emit and table clears.  What a quad
	// needs after its second look (list position, threshold, budget, ...) is copied out of the carried state at the top of its iteration.
	constexpr uint32_t NB = MODE == 0 ? 7u : 15u, KB = MODE == 0 ? 3u : 4u;      // list ends that matter / bits of a list number
	uint32_t rc[4][R];                                                // records of the resident rows of the quad that is looked at next
	bool n_live = false; uint32_t n_li = 0, n_hx = 0, n_hy = 0, n_eend = 0, n_krp[4] = {0u, 0u, 0u, 0u}; unsigned long long n_ab = 0;
	auto row_rec_of = [&](uint32_t q, uint32_t T, const uint32_t (&eb)[NB], uint32_t r, uint32_t &kreg, unsigned long long ab_) -> uint32_t {
		const uint32_t i = r * 64u + lane;
		const uint32_t ic = i < T ? i : T - 1u;                       // beyond the stream: its last record once more (OR is idempotent; the second look tests i < T)
		uint32_t kk = 0;
		#pragma unroll
		for (uint32_t j = 0; j < NB; ++j) kk += eb[j] <= ic ? 1u : 0u;          // lists that end at or before the position = its list
		const uint32_t src = q * 16u + kk;
		const uint32_t a_lo = (uint32_t)__shfl((int)(uint32_t)ab_, (int)src, 64), a_hi = (uint32_t)__shfl((int)(uint32_t)(ab_ >> 32), (int)src, 64);
		kreg = kk;
		return ((bhip_gptr_t)(uintptr_t)(((unsigned long long)a_hi << 32 | a_lo) + 4ull * ic))[0];
	};
	auto list_ends_of = [&](uint32_t q, uint32_t (&eb)[NB], uint32_t eend_) {           // ends of the query's lists but the last: wave-uniform
		#pragma unroll
		for (uint32_t j = 0; j < NB; ++j) eb[j] = (uint32_t)__builtin_amdgcn_readlane((int)eend_, (int)(q * 16u + j));
	};
	// the lists of quad `quad`'s queries (lane gl of group g = list gl of query g; eend = end of the list in its query's flattened stream;
	// ab = biased address: the record at stream position i of the list is at ab + 4 i), the loads of the first R rows of ALL four queries
	// (their gathers are in flight together), and the header / ranges of the quad after it
	auto prep = [&](uint32_t quad) {
		n_live = quad * 4 + g < n_items;
		n_li = sel ? (n_live ? sel[quad * 4 + g] : 0u) : quad * 4 + g;      // list position of this group's query
		n_hx = (uint32_t)h_n; n_hy = (uint32_t)(h_n >> 32);
		const unsigned long long r_c = r_n;
		fetch(quad + gridDim.x, h_n, r_n);                                // one quad ahead
		const uint32_t rx = (uint32_t)r_c, ry = (uint32_t)(r_c >> 32);
		const uint32_t n0 = (n_live && gl < W16) ? ry & 0xFFFFFFu : 0u;
		const unsigned long long beg = (unsigned long long)rx | (unsigned long long)(ry >> 24) << 32;
		n_eend = group_scan(n0);
		n_ab = (unsigned long long)(uintptr_t)ent + 4ull * (beg - (unsigned long long)(n_eend - n0));
		#pragma unroll
		for (uint32_t q = 0; q < 4; ++q) {
			const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)n_eend, (int)(q * 16u + 15u));
			n_krp[q] = 0;
			if (T == 0u) continue;                                        // (wave-uniform)
			my_ent += T;
			uint32_t eb[NB];
			list_ends_of(q, eb, n_eend);
			const uint32_t rows = (T + 63u) >> 6;
			#pragma unroll
			for (uint32_t r = 0; r < R; ++r) if (r < rows) { uint32_t kr; rc[q][r] = row_rec_of(q, T, eb, r, kr, n_ab); n_krp[q] |= kr << (KB * r); }
		}
	};
	if (blockIdx.x < n_quads) prep(blockIdx.x);
	for (uint32_t quad = blockIdx.x; quad < n_quads; quad += gridDim.x) {
		const bool live = n_live;
		const uint32_t li = n_li;
		const uint2 hd = make_uint2(n_hx, n_hy);
		const uint32_t need = hd.x & 0xFFFFu, len = hd.y & 0xFFFu;
		const uint32_t budget = (hd.y >> 12) & 255u, dper = (hd.y >> 20) & 15u ? (hd.y >> 20) & 15u : 1u;
		const uint32_t thr = need ? need : 1u;                            // (group-uniform)
		const uint32_t eend = n_eend;
		const unsigned long long ab = n_ab;
		const uint32_t krp[4] = {n_krp[0], n_krp[1], n_krp[2], n_krp[3]};
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
		// ---- the record streams, 64 stream positions per row: the first R rows of the four queries are in `rc` (issued by prep one quad ago)
		uint32_t Tq[4];
		#pragma unroll
		for (uint32_t q = 0; q < 4; ++q) Tq[q] = (uint32_t)__builtin_amdgcn_readlane((int)eend, (int)(q * 16u + 15u));
		auto row_rec = [&](uint32_t q, uint32_t T, const uint32_t (&eb)[NB], uint32_t r, uint32_t &kreg) -> uint32_t { return row_rec_of(q, T, eb, r, kreg, ab); };
		auto list_ends = [&](uint32_t q, uint32_t (&eb)[NB]) { list_ends_of(q, eb, eend); };
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
		if (quad + gridDim.x < n_quads) prep(quad + gridDim.x);      // the record registers are free: the next quad's gathers fly during the rounds, the emit and the clears below
		PFM_T(0);
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


#ifdef PFM_PROF
extern "C" BHIP_API int bhip_debug_prof(unsigned long long *out, int reset) {
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pfm_prof), 64) != hipSuccess) return -1;
	if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pfm_prof), z, 64) != hipSuccess) return -1; }
	return 0;
}
#endif

