// burst_amd/csrc/bhip_internal.h -- types shared between the kernels and the C-ABI host code.
#ifndef BHIP_INTERNAL_H
#define BHIP_INTERNAL_H
#include <stdint.h>

// match_mask[a] bit b = 1 iff cost(a, b) == 0 in the 16x16 table (burst.c:1310-1328 after setScore()).
struct BhipMatchMask { uint16_t m[16]; };

// One (query entry, reference lane) with ed <= budget, as left by k_myers for k_rescore.
struct BhipRawHit {
	uint32_t q;        // batch index of the query entry
	uint32_t refIx;    // 16*clump + lane
	uint32_t ed;
	uint32_t e_first;  // first / last 1-based end column whose last-row score equals ed
	uint32_t e_last;   // (may run into trailing pad columns; k_rescore clamps to ClumpLen)
	// what the re-scoring kernels would otherwise look up through four more random sectors per hit (query offsets, shared-slot
	// table, clump lengths, clump offsets): the sweeps have these at hand when they write the record
	uint32_t m;        // query length (k_junk_adjust_raw adds the symbols the search view left out)
	uint32_t L;        // ClumpLen of the lane's clump
	uint32_t six;      // shared slot of the query (index into best[])
	uint64_t rbase;    // uint4 index of the lane's first chunk in the lane-major copy of the references
};

// A reference lane whose prefix filter fired: the first and the last group of 8 columns (columns 8g + 1 .. 8g + 8, 1-based) that
// hold a column with prefix score <= budget.  (32-column flags cost the full-length stage 24 columns of slack per window.)
// The record carries what the full-length stage needs to start (query number, length, budget, where the lane's symbols are),
// so that stage opens with one load instead of a chain of four dependent ones.
#define BHIP_WIN_GMASK 0x3FFFFFFFu  // g_first proper; its two top bits: the band class (0 .. 2: a band of 2 .. 4 words holds the flagged diagonals, 3: none does)
struct BhipWin {
	uint32_t li;       // list position of the query (peq row)
	uint32_t refIx;
	uint32_t g_first, g_last;
	uint32_t q;        // batch index of the query entry
	uint32_t mE;       // query length | budget << 16
	uint32_t nchunks;  // 32-column chunks of the lane
	uint32_t L;        // ClumpLen of the lane's clump
	uint64_t rbase;    // uint4 index of the lane's first chunk in the lane-major copy of the references
	uint32_t six;      // shared slot of the query (index into best[])
	uint32_t pad2;
};
// a band of BW words holds 32 (BW - 1) - 6 diagonals (k_myers_window_band); the flagged ones are 8 (g_last - g_first) + 8 + 2 E
__host__ __device__ inline uint32_t bhip_win_class(uint32_t g_first, uint32_t g_last, uint32_t E) {
	const uint64_t diags = 8ull * (g_last - g_first) + 8u + 2u * E;
	return diags <= 26 ? 0u : diags <= 58 ? 1u : diags <= 90 ? 2u : 3u;
}

// ------------------------------------------------------------------------------------------------
// Seed plan of one query entry: plan = stride | need << 8 | x << 24 | used << 28.
//   stride  sampled words start at 0, s, 2s, ...
//   need    words an alignment within budget is guaranteed to keep = (voting words) - E * ceil(K / s); 0 = none (exhaustive route)
//   x, used only when stride == K (the words do not overlap): x words that hold exactly ONE ambiguous query symbol with 2..4
//           compatible bases vote through their EXPANSIONS (the reference expands ambiguous query words too: storeAmbigWords,
//           burst.c:3232-3236) -- the first alternative in the word's own slot, the others in `used` (<= BHIP_EXPAND_SLOTS) extra
//           word slots behind the positional ones.  An expansion is a vote of its own: a reference holding two expansions of a word
//           counts twice, which only adds candidates.  `need` includes the x words; the kernels that count strictly (words of
//           A/C/G/T only: the clump-level fallbacks) use need - x and take every clump when that is below 1.
// Words with more ambiguity, with a symbol outside the alphabet, or beyond the slot budget do not vote, as before.
// ------------------------------------------------------------------------------------------------
#define BHIP_PLAN_STRIDE(p) ((p) & 255u)
#define BHIP_PLAN_NEED(p)   (((p) >> 8) & 0xFFFFu)
#define BHIP_PLAN_X(p)      (((p) >> 24) & 15u)
#define BHIP_PLAN_USED(p)   ((p) >> 28)
#define BHIP_EXPAND_SLOTS 8u
// compatible bases of a query symbol code: n[c] = how many of A/C/G/T cost 0 against it (0: a symbol that matches nothing, 1: a base),
// base[c] = those bases, two bits each, first alternative in the low bits.  Made from the cost table at bhip_init.
struct BhipAlt { uint8_t n[16]; uint8_t base[16]; };

// class of the word of K symbols starting at p: 0 does not vote, 1 A/C/G/T only, 2 expandable (amb_k = offset of its ambiguous symbol, extra = alternatives - 1)
template <class Sym>
__host__ __device__ inline uint32_t bhip_word_class(Sym sym, uint32_t p, uint32_t K, const BhipAlt &A, uint32_t &amb_k, uint32_t &extra) {
	uint32_t namb = 0, bad = 0;
	amb_k = 0; extra = 0;
	for (uint32_t k = 0; k < K; ++k) {
		const uint32_t n = A.n[sym(p + k) & 15u];
		if (n == 0u) bad = 1u;
		else if (n > 1u) { ++namb; amb_k = k; extra = n - 1u; }
	}
	return (bad || namb > 1u) ? 0u : (namb ? 2u : 1u);
}
// walk over the non-overlapping words 0, K, 2K, ... [0, upto): voting words (strict + expanded within the slot budget), expanded words, slots used.
// The SAME walk decides everywhere (plan, seed lookups) which expandable words fit the budget: position order, first come first served.
template <class Sym>
__host__ __device__ inline void bhip_expand_walk(Sym sym, uint32_t K, uint32_t upto, const BhipAlt &A, uint32_t &w_strict, uint32_t &x, uint32_t &used) {
	w_strict = 0; x = 0; used = 0;
	for (uint32_t j = 0; j < upto; ++j) {
		uint32_t ak, ex;
		const uint32_t c = bhip_word_class(sym, j * K, K, A, ak, ex);
		if (c == 1u) ++w_strict;
		else if (c == 2u && used + ex <= BHIP_EXPAND_SLOTS) { ++x; used += ex; }
	}
}

// Routing of a staged batch, filled on the device by k_route (or by the host pass that handles batches with query symbols
// outside the alphabet): a query entry belongs to the list of its key = ((lane * BHIP_N_CLASSES + length class) * 2 + exhaustive).
#define BHIP_N_CLASSES  8            // length classes: vectors of 2, 4, 6, 8, 10, 16, 32 words, and the queries beyond 1 024 symbols
#define BHIP_MAX_LANES  15           // sub-pipelines of a batch (the keys are bytes and 255 means "in no list")
#define BHIP_ROUTE_KEYS 240          // 15 lanes x 8 classes x {prefilter, exhaustive}
#define BHIP_ROUTE_SKIP 255          // empty entries (and entries a batch error was raised for): in no list
struct BhipStageInfo {
	uint32_t count[256];               // entries per key
	uint32_t maxE[BHIP_ROUTE_KEYS / 2], maxwords[BHIP_ROUTE_KEYS / 2];
	unsigned long long seed_words[BHIP_ROUTE_KEYS / 2];
	uint32_t maxlen_lane[16], n_entries_lane[16];
	uint32_t junk;                     // entries with symbols of code 0 (the host pass takes over)
	uint32_t err, err_i, err_len;      // 1 = query longer than BHIP_MAX_QLEN, 2 = shared slot out of range
	uint32_t maxE_all, maxlen_all;
};

// A launch of one of the SUPERSEDED counting-filter prefilter kernels (k_prefilter_cf<CB, RB>, k_prefilter_cw<0 / 1, BIG>: options
// prefilter_cw = 0 / 1).  They are not part of libburst_hip.so: bhip_prefilter_legacy.hip builds them into libburst_hip_legacy.so, which
// the tests load in front of the product library; the product reaches them through these two WEAK symbols and refuses the options when
// they are not there.  Same arguments as k_prefilter_cq.
struct BhipPfLaunch {
	int kind;                              // 0 = k_prefilter_cf<htb, rb>, 1 = k_prefilter_cw<cw_mode, big>
	int htb, rb, cw_mode, big;
	uint32_t grid; void *stream;
	const uint2 *ranges, *hdr; uint32_t W16, n_list; const uint32_t *ent, *bad; uint32_t n_bad; const uint32_t *clump_len; uint32_t tot_refs;
	uint2 *tasks; uint32_t *n_tasks; uint32_t task_cap; unsigned long long *ent_read; uint32_t *fb, *n_fb;
	unsigned long long *unit_sum, *col_sum, *qlen_sum, *surv_sum; uint2 *tasks2; uint32_t *n_tasks2; int prune; const uint32_t *sel, *n_sel; int bytes;
};
extern "C" int bhip_legacy_pf_attrs(int kind, int htb, int rb, int cw_mode, size_t *lds_bytes, int *regs) __attribute__((weak));      // 0 = ok
extern "C" int bhip_legacy_pf_launch(const BhipPfLaunch *a) __attribute__((weak));                                                   // hipError_t of the launch

#define BHIP_RESCORE_WMAX 48   // band widths up to this many diagonals are handled in LDS

// ------------------------------------------------------------------------------------------------
// Accelerator (.acx, burst.c:3535-3594) as it lives in HBM.
//   Lists: one aligned 4-byte record per list entry, in the file's word order: bits 0-23 = clump id (the 24 bits of the LARGE
//   format, burst.c:3245-3248), bits 24-31 = a LANE-SET CODE: which of the clump's 16 lanes really hold the word (the .acx is
//   clump-granular; the lanes are worked out on the device).  The code names a superset of the true lane set that is exact for
//   one lane (codes 0..15) and for two (16..135, the 120 pairs) -- 99.9 % of the entries of a low-redundancy database --, and the
//   smallest enclosing quad pattern beyond: three or four lanes of one aligned quad (136..155), a union of quads (156..166; 166 =
//   every lane, also what records without lane information carry).  A superset never loses a candidate lane, it only costs a
//   sweep.  Round 3 kept a full 16-bit mask (5 bytes, two loads and a funnel shift per record); 4 bytes are what lets the
//   metric's database (~54 G entries) stay resident on one 288 GB device, and one dword load per record.
//   Offsets: the file's Lens[4^K] (burst.c:3558) becomes a table of 64-byte LINES, one per block of 14
//   words: a 64-bit entry number of the block's first list followed by the fourteen 32-bit inclusive prefix sums of the list
//   lengths inside the block -- the range of a word is ONE 64-byte sector at a random address (a flat offset array costs two,
//   a two-level table three).  A list is shorter than 2^24 entries (one per clump at most, burst.c:3385-3386, clump ids have
//   24 bits), so 14 of them always fit 32 bits, while the whole accelerator may hold far more than 2^32 entries (RefSeq
//   scale: ~5 * 10^10).
//   `rec` may point BEFORE the allocation (test hook BHIP_TEST_ENTRY_BIAS: entry numbers start at the bias, so that small
//   databases exercise offsets beyond 2^32); only entry numbers >= the bias are ever dereferenced.
// ------------------------------------------------------------------------------------------------
#define BHIP_ACX_LINE_WORDS 14u
#define BHIP_REC_BYTES 4
#define BHIP_REC_PAD 0xFFFFFFFFu     // not a record (code 255 is never stored)
struct BhipAcxView {
	const uint4 *lines;                  // [ceil(n_words / 14)] x 64 bytes
	const uint32_t *rec;                 // 4 bytes per entry
};

#include "bhip_lanecode.h"

#ifdef __HIPCC__
typedef const uint32_t __attribute__((address_space(1))) *bhip_gptr_t;
// entry range of word w: first entry and length
__device__ __forceinline__ void bhip_acx_range(const BhipAcxView &a, uint32_t w, unsigned long long &beg, uint32_t &n) {
	const uint32_t blk = w / BHIP_ACX_LINE_WORDS, j = w - blk * BHIP_ACX_LINE_WORDS;
	const uint32_t *L = (const uint32_t *)(a.lines + 4ull * blk);
	const unsigned long long b0 = (unsigned long long)L[0] | (unsigned long long)L[1] << 32;
	const uint32_t hi = L[2 + j], lo_raw = L[1 + j];      // (for j = 0 lo_raw is the high word of the base: not used)
	const uint32_t lo = j ? lo_raw : 0u;
	beg = b0 + lo;
	n = hi - lo;
}
// record e as (clump, lane mask) -- the decoding of the code is a loop: for the kernels off the hot path
__device__ __forceinline__ uint2 bhip_acx_rec(const uint32_t *rec, unsigned long long e) {
	const uint32_t v = rec[e];
	return make_uint2(v & 0xFFFFFFu, bhip_lane_code_mask(v >> 24));
}
__device__ __forceinline__ uint32_t bhip_acx_clump(const uint32_t *rec, unsigned long long e) { return rec[e] & 0xFFFFFFu; }
// The raw record word for the hot kernels, without a branch around the load: lanes without a record read `dummy` (any mapped,
// 4-byte readable address) and get BHIP_REC_PAD.  A conditional load is compiled as a divergent branch with its own
// s_waitcnt vmcnt(0) -- a sequence of them is fully serialised, one memory latency each -- and the compiler turns a select
// around a load back into that branch unless the loaded word is used on every path: it is folded into `sink`, which the caller
// stores under a condition that never holds.  (A global-address-space pointer: a flat load also counts as an LDS operation,
// and every s_waitcnt lgkmcnt(0) in front of the next LDS round trip would wait for it.)
__device__ __forceinline__ uint32_t bhip_acx_raw_issue(const uint32_t *rec, unsigned long long e, bool valid, const void *dummy) {
	const uintptr_t addr = valid ? (uintptr_t)(rec + e) : (uintptr_t)dummy;
	return ((bhip_gptr_t)addr)[0];
}
__device__ __forceinline__ uint32_t bhip_acx_raw_finish(uint32_t raw, bool valid, uint32_t &sink) {
	sink ^= raw;
	return valid ? raw : BHIP_REC_PAD;
}
__device__ __forceinline__ uint32_t bhip_acx_raw_or_pad(const uint32_t *rec, unsigned long long e, bool valid, const void *dummy, uint32_t &sink) {
	return bhip_acx_raw_finish(bhip_acx_raw_issue(rec, e, valid, dummy), valid, sink);
}
// Counting sort of the records by query, first half: behind the shared record counter (n_out, err) sit the pointers to the
// per-query counters and to the rank array (SharedCtr in bhip_api.hip; null when the caller sorts differently).  Called by the
// kernels that write BhipHit records, with the position they reserved.
__device__ __forceinline__ void bhip_hit_rank(uint32_t *n_out, uint32_t pos, uint32_t q) {
	uint32_t *const *pp = (uint32_t *const *)(n_out + 2);
	uint32_t *cnt = pp[0], *rank = pp[1];
	if (cnt) rank[pos] = atomicAdd(&cnt[q], 1u);
}
#endif

#endif
