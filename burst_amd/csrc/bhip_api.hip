// burst_amd/csrc/bhip_api.hip -- C-ABI entry points of libburst_hip.so (include/burst_hip.h).
// Owns device memory, one HIP stream per handle, HIP-event timing of every phase, and the batch driver
// that replaces the bodies of the two OpenMP loops in do_alignments (burst.c:4077-4289, 4343-4484).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <string>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include <chrono>
#include <thread>
#include "burst_hip.h"
#include "bhip_internal.h"

// ---- kernels (bhip_kernels.hip) -------------------------------------------------------------------
__global__ void k_acx_offsets(const uint32_t *, uint64_t, int, uint32_t *, unsigned long long *);
__global__ void k_acx_lines(const uint32_t *, uint64_t, int, unsigned long long *, uint4 *);
__global__ void k_acx_decode(const uint8_t *, const unsigned long long *, const uint32_t *, BhipAcxView, uint64_t, int, uint32_t, uint8_t *, uint32_t *);
__global__ void k_transpose_refs(const uint8_t *, const uint64_t *, const uint32_t *, const uint64_t *, uint32_t, uint4 *, uint4 *);
__global__ void k_build_peq(const uint8_t *, const uint64_t *, const uint32_t *, uint32_t, int, int, BhipMatchMask, uint32_t *, const uint32_t *, uint32_t);
template <bool LDS_CNT> __global__ void k_prefilter(const uint8_t *, const uint64_t *, const uint16_t *, const uint32_t *, uint32_t,
	BhipAcxView, int, uint32_t, uint32_t *, const uint32_t *, uint32_t, uint2 *, uint32_t *, uint32_t *, uint32_t,
	unsigned long long *, const uint32_t *, const uint32_t *, const uint32_t *);
__global__ void k_prefilter_hash(const uint8_t *, const uint64_t *, const uint16_t *, const uint32_t *, uint32_t, BhipAcxView, int,
	const uint32_t *, uint32_t, uint2 *, uint32_t *, uint32_t *, uint32_t, unsigned long long *, const uint32_t *, uint32_t *, uint32_t *);
template <typename CNT> __global__ void k_prefilter_wave(const uint8_t *, const uint64_t *, const uint16_t *, const uint32_t *, uint32_t,
	BhipAcxView, int, uint32_t, const uint32_t *, uint32_t, uint2 *, uint32_t *, uint32_t *, uint32_t, unsigned long long *, const uint32_t *,
	const uint32_t *, const uint32_t *);
template <int NW> __global__ void k_myers(const uint2 *, const uint32_t *, uint64_t, uint32_t, uint32_t, const uint32_t *, const uint32_t *,
	const uint64_t *, const uint16_t *, const uint32_t *, const uint4 *, const uint64_t *, const uint32_t *, uint32_t,
	BhipRawHit *, uint32_t *, uint32_t, uint32_t *, uint8_t *, unsigned long long *, unsigned long long *);
template <int NWP> __global__ void k_myers_prefix(const uint2 *, const uint32_t *, uint64_t, uint32_t, uint32_t, const uint32_t *, const uint32_t *,
	const uint64_t *, const uint16_t *, const uint4 *, const uint64_t *, const uint32_t *, uint32_t, BhipWin *, uint32_t *, uint32_t,
	unsigned long long *, unsigned long long *);
template <int NW> __global__ void k_myers_window(const BhipWin *, const uint32_t *, uint32_t, int, const uint32_t *, const uint32_t *, const uint64_t *,
	const uint16_t *, const uint32_t *, const uint4 *, const uint64_t *, const uint32_t *, BhipRawHit *, uint32_t *, uint32_t, uint32_t *,
	unsigned long long *);
__global__ void k_extract_kmers(const uint4 *, const uint64_t *, const uint32_t *, const uint64_t *, uint32_t, uint32_t, int, unsigned long long *, uint16_t *, uint32_t *);
__global__ void k_attach_masks(BhipAcxView, uint64_t, const unsigned long long *, const uint16_t *, uint32_t, const uint32_t *, uint8_t *, uint32_t, uint32_t);
template <int HTB> __global__ void k_prefilter_mask(const uint2 *, const uint2 *, uint32_t, uint32_t, const uint8_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t,
	uint2 *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, unsigned long long *,
	uint2 *, uint32_t *, uint32_t);
template <int CB> __global__ void k_prefilter_cf(const uint2 *, const uint2 *, uint32_t, uint32_t, const uint8_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t,
	uint2 *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, unsigned long long *, unsigned long long *,
	uint2 *, uint32_t *, int);
__global__ void k_task_filter(const uint2 *, const uint32_t *, uint32_t, const uint32_t *, const uint32_t *, const uint32_t *, uint2 *, uint32_t *);
__global__ void k_seed_ranges(const uint8_t *, const uint64_t *, const uint32_t *, uint32_t, BhipAcxView, int, const uint32_t *, uint32_t, uint2 *, uint2 *, const uint32_t *, uint32_t, const uint16_t *);
template <int NWP> __global__ void k_myers_prefix_task(const uint2 *, const uint32_t *, uint32_t, const uint32_t *, const uint32_t *, const uint64_t *,
	const uint16_t *, const uint4 *, const uint64_t *, const uint32_t *, BhipWin *, uint32_t *, uint32_t, unsigned long long *);
template <bool WIDE> __global__ void k_rescore(const BhipRawHit *, const uint32_t *, uint32_t, const uint32_t *, const uint32_t *,
	const uint32_t *, int, const uint8_t *, const uint64_t *, const uint32_t *, const uint8_t *, const uint8_t *, const uint64_t *,
	const uint32_t *, const uint8_t *, BhipHit *, uint32_t *, uint32_t, uint32_t *, uint32_t *, uint32_t *, unsigned long long *,
	unsigned long long, uint32_t *, const uint32_t *, uint32_t, uint32_t, uint32_t);
__global__ void k_pack_queries(const uint8_t *, const uint64_t *, uint32_t, uint32_t, uint32_t *);
__global__ void k_unpack4(const uint8_t *, uint64_t, uint64_t, uint8_t *);
__global__ void k_span_fill(const uint64_t *, uint32_t, uint32_t, uint64_t, uint32_t, uint64_t *, uint32_t *, uint32_t *);
__global__ void k_route(const uint64_t *, const uint32_t *, uint32_t, const uint16_t *, const uint32_t *, const uint8_t *, uint32_t, uint32_t, uint32_t, int, int, int,
	uint32_t *, uint8_t *, uint32_t *, BhipStageInfo *);
__global__ void k_rescore_classify(const BhipRawHit *, const uint32_t *, uint32_t, const uint32_t *, int, const uint64_t *, const uint32_t *, const uint8_t *,
	const uint32_t *, BhipHit *, uint32_t *, uint32_t, uint32_t *, uint32_t *, uint32_t *, uint32_t *, uint32_t, int);
template <int SET> __global__ void k_rescore_reg(const BhipRawHit *, const uint32_t *, const uint32_t *, uint32_t, const uint64_t *, const uint8_t *, const uint32_t *, uint32_t,
	const uint8_t *, const uint64_t *, const uint32_t *, const uint8_t *, BhipHit *, uint32_t *, uint32_t, uint32_t *);

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
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) raw[i].ed += nx[raw[i].q];
}
__global__ void k_junk_adjust_best(uint32_t *__restrict__ best, const uint8_t *__restrict__ nx_six, uint32_t s0, uint32_t s1) {
	for (uint32_t s = s0 + blockIdx.x * blockDim.x + threadIdx.x; s < s1; s += gridDim.x * blockDim.x)
		if (nx_six[s] && best[s] != 0xFFFFFFFFu) best[s] += nx_six[s];
}

__global__ void k_hit_fix(BhipHit *__restrict__ out, const uint32_t *__restrict__ off, const uint32_t *__restrict__ cnt, uint32_t n_q) {
	for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n_q; q += gridDim.x * blockDim.x) {
		const uint32_t n = cnt[q];
		if (n < 2) continue;
		BhipHit *a = out + off[q];
		uint32_t gap = 1;
		while (gap < n / 3) gap = 3 * gap + 1;          // Shell sort (plain insertion sort for the usual 2..4 records)
		for (; gap >= 1; gap /= 3)
			for (uint32_t i = gap; i < n; ++i) {
				const BhipHit v = a[i];
				uint32_t j = i;
				for (; j >= gap && a[j - gap].refIx > v.refIx; j -= gap) a[j] = a[j - gap];
				a[j] = v;
			}
	}
}

static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
static int fail(int code, const char *fmt, ...) {
	va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
	return code;
}
// (for the other translation units of the library)
int bhip_fail_msg(int code, const char *fmt, ...) {
	va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
	return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
	return fail(BHIP_E_DEVICE, "%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); } while (0)

// grow-only device buffer
struct DBuf {
	void *p = nullptr; size_t cap = 0;
	int reserve(size_t bytes) {
		if (bytes <= cap) return 0;
		if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
		size_t want = bytes + bytes / 4 + 256;
		hipError_t e = hipMalloc(&p, want);
		if (e != hipSuccess) { p = nullptr; return fail(BHIP_E_DEVICE, "hipMalloc(%zu): %s", want, hipGetErrorString(e)); }
		cap = want; return 0;
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
	template <class T> T *as() const { return (T *)p; }
};

static const int kClasses[] = {2, 4, 6, 8, 10, 16, 32};
static const int kNumClasses = 7;
static int class_of_len(uint32_t len) {
	for (int i = 0; i < kNumClasses; ++i) if (len <= 32u * kClasses[i]) return i;
	return -1;
}

// device-side counters, one block copied back per call
struct Counters {
	uint32_t n_cand, n_raw, n_out, n_wide, err, pad0;
	uint32_t n_cand_cls[8];
	uint32_t n_wins_cls[8];
	uint32_t n_fb, pad2;
	uint32_t n_tasks_cls[8];
	uint32_t n_tasks2_cls[8];  // deferred lane tasks (lower bound above the query's best bound)
	uint32_t n_tasks2k_cls[8]; // ... of which kept by k_task_filter
	uint32_t n_wins2_cls[8];   // windows flagged by the second sweep
	uint32_t n_rs[12];         // re-scorer buckets: hits per band-width class
	unsigned long long wcol_sum, tcol_sum, unit_sum;
	unsigned long long col_sum, qlen_sum, ent_read, scratch_used;
	unsigned long long surv_sum;           // list records that passed the counting filter (k_prefilter_cf)
};

// One staged batch.  Query symbols with code 0 (anything outside the IUPAC nucleotide alphabet) cost 255 against every
// reference symbol (burst.c:170-190): such a symbol can only be aligned opposite a gap, so the edit distance of the query is
// (number of such symbols) + edit distance of the query without them, end columns unchanged.  When a staged batch holds any,
// the SEARCH kernels (seeds, prefilter, profiles, sweeps) work on a second view of the batch with those symbols removed and
// the budgets reduced (qcodes_s ...); k_junk_adjust adds the counts back before the re-scorer, which works on the original
// queries with the real cost table.  Without such symbols the search view is the batch itself.
struct StageSlot {
	DBuf qcodes, qcodes4, qoff, qemac, qsix, qrc, qflags, qmap, off_raw, plan, qpack, key, key_sorted, idx, idx_sorted, sort_tmp, info;
	DBuf qcodes_s, qoff_s, qemac_s, qpack_s, nx, nx_six;
	BhipStageInfo *info_pinned = nullptr;
	hipEvent_t ev_begin = nullptr, ev_done = nullptr;
	int state = 0;                        // 0 empty, 1 staged (not aligned yet), 2 active (aligned; may be run again)
	uint64_t seq = 0;
	bool resolved = false;                // routing read back, lists assigned
	bool st_valid = false, st_has_six = false, st_has_rc = false, st_has_junk = false, has_flags = false, has_qmap = false;
	uint32_t st_nq = 0, st_nshared = 0, st_maxlen = 0, st_maxE = 0, st_lanes = 1;
	float st_ms_h2d = 0;
	std::vector<BhipQuerySpan> spans;     // the caller's arrays (valid until the batch has been aligned): the host pass reads them
	const uint32_t *six_explicit = nullptr;
	uint32_t npf[16][7], nex[16][7], maxE[16][7], maxwords[16][7], qlist_off[16][7], maxlen_lane[16], n_entries_lane[16];
	uint64_t seed_words[16][7];
	void release_all() {
		DBuf *b[] = {&qcodes, &qcodes4, &qoff, &qemac, &qsix, &qrc, &qflags, &qmap, &off_raw, &plan, &qpack, &key, &key_sorted, &idx, &idx_sorted,
			&sort_tmp, &info, &qcodes_s, &qoff_s, &qemac_s, &qpack_s, &nx, &nx_six};
		for (DBuf *x : b) x->release();
		if (info_pinned) { (void)hipHostFree(info_pinned); info_pinned = nullptr; }
		if (ev_begin) { (void)hipEventDestroy(ev_begin); ev_begin = nullptr; }
		if (ev_done) { (void)hipEventDestroy(ev_done); ev_done = nullptr; }
	}
};

// One independent sub-pipeline of a staged batch: its own stream and scratch, a contiguous range of shared slots
// (so a forward entry and its reverse-complement twin are always in the same lane and `best[six]` is final when the
// lane's re-scorer runs).  Lanes overlap each other's latency-bound kernels (prefilter, window, re-scorer) with the
// VALU-bound column sweep, which itself is serialised on one dedicated stream (Handle::sweep_stream).
struct Lane {
	hipStream_t stream = nullptr;
	hipEvent_t ev_cls[kNumClasses][8];   // per class: 0 start, 1 peq done (sweep stream), 7 prefilter start, 2 prefilter done, 6 sweep start, 3 sweep(pf) done, 4 sweep(ex) done, 5 window done
	hipEvent_t ev_rs[2];
	hipEvent_t ev_ph[kNumClasses][2];    // per class: first window sweep done, second task sweep done
	hipEvent_t ev_pf[kNumClasses][3];    // per class: seed lookup start, hash kernel start, hash kernel done
	uint64_t seed_words[kNumClasses] = {0};
	uint32_t pf_launches = 0;
	bool pf_masked[kNumClasses] = {false};
	bool pruned[kNumClasses] = {false};   // a second (filtered) sweep ran for this class
	int pf_algo_used = 0;
	int pf_algo = 0;              // algorithm of this lane's next prefilter launches (follows opt_pf_algo: -1 = adapt)
	const uint32_t *qlist[kNumClasses] = {nullptr};      // (lane, class) lists of the current batch: entries of the slot's sorted index array
	DBuf peq, peqp, cand, candcnt, wins, raw, wide, scratch, fb_list, gcnt, counters, tasks, tasks2, tasks2k, wins2, rs_lists;
	// seed lookups (k_seed_ranges) per class: list ranges + query headers for the prefilter.  They depend on the staged batch alone,
	// so the lookups of batch k+1 run on the prefilter stream WHILE batch k is swept and re-scored (seed_ahead): memory-latency-bound
	// work beside VALU-bound work.  seeded_* say which staged batch the buffers of a class hold.
	DBuf ranges_c[kNumClasses], hdr_c[kNumClasses];
	bool seeded_ok[kNumClasses] = {false};
	uint64_t seeded_seq[kNumClasses] = {0};
	uint32_t seeded_n[kNumClasses] = {0}, seeded_W16[kNumClasses] = {0};
	hipEvent_t ev_seed[2][kNumClasses][2];   // [batch parity][class]: seed lookup start, done
	// match profiles built ahead for the next staged batch (when it has a single class in this lane): swapped in by enqueue_lane
	DBuf peq_alt, peqp_alt;
	bool alt_ok = false; uint64_t alt_seq = 0; int alt_cls = 0, alt_nwp = 0; uint32_t alt_n = 0;
	hipEvent_t ev_peq_alt[2], ev_peq_cur[2];  // profile build start, done: of the buffers built ahead / of the ones in use
	bool peq_ahead[kNumClasses] = {false};    // this batch's profiles of the class came from the build ahead
	uint64_t task_cap = 1 << 20;
	uint64_t cand_cap = 1 << 18, raw_cap = 1 << 18, win_cap = 1 << 20, scratch_cap = 1 << 18;
	uint32_t npf[kNumClasses] = {0}, nex[kNumClasses] = {0}, maxE[kNumClasses] = {0}, maxwords[kNumClasses] = {0}, maxlen = 0, n_entries = 0;
	Counters *hc_pinned = nullptr;        // pinned, so that the read-back of the counters does not block the enqueueing thread
	Counters hc;
	uint32_t launches = 0, prefix_words = 0;
	uint64_t n_pairs_ex = 0;
	bool masked = false;
};

// counters all lanes of a batch share; behind them the per-query record counters and the rank array of the counting sort: the
// re-scoring kernels take rank[pos] = cnt[q]++ when they write a record (bhip_hit_rank in bhip_internal.h reads the two pointers
// through the n_out pointer they already get), so that no separate counting pass runs between the re-scorer and the scatter
struct SharedCtr { uint32_t n_out, err; uint32_t *cnt; uint32_t *rank; };
__global__ void k_set_rank_ptrs(SharedCtr *sc, uint32_t *cnt, uint32_t *rank) { sc->cnt = cnt; sc->rank = rank; }
struct Handle {
	int device = 0, n_cu = 0;
	char dev_name[256];
	uint64_t hbm = 0;
	hipStream_t stream = nullptr;         // staging, sort, copies
	// software pipeline over lanes: stage streams run the same stage of consecutive lanes back to back, so that lane k+1's
	// prefilter and lane k-1's window/re-scoring overlap lane k's column sweep
	hipStream_t pf_stream = nullptr;      // peq + prefilter of every lane, in lane order
	hipStream_t sweep_stream = nullptr;   // every k_myers_prefix / k_myers launch, in lane order
	hipStream_t post_stream = nullptr;    // window stage + re-scorer + counter read-back, in lane order
	hipEvent_t ev[10];
	// database
	uint32_t n_clumps = 0, tot_refs = 0, max_clump_len = 0;
	DBuf ref, ref_lane, ref_off, clump_len, lut;      // ref: 16 lanes interleaved per 32-column chunk; ref_lane: each lane contiguous
	BhipMatchMask mm;
	bool has_acx = false; int K = 0;
	// accelerator: two-level offsets + 5-byte (clump, lane mask) records (bhip_internal.h); entry numbers start at acx_bias
	// (0, or the test hook BHIP_TEST_ENTRY_BIAS that pushes a small database's offsets beyond 2^32)
	DBuf acx_lines, acx_rec, bad; uint32_t n_bad = 0; uint64_t n_ent = 0, acx_bias = 0;
	BhipAcxView acx_view() const {
		BhipAcxView v; v.lines = acx_lines.as<uint4>();
		v.rec = acx_rec.as<uint8_t>() - acx_bias * (uint64_t)BHIP_REC_BYTES; return v;
	}
	bool has_masks = false;       // per-entry lane masks were built at upload (lane-resolved prefilter)
	int opt_lane_masks = 1;       // use them
	// staged batches: two slots, so that the upload and routing of batch k+1 (stage_stream) run while batch k is aligned
	StageSlot slots[3];                   // one batch being aligned, one staged (its seed lookups and profiles run ahead), one being staged
	StageSlot *cur = &slots[0];           // slot of the batch being aligned
	uint64_t stage_seq = 0;
	hipStream_t stage_stream = nullptr;
	// records of the last aligned batch, complete but not delivered (the caller's buffer was too small): delivered by the next call
	bool res_valid = false; uint64_t res_seq = 0; int res_all_hits = 0; uint32_t res_n = 0; BhipStats res_stats;
	int opt_host_routing = 0;             // 1 = route every batch on the host (the pass that handles symbols of code 0); test hook
	// batch-wide buffers
	DBuf best, out, shared_ctr, mins, pairs;
	SharedCtr *hsc_pinned = nullptr;      // read-back of shared_ctr behind the chain (pinned: no blocking copy on the way out of a batch)
	const uint8_t *s_codes() const { return cur->st_has_junk ? cur->qcodes_s.as<uint8_t>() : cur->qcodes.as<uint8_t>(); }
	const uint64_t *s_off() const { return cur->st_has_junk ? cur->qoff_s.as<uint64_t>() : cur->qoff.as<uint64_t>(); }
	const uint16_t *s_emac() const { return cur->st_has_junk ? cur->qemac_s.as<uint16_t>() : cur->qemac.as<uint16_t>(); }
	const uint32_t *s_pack() const { return cur->st_has_junk ? cur->qpack_s.as<uint32_t>() : cur->qpack.as<uint32_t>(); }
	DBuf sort_keys, sort_keys2, sort_idx, sort_tmp, out_sorted, out_sorted2;   // sort_keys / sort_keys2: per-query record counts / offsets; sort_idx: rank of a record inside its query
	uint64_t out_cap = 1 << 20;
	std::vector<uint32_t> h_clump_len;
	BhipStats stats;
	std::vector<Lane *> lanes;
	int opt_two_stage = 1;        // 1 = prefix filter + windowed full-length stage when it pays, 0 = always the one-stage sweep
	int opt_prefilter_stride = 0; // 0 = automatic sparse seeds, s > 0 = every s-th word (1 = the reference's scheme)
	int opt_lanes = 1;            // sub-pipelines per staged batch (the stage kernels fill the chip on their own; > 1 only helps small batches)
	int opt_sweep_blocks = 8;     // 256-thread blocks per CU of the column-sweep kernels
	// asynchronous hand-over of the records (option "async_d2h"): two device buffers alternate, the copy of call k runs on its
	// own stream while call k+1 computes; the caller's buffers are page-locked once and stay registered
	int opt_async_d2h = 0, out_idx = 0;
	hipStream_t copy_stream = nullptr;
	hipEvent_t ev_sorted = nullptr, ev_copied[2] = {nullptr, nullptr};
	bool copy_pending[2] = {false, false};
	void *reg_ptr[2] = {nullptr, nullptr}; size_t reg_bytes[2] = {0, 0};
	int last_out = 0;             // which of the two sorted buffers holds the last call's records
	uint64_t last_n_out = 0;      // records of the last bhip_align_staged call, still resident (sorted) in out_sorted
	int opt_prune = 1;            // second sweep for lanes whose seed count bounds their edit distance above the first sweep's best
	int opt_lane_min = 32768;     // fewest entries a sub-pipeline is worth opening for
	int opt_rescore_reg = 1;      // register-band re-scorer for narrow bands (0 = LDS band only)
	int opt_pf_waves = 0;         // single-wave blocks per CU of the lane-resolved prefilter (0 = as many as the LDS allows, <= 12)
	int opt_pf_algo = -1;         // 0 = counting filter + exact lane table (k_prefilter_cf), 1 = exact clump hash table in two passes
	                              // (k_prefilter_mask), -1 = start with 0 and switch a lane to 1 when more than 20 % of its records survive the filter
	int opt_pf_table = 0;         // log2 of the per-query hash table (0 = from the workload: 9, 10 or 11)
	int opt_seed_ahead = 1;       // seed lookups of the next staged batch run while the current one is swept
	int opt_seed_ahead_blocks = 2; // 256-thread blocks per CU of a seed kernel that runs ahead (0 = one block per 256 lookups, as in place); 2: +2.3 % on the bench
	int opt_peq_ahead_blocks = 16; // 256-thread blocks per CU of a profile build that runs ahead
	double acx_wmean = 0.0;       // occurrence-weighted mean .acx list length
};

static float ev_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; }

extern "C" const char *bhip_last_error(void) { return g_err; }
extern "C" int bhip_abi_version(void) { return BHIP_ABI_VERSION; }

static void lane_destroy(Lane *L) {
	if (!L) return;
	DBuf *all[] = {&L->peq, &L->peqp, &L->cand, &L->candcnt, &L->wins, &L->raw, &L->wide, &L->scratch, &L->fb_list, &L->gcnt, &L->counters, &L->tasks, &L->tasks2, &L->tasks2k, &L->wins2, &L->rs_lists};
	for (DBuf *b : all) b->release();
	L->peq_alt.release(); L->peqp_alt.release();
	for (auto &e : L->ev_peq_alt) if (e) (void)hipEventDestroy(e);
	for (auto &e : L->ev_peq_cur) if (e) (void)hipEventDestroy(e);
	for (DBuf &b : L->ranges_c) b.release();
	for (DBuf &b : L->hdr_c) b.release();
	for (auto &pe : L->ev_seed) for (auto &ce : pe) for (auto &e : ce) if (e) (void)hipEventDestroy(e);
	for (auto &ce : L->ev_cls) for (auto &e : ce) if (e) (void)hipEventDestroy(e);
	for (auto &e : L->ev_rs) if (e) (void)hipEventDestroy(e);
	for (auto &ce : L->ev_pf) for (auto &e : ce) if (e) (void)hipEventDestroy(e);
	for (auto &ce : L->ev_ph) for (auto &e : ce) if (e) (void)hipEventDestroy(e);
	if (L->hc_pinned) (void)hipHostFree(L->hc_pinned);
	delete L;
}

static int lane_create(Handle *h, Lane **out) {
	Lane *L = new Lane();
	memset(L->ev_cls, 0, sizeof L->ev_cls); memset(L->ev_rs, 0, sizeof L->ev_rs); memset(L->ev_pf, 0, sizeof L->ev_pf); memset(L->ev_ph, 0, sizeof L->ev_ph); memset(L->ev_seed, 0, sizeof L->ev_seed); memset(L->ev_peq_alt, 0, sizeof L->ev_peq_alt); memset(L->ev_peq_cur, 0, sizeof L->ev_peq_cur);
	L->stream = h->stream;      // (the kernel-level entry points run a lane on the handle's own stream)
	for (auto &ce : L->ev_cls) for (auto &e : ce) if (hipEventCreate(&e) != hipSuccess) { lane_destroy(L); return fail(BHIP_E_DEVICE, "hipEventCreate failed"); }
	for (auto &e : L->ev_rs) if (hipEventCreate(&e) != hipSuccess) { lane_destroy(L); return fail(BHIP_E_DEVICE, "hipEventCreate failed"); }
	for (auto &ce : L->ev_pf) for (auto &e : ce) if (hipEventCreate(&e) != hipSuccess) { lane_destroy(L); return fail(BHIP_E_DEVICE, "hipEventCreate failed"); }
	for (auto &ce : L->ev_ph) for (auto &e : ce) if (hipEventCreate(&e) != hipSuccess) { lane_destroy(L); return fail(BHIP_E_DEVICE, "hipEventCreate failed"); }
	for (auto &pe : L->ev_seed) for (auto &ce : pe) for (auto &e : ce) if (hipEventCreate(&e) != hipSuccess) { lane_destroy(L); return fail(BHIP_E_DEVICE, "hipEventCreate failed"); }
	for (auto &e : L->ev_peq_alt) if (hipEventCreate(&e) != hipSuccess) { lane_destroy(L); return fail(BHIP_E_DEVICE, "hipEventCreate failed"); }
	for (auto &e : L->ev_peq_cur) if (hipEventCreate(&e) != hipSuccess) { lane_destroy(L); return fail(BHIP_E_DEVICE, "hipEventCreate failed"); }
	int rc = L->counters.reserve(sizeof(Counters));
	if (rc) { lane_destroy(L); return rc; }
	if (hipHostMalloc((void **)&L->hc_pinned, sizeof(Counters), hipHostMallocDefault) != hipSuccess) { lane_destroy(L); return fail(BHIP_E_DEVICE, "hipHostMalloc failed"); }
	(void)h;
	*out = L;
	return 0;
}

extern "C" void bhip_destroy(void *handle) {
	Handle *h = (Handle *)handle;
	if (!h) return;
	(void)hipSetDevice(h->device);
	if (h->stream) (void)hipStreamSynchronize(h->stream);
	if (h->sweep_stream) (void)hipStreamSynchronize(h->sweep_stream);
	if (h->pf_stream) (void)hipStreamSynchronize(h->pf_stream);
	if (h->post_stream) (void)hipStreamSynchronize(h->post_stream);
	if (h->stage_stream) (void)hipStreamSynchronize(h->stage_stream);
	for (StageSlot &S : h->slots) S.release_all();
	if (h->hsc_pinned) (void)hipHostFree(h->hsc_pinned);
	for (Lane *L : h->lanes) lane_destroy(L);
	DBuf *all[] = {&h->ref, &h->ref_lane, &h->ref_off, &h->clump_len, &h->lut, &h->acx_lines, &h->acx_rec, &h->bad,
		&h->best, &h->out, &h->shared_ctr, &h->mins, &h->pairs, &h->sort_keys, &h->sort_keys2, &h->sort_idx,
		&h->sort_tmp, &h->out_sorted, &h->out_sorted2};
	for (int o = 0; o < 2; ++o) {
		if (h->copy_pending[o] && h->ev_copied[o]) (void)hipEventSynchronize(h->ev_copied[o]);
		if (h->reg_ptr[o]) (void)hipHostUnregister(h->reg_ptr[o]);
		if (h->ev_copied[o]) (void)hipEventDestroy(h->ev_copied[o]);
	}
	if (h->ev_sorted) (void)hipEventDestroy(h->ev_sorted);
	if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
	for (DBuf *b : all) b->release();
	for (auto &e : h->ev) if (e) (void)hipEventDestroy(e);
	if (h->stream) (void)hipStreamDestroy(h->stream);        // sweep_stream and post_stream are aliases of it
	if (h->pf_stream) (void)hipStreamDestroy(h->pf_stream);
	if (h->stage_stream) (void)hipStreamDestroy(h->stage_stream);
	delete h;
}

static int ensure_lanes(Handle *h, uint32_t n);

// per-entry lane masks (see bhip_kernels.hip, "Lane-resolved accelerator").  Skipped (has_masks stays false, clump-level
// behaviour) when the database has >= 2^31 reference positions (device-sort item limit) or the scratch does not fit.
struct BitOrU16 { __host__ __device__ uint16_t operator()(const uint16_t &a, const uint16_t &b) const { return (uint16_t)(a | b); } };
static int build_lane_masks(Handle *h, const std::vector<uint64_t> &chunk_off) {
	(void)chunk_off;
	const uint32_t nC = h->n_clumps;
	std::vector<uint64_t> key_off(nC + 1);
	key_off[0] = 0;
	for (uint32_t c = 0; c < nC; ++c) key_off[c + 1] = key_off[c] + 16ull * h->h_clump_len[c];
	if (key_off[nC] == 0 || !h->n_ent) return 0;
	// The (word, clump, lane) tuples of the whole database may not fit next to it (26 bytes of sort space per reference
	// position): the clumps are processed in slices, each slice sorted and folded on its own and joined to the list entries
	// of its clumps.  BHIP_MASK_SLICE (reference positions per slice) is the test hook for small databases.
	size_t free_b = 0, total_b = 0;
	if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
	const double after_masks = (double)free_b - 4.0 * (double)nC - 65536.0;      // the masks go into the records that are already there
	if (after_masks <= 0) return 0;
	uint64_t slice_items = (uint64_t)std::min<double>(2147483000.0, after_masks * 0.8 / 26.0);
	if (const char *ev = getenv("BHIP_MASK_SLICE")) { const long long v = atoll(ev); if (v > 0) slice_items = (uint64_t)v; }
	uint64_t biggest = 0;
	for (uint32_t c = 0; c < nC; ++c) biggest = std::max(biggest, key_off[c + 1] - key_off[c]);
	if (slice_items < biggest) { if ((double)biggest * 26.0 > after_masks) return 0; slice_items = biggest; }
	DBuf d_koff, k0, k1, v0, v1, nruns, tmp, amb;
	int rc;
	#define BLM(x) do { if ((rc = (x))) { d_koff.release(); k0.release(); k1.release(); v0.release(); v1.release(); nruns.release(); tmp.release(); amb.release(); return rc; } } while (0)
	#define BLMH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { BLM(fail(BHIP_E_DEVICE, "%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_))); } } while (0)
	const uint64_t cap_items = std::min<uint64_t>(slice_items, key_off[nC]);
	BLM(d_koff.reserve((nC + 1) * 8)); BLM(k0.reserve(cap_items * 8)); BLM(k1.reserve(cap_items * 8)); BLM(v0.reserve(cap_items * 2)); BLM(v1.reserve(cap_items * 2));
	BLM(nruns.reserve(16)); BLM(amb.reserve((size_t)nC * 4 + 16));
	BLMH(hipMemsetAsync(amb.p, 0, (size_t)nC * 4, h->stream));
	BLMH(hipMemcpyAsync(d_koff.p, key_off.data(), (nC + 1) * 8, hipMemcpyHostToDevice, h->stream));
	const int end_bit = 2 * h->K + 24;
	uint32_t n_slices = 0;
	for (uint32_t c0 = 0; c0 < nC;) {
		uint32_t c1 = c0 + 1;
		while (c1 < nC && key_off[c1 + 1] - key_off[c0] <= slice_items) ++c1;
		const uint64_t n_items = key_off[c1] - key_off[c0];
		++n_slices;
		hipLaunchKernelGGL(k_extract_kmers, dim3(std::min<uint32_t>(((c1 - c0) * 16 + 255) / 256, (uint32_t)h->n_cu * 16)), dim3(256), 0, h->stream, h->ref.as<uint4>(),
			h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), d_koff.as<uint64_t>(), c0, c1, h->K, k0.as<unsigned long long>(), v0.as<uint16_t>(), amb.as<uint32_t>());
		BLMH(hipGetLastError());
		size_t tb = 0;
		hipcub::DoubleBuffer<unsigned long long> dk(k0.as<unsigned long long>(), k1.as<unsigned long long>());
		hipcub::DoubleBuffer<uint16_t> dv(v0.as<uint16_t>(), v1.as<uint16_t>());
		BLMH(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, dv, (int)n_items, 0, end_bit, h->stream));
		BLM(tmp.reserve(tb));
		BLMH(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, dk, dv, (int)n_items, 0, end_bit, h->stream));
		// invalid slots carry key ~0, which after masking to end_bit sorts last (all ones) -- their run is simply never looked up
		unsigned long long *skeys = dk.Current(); uint16_t *svals = dv.Current();
		unsigned long long *ukeys = dk.Alternate(); uint16_t *umasks = dv.Alternate();
		size_t tb2 = 0;
		BLMH(hipcub::DeviceReduce::ReduceByKey(nullptr, tb2, skeys, ukeys, svals, umasks, nruns.as<uint32_t>(), BitOrU16(), (int)n_items, h->stream));
		BLM(tmp.reserve(tb2));
		BLMH(hipcub::DeviceReduce::ReduceByKey(tmp.p, tb2, skeys, ukeys, svals, umasks, nruns.as<uint32_t>(), BitOrU16(), (int)n_items, h->stream));
		uint32_t n_unique = 0;
		BLMH(hipMemcpyAsync(&n_unique, nruns.p, 4, hipMemcpyDeviceToHost, h->stream));
		BLMH(hipStreamSynchronize(h->stream));
		hipLaunchKernelGGL(k_attach_masks, dim3((uint32_t)h->n_cu * 32), dim3(256), 0, h->stream,
			h->acx_view(), (uint64_t)(1ull << (2 * h->K)), ukeys, umasks, n_unique, amb.as<uint32_t>(), (uint8_t *)h->acx_view().rec, c0, c1);
		BLMH(hipGetLastError());
		BLMH(hipStreamSynchronize(h->stream));
		c0 = c1;
	}
	#undef BLM
	#undef BLMH
	d_koff.release(); k0.release(); k1.release(); v0.release(); v1.release(); nruns.release(); tmp.release(); amb.release();
	if (getenv("BHIP_DEBUG")) fprintf(stderr, "[bhip] lane masks: %llu reference positions in %u slice(s)\n", (unsigned long long)key_off[nC], n_slices);
	h->has_masks = true;
	return 0;
}

extern "C" int bhip_init(int device, const void *edx_packed, const uint32_t *clump_len, uint32_t n_clumps, uint32_t tot_refs,
                         const uint32_t *acx_lens, const void *acx_lists, int acx_fmt, int K,
                         const uint32_t *badlist, uint32_t n_bad, const uint8_t score_lut[256], int xalpha, void **handle) {
	if (!handle) return fail(BHIP_E_ARG, "handle is NULL");
	*handle = nullptr;
	if (xalpha) return fail(BHIP_E_ARG, "xalpha (-x) databases are not supported on the device");
	if (!edx_packed || !clump_len || !n_clumps || !score_lut) return fail(BHIP_E_ARG, "empty database");
	if (acx_lens && (K < 4 || K > 15 || !acx_lists || (acx_fmt != 0 && acx_fmt != 1)))
		return fail(BHIP_E_ARG, "bad accelerator arguments (K=%d fmt=%d)", K, acx_fmt);
	int ndev = 0;
	HIPCHK(hipGetDeviceCount(&ndev));
	if (device < 0 || device >= ndev) return fail(BHIP_E_DEVICE, "device %d not present (%d visible)", device, ndev);
	HIPCHK(hipSetDevice(device));
	Handle *h = new Handle();
	memset(h->ev, 0, sizeof h->ev);
	memset(&h->stats, 0, sizeof h->stats);
	h->device = device;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
		h->n_cu = prop.multiProcessorCount; h->hbm = prop.totalGlobalMem;
		snprintf(h->dev_name, sizeof h->dev_name, "%s (%s)", prop.name, prop.gcnArchName);
	}
	if (h->n_cu <= 0) h->n_cu = 256;
	#define INITCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
		fail(BHIP_E_DEVICE, "%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); bhip_destroy(h); return BHIP_E_DEVICE; } } while (0)
	#define INITRC(x) do { int rc_ = (x); if (rc_) { bhip_destroy(h); return rc_; } } while (0)
	// Four streams in all -- the HIP runtime multiplexes streams onto 4 hardware queues by default (GPU_MAX_HW_QUEUES), and
	// streams that share a queue serialise (measured: with seven streams the staging kernels of batch k+1 delayed the window
	// sweep of batch k by 0.25 ms and the hand-over copy sat in front of the next batch: 248 -> 326 M reads/s with more queues).
	// One chain (profiles, sweeps, re-scoring, sort), the seed/prefilter stream beside it, staging, record hand-over.
	INITCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
	h->sweep_stream = h->stream; h->post_stream = h->stream;
	INITCHK(hipStreamCreateWithFlags(&h->pf_stream, hipStreamNonBlocking));
	INITCHK(hipStreamCreateWithFlags(&h->stage_stream, hipStreamNonBlocking));
	for (auto &e : h->ev) INITCHK(hipEventCreate(&e));
	h->n_clumps = n_clumps; h->tot_refs = tot_refs;
	h->h_clump_len.assign(clump_len, clump_len + n_clumps);
	for (int a = 0; a < 16; ++a) {
		uint16_t m = 0;
		for (int b = 0; b < 16; ++b) if (score_lut[16 * a + b] == 0) m |= (uint16_t)(1u << b);
		h->mm.m[a] = m;
	}
	// reference area: upload as on disk, transpose on the device
	std::vector<uint64_t> src_off(n_clumps + 1), dst_off(n_clumps + 1);
	src_off[0] = dst_off[0] = 0;
	for (uint32_t c = 0; c < n_clumps; ++c) {
		src_off[c + 1] = src_off[c] + clump_len[c] / 2u + (clump_len[c] & 1);
		dst_off[c + 1] = dst_off[c] + (clump_len[c] + 31) / 32u;
		if (clump_len[c] > h->max_clump_len) h->max_clump_len = clump_len[c];
	}
	{
		DBuf d_src, d_srcoff;
		INITRC(d_src.reserve(src_off[n_clumps] * 16 + 16));
		INITRC(d_srcoff.reserve((n_clumps + 1) * sizeof(uint64_t)));
		INITRC(h->ref.reserve(dst_off[n_clumps] * 256 + 256));
		INITRC(h->ref_lane.reserve(dst_off[n_clumps] * 256 + 256));
		INITRC(h->ref_off.reserve((n_clumps + 1) * sizeof(uint64_t)));
		INITRC(h->clump_len.reserve(n_clumps * sizeof(uint32_t)));
		INITRC(h->lut.reserve(256));
		INITCHK(hipMemcpyAsync(d_src.p, edx_packed, src_off[n_clumps] * 16, hipMemcpyHostToDevice, h->stream));
		INITCHK(hipMemcpyAsync(d_srcoff.p, src_off.data(), (n_clumps + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
		INITCHK(hipMemcpyAsync(h->ref_off.p, dst_off.data(), (n_clumps + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
		INITCHK(hipMemcpyAsync(h->clump_len.p, clump_len, n_clumps * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
		INITCHK(hipMemcpyAsync(h->lut.p, score_lut, 256, hipMemcpyHostToDevice, h->stream));
		const uint32_t grid = std::min<uint32_t>(n_clumps, (uint32_t)h->n_cu * 8);
		hipLaunchKernelGGL(k_transpose_refs, dim3(grid), dim3(256), 0, h->stream, d_src.as<uint8_t>(), d_srcoff.as<uint64_t>(),
			h->clump_len.as<uint32_t>(), h->ref_off.as<uint64_t>(), n_clumps, h->ref.as<uint4>(), h->ref_lane.as<uint4>());
		INITCHK(hipGetLastError());
		INITCHK(hipStreamSynchronize(h->stream));
		d_src.release(); d_srcoff.release();
	}
	if (acx_lens) {
		const uint64_t nw = 1ull << (2 * K), nblk = (nw + 255) >> 8;
		// Lens[4^K] goes up as it is; block-relative offsets, block bases, byte offsets of the packed lists, the total and the
		// occurrence-weighted mean list length are scans / reductions on the device (at K = 15 the table has 2^30 words)
		DBuf d_lens, d_red, d_tmp, d_bsum, d_bdelta, d_bbase, d_lsum, d_lbase;
		#define ACXFREE() do { d_lens.release(); d_red.release(); d_tmp.release(); d_bsum.release(); d_bdelta.release(); d_bbase.release(); d_lsum.release(); d_lbase.release(); } while (0)
		#define ACXRC(x) do { int rc_ = (x); if (rc_) { ACXFREE(); bhip_destroy(h); return rc_; } } while (0)
		#define ACXCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ACXFREE(); \
			fail(BHIP_E_DEVICE, "%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); bhip_destroy(h); return BHIP_E_DEVICE; } } while (0)
		if (const char *ev = getenv("BHIP_TEST_ENTRY_BIAS")) h->acx_bias = strtoull(ev, nullptr, 0);
		ACXRC(d_lens.reserve(nw * sizeof(uint32_t)));
		ACXRC(d_red.reserve(64));
		ACXRC(d_bsum.reserve((nblk + 2) * 8));
		ACXRC(d_bdelta.reserve(nw * sizeof(uint32_t) + 16));
		ACXRC(d_bbase.reserve((nblk + 2) * 8));
		const uint64_t n_lines = (nw + BHIP_ACX_LINE_WORDS - 1) / BHIP_ACX_LINE_WORDS;
		ACXRC(h->acx_lines.reserve((n_lines + 1) * 64));
		ACXRC(d_lsum.reserve((n_lines + 2) * 8));
		ACXRC(d_lbase.reserve((n_lines + 2) * 8));
		ACXCHK(hipMemcpyAsync(d_lens.p, acx_lens, nw * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
		auto to_sq = [] __host__ __device__(uint32_t n) -> double { return (double)n * (double)n; };
		hipcub::TransformInputIterator<double, decltype(to_sq), const uint32_t *> it_sq(d_lens.as<uint32_t>(), to_sq);
		unsigned long long *r_tot = d_red.as<unsigned long long>(); double *r_sq = (double *)(r_tot + 1); uint32_t *r_max = (uint32_t *)(r_tot + 2);
		size_t tb = 0, tb1 = 0;
		ACXCHK(hipcub::DeviceReduce::Sum(nullptr, tb1, it_sq, r_sq, (int)nw, h->stream)); tb = std::max(tb, tb1);
		ACXCHK(hipcub::DeviceReduce::Max(nullptr, tb1, d_lens.as<uint32_t>(), r_max, (int)nw, h->stream)); tb = std::max(tb, tb1);
		ACXCHK(hipcub::DeviceScan::ExclusiveScan(nullptr, tb1, d_lsum.as<unsigned long long>(), d_lbase.as<unsigned long long>(), hipcub::Sum(), 0ull, (int)(n_lines + 1), h->stream)); tb = std::max(tb, tb1);
		ACXRC(d_tmp.reserve(tb + 16));
		tb1 = tb; ACXCHK(hipcub::DeviceReduce::Sum(d_tmp.p, tb1, it_sq, r_sq, (int)nw, h->stream));
		tb1 = tb; ACXCHK(hipcub::DeviceReduce::Max(d_tmp.p, tb1, d_lens.as<uint32_t>(), r_max, (int)nw, h->stream));
		const uint32_t og = (uint32_t)std::min<uint64_t>(nblk, (uint64_t)h->n_cu * 16);
		// entry offsets: block sums -> 64-bit bases -> lines
		{
			const uint32_t lg = (uint32_t)std::min<uint64_t>((n_lines + 255) / 256, (uint64_t)h->n_cu * 32);
			ACXCHK(hipMemsetAsync(d_lsum.p, 0, (n_lines + 2) * 8, h->stream));
			hipLaunchKernelGGL(k_acx_lines, dim3(lg), dim3(256), 0, h->stream, d_lens.as<uint32_t>(), nw, 0, d_lsum.as<unsigned long long>(), h->acx_lines.as<uint4>());
			ACXCHK(hipGetLastError());
			tb1 = tb; ACXCHK(hipcub::DeviceScan::ExclusiveScan(d_tmp.p, tb1, d_lsum.as<unsigned long long>(), d_lbase.as<unsigned long long>(), hipcub::Sum(), (unsigned long long)h->acx_bias, (int)(n_lines + 1), h->stream));
			hipLaunchKernelGGL(k_acx_lines, dim3(lg), dim3(256), 0, h->stream, d_lens.as<uint32_t>(), nw, 1, d_lbase.as<unsigned long long>(), h->acx_lines.as<uint4>());
			ACXCHK(hipGetLastError());
		}
		// byte offsets of the packed lists on disk
		ACXCHK(hipMemsetAsync(d_bsum.p, 0, (nblk + 2) * 8, h->stream));
		hipLaunchKernelGGL(k_acx_offsets, dim3(og), dim3(256), 0, h->stream, d_lens.as<uint32_t>(), nw, acx_fmt == 1 ? 2 : 1, d_bdelta.as<uint32_t>(), d_bsum.as<unsigned long long>());
		ACXCHK(hipGetLastError());
		tb1 = tb; ACXCHK(hipcub::DeviceScan::ExclusiveScan(d_tmp.p, tb1, d_bsum.as<unsigned long long>(), d_bbase.as<unsigned long long>(), hipcub::Sum(), 0ull, (int)(nblk + 1), h->stream));
		unsigned long long red[3], tot_b = 0, bytes = 0;
		ACXCHK(hipMemcpyAsync(red, d_red.p, 24, hipMemcpyDeviceToHost, h->stream));
		ACXCHK(hipMemcpyAsync(&tot_b, d_lbase.as<unsigned long long>() + n_lines, 8, hipMemcpyDeviceToHost, h->stream));
		ACXCHK(hipMemcpyAsync(&bytes, d_bbase.as<unsigned long long>() + nblk, 8, hipMemcpyDeviceToHost, h->stream));
		ACXCHK(hipStreamSynchronize(h->stream));
		d_lsum.release(); d_lbase.release();
		const uint64_t tot = tot_b - h->acx_bias;
		double sq; memcpy(&sq, &red[1], 8);
		uint32_t maxlen; memcpy(&maxlen, &red[2], 4);
		h->acx_wmean = tot ? sq / (double)tot : 0.0;
		if (maxlen > n_clumps || maxlen >= (1u << 24)) { ACXFREE(); fail(BHIP_E_ARG, "an accelerator list has %u entries, the database %u clumps (wrong K for this file?)", maxlen, n_clumps); bhip_destroy(h); return BHIP_E_ARG; }
		if (tot_b >= (1ull << 40)) { ACXFREE(); fail(BHIP_E_ARG, "accelerator with %llu entries exceeds the 40-bit entry numbers of the device layout", (unsigned long long)tot); bhip_destroy(h); return BHIP_E_ARG; }
		// the packed list area goes up as it is on disk and is decoded to 5-byte records by the device
		ACXRC(h->acx_rec.reserve(tot * BHIP_REC_BYTES + 16));
		{
			DBuf d_lists, d_flag;
			int rcl;
			if ((rcl = d_lists.reserve(bytes + 16)) || (rcl = d_flag.reserve(16))) { d_lists.release(); d_flag.release(); ACXRC(rcl); }
			#define ACXCHK2(x) do { hipError_t e2_ = (x); if (e2_ != hipSuccess) { d_lists.release(); d_flag.release(); ACXFREE(); \
				fail(BHIP_E_DEVICE, "%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e2_)); bhip_destroy(h); return BHIP_E_DEVICE; } } while (0)
			if (bytes) ACXCHK2(hipMemcpyAsync(d_lists.p, acx_lists, bytes, hipMemcpyHostToDevice, h->stream));
			ACXCHK2(hipMemsetAsync(d_flag.p, 0, 16, h->stream));
			hipLaunchKernelGGL(k_acx_decode, dim3((uint32_t)h->n_cu * 32), dim3(256), 0, h->stream, d_lists.as<uint8_t>(), d_bbase.as<unsigned long long>(),
				d_bdelta.as<uint32_t>(), h->acx_view(), nw, acx_fmt, n_clumps, (uint8_t *)h->acx_view().rec, d_flag.as<uint32_t>());
			ACXCHK2(hipGetLastError());
			uint32_t worst = 0;
			ACXCHK2(hipMemcpyAsync(&worst, d_flag.p, 4, hipMemcpyDeviceToHost, h->stream));
			ACXCHK2(hipStreamSynchronize(h->stream));
			#undef ACXCHK2
			d_lists.release(); d_flag.release();
			ACXFREE();
			if (worst) { fail(BHIP_E_ARG, "an accelerator entry refers to clump %u >= %u", worst, n_clumps); bhip_destroy(h); return BHIP_E_ARG; }
		}
		#undef ACXFREE
		#undef ACXRC
		#undef ACXCHK
		if (getenv("BHIP_DEBUG")) fprintf(stderr, "[bhip] accelerator: K=%d, %llu entries (first entry number %llu), %.2f B per entry on the device (records %d B + offsets)\n", K,
			(unsigned long long)tot, (unsigned long long)h->acx_bias, tot ? (double)(tot * BHIP_REC_BYTES + n_lines * 64) / (double)tot : 0.0, BHIP_REC_BYTES);
		h->n_bad = n_bad;
		INITRC(h->bad.reserve((n_bad + 1) * sizeof(uint32_t)));
		if (n_bad) {
			for (uint32_t i = 0; i < n_bad; ++i) if (badlist[i] >= n_clumps) { fail(BHIP_E_ARG, "BadList entry out of range"); bhip_destroy(h); return BHIP_E_ARG; }
			INITCHK(hipMemcpy(h->bad.p, badlist, n_bad * sizeof(uint32_t), hipMemcpyHostToDevice));
		}
		h->has_acx = true; h->K = K; h->n_ent = tot;
		if (!getenv("BHIP_NO_LANE_MASKS")) { int rcm = build_lane_masks(h, dst_off); if (rcm) { bhip_destroy(h); return rcm; } }
	}
	INITRC(ensure_lanes(h, 1));
	*handle = h;
	return BHIP_OK;
}


extern "C" int bhip_device_info(void *handle, char *name, int name_cap, int *n_cu, uint64_t *hbm_bytes) {
	Handle *h = (Handle *)handle;
	if (!h) return fail(BHIP_E_ARG, "null handle");
	if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s", h->dev_name);
	if (n_cu) *n_cu = h->n_cu;
	if (hbm_bytes) *hbm_bytes = h->hbm;
	return BHIP_OK;
}

extern "C" int bhip_set_option(void *handle, const char *name, long long value) {
	Handle *h = (Handle *)handle;
	if (!h || !name) return fail(BHIP_E_ARG, "null argument");
	if (!strcmp(name, "prefilter_stride")) {
		if (value < 0 || value > 64) return fail(BHIP_E_ARG, "prefilter_stride must be 0 (auto) .. 64");
		h->opt_prefilter_stride = (int)value; return BHIP_OK;
	}
	if (!strcmp(name, "two_stage")) { h->opt_two_stage = value != 0; return BHIP_OK; }
	if (!strcmp(name, "host_routing")) { h->opt_host_routing = value != 0; return BHIP_OK; }
	if (!strcmp(name, "discard_staged")) { for (StageSlot &S : h->slots) if (S.state == 1) S.state = 0; h->res_valid = false; return BHIP_OK; }
	if (!strcmp(name, "lane_masks")) { h->opt_lane_masks = value != 0; return BHIP_OK; }
	if (!strcmp(name, "sweep_blocks")) { if (value < 1 || value > 8) return fail(BHIP_E_ARG, "sweep_blocks must be 1 .. 8"); h->opt_sweep_blocks = (int)value; return BHIP_OK; }
	if (!strcmp(name, "lane_min_entries")) { if (value < 1) return fail(BHIP_E_ARG, "lane_min_entries must be >= 1"); h->opt_lane_min = (int)value; return BHIP_OK; }
	if (!strcmp(name, "async_d2h")) { h->opt_async_d2h = value != 0; return BHIP_OK; }
	if (!strcmp(name, "seed_ahead")) { h->opt_seed_ahead = value != 0; return BHIP_OK; }
	if (!strcmp(name, "peq_ahead_blocks")) { if (value < 1 || value > 16) return fail(BHIP_E_ARG, "peq_ahead_blocks must be 1 .. 16"); h->opt_peq_ahead_blocks = (int)value; return BHIP_OK; }
	if (!strcmp(name, "seed_ahead_blocks")) { if (value < 0 || value > 64) return fail(BHIP_E_ARG, "seed_ahead_blocks must be 0 .. 64"); h->opt_seed_ahead_blocks = (int)value; return BHIP_OK; }
	if (!strcmp(name, "prune")) { h->opt_prune = value != 0; return BHIP_OK; }
	if (!strcmp(name, "rescore_reg")) { h->opt_rescore_reg = value != 0; return BHIP_OK; }
	if (!strcmp(name, "prefilter_waves")) { if (value < 0 || value > 16) return fail(BHIP_E_ARG, "prefilter_waves must be 0 .. 16"); h->opt_pf_waves = (int)value; return BHIP_OK; }
	if (!strcmp(name, "prefilter_algo")) { if (value < -1 || value > 1) return fail(BHIP_E_ARG, "prefilter_algo must be -1, 0 or 1"); h->opt_pf_algo = (int)value; return BHIP_OK; }
	if (!strcmp(name, "prefilter_table")) { if (value != 0 && (value < 9 || value > 11)) return fail(BHIP_E_ARG, "prefilter_table must be 0, 9, 10 or 11"); h->opt_pf_table = (int)value; return BHIP_OK; }
	if (!strcmp(name, "lanes")) {
		if (value < 1 || value > 16) return fail(BHIP_E_ARG, "lanes must be 1 .. 16");
		h->opt_lanes = (int)value; for (StageSlot &S : h->slots) { S.state = 0; S.st_valid = false; } return BHIP_OK;
	}
	return fail(BHIP_E_ARG, "unknown option '%s'", name);
}

extern "C" int bhip_get_stats(void *handle, BhipStats *out) {
	Handle *h = (Handle *)handle;
	if (!h || !out) return fail(BHIP_E_ARG, "null argument");
	*out = h->stats;
	return BHIP_OK;
}

// ---- kernel launch helpers (st = stream to launch on) ---------------------------------------------------------------
static void launch_myers(Handle *h, Lane *L, hipStream_t st, int cls, uint32_t grid, const uint2 *pairs, const uint32_t *n_pairs_dev,
		uint64_t n_pairs_host, uint32_t li_base, const uint32_t *qlist, BhipRawHit *raw, uint32_t *n_raw, uint32_t raw_cap, uint32_t *best,
		uint8_t *mins, Counters *dc) {
	#define LM(N) hipLaunchKernelGGL(k_myers<N>, dim3(grid), dim3(256), 0, st, pairs, n_pairs_dev, n_pairs_host, h->n_clumps, li_base, qlist, \
		L->peq.as<uint32_t>(), h->s_off(), h->s_emac(), (best && h->cur->st_has_six) ? h->cur->qsix.as<uint32_t>() : nullptr, \
		h->ref.as<uint4>(), h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), h->tot_refs, raw, n_raw, raw_cap, best, mins, &dc->col_sum, &dc->qlen_sum)
	switch (kClasses[cls]) { case 2: LM(2); break; case 4: LM(4); break; case 6: LM(6); break; case 8: LM(8); break; case 10: LM(10); break;
		case 16: LM(16); break; default: LM(32); break; }
	#undef LM
}
static void launch_prefix(Handle *h, Lane *L, hipStream_t st, int NWP, uint32_t grid, const uint2 *pairs, const uint32_t *n_pairs_dev,
		uint64_t n_pairs_host, uint32_t li_base, const uint32_t *qlist, uint32_t *n_wins, Counters *dc) {
	#define LP(N) hipLaunchKernelGGL(k_myers_prefix<N>, dim3(grid), dim3(256), 0, st, pairs, n_pairs_dev, n_pairs_host, h->n_clumps, li_base, qlist, \
		L->peqp.as<uint32_t>(), h->s_off(), h->s_emac(), h->ref.as<uint4>(), h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), \
		h->tot_refs, L->wins.as<BhipWin>(), n_wins, (uint32_t)L->win_cap, &dc->col_sum, &dc->qlen_sum)
	if (NWP == 1) LP(1); else if (NWP == 2) LP(2); else if (NWP == 3) LP(3); else if (NWP == 4) LP(4); else LP(6);
	#undef LP
}
static void launch_prefix_task(Handle *h, Lane *L, hipStream_t st, int NWP, uint32_t grid, const uint2 *tasks, const uint32_t *n_tasks_dev, const uint32_t *qlist,
		BhipWin *wins, uint32_t *n_wins, Counters *dc) {
	#define LT(N) hipLaunchKernelGGL(k_myers_prefix_task<N>, dim3(grid), dim3(64), 0, st, tasks, n_tasks_dev, (uint32_t)L->task_cap, qlist, \
		L->peqp.as<uint32_t>(), h->s_off(), h->s_emac(), h->ref_lane.as<uint4>(), h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), \
		wins, n_wins, (uint32_t)L->win_cap, &dc->tcol_sum)
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

static void launch_window(Handle *h, Lane *L, hipStream_t wst, int cls, int NWP, uint32_t grid_cap, const uint32_t *qlist, const BhipWin *wins, const uint32_t *n_wins, Counters *dc) {
	#define LW(N) { const uint32_t thr = (N) <= 8 ? 64u : 256u;      /* NW <= 8: per-thread A/C/G/T profile rows in LDS, 64-thread blocks */ \
		const uint32_t grid = std::min<uint32_t>(grid_cap * (256u / thr), (uint32_t)h->n_cu * blocks_per_cu((const void *)k_myers_window<N>, thr, 0)); \
		hipLaunchKernelGGL(k_myers_window<N>, dim3(grid), dim3(thr), 0, wst, wins, n_wins, (uint32_t)L->win_cap, NWP, qlist, \
		L->peq.as<uint32_t>(), h->s_off(), h->s_emac(), h->cur->st_has_six ? h->cur->qsix.as<uint32_t>() : nullptr, h->ref_lane.as<uint4>(), \
		h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), L->raw.as<BhipRawHit>(), &dc->n_raw, (uint32_t)L->raw_cap, h->best.as<uint32_t>(), &dc->wcol_sum); }
	switch (kClasses[cls]) { case 2: LW(2); break; case 4: LW(4); break; case 6: LW(6); break; case 8: LW(8); break; case 10: LW(10); break;
		case 16: LW(16); break; default: LW(32); break; }
	#undef LW
}

// Seed plan of one query entry (see bhip_kernels.hip): returns stride | need << 8, need = 0 when no stride guarantees a
// surviving word (the caller then aligns the entry against every clump).  stride_opt > 0 forces the stride.
static uint32_t make_seed_plan(const uint8_t *s, uint32_t len, uint32_t E, uint32_t K, int stride_opt) {
	if (len < K) return 1u;
	const uint32_t npos = len - K + 1;
	bool clean = true;
	for (uint32_t i = 0; i < len; ++i) if ((uint32_t)(s[i] - 1u) >= 4u) { clean = false; break; }
	std::vector<uint8_t> valid;
	if (!clean) {           // valid[p] = word starting at p contains only A/C/G/T
		valid.assign(npos, 0);
		uint32_t run = 0;
		for (uint32_t i = 0; i < len; ++i) { run = ((uint32_t)(s[i] - 1u) < 4u) ? run + 1 : 0; if (i + 1 >= K && run >= K) valid[i + 1 - K] = 1; }
	}
	auto need_of = [&](uint32_t st) -> int {
		uint32_t W = 0;
		if (clean) W = (len - K) / st + 1; else for (uint32_t p = 0; p < npos; p += st) W += valid[p];
		return (int)W - (int)(E * ((K + st - 1) / st));
	};
	const uint32_t smin = (len - K) / 254 + 1;      // keeps the number of sampled words <= 255 (8-bit counters)
	uint32_t best_s = 0; int best_n = 0;
	if (stride_opt > 0) { best_s = std::max<uint32_t>((uint32_t)stride_opt, smin); best_n = need_of(best_s); }
	else {
		for (uint32_t st = std::max(K, smin); st >= smin; --st) { const int n = need_of(st); if (n >= 3) { best_s = st; best_n = n; break; } if (st == smin) break; }
		if (!best_s) for (uint32_t st = smin; st <= std::max(K, smin); ++st) { const int n = need_of(st); if (n > best_n) { best_n = n; best_s = st; } }
		if (!best_s) { best_s = smin; best_n = need_of(smin); }
	}
	if (best_n < 1) best_n = 0;
	if (best_n > 0xFFFF) best_n = 0xFFFF;
	return (best_s & 255u) | ((uint32_t)best_n << 8);
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
	if (!h->opt_two_stage) return 0;
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
		hipLaunchKernelGGL(k_build_peq, dim3(grid), dim3(256), 0, st, codes, off, d_qlist, n_list, NW, 0, h->mm, peq.as<uint32_t>(), pack, (S->st_maxlen + 7) / 8);
		HIPCHK(hipGetLastError());
	}
	if (NWP) {
		const uint32_t qb = 256u / (uint32_t)NWP;
		const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)n_list + qb - 1) / qb, (uint64_t)h->n_cu * blocks_per_cu);
		hipLaunchKernelGGL(k_build_peq, dim3(grid), dim3(256), 0, st, codes, off, d_qlist, n_list, NWP, 32 * NWP, h->mm, peqp.as<uint32_t>(), pack, (S->st_maxlen + 7) / 8);
		HIPCHK(hipGetLastError());
	}
	return 0;
}
// k_seed_ranges for one (lane, class) list of staged batch S into the lane's per-class buffers
static int launch_seed(Handle *h, Lane *L, hipStream_t st, StageSlot *S, int cls, const uint32_t *d_qlist, uint32_t n_list, uint32_t maxwords, bool ahead = false) {
	int rc;
	const uint32_t W16 = seed_row_words(maxwords);
	L->seeded_ok[cls] = false;
	if ((rc = L->ranges_c[cls].reserve((size_t)n_list * W16 * 8 + 16))) return rc;
	if ((rc = L->hdr_c[cls].reserve((size_t)n_list * 8 + 16))) return rc;
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
		junk ? S->qpack_s.as<uint32_t>() : S->qpack.as<uint32_t>(), (S->st_maxlen + 7) / 8, junk ? S->qemac_s.as<uint16_t>() : S->qemac.as<uint16_t>());
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(ev[1], st));
	L->seeded_ok[cls] = true; L->seeded_seq[cls] = S->seq; L->seeded_n[cls] = n_list; L->seeded_W16[cls] = W16;
	return 0;
}

// dense clump-level kernels into L->cand as (list position, clump) pairs
static int launch_prefilter_mask(Handle *h, Lane *L, hipStream_t st, int cls, const uint32_t *d_qlist, uint32_t n_list, uint32_t maxwords, uint32_t *n_tasks_dev,
                                 uint32_t *n_cand_dev, Counters *dc, int prune) {
	int rc;
	if ((rc = L->fb_list.reserve((size_t)n_list * 4 + 16))) return rc;
	if (L->pf_launches) HIPCHK(hipMemsetAsync(&dc->n_fb, 0, 4, st));      // (the lane's first class finds the whole counter block zeroed by enqueue_lane)
	const uint32_t W16 = seed_row_words(maxwords);
	// the lookups of this batch may have run ahead (seed_next_batch, during the previous call)
	if (!(L->seeded_ok[cls] && L->seeded_seq[cls] == h->cur->seq && L->seeded_n[cls] == n_list && L->seeded_W16[cls] == W16))
		if ((rc = launch_seed(h, L, st, h->cur, cls, d_qlist, n_list, maxwords))) return rc;
	const int algo = h->opt_pf_algo >= 0 ? h->opt_pf_algo : L->pf_algo;
	const uint32_t n_quads = (n_list + 3) / 4;
	// hash table size per query from the expected number of distinct clumps (sampled words x occurrence-weighted mean list
	// length): 512 slots keep 12 single-wave blocks on a CU, 1024 -> 7, 2048 -> 4
	const double expect = (n_list ? (double)L->seed_words[cls] / (double)n_list : (double)maxwords) * h->acx_wmean;   // mean, not max: outliers use the fallback
	// (the touched list holds half the slots; a query that exceeds it is re-done by the dense fallback, so the estimate -- an
	// upper bound, every repeated clump counted once per word -- may be cut close)
	// (counting filter: the approximate counters tolerate a load around 1 -- false survivors only cost work)
	const int htb = h->opt_pf_table ? h->opt_pf_table : algo == 0 ? (expect <= 600.0 ? 9 : expect <= 1200.0 ? 10 : 11) : (expect <= 230.0 ? 9 : expect <= 470.0 ? 10 : 11);
	// resident single-wave blocks per CU from the kernel's static LDS / register use (measured on gfx950: 11 blocks of 13 144 B
	// fit a CU and 12 do not, 10 of 14 168 B fit and 11 do not: about 148 KB of the 160 KB are available to them; 512 VGPRs per SIMD lane in steps of 8).  The kernel is a persistent loop over a static
	// partition of the list: one block too many per CU would run after the others and double the time.
	hipFuncAttributes fa;
	memset(&fa, 0, sizeof fa);
	{
		const void *fp = algo == 0
			? (htb == 9 ? (const void *)k_prefilter_cf<9> : htb == 10 ? (const void *)k_prefilter_cf<10> : (const void *)k_prefilter_cf<11>)
			: (htb == 9 ? (const void *)k_prefilter_mask<9> : htb == 10 ? (const void *)k_prefilter_mask<10> : (const void *)k_prefilter_mask<11>);
		if (hipFuncGetAttributes(&fa, fp) != hipSuccess) { fa.sharedSizeBytes = 48 * 1024; fa.numRegs = 128; }
	}
	const uint32_t by_lds = (148u * 1024u) / (uint32_t)std::max<size_t>(512, (fa.sharedSizeBytes + 511) & ~(size_t)511);
	const uint32_t by_reg = 4u * (512u / (uint32_t)std::max(8, (fa.numRegs + 7) & ~7));
	const uint32_t fit = std::max<uint32_t>(1u, std::min<uint32_t>(12u, std::min(by_lds, by_reg)));
	if (getenv("BHIP_DEBUG")) fprintf(stderr, "[bhip] prefilter kernel: table 2^%d, %zu B LDS, %d VGPRs -> %u blocks per CU\n", htb, fa.sharedSizeBytes, fa.numRegs, fit);
	const uint32_t waves = h->opt_pf_waves ? std::min<uint32_t>((uint32_t)h->opt_pf_waves, fit) : fit;
	const uint32_t grid = std::min<uint32_t>(n_quads, (uint32_t)h->n_cu * waves);
	HIPCHK(hipEventRecord(L->ev_pf[cls][1], st));
	if (algo == 0) {
#define PFC_LAUNCH(B) hipLaunchKernelGGL(k_prefilter_cf<B>, dim3(grid), dim3(64), 0, st, L->ranges_c[cls].as<uint2>(), L->hdr_c[cls].as<uint2>(), W16, n_list, \
		h->acx_view().rec, h->bad.as<uint32_t>(), h->n_bad, \
		h->clump_len.as<uint32_t>(), h->tot_refs, L->tasks.as<uint2>(), n_tasks_dev, (uint32_t)L->task_cap, &dc->ent_read, \
		L->fb_list.as<uint32_t>(), &dc->n_fb, &dc->unit_sum, &dc->col_sum, &dc->qlen_sum, &dc->surv_sum, \
		L->tasks2.as<uint2>(), &dc->n_tasks2_cls[cls], prune)
		if (htb == 9) PFC_LAUNCH(9); else if (htb == 10) PFC_LAUNCH(10); else PFC_LAUNCH(11);
#undef PFC_LAUNCH
	} else {
#define PFM_LAUNCH(B) hipLaunchKernelGGL(k_prefilter_mask<B>, dim3(grid), dim3(64), 0, st, L->ranges_c[cls].as<uint2>(), L->hdr_c[cls].as<uint2>(), W16, n_list, \
		h->acx_view().rec, h->bad.as<uint32_t>(), h->n_bad, \
		h->clump_len.as<uint32_t>(), h->tot_refs, L->tasks.as<uint2>(), n_tasks_dev, (uint32_t)L->task_cap, &dc->ent_read, \
		L->fb_list.as<uint32_t>(), &dc->n_fb, &dc->unit_sum, &dc->col_sum, &dc->qlen_sum, L->cand.as<uint2>(), n_cand_dev, (uint32_t)L->cand_cap)
		if (htb == 9) PFM_LAUNCH(9); else if (htb == 10) PFM_LAUNCH(10); else PFM_LAUNCH(11);
#undef PFM_LAUNCH
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(L->ev_pf[cls][2], st));
	++L->pf_launches;
	L->pf_algo_used = algo;
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
			(uint32_t *)nullptr, n_cand_dev, (uint32_t)L->cand_cap, &dc->ent_read, h->cur->plan.as<uint32_t>(), L->fb_list.as<uint32_t>(), &dc->n_fb);
		else hipLaunchKernelGGL(k_prefilter_wave<uint16_t>, dim3(g2), dim3(64), lds_w, st, h->s_codes(), h->s_off(),
			h->s_emac(), d_qlist, n_list, h->acx_view(), h->K, h->n_clumps, bad, h->n_bad, L->cand.as<uint2>(),
			(uint32_t *)nullptr, n_cand_dev, (uint32_t)L->cand_cap, &dc->ent_read, h->cur->plan.as<uint32_t>(), L->fb_list.as<uint32_t>(), &dc->n_fb);
	} else {
		uint32_t g2 = std::min<uint32_t>(n_list, (uint32_t)h->n_cu * 2);
		if ((rc = L->gcnt.reserve((size_t)g2 * nw32 * 4))) return rc;
		hipLaunchKernelGGL(k_prefilter<false>, dim3(g2), dim3(256), 0, st, h->s_codes(), h->s_off(),
			h->s_emac(), d_qlist, n_list, h->acx_view(), h->K, h->n_clumps,
			L->gcnt.as<uint32_t>(), bad, h->n_bad, L->cand.as<uint2>(), (uint32_t *)nullptr, n_cand_dev, (uint32_t)L->cand_cap, &dc->ent_read,
			L->fb_list.as<uint32_t>(), &dc->n_fb, h->cur->plan.as<uint32_t>());
	}
	HIPCHK(hipGetLastError());
	return 0;
}

static int ensure_lanes(Handle *h, uint32_t n) {
	while (h->lanes.size() < n) { Lane *L = nullptr; int rc = lane_create(h, &L); if (rc) return rc; h->lanes.push_back(L); }
	return 0;
}

// ---- staged batches -------------------------------------------------------------------------------------------------
// A batch is staged into one of two slots.  Everything is enqueued on stage_stream and nothing is waited for: the copies
// of the caller's arrays, k_span_fill (batch offsets, shared slots, reported query numbers), k_pack_queries, k_route
// (class / lane / route / seed plan per entry, per-list counts) and a stable radix sort of the entry numbers by list key.
// The batch that is being aligned meanwhile uses the other slot and other streams.  resolve_slot() -- called when the
// batch is about to be aligned -- waits for the slot's event, reads the routing summary from pinned memory and, for the
// rare batch with query symbols of code 0, runs the host pass that builds the search view.
static int slot_init(StageSlot *S) {
	if (S->ev_done) return 0;
	if (hipEventCreate(&S->ev_begin) != hipSuccess || hipEventCreate(&S->ev_done) != hipSuccess) return fail(BHIP_E_DEVICE, "hipEventCreate failed");
	if (hipHostMalloc((void **)&S->info_pinned, sizeof(BhipStageInfo), hipHostMallocDefault) != hipSuccess) return fail(BHIP_E_DEVICE, "hipHostMalloc failed");
	int rc = S->info.reserve(sizeof(BhipStageInfo));
	return rc;
}

static uint32_t lanes_for(const Handle *h, uint32_t n_q) {
	uint32_t nl = (uint32_t)h->opt_lanes;
	while (nl > 1 && n_q / nl < (uint32_t)h->opt_lane_min) --nl;
	return nl;
}

static int stage_enqueue(Handle *h, StageSlot *S, const BhipQuerySpan *spans, uint32_t n_spans, const uint32_t *six_explicit, bool share_by_position,
                         uint32_t n_shared, uint32_t max_len) {
	int rc;
	if ((rc = slot_init(S))) return rc;
	hipStream_t st = h->stage_stream;
	uint64_t n_q64 = 0, nb = 0;
	bool any_rc = false, any_flags = false, all_flags = true, any_qbase = false, all_packed = true;
	for (uint32_t k = 0; k < n_spans; ++k) {
		const BhipQuerySpan &sp = spans[k];
		if (!sp.n) continue;
		if (!sp.codes || !sp.off || !sp.emac) return fail(BHIP_E_ARG, "null query arrays");
		if (share_by_position && sp.n > n_shared) return fail(BHIP_E_ARG, "span %u has %u entries for %u shared slots", k, sp.n, n_shared);
		n_q64 += sp.n; nb += sp.off[sp.n] - sp.off[0];
		all_packed &= sp.codes4 != nullptr;
		any_rc |= sp.rc != nullptr; any_flags |= sp.flags != nullptr; all_flags &= sp.flags != nullptr; any_qbase |= sp.q_base != 0 || k > 0;
	}
	if (n_q64 > 0xFFFFFFF0ull) return fail(BHIP_E_ARG, "too many entries in one batch");
	if (any_flags && !all_flags) return fail(BHIP_E_ARG, "q_flags given for some spans only");
	const uint32_t n_q = (uint32_t)n_q64;
	S->spans.assign(spans, spans + n_spans);
	S->six_explicit = six_explicit;
	S->resolved = false; S->st_valid = false; S->st_has_junk = false;
	S->st_nq = n_q; S->st_has_six = six_explicit != nullptr || share_by_position; S->st_has_rc = any_rc; S->has_flags = any_flags; S->has_qmap = any_qbase;
	S->st_nshared = S->st_has_six ? n_shared : n_q;
	S->st_lanes = n_q ? lanes_for(h, n_q) : 0;
	S->seq = ++h->stage_seq;
	if (!n_q) { S->st_maxlen = 0; return 0; }
	if ((rc = ensure_lanes(h, S->st_lanes))) return rc;
	if (!max_len) {         // longest entry: from the offsets (host pass over 8 bytes per entry)
		for (uint32_t k = 0; k < n_spans; ++k) for (uint32_t j = 0; j < spans[k].n; ++j) {
			const uint64_t len = spans[k].off[j + 1] - spans[k].off[j];
			if (len > BHIP_MAX_QLEN) return fail(BHIP_E_QUERYLEN, "query %u has %llu symbols (max %d)", j, (unsigned long long)len, BHIP_MAX_QLEN);
			max_len = std::max<uint32_t>(max_len, (uint32_t)len);
		}
	}
	if (max_len > BHIP_MAX_QLEN) return fail(BHIP_E_QUERYLEN, "queries of up to %u symbols (max %d)", max_len, BHIP_MAX_QLEN);
	S->st_maxlen = max_len;
	const uint32_t qw = (max_len + 7) / 8;
	if (all_packed && (rc = S->qcodes4.reserve(nb / 2 + 2 * (size_t)n_spans + 64))) return rc;
	if ((rc = S->qcodes.reserve(nb + 2 * (size_t)n_spans + 64)) || (rc = S->qoff.reserve(((size_t)n_q + 1) * 8)) || (rc = S->qemac.reserve(((size_t)n_q + 1) * 2)) ||
	    (rc = S->qsix.reserve(((size_t)n_q + 1) * 4)) || (rc = S->qrc.reserve((size_t)n_q + 1)) || (rc = S->qflags.reserve((size_t)n_q + 1)) ||
	    (rc = S->qmap.reserve(((size_t)n_q + 1) * 4)) || (rc = S->off_raw.reserve(((size_t)n_q + n_spans + 1) * 8)) || (rc = S->plan.reserve((size_t)n_q * 4 + 16)) ||
	    (rc = S->qpack.reserve((size_t)n_q * qw * 4 + 64)) || (rc = S->key.reserve((size_t)n_q + 16)) || (rc = S->key_sorted.reserve((size_t)n_q + 16)) ||
	    (rc = S->idx.reserve((size_t)n_q * 4 + 16)) || (rc = S->idx_sorted.reserve((size_t)n_q * 4 + 16))) return rc;
	HIPCHK(hipEventRecord(S->ev_begin, st));
	uint32_t ebase = 0; uint64_t pos = 0, pos4 = 0;      // pos: symbols of the batch so far; pos4: nibbles of the packed staging area
	for (uint32_t k = 0; k < n_spans; ++k) {
		const BhipQuerySpan &sp = spans[k];
		if (!sp.n) continue;
		const uint64_t bytes = sp.off[sp.n] - sp.off[0];
		if (all_packed) {      // in the staging area the span starts on a byte of its own, at the nibble parity it has in the caller's array
			pos4 = ((pos4 + 1) & ~1ull) + (sp.off[0] & 1ull);
			if (bytes) {
				HIPCHK(hipMemcpyAsync(S->qcodes4.as<uint8_t>() + (pos4 >> 1), sp.codes4 + (sp.off[0] >> 1), ((sp.off[sp.n] + 1) >> 1) - (sp.off[0] >> 1), hipMemcpyHostToDevice, st));
				hipLaunchKernelGGL(k_unpack4, dim3((uint32_t)std::min<uint64_t>((bytes / 4 + 256) / 256, (uint64_t)h->n_cu * 16)), dim3(256), 0, st, S->qcodes4.as<uint8_t>(), pos4, bytes,
					S->qcodes.as<uint8_t>() + pos);
				HIPCHK(hipGetLastError());
			}
			pos4 += bytes;
		} else if (bytes) HIPCHK(hipMemcpyAsync(S->qcodes.as<uint8_t>() + pos, sp.codes + sp.off[0], bytes, hipMemcpyHostToDevice, st));
		uint64_t *raw = S->off_raw.as<uint64_t>() + ebase + k;
		HIPCHK(hipMemcpyAsync(raw, sp.off, ((size_t)sp.n + 1) * 8, hipMemcpyHostToDevice, st));
		HIPCHK(hipMemcpyAsync(S->qemac.as<uint16_t>() + ebase, sp.emac, (size_t)sp.n * 2, hipMemcpyHostToDevice, st));
		if (sp.rc) HIPCHK(hipMemcpyAsync(S->qrc.as<uint8_t>() + ebase, sp.rc, sp.n, hipMemcpyHostToDevice, st));
		else if (any_rc) HIPCHK(hipMemsetAsync(S->qrc.as<uint8_t>() + ebase, 0, sp.n, st));
		if (sp.flags) HIPCHK(hipMemcpyAsync(S->qflags.as<uint8_t>() + ebase, sp.flags, sp.n, hipMemcpyHostToDevice, st));
		if (six_explicit) HIPCHK(hipMemcpyAsync(S->qsix.as<uint32_t>() + ebase, six_explicit + ebase, (size_t)sp.n * 4, hipMemcpyHostToDevice, st));
		hipLaunchKernelGGL(k_span_fill, dim3(std::min<uint32_t>((sp.n + 256) / 256, (uint32_t)h->n_cu * 4)), dim3(256), 0, st, raw, sp.n, ebase, pos, sp.q_base,
			S->qoff.as<uint64_t>(), share_by_position ? S->qsix.as<uint32_t>() : (uint32_t *)nullptr, S->qmap.as<uint32_t>());
		HIPCHK(hipGetLastError());
		ebase += sp.n; pos += bytes;
	}
	{	// 4-bit packed copy of the queries at a fixed stride (layout used by the routing, seed, profile and re-scoring kernels)
		const uint64_t total = (uint64_t)n_q * qw;
		if (total) hipLaunchKernelGGL(k_pack_queries, dim3((uint32_t)std::min<uint64_t>((total + 255) / 256, (uint64_t)h->n_cu * 16)), dim3(256), 0, st,
			S->qcodes.as<uint8_t>(), S->qoff.as<uint64_t>(), n_q, qw, S->qpack.as<uint32_t>());
		HIPCHK(hipGetLastError());
	}
	HIPCHK(hipMemsetAsync(S->info.p, 0, sizeof(BhipStageInfo), st));
	hipLaunchKernelGGL(k_route, dim3(std::min<uint32_t>((n_q + 255) / 256, (uint32_t)h->n_cu * 8)), dim3(256), 0, st, S->qoff.as<uint64_t>(), S->qpack.as<uint32_t>(), qw,
		S->qemac.as<uint16_t>(), S->st_has_six ? S->qsix.as<uint32_t>() : (const uint32_t *)nullptr, any_flags ? S->qflags.as<uint8_t>() : (const uint8_t *)nullptr,
		n_q, S->st_nshared, S->st_lanes, h->has_acx ? 1 : 0, h->K, h->opt_prefilter_stride, S->plan.as<uint32_t>(), S->key.as<uint8_t>(), S->idx.as<uint32_t>(),
		S->info.as<BhipStageInfo>());
	HIPCHK(hipGetLastError());
	{
		size_t tb = 0;
		HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, S->key.as<uint8_t>(), S->key_sorted.as<uint8_t>(), S->idx.as<uint32_t>(), S->idx_sorted.as<uint32_t>(), (int)n_q, 0, 8, st));
		if ((rc = S->sort_tmp.reserve(tb + 16))) return rc;
		HIPCHK(hipcub::DeviceRadixSort::SortPairs(S->sort_tmp.p, tb, S->key.as<uint8_t>(), S->key_sorted.as<uint8_t>(), S->idx.as<uint32_t>(), S->idx_sorted.as<uint32_t>(), (int)n_q, 0, 8, st));
	}
	HIPCHK(hipMemcpyAsync(S->info_pinned, S->info.p, sizeof(BhipStageInfo), hipMemcpyDeviceToHost, st));
	HIPCHK(hipEventRecord(S->ev_done, st));
	return 0;
}

// lists and maxima of a slot from a routing summary
static void slot_take_info(StageSlot *S, const BhipStageInfo &I) {
	uint32_t off = 0;
	for (uint32_t l = 0; l < 16; ++l) {
		for (int c = 0; c < kNumClasses; ++c) {
			const uint32_t lc = l * 7 + (uint32_t)c;
			S->npf[l][c] = I.count[lc * 2]; S->nex[l][c] = I.count[lc * 2 + 1];
			S->qlist_off[l][c] = off; off += S->npf[l][c] + S->nex[l][c];
			S->maxE[l][c] = I.maxE[lc]; S->maxwords[l][c] = I.maxwords[lc]; S->seed_words[l][c] = I.seed_words[lc];
		}
		S->maxlen_lane[l] = I.maxlen_lane[l]; S->n_entries_lane[l] = I.n_entries_lane[l];
	}
	S->st_maxE = I.maxE_all;
}

// Host pass: routing, seed plans and -- when the batch holds symbols of code 0 -- the search view without them, from the
// caller's arrays (the device copies of the batch itself are already in place).
static int host_route(Handle *h, StageSlot *S) {
	const uint32_t n_q = S->st_nq, nl = S->st_lanes, nsh = S->st_nshared;
	int rc;
	// flat host view of the batch
	std::vector<uint8_t> f_codes, f_rc, f_flags; std::vector<uint64_t> f_off; std::vector<uint16_t> f_emac; std::vector<uint32_t> f_six;
	const uint8_t *q_codes; const uint64_t *q_off; const uint16_t *q_emac; const uint32_t *q_six = S->six_explicit; const uint8_t *q_flags = nullptr;
	uint32_t n_live = 0, first = 0;
	for (uint32_t k = 0; k < S->spans.size(); ++k) if (S->spans[k].n) { if (!n_live) first = k; ++n_live; }
	const bool share_by_position = S->st_has_six && !S->six_explicit;
	if (n_live == 1 && S->spans[first].off[0] == 0 && !share_by_position) {
		q_codes = S->spans[first].codes; q_off = S->spans[first].off; q_emac = S->spans[first].emac; q_flags = S->spans[first].flags;
	} else {
		f_off.assign(1, 0);
		for (const BhipQuerySpan &sp : S->spans) {
			if (!sp.n) continue;
			f_codes.insert(f_codes.end(), sp.codes + sp.off[0], sp.codes + sp.off[sp.n]);
			const uint64_t base = f_off.back() - sp.off[0];
			for (uint32_t j = 0; j < sp.n; ++j) { f_off.push_back(sp.off[j + 1] + base); if (share_by_position) f_six.push_back(j); }
			f_emac.insert(f_emac.end(), sp.emac, sp.emac + sp.n);
			if (S->has_flags) f_flags.insert(f_flags.end(), sp.flags, sp.flags + sp.n);
		}
		f_codes.resize(f_codes.size() + 16, 0);
		q_codes = f_codes.data(); q_off = f_off.data(); q_emac = f_emac.data();
		if (share_by_position) q_six = f_six.data();
		if (S->has_flags) q_flags = f_flags.data();
	}
	const size_t n_keys = (size_t)nl * kNumClasses * 2;
	std::vector<uint32_t> plan(n_q, 1u);
	std::vector<uint8_t> nxv(n_q, 0);          // symbols of code 0 per entry (255 = more than any budget: never searched)
	struct Part {
		std::vector<std::vector<uint32_t>> lists;
		std::vector<uint32_t> maxE, maxwords, maxlen, n_entries;
		std::vector<uint64_t> seed_words;
		int err = 0; uint32_t err_i = 0; uint64_t err_len = 0;
		bool junk = false;
	};
	uint32_t n_thr = n_q < 65536 ? 1u : std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
	if (const char *ev = getenv("BHIP_STAGE_THREADS")) { const int v = atoi(ev); if (v > 0 && n_q >= 65536) n_thr = (uint32_t)std::min(v, 64); }      // tuning hook
	std::vector<Part> parts(n_thr);
	auto work = [&](uint32_t t) {
		Part &P = parts[t];
		P.lists.resize(n_keys); P.maxE.assign((size_t)nl * kNumClasses, 0); P.maxwords.assign((size_t)nl * kNumClasses, 0);
		P.seed_words.assign((size_t)nl * kNumClasses, 0); P.maxlen.assign(nl, 0); P.n_entries.assign(nl, 0);
		const uint32_t i0 = (uint32_t)((uint64_t)n_q * t / n_thr), i1 = (uint32_t)((uint64_t)n_q * (t + 1) / n_thr);
		std::vector<uint8_t> clean;
		for (uint32_t i = i0; i < i1; ++i) {
			const uint64_t len_all = q_off[i + 1] - q_off[i];
			if (len_all == 0) continue;
			if (len_all > BHIP_MAX_QLEN) { P.err = 1; P.err_i = i; P.err_len = len_all; return; }
			if (q_six && q_six[i] >= nsh) { P.err = 2; P.err_i = i; return; }
			// search view of the entry: symbols of code 0 removed, budget reduced by their number
			const uint8_t *codes_i = q_codes + q_off[i];
			uint64_t len = len_all;
			uint32_t E_i = q_emac[i];
			if (memchr(codes_i, 0, len_all)) {
				clean.clear();
				for (uint64_t k = 0; k < len_all; ++k) if (codes_i[k]) clean.push_back(codes_i[k]);
				const uint64_t nx_i = len_all - clean.size();
				P.junk = true;
				nxv[i] = (uint8_t)std::min<uint64_t>(nx_i, 255);
				if (nx_i > E_i || clean.empty()) { nxv[i] = 255; continue; }      // cannot be aligned within its budget
				E_i -= (uint32_t)nx_i; len = clean.size(); codes_i = clean.data();
			}
			const uint32_t six = q_six ? q_six[i] : i;
			const uint32_t l = (uint32_t)(((uint64_t)six * nl) / nsh);
			const int cls = class_of_len((uint32_t)len);
			int ex = q_flags ? (q_flags[i] == BHIP_Q_EXHAUSTIVE) : !h->has_acx;
			if (!h->has_acx) ex = 1;
			if (!ex) {
				plan[i] = make_seed_plan(codes_i, (uint32_t)len, E_i, (uint32_t)h->K, h->opt_prefilter_stride);
				if ((plan[i] >> 8) == 0) ex = 1;           // no word is guaranteed to survive: exhaustive (burst.c:3130-3131 does the same for "bad" queries)
			}
			const size_t lc = (size_t)l * kNumClasses + cls;
			P.lists[lc * 2 + ex].push_back(i);
			P.maxE[lc] = std::max<uint32_t>(P.maxE[lc], E_i);
			if (!ex && len >= (uint64_t)h->K) {
				const uint32_t nwd = (uint32_t)((len - h->K) / (plan[i] & 255u) + 1);
				P.maxwords[lc] = std::max<uint32_t>(P.maxwords[lc], nwd);
				P.seed_words[lc] += nwd;
			}
			P.maxlen[l] = std::max<uint32_t>(P.maxlen[l], (uint32_t)len_all);
			++P.n_entries[l];
		}
	};
	if (n_thr == 1) work(0);
	else {
		std::vector<std::thread> th;
		for (uint32_t t = 0; t < n_thr; ++t) th.emplace_back(work, t);
		for (auto &x : th) x.join();
	}
	for (const Part &P : parts) {
		if (P.err == 1) return fail(BHIP_E_QUERYLEN, "query %u has %llu symbols (max %d)", P.err_i, (unsigned long long)P.err_len, BHIP_MAX_QLEN);
		if (P.err == 2) return fail(BHIP_E_ARG, "q_six[%u] out of range", P.err_i);
	}
	// summary + sorted entry numbers in key order (thread order = entry order: the lists come out as a single pass would build them)
	BhipStageInfo I;
	memset(&I, 0, sizeof I);
	std::vector<uint32_t> sorted; sorted.reserve(n_q);
	for (uint32_t l = 0; l < nl; ++l) for (int c = 0; c < kNumClasses; ++c) for (int ex = 0; ex < 2; ++ex) {
		const size_t lc = (size_t)l * kNumClasses + c, k = lc * 2 + ex;
		for (const Part &P : parts) { sorted.insert(sorted.end(), P.lists[k].begin(), P.lists[k].end()); I.count[(l * 7 + c) * 2 + ex] += (uint32_t)P.lists[k].size(); }
		if (!ex) for (const Part &P : parts) {
			I.maxE[l * 7 + c] = std::max(I.maxE[l * 7 + c], P.maxE[lc]); I.maxwords[l * 7 + c] = std::max(I.maxwords[l * 7 + c], P.maxwords[lc]);
			I.seed_words[l * 7 + c] += P.seed_words[lc];
		}
	}
	for (uint32_t l = 0; l < nl; ++l) for (const Part &P : parts) { I.maxlen_lane[l] = std::max(I.maxlen_lane[l], P.maxlen[l]); I.n_entries_lane[l] += P.n_entries[l]; }
	for (uint32_t i = 0; i < n_q; ++i) I.maxE_all = std::max<uint32_t>(I.maxE_all, q_emac[i]);
	slot_take_info(S, I);
	hipStream_t st = h->stage_stream;
	HIPCHK(hipMemcpyAsync(S->plan.p, plan.data(), (size_t)n_q * 4, hipMemcpyHostToDevice, st));
	if (!sorted.empty()) HIPCHK(hipMemcpyAsync(S->idx_sorted.p, sorted.data(), sorted.size() * 4, hipMemcpyHostToDevice, st));
	S->st_has_junk = false;
	for (const Part &P : parts) S->st_has_junk |= P.junk;
	if (S->st_has_junk) {       // rare: second view of the batch without the symbols of code 0 (see StageSlot)
		std::vector<uint64_t> off_s((size_t)n_q + 1, 0);
		std::vector<uint16_t> emac_s(n_q);
		std::vector<uint8_t> codes_s; codes_s.reserve(q_off[n_q] + 16);
		std::vector<uint8_t> nxs(nsh + 1, 0);
		for (uint32_t i = 0; i < n_q; ++i) {
			const uint8_t *c = q_codes + q_off[i]; const uint64_t len = q_off[i + 1] - q_off[i];
			if (!nxv[i]) codes_s.insert(codes_s.end(), c, c + len);
			else if (nxv[i] != 255) for (uint64_t k = 0; k < len; ++k) { if (c[k]) codes_s.push_back(c[k]); }
			off_s[i + 1] = codes_s.size();
			emac_s[i] = (uint16_t)(nxv[i] == 255 ? 0 : q_emac[i] - nxv[i]);
			if (nxv[i] == 255) nxv[i] = 0;            // never searched: nothing to add back
			nxs[q_six ? q_six[i] : i] = nxv[i];
		}
		codes_s.resize(codes_s.size() + 16, 0);
		if ((rc = S->qcodes_s.reserve(codes_s.size()))) return rc;
		if ((rc = S->qoff_s.reserve(((size_t)n_q + 1) * 8))) return rc;
		if ((rc = S->qemac_s.reserve(((size_t)n_q + 1) * 2))) return rc;
		if ((rc = S->nx.reserve((size_t)n_q + 16))) return rc;
		if ((rc = S->nx_six.reserve((size_t)nsh + 16))) return rc;
		HIPCHK(hipMemcpyAsync(S->qcodes_s.p, codes_s.data(), codes_s.size(), hipMemcpyHostToDevice, st));
		HIPCHK(hipMemcpyAsync(S->qoff_s.p, off_s.data(), ((size_t)n_q + 1) * 8, hipMemcpyHostToDevice, st));
		HIPCHK(hipMemcpyAsync(S->qemac_s.p, emac_s.data(), (size_t)n_q * 2, hipMemcpyHostToDevice, st));
		HIPCHK(hipMemcpyAsync(S->nx.p, nxv.data(), n_q, hipMemcpyHostToDevice, st));
		HIPCHK(hipMemcpyAsync(S->nx_six.p, nxs.data(), nsh, hipMemcpyHostToDevice, st));
		const uint32_t qw_g = (S->st_maxlen + 7) / 8;
		if ((rc = S->qpack_s.reserve((size_t)n_q * qw_g * 4 + 64))) return rc;
		const uint64_t total = (uint64_t)n_q * qw_g;
		if (total) hipLaunchKernelGGL(k_pack_queries, dim3((uint32_t)std::min<uint64_t>((total + 255) / 256, (uint64_t)h->n_cu * 16)), dim3(256), 0, st,
			S->qcodes_s.as<uint8_t>(), S->qoff_s.as<uint64_t>(), n_q, qw_g, S->qpack_s.as<uint32_t>());
		HIPCHK(hipGetLastError());
	}
	HIPCHK(hipStreamSynchronize(st));      // the host vectors go out of scope
	return 0;
}

static int resolve_slot(Handle *h, StageSlot *S) {
	if (S->resolved) return 0;
	if (!S->st_nq) { S->resolved = true; S->st_valid = true; S->st_lanes = 0; return 0; }
	HIPCHK(hipEventSynchronize(S->ev_done));
	const BhipStageInfo &I = *S->info_pinned;
	if (I.err == 1) return fail(BHIP_E_QUERYLEN, "query %u has %u symbols (max %d)", I.err_i, I.err_len, BHIP_MAX_QLEN);
	if (I.err == 2) return fail(BHIP_E_ARG, "q_six[%u] out of range", I.err_i);
	S->st_ms_h2d = ev_ms(S->ev_begin, S->ev_done);
	if (I.junk || h->opt_host_routing) { int rc = host_route(h, S); if (rc) return rc; }
	else slot_take_info(S, I);
	S->resolved = true; S->st_valid = true;
	if (getenv("BHIP_DEBUG")) fprintf(stderr, "[bhip] stage: %u entries, %s routing, copies + routing %.2f ms on the staging stream\n", S->st_nq,
		(I.junk || h->opt_host_routing) ? "host" : "device", S->st_ms_h2d);
	return 0;
}

// Grow-only buffers start at sizes a batch of n entries normally stays within (a couple of lane tasks, windows and records
// per entry): the capacity check of bhip_align_staged re-runs a batch whose buffers overflowed, which is what a first batch
// sized for nothing would always do.
static void lane_capacity_floor(Handle *h, Lane *L, uint64_t n) {
	L->task_cap = std::max<uint64_t>(L->task_cap, 3 * n + 4096); L->win_cap = std::max<uint64_t>(L->win_cap, 2 * n + 4096);
	L->raw_cap = std::max<uint64_t>(L->raw_cap, 2 * n + 4096); L->cand_cap = std::max<uint64_t>(L->cand_cap, n / 4 + 4096);
	h->out_cap = std::max<uint64_t>(h->out_cap, 2 * n + 4096);
}

// the slot becomes the batch the alignment kernels work on
static void apply_slot(Handle *h, StageSlot *S) {
	h->cur = S;
	for (uint32_t l = 0; l < S->st_lanes && l < h->lanes.size(); ++l) {
		Lane *L = h->lanes[l];
		for (int c = 0; c < kNumClasses; ++c) {
			L->npf[c] = S->npf[l][c]; L->nex[c] = S->nex[l][c]; L->maxE[c] = S->maxE[l][c]; L->maxwords[c] = S->maxwords[l][c]; L->seed_words[c] = S->seed_words[l][c];
			L->qlist[c] = S->idx_sorted.as<uint32_t>() + S->qlist_off[l][c];
		}
		L->maxlen = S->maxlen_lane[l]; L->n_entries = S->n_entries_lane[l];
		lane_capacity_floor(h, L, L->n_entries);
	}
}

static StageSlot *free_slot(Handle *h) {        // a slot that holds no batch waiting to be aligned (an already aligned batch may be overwritten)
	for (StageSlot &S : h->slots) if (S.state == 0) return &S;
	for (StageSlot &S : h->slots) if (S.state == 2) return &S;
	return nullptr;
}

extern "C" int bhip_stage_spans(void *handle, const BhipQuerySpan *spans, uint32_t n_spans, uint32_t n_shared, uint32_t max_len) {
	Handle *h = (Handle *)handle;
	if (!h || (!spans && n_spans)) return fail(BHIP_E_ARG, "null argument");
	HIPCHK(hipSetDevice(h->device));
	StageSlot *S = free_slot(h);
	if (!S) return fail(BHIP_E_ARG, "every staging slot holds a batch that has not been aligned yet: align one first");
	S->state = 0;
	int rc = stage_enqueue(h, S, spans, n_spans, nullptr, true, n_shared, max_len);
	if (rc) return rc;
	S->state = 1;
	return BHIP_OK;
}

extern "C" int bhip_stage_queries(void *handle, const uint8_t *q_codes, const uint64_t *q_off, const uint16_t *q_emac,
                                  const uint32_t *q_six, const uint8_t *q_rc, const uint8_t *q_flags, uint32_t n_q, uint32_t n_shared) {
	Handle *h = (Handle *)handle;
	if (!h) return fail(BHIP_E_ARG, "null handle");
	if (n_q && (!q_codes || !q_off || !q_emac)) return fail(BHIP_E_ARG, "null query arrays");
	HIPCHK(hipSetDevice(h->device));
	for (StageSlot &S : h->slots) if (S.state == 1) S.state = 0;      // this entry point replaces whatever was waiting
	StageSlot *S = free_slot(h);
	S->state = 0;
	BhipQuerySpan sp;
	memset(&sp, 0, sizeof sp);
	sp.codes = q_codes; sp.off = q_off; sp.emac = q_emac; sp.rc = q_rc; sp.flags = q_flags; sp.n = n_q; sp.q_base = 0;
	int rc = stage_enqueue(h, S, &sp, n_q ? 1u : 0u, q_six, false, n_shared, 0);
	if (rc) return rc;
	if ((rc = resolve_slot(h, S))) return rc;      // synchronous: the caller's arrays are free again at return
	S->spans.clear(); S->six_explicit = nullptr;
	S->state = 1;
	return BHIP_OK;
}

// Allocate, ahead of the first batch, what batches of up to n_entries entries of up to max_len symbols need (both staging slots,
// the scratch of the alignment kernels, the record buffers): a batch scheduler calls it once so that no allocation -- each one
// synchronises the device -- falls into its first batches.
extern "C" int bhip_reserve(void *handle, uint32_t n_entries, uint32_t max_len) {
	Handle *h = (Handle *)handle;
	if (!h) return fail(BHIP_E_ARG, "null handle");
	if (!n_entries) return BHIP_OK;
	if (!max_len || max_len > BHIP_MAX_QLEN) max_len = BHIP_MAX_QLEN;
	HIPCHK(hipSetDevice(h->device));
	int rc;
	const size_t n = n_entries, nb = n * max_len, qw = (max_len + 7) / 8;
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
	if ((rc = ensure_lanes(h, 1))) return rc;
	Lane *L = h->lanes[0];
	lane_capacity_floor(h, L, n);
	const int cls = class_of_len(max_len);
	if ((rc = L->cand.reserve(L->cand_cap * sizeof(uint2))) || (rc = L->raw.reserve(L->raw_cap * sizeof(BhipRawHit))) || (rc = L->wide.reserve(L->raw_cap * sizeof(uint32_t))) ||
	    (rc = L->rs_lists.reserve(L->raw_cap * sizeof(uint32_t) * 10)) || (rc = L->scratch.reserve(L->scratch_cap * sizeof(uint32_t))) || (rc = L->wins.reserve(L->win_cap * sizeof(BhipWin))) ||
	    (rc = L->tasks.reserve(L->task_cap * sizeof(uint2))) || (rc = L->tasks2.reserve(L->task_cap * sizeof(uint2))) || (rc = L->tasks2k.reserve(L->task_cap * sizeof(uint2))) ||
	    (rc = L->wins2.reserve(L->win_cap * sizeof(BhipWin))) || (rc = L->peq.reserve(n * 16 * kClasses[cls] * 4)) || (rc = L->peqp.reserve(n * 16 * 6 * 4)) ||
	    (rc = L->peq_alt.reserve(n * 16 * kClasses[cls] * 4)) || (rc = L->peqp_alt.reserve(n * 16 * 6 * 4)) ||
	    (rc = L->fb_list.reserve(n * 4 + 16)) || (rc = L->ranges_c[cls].reserve(n * 16 * 8 + 16)) || (rc = L->hdr_c[cls].reserve(n * 8 + 16))) return rc;
	if ((rc = h->best.reserve((n + 1) * 4)) || (rc = h->out.reserve(h->out_cap * sizeof(BhipHit))) || (rc = h->shared_ctr.reserve(sizeof(SharedCtr))) ||
	    (rc = h->sort_idx.reserve(h->out_cap * 4)) || (rc = h->sort_keys.reserve((n + 1) * 4)) || (rc = h->sort_keys2.reserve((n + 1) * 4)) ||
	    (rc = h->out_sorted.reserve(h->out_cap * sizeof(BhipHit))) || (rc = h->out_sorted2.reserve(h->out_cap * sizeof(BhipHit)))) return rc;
	{
		size_t tb = 0;
		HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, h->sort_keys.as<uint32_t>(), h->sort_keys2.as<uint32_t>(), (int)(n + 1), h->stream));
		if ((rc = h->sort_tmp.reserve(tb))) return rc;
	}
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

// page-locked host memory for the arrays handed to bhip_stage_spans and the result buffers of bhip_align_staged
extern "C" void *bhip_alloc_host(uint64_t bytes) {
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
	// The first ASYNCHRONOUS copy into a new page-locked allocation blocks its caller for tens of milliseconds (measured: 19 ms
	// in front of the first hand-over copy of a run into a 260 MB buffer, 7 us for every later copy into the same allocation;
	// a synchronous hipMemcpy does not take that path): make that first copy here.
	void *d = nullptr; hipStream_t st = nullptr;
	if (hipMalloc(&d, 256) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) {
		(void)hipMemcpyAsync(p, d, bytes < 64 ? bytes : 64, hipMemcpyDeviceToHost, st);
		(void)hipMemcpyAsync(d, p, bytes < 64 ? bytes : 64, hipMemcpyHostToDevice, st);
		(void)hipStreamSynchronize(st);
		// ... and so do the first query of the allocation's attributes (15-29 ms: bhip_align_staged asks whether its record buffer is
		// page-locked) and the first copy to an address INSIDE the allocation (10-16 ms, measured in front of the second batch's
		// hand-over copy): both here, once
		hipPointerAttribute_t at;
		memset(&at, 0, sizeof at);
		(void)hipPointerGetAttributes(&at, p);
		if (bytes > 4096) {
			char *mid = (char *)p + ((bytes / 2) & ~(uint64_t)63);
			(void)hipPointerGetAttributes(&at, mid);
			(void)hipMemcpyAsync(mid, d, 64, hipMemcpyDeviceToHost, st);
			(void)hipMemcpyAsync(d, mid, 64, hipMemcpyHostToDevice, st);
			(void)hipStreamSynchronize(st);
		}
	}
	(void)hipGetLastError();
	if (st) (void)hipStreamDestroy(st);
	if (d) (void)hipFree(d);
	return p;
}
extern "C" void bhip_free_host(void *p) { if (p) (void)hipHostFree(p); }
extern "C" int bhip_host_register(void *p, uint64_t bytes) {
	if (!p || !bytes) return BHIP_OK;
	if (hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); return fail(BHIP_E_DEVICE, "hipHostRegister(%llu bytes) failed", (unsigned long long)bytes); }
	return BHIP_OK;
}
extern "C" int bhip_host_unregister(void *p) {
	if (p && hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return fail(BHIP_E_DEVICE, "hipHostUnregister failed"); }
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
	L->launches = 0; L->prefix_words = 0; L->n_pairs_ex = 0; L->pf_launches = 0;
	for (int c = 0; c < kNumClasses; ++c) { L->pf_masked[c] = false; L->pruned[c] = false; }
	const uint32_t grid_my = (uint32_t)h->n_cu * (uint32_t)h->opt_sweep_blocks;   // < 8 leaves wave slots for the other stages' kernels
	(void)start;
	for (int cls = 0; cls < kNumClasses; ++cls) {
		const uint32_t n_pf = L->npf[cls], n_ex = L->nex[cls], n_list = n_pf + n_ex;
		if (!n_list) continue;
		const int NW = kClasses[cls];
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
		L->masked = masked;
		L->pf_masked[cls] = masked && n_pf;
		HIPCHK(hipEventRecord(ce[2], pf));
		// column sweep on the sweep stream, behind this lane's prefilter
		HIPCHK(hipStreamWaitEvent(sw, ce[2], 0));
		HIPCHK(hipEventRecord(ce[6], sw));
		if (n_pf) {
			if (masked) launch_prefix_task(h, L, sw, NWP, (uint32_t)h->n_cu * 4u * (uint32_t)h->opt_sweep_blocks, L->tasks.as<uint2>(), &dc->n_tasks_cls[cls], qlist, L->wins.as<BhipWin>(), &dc->n_wins_cls[cls], dc);
			if (NWP) launch_prefix(h, L, sw, NWP, grid_my, L->cand.as<uint2>(), &dc->n_cand_cls[cls], L->cand_cap, 0, qlist, &dc->n_wins_cls[cls], dc);
			else launch_myers(h, L, sw, cls, grid_my, L->cand.as<uint2>(), &dc->n_cand_cls[cls], L->cand_cap, 0, qlist, L->raw.as<BhipRawHit>(),
				&dc->n_raw, (uint32_t)L->raw_cap, h->best.as<uint32_t>(), nullptr, dc);
			HIPCHK(hipGetLastError());
			++L->launches;
		}
		HIPCHK(hipEventRecord(ce[3], sw));
		if (n_ex) {
			const uint64_t np = (uint64_t)n_ex * h->n_clumps;
			const uint32_t g = (uint32_t)std::min<uint64_t>((np + 15) / 16, grid_my);
			if (NWP) launch_prefix(h, L, sw, NWP, g, nullptr, nullptr, np, n_pf, qlist, &dc->n_wins_cls[cls], dc);
			else launch_myers(h, L, sw, cls, g, nullptr, nullptr, np, n_pf, qlist, L->raw.as<BhipRawHit>(), &dc->n_raw, (uint32_t)L->raw_cap,
				h->best.as<uint32_t>(), nullptr, dc);
			HIPCHK(hipGetLastError());
			++L->launches;
			L->n_pairs_ex += np;
		}
		HIPCHK(hipEventRecord(ce[4], sw));
		HIPCHK(hipStreamWaitEvent(po, ce[4], 0));
		if (NWP) { launch_window(h, L, po, cls, NWP, grid_my, qlist, L->wins.as<BhipWin>(), &dc->n_wins_cls[cls], dc); HIPCHK(hipGetLastError()); }
		L->pruned[cls] = masked && prune && n_pf;
		if (masked && prune && n_pf) {
			// second sweep: the deferred lanes whose lower bound is not above the minimum found by the first sweep
			HIPCHK(hipEventRecord(L->ev_ph[cls][0], po));
			HIPCHK(hipStreamWaitEvent(sw, L->ev_ph[cls][0], 0));
			hipLaunchKernelGGL(k_task_filter, dim3((uint32_t)h->n_cu * 8), dim3(256), 0, sw, L->tasks2.as<uint2>(), &dc->n_tasks2_cls[cls], (uint32_t)L->task_cap, qlist,
				h->cur->st_has_six ? h->cur->qsix.as<uint32_t>() : nullptr, h->best.as<uint32_t>(), L->tasks2k.as<uint2>(), &dc->n_tasks2k_cls[cls]);
			launch_prefix_task(h, L, sw, NWP, (uint32_t)h->n_cu * 4u * (uint32_t)h->opt_sweep_blocks, L->tasks2k.as<uint2>(), &dc->n_tasks2k_cls[cls], qlist, L->wins2.as<BhipWin>(), &dc->n_wins2_cls[cls], dc);
			HIPCHK(hipGetLastError());
			HIPCHK(hipEventRecord(L->ev_ph[cls][1], sw));
			HIPCHK(hipStreamWaitEvent(po, L->ev_ph[cls][1], 0));
			launch_window(h, L, po, cls, NWP, grid_my, qlist, L->wins2.as<BhipWin>(), &dc->n_wins2_cls[cls], dc);
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
#define RS_LAUNCH(SET, BLOCKS) hipLaunchKernelGGL(k_rescore_reg<SET>, dim3((uint32_t)h->n_cu * std::min<uint32_t>(32u, blocks_per_cu((const void *)k_rescore_reg<SET>, 64, 0))), dim3(64), 0, po, L->raw.as<BhipRawHit>(), L->rs_lists.as<uint32_t>(), dc->n_rs, (uint32_t)L->raw_cap, \
			h->cur->qoff.as<uint64_t>(), h->cur->st_has_rc ? h->cur->qrc.as<uint8_t>() : nullptr, h->cur->qpack.as<uint32_t>(), qw_g, \
			h->ref_lane.as<uint8_t>(), h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), h->lut.as<uint8_t>(), h->out.as<BhipHit>(), &sc->n_out, (uint32_t)h->out_cap, &sc->err)
		RS_LAUNCH(0, 16);
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
		h->cur->st_has_six ? h->cur->qsix.as<uint32_t>() : nullptr, h->cur->st_has_rc ? h->cur->qrc.as<uint8_t>() : nullptr, h->ref.as<uint8_t>(), h->ref_off.as<uint64_t>(),
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
			if (!n_pf || !class_prefix_words(h, N->maxE[l][cls], kClasses[cls])) continue;      // (lane-resolved prefilter only)
			if (L->seeded_ok[cls] && L->seeded_seq[cls] == N->seq) continue;
			if (launch_seed(h, L, h->pf_stream, N, cls, N->idx_sorted.as<uint32_t>() + N->qlist_off[l][cls], n_pf, N->maxwords[l][cls], true)) { (void)hipGetLastError(); return; }
		}
		// the match profiles as well, when the lane has a single class (its two buffer pairs then simply alternate): built in place
		// they would run beside the prefilter -- which no longer has its seed lookups in front -- and slow it down
		int only = -1, n_cls = 0;
		for (int cls = 0; cls < kNumClasses; ++cls) if (N->npf[l][cls] + N->nex[l][cls]) { only = cls; ++n_cls; }
		if (n_cls == 1 && !(L->alt_ok && L->alt_seq == N->seq)) {
			const uint32_t n_list = N->npf[l][only] + N->nex[l][only];
			const int NW = kClasses[only], NWP = class_prefix_words(h, N->maxE[l][only], NW);
			L->alt_ok = false;
			if (hipEventRecord(L->ev_peq_alt[0], h->pf_stream) != hipSuccess ||
			    launch_peq(h, h->pf_stream, N, N->idx_sorted.as<uint32_t>() + N->qlist_off[l][only], n_list, NW, NWP, L->peq_alt, L->peqp_alt, (uint32_t)h->opt_peq_ahead_blocks) ||
			    hipEventRecord(L->ev_peq_alt[1], h->pf_stream) != hipSuccess) { (void)hipGetLastError(); return; }
			L->alt_ok = true; L->alt_seq = N->seq; L->alt_cls = only; L->alt_nwp = NWP; L->alt_n = n_list;
		}
	}
}

extern "C" int bhip_align_staged(void *handle, int all_hits, BhipHit *hits, uint64_t cap, uint64_t *n_hits) {
	Handle *h = (Handle *)handle;
	if (!h || !n_hits) return fail(BHIP_E_ARG, "null argument");
	*n_hits = 0;
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
	bool sorted_ahead = false; int o_ahead = 0;
	for (int attempt = 0; attempt < 24; ++attempt) {
		int rc;
		sorted_ahead = false;
		// the records of this batch are still resident when the previous call only failed for the size of the caller's buffer
		if (h->res_valid && h->res_seq == slot->seq && h->res_all_hits == all_hits) { hsc.n_out = h->res_n; hsc.err = 0; h->stats = h->res_stats; *n_hits = hsc.n_out; }
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
			    (rc = sorted.reserve((size_t)h->out_cap * sizeof(BhipHit)))) return rc;
			cnt = h->sort_keys.as<uint32_t>(); off = h->sort_keys2.as<uint32_t>(); rank = h->sort_idx.as<uint32_t>();
			HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, cnt, off, (int)(n_q + 1), h->post_stream));
			if ((rc = h->sort_tmp.reserve(tmp_bytes))) return rc;
			SharedCtr *sc = h->shared_ctr.as<SharedCtr>();
			const uint32_t g = (uint32_t)h->n_cu * 8;
			if (h->copy_pending[o_ahead]) HIPCHK(hipStreamWaitEvent(h->post_stream, h->ev_copied[o_ahead], 0));      // the copy that last read this buffer
			HIPCHK(hipEventRecord(h->ev[4], h->post_stream));
			// (counts and ranks were taken by the re-scoring kernels as they wrote the records)
			HIPCHK(hipcub::DeviceScan::ExclusiveSum(h->sort_tmp.p, tmp_bytes, cnt, off, (int)(n_q + 1), h->post_stream));
			hipLaunchKernelGGL(k_hit_scatter, dim3(g), dim3(256), 0, h->post_stream, h->out.as<BhipHit>(), (uint32_t)h->out_cap, &sc->n_out, off, rank, sorted.as<BhipHit>(),
				h->cur->has_qmap ? h->cur->qmap.as<uint32_t>() : (const uint32_t *)nullptr);
			hipLaunchKernelGGL(k_hit_fix, dim3(std::min<uint32_t>((n_q + 255) / 256, (uint32_t)h->n_cu * 8)), dim3(256), 0, h->post_stream, sorted.as<BhipHit>(), off, cnt, n_q);
			HIPCHK(hipGetLastError());
			HIPCHK(hipEventRecord(h->ev[5], h->post_stream));
			sorted_ahead = true;
		}
		if (!h->hsc_pinned) HIPCHK(hipHostMalloc((void **)&h->hsc_pinned, sizeof(SharedCtr), hipHostMallocDefault));
		HIPCHK(hipMemcpyAsync(h->hsc_pinned, h->shared_ctr.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, h->post_stream));
		HIPCHK(hipEventRecord(h->ev[3], h->post_stream));
		seed_next_batch(h, slot, h->ev[3]);
		HIPCHK(hipEventSynchronize(h->ev[2]));
		HIPCHK(hipStreamSynchronize(h->sweep_stream));
		HIPCHK(hipStreamSynchronize(h->post_stream));
		for (uint32_t l = 0; l < nl; ++l) if (h->lanes[l]->n_entries) h->lanes[l]->hc = *h->lanes[l]->hc_pinned;
		for (uint32_t l = 0; l < nl; ++l) {      // a lane whose records mostly survive the counting filter does better with the exact table
			Lane *L = h->lanes[l];
			// (with the minimum-only semantics the counting-filter kernel also splits off the lanes that cannot hold a minimum --
			// the second sweep -- which the exact-table kernel does not: it only takes over when most records survive)
			if (L->n_entries && L->pf_algo == 0 && L->hc.ent_read > 100000 && (double)L->hc.surv_sum > (all_hits ? 0.20 : 0.50) * (double)L->hc.ent_read) L->pf_algo = 1;
		}
		if (getenv("BHIP_DEBUG")) for (uint32_t l = 0; l < nl; ++l) {
			const Lane *L = h->lanes[l];
			if (!L->n_entries) continue;
			fprintf(stderr, "[bhip] lane %u: %llu list records, %llu survived the counting filter, next prefilter algorithm %d\n", l, (unsigned long long)L->hc.ent_read, (unsigned long long)L->hc.surv_sum, L->pf_algo);
			for (int cls = 0; cls < kNumClasses; ++cls) if (L->npf[cls] + L->nex[cls])
				fprintf(stderr, "[bhip] lane %u class NW=%d: prefiltered %u exhaustive %u maxE %u maxwords %u | tasks %u + deferred %u (kept %u) clump pairs %u windows %u + %u | fallback queries(last class) %u raw %u\n",
					l, kClasses[cls], L->npf[cls], L->nex[cls], L->maxE[cls], L->maxwords[cls], L->hc.n_tasks_cls[cls], L->hc.n_tasks2_cls[cls], L->hc.n_tasks2k_cls[cls], L->hc.n_cand_cls[cls], L->hc.n_wins_cls[cls], L->hc.n_wins2_cls[cls], L->hc.n_fb, L->hc.n_raw);
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
				h->ref.as<uint8_t>(), h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), h->lut.as<uint8_t>(), h->out.as<BhipHit>(),
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
		*n_hits = hsc.n_out;
		h->last_n_out = 0;
		// statistics
		BhipStats &S = h->stats;
		S.n_queries = n_q; S.n_hits = hsc.n_out;
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
		h->res_valid = true; h->res_seq = slot->seq; h->res_all_hits = all_hits; h->res_n = hsc.n_out; h->res_stats = h->stats;
		}
		BhipStats &S = h->stats;
		if (hits && hsc.n_out > cap) return fail(BHIP_E_CAPACITY, "hit buffer holds %llu records, %u needed", (unsigned long long)cap, hsc.n_out);
		HIPCHK(hipEventRecord(h->ev[8], h->stream));
		const bool dbg_t = getenv("BHIP_DEBUG_TIMES") != nullptr;
		const auto tq0 = std::chrono::steady_clock::now();
		auto tq = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq0).count(); };
		double tq1 = 0, tq2 = 0, tq3 = 0, tq4 = 0;
		if (hsc.n_out) {
			const uint32_t n = hsc.n_out;
			if ((rc = h->sort_idx.reserve((size_t)n * 4)) || (rc = h->sort_keys.reserve((size_t)(n_q + 1) * 4)) || (rc = h->sort_keys2.reserve((size_t)(n_q + 1) * 4)) ||
			    0) return rc;
			const bool async = h->opt_async_d2h && hits;
			const int o = async ? (h->out_idx ^= 1) : 0;
			DBuf &sorted = o ? h->out_sorted2 : h->out_sorted;
			if (sorted_ahead && o == o_ahead) tq1 = tq();          // grouped already, behind the re-scorer
			else {
			if (h->copy_pending[o]) { HIPCHK(hipEventSynchronize(h->ev_copied[o])); h->copy_pending[o] = false; }    // the copy that last read this buffer
			if ((rc = sorted.reserve((size_t)n * sizeof(BhipHit)))) return rc;
			uint32_t *cnt = h->sort_keys.as<uint32_t>(), *off = h->sort_keys2.as<uint32_t>(), *rank = h->sort_idx.as<uint32_t>();
			const uint32_t g = std::min<uint32_t>((n + 255) / 256, (uint32_t)h->n_cu * 8);
			HIPCHK(hipMemsetAsync(cnt, 0, (size_t)(n_q + 1) * 4, h->stream));
			tq1 = tq();
			hipLaunchKernelGGL(k_hit_count, dim3(g), dim3(256), 0, h->stream, h->out.as<BhipHit>(), n, (const uint32_t *)nullptr, cnt, rank);
			size_t tmp_bytes = 0;
			HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, cnt, off, (int)(n_q + 1), h->stream));
			if ((rc = h->sort_tmp.reserve(tmp_bytes))) return rc;
			HIPCHK(hipcub::DeviceScan::ExclusiveSum(h->sort_tmp.p, tmp_bytes, cnt, off, (int)(n_q + 1), h->stream));
			hipLaunchKernelGGL(k_hit_scatter, dim3(g), dim3(256), 0, h->stream, h->out.as<BhipHit>(), n, (const uint32_t *)nullptr, off, rank, sorted.as<BhipHit>(),
				h->cur->has_qmap ? h->cur->qmap.as<uint32_t>() : (const uint32_t *)nullptr);
			hipLaunchKernelGGL(k_hit_fix, dim3(std::min<uint32_t>((n_q + 255) / 256, (uint32_t)h->n_cu * 8)), dim3(256), 0, h->stream, sorted.as<BhipHit>(), off, cnt, n_q);
			HIPCHK(hipGetLastError());
			}
			tq2 = tq();
			const size_t bytes = (size_t)n * sizeof(BhipHit);
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
			h->last_n_out = n; h->last_out = o;
		}
		tq4 = tq();
		HIPCHK(hipEventRecord(h->ev[9], h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		if (dbg_t) fprintf(stderr, "[bhip] delivery host ms: reserve+memset %.3f, sort launches %.3f, pointer check %.3f, copy enqueue %.3f, sync %.3f\n", tq1, tq2 - tq1, tq3 - tq2, tq4 - tq3, tq() - tq4);
		S.ms_h2d = h->cur->st_ms_h2d; S.ms_d2h = ev_ms(h->ev[8], h->ev[9]) + (sorted_ahead ? ev_ms(h->ev[4], h->ev[5]) : 0.0f); S.ms_total = ev_ms(h->ev[0], h->ev[9]);
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
	const int cls = class_of_len(std::max<uint32_t>(maxlen, 1)), NW = kClasses[cls];
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
		h->cur->qcodes.as<uint8_t>(), h->cur->qoff.as<uint64_t>(), (const uint32_t *)nullptr, n_q, NW, 0, h->mm, L->peq.as<uint32_t>(), (const uint32_t *)nullptr, 0u);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(h->ev[0], st));
	launch_myers(h, L, st, cls, (uint32_t)std::min<uint64_t>((n_pairs + 15) / 16, (uint64_t)h->n_cu * 8), h->pairs.as<uint2>(), nullptr, n_pairs, 0, nullptr,
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
			for (uint32_t i = 0; i < n_q; ++i) plan[i] = make_seed_plan(q_codes + q_off[i], (uint32_t)(q_off[i + 1] - q_off[i]), q_emac[i], (uint32_t)h->K, h->opt_prefilter_stride);
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
	if (!dst_device) return fail(BHIP_E_ARG, "null destination");
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
