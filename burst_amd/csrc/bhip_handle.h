// burst_amd/csrc/bhip_handle.h -- the device handle behind the C ABI (include/burst_hip.h) and what the translation units of
// libburst_hip.so share: kernel declarations (bhip_kernels.hip), the grow-only device buffer, the staged-batch slots, the
// sub-pipeline lanes, the handle itself and the few host functions that cross files (bhip_init.hip: handle life cycle, database
// upload, accelerator; bhip_stage.hip: staging and routing of batches; bhip_align.hip: the alignment chain and the kernel-level
// entry points).
#ifndef BHIP_HANDLE_H
#define BHIP_HANDLE_H
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
#include <mutex>
#include <atomic>
#include <chrono>
#include <thread>
#include "burst_hip.h"
#include "bhip_internal.h"

// ---- kernels (bhip_kernels.hip) -------------------------------------------------------------------
__global__ void k_acx_offsets(const uint32_t *, uint64_t, int, uint32_t *, unsigned long long *);
__global__ void k_acx_lines(const uint32_t *, uint64_t, int, unsigned long long *, uint4 *);
__global__ void k_acx_decode(const uint8_t *, const unsigned long long *, const uint32_t *, BhipAcxView, uint64_t, int, uint32_t, uint32_t *, uint32_t *);
__global__ void k_transpose_refs(const uint8_t *, const uint64_t *, const uint32_t *, const uint64_t *, uint32_t, uint4 *);
__global__ void k_build_peq(const uint8_t *, const uint64_t *, const uint32_t *, uint32_t, int, int, BhipMatchMask, uint32_t *, const uint32_t *, uint32_t, uint32_t);
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
__global__ void k_myers_long(const uint2 *, const uint32_t *, uint64_t, uint32_t, uint32_t, const uint32_t *, const uint32_t *,
	const uint64_t *, const uint16_t *, const uint32_t *, const uint4 *, const uint64_t *, const uint32_t *, uint32_t,
	BhipRawHit *, uint32_t *, uint32_t, uint32_t *, uint8_t *, unsigned long long *, unsigned long long *, uint32_t);
template <int NWP> __global__ void k_myers_prefix(const uint2 *, const uint32_t *, uint64_t, uint32_t, uint32_t, const uint32_t *, const uint32_t *,
	const uint64_t *, const uint16_t *, const uint4 *, const uint64_t *, const uint32_t *, uint32_t, BhipWin *, uint32_t *, uint32_t,
	unsigned long long *, unsigned long long *, uint32_t *, const uint32_t *);
template <int NW> __global__ void k_myers_window(const BhipWin *, const uint32_t *, uint32_t, int, int, const uint32_t *, const uint32_t *,
	const uint4 *, BhipRawHit *, uint32_t *, uint32_t, uint32_t *, unsigned long long *, const uint32_t *);
template <int BW> __global__ void k_myers_window_band(const BhipWin *, const uint32_t *, uint32_t, int, int, const uint32_t *, const uint32_t *,
	const uint4 *, BhipRawHit *, uint32_t *, uint32_t, uint32_t *, unsigned long long *, const uint32_t *);
__global__ void k_extract_kmers(const uint4 *, const uint64_t *, const uint32_t *, const uint64_t *, uint32_t, uint32_t, int, unsigned long long *, uint16_t *, uint32_t *);
__global__ void k_attach_masks(BhipAcxView, uint64_t, const unsigned long long *, const uint16_t *, uint32_t, const uint32_t *, uint32_t *, uint32_t, uint32_t);
template <int HTB> __global__ void k_prefilter_mask(const uint2 *, const uint2 *, uint32_t, uint32_t, const uint32_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t,
	uint2 *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, unsigned long long *,
	uint2 *, uint32_t *, uint32_t);
template <int CB, int RBT> __global__ void k_prefilter_cf(const uint2 *, const uint2 *, uint32_t, uint32_t, const uint32_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t,
	uint2 *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, unsigned long long *, unsigned long long *,
	uint2 *, uint32_t *, int, const uint32_t *, const uint32_t *, int);
template <int MODE, int BIG> __global__ void k_prefilter_cq(const uint2 *, const uint2 *, uint32_t, uint32_t, const uint32_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t,
	uint2 *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, unsigned long long *, unsigned long long *,
	uint2 *, uint32_t *, int, const uint32_t *, const uint32_t *, int);
template <int MODE, int BIG> __global__ void k_prefilter_cw(const uint2 *, const uint2 *, uint32_t, uint32_t, const uint32_t *, const uint32_t *, uint32_t, const uint32_t *, uint32_t,
	uint2 *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, uint32_t *, unsigned long long *, unsigned long long *, unsigned long long *, unsigned long long *,
	uint2 *, uint32_t *, int, const uint32_t *, const uint32_t *, int);
__global__ void k_task_filter(const uint2 *, const uint32_t *, uint32_t, const uint32_t *, const uint32_t *, const uint32_t *, uint2 *, uint32_t *);
__global__ void k_seed_ranges(const uint8_t *, const uint64_t *, const uint32_t *, uint32_t, BhipAcxView, int, const uint32_t *, uint32_t, uint2 *, uint2 *, const uint32_t *, uint32_t, const uint16_t *, uint4 *, const uint32_t *, uint32_t, uint32_t, BhipAlt);
template <int NWP> __global__ void k_myers_prefix_task(const uint2 *, const uint32_t *, uint32_t, const uint32_t *, const uint32_t *, const uint64_t *,
	const uint16_t *, const uint4 *, const uint64_t *, const uint32_t *, BhipWin *, uint32_t *, uint32_t, unsigned long long *, uint32_t *, const uint4 *, const uint32_t *);
template <bool WIDE> __global__ void k_rescore(const BhipRawHit *, const uint32_t *, uint32_t, const uint32_t *, const uint32_t *,
	const uint32_t *, int, const uint8_t *, const uint64_t *, const uint32_t *, const uint8_t *, const uint8_t *, const uint64_t *,
	const uint32_t *, const uint8_t *, BhipHit *, uint32_t *, uint32_t, uint32_t *, uint32_t *, uint32_t *, unsigned long long *,
	unsigned long long, uint32_t *, const uint32_t *, uint32_t, uint32_t, uint32_t);
__global__ void k_pack_queries(const uint8_t *, const uint64_t *, uint32_t, uint32_t, uint32_t *);
__global__ void k_unpack4(const uint8_t *, uint64_t, uint64_t, uint8_t *);
__global__ void k_unpack2(const uint8_t *, uint64_t, uint64_t, uint8_t *);
__global__ void k_span_fill(const uint64_t *, uint32_t, uint32_t, uint64_t, uint32_t, uint64_t *, uint32_t *, uint32_t *);
__global__ void k_route(const uint64_t *, const uint32_t *, uint32_t, const uint16_t *, const uint32_t *, const uint8_t *, uint32_t, uint32_t, uint32_t, int, int, int,
	uint32_t *, uint8_t *, uint32_t *, BhipStageInfo *, BhipAlt);
__global__ void k_rescore_classify(const BhipRawHit *, const uint32_t *, uint32_t, const uint32_t *, int, const uint64_t *, const uint32_t *, const uint8_t *,
	const uint32_t *, BhipHit *, uint32_t *, uint32_t, uint32_t *, uint32_t *, uint32_t *, uint32_t *, uint32_t, int);
template <int SET> __global__ void k_rescore_reg(const BhipRawHit *, const uint32_t *, const uint32_t *, uint32_t, const uint64_t *, const uint8_t *, const uint32_t *, uint32_t,
	const uint8_t *, const uint64_t *, const uint32_t *, const uint8_t *, BhipHit *, uint32_t *, uint32_t, uint32_t *);
// error text of the calling thread (bhip_last_error); defined in bhip_init.hip
int bhip_fail_msg(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#define fail(...) bhip_fail_msg(__VA_ARGS__)
// BHIP_TRACE_SYNC=1 (debugging a device fault, which ends the process without a Python frame): every checked call is followed by a line on
// stderr and a device synchronisation -- the last line names the call behind which the device died
inline bool bhip_trace_sync() { static const bool on = getenv("BHIP_TRACE_SYNC") != nullptr; return on; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
	return fail(BHIP_E_DEVICE, "%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
	if (bhip_trace_sync()) { fprintf(stderr, "[bhip sync] %s:%d\n", __FILE__, __LINE__); (void)hipDeviceSynchronize(); } } while (0)

// grow-only device buffer
// (address-range reservations, mappings and unmappings of ALL handles of the process go one at a time: ranks that share a process -- one
// thread each -- build their accelerators at the same moment)
inline std::mutex &bhip_vmm_mutex() { static std::mutex m; return m; }
// (set by bhip_team_create: the ranks are threads of THIS process on several devices and will copy regions out of each other's record
// areas -- only then are a range's chunks made accessible to the peer devices; a process with one rank never touches the other devices)
inline std::atomic<int> &bhip_vmm_peer_access() { static std::atomic<int> f(0); return f; }
struct DBuf {
	void *p = nullptr; size_t cap = 0;
	// growable variant (reserve_growable / grow_to): ONE address range whose physical memory is mapped chunk by chunk as the array
	// grows (HIP virtual memory management) -- for the accelerator's record area, whose size is only known when it has been built.
	// Chunks can also be mapped from the TOP of the range downwards (grow_top_to): the word-sliced accelerator build sorts in the part of
	// the record area its records have not reached yet.  A chunk that has been unmapped is never mapped again while the range lives (an
	// address that changed its memory under running ranks read back as zeros in one run of three: tests/test_gpu_acx.py, round 5) --
	// shrink_to is for the END of a build.
	bool vmm = false; size_t va_size = 0; int vmm_device = 0;
	std::vector<hipMemGenericAllocationHandle_t> chunks;      // one per chunk of the range, nullptr = not mapped
	size_t n_mapped = 0;
	static constexpr size_t kChunk = 1ull << 30;
	// Memory that has just been given back -- by this process (the accelerator build returns the 39 GB its sort worked in; the upload buffer
	// of the references) or by the process that held the device before -- is not allocatable at once: hipMemGetInfo sees it come back over
	// some hundred milliseconds (round 6: `burst_hip` at the metric's size, 255 GB resident, found 16 GB free right behind its build where
	// 55 GB are free a moment later, and an 8 GB reservation of batch buffers failed by 100 MB).  An allocation that fails for memory is
	// therefore tried again for a few seconds.
	static hipError_t malloc_patiently(void **q, size_t want) {
		hipError_t e = hipMalloc(q, want);
		if (e != hipErrorOutOfMemory) return e;
		(void)hipGetLastError();
		size_t last_free = 0, total = 0;
		(void)hipMemGetInfo(&last_free, &total);
		const bool dbg = getenv("BHIP_DEBUG") != nullptr;
		if (dbg) fprintf(stderr, "[bhip] hipMalloc(%zu) found no memory with %.2f GB reported free: waiting for memory that is on its way back\n", want, last_free / 1e9);
		for (int tries = 0; tries < 80; ++tries) {      // at most ~8 s (no device synchronisation here: the caller may have work enqueued that waits for events it has yet to record)
			std::this_thread::sleep_for(std::chrono::milliseconds(100));
			e = hipMalloc(q, want);
			if (e != hipErrorOutOfMemory) { if (dbg) fprintf(stderr, "[bhip] ... there after %d ms\n", 100 * (tries + 1)); return e; }
			(void)hipGetLastError();
			if (dbg && tries % 10 == 9) { size_t f = 0; (void)hipMemGetInfo(&f, &total); fprintf(stderr, "[bhip] ... %d ms: %.2f GB reported free\n", 100 * (tries + 1), f / 1e9); }
		}
		return e;
	}
	// BHIP_POISON=<byte> (tests, tools/fuzz_repro.sh): fresh device memory is filled with that byte instead of whatever the last owner left
	// there -- a kernel that reads what nobody wrote then fails at once and not in the 235th configuration of one process
	static void poison(void *q, size_t n) {
		static const char *ev = getenv("BHIP_POISON");
		if (ev && q) { (void)hipMemset(q, atoi(ev) & 255, n); (void)hipDeviceSynchronize(); if (getenv("BHIP_DEBUG")) fprintf(stderr, "[bhip] buffer %p .. %p (%zu bytes)\n", q, (void *)((char *)q + n), n); }
	}
	int reserve(size_t bytes) {
		if (bytes <= cap) return 0;
		release();
		size_t want = bytes + bytes / 4 + 256;
		hipError_t e = malloc_patiently(&p, want);
		if (e != hipSuccess) { p = nullptr; return fail(BHIP_E_DEVICE, "hipMalloc(%zu): %s", want, hipGetErrorString(e)); }
		poison(p, want);
		cap = want; return 0;
	}
	// the same without the growth slack: for the database-sized buffers that are allocated once (a quarter more of a 170 GB record
	// area is what decides whether a database fits the device)
	int reserve_exact(size_t bytes) {
		if (bytes <= cap) return 0;
		release();
		hipError_t e = malloc_patiently(&p, bytes);
		if (e != hipSuccess) { p = nullptr; return fail(BHIP_E_DEVICE, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
		poison(p, bytes);
		cap = bytes; return 0;
	}
	// an address range for up to max_bytes with nothing behind it yet; non-zero (and no error text) when the runtime cannot do it
	int reserve_growable(size_t max_bytes, int device) {
		release();
		std::lock_guard<std::mutex> lk(bhip_vmm_mutex());
		int ok = 0;
		if (hipDeviceGetAttribute(&ok, hipDeviceAttributeVirtualMemoryManagementSupported, device) != hipSuccess || !ok) { (void)hipGetLastError(); return 1; }
		const size_t sz = ((max_bytes + kChunk - 1) / kChunk + 1) * kChunk;
		if (hipMemAddressReserve(&p, sz, kChunk, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return 1; }
		vmm = true; va_size = sz; vmm_device = device; cap = 0; n_mapped = 0;
		chunks.assign(sz / kChunk, nullptr);
		// (devices that can read this device's memory over xGMI get access to the range as well: the peers of a cooperative accelerator
		// build copy regions out of it, bhip_team_share)
		peers.clear();
		int n_dev = 0;
		if (bhip_vmm_peer_access().load() && hipGetDeviceCount(&n_dev) == hipSuccess) for (int d = 0; d < n_dev; ++d) { int can = 0; if (d != device && hipDeviceCanAccessPeer(&can, d, device) == hipSuccess && can) peers.push_back(d); }
		(void)hipGetLastError();
		return 0;
	}
	std::vector<int> peers;
	// memory behind the chunks [c0, c1) that have none yet
	int map_chunks(size_t c0, size_t c1) {
		if (!vmm) return fail(BHIP_E_INTERNAL, "map_chunks on a fixed buffer");
		if (c1 > chunks.size()) return fail(BHIP_E_DEVICE, "record area: chunk %zu wanted, %zu reserved", c1, chunks.size());
		hipMemAllocationProp prop = {};
		prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = vmm_device;
		hipMemAccessDesc acc = {};
		acc.location.type = hipMemLocationTypeDevice; acc.location.id = vmm_device; acc.flags = hipMemAccessFlagsProtReadWrite;
		for (size_t c = c0; c < c1; ++c) {
			if (chunks[c]) continue;
			std::lock_guard<std::mutex> lk(bhip_vmm_mutex());
			hipMemGenericAllocationHandle_t hnd;
			hipError_t e = hipMemCreate(&hnd, kChunk, &prop, 0);
			if (e != hipSuccess) return fail(BHIP_E_DEVICE, "hipMemCreate(%zu) with %zu chunks mapped: %s", kChunk, n_mapped, hipGetErrorString(e));
			e = hipMemMap((char *)p + c * kChunk, kChunk, 0, hnd, 0);
			if (e == hipSuccess && !peers.empty()) {
				std::vector<hipMemAccessDesc> all(1 + peers.size(), acc);
				for (size_t k = 0; k < peers.size(); ++k) all[1 + k].location.id = peers[k];
				if (hipMemSetAccess((char *)p + c * kChunk, kChunk, all.data(), all.size()) != hipSuccess) { (void)hipGetLastError(); peers.clear(); }      // (own device only, then)
			}
			if (e == hipSuccess && peers.empty()) e = hipMemSetAccess((char *)p + c * kChunk, kChunk, &acc, 1);
			if (e != hipSuccess) { (void)hipMemRelease(hnd); return fail(BHIP_E_DEVICE, "hipMemMap with %zu chunks mapped: %s", n_mapped, hipGetErrorString(e)); }
			chunks[c] = hnd; ++n_mapped;
		}
		size_t pre = cap / kChunk;
		while (pre < chunks.size() && chunks[pre]) ++pre;
		cap = pre * kChunk;      // (the mapped prefix: what the array may use)
		return 0;
	}
	int grow_to(size_t bytes) {
		if (!vmm) return fail(BHIP_E_INTERNAL, "grow_to on a fixed buffer");
		if (bytes > va_size) return fail(BHIP_E_DEVICE, "record area: %zu bytes wanted, %zu reserved", bytes, va_size);
		return map_chunks(0, (bytes + kChunk - 1) / kChunk);
	}
	// the last `bytes` of the range
	int grow_top_to(size_t bytes) {
		if (!vmm) return fail(BHIP_E_INTERNAL, "grow_top_to on a fixed buffer");
		if (bytes > va_size) return fail(BHIP_E_DEVICE, "record area: %zu bytes wanted at its top, %zu reserved", bytes, va_size);
		return map_chunks(chunks.size() - (bytes + kChunk - 1) / kChunk, chunks.size());
	}
	// give back every chunk beyond `bytes` (the array was mapped ahead of knowing its size)
	void shrink_to(size_t bytes) {
		if (!vmm) return;
		const size_t keep = (bytes + kChunk - 1) / kChunk;
		std::lock_guard<std::mutex> lk(bhip_vmm_mutex());
		for (size_t c = keep; c < chunks.size(); ++c) if (chunks[c]) {
			(void)hipMemUnmap((char *)p + c * kChunk, kChunk); (void)hipMemRelease(chunks[c]);
			chunks[c] = nullptr; --n_mapped;
		}
		if (cap > keep * kChunk) cap = keep * kChunk;
	}
	void release() {
		if (vmm) {
			std::lock_guard<std::mutex> lk(bhip_vmm_mutex());
			for (size_t c = 0; c < chunks.size(); ++c) if (chunks[c]) { (void)hipMemUnmap((char *)p + c * kChunk, kChunk); (void)hipMemRelease(chunks[c]); }
			chunks.clear(); n_mapped = 0;
			if (p) (void)hipMemAddressFree(p, va_size);
			vmm = false; va_size = 0;
		} else if (p) (void)hipFree(p);
		p = nullptr; cap = 0;
	}
	template <class T> T *as() const { return (T *)p; }
};

static const int kClasses[] = {2, 4, 6, 8, 10, 16, 32, 128};      // (the last: 1 025 .. BHIP_MAX_QLEN symbols, as many words as the lane's longest query needs)
static const int kNumClasses = BHIP_N_CLASSES;
static inline int class_of_len(uint32_t len) {
	for (int i = 0; i < kNumClasses - 1; ++i) if (len <= 32u * kClasses[i]) return i;
	return kNumClasses - 1;      // 1 025 .. BHIP_MAX_QLEN symbols: a class of their own (round 6: one such read used to drag every 513 .. 1 024-symbol query of its lane into k_myers_long)
}
static inline int class_words(int cls, uint32_t maxlen) { return cls == kNumClasses - 1 ? (int)((std::max<uint32_t>(maxlen, 1025u) + 31u) / 32u) : kClasses[cls]; }

// device-side counters, one block copied back per call
struct Counters {
	uint32_t n_cand, n_raw, n_out, n_wide, err, pad0;
	uint32_t n_cand_cls[8];
	uint32_t n_wins_cls[8];
	uint32_t n_fb, n_fb2;      // queries that overflowed the first / the second (largest tables) pass of the counting-filter prefilter
	uint32_t n_tasks_cls[8];
	uint32_t n_tasks2_cls[8];  // deferred lane tasks (lower bound above the query's best bound)
	uint32_t n_tasks2k_cls[8]; // ... of which kept by k_task_filter
	uint32_t n_wins2_cls[8];   // windows flagged by the second sweep
	uint32_t n_rs[12];         // re-scorer buckets: hits per band-width class
	unsigned long long wcol_sum, tcol_sum, unit_sum;
	unsigned long long col_sum, qlen_sum, ent_read, scratch_used;
	unsigned long long surv_sum;           // list records that passed the counting filter (k_prefilter_cf)
	uint32_t win_class_seen[4];            // [c] != 0: a window of band class c >= 1 was flagged in this call (kernels of unused classes return at once)
};

// One staged batch.  Query symbols with code 0 (anything outside the IUPAC nucleotide alphabet) cost 255 against every
// reference symbol (burst.c:170-190): such a symbol can only be aligned opposite a gap, so the edit distance of the query is
// (number of such symbols) + edit distance of the query without them, end columns unchanged.  When a staged batch holds any,
// the SEARCH kernels (seeds, prefilter, profiles, sweeps) work on a second view of the batch with those symbols removed and
// the budgets reduced (qcodes_s ...); k_junk_adjust adds the counts back before the re-scorer, which works on the original
// queries with the real cost table.  Without such symbols the search view is the batch itself.
struct StageSlot {
	DBuf qcodes, qcodes4, qlen16, qoff, qemac, qsix, qrc, qflags, qmap, off_raw, plan, qpack, key, key_sorted, idx, idx_sorted, sort_tmp, info;
	DBuf qcodes_s, qoff_s, qemac_s, qpack_s, nx, nx_six;
	BhipStageInfo *info_pinned = nullptr;
	hipEvent_t ev_begin = nullptr, ev_done = nullptr, ev_copied = nullptr;      // staging: start, end, and the point between the copies and the routing kernels
	int state = 0;                        // 0 empty, 1 staged (not aligned yet), 2 active (aligned; may be run again)
	uint64_t seq = 0;
	bool resolved = false;                // routing read back, lists assigned
	bool st_valid = false, st_has_six = false, st_has_rc = false, st_has_junk = false, has_flags = false, has_qmap = false;
	uint32_t st_nq = 0, st_nshared = 0, st_maxlen = 0, st_maxE = 0, st_lanes = 1;
	float st_ms_h2d = 0, st_ms_copy = 0, st_ms_route = 0;
	std::vector<BhipQuerySpan> spans;     // the caller's arrays (valid until the batch has been aligned): the host pass reads them
	const uint32_t *six_explicit = nullptr;
	uint32_t npf[16][BHIP_N_CLASSES], nex[16][BHIP_N_CLASSES], maxE[16][BHIP_N_CLASSES], maxwords[16][BHIP_N_CLASSES], qlist_off[16][BHIP_N_CLASSES], maxlen_lane[16], n_entries_lane[16];
	uint64_t seed_words[16][BHIP_N_CLASSES];
	void release_all() {
		DBuf *b[] = {&qcodes, &qcodes4, &qlen16, &qoff, &qemac, &qsix, &qrc, &qflags, &qmap, &off_raw, &plan, &qpack, &key, &key_sorted, &idx, &idx_sorted,
			&sort_tmp, &info, &qcodes_s, &qoff_s, &qemac_s, &qpack_s, &nx, &nx_six};
		for (DBuf *x : b) x->release();
		if (info_pinned) { (void)hipHostFree(info_pinned); info_pinned = nullptr; }
		if (ev_begin) { (void)hipEventDestroy(ev_begin); ev_begin = nullptr; }
		if (ev_done) { (void)hipEventDestroy(ev_done); ev_done = nullptr; }
		if (ev_copied) { (void)hipEventDestroy(ev_copied); ev_copied = nullptr; }
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
	bool fb_dirty = false;                 // the clump-level prefilter of an earlier class of this call left its overflow count in the shared counter
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
	DBuf qmeta_c[2][kNumClasses];          // (query, length | budget << 16, shared slot) per list position, by batch parity: the prefix sweeps of batch k read theirs while the seeds of batch k + 1 are written
	uint64_t qmeta_seq[2][kNumClasses] = {};   // batch the entries belong to (+1; 0 = none)
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
__global__ void k_set_rank_ptrs(SharedCtr *sc, uint32_t *cnt, uint32_t *rank);
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
	DBuf ref_lane, ref_off, clump_len, lut;           // ref_lane: [clump][lane][32-column chunk][16 B], each lane contiguous inside its clump's area
	BhipMatchMask mm;
	BhipAlt alt;                  // compatible bases of every query symbol code (from the cost table): which ambiguous query words can vote through expansions
	bool has_acx = false; int K = 0;
	int acx_z = 0;                // N penalised (the cost table's N-against-N entry): what an accelerator built on the device expands
	// accelerator: offset lines + 4-byte (clump, lane-set code) records (bhip_internal.h); entry numbers start at acx_bias
	// (0, or the test hook BHIP_TEST_ENTRY_BIAS that pushes a small database's offsets beyond 2^32)
	DBuf acx_lines, acx_rec, bad; uint32_t n_bad = 0; uint64_t n_ent = 0, acx_bias = 0;
	BhipAcxView acx_view() const {
		BhipAcxView v; v.lines = acx_lines.as<uint4>();
		v.rec = acx_rec.as<uint32_t>() - acx_bias; return v;
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
	// BEST on the device (bhip_align_staged with all_hits = BHIP_HITS_BEST): RefIxSrt per reference, the per-entry minimum key of a batch
	DBuf ref_order, best_key; uint32_t n_order = 0;
	uint32_t *nsel_pinned = nullptr;      // read-back of the number of selected records
	bool res_sel = false; uint32_t res_n_raw = 0;      // the resident records of a batch that did not fit the caller's buffer: selected? how many before the selection?
	SharedCtr *hsc_pinned = nullptr;      // read-back of shared_ctr behind the chain (pinned: no blocking copy on the way out of a batch)
	const uint8_t *s_codes() const { return cur->st_has_junk ? cur->qcodes_s.as<uint8_t>() : cur->qcodes.as<uint8_t>(); }
	const uint64_t *s_off() const { return cur->st_has_junk ? cur->qoff_s.as<uint64_t>() : cur->qoff.as<uint64_t>(); }
	const uint16_t *s_emac() const { return cur->st_has_junk ? cur->qemac_s.as<uint16_t>() : cur->qemac.as<uint16_t>(); }
	const uint32_t *s_pack() const { return cur->st_has_junk ? cur->qpack_s.as<uint32_t>() : cur->qpack.as<uint32_t>(); }
	DBuf sort_keys, sort_keys2, sort_idx, sort_tmp, out_sorted, out_sorted2, sort_scratch;   // sort_keys / sort_keys2: per-query record counts / offsets; sort_idx: rank of a record inside its query
	uint64_t out_cap = 1 << 20;
	std::vector<uint32_t> h_clump_len;
	BhipStats stats;
	std::vector<Lane *> lanes;
	int opt_two_stage = 1;        // 1 = prefix filter + windowed full-length stage when it pays, 0 = always the one-stage sweep
	int opt_prefilter_stride = 0; // 0 = automatic sparse seeds, s > 0 = every s-th word (1 = the reference's scheme)
	int opt_lanes = 1;            // sub-pipelines per staged batch (the stage kernels fill the chip on their own; > 1 only helps small batches)
	int opt_sweep_blocks = 8;     // 256-thread blocks per CU of the column-sweep kernels
	uint32_t peq_rows = 16;       // rows of a match profile that are built: 5 (pad, A, C, G, T) when no reference holds another symbol
	int opt_oversub = 2;          // blocks launched per resident block slot of the per-item kernels (prefix tasks, windows, re-scoring)
	int opt_band_blocks = 0;      // 64-thread blocks per CU of k_myers_window_band (0: as many as fit)
	int opt_no_band = 0;          // 1 = every window through the full-column kernel (option "band" 0; the parity tests run both)
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
	int opt_pf_cw = 2;            // the counting filter as: 0 k_prefilter_cf (four queries per wave, 16 lanes each), 1 k_prefilter_cw (one query per wave, list-mask slots),
	                              // 2 k_prefilter_cq (four queries per wave, their record streams walked by the whole wave; plans beyond 16 lists: k_prefilter_cw)
	int opt_pf_bytes = 1;         // byte counters (twice as many) for queries whose record stream is at most 255 records
	int opt_pf_rb = 0;            // 64-record blocks per query the counting-filter kernel fetches a quad ahead and keeps in registers (0 = from the workload: 2, 3 or 4)
	int opt_seed_ahead = 1;       // seed lookups of the next staged batch run while the current one is swept
	int opt_seed_ahead_blocks = 2; // 256-thread blocks per CU of a seed kernel that runs ahead (0 = one block per 256 lookups, as in place); 2: +2.3 % on the bench
	int opt_peq_ahead_blocks = 16; // 256-thread blocks per CU of a profile build that runs ahead
	void (*enqueued_hook)(void *) = nullptr; void *enqueued_ctx = nullptr;      // bhip_set_enqueued_hook
	int opt_seed_min_need = -1;   // the longest lists of a query's sampled words are left out while its guaranteed count stays >= this (0 = keep every list, -1 = 3 when the rule of seed_min_need_for says it pays)
	int opt_seed_drop_len = 8;    // ... lists shorter than this are always kept (leaving them out saves nothing and costs selectivity)
	double acx_wmean = 0.0;       // occurrence-weighted mean .acx list length
};

static inline float ev_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; }

// ---- host functions shared between the files of the library ----
int  ensure_lanes(Handle *h, uint32_t n);                                    // bhip_init.hip
uint32_t make_seed_plan(const uint8_t *s, uint32_t len, uint32_t E, uint32_t K, int stride_opt, const BhipAlt &A);   // bhip_stage.hip
int  slot_init(StageSlot *S);
int  resolve_slot(Handle *h, StageSlot *S);
void apply_slot(Handle *h, StageSlot *S);
void lane_capacity_floor(Handle *h, Lane *L, uint64_t n);
int  bhip_load_accelerator(Handle *h, const uint32_t *acx_lens, const void *acx_lists, int acx_fmt, int K, const uint32_t *badlist, uint32_t n_bad);   // bhip_acx.hip
int  bhip_build_accelerator(Handle *h, int K, int z);
#endif
