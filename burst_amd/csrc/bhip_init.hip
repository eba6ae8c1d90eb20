// burst_amd/csrc/bhip_init.hip -- life cycle of a device handle (include/burst_hip.h): database upload (the reference layouts,
// the accelerator as 5-byte records + 64-byte offset lines, per-entry lane masks), options, statistics, destruction.
#include "bhip_handle.h"

static thread_local char g_err[512] = "";

// *flag != 0 afterwards iff some 4-bit symbol of the packed references (two per byte, as in the .edx) is not pad, A, C, G or T
__global__ void k_ref_symbol_scan(const uint32_t *__restrict__ w, uint64_t n_words, uint32_t *__restrict__ flag) {
	uint32_t any = 0;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t d = w[i];
		any |= (d | ((d << 1) & ((d << 2) | (d << 3)))) & 0x88888888u;      // code >= 5: bit 3, or bit 2 with bit 1 or bit 0
	}
	if (any) *flag = 1u;
}
int bhip_fail_msg(int code, const char *fmt, ...) {
	va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
	return code;
}

extern "C" const char *bhip_last_error(void) { return g_err; }
extern "C" int bhip_abi_version(void) { return BHIP_ABI_VERSION; }
std::atomic<int> &bhip_vmm_peer_access_flag() { return bhip_vmm_peer_access(); }      // (for bhip_comm.hip, which does not see the handle's header)

static void lane_destroy(Lane *L) {
	if (!L) return;
	DBuf *all[] = {&L->peq, &L->peqp, &L->cand, &L->candcnt, &L->wins, &L->raw, &L->wide, &L->scratch, &L->fb_list, &L->gcnt, &L->counters, &L->tasks, &L->tasks2, &L->tasks2k, &L->wins2, &L->rs_lists};
	for (DBuf *b : all) b->release();
	L->peq_alt.release(); L->peqp_alt.release();
	for (auto &e : L->ev_peq_alt) if (e) (void)hipEventDestroy(e);
	for (auto &e : L->ev_peq_cur) if (e) (void)hipEventDestroy(e);
	for (DBuf &b : L->ranges_c) b.release();
	for (DBuf &b : L->hdr_c) b.release();
	for (auto &pb : L->qmeta_c) for (DBuf &b : pb) b.release();
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
	if (h->nsel_pinned) (void)hipHostFree(h->nsel_pinned);
	for (Lane *L : h->lanes) lane_destroy(L);
	DBuf *all[] = {&h->ref_lane, &h->ref_off, &h->clump_len, &h->lut, &h->acx_lines, &h->acx_rec, &h->bad,
		&h->best, &h->out, &h->shared_ctr, &h->mins, &h->pairs, &h->sort_keys, &h->sort_keys2, &h->sort_idx,
		&h->sort_tmp, &h->out_sorted, &h->out_sorted2, &h->sort_scratch, &h->ref_order, &h->best_key};
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


extern "C" int bhip_init(int device, const void *edx_packed, const uint32_t *clump_len, uint32_t n_clumps, uint32_t tot_refs,
                         const uint32_t *acx_lens, const void *acx_lists, int acx_fmt, int K,
                         const uint32_t *badlist, uint32_t n_bad, const uint8_t score_lut[256], int xalpha, void **handle) {
	if (!handle) return fail(BHIP_E_ARG, "handle is NULL");
	*handle = nullptr;
	if (xalpha) return fail(BHIP_E_ARG, "xalpha (-x) databases are not supported on the device");
	if (!edx_packed || !clump_len || !n_clumps || !score_lut) return fail(BHIP_E_ARG, "empty database");
	if (acx_lens && (K < 4 || K > 15 || !acx_lists || (acx_fmt != 0 && acx_fmt != 1)))
		return fail(BHIP_E_ARG, "bad accelerator arguments (K=%d fmt=%d)", K, acx_fmt);
	if (!acx_lens && K && (K < 4 || K > 15)) return fail(BHIP_E_ARG, "bad accelerator word length K=%d", K);
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
		// the bases among A, C, G, T (codes 1..4) that cost nothing against query code a
		uint8_t n = 0, bases = 0;
		for (int b = 1; b <= 4; ++b) if (m & (1u << b)) { bases |= (uint8_t)((b - 1) << (2 * n)); ++n; }
		h->alt.n[a] = n; h->alt.base[a] = bases;
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
		INITRC(h->ref_lane.reserve_exact(dst_off[n_clumps] * 256 + 256));
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
			h->clump_len.as<uint32_t>(), h->ref_off.as<uint64_t>(), n_clumps, h->ref_lane.as<uint4>());
		INITCHK(hipGetLastError());
		// any reference symbol beyond A/C/G/T?  (decides how many rows of the match profiles are built per batch)
		DBuf d_flag;
		INITRC(d_flag.reserve(sizeof(uint32_t)));
		INITCHK(hipMemsetAsync(d_flag.p, 0, sizeof(uint32_t), h->stream));
		hipLaunchKernelGGL(k_ref_symbol_scan, dim3((uint32_t)h->n_cu * 8), dim3(256), 0, h->stream, d_src.as<uint32_t>(), src_off[n_clumps] * 4, d_flag.as<uint32_t>());
		INITCHK(hipGetLastError());
		uint32_t any_other = 1;
		INITCHK(hipMemcpyAsync(&any_other, d_flag.p, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
		INITCHK(hipStreamSynchronize(h->stream));
		h->peq_rows = any_other ? 16u : 5u;
		d_src.release(); d_srcoff.release(); d_flag.release();
	}
	if (getenv("BHIP_DEBUG")) { size_t f_ = 0, t_ = 0; if (hipMemGetInfo(&f_, &t_) == hipSuccess) fprintf(stderr, "[bhip] references on the device (%.2f GB, upload buffer released): %.2f GB of the device's %.2f free\n", h->ref_lane.cap / 1e9, f_ / 1e9, t_ / 1e9); }
	// accelerator: from the file's tables, or -- acx_lens == NULL and K given -- built here from the references alone
	h->acx_z = score_lut[16 * 5 + 5] != 0;      // N penalised (burst.c:164, -y clears it): N costs 1 even against N
	if (acx_lens) INITRC(bhip_load_accelerator(h, acx_lens, acx_lists, acx_fmt, K, badlist, n_bad));
	else if (K) INITRC(bhip_build_accelerator(h, K, h->acx_z));
	INITRC(ensure_lanes(h, 1));
	// BHIP_OPTS="name=value,name=value": tuning options for callers that have no other way to pass them (A/B runs of the command line, the
	// tests).  Checked BEFORE the upload (bhip_opts_check below: a misspelt name must not cost a database build); here an entry that does
	// not parse or that the library does not know is skipped with a warning.
	if (const char *ev = getenv("BHIP_OPTS")) {
		std::string all(ev);
		for (size_t a = 0; a < all.size();) {
			size_t b = all.find(',', a); if (b == std::string::npos) b = all.size();
			const std::string kv = all.substr(a, b - a); a = b + 1;
			const size_t eq = kv.find('=');
			char *end = nullptr;
			const long long v = eq == std::string::npos ? 0 : strtoll(kv.c_str() + eq + 1, &end, 0);
			if (eq == std::string::npos || eq == 0 || !end || end == kv.c_str() + eq + 1 || *end) { fprintf(stderr, "[bhip] BHIP_OPTS: '%s' is not name=<integer>: ignored\n", kv.c_str()); continue; }
			if (bhip_set_option(h, kv.substr(0, eq).c_str(), v)) fprintf(stderr, "[bhip] BHIP_OPTS: %s: ignored\n", bhip_last_error());
		}
	}
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
	if (!strcmp(name, "band_blocks")) { if (value < 0 || value > 1024) return fail(BHIP_E_ARG, "band_blocks must be 0 .. 1024"); h->opt_band_blocks = (int)value; return BHIP_OK; }
	if (!strcmp(name, "oversub")) { if (value < 1 || value > 16) return fail(BHIP_E_ARG, "oversub must be 1 .. 16"); h->opt_oversub = (int)value; return BHIP_OK; }
	if (!strcmp(name, "band")) { h->opt_no_band = value == 0; return BHIP_OK; }
	if (!strcmp(name, "prune")) { h->opt_prune = value != 0; return BHIP_OK; }
	if (!strcmp(name, "rescore_reg")) { h->opt_rescore_reg = value != 0; return BHIP_OK; }
	if (!strcmp(name, "seed_min_need")) { if (value < -1 || value > 255) return fail(BHIP_E_ARG, "seed_min_need must be -1 (by rule) or 0 .. 255"); h->opt_seed_min_need = (int)value; return BHIP_OK; }
	if (!strcmp(name, "seed_drop_len")) { if (value < 0 || value > (1 << 24)) return fail(BHIP_E_ARG, "seed_drop_len must be 0 .. 2^24"); h->opt_seed_drop_len = (int)value; return BHIP_OK; }
	if (!strcmp(name, "prefilter_waves")) { if (value < 0 || value > 16) return fail(BHIP_E_ARG, "prefilter_waves must be 0 .. 16"); h->opt_pf_waves = (int)value; return BHIP_OK; }
	if (!strcmp(name, "prefilter_algo")) { if (value < -1 || value > 1) return fail(BHIP_E_ARG, "prefilter_algo must be -1, 0 or 1"); h->opt_pf_algo = (int)value; return BHIP_OK; }
	if (!strcmp(name, "prefilter_table")) { if (value != 0 && (value < 9 || value > 11)) return fail(BHIP_E_ARG, "prefilter_table must be 0, 9, 10 or 11"); h->opt_pf_table = (int)value; return BHIP_OK; }
	if (!strcmp(name, "prefilter_cw")) { if (value < 0 || value > 2) return fail(BHIP_E_ARG, "prefilter_cw must be 0, 1 or 2"); h->opt_pf_cw = (int)value; return BHIP_OK; }
	if (!strcmp(name, "prefilter_bytes")) { h->opt_pf_bytes = value != 0; return BHIP_OK; }
	if (!strcmp(name, "prefilter_rb")) { if (value != 0 && (value < 2 || value > 4)) return fail(BHIP_E_ARG, "prefilter_rb must be 0, 2, 3 or 4"); h->opt_pf_rb = (int)value; return BHIP_OK; }
	if (!strcmp(name, "lanes")) {
		if (value < 1 || value > BHIP_MAX_LANES) return fail(BHIP_E_ARG, "lanes must be 1 .. %d", BHIP_MAX_LANES);
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

int ensure_lanes(Handle *h, uint32_t n) {
	while (h->lanes.size() < n) { Lane *L = nullptr; int rc = lane_create(h, &L); if (rc) return rc; h->lanes.push_back(L); }
	return 0;
}
