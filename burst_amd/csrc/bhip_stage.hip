// burst_amd/csrc/bhip_stage.hip -- staging of query batches (include/burst_hip.h: bhip_stage_spans, bhip_stage_queries): copies
// of the caller's arrays, device-side routing (length classes, prefilter / exhaustive route, seed plans, per-list counts), the
// host pass for batches with symbols outside the alphabet, and the page-locked host memory helpers.
#include "bhip_handle.h"

// Seed plan of one query entry (bhip_internal.h: stride | need << 8 | x << 24 | used << 28), need = 0 when no stride guarantees a
// surviving word (the caller then aligns the entry against every clump).  stride_opt > 0 forces the stride.  The same choice as
// bhip_seed_plan (k_route, bhip_kernels.hip).
uint32_t make_seed_plan(const uint8_t *s, uint32_t len, uint32_t E, uint32_t K, int stride_opt, const BhipAlt &A) {
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
	uint32_t xk = 0, usedk = 0;      // expansions: with non-overlapping words (stride K) only
	auto sym = [&](uint32_t i) -> uint32_t { return s[i]; };
	auto need_of = [&](uint32_t st) -> int {
		uint32_t W = 0;
		if (clean) W = (len - K) / st + 1;
		else if (st == K) { uint32_t ws; bhip_expand_walk(sym, K, (len - K) / K + 1, A, ws, xk, usedk); W = ws + xk; }
		else for (uint32_t p = 0; p < npos; p += st) W += valid[p];
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
	const bool ex = !clean && best_s == K && best_n > 0 && xk > 0;
	return (best_s & 255u) | ((uint32_t)best_n << 8) | (ex ? xk << 24 | usedk << 28 : 0u);
}

// ---- staged batches -------------------------------------------------------------------------------------------------
// A batch is staged into one of two slots.  Everything is enqueued on stage_stream and nothing is waited for: the copies
// of the caller's arrays, k_span_fill (batch offsets, shared slots, reported query numbers), k_pack_queries, k_route
// (class / lane / route / seed plan per entry, per-list counts) and a stable radix sort of the entry numbers by list key.
// The batch that is being aligned meanwhile uses the other slot and other streams.  resolve_slot() -- called when the
// batch is about to be aligned -- waits for the slot's event, reads the routing summary from pinned memory and, for the
// rare batch with query symbols of code 0, runs the host pass that builds the search view.
int slot_init(StageSlot *S) {
	if (S->ev_done) return 0;
	if (hipEventCreate(&S->ev_begin) != hipSuccess || hipEventCreate(&S->ev_done) != hipSuccess || hipEventCreate(&S->ev_copied) != hipSuccess) return fail(BHIP_E_DEVICE, "hipEventCreate failed");
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
	bool any_rc = false, any_flags = false, all_flags = true, any_qbase = false, all_packed = true, all_packed2 = true, all_len = true;
	for (uint32_t k = 0; k < n_spans; ++k) {
		const BhipQuerySpan &sp = spans[k];
		if (!sp.n) continue;
		if (!sp.codes || !sp.off || !sp.emac) return fail(BHIP_E_ARG, "null query arrays");
		if (share_by_position && sp.n > n_shared) return fail(BHIP_E_ARG, "span %u has %u entries for %u shared slots", k, sp.n, n_shared);
		n_q64 += sp.n; nb += sp.off[sp.n] - sp.off[0];
		all_packed &= sp.codes4 != nullptr; all_packed2 &= sp.codes2 != nullptr; all_len &= sp.len != nullptr;
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
	if (all_packed2) all_packed = false;      // four symbols per byte beat two
	if ((all_packed || all_packed2) && (rc = S->qcodes4.reserve(nb / 2 + 2 * (size_t)n_spans + 64))) return rc;
	if (all_len && (rc = S->qlen16.reserve(((size_t)n_q + n_spans + 1) * 2 + 64))) return rc;
	if ((rc = S->qcodes.reserve(nb + 2 * (size_t)n_spans + 64)) || (rc = S->qoff.reserve(((size_t)n_q + 1) * 8)) || (rc = S->qemac.reserve(((size_t)n_q + 1) * 2)) ||
	    (rc = S->qsix.reserve(((size_t)n_q + 1) * 4)) || (rc = S->qrc.reserve((size_t)n_q + 1)) || (rc = S->qflags.reserve((size_t)n_q + 1)) ||
	    (rc = S->qmap.reserve(((size_t)n_q + 1) * 4)) || (rc = S->off_raw.reserve(((size_t)n_q + n_spans + 1) * 8)) || (rc = S->plan.reserve((size_t)n_q * 4 + 16)) ||
	    (rc = S->qpack.reserve((size_t)n_q * qw * 4 + 64)) || (rc = S->key.reserve((size_t)n_q + 16)) || (rc = S->key_sorted.reserve((size_t)n_q + 16)) ||
	    (rc = S->idx.reserve((size_t)n_q * 4 + 16)) || (rc = S->idx_sorted.reserve((size_t)n_q * 4 + 16))) return rc;
	HIPCHK(hipEventRecord(S->ev_begin, st));
	uint32_t ebase = 0; uint64_t pos = 0, pos4 = 0;      // pos: symbols of the batch so far; pos4: nibbles (or 2-bit symbols) of the packed staging area
	for (uint32_t k = 0; k < n_spans; ++k) {
		const BhipQuerySpan &sp = spans[k];
		if (!sp.n) continue;
		const uint64_t bytes = sp.off[sp.n] - sp.off[0];
		if (all_packed2) {     // four symbols per byte: the span starts on a byte of its own, at the phase (symbol number mod 4) it has in the caller's array
			pos4 = ((pos4 + 3) & ~3ull) + (sp.off[0] & 3ull);
			if (bytes) {
				HIPCHK(hipMemcpyAsync(S->qcodes4.as<uint8_t>() + (pos4 >> 2), sp.codes2 + (sp.off[0] >> 2), ((sp.off[sp.n] + 3) >> 2) - (sp.off[0] >> 2), hipMemcpyHostToDevice, st));
				hipLaunchKernelGGL(k_unpack2, dim3((uint32_t)std::min<uint64_t>((bytes / 4 + 256) / 256, (uint64_t)h->n_cu * 16)), dim3(256), 0, st, S->qcodes4.as<uint8_t>(), pos4, bytes,
					S->qcodes.as<uint8_t>() + pos);
				HIPCHK(hipGetLastError());
			}
			pos4 += bytes;
		} else if (all_packed) {      // in the staging area the span starts on a byte of its own, at the nibble parity it has in the caller's array
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
		if (all_len) {         // lengths up (2 bytes per entry), offsets by a prefix sum on the device: raw[0 .. n] relative to the span's start
			uint16_t *dl = S->qlen16.as<uint16_t>() + ebase + k;
			HIPCHK(hipMemcpyAsync(dl, sp.len, (size_t)sp.n * 2, hipMemcpyHostToDevice, st));
			HIPCHK(hipMemsetAsync(dl + sp.n, 0, 2, st));
			auto widen = [] __host__ __device__(uint16_t v) -> unsigned long long { return (unsigned long long)v; };
			hipcub::TransformInputIterator<unsigned long long, decltype(widen), const uint16_t *> it(dl, widen);
			size_t tb = 0;
			HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, it, (unsigned long long *)raw, (int)(sp.n + 1), st));
			if ((rc = S->sort_tmp.reserve(tb + 16))) return rc;
			HIPCHK(hipcub::DeviceScan::ExclusiveSum(S->sort_tmp.p, tb, it, (unsigned long long *)raw, (int)(sp.n + 1), st));
		} else HIPCHK(hipMemcpyAsync(raw, sp.off, ((size_t)sp.n + 1) * 8, hipMemcpyHostToDevice, st));
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
	HIPCHK(hipEventRecord(S->ev_copied, st));      // (ms_stage_copy ends / ms_stage_route starts here)
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
		S->info.as<BhipStageInfo>(), h->alt);
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
			const uint32_t lc = l * BHIP_N_CLASSES + (uint32_t)c;
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
				plan[i] = make_seed_plan(codes_i, (uint32_t)len, E_i, (uint32_t)h->K, h->opt_prefilter_stride, h->alt);
				if (BHIP_PLAN_NEED(plan[i]) == 0) ex = 1;           // no word is guaranteed to survive: exhaustive (burst.c:3130-3131 does the same for "bad" queries)
			}
			const size_t lc = (size_t)l * kNumClasses + cls;
			P.lists[lc * 2 + ex].push_back(i);
			P.maxE[lc] = std::max<uint32_t>(P.maxE[lc], E_i);
			if (!ex && len >= (uint64_t)h->K) {
				const uint32_t nwd = (uint32_t)((len - h->K) / (plan[i] & 255u) + 1) + BHIP_PLAN_USED(plan[i]);      // (+ the slots of expanded words)
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
		for (const Part &P : parts) { sorted.insert(sorted.end(), P.lists[k].begin(), P.lists[k].end()); I.count[(l * BHIP_N_CLASSES + c) * 2 + ex] += (uint32_t)P.lists[k].size(); }
		if (!ex) for (const Part &P : parts) {
			I.maxE[l * BHIP_N_CLASSES + c] = std::max(I.maxE[l * BHIP_N_CLASSES + c], P.maxE[lc]); I.maxwords[l * BHIP_N_CLASSES + c] = std::max(I.maxwords[l * BHIP_N_CLASSES + c], P.maxwords[lc]);
			I.seed_words[l * BHIP_N_CLASSES + c] += P.seed_words[lc];
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

int resolve_slot(Handle *h, StageSlot *S) {
	if (S->resolved) return 0;
	if (!S->st_nq) { S->resolved = true; S->st_valid = true; S->st_lanes = 0; return 0; }
	HIPCHK(hipEventSynchronize(S->ev_done));
	const BhipStageInfo &I = *S->info_pinned;
	if (I.err == 1) return fail(BHIP_E_QUERYLEN, "query %u has %u symbols (max %d)", I.err_i, I.err_len, BHIP_MAX_QLEN);
	if (I.err == 2) return fail(BHIP_E_ARG, "q_six[%u] out of range", I.err_i);
	S->st_ms_h2d = ev_ms(S->ev_begin, S->ev_done);
	S->st_ms_copy = ev_ms(S->ev_begin, S->ev_copied); S->st_ms_route = ev_ms(S->ev_copied, S->ev_done);
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
void lane_capacity_floor(Handle *h, Lane *L, uint64_t n) {
	L->task_cap = std::max<uint64_t>(L->task_cap, 3 * n + 4096); L->win_cap = std::max<uint64_t>(L->win_cap, 2 * n + 4096);
	L->raw_cap = std::max<uint64_t>(L->raw_cap, 2 * n + 4096); L->cand_cap = std::max<uint64_t>(L->cand_cap, n / 4 + 4096);
	h->out_cap = std::max<uint64_t>(h->out_cap, 2 * n + 4096);
}

// the slot becomes the batch the alignment kernels work on
void apply_slot(Handle *h, StageSlot *S) {
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
