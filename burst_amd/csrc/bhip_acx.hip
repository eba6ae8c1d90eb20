// burst_amd/csrc/bhip_acx.hip -- the accelerator (.acx, burst.c:3535-3594) as it lives in HBM (BhipAcxView, bhip_internal.h):
// 4-byte (clump, lane-set code) records in word order + one 64-byte offset line per 14 words.  Three ways to get there:
//   bhip_load_accelerator   from the file's Lens[4^K] + packed lists (read_accelerator, burst.c:3535-3594): decoded on the device,
//                           lane masks derived from the references (build_lane_masks);
//   bhip_build_accelerator  from the references alone, on the device (make_accelerator, burst.c:3304-3532): every K-mer of every
//                           lane -- expanded over IUPAC codes, clumps whose expansion exceeds the reference's budget on the BadList --
//                           as (word, clump, lane) tuples, radix-sorted and folded to one record per (word, clump) with its lane set:
//                           the same entries in the same order as the file, without the file;
//   bhip_acx_export         back to the host as Lens[4^K] + clump ids (+ masks, BadList), e.g. to write the .acx.
#include "bhip_handle.h"
#include "bhip_acx_words.h"

// ------------------------------------------------------------------------------------------------
// .acx offsets: Lens[4^K] (burst.c:3558) -> exclusive prefix inside each block of 256 words (`delta`) and the block sums
// (scanned to 64-bit block bases by the caller).  WHAT = 0: list lengths (entries); 1: bytes of the packed SMALL lists
// (pairs of 20-bit ids in 5 bytes with a 3-byte odd tail, burst.c:3516-3527); 2: bytes of the LARGE lists (3 per id).
// One 256-thread workgroup per block of words.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_acx_offsets(const uint32_t *__restrict__ lens, uint64_t n_words, int what,
                                                     uint32_t *__restrict__ delta, unsigned long long *__restrict__ blocksum) {
	__shared__ uint32_t s_w[4];
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint64_t n_blocks = (n_words + 255) >> 8;
	for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
		const uint64_t w = blk * 256 + tid;
		const uint32_t len = w < n_words ? lens[w] : 0u;
		const uint32_t v = what == 0 ? len : what == 1 ? 5u * (len >> 1) + 3u * (len & 1u) : 3u * len;
		uint32_t ps = v;
		#pragma unroll
		for (uint32_t o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(ps, o); if (lane >= o) ps += t; }
		__syncthreads();
		if (lane == 63) s_w[wv] = ps;
		__syncthreads();
		uint32_t add = 0;
		for (uint32_t k = 0; k < wv; ++k) add += s_w[k];
		if (w < n_words) delta[w] = add + ps - v;
		if (tid == 255) blocksum[blk] = (unsigned long long)add + ps;
	}
}

// .acx offsets as 64-byte lines (BhipAcxView): pass 0 writes the sum of every block of 14 lengths, the caller scans them into
// 64-bit bases (hipCUB), pass 1 writes base + inclusive prefix sums.  One thread per line.
__global__ void k_acx_lines(const uint32_t *__restrict__ lens, uint64_t n_words, int pass, unsigned long long *__restrict__ blk, uint4 *__restrict__ lines) {
	const uint64_t n_lines = (n_words + BHIP_ACX_LINE_WORDS - 1) / BHIP_ACX_LINE_WORDS;
	for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < n_lines; b += (uint64_t)gridDim.x * blockDim.x) {
		uint32_t d[14], run = 0;
		#pragma unroll
		for (uint32_t j = 0; j < 14; ++j) { const uint64_t w = b * BHIP_ACX_LINE_WORDS + j; run += w < n_words ? lens[w] : 0u; d[j] = run; }
		if (pass == 0) { blk[b] = run; continue; }
		const unsigned long long base = blk[b];
		lines[4 * b + 0] = make_uint4((uint32_t)base, (uint32_t)(base >> 32), d[0], d[1]);
		lines[4 * b + 1] = make_uint4(d[2], d[3], d[4], d[5]);
		lines[4 * b + 2] = make_uint4(d[6], d[7], d[8], d[9]);
		lines[4 * b + 3] = make_uint4(d[10], d[11], d[12], d[13]);
	}
}

// ------------------------------------------------------------------------------------------------
// .acx list area -> 4-byte records (24-bit clump id, lane-set code preset to "every lane"), on the device (the packed bytes
// are what is uploaded): SMALL lists are pairs of 20-bit ids in 5 bytes with a 3-byte odd tail (burst.c:3265-3274), LARGE
// lists 3 bytes per id (3245-3248).  One thread per word; `bad` is raised when an id is not a clump of the database.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bhip_rec_store(uint32_t *rec, unsigned long long e, uint32_t clump, uint32_t mask) {
	rec[e] = (clump & 0xFFFFFFu) | bhip_lane_mask_code(mask) << 24;
}
__global__ void k_acx_decode(const uint8_t *__restrict__ lists, const unsigned long long *__restrict__ byte_base, const uint32_t *__restrict__ byte_delta,
                             BhipAcxView acx, uint64_t n_words, int fmt, uint32_t n_clumps, uint32_t *__restrict__ rec, uint32_t *__restrict__ bad) {
	for (uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
		unsigned long long e; uint32_t n;
		bhip_acx_range(acx, (uint32_t)w, e, n);
		if (!n) continue;
		const uint8_t *p = lists + byte_base[w >> 8] + byte_delta[w];
		uint32_t worst = 0;
		if (fmt == 1) {
			for (; n; --n, p += 3) { const uint32_t v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); bhip_rec_store(rec, e++, v, 0xFFFFu); worst = v > worst ? v : worst; }
		} else {
			for (; n >= 2; n -= 2, p += 5) {
				const unsigned long long v = (unsigned long long)p[0] | ((unsigned long long)p[1] << 8) | ((unsigned long long)p[2] << 16) |
				                             ((unsigned long long)p[3] << 24) | ((unsigned long long)p[4] << 32);
				const uint32_t a = (uint32_t)(v & 0xFFFFF), b = (uint32_t)((v >> 20) & 0xFFFFF);
				bhip_rec_store(rec, e++, a, 0xFFFFu); bhip_rec_store(rec, e++, b, 0xFFFFu);
				worst = a > worst ? a : worst; worst = b > worst ? b : worst;
			}
			if (n) { const uint32_t v = ((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16)) & 0xFFFFF; bhip_rec_store(rec, e++, v, 0xFFFFu); worst = v > worst ? v : worst; }
		}
		if (worst >= n_clumps) atomicMax(bad, worst);
	}
}

// ------------------------------------------------------------------------------------------------
// Lane-resolved accelerator.  The .acx says which CLUMPS contain a word; at upload we also work out which of the 16
// LANES of the clump contain it (a 16-bit mask per list entry), so that the prefilter can count seed words per reference
// lane and hand only the lanes that can hold an alignment to the edit-distance kernels (2-3 lanes per candidate clump
// instead of all 16).  k_extract_kmers emits (word << 24 | clump, 1 << lane) for every A/C/G/T-only K-mer of every lane,
// a device radix sort + OR-reduce-by-key gives one mask per (word, clump), and k_attach_masks looks every .acx entry up.
// Entries the extraction does not know (words the reference added by IUPAC expansion, burst.c:3368-3377) get 0xFFFF:
// every lane, i.e. the clump-level behaviour.  Masks only ever add lanes, never drop one, so results are unchanged.
// ------------------------------------------------------------------------------------------------
__global__ void k_extract_kmers(const uint4 *__restrict__ ref, const uint64_t *__restrict__ ref_off, const uint32_t *__restrict__ clump_len,
                                const uint64_t *__restrict__ key_off, uint32_t c0, uint32_t c1, int K,     // clumps [c0, c1): one slice of the database
                                unsigned long long *__restrict__ keys, uint16_t *__restrict__ vals, uint32_t *__restrict__ ambig_lanes) {
	const uint64_t n_threads = (uint64_t)(c1 - c0) * 16;
	const uint32_t wmask = K == 16 ? 0xFFFFFFFFu : ((1u << (2 * K)) - 1u);
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_threads; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t c = c0 + (uint32_t)(i >> 4), z = (uint32_t)(i & 15), L = clump_len[c], nchunks = (L + 31) >> 5;
		const uint4 *rp = ref + ref_off[c] * 16 + (uint64_t)z * nchunks;       // lane-major: [clump][lane][chunk]
		unsigned long long *kout = keys + (key_off[c] - key_off[c0]) + (uint64_t)z * L;
		uint16_t *vout = vals + (key_off[c] - key_off[c0]) + (uint64_t)z * L;
		uint32_t w = 0, run = 0, amb = 0;
		for (uint32_t t = 0; t < nchunks; ++t) {
			const uint4 ch = rp[t];
			const uint32_t dw[4] = {ch.x, ch.y, ch.z, ch.w};
			for (uint32_t k = 0; k < 32; ++k) {
				const uint32_t pos = t * 32 + k;
				if (pos >= L) break;
				const uint32_t sym = (dw[k >> 3] >> (4 * (k & 7))) & 15u;
				amb |= sym > 4u;
				run = (sym - 1u) < 4u ? run + 1 : 0;
				w = ((w << 2) | ((sym - 1u) & 3u)) & wmask;
				// the word ENDING at pos is stored in slot pos (slots 0..K-2 and words with other symbols are invalid)
				kout[pos] = run >= (uint32_t)K ? (((unsigned long long)w << 24) | c) : ~0ull;
				vout[pos] = (uint16_t)(1u << z);
			}
		}
		// a lane with IUPAC / N symbols can match words it does not literally contain (the .acx lists them for the clump
		// through the reference's expansion): such a lane takes part in every entry of its clump
		if (amb) atomicOr(&ambig_lanes[c], 1u << z);
	}
}

__global__ void k_attach_masks(BhipAcxView acx, uint64_t n_words,
                               const unsigned long long *__restrict__ ukeys, const uint16_t *__restrict__ umasks, uint32_t n_unique,
                               const uint32_t *__restrict__ ambig_lanes, uint32_t *__restrict__ rec,    // the code bytes of the records are rewritten
                               uint32_t c0, uint32_t c1) {                                                 // only entries of clumps [c0, c1)
	// one thread per word walks its list (a few entries); the key of an entry is (word, clump), looked up in the folded tuples
	for (uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
		unsigned long long e; uint32_t n;
		bhip_acx_range(acx, (uint32_t)w, e, n);
		for (; n; --n, ++e) {
			const uint32_t ce = bhip_acx_clump(acx.rec, e);
			if (ce < c0 || ce >= c1) continue;
			const unsigned long long key = ((unsigned long long)w << 24) | ce;
			uint32_t a = 0, b = n_unique;
			while (a < b) { const uint32_t mid = a + ((b - a) >> 1); if (ukeys[mid] < key) a = mid + 1; else b = mid; }
			const uint32_t m = ((a < n_unique && ukeys[a] == key) ? (uint32_t)umasks[a] : 0xFFFFu) | (ambig_lanes[ce] & 0xFFFFu);
			bhip_rec_store(rec, e, ce, m);
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Host side.  Temporary device buffers are released when they go out of scope.
// ------------------------------------------------------------------------------------------------
struct DTmp : DBuf { ~DTmp() { release(); } };
#define ARC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// Offset lines, total and statistics from the list lengths on the device (Lens[4^K], burst.c:3558): sums of 14 -> 64-bit bases
// (hipCUB scan) -> lines; occurrence-weighted mean list length (sizes the prefilter's per-query tables) and the longest list.
// (`scratch`: memory the caller took while the device still had room -- acx_lines_scratch_bytes of it: an allocation next to 216 GB of mapped
// records takes half a second each, 1.1 s of a 4.2 s build for these four)
static size_t acx_lines_scratch_bytes(uint64_t nw) { return (size_t)(((nw + BHIP_ACX_LINE_WORDS - 1) / BHIP_ACX_LINE_WORDS + 2) * 16 + ((size_t)64 << 20)); }
static int acx_lines_from_lens(Handle *h, const uint32_t *d_lens, uint64_t nw, uint64_t *tot_out, uint32_t *maxlen_out, DBuf *scratch = nullptr) {
	const uint64_t n_lines = (nw + BHIP_ACX_LINE_WORDS - 1) / BHIP_ACX_LINE_WORDS;
	DTmp d_red, d_tmp, d_lsum, d_lbase;
	const auto t_a0 = std::chrono::steady_clock::now();
	ARC(h->acx_lines.reserve_exact((n_lines + 1) * 64));
	auto to_sq = [] __host__ __device__(uint32_t n) -> double { return (double)n * (double)n; };
	hipcub::TransformInputIterator<double, decltype(to_sq), const uint32_t *> it_sq(d_lens, to_sq);
	size_t tb = 0, tb1 = 0;
	HIPCHK(hipcub::DeviceReduce::Sum(nullptr, tb1, it_sq, (double *)nullptr, (int)nw, h->stream)); tb = std::max(tb, tb1);
	HIPCHK(hipcub::DeviceReduce::Max(nullptr, tb1, d_lens, (uint32_t *)nullptr, (int)nw, h->stream)); tb = std::max(tb, tb1);
	HIPCHK(hipcub::DeviceScan::ExclusiveScan(nullptr, tb1, (unsigned long long *)nullptr, (unsigned long long *)nullptr, hipcub::Sum(), 0ull, (int)(n_lines + 1), h->stream)); tb = std::max(tb, tb1);
	const size_t sum_b = ((n_lines + 2) * 8 + 255) & ~(size_t)255;
	const bool carved = scratch && scratch->p && scratch->cap >= 256 + 2 * sum_b + tb + 16;
	if (carved) {          // (borrowed: the DTmp wrappers must not free them)
		d_red.p = scratch->p; d_lsum.p = (char *)scratch->p + 256; d_lbase.p = (char *)d_lsum.p + sum_b; d_tmp.p = (char *)d_lbase.p + sum_b;
	} else {
		ARC(d_red.reserve(64)); ARC(d_lsum.reserve((n_lines + 2) * 8)); ARC(d_lbase.reserve((n_lines + 2) * 8)); ARC(d_tmp.reserve(tb + 16));
	}
	struct Unborrow { DTmp &a, &b, &c, &d; bool on; ~Unborrow() { if (on) { a.p = b.p = c.p = d.p = nullptr; a.cap = b.cap = c.cap = d.cap = 0; } } } unborrow{d_red, d_lsum, d_lbase, d_tmp, carved};
	if (getenv("BHIP_DEBUG")) fprintf(stderr, "[bhip] offset lines: %.3f s allocating (%s)\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a0).count(), carved ? "scratch taken before the build" : "now");
	unsigned long long *r_tot = d_red.as<unsigned long long>(); double *r_sq = (double *)(r_tot + 1); uint32_t *r_max = (uint32_t *)(r_tot + 2);
	tb1 = tb; HIPCHK(hipcub::DeviceReduce::Sum(d_tmp.p, tb1, it_sq, r_sq, (int)nw, h->stream));
	tb1 = tb; HIPCHK(hipcub::DeviceReduce::Max(d_tmp.p, tb1, d_lens, r_max, (int)nw, h->stream));
	const uint32_t lg = (uint32_t)std::min<uint64_t>((n_lines + 255) / 256, (uint64_t)h->n_cu * 32);
	HIPCHK(hipMemsetAsync(d_lsum.p, 0, (n_lines + 2) * 8, h->stream));
	hipLaunchKernelGGL(k_acx_lines, dim3(lg), dim3(256), 0, h->stream, d_lens, nw, 0, d_lsum.as<unsigned long long>(), h->acx_lines.as<uint4>());
	HIPCHK(hipGetLastError());
	tb1 = tb; HIPCHK(hipcub::DeviceScan::ExclusiveScan(d_tmp.p, tb1, d_lsum.as<unsigned long long>(), d_lbase.as<unsigned long long>(), hipcub::Sum(), (unsigned long long)h->acx_bias, (int)(n_lines + 1), h->stream));
	hipLaunchKernelGGL(k_acx_lines, dim3(lg), dim3(256), 0, h->stream, d_lens, nw, 1, d_lbase.as<unsigned long long>(), h->acx_lines.as<uint4>());
	HIPCHK(hipGetLastError());
	unsigned long long red[3], tot_b = 0;
	HIPCHK(hipMemcpyAsync(red, d_red.p, 24, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipMemcpyAsync(&tot_b, d_lbase.as<unsigned long long>() + n_lines, 8, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	const uint64_t tot = tot_b - h->acx_bias;
	double sq; memcpy(&sq, &red[1], 8);
	uint32_t maxlen; memcpy(&maxlen, &red[2], 4);
	h->acx_wmean = tot ? sq / (double)tot : 0.0;
	if (maxlen > h->n_clumps || maxlen >= (1u << 24)) return fail(BHIP_E_ARG, "an accelerator list has %u entries, the database %u clumps (wrong K for this file?)", maxlen, h->n_clumps);
	if (tot_b >= (1ull << 40)) return fail(BHIP_E_ARG, "accelerator with %llu entries exceeds the 40-bit entry numbers of the device layout", (unsigned long long)tot);
	*tot_out = tot; *maxlen_out = maxlen;
	return 0;
}

static int set_badlist(Handle *h, const uint32_t *badlist, uint32_t n_bad) {
	h->n_bad = n_bad;
	ARC(h->bad.reserve(((size_t)n_bad + 1) * sizeof(uint32_t)));
	for (uint32_t i = 0; i < n_bad; ++i) if (badlist[i] >= h->n_clumps) return fail(BHIP_E_ARG, "BadList entry out of range");
	if (n_bad) HIPCHK(hipMemcpy(h->bad.p, badlist, (size_t)n_bad * sizeof(uint32_t), hipMemcpyHostToDevice));
	return 0;
}

// per-entry lane masks of a LOADED accelerator (see "Lane-resolved accelerator" above).  Skipped (has_masks stays false, clump-level
// behaviour) when the scratch does not fit.
struct BitOrU16 { __host__ __device__ uint16_t operator()(const uint16_t &a, const uint16_t &b) const { return (uint16_t)(a | b); } };
static int build_lane_masks(Handle *h) {
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
	DTmp d_koff, k0, k1, v0, v1, nruns, tmp, amb;
	const uint64_t cap_items = std::min<uint64_t>(slice_items, key_off[nC]);
	ARC(d_koff.reserve((nC + 1) * 8)); ARC(k0.reserve(cap_items * 8)); ARC(k1.reserve(cap_items * 8)); ARC(v0.reserve(cap_items * 2)); ARC(v1.reserve(cap_items * 2));
	ARC(nruns.reserve(16)); ARC(amb.reserve((size_t)nC * 4 + 16));
	HIPCHK(hipMemsetAsync(amb.p, 0, (size_t)nC * 4, h->stream));
	HIPCHK(hipMemcpyAsync(d_koff.p, key_off.data(), (nC + 1) * 8, hipMemcpyHostToDevice, h->stream));
	const int end_bit = 2 * h->K + 24;
	uint32_t n_slices = 0;
	for (uint32_t c0 = 0; c0 < nC;) {
		uint32_t c1 = c0 + 1;
		while (c1 < nC && key_off[c1 + 1] - key_off[c0] <= slice_items) ++c1;
		const uint64_t n_items = key_off[c1] - key_off[c0];
		++n_slices;
		hipLaunchKernelGGL(k_extract_kmers, dim3(std::min<uint32_t>(((c1 - c0) * 16 + 255) / 256, (uint32_t)h->n_cu * 16)), dim3(256), 0, h->stream, h->ref_lane.as<uint4>(),
			h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), d_koff.as<uint64_t>(), c0, c1, h->K, k0.as<unsigned long long>(), v0.as<uint16_t>(), amb.as<uint32_t>());
		HIPCHK(hipGetLastError());
		size_t tb = 0;
		hipcub::DoubleBuffer<unsigned long long> dk(k0.as<unsigned long long>(), k1.as<unsigned long long>());
		hipcub::DoubleBuffer<uint16_t> dv(v0.as<uint16_t>(), v1.as<uint16_t>());
		HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, dv, (int)n_items, 0, end_bit, h->stream));
		ARC(tmp.reserve(tb));
		HIPCHK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, dk, dv, (int)n_items, 0, end_bit, h->stream));
		// invalid slots carry key ~0, which after masking to end_bit sorts last (all ones) -- their run is simply never looked up
		unsigned long long *skeys = dk.Current(); uint16_t *svals = dv.Current();
		unsigned long long *ukeys = dk.Alternate(); uint16_t *umasks = dv.Alternate();
		size_t tb2 = 0;
		HIPCHK(hipcub::DeviceReduce::ReduceByKey(nullptr, tb2, skeys, ukeys, svals, umasks, nruns.as<uint32_t>(), BitOrU16(), (int)n_items, h->stream));
		ARC(tmp.reserve(tb2));
		HIPCHK(hipcub::DeviceReduce::ReduceByKey(tmp.p, tb2, skeys, ukeys, svals, umasks, nruns.as<uint32_t>(), BitOrU16(), (int)n_items, h->stream));
		uint32_t n_unique = 0;
		HIPCHK(hipMemcpyAsync(&n_unique, nruns.p, 4, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		hipLaunchKernelGGL(k_attach_masks, dim3((uint32_t)h->n_cu * 32), dim3(256), 0, h->stream,
			h->acx_view(), (uint64_t)(1ull << (2 * h->K)), ukeys, umasks, n_unique, amb.as<uint32_t>(), (uint32_t *)h->acx_view().rec, c0, c1);
		HIPCHK(hipGetLastError());
		HIPCHK(hipStreamSynchronize(h->stream));
		c0 = c1;
	}
	if (getenv("BHIP_DEBUG")) fprintf(stderr, "[bhip] lane masks: %llu reference positions in %u slice(s)\n", (unsigned long long)key_off[nC], n_slices);
	h->has_masks = true;
	return 0;
}

// The accelerator of a file: Lens[4^K] goes up as it is; offsets, the byte positions of the packed lists, the total and the
// statistics are scans / reductions on the device (at K = 15 the table has 2^30 words); the packed list area goes up as it is
// on disk and is decoded to 5-byte records by the device.
int bhip_load_accelerator(Handle *h, const uint32_t *acx_lens, const void *acx_lists, int acx_fmt, int K, const uint32_t *badlist, uint32_t n_bad) {
	const uint64_t nw = 1ull << (2 * K), nblk = (nw + 255) >> 8;
	DTmp d_lens, d_tmp, d_bsum, d_bdelta, d_bbase, d_lists, d_flag;
	if (const char *ev = getenv("BHIP_TEST_ENTRY_BIAS")) h->acx_bias = strtoull(ev, nullptr, 0);
	ARC(d_lens.reserve(nw * sizeof(uint32_t)));
	ARC(d_bsum.reserve((nblk + 2) * 8));
	ARC(d_bdelta.reserve(nw * sizeof(uint32_t) + 16));
	ARC(d_bbase.reserve((nblk + 2) * 8));
	HIPCHK(hipMemcpyAsync(d_lens.p, acx_lens, nw * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
	uint64_t tot = 0; uint32_t maxlen = 0;
	ARC(acx_lines_from_lens(h, d_lens.as<uint32_t>(), nw, &tot, &maxlen));
	// byte offsets of the packed lists on disk
	const uint32_t og = (uint32_t)std::min<uint64_t>(nblk, (uint64_t)h->n_cu * 16);
	HIPCHK(hipMemsetAsync(d_bsum.p, 0, (nblk + 2) * 8, h->stream));
	hipLaunchKernelGGL(k_acx_offsets, dim3(og), dim3(256), 0, h->stream, d_lens.as<uint32_t>(), nw, acx_fmt == 1 ? 2 : 1, d_bdelta.as<uint32_t>(), d_bsum.as<unsigned long long>());
	HIPCHK(hipGetLastError());
	size_t tb = 0;
	HIPCHK(hipcub::DeviceScan::ExclusiveScan(nullptr, tb, d_bsum.as<unsigned long long>(), d_bbase.as<unsigned long long>(), hipcub::Sum(), 0ull, (int)(nblk + 1), h->stream));
	ARC(d_tmp.reserve(tb + 16));
	HIPCHK(hipcub::DeviceScan::ExclusiveScan(d_tmp.p, tb, d_bsum.as<unsigned long long>(), d_bbase.as<unsigned long long>(), hipcub::Sum(), 0ull, (int)(nblk + 1), h->stream));
	unsigned long long bytes = 0;
	HIPCHK(hipMemcpyAsync(&bytes, d_bbase.as<unsigned long long>() + nblk, 8, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	ARC(h->acx_rec.reserve_exact(tot * BHIP_REC_BYTES + 16));
	ARC(d_lists.reserve(bytes + 16)); ARC(d_flag.reserve(16));
	if (bytes) HIPCHK(hipMemcpyAsync(d_lists.p, acx_lists, bytes, hipMemcpyHostToDevice, h->stream));
	HIPCHK(hipMemsetAsync(d_flag.p, 0, 16, h->stream));
	hipLaunchKernelGGL(k_acx_decode, dim3((uint32_t)h->n_cu * 32), dim3(256), 0, h->stream, d_lists.as<uint8_t>(), d_bbase.as<unsigned long long>(),
		d_bdelta.as<uint32_t>(), h->acx_view(), nw, acx_fmt, h->n_clumps, (uint32_t *)h->acx_view().rec, d_flag.as<uint32_t>());
	HIPCHK(hipGetLastError());
	uint32_t worst = 0;
	HIPCHK(hipMemcpyAsync(&worst, d_flag.p, 4, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(hipStreamSynchronize(h->stream));
	if (worst) return fail(BHIP_E_ARG, "an accelerator entry refers to clump %u >= %u", worst, h->n_clumps);
	d_lists.release(); d_lens.release(); d_bdelta.release(); d_bsum.release(); d_bbase.release();
	if (getenv("BHIP_DEBUG")) fprintf(stderr, "[bhip] accelerator: K=%d, %llu entries (first entry number %llu), %.2f B per entry on the device (records %d B + offsets)\n", K,
		(unsigned long long)tot, (unsigned long long)h->acx_bias, tot ? (double)(tot * BHIP_REC_BYTES + ((nw + BHIP_ACX_LINE_WORDS - 1) / BHIP_ACX_LINE_WORDS) * 64) / (double)tot : 0.0, BHIP_REC_BYTES);
	ARC(set_badlist(h, badlist, n_bad));
	h->has_acx = true; h->K = K; h->n_ent = tot;
	if (!getenv("BHIP_NO_LANE_MASKS")) ARC(build_lane_masks(h));
	return 0;
}

// ------------------------------------------------------------------------------------------------
// Accelerator built on the device from the references alone (make_accelerator, burst.c:3304-3532).
// A list entry (word, clump) exists iff the word is one of the IUPAC expansions (AMBIGS, burst.c:1372-1375) of a K-symbol
// window of one of the clump's lanes; windows with a symbol of code 0 have none, windows with N none when N is penalised
// (burst.c:3368-3374).  The reference first estimates every clump's expansion -- sum over the windows of 3^a (N penalised) or
// "4^a" (its table says 61 for a = 3) where a counts the ambiguous symbols among the K - 1 symbols BEFORE the window's last
// (burst.c:3343-3351: the count is taken before the last symbol is added) -- and puts a clump whose estimate reaches 2^24
// (2^31 - 1 for K = 15) on the BadList instead of indexing it (burst.c:3322, 3351).  Reproduced literally: same entries, same
// BadList.  One thread per (clump, lane), symbol after symbol, the last K symbol codes in one 64-bit register.
// ------------------------------------------------------------------------------------------------
__device__ const unsigned long long c_ipow[2][16] = {
	{1, 4, 16, 61, 256, 1024, 4096, 16384, 65536, 262144, 1048576, 4194304, 16777216, 67108864, 268435456, 1073741824},      // N matches everything (-y)
	{1, 3, 9, 27, 81, 243, 729, 2187, 6561, 19683, 59049, 177147, 531441, 1594323, 4782969, 14348907}};                     // N penalised (default)

// per clump: the reference's expansion estimate (tsum) and the true number of words its ambiguous windows expand to (nexp)
__global__ void k_acx_budget(const uint4 *__restrict__ ref, const uint64_t *__restrict__ ref_off, const uint32_t *__restrict__ clump_len, uint32_t n_clumps, uint32_t tot_refs,
                             int K, int z, unsigned long long *__restrict__ tsum, unsigned long long *__restrict__ nexp) {
	const uint64_t n_threads = (uint64_t)n_clumps * 16;
	const uint32_t AMBIG = 4u + (uint32_t)z;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_threads; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t c = (uint32_t)(i >> 4), zz = (uint32_t)(i & 15);
		if (16ull * c + zz >= tot_refs) continue;                                    // burst.c:3336: lanes beyond the last reference
		const uint32_t L = clump_len[c], nchunks = (L + 31) >> 5;
		const uint4 *rp = ref + ref_off[c] * 16 + (uint64_t)zz * nchunks;
		// the lane's own length: pads (code 0) at the end do not count
		uint32_t ll = 0;
		for (uint32_t t = nchunks; t-- > 0 && !ll;) {
			const uint4 ch = rp[t];
			const uint32_t dw[4] = {ch.x, ch.y, ch.z, ch.w};
			for (int k = 31; k >= 0; --k) if ((dw[k >> 3] >> (4 * (k & 7))) & 15u) { ll = t * 32 + (uint32_t)k + 1; break; }
		}
		if (ll > L) ll = L;
		if (ll < (uint32_t)K) continue;
		unsigned long long win = 0, ts = 0, nx = 0;
		uint32_t asum = 0, run = 0, lit = 0;
		for (uint32_t t = 0; t * 32 < ll; ++t) {
			const uint4 ch = rp[t];
			const uint32_t dw[4] = {ch.x, ch.y, ch.z, ch.w};
			for (uint32_t k = 0; k < 32 && t * 32 + k < ll; ++k) {
				const uint32_t j = t * 32 + k, sym = (dw[k >> 3] >> (4 * (k & 7))) & 15u;
				if (j >= (uint32_t)K - 1) { ts += c_ipow[z ? 1 : 0][asum & 15u]; if (((uint32_t)(win >> (4 * (K - 2))) & 15u) > AMBIG) --asum; }
				if (sym > AMBIG) ++asum;
				run = (sym >= 1u && !(z && sym == 5u)) ? run + 1 : 0;
				lit = (sym - 1u) < 4u ? lit + 1 : 0;
				win = (win << 4) | sym;
				if (run >= (uint32_t)K && lit < (uint32_t)K) nx += amb_product(win, K);
			}
		}
		atomicAdd(&tsum[c], ts);
		if (nx) atomicAdd(&nexp[c], nx);
	}
}

// One 64-bit tuple for every window of the clumps [c0, c1): lane bit << 48 | word << cb | (clump - c0), cb = bits of a clump number
// inside the slice (2 K + cb <= 47).  The lane bit rides above the 48 bits the radix sort looks at -- keys only, six passes, no value
// array -- and is OR-ed over equal (word, clump) by the fold behind the sort.  Unambiguous windows go into the slot of their last
// position (slot_off: 16 x ClumpLen slots per clump), the expansions of ambiguous ones are appended behind the slots of the slice
// (`extra`, one atomic reservation per window); windows without a word, and every window of a BadList clump, leave BHIP_ACX_NOKEY
// (bit 47: sorts behind every word).
#define BHIP_ACX_NOKEY (1ull << 47)
#define BHIP_ACX_KEYBITS 48
// Round 5: one work item = EIGHT consecutive positions of one lane (one dword of 4-bit symbols), the items of a lane on consecutive
// threads: a thread writes 64 contiguous bytes and a wave 4 KB (the first version gave every thread a whole lane -- 64 different
// 64-byte sectors per store instruction, 0.75 s per pass over the metric's database instead of the 0.14 s its 474 GB of tuples take).
// The window state at the item's first position is rebuilt from the two dwords in front of it (16 symbols >= K - 1: the run lengths
// only matter up to K), 24 symbol steps for 8 tuples.  One block works on one clump at a time.
__global__ __launch_bounds__(256) void k_acx_extract(const uint4 *__restrict__ ref, const uint64_t *__restrict__ ref_off, const uint32_t *__restrict__ clump_len,
                              const uint64_t *__restrict__ slot_off, const uint8_t *__restrict__ is_bad, uint32_t c0, uint32_t c1, uint32_t tot_refs, int K, int z, int cb,
                              unsigned long long *__restrict__ keys, unsigned long long extra_base, unsigned long long *__restrict__ extra_cursor) {
	const uint32_t wmask = (1u << (2 * K)) - 1u;
	for (uint32_t c = c0 + blockIdx.x; c < c1; c += gridDim.x) {
		const uint32_t L = clump_len[c], nchunks = (L + 31) >> 5, ng = (L + 7) >> 3;      // ng: dwords of symbols = items per lane
		const uint32_t *rw = (const uint32_t *)(ref + ref_off[c] * 16);
		unsigned long long *kc = keys + (slot_off[c] - slot_off[c0]);
		const bool bad = is_bad[c] != 0;
		for (uint32_t item = threadIdx.x; item < 16u * ng; item += 256) {
			const uint32_t zz = item / ng, g = item - zz * ng;
			const uint32_t *lw = rw + (size_t)zz * nchunks * 4;
			unsigned long long *kout = kc + (uint64_t)zz * L;
			const bool live = !bad && 16ull * c + zz < tot_refs;
			const unsigned long long tag = (1ull << (48 + zz)) | (unsigned long long)(c - c0);      // lane bit and slice-local clump number
			unsigned long long win = 0;
			uint32_t w = 0, run = 0, lit = 0;
			for (uint32_t d = g >= 2 ? g - 2 : 0; d <= g; ++d) {
				const uint32_t dw = lw[d];
				#pragma unroll
				for (uint32_t k = 0; k < 8; ++k) {
					const uint32_t pos = d * 8 + k;
					const uint32_t sym = (dw >> (4 * k)) & 15u;
					run = (sym >= 1u && !(z && sym == 5u)) ? run + 1 : 0;
					lit = (sym - 1u) < 4u ? lit + 1 : 0;
					w = ((w << 2) | ((sym - 1u) & 3u)) & wmask;
					win = (win << 4) | sym;
					if (d != g || pos >= L) continue;
					kout[pos] = (live && lit >= (uint32_t)K) ? (((unsigned long long)w << cb) | tag) : BHIP_ACX_NOKEY;
					if (live && run >= (uint32_t)K && lit < (uint32_t)K) {
						const unsigned long long prod = amb_product(win, K);
						unsigned long long e = extra_base + atomicAdd(extra_cursor, prod);
						for (unsigned long long idx = 0; idx < prod; ++idx, ++e) {
							unsigned long long r = idx; uint32_t word = 0;
							for (int s = 0; s < K; ++s) {          // symbol s counted from the window's end: 2-bit digit s of the word
								const uint32_t code = (uint32_t)(win >> (4 * s)) & 15u, n = amb_count(code), dgt = (uint32_t)(r % n);
								r /= n;
								word |= ((amb_bases(code) >> (2u * dgt)) & 3u) << (2 * s);
							}
							keys[e] = ((unsigned long long)word << cb) | tag;
						}
					}
				}
			}
		}
	}
}
// what the fold reads out of a sorted tuple: the 48 sorted bits as the key, the lane bits as the value
struct AcxKeyOf { __host__ __device__ unsigned long long operator()(const unsigned long long &t) const { return t & ((1ull << BHIP_ACX_KEYBITS) - 1ull); } };
struct AcxLanesOf { __host__ __device__ uint16_t operator()(const unsigned long long &t) const { return (uint16_t)(t >> 48); } };

// list lengths: one count per distinct (word, clump); the run of tuples that are not words (BHIP_ACX_NOKEY) is skipped
__global__ void k_acx_hist(const unsigned long long *__restrict__ ukeys, uint32_t n_unique, int cb, uint32_t *__restrict__ lens) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_unique; i += gridDim.x * blockDim.x) {
		const unsigned long long key = ukeys[i];
		if (key >> 47) continue;
		atomicAdd(&lens[(uint32_t)(key >> cb)], 1u);
	}
}
// the same straight from the SORTED tuples of a slice, without folding them first (the first pass only wants the lengths): one count
// per tuple whose (word, clump) differs from its predecessor's
__global__ __launch_bounds__(256) void k_acx_hist_sorted(const unsigned long long *__restrict__ skeys, uint32_t n, int cb, uint32_t *__restrict__ lens) {
	// (the tuples of a word are neighbours: a wave adds ONE number per word it sees -- the first lane of the word's run counts the run's
	// distinct tuples from two ballots -- instead of one atomic per tuple on the same address: 54 G atomics took 1.1 s of the build)
	const unsigned long long KM = (1ull << BHIP_ACX_KEYBITS) - 1ull;
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t n64 = (n + 63u) & ~63u;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n64; i += gridDim.x * blockDim.x) {
		const bool in = i < n;
		const unsigned long long key = in ? skeys[i] & KM : ~0ull;
		const unsigned long long prev = (in && i) ? skeys[i - 1] & KM : ~0ull;
		const bool isw = in && !(key >> 47);
		const bool uniq = isw && key != prev;
		const bool lead = isw && (lane == 0 || (key >> cb) != (prev >> cb));
		const unsigned long long um = __ballot(uniq), lm = __ballot(lead);
		if (lead) {
			const unsigned long long rest = lane < 63u ? lm >> (lane + 1u) : 0ull;
			const uint32_t hi = rest ? lane + 1u + (uint32_t)__builtin_ctzll(rest) : 64u;      // the next word's first lane
			const unsigned long long range = (hi == 64u ? ~0ull : (1ull << hi) - 1ull) & ~((1ull << lane) - 1ull);
			const uint32_t cnt = (uint32_t)__popcll(um & range);
			if (cnt) atomicAdd(&lens[(uint32_t)(key >> cb)], cnt);
		}
	}
}
// head[i] = i for the first tuple of every word, 0 elsewhere: an inclusive max-scan turns it into "first tuple of my word"
__global__ void k_acx_heads(const unsigned long long *__restrict__ ukeys, uint32_t n_unique, int cb, uint32_t *__restrict__ head) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_unique; i += gridDim.x * blockDim.x)
		head[i] = (i && (ukeys[i] >> cb) != (ukeys[i - 1] >> cb)) ? i : 0u;
}
// records of one slice: tuple i of word w goes to entry first(w) + (entries of w written by earlier slices) + (rank inside the word);
// the tuples are sorted by (word, clump) and the slices are ascending clump ranges, so every list ends up in ascending clump order
// (what the reference writes with one thread)
__global__ void k_acx_fill(BhipAcxView acx, const unsigned long long *__restrict__ ukeys, const uint16_t *__restrict__ umasks, const uint32_t *__restrict__ head,
                           uint32_t n_unique, int cb, uint32_t c0, const uint32_t *__restrict__ cursor, uint32_t *__restrict__ rec, uint32_t all_lanes) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_unique; i += gridDim.x * blockDim.x) {
		const unsigned long long key = ukeys[i];
		if (key >> 47) continue;
		const uint32_t w = (uint32_t)(key >> cb);
		unsigned long long beg; uint32_t n;
		bhip_acx_range(acx, w, beg, n);
		bhip_rec_store(rec, beg + (cursor ? cursor[w] : 0u) + (i - head[i]), c0 + ((uint32_t)key & ((1u << cb) - 1u)), all_lanes ? 0xFFFFu : (uint32_t)umasks[i]);
	}
}
__global__ void k_acx_advance(const unsigned long long *__restrict__ ukeys, const uint32_t *__restrict__ head, uint32_t n_unique, int cb, uint32_t *__restrict__ cursor) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_unique; i += gridDim.x * blockDim.x) {
		const unsigned long long key = ukeys[i];
		if (key >> 47) continue;
		if (i + 1 == n_unique || (ukeys[i + 1] >> cb) != (key >> cb)) cursor[(uint32_t)(key >> cb)] += i - head[i] + 1;      // last tuple of its word in this slice
	}
}
// back to the file's tables
__global__ void k_acx_lens_from_lines(BhipAcxView acx, uint64_t n_words, uint32_t *__restrict__ lens) {
	for (uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
		unsigned long long e; uint32_t n;
		bhip_acx_range(acx, (uint32_t)w, e, n);
		lens[w] = n;
	}
}
__global__ void k_acx_rec_export(const uint32_t *__restrict__ rec, unsigned long long e0, uint64_t n, uint32_t *__restrict__ clumps, uint16_t *__restrict__ masks) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint2 r = bhip_acx_rec(rec, e0 + i);
		clumps[i] = r.x;
		if (masks) masks[i] = (uint16_t)r.y;
	}
}

// ------------------------------------------------------------------------------------------------
// The same accelerator built in slices of the WORD space (round 5).  The builder above cuts the database into ranges of clumps: a
// word's list then gets entries from every slice, so the list lengths must be known before a record can be placed -- every slice
// is extracted, sorted over 48 bits and folded TWICE (46 % of the build was the radix sort).  Cut by word ranges instead, slice s
// holds EVERY tuple of the words [w0, w1): sorted and folded once, its unique tuples ARE the records of those words, in place
// and in order -- no second pass, no cursor per word, and:
//  * the tuples of a slice are written in ascending clump order (a per-clump offset from one counting scan over the database), so
//    the stable radix sort only has to order the word bits of the slice -- at most 26 bits, four passes, instead of 48 bits in six;
//  * what a slice costs instead is one more scan over the references (33 GB at the metric's size: 36 ms) to pick out its words.
// The record area is one address range (DBuf::reserve_growable) whose memory is mapped as the records come -- its size is known when
// it is full -- and the sort of a slice works at the range's top, in the part the records have not reached yet (below).
// The default builder since the end of round 5: 3.1 s at the metric's size.  Returns 1 (nothing changed) when this box cannot do it
// -- no virtual memory management, more than 2^24 clumps, a bucket of words with more tuples than a sort takes, not enough room --
// and the clump-sliced builder takes over.
// ------------------------------------------------------------------------------------------------
// tuples per bucket of 2^shift words, whole database (one LDS histogram per block)
__global__ __launch_bounds__(256) void k_acx_whist(const uint4 *__restrict__ ref, const uint64_t *__restrict__ ref_off, const uint32_t *__restrict__ clump_len,
		const uint8_t *__restrict__ is_bad, uint32_t n_clumps, uint32_t tot_refs, int K, int z, uint32_t shift, uint32_t n_buckets, unsigned long long *__restrict__ hist) {
	__shared__ uint32_t s_h[4096];
	for (uint32_t i = threadIdx.x; i < n_buckets; i += 256) s_h[i] = 0;
	__syncthreads();
	const uint64_t n_threads = (uint64_t)n_clumps * 16;
	for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n_threads; i += (uint64_t)gridDim.x * 256ull) {
		const uint32_t c = (uint32_t)(i >> 4), zz = (uint32_t)(i & 15);
		if (is_bad[c] || 16ull * c + zz >= tot_refs) continue;
		const uint32_t L = clump_len[c], nchunks = (L + 31) >> 5;
		acx_lane_words(ref + ref_off[c] * 16 + (uint64_t)zz * nchunks, L, nchunks, K, z, [&](uint32_t w) { atomicAdd(&s_h[w >> shift], 1u); });
	}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < n_buckets; i += 256) if (s_h[i]) atomicAdd(&hist[i], (unsigned long long)s_h[i]);
}
// tuples per (slice, clump): counts[s * n_clumps + c]; a block = 16 clumps at a time
#define BHIP_ACX_MAX_SLICES 256u
__global__ __launch_bounds__(256) void k_acx_wcount(const uint4 *__restrict__ ref, const uint64_t *__restrict__ ref_off, const uint32_t *__restrict__ clump_len,
		const uint8_t *__restrict__ is_bad, uint32_t n_clumps, uint32_t tot_refs, int K, int z, uint32_t shift, uint32_t n_buckets, const uint8_t *__restrict__ b2s, uint32_t n_slices,
		uint32_t *__restrict__ counts) {
	__shared__ uint8_t s_b2s[4096];
	__shared__ uint32_t s_cnt[16][BHIP_ACX_MAX_SLICES];
	for (uint32_t i = threadIdx.x; i < n_buckets; i += 256) s_b2s[i] = b2s[i];
	const uint32_t cl = threadIdx.x >> 4, zz = threadIdx.x & 15u;
	const uint32_t n_groups = (n_clumps + 15u) >> 4;
	for (uint32_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
		for (uint32_t i = threadIdx.x; i < 16u * n_slices; i += 256) s_cnt[i / n_slices][i % n_slices] = 0;
		__syncthreads();
		const uint32_t c = grp * 16u + cl;
		if (c < n_clumps && !is_bad[c] && 16ull * c + zz < tot_refs) {
			const uint32_t L = clump_len[c], nchunks = (L + 31) >> 5;
			acx_lane_words(ref + ref_off[c] * 16 + (uint64_t)zz * nchunks, L, nchunks, K, z, [&](uint32_t w) { const uint32_t sl = s_b2s[w >> shift]; if (sl != 0xFFu) atomicAdd(&s_cnt[cl][sl], 1u); });      // (0xFF: another rank's words, cooperative build)
		}
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < 16u * n_slices; i += 256) {
			const uint32_t s = i >> 4, k = i & 15u;
			if (grp * 16u + k < n_clumps) counts[(uint64_t)s * n_clumps + grp * 16u + k] = s_cnt[k][s];
		}
		__syncthreads();
	}
}
// the tuples of the words [w0, w1): lane << 60 | (word - w0) << cbits | clump, those of clump c at off[c] .. (any order inside a clump)
__global__ __launch_bounds__(256) void k_acx_wwrite(const uint4 *__restrict__ ref, const uint64_t *__restrict__ ref_off, const uint32_t *__restrict__ clump_len,
		const uint8_t *__restrict__ is_bad, uint32_t n_clumps, uint32_t tot_refs, int K, int z, uint32_t w0, uint32_t w1, uint32_t cbits, const uint32_t *__restrict__ off,
		unsigned long long *__restrict__ keys) {
	__shared__ uint32_t s_cur[16];
	const uint32_t cl = threadIdx.x >> 4, zz = threadIdx.x & 15u;
	const uint32_t n_groups = (n_clumps + 15u) >> 4;
	for (uint32_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
		if (threadIdx.x < 16) s_cur[threadIdx.x] = 0;
		__syncthreads();
		const uint32_t c = grp * 16u + cl;
		if (c < n_clumps && !is_bad[c] && 16ull * c + zz < tot_refs) {
			const uint32_t L = clump_len[c], nchunks = (L + 31) >> 5, base = off[c];
			const unsigned long long tag = (unsigned long long)zz << 60 | c;
			acx_lane_words(ref + ref_off[c] * 16 + (uint64_t)zz * nchunks, L, nchunks, K, z, [&](uint32_t w) {
				if (w >= w0 && w < w1) keys[base + atomicAdd(&s_cur[cl], 1u)] = tag | (unsigned long long)(w - w0) << cbits;
			});
		}
		__syncthreads();
	}
}
struct AcxWKeyOf { __host__ __device__ unsigned long long operator()(const unsigned long long &t) const { return t & ((1ull << 60) - 1ull); } };
struct AcxWLaneOf { __host__ __device__ uint16_t operator()(const unsigned long long &t) const { return (uint16_t)(1u << (uint32_t)(t >> 60)); } };
// unique tuple i of a slice = record rec0 + i, and one count for its word's list length -- added per WORD and wave (the tuples of a word
// are neighbours: the first lane of a word's run inside the wave adds the run's length; one atomic per record on 50 equal addresses in a
// row made this kernel 2.75 s of a 6.8 s build)
__global__ __launch_bounds__(256) void k_acx_wfill(const unsigned long long *__restrict__ ukeys, const uint16_t *__restrict__ umasks, uint32_t n_unique, uint32_t w0, uint32_t cbits,
		uint32_t *__restrict__ rec, uint32_t *__restrict__ lens, uint32_t all_lanes) {
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t n64 = (n_unique + 63u) & ~63u;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n64; i += gridDim.x * blockDim.x) {
		const bool in = i < n_unique;
		const unsigned long long key = in ? ukeys[i] : 0ull;
		const uint32_t word = (uint32_t)(key >> cbits);
		if (in) bhip_rec_store(rec, i, (uint32_t)key & ((1u << cbits) - 1u), all_lanes ? 0xFFFFu : (uint32_t)umasks[i]);
		const uint32_t before = __shfl_up(word, 1);
		const bool lead = in && (lane == 0 || word != before);
		const unsigned long long lm = __ballot(lead), im = __ballot(in);
		if (lead) {
			const unsigned long long rest = lane < 63u ? lm >> (lane + 1u) : 0ull;
			const uint32_t hi = rest ? lane + 1u + (uint32_t)__builtin_ctzll(rest) : 64u;      // the next word's first lane
			const unsigned long long range = (hi == 64u ? ~0ull : (1ull << hi) - 1ull) & ~((1ull << lane) - 1ull);
			atomicAdd(&lens[w0 + word], (uint32_t)__popcll(im & range));
		}
	}
}

// Cooperative form (n_parts > 1; bhip_build_accelerator_shared): the database is replicated over n_parts devices and every one of them needs
// the whole accelerator.  Rank `part` builds the lists of ITS run of word buckets only -- the runs are cut from the bucket histogram, which
// every rank computes alike, so that they hold equal numbers of tuples; a run's records are one contiguous region of the record area and its
// list lengths one contiguous range of Lens -- and the ranks then complete each other's arrays through `share` (bhip_share_fn: RCCL
// broadcasts over xGMI, peer copies between the threads of one process, or whatever the caller has): first Lens (4^K x 4 bytes in all),
// from which everyone derives the same offset lines and the regions' places, then the records.  A rank's work is 1/n_parts of the sort
// and fold plus the scans over the whole database (the histogram, the counts, one per slice); what crosses the links is (n - 1)/n of the
// tables INTO every rank.  A rank that cannot do its share says so in the first exchange and ALL ranks return 1 (the caller builds alone).
struct U32To64 { __host__ __device__ unsigned long long operator()(const uint32_t &x) const { return (unsigned long long)x; } };
// [0, bytes) of `base` moved up by `shift` bytes (the ranges may overlap: from the top down, in pieces no longer than the shift; a small
// shift goes through a bounce buffer)
static int acx_move_up(Handle *h, char *base, size_t bytes, size_t shift) {
	if (!shift || !bytes) return 0;
	const size_t kPiece = (size_t)256 << 20;
	if (shift >= bytes) { HIPCHK(hipMemcpyAsync(base + shift, base, bytes, hipMemcpyDeviceToDevice, h->stream)); return 0; }
	if (shift >= ((size_t)16 << 20)) {
		const size_t piece = std::min(shift, kPiece);
		for (size_t end = bytes; end > 0;) { const size_t n = std::min(piece, end); end -= n; HIPCHK(hipMemcpyAsync(base + end + shift, base + end, n, hipMemcpyDeviceToDevice, h->stream)); }
		return 0;
	}
	DTmp bounce;
	ARC(bounce.reserve(kPiece));
	for (size_t end = bytes; end > 0;) {
		const size_t n = std::min(kPiece, end); end -= n;
		HIPCHK(hipMemcpyAsync(bounce.p, base + end, n, hipMemcpyDeviceToDevice, h->stream));
		HIPCHK(hipMemcpyAsync(base + end + shift, bounce.p, n, hipMemcpyDeviceToDevice, h->stream));
	}
	HIPCHK(hipStreamSynchronize(h->stream));
	return 0;
}
static int build_accelerator_by_words(Handle *h, int K, int z, int part = 0, int n_parts = 1, bhip_share_fn share = nullptr, void *share_ctx = nullptr) {
	const bool coop = n_parts > 1;
	const uint32_t nC = h->n_clumps;
	const uint64_t nw = 1ull << (2 * K);
	const bool dbg = getenv("BHIP_DEBUG") != nullptr;
	const auto t_begin = std::chrono::steady_clock::now();
	auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
	uint32_t cbits = 1; while ((1ull << cbits) < nC) ++cbits;
	const uint32_t bb = (uint32_t)std::min(12, 2 * K), shift = (uint32_t)(2 * K) - bb, n_buckets = 1u << bb;
	const uint32_t g = (uint32_t)h->n_cu * 16;
	const uint64_t n_lines = (nw + BHIP_ACX_LINE_WORDS - 1) / BHIP_ACX_LINE_WORDS;
	const uint32_t all_lanes = getenv("BHIP_NO_LANE_MASKS") ? 1u : 0u;
	std::vector<uint32_t> badlist;
	std::vector<uint32_t> rb((size_t)n_parts + 1, 0);      // bucket boundaries of the ranks' runs
	DTmp d_lens, lines_scratch;
	uint64_t rec_n = 0, total = 0, cap_items = 0;
	uint32_t n_slices = 0;
	double t_hist = 0, t_count = 0, t_sort = 0;
	// everything a rank does on its own: 0 = its lists are built (records [0, rec_n) of the record area, their lengths in d_lens),
	// 1 = cannot run here, < 0 = error
	auto local_part = [&]() -> int {
	if (cbits > 24) return 1;
	if (coop) if (const char *ev = getenv("BHIP_TEST_COOP_FAIL_RANK")) if (atoi(ev) == part) return 1;      // (test hook: this rank cannot -> all ranks build alone)
	// 1. expansion estimate per clump -> BadList (as the clump-sliced builder)
	std::vector<unsigned long long> tsum(nC), nexp(nC);
	std::vector<uint8_t> is_bad(nC, 0);
	DTmp d_bad;
	{
		DTmp d_ts, d_nx;
		ARC(d_ts.reserve((size_t)nC * 8)); ARC(d_nx.reserve((size_t)nC * 8));
		HIPCHK(hipMemsetAsync(d_ts.p, 0, (size_t)nC * 8, h->stream)); HIPCHK(hipMemsetAsync(d_nx.p, 0, (size_t)nC * 8, h->stream));
		hipLaunchKernelGGL(k_acx_budget, dim3(std::min<uint32_t>((nC * 16u + 255u) / 256u, (uint32_t)h->n_cu * 16)), dim3(256), 0, h->stream, h->ref_lane.as<uint4>(), h->ref_off.as<uint64_t>(),
			h->clump_len.as<uint32_t>(), nC, h->tot_refs, K, z ? 1 : 0, d_ts.as<unsigned long long>(), d_nx.as<unsigned long long>());
		HIPCHK(hipGetLastError());
		HIPCHK(hipMemcpyAsync(tsum.data(), d_ts.p, (size_t)nC * 8, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipMemcpyAsync(nexp.data(), d_nx.p, (size_t)nC * 8, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	}
	const unsigned long long full_size = K > 14 ? 0x7FFFFFFFull : (1ull << 24);      // burst.c:3322
	for (uint32_t c = 0; c < nC; ++c) {
		if (tsum[c] >= full_size) { is_bad[c] = 1; badlist.push_back(c); continue; }
		if (16ull * h->h_clump_len[c] + nexp[c] >= (1ull << 31)) return 1;
	}
	ARC(d_bad.reserve((size_t)nC + 16));
	HIPCHK(hipMemcpyAsync(d_bad.p, is_bad.data(), nC, hipMemcpyHostToDevice, h->stream));
	// 2. tuples per bucket of words -> the ranks' runs of buckets (equal numbers of tuples), and inside this rank's run the slices: runs of
	// buckets with at most `target` tuples and at most 2^24 words (three sort passes)
	std::vector<unsigned long long> hist(n_buckets);
	{
		DTmp d_hist;
		ARC(d_hist.reserve((size_t)n_buckets * 8));
		HIPCHK(hipMemsetAsync(d_hist.p, 0, (size_t)n_buckets * 8, h->stream));
		hipLaunchKernelGGL(k_acx_whist, dim3(g), dim3(256), 0, h->stream, h->ref_lane.as<uint4>(), h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), d_bad.as<uint8_t>(),
			nC, h->tot_refs, K, z ? 1 : 0, shift, n_buckets, d_hist.as<unsigned long long>());
		HIPCHK(hipGetLastError());
		HIPCHK(hipMemcpyAsync(hist.data(), d_hist.p, (size_t)n_buckets * 8, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	}
	t_hist = since();
	uint64_t biggest = 0;
	total = 0;
	for (uint32_t b = 0; b < n_buckets; ++b) total += hist[b];
	if (!total) return 1;
	{
		uint64_t run = 0; int r = 1;
		for (uint32_t b = 0; b < n_buckets && r < n_parts; ++b) {
			while (r < n_parts && run >= (uint64_t)((double)total * r / n_parts)) rb[r++] = b;      // (run = tuples of the buckets before b)
			run += hist[b];
		}
		for (; r < n_parts; ++r) rb[r] = n_buckets;
		rb[n_parts] = n_buckets;
	}
	const uint32_t own0 = rb[part], own1 = rb[part + 1];
	uint64_t total_own = 0;
	for (uint32_t b = own0; b < own1; ++b) { total_own += hist[b]; biggest = std::max<uint64_t>(biggest, hist[b]); }
	size_t free_b = 0, total_b = 0;
	HIPCHK(hipMemGetInfo(&free_b, &total_b));
	long long forced_slice = 0;
	if (const char *ev = getenv("BHIP_MASK_SLICE")) forced_slice = atoll(ev);
	// The record area is ONE address range (DBuf::reserve_growable) and the sort works inside it: the records fill it from the bottom, the
	// slice at hand is sorted and folded at its TOP -- two 8-byte tuple arrays and the folded lane masks, 18 bytes per tuple -- in the part
	// the records have not reached yet.  So a slice may have as many tuples as fit between the records so far and the top,
	// 4 (before + n) + 18 n <= range: the early slices take what one sort call takes (2^31 tuples), the last ones what is left beside
	// 216 GB of records -- 28 slices at the metric's size, where "one record per tuple, and the sort buffers beside them" gave 64.  Every
	// slice costs one scan of the references, so fewer, larger slices are what makes this builder cheap.  Nothing is unmapped before the
	// build is over (first version: sort buffers of their own, mapped and unmapped slice by slice -- ranks building side by side read each
	// other's regions back as zeros once in three runs; with ordinary allocations of a fixed size: never in 48).
	// A slice spans at most 2^26 words (four sort passes).
	std::vector<uint32_t> cuts;        // bucket boundaries of the slices
	std::vector<uint64_t> slice_items;
	const uint32_t max_b = 26 > shift ? 1u << (26 - shift) : 1u;
	auto buf_bytes = [](uint64_t n) -> size_t { return 2 * ((size_t)(n * 8 + 16 + 255) & ~(size_t)255) + ((size_t)(n * 2 + 16 + 255) & ~(size_t)255); };
	// the range: the final size (a record per tuple of the whole database at most) and room for the largest slice's sort on top of the own
	// records, as far as the device has it
	// (round 6: 24 GB instead of 3 left alone beyond the tables.  What the sort's part of the range gives back at the end of the build is not
	// allocatable at once -- nor is all of what a process that held the device a moment ago gave back, whatever hipMemGetInfo says: behind a
	// parent that had opened and closed the same database, `burst_hip` at the metric's size could allocate 2.5 GB with 16 GB reported free -- and
	// the caller's batch buffers, 7.4 GB for 2 M-entry batches, are reserved right behind the build.  The price: the last slices sort in less room.)
	const double other = (double)nw * 4.0 + (double)(n_lines + 1) * 64.0 + (double)acx_lines_scratch_bytes(nw) + (double)nC * 8.0 + 40.0 * nC * 4.0 + (double)(24ull << 30);
	const double avail = (double)free_b - other;
	const uint64_t biggest_sort = std::min<uint64_t>(total_own, 2147483000ull);
	double want_va = std::max((double)total * BHIP_REC_BYTES + 16.0, (double)total_own * BHIP_REC_BYTES + (double)buf_bytes(biggest_sort)) + 4096.0;
	if (want_va > avail) want_va = avail;
	if (want_va < (double)total * BHIP_REC_BYTES + 16.0 + 4096.0) return 1;      // (not even the records fit)
	// (reserve_growable rounds up to whole chunks and adds one: planned with the two chunks taken off)
	const size_t va_plan = (size_t)want_va > 2 * DBuf::kChunk + ((size_t)total * BHIP_REC_BYTES + 16) ? (size_t)want_va - 2 * DBuf::kChunk : (size_t)total * BHIP_REC_BYTES + 16;
	if (h->acx_rec.reserve_growable(va_plan, h->device)) return 1;
	const size_t va_size = h->acx_rec.va_size;
	if (getenv("BHIP_DEBUG")) fprintf(stderr, "[bhip] word-sliced build: %.2f GB free at its start, %.2f GB set aside for the rest, range of %.2f GB for %.2f GB of records at most\n", free_b / 1e9, other / 1e9, va_size / 1e9, (double)total * BHIP_REC_BYTES / 1e9);
	auto plan = [&](uint64_t target) -> uint32_t {      // (BHIP_MASK_SLICE: slices of a given size)
		cuts.assign(1, own0); slice_items.clear(); cap_items = 0;
		uint64_t before = 0;
		for (uint32_t b0 = own0; b0 < own1;) {
			uint32_t b1 = b0 + 1; uint64_t n = hist[b0];
			while (b1 < own1 && b1 - b0 < max_b && n + hist[b1] <= target) n += hist[b1++];
			if ((before + n) * BHIP_REC_BYTES + 16 + buf_bytes(n) > va_size) return 0;
			cuts.push_back(b1); slice_items.push_back(n); cap_items = std::max(cap_items, n); before += n; b0 = b1;
		}
		return (uint32_t)cuts.size() - 1;
	};
	auto plan_by_room = [&]() -> uint32_t {
		cuts.assign(1, own0); slice_items.clear(); cap_items = 0;
		uint64_t before = 0;
		for (uint32_t b0 = own0; b0 < own1;) {
			const double room = (double)va_size - (double)before * BHIP_REC_BYTES - 4096.0;
			const uint64_t target = room > 0 ? (uint64_t)std::min(2147483000.0, room / 22.0) : 0;
			if (hist[b0] > target) return 0;
			uint32_t b1 = b0 + 1; uint64_t n = hist[b0];
			while (b1 < own1 && b1 - b0 < max_b && n + hist[b1] <= target) n += hist[b1++];
			cuts.push_back(b1); slice_items.push_back(n); cap_items = std::max(cap_items, n); before += n; b0 = b1;
		}
		return (uint32_t)cuts.size() - 1;
	};
	const uint32_t max_slices = BHIP_ACX_MAX_SLICES - 1u;      // (slice number 0xFF marks "not this build's bucket" in the one-byte bucket -> slice table, in every mode)
	if (own1 > own0 && total_own) {
		n_slices = forced_slice > 0 ? plan(std::max<uint64_t>((uint64_t)forced_slice, biggest)) : plan_by_room();
		if (!n_slices || n_slices > max_slices || cap_items >= 2147483000ull) return 1;
	}
	if (const char *ev = getenv("BHIP_TEST_ENTRY_BIAS")) h->acx_bias = strtoull(ev, nullptr, 0);
	h->K = K;
	ARC(d_lens.reserve(nw * 4 + 16));
	HIPCHK(hipMemsetAsync(d_lens.p, 0, nw * 4, h->stream));
	ARC(h->acx_lines.reserve_exact((n_lines + 1) * 64));      // (now, while the device has room: an allocation next to 216 GB of mapped records takes half a second)
	ARC(lines_scratch.reserve(acx_lines_scratch_bytes(nw)));
	if (!n_slices) { HIPCHK(hipStreamSynchronize(h->stream)); return 0; }      // (a rank whose run of buckets is empty: it only receives)
	// 3. tuples per (slice, clump) in one scan
	std::vector<uint8_t> b2s(n_buckets, 0xFF);
	for (uint32_t s = 0; s < n_slices; ++s) for (uint32_t b = cuts[s]; b < cuts[s + 1]; ++b) b2s[b] = (uint8_t)s;
	DTmp d_b2s, d_counts, d_off, nruns, tmp;
	ARC(d_b2s.reserve(n_buckets)); ARC(d_counts.reserve_exact((size_t)n_slices * nC * 4 + 16)); ARC(d_off.reserve((size_t)nC * 4 + 16));
	HIPCHK(hipMemcpyAsync(d_b2s.p, b2s.data(), n_buckets, hipMemcpyHostToDevice, h->stream));
	hipLaunchKernelGGL(k_acx_wcount, dim3(g), dim3(256), 0, h->stream, h->ref_lane.as<uint4>(), h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), d_bad.as<uint8_t>(),
		nC, h->tot_refs, K, z ? 1 : 0, shift, n_buckets, d_b2s.as<uint8_t>(), n_slices, d_counts.as<uint32_t>());
	HIPCHK(hipGetLastError());
	ARC(nruns.reserve(16));
	{	// the library calls' own scratch for the largest slice, once
		size_t tb = 0, tb1 = 0;
		hipcub::DoubleBuffer<unsigned long long> dk((unsigned long long *)nullptr, (unsigned long long *)nullptr);
		HIPCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, tb1, dk, (int)cap_items, (int)cbits, (int)(cbits + 26), h->stream)); tb = std::max(tb, tb1);
		hipcub::TransformInputIterator<unsigned long long, AcxWKeyOf, const unsigned long long *> kin((const unsigned long long *)nullptr, AcxWKeyOf());
		hipcub::TransformInputIterator<uint16_t, AcxWLaneOf, const unsigned long long *> vin((const unsigned long long *)nullptr, AcxWLaneOf());
		HIPCHK(hipcub::DeviceReduce::ReduceByKey(nullptr, tb1, kin, (unsigned long long *)nullptr, vin, (uint16_t *)nullptr, (uint32_t *)nullptr, BitOrU16(), (int)cap_items, h->stream)); tb = std::max(tb, tb1);
		HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb1, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)nC, h->stream)); tb = std::max(tb, tb1);
		ARC(tmp.reserve(tb + tb / 4 + 4096));
	}
	HIPCHK(hipStreamSynchronize(h->stream));
	t_count = since();
	// 4. slice after slice: offsets, tuples, sort over the slice's word bits, fold, records.  The range's memory is mapped by a thread of
	// its own -- the top for the slice's sort, the bottom for the records so far plus this slice's (at most its tuples) -- while the slice
	// before is at work: mapping 230 GB in 1 GiB chunks takes seconds, which lie beside the kernels.
	std::atomic<int> map_failed(0), stop(0);
	std::atomic<size_t> mapped_lo(0), mapped_top(0), want_lo(0), want_top(0);
	std::thread mapper;
	char map_err[400] = "";      // the mapper thread's own error text (hipMemCreate / hipMemMap ...), written before map_failed is set
	struct JoinMapper { std::thread &t; std::atomic<int> &stop; ~JoinMapper() { stop = 1; if (t.joinable()) t.join(); } } join_mapper{mapper, stop};
	{
		DBuf *rec = &h->acx_rec;
		const int dev = h->device;
		char *const merr = map_err;
		mapper = std::thread([rec, dev, merr, &want_lo, &want_top, &mapped_lo, &mapped_top, &map_failed, &stop]() {
			auto give_up = [&](const char *what) { snprintf(merr, 400, "%s: %s", what, bhip_last_error()); (void)hipGetLastError(); map_failed = 1; };
			if (hipSetDevice(dev) != hipSuccess) { give_up("hipSetDevice in the mapping thread"); return; }
			size_t top = 0;
			while (!stop.load()) {
				const size_t wt = want_top.load(), wl = want_lo.load();
				if (wt > top) { if (rec->grow_top_to(wt)) { give_up("mapping the top of the range"); return; } top = (wt + DBuf::kChunk - 1) / DBuf::kChunk * DBuf::kChunk; }
				else if (wl > rec->cap) { if (rec->grow_to(wl)) { give_up("mapping the records' part of the range"); return; } }
				else std::this_thread::yield();
				mapped_top = top; mapped_lo = rec->cap;      // (both, every time: in a small range the top's chunks ARE the records' -- the prefix grows without grow_to)
			}
		});
	}
	char *const va_end = h->acx_rec.as<char>() + va_size;
	double t_map = 0, t_scan = 0, t_fill = 0, t_wait = 0;
	auto lap = [&](double &acc, const std::chrono::steady_clock::time_point &from) { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - from).count(); };
	for (uint32_t s = 0; s < n_slices; ++s) {
		const uint64_t n_items = slice_items[s];
		if (!n_items) continue;
		want_top = buf_bytes(n_items);
		want_lo = (size_t)(rec_n + n_items) * BHIP_REC_BYTES + 16;      // (the records so far are known, this slice's are at most its tuples)
		if ((size_t)(rec_n + n_items) * BHIP_REC_BYTES + 16 + buf_bytes(n_items) > va_size) return fail(BHIP_E_INTERNAL, "accelerator build: slice %u does not fit its plan", s);
		const auto tm0 = std::chrono::steady_clock::now();
		while (mapped_top.load() < buf_bytes(n_items) && !map_failed.load()) std::this_thread::yield();
		lap(t_map, tm0);
		if (map_failed.load()) return fail(BHIP_E_DEVICE, "accelerator build: the sort's part of the record area could not be mapped (%s)", map_err);
		unsigned long long *const k0 = (unsigned long long *)(va_end - buf_bytes(n_items));
		unsigned long long *const k1 = (unsigned long long *)((char *)k0 + ((size_t)(n_items * 8 + 16 + 255) & ~(size_t)255));
		uint16_t *const v0 = (uint16_t *)((char *)k1 + ((size_t)(n_items * 8 + 16 + 255) & ~(size_t)255));
		const uint64_t w0 = (uint64_t)cuts[s] << shift, w1 = (uint64_t)cuts[s + 1] << shift;
		uint32_t wl = 1; while ((1ull << wl) < w1 - w0) ++wl;
		const uint32_t *cnt_s = d_counts.as<uint32_t>() + (size_t)s * nC;
		size_t tb = tmp.cap;
		HIPCHK(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, cnt_s, d_off.as<uint32_t>(), (int)nC, h->stream));
		hipLaunchKernelGGL(k_acx_wwrite, dim3(g), dim3(256), 0, h->stream, h->ref_lane.as<uint4>(), h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), d_bad.as<uint8_t>(),
			nC, h->tot_refs, K, z ? 1 : 0, (uint32_t)w0, (uint32_t)std::min<uint64_t>(w1, 0xFFFFFFFFull), cbits, d_off.as<uint32_t>(), k0);
		HIPCHK(hipGetLastError());
		if (dbg) { const auto tc0 = std::chrono::steady_clock::now(); HIPCHK(hipStreamSynchronize(h->stream)); lap(t_scan, tc0); }      // (BHIP_DEBUG: the scan timed on its own)
		const auto ts0 = std::chrono::steady_clock::now();
		hipcub::DoubleBuffer<unsigned long long> dk(k0, k1);
		tb = tmp.cap;
		HIPCHK(hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, dk, (int)n_items, (int)cbits, (int)(cbits + wl), h->stream));
		unsigned long long *skeys = dk.Current(), *ukeys = dk.Alternate();
		hipcub::TransformInputIterator<unsigned long long, AcxWKeyOf, const unsigned long long *> kin(skeys, AcxWKeyOf());
		hipcub::TransformInputIterator<uint16_t, AcxWLaneOf, const unsigned long long *> vin(skeys, AcxWLaneOf());
		tb = tmp.cap;
		HIPCHK(hipcub::DeviceReduce::ReduceByKey(tmp.p, tb, kin, ukeys, vin, v0, nruns.as<uint32_t>(), BitOrU16(), (int)n_items, h->stream));
		uint32_t n_unique = 0, last_off = 0, last_cnt = 0;
		HIPCHK(hipMemcpyAsync(&n_unique, nruns.p, 4, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipMemcpyAsync(&last_off, d_off.as<uint32_t>() + (nC - 1), 4, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipMemcpyAsync(&last_cnt, cnt_s + (nC - 1), 4, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		// (the tuples the counting scan attributed to this slice are the tuples the plan sized its buffers for: anything else means
		// the write pass and the sort did not see the same data -- stop before the records are stored)
		if ((uint64_t)last_off + last_cnt != n_items) return fail(BHIP_E_INTERNAL, "accelerator build: slice %u holds %llu tuples, its plan says %llu", s, (unsigned long long)last_off + last_cnt, (unsigned long long)n_items);
		if (dbg) t_sort += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts0).count();
		const auto tw0 = std::chrono::steady_clock::now();
		while (mapped_lo.load() < (rec_n + n_unique) * BHIP_REC_BYTES + 16 && !map_failed.load()) std::this_thread::yield();
		lap(t_wait, tw0);
		if (map_failed.load()) return fail(BHIP_E_DEVICE, "accelerator build: the record area could not be mapped (%llu records so far; %s)", (unsigned long long)rec_n, map_err);
		const auto tf0 = std::chrono::steady_clock::now();
		hipLaunchKernelGGL(k_acx_wfill, dim3(g), dim3(256), 0, h->stream, ukeys, v0, n_unique, (uint32_t)w0, cbits,
			h->acx_rec.as<uint32_t>() + rec_n, d_lens.as<uint32_t>(), all_lanes);
		HIPCHK(hipGetLastError());
		rec_n += n_unique;
		HIPCHK(hipStreamSynchronize(h->stream));
		lap(t_fill, tf0);
	}
	stop = 1;
	if (mapper.joinable()) mapper.join();
	// (nothing is unmapped here: the range is cut back to its final size when that is known -- behind the offset lines)
	if (dbg) fprintf(stderr, "[bhip] word-sliced build, inside the slices: %.2f s scans, %.2f s sort + fold, %.2f s records, %.2f s waiting for the sort's part of the range, %.2f s for the records' part\n", t_scan, t_sort, t_fill, t_map, t_wait);
	return 0;
	};
	int rc = local_part();
	std::vector<uint64_t> boff((size_t)n_parts + 1, 0);
	if (coop) {      // the list lengths of the other ranks' words (a rank that could not build its own says so: everyone leaves)
		for (int r = 0; r <= n_parts; ++r) boff[r] = ((uint64_t)rb[r] << shift) * 4ull;
		const int st = share(share_ctx, d_lens.p, boff.data(), part, n_parts, rc != 0);
		if (rc < 0) return rc;
		if (st < 0) return fail(BHIP_E_DEVICE, "cooperative accelerator build: the exchange of the list lengths failed");
		if (rc || st) return 1;
	} else if (rc) return rc;
	const double t_own = since();
	uint64_t tot = 0; uint32_t maxlen = 0;
	rc = acx_lines_from_lens(h, d_lens.as<uint32_t>(), nw, &tot, &maxlen, &lines_scratch);
	const double t_lines = since() - t_own;
	std::vector<unsigned long long> eoff((size_t)n_parts + 1, 0);      // first record of every rank's region
	if (coop && !rc) {
		DTmp d_sum, tmp;
		rc = d_sum.reserve((size_t)n_parts * 8);
		hipcub::TransformInputIterator<unsigned long long, U32To64, const uint32_t *> in(d_lens.as<uint32_t>(), U32To64());
		for (int r = 0; r < n_parts && !rc; ++r) {
			const uint64_t w0 = (uint64_t)rb[r] << shift, w1 = (uint64_t)rb[r + 1] << shift;
			size_t tb = 0;
			if (hipcub::DeviceReduce::Sum(nullptr, tb, in + w0, d_sum.as<unsigned long long>() + r, (int)(w1 - w0), h->stream) != hipSuccess || (rc = tmp.reserve(tb)) ||
			    hipcub::DeviceReduce::Sum(tmp.p, tb, in + w0, d_sum.as<unsigned long long>() + r, (int)(w1 - w0), h->stream) != hipSuccess) { if (!rc) rc = fail(BHIP_E_DEVICE, "cooperative accelerator build: region sizes"); }
		}
		if (!rc && (hipMemcpyAsync(eoff.data() + 1, d_sum.p, (size_t)n_parts * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess))
			rc = fail(BHIP_E_DEVICE, "cooperative accelerator build: region sizes");
		for (int r = 0; r < n_parts; ++r) eoff[r + 1] += eoff[r];
		if (!rc && (eoff[n_parts] != tot || eoff[part + 1] - eoff[part] != rec_n))
			rc = fail(BHIP_E_INTERNAL, "cooperative accelerator build: rank %d wrote %llu records, its lists add up to %llu (all: %llu of %llu)", part, (unsigned long long)rec_n,
				(unsigned long long)(eoff[part + 1] - eoff[part]), (unsigned long long)eoff[n_parts], (unsigned long long)tot);
		// the whole record area, this rank's region moved to its place, the other regions from their builders
		if (!rc) rc = h->acx_rec.grow_to(tot * BHIP_REC_BYTES + 16);
		if (!rc) rc = acx_move_up(h, h->acx_rec.as<char>(), rec_n * BHIP_REC_BYTES, eoff[part] * BHIP_REC_BYTES);
		if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(BHIP_E_DEVICE, "cooperative accelerator build: moving the region");
		for (int r = 0; r <= n_parts; ++r) boff[r] = eoff[r] * BHIP_REC_BYTES;
		const int st = share(share_ctx, h->acx_rec.p, boff.data(), part, n_parts, rc != 0);
		if (rc < 0) { h->acx_rec.release(); h->acx_lines.release(); return rc; }
		if (st < 0) { h->acx_rec.release(); h->acx_lines.release(); return fail(BHIP_E_DEVICE, "cooperative accelerator build: the exchange of the records failed"); }
		if (st) return 1;
	} else if (coop) {      // (the offset lines failed here: the others must not wait)
		(void)share(share_ctx, h->acx_rec.p, boff.data(), part, n_parts, 1);
		return rc;
	}
	if (rc) return rc;
	d_lens.release();
	if (!coop && tot != rec_n) return fail(BHIP_E_INTERNAL, "accelerator build: %llu records written, the list lengths add up to %llu", (unsigned long long)rec_n, (unsigned long long)tot);
	ARC(h->acx_rec.grow_to(tot * BHIP_REC_BYTES + 16));
	h->acx_rec.shrink_to(tot * BHIP_REC_BYTES + 16);      // (the sort's part of the range and what was mapped ahead of the records go back)
	ARC(set_badlist(h, badlist.data(), (uint32_t)badlist.size()));
	h->has_acx = true; h->n_ent = tot; h->has_masks = !all_lanes;
	if (dbg) fprintf(stderr, "[bhip] accelerator built on the device by word ranges%s: K=%d, %llu entries from %llu word tuples, %u slice(s) of at most %llu tuples here (%llu records), %zu clump(s) on the BadList, %.2f B per entry; "
		"%.2f s (%.2f s histogram, %.2f s counts, %.2f s sort + fold, %.2f s until the own lists stood, %.2f s offset lines)\n", coop ? " (cooperative)" : "", K, (unsigned long long)tot, (unsigned long long)total, n_slices,
		(unsigned long long)cap_items, (unsigned long long)rec_n, badlist.size(), tot ? (double)(h->acx_rec.cap + n_lines * 64) / (double)tot : 0.0, since(), t_hist, t_count - t_hist, t_sort, t_own, t_lines);
	if (dbg) { size_t f_ = 0, t_ = 0; if (hipMemGetInfo(&f_, &t_) == hipSuccess) fprintf(stderr, "[bhip] after the build: %.2f GB of the device's %.2f free (record area %.2f GB mapped of a %.2f GB range)\n", f_ / 1e9, t_ / 1e9, h->acx_rec.cap / 1e9, h->acx_rec.va_size / 1e9); }
	if (dbg && coop) fprintf(stderr, "[bhip] rank %d of %d: words [%llu, %llu), records [%llu, %llu)\n", part, n_parts, (unsigned long long)rb[part] << shift, (unsigned long long)rb[part + 1] << shift,
		eoff[part], eoff[part + 1]);
	return 0;
}

static int build_accelerator_by_clumps(Handle *h, int K, int z);
// The word-sliced builder is the default since the end of round 5 (3.1 s against 7.7 s at the metric's size on a quiet device,
// profiles/r05t_cli_quiet_device.txt, profiles/r05m_builders_ab.txt); where it cannot run (no virtual memory management, more than 2^24 clumps, not enough room for its plan) the clump-sliced
// one takes over.  BHIP_ACX_BUILD=clumps asks for that one (and so do the test hooks of its large-database path), =words for the other only.
int bhip_build_accelerator(Handle *h, int K, int z) {
	const char *how = getenv("BHIP_ACX_BUILD");
	const bool only_words = how && !strcmp(how, "words");
	const bool clumps = (how && !strcmp(how, "clumps")) || ((getenv("BHIP_TEST_TWO_PLANS") || getenv("BHIP_ACX_NO_PREMAP")) && !only_words);
	if (!clumps) {
		const int rc = build_accelerator_by_words(h, K, z);
		if (rc == 0 || (rc < 0 && rc != BHIP_E_DEVICE)) return rc;      // (1: cannot run here; a device error -- memory, most likely -- : the other builder plans differently)
		const std::string why = rc < 0 ? bhip_last_error() : "";
		h->acx_rec.release(); h->acx_lines.release(); h->has_acx = false;
		(void)hipGetLastError();
		if (only_words) return fail(BHIP_E_DEVICE, "BHIP_ACX_BUILD=words: the word-sliced accelerator build cannot run here (%s)", rc == 1 ? "its preconditions" : why.c_str());
		if (getenv("BHIP_DEBUG")) fprintf(stderr, "[bhip] word-sliced accelerator build not possible here (%s): clump-sliced build\n", rc == 1 ? "its preconditions" : why.c_str());
	}
	return build_accelerator_by_clumps(h, K, z);
}
// The accelerator of a handle that has none (bhip_init with K = 0), built together with the other handles of a replicated database
// (include/burst_hip.h).  1 from the cooperative builder = not possible here, on some rank: every rank then builds the whole thing itself.
extern "C" int bhip_build_accelerator_shared(void *handle, int K, int part, int n_parts, bhip_share_fn share, void *ctx) {
	Handle *h = (Handle *)handle;
	if (!h) return fail(BHIP_E_ARG, "null handle");
	if (h->has_acx) return fail(BHIP_E_ARG, "the handle has an accelerator already");
	if (K < 4 || K > 15 || n_parts < 1 || part < 0 || part >= n_parts || (n_parts > 1 && !share)) return fail(BHIP_E_ARG, "bad arguments of the cooperative accelerator build");
	HIPCHK(hipSetDevice(h->device));
	if (n_parts > 1) {
		const int rc = build_accelerator_by_words(h, K, h->acx_z, part, n_parts, share, ctx);
		if (rc <= 0) return rc;
		h->acx_rec.release(); h->acx_lines.release(); h->has_acx = false;
		(void)hipGetLastError();
		if (getenv("BHIP_DEBUG")) fprintf(stderr, "[bhip] cooperative accelerator build not possible here (rank %d of %d): every rank builds alone\n", part, n_parts);
	}
	return bhip_build_accelerator(h, K, h->acx_z);
}

static int build_accelerator_by_clumps(Handle *h, int K, int z) {
	const uint32_t nC = h->n_clumps;
	const uint64_t nw = 1ull << (2 * K);
	const int cb = std::min(24, 47 - 2 * K);      // bits of a clump number inside a slice: word << cb | clump fits below the lane bits and the no-word bit
	if (cb < 1) return fail(BHIP_E_ARG, "word length %d is beyond the device builder", K);
	const bool dbg = getenv("BHIP_DEBUG") != nullptr;
	const auto t_begin = std::chrono::steady_clock::now();
	auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
	if (const char *ev = getenv("BHIP_TEST_ENTRY_BIAS")) h->acx_bias = strtoull(ev, nullptr, 0);
	h->K = K;
	// 1. expansion estimate and true expansion per clump -> BadList
	std::vector<unsigned long long> tsum(nC), nexp(nC);
	std::vector<uint8_t> is_bad(nC, 0);
	std::vector<uint32_t> badlist;
	DTmp d_bad, d_soff;
	{
		DTmp d_ts, d_nx;
		ARC(d_ts.reserve((size_t)nC * 8)); ARC(d_nx.reserve((size_t)nC * 8));
		HIPCHK(hipMemsetAsync(d_ts.p, 0, (size_t)nC * 8, h->stream)); HIPCHK(hipMemsetAsync(d_nx.p, 0, (size_t)nC * 8, h->stream));
		hipLaunchKernelGGL(k_acx_budget, dim3(std::min<uint32_t>((nC * 16u + 255u) / 256u, (uint32_t)h->n_cu * 16)), dim3(256), 0, h->stream, h->ref_lane.as<uint4>(), h->ref_off.as<uint64_t>(),
			h->clump_len.as<uint32_t>(), nC, h->tot_refs, K, z ? 1 : 0, d_ts.as<unsigned long long>(), d_nx.as<unsigned long long>());
		HIPCHK(hipGetLastError());
		HIPCHK(hipMemcpyAsync(tsum.data(), d_ts.p, (size_t)nC * 8, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipMemcpyAsync(nexp.data(), d_nx.p, (size_t)nC * 8, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	}
	const unsigned long long full_size = K > 14 ? 0x7FFFFFFFull : (1ull << 24);      // burst.c:3322
	for (uint32_t c = 0; c < nC; ++c) if (tsum[c] >= full_size) { is_bad[c] = 1; badlist.push_back(c); nexp[c] = 0; }
	std::vector<uint64_t> slot_off(nC + 1), item_off(nC + 1);
	slot_off[0] = item_off[0] = 0;
	for (uint32_t c = 0; c < nC; ++c) { slot_off[c + 1] = slot_off[c] + 16ull * h->h_clump_len[c]; item_off[c + 1] = item_off[c] + 16ull * h->h_clump_len[c] + nexp[c]; }
	ARC(d_bad.reserve((size_t)nC + 16)); ARC(d_soff.reserve(((size_t)nC + 1) * 8));
	HIPCHK(hipMemcpyAsync(d_bad.p, is_bad.data(), nC, hipMemcpyHostToDevice, h->stream));
	HIPCHK(hipMemcpyAsync(d_soff.p, slot_off.data(), ((size_t)nC + 1) * 8, hipMemcpyHostToDevice, h->stream));
	// 2. slices of clumps whose tuples fit the sort buffers (two 8-byte tuple arrays + the folded lane masks + the sort's own scratch; ONE budget: 33 bytes per tuple, planned
	// with slack) next to what is resident at that time, of at most 2^cb clumps each (the slice-local clump number in the tuple).  The records
	// (4 bytes per entry; their number is only known after the first pass) are not there yet while the lists are counted: the first
	// pass runs over slices as large as the sort allows, the second over slices that fit next to the records -- unless everything
	// fits at once under the safe estimate "one record per tuple", in which case one slice serves both passes and is sorted once.
	DTmp d_lens, d_cursor, k0, k1, v0, v1, nruns, tmp, d_xcur;
	ARC(d_lens.reserve(nw * 4 + 16));
	HIPCHK(hipMemsetAsync(d_lens.p, 0, nw * 4, h->stream));
	size_t free_b = 0, total_b = 0;
	HIPCHK(hipMemGetInfo(&free_b, &total_b));
	const uint64_t n_lines = (nw + BHIP_ACX_LINE_WORDS - 1) / BHIP_ACX_LINE_WORDS;
	uint64_t biggest = 0;
	for (uint32_t c = 0; c < nC; ++c) biggest = std::max(biggest, item_off[c + 1] - item_off[c]);
	if (biggest > 2147483000ull) return fail(BHIP_E_DEVICE, "a clump alone has %llu word tuples", (unsigned long long)biggest);
	long long forced_slice = 0;
	if (const char *ev = getenv("BHIP_MASK_SLICE")) forced_slice = atoll(ev);
	std::vector<uint32_t> cuts;
	uint32_t n_slices = 0;
	uint64_t cap_items = 0;
	auto plan_slices = [&](double room) -> int {       // (33 B of room per tuple: 18 B of tuple arrays and masks, rocPRIM's scratch, and 20 % of slack for whoever else allocates meanwhile)
		uint64_t slice_items = room > 0 ? (uint64_t)std::min<double>(2147483000.0, room * 0.8 / 33.0) : 0;
		if (forced_slice > 0) slice_items = std::max<uint64_t>((uint64_t)forced_slice, biggest);
		if (slice_items < biggest)
			return fail(BHIP_E_DEVICE, "not enough device memory to build the accelerator (a clump alone has %llu word tuples, %.1f GB to sort in)", (unsigned long long)biggest, room / 1e9);
		cuts.assign(1, 0);
		for (uint32_t c0 = 0; c0 < nC;) {
			uint32_t c1 = c0 + 1;
			while (c1 < nC && item_off[c1 + 1] - item_off[c0] <= slice_items && c1 - c0 < (1u << cb)) ++c1;
			cuts.push_back(c1); c0 = c1;
		}
		n_slices = (uint32_t)cuts.size() - 1;
		cap_items = 0;
		for (uint32_t s = 0; s < n_slices; ++s) cap_items = std::max(cap_items, item_off[cuts[s + 1]] - item_off[cuts[s]]);
		k0.release(); k1.release(); v0.release(); v1.release(); tmp.release();
		if (k0.reserve_exact(cap_items * 8 + 16) || k1.reserve_exact(cap_items * 8 + 16) || v0.reserve_exact(cap_items * 2 + 16)) {
			// somebody else took memory of this device since it was measured (a query sort on the ingest thread, another rank): smaller slices
			k0.release(); k1.release(); v0.release(); v1.release();
			(void)hipGetLastError();
			return 1;
		}
		return 0;
	};
	auto plan_slices_retry = [&](double room) -> int {
		for (int attempt = 0; attempt < 5; ++attempt, room *= 0.5) {
			const int rc = plan_slices(room);
			if (rc <= 0) return rc;
			if (forced_slice > 0) break;
		}
		return fail(BHIP_E_DEVICE, "not enough device memory for the accelerator build's sort buffers");
	};
	const double room_once = (double)free_b - (double)n_lines * 64.0 - (double)item_off[nC] * BHIP_REC_BYTES - (double)(256u << 20);
	bool one_plan = forced_slice > 0 || (room_once > 0 && room_once * 0.8 / 33.0 >= (double)item_off[nC] && item_off[nC] <= 2147483000ull);
	if (getenv("BHIP_TEST_TWO_PLANS") && forced_slice <= 0) one_plan = false;      // (test hook: a small database through the large databases' path -- counting pass, record area mapped beside it, second plan)
	ARC(plan_slices_retry(one_plan ? room_once : (double)free_b - (double)n_lines * 64.0 - (double)(256u << 20)));
	ARC(nruns.reserve(16)); ARC(d_xcur.reserve(16));
	// The record area is taken WHILE the first pass counts: one address range for the upper bound (a record per tuple), its memory mapped
	// chunk by chunk by a thread of its own (DBuf::reserve_growable) -- a process that follows another one on the device waits seconds for
	// its first large allocations (the memory of the process before is not back at once: 0.8 .. 5.5 s for these 216 GB, measured), and
	// that wait now lies beside 3.7 s of sorting instead of behind them.  What is mapped beyond the real size goes back after the pass.
	// Not with one plan (everything fits at once: the records are small) and not where the runtime has no virtual memory management.
	std::thread mapper;
	int map_rc = 0;
	bool rec_vmm = false;
	if (!one_plan && !getenv("BHIP_ACX_NO_PREMAP")) {
		size_t fb2 = 0, tb2 = 0;
		if (hipMemGetInfo(&fb2, &tb2) == hipSuccess && h->acx_rec.reserve_growable((size_t)item_off[nC] * BHIP_REC_BYTES + 16, h->device) == 0) {
			rec_vmm = true;
			const size_t spare_b = (size_t)6 << 30;      // (rocPRIM's scratch, the offset lines' scratch)
			const size_t limit = fb2 > spare_b ? std::min<size_t>((size_t)item_off[nC] * BHIP_REC_BYTES + 16, fb2 - spare_b) : 0;
			const int dev = h->device;
			DBuf *rec = &h->acx_rec;
			if (limit) mapper = std::thread([rec, limit, dev, &map_rc]() { if (hipSetDevice(dev) != hipSuccess) { map_rc = 1; return; } map_rc = rec->grow_to(limit) ? 1 : 0; });
		}
	}
	struct JoinMapper { std::thread &t; ~JoinMapper() { if (t.joinable()) t.join(); } } join_mapper{mapper};
	unsigned long long *ukeys = nullptr; uint16_t *umasks = nullptr; unsigned long long *spare = nullptr; uint32_t n_unique = 0;
	// tuples of slice s, sorted and folded: ukeys / umasks / n_unique (spare = the other key buffer, free for scratch)
	auto fold_slice = [&](uint32_t s, bool count_only) -> int {
		const uint32_t c0 = cuts[s], c1 = cuts[s + 1];
		const uint64_t n_slots = slot_off[c1] - slot_off[c0], n_items = item_off[c1] - item_off[c0];
		n_unique = 0;
		if (!n_items) return 0;
		HIPCHK(hipMemsetAsync(d_xcur.p, 0, 8, h->stream));
		hipLaunchKernelGGL(k_acx_extract, dim3(std::min<uint32_t>(c1 - c0, (uint32_t)h->n_cu * 32)), dim3(256), 0, h->stream, h->ref_lane.as<uint4>(),
			h->ref_off.as<uint64_t>(), h->clump_len.as<uint32_t>(), d_soff.as<uint64_t>(), d_bad.as<uint8_t>(), c0, c1, h->tot_refs, K, z ? 1 : 0,
			cb, k0.as<unsigned long long>(), (unsigned long long)n_slots, d_xcur.as<unsigned long long>());
		HIPCHK(hipGetLastError());
		size_t tb = 0;
		hipcub::DoubleBuffer<unsigned long long> dk(k0.as<unsigned long long>(), k1.as<unsigned long long>());
		// The slots of a slice are laid out clump after clump, and the radix sort is stable: ordering the WORD bits (and the no-word bit
		// above them) leaves the tuples of a word in ascending clump order -- 2 K + 1 bits, four passes, instead of 48 bits in six.  Only
		// a slice with IUPAC expansions (appended behind the slots, out of clump order) needs the clump bits sorted as well.
		const int bit0 = n_items == n_slots ? cb : 0;
		HIPCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, dk, (int)n_items, bit0, BHIP_ACX_KEYBITS, h->stream));
		ARC(tmp.reserve(tb));
		HIPCHK(hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, dk, (int)n_items, bit0, BHIP_ACX_KEYBITS, h->stream));
		unsigned long long *skeys = dk.Current();
		if (count_only) {      // first pass: the list lengths from the sorted tuples themselves (no folded copy is written or read)
			hipLaunchKernelGGL(k_acx_hist_sorted, dim3((uint32_t)h->n_cu * 16), dim3(256), 0, h->stream, skeys, (uint32_t)n_items, cb, d_lens.as<uint32_t>());
			HIPCHK(hipGetLastError());
			HIPCHK(hipStreamSynchronize(h->stream));
			return 0;
		}
		ukeys = dk.Alternate(); umasks = v0.as<uint16_t>(); spare = skeys;
		hipcub::TransformInputIterator<unsigned long long, AcxKeyOf, const unsigned long long *> kin(skeys, AcxKeyOf());
		hipcub::TransformInputIterator<uint16_t, AcxLanesOf, const unsigned long long *> vin(skeys, AcxLanesOf());
		size_t tb2 = 0;
		HIPCHK(hipcub::DeviceReduce::ReduceByKey(nullptr, tb2, kin, ukeys, vin, umasks, nruns.as<uint32_t>(), BitOrU16(), (int)n_items, h->stream));
		ARC(tmp.reserve(tb2));
		HIPCHK(hipcub::DeviceReduce::ReduceByKey(tmp.p, tb2, kin, ukeys, vin, umasks, nruns.as<uint32_t>(), BitOrU16(), (int)n_items, h->stream));
		HIPCHK(hipMemcpyAsync(&n_unique, nruns.p, 4, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
		return 0;
	};
	const uint32_t g = (uint32_t)h->n_cu * 16;
	// 3. first pass: list lengths
	// (one slice serves both passes when everything fits at once: it is folded, and its folded tuples stay for the second pass)
	for (uint32_t s = 0; s < n_slices; ++s) {
		const bool keep = one_plan && n_slices == 1;
		ARC(fold_slice(s, !keep));
		if (keep && n_unique) { hipLaunchKernelGGL(k_acx_hist, dim3(g), dim3(256), 0, h->stream, ukeys, n_unique, cb, d_lens.as<uint32_t>()); HIPCHK(hipGetLastError()); }
	}
	const double t_pass1 = since();
	if (!one_plan) { k0.release(); k1.release(); v0.release(); tmp.release(); }      // the first pass's sort buffers make room for the offset lines and the records
	uint64_t tot = 0; uint32_t maxlen = 0;
	ARC(acx_lines_from_lens(h, d_lens.as<uint32_t>(), nw, &tot, &maxlen));
	const uint32_t n_slices_1 = n_slices;
	const double t_a0 = since();
	if (mapper.joinable()) mapper.join();
	if (rec_vmm && !map_rc) {
		h->acx_rec.shrink_to(tot * BHIP_REC_BYTES + 16);
		if (h->acx_rec.grow_to(tot * BHIP_REC_BYTES + 16)) { h->acx_rec.release(); rec_vmm = false; (void)hipGetLastError(); }
	} else if (rec_vmm) { h->acx_rec.release(); rec_vmm = false; (void)hipGetLastError(); }
	if (!rec_vmm) ARC(h->acx_rec.reserve_exact(tot * BHIP_REC_BYTES + 16));
	if (!one_plan) {
		HIPCHK(hipMemGetInfo(&free_b, &total_b));
		ARC(plan_slices_retry((double)free_b - (double)(256u << 20)));
	}
	if (n_slices > 1 || !one_plan) { d_cursor.p = d_lens.p; d_cursor.cap = d_lens.cap; d_lens.p = nullptr; d_lens.cap = 0; HIPCHK(hipMemsetAsync(d_cursor.p, 0, nw * 4, h->stream)); }      // (the length table's memory)
	else d_lens.release();
	const bool refold = n_slices > 1 || !one_plan;
	const double t_alloc = since() - t_a0;      // the record area and the second plan's sort buffers: allocations
	// 4. second pass: the records (one slice: the folded tuples are still there)
	const uint32_t all_lanes = getenv("BHIP_NO_LANE_MASKS") ? 1u : 0u;
	for (uint32_t s = 0; s < n_slices; ++s) {
		if (refold) ARC(fold_slice(s, false));
		if (!n_unique) continue;
		uint32_t *head_in = (uint32_t *)spare, *head = head_in + n_unique;      // 8 bytes per tuple of scratch: the sorted key buffer
		hipLaunchKernelGGL(k_acx_heads, dim3(g), dim3(256), 0, h->stream, ukeys, n_unique, cb, head_in);
		HIPCHK(hipGetLastError());
		size_t tb = 0;
		HIPCHK(hipcub::DeviceScan::InclusiveScan(nullptr, tb, head_in, head, hipcub::Max(), (int)n_unique, h->stream));
		ARC(tmp.reserve(tb));
		HIPCHK(hipcub::DeviceScan::InclusiveScan(tmp.p, tb, head_in, head, hipcub::Max(), (int)n_unique, h->stream));
		hipLaunchKernelGGL(k_acx_fill, dim3(g), dim3(256), 0, h->stream, h->acx_view(), ukeys, umasks, head, n_unique, cb, cuts[s],
			refold ? d_cursor.as<uint32_t>() : (const uint32_t *)nullptr, (uint32_t *)h->acx_view().rec, all_lanes);
		HIPCHK(hipGetLastError());
		if (refold) { hipLaunchKernelGGL(k_acx_advance, dim3(g), dim3(256), 0, h->stream, ukeys, head, n_unique, cb, d_cursor.as<uint32_t>()); HIPCHK(hipGetLastError()); }
		HIPCHK(hipStreamSynchronize(h->stream));
	}
	ARC(set_badlist(h, badlist.data(), (uint32_t)badlist.size()));
	h->has_acx = true; h->n_ent = tot; h->has_masks = !all_lanes;
	if (dbg) fprintf(stderr, "[bhip] accelerator built on the device: K=%d, %llu entries from %llu word tuples in %u + %u slice(s), %zu clump(s) on the BadList, %.2f B per entry; %.2f s (%.2f s for the list lengths, %.2f s allocating the records and the second pass's buffers)\n",
		K, (unsigned long long)tot, (unsigned long long)item_off[nC], n_slices_1, n_slices, badlist.size(), tot ? (double)(h->acx_rec.cap + n_lines * 64) / (double)tot : 0.0, since(), t_pass1, t_alloc);
	if (dbg && rec_vmm) fprintf(stderr, "[bhip] record area: %zu chunks of 1 GiB mapped beside the first pass\n", h->acx_rec.n_mapped);
	return 0;
}

// a piece of a device array to or from host memory (for exchanges the caller stages itself)
extern "C" int bhip_device_copy(void *dst, const void *src, uint64_t bytes, int to_device) {
	if (bytes && (!dst || !src)) return fail(BHIP_E_ARG, "bad copy arguments");
	if (bytes) HIPCHK(hipMemcpy(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
	return BHIP_OK;
}

// entries [first, first + n) of the record area (word order), as clump ids (+ lane masks): for callers that write the lists piece by
// piece instead of holding all of them (33 G entries are 134 GB as 32-bit numbers)
extern "C" int bhip_acx_export_entries(void *handle, uint64_t first, uint64_t n_entries, uint32_t *clumps, uint16_t *masks) {
	Handle *h = (Handle *)handle;
	if (!h || (!clumps && n_entries)) return fail(BHIP_E_ARG, "null argument");
	if (!h->has_acx) return fail(BHIP_E_ARG, "handle has no accelerator");
	if (first > h->n_ent || n_entries > h->n_ent - first) return fail(BHIP_E_ARG, "entries [%llu, +%llu) beyond the accelerator's %llu", (unsigned long long)first, (unsigned long long)n_entries, (unsigned long long)h->n_ent);
	HIPCHK(hipSetDevice(h->device));
	const uint64_t piece = 1ull << 26;
	DTmp dc, dm;
	ARC(dc.reserve(std::min(piece, n_entries + 1) * 4)); if (masks) ARC(dm.reserve(std::min(piece, n_entries + 1) * 2));
	for (uint64_t e = 0; e < n_entries; e += piece) {
		const uint64_t n = std::min(piece, n_entries - e);
		hipLaunchKernelGGL(k_acx_rec_export, dim3((uint32_t)h->n_cu * 16), dim3(256), 0, h->stream, h->acx_view().rec, (unsigned long long)(h->acx_bias + first + e), n,
			dc.as<uint32_t>(), masks ? dm.as<uint16_t>() : (uint16_t *)nullptr);
		HIPCHK(hipGetLastError());
		HIPCHK(hipMemcpyAsync(clumps + e, dc.p, n * 4, hipMemcpyDeviceToHost, h->stream));
		if (masks) HIPCHK(hipMemcpyAsync(masks + e, dm.p, n * 2, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	}
	return BHIP_OK;
}

// the accelerator of a handle in the file's terms: Lens[4^K] (burst.c:3558), the clump ids of all lists in word order (and
// their lane masks), the BadList.  Any pointer may be NULL; *n_entries / *n_bad are always set.
extern "C" int bhip_acx_export(void *handle, uint32_t *lens, uint32_t *clumps, uint16_t *masks, uint64_t cap_entries, uint64_t *n_entries,
                               uint32_t *badlist, uint32_t cap_bad, uint32_t *n_bad) {
	Handle *h = (Handle *)handle;
	if (!h) return fail(BHIP_E_ARG, "null handle");
	if (!h->has_acx) return fail(BHIP_E_ARG, "handle has no accelerator");
	HIPCHK(hipSetDevice(h->device));
	if (n_entries) *n_entries = h->n_ent;
	if (n_bad) *n_bad = h->n_bad;
	const uint64_t nw = 1ull << (2 * h->K);
	if (lens) {
		DTmp d;
		ARC(d.reserve(nw * 4));
		hipLaunchKernelGGL(k_acx_lens_from_lines, dim3((uint32_t)h->n_cu * 32), dim3(256), 0, h->stream, h->acx_view(), nw, d.as<uint32_t>());
		HIPCHK(hipGetLastError());
		HIPCHK(hipMemcpyAsync(lens, d.p, nw * 4, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(hipStreamSynchronize(h->stream));
	}
	if (clumps) {
		if (cap_entries < h->n_ent) return fail(BHIP_E_CAPACITY, "entry buffer holds %llu, %llu needed", (unsigned long long)cap_entries, (unsigned long long)h->n_ent);
		ARC(bhip_acx_export_entries(handle, 0, h->n_ent, clumps, masks));
	}
	if (badlist) {
		if (cap_bad < h->n_bad) return fail(BHIP_E_CAPACITY, "BadList buffer holds %u, %u needed", cap_bad, h->n_bad);
		if (h->n_bad) HIPCHK(hipMemcpy(badlist, h->bad.p, (size_t)h->n_bad * 4, hipMemcpyDeviceToHost));
	}
	return BHIP_OK;
}
