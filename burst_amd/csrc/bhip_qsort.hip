// burst_amd/csrc/bhip_qsort.hip -- query sort + duplicate marking on the device (gfx950).
// The reference sorts the query records by sequence and folds identical ones (burst.c:636-690 qsort with the comparator of
// 363-366, uniqueness 3036-3053); the C host does the same on 32 threads (host/bh_queries.c: bucket by the first five symbols +
// qsort).  Here: an LSD radix sort over the 4-bit symbol codes, sixteen symbols (one 64-bit key) per pass, most significant
// chunk last, with the length as the least significant key and the record number as the implicit last one (every pass is
// stable).  A position past a record's end reads as code 0, the smallest code: two records then compare as memcmp over the
// shorter length followed by "shorter first" -- also when real symbols of code 0 are present, because equal padded keys are
// separated by the length pass.  Duplicates = neighbours of equal length and bytes.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include "burst_hip.h"

int bhip_fail_msg(int code, const char *fmt, ...);      // bhip_api.hip: sets the calling thread's error text

__global__ void k_qs_iota(uint32_t *idx, uint32_t n) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) idx[i] = i;
}
__global__ void k_qs_len_keys(const uint32_t *__restrict__ idx, const uint32_t *__restrict__ len, uint32_t n, uint32_t *__restrict__ keys) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) keys[i] = len[idx[i]];
}
// key of record idx[i] for symbols [16 * chunk, 16 * chunk + 16): first symbol in the top nibble
__global__ void k_qs_chunk_keys(const uint32_t *__restrict__ idx, const uint64_t *__restrict__ start, const uint32_t *__restrict__ len,
                                const uint8_t *__restrict__ codes, uint32_t n, uint32_t chunk, unsigned long long *__restrict__ keys) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t r = idx[i], L = len[r], p0 = chunk * 16u;
		const uint8_t *s = codes + start[r];
		unsigned long long k = 0;
		#pragma unroll
		for (uint32_t j = 0; j < 16; ++j) k = (k << 4) | (unsigned long long)(p0 + j < L ? (s[p0 + j] & 15u) : 0u);
		keys[i] = k;
	}
}
__global__ void k_qs_is_new(const uint32_t *__restrict__ perm, const uint64_t *__restrict__ start, const uint32_t *__restrict__ len,
                            const uint8_t *__restrict__ codes, uint32_t n, uint8_t *__restrict__ is_new) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		uint8_t fresh = 1;
		if (i) {
			const uint32_t a = perm[i], b = perm[i - 1], L = len[a];
			if (L == len[b]) {
				const uint8_t *x = codes + start[a], *y = codes + start[b];
				uint32_t k = 0;
				while (k < L && x[k] == y[k]) ++k;
				fresh = k < L;
			}
		}
		is_new[i] = fresh;
	}
}

#define QCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = bhip_fail_msg(BHIP_E_DEVICE, "%s: %s", #x, hipGetErrorString(e_)); goto done; } } while (0)

extern "C" int bhip_sort_queries(int device, const uint8_t *codes, uint64_t codes_bytes, const uint64_t *start, const uint32_t *len,
                                 uint64_t n64, uint32_t max_len, uint32_t *perm, uint8_t *is_new) {
	if (!codes || !start || !len || !perm || !is_new) return bhip_fail_msg(BHIP_E_ARG, "null argument");
	if (n64 >= 0x7FFFFFFFull) return bhip_fail_msg(BHIP_E_ARG, "too many records for the device sort");
	const uint32_t n = (uint32_t)n64;
	if (!n) return BHIP_OK;
	int rc = BHIP_OK, n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) { (void)hipGetLastError(); return bhip_fail_msg(BHIP_E_DEVICE, "no such device"); }
	uint8_t *d_codes = nullptr, *d_new = nullptr; uint64_t *d_start = nullptr; uint32_t *d_len = nullptr, *d_idx[2] = {nullptr, nullptr}, *d_k32[2] = {nullptr, nullptr};
	unsigned long long *d_k64[2] = {nullptr, nullptr};
	void *d_tmp = nullptr; size_t tmp_bytes = 0, tb = 0;
	const uint32_t grid = 256 * 16, chunks = (max_len + 15) / 16;
	int cur = 0;
	const bool dbg = getenv("BHIP_DEBUG") != nullptr || getenv("BURST_HOST_DEBUG") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
	double t_alloc = 0, t_up = 0, t_sort = 0;
	bool pinned = false;
	QCHK(hipSetDevice(device));
	QCHK(hipMalloc(&d_codes, codes_bytes + 16)); QCHK(hipMalloc(&d_start, (size_t)n * 8)); QCHK(hipMalloc(&d_len, (size_t)n * 4));
	for (int b = 0; b < 2; ++b) { QCHK(hipMalloc(&d_idx[b], (size_t)n * 4)); QCHK(hipMalloc(&d_k64[b], (size_t)n * 8)); }
	d_k32[0] = (uint32_t *)d_k64[0]; d_k32[1] = (uint32_t *)d_k64[1];
	QCHK(hipMalloc(&d_new, n));
	QCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_k64[0], d_k64[1], d_idx[0], d_idx[1], (int)n, 0, 64, (hipStream_t)0));
	QCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, d_k32[0], d_k32[1], d_idx[0], d_idx[1], (int)n, 0, 32, (hipStream_t)0));
	if (tb > tmp_bytes) tmp_bytes = tb;
	QCHK(hipMalloc(&d_tmp, tmp_bytes + 16));
	t_alloc = ms();
	// (page-locking the text for the one copy pays: ~15 ms per 500 MB against a pageable copy at a fifth of the link rate)
	pinned = hipHostRegister((void *)codes, codes_bytes, hipHostRegisterDefault) == hipSuccess;
	if (!pinned) (void)hipGetLastError();
	QCHK(hipMemcpy(d_codes, codes, codes_bytes, hipMemcpyHostToDevice));
	QCHK(hipMemcpy(d_start, start, (size_t)n * 8, hipMemcpyHostToDevice));
	QCHK(hipMemcpy(d_len, len, (size_t)n * 4, hipMemcpyHostToDevice));
	t_up = ms();
	hipLaunchKernelGGL(k_qs_iota, dim3(grid), dim3(256), 0, 0, d_idx[0], n);
	{	// least significant key: the length (bits that can be set only)
		int bits = 1; while (bits < 32 && (max_len >> bits)) ++bits;
		hipLaunchKernelGGL(k_qs_len_keys, dim3(grid), dim3(256), 0, 0, d_idx[cur], d_len, n, d_k32[cur]);
		tb = tmp_bytes;
		QCHK(hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, d_k32[cur], d_k32[cur ^ 1], d_idx[cur], d_idx[cur ^ 1], (int)n, 0, bits, (hipStream_t)0));
		cur ^= 1;
	}
	for (uint32_t c = chunks; c-- > 0;) {
		hipLaunchKernelGGL(k_qs_chunk_keys, dim3(grid), dim3(256), 0, 0, d_idx[cur], d_start, d_len, d_codes, n, c, d_k64[cur]);
		tb = tmp_bytes;
		QCHK(hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, d_k64[cur], d_k64[cur ^ 1], d_idx[cur], d_idx[cur ^ 1], (int)n, 0, 64, (hipStream_t)0));
		cur ^= 1;
	}
	hipLaunchKernelGGL(k_qs_is_new, dim3(grid), dim3(256), 0, 0, d_idx[cur], d_start, d_len, d_codes, n, d_new);
	QCHK(hipGetLastError());
	QCHK(hipDeviceSynchronize());
	t_sort = ms();
	QCHK(hipMemcpy(perm, d_idx[cur], (size_t)n * 4, hipMemcpyDeviceToHost));
	QCHK(hipMemcpy(is_new, d_new, n, hipMemcpyDeviceToHost));
	if (dbg) fprintf(stderr, "[bhip] query sort on the device: %u records, %u key passes; allocations %.1f ms, copies in %.1f ms (%s), sort + duplicate marks %.1f ms, copies out %.1f ms\n",
		n, chunks + 1, t_alloc, t_up - t_alloc, pinned ? "page-locked" : "pageable", t_sort - t_up, ms() - t_sort);
done:
	if (pinned) (void)hipHostUnregister((void *)codes);
	(void)hipFree(d_codes); (void)hipFree(d_start); (void)hipFree(d_len); (void)hipFree(d_idx[0]); (void)hipFree(d_idx[1]);
	(void)hipFree(d_k64[0]); (void)hipFree(d_k64[1]); (void)hipFree(d_new); (void)hipFree(d_tmp);
	if (rc) (void)hipGetLastError();
	return rc;
}
