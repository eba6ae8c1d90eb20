// tools/ubench/run_gather.hip -- the prefilter's access pattern with KNOWN traffic: a wave walks a flattened stream of RUNS of `run`
// consecutive dwords, every run at a random 4-byte-aligned address of a large buffer (an accelerator list of ~40 four-byte records),
// 64 adjacent lanes = 64 adjacent stream positions (a row of k_prefilter_cq: ~1.6 lists).  The host knows, from the same hash, how many
// distinct 32-byte / 64-byte / 128-byte pieces the runs touch: under `rocprofv3 --pmc FETCH_SIZE` the counter is compared with those
// three predictions (tools/calibrate_fetch.sh) -- which granularity FETCH_SIZE tallies for THIS pattern, and hence what factor turns it
// into bytes moved.  Also prints the rate (useful GB/s and 64-byte sectors/s) the pattern reaches.
//   hipcc --offload-arch=gfx950 -O3 run_gather.hip -o run_gather;  ./run_gather [GiB=12] [run=40] [iters=64] [waves_per_simd=8]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
__host__ __device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
template <int U>
__global__ __launch_bounds__(256) void k_runs(const uint32_t *__restrict__ buf, uint64_t n_dwords, uint32_t run, int iters, uint32_t *out, uint64_t seed) {
	const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t acc = 0;
	for (int it = 0; it < iters; ++it) {
		uint32_t v[U];
		#pragma unroll
		for (int u = 0; u < U; ++u) {
			const uint64_t p = ((wave * iters + it) * U + u) * 64 + lane;      // stream position
			const uint64_t r = p / run, o = p % run;
			const uint64_t base = mix(seed + r) % (n_dwords - run);
			v[u] = buf[base + o];
		}
		#pragma unroll
		for (int u = 0; u < U; ++u) acc += v[u];
	}
	out[(uint64_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main(int argc, char **argv) {
	const uint64_t ws = (argc > 1 ? strtoull(argv[1], 0, 10) : 12ull) << 30;
	const uint32_t run = argc > 2 ? (uint32_t)atoi(argv[2]) : 40u;
	const int iters = argc > 3 ? atoi(argv[3]) : 64, wps = argc > 4 ? atoi(argv[4]) : 8;
	constexpr int U = 4;
	uint32_t *buf, *out;
	if (hipMalloc(&buf, ws) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
	const int blocks = 256 * wps, threads = 256;
	(void)hipMalloc(&out, (size_t)blocks * threads * 4);
	(void)hipMemset(buf, 1, ws);
	const uint64_t n_dwords = ws / 4, seed = 12345;
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	k_runs<U><<<blocks, threads>>>(buf, n_dwords, run, iters, out, 1); (void)hipDeviceSynchronize();      // warm-up (another seed: other addresses)
	(void)hipEventRecord(e0);
	k_runs<U><<<blocks, threads>>>(buf, n_dwords, run, iters, out, seed);                             // the measured launch = the LAST dispatch of k_runs
	(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
	float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
	// what the measured launch touched
	const uint64_t positions = (uint64_t)blocks * threads / 64 * iters * U * 64;
	const uint64_t n_runs = (positions + run - 1) / run;
	uint64_t s32 = 0, s64 = 0, s128 = 0;
	for (uint64_t r = 0; r < n_runs; ++r) {
		const uint64_t b = (mix(seed + r) % (n_dwords - run)) * 4;
		const uint64_t len = (r + 1 == n_runs && positions % run ? positions % run : run) * 4ull;
		s32 += (b % 32 + len + 31) / 32; s64 += (b % 64 + len + 63) / 64; s128 += (b % 128 + len + 127) / 128;
	}
	// (a run split over two rows of 64 positions is fetched by two instructions: pieces at the seam may be requested twice; upper bound of that effect)
	const uint64_t seams = positions / 64;
	printf("run_gather: work set %.1f GiB, runs of %u dwords, %llu positions in %llu runs, %d waves/SIMD, %.3f ms\n", ws / 1073741824.0, run, (unsigned long long)positions, (unsigned long long)n_runs, wps, ms);
	printf("useful_bytes %llu  bytes_as_32B_pieces %llu  bytes_as_64B_pieces %llu  bytes_as_128B_pieces %llu  row_seams %llu\n",
		(unsigned long long)(positions * 4), (unsigned long long)(s32 * 32), (unsigned long long)(s64 * 64), (unsigned long long)(s128 * 128), (unsigned long long)seams);
	printf("rate: %.1f GB/s useful, %.2f G 64-byte sectors/s (%.1f GB/s of sectors)\n", positions * 4 / (ms * 1e-3) / 1e9, s64 / (ms * 1e-3) / 1e9, s64 * 64 / (ms * 1e-3) / 1e9);
	return 0;
}
