// tools/ubench/gather_rate.hip -- rate of random 64-byte sector reads (one dword per lane, every lane its own sector) as a
// function of the working set and of the waves in flight: what the seed lookup and the list gather of the prefilter can get
// from HBM + address translation on gfx950.  hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
template <int U>
__global__ __launch_bounds__(256) void k_gather(const uint32_t *__restrict__ buf, uint64_t n_sectors, int iters, uint32_t *out, uint64_t seed) {
	const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t acc = 0;
	for (int it = 0; it < iters; ++it) {
		uint32_t v[U];
		#pragma unroll
		for (int u = 0; u < U; ++u) {
			const uint64_t s = mix(seed + (tid * iters + it) * U + u) % n_sectors;
			v[u] = buf[s * 16];
		}
		#pragma unroll
		for (int u = 0; u < U; ++u) acc += v[u];
	}
	out[tid] = acc;
}
template <int U> double run(const uint32_t *buf, uint64_t ws_bytes, int waves_per_simd, uint32_t *out) {
	const int blocks = 256 * waves_per_simd, threads = 256, iters = 64;      // 4 waves per block = one per SIMD
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k_gather<U><<<blocks, threads>>>(buf, ws_bytes / 64, iters, out, 1); hipDeviceSynchronize();
	hipEventRecord(e0);
	k_gather<U><<<blocks, threads>>>(buf, ws_bytes / 64, iters, out, 12345);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	return (double)blocks * threads * iters * U / (ms * 1e-3) / 1e9;      // G sector reads per second
}
int main(int argc, char **argv) {
	const uint64_t max_ws = (argc > 1 ? strtoull(argv[1], 0, 10) : 12ull) << 30;
	uint32_t *buf, *out;
	if (hipMalloc(&buf, max_ws) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
	hipMalloc(&out, 256 * 8 * 256 * 4);
	hipMemset(buf, 1, max_ws);
	printf("random 64-byte sector reads, one dword per lane; G sectors/s (x64 = GB/s)\n%10s %6s %8s %8s %8s\n", "work set", "w/SIMD", "U=1", "U=4", "U=8");
	const uint64_t sizes[] = {64ull << 20, 512ull << 20, 2ull << 30, 6ull << 30, max_ws};
	for (uint64_t ws : sizes) for (int w : {2, 4, 8}) {
		if (ws > max_ws) continue;
		printf("%8.2fGB %6d %8.1f %8.1f %8.1f\n", ws / 1073741824.0, w, run<1>(buf, ws, w, out), run<4>(buf, ws, w, out), run<8>(buf, ws, w, out));
	}
	return 0;
}
