// Does this box's HIP runtime support growing ONE device array in place (virtual memory management)?  Reserves a 256 GB address range,
// maps physical chunks behind each other, runs a kernel across the seams, times the mapping.  hipcc --offload-arch=gfx950 -O2 vmm_probe.hip -o vmm_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_fill(unsigned *p, size_t n, unsigned v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (unsigned)i; }
__global__ void k_sum(const unsigned *p, size_t n, unsigned long long *out) {
	unsigned long long s = 0;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
	atomicAdd(out, s);
}
int main() {
	int dev = 0; CK(hipSetDevice(dev));
	int vmm = 0; CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
	printf("hipDeviceAttributeVirtualMemoryManagementSupported = %d\n", vmm);
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
	size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
	printf("granularity %zu\n", gran);
	const size_t VA = 256ull << 30, CH = 1ull << 30;
	void *base = nullptr; CK(hipMemAddressReserve(&base, VA, gran, nullptr, 0));
	printf("reserved %zu GB at %p\n", VA >> 30, base);
	std::vector<hipMemGenericAllocationHandle_t> hs;
	hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
	auto t0 = std::chrono::steady_clock::now();
	const int NCH = 24;
	for (int i = 0; i < NCH; ++i) {
		hipMemGenericAllocationHandle_t hnd; CK(hipMemCreate(&hnd, CH, &prop, 0));
		CK(hipMemMap((char *)base + i * CH, CH, 0, hnd, 0));
		CK(hipMemSetAccess((char *)base + i * CH, CH, &acc, 1));
		hs.push_back(hnd);
		if (i == 1) {      // a kernel over the first two chunks while more get mapped behind them
			hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned *)base, 2 * CH / 4, 7u);
			CK(hipGetLastError());
		}
	}
	CK(hipDeviceSynchronize());
	double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	printf("mapped %d x 1 GB in %.3f s (%.1f ms per GB)\n", NCH, dt, dt * 1e3 / NCH);
	hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (unsigned *)base, NCH * CH / 4, 1u);
	unsigned long long *d_s, h_s = 0; CK(hipMalloc(&d_s, 8)); CK(hipMemset(d_s, 0, 8));
	t0 = std::chrono::steady_clock::now();
	hipLaunchKernelGGL(k_sum, dim3(8192), dim3(256), 0, 0, (const unsigned *)base, NCH * CH / 4, d_s);
	CK(hipMemcpy(&h_s, d_s, 8, hipMemcpyDeviceToHost));
	dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	const size_t n = NCH * CH / 4;
	unsigned long long expect = 0; for (size_t i = 0; i < n; ++i) expect += (unsigned)(1u + (unsigned)i);
	printf("sum over %d GB across the seams: %s (%.1f GB/s read)\n", NCH, h_s == expect ? "correct" : "WRONG", NCH * (double)CH / dt / 1e9);
	size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot)); printf("free %zu MB of %zu MB with %d GB mapped\n", fr >> 20, tot >> 20, NCH);
	for (int i = 0; i < NCH; ++i) { CK(hipMemUnmap((char *)base + i * CH, CH)); CK(hipMemRelease(hs[i])); }
	CK(hipMemAddressFree(base, VA));
	CK(hipMemGetInfo(&fr, &tot)); printf("free %zu MB after release\n", fr >> 20);
	printf("vmm probe ok\n");
	return 0;
}
