// burst_amd/csrc/bhip_pf_common.h -- what the prefilter kernels of the three translation units share (bhip_prefilter.hip: the product's
// k_seed_ranges + k_prefilter_cq; bhip_prefilter_alt.hip: the fallbacks the product reaches; bhip_prefilter_legacy.hip: the superseded
// counting-filter kernels, a test-only library): phase timers, wave-order fence, wave-wide scans.
#ifndef BHIP_PF_COMMON_H
#define BHIP_PF_COMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "burst_hip.h"
#include "bhip_internal.h"
#ifdef PFM_PROF
extern __device__ unsigned long long g_pfm_prof[8];      // (defined by the one translation unit that times its phases: bhip_prefilter.hip)
#if PFM_PROF == 2      // without draining the memory pipeline: issue + stall time of each phase as it really runs
#define PFM_T(i) do { const unsigned long long t_ = wall_clock64(); if (lane == 0) my_t[i] += t_ - t_last; t_last = t_; } while (0)
#else
#define PFM_T(i) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = wall_clock64(); if (lane == 0) my_t[i] += t_ - t_last; t_last = t_; } while (0)
#endif
#else
#define PFM_T(i) do {} while (0)
#endif
#define PFM_STAGE 128u
#define PFM_RB 3u            // blocks of 64 records per query kept in registers between the passes
__device__ __forceinline__ unsigned long long spread8(uint32_t m8) {   // bit i of m8 -> bit 8*i
	unsigned long long x = m8;
	x = (x | (x << 28)) & 0x0000000F0000000Full;
	x = (x | (x << 14)) & 0x0003000300030003ull;
	x = (x | (x << 7)) & 0x0101010101010101ull;
	return x;
}
// Four hash-table updates in lock step (independent LDS round trips overlap).  CAS first: most updates of a
// query are first sightings of a clump, which complete in one round trip; a key hit costs one more (no-return) add.
template <int HTB>
__device__ __forceinline__ void pfm_bump4(uint32_t *tab, uint32_t *dummy, const uint32_t (&c)[4], const bool (&valid)[4], uint32_t (&slot)[4], bool (&ins)[4], bool &fail) {
	uint32_t key[4]; bool act[4];
	#pragma unroll
	for (int k = 0; k < 4; ++k) { key[k] = (c[k] + 1u) << 8; slot[k] = (c[k] * 0x9E3779B1u) >> (32 - HTB); act[k] = valid[k]; ins[k] = false; }
	bool any = valid[0] | valid[1] | valid[2] | valid[3];
	for (uint32_t probes = 0; any && probes < (1u << HTB); ++probes) {
		uint32_t old[4];
		// finished chains compare-and-swap a private dummy word with a value that never matches: no branches between the
		// four LDS round trips, so they are in flight together
		#pragma unroll
		for (int k = 0; k < 4; ++k) old[k] = atomicCAS(act[k] ? &tab[slot[k]] : dummy, act[k] ? 0u : 0xFFFFFFFFu, key[k] | 1u);
		any = false;
		#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const bool hit = act[k] && (old[k] & 0xFFFFFF00u) == key[k];
			const bool fresh = act[k] && old[k] == 0;
			const bool step = act[k] && !hit && !fresh;
			if (hit) atomicAdd(&tab[slot[k]], 1u);
			ins[k] |= fresh;
			slot[k] = step ? (slot[k] + 1) & ((1u << HTB) - 1) : slot[k];
			act[k] = step;
			any |= step;
		}
	}
	fail = any;
}
// The workgroup of this kernel is ONE wave: its LDS operations are issued and completed in program order, so a later read sees an
// earlier update by any lane without a barrier.  __syncthreads() would still cost an s_waitcnt vmcnt(0) lgkmcnt(0) -- a wait for
// every load in flight, i.e. for the record prefetch of the NEXT quad that the software pipeline has just issued.  What the phases
// need between them is only that the compiler keeps their LDS accesses in order.  (-DCF_BARRIERS=1 puts the barriers back.)
#if defined(CF_BARRIERS) && CF_BARRIERS
#define CF_WAVE_ORDER() __syncthreads()
#else
#define CF_WAVE_ORDER() __asm__ volatile("" ::: "memory")
#endif
// inclusive prefix sum over the 64 lanes of a wave (all lanes active): four shifts inside each row of 16 lanes, then lane 15 of
// rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2 and 3 -- data-parallel-primitive moves, no LDS round trip
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x) {
	int v = (int)x;
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);    // row_shr:1
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);    // row_shr:2
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);    // row_shr:4
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);    // row_shr:8
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);    // row_bcast:15 -> rows 1, 3
	v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);    // row_bcast:31 -> rows 2, 3
	return (uint32_t)v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {      // maximum over the 64 lanes (all active), in every lane
	int v = (int)x, t;
	t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false); v = (uint32_t)t > (uint32_t)v ? t : v;    // row_shr:1
	t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false); v = (uint32_t)t > (uint32_t)v ? t : v;    // row_shr:2
	t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false); v = (uint32_t)t > (uint32_t)v ? t : v;    // row_shr:4
	t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false); v = (uint32_t)t > (uint32_t)v ? t : v;    // row_shr:8
	const uint32_t a = (uint32_t)__builtin_amdgcn_readlane(v, 15), b = (uint32_t)__builtin_amdgcn_readlane(v, 31),
		c = (uint32_t)__builtin_amdgcn_readlane(v, 47), d = (uint32_t)__builtin_amdgcn_readlane(v, 63);
	const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
	return ab > cd ? ab : cd;
}
#endif
