// burst_amd/csrc/bhip_lanecode.h -- the lane-set code in the top byte of a 4-byte accelerator record (bhip_internal.h): plain C,
// so that the CPU tests can compile it on its own (tests/test_host_cpu.py checks every one of the 65 536 masks).
#ifndef BHIP_LANECODE_H
#define BHIP_LANECODE_H
#include <stdint.h>
#ifndef __HIPCC__
#define BHIP_HD static inline
#else
#define BHIP_HD __host__ __device__ inline
#endif
#define BHIP_LANES_ALL 166u          /* lane-set code "every lane" */

// lane-set code -> 16-bit lane mask
BHIP_HD uint32_t bhip_lane_code_mask(uint32_t code) {
	if (code < 16u) return 1u << code;
	if (code < 136u) {                                   // pair a < b: 16 + a (31 - a) / 2 + (b - a - 1)
		uint32_t idx = code - 16u, a = 0;
		while (idx >= 15u - a) { idx -= 15u - a; ++a; }
		return (1u << a) | (1u << (a + 1u + idx));
	}
	if (code < 156u) {                                   // inside quad q: t = 0..3 all but lane t, t = 4 the whole quad
		const uint32_t q = (code - 136u) / 5u, t = (code - 136u) % 5u;
		return (t == 4u ? 0xFu : (0xFu & ~(1u << t))) << (4u * q);
	}
	if (code <= 166u) {                                  // union of two or more quads, by the 4-bit set of quads
		const uint32_t qm = (uint32_t)((0xFEDCBA97653ull >> (4u * (code - 156u))) & 15u);
		return ((qm & 1u) ? 0x000Fu : 0u) | ((qm & 2u) ? 0x00F0u : 0u) | ((qm & 4u) ? 0x0F00u : 0u) | ((qm & 8u) ? 0xF000u : 0u);
	}
	return 0xFFFFu;
}
// 16-bit lane mask -> the code of its smallest superset (0 = "not known" -> every lane)
BHIP_HD uint32_t bhip_lane_mask_code(uint32_t mask) {
	mask &= 0xFFFFu;
	const uint32_t pc = (uint32_t)__builtin_popcount(mask);
	if (pc == 0u) return BHIP_LANES_ALL;
	const uint32_t a = (uint32_t)__builtin_ctz(mask);
	if (pc == 1u) return a;
	if (pc == 2u) { const uint32_t b = 31u - (uint32_t)__builtin_clz(mask); return 16u + a * (31u - a) / 2u + (b - a - 1u); }
	const uint32_t qm = ((mask & 0x000Fu) ? 1u : 0u) | ((mask & 0x00F0u) ? 2u : 0u) | ((mask & 0x0F00u) ? 4u : 0u) | ((mask & 0xF000u) ? 8u : 0u);
	if ((qm & (qm - 1u)) == 0u) {                         // one quad
		const uint32_t q = (uint32_t)__builtin_ctz(qm), sub = (mask >> (4u * q)) & 15u;
		return 136u + 5u * q + (sub == 15u ? 4u : (uint32_t)__builtin_ctz(~sub & 15u));
	}
	// quad sets 3 5 6 7 9 10 11 12 13 14 15 -> 0 .. 10
	return 156u + (uint32_t)((0xA987654032100000ull >> (4u * qm)) & 15u);
}

#endif
