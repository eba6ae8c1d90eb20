// burst_amd/csrc/bhip_internal.h -- types shared between the kernels and the C-ABI host code.
#ifndef BHIP_INTERNAL_H
#define BHIP_INTERNAL_H
#include <stdint.h>

// match_mask[a] bit b = 1 iff cost(a, b) == 0 in the 16x16 table (burst.c:1310-1328 after setScore()).
struct BhipMatchMask { uint16_t m[16]; };

// One (query entry, reference lane) with ed <= budget, as left by k_myers for k_rescore.
struct BhipRawHit {
	uint32_t q;        // batch index of the query entry
	uint32_t refIx;    // 16*clump + lane
	uint32_t ed;
	uint32_t e_first;  // first / last 1-based end column whose last-row score equals ed
	uint32_t e_last;   // (may run into trailing pad columns; k_rescore clamps to ClumpLen)
};

// A reference lane whose prefix filter fired: flags bit b covers chunks [b << fshift, (b+1) << fshift) of 32 columns.
struct BhipWin {
	uint32_t li;       // list position of the query (peq row)
	uint32_t refIx;
	uint32_t flags;
};

#define BHIP_RESCORE_WMAX 48   // band widths up to this many diagonals are handled in LDS

#endif
