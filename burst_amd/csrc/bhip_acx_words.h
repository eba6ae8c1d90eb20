// burst_amd/csrc/bhip_acx_words.h -- the words of one reference lane (make_accelerator's window walk, burst.c:3343-3377), shared by
// the accelerator builders' kernels (bhip_acx.hip) and by a host program that checks the dword-at-a-time walk against the
// symbol-by-symbol one (tests/csrc/acx_words_host.cpp, tests/test_host_cpu.py: no device needed for the bit tricks).
#pragma once
#include <stdint.h>
#ifndef BHIP_WORDS_FN
#define BHIP_WORDS_FN __host__ __device__ __forceinline__
#endif
#ifndef __HIPCC__      /* (the host check is plain C++: the 16-byte load of the kernels as a struct) */
struct uint4 { uint32_t x, y, z, w; };
#endif
BHIP_WORDS_FN uint32_t amb_count(uint32_t code) { return (uint32_t)((0x3333222222411110ull >> (4u * code)) & 15u); }
BHIP_WORDS_FN uint32_t amb_bases(uint32_t code) {      // up to four 2-bit bases, first option in the low bits
	return (uint32_t)(((code < 8 ? 0x040EE40302010000ull : 0x383424390C090D08ull) >> (8u * (code & 7u))) & 255u);
}
BHIP_WORDS_FN unsigned long long amb_product(unsigned long long win, int K) {
	unsigned long long p = 1;
	for (int t = 0; t < K; ++t) p *= amb_count((uint32_t)(win >> (4 * t)) & 15u);
	return p;
}

// every word of one reference lane: emit(word) for each window of K symbols A/C/G/T, and for each IUPAC expansion of an ambiguous one
// (the order of the calls is not the order of the positions' -- nobody needs it).
// A dword of eight symbols at a time: where all eight are A/C/G/T (every dword of a database without ambiguity codes, nearly every one of
// a real one) their 2-bit codes are packed with bit tricks, newest symbol lowest, behind the codes of the dwords before in one 64-bit
// register, and -- with K - 1 such symbols in front of the dword -- the eight windows are eight shifts of it -- 8 instructions per position instead of 22 for the symbol-by-symbol walk, which
// is what a scan of the references costs (and this builder scans them once per slice).  A dword with anything else in it (ambiguity codes,
// padding, the lane's end) takes the symbol-by-symbol walk, its window state rebuilt from the two dwords before (16 symbols >= K - 1).
template <class F>
BHIP_WORDS_FN void acx_lane_words(const uint4 *__restrict__ rp, uint32_t L, uint32_t nchunks, int K, int z, F &&emit) {
	const uint32_t wmask = (1u << (2 * K)) - 1u;
	unsigned long long S = 0;          // 2-bit codes of the symbols so far, the newest in the lowest bits
	uint32_t litrun = 0;               // A/C/G/T symbols in a row up to the end of the dword before (saturating)
	uint32_t prev1 = 0, prev2 = 0;     // the two dwords before (prev1 the nearer)
	for (uint32_t t = 0; t < nchunks; ++t) {
		const uint4 ch = rp[t];
		const uint32_t dw[4] = {ch.x, ch.y, ch.z, ch.w};
		#pragma unroll
		for (uint32_t j = 0; j < 4; ++j) {
			const uint32_t base = t * 32 + j * 8;
			if (base >= L) return;
			const uint32_t x = dw[j];
			const uint32_t y = (x | 0x88888888u) - 0x11111111u;                       // per nibble: (symbol - 1) mod 8 in the low three bits
			const uint32_t litm = (~y >> 2) & (~x >> 3) & 0x11111111u;                // bit 4k: symbol k is one of 1..4
			uint32_t p = y & 0x33333333u;                                             // 2-bit codes, one per nibble ...
			p = (p | (p >> 2)) & 0x0F0F0F0Fu; p = (p | (p >> 4)) & 0x00FF00FFu; p = (p | (p >> 8)) & 0xFFFFu;      // ... side by side, symbol 0 lowest
			uint32_t r = __builtin_bitreverse32(p) >> 16;                                             // symbol 7 lowest (the bits of a code swapped: put back)
			r = ((r & 0x5555u) << 1) | ((r >> 1) & 0x5555u);
			S = (S << 16) | r;
			if (litm == 0x11111111u && base + 8 <= L && litrun + 1 >= (uint32_t)K) {      // (K - 1 literals before the dword: none of its windows reaches an ambiguity code)
				#pragma unroll
				for (uint32_t k = 0; k < 8; ++k) emit((uint32_t)(S >> (2u * (7u - k))) & wmask);
				litrun = litrun + 8 > 64 ? 64 : litrun + 8;
			} else {
				unsigned long long win = 0;
				uint32_t w = 0, run = 0, lit = 0;
				const uint32_t g = base >> 3;
				for (uint32_t d = g >= 2 ? 0u : 2u - g; d < 3; ++d) {      // d = 0, 1: warm-up over prev2, prev1 (as far as the lane has them); 2: this dword
					const uint32_t v = d == 0 ? prev2 : d == 1 ? prev1 : x;
					#pragma unroll
					for (uint32_t k = 0; k < 8; ++k) {
						const uint32_t sym = (v >> (4 * k)) & 15u;
						run = (sym >= 1u && !(z && sym == 5u)) ? run + 1 : 0;
						lit = (sym - 1u) < 4u ? lit + 1 : 0;
						w = ((w << 2) | ((sym - 1u) & 3u)) & wmask;
						win = (win << 4) | sym;
						if (d != 2 || base + k >= L) continue;
						if (lit >= (uint32_t)K) emit(w);
						else if (run >= (uint32_t)K) {
							const unsigned long long prod = amb_product(win, K);
							for (unsigned long long idx = 0; idx < prod; ++idx) {
								unsigned long long rr = idx; uint32_t word = 0;
								for (int q = 0; q < K; ++q) {          // symbol q counted from the window's end: 2-bit digit q of the word
									const uint32_t code = (uint32_t)(win >> (4 * q)) & 15u, n = amb_count(code), dgt = (uint32_t)(rr % n);
									rr /= n;
									word |= ((amb_bases(code) >> (2u * dgt)) & 3u) << (2 * q);
								}
								emit(word);
							}
						}
					}
				}
				const uint32_t inv = ~litm & 0x11111111u;      // literals at the dword's end: the nibbles above the highest one that is not
				litrun = inv ? (uint32_t)(__builtin_clz(inv) - 3) >> 2 : (litrun + 8 > 64 ? 64 : litrun + 8);
			}
			prev2 = prev1; prev1 = x;
		}
	}
}
