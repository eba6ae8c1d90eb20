// The dword-at-a-time window walk of the accelerator builders (burst_amd/csrc/bhip_acx_words.h: acx_lane_words) against the
// symbol-by-symbol walk it replaced (make_accelerator's loop, burst.c:3343-3377), on the host: random lanes with and without
// ambiguity codes, padding, junk behind the lane's end, K = 4 .. 15, both N rules.  The multiset of emitted words must be the same.
// Test infrastructure (tests/test_host_cpu.py): g++ -O1, no device.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#define BHIP_WORDS_FN static inline
#define __restrict__
static inline uint32_t bitrev32(uint32_t x) { uint32_t r = 0; for (int i = 0; i < 32; ++i) if (x >> i & 1) r |= 1u << (31 - i); return r; }
#define __builtin_bitreverse32 bitrev32
#include "bhip_acx_words.h"

// the walk of the first builders: one symbol at a time, run lengths of literal / unpenalised symbols, IUPAC expansion of ambiguous windows
template <class F> static void serial_words(const uint32_t *dw, uint32_t L, int K, int z, F &&emit) {
	const uint32_t wmask = (1u << (2 * K)) - 1u;
	unsigned long long win = 0; uint32_t w = 0, run = 0, lit = 0;
	for (uint32_t pos = 0; pos < L; ++pos) {
		const uint32_t sym = (dw[pos >> 3] >> (4 * (pos & 7))) & 15u;
		run = (sym >= 1u && !(z && sym == 5u)) ? run + 1 : 0;
		lit = (sym - 1u) < 4u ? lit + 1 : 0;
		w = ((w << 2) | ((sym - 1u) & 3u)) & wmask;
		win = (win << 4) | sym;
		if (lit >= (uint32_t)K) emit(w);
		else if (run >= (uint32_t)K) {
			const unsigned long long prod = amb_product(win, K);
			for (unsigned long long idx = 0; idx < prod; ++idx) {
				unsigned long long rr = idx; uint32_t word = 0;
				for (int q = 0; q < K; ++q) { const uint32_t code = (uint32_t)(win >> (4 * q)) & 15u, n = amb_count(code), d = (uint32_t)(rr % n); rr /= n; word |= ((amb_bases(code) >> (2u * d)) & 3u) << (2 * q); }
				emit(word);
			}
		}
	}
}

int main(int argc, char **argv) {
	const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
	std::mt19937 rng(argc > 2 ? (unsigned)atoi(argv[2]) : 5u);
	long long words = 0; int bad = 0;
	for (int it = 0; it < rounds; ++it) {
		const uint32_t L = 1 + rng() % 700; const int K = 4 + (int)(rng() % 12), z = (int)(rng() & 1);
		const double amb = it % 4 == 0 ? 0.0 : it % 4 == 1 ? 0.004 : it % 4 == 2 ? 0.03 : 0.2;
		const uint32_t nchunks = (L + 31) / 32;
		std::vector<uint32_t> dw((size_t)nchunks * 4, 0);
		for (uint32_t p = 0; p < L; ++p) {
			uint32_t sym = 1 + rng() % 4;
			if ((rng() % 100000) / 100000.0 < amb) { sym = rng() % 16; if (amb > 0.1 && rng() % 3 == 0) sym = 5 + rng() % 11; }      // (0 = padding inside a lane: a shorter sequence of the clump)
			dw[p >> 3] |= sym << (4 * (p & 7));
		}
		if (rng() & 1) for (uint32_t p = L; p < nchunks * 32; ++p) dw[p >> 3] |= (1 + rng() % 4) << (4 * (p & 7));      // junk behind the end must not matter
		std::vector<uint32_t> a, b;
		serial_words(dw.data(), L, K, z, [&](uint32_t w) { a.push_back(w); });
		acx_lane_words((const uint4 *)dw.data(), L, nchunks, K, z, [&](uint32_t w) { b.push_back(w); });
		std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
		words += (long long)a.size();
		if (a != b) { if (bad < 5) printf("MISMATCH round %d: L=%u K=%d z=%d amb=%g: %zu words vs %zu\n", it, L, K, z, amb, a.size(), b.size()); ++bad; }
	}
	printf("%d rounds, %lld words, %d mismatches\n", rounds, words, bad);
	return bad ? 1 : 0;
}
