// tools/ubench/valu_rate.hip -- issue rate of the integer VALU ops the bit-vector kernels use (gfx950).
// 8 independent accumulators per lane, fully unrolled; reports cycles per wave64 instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 64
#define ITER 2000
#define OP8(STR) asm volatile(STR "\n" STR2(STR) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
template <int OP> __global__ void k(uint32_t *out, uint32_t b, uint32_t c) {
	uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	for (int it = 0; it < ITER; ++it) {
		#pragma unroll
		for (int r = 0; r < REP / 8; ++r) {
			if (OP == 0) asm volatile("v_and_b32 %0, %0, %8\nv_and_b32 %1, %1, %8\nv_and_b32 %2, %2, %8\nv_and_b32 %3, %3, %8\nv_and_b32 %4, %4, %8\nv_and_b32 %5, %5, %8\nv_and_b32 %6, %6, %8\nv_and_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
			if (OP == 1) asm volatile("v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\nv_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\nv_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\nv_bitop3_b32 %3, %3, %8, %9 bitop3:0x96\nv_bitop3_b32 %4, %4, %8, %9 bitop3:0x96\nv_bitop3_b32 %5, %5, %8, %9 bitop3:0x96\nv_bitop3_b32 %6, %6, %8, %9 bitop3:0x96\nv_bitop3_b32 %7, %7, %8, %9 bitop3:0x96" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
			if (OP == 2) asm volatile("v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\nv_add_u32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
			if (OP == 3) asm volatile("v_add3_u32 %0, %0, %8, %9\nv_add3_u32 %1, %1, %8, %9\nv_add3_u32 %2, %2, %8, %9\nv_add3_u32 %3, %3, %8, %9\nv_add3_u32 %4, %4, %8, %9\nv_add3_u32 %5, %5, %8, %9\nv_add3_u32 %6, %6, %8, %9\nv_add3_u32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
			if (OP == 4) asm volatile("v_lshlrev_b32 %0, 1, %0\nv_lshlrev_b32 %1, 1, %1\nv_lshlrev_b32 %2, 1, %2\nv_lshlrev_b32 %3, 1, %3\nv_lshlrev_b32 %4, 1, %4\nv_lshlrev_b32 %5, 1, %5\nv_lshlrev_b32 %6, 1, %6\nv_lshlrev_b32 %7, 1, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
			if (OP == 5) asm volatile("v_alignbit_b32 %0, %0, %8, 31\nv_alignbit_b32 %1, %1, %8, 31\nv_alignbit_b32 %2, %2, %8, 31\nv_alignbit_b32 %3, %3, %8, 31\nv_alignbit_b32 %4, %4, %8, 31\nv_alignbit_b32 %5, %5, %8, 31\nv_alignbit_b32 %6, %6, %8, 31\nv_alignbit_b32 %7, %7, %8, 31" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
			if (OP == 6) asm volatile("v_bfi_b32 %0, %0, %8, %9\nv_bfi_b32 %1, %1, %8, %9\nv_bfi_b32 %2, %2, %8, %9\nv_bfi_b32 %3, %3, %8, %9\nv_bfi_b32 %4, %4, %8, %9\nv_bfi_b32 %5, %5, %8, %9\nv_bfi_b32 %6, %6, %8, %9\nv_bfi_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
			if (OP == 7) asm volatile("v_min3_i32 %0, %0, %8, %9\nv_min3_i32 %1, %1, %8, %9\nv_min3_i32 %2, %2, %8, %9\nv_min3_i32 %3, %3, %8, %9\nv_min3_i32 %4, %4, %8, %9\nv_min3_i32 %5, %5, %8, %9\nv_min3_i32 %6, %6, %8, %9\nv_min3_i32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
			if (OP == 8) asm volatile("v_and_or_b32 %0, %0, %8, %9\nv_and_or_b32 %1, %1, %8, %9\nv_and_or_b32 %2, %2, %8, %9\nv_and_or_b32 %3, %3, %8, %9\nv_and_or_b32 %4, %4, %8, %9\nv_and_or_b32 %5, %5, %8, %9\nv_and_or_b32 %6, %6, %8, %9\nv_and_or_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
			if (OP == 9) asm volatile("v_addc_co_u32 %0, vcc, %0, %8, vcc\nv_addc_co_u32 %1, vcc, %1, %8, vcc\nv_addc_co_u32 %2, vcc, %2, %8, vcc\nv_addc_co_u32 %3, vcc, %3, %8, vcc\nv_addc_co_u32 %4, vcc, %4, %8, vcc\nv_addc_co_u32 %5, vcc, %5, %8, vcc\nv_addc_co_u32 %6, vcc, %6, %8, vcc\nv_addc_co_u32 %7, vcc, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
			if (OP == 10) asm volatile("v_add_co_u32_e64 %0, s[20:21], %0, %8\nv_add_co_u32_e64 %1, s[20:21], %1, %8\nv_add_co_u32_e64 %2, s[20:21], %2, %8\nv_add_co_u32_e64 %3, s[20:21], %3, %8\nv_add_co_u32_e64 %4, s[20:21], %4, %8\nv_add_co_u32_e64 %5, s[20:21], %5, %8\nv_add_co_u32_e64 %6, s[20:21], %6, %8\nv_add_co_u32_e64 %7, s[20:21], %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21");
			if (OP == 11) asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\nv_cndmask_b32_e64 %1, %1, %8, s[20:21]\nv_cndmask_b32_e64 %2, %2, %8, s[20:21]\nv_cndmask_b32_e64 %3, %3, %8, s[20:21]\nv_cndmask_b32_e64 %4, %4, %8, s[20:21]\nv_cndmask_b32_e64 %5, %5, %8, s[20:21]\nv_cndmask_b32_e64 %6, %6, %8, s[20:21]\nv_cndmask_b32_e64 %7, %7, %8, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21");
			if (OP == 12) asm volatile("v_lshl_or_b32 %0, %0, 1, %8\nv_lshl_or_b32 %1, %1, 1, %8\nv_lshl_or_b32 %2, %2, 1, %8\nv_lshl_or_b32 %3, %3, 1, %8\nv_lshl_or_b32 %4, %4, 1, %8\nv_lshl_or_b32 %5, %5, 1, %8\nv_lshl_or_b32 %6, %6, 1, %8\nv_lshl_or_b32 %7, %7, 1, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21");
			if (OP == 13) asm volatile("v_lshrrev_b32_e32 %0, 1, %0\nv_lshrrev_b32_e32 %1, 1, %1\nv_lshrrev_b32_e32 %2, 1, %2\nv_lshrrev_b32_e32 %3, 1, %3\nv_lshrrev_b32_e32 %4, 1, %4\nv_lshrrev_b32_e32 %5, 1, %5\nv_lshrrev_b32_e32 %6, 1, %6\nv_lshrrev_b32_e32 %7, 1, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21");
			if (OP == 14) asm volatile("v_mul_lo_u32 %0, %0, %8\nv_mul_lo_u32 %1, %1, %8\nv_mul_lo_u32 %2, %2, %8\nv_mul_lo_u32 %3, %3, %8\nv_mul_lo_u32 %4, %4, %8\nv_mul_lo_u32 %5, %5, %8\nv_mul_lo_u32 %6, %6, %8\nv_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21");
			if (OP == 15) asm volatile("v_or_b32_e32 %0, %0, %8\nv_or_b32_e32 %1, %1, %8\nv_or_b32_e32 %2, %2, %8\nv_or_b32_e32 %3, %3, %8\nv_or_b32_e32 %4, %4, %8\nv_or_b32_e32 %5, %5, %8\nv_or_b32_e32 %6, %6, %8\nv_or_b32_e32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21");
			if (OP == 16) asm volatile("v_addc_co_u32_e64 %0, s[20:21], %0, %8, s[20:21]\nv_addc_co_u32_e64 %1, s[20:21], %1, %8, s[20:21]\nv_addc_co_u32_e64 %2, s[20:21], %2, %8, s[20:21]\nv_addc_co_u32_e64 %3, s[20:21], %3, %8, s[20:21]\nv_addc_co_u32_e64 %4, s[20:21], %4, %8, s[20:21]\nv_addc_co_u32_e64 %5, s[20:21], %5, %8, s[20:21]\nv_addc_co_u32_e64 %6, s[20:21], %6, %8, s[20:21]\nv_addc_co_u32_e64 %7, s[20:21], %7, %8, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21");
			if (OP == 17) asm volatile("v_and_or_b32 %0, %0, %8, %9\nv_and_or_b32 %1, %1, %8, %9\nv_and_or_b32 %2, %2, %8, %9\nv_and_or_b32 %3, %3, %8, %9\nv_and_or_b32 %4, %4, %8, %9\nv_and_or_b32 %5, %5, %8, %9\nv_and_or_b32 %6, %6, %8, %9\nv_and_or_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21");
			if (OP == 18) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21");
			if (OP == 19) asm volatile("v_xad_u32 %0, %0, %8, %9\nv_xad_u32 %1, %1, %8, %9\nv_xad_u32 %2, %2, %8, %9\nv_xad_u32 %3, %3, %8, %9\nv_xad_u32 %4, %4, %8, %9\nv_xad_u32 %5, %5, %8, %9\nv_xad_u32 %6, %6, %8, %9\nv_xad_u32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21");
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP> void run(const char *name, uint32_t *d) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int blocks = 256 * 8, threads = 256;   // 8 waves per SIMD
	k<OP><<<blocks, threads>>>(d, 0x12345, 7); hipDeviceSynchronize();
	hipEventRecord(e0); k<OP><<<blocks, threads>>>(d, 0x12345, 7); hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	double winstr = (double)blocks * (threads / 64) * (double)ITER * REP;
	printf("%-16s %8.3f ms  %7.1f G wave-instr/s  -> %.2f cycles/instr/SIMD at 2.4 GHz (%.2f at 2.1)\n", name, ms, winstr / ms / 1e6,
	       1024.0 * 2.4e9 / (winstr / (ms * 1e-3)), 1024.0 * 2.1e9 / (winstr / (ms * 1e-3)));
}
int main() {
	uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
	run<0>("v_and_b32", d); run<1>("v_bitop3_b32", d); run<2>("v_add_u32", d); run<3>("v_add3_u32", d); run<4>("v_lshlrev_b32", d);
	run<5>("v_alignbit_b32", d); run<6>("v_bfi_b32", d); run<7>("v_min3_i32", d); run<8>("v_and_or_b32", d); run<9>("v_addc_co_u32", d);
	run<10>("v_add_co_u32_e64 sgpr", d); run<11>("v_cndmask_e64 sgpr", d); run<12>("v_lshl_or_b32", d); run<13>("v_lshrrev_b32", d); run<14>("v_mul_lo_u32", d);
	run<15>("v_or_b32", d); run<16>("v_addc_co_e64 sgpr", d); run<17>("v_and_or_b32", d); run<18>("v_mov_b32_dpp", d); run<19>("v_xad_u32", d);
	return 0;
}
