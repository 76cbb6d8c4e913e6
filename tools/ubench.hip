// tools/ubench.hip — instruction-rate microbenchmarks that decide the fused kernel's design
// (packed vs scalar fp32 FMA, SGPR-broadcast operands, LDS read / bpermute / DPP costs) on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float v2f __attribute__ ((ext_vector_type (2)));

#define REP8(X) X X X X X X X X

template <int MODE>
__global__ __launch_bounds__ (256) void k_rate (float* out, int iters, float s0)
{
	v2f a0 = {1, 2}, a1 = {3, 4}, a2 = {5, 6}, a3 = {7, 8}, a4 = {1, 3}, a5 = {2, 4}, a6 = {5, 7}, a7 = {6, 8};
	v2f x = {threadIdx.x * 1e-3f, 0.5f};
	float f0 = 1, f1 = 2, f2 = 3, f3 = 4, f4 = 5, f5 = 6, f6 = 7, f7 = 8;
	float fx = threadIdx.x * 1e-3f;
	__shared__ v2f lds[4096];
	for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = v2f{(float) i, 1.f};
	__syncthreads ();
	const int lane = threadIdx.x & 63;
	const v2f* lp = lds + (threadIdx.x >> 6) * 1000 + lane * 13;
	int idx = (lane * 4);
	for (int it = 0; it < iters; ++it) {
		if (MODE == 0) {   // v_pk_fma_f32, VGPR operands, 8 independent chains
			REP8 (asm volatile ("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
			                    "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n"
			                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
		} else if (MODE == 1) {   // v_fma_f32
			REP8 (asm volatile ("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
			                    "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
			                    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fx));)
		} else if (MODE == 2) {   // v_pk_fma_f32 with an SGPR-pair broadcast operand (the FIR form)
			REP8 (asm volatile ("v_pk_fma_f32 %0, %8, s[20:21], %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %8, s[22:23], %1 op_sel_hi:[1,0,1]\n"
			                    "v_pk_fma_f32 %2, %8, s[24:25], %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %8, s[26:27], %3 op_sel_hi:[1,0,1]\n"
			                    "v_pk_fma_f32 %4, %8, s[28:29], %4 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %8, s[30:31], %5 op_sel_hi:[1,0,1]\n"
			                    "v_pk_fma_f32 %6, %8, s[32:33], %6 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %7, %8, s[34:35], %7 op_sel_hi:[1,0,1]\n"
			                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x)
			                    : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");)
		} else if (MODE == 3) {   // v_fma_f32 with SGPR operand
			REP8 (asm volatile ("v_fma_f32 %0, %8, s20, %0\n v_fma_f32 %1, %8, s21, %1\n v_fma_f32 %2, %8, s22, %2\n v_fma_f32 %3, %8, s23, %3\n"
			                    "v_fma_f32 %4, %8, s24, %4\n v_fma_f32 %5, %8, s25, %5\n v_fma_f32 %6, %8, s26, %6\n v_fma_f32 %7, %8, s27, %7\n"
			                    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fx)
			                    : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
		} else if (MODE == 4) {   // v_pk_add_f32
			REP8 (asm volatile ("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
			                    "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
			                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
		} else if (MODE == 5) {   // v_max3_f32 with abs modifiers
			REP8 (asm volatile ("v_max3_f32 %0, %0, |%8|, |%1|\n v_max3_f32 %1, %1, |%8|, |%2|\n v_max3_f32 %2, %2, |%8|, |%3|\n v_max3_f32 %3, %3, |%8|, |%4|\n"
			                    "v_max3_f32 %4, %4, |%8|, |%5|\n v_max3_f32 %5, %5, |%8|, |%6|\n v_max3_f32 %6, %6, |%8|, |%7|\n v_max3_f32 %7, %7, |%8|, |%0|\n"
			                    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fx));)
		} else if (MODE == 6) {   // ds_read_b64 at lane stride 13 slots, 8 per group, one wait
			REP8 (asm volatile ("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:16\n ds_read_b64 %3, %8 offset:24\n"
			                    "ds_read_b64 %4, %8 offset:32\n ds_read_b64 %5, %8 offset:40\n ds_read_b64 %6, %8 offset:48\n ds_read_b64 %7, %8 offset:56\n s_waitcnt lgkmcnt(0)\n"
			                    : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"((unsigned) (size_t) lp));)
		} else if (MODE == 7) {   // ds_bpermute_b32 throughput (independent)
			REP8 (asm volatile ("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
			                    "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n"
			                    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(idx));)
		} else if (MODE == 8) {   // ds_bpermute_b32 dependent latency
			REP8 (asm volatile ("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n"
			                    "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n"
			                    "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n"
			                    "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n"
			                    : "+v"(f0) : "v"(idx));)
		} else if (MODE == 9) {   // DPP row_shr:1 mov (cross-lane without LDS)
			REP8 (asm volatile ("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
			                    "v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
			                    "v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
			                    "v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
			                    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));)
		} else if (MODE == 10) {  // dependent v_pk_fma_f32 chain (latency)
			REP8 (asm volatile ("v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n"
			                    "v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n"
			                    : "+v"(a0) : "v"(x));)
		} else if (MODE == 11) {  // dependent v_fma_f32 chain (latency)
			REP8 (asm volatile ("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n"
			                    "v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n"
			                    : "+v"(f0) : "v"(fx));)
		} else if (MODE == 12) {  // v_pk_mul_f32
			REP8 (asm volatile ("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
			                    "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
			                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
		}
	}
	v2f r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
	out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + s0;
}

template <int MODE>
static void run (const char* name, int blocks_per_cu, float* d_out)
{
	const int iters = 2000, grid = 256 * blocks_per_cu;
	hipEvent_t e0, e1;
	hipEventCreate (&e0); hipEventCreate (&e1);
	hipLaunchKernelGGL (k_rate<MODE>, dim3 (grid), dim3 (256), 0, 0, d_out, 10, 0.f);
	hipDeviceSynchronize ();
	hipEventRecord (e0);
	hipLaunchKernelGGL (k_rate<MODE>, dim3 (grid), dim3 (256), 0, 0, d_out, iters, 0.f);
	hipEventRecord (e1);
	hipEventSynchronize (e1);
	float ms;
	hipEventElapsedTime (&ms, e0, e1);
	const double winstr = (double) grid * 4 * iters * 64;            // wave-instructions
	const double per_simd = winstr / (256.0 * 4);                     // per SIMD
	const double cyc = ms * 1e-3 * 2.4e9 / per_simd;                  // cycles per wave-instr per SIMD at 2.4 GHz
	printf ("%-44s blocks/CU=%d  %8.3f ms  %6.2f cyc/wave-instr/SIMD @2.4GHz  %7.2f T lane-instr/s\n",
	        name, blocks_per_cu, ms, cyc, winstr * 64 / (ms * 1e-3) / 1e12);
}

int main ()
{
	float* d;
	hipMalloc (&d, 256 * 8 * 256 * 4);
	for (int b : {1, 2, 4}) {
		run<0> ("v_pk_fma_f32 vgpr", b, d);
		run<1> ("v_fma_f32 vgpr", b, d);
		run<2> ("v_pk_fma_f32 sgpr-broadcast", b, d);
		run<3> ("v_fma_f32 sgpr", b, d);
		run<4> ("v_pk_add_f32", b, d);
		run<12> ("v_pk_mul_f32", b, d);
		run<5> ("v_max3_f32 |abs|", b, d);
		run<6> ("ds_read_b64 stride13 (8 per wait)", b, d);
		run<7> ("ds_bpermute_b32 (8 per wait)", b, d);
		run<9> ("v_mov_b32_dpp row_shr:1", b, d);
	}
	run<8> ("ds_bpermute_b32 dependent (1 wave/SIMD)", 1, d);
	run<10> ("v_pk_fma_f32 dependent chain (1 wave/SIMD)", 1, d);
	run<11> ("v_fma_f32 dependent chain (1 wave/SIMD)", 1, d);
	return 0;
}
