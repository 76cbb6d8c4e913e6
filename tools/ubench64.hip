// tools/ubench64.hip — fp64 VALU instruction rates on gfx950, for k_bank's instruction floor: cycles a SIMD
// needs per wave-instruction with W waves resident (W = 1, 2, 4), measured with s_memtime around 8192 instructions per wave.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench64.hip -o tools/ubench64 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(X) X X X X X X X X
#define REP4(X) X X X X

template <int MODE>
__global__ __launch_bounds__ (256) void k_rate (unsigned long long* cyc, double* sink, int iters)
{
	double a0 = 1.0 + threadIdx.x * 1e-9, a1 = 1.1, a2 = 1.2, a3 = 1.3, a4 = 1.4, a5 = 1.5, a6 = 1.6, a7 = 1.7;
	const double x = 0.999999 + threadIdx.x * 1e-12, y = 1e-7;
	float f0 = 1.f, f1 = 2.f, f2 = 3.f, f3 = 4.f, f4 = 5.f, f5 = 6.f, f6 = 7.f, f7 = 8.f;
	unsigned long long t0, t1;
	asm volatile ("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
	for (int it = 0; it < iters; ++it) {
		if (MODE == 0) {          // v_fma_f64, 8 independent chains
			REP4 (asm volatile ("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
			                    "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
			                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
		} else if (MODE == 1) {   // v_add_f64
			REP4 (asm volatile ("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
			                    "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
			                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));)
		} else if (MODE == 2) {   // v_mul_f64
			REP4 (asm volatile ("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
			                    "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
			                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
		} else if (MODE == 3) {   // v_cvt_f32_f64
			REP4 (asm volatile ("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n"
			                    "v_cvt_f32_f64 %4, %12\n v_cvt_f32_f64 %5, %13\n v_cvt_f32_f64 %6, %14\n v_cvt_f32_f64 %7, %15\n"
			                    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
			                    : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
		} else if (MODE == 4) {   // v_fma_f32 (reference: a full-rate instruction)
			REP4 (asm volatile ("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
			                    "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
			                    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(f0));)
		} else if (MODE == 5) {   // dependent v_fma_f64 chain (latency)
			REP4 (asm volatile ("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
			                    "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
			                    : "+v"(a0) : "v"(x), "v"(y));)
		} else if (MODE == 6) {   // k_bank's mix per frame: 19 fma + 5 add + 1 mul (f64), cvt, 2 fma f32, max: dependent as in the bank (one section after the other)
			REP4 (asm volatile ("v_mul_f64 %1, %0, %8\n v_add_f64 %2, %1, %3\n v_fma_f64 %3, %2, %9, %4\n v_fma_f64 %4, %2, %9, %1\n"
			                    "v_add_f64 %5, %2, %6\n v_fma_f64 %6, %5, %9, %7\n v_fma_f64 %7, %5, %9, %2\n v_fma_f64 %0, %5, %9, %0\n"
			                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
		}
	}
	asm volatile ("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
	if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
	sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

template <int MODE>
static void run (const char* name, int waves_per_simd)
{
	const int iters = 16384, per_iter = 32;          // 524 288 instructions per wave: milliseconds, the launch does not count
	const int blocks = 256 * waves_per_simd;        // 256-thread blocks: one wave per SIMD of a CU each
	unsigned long long* cyc; double* sink;
	hipMalloc (&cyc, blocks * 4 * sizeof (unsigned long long));
	hipMalloc (&sink, (size_t) blocks * 256 * sizeof (double));
	hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
	float best = 1e9f;
	std::vector<unsigned long long> h (blocks * 4);
	for (int rep = 0; rep < 3; ++rep) {
		hipEventRecord (e0);
		hipLaunchKernelGGL ((k_rate<MODE>), dim3 (blocks), dim3 (256), 0, 0, cyc, sink, iters);
		hipEventRecord (e1); hipEventSynchronize (e1);
		float ms; hipEventElapsedTime (&ms, e0, e1); best = std::min (best, ms);
	}
	hipMemcpy (h.data (), cyc, h.size () * 8, hipMemcpyDeviceToHost);
	std::sort (h.begin (), h.end ());
	const double med = (double) h[h.size () / 2];
	// s_memtime counts at a constant 100 MHz on gfx950 (REFCLK): convert with the wall time of the launch instead
	const double insts = (double) iters * per_iter * waves_per_simd;          // wave-instructions per SIMD
	printf ("%-44s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD  (= %.2f cycles at 2.4 GHz, %.2f at 2.0 GHz; memtime ticks median %.0f)\n",
	        name, waves_per_simd, best, best * 1e6 / insts, best * 1e6 / insts * 2.4, best * 1e6 / insts * 2.0, med);
	hipFree (cyc); hipFree (sink);
}

int main ()
{
	for (int w : {1, 2, 3, 4, 8}) {
		run<4> ("v_fma_f32 (8 chains)", w);
		run<0> ("v_fma_f64 (8 chains)", w);
		run<1> ("v_add_f64 (8 chains)", w);
		run<2> ("v_mul_f64 (8 chains)", w);
		run<3> ("v_cvt_f32_f64", w);
		run<5> ("v_fma_f64 dependent chain", w);
		run<6> ("mul/add/fma f64 mix, dependent like the bank", w);
	}
	return 0;
}
