// tools/mfma_probe.hip — checks mtr_mfma_fir.h on the GPU: operand / result layouts of
// v_mfma_f32_32x32x16_f16 as used there, the accuracy of the split-f16 interpolator against a double
// precision FIR with the real taps, and whether f16 subnormal operands survive the matrix pipe.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imeters.lv2_amd/csrc tools/mfma_probe.hip \
//        -Lmeters.lv2_amd/lib -lmtr_engine -Wl,-rpath,'$ORIGIN/../meters.lv2_amd/lib' -o tools/mfma_probe
#include <cmath>
#include <cstdio>
#include <vector>

#include "mtr_internal.h"
#include "mtr_mfma_fir.h"

__global__ void k_probe (const float* x, int n, const uint16_t* atab, float* out)
{
	__shared__ __attribute__ ((aligned (16))) uint32_t W[512];
	const int lane = threadIdx.x;
	for (int i = lane; i < 512; i += 64) W[i] = i < n ? mfir::split_word (x[i]) : 0u;
	__syncthreads ();
	mfir::AFrag A;
	A.load (atab, lane);
	const mfir::f16x acc = mfir::tile (A, W, 0, lane);
	for (int r = 0; r < 16; ++r) out[r * 64 + lane] = acc[r];
}

int main ()
{
	float tab[120], g[3][48];
	mtr_setup_fir_table (tab);
	for (int ph = 1; ph <= 3; ++ph)
		for (int i = 0; i < 48; ++i) g[ph - 1][i] = (i < 24) ? tab[24 * ph + i] : tab[24 * (4 - ph) + (47 - i)];
	std::vector<uint16_t> a (MTR_MFMA_A_HALVES);
	mtr_mfma_build_a (&g[0][0], a.data ());
	const int n = 256 + 56;
	for (int pass = 0; pass < 3; ++pass) {
		const float scale = pass == 0 ? 1.f : (pass == 1 ? 1e-3f : 3e-6f);
		std::vector<float> x (n);
		uint32_t r = 777u + pass;
		for (auto& v : x) { r = r * 1664525u + 1013904223u; v = scale * (((int) (r >> 8) - (1 << 23)) / 8388608.f); }
		float *dx, *dout; uint16_t* da;
		hipMalloc (&dx, n * 4); hipMalloc (&dout, 1024 * 4); hipMalloc (&da, a.size () * 2);
		hipMemcpy (dx, x.data (), n * 4, hipMemcpyHostToDevice);
		hipMemcpy (da, a.data (), a.size () * 2, hipMemcpyHostToDevice);
		hipLaunchKernelGGL (k_probe, dim3 (1), dim3 (64), 0, 0, dx, n, da, dout);
		std::vector<float> out (1024);
		hipMemcpy (out.data (), dout, 1024 * 4, hipMemcpyDeviceToHost);
		double worst = 0, worst_f16tap = 0, peak = 0;
		for (int lane = 0; lane < 64; ++lane)
			for (int rr = 0; rr < 16; ++rr) {
				const int p = rr >> 2, f = 8 * (lane & 31) + (rr & 3) + 4 * (lane >> 5);   // output frame f: window x[f .. f+47]
				double ref = 0;
				if (p == 0) ref = x[f + 23];
				else for (int i = 0; i < 48; ++i) ref += (double) g[p - 1][i] * (double) x[f + i];
				const double e = fabs ((double) out[rr * 64 + lane] / (double) (1 << MTR_MFMA_TAP_SHIFT) - ref);
				if (e > worst) worst = e;
				if (fabs (ref) > peak) peak = fabs (ref);
			}
		printf ("scale %.1e: peak |y| %.6g, worst |mfma - f64| %.3g = %.3g of the peak (%.5f dB)\n", scale, peak, worst, worst / peak,
		        20 * log10 (1 + worst / peak));
		(void) worst_f16tap;
	}
	return 0;
}
