// tools/mfma_power.hip — is v_mfma_f32_32x32x16_f16 cheaper per MAC than v_mfma_f32_16x16x32_f16 under the socket's power cap?
// Same MACs, noise operands (toggling), one wave per SIMD (k_seg's occupancy); duration by HIP events.  A 32x32x16 does 16384
// MACs with 16 + 16 operand bytes per lane, a 16x16x32 8192 MACs with the same operand bytes.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_power.hip -o tools/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__ ((ext_vector_type (8)));
typedef float f4 __attribute__ ((ext_vector_type (4)));
typedef float f16v __attribute__ ((ext_vector_type (16)));

template <int MODE>
__global__ __launch_bounds__ (64) void k (const uint4* src, float* out, int iters)
{
	const int lane = threadIdx.x;
	uint4 a[6], b[4];
#pragma unroll
	for (int i = 0; i < 6; ++i) a[i] = src[(blockIdx.x * 16 + i) * 64 + lane];
#pragma unroll
	for (int i = 0; i < 4; ++i) b[i] = src[(blockIdx.x * 16 + 6 + i) * 64 + lane];
	if (MODE == 0) {
		f4 acc[6];
#pragma unroll
		for (int i = 0; i < 6; ++i) acc[i] = f4{0, 0, 0, 0};
		for (int it = 0; it < iters; ++it) {
#pragma unroll
			for (int r = 0; r < 4; ++r)
#pragma unroll
				for (int i = 0; i < 6; ++i)        // 24 MFMAs of 8192 MACs per trip
					acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16 (__builtin_bit_cast (h8, a[i]), __builtin_bit_cast (h8, b[(i + r) & 3]), acc[i], 0, 0, 0);
		}
		float s = 0;
#pragma unroll
		for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
		out[blockIdx.x * 64 + lane] = s;
	} else {
		f16v acc[3];
#pragma unroll
		for (int i = 0; i < 3; ++i) acc[i] = f16v{0};
		for (int it = 0; it < iters; ++it) {
#pragma unroll
			for (int r = 0; r < 4; ++r)
#pragma unroll
				for (int i = 0; i < 3; ++i)        // 12 MFMAs of 16384 MACs per trip
					acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16 (__builtin_bit_cast (h8, a[i + (r & 1) * 3]), __builtin_bit_cast (h8, b[(i + r) & 3]), acc[i], 0, 0, 0);
		}
		float s = 0;
#pragma unroll
		for (int i = 0; i < 3; ++i)
#pragma unroll
			for (int j = 0; j < 16; ++j) s += acc[i][j];
		out[blockIdx.x * 64 + lane] = s;
	}
}

int main (int argc, char** argv)
{
	const int waves = argc > 1 ? atoi (argv[1]) : 1024, iters = argc > 2 ? atoi (argv[2]) : 200000;
	uint4* src; float* out;
	hipMalloc (&src, (size_t) waves * 16 * 64 * 16); hipMalloc (&out, (size_t) waves * 64 * 4);
	uint32_t* h = (uint32_t*) malloc ((size_t) waves * 16 * 64 * 16);
	uint32_t r = 1234567;
	for (size_t i = 0; i < (size_t) waves * 16 * 64 * 4; ++i) {
		// pairs of f16 noise in [-2, 2): sign, exponent 13..16, random mantissa
		uint32_t w = 0;
		for (int k = 0; k < 2; ++k) { r = r * 1664525u + 1013904223u; const uint32_t e = 13 + ((r >> 20) & 3); w |= (((r >> 31) << 15) | (e << 10) | ((r >> 8) & 0x3ff)) << (16 * k); }
		h[i] = w;
	}
	hipMemcpy (src, h, (size_t) waves * 16 * 64 * 16, hipMemcpyHostToDevice);
	hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
	for (int rep = 0; rep < 3; ++rep)
		for (int mode = 0; mode < 2; ++mode) {
			hipEventRecord (e0);
			if (mode == 0) hipLaunchKernelGGL (k<0>, dim3 (waves), dim3 (64), 0, 0, src, out, iters);
			else           hipLaunchKernelGGL (k<1>, dim3 (waves), dim3 (64), 0, 0, src, out, iters);
			hipEventRecord (e1); hipDeviceSynchronize ();
			float ms; hipEventElapsedTime (&ms, e0, e1);
			const double macs = (double) waves * iters * 24 * 8192;
			printf ("%s: %d waves x %d trips: %.2f ms, %.1f T MAC/s (dense peak 1250 T MAC/s at 2.4 GHz)\n", mode ? "32x32x16" : "16x16x32", waves, iters, ms, macs / ms / 1e9);
		}
	return 0;
}
