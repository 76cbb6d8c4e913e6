// tools/coissue.hip — does f32 VALU work run BESIDE f16 MFMAs on one SIMD of gfx950, and in which form?
// Decides the structure of the matrix-pipe true-peak kernel (mtr_fused4.hip):
//   * MFMA alone (16x16x32 and 32x32x16 f16), VALU alone (v_fma_f32, v_pk_fma_f32),
//   * both in ONE wave, interleaved by the compiler's sched_group_barrier at several VALU : MFMA ratios,
//   * both on one SIMD from TWO waves (a 512-thread workgroup: waves w and w + 4 share a SIMD), one role each,
//   * with the operand traffic of the real kernel (ds_read_b128 per MFMA group, v_max3 epilogue).
// Output: shader cycles (s_memtime) per iteration for one wave, wall time, and what the sum of the parts would be.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/coissue.hip -o tools/coissue ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__ ((ext_vector_type (2)));
typedef float f4 __attribute__ ((ext_vector_type (4)));
typedef float f16v __attribute__ ((ext_vector_type (16)));
typedef _Float16 h8 __attribute__ ((ext_vector_type (8)));

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier (mask, n, 0)
constexpr int M_VALU = 0x2, M_MFMA = 0x8, M_DSR = 0x100;

struct Res { unsigned long long cyc; };

// NM MFMAs per iteration in six accumulator chains; NV scalar FMAs (PK = false) or NV / 2 packed FMAs (PK = true),
// interleaved R per MFMA.  ROLE: 0 = both in this wave, 1 = MFMA only, 2 = VALU only, 3 = by wave (w < 4: MFMA, else VALU)
template <int SHAPE, int NM, int NV, bool PK, int ROLE, bool LDS_OPS>
__global__ void k_co (float* out, unsigned long long* cyc, int iters, float seed)
{
	extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint4* const L = reinterpret_cast<uint4*> (smem);
	for (int i = threadIdx.x; i < 1024; i += blockDim.x) L[i] = uint4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
	__syncthreads ();
	h8 a[3], b[2];
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16) (0.001f * (lane + i + j));
	for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (_Float16) (0.002f * (lane - i + j));
	f4 c4[6];
	f16v c16[2];
	for (int i = 0; i < 6; ++i) c4[i] = f4{0, 0, 0, 0};
	for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) c16[i][j] = 0;
	float f[8];
	v2f p[8];
	for (int i = 0; i < 8; ++i) { f[i] = seed + i + lane; p[i] = v2f{seed + i, seed - lane}; }
	const float m = 0.999f + seed;
	const v2f mp = v2f{m, m};
	float pk = 0.f;
	const bool do_m = ROLE == 0 || ROLE == 1 || (ROLE == 3 && wave < 4);
	const bool do_v = ROLE == 0 || ROLE == 2 || (ROLE == 3 && wave >= 4);
	const uint4* const lp = L + lane;

	unsigned long long t0 = __builtin_readcyclecounter ();
	for (int it = 0; it < iters; ++it) {
		if constexpr (ROLE == 0) {
			// one stream: per MFMA, R VALU ops
			constexpr int R = NM > 0 ? NV / (NM > 0 ? NM : 1) : 0;
#pragma unroll
			for (int k = 0; k < NM; ++k) {
				if (LDS_OPS && (k % 6) == 0) b[(k / 6) & 1] = __builtin_bit_cast (h8, lp[64 * ((k / 6) & 7)]);
				if (SHAPE == 16) c4[k % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16 (a[k % 3], b[(k / 3) & 1], c4[k % 6], 0, 0, 0);
				else             c16[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a[k % 3], b[k & 1], c16[k & 1], 0, 0, 0);
#pragma unroll
				for (int r = 0; r < R; ++r) {
					const int q = (k * R + r) & 7;
					if (PK) { if ((r & 1) == 0) p[q] = p[q] * mp + mp; }
					else    f[q] = __builtin_fmaf (f[q], m, m);
				}
			}
#pragma unroll
			for (int k = 0; k < NM; ++k) {
				if (LDS_OPS && (k % 6) == 0) SGB (M_DSR, 1);
				SGB (M_MFMA, 1);
				SGB (M_VALU, (PK ? (R + 1) / 2 : R));
			}
		} else {
			if (do_m) {
#pragma unroll
				for (int k = 0; k < NM; ++k) {
					if (LDS_OPS && (k % 6) == 0) b[(k / 6) & 1] = __builtin_bit_cast (h8, lp[64 * ((k / 6) & 7)]);
					if (SHAPE == 16) c4[k % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16 (a[k % 3], b[(k / 3) & 1], c4[k % 6], 0, 0, 0);
					else             c16[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a[k % 3], b[k & 1], c16[k & 1], 0, 0, 0);
				}
			}
			if (do_v) {
#pragma unroll
				for (int k = 0; k < NV; ++k) {
					if (PK) { if ((k & 1) == 0) p[(k >> 1) & 7] = p[(k >> 1) & 7] * mp + mp; }
					else    f[k & 7] = __builtin_fmaf (f[k & 7], m, m);
				}
			}
		}
		if (LDS_OPS && do_m) {
			// the epilogue of the real kernel: |max| over the accumulators, then restart them
#pragma unroll
			for (int i = 0; i < 6; ++i) {
				pk = fmaxf (fmaxf (pk, fabsf (c4[i][0])), fabsf (c4[i][1]));
				pk = fmaxf (fmaxf (pk, fabsf (c4[i][2])), fabsf (c4[i][3]));
				c4[i] = f4{0, 0, 0, 0};
			}
		}
	}
	unsigned long long t1 = __builtin_readcyclecounter ();
	float r = pk;
	for (int i = 0; i < 8; ++i) r += f[i] + p[i].x + p[i].y;
	for (int i = 0; i < 6; ++i) r += c4[i][0] + c4[i][1] + c4[i][2] + c4[i][3];
	for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) r += c16[i][j];
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
	if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int SHAPE, int NM, int NV, bool PK, int ROLE, bool LDS_OPS>
static void run (const char* name, int threads, int wg_per_cu, float* d_out, unsigned long long* d_cyc)
{
	const int iters = 400;
	const int grid = 256 * wg_per_cu;
	const size_t lds = 160 * 1024 / wg_per_cu;               // pins the residency: exactly wg_per_cu workgroups per CU
	auto kern = k_co<SHAPE, NM, NV, PK, ROLE, LDS_OPS>;
	hipFuncSetAttribute ((const void*) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
	hipEvent_t e0, e1;
	hipEventCreate (&e0); hipEventCreate (&e1);
	hipLaunchKernelGGL (kern, dim3 (grid), dim3 (threads), lds, 0, d_out, d_cyc, 4, 0.f);
	hipDeviceSynchronize ();
	hipEventRecord (e0);
	hipLaunchKernelGGL (kern, dim3 (grid), dim3 (threads), lds, 0, d_out, d_cyc, iters, 0.f);
	hipEventRecord (e1);
	hipEventSynchronize (e1);
	float ms;
	hipEventElapsedTime (&ms, e0, e1);
	unsigned long long c[8] = {0};
	hipMemcpy (c, d_cyc, sizeof (c), hipMemcpyDeviceToHost);
	const int waves_per_simd = threads / 64 * wg_per_cu / 4;
	printf ("%-58s w/SIMD=%d  %8.3f ms  wall/iter/SIMD-slot %7.1f ns  memtime/iter w0 %8.1f w4 %8.1f\n", name, waves_per_simd, ms,
	        ms * 1e6 / iters, (double) c[0] / iters, (double) c[4] / iters);
	fflush (stdout);
}

int main ()
{
	float* d;
	unsigned long long* dc;
	hipMalloc (&d, 256 * 16 * 512 * 4);
	hipMalloc (&dc, 64);
	hipMemset (dc, 0, 64);
	// --- the parts alone, one wave per workgroup, 1 / 2 / 4 waves per SIMD
	for (int w : {4, 8, 16}) {
		run<16, 36, 0, false, 1, false> ("36 mfma16x16x32 only", 64, w, d, dc);
		run<32, 22, 0, false, 1, false> ("22 mfma32x32x16 only", 64, w, d, dc);
		run<16, 0, 144, false, 2, false> ("144 v_fma_f32 only", 64, w, d, dc);
		run<16, 0, 144, true, 2, false> ("72 v_pk_fma_f32 only", 64, w, d, dc);
	}
	// --- both in one wave
	for (int w : {4, 8, 16}) {
		run<16, 36, 72, false, 0, false> ("36 mfma16 + 72 v_fma (2 per mfma), one stream", 64, w, d, dc);
		run<16, 36, 144, false, 0, false> ("36 mfma16 + 144 v_fma (4 per mfma), one stream", 64, w, d, dc);
		run<16, 36, 216, false, 0, false> ("36 mfma16 + 216 v_fma (6 per mfma), one stream", 64, w, d, dc);
		run<16, 36, 288, false, 0, false> ("36 mfma16 + 288 v_fma (8 per mfma), one stream", 64, w, d, dc);
		run<16, 36, 72, true, 0, false> ("36 mfma16 + 36 v_pk_fma (1 per mfma), one stream", 64, w, d, dc);
		run<16, 36, 144, true, 0, false> ("36 mfma16 + 72 v_pk_fma (2 per mfma), one stream", 64, w, d, dc);
		run<16, 36, 288, true, 0, false> ("36 mfma16 + 144 v_pk_fma (4 per mfma), one stream", 64, w, d, dc);
		run<32, 22, 176, false, 0, false> ("22 mfma32 + 176 v_fma (8 per mfma), one stream", 64, w, d, dc);
		run<32, 22, 176, true, 0, false> ("22 mfma32 + 88 v_pk_fma (4 per mfma), one stream", 64, w, d, dc);
		run<16, 36, 144, false, 0, true> ("36 mfma16 + 144 v_fma + 6 ds_read_b128 + 12 max3", 64, w, d, dc);
	}
	// --- one role per wave, two (four) waves per SIMD: 512-thread workgroups, waves w and w + 4 on one SIMD
	for (int w : {1, 2}) {
		run<16, 36, 144, false, 3, false> ("wave<4: 36 mfma16 | wave>=4: 144 v_fma", 512, w, d, dc);
		run<16, 36, 288, false, 3, false> ("wave<4: 36 mfma16 | wave>=4: 288 v_fma", 512, w, d, dc);
		run<16, 36, 144, true, 3, false> ("wave<4: 36 mfma16 | wave>=4: 72 v_pk_fma", 512, w, d, dc);
		run<16, 36, 288, true, 3, false> ("wave<4: 36 mfma16 | wave>=4: 144 v_pk_fma", 512, w, d, dc);
		run<32, 22, 288, false, 3, false> ("wave<4: 22 mfma32 | wave>=4: 288 v_fma", 512, w, d, dc);
		run<32, 22, 288, true, 3, false> ("wave<4: 22 mfma32 | wave>=4: 144 v_pk_fma", 512, w, d, dc);
	}
	return 0;
}
