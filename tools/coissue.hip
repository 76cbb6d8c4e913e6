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


// ---- the real mix: per block 36 MFMAs (two channels x 18, operands from LDS) with NPK packed K-filter steps' worth of
// dependent v_pk_fma_f32 (11 per step, as KW_STEP) and NSC scalar v_max3 spread between them
template <int NSTEP, int NMAX, int PAT>
__global__ void k_mix (float* out, unsigned long long* cyc, int iters, float seed)
{
	extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
	const int lane = threadIdx.x & 63;
	uint4* const L = reinterpret_cast<uint4*> (smem);
	for (int i = threadIdx.x; i < 1024; i += blockDim.x) L[i] = uint4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
	__syncthreads ();
	h8 a[6];
	for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16) (0.001f * (lane + i + j));
	v2f z1 = {seed, seed}, z2 = z1, z3 = z1, z4 = z1, sj = {0, 0};
	const v2f a0 = {1.f + seed, 1.f}, a1 = {-1.9f, -1.9f}, a2 = {0.9f, 0.9f}, b1 = {-1.6f, -1.6f}, b2 = {0.7f, 0.7f}, c3 = {0.01f, 0.01f}, c4 = {2e-5f, 2e-5f};
	v2f xs[4] = {{seed + lane, 1.f}, {2.f, seed}, {3.f, 1.f}, {seed, 4.f}};
	float pk = 0.f;
	const uint4* const lp = L + lane;
	unsigned long long t0 = __builtin_readcyclecounter ();
	for (int it = 0; it < iters; ++it) {
		uint4 b0 = lp[0], b1_ = lp[64], b2_ = lp[128], b3 = lp[192];
		f4 y[6];
#pragma unroll
		for (int i = 0; i < 6; ++i) y[i] = f4{0, 0, 0, 0};
#pragma unroll
		for (int k = 0; k < 36; ++k) {
			const uint4& bb = (k / 9) == 0 ? b0 : (k / 9) == 1 ? b1_ : (k / 9) == 2 ? b2_ : b3;
			y[k % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16 (a[k % 6], __builtin_bit_cast (h8, bb), y[k % 6], 0, 0, 0);
		}
#pragma unroll
		for (int sidx = 0; sidx < NSTEP; ++sidx) {
			v2f t_ = xs[sidx & 3] + 1e-15f;
			t_ = t_ - b2 * z2;
			const v2f x_ = t_ - b1 * z1;
			v2f u_ = a1 * z1;
			u_ = u_ + a2 * z2;
			u_ = u_ - c4 * z4;
			u_ = u_ - c3 * z3;
			const v2f yy = a0 * x_ + u_;
			z2 = z1; z1 = x_; z4 += z3; z3 += yy;
			sj += yy * yy;
		}
#pragma unroll
		for (int i = 0; i < NMAX; ++i) pk = fmaxf (fmaxf (pk, fabsf (y[i % 6][(2 * i / 6) & 3])), fabsf (y[i % 6][((2 * i / 6) + 1) & 3]));
		if (PAT == 1) {
			// one VALU after every MFMA
#pragma unroll
			for (int k = 0; k < 36; ++k) { SGB (M_MFMA, 1); SGB (M_VALU, 1); }
		} else if (PAT == 2) {
			// all MFMAs, then all VALU
			SGB (M_MFMA, 36);
		}
	}
	unsigned long long t1 = __builtin_readcyclecounter ();
	float r = pk + z1.x + z1.y + z2.x + z3.x + z3.y + z4.x + z4.y + sj.x + sj.y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
	if (blockIdx.x == 0 && lane == 0) cyc[0] = t1 - t0;
}

template <int NSTEP, int NMAX, int PAT>
static void run_mix (const char* name, int wg_per_cu, float* d_out, unsigned long long* d_cyc)
{
	const int iters = 400, grid = 256 * wg_per_cu;
	const size_t lds = 160 * 1024 / wg_per_cu;
	auto kern = k_mix<NSTEP, NMAX, PAT>;
	hipFuncSetAttribute ((const void*) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
	hipEvent_t e0, e1;
	hipEventCreate (&e0); hipEventCreate (&e1);
	hipLaunchKernelGGL (kern, dim3 (grid), dim3 (64), lds, 0, d_out, d_cyc, 4, 0.f);
	hipDeviceSynchronize ();
	hipEventRecord (e0);
	hipLaunchKernelGGL (kern, dim3 (grid), dim3 (64), lds, 0, d_out, d_cyc, iters, 0.f);
	hipEventRecord (e1);
	hipEventSynchronize (e1);
	float ms;
	hipEventElapsedTime (&ms, e0, e1);
	unsigned long long c = 0;
	hipMemcpy (&c, d_cyc, sizeof (c), hipMemcpyDeviceToHost);
	printf ("%-64s w/SIMD=%d  wall/iter/SIMD %7.1f ns  memtime/iter(w0) %8.1f\n", name, wg_per_cu / 4, ms * 1e6 / iters / (wg_per_cu / 4), (double) c / iters);
	fflush (stdout);
}

// ---- the exact alternative: the interpolator on v_mfma_f32_32x32x2_f32 (f32 in, bit-for-bit an fmaf
// chain) beside a wave of packed-f32 K-filter work on the same SIMD.  One 32x32x2 MFMA = 2048 MACs in 64 cycles.
template <int ROLE>      // 1 = MFMA only, 2 = VALU only, 3 = waves 0-3 MFMA, waves 4-7 VALU
__global__ void k_f32mfma (float* out, unsigned long long* cyc, int iters, float seed)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	f16v c[4];
	for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0;
	float a = 0.001f * lane + seed, b = 0.002f * lane - seed;
	v2f p[8];
	for (int i = 0; i < 8; ++i) p[i] = v2f{seed + i, seed - lane};
	const v2f mp = v2f{0.999f + seed, 0.999f + seed};
	const bool do_m = ROLE == 1 || (ROLE == 3 && wave < 4), do_v = ROLE == 2 || (ROLE == 3 && wave >= 4);
	unsigned long long t0 = __builtin_readcyclecounter ();
	for (int it = 0; it < iters; ++it) {
		if (do_m) {
#pragma unroll
			for (int k = 0; k < 16; ++k) c[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32 (a, b, c[k & 3], 0, 0, 0);
		}
		if (do_v) {
#pragma unroll
			for (int k = 0; k < 128; ++k) p[k & 7] = p[k & 7] * mp + mp;
		}
	}
	unsigned long long t1 = __builtin_readcyclecounter ();
	float r = 0;
	for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
	for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r += c[i][j];
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
	if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int ROLE>
static void run_f32 (const char* name, int threads, float* d_out, unsigned long long* d_cyc)
{
	const int iters = 400, grid = 256;
	hipEvent_t e0, e1;
	hipEventCreate (&e0); hipEventCreate (&e1);
	hipLaunchKernelGGL (k_f32mfma<ROLE>, dim3 (grid), dim3 (threads), 0, 0, d_out, d_cyc, 4, 0.f);
	hipDeviceSynchronize ();
	hipEventRecord (e0);
	hipLaunchKernelGGL (k_f32mfma<ROLE>, dim3 (grid), dim3 (threads), 0, 0, d_out, d_cyc, iters, 0.f);
	hipEventRecord (e1);
	hipEventSynchronize (e1);
	float ms;
	hipEventElapsedTime (&ms, e0, e1);
	unsigned long long c[8] = {0};
	hipMemcpy (c, d_cyc, sizeof (c), hipMemcpyDeviceToHost);
	printf ("%-72s %8.3f ms  wall/iter %7.1f ns  memtime/iter w0 %8.1f w4 %8.1f\n", name, ms, ms * 1e6 / iters, (double) c[0] / iters, (double) c[4] / iters);
	fflush (stdout);
}

int main (int argc, char** argv)
{
	float* d;
	unsigned long long* dc;
	hipMalloc (&d, 256 * 16 * 512 * 4);
	hipMalloc (&dc, 64);
	hipMemset (dc, 0, 64);
	if (argc > 1 && argv[1][0] == 'f') {
		run_f32<1> ("16 v_mfma_f32_32x32x2_f32 per iteration, one wave per SIMD", 256, d, dc);
		run_f32<2> ("128 v_pk_fma_f32 per iteration, one wave per SIMD", 256, d, dc);
		run_f32<3> ("waves 0-3: 16 f32 MFMA | waves 4-7: 128 v_pk_fma_f32 (same SIMDs)", 512, d, dc);
		return 0;
	}
	if (argc > 1) {
		for (int w : {4, 8}) {
			run_mix<0, 0, 0> ("36 mfma16 + 4 ds_read_b128", w, d, dc);
			run_mix<0, 12, 0> ("36 mfma16 + 12 max3 (compiler order)", w, d, dc);
			run_mix<0, 12, 1> ("36 mfma16 + 12 max3 (1 VALU after each MFMA)", w, d, dc);
			run_mix<1, 0, 0> ("36 mfma16 + 1 K-filter step (12 pk), compiler order", w, d, dc);
			run_mix<2, 0, 0> ("36 mfma16 + 2 K-filter steps (24 pk), compiler order", w, d, dc);
			run_mix<2, 0, 1> ("36 mfma16 + 2 K-filter steps (24 pk), 1 VALU per MFMA", w, d, dc);
			run_mix<3, 0, 1> ("36 mfma16 + 3 K-filter steps (36 pk), 1 VALU per MFMA", w, d, dc);
			run_mix<3, 0, 2> ("36 mfma16 then 3 K-filter steps (36 pk) (serial)", w, d, dc);
			run_mix<3, 12, 1> ("36 mfma16 + 3 steps (36 pk) + 12 max3, 1 VALU per MFMA", w, d, dc);
			run_mix<4, 12, 1> ("36 mfma16 + 4 steps (48 pk) + 12 max3, 1 VALU per MFMA", w, d, dc);
			run_mix<4, 12, 0> ("36 mfma16 + 4 steps (48 pk) + 12 max3, compiler order", w, d, dc);
			run_mix<4, 12, 2> ("36 mfma16 then 4 steps (48 pk) + 12 max3 (serial)", w, d, dc);
			run_mix<4, 0, 2> ("(36 mfma16 then) 4 steps (48 pk) serial", w, d, dc);
		}
		return 0;
	}
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
