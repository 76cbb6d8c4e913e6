// tools/issue_model.hip — the issue model of one gfx950 wave that mixes MFMAs with VALU / LDS work, with the ORDER PINNED
// (every instruction is its own `asm volatile`: no scheduler, no unpacking of packed ops — tools/coissue.hip left the order
// to sched_group_barrier, which ignores packed-f32 instructions, so its packed rows measured the compiler's order).
//
// One wave per SIMD (256-thread workgroups, one per CU), GROUPS groups per iteration, every group =
//     1 MFMA (16x16x32 f16 or 32x32x16 f16, or none)  +  NS scalar v_fma_f32  +  NP v_pk_fma_f32  +  NL ds_read_b128
// DEP = 0: VALU ops rotate over 8 independent registers; DEP = 1: one dependent chain; DEP = 2: two interleaved chains.
// Output: shader cycles per group (s_memtime of wave 0 over the loop / groups).
// Build: hipcc --offload-arch=gfx950 -O3 tools/issue_model.hip -o tools/issue_model ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float v2f __attribute__ ((ext_vector_type (2)));
typedef float f4 __attribute__ ((ext_vector_type (4)));
typedef float f16v __attribute__ ((ext_vector_type (16)));
typedef _Float16 h8 __attribute__ ((ext_vector_type (8)));

constexpr int GROUPS = 12;

template <int SHAPE, int NS, int NP, int NL, int DEP, bool AGPR>
__global__ __launch_bounds__ (256) void k_im (float* out, unsigned long long* cyc, int iters, float seed)
{
	__shared__ __attribute__ ((aligned (16))) uint4 L[1024];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (int i = threadIdx.x; i < 1024; i += blockDim.x) L[i] = uint4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
	__syncthreads ();
	h8 a[3], b[2];
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16) (0.001f * (lane + i + j));
	for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (_Float16) (0.002f * (lane - i + j));
	if (AGPR) { asm volatile ("" : "+a"(a[0])); asm volatile ("" : "+a"(a[1])); asm volatile ("" : "+a"(a[2])); }
	f4 c4[6];
	f16v c16[3];
	for (int i = 0; i < 6; ++i) c4[i] = f4{0, 0, 0, 0};
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 16; ++j) c16[i][j] = 0;
	float f[8];
	v2f p[8];
	for (int i = 0; i < 8; ++i) { f[i] = seed + i + lane; p[i] = v2f{seed + i, seed - lane}; }
	const float m = 0.999f + seed;
	const v2f mp = v2f{m, m};
	uint4 ld[4] = {};
	const uint32_t la = (uint32_t) (lane * 16);

	unsigned long long t0, t1;
	asm volatile ("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int g = 0; g < GROUPS; ++g) {
			if constexpr (SHAPE == 16) {
				if constexpr (AGPR) asm volatile ("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c4[g % 6]) : "a"(a[g % 3]), "v"(b[g & 1]));
				else                asm volatile ("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c4[g % 6]) : "v"(a[g % 3]), "v"(b[g & 1]));
			} else if constexpr (SHAPE == 32) {
				if constexpr (AGPR) asm volatile ("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c16[g % 3]) : "a"(a[g % 3]), "v"(b[g & 1]));
				else                asm volatile ("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c16[g % 3]) : "v"(a[g % 3]), "v"(b[g & 1]));
			}
#pragma unroll
			for (int l = 0; l < NL; ++l)
				asm volatile ("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[(g * NL + l) & 3]) : "v"(la), "n"(1024 * ((g * NL + l) & 7)));
#pragma unroll
			for (int s = 0; s < NS; ++s) {
				const int q = DEP == 1 ? 0 : DEP == 2 ? (s & 1) : ((g * NS + s) & 7);
				asm volatile ("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q]) : "v"(m));
			}
#pragma unroll
			for (int s = 0; s < NP; ++s) {
				const int q = DEP == 1 ? 0 : DEP == 2 ? (s & 1) : ((g * NP + s) & 7);
				asm volatile ("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[q]) : "v"(mp));
			}
		}
		if (NL) asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
	}
	asm volatile ("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
	float r = 0;
	for (int i = 0; i < 8; ++i) r += f[i] + p[i].x + p[i].y;
	for (int i = 0; i < 6; ++i) r += c4[i][0] + c4[i][1] + c4[i][2] + c4[i][3];
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 16; ++j) r += c16[i][j];
	for (int i = 0; i < 4; ++i) r += (float) (ld[i].x + ld[i].y + ld[i].z + ld[i].w);
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
	if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}


// Which scalar instructions are "free" behind an MFMA?  Group = 1 MFMA (SHAPE) + N instructions of KIND, independent:
// 0 v_fma_f32, 1 v_fma_mixlo_f16 (f32 sources), 2 v_max3_f32 with |.| modifiers, 3 v_cvt_pk_f16_f32, 4 v_permlane16_swap_b32,
// 5 v_mul_f32, 6 v_fma_mixlo_f16 + v_fma_mixhi_f16 on ONE word (dependent halves), 7 v_accvgpr_read
template <int SHAPE, int KIND, int N>
__global__ __launch_bounds__ (256) void k_kind (float* out, unsigned long long* cyc, int iters, float seed)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	h8 a[3], b[2];
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16) (0.001f * (lane + i + j));
	for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (_Float16) (0.002f * (lane - i + j));
	f4 c4[6];
	f16v c16[3];
	for (int i = 0; i < 6; ++i) c4[i] = f4{0, 0, 0, 0};
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 16; ++j) c16[i][j] = 0;
	float f[8];
	uint32_t u[8];
	for (int i = 0; i < 8; ++i) { f[i] = seed + i + lane; u[i] = lane * 77 + i; }
	const float m = 0.999f + seed;
	float acc = seed;
	asm volatile ("" : "+a"(acc));
	unsigned long long t0, t1;
	asm volatile ("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int g = 0; g < GROUPS; ++g) {
			if constexpr (SHAPE == 16) asm volatile ("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c4[g % 6]) : "v"(a[g % 3]), "v"(b[g & 1]));
			else if constexpr (SHAPE == 32) asm volatile ("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c16[g % 3]) : "v"(a[g % 3]), "v"(b[g & 1]));
#pragma unroll
			for (int s = 0; s < N; ++s) {
				const int q = (g * N + s) & 7;
				if constexpr (KIND == 0) asm volatile ("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q]) : "v"(m));
				else if constexpr (KIND == 1) asm volatile ("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(u[q]) : "v"(f[q]), "v"(m));
				else if constexpr (KIND == 2) asm volatile ("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(f[q]) : "v"(f[(q + 1) & 7]), "v"(m));
				else if constexpr (KIND == 3) asm volatile ("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[q]) : "v"(f[q]), "v"(m));
				else if constexpr (KIND == 4) asm volatile ("v_permlane16_swap_b32 %0, %1" : "+v"(u[q]), "+v"(u[(q + 4) & 7]));
				else if constexpr (KIND == 5) asm volatile ("v_mul_f32 %0, %0, %1" : "+v"(f[q]) : "v"(m));
				else if constexpr (KIND == 6) {
					if (s & 1) asm volatile ("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(u[(q - 1) & 7]) : "v"(u[(q + 3) & 7]), "v"(m));
					else       asm volatile ("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(u[q]) : "v"(u[(q + 4) & 7]), "v"(m));
				} else asm volatile ("v_accvgpr_read_b32 %0, %1" : "=v"(f[q]) : "a"(acc));
			}
		}
	}
	asm volatile ("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
	float r = 0;
	for (int i = 0; i < 8; ++i) r += f[i] + (float) u[i];
	for (int i = 0; i < 6; ++i) r += c4[i][0] + c4[i][1] + c4[i][2] + c4[i][3];
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 16; ++j) r += c16[i][j];
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
	if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int SHAPE, int KIND, int N>
static void run_kind (float* d_out, unsigned long long* d_cyc)
{
	static const char* names[] = { "v_fma_f32", "v_fma_mixlo_f16", "v_max3_f32 |.|", "v_cvt_pk_f16_f32", "v_permlane16_swap_b32", "v_mul_f32", "v_fma_mixlo+mixhi (one word)", "v_accvgpr_read_b32" };
	const int iters = 300;
	auto kern = k_kind<SHAPE, KIND, N>;
	hipLaunchKernelGGL (kern, dim3 (256), dim3 (256), 0, 0, d_out, d_cyc, 4, 0.f);
	hipDeviceSynchronize ();
	hipLaunchKernelGGL (kern, dim3 (256), dim3 (256), 0, 0, d_out, d_cyc, iters, 0.f);
	hipDeviceSynchronize ();
	unsigned long long c[4] = {0};
	hipMemcpy (c, d_cyc, sizeof (c), hipMemcpyDeviceToHost);
	printf ("mfma%-2d + %d x %-30s : cycles/group %7.2f\n", SHAPE, N, names[KIND], (double) c[0] / iters / GROUPS);
	fflush (stdout);
}

template <int SHAPE, int NS, int NP, int NL, int DEP, bool AGPR>
static void run (float* d_out, unsigned long long* d_cyc)
{
	const int iters = 300;
	auto kern = k_im<SHAPE, NS, NP, NL, DEP, AGPR>;
	hipLaunchKernelGGL (kern, dim3 (256), dim3 (256), 0, 0, d_out, d_cyc, 4, 0.f);
	hipDeviceSynchronize ();
	hipEvent_t e0, e1;
	hipEventCreate (&e0); hipEventCreate (&e1);
	hipEventRecord (e0);
	hipLaunchKernelGGL (kern, dim3 (256), dim3 (256), 0, 0, d_out, d_cyc, iters, 0.f);
	hipEventRecord (e1);
	hipEventSynchronize (e1);
	float ms;
	hipEventElapsedTime (&ms, e0, e1);
	unsigned long long c[4] = {0};
	hipMemcpy (c, d_cyc, sizeof (c), hipMemcpyDeviceToHost);
	const double per = (double) c[0] / iters / GROUPS;
	// (s_memtime counts shader cycles on gfx950: 16.3 per 16x16x32 MFMA alone)
	printf ("mfma%-2d%s + %d v_fma + %d v_pk_fma + %d ds_read_b128  dep=%d : memtime/group %7.2f   ns/group %7.2f\n", SHAPE, AGPR ? "(A in AGPR)" : "",
	        NS, NP, NL, DEP, per, ms * 1e6 / iters / GROUPS);
	fflush (stdout);
}

int main ()
{
	float* d;
	unsigned long long* dc;
	hipMalloc (&d, 256 * 256 * 4);
	hipMalloc (&dc, 64);
	hipMemset (dc, 0, 64);
	puts ("# --- MFMA alone");
	run<16, 0, 0, 0, 0, false> (d, dc);
	run<16, 0, 0, 0, 0, true> (d, dc);
	run<32, 0, 0, 0, 0, false> (d, dc);
	puts ("# --- VALU alone: 8 ops per group");
	run<0, 8, 0, 0, 0, false> (d, dc);
	run<0, 8, 0, 0, 1, false> (d, dc);
	run<0, 8, 0, 0, 2, false> (d, dc);
	run<0, 0, 8, 0, 0, false> (d, dc);
	run<0, 0, 8, 0, 1, false> (d, dc);
	run<0, 0, 8, 0, 2, false> (d, dc);
	run<0, 0, 0, 4, 0, false> (d, dc);
	puts ("# --- 16x16x32 + n scalar");
	run<16, 1, 0, 0, 0, false> (d, dc);
	run<16, 2, 0, 0, 0, false> (d, dc);
	run<16, 3, 0, 0, 0, false> (d, dc);
	run<16, 4, 0, 0, 0, false> (d, dc);
	run<16, 6, 0, 0, 0, false> (d, dc);
	run<16, 2, 0, 0, 1, false> (d, dc);
	run<16, 3, 0, 0, 1, false> (d, dc);
	puts ("# --- 16x16x32 + n packed");
	run<16, 0, 1, 0, 0, false> (d, dc);
	run<16, 0, 2, 0, 0, false> (d, dc);
	run<16, 0, 3, 0, 0, false> (d, dc);
	run<16, 0, 4, 0, 0, false> (d, dc);
	run<16, 0, 1, 0, 1, false> (d, dc);
	run<16, 0, 2, 0, 1, false> (d, dc);
	run<16, 1, 1, 0, 0, false> (d, dc);
	run<16, 2, 1, 0, 0, false> (d, dc);
	puts ("# --- 16x16x32 + LDS");
	run<16, 2, 0, 1, 0, false> (d, dc);
	run<16, 0, 0, 1, 0, false> (d, dc);
	run<16, 0, 0, 2, 0, false> (d, dc);
	puts ("# --- 32x32x16 + n");
	run<32, 2, 0, 0, 0, false> (d, dc);
	run<32, 4, 0, 0, 0, false> (d, dc);
	run<32, 5, 0, 0, 0, false> (d, dc);
	run<32, 6, 0, 0, 0, false> (d, dc);
	run<32, 8, 0, 0, 0, false> (d, dc);
	run<32, 0, 1, 0, 0, false> (d, dc);
	run<32, 0, 2, 0, 0, false> (d, dc);
	run<32, 0, 3, 0, 0, false> (d, dc);
	run<32, 0, 4, 0, 0, false> (d, dc);
	run<32, 0, 2, 0, 1, false> (d, dc);
	run<32, 0, 3, 0, 1, false> (d, dc);
	run<32, 4, 0, 1, 0, false> (d, dc);
	run<32, 4, 0, 0, 0, true> (d, dc);
	puts ("# --- which instructions ride behind an MFMA");
	run_kind<16, 0, 2> (d, dc); run_kind<16, 1, 2> (d, dc); run_kind<16, 2, 2> (d, dc); run_kind<16, 3, 2> (d, dc);
	run_kind<16, 4, 2> (d, dc); run_kind<16, 5, 2> (d, dc); run_kind<16, 6, 2> (d, dc); run_kind<16, 7, 2> (d, dc);
	run_kind<16, 4, 1> (d, dc); run_kind<16, 1, 1> (d, dc);
	run_kind<0, 1, 8> (d, dc); run_kind<0, 2, 8> (d, dc); run_kind<0, 3, 8> (d, dc); run_kind<0, 4, 8> (d, dc);
	run_kind<32, 0, 5> (d, dc); run_kind<32, 1, 4> (d, dc); run_kind<32, 2, 5> (d, dc); run_kind<32, 4, 4> (d, dc); run_kind<32, 4, 5> (d, dc);
	return 0;
}
