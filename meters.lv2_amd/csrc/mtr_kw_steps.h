// mtr_kw_steps.h — the K-weighting step (ebu_r128_proc.cc:321-326) for both channels of a frame: the plain form and the
// hand-scheduled pair used by pass 2 of the time-parallel K-filter (k_kwtp16, k_kw).
#pragma once
#include <hip/hip_runtime.h>

#include "mtr_wave.h"

// (explicit fused multiply-adds, in kw_pair's association: the tail step of k_kw and the pairs round the same way whatever
// -ffp-contract says — ADVICE r2)
#define KW_STEP(p, y)                                                   \
	{                                                                   \
		v2f t_ = (p) + 1e-15f;                                          \
		v2f u_ = (v2f) (a1) * z1;                                       \
		t_ = __builtin_elementwise_fma (-(v2f) (b2), z2, t_);                   \
		u_ = __builtin_elementwise_fma ((v2f) (a2), z2, u_);                    \
		const v2f x_ = __builtin_elementwise_fma (-(v2f) (b1), z1, t_);         \
		u_ = __builtin_elementwise_fma (-(v2f) (c4), z4, u_);                   \
		const v2f z4n_ = z4 + z3;                                       \
		u_ = __builtin_elementwise_fma (-(v2f) (c3), z3, u_);                   \
		y = __builtin_elementwise_fma ((v2f) (a0), x_, u_);                     \
		z2 = z1; z1 = x_; z4 = z4n_; z3 += y;                           \
	}

// Two K-weighting steps (frames n, n + 1 of every lane's run) as one hand-scheduled block.  What hipcc makes of the
// masked C++ loop (one lane per tile has a partial run) is, per step, a scalar branch, an exec save / restore, five
// 64-bit moves for the phi nodes and a wait state between every pair of dependent packed instructions; here the
// lane mask is picked without a branch (steps n < rl run under `upto`, the others under `before`: two SALU
// instructions), the shelving states ping-pong between two registers (x_ of step n overwrites z2, which is z1 of
// step n + 1), and the next step's first instructions fill the slots behind the dependent ones: 22 packed
// instructions, 2 wait states, 5 SALU per pair.  Same operations in the same association as KW_STEP.
// On entry zA = z1, zB = z2; a lane that ran an odd number of steps holds them swapped (the caller picks).
template <int N>
__device__ __forceinline__ void kw_pair (v2f x0, v2f x1, v2f& zA, v2f& zB, v2f& z3, v2f& z4, v2f& sj, v2f a0, v2f a1, v2f a2,
                                         v2f b1, v2f b2, v2f c3, v2f c4, v2f eps, uint64_t upto, uint64_t before, int rl)
{
	v2f t, u, y, t2, u2;
	uint64_t ex;                                   // the caller's lane mask: restored on the way out, whatever it was
	asm volatile (
		"s_mov_b64 %[ex], exec\n\t"
		"s_cmp_gt_i32 %[rl], %[n0]\n\t"
		"s_cselect_b64 exec, %[upto], %[before]\n\t"
		"v_pk_add_f32 %[t], %[x0], %[eps]\n\t"
		"v_pk_mul_f32 %[u], %[a1], %[zA]\n\t"
		"v_pk_fma_f32 %[t], %[b2], %[zB], %[t] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
		"v_pk_fma_f32 %[u], %[a2], %[zB], %[u]\n\t"
		"v_pk_fma_f32 %[zB], %[b1], %[zA], %[t] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
		"v_pk_fma_f32 %[u], %[c4], %[z4], %[u] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
		"v_pk_add_f32 %[z4], %[z4], %[z3]\n\t"
		"v_pk_fma_f32 %[u], %[c3], %[z3], %[u] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
		"v_pk_add_f32 %[t2], %[x1], %[eps]\n\t"
		"v_pk_fma_f32 %[y], %[a0], %[zB], %[u]\n\t"
		"v_pk_mul_f32 %[u2], %[a1], %[zB]\n\t"
		"v_pk_add_f32 %[z3], %[z3], %[y]\n\t"
		"v_pk_fma_f32 %[sj], %[y], %[y], %[sj]\n\t"
		"s_cmp_gt_i32 %[rl], %[n1]\n\t"
		"s_cselect_b64 exec, %[upto], %[before]\n\t"
		"v_pk_fma_f32 %[t2], %[b2], %[zA], %[t2] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
		"v_pk_fma_f32 %[u2], %[a2], %[zA], %[u2]\n\t"
		"v_pk_fma_f32 %[zA], %[b1], %[zB], %[t2] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
		"v_pk_fma_f32 %[u2], %[c4], %[z4], %[u2] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
		"v_pk_add_f32 %[z4], %[z4], %[z3]\n\t"
		"v_pk_fma_f32 %[u2], %[c3], %[z3], %[u2] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
		"s_nop 0\n\t"
		"v_pk_fma_f32 %[y], %[a0], %[zA], %[u2]\n\t"
		"s_nop 0\n\t"
		"v_pk_add_f32 %[z3], %[z3], %[y]\n\t"
		"v_pk_fma_f32 %[sj], %[y], %[y], %[sj]\n\t"
		"s_mov_b64 exec, %[ex]"
		: [zA] "+v"(zA), [zB] "+v"(zB), [z3] "+v"(z3), [z4] "+v"(z4), [sj] "+v"(sj),
		  [t] "=&v"(t), [u] "=&v"(u), [y] "=&v"(y), [t2] "=&v"(t2), [u2] "=&v"(u2), [ex] "=&s"(ex)
		: [x0] "v"(x0), [x1] "v"(x1), [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [b1] "v"(b1), [b2] "v"(b2), [c3] "v"(c3), [c4] "v"(c4),
		  [eps] "v"(eps), [upto] "s"(upto), [before] "s"(before), [rl] "s"(rl), [n0] "n"(N), [n1] "n"(N + 1)
		: "scc");      // (exec is saved and restored inside the block: unchanged as far as the compiler can see)
}

