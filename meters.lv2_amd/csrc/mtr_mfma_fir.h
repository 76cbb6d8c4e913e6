// mtr_mfma_fir.h — the 4x interpolator as a block-Toeplitz product on the matrix pipe (gfx950).
//
// Optional true-peak form (tune_fir = 2), NOT the default: the exact-f32 VALU interpolator of
// mtr_fused2.hip is fp32-VALU bound at ~17 % of the HBM roofline; this one moves the 144 MACs per
// channel-sample to v_mfma_f32_32x32x16_f16 with the samples split into two halves (x = hi + lo,
// 22 significand bits) and f32 accumulation.  The taps are rounded to f16, which bounds the deviation
// from the f32 result by 2^-12 * L1 * max|x| = 6.3e-4 relative = 0.0055 dB in the worst case (every
// rounding error aligned with the sign of its sample) — inside the +-0.01 dB the parity clause states
// for dBTP, and ~1e-4 dB on programme material (tests/test_gpu_mfma.py measures both).
//
// One MFMA tile = 256 output frames x 4 phases of ONE channel:
//     row m = 8 p + b  (phase p = 0..3, offset b = 0..7),  column c = 0..31:  output frame 8 c + b
//     Y[m][c] = sum_k A[m][k] * X[k][c],   A[8p+b][k] = h_p[k - b] (0 <= k - b < 48),  X[k][c] = x[n0 - 47 + 8 c + k]
// so X is just the sample array read at stride 8: lane (c, g) of step j needs the 4 samples
// 8 c + 8 j + 4 g + 0..3 — with the two halves of a sample stored side by side that is ONE 16-byte LDS
// read per MFMA.  K = 56 samples x {hi, lo} = 112 = 7 steps of 16.  Phase 0 (the identity branch,
// x[n - 24]) rides along as rows 0..7, so |x| needs no separate pass.
#pragma once
#include <stdint.h>
#include <string.h>

#define MTR_MFMA_STEPS     7      /* K = 112 = 7 x 16 */
/* The residual is stored as lo * 2^11 (|lo| <= 2^-11 |x|, so it spans the range of x itself and keeps 11 bits
 * down to |x| = 2^-25: a stream that peaks at -110 dBFS is still read to 1e-4 dB) and its taps as h * 2^-11;
 * those are f16 subnormals for |h| < 0.125, which the matrix pipe does not flush (tools/mfma_probe.hip). */
#define MTR_MFMA_LO_SHIFT  11
/* The taps are stored as h * 2^10 (the result is scaled back, exactly, after the maximum): the outermost taps
 * are ~1e-6 and would be f16 subnormals with a handful of bits — visible when a call's peak is nothing but
 * the pre-ringing of one impulse at its edge (tests/test_gpu_mfma.py::test_mfma_layout_edge_signals). */
#define MTR_MFMA_TAP_SHIFT 10
#define MTR_MFMA_A_HALVES  (MTR_MFMA_STEPS * 64 * 8)

/* float -> IEEE half, round to nearest even (host side, table set-up only) */
static inline uint16_t mtr_f32_to_f16 (float f)
{
	uint32_t x;
	memcpy (&x, &f, 4);
	const uint32_t sign = (x >> 16) & 0x8000u;
	const int32_t e = (int32_t) ((x >> 23) & 0xff) - 127 + 15;
	uint32_t m = x & 0x7fffffu;
	if (((x >> 23) & 0xff) == 0xff) return (uint16_t) (sign | 0x7c00u | (m ? 0x200u : 0));
	if (e >= 31) return (uint16_t) (sign | 0x7c00u);
	if (e <= 0) {
		if (e < -10) return (uint16_t) sign;
		m |= 0x800000u;
		const int sh = 14 - e;                       /* 14 .. 24 */
		uint32_t r = m >> sh;
		const uint32_t rem = m & ((1u << sh) - 1), half = 1u << (sh - 1);
		if (rem > half || (rem == half && (r & 1))) ++r;
		return (uint16_t) (sign | r);
	}
	uint32_t r = ((uint32_t) e << 10) | (m >> 13);
	const uint32_t rem = m & 0x1fffu;
	if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;   /* may carry into the exponent: still right */
	return (uint16_t) (sign | r);
}

/* A fragments for the 7 steps: out[(j * 64 + lane) * 8 + i] = A[m = lane & 31][k' = 16 j + 8 (lane >> 5) + i],
 * k' = 2 t + part: sample t of the 56-sample window, part 0 = multiplies hi, part 1 = multiplies lo * 2^11.
 * g = the 48-tap kernels of phases 1..3 in window order (g[ph-1][i] multiplies x[n - 47 + i]). */
static inline void mtr_mfma_build_a (const float* g /* [3][48] */, uint16_t* out /* [MTR_MFMA_A_HALVES] */)
{
	for (int j = 0; j < MTR_MFMA_STEPS; ++j)
		for (int lane = 0; lane < 64; ++lane)
			for (int i = 0; i < 8; ++i) {
				const int m = lane & 31, p = m >> 3, b = m & 7;
				const int kp = 16 * j + 8 * (lane >> 5) + i, t = kp >> 1, part = kp & 1;
				const int tau = t - b;
				float h = 0.f;
				if (tau >= 0 && tau < 48) h = p == 0 ? (tau == 23 ? 1.f : 0.f) : g[48 * (p - 1) + tau];
				h *= (float) (1 << MTR_MFMA_TAP_SHIFT);
				if (part) h *= 1.f / (float) (1 << MTR_MFMA_LO_SHIFT);
				out[(j * 64 + lane) * 8 + i] = mtr_f32_to_f16 (h);
			}
}

#ifdef __HIPCC__
#include <hip/hip_runtime.h>

namespace mfir {

typedef _Float16 h8 __attribute__ ((ext_vector_type (8)));
typedef float f16x __attribute__ ((ext_vector_type (16)));

struct AFrag {
	h8 a[MTR_MFMA_STEPS];
	__device__ __forceinline__ void load (const uint16_t* tab, int lane)
	{
#pragma unroll
		for (int j = 0; j < MTR_MFMA_STEPS; ++j) a[j] = *reinterpret_cast<const h8*> (tab + (j * 64 + lane) * 8);
	}
};

// {hi, lo * 2^11} of one sample as one LDS word (hi in the low half)
__device__ __forceinline__ uint32_t split_word (float x)
{
	const _Float16 hi = (_Float16) x;
	const _Float16 lo = (_Float16) ((x - (float) hi) * (float) (1 << MTR_MFMA_LO_SHIFT));
	return (uint32_t) __builtin_bit_cast (uint16_t, hi) | ((uint32_t) __builtin_bit_cast (uint16_t, lo) << 16);
}

// Both channels of one frame at once: {hi, lo} words of the left and of the right sample.  Round-toward-zero
// packs (one instruction per pair): hi then carries 10 bits and lo the rest, |lo * 2^11| < 2 |x|.
__device__ __forceinline__ void split_words (float xl, float xr, uint32_t& wl, uint32_t& wr)
{
	typedef __fp16 hf2 __attribute__ ((ext_vector_type (2)));
	const hf2 hi = __builtin_amdgcn_cvt_pkrtz (xl, xr);
	const float s = (float) (1 << MTR_MFMA_LO_SHIFT);
	const hf2 lo = __builtin_amdgcn_cvt_pkrtz ((xl - (float) hi[0]) * s, (xr - (float) hi[1]) * s);
	const uint32_t h = __builtin_bit_cast (uint32_t, hi), l = __builtin_bit_cast (uint32_t, lo);
	wl = __builtin_amdgcn_perm (l, h, 0x05040100u);              // {hi left, lo left}
	wr = __builtin_amdgcn_perm (l, h, 0x07060302u);              // {hi right, lo right}
}

// One tile: W = the channel's word array, base = index of the window start of output frame 0
// (a multiple of 4: 16-byte reads).  Result: acc[r] = Y[(r & 3) + 8 (r >> 2) + 4 (lane >> 5)][lane & 31],
// i.e. phase r >> 2 of output frame 8 (lane & 31) + (r & 3) + 4 (lane >> 5), times 2^MTR_MFMA_TAP_SHIFT.
__device__ __forceinline__ f16x tile (const AFrag& A, const uint32_t* W, int base, int lane)
{
	const uint4* const p = reinterpret_cast<const uint4*> (W + base + 8 * (lane & 31) + 4 * (lane >> 5));
	f16x acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
	for (int j = 0; j < MTR_MFMA_STEPS; ++j) {
		const uint4 w = p[2 * j];                                    // 8 words further per step
		acc = __builtin_amdgcn_mfma_f32_32x32x16_f16 (A.a[j], __builtin_bit_cast (h8, w), acc, 0, 0, 0);
	}
	return acc;
}

// The operands of one tile (7 x 16 bytes per lane) and the product for both channels with the two
// accumulator chains interleaved (a dependent MFMA waits for its predecessor; two chains keep the pipe busy).
struct BFrag { uint4 w[MTR_MFMA_STEPS]; };
// `last` = the highest word index a fragment may start at (array words - 52): columns past the end of a short
// tile are masked by the caller, what they read only has to lie inside the array
__device__ __forceinline__ void fetch_b (BFrag& B, const uint32_t* W, int base, int lane, int last)
{
	const int o = base + 8 * (lane & 31) + 4 * (lane >> 5);
	const uint4* const p = reinterpret_cast<const uint4*> (W + (o < last ? o : last));
#pragma unroll
	for (int j = 0; j < MTR_MFMA_STEPS; ++j) B.w[j] = p[2 * j];
}
__device__ __forceinline__ void tile2 (const AFrag& A, const BFrag& L, const BFrag& Rr, f16x& yl, f16x& yr)
{
	const f16x z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	yl = z; yr = z;
#pragma unroll
	for (int j = 0; j < MTR_MFMA_STEPS; ++j) {
		yl = __builtin_amdgcn_mfma_f32_32x32x16_f16 (A.a[j], __builtin_bit_cast (h8, L.w[j]), yl, 0, 0, 0);
		yr = __builtin_amdgcn_mfma_f32_32x32x16_f16 (A.a[j], __builtin_bit_cast (h8, Rr.w[j]), yr, 0, 0, 0);
	}
}

}  // namespace mfir
#endif
