// mtr_mfma16_fir.h — the 4x interpolator on the matrix pipe at f32 grade (gfx950): layouts 6 and 7 (k_kwtp16, k_seg) and k_tpb.
//
// y_p[n] = sum_i g_p[i] x[n - 47 + i]  (p = 1..3; phase 0 is x[n - 24] itself and stays on the VALU) is a
// block-Toeplitz product.  Samples AND taps are carried as two f16 halves each,
//     x * 2^s = Xhi + Xlo,      g * 2^15 = Ghi + Glo        (s: per tile and channel, from max |x|)
// and the product keeps the three terms that matter,
//     y * 2^(s+15) = sum Ghi Xhi + sum Ghi Xlo + sum Glo Xhi            (Glo Xlo < 2^-22 of the first: dropped)
// accumulated in f32 by v_mfma_f32_16x16x32_f16.  Every f16 product is exact in f32, so what is left is the
// 2^-23-relative truncation of each factor — the same order as the f32 fmaf chain's own rounding
// (48 x 2^-24 per output in the worst case).  tests/test_gpu_parity.py holds this path to the same 2e-6
// relative bound as the exact-f32 VALU interpolator.
//
// One MFMA = 16 offsets x 16 columns x 32 window samples of ONE phase and channel:
//     row m (0..15): output frame 16 c + m of column c;  column c reads the 64-sample window that starts
//     at array position 16 c (position i <-> frame t0 - 48 + i), in two steps of 32 samples:
//     A[m][t] = g_p[t - 1 - m]  (0 <= t - 1 - m < 48),   B[t][c] = X[16 c + t].
// Samples are stored two per 32-bit word, {X[2w], X[2w+1]}, hi and lo halves in separate arrays per channel:
// lane (c, kg = lane >> 4) reads words 8 c + 16 step + 4 kg .. + 3 — ONE aligned ds_read_b128, consecutive
// lanes 32 bytes apart, conflict-free in the b128 lane groups — and the hi fragments serve two products.
// Per 256 output frames and channel: 4 operand reads, 3 phases x 6 MFMAs.
#pragma once
#include <stdint.h>
#include <string.h>


#define MTR_M16_TAP_SHIFT 15        /* taps as g * 2^15: the largest (0.9) stays below the f16 maximum */
#define MTR_M16_FRAGS     12        /* [phase 3][step 2][part 2: hi, lo] */
#define MTR_M16_A_HALVES  (MTR_M16_FRAGS * 64 * 8)
#define MTR_M16_HALO      48        /* positions in front of a tile (47 frames of history + 1: pairs stay aligned) */

/* float -> IEEE half (round to nearest even) and back, for the host-side tap tables */
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

static inline float mtr_f16_to_f32 (uint16_t h)
{
	const uint32_t sign = (uint32_t) (h & 0x8000u) << 16;
	const int e = (h >> 10) & 31;
	uint32_t m = h & 0x3ffu, x;
	if (e == 0) {
		if (m == 0) x = sign;
		else {
			int sh = 0;
			while (!(m & 0x400u)) { m <<= 1; ++sh; }
			x = sign | ((uint32_t) (127 - 15 - sh + 1) << 23) | ((m & 0x3ffu) << 13);
		}
	} else if (e == 31) x = sign | 0x7f800000u | (m << 13);
	else x = sign | ((uint32_t) (e - 15 + 127) << 23) | (m << 13);
	float f;
	memcpy (&f, &x, 4);
	return f;
}

/* A fragments: out[((p * 2 + step) * 2 + part) * 512 + lane * 8 + e] = part of A[m = lane & 15][t = 32 step + 8 (lane >> 4) + e].
 * g = the 48-tap kernels of phases 1..3 in window order (g[p][i] multiplies x[n - 47 + i]). */
static inline void mtr_m16_build_a (const float* g /* [3][48] */, uint16_t* out /* [MTR_M16_A_HALVES] */)
{
	for (int p = 0; p < 3; ++p)
		for (int step = 0; step < 2; ++step)
			for (int lane = 0; lane < 64; ++lane)
				for (int e = 0; e < 8; ++e) {
					const int m = lane & 15, t = 32 * step + 8 * (lane >> 4) + e, i = t - 1 - m;
					float h = (i >= 0 && i < 48) ? g[48 * p + i] : 0.f;
					h *= (float) (1 << MTR_M16_TAP_SHIFT);                    /* exact */
					const uint16_t hi = mtr_f32_to_f16 (h);
					const uint16_t lo = mtr_f32_to_f16 (h - mtr_f16_to_f32 (hi));   /* exact difference, then one rounding */
					out[((p * 2 + step) * 2 + 0) * 512 + lane * 8 + e] = hi;
					out[((p * 2 + step) * 2 + 1) * 512 + lane * 8 + e] = lo;
				}
}

#ifdef __HIPCC__
#include <hip/hip_runtime.h>

namespace m16 {

typedef _Float16 h8 __attribute__ ((ext_vector_type (8)));
typedef _Float16 h2 __attribute__ ((ext_vector_type (2)));
typedef float f4 __attribute__ ((ext_vector_type (4)));
typedef float v2f_ __attribute__ ((ext_vector_type (2)));

struct AFrag {
	h8 a[MTR_M16_FRAGS];
	__device__ __forceinline__ void load (const uint16_t* tab, int lane)
	{
#pragma unroll
		for (int f = 0; f < MTR_M16_FRAGS; ++f) a[f] = *reinterpret_cast<const h8*> (tab + f * 512 + lane * 8);
	}
};

// Two consecutive samples of one channel, already scaled: hi = round-to-nearest f16 pair, lo = the exact f32
// remainders rounded to f16 (v_fma_mix: f16 source, f32 addend, f16 result — one instruction per half).
__device__ __forceinline__ uint32_t hi_pair (float x0, float x1)
{
	return __builtin_bit_cast (uint32_t, __builtin_convertvector (v2f_{x0, x1}, h2));           // v_cvt_pk_f16_f32
}
__device__ __forceinline__ uint32_t lo_pair (uint32_t hw, float x0, float x1)
{
	uint32_t lw;
	asm ("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
	     "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
	     : "=&v"(lw) : "v"(hw), "v"(x0), "v"(x1));
	return lw;
}
__device__ __forceinline__ void split_pair (float x0, float x1, uint32_t& hi, uint32_t& lo)
{
	hi = hi_pair (x0, x1);
	lo = lo_pair (hi, x0, x1);
}

// the four operand fragments of one channel for one 256-frame block: hi / lo x window steps 0, 1
struct BFrag { uint4 h0, h1, l0, l1; };
__device__ __forceinline__ void fetch_b (BFrag& B, const uint32_t* H, const uint32_t* L, int w)
{
	B.h0 = *reinterpret_cast<const uint4*> (H + w);
	B.h1 = *reinterpret_cast<const uint4*> (H + w + 16);
	B.l0 = *reinterpret_cast<const uint4*> (L + w);
	B.l1 = *reinterpret_cast<const uint4*> (L + w + 16);
}

#define M16_MFMA(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_f16 (A_, __builtin_bit_cast (m16::h8, B_), C_, 0, 0, 0)

// y[p][r] = phase p + 1 of output frame 16 (16 blk + (lane & 15)) + 4 (lane >> 4) + r, times 2^(s + 15)
__device__ __forceinline__ void block (const AFrag& A, const BFrag& B, f4 (&y)[3])
{
#pragma unroll
	for (int p = 0; p < 3; ++p) y[p] = f4{0.f, 0.f, 0.f, 0.f};
	// ordered so that consecutive MFMAs write different accumulators (a dependent MFMA waits for its predecessor)
#pragma unroll
	for (int p = 0; p < 3; ++p) M16_MFMA (A.a[(p * 2 + 0) * 2 + 0], B.h0, y[p]);
#pragma unroll
	for (int p = 0; p < 3; ++p) M16_MFMA (A.a[(p * 2 + 1) * 2 + 0], B.h1, y[p]);
#pragma unroll
	for (int p = 0; p < 3; ++p) M16_MFMA (A.a[(p * 2 + 0) * 2 + 0], B.l0, y[p]);
#pragma unroll
	for (int p = 0; p < 3; ++p) M16_MFMA (A.a[(p * 2 + 1) * 2 + 0], B.l1, y[p]);
#pragma unroll
	for (int p = 0; p < 3; ++p) M16_MFMA (A.a[(p * 2 + 0) * 2 + 1], B.h0, y[p]);
#pragma unroll
	for (int p = 0; p < 3; ++p) M16_MFMA (A.a[(p * 2 + 1) * 2 + 1], B.h1, y[p]);
}

// MFMA number I (0..17) of block (), for a caller that places other work between the products by hand (mtr_seg.hip);
// I < 3 start from a zero accumulator.
template <int I>
__device__ __forceinline__ void block_mfma (const AFrag& A, const BFrag& B, f4 (&y)[3])
{
	constexpr int g = I / 3, p = I % 3;
	constexpr int ai = (p * 2 + (g & 1)) * 2 + (g >= 4 ? 1 : 0);
	const uint4& b = g == 0 ? B.h0 : g == 1 ? B.h1 : g == 2 ? B.l0 : g == 3 ? B.l1 : g == 4 ? B.h0 : B.h1;
	const f4 c = I < 3 ? f4{0.f, 0.f, 0.f, 0.f} : y[p];
	y[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16 (A.a[ai], __builtin_bit_cast (h8, b), c, 0, 0, 0);
}

// The same products in two instalments, for the refinement form of exact pruning (tune_prune = 2): block_first is
// Ghi Xhi alone, block_rest adds Ghi Xlo and Glo Xhi — the MFMAs of block () in the same order on the same accumulators,
// so first + rest is bit-for-bit block ().
__device__ __forceinline__ void block_first (const AFrag& A, const uint4& h0, const uint4& h1, f4 (&y)[3])
{
#pragma unroll
	for (int p = 0; p < 3; ++p) y[p] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
	for (int p = 0; p < 3; ++p) M16_MFMA (A.a[(p * 2 + 0) * 2 + 0], h0, y[p]);
#pragma unroll
	for (int p = 0; p < 3; ++p) M16_MFMA (A.a[(p * 2 + 1) * 2 + 0], h1, y[p]);
}
__device__ __forceinline__ void block_rest (const AFrag& A, const uint4& h0, const uint4& h1, const uint4& l0, const uint4& l1, f4 (&y)[3])
{
#pragma unroll
	for (int p = 0; p < 3; ++p) M16_MFMA (A.a[(p * 2 + 0) * 2 + 0], l0, y[p]);
#pragma unroll
	for (int p = 0; p < 3; ++p) M16_MFMA (A.a[(p * 2 + 1) * 2 + 0], l1, y[p]);
#pragma unroll
	for (int p = 0; p < 3; ++p) M16_MFMA (A.a[(p * 2 + 0) * 2 + 1], h0, y[p]);
#pragma unroll
	for (int p = 0; p < 3; ++p) M16_MFMA (A.a[(p * 2 + 1) * 2 + 1], h1, y[p]);
}

}  // namespace m16
#endif
