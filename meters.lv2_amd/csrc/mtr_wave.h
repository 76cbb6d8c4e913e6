// mtr_wave.h — wave64 cross-lane primitives for the K-weighting scan, on the DPP path (gfx950).
//
// ds_bpermute_b32 goes through the LDS crossbar: measured 24 cycles of issue per SIMD and 61 cycles of
// dependent latency (tools/ubench.hip).  The data-parallel-primitive modifiers move data inside the
// VALU instead (a v_mov_b32_dpp issues like any other VALU op), but only along fixed patterns:
// shifts inside a row of 16 lanes, "lane 15 of every row to the next row", "lane 31 to rows 2-3",
// and a whole-wave shift by one.  That is exactly enough for a prefix scan of a linear recurrence:
// four in-row Hillis-Steele steps (offsets 1, 2, 4, 8; out-of-row sources read as zero), then two
// row-total broadcasts whose matrices depend on the receiving lane's position (M^(i+1) with
// i = lane & 15, and M^(lane-31)), held in 24 VGPRs.
#pragma once
#include <hip/hip_runtime.h>

typedef float v2f __attribute__ ((ext_vector_type (2)));

namespace mtrw {

// dpp_ctrl encodings (GFX9 ISA, "DPP_CTRL"): row_shr:n = 0x110+n, wave_shr:1 = 0x138,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp0 (float v)
{
	// old = 0 and bound_ctrl: lanes without a source (out of row, masked row) read 0
	return __int_as_float (__builtin_amdgcn_update_dpp (0, __float_as_int (v), CTRL, ROW_MASK, 0xF, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ v2f dpp0 (v2f v) { return v2f{dpp0<CTRL, ROW_MASK> (v.x), dpp0<CTRL, ROW_MASK> (v.y)}; }

// value of the lane to the left; lane 0 reads 0
__device__ __forceinline__ v2f from_left (v2f v) { return dpp0<0x138, 0xF> (v); }

// wave-uniform lane pick (scalar lane index): v_readlane_b32, the result lives in an SGPR
__device__ __forceinline__ v2f pick (v2f v, int l)
{
	return v2f{__int_as_float (__builtin_amdgcn_readlane (__float_as_int (v.x), l)),
	           __int_as_float (__builtin_amdgcn_readlane (__float_as_int (v.y), l))};
}

// sum over the wave, result valid in lane 63 only and returned from there
__device__ __forceinline__ float sum63 (float v)
{
	v += dpp0<0x111, 0xF> (v);
	v += dpp0<0x112, 0xF> (v);
	v += dpp0<0x114, 0xF> (v);
	v += dpp0<0x118, 0xF> (v);
	v += dpp0<0x142, 0xA> (v);
	v += dpp0<0x143, 0xC> (v);
	return __int_as_float (__builtin_amdgcn_readlane (__float_as_int (v), 63));
}

// max over the wave of NON-NEGATIVE values (lanes without a source read 0), returned wave-uniform
__device__ __forceinline__ float max63 (float v)
{
	v = fmaxf (v, dpp0<0x111, 0xF> (v));
	v = fmaxf (v, dpp0<0x112, 0xF> (v));
	v = fmaxf (v, dpp0<0x114, 0xF> (v));
	v = fmaxf (v, dpp0<0x118, 0xF> (v));
	v = fmaxf (v, dpp0<0x142, 0xA> (v));
	v = fmaxf (v, dpp0<0x143, 0xC> (v));
	return __int_as_float (__builtin_amdgcn_readlane (__float_as_int (v), 63));
}

// componentwise max of two 16-bit fields per lane over the wave (v_pk_max_u16), returned wave-uniform
__device__ __forceinline__ uint32_t max63_u16x2 (uint32_t v)
{
	typedef unsigned short us2 __attribute__ ((ext_vector_type (2)));
#define MTRW_PKMAX(CTRL, MASK)                                                                                  \
	{                                                                                                           \
		const uint32_t o = (uint32_t) __builtin_amdgcn_update_dpp (0, (int) v, CTRL, MASK, 0xF, true);         \
		v = __builtin_bit_cast (uint32_t, __builtin_elementwise_max (__builtin_bit_cast (us2, v), __builtin_bit_cast (us2, o))); \
	}
	MTRW_PKMAX (0x111, 0xF) MTRW_PKMAX (0x112, 0xF) MTRW_PKMAX (0x114, 0xF) MTRW_PKMAX (0x118, 0xF)
	MTRW_PKMAX (0x142, 0xA) MTRW_PKMAX (0x143, 0xC)
#undef MTRW_PKMAX
	return (uint32_t) __builtin_amdgcn_readlane ((int) v, 63);
}

// One application of a K-weighting transition-matrix power (block lower triangular: the shelving
// stage does not see the integrators): z += M w.
#define MTRW_APPLY(M, w1, w2, w3, w4)                                                     \
	{                                                                                     \
		z1 += (M)[0] * (w1);  z1 += (M)[1] * (w2);                                        \
		z2 += (M)[4] * (w1);  z2 += (M)[5] * (w2);                                        \
		z3 += (M)[8] * (w1);  z3 += (M)[9] * (w2);  z3 += (M)[10] * (w3); z3 += (M)[11] * (w4); \
		z4 += (M)[12] * (w1); z4 += (M)[13] * (w2); z4 += (M)[14] * (w3); z4 += (M)[15] * (w4); \
	}

// Per-lane matrices of the two row-broadcast steps: 12 non-zero entries each.
struct RowMats {
	float p[12], q[12];
	// pw = table of M^1 .. M^32 (16 floats each, row major)
	__device__ __forceinline__ void load (const float* pw, int lane)
	{
		const float* P = pw + 16 * (lane & 15);                 // M^((lane & 15) + 1)
		const float* Q = pw + 16 * (lane >= 32 ? lane - 32 : 0); // M^(lane - 31), lanes 32..63
		p[0] = P[0]; p[1] = P[1]; p[2] = P[4]; p[3] = P[5];
		q[0] = Q[0]; q[1] = Q[1]; q[2] = Q[4]; q[3] = Q[5];
		for (int i = 0; i < 8; ++i) { p[4 + i] = P[8 + i]; q[4 + i] = Q[8 + i]; }
	}
};

// In-place inclusive scan: z_l <- sum_{j <= l} M^(l-j) z_j.  M1248 = M^1, M^2, M^4, M^8 (16 floats each,
// wave-uniform); rm = the per-lane matrices above.
template <typename FP>
__device__ __forceinline__ void scan (v2f& z1, v2f& z2, v2f& z3, v2f& z4, const FP M1248, const RowMats& rm)
{
#define MTRW_ROW_STEP(CTRL, d)                                                            \
	{                                                                                     \
		const v2f w1 = dpp0<CTRL, 0xF> (z1), w2 = dpp0<CTRL, 0xF> (z2);                   \
		const v2f w3 = dpp0<CTRL, 0xF> (z3), w4 = dpp0<CTRL, 0xF> (z4);                   \
		const FP M = M1248 + 16 * (d);                                                    \
		MTRW_APPLY (M, w1, w2, w3, w4)                                                    \
	}
	MTRW_ROW_STEP (0x111, 0)
	MTRW_ROW_STEP (0x112, 1)
	MTRW_ROW_STEP (0x114, 2)
	MTRW_ROW_STEP (0x118, 3)
#undef MTRW_ROW_STEP
#define MTRW_BCAST_STEP(CTRL, MASK, m)                                                    \
	{                                                                                     \
		const v2f w1 = dpp0<CTRL, MASK> (z1), w2 = dpp0<CTRL, MASK> (z2);                 \
		const v2f w3 = dpp0<CTRL, MASK> (z3), w4 = dpp0<CTRL, MASK> (z4);                 \
		z1 += (m)[0] * w1;  z1 += (m)[1] * w2;                                            \
		z2 += (m)[2] * w1;  z2 += (m)[3] * w2;                                            \
		z3 += (m)[4] * w1;  z3 += (m)[5] * w2;  z3 += (m)[6] * w3;  z3 += (m)[7] * w4;    \
		z4 += (m)[8] * w1;  z4 += (m)[9] * w2;  z4 += (m)[10] * w3; z4 += (m)[11] * w4;   \
	}
	MTRW_BCAST_STEP (0x142, 0xA, rm.p)      // rows 1, 3 <- complete scan of lane 15 / 47's row
	MTRW_BCAST_STEP (0x143, 0xC, rm.q)      // rows 2, 3 <- complete scan at lane 31
#undef MTRW_BCAST_STEP
}

}  // namespace mtrw
