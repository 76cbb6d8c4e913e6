// mtr_mfma_fir.h — float -> IEEE half conversion for the host-side tables of the matrix-pipe interpolator
// (mtr_mfma16_fir.h).  Round 1's single-f16-tap interpolator (layout 5), which this header used to carry, is gone: it was
// narrower than the reference's f32 chain.
#pragma once
#include <stdint.h>
#include <string.h>

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
