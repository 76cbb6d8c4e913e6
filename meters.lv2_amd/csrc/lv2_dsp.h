/* lv2_dsp.h — host-side envelope followers shared by more than one plugin translation unit.
 * CPU plumbing: the reference's own arithmetic order, so the port values are
 * bit-identical to the reference build (tests/test_needle_golden.py). */
#ifndef MTR_LV2_DSP_H
#define MTR_LV2_DSP_H

#include <math.h>
#include <string.h>

/* ---- K-meter: RMS through two cascaded one-poles + digital peak with hold and fallback, kmeterdsp.cc:47-160 ---- */
typedef struct { float z1, z2, rms, peak, fall, fsamp, omega; int cnt, fpp, hold, flag; } Kmeter;

static inline void km_reset (Kmeter* k) { k->z1 = k->z2 = k->rms = k->peak = 0; k->cnt = 0; k->flag = 0; }

static inline void km_init (Kmeter* k, float fsamp)          /* kmeterdsp.cc:47-54 */
{
	memset (k, 0, sizeof (*k));
	k->fsamp = fsamp;
	k->hold = (int) (0.5f * fsamp + 0.5f);                   /* samples to hold the peak */
	k->omega = 9.72f / fsamp;                                /* ballistic filter coefficient */
}

static inline void km_process (Kmeter* k, const float* p, int n)
{
	if (k->fpp != n) {                                       /* per-period fallback multiplier: 15 dB/s */
		k->fall = powf (10.0f, -0.05f * 15.0f * ((float) n / k->fsamp));
		k->fpp = n;
	}
	float t = 0;
	float z1 = k->z1 > 50 ? 50 : (k->z1 < 0 ? 0 : k->z1);
	float z2 = k->z2 > 50 ? 50 : (k->z2 < 0 ? 0 : k->z2);
	for (n /= 4; n > 0; --n) {                               /* groups of four; the second filter runs once per group */
		for (int q = 0; q < 4; ++q) {
			float s = *p++;
			s *= s;
			if (t < s) t = s;
			z1 += k->omega * (s - z1);
		}
		z2 += 4 * k->omega * (z1 - z2);
	}
	if (isnan (z1)) z1 = 0;
	if (isnan (z2)) z2 = 0;
	if (!isfinite (t)) t = 0;
	k->z1 = z1 + 1e-20f;
	k->z2 = z2 + 1e-20f;
	const float s = sqrtf (2.0f * z2);
	t = sqrtf (t);
	if (k->flag) { k->rms = s; k->flag = 0; }                /* the value has been read: start a new maximum */
	else if (s > k->rms) k->rms = s;
	if (t >= k->peak) { k->peak = t; k->cnt = k->hold; }
	else if (k->cnt > 0) k->cnt -= k->fpp;
	else { k->peak *= k->fall; k->peak += 1e-10f; }
}

/* Kmeterdsp::read (rms, peak), kmeterdsp.cc:148-153 */
static inline void km_read (Kmeter* k, float* rms, float* peak) { *rms = k->rms; *peak = k->peak; k->flag = 1; }

#endif
