/* mtr_setup.c — one-off host-side set-up math of the engine (plain C, built with gcc,
 * -ffp-contract=off): filter coefficients and tables, computed in the precision and order the
 * reference uses so that the device kernels start from identical constants.
 *
 *   K-weighting coefficients   Ebu_r128_proc::detect_init      ebumeter/ebu_r128_proc.cc:263-293
 *   polyphase FIR table        Resampler_table::Resampler_table zita-resampler/resampler-table.cc:52-75
 *   band-pass sections         bandpass_setup                   src/spectr.c:89-206
 *   histogram bin powers       Ebu_r128_hist::initstat          ebumeter/ebu_r128_proc.cc:54-63
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "mtr_internal.h"

/* High-shelf biquad (stage 1) and RLB high-pass written as a double integrator in the
 * feedback path (stage 2); all float, tanf because the reference TU is C++ (float overload). */
void mtr_setup_kweight (float fs, float* o)
{
	const float q  = 1 / tanf (4712.3890f / fs);
	const float wa = q / 1.12201f, wb = q * 1.12201f;
	const float u  = 1.4085f + 210.0f / fs;
	const float pa = u * wa, pb = wa * wa;
	const float pc = u * wb, pd = wb * wb;
	const float den = 1 + pa + pb;
	float a0 = (1 + pc + pd) / den;
	float a1 = (2 - 2 * pd) / den;
	float a2 = (1 - pc + pd) / den;
	const float b1 = (2 - 2 * pb) / den;
	const float b2 = (1 - pa + pb) / den;
	const float r  = 48.0f / fs;
	float ha = 4.9886075f * r;
	float hb = 6.2298014f * r * r;
	const float hd = 1 + ha + hb;
	ha *= 2 / hd;
	hb *= 4 / hd;
	const float g = 1.004995f / hd;
	a0 *= g; a1 *= g; a2 *= g;
	o[0] = a0; o[1] = a1; o[2] = a2; o[3] = b1; o[4] = b2; o[5] = ha + hb; o[6] = hb;
}

/* State-space form of one K-weighting step, s = [z1 z2 z3 z4]:  s' = A s + B p   */
void mtr_setup_kweight_matrix (const float* k, double* A, double* B)
{
	const double a0 = k[0], a1 = k[1], a2 = k[2], b1 = k[3], b2 = k[4], c3 = k[5], c4 = k[6];
	const double M[16] = {
		-b1,          -b2,          0.0,      0.0,
		1.0,          0.0,          0.0,      0.0,
		a1 - a0 * b1, a2 - a0 * b2, 1.0 - c3, -c4,
		0.0,          0.0,          1.0,      1.0,
	};
	memcpy (A, M, sizeof (M));
	B[0] = 1.0; B[1] = 0.0; B[2] = a0; B[3] = 0.0;
}

static double sinc_pi (double x)
{
	x = fabs (x);
	if (x < 1e-6) return 1.0;
	x *= M_PI;
	return sin (x) / x;
}

static double win3 (double x)
{
	x = fabs (x);
	if (x >= 1.0) return 0.0f;
	x *= M_PI;
	return 0.384 + 0.500 * cos (x) + 0.116 * cos (2 * x);
}

/* np + 1 = 5 rows of hl = 24 taps, relative cut-off 1.0; row j is the fractional delay j/4,
 * stored with the tap for distance i at index hl-1-i. */
void mtr_setup_fir_table (float* out)
{
	const unsigned hl = 24, np = 4;
	const double fr = 1.0;
	for (unsigned j = 0; j <= np; j++) {
		double t = (double) j / (double) np;
		for (unsigned i = 0; i < hl; i++) {
			out[j * hl + (hl - 1 - i)] = (float) (fr * sinc_pi (t * fr) * win3 (t / hl));
			t += 1;
		}
	}
}

void mtr_setup_bin_power (float* out)
{
	for (int j = 0; j < 100; ++j) out[j] = powf (10.0f, j / 100.0f);
}

/* 6th-order Butterworth band-pass -> 6 biquads by the complex bilinear transform, normalised to
 * unity gain at the geometric band centre through section 0's numerator.
 * out36 = [section][a0 a1 a2 b0 b1 b2]. */
void mtr_setup_band (double rate, uint32_t band, double* out)
{
	const int order = 6;
	const double f_m  = pow (2, ((int) band - 16) / 3.) * 1000;
	const double bw   = f_m * pow (2, 1. / 6.) - f_m * pow (2, -1. / 6.);
	const double wc = 2. * M_PI * f_m / rate;
	const double ww = 2. * M_PI * bw / rate;
	double lo = wc - (ww / 2.), hi = wc + (ww / 2.);
	if (hi > M_PI - 1e-9) hi = M_PI - 1e-9;
	if (lo < 1e-9) lo = 1e-9;
	hi *= .5; lo *= .5;

	const double ca  = cos (hi + lo) / cos (hi - lo);
	const double cb  = 1. / tan (hi - lo);
	const double w0  = 2. * atan (sqrt (tan (hi) * tan (lo)));
	const double ca2 = ca * ca, cb2 = cb * cb, ab2 = 2. * ca * cb;
	double (*W)[6] = (double (*)[6]) out;

	for (int i = 0; i < order / 2; ++i) {
		const double th = M_PI_2 + (2 * i + 1) * M_PI / (2. * (double) order);
		const double complex p = CMPLX (cos (th), sin (th));
		const double complex c = (1. + p) / (1. - p);
		const double complex d = 2 * (cb - 1) * c + 2 * (1 + cb);
		double complex v = (4 * (cb2 * (ca2 - 1) + 1)) * c;
		v += 8 * (cb2 * (ca2 - 1) - 1);
		v *= c;
		v += 4 * (cb2 * (ca2 - 1) + 1);
		v  = csqrt (v);
		const double complex nv = v * -1.;
		const double complex root[2] = {
			CMPLX (ab2 + creal (nv) + ab2 * creal (c), cimag (nv) + ab2 * cimag (c)) / d,
			CMPLX (ab2 + creal (v)  + ab2 * creal (c), cimag (v)  + ab2 * cimag (c)) / d,
		};
		for (int k = 0; k < 2; ++k) {
			double* w = W[2 * i + k];
			w[0] = 1.;
			w[1] = -2 * creal (root[k]);
			w[2] = creal (root[k]) * creal (root[k]) + cimag (root[k]) * cimag (root[k]);
			w[3] = 1.;
			w[4] = k ? -2. : 2.;
			w[5] = 1.;
		}
	}

	const double cw = cos (-w0), sw = sin (-w0), cw2 = cos (-2. * w0), sw2 = sin (-2. * w0);
	double complex num = 1, den = 1;
	for (int i = 0; i < order; ++i) {
		num *= CMPLX ((1 + W[i][4] * cw) + cw2, (W[i][4] * sw) + sw2);
		den *= CMPLX ((1 + W[i][1] * cw) + W[i][2] * cw2, (W[i][1] * sw) + W[i][2] * sw2);
	}
	const double g = creal (den / num);
	W[0][3] *= g; W[0][4] *= g; W[0][5] *= g;
}

/* Programme loudness / range from (possibly summed) histograms: Ebu_r128_hist::integrate,
 * calc_integ, calc_range (ebumeter/ebu_r128_proc.cc:82-150) on plain int32[751] arrays. */
static float hl_integrate (const int32_t* h, const float* bp, int i)
{
	int   j = i % 100, n = 0;
	float s = 0;
	while (i <= 750) {
		const int k = h[i++];
		n += k;
		s += k * bp[j++];
		if (j == 100) { j = 0; s /= 10.0f; }
	}
	return s / n;
}

void mtr_setup_hist_loudness (const int32_t* hm, const int32_t* hs, float* integ, float* integ_thr,
                              float* rmin, float* rmax, float* rthr)
{
	float bp[100];
	mtr_setup_bin_power (bp);
	*integ = *integ_thr = *rmin = *rmax = *rthr = -200.0f;
	if (hm) {
		long cnt = 0;
		for (int i = 0; i <= 750; ++i) cnt += hm[i];
		if (cnt >= 50) {
			float s = hl_integrate (hm, bp, 0);
			*integ_thr = 10 * log10f (s) - 10.0f;
			int k = (int) (floorf (100 * log10f (s) + 0.5f)) + 600;
			if (k < 0) k = 0;
			s = hl_integrate (hm, bp, k);
			*integ = 10 * log10f (s);
		}
	}
	if (hs) {
		long cnt = 0;
		for (int i = 0; i <= 750; ++i) cnt += hs[i];
		if (cnt >= 20) {
			float s = hl_integrate (hs, bp, 0);
			*rthr = 10 * log10f (s) - 20.0f;
			int k = (int) (floorf (100 * log10f (s) + 0.5)) + 500;
			if (k < 0) k = 0;
			int i, j, n = 0;
			for (i = k; i <= 750; i++) n += hs[i];
			const float a = 0.10f * n, b = 0.95f * n;
			for (i = k, s = 0; s < a; i++) s += hs[i];
			for (j = 750, s = n; s > b; j--) s -= hs[j];
			*rmin = (i - 701) / 10.0f;
			*rmax = (j - 699) / 10.0f;
		}
	}
}
