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

/* Programme loudness / range from (possibly summed) histograms — what Ebu_r128_hist::integrate, calc_integ and calc_range
 * (ebumeter/ebu_r128_proc.cc:82-150) give for the same counts, on plain int32[751] arrays.  Bin b stands for the loudness
 * (b - 700) / 10 dB; its power weight is 10^((b % 100) / 100) scaled by ten per hundred bins below the top. */

/* Mean power of the points in bins [first, 750].  The float sum runs over the bins in ascending order and is divided by ten
 * at the end of every hundred — the rounding sequence the reference's table walk has, so the result is its float. */
static float gated_mean_power (const int32_t* h, const float* weight100, int first)
{
	int   points = 0;
	float acc = 0.f;
	for (int base = first - first % 100; base <= 750; base += 100) {
		const int from = base < first ? first : base;
		const int to = base + 99 < 750 ? base + 99 : 750;
		for (int b = from; b <= to; ++b) {
			points += h[b];
			acc += h[b] * weight100[b - base];
		}
		if (to == base + 99) acc /= 10.0f;
	}
	return acc / points;
}

static long points_in (const int32_t* h, int from)
{
	long n = 0;
	for (int b = from; b <= 750; ++b) n += h[b];
	return n;
}

/* First bin of the relative gate: `db_below` under the ungated mean, in hundredths of a decade of power.  The rounding
 * constant is a float for the integrated loudness and a double for the range (:121 writes 0.5f, :141 writes 0.5). */
static int gate_bin (float mean_power, int offset, int half_is_double)
{
	const float c = 100 * log10f (mean_power);
	const int k = (int) floorf (half_is_double ? c + 0.5 : c + 0.5f) + offset;
	return k < 0 ? 0 : k;
}

void mtr_setup_hist_loudness (const int32_t* hm, const int32_t* hs, float* integ, float* integ_thr,
                              float* rmin, float* rmax, float* rthr)
{
	float w[100];
	mtr_setup_bin_power (w);
	*integ = *integ_thr = *rmin = *rmax = *rthr = -200.0f;
	if (hm && points_in (hm, 0) >= 50) {                         /* integrated loudness: the mean above (ungated mean - 10 dB) */
		const float all = gated_mean_power (hm, w, 0);
		*integ_thr = 10 * log10f (all) - 10.0f;
		*integ = 10 * log10f (gated_mean_power (hm, w, gate_bin (all, 600, 0)));
	}
	if (hs && points_in (hs, 0) >= 20) {                         /* loudness range: 10th .. 95th percentile above (mean - 20 dB) */
		const float all = gated_mean_power (hs, w, 0);
		*rthr = 10 * log10f (all) - 20.0f;
		const int gate = gate_bin (all, 500, 1);
		const int n = (int) points_in (hs, gate);
		const float low = 0.10f * n, high = 0.95f * n;
		/* counts are integers and a float holds them exactly up to 2^24 points, so whole-number running counts compared as
		 * floats decide every bin as the reference's float accumulators do */
		long below = 0;                                            /* points in bins [gate, lo) */
		int  lo = gate;
		while ((float) below < low) below += hs[lo++];
		long kept = n;                                             /* points in bins [gate, hi] */
		int  hi = 750;
		while ((float) kept > high) kept -= hs[hi--];
		*rmin = (lo - 701) / 10.0f;
		*rmax = (hi - 699) / 10.0f;
	}
}
