/* mtr_oracle.c — CPU restatement of the meters.lv2 DSP hot path (see mtr_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY — never linked into the product (meters.lv2_amd/).
 *
 * Written from the equations of SURVEY.md Appendix A and the reference sources
 * cited per function; operation order and precision follow the reference so
 * that, with -ffp-contract=off and no FMA ISA, results are bit-identical to the
 * reference objects on x86-64 (checked in tests/test_oracle_vs_ref.py).
 */
#define _GNU_SOURCE
#include "mtr_oracle.h"

#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================
 * EBU R128
 * ====================================================================== */

/* channel weights L R C Ls Rs — ebu_r128_proc.cc:29 */
static const float mo_chan_gain[MO_MAXCH] = { 1.0f, 1.0f, 1.0f, 1.41f, 1.41f };

/* 10^(j/100), j = 0..99 — ebu_r128_proc.cc:54-63 (powf, float argument) */
static float mo_bin_power[100];
static int   mo_bin_power_ready = 0;

static void mo_bin_power_init (void)
{
	if (mo_bin_power_ready) return;
	for (int j = 0; j < 100; ++j) mo_bin_power[j] = powf (10.0f, j / 100.0f);
	mo_bin_power_ready = 1;
}

/* ebu_r128_proc.cc:263-293.  The reference is C++: `tan (float)` resolves to the
 * float overload, hence tanf here; everything stays in float. */
void mo_kw_init (mo_kw_coef* k, float fsamp)
{
	float r  = 1 / tanf (4712.3890f / fsamp);
	float w1 = r / 1.12201f;
	float w2 = r * 1.12201f;
	float u  = 1.4085f + 210.0f / fsamp;
	float a  = u * w1, b = w1 * w1;
	float c  = u * w2, d = w2 * w2;
	r = 1 + a + b;
	k->a0 = (1 + c + d) / r;
	k->a1 = (2 - 2 * d) / r;
	k->a2 = (1 - c + d) / r;
	k->b1 = (2 - 2 * b) / r;
	k->b2 = (1 - a + b) / r;
	/* RLB high-pass as a double integrator in the feedback path */
	r = 48.0f / fsamp;
	a = 4.9886075f * r;
	b = 6.2298014f * r * r;
	r = 1 + a + b;
	a *= 2 / r;
	b *= 4 / r;
	k->c3 = a + b;
	k->c4 = b;
	r = 1.004995f / r;
	k->a0 *= r;
	k->a1 *= r;
	k->a2 *= r;
}

void mo_hist_reset (mo_hist* h)
{
	memset (h->histc, 0, sizeof (h->histc));
	h->count = 0;
	h->error = 0;
}

/* ebu_r128_proc.cc:66-79 — 0.1 dB bins, bin 0 = -70.0, bin 700 = 0.0, clamp high */
void mo_hist_addpoint (mo_hist* h, float v)
{
	int k = (int) floorf (10 * v + 700.5f);
	if (k < 0) return;
	if (k > 750) { k = 750; h->error++; }
	h->histc[k]++;
	h->count++;
}

/* ebu_r128_proc.cc:82-102 — power-domain mean from bin `i` upwards, with the
 * running /10 renormalisation each time the sub-index wraps */
float mo_hist_integrate (const mo_hist* h, int i)
{
	int   j = i % 100, n = 0;
	float s = 0;
	mo_bin_power_init ();
	while (i <= 750) {
		int k = h->histc[i++];
		n += k;
		s += k * mo_bin_power[j++];
		if (j == 100) { j = 0; s /= 10.0f; }
	}
	return s / n;
}

/* ebu_r128_proc.cc:105-125 */
void mo_hist_calc_integ (const mo_hist* h, float* vi, float* th)
{
	if (h->count < 50) { *vi = -200.0f; return; }
	float s = mo_hist_integrate (h, 0);
	if (th) *th = 10 * log10f (s) - 10.0f;
	int k = (int)(floorf (100 * log10f (s) + 0.5f)) + 600;
	if (k < 0) k = 0;
	s = mo_hist_integrate (h, k);
	*vi = 10 * log10f (s);
}

/* ebu_r128_proc.cc:128-150 — note the double 0.5 at :141 */
void mo_hist_calc_range (const mo_hist* h, float* v0, float* v1, float* th)
{
	int   i, j, k, n;
	float a, b, s;
	if (h->count < 20) { *v0 = -200.0f; *v1 = -200.0f; return; }
	s = mo_hist_integrate (h, 0);
	if (th) *th = 10 * log10f (s) - 20.0f;
	k = (int)(floorf (100 * log10f (s) + 0.5)) + 500;
	if (k < 0) k = 0;
	for (i = k, n = 0; i <= 750; i++) n += h->histc[i];
	a = 0.10f * n;
	b = 0.95f * n;
	for (i = k,   s = 0; s < a; i++) s += h->histc[i];
	for (j = 750, s = n; s > b; j--) s -= h->histc[j];
	*v0 = (i - 701) / 10.0f;
	*v1 = (j - 699) / 10.0f;
}

void mo_ebu_integr_reset (mo_ebu* e)
{
	mo_hist_reset (&e->hist_M);
	mo_hist_reset (&e->hist_S);
	e->maxloudn_M = e->maxloudn_S = -200.0f;
	e->integrated = e->integ_thr  = -200.0f;
	e->range_min  = e->range_max  = e->range_thr = -200.0f;
	e->div1 = e->div2 = 0;
}

void mo_ebu_reset (mo_ebu* e)
{
	e->integr = 0;
	e->frcnt  = e->fragm;
	e->frpwr  = 1e-30f;
	e->wrind  = 0;
	e->div1 = e->div2 = 0;
	e->loudness_M = e->loudness_S = -200.0f;
	memset (e->power, 0, sizeof (e->power));
	mo_ebu_integr_reset (e);
	memset (e->z, 0, sizeof (e->z));
}

void mo_ebu_init (mo_ebu* e, int nchan, float fsamp)
{
	memset (e, 0, sizeof (*e));
	mo_bin_power_init ();
	e->nchan = nchan;
	e->fsamp = fsamp;
	e->fragm = (int) fsamp / 20;
	mo_kw_init (&e->k, fsamp);
	mo_ebu_reset (e);
}

void mo_ebu_integr_start (mo_ebu* e) { e->integr = 1; }
void mo_ebu_integr_pause (mo_ebu* e) { e->integr = 0; }

/* ebu_r128_proc.cc:251-260 */
static float mo_ebu_addfrags (const mo_ebu* e, int nfrag)
{
	float s = 0;
	int   k = (e->wrind - nfrag) & 63;
	for (int i = 0; i < nfrag; i++) s += e->power[(i + k) & 63];
	return -0.6976f + 10 * log10f (s / nfrag);
}

/* ebu_r128_proc.cc:302-337 — K-weighting recurrence, SURVEY.md A.1 */
float mo_ebu_detect (mo_ebu* e, int nfram, const float* const* input)
{
	const mo_kw_coef c = e->k;
	float si = 0;
	for (int ch = 0; ch < e->nchan; ch++) {
		float z1 = e->z[ch][0], z2 = e->z[ch][1], z3 = e->z[ch][2], z4 = e->z[ch][3];
		const float* p = input[ch];
		float sj = 0;
		for (int j = 0; j < nfram; j++) {
			float x = p[j] - c.b1 * z1 - c.b2 * z2 + 1e-15f;
			float y = c.a0 * x + c.a1 * z1 + c.a2 * z2 - c.c3 * z3 - c.c4 * z4;
			z2 = z1;
			z1 = x;
			z4 += z3;
			z3 += y;
			sj += y * y;
		}
		if (e->nchan == 1) si = 2 * sj;
		else               si += mo_chan_gain[ch] * sj;
		e->z[ch][0] = isfinite (z1) ? z1 : 0;
		e->z[ch][1] = isfinite (z2) ? z2 : 0;
		e->z[ch][2] = isfinite (z3) ? z3 : 0;
		e->z[ch][3] = isfinite (z4) ? z4 : 0;
	}
	return si;
}

/* ebu_r128_proc.cc:207-248 */
void mo_ebu_process (mo_ebu* e, int nfram, const float* const* input)
{
	const float* ipp[MO_MAXCH];
	for (int i = 0; i < e->nchan; i++) ipp[i] = input[i];
	while (nfram) {
		int k = (e->frcnt < nfram) ? e->frcnt : nfram;
		e->frpwr += mo_ebu_detect (e, k, ipp);
		e->frcnt -= k;
		if (e->frcnt == 0) {
			e->power[e->wrind++] = e->frpwr / e->fragm;
			e->frcnt  = e->fragm;
			e->frpwr  = 1e-30f;
			e->wrind &= 63;
			e->loudness_M = mo_ebu_addfrags (e, 8);
			e->loudness_S = mo_ebu_addfrags (e, 60);
			if (!isfinite (e->loudness_M) || e->loudness_M < -200.f) e->loudness_M = -200.0f;
			if (!isfinite (e->loudness_S) || e->loudness_S < -200.f) e->loudness_S = -200.0f;
			if (e->loudness_M > e->maxloudn_M) e->maxloudn_M = e->loudness_M;
			if (e->loudness_S > e->maxloudn_S) e->maxloudn_S = e->loudness_S;
			if (e->integr) {
				if (++e->div1 == 2) {
					mo_hist_addpoint (&e->hist_M, e->loudness_M);
					e->div1 = 0;
				}
				if (++e->div2 == 10) {
					mo_hist_addpoint (&e->hist_S, e->loudness_S);
					e->div2 = 0;
					mo_hist_calc_integ (&e->hist_M, &e->integrated, &e->integ_thr);
					mo_hist_calc_range (&e->hist_S, &e->range_min, &e->range_max, &e->range_thr);
				}
			}
		}
		for (int i = 0; i < e->nchan; i++) ipp[i] += k;
		nfram -= k;
	}
}

/* ======================================================================
 * True peak: polyphase table + streaming FIR + ballistics
 * ====================================================================== */

static float mo_tp_ctab[(MO_TP_NP + 1) * MO_TP_HL];
static int   mo_tp_ctab_ready = 0;

/* resampler-table.cc:29-44 */
static double mo_sinc (double x)
{
	x = fabs (x);
	if (x < 1e-6) return 1.0;
	x *= M_PI;
	return sin (x) / x;
}
static double mo_wind (double x)
{
	x = fabs (x);
	if (x >= 1.0) return 0.0f;
	x *= M_PI;
	return 0.384 + 0.500 * cos (x) + 0.116 * cos (2 * x);
}

/* resampler-table.cc:52-75 with fr = 1.0, hl = 24, np = 4 (truepeakdsp.cc:150);
 * row j holds phase j/np, stored newest-tap-last (index hl-1-i). */
const float* mo_tp_table (void)
{
	if (!mo_tp_ctab_ready) {
		const double fr = 1.0;
		float* p = mo_tp_ctab;
		for (unsigned j = 0; j <= MO_TP_NP; j++) {
			double t = (double) j / (double) MO_TP_NP;
			for (unsigned i = 0; i < MO_TP_HL; i++) {
				p[MO_TP_HL - i - 1] = (float)(fr * mo_sinc (t * fr) * mo_wind (t / MO_TP_HL));
				t += 1;
			}
			p += MO_TP_HL;
		}
		mo_tp_ctab_ready = 1;
	}
	return mo_tp_ctab;
}

/* truepeakdsp.cc:148-169.  The 8192-zero pre-roll leaves the resampler with an
 * all-zero 48-sample window, nread = 1, phase = 0. */
void mo_tp_init (mo_tp* t, float fsamp)
{
	memset (t, 0, sizeof (*t));
	(void) mo_tp_table ();
	t->res = 1;
	t->z1 = t->z2 = .0f;
	t->w1 = 4000.0f / fsamp / 4.0;
	t->w2 = 17200.0f / fsamp / 4.0;
	t->w3 = 1.0f - 7.0f / fsamp / 4.0;
	t->g  = 0.502f;
}

/* resampler.cc:211-235 — one input in, four outputs out.  win[0] is the oldest
 * of the 48 samples, win[47] the newest (the one just read). */
void mo_tp_resample (mo_tp* t, const float* in, int n, float* out)
{
	const float* ctab = mo_tp_table ();
	float* w = t->win;
	for (int k = 0; k < n; k++) {
		memmove (w, w + 1, (2 * MO_TP_HL - 1) * sizeof (float));
		w[2 * MO_TP_HL - 1] = in[k];
		for (unsigned ph = 0; ph < MO_TP_NP; ph++) {
			const float* c1 = ctab + MO_TP_HL * ph;
			const float* c2 = ctab + MO_TP_HL * (MO_TP_NP - ph);
			float s = 1e-20f;
			for (unsigned i = 0; i < MO_TP_HL; i++) {
				s += w[i] * c1[i] + w[2 * MO_TP_HL - 1 - i] * c2[i];
			}
			*out++ = s - 1e-20f;
		}
	}
}

#define MO_TP_MAXBLK 8192   /* truepeakdsp.cc:44,103 */

/* truepeakdsp.cc:101-124 */
void mo_tp_process_max (mo_tp* t, const float* in, int n)
{
	static __thread float buf[4 * MO_TP_MAXBLK];
	mo_tp_resample (t, in, n, buf);
	float m = t->res ? 0 : t->m;
	const float* b = buf;
	while (n--) {
		for (int q = 0; q < 4; q++) {
			float v = fabsf (*b++);
			if (v > m) m = v;
		}
	}
	t->m = m;
}

/* truepeakdsp.cc:41-99 */
void mo_tp_process (mo_tp* t, const float* in, int n)
{
	static __thread float buf[4 * MO_TP_MAXBLK];
	mo_tp_resample (t, in, n, buf);
	float m  = t->res ? 0 : t->m;
	float p  = t->res ? 0 : t->p;
	float z1 = t->z1 > 20 ? 20 : (t->z1 < 0 ? 0 : t->z1);
	float z2 = t->z2 > 20 ? 20 : (t->z2 < 0 ? 0 : t->z2);
	const float* b = buf;
	while (n--) {
		z1 *= t->w3;
		z2 *= t->w3;
		for (int q = 0; q < 4; q++) {
			float v = fabsf (*b++);
			if (v > z1) z1 += t->w1 * (v - z1);
			if (v > z2) z2 += t->w2 * (v - z2);
			if (v > p)  p = v;
		}
		float v = z1 + z2;
		if (v > m) m = v;
	}
	t->z1 = z1 + 1e-20f;
	t->z2 = z2 + 1e-20f;
	m *= t->g;
	if (t->res) {
		t->m = m;
		t->p = p;
		t->res = 0;
	} else {
		if (m > t->m) t->m = m;
		if (p > t->p) t->p = p;
	}
}

float mo_tp_read (mo_tp* t) { t->res = 1; return t->m; }
void  mo_tp_read2 (mo_tp* t, float* m, float* p) { t->res = 1; *m = t->m; *p = t->p; }
void  mo_tp_reset (mo_tp* t) { t->res = 1; t->m = 0; t->p = 0; }

/* ======================================================================
 * 30-band bank
 * ====================================================================== */

enum { W_a0 = 0, W_a1, W_a2, W_b0, W_b1, W_b2 };   /* spectr.c:51 */

/* spectr.c:89-206 — order-`order` Butterworth band-pass as `order` biquads via
 * the complex bilinear transform; unity gain at the geometric band centre by
 * scaling section 0's numerator with Re(cb/ch). */
void mo_band_setup (mo_band* fb, double rate, double freq, double band, int order)
{
	fb->stages = (uint32_t) order;
	for (uint32_t i = 0; i < fb->stages; ++i) fb->f[i].z[0] = fb->f[i].z[1] = 0;

	const double wc = 2. * M_PI * freq / rate;
	const double ww = 2. * M_PI * band / rate;
	double wl = wc - (ww / 2.);
	double wu = wc + (ww / 2.);
	if (wu > M_PI - 1e-9) wu = M_PI - 1e-9;   /* :113-122 (stderr warning omitted) */
	if (wl < 1e-9)        wl = 1e-9;          /* :123-131 */
	wu *= .5; wl *= .5;

	const double c_a = cos (wu + wl) / cos (wu - wl);
	const double c_b = 1. / tan (wu - wl);
	const double w   = 2. * atan (sqrt (tan (wu) * tan (wl)));
	const double c_a2 = c_a * c_a;
	const double c_b2 = c_b * c_b;
	const double ab_2 = 2. * c_a * c_b;

	for (uint32_t i = 0; i < fb->stages / 2; ++i) {
		const double omega = M_PI_2 + (2 * i + 1) * M_PI / (2. * (double) fb->stages);
		const double complex p = CMPLX (cos (omega), sin (omega));
		const double complex c = (1. + p) / (1. - p);
		const double complex d = 2 * (c_b - 1) * c + 2 * (1 + c_b);
		double complex v;
		v  = (4 * (c_b2 * (c_a2 - 1) + 1)) * c;
		v += 8 * (c_b2 * (c_a2 - 1) - 1);
		v *= c;
		v += 4 * (c_b2 * (c_a2 - 1) + 1);
		v  = csqrt (v);

		const double complex vm = v * -1.;
		const double complex u0 = CMPLX (ab_2 + creal (vm) + ab_2 * creal (c), cimag (vm) + ab_2 * cimag (c));
		const double complex u1 = CMPLX (ab_2 + creal (v)  + ab_2 * creal (c), cimag (v)  + ab_2 * cimag (c));

		const double complex P[2] = { u0 / d, u1 / d };
		for (int q = 0; q < 2; ++q) {
			mo_biquad* f = &fb->f[2 * i + q];
			f->W[W_a0] = 1.;
			f->W[W_a1] = -2 * creal (P[q]);
			f->W[W_a2] = creal (P[q]) * creal (P[q]) + cimag (P[q]) * cimag (P[q]);
			f->W[W_b0] = 1.;
			f->W[W_b1] = q ? -2. : 2.;
			f->W[W_b2] = 1.;
		}
	}

	/* normalise at e^{-jw} — spectr.c:173-190 */
	const double cos_w = cos (-w), sin_w = sin (-w);
	const double cos_w2 = cos (-2. * w), sin_w2 = sin (-2. * w);
	double complex ch = 1, cb = 1;
	for (uint32_t i = 0; i < fb->stages; ++i) {
		const mo_biquad* f = &fb->f[i];
		ch *= CMPLX ((1 + f->W[W_b1] * cos_w) + cos_w2, (f->W[W_b1] * sin_w) + sin_w2);
		cb *= CMPLX ((1 + f->W[W_a1] * cos_w) + f->W[W_a2] * cos_w2,
		             (f->W[W_a1] * sin_w) + f->W[W_a2] * sin_w2);
	}
	const double complex scale = cb / ch;
	fb->f[0].W[W_b0] *= creal (scale);
	fb->f[0].W[W_b1] *= creal (scale);
	fb->f[0].W[W_b2] *= creal (scale);
}

/* spectr.c:68-87 — TDF-II cascade in double with the +/-1e-12 anti-denormal toggle */
float mo_band_process (mo_band* fb, float in)
{
	fb->ac = !fb->ac;
	double out = in + (fb->ac ? 1e-12 : -1e-12);
	for (uint32_t i = 0; i < fb->stages; ++i) {
		mo_biquad* f = &fb->f[i];
		const double y = f->W[W_b0] * out + f->z[0];
		f->z[0] = f->W[W_b1] * out - f->W[W_a1] * y + f->z[1];
		f->z[1] = f->W[W_b2] * out - f->W[W_a2] * y;
		out = y;
	}
	return out;
}

/* spectrumlv2.c:73-121 */
void mo_spectr_init (mo_spectr* s, uint32_t nchannels, double rate)
{
	memset (s, 0, sizeof (*s));
	s->nchannels = nchannels;
	s->rate = rate;
	s->omega = 1.0f - expf (-2.0 * M_PI * 1.0 / rate);
	const double f_r = 1000, b = 3;
	const double f1f = pow (2, -1. / (2. * b));
	const double f2f = pow (2,  1. / (2. * b));
	for (uint32_t i = 0; i < MO_NBANDS; ++i) {
		const int    x   = (int) i - 16;
		const double f_m = pow (2, x / b) * f_r;
		const double bw  = f_m * f2f - f_m * f1f;
		mo_band_setup (&s->flt[i], rate, f_m, bw, 6);
	}
}

/* spectrumlv2.c:170-177 */
void mo_spectr_set_speed (mo_spectr* s, float v)
{
	if (v < 0.01) v = 0.01;
	if (v > 15.0) v = 15.0;
	s->omega = 1.0f - expf (-2.0 * M_PI * v / s->rate);
}

void mo_spectr_reset_peak (mo_spectr* s)
{
	for (int i = 0; i < MO_NBANDS; ++i) s->max_f[i] = 0;
}

/* spectrumlv2.c:208-248 */
void mo_spectr_run (mo_spectr* s, const float* inL, const float* inR, uint32_t n)
{
	float val_f[MO_NBANDS], max_f[MO_NBANDS];
	const float omega = s->omega;
	const int stereo = s->nchannels == 2;
	memcpy (val_f, s->val_f, sizeof (val_f));
	memcpy (max_f, s->max_f, sizeof (max_f));

	for (uint32_t j = 0; j < n; ++j) {
		float in;
		if (stereo) {
			const float L = inL[j], R = inR[j];
			in = (L + R) / 2.0f;
		} else {
			in = inL[j];
		}
		for (int i = 0; i < MO_NBANDS; ++i) {
			const float v = mo_band_process (&s->flt[i], in);
			const float q = v * v;
			val_f[i] += omega * (q - val_f[i]);
			if (val_f[i] > max_f[i]) max_f[i] = val_f[i];
		}
	}

	for (int i = 0; i < MO_NBANDS; ++i) {
		if (!isfinite (val_f[i])) val_f[i] = 0;
		if (!isfinite (max_f[i])) max_f[i] = 0;
		for (uint32_t j = 0; j < s->flt[i].stages; ++j) {
			if (!isfinite (s->flt[i].f[j].z[0])) s->flt[i].f[j].z[0] = 0;
			if (!isfinite (s->flt[i].f[j].z[1])) s->flt[i].f[j].z[1] = 0;
		}
		s->val_f[i] = val_f[i] + 1e-20f;
		s->max_f[i] = max_f[i];
		const float vs = sqrtf (2. * val_f[i]);
		const float mx = sqrtf (2. * max_f[i]);
		s->spec_db[i] = vs > .00001f ? 20.0 * log10f (vs) : -100.0;
		s->max_db[i]  = mx > .00001f ? 20.0 * log10f (mx) : -100.0;
	}
}

/* ======================================================================
 * VU
 * ====================================================================== */

void mo_vu_init (mo_vu* v, float fsamp)
{
	v->z1 = v->z2 = v->m = 0;
	v->res = 1;
	v->w = 11.1f / fsamp;
	v->g = 1.5f * 1.571f;
}

/* vumeterdsp.cc:45-73 — groups of four; n mod 4 trailing samples are dropped */
void mo_vu_process (mo_vu* v, const float* p, int n)
{
	float z1 = v->z1 > 20 ? 20 : (v->z1 < -20 ? -20 : v->z1);
	float z2 = v->z2 > 20 ? 20 : (v->z2 < -20 ? -20 : v->z2);
	float m  = v->res ? 0 : v->m;
	const float w = v->w;
	v->res = 0;
	n /= 4;
	while (n--) {
		const float t2 = z2 / 2;
		for (int q = 0; q < 4; q++) {
			const float t1 = fabsf (*p++) - t2;
			z1 += w * (t1 - z1);
		}
		z2 += 4 * w * (z1 - z2);
		if (z2 > m) m = z2;
	}
	if (!isfinite (z1)) { v->z1 = 0; m = INFINITY; } else v->z1 = z1;
	if (!isfinite (z2)) { v->z2 = 0; m = INFINITY; } else v->z2 = z2 + 1e-10f;
	v->m = m;
}

float mo_vu_read (mo_vu* v) { v->res = 1; return v->g * v->m; }

/* ======================================================================
 * PPM (IEC 268-10 type I: DIN / Nordic; type II: BBC / EBU), M/S PPM,
 * stereo correlation, K-meter
 * ====================================================================== */

static void ppm_zero (mo_ppm* p) { p->z1 = p->z2 = p->m = 0; p->res = 1; }

void mo_ppm_init_iec1 (mo_ppm* p, float fsamp)
{
	ppm_zero (p);
	p->w1 = 450.0f / fsamp;
	p->w2 = 1300.0f / fsamp;
	p->w3 = 1.0f - 5.4f / fsamp;
	p->g  = 0.5108f;
}

void mo_ppm_init_iec2 (mo_ppm* p, float fsamp)
{
	ppm_zero (p);
	p->w1 = 200.0f / fsamp;
	p->w2 = 860.0f / fsamp;
	p->w3 = 1.0f - 4.0f / fsamp;
	p->g  = 0.5141f;
}

/* one group of four rectified samples t[0..3] through the two attack filters; the decay is applied
 * once per group (iec1ppmdsp.cc:58-74) */
#define PPM_GROUP(T0, T1, T2, T3)                                  \
	{                                                              \
		float t;                                                   \
		z1 *= w3; z2 *= w3;                                        \
		t = (T0); if (t > z1) z1 += w1 * (t - z1); if (t > z2) z2 += w2 * (t - z2); \
		t = (T1); if (t > z1) z1 += w1 * (t - z1); if (t > z2) z2 += w2 * (t - z2); \
		t = (T2); if (t > z1) z1 += w1 * (t - z1); if (t > z2) z2 += w2 * (t - z2); \
		t = (T3); if (t > z1) z1 += w1 * (t - z1); if (t > z2) z2 += w2 * (t - z2); \
		t = z1 + z2;                                               \
		if (t > m) m = t;                                          \
	}

void mo_ppm_process (mo_ppm* p, const float* in, int n)
{
	float z1 = p->z1 > 20 ? 20 : (p->z1 < 0 ? 0 : p->z1);
	float z2 = p->z2 > 20 ? 20 : (p->z2 < 0 ? 0 : p->z2);
	float m  = p->res ? 0 : p->m;
	const float w1 = p->w1, w2 = p->w2, w3 = p->w3;
	p->res = 0;
	n /= 4;
	while (n--) {
		PPM_GROUP (fabsf (in[0]), fabsf (in[1]), fabsf (in[2]), fabsf (in[3]))
		in += 4;
	}
	p->z1 = z1 + 1e-10f;
	p->z2 = z2 + 1e-10f;
	p->m = m;
}

float mo_ppm_read (mo_ppm* p) { p->res = 1; return p->g * p->m; }

void mo_msppm_set_gain (mo_msppm* p, float db)
{
	if (p->db == db) return;
	p->db = db;
	p->mv = powf (10, .05 * db);
}

void mo_msppm_init (mo_msppm* p, float fsamp, float mdb)
{
	mo_ppm_init_iec2 (&p->p, fsamp);           /* the same constants, msppmdsp.cc:131-137 */
	p->db = 0;
	p->mv = 1.0f;
	mo_msppm_set_gain (p, mdb);
}

void mo_msppm_process (mo_msppm* q, const float* l, const float* r, int n, int side)
{
	mo_ppm* p = &q->p;
	float z1 = p->z1 > 20 ? 20 : (p->z1 < 0 ? 0 : p->z1);
	float z2 = p->z2 > 20 ? 20 : (p->z2 < 0 ? 0 : p->z2);
	float m  = p->res ? 0 : p->m;
	const float w1 = p->w1, w2 = p->w2, w3 = p->w3, mv = q->mv;
	p->res = 0;
	n /= 4;
	while (n--) {
		if (side) {
			PPM_GROUP (mv * fabsf (l[0] - r[0]), mv * fabsf (l[1] - r[1]), mv * fabsf (l[2] - r[2]), mv * fabsf (l[3] - r[3]))
		} else {
			PPM_GROUP (mv * fabsf (l[0] + r[0]), mv * fabsf (l[1] + r[1]), mv * fabsf (l[2] + r[2]), mv * fabsf (l[3] + r[3]))
		}
		l += 4; r += 4;
	}
	p->z1 = z1 + 1e-10f;
	p->z2 = z2 + 1e-10f;
	p->m = m;
}

float mo_msppm_read (mo_msppm* p) { return mo_ppm_read (&p->p); }

void mo_stcorr_init (mo_stcorr* c, int fsamp, float flp, float tcf)
{
	c->zl = c->zr = c->zlr = c->zll = c->zrr = 0;
	c->w1 = 6.28f * flp / fsamp;
	c->w2 = 1 / (tcf * fsamp);
}

void mo_stcorr_process (mo_stcorr* c, const float* pl, const float* pr, int n)
{
	float zl = c->zl, zr = c->zr, zlr = c->zlr, zll = c->zll, zrr = c->zrr;
	const float w1 = c->w1, w2 = c->w2;
	while (n--) {
		zl += w1 * (*pl++ - zl) + 1e-20f;
		zr += w1 * (*pr++ - zr) + 1e-20f;
		zlr += w2 * (zl * zr - zlr);
		zll += w2 * (zl * zl - zll);
		zrr += w2 * (zr * zr - zrr);
	}
	if (!isfinite (zl)) zl = 0;
	if (!isfinite (zr)) zr = 0;
	if (!isfinite (zlr)) zlr = 0;
	if (!isfinite (zll)) zll = 0;
	if (!isfinite (zrr)) zrr = 0;
	c->zl = zl;
	c->zr = zr;
	c->zlr = zlr + 1e-10f;
	c->zll = zll + 1e-10f;
	c->zrr = zrr + 1e-10f;
}

float mo_stcorr_read (const mo_stcorr* c) { return c->zlr / sqrtf (c->zll * c->zrr + 1e-10f); }

void mo_kmeter_reset (mo_kmeter* k)
{
	k->z1 = k->z2 = k->rms = k->peak = .0f;
	k->cnt = 0;
	k->flag = 0;
}

void mo_kmeter_init (mo_kmeter* k, float fsamp)
{
	mo_kmeter_reset (k);
	k->fpp = 0;
	k->fall = 0;
	k->fsamp = fsamp;
	k->hold = (int) (0.5f * fsamp + 0.5f);       /* samples to hold the peak */
	k->omega = 9.72f / fsamp;                    /* ballistic filter coefficient */
}

void mo_kmeter_process (mo_kmeter* k, const float* p, int n)
{
	if (k->fpp != n) {
		const float fall = 15.0f;
		const float tme = (float) n / k->fsamp;  /* period time in seconds */
		k->fall = powf (10.0f, -0.05f * fall * tme);
		k->fpp = n;
	}
	float s, t = 0;
	float z1 = k->z1 > 50 ? 50 : (k->z1 < 0 ? 0 : k->z1);
	float z2 = k->z2 > 50 ? 50 : (k->z2 < 0 ? 0 : k->z2);
	const float omega = k->omega;
	n /= 4;
	while (n--) {
		for (int q = 0; q < 4; ++q) {
			s = *p++;
			s *= s;
			if (t < s) t = s;                    /* digital peak */
			z1 += omega * (s - z1);              /* first filter */
		}
		z2 += 4 * omega * (z1 - z2);             /* second filter, every 4th sample */
	}
	if (isnan (z1)) z1 = 0;
	if (isnan (z2)) z2 = 0;
	if (!isfinite (t)) t = 0;
	k->z1 = z1 + 1e-20f;
	k->z2 = z2 + 1e-20f;
	s = sqrtf (2.0f * z2);
	t = sqrtf (t);
	if (k->flag) {                               /* the display has read the rms value */
		k->rms = s;
		k->flag = 0;
	} else if (s > k->rms) {
		k->rms = s;
	}
	if (t >= k->peak) {                          /* peak hold and fallback */
		k->peak = t;
		k->cnt = k->hold;
	} else if (k->cnt > 0) {
		k->cnt -= k->fpp;
	} else {
		k->peak *= k->fall;
		k->peak += 1e-10f;
	}
}

void mo_kmeter_read (mo_kmeter* k, float* rms, float* peak)
{
	*rms = k->rms;
	*peak = k->peak;
	k->flag = 1;
}

/* ======================================================================
 * DR-14 / TP+RMS (src/dr14.c)
 * ====================================================================== */

static float dr_coeff_to_db (const float coeff)              /* dr14.c:235-238 */
{
	if (coeff < .0001) return -80;
	return 20 * log10f (coeff);
}

static float dr_db_to_coeff (const float db)                 /* dr14.c:240-243 */
{
	if (db <= -80) return 0;
	return powf (10, 0.05 * db);
}

void mo_dr14_reset (mo_dr14* d)
{
	for (int c = 0; c < d->n_channels; ++c) {
		d->m_peak[c] = -81;
		d->m_rms[c] = -81;
		d->m_dbtp[c] = 0;
		d->rms_sum[c] = 0;
		d->peak_cur[c] = 0;
		d->peak_hist[c][0] = d->peak_hist[c][1] = 0;
		mo_kmeter_reset (&d->km[c]);
		if (d->dr_mode) memset (d->hist[c], 0, sizeof (d->hist[c]));
	}
	d->sample_count = 0;
	d->num_fragments = 0;
}

void mo_dr14_init (mo_dr14* d, int n_channels, int dr_mode, double rate)
{
	memset (d, 0, sizeof (*d));
	d->n_channels = n_channels;
	d->dr_mode = dr_mode;
	d->rate = rate;
	d->n_sample_cnt = rintf (rate * 3.0);
	for (int c = 0; c < n_channels; ++c) {
		mo_kmeter_init (&d->km[c], rate);
		mo_tp_init (&d->tp[c], rate);
		d->m_rms[c] = -81;
		d->m_peak[c] = -81;
	}
}

/* dr14_calc_rms_score  dr14.c:283-352 */
static void dr14_window (mo_dr14* d)
{
	int silent = 1;
	for (int c = 0; c < d->n_channels; ++c)
		if (d->rms_sum[c] > 1e-9 * (float) d->n_sample_cnt) silent = 0;
	if (silent) {
		for (int c = 0; c < d->n_channels; ++c) d->rms_sum[c] = 0;
		return;
	}
	d->num_fragments++;
	float cutf = floorf (d->num_fragments / 5.0);
	const uint32_t m_cut = cutf > 1 ? cutf : 1;
	for (int c = 0; c < d->n_channels; ++c) {
		float rms = sqrt (2.f * d->rms_sum[c] / (float) d->n_sample_cnt);
		d->rms_sum[c] = 0;
		int bin = rintf (100.f * (80.f + dr_coeff_to_db (rms))) - 1;
		if (bin >= MO_DR_HISTBINS) bin = MO_DR_HISTBINS - 1;
		if (bin > 0) d->hist[c][bin]++;
		uint32_t n_cut = 0;
		float rms_score = 0;
		if (d->num_fragments > 2) {
			for (int32_t b = MO_DR_HISTBINS - 1; b > 0 && n_cut < m_cut; --b) {
				const uint32_t bc = d->hist[c][b];
				if (bc == 0) continue;
				const float cd = dr_db_to_coeff ((b - MO_DR_HISTBINS + 1) / 100.0);
				rms_score += cd * cd * (float) bc;
				n_cut += bc;
			}
		}
		if (n_cut > 0) rms_score = dr_coeff_to_db (sqrtf (rms_score / n_cut));
		else rms_score = -81;
		d->m_rms[c] = rms_score;
		if (d->peak_cur[c] >= d->peak_hist[c][0]) {
			d->peak_hist[c][1] = d->peak_hist[c][0];
			d->peak_hist[c][0] = d->peak_cur[c];
		} else if (d->peak_cur[c] > d->peak_hist[c][1]) {
			d->peak_hist[c][1] = d->peak_cur[c];
		}
		d->peak_cur[c] = 0;
		d->m_peak[c] = d->num_fragments > 2 ? dr_coeff_to_db (d->peak_hist[c][1]) : -81;
	}
}

void mo_dr14_run (mo_dr14* d, const float* const* in, uint32_t n, mo_dr14_ports* out)
{
	for (int c = 0; c < d->n_channels; ++c) {
		mo_kmeter_process (&d->km[c], in[c], n);
		mo_tp_process (&d->tp[c], in[c], n);
	}
	if (d->dr_mode) {
		uint64_t scnt = d->sample_count;
		for (uint32_t s = 0; s < n; ++s) {
			for (int c = 0; c < d->n_channels; ++c) {
				const float v = in[c][s];
				d->rms_sum[c] += v * v;
				d->peak_cur[c] = d->peak_cur[c] > v ? d->peak_cur[c] : v;
			}
			if (++scnt > d->n_sample_cnt) {
				dr14_window (d);
				scnt = 0;
			}
		}
		d->sample_count = scnt;
	}
	float dr_total = 0;
	int dr_valid = 0;
	memset (out, 0, sizeof (*out));
	for (int c = 0; c < d->n_channels; ++c) {
		float rv, rp, pv, pp;
		mo_tp_read2 (&d->tp[c], &pv, &pp);
		mo_kmeter_read (&d->km[c], &rv, &rp);
		if (pp > d->m_dbtp[c]) d->m_dbtp[c] = pp;
		out->v_rms[c] = dr_coeff_to_db (rv);
		out->v_peak[c] = dr_coeff_to_db (pv);
		out->m_peak[c] = dr_coeff_to_db (d->m_dbtp[c]);
		if (d->dr_mode) {
			const float rdb = d->m_rms[c], pdb = d->m_peak[c];
			const float dr = (0 < pdb ? 0 : pdb) - rdb;
			if (rdb > -80 && pdb > -80) { dr_total += dr; dr_valid++; }
			const float cl = 20 < dr ? 20 : dr;
			out->dr[c] = (rdb > -80 && pdb > -80) ? (1 > cl ? 1 : cl) : 21;
			out->m_rms[c] = rdb;
		} else {
			out->m_rms[c] = dr_coeff_to_db (rp);
		}
	}
	if (d->n_channels > 1 && d->dr_mode) {
		if (dr_valid > 0) {
			const float a = dr_total / (float) dr_valid;
			const float cl = 20 < a ? 20 : a;
			out->dr_total = 1 > cl ? 1 : cl;
		} else {
			out->dr_total = 21;
		}
	}
	out->block_count = 3.0 * d->num_fragments;
}

/* ======================================================================
 * Integer paths
 * ====================================================================== */

#define BIM_DHIT 0
#define BIM_NHIT 23
#define BIM_DONE 280
#define BIM_NONE 303
#define BIM_DSET 560

void mo_bitstats_reset (mo_bitstats* b)
{
	memset (b, 0, sizeof (*b));
	b->vmin = INFINITY;
	b->vmax = 0;
}

/* bitmeter.c:63-105 */
void mo_bitstats_run (mo_bitstats* b, const float* x, uint32_t n)
{
	for (uint32_t s = 0; s < n; ++s) {
		uint32_t bits;
		memcpy (&bits, &x[s], 4);
		uint32_t ex   = (bits & 0x7f800000u) >> 23;
		const int neg = (bits & 0x80000000u) != 0;
		const uint32_t man = bits & 0x7fffffu;

		if (ex == 255) {
			if (man == 0) ++b->n_inf; else ++b->n_nan;
			continue;
		}
		if (ex == 0 && man == 0) { ++b->n_zero; continue; }
		if (ex == 0) ++b->n_den;
		if (!neg) ++b->n_pos;
		if (ex > 0) {
			const float v = fabsf (x[s]);
			if (v > b->vmax) b->vmax = v;
			if (v < b->vmin) b->vmin = v;
			++b->hist[BIM_NHIT + ex];
			++b->hist[BIM_NONE + ex];
		} else {
			ex = 1;   /* denormals sit at 2^-126 */
		}
		for (int k = 0; k < 23; ++k) {
			++b->hist[BIM_DHIT + ex + k];
			if (man & (1u << k)) {
				++b->hist[BIM_DONE + ex + k];
				++b->hist[BIM_DSET + k];
			}
		}
	}
}

/* sdh_reset  sigdistlv2.c:49-61 (hist_peakS = -1: no peak yet) */
void mo_sigdist_reset (mo_sigdist* d) { memset (d, 0, sizeof (*d)); d->peak_bin = -1; }

/* sigdistlv2.c:303-318 */
void mo_sigdist_run (mo_sigdist* d, const float* x, uint32_t n)
{
	for (uint32_t s = 0; s < n; ++s) {
		const float val = x[s];
		int bin = rintf (180.f + val * 150.f);
		if (bin < 0) continue;
		if (bin >= MO_DIST_BIN) continue;
		if ((++d->bins[bin]) > d->peak_cnt) {
			d->peak_cnt = d->bins[bin];
			d->peak_bin = bin;
		}
		d->avg += val;
		const double var_m1 = d->var_m;
		const double cnt_a  = d->count + s + 1;
		d->var_m = d->var_m + ((double) val - d->var_m) / cnt_a;
		d->var_s = d->var_s + ((double) val - d->var_m) * ((double) val - var_m1);
	}
	d->count += n;
}

/* ======================================================================
 * Batch glue for tests / bench (not a restatement of anything)
 * ====================================================================== */

void mo_fill_lcg (float* x, uint32_t T, uint32_t seed, float gain)
{
	uint32_t s = seed;
	for (uint32_t i = 0; i < 2u * T; ++i) {
		s = 1664525u * s + 1013904223u;
		x[i] = gain * (((float)(int32_t)((s >> 8) - 8388608u)) / 8388608.0f);
	}
}

static void mo_deinterleave (const float* x, uint32_t n, float* L, float* R)
{
	for (uint32_t i = 0; i < n; ++i) { L[i] = x[2 * i]; R[i] = x[2 * i + 1]; }
}

void mo_batch_ebu (const float* x, uint32_t T, float fsamp, uint32_t block,
                   float* out9, int32_t* hist_M, int32_t* hist_S, int32_t* counts2,
                   float* frag_power)
{
	mo_ebu* e = (mo_ebu*) malloc (sizeof (mo_ebu));
	float* L = (float*) malloc (sizeof (float) * block);
	float* R = (float*) malloc (sizeof (float) * block);
	mo_ebu_init (e, 2, fsamp);
	mo_ebu_integr_start (e);
	uint32_t nf = 0;
	for (uint32_t pos = 0; pos < T; pos += block) {
		uint32_t n = (T - pos < block) ? T - pos : block;
		mo_deinterleave (x + 2 * (size_t) pos, n, L, R);
		if (frag_power) {
			/* walk fragment ends one at a time so each ring write can be captured */
			uint32_t done = 0;
			while (done < n) {
				uint32_t k = (uint32_t) e->frcnt < n - done ? (uint32_t) e->frcnt : n - done;
				const float* in[2] = { L + done, R + done };
				int w0 = e->wrind;
				mo_ebu_process (e, (int) k, in);
				if (e->wrind != w0) frag_power[nf++] = e->power[w0];
				done += k;
			}
		} else {
			const float* in[2] = { L, R };
			mo_ebu_process (e, (int) n, in);
		}
	}
	out9[0] = e->loudness_M; out9[1] = e->maxloudn_M;
	out9[2] = e->loudness_S; out9[3] = e->maxloudn_S;
	out9[4] = e->integrated; out9[5] = e->integ_thr;
	out9[6] = e->range_min;  out9[7] = e->range_max; out9[8] = e->range_thr;
	if (hist_M) memcpy (hist_M, e->hist_M.histc, sizeof (e->hist_M.histc));
	if (hist_S) memcpy (hist_S, e->hist_S.histc, sizeof (e->hist_S.histc));
	if (counts2) { counts2[0] = e->hist_M.count; counts2[1] = e->hist_S.count; }
	free (L); free (R); free (e);
}

void mo_batch_tp (const float* x, uint32_t T, float fsamp, uint32_t block, float* peak2)
{
	mo_tp tl, tr;
	if (block > MO_TP_MAXBLK) block = MO_TP_MAXBLK;
	float* L = (float*) malloc (sizeof (float) * block);
	float* R = (float*) malloc (sizeof (float) * block);
	mo_tp_init (&tl, fsamp);
	mo_tp_init (&tr, fsamp);
	peak2[0] = peak2[1] = 0;
	for (uint32_t pos = 0; pos < T; pos += block) {
		uint32_t n = (T - pos < block) ? T - pos : block;
		mo_deinterleave (x + 2 * (size_t) pos, n, L, R);
		mo_tp_process_max (&tl, L, (int) n);
		mo_tp_process_max (&tr, R, (int) n);
		float a = mo_tp_read (&tl), b = mo_tp_read (&tr);
		if (a > peak2[0]) peak2[0] = a;
		if (b > peak2[1]) peak2[1] = b;
	}
	free (L); free (R);
}

void mo_batch_spectr (const float* x, uint32_t T, double rate, uint32_t block,
                      float* val30, float* max30, float* valdb30, float* maxdb30)
{
	mo_spectr* s = (mo_spectr*) malloc (sizeof (mo_spectr));
	float* L = (float*) malloc (sizeof (float) * block);
	float* R = (float*) malloc (sizeof (float) * block);
	mo_spectr_init (s, 2, rate);
	for (uint32_t pos = 0; pos < T; pos += block) {
		uint32_t n = (T - pos < block) ? T - pos : block;
		mo_deinterleave (x + 2 * (size_t) pos, n, L, R);
		mo_spectr_run (s, L, R, n);
	}
	if (val30)   memcpy (val30, s->val_f, sizeof (s->val_f));
	if (max30)   memcpy (max30, s->max_f, sizeof (s->max_f));
	if (valdb30) memcpy (valdb30, s->spec_db, sizeof (s->spec_db));
	if (maxdb30) memcpy (maxdb30, s->max_db, sizeof (s->max_db));
	free (L); free (R); free (s);
}
