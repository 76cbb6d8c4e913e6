/* lv2_needle.c — the needle meters of lib/meters_amd.so besides VU and dBTP: CPU plumbing, as in the reference.
 *
 *   BBCmono/stereo, EBUmono/stereo   IEC 268-10 type II PPM   jmeters/iec2ppmdsp.cc   run       src/meters.cc:298-331
 *   DINmono/stereo, NORmono/stereo   IEC 268-10 type I PPM    jmeters/iec1ppmdsp.cc   run
 *   COR                              stereo phase correlation jmeters/stcorrdsp.cc    cor_run   src/meters.cc:566-588
 *   BBCM6                            M/S PPM                  jmeters/msppmdsp.cc     bbcm_run  src/meters.cc:603-637
 *   K12/K14/K20 mono/stereo          K-system RMS + peak      jmeters/kmeterdsp.cc    kmeter_run src/meters.cc:333-412
 *
 * These are one- and two-pole envelope followers on a handful of samples per run(): nothing for a GPU to do
 * at batch = 1 (SURVEY.md §8f rank 4 "DSP is trivial"), and the reference's own arithmetic order is kept so
 * the port values are bit-identical to the reference build (tests/test_needle_golden.py).  The BBC / EBU /
 * DIN / Nordic variants differ only in the scale the GUI paints, exactly as in the reference.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "lv2_min.h"
#include "lv2_plugins.h"
#include "lv2_dsp.h"

#define MTR_URI "http://gareus.org/oss/lv2/meters#"

/* ---- peak programme meter: two attack filters with a common decay, evaluated on groups of four ---- */
typedef struct { float z1, z2, m, w1, w2, w3, g; int res; } Ppm;

static void ppm_init (Ppm* p, int type, float fs)
{
	memset (p, 0, sizeof (*p));
	p->res = 1;
	if (type == 1) { p->w1 = 450.0f / fs; p->w2 = 1300.0f / fs; p->w3 = 1.0f - 5.4f / fs; p->g = 0.5108f; }   /* iec1ppmdsp.cc:89-95 */
	else           { p->w1 = 200.0f / fs; p->w2 = 860.0f / fs;  p->w3 = 1.0f - 4.0f / fs; p->g = 0.5141f; }   /* iec2ppmdsp.cc:89-95 */
}
static inline float clamp20 (float z) { return z > 20 ? 20 : (z < 0 ? 0 : z); }
static inline void ppm_hit (const Ppm* p, float t, float* z1, float* z2)
{
	if (t > *z1) *z1 += p->w1 * (t - *z1);
	if (t > *z2) *z2 += p->w2 * (t - *z2);
}
/* mode 0: |a|; 1: mv |a + b|; 2: mv |a - b|  (iec*ppmdsp.cc:47-79, msppmdsp.cc:50-114) */
static void ppm_process (Ppm* p, const float* a, const float* b, int n, int mode, float mv)
{
	float z1 = clamp20 (p->z1), z2 = clamp20 (p->z2);
	float m = p->res ? 0 : p->m;
	p->res = 0;
	for (n /= 4; n > 0; --n) {
		z1 *= p->w3;
		z2 *= p->w3;
		for (int q = 0; q < 4; ++q) {
			float t;
			if (mode == 0)      t = fabsf (*a++);
			else if (mode == 1) t = mv * fabsf (*a++ + *b++);
			else                t = mv * fabsf (*a++ - *b++);
			ppm_hit (p, t, &z1, &z2);
		}
		const float s = z1 + z2;
		if (s > m) m = s;
	}
	p->z1 = z1 + 1e-10f;
	p->z2 = z2 + 1e-10f;
	p->m = m;
}
static float ppm_read (Ppm* p) { p->res = 1; return p->g * p->m; }

/* ---- stereo correlation (Stcorrdsp, jmeters/stcorrdsp.cc:47-93): each channel through a one-pole low-pass of 2 kHz, then the
 * product and the two squares each through a one-pole of 0.3 s; read () divides the first by the geometric mean of the
 * others.  Five leaky integrators z <- z + w (target - z); the two in front carry the anti-denormal bias. ---- */
typedef struct { float zl, zr, zlr, zll, zrr, w1, w2; } Cor;
static inline float leak_to (float z, float w, float target) { return z + w * (target - z); }
static inline float leak_biased (float z, float w, float target) { return z + (w * (target - z) + 1e-20f); }
static inline float finite_or_zero (float v) { return isfinite (v) ? v : 0.f; }
static void cor_process (Cor* c, const float* pl, const float* pr, int n)
{
	const float w_in = c->w1, w_out = c->w2;
	float l = c->zl, r = c->zr, lr = c->zlr, ll = c->zll, rr = c->zrr;
	for (int i = 0; i < n; ++i) {
		l = leak_biased (l, w_in, pl[i]);
		r = leak_biased (r, w_in, pr[i]);
		lr = leak_to (lr, w_out, l * r);
		ll = leak_to (ll, w_out, l * l);
		rr = leak_to (rr, w_out, r * r);
	}
	c->zl = finite_or_zero (l);
	c->zr = finite_or_zero (r);
	c->zlr = finite_or_zero (lr) + 1e-10f;                     /* (the denominator of read () never vanishes) */
	c->zll = finite_or_zero (ll) + 1e-10f;
	c->zrr = finite_or_zero (rr) + 1e-10f;
}

/* ---- K-meter: lv2_dsp.h (shared with the DR14 / TP+RMS plugins) ---- */

/* ---- the instance, LV2meter of src/meters.cc:91-148 reduced to what these plugins use ---- */
enum { P_REFLEVEL = 0, P_INPUT0, P_OUTPUT0, P_LEVEL0, P_INPUT1, P_OUTPUT1, P_LEVEL1, P_PEAK0, P_PEAK1, P_HOLD };   /* :59-70 */
enum { T_PPM, T_COR, T_BM6, T_KM };

typedef struct {
	int type;
	uint32_t chn;
	float rlgain, p_refl, peak_hold;
	float* reflvl;
	float* input[2];
	float* output[2];
	float* level[2];
	float* peak[2];
	float* hold;
	Ppm ppm[2];
	float ms_db;                                 /* gain of the S needle of BBCM6, in dB */
	float ms_mv[2];
	Cor cor;
	Kmeter km[2];
} Needle;

LV2_Handle needle_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features)
{
	(void) path; (void) features;
	static const struct { const char* name; int type, ppm; } kinds[] = {
		{ "BBC", T_PPM, 2 }, { "EBU", T_PPM, 2 }, { "DIN", T_PPM, 1 }, { "NOR", T_PPM, 1 },
		{ "K12", T_KM, 0 }, { "K14", T_KM, 0 }, { "K20", T_KM, 0 } };
	if (strncmp (d->URI, MTR_URI, sizeof (MTR_URI) - 1)) return NULL;
	const char* name = d->URI + sizeof (MTR_URI) - 1;
	Needle* self = (Needle*) calloc (1, sizeof (Needle));
	if (!self) return NULL;
	self->rlgain = 1.0f;
	self->p_refl = -9999;
	if (!strcmp (name, "COR")) {                 /* src/meters.cc:202-207: init (rate, 2e3f, 0.3f) */
		self->type = T_COR; self->chn = 2;
		self->cor.w1 = 6.28f * 2e3f / (int) rate;
		self->cor.w2 = 1 / (0.3f * (int) rate);
		return self;
	}
	if (!strcmp (name, "BBCM6")) {               /* :208-214: two Msppmdsp (-6) */
		self->type = T_BM6; self->chn = 2;
		ppm_init (&self->ppm[0], 2, (float) rate);
		ppm_init (&self->ppm[1], 2, (float) rate);
		self->ms_db = -6;
		self->ms_mv[0] = self->ms_mv[1] = powf (10, .05 * -6);
		return self;
	}
	for (size_t i = 0; i < sizeof (kinds) / sizeof (kinds[0]); ++i) {
		const size_t l = strlen (kinds[i].name);
		if (strncmp (name, kinds[i].name, l)) continue;
		if (!strcmp (name + l, "mono")) self->chn = 1;
		else if (!strcmp (name + l, "stereo")) self->chn = 2;
		else continue;
		self->type = kinds[i].type;
		for (uint32_t c = 0; c < self->chn; ++c) {
			if (self->type == T_PPM) ppm_init (&self->ppm[c], kinds[i].ppm, (float) rate);
			else km_init (&self->km[c], (float) rate);
		}
		return self;
	}
	free (self);
	return NULL;
}

void needle_connect_port (LV2_Handle h, uint32_t port, void* data)
{
	Needle* self = (Needle*) h;
	switch (port) {
	case P_REFLEVEL: self->reflvl = (float*) data; break;
	case P_INPUT0:   self->input[0] = (float*) data; break;
	case P_OUTPUT0:  self->output[0] = (float*) data; break;
	case P_LEVEL0:   self->level[0] = (float*) data; break;
	case P_INPUT1:   self->input[1] = (float*) data; break;
	case P_OUTPUT1:  self->output[1] = (float*) data; break;
	case P_LEVEL1:   self->level[1] = (float*) data; break;
	case P_PEAK0:    self->peak[0] = (float*) data; break;
	case P_PEAK1:    self->peak[1] = (float*) data; break;
	case P_HOLD:     self->hold = (float*) data; break;
	default: break;
	}
}

static void reference_level (Needle* self)       /* src/meters.cc:303-306 */
{
	if (self->p_refl != *self->reflvl) {
		self->p_refl = *self->reflvl;
		self->rlgain = powf (10.0f, 0.05f * (self->p_refl + 18.0));
	}
}
static void pass_through (Needle* self, uint32_t c, uint32_t n)
{
	if (self->input[c] != self->output[c]) memcpy (self->output[c], self->input[c], sizeof (float) * n);
}

void needle_run (LV2_Handle h, uint32_t n_samples)  /* run, src/meters.cc:298-331 */
{
	Needle* self = (Needle*) h;
	reference_level (self);
	for (uint32_t c = 0; c < self->chn; ++c) {
		ppm_process (&self->ppm[c], self->input[c], NULL, (int) n_samples, 0, 1.0f);
		*self->level[c] = self->rlgain * ppm_read (&self->ppm[c]);
		pass_through (self, c, n_samples);
	}
}

void cor_run (LV2_Handle h, uint32_t n_samples)     /* :566-588 */
{
	Needle* self = (Needle*) h;
	cor_process (&self->cor, self->input[0], self->input[1], (int) n_samples);
	*self->level[0] = self->cor.zlr / sqrtf (self->cor.zll * self->cor.zrr + 1e-10f);
	pass_through (self, 0, n_samples);
	pass_through (self, 1, n_samples);
}

void bbcm_run (LV2_Handle h, uint32_t n_samples)    /* :603-637 */
{
	Needle* self = (Needle*) h;
	reference_level (self);
	const float db = (*self->peak[0] > 0.5) ? +14 : -6;            /* port 7: the S needle's +20 dB switch */
	if (self->ms_db != db) { self->ms_db = db; self->ms_mv[1] = powf (10, .05 * db); }
	ppm_process (&self->ppm[0], self->input[0], self->input[1], (int) n_samples, 1, self->ms_mv[0]);
	*self->level[0] = self->rlgain * ppm_read (&self->ppm[0]);
	ppm_process (&self->ppm[1], self->input[0], self->input[1], (int) n_samples, 2, self->ms_mv[1]);
	*self->level[1] = self->rlgain * ppm_read (&self->ppm[1]);
	pass_through (self, 0, n_samples);
	pass_through (self, 1, n_samples);
}

void kmeter_run (LV2_Handle h, uint32_t n_samples)  /* :333-412 */
{
	Needle* self = (Needle*) h;
	int reinit_gui = 0;
	/* port 0 doubles as the UI's request channel for the peak values */
	if (self->p_refl != *self->reflvl) {
		if (fabsf (*self->reflvl) < 3) {         /* reset peak-hold */
			self->peak_hold = 0;
			reinit_gui = 1;
			for (uint32_t c = 0; c < self->chn; ++c) km_reset (&self->km[c]);
		}
		if (fabsf (*self->reflvl) == 3) reinit_gui = 1;            /* re-notify until the UI acknowledges */
		else self->p_refl = *self->reflvl;
	}
	for (uint32_t c = 0; c < self->chn; ++c) {
		km_process (&self->km[c], self->input[c], (int) n_samples);
		pass_through (self, c, n_samples);
	}
	if (reinit_gui) {                            /* force a parameter change */
		if (self->chn == 1) *self->output[1] = -1 - (rand () & 0xffff);   /* port 5 */
		else *self->hold = -1 - (rand () & 0xffff);
		return;
	}
	if (self->chn == 1) {                        /* mono re-uses ports 4 and 5 for peak and hold */
		self->km[0].flag = 1;
		*self->level[0] = self->rlgain * self->km[0].rms;
		*self->input[1] = self->rlgain * self->km[0].peak;
		if (*self->input[1] > self->peak_hold) self->peak_hold = *self->input[1];
		*self->output[1] = self->peak_hold;
	} else {
		for (uint32_t c = 0; c < 2; ++c) {
			self->km[c].flag = 1;
			*self->level[c] = self->rlgain * self->km[c].rms;
			*self->peak[c] = self->rlgain * self->km[c].peak;
			if (*self->peak[c] > self->peak_hold) self->peak_hold = *self->peak[c];
		}
		*self->hold = self->peak_hold;
	}
}

void needle_cleanup (LV2_Handle h) { free (h); }

/* ======================================================================================
 * surround3 .. surround8 (src/surmeter.c): a K-meter per channel (level + peak) and up to four
 * correlation meters whose input pairs are chosen on control ports.  Ports: 0 reference level,
 * 1..12 = four x (channel a, channel b, correlation out), then per channel in, out, level, peak.
 * ====================================================================================== */
#define SUR_MAXCH 8
typedef struct {
	uint32_t chn;
	float* reflvl;
	float* surc_a[4];
	float* surc_b[4];
	float* surc_c[4];
	float* input[SUR_MAXCH];
	float* output[SUR_MAXCH];
	float* level[SUR_MAXCH];
	float* peak[SUR_MAXCH];
	Kmeter km[SUR_MAXCH];
	Cor cor4[4];
} Surround;

LV2_Handle sur_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features)
{
	(void) path; (void) features;
	const size_t l = sizeof (MTR_URI "surround") - 1;
	if (strncmp (d->URI, MTR_URI "surround", l) || d->URI[l] < '3' || d->URI[l] > '8' || d->URI[l + 1]) return NULL;
	Surround* self = (Surround*) calloc (1, sizeof (Surround));
	if (!self) return NULL;
	self->chn = (uint32_t) (d->URI[l] - '0');
	for (uint32_t c = 0; c < self->chn; ++c) km_init (&self->km[c], (float) rate);
	for (uint32_t c = 0; c < 4; ++c) {                        /* Stcorrdsp::init (rate, 2e3f, 0.3f), surmeter.c:64-67 */
		self->cor4[c].w1 = 6.28f * 2e3f / (int) rate;
		self->cor4[c].w2 = 1 / (0.3f * (int) rate);
	}
	return self;
}

void sur_connect_port (LV2_Handle h, uint32_t port, void* data)   /* surmeter.c:74-113 */
{
	Surround* self = (Surround*) h;
	if (port == 0) {
		self->reflvl = (float*) data;
	} else if (port <= 12) {
		const int cor = (port - 1) / 3;
		switch (port % 3) {
		case 1: self->surc_a[cor] = (float*) data; break;
		case 2: self->surc_b[cor] = (float*) data; break;
		default: self->surc_c[cor] = (float*) data; break;
		}
	} else if (port <= 12 + 4 * self->chn) {
		const int chan = (port - 13) / 4;
		switch (port % 4) {
		case 1: self->input[chan] = (float*) data; break;
		case 2: self->output[chan] = (float*) data; break;
		case 3: self->level[chan] = (float*) data; break;
		default: self->peak[chan] = (float*) data; break;
		}
	}
}

void sur_run (LV2_Handle h, uint32_t n_samples)               /* surmeter.c:115-144 */
{
	Surround* self = (Surround*) h;
	const uint32_t cors = self->chn > 3 ? 4 : 3;
	for (uint32_t c = 0; c < cors; ++c) {
		uint32_t in_a = rintf (*self->surc_a[c]);
		uint32_t in_b = rintf (*self->surc_b[c]);
		if (in_a >= self->chn) in_a = self->chn - 1;
		if (in_b >= self->chn) in_b = self->chn - 1;
		cor_process (&self->cor4[c], self->input[in_a], self->input[in_b], (int) n_samples);
		*self->surc_c[c] = self->cor4[c].zlr / sqrtf (self->cor4[c].zll * self->cor4[c].zrr + 1e-10f);
	}
	for (uint32_t c = 0; c < self->chn; ++c) {
		float m, p;
		km_process (&self->km[c], self->input[c], (int) n_samples);
		km_read (&self->km[c], &m, &p);
		*self->level[c] = m;
		*self->peak[c] = p;
		if (self->input[c] != self->output[c]) memcpy (self->output[c], self->input[c], sizeof (float) * n_samples);
	}
}
