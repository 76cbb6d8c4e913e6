/* lv2_plugin.c — the LV2 plugin surface (lib/meters_amd.so) of the MI355X meters engine.
 *
 * Host code in plain C; each instance is an n_streams = 1 client of libmtr_engine.so.  It mirrors the
 * reference's descriptors, port maps and per-block run() semantics for the in-scope URIs:
 *
 *   index  URI (prefix http://gareus.org/oss/lv2/meters#)   reference
 *   0      VUmono          src/meters.cc:298-331 (run), jmeters/vumeterdsp.cc   — CPU plumbing (config 0)
 *   1      VUstereo
 *   2      EBUr128         src/ebulv2.cc (lv2_ebur128.c) — K-weighting + true peak on the GPU, full UI protocol + State
 *   3      spectr30mono    src/spectrumlv2.c:159-257    — 30-band bank on the GPU
 *   4      dBTPmono        src/meters.cc:438-508        — TruePeakdsp::process on the GPU
 *   5      dBTPstereo
 *   6      spectr30stereo
 *   7      SigDistHist     src/sigdistlv2.c (lv2_intstat.c) — signal-distribution histogram on the GPU, UI protocol + State
 *   8      bitmeter        src/bitmeter.c (lv2_intstat.c)   — IEEE-754 bit statistics on the GPU, UI protocol + State
 *   9-24   BBC / EBU / DIN / NOR mono+stereo, COR, BBCM6, K12 / K14 / K20 mono+stereo (lv2_needle.c) — CPU plumbing
 *   25-28  dr14mono/stereo, TPnRMSmono/stereo   src/dr14.c (lv2_dr14.c) — true-peak ballistics on the GPU
 *   29-34  surround8 .. surround3               src/surmeter.c (lv2_needle.c) — CPU plumbing
 *
 * The reference enumerates 38 plugins (src/meters.cc:745-792); LV2 hosts match by URI and stop at
 * the first NULL, so the in-scope subset is enumerated densely.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lv2_min.h"
#include "mtr_engine.h"
#include "lv2_plugins.h"

#define MTR_URI "http://gareus.org/oss/lv2/meters#"

static const void* no_extension (const char* uri) { (void) uri; return NULL; }
static int finitef_ (float v) { return isfinite (v); }

/* ======================================================================================
 * VU (CPU): Vumeterdsp, jmeters/vumeterdsp.cc:45-87.  _w and _g are class statics in the
 * reference (one sample rate per process); kept as file statics here.
 * ====================================================================================== */
static float vu_w = 0, vu_g = 0;
typedef struct { float z1, z2, m; int res; } VuDsp;

static void vu_process (VuDsp* v, const float* p, int n)
{
	float z1 = v->z1 > 20 ? 20 : (v->z1 < -20 ? -20 : v->z1);
	float z2 = v->z2 > 20 ? 20 : (v->z2 < -20 ? -20 : v->z2);
	float m = v->res ? 0 : v->m;
	v->res = 0;
	for (n /= 4; n > 0; --n) {          /* groups of four; n mod 4 trailing samples are dropped */
		const float t2 = z2 / 2;
		for (int q = 0; q < 4; ++q) { const float t1 = fabsf (*p++) - t2; z1 += vu_w * (t1 - z1); }
		z2 += 4 * vu_w * (z1 - z2);
		if (z2 > m) m = z2;
	}
	if (!finitef_ (z1)) { v->z1 = 0; m = INFINITY; } else v->z1 = z1;
	if (!finitef_ (z2)) { v->z2 = 0; m = INFINITY; } else v->z2 = z2 + 1e-10f;
	v->m = m;
}
static float vu_read (VuDsp* v) { v->res = 1; return vu_g * v->m; }

/* ======================================================================================
 * needle-meter / dBTP instance (LV2meter, src/meters.cc:91-175; ports :59-70)
 * ====================================================================================== */
enum { MTR_REFLEVEL = 0, MTR_INPUT0, MTR_OUTPUT0, MTR_LEVEL0, MTR_INPUT1, MTR_OUTPUT1, MTR_LEVEL1, MTR_PEAK0, MTR_PEAK1 };
enum { KIND_VU, KIND_DBTP };

typedef struct {
	int      kind;
	uint32_t chn;
	float    rlgain, p_refl;
	float*   reflvl;
	float*   level[2];
	float*   input[2];
	float*   output[2];
	float*   peak[2];
	float    peak_max[2];
	int      unread;            /* a reset run returned before read(m, p): that block's values are still pending */
	float    unread_m[2], unread_p[2];
	VuDsp    vu[2];
	mtr_engine* amd;
	int failing;               /* an engine call failed in the last run (): lv2_engine_ok */
} Meter;

static LV2_Handle meter_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* f)
{
	(void) path; (void) f;
	Meter* self = (Meter*) calloc (1, sizeof (Meter));
	if (!self) return NULL;
	const char* name = d->URI + strlen (MTR_URI);
	if (!strcmp (name, "VUmono"))          { self->kind = KIND_VU;   self->chn = 1; }
	else if (!strcmp (name, "VUstereo"))   { self->kind = KIND_VU;   self->chn = 2; }
	else if (!strcmp (name, "dBTPmono"))   { self->kind = KIND_DBTP; self->chn = 1; }
	else if (!strcmp (name, "dBTPstereo")) { self->kind = KIND_DBTP; self->chn = 2; }
	else { free (self); return NULL; }                         /* src/meters.cc:224-227 */
	if (self->kind == KIND_VU) {
		vu_w = 11.1f / (float) rate;                           /* Vumeterdsp::init, vumeterdsp.cc:82-86 */
		vu_g = 1.5f * 1.571f;
		self->vu[0].res = self->vu[1].res = 1;
	} else {
		mtr_config cfg;
		memset (&cfg, 0, sizeof (cfg));
		cfg.struct_size = sizeof (cfg);
		cfg.meters = MTR_METER_TPBALLIST;
		cfg.n_streams = 1;
		cfg.n_channels = self->chn;
		cfg.sample_rate = (float) rate;
		if (lv2_engine_open (&cfg, f, &self->amd) != MTR_OK) {
			fprintf (stderr, "meters_amd: dBTP: %s\n", mtr_last_error ());
			free (self);
			return NULL;
		}
	}
	self->rlgain = 1.0;
	self->p_refl = -9999;
	return self;
}

static void meter_connect_port (LV2_Handle h, uint32_t port, void* data)
{
	Meter* self = (Meter*) h;
	switch (port) {
	case MTR_REFLEVEL: self->reflvl = (float*) data; break;
	case MTR_INPUT0:   self->input[0] = (float*) data; break;
	case MTR_OUTPUT0:  self->output[0] = (float*) data; break;
	case MTR_LEVEL0:   self->level[0] = (float*) data; break;
	case MTR_INPUT1:   self->input[1] = (float*) data; break;
	case MTR_OUTPUT1:  self->output[1] = (float*) data; break;
	case MTR_LEVEL1:   self->level[1] = (float*) data; break;
	case MTR_PEAK0:    self->peak[0] = (float*) data; break;
	case MTR_PEAK1:    self->peak[1] = (float*) data; break;
	default: break;
	}
}

/* src/meters.cc:298-331 */
static void vu_run (LV2_Handle h, uint32_t n_samples)
{
	Meter* self = (Meter*) h;
	if (self->p_refl != *self->reflvl) {
		self->p_refl = *self->reflvl;
		self->rlgain = powf (10.0f, 0.05f * (self->p_refl + 18.0));
	}
	for (uint32_t c = 0; c < self->chn; ++c) {
		vu_process (&self->vu[c], self->input[c], (int) n_samples);
		*self->level[c] = self->rlgain * vu_read (&self->vu[c]);
		if (self->input[c] && self->output[c] && self->input[c] != self->output[c]) memcpy (self->output[c], self->input[c], sizeof (float) * n_samples);
	}
}

/* src/meters.cc:438-508 */
static void dbtp_run (LV2_Handle h, uint32_t n_samples)
{
	Meter* self = (Meter*) h;
	int reinit_gui = 0;
	if (self->p_refl != *self->reflvl) {
		if (fabsf (*self->reflvl) < 3) {                    /* reset peak-hold */
			reinit_gui = 1;
			self->peak_max[0] = self->peak_max[1] = 0;
			self->unread = 0;
			mtr_engine_truepeak_reset (self->amd);
		}
		if (fabsf (*self->reflvl) != 3) self->p_refl = *self->reflvl;
	}
	if (fabsf (*self->reflvl) == 3) reinit_gui = 1;

	const float* in[2] = { self->input[0], self->input[1] };
	int rc = MTR_OK;
	if (n_samples > 0) rc = mtr_engine_process_planar_host (self->amd, in, n_samples);
	for (uint32_t c = 0; c < self->chn; ++c)
		if (self->input[c] && self->output[c] && self->input[c] != self->output[c]) memcpy (self->output[c], self->input[c], sizeof (float) * n_samples);

	mtr_stream_result r;
	memset (&r, 0, sizeof (r));
	if (n_samples > 0 && rc == MTR_OK) rc = mtr_engine_results (self->amd, 0, 1, &r);
	if (!lv2_engine_ok (rc, &self->failing, "dBTP")) {
		for (uint32_t c = 0; c < self->chn; ++c) *self->level[c] = MTR_LV2_NO_DATA;
		if (self->chn == 1) *self->input[1] = MTR_LV2_NO_DATA;      /* port index 4: the mono peak output */
		else { *self->peak[0] = MTR_LV2_NO_DATA; *self->peak[1] = MTR_LV2_NO_DATA; }
		return;
	}
	/* TruePeakdsp keeps max-accumulating until read (m, p) is called (truepeakdsp.cc:91-98): a block
	 * whose run() returned early below is folded into the next read. */
	float m[2], p[2];
	for (uint32_t c = 0; c < self->chn; ++c) {
		m[c] = r.tpb_level[c]; p[c] = r.tpb_peak[c];
		if (self->unread) {
			if (self->unread_m[c] > m[c]) m[c] = self->unread_m[c];
			if (self->unread_p[c] > p[c]) p[c] = self->unread_p[c];
		}
	}
	if (reinit_gui) {                                       /* force a parameter change, :477-489 */
		for (uint32_t c = 0; c < self->chn; ++c) { self->unread_m[c] = m[c]; self->unread_p[c] = p[c]; }
		self->unread = 1;
		*self->level[0] = -500 - (rand () & 0xffff);
		if (self->chn == 1) {
			*self->input[1] = -500 - (rand () & 0xffff);    /* port index 4 */
		} else {
			*self->level[1] = -500 - (rand () & 0xffff);
			*self->peak[0] = -500 - (rand () & 0xffff);
			*self->peak[1] = -500 - (rand () & 0xffff);
		}
		return;
	}
	self->unread = 0;
	for (uint32_t c = 0; c < self->chn; ++c) {              /* TruePeakdsp::read (m, p), :491-507 */
		if (self->peak_max[c] < self->rlgain * p[c]) self->peak_max[c] = self->rlgain * p[c];
		*self->level[c] = self->rlgain * m[c];
	}
	if (self->chn == 1) *self->input[1] = self->peak_max[0];    /* port index 4, :496 */
	else { *self->peak[0] = self->peak_max[0]; *self->peak[1] = self->peak_max[1]; }
}

static void meter_cleanup (LV2_Handle h)
{
	Meter* self = (Meter*) h;
	if (self->amd) mtr_engine_destroy (self->amd);
	free (self);
}

/* ======================================================================================
 * spectr30 (LV2spec, src/spectrumlv2.c:35-157)
 * ====================================================================================== */
enum { SA_SPEED = 60, SA_RESET = 61, SA_AMP = 62, SA_STATE = 63, SA_INPUT0 = 64, SA_OUTPUT0 = 65, SA_INPUT1 = 66, SA_OUTPUT1 = 67 };

typedef struct {
	float* input[2];
	float* output[2];
	float* spec[MTR_NBANDS];
	float* maxf[MTR_NBANDS];
	float* rst_p;
	float* spd_p;
	float  rst_h, spd_h;
	uint32_t nchannels;
	mtr_engine* amd;
	int failing;               /* an engine call failed in the last run (): lv2_engine_ok */
} Spec;

static LV2_Handle spectrum_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* f)
{
	(void) path; (void) f;
	uint32_t nch;
	if (!strcmp (d->URI, MTR_URI "spectr30stereo")) nch = 2;
	else if (!strcmp (d->URI, MTR_URI "spectr30mono")) nch = 1;
	else return NULL;
	Spec* self = (Spec*) calloc (1, sizeof (Spec));
	if (!self) return NULL;
	self->nchannels = nch;
	self->rst_h = -4;
	self->spd_h = 1.0;
	mtr_config cfg;
	memset (&cfg, 0, sizeof (cfg));
	cfg.struct_size = sizeof (cfg);
	cfg.meters = MTR_METER_SPECTR30;
	cfg.n_streams = 1;
	cfg.n_channels = nch;
	cfg.sample_rate = (float) rate;
	if (lv2_engine_open (&cfg, f, &self->amd) != MTR_OK) {
		fprintf (stderr, "meters_amd: spectr30: %s\n", mtr_last_error ());
		free (self);
		return NULL;
	}
	return self;
}

static void spectrum_connect_port (LV2_Handle h, uint32_t port, void* data)
{
	Spec* self = (Spec*) h;
	switch (port) {
	case SA_INPUT0:  self->input[0] = (float*) data; break;
	case SA_OUTPUT0: self->output[0] = (float*) data; break;
	case SA_INPUT1:  self->input[1] = (float*) data; break;
	case SA_OUTPUT1: self->output[1] = (float*) data; break;
	case SA_RESET:   self->rst_p = (float*) data; break;
	case SA_SPEED:   self->spd_p = (float*) data; break;
	case SA_AMP: case SA_STATE: break;
	default:
		if (port < 30) self->spec[port] = (float*) data;
		else if (port < 60) self->maxf[port - 30] = (float*) data;
		break;
	}
}

/* src/spectrumlv2.c:159-257 */
static void spectrum_run (LV2_Handle h, uint32_t n_samples)
{
	Spec* self = (Spec*) h;
	int reinit_gui = 0;
	if (self->spd_h != *self->spd_p) {
		self->spd_h = *self->spd_p;
		mtr_engine_spectr_set_speed (self->amd, self->spd_h);   /* clamps to [0.01, 15] like :172-175 */
		self->rst_h = 0;
	}
	if (self->rst_h != *self->rst_p) {
		if (fabsf (*self->rst_p) < 3 || self->rst_h == 0) {
			reinit_gui = 1;
			mtr_engine_spectr_reset_peak (self->amd);
		}
		if (fabsf (*self->rst_p) != 3) self->rst_h = *self->rst_p;
	}
	if (fabsf (*self->rst_p) == 3) reinit_gui = 1;

	const float* in[2] = { self->input[0], self->input[1] };
	int rc = MTR_OK;
	if (n_samples > 0) rc = mtr_engine_process_planar_host (self->amd, in, n_samples);

	float val_db[MTR_NBANDS], max_db[MTR_NBANDS];
	if (rc == MTR_OK) rc = mtr_engine_spectrum (self->amd, 0, 1, NULL, NULL, val_db, max_db);
	if (lv2_engine_ok (rc, &self->failing, "spectr30")) {
		for (int i = 0; i < MTR_NBANDS; ++i) {
			*self->spec[i] = val_db[i];
			*self->maxf[i] = reinit_gui ? (float) (-500 - (rand () & 0xffff)) : max_db[i];
		}
	} else {
		for (int i = 0; i < MTR_NBANDS; ++i) { *self->spec[i] = MTR_LV2_NO_DATA; *self->maxf[i] = MTR_LV2_NO_DATA; }
	}
	for (uint32_t c = 0; c < self->nchannels; ++c)
		if (self->input[c] && self->output[c] && self->input[c] != self->output[c]) memcpy (self->output[c], self->input[c], sizeof (float) * n_samples);
}

static void spectrum_cleanup (LV2_Handle h)
{
	Spec* self = (Spec*) h;
	if (self->amd) mtr_engine_destroy (self->amd);
	free (self);
}

/* ======================================================================================
 * descriptors (8 positional members, src/meters.cc:683-693)
 * ====================================================================================== */
static const LV2_Descriptor descriptors[] = {
	{ MTR_URI "VUmono",         meter_instantiate,    meter_connect_port,    NULL, vu_run,       NULL, meter_cleanup,    no_extension },
	{ MTR_URI "VUstereo",       meter_instantiate,    meter_connect_port,    NULL, vu_run,       NULL, meter_cleanup,    no_extension },
	{ MTR_URI "EBUr128",        ebur128_instantiate,  ebur128_connect_port,  NULL, ebur128_run,  NULL, ebur128_cleanup,  ebur128_extension_data },
	{ MTR_URI "spectr30mono",   spectrum_instantiate, spectrum_connect_port, NULL, spectrum_run, NULL, spectrum_cleanup, no_extension },
	{ MTR_URI "dBTPmono",       meter_instantiate,    meter_connect_port,    NULL, dbtp_run,     NULL, meter_cleanup,    no_extension },
	{ MTR_URI "dBTPstereo",     meter_instantiate,    meter_connect_port,    NULL, dbtp_run,     NULL, meter_cleanup,    no_extension },
	{ MTR_URI "spectr30stereo", spectrum_instantiate, spectrum_connect_port, NULL, spectrum_run, NULL, spectrum_cleanup, no_extension },
	{ MTR_URI "SigDistHist",    sdh_instantiate,      intstat_connect_port,  NULL, sdh_run,      NULL, intstat_cleanup,  sdh_extension_data },
	{ MTR_URI "bitmeter",       bim_instantiate,      intstat_connect_port,  NULL, bim_run,      NULL, intstat_cleanup,  bim_extension_data },
#define NEEDLE(name, runfn) { MTR_URI name, needle_instantiate, needle_connect_port, NULL, runfn, NULL, needle_cleanup, no_extension }
	NEEDLE ("BBCmono", needle_run), NEEDLE ("BBCstereo", needle_run), NEEDLE ("EBUmono", needle_run), NEEDLE ("EBUstereo", needle_run),
	NEEDLE ("DINmono", needle_run), NEEDLE ("DINstereo", needle_run), NEEDLE ("NORmono", needle_run), NEEDLE ("NORstereo", needle_run),
	NEEDLE ("COR", cor_run), NEEDLE ("BBCM6", bbcm_run),
	NEEDLE ("K12mono", kmeter_run), NEEDLE ("K14mono", kmeter_run), NEEDLE ("K20mono", kmeter_run),
	NEEDLE ("K12stereo", kmeter_run), NEEDLE ("K14stereo", kmeter_run), NEEDLE ("K20stereo", kmeter_run),
#undef NEEDLE
#define DR14(name) { MTR_URI name, dr14_instantiate, dr14_connect_port, NULL, dr14_run, NULL, dr14_cleanup, no_extension }
	DR14 ("dr14mono"), DR14 ("dr14stereo"), DR14 ("TPnRMSmono"), DR14 ("TPnRMSstereo"),
#undef DR14
#define SUR(name) { MTR_URI name, sur_instantiate, sur_connect_port, NULL, sur_run, NULL, needle_cleanup, no_extension }
	SUR ("surround8"), SUR ("surround7"), SUR ("surround6"), SUR ("surround5"), SUR ("surround4"), SUR ("surround3"),
#undef SUR
};

LV2_SYMBOL_EXPORT const LV2_Descriptor* lv2_descriptor (uint32_t index)
{
	return index < sizeof (descriptors) / sizeof (descriptors[0]) ? &descriptors[index] : NULL;
}
