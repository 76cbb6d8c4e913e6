/* lv2_plugin.c — the LV2 plugin surface (lib/meters_amd.so) of the MI355X meters engine.
 *
 * Host code in plain C; each instance is an n_streams = 1 client of libmtr_engine.so.  It mirrors the
 * reference's descriptors, port maps and per-block run() semantics for the in-scope URIs:
 *
 *   index  URI (prefix http://gareus.org/oss/lv2/meters#)   reference
 *   0      VUmono          src/meters.cc:298-331 (run), jmeters/vumeterdsp.cc   — CPU plumbing (config 0)
 *   1      VUstereo
 *   2      EBUr128         src/ebulv2.cc:239-498        — K-weighting + true peak on the GPU
 *   3      spectr30mono    src/spectrumlv2.c:159-257    — 30-band bank on the GPU
 *   4      dBTPmono        src/meters.cc:438-508        — TruePeakdsp::process on the GPU
 *   5      dBTPstereo
 *   6      spectr30stereo
 *
 * The reference enumerates 38 plugins (src/meters.cc:745-792); LV2 hosts match by URI and stop at
 * the first NULL, so the in-scope subset is enumerated densely.  Not mirrored:
 * the EBU radar / histogram-diff messages, transport sync and LV2 State; the EBU plugin speaks the
 * subset of the atom protocol a headless host needs: meteron/meteroff, metercfg {START, PAUSE, RESET,
 * UISETTINGS} in, the `ebulevels` object out.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lv2_min.h"
#include "mtr_engine.h"

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
		if (mtr_engine_create (&cfg, &self->amd) != MTR_OK) {
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
		if (self->input[c] != self->output[c]) memcpy (self->output[c], self->input[c], sizeof (float) * n_samples);
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
	if (n_samples > 0) mtr_engine_process_planar_host (self->amd, in, n_samples);
	for (uint32_t c = 0; c < self->chn; ++c)
		if (self->input[c] != self->output[c]) memcpy (self->output[c], self->input[c], sizeof (float) * n_samples);

	mtr_stream_result r;
	memset (&r, 0, sizeof (r));
	if (n_samples > 0 && mtr_engine_results (self->amd, 0, 1, &r) != MTR_OK) return;
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
	if (mtr_engine_create (&cfg, &self->amd) != MTR_OK) {
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
	if (n_samples > 0) mtr_engine_process_planar_host (self->amd, in, n_samples);

	float val_db[MTR_NBANDS], max_db[MTR_NBANDS];
	if (mtr_engine_spectrum (self->amd, 0, 1, NULL, NULL, val_db, max_db) == MTR_OK) {
		for (int i = 0; i < MTR_NBANDS; ++i) {
			*self->spec[i] = val_db[i];
			*self->maxf[i] = reinit_gui ? (float) (-500 - (rand () & 0xffff)) : max_db[i];
		}
	}
	for (uint32_t c = 0; c < self->nchannels; ++c)
		if (self->input[c] != self->output[c]) memcpy (self->output[c], self->input[c], sizeof (float) * n_samples);
}

static void spectrum_cleanup (LV2_Handle h)
{
	Spec* self = (Spec*) h;
	if (self->amd) mtr_engine_destroy (self->amd);
	free (self);
}

/* ======================================================================================
 * EBUr128 (src/ebulv2.cc) with the headless subset of the atom protocol
 * ====================================================================================== */
enum { EBU_CONTROL = 0, EBU_NOTIFY, EBU_INPUT0, EBU_OUTPUT0, EBU_INPUT1, EBU_OUTPUT1 };
enum { CTL_START = 1, CTL_PAUSE, CTL_RESET, CTL_TRANSPORTSYNC, CTL_AUTORESET, CTL_RADARTIME, CTL_UISETTINGS };   /* src/uris.h:187-203 */

typedef struct {
	LV2_URID atom_Blank, atom_Object, atom_Int, atom_Float, atom_Bool, atom_Sequence;
	LV2_URID mtr_ebulevels, ebu_loudnessM, ebu_maxloudnM, ebu_loudnessS, ebu_maxloudnS;
	LV2_URID ebu_integrated, ebu_range_min, ebu_range_max, ebu_integrating, ebu_integr_time, mtr_truepeak;
	LV2_URID mtr_cckey, mtr_ccval, mtr_control, mtr_meters_on, mtr_meters_off, mtr_meters_cfg;
} Urids;

typedef struct {
	float* input[2];
	float* output[2];
	const LV2_Atom_Sequence* control;
	LV2_Atom_Sequence* notify;
	LV2_URID_Map* map;
	Urids u;
	double rate;
	int ui_active, ebu_integrating, dbtp_enable;
	uint32_t ui_settings;
	uint64_t integration_time;
	float tp_max;
	mtr_engine* amd;
} Ebu;

static LV2_Handle ebur128_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features)
{
	(void) path;
	if (strcmp (d->URI, MTR_URI "EBUr128")) return NULL;
	Ebu* self = (Ebu*) calloc (1, sizeof (Ebu));
	if (!self) return NULL;
	for (int i = 0; features && features[i]; ++i)
		if (!strcmp (features[i]->URI, LV2_URID__map)) self->map = (LV2_URID_Map*) features[i]->data;
	if (!self->map) {                                        /* src/ebulv2.cc:140-144 */
		fprintf (stderr, "EBUrLV2 error: Host does not support urid:map\n");
		free (self);
		return NULL;
	}
#define MAP(field, uri) self->u.field = self->map->map (self->map->handle, uri)
	MAP (atom_Blank, LV2_ATOM__Blank); MAP (atom_Object, LV2_ATOM__Object); MAP (atom_Int, LV2_ATOM__Int);
	MAP (atom_Float, LV2_ATOM__Float); MAP (atom_Bool, LV2_ATOM__Bool); MAP (atom_Sequence, LV2_ATOM__Sequence);
	MAP (mtr_ebulevels, MTR_URI "ebulevels");
	MAP (ebu_loudnessM, MTR_URI "ebu_loudnessM"); MAP (ebu_maxloudnM, MTR_URI "ebu_maxloudnM");
	MAP (ebu_loudnessS, MTR_URI "ebu_loudnessS"); MAP (ebu_maxloudnS, MTR_URI "ebu_maxloudnS");
	MAP (ebu_integrated, MTR_URI "ebu_integrated"); MAP (ebu_range_min, MTR_URI "ebu_range_min");
	MAP (ebu_range_max, MTR_URI "ebu_range_max"); MAP (ebu_integrating, MTR_URI "ebu_integrating");
	MAP (ebu_integr_time, MTR_URI "ebu_integr_time"); MAP (mtr_truepeak, MTR_URI "truepeak");
	MAP (mtr_cckey, MTR_URI "controlkey"); MAP (mtr_ccval, MTR_URI "controlval"); MAP (mtr_control, MTR_URI "control");
	MAP (mtr_meters_on, MTR_URI "meteron"); MAP (mtr_meters_off, MTR_URI "meteroff"); MAP (mtr_meters_cfg, MTR_URI "metercfg");
#undef MAP
	self->rate = rate;
	self->ui_settings = 8;
	self->tp_max = -INFINITY;
	mtr_config cfg;
	memset (&cfg, 0, sizeof (cfg));
	cfg.struct_size = sizeof (cfg);
	cfg.meters = MTR_METER_EBU | MTR_METER_TRUEPEAK;
	cfg.n_streams = 1;
	cfg.n_channels = 2;
	cfg.sample_rate = (float) rate;
	if (mtr_engine_create (&cfg, &self->amd) != MTR_OK) {
		fprintf (stderr, "meters_amd: EBUr128: %s\n", mtr_last_error ());
		free (self);
		return NULL;
	}
	return self;
}

static void ebur128_connect_port (LV2_Handle h, uint32_t port, void* data)
{
	Ebu* self = (Ebu*) h;
	switch (port) {
	case EBU_INPUT0:  self->input[0] = (float*) data; break;
	case EBU_OUTPUT0: self->output[0] = (float*) data; break;
	case EBU_INPUT1:  self->input[1] = (float*) data; break;
	case EBU_OUTPUT1: self->output[1] = (float*) data; break;
	case EBU_NOTIFY:  self->notify = (LV2_Atom_Sequence*) data; break;
	case EBU_CONTROL: self->control = (const LV2_Atom_Sequence*) data; break;
	default: break;
	}
}

/* ---- a forge for exactly what `ebulevels` needs ------------------------------------------- */
typedef struct { uint8_t* buf; uint32_t cap, pos; } Forge;
static uint32_t pad8 (uint32_t n) { return (n + 7u) & ~7u; }
static void* forge_raw (Forge* f, uint32_t n)
{
	if (f->pos + pad8 (n) > f->cap) return NULL;
	void* p = f->buf + f->pos;
	memset (p, 0, pad8 (n));
	f->pos += pad8 (n);
	return p;
}
static void forge_prop_f32 (Forge* f, LV2_URID key, LV2_URID type, float v)
{
	LV2_Atom_Property_Body* p = (LV2_Atom_Property_Body*) forge_raw (f, sizeof (LV2_Atom_Property_Body) + 4);
	if (!p) return;
	p->key = key; p->context = 0; p->value.size = 4; p->value.type = type;
	memcpy (p + 1, &v, 4);
}
static void forge_prop_i32 (Forge* f, LV2_URID key, LV2_URID type, int32_t v)
{
	LV2_Atom_Property_Body* p = (LV2_Atom_Property_Body*) forge_raw (f, sizeof (LV2_Atom_Property_Body) + 4);
	if (!p) return;
	p->key = key; p->context = 0; p->value.size = 4; p->value.type = type;
	memcpy (p + 1, &v, 4);
}

/* value of property `key` inside an object body, or NULL */
static const LV2_Atom* object_get (const LV2_Atom_Object* obj, LV2_URID key)
{
	const uint8_t* p = (const uint8_t*) (&obj->body + 1);
	const uint8_t* end = (const uint8_t*) &obj->body + obj->atom.size;
	while (p + sizeof (LV2_Atom_Property_Body) <= end) {
		const LV2_Atom_Property_Body* pb = (const LV2_Atom_Property_Body*) p;
		if (pb->key == key) return &pb->value;
		p += pad8 ((uint32_t) sizeof (LV2_Atom_Property_Body) + pb->value.size);
	}
	return NULL;
}

/* src/ebulv2.cc:239-498 (the parts named in the file header) */
static void ebur128_run (LV2_Handle h, uint32_t n_samples)
{
	Ebu* self = (Ebu*) h;
	const uint32_t capacity = self->notify->atom.size;       /* the host presets the capacity, :244 */
	Forge fg = { (uint8_t*) self->notify, capacity + (uint32_t) sizeof (LV2_Atom), 0 };
	LV2_Atom_Sequence* seq = (LV2_Atom_Sequence*) forge_raw (&fg, sizeof (LV2_Atom_Sequence));
	if (seq) { seq->atom.type = self->u.atom_Sequence; seq->atom.size = sizeof (LV2_Atom_Sequence_Body); }

	/* incoming events, :258-331 */
	if (self->control) {
		const uint8_t* p = (const uint8_t*) (&self->control->body + 1);
		const uint8_t* end = (const uint8_t*) &self->control->body + self->control->atom.size;
		while (p + sizeof (LV2_Atom_Event) <= end) {
			const LV2_Atom_Event* ev = (const LV2_Atom_Event*) p;
			if (ev->body.type == self->u.atom_Blank || ev->body.type == self->u.atom_Object) {
				const LV2_Atom_Object* obj = (const LV2_Atom_Object*) &ev->body;
				if (obj->body.otype == self->u.mtr_meters_on) self->ui_active = 1;
				else if (obj->body.otype == self->u.mtr_meters_off) self->ui_active = 0;
				else if (obj->body.otype == self->u.mtr_meters_cfg) {
					const LV2_Atom* k = object_get (obj, self->u.mtr_cckey);
					const LV2_Atom* v = object_get (obj, self->u.mtr_ccval);
					if (k && v) {
						const int key = ((const LV2_Atom_Int*) k)->body;
						const float val = ((const LV2_Atom_Float*) v)->body;
						switch (key) {
						case CTL_START: if (!self->ebu_integrating) { mtr_engine_integr_start (self->amd); self->ebu_integrating = 1; } break;
						case CTL_PAUSE: if (self->ebu_integrating) { mtr_engine_integr_pause (self->amd); self->ebu_integrating = 0; } break;
						case CTL_RESET:                                     /* ebu_reset, :47-63 */
							mtr_engine_integr_reset (self->amd);
							mtr_engine_truepeak_reset (self->amd);
							self->integration_time = 0;
							self->tp_max = -INFINITY;
							break;
						case CTL_UISETTINGS:
							self->ui_settings = (uint32_t) val;
							self->dbtp_enable = (self->ui_settings & 64) ? 1 : 0;
							break;
						default: break;
						}
					} else {
						fprintf (stderr, "MTRlv2: Malformed ctrl message has no key or value.\n");
					}
				}
			}
			p += pad8 ((uint32_t) sizeof (LV2_Atom_Event) + ev->body.size);
		}
	}

	/* audio, :340-347 */
	const float* in[2] = { self->input[0], self->input[1] };
	if (n_samples > 0) mtr_engine_process_planar_host (self->amd, in, n_samples);
	mtr_stream_result r;
	memset (&r, 0, sizeof (r));
	mtr_engine_results (self->amd, 0, 1, &r);

	if (self->dbtp_enable) {                                  /* :360-367 */
		const float tp0 = r.truepeak_call[0], tp1 = r.truepeak_call[1];
		const float tpm = tp0 > tp1 ? tp0 : tp1;
		const float tp = tpm == 0 ? -INFINITY : (float) (20.0 * log10f (tpm));
		if (tp > self->tp_max) self->tp_max = tp;
	} else {
		self->tp_max = -INFINITY;
	}
	if (self->ebu_integrating) self->integration_time += n_samples;

	/* `ebulevels` to the UI, :465-482 */
	if (self->ui_active && seq) {
		const uint32_t ev_pos = fg.pos;
		LV2_Atom_Event* ev = (LV2_Atom_Event*) forge_raw (&fg, sizeof (LV2_Atom_Event) + sizeof (LV2_Atom_Object_Body));
		if (ev) {
			ev->frames = 0;
			ev->body.type = self->u.atom_Object;
			LV2_Atom_Object_Body* ob = (LV2_Atom_Object_Body*) (ev + 1);
			ob->id = 1; ob->otype = self->u.mtr_ebulevels;
			const uint32_t body0 = fg.pos - (uint32_t) sizeof (LV2_Atom_Object_Body);
			forge_prop_f32 (&fg, self->u.ebu_loudnessM, self->u.atom_Float, r.loudness_M);
			forge_prop_f32 (&fg, self->u.ebu_maxloudnM, self->u.atom_Float, r.maxloudn_M);
			forge_prop_f32 (&fg, self->u.ebu_loudnessS, self->u.atom_Float, r.loudness_S);
			forge_prop_f32 (&fg, self->u.ebu_maxloudnS, self->u.atom_Float, r.maxloudn_S);
			forge_prop_f32 (&fg, self->u.ebu_integrated, self->u.atom_Float, r.integrated);
			forge_prop_f32 (&fg, self->u.ebu_range_min, self->u.atom_Float, r.range_min);
			forge_prop_f32 (&fg, self->u.ebu_range_max, self->u.atom_Float, r.range_max);
			forge_prop_f32 (&fg, self->u.mtr_truepeak, self->u.atom_Float, self->tp_max);
			forge_prop_i32 (&fg, self->u.ebu_integrating, self->u.atom_Bool, self->ebu_integrating);
			forge_prop_f32 (&fg, self->u.ebu_integr_time, self->u.atom_Float, (float) (self->integration_time / self->rate));
			ev->body.size = fg.pos - body0;
			seq->atom.size += fg.pos - ev_pos;
		}
	}

	for (int c = 0; c < 2; ++c)
		if (self->input[c] != self->output[c]) memcpy (self->output[c], self->input[c], sizeof (float) * n_samples);
}

static void ebur128_cleanup (LV2_Handle h)
{
	Ebu* self = (Ebu*) h;
	if (self->amd) mtr_engine_destroy (self->amd);
	free (self);
}

/* ======================================================================================
 * descriptors (8 positional members, src/meters.cc:683-693)
 * ====================================================================================== */
static const LV2_Descriptor descriptors[] = {
	{ MTR_URI "VUmono",         meter_instantiate,    meter_connect_port,    NULL, vu_run,       NULL, meter_cleanup,    no_extension },
	{ MTR_URI "VUstereo",       meter_instantiate,    meter_connect_port,    NULL, vu_run,       NULL, meter_cleanup,    no_extension },
	{ MTR_URI "EBUr128",        ebur128_instantiate,  ebur128_connect_port,  NULL, ebur128_run,  NULL, ebur128_cleanup,  no_extension },
	{ MTR_URI "spectr30mono",   spectrum_instantiate, spectrum_connect_port, NULL, spectrum_run, NULL, spectrum_cleanup, no_extension },
	{ MTR_URI "dBTPmono",       meter_instantiate,    meter_connect_port,    NULL, dbtp_run,     NULL, meter_cleanup,    no_extension },
	{ MTR_URI "dBTPstereo",     meter_instantiate,    meter_connect_port,    NULL, dbtp_run,     NULL, meter_cleanup,    no_extension },
	{ MTR_URI "spectr30stereo", spectrum_instantiate, spectrum_connect_port, NULL, spectrum_run, NULL, spectrum_cleanup, no_extension },
};

LV2_SYMBOL_EXPORT const LV2_Descriptor* lv2_descriptor (uint32_t index)
{
	return index < sizeof (descriptors) / sizeof (descriptors[0]) ? &descriptors[index] : NULL;
}
