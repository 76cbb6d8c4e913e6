/* lv2_ebur128.c — the EBUr128 plugin (http://gareus.org/oss/lv2/meters#EBUr128) of lib/meters_amd.so.
 *
 * Mirrors src/ebulv2.cc of the reference: port map (:31-38), instantiate (:118-199), run (:239-498),
 * LV2 State (:514-566), with the DSP (Ebu_r128_proc::process + 2 x TruePeakdsp::process_max) on the GPU
 * through the engine's C ABI.  The whole plugin <-> UI atom protocol is spoken, so the reference's GUI
 * (or any host-side consumer of it) sees the same messages:
 *
 *   in   time:Position {speed}                transport follow (update_position, :84-112)
 *        meteron / meteroff                   UI attached / detached; meteron resyncs radar + histogram
 *        metercfg {key, value}                START PAUSE RESET TRANSPORTSYNC AUTORESET RADARTIME UISETTINGS
 *   out  control {key, value}                 LV2_FTM, LV2_RADARTIME, UISETTINGS (state to UI), LV2_RESETRADAR,
 *                                             LV2_RESYNCDONE
 *        rdr_radarpoint {M, S, pos, cur, max} radar ring: resync batches and one point per radar step
 *        rdr_histpoint {M, S, pos}            histogram bins that changed (<= 17 per cycle), bins 110..649
 *        rdr_histogram {M, S}                 new histogram maxima
 *        ebulevels {10 values}                every cycle while the UI is attached
 *
 * No LV2 SDK in this image: the atom layouts come from include/lv2_min.h, the writer is lv2_forge.h.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lv2_min.h"
#include "mtr_engine.h"
#include "lv2_forge.h"
#include "lv2_plugins.h"

#define HIST_LEN 751                                         /* src/uris.h:45 */

enum { EBU_CONTROL = 0, EBU_NOTIFY, EBU_INPUT0, EBU_OUTPUT0, EBU_INPUT1, EBU_OUTPUT1 };

typedef struct {
	ForgeUrids f;
	LV2_URID mtr_ebulevels, ebu_loudnessM, ebu_maxloudnM, ebu_loudnessS, ebu_maxloudnS;
	LV2_URID ebu_integrated, ebu_range_min, ebu_range_max, ebu_integrating, ebu_integr_time, mtr_truepeak;
	LV2_URID ebu_state;
	LV2_URID rdr_histogram, rdr_histpoint, rdr_radarpoint, rdr_pointpos, rdr_pos_cur, rdr_pos_max;
} Urids;

typedef struct {
	float* input[2];
	float* output[2];
	const LV2_Atom_Sequence* control;
	LV2_Atom_Sequence* notify;
	LV2_URID_Map* map;
	Urids u;
	Forge fg;
	double rate;

	int ui_active, send_state_to_ui;
	int follow_transport_mode, tranport_rolling;             /* (sic) the reference's spelling */
	int ebu_integrating, dbtp_enable;
	uint32_t ui_settings;

	float *radarS, *radarM;
	float radarSC, radarMC;
	int radar_pos_cur, radar_pos_max;
	uint32_t radar_spd_cur, radar_spd_max;
	int radar_resync;
	uint64_t integration_time;
	int32_t histM[HIST_LEN], histS[HIST_LEN];
	int32_t hist_maxM, hist_maxS;
	float tp_max;

	mtr_engine* amd;
	int32_t devM[HIST_LEN], devS[HIST_LEN];                  /* the engine's histograms, fetched per cycle */
} Ebu;

/* ---- helpers of the reference, src/ebulv2.cc:44-112 ---------------------------------------------- */
static void ebu_reset (Ebu* self)
{
	mtr_engine_integr_reset (self->amd);
	/* Not in the reference's ebu_reset: its TruePeakdsp objects are never reset, but their read() returns
	 * the peak since the previous read, so tp_max = -inf below forgets the past there too. */
	mtr_engine_truepeak_reset (self->amd);
	kv_message (&self->fg, CTL_LV2_RESETRADAR, 0);
	for (int i = 0; i < self->radar_pos_max; ++i) { self->radarS[i] = -INFINITY; self->radarM[i] = -INFINITY; }
	memset (self->histM, 0, sizeof (self->histM));
	memset (self->histS, 0, sizeof (self->histS));
	self->radar_pos_cur = 0;
	self->integration_time = 0;
	self->hist_maxM = 0;
	self->hist_maxS = 0;
	self->tp_max = -INFINITY;
}

static void ebu_integrate (Ebu* self, int on)
{
	if (self->ebu_integrating == on) return;
	if (on) {
		if (self->follow_transport_mode & 2) ebu_reset (self);
		mtr_engine_integr_start (self->amd);
		self->ebu_integrating = 1;
	} else {
		mtr_engine_integr_pause (self->amd);
		self->ebu_integrating = 0;
	}
}

static void ebu_set_radarspeed (Ebu* self, float seconds)
{
	self->radar_spd_max = (uint32_t) rint (seconds * self->rate / self->radar_pos_max);
	if (self->radar_spd_max < 4096) self->radar_spd_max = 4096;
}

static void update_position (Ebu* self, const LV2_Atom_Object* obj)
{
	const LV2_Atom* speed = object_get (obj, self->u.f.time_speed);
	if (speed && speed->type == self->u.f.atom_Float) {
		const float ts = ((const LV2_Atom_Float*) speed)->body;
		if (ts != 0 && !self->tranport_rolling) { if (self->follow_transport_mode & 1) ebu_integrate (self, 1); }
		if (ts == 0 && self->tranport_rolling)  { if (self->follow_transport_mode & 1) ebu_integrate (self, 0); }
		self->tranport_rolling = (ts != 0);
	}
}

/* ---- LV2 callbacks ----------------------------------------------------------------------------- */
LV2_Handle ebur128_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features)
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
	forge_map_urids (self->map, &self->u.f);
	MAP (mtr_ebulevels, MTR_URI "ebulevels");
	MAP (ebu_loudnessM, MTR_URI "ebu_loudnessM"); MAP (ebu_maxloudnM, MTR_URI "ebu_maxloudnM");
	MAP (ebu_loudnessS, MTR_URI "ebu_loudnessS"); MAP (ebu_maxloudnS, MTR_URI "ebu_maxloudnS");
	MAP (ebu_integrated, MTR_URI "ebu_integrated"); MAP (ebu_range_min, MTR_URI "ebu_range_min");
	MAP (ebu_range_max, MTR_URI "ebu_range_max"); MAP (ebu_integrating, MTR_URI "ebu_integrating");
	MAP (ebu_integr_time, MTR_URI "ebu_integr_time"); MAP (mtr_truepeak, MTR_URI "truepeak");
	MAP (ebu_state, MTR_URI "ebu_state");
	MAP (rdr_histogram, MTR_URI "rdr_histogram"); MAP (rdr_histpoint, MTR_URI "rdr_histpoint");
	MAP (rdr_radarpoint, MTR_URI "rdr_radarpoint"); MAP (rdr_pointpos, MTR_URI "rdr_pointpos");
	MAP (rdr_pos_cur, MTR_URI "rdr_pos_cur"); MAP (rdr_pos_max, MTR_URI "rdr_pos_max");
#undef MAP
	self->rate = rate;
	self->radar_pos_max = 360;
	self->radar_resync = -1;
	self->ui_settings = 8;
	self->radarS = (float*) malloc (self->radar_pos_max * sizeof (float));
	self->radarM = (float*) malloc (self->radar_pos_max * sizeof (float));
	if (!self->radarS || !self->radarM) { free (self->radarS); free (self->radarM); free (self); return NULL; }
	self->radarSC = self->radarMC = -INFINITY;
	for (int i = 0; i < self->radar_pos_max; ++i) { self->radarS[i] = -INFINITY; self->radarM[i] = -INFINITY; }
	ebu_set_radarspeed (self, 2.0f * 60.0f);
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
		free (self->radarS); free (self->radarM); free (self);
		return NULL;
	}
	return self;
}

void ebur128_connect_port (LV2_Handle h, uint32_t port, void* data)
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

static void radar_point (Ebu* self, float m, float s, int pos)
{
	ObjFrame fr;
	if (!obj_begin (&self->fg, &fr, self->u.rdr_radarpoint)) return;
	prop_f (&self->fg, self->u.ebu_loudnessM, m);
	prop_f (&self->fg, self->u.ebu_loudnessS, s);
	prop_i (&self->fg, self->u.rdr_pointpos, pos);
	prop_i (&self->fg, self->u.rdr_pos_cur, self->radar_pos_cur);
	prop_i (&self->fg, self->u.rdr_pos_max, self->radar_pos_max);
	obj_end (&self->fg, &fr);
}

void ebur128_run (LV2_Handle h, uint32_t n_samples)
{
	Ebu* self = (Ebu*) h;
	const uint32_t capacity = self->notify->atom.size;
	Forge* const fg = &self->fg;
	forge_begin (fg, self->notify, &self->u.f);

	if (self->send_state_to_ui && self->ui_active) {          /* :248-255 */
		self->send_state_to_ui = 0;
		kv_message (fg, CTL_LV2_FTM, (float) self->follow_transport_mode);
		kv_message (fg, CTL_LV2_RADARTIME, (float) (self->radar_pos_max * self->radar_spd_max / self->rate));
		kv_message (fg, CTL_UISETTINGS, (float) self->ui_settings);
	}

	/* incoming events, :257-331 */
	if (self->control) {
		FORGE_FOREACH_OBJECT (self->control, &self->u.f, obj) {
			if (obj->body.otype == self->u.f.time_Position) {
				update_position (self, obj);
			} else if (obj->body.otype == self->u.f.mtr_meters_on) {
				self->ui_active = 1;
				self->send_state_to_ui = 1;
				self->radar_resync = 0;
				memset (self->histM, 0, sizeof (self->histM));          /* resync histogram */
				memset (self->histS, 0, sizeof (self->histS));
				self->hist_maxM = 0;
				self->hist_maxS = 0;
			} else if (obj->body.otype == self->u.f.mtr_meters_off) {
				self->ui_active = 0;
			} else if (obj->body.otype == self->u.f.mtr_meters_cfg) {
				int key; float val;
				get_cc_key_value (&self->u.f, obj, &key, &val);
				switch (key) {
				case CTL_START: ebu_integrate (self, 1); break;
				case CTL_PAUSE: ebu_integrate (self, 0); break;
				case CTL_RESET: ebu_reset (self); break;
				case CTL_TRANSPORTSYNC:
					if (val == 1) {
						self->follow_transport_mode |= 1;
						if (self->tranport_rolling != self->ebu_integrating) ebu_integrate (self, self->tranport_rolling);
					} else {
						self->follow_transport_mode &= ~1;
					}
					break;
				case CTL_AUTORESET:
					if (val == 1) self->follow_transport_mode |= 2; else self->follow_transport_mode &= ~2;
					break;
				case CTL_RADARTIME:
					if (val >= 30 && val <= 600) {
						ebu_set_radarspeed (self, val);
						if (self->radar_spd_max < 2 * n_samples) self->radar_spd_max = 2 * n_samples;
					}
					kv_message (fg, CTL_LV2_RADARTIME, (float) (self->radar_pos_max * self->radar_spd_max / self->rate));
					break;
				case CTL_UISETTINGS:
					self->ui_settings = (uint32_t) val;
					self->dbtp_enable = (self->ui_settings & 64) ? 1 : 0;
					break;
				default: break;
				}
			}
		}
	}

	/* audio, :340-347: one batch-of-one launch; results come back in one record */
	const float* in[2] = { self->input[0], self->input[1] };
	if (n_samples > 0) mtr_engine_process_planar_host (self->amd, in, n_samples);
	mtr_stream_result r;
	memset (&r, 0, sizeof (r));
	mtr_engine_results (self->amd, 0, 1, &r);
	const float lm = r.loudness_M, ls = r.loudness_S;

	if (self->dbtp_enable) {                                  /* :360-367 */
		const float tp0 = r.truepeak_call[0], tp1 = r.truepeak_call[1];
		const float tpm = tp0 > tp1 ? tp0 : tp1;
		const float tp = tpm == 0 ? -INFINITY : (float) (20.0 * log10f (tpm));
		if (tp > self->tp_max) self->tp_max = tp;
	} else {
		self->tp_max = -INFINITY;
	}

	if (self->radar_resync >= 0) {                            /* :369-388 */
		int batch = ((int) capacity - 512) / 192;
		if (batch > 16) batch = 16;
		for (int i = 0; i < batch; i++, self->radar_resync++) {
			if (self->radar_resync >= self->radar_pos_max) {
				self->radar_resync = -1;
				kv_message (fg, CTL_LV2_RESYNCDONE, 0);
				break;
			}
			radar_point (self, self->radarM[self->radar_resync], self->radarS[self->radar_resync], self->radar_resync);
		}
	}

	/* radar history, :390-423 (the second test reads `lm` in the reference too) */
	if (lm > self->radarMC) self->radarMC = lm;
	if (lm > self->radarSC) self->radarSC = ls;
	if (self->ebu_integrating) self->integration_time += n_samples;
	self->radar_spd_cur += n_samples;
	if (self->radar_spd_cur > self->radar_spd_max) {
		if (self->ui_active) radar_point (self, self->radarMC, self->radarSC, self->radar_pos_cur);
		self->radarM[self->radar_pos_cur] = self->radarMC;
		self->radarS[self->radar_pos_cur] = self->radarSC;
		self->radar_spd_cur = self->radar_spd_cur % self->radar_spd_max;
		self->radar_pos_cur = (self->radar_pos_cur + 1) % self->radar_pos_max;
		self->radarSC = self->radarMC = -INFINITY;
	}

	if (self->ui_active) {                                    /* histogram diffs, :425-462 */
		int msgtx = 0;
		if (r.hist_M_count > 10 && r.hist_S_count > 10
		    && mtr_engine_histograms (self->amd, 0, 1, self->devM, self->devS) == MTR_OK) {
			int max_changed = 0;
			for (int i = 110; i < 650; i++) {
				const int32_t vm = self->devM[i], vs = self->devS[i];
				if (capacity - forge_used (fg) <= 512) break;
				if (self->histM[i] != vm || self->histS[i] != vs) {
					if (msgtx++ > 16) break;                  /* limit max data-rate */
					self->histM[i] = vm;
					self->histS[i] = vs;
					ObjFrame fr;
					if (obj_begin (fg, &fr, self->u.rdr_histpoint)) {
						prop_i (fg, self->u.ebu_loudnessM, vm);
						prop_i (fg, self->u.ebu_loudnessS, vs);
						prop_i (fg, self->u.rdr_pointpos, i);
						obj_end (fg, &fr);
					}
				}
				if (vm > self->hist_maxM) { self->hist_maxM = vm; max_changed = 1; }
				if (vs > self->hist_maxS) { self->hist_maxS = vs; max_changed = 1; }
			}
			if (max_changed) {
				ObjFrame fr;
				if (obj_begin (fg, &fr, self->u.rdr_histogram)) {
					prop_i (fg, self->u.ebu_loudnessM, self->hist_maxM);
					prop_i (fg, self->u.ebu_loudnessS, self->hist_maxS);
					obj_end (fg, &fr);
				}
			}
		}
	}

	if (self->ui_active) {                                    /* `ebulevels`, :464-482 */
		ObjFrame fr;
		if (obj_begin (fg, &fr, self->u.mtr_ebulevels)) {
			prop_f (fg, self->u.ebu_loudnessM, lm);
			prop_f (fg, self->u.ebu_maxloudnM, r.maxloudn_M);
			prop_f (fg, self->u.ebu_loudnessS, ls);
			prop_f (fg, self->u.ebu_maxloudnS, r.maxloudn_S);
			prop_f (fg, self->u.ebu_integrated, r.integrated);
			prop_f (fg, self->u.ebu_range_min, r.range_min);
			prop_f (fg, self->u.ebu_range_max, r.range_max);
			prop_f (fg, self->u.mtr_truepeak, self->tp_max);
			prop_b (fg, self->u.ebu_integrating, self->ebu_integrating);
			prop_f (fg, self->u.ebu_integr_time, (float) (self->integration_time / self->rate));
			obj_end (fg, &fr);
		}
	}

	for (int c = 0; c < 2; ++c)
		if (self->input[c] != self->output[c]) memcpy (self->output[c], self->input[c], sizeof (float) * n_samples);
}

void ebur128_cleanup (LV2_Handle h)
{
	Ebu* self = (Ebu*) h;
	if (self->amd) mtr_engine_destroy (self->amd);
	free (self->radarS);
	free (self->radarM);
	free (self);
}

/* ---- LV2 State, src/ebulv2.cc:514-553: one atom:Int = ui_settings | follow_transport_mode << 8 | radar_spd_max << 16 */
static LV2_State_Status ebur128_save (LV2_Handle h, LV2_State_Store_Function store, LV2_State_Handle handle,
                                      uint32_t flags, const LV2_Feature* const* features)
{
	(void) flags; (void) features;
	Ebu* self = (Ebu*) h;
	uint32_t cfg = self->ui_settings;
	cfg |= (uint32_t) self->follow_transport_mode << 8;
	cfg |= self->radar_spd_max << 16;
	store (handle, self->u.ebu_state, (void*) &cfg, sizeof (uint32_t), self->u.f.atom_Int, LV2_STATE_IS_POD | LV2_STATE_IS_PORTABLE);
	return LV2_STATE_SUCCESS;
}

static LV2_State_Status ebur128_restore (LV2_Handle h, LV2_State_Retrieve_Function retrieve, LV2_State_Handle handle,
                                         uint32_t flags, const LV2_Feature* const* features)
{
	(void) flags; (void) features;
	Ebu* self = (Ebu*) h;
	size_t size;
	uint32_t type, valflags;
	const void* value = retrieve (handle, self->u.ebu_state, &size, &type, &valflags);
	if (value && size == sizeof (uint32_t) && type == self->u.f.atom_Int) {
		const uint32_t cfg = *((const uint32_t*) value);
		self->ui_settings = cfg & 0xff;
		self->follow_transport_mode = (cfg >> 8) & 0x3;
		self->radar_spd_max = cfg >> 16;
		self->dbtp_enable = (self->ui_settings & 64) ? 1 : 0;
		self->send_state_to_ui = 1;
	}
	return LV2_STATE_SUCCESS;
}

const void* ebur128_extension_data (const char* uri)
{
	static const LV2_State_Interface state = { ebur128_save, ebur128_restore };
	if (!strcmp (uri, LV2_STATE__interface)) return &state;
	return NULL;
}
