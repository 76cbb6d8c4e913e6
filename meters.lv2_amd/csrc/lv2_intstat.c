/* lv2_intstat.c — the two integer-statistics plugins of lib/meters_amd.so:
 *
 *   bitmeter     (http://gareus.org/oss/lv2/meters#bitmeter)     src/bitmeter.c:112-397
 *   SigDistHist  (http://gareus.org/oss/lv2/meters#SigDistHist)  src/sigdistlv2.c:110-460
 *
 * Both are mono, both talk to their UI over an atom control / notify port pair with the EBU plugin's
 * vocabulary (meteron / meteroff / metercfg in, objects out), both keep their counting on the GPU
 * (k_bitstats / k_sigdist, bit-exact tables) and everything the reference keeps per instance besides the
 * tables — integration switch and time, the ~5 fps / 25 fps message cadence, the windowed clear of the
 * bit meter, transport follow of the histogram, LV2 State — here, in plain C.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lv2_min.h"
#include "mtr_engine.h"
#include "lv2_forge.h"
#include "lv2_plugins.h"

enum { IS_CONTROL = 0, IS_NOTIFY, IS_INPUT0, IS_OUTPUT0, IS_INPUT1, IS_OUTPUT1 };   /* BIMPortIndex / SDHPortIndex */

#define BIM_LAST 584                                         /* src/uris.h:60 */
#define DIST_BIN 361                                         /* src/uris.h:47 */

typedef struct {
	ForgeUrids f;
	LV2_URID ebu_integrating, ebu_integr_time;
	LV2_URID bim_state, bim_information, bim_averaging, bim_stats, bim_data;
	LV2_URID bim_zero, bim_pos, bim_min, bim_max, bim_nan, bim_inf, bim_den;
	LV2_URID sdh_state, sdh_histogram, sdh_hist_max, sdh_hist_var, sdh_hist_avg, sdh_hist_peak, sdh_hist_data, sdh_information;
} Urids;

typedef struct {
	float* input[2];
	float* output[2];
	uint32_t chn;
	const LV2_Atom_Sequence* control;
	LV2_Atom_Sequence* notify;
	LV2_URID_Map* map;
	Urids u;
	Forge fg;
	double rate;
	mtr_engine* amd;

	int ui_active, send_state_to_ui, integrating;
	int64_t integration_time;
	int radar_resync;                                        /* samples since the last message (the reference's name) */

	/* bitmeter */
	int bim_average;
	int32_t nan_base, inf_base, den_base;                    /* counted before the last windowed clear */
	/* SigDistHist */
	int follow_transport_mode, tranport_rolling;
	uint32_t ui_settings;
} IntStat;

static IntStat* common_instantiate (const LV2_Feature* const* features, double rate, uint32_t meter, const char* name)
{
	IntStat* self = (IntStat*) calloc (1, sizeof (IntStat));
	if (!self) return NULL;
	for (int i = 0; features && features[i]; ++i)
		if (!strcmp (features[i]->URI, LV2_URID__map)) self->map = (LV2_URID_Map*) features[i]->data;
	if (!self->map) {
		fprintf (stderr, "%s error: Host does not support urid:map\n", name);
		free (self);
		return NULL;
	}
	forge_map_urids (self->map, &self->u.f);
#define MAP(field, uri) self->u.field = self->map->map (self->map->handle, MTR_URI uri)
	MAP (ebu_integrating, "ebu_integrating"); MAP (ebu_integr_time, "ebu_integr_time");
	MAP (bim_state, "bim_state"); MAP (bim_information, "bim_information"); MAP (bim_averaging, "bim_averaging");
	MAP (bim_stats, "bim_stats"); MAP (bim_data, "bim_data"); MAP (bim_zero, "bim_zero"); MAP (bim_pos, "bim_pos");
	MAP (bim_min, "bim_min"); MAP (bim_max, "bim_max"); MAP (bim_nan, "bim_nan"); MAP (bim_inf, "bim_inf"); MAP (bim_den, "bim_den");
	MAP (sdh_state, "sdh_state"); MAP (sdh_histogram, "sdh_histogram"); MAP (sdh_hist_max, "sdh_hist_max");
	MAP (sdh_hist_var, "sdh_hist_var"); MAP (sdh_hist_avg, "sdh_hist_avg"); MAP (sdh_hist_peak, "sdh_hist_peak");
	MAP (sdh_hist_data, "sdh_hist_data"); MAP (sdh_information, "sdh_information");
#undef MAP
	self->rate = rate;
	self->chn = 1;
	mtr_config cfg;
	memset (&cfg, 0, sizeof (cfg));
	cfg.struct_size = sizeof (cfg);
	cfg.meters = meter;
	cfg.n_streams = 1;
	cfg.n_channels = 1;
	cfg.sample_rate = (float) rate;
	if (mtr_engine_create (&cfg, &self->amd) != MTR_OK) {
		fprintf (stderr, "meters_amd: %s: %s\n", name, mtr_last_error ());
		free (self);
		return NULL;
	}
	return self;
}

void intstat_connect_port (LV2_Handle h, uint32_t port, void* data)
{
	IntStat* self = (IntStat*) h;
	switch (port) {
	case IS_INPUT0:  self->input[0] = (float*) data; break;
	case IS_OUTPUT0: self->output[0] = (float*) data; break;
	case IS_INPUT1:  self->input[1] = (float*) data; break;
	case IS_OUTPUT1: self->output[1] = (float*) data; break;
	case IS_NOTIFY:  self->notify = (LV2_Atom_Sequence*) data; break;
	case IS_CONTROL: self->control = (const LV2_Atom_Sequence*) data; break;
	default: break;
	}
}

void intstat_cleanup (LV2_Handle h)
{
	IntStat* self = (IntStat*) h;
	if (self->amd) mtr_engine_destroy (self->amd);
	free (self);
}

/* the data-acquisition guard both plugins share (src/bitmeter.c:246-259, src/sigdistlv2.c:285-296): the
 * tables are int32, so counting stops for good at 2^31 - 1 samples */
static void acquire (IntStat* self, uint32_t n_samples)
{
	if (!self->integrating || self->integration_time >= 2147483647) return;
	if (self->integration_time > 2147483647 - (int64_t) n_samples) {
		self->integration_time = 2147483647;
		return;
	}
	const float* in[1] = { self->input[0] };
	if (n_samples > 0) mtr_engine_process_planar_host (self->amd, in, n_samples);
	self->integration_time += n_samples;
}

static void forward_audio (IntStat* self, uint32_t n_samples)
{
	if (self->input[0] != self->output[0]) memcpy (self->output[0], self->input[0], sizeof (float) * n_samples);
	if (self->chn > 1 && self->input[1] != self->output[1]) memcpy (self->output[1], self->input[1], sizeof (float) * n_samples);
}

/* ======================================================================================
 * bitmeter
 * ====================================================================================== */

/* bim_clear, src/bitmeter.c:47-55: the table, min / max, zero / pos and the sample count start over; the
 * special-value counters do not (they last until bim_reset) */
static void bim_clear (IntStat* self, const int32_t* counters)
{
	if (counters) { self->nan_base += counters[2]; self->inf_base += counters[3]; self->den_base += counters[4]; }
	mtr_engine_intstat_reset (self->amd);
	self->integration_time = 0;
}
static void bim_reset (IntStat* self)                          /* :57-60 */
{
	bim_clear (self, NULL);
	self->nan_base = self->inf_base = self->den_base = 0;
}

LV2_Handle bim_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features)
{
	(void) path;
	if (strcmp (d->URI, MTR_URI "bitmeter")) return NULL;
	IntStat* self = common_instantiate (features, rate, MTR_METER_BITSTATS, "Bitmeter");
	if (!self) return NULL;
	self->integrating = 1;                                   /* :150 */
	return self;
}

void bim_run (LV2_Handle h, uint32_t n_samples)
{
	IntStat* self = (IntStat*) h;
	Forge* const fg = &self->fg;
	forge_begin (fg, self->notify, &self->u.f);

	if (self->send_state_to_ui && self->ui_active) {          /* :190-193 */
		self->send_state_to_ui = 0;
		kv_message (fg, CTL_SAMPLERATE, (float) self->rate);
	}
	if (self->control) {                                      /* :195-235 */
		FORGE_FOREACH_OBJECT (self->control, &self->u.f, obj) {
			if (obj->body.otype == self->u.f.mtr_meters_on) {
				self->ui_active = 1;
				self->send_state_to_ui = 1;
			} else if (obj->body.otype == self->u.f.mtr_meters_off) {
				self->ui_active = 0;
			} else if (obj->body.otype == self->u.f.mtr_meters_cfg) {
				int k; float v;
				get_cc_key_value (&self->u.f, obj, &k, &v);
				switch (k) {
				case CTL_START:    self->integrating = 1; break;
				case CTL_PAUSE:    self->integrating = 0; break;
				case CTL_RESET:    bim_reset (self); self->send_state_to_ui = 1; break;
				case CTL_AVERAGE:  self->bim_average = 1; break;
				case CTL_WINDOWED: self->bim_average = 0; break;
				default: break;
				}
			}
		}
	}

	acquire (self, n_samples);

	const int fps_limit = (int) (n_samples * ceil (self->rate / (5.f * n_samples)));   /* ~ 5 fps, :261 */
	self->radar_resync += (int) n_samples;
	if (fps_limit > 0 && (self->radar_resync >= fps_limit || self->send_state_to_ui)) {
		int32_t hist[BIM_LAST], counters[5];
		float minmax[2];
		const int have = mtr_engine_bitstats (self->amd, 0, 1, hist, counters, minmax) == MTR_OK;
		if (have && self->ui_active && (self->integrating || self->send_state_to_ui)) {
			ObjFrame fr;
			if (obj_begin (fg, &fr, self->u.bim_stats)) {      /* :266-292 */
				prop_l (fg, self->u.ebu_integr_time, self->integration_time);
				prop_i (fg, self->u.bim_zero, counters[0]);
				prop_i (fg, self->u.bim_pos, counters[1]);
				prop_d (fg, self->u.bim_max, (double) minmax[1]);
				prop_d (fg, self->u.bim_min, (double) minmax[0]);
				prop_i (fg, self->u.bim_nan, self->nan_base + counters[2]);
				prop_i (fg, self->u.bim_inf, self->inf_base + counters[3]);
				prop_i (fg, self->u.bim_den, self->den_base + counters[4]);
				prop_vec_i32 (fg, self->u.bim_data, hist, BIM_LAST);
				obj_end (fg, &fr);
			}
		}
		if (self->radar_resync >= fps_limit) {
			self->radar_resync = self->radar_resync % fps_limit;
			if (self->ui_active) {                            /* :315-324 */
				ObjFrame fr;
				if (obj_begin (fg, &fr, self->u.bim_information)) {
					prop_b (fg, self->u.ebu_integrating, self->integrating);
					prop_b (fg, self->u.bim_averaging, self->bim_average);
					obj_end (fg, &fr);
				}
			}
			if (!self->bim_average) bim_clear (self, have ? counters : NULL);   /* :326-328 */
		}
	}
	forward_audio (self, n_samples);
}

static LV2_State_Status bim_save (LV2_Handle h, LV2_State_Store_Function store, LV2_State_Handle handle,
                                  uint32_t flags, const LV2_Feature* const* features)
{
	(void) flags; (void) features;
	IntStat* self = (IntStat*) h;
	uint32_t cfg = self->bim_average ? 1 : 0;
	store (handle, self->u.bim_state, (void*) &cfg, sizeof (uint32_t), self->u.f.atom_Int, LV2_STATE_IS_POD | LV2_STATE_IS_PORTABLE);
	return LV2_STATE_SUCCESS;
}
static LV2_State_Status bim_restore (LV2_Handle h, LV2_State_Retrieve_Function retrieve, LV2_State_Handle handle,
                                     uint32_t flags, const LV2_Feature* const* features)
{
	(void) flags; (void) features;
	IntStat* self = (IntStat*) h;
	size_t size; uint32_t type, valflags;
	const void* value = retrieve (handle, self->u.bim_state, &size, &type, &valflags);
	if (value && size == sizeof (uint32_t) && type == self->u.f.atom_Int) {
		const uint32_t cfg = *((const uint32_t*) value);
		self->bim_average = (cfg & 0x1) ? 1 : 0;
		self->send_state_to_ui = 1;
	}
	return LV2_STATE_SUCCESS;
}
const void* bim_extension_data (const char* uri)
{
	static const LV2_State_Interface state = { bim_save, bim_restore };
	if (!strcmp (uri, LV2_STATE__interface)) return &state;
	return NULL;
}

/* ======================================================================================
 * SigDistHist
 * ====================================================================================== */

static void sdh_reset (IntStat* self)                          /* src/sigdistlv2.c:49-61 */
{
	kv_message (&self->fg, CTL_LV2_RESETRADAR, 0);
	mtr_engine_intstat_reset (self->amd);
	self->integration_time = 0;
	self->radar_resync = 0;
}
static void sdh_integrate (IntStat* self, int on)              /* :63-73 */
{
	if (self->integrating == on) return;
	if (on && (self->follow_transport_mode & 2)) sdh_reset (self);
	self->integrating = on;
}
static void sdh_update_position (IntStat* self, const LV2_Atom_Object* obj)   /* :79-101 */
{
	const LV2_Atom* speed = object_get (obj, self->u.f.time_speed);
	if (speed && speed->type == self->u.f.atom_Float) {
		const float ts = ((const LV2_Atom_Float*) speed)->body;
		if (ts != 0 && !self->tranport_rolling) { if (self->follow_transport_mode & 1) sdh_integrate (self, 1); }
		if (ts == 0 && self->tranport_rolling)  { if (self->follow_transport_mode & 1) sdh_integrate (self, 0); }
		self->tranport_rolling = (ts != 0);
	}
}

LV2_Handle sdh_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features)
{
	(void) path;
	if (strcmp (d->URI, MTR_URI "SigDistHist")) return NULL;
	return common_instantiate (features, rate, MTR_METER_SIGDIST, "SigDistHist");
}

void sdh_run (LV2_Handle h, uint32_t n_samples)
{
	IntStat* self = (IntStat*) h;
	Forge* const fg = &self->fg;
	forge_begin (fg, self->notify, &self->u.f);

	if (self->send_state_to_ui && self->ui_active) {          /* :213-218 */
		self->send_state_to_ui = 0;
		kv_message (fg, CTL_LV2_FTM, (float) self->follow_transport_mode);
		kv_message (fg, CTL_SAMPLERATE, (float) self->rate);
		kv_message (fg, CTL_UISETTINGS, (float) self->ui_settings);
	}
	if (self->control) {                                      /* :220-274 */
		FORGE_FOREACH_OBJECT (self->control, &self->u.f, obj) {
			if (obj->body.otype == self->u.f.time_Position) {
				sdh_update_position (self, obj);
			} else if (obj->body.otype == self->u.f.mtr_meters_on) {
				self->ui_active = 1;
				self->send_state_to_ui = 1;
			} else if (obj->body.otype == self->u.f.mtr_meters_off) {
				self->ui_active = 0;
			} else if (obj->body.otype == self->u.f.mtr_meters_cfg) {
				int k; float v;
				get_cc_key_value (&self->u.f, obj, &k, &v);
				switch (k) {
				case CTL_START: sdh_integrate (self, 1); break;
				case CTL_PAUSE: sdh_integrate (self, 0); break;
				case CTL_RESET: sdh_reset (self); break;
				case CTL_TRANSPORTSYNC:
					if (v == 1) {
						self->follow_transport_mode |= 1;
						if (self->tranport_rolling != self->integrating) sdh_integrate (self, self->tranport_rolling);
					} else {
						self->follow_transport_mode &= ~1;
					}
					break;
				case CTL_AUTORESET:
					if (v == 1) self->follow_transport_mode |= 2; else self->follow_transport_mode &= ~2;
					break;
				case CTL_UISETTINGS: self->ui_settings = (uint32_t) v; break;
				default: break;
				}
			}
		}
	}

	acquire (self, n_samples);

	const float fps = (float) (self->rate / 25.f);                                      /* :330 */
	const int fps_limit = (int) (fps > (float) n_samples ? fps : (float) n_samples);
	self->radar_resync += (int) n_samples;
	if (fps_limit > 0 && (self->radar_resync >= fps_limit || self->send_state_to_ui)) {
		self->radar_resync = self->radar_resync % fps_limit;
		if (self->ui_active && (self->integrating || self->send_state_to_ui)) {
			int32_t bins[DIST_BIN], peak[2];
			double mom[3];
			int64_t n;
			if (mtr_engine_sigdist (self->amd, 0, 1, bins, peak, mom, &n) == MTR_OK) {
				ObjFrame fr;
				if (obj_begin (fg, &fr, self->u.sdh_histogram)) {   /* :339-358 */
					prop_i (fg, self->u.sdh_hist_max, peak[0]);
					prop_d (fg, self->u.sdh_hist_avg, mom[0]);
					prop_d (fg, self->u.sdh_hist_var, mom[2]);
					prop_i (fg, self->u.sdh_hist_peak, peak[1]);
					prop_vec_i32 (fg, self->u.sdh_hist_data, bins, DIST_BIN);
					obj_end (fg, &fr);
				}
			}
		}
		if (self->ui_active) {                                /* :360-370 */
			ObjFrame fr;
			if (obj_begin (fg, &fr, self->u.sdh_information)) {
				prop_b (fg, self->u.ebu_integrating, self->integrating);
				prop_l (fg, self->u.ebu_integr_time, self->integration_time);
				obj_end (fg, &fr);
			}
		}
	}
	forward_audio (self, n_samples);
}

static LV2_State_Status sdh_save (LV2_Handle h, LV2_State_Store_Function store, LV2_State_Handle handle,
                                  uint32_t flags, const LV2_Feature* const* features)
{
	(void) flags; (void) features;
	IntStat* self = (IntStat*) h;
	uint32_t cfg = self->ui_settings;
	cfg |= (uint32_t) self->follow_transport_mode << 8;
	store (handle, self->u.sdh_state, (void*) &cfg, sizeof (uint32_t), self->u.f.atom_Int, LV2_STATE_IS_POD | LV2_STATE_IS_PORTABLE);
	return LV2_STATE_SUCCESS;
}
static LV2_State_Status sdh_restore (LV2_Handle h, LV2_State_Retrieve_Function retrieve, LV2_State_Handle handle,
                                     uint32_t flags, const LV2_Feature* const* features)
{
	(void) flags; (void) features;
	IntStat* self = (IntStat*) h;
	size_t size; uint32_t type, valflags;
	const void* value = retrieve (handle, self->u.sdh_state, &size, &type, &valflags);
	if (value && size == sizeof (uint32_t) && type == self->u.f.atom_Int) {
		const uint32_t cfg = *((const uint32_t*) value);
		self->ui_settings = cfg & 0xff;
		self->follow_transport_mode = (cfg >> 8) & 0x3;
		self->send_state_to_ui = 1;
	}
	return LV2_STATE_SUCCESS;
}
const void* sdh_extension_data (const char* uri)
{
	static const LV2_State_Interface state = { sdh_save, sdh_restore };
	if (!strcmp (uri, LV2_STATE__interface)) return &state;
	return NULL;
}
