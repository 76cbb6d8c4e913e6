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
	int failing;               /* an engine call failed in the last run (): lv2_engine_ok */

	int ui_active, send_state_to_ui, integrating;
	int64_t integ_frames;
	int since_notify;                                        /* samples since the last message (the reference's name) */

	/* bitmeter */
	int averaging;
	int32_t nan_base, inf_base, den_base;                    /* counted before the last windowed clear */
	/* SigDistHist */
	int transport_mode, rolling;
	uint32_t ui_flags;
} IntStat;

static IntStat* common_instantiate (const LV2_Feature* const* features, double rate, uint32_t meter, const char* name)
{
	IntStat* p = (IntStat*) calloc (1, sizeof (IntStat));
	if (!p) return NULL;
	for (int i = 0; features && features[i]; ++i)
		if (!strcmp (features[i]->URI, LV2_URID__map)) p->map = (LV2_URID_Map*) features[i]->data;
	if (!p->map) {
		fprintf (stderr, "%s error: Host does not support urid:map\n", name);
		free (p);
		return NULL;
	}
	forge_map_urids (p->map, &p->u.f);
#define MAP(field, uri) p->u.field = p->map->map (p->map->handle, MTR_URI uri)
	MAP (ebu_integrating, "ebu_integrating"); MAP (ebu_integr_time, "ebu_integr_time");
	MAP (bim_state, "bim_state"); MAP (bim_information, "bim_information"); MAP (bim_averaging, "bim_averaging");
	MAP (bim_stats, "bim_stats"); MAP (bim_data, "bim_data"); MAP (bim_zero, "bim_zero"); MAP (bim_pos, "bim_pos");
	MAP (bim_min, "bim_min"); MAP (bim_max, "bim_max"); MAP (bim_nan, "bim_nan"); MAP (bim_inf, "bim_inf"); MAP (bim_den, "bim_den");
	MAP (sdh_state, "sdh_state"); MAP (sdh_histogram, "sdh_histogram"); MAP (sdh_hist_max, "sdh_hist_max");
	MAP (sdh_hist_var, "sdh_hist_var"); MAP (sdh_hist_avg, "sdh_hist_avg"); MAP (sdh_hist_peak, "sdh_hist_peak");
	MAP (sdh_hist_data, "sdh_hist_data"); MAP (sdh_information, "sdh_information");
#undef MAP
	p->rate = rate;
	p->chn = 1;
	mtr_config cfg;
	memset (&cfg, 0, sizeof (cfg));
	cfg.struct_size = sizeof (cfg);
	cfg.meters = meter;
	cfg.n_streams = 1;
	cfg.n_channels = 1;
	cfg.sample_rate = (float) rate;
	if (lv2_engine_open (&cfg, features, &p->amd) != MTR_OK) {
		fprintf (stderr, "meters_amd: %s: %s\n", name, mtr_last_error ());
		free (p);
		return NULL;
	}
	return p;
}

void intstat_connect_port (LV2_Handle h, uint32_t port, void* data)
{
	IntStat* p = (IntStat*) h;
	switch (port) {
	case IS_INPUT0:  p->input[0] = (float*) data; break;
	case IS_OUTPUT0: p->output[0] = (float*) data; break;
	case IS_INPUT1:  p->input[1] = (float*) data; break;
	case IS_OUTPUT1: p->output[1] = (float*) data; break;
	case IS_NOTIFY:  p->notify = (LV2_Atom_Sequence*) data; break;
	case IS_CONTROL: p->control = (const LV2_Atom_Sequence*) data; break;
	default: break;
	}
}

void intstat_cleanup (LV2_Handle h)
{
	IntStat* p = (IntStat*) h;
	if (p->amd) mtr_engine_destroy (p->amd);
	free (p);
}

/* the data-acquisition guard both plugins share (src/bitmeter.c:246-259, src/sigdistlv2.c:285-296): the
 * tables are int32, so counting stops for good at 2^31 - 1 samples */
static void acquire (IntStat* p, uint32_t n_samples)
{
	if (!p->integrating || p->integ_frames >= 2147483647) return;
	if (p->integ_frames > 2147483647 - (int64_t) n_samples) {
		p->integ_frames = 2147483647;
		return;
	}
	const float* in[1] = { p->input[0] };
	/* a block the engine could not take is not counted: the tables (integer counts, sent to the UI as such) stay what
	 * they were and integration time does not advance — reported once per failure streak (lv2_plugins.h) */
	if (n_samples > 0 && !lv2_engine_ok (mtr_engine_process_planar_host (p->amd, in, n_samples), &p->failing, "bitmeter / SigDistHist")) return;
	p->integ_frames += n_samples;
}

static void forward_audio (IntStat* p, uint32_t n_samples)
{
	if (p->input[0] != p->output[0]) memcpy (p->output[0], p->input[0], sizeof (float) * n_samples);
	if (p->chn > 1 && p->input[1] != p->output[1]) memcpy (p->output[1], p->input[1], sizeof (float) * n_samples);
}

/* ======================================================================================
 * bitmeter
 * ====================================================================================== */

/* bim_clear, src/bitmeter.c:47-55: the table, min / max, zero / pos and the sample count start over; the
 * special-value counters do not (they last until bim_reset) */
static void bim_clear (IntStat* p, const int32_t* counters)
{
	if (counters) { p->nan_base += counters[2]; p->inf_base += counters[3]; p->den_base += counters[4]; }
	mtr_engine_intstat_reset (p->amd);
	p->integ_frames = 0;
}
static void bim_reset (IntStat* p)                          /* :57-60 */
{
	bim_clear (p, NULL);
	p->nan_base = p->inf_base = p->den_base = 0;
}

LV2_Handle bim_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features)
{
	(void) path;
	if (strcmp (d->URI, MTR_URI "bitmeter")) return NULL;
	IntStat* p = common_instantiate (features, rate, MTR_METER_BITSTATS, "Bitmeter");
	if (!p) return NULL;
	p->integrating = 1;                                   /* :150 */
	return p;
}

void bim_run (LV2_Handle h, uint32_t n_samples)
{
	IntStat* p = (IntStat*) h;
	Forge* const fg = &p->fg;
	forge_begin (fg, p->notify, &p->u.f);

	if (p->send_state_to_ui && p->ui_active) {          /* :190-193 */
		p->send_state_to_ui = 0;
		kv_message (fg, CTL_SAMPLERATE, (float) p->rate);
	}
	if (p->control) {                                      /* :195-235 */
		FORGE_FOREACH_OBJECT (p->control, &p->u.f, obj) {
			if (obj->body.otype == p->u.f.mtr_meters_on) {
				p->ui_active = 1;
				p->send_state_to_ui = 1;
			} else if (obj->body.otype == p->u.f.mtr_meters_off) {
				p->ui_active = 0;
			} else if (obj->body.otype == p->u.f.mtr_meters_cfg) {
				int k; float v;
				get_cc_key_value (&p->u.f, obj, &k, &v);
				switch (k) {
				case CTL_START:    p->integrating = 1; break;
				case CTL_PAUSE:    p->integrating = 0; break;
				case CTL_RESET:    bim_reset (p); p->send_state_to_ui = 1; break;
				case CTL_AVERAGE:  p->averaging = 1; break;
				case CTL_WINDOWED: p->averaging = 0; break;
				default: break;
				}
			}
		}
	}

	acquire (p, n_samples);

	const int fps_limit = (int) (n_samples * ceil (p->rate / (5.f * n_samples)));   /* ~ 5 fps, :261 */
	p->since_notify += (int) n_samples;
	if (fps_limit > 0 && (p->since_notify >= fps_limit || p->send_state_to_ui)) {
		int32_t hist[BIM_LAST], counters[5];
		float minmax[2];
		const int have = mtr_engine_bitstats (p->amd, 0, 1, hist, counters, minmax) == MTR_OK;
		if (have && p->ui_active && (p->integrating || p->send_state_to_ui)) {
			ObjFrame fr;
			if (obj_begin (fg, &fr, p->u.bim_stats)) {      /* :266-292 */
				prop_l (fg, p->u.ebu_integr_time, p->integ_frames);
				prop_i (fg, p->u.bim_zero, counters[0]);
				prop_i (fg, p->u.bim_pos, counters[1]);
				prop_d (fg, p->u.bim_max, (double) minmax[1]);
				prop_d (fg, p->u.bim_min, (double) minmax[0]);
				prop_i (fg, p->u.bim_nan, p->nan_base + counters[2]);
				prop_i (fg, p->u.bim_inf, p->inf_base + counters[3]);
				prop_i (fg, p->u.bim_den, p->den_base + counters[4]);
				prop_vec_i32 (fg, p->u.bim_data, hist, BIM_LAST);
				obj_end (fg, &fr);
			}
		}
		if (p->since_notify >= fps_limit) {
			p->since_notify = p->since_notify % fps_limit;
			if (p->ui_active) {                            /* :315-324 */
				ObjFrame fr;
				if (obj_begin (fg, &fr, p->u.bim_information)) {
					prop_b (fg, p->u.ebu_integrating, p->integrating);
					prop_b (fg, p->u.bim_averaging, p->averaging);
					obj_end (fg, &fr);
				}
			}
			if (!p->averaging) bim_clear (p, have ? counters : NULL);   /* :326-328 */
		}
	}
	forward_audio (p, n_samples);
}

static LV2_State_Status bim_save (LV2_Handle h, LV2_State_Store_Function store, LV2_State_Handle handle,
                                  uint32_t flags, const LV2_Feature* const* features)
{
	(void) flags; (void) features;
	IntStat* p = (IntStat*) h;
	uint32_t cfg = p->averaging ? 1 : 0;
	store (handle, p->u.bim_state, (void*) &cfg, sizeof (uint32_t), p->u.f.atom_Int, LV2_STATE_IS_POD | LV2_STATE_IS_PORTABLE);
	return LV2_STATE_SUCCESS;
}
static LV2_State_Status bim_restore (LV2_Handle h, LV2_State_Retrieve_Function retrieve, LV2_State_Handle handle,
                                     uint32_t flags, const LV2_Feature* const* features)
{
	(void) flags; (void) features;
	IntStat* p = (IntStat*) h;
	size_t size; uint32_t type, valflags;
	const void* value = retrieve (handle, p->u.bim_state, &size, &type, &valflags);
	if (value && size == sizeof (uint32_t) && type == p->u.f.atom_Int) {
		const uint32_t cfg = *((const uint32_t*) value);
		p->averaging = (cfg & 0x1) ? 1 : 0;
		p->send_state_to_ui = 1;
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

static void sdh_reset (IntStat* p)                          /* src/sigdistlv2.c:49-61 */
{
	kv_message (&p->fg, CTL_LV2_RESETRADAR, 0);
	mtr_engine_intstat_reset (p->amd);
	p->integ_frames = 0;
	p->since_notify = 0;
}
static void sdh_integrate (IntStat* p, int on)              /* :63-73 */
{
	if (p->integrating == on) return;
	if (on && (p->transport_mode & 2)) sdh_reset (p);
	p->integrating = on;
}
static void sdh_on_position (IntStat* p, const LV2_Atom_Object* obj)   /* :79-101 */
{
	const LV2_Atom* speed = object_get (obj, p->u.f.time_speed);
	if (speed && speed->type == p->u.f.atom_Float) {
		const float ts = ((const LV2_Atom_Float*) speed)->body;
		if (ts != 0 && !p->rolling) { if (p->transport_mode & 1) sdh_integrate (p, 1); }
		if (ts == 0 && p->rolling)  { if (p->transport_mode & 1) sdh_integrate (p, 0); }
		p->rolling = (ts != 0);
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
	IntStat* p = (IntStat*) h;
	Forge* const fg = &p->fg;
	forge_begin (fg, p->notify, &p->u.f);

	if (p->send_state_to_ui && p->ui_active) {          /* :213-218 */
		p->send_state_to_ui = 0;
		kv_message (fg, CTL_LV2_FTM, (float) p->transport_mode);
		kv_message (fg, CTL_SAMPLERATE, (float) p->rate);
		kv_message (fg, CTL_UISETTINGS, (float) p->ui_flags);
	}
	if (p->control) {                                      /* :220-274 */
		FORGE_FOREACH_OBJECT (p->control, &p->u.f, obj) {
			if (obj->body.otype == p->u.f.time_Position) {
				sdh_on_position (p, obj);
			} else if (obj->body.otype == p->u.f.mtr_meters_on) {
				p->ui_active = 1;
				p->send_state_to_ui = 1;
			} else if (obj->body.otype == p->u.f.mtr_meters_off) {
				p->ui_active = 0;
			} else if (obj->body.otype == p->u.f.mtr_meters_cfg) {
				int k; float v;
				get_cc_key_value (&p->u.f, obj, &k, &v);
				switch (k) {
				case CTL_START: sdh_integrate (p, 1); break;
				case CTL_PAUSE: sdh_integrate (p, 0); break;
				case CTL_RESET: sdh_reset (p); break;
				case CTL_TRANSPORTSYNC:
					if (v == 1) {
						p->transport_mode |= 1;
						if (p->rolling != p->integrating) sdh_integrate (p, p->rolling);
					} else {
						p->transport_mode &= ~1;
					}
					break;
				case CTL_AUTORESET:
					if (v == 1) p->transport_mode |= 2; else p->transport_mode &= ~2;
					break;
				case CTL_UISETTINGS: p->ui_flags = (uint32_t) v; break;
				default: break;
				}
			}
		}
	}

	acquire (p, n_samples);

	const float fps = (float) (p->rate / 25.f);                                      /* :330 */
	const int fps_limit = (int) (fps > (float) n_samples ? fps : (float) n_samples);
	p->since_notify += (int) n_samples;
	if (fps_limit > 0 && (p->since_notify >= fps_limit || p->send_state_to_ui)) {
		p->since_notify = p->since_notify % fps_limit;
		if (p->ui_active && (p->integrating || p->send_state_to_ui)) {
			int32_t bins[DIST_BIN], peak[2];
			double mom[3];
			int64_t n;
			if (mtr_engine_sigdist (p->amd, 0, 1, bins, peak, mom, &n) == MTR_OK) {
				ObjFrame fr;
				if (obj_begin (fg, &fr, p->u.sdh_histogram)) {   /* :339-358 */
					prop_i (fg, p->u.sdh_hist_max, peak[0]);
					prop_d (fg, p->u.sdh_hist_avg, mom[0]);
					prop_d (fg, p->u.sdh_hist_var, mom[2]);
					prop_i (fg, p->u.sdh_hist_peak, peak[1]);
					prop_vec_i32 (fg, p->u.sdh_hist_data, bins, DIST_BIN);
					obj_end (fg, &fr);
				}
			}
		}
		if (p->ui_active) {                                /* :360-370 */
			ObjFrame fr;
			if (obj_begin (fg, &fr, p->u.sdh_information)) {
				prop_b (fg, p->u.ebu_integrating, p->integrating);
				prop_l (fg, p->u.ebu_integr_time, p->integ_frames);
				obj_end (fg, &fr);
			}
		}
	}
	forward_audio (p, n_samples);
}

static LV2_State_Status sdh_save (LV2_Handle h, LV2_State_Store_Function store, LV2_State_Handle handle,
                                  uint32_t flags, const LV2_Feature* const* features)
{
	(void) flags; (void) features;
	IntStat* p = (IntStat*) h;
	uint32_t cfg = p->ui_flags;
	cfg |= (uint32_t) p->transport_mode << 8;
	store (handle, p->u.sdh_state, (void*) &cfg, sizeof (uint32_t), p->u.f.atom_Int, LV2_STATE_IS_POD | LV2_STATE_IS_PORTABLE);
	return LV2_STATE_SUCCESS;
}
static LV2_State_Status sdh_restore (LV2_Handle h, LV2_State_Retrieve_Function retrieve, LV2_State_Handle handle,
                                     uint32_t flags, const LV2_Feature* const* features)
{
	(void) flags; (void) features;
	IntStat* p = (IntStat*) h;
	size_t size; uint32_t type, valflags;
	const void* value = retrieve (handle, p->u.sdh_state, &size, &type, &valflags);
	if (value && size == sizeof (uint32_t) && type == p->u.f.atom_Int) {
		const uint32_t cfg = *((const uint32_t*) value);
		p->ui_flags = cfg & 0xff;
		p->transport_mode = (cfg >> 8) & 0x3;
		p->send_state_to_ui = 1;
	}
	return LV2_STATE_SUCCESS;
}
const void* sdh_extension_data (const char* uri)
{
	static const LV2_State_Interface state = { sdh_save, sdh_restore };
	if (!strcmp (uri, LV2_STATE__interface)) return &state;
	return NULL;
}
