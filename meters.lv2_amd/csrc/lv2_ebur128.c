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
	int transport_mode, rolling;             /* (sic) the reference's spelling */
	int integrating, dbtp_enable;
	int failing;               /* an engine call failed in the last run (): lv2_engine_ok */
	uint32_t ui_flags;

	float *ring_s, *ring_m;
	float acc_s, acc_m;
	int ring_pos, ring_len;
	uint32_t ring_acc, ring_span;
	int resend_at;
	uint64_t integ_frames;
	int32_t sent_m[HIST_LEN], sent_s[HIST_LEN];
	int32_t hmax_m, hmax_s;
	float peak_db;

	mtr_engine* amd;
	int32_t devM[HIST_LEN], devS[HIST_LEN];                  /* the engine's histograms, fetched per cycle */
} Ebu;

/* ---- helpers of the reference, src/ebulv2.cc:44-112 ---------------------------------------------- */
static void ebu_reset (Ebu* p)
{
	mtr_engine_integr_reset (p->amd);
	/* Not in the reference's ebu_reset: its TruePeakdsp objects are never reset, but their read() returns
	 * the peak since the previous read, so peak_db = -inf below forgets the past there too. */
	mtr_engine_truepeak_reset (p->amd);
	kv_message (&p->fg, CTL_LV2_RESETRADAR, 0);
	for (int i = 0; i < p->ring_len; ++i) { p->ring_s[i] = -INFINITY; p->ring_m[i] = -INFINITY; }
	memset (p->sent_m, 0, sizeof (p->sent_m));
	memset (p->sent_s, 0, sizeof (p->sent_s));
	p->ring_pos = 0;
	p->integ_frames = 0;
	p->hmax_m = 0;
	p->hmax_s = 0;
	p->peak_db = -INFINITY;
}

static void ebu_integrate (Ebu* p, int on)
{
	if (p->integrating == on) return;
	if (on) {
		if (p->transport_mode & 2) ebu_reset (p);
		mtr_engine_integr_start (p->amd);
		p->integrating = 1;
	} else {
		mtr_engine_integr_pause (p->amd);
		p->integrating = 0;
	}
}

static void ebu_set_radarspeed (Ebu* p, float seconds)
{
	p->ring_span = (uint32_t) rint (seconds * p->rate / p->ring_len);
	if (p->ring_span < 4096) p->ring_span = 4096;
}

static void update_position (Ebu* p, const LV2_Atom_Object* obj)
{
	const LV2_Atom* speed = object_get (obj, p->u.f.time_speed);
	if (speed && speed->type == p->u.f.atom_Float) {
		const float ts = ((const LV2_Atom_Float*) speed)->body;
		if (ts != 0 && !p->rolling) { if (p->transport_mode & 1) ebu_integrate (p, 1); }
		if (ts == 0 && p->rolling)  { if (p->transport_mode & 1) ebu_integrate (p, 0); }
		p->rolling = (ts != 0);
	}
}

/* ---- LV2 callbacks ----------------------------------------------------------------------------- */
LV2_Handle ebur128_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features)
{
	(void) path;
	if (strcmp (d->URI, MTR_URI "EBUr128")) return NULL;
	Ebu* p = (Ebu*) calloc (1, sizeof (Ebu));
	if (!p) return NULL;
	for (int i = 0; features && features[i]; ++i)
		if (!strcmp (features[i]->URI, LV2_URID__map)) p->map = (LV2_URID_Map*) features[i]->data;
	if (!p->map) {                                        /* src/ebulv2.cc:140-144 */
		fprintf (stderr, "EBUrLV2 error: Host does not support urid:map\n");
		free (p);
		return NULL;
	}
#define MAP(field, uri) p->u.field = p->map->map (p->map->handle, uri)
	forge_map_urids (p->map, &p->u.f);
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
	p->rate = rate;
	p->ring_len = 360;
	p->resend_at = -1;
	p->ui_flags = 8;
	p->ring_s = (float*) malloc (p->ring_len * sizeof (float));
	p->ring_m = (float*) malloc (p->ring_len * sizeof (float));
	if (!p->ring_s || !p->ring_m) { free (p->ring_s); free (p->ring_m); free (p); return NULL; }
	p->acc_s = p->acc_m = -INFINITY;
	for (int i = 0; i < p->ring_len; ++i) { p->ring_s[i] = -INFINITY; p->ring_m[i] = -INFINITY; }
	ebu_set_radarspeed (p, 2.0f * 60.0f);
	p->peak_db = -INFINITY;

	mtr_config cfg;
	memset (&cfg, 0, sizeof (cfg));
	cfg.struct_size = sizeof (cfg);
	cfg.meters = MTR_METER_EBU | MTR_METER_TRUEPEAK;
	cfg.n_streams = 1;
	cfg.n_channels = 2;
	cfg.sample_rate = (float) rate;
	if (lv2_engine_open (&cfg, features, &p->amd) != MTR_OK) {
		fprintf (stderr, "meters_amd: EBUr128: %s\n", mtr_last_error ());
		free (p->ring_s); free (p->ring_m); free (p);
		return NULL;
	}
	return p;
}

void ebur128_connect_port (LV2_Handle h, uint32_t port, void* data)
{
	Ebu* p = (Ebu*) h;
	switch (port) {
	case EBU_INPUT0:  p->input[0] = (float*) data; break;
	case EBU_OUTPUT0: p->output[0] = (float*) data; break;
	case EBU_INPUT1:  p->input[1] = (float*) data; break;
	case EBU_OUTPUT1: p->output[1] = (float*) data; break;
	case EBU_NOTIFY:  p->notify = (LV2_Atom_Sequence*) data; break;
	case EBU_CONTROL: p->control = (const LV2_Atom_Sequence*) data; break;
	default: break;
	}
}

static void radar_point (Ebu* p, float m, float s, int pos)
{
	ObjFrame fr;
	if (!obj_begin (&p->fg, &fr, p->u.rdr_radarpoint)) return;
	prop_f (&p->fg, p->u.ebu_loudnessM, m);
	prop_f (&p->fg, p->u.ebu_loudnessS, s);
	prop_i (&p->fg, p->u.rdr_pointpos, pos);
	prop_i (&p->fg, p->u.rdr_pos_cur, p->ring_pos);
	prop_i (&p->fg, p->u.rdr_pos_max, p->ring_len);
	obj_end (&p->fg, &fr);
}

void ebur128_run (LV2_Handle h, uint32_t n_samples)
{
	Ebu* p = (Ebu*) h;
	const uint32_t capacity = p->notify->atom.size;
	Forge* const fg = &p->fg;
	forge_begin (fg, p->notify, &p->u.f);

	if (p->send_state_to_ui && p->ui_active) {          /* :248-255 */
		p->send_state_to_ui = 0;
		kv_message (fg, CTL_LV2_FTM, (float) p->transport_mode);
		kv_message (fg, CTL_LV2_RADARTIME, (float) (p->ring_len * p->ring_span / p->rate));
		kv_message (fg, CTL_UISETTINGS, (float) p->ui_flags);
	}

	/* incoming events, :257-331 */
	if (p->control) {
		FORGE_FOREACH_OBJECT (p->control, &p->u.f, obj) {
			if (obj->body.otype == p->u.f.time_Position) {
				update_position (p, obj);
			} else if (obj->body.otype == p->u.f.mtr_meters_on) {
				p->ui_active = 1;
				p->send_state_to_ui = 1;
				p->resend_at = 0;
				memset (p->sent_m, 0, sizeof (p->sent_m));          /* resync histogram */
				memset (p->sent_s, 0, sizeof (p->sent_s));
				p->hmax_m = 0;
				p->hmax_s = 0;
			} else if (obj->body.otype == p->u.f.mtr_meters_off) {
				p->ui_active = 0;
			} else if (obj->body.otype == p->u.f.mtr_meters_cfg) {
				int key; float val;
				get_cc_key_value (&p->u.f, obj, &key, &val);
				switch (key) {
				case CTL_START: ebu_integrate (p, 1); break;
				case CTL_PAUSE: ebu_integrate (p, 0); break;
				case CTL_RESET: ebu_reset (p); break;
				case CTL_TRANSPORTSYNC:
					if (val == 1) {
						p->transport_mode |= 1;
						if (p->rolling != p->integrating) ebu_integrate (p, p->rolling);
					} else {
						p->transport_mode &= ~1;
					}
					break;
				case CTL_AUTORESET:
					if (val == 1) p->transport_mode |= 2; else p->transport_mode &= ~2;
					break;
				case CTL_RADARTIME:
					if (val >= 30 && val <= 600) {
						ebu_set_radarspeed (p, val);
						if (p->ring_span < 2 * n_samples) p->ring_span = 2 * n_samples;
					}
					kv_message (fg, CTL_LV2_RADARTIME, (float) (p->ring_len * p->ring_span / p->rate));
					break;
				case CTL_UISETTINGS:
					p->ui_flags = (uint32_t) val;
					p->dbtp_enable = (p->ui_flags & 64) ? 1 : 0;
					break;
				default: break;
				}
			}
		}
	}

	/* audio, :340-347: one batch-of-one launch; results come back in one record */
	const float* in[2] = { p->input[0], p->input[1] };
	int rc = MTR_OK;
	if (n_samples > 0) rc = mtr_engine_process_planar_host (p->amd, in, n_samples);
	mtr_stream_result r;
	memset (&r, 0, sizeof (r));
	if (rc == MTR_OK) rc = mtr_engine_results (p->amd, 0, 1, &r);
	if (!lv2_engine_ok (rc, &p->failing, "EBUr128")) {
		/* no measurement this cycle: every level of the record is NaN (lv2_plugins.h), nothing is accumulated */
		r.loudness_M = r.maxloudn_M = r.loudness_S = r.maxloudn_S = r.integrated = MTR_LV2_NO_DATA;
		r.range_min = r.range_max = r.integ_thr = r.range_thr = MTR_LV2_NO_DATA;
	}
	const float lm = r.loudness_M, ls = r.loudness_S;

	if (p->dbtp_enable) {                                  /* :360-367 */
		const float tp0 = r.truepeak_call[0], tp1 = r.truepeak_call[1];
		const float tpm = tp0 > tp1 ? tp0 : tp1;
		const float tp = tpm == 0 ? -INFINITY : (float) (20.0 * log10f (tpm));
		if (tp > p->peak_db) p->peak_db = tp;
	} else {
		p->peak_db = -INFINITY;
	}

	if (p->resend_at >= 0) {                            /* :369-388 */
		int batch = ((int) capacity - 512) / 192;
		if (batch > 16) batch = 16;
		for (int i = 0; i < batch; i++, p->resend_at++) {
			if (p->resend_at >= p->ring_len) {
				p->resend_at = -1;
				kv_message (fg, CTL_LV2_RESYNCDONE, 0);
				break;
			}
			radar_point (p, p->ring_m[p->resend_at], p->ring_s[p->resend_at], p->resend_at);
		}
	}

	/* radar history, :390-423 (the second test reads `lm` in the reference too) */
	if (lm > p->acc_m) p->acc_m = lm;
	if (lm > p->acc_s) p->acc_s = ls;
	if (p->integrating) p->integ_frames += n_samples;
	p->ring_acc += n_samples;
	if (p->ring_acc > p->ring_span) {
		if (p->ui_active) radar_point (p, p->acc_m, p->acc_s, p->ring_pos);
		p->ring_m[p->ring_pos] = p->acc_m;
		p->ring_s[p->ring_pos] = p->acc_s;
		p->ring_acc = p->ring_acc % p->ring_span;
		p->ring_pos = (p->ring_pos + 1) % p->ring_len;
		p->acc_s = p->acc_m = -INFINITY;
	}

	if (p->ui_active) {                                    /* histogram diffs, :425-462 */
		int msgtx = 0;
		if (r.hist_M_count > 10 && r.hist_S_count > 10
		    && mtr_engine_histograms (p->amd, 0, 1, p->devM, p->devS) == MTR_OK) {
			int max_changed = 0;
			for (int i = 110; i < 650; i++) {
				const int32_t vm = p->devM[i], vs = p->devS[i];
				if (capacity - forge_used (fg) <= 512) break;
				if (p->sent_m[i] != vm || p->sent_s[i] != vs) {
					if (msgtx++ > 16) break;                  /* limit max data-rate */
					p->sent_m[i] = vm;
					p->sent_s[i] = vs;
					ObjFrame fr;
					if (obj_begin (fg, &fr, p->u.rdr_histpoint)) {
						prop_i (fg, p->u.ebu_loudnessM, vm);
						prop_i (fg, p->u.ebu_loudnessS, vs);
						prop_i (fg, p->u.rdr_pointpos, i);
						obj_end (fg, &fr);
					}
				}
				if (vm > p->hmax_m) { p->hmax_m = vm; max_changed = 1; }
				if (vs > p->hmax_s) { p->hmax_s = vs; max_changed = 1; }
			}
			if (max_changed) {
				ObjFrame fr;
				if (obj_begin (fg, &fr, p->u.rdr_histogram)) {
					prop_i (fg, p->u.ebu_loudnessM, p->hmax_m);
					prop_i (fg, p->u.ebu_loudnessS, p->hmax_s);
					obj_end (fg, &fr);
				}
			}
		}
	}

	if (p->ui_active) {                                    /* `ebulevels`, :464-482 */
		ObjFrame fr;
		if (obj_begin (fg, &fr, p->u.mtr_ebulevels)) {
			prop_f (fg, p->u.ebu_loudnessM, lm);
			prop_f (fg, p->u.ebu_maxloudnM, r.maxloudn_M);
			prop_f (fg, p->u.ebu_loudnessS, ls);
			prop_f (fg, p->u.ebu_maxloudnS, r.maxloudn_S);
			prop_f (fg, p->u.ebu_integrated, r.integrated);
			prop_f (fg, p->u.ebu_range_min, r.range_min);
			prop_f (fg, p->u.ebu_range_max, r.range_max);
			prop_f (fg, p->u.mtr_truepeak, p->peak_db);
			prop_b (fg, p->u.ebu_integrating, p->integrating);
			prop_f (fg, p->u.ebu_integr_time, (float) (p->integ_frames / p->rate));
			obj_end (fg, &fr);
		}
	}

	for (int c = 0; c < 2; ++c)
		if (p->input[c] != p->output[c]) memcpy (p->output[c], p->input[c], sizeof (float) * n_samples);
}

void ebur128_cleanup (LV2_Handle h)
{
	Ebu* p = (Ebu*) h;
	if (p->amd) mtr_engine_destroy (p->amd);
	free (p->ring_s);
	free (p->ring_m);
	free (p);
}

/* ---- LV2 State, src/ebulv2.cc:514-553: one atom:Int = ui_flags | transport_mode << 8 | ring_span << 16 */
static LV2_State_Status ebur128_save (LV2_Handle h, LV2_State_Store_Function store, LV2_State_Handle handle,
                                      uint32_t flags, const LV2_Feature* const* features)
{
	(void) flags; (void) features;
	Ebu* p = (Ebu*) h;
	uint32_t cfg = p->ui_flags;
	cfg |= (uint32_t) p->transport_mode << 8;
	cfg |= p->ring_span << 16;
	store (handle, p->u.ebu_state, (void*) &cfg, sizeof (uint32_t), p->u.f.atom_Int, LV2_STATE_IS_POD | LV2_STATE_IS_PORTABLE);
	return LV2_STATE_SUCCESS;
}

static LV2_State_Status ebur128_restore (LV2_Handle h, LV2_State_Retrieve_Function retrieve, LV2_State_Handle handle,
                                         uint32_t flags, const LV2_Feature* const* features)
{
	(void) flags; (void) features;
	Ebu* p = (Ebu*) h;
	size_t size;
	uint32_t type, valflags;
	const void* value = retrieve (handle, p->u.ebu_state, &size, &type, &valflags);
	if (value && size == sizeof (uint32_t) && type == p->u.f.atom_Int) {
		const uint32_t cfg = *((const uint32_t*) value);
		p->ui_flags = cfg & 0xff;
		p->transport_mode = (cfg >> 8) & 0x3;
		p->ring_span = cfg >> 16;
		p->dbtp_enable = (p->ui_flags & 64) ? 1 : 0;
		p->send_state_to_ui = 1;
	}
	return LV2_STATE_SUCCESS;
}

const void* ebur128_extension_data (const char* uri)
{
	static const LV2_State_Interface state = { ebur128_save, ebur128_restore };
	if (!strcmp (uri, LV2_STATE__interface)) return &state;
	return NULL;
}
