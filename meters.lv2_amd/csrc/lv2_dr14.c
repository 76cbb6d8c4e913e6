/* lv2_dr14.c — the DR-14 and "true peak + RMS" plugins of lib/meters_amd.so (src/dr14.c of the reference):
 *
 *   dr14mono / dr14stereo       crest-factor "dynamic range" over 3 s windows (top 20 % of the RMS histogram
 *                               against the second-highest window peak), plus the bars below
 *   TPnRMSmono / TPnRMSstereo   the bars only: K-meter RMS and ballistic true peak with max hold
 *
 * Per channel the reference runs a Kmeterdsp and a TruePeakdsp::process.  The true-peak meter (4x
 * interpolation + attack / release ballistics) is the engine's TPBALLIST kernel on the GPU; the K-meter
 * one-poles and the window bookkeeping (one multiply-add per sample, a histogram insert every 3 s) stay on
 * the host like the other needle ballistics.  Ports as in src/dr14.c:27-43.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lv2_min.h"
#include "mtr_engine.h"
#include "lv2_forge.h"
#include "lv2_plugins.h"
#include "lv2_dsp.h"

enum { DR_CONTROL = 0, DR_HOST_TRANSPORT, DR_RESET, DR_BLKCNT, DR_INPUT0, DR_OUTPUT0, DR_V_PEAK0, DR_M_PEAK0,
       DR_V_RMS0, DR_M_RMS0, DR_DR0, DR_INPUT1, DR_OUTPUT1, DR_V_PEAK1, DR_M_PEAK1, DR_V_RMS1, DR_M_RMS1, DR_DR1, DR_TOTAL };

#define DR_CHANNELS 2
#define DR_HISTBINS 8000                                     /* -80 dB .. 0 dB in .01 dB steps */
#define MAXF(a, b) ((a) > (b) ? (a) : (b))
#define MINF(a, b) ((a) < (b) ? (a) : (b))

typedef struct {
	const LV2_Atom_Sequence* control;
	float* p_follow_host_transport;
	float* p_reset_button;
	float* p_block_count;
	float* p_input[DR_CHANNELS];
	float* p_output[DR_CHANNELS];
	float* p_v_rms[DR_CHANNELS];
	float* p_v_peak[DR_CHANNELS];
	float* p_m_rms[DR_CHANNELS];
	float* p_m_peak[DR_CHANNELS];
	float* p_dr[DR_CHANNELS];
	float* p_dr_total;

	ForgeUrids u;
	LV2_URID mtr_dr14reset;
	uint32_t n_channels;
	double rate;
	uint64_t n_sample_cnt;
	int follow_host_transport, tranport_rolling, reinit_gui, dr_operation_mode;

	float m_dbtp[DR_CHANNELS], m_peak[DR_CHANNELS], m_rms[DR_CHANNELS];
	uint64_t sample_count, num_fragments;
	Kmeter km[DR_CHANNELS];
	float rms_sum[DR_CHANNELS], peak_cur[DR_CHANNELS], peak_hist[DR_CHANNELS][2];
	uint32_t* hist[DR_CHANNELS];
	mtr_engine* amd;
} Dr14;

static float coeff_to_db (const float coeff) { return coeff < .0001 ? -80 : 20 * log10f (coeff); }   /* :235-238 */
static float db_to_coeff (const float db) { return db <= -80 ? 0 : powf (10, 0.05 * db); }          /* :240-243 */

static void reset_peaks (Dr14* self)                         /* :245-260 */
{
	for (uint32_t c = 0; c < self->n_channels; ++c) {
		self->m_peak[c] = -81;
		self->m_rms[c] = -81;
		self->m_dbtp[c] = 0;
		self->rms_sum[c] = 0;
		self->peak_cur[c] = 0;
		self->peak_hist[c][0] = self->peak_hist[c][1] = 0;
		km_reset (&self->km[c]);
		if (self->dr_operation_mode) memset (self->hist[c], 0, DR_HISTBINS * sizeof (int32_t));
	}
	self->sample_count = 0;
	self->num_fragments = 0;
}

/* one 3 s window is complete, :283-352 */
static void calc_rms_score (Dr14* self)
{
	int silent = 1;
	for (uint32_t c = 0; c < self->n_channels; ++c)
		if (self->rms_sum[c] > 1e-9 * (float) self->n_sample_cnt) silent = 0;
	if (silent) {                                            /* silence is not added to the histogram */
		for (uint32_t c = 0; c < self->n_channels; ++c) self->rms_sum[c] = 0;
		return;
	}
	self->num_fragments++;
	const uint32_t m_cut = MAXF (1, floorf (self->num_fragments / 5.0));   /* top 20 % */
	for (uint32_t c = 0; c < self->n_channels; ++c) {
		const float rms = sqrt (2.f * self->rms_sum[c] / (float) self->n_sample_cnt);
		self->rms_sum[c] = 0;
		int bin = rintf (100.f * (80.f + coeff_to_db (rms))) - 1;
		if (bin >= DR_HISTBINS) bin = DR_HISTBINS - 1;
		if (bin > 0) self->hist[c][bin]++;

		uint32_t n_cut = 0;
		float rms_score = 0;
		if (self->num_fragments > 2) {                       /* RMS average of the top bins, via coefficients */
			for (int32_t b = DR_HISTBINS - 1; b > 0 && n_cut < m_cut; --b) {
				const uint32_t bc = self->hist[c][b];
				if (bc == 0) continue;
				const float cd = db_to_coeff ((b - DR_HISTBINS + 1) / 100.0);
				rms_score += cd * cd * (float) bc;
				n_cut += bc;
			}
		}
		self->m_rms[c] = n_cut > 0 ? coeff_to_db (sqrtf (rms_score / n_cut)) : -81;

		/* the second-highest raw peak of all windows */
		if (self->peak_cur[c] >= self->peak_hist[c][0]) {
			self->peak_hist[c][1] = self->peak_hist[c][0];
			self->peak_hist[c][0] = self->peak_cur[c];
		} else if (self->peak_cur[c] > self->peak_hist[c][1]) {
			self->peak_hist[c][1] = self->peak_cur[c];
		}
		self->peak_cur[c] = 0;
		self->m_peak[c] = self->num_fragments > 2 ? coeff_to_db (self->peak_hist[c][1]) : -81;
	}
}

LV2_Handle dr14_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features)
{
	(void) path;
	uint32_t n_channels;
	int dr_mode;
	if (!strcmp (d->URI, MTR_URI "dr14stereo"))        { n_channels = 2; dr_mode = 1; }
	else if (!strcmp (d->URI, MTR_URI "dr14mono"))     { n_channels = 1; dr_mode = 1; }
	else if (!strcmp (d->URI, MTR_URI "TPnRMSstereo")) { n_channels = 2; dr_mode = 0; }
	else if (!strcmp (d->URI, MTR_URI "TPnRMSmono"))   { n_channels = 1; dr_mode = 0; }
	else return NULL;
	LV2_URID_Map* map = NULL;
	for (int i = 0; features && features[i]; ++i)
		if (!strcmp (features[i]->URI, LV2_URID__map)) map = (LV2_URID_Map*) features[i]->data;
	if (!map) {
		fprintf (stderr, "DR14LV2 error: Host does not support urid:map\n");
		return NULL;
	}
	Dr14* self = (Dr14*) calloc (1, sizeof (Dr14));
	if (!self) return NULL;
	self->n_channels = n_channels;
	self->dr_operation_mode = dr_mode;
	self->rate = rate;
	forge_map_urids (map, &self->u);
	self->mtr_dr14reset = map->map (map->handle, MTR_URI "dr14reset");
	self->follow_host_transport = 1;
	self->n_sample_cnt = (uint64_t) rintf (rate * 3.0);
	for (uint32_t c = 0; c < n_channels; ++c) {
		km_init (&self->km[c], (float) rate);
		self->m_rms[c] = -81;
		self->m_peak[c] = -81;
		if (dr_mode) {
			self->hist[c] = (uint32_t*) calloc (DR_HISTBINS, sizeof (uint32_t));
			if (!self->hist[c]) { free (self->hist[0]); free (self); return NULL; }
		}
	}
	mtr_config cfg;
	memset (&cfg, 0, sizeof (cfg));
	cfg.struct_size = sizeof (cfg);
	cfg.meters = MTR_METER_TPBALLIST;
	cfg.n_streams = 1;
	cfg.n_channels = n_channels;
	cfg.sample_rate = (float) rate;
	if (mtr_engine_create (&cfg, &self->amd) != MTR_OK) {
		fprintf (stderr, "meters_amd: %s: %s\n", d->URI, mtr_last_error ());
		free (self->hist[0]); free (self->hist[1]); free (self);
		return NULL;
	}
	return self;
}

void dr14_connect_port (LV2_Handle h, uint32_t port, void* data)
{
	Dr14* self = (Dr14*) h;
	switch (port) {
	case DR_CONTROL:        self->control = (const LV2_Atom_Sequence*) data; break;
	case DR_HOST_TRANSPORT: self->p_follow_host_transport = (float*) data; break;
	case DR_RESET:          self->p_reset_button = (float*) data; break;
	case DR_BLKCNT:         self->p_block_count = (float*) data; break;
	case DR_INPUT0:  self->p_input[0] = (float*) data; break;
	case DR_OUTPUT0: self->p_output[0] = (float*) data; break;
	case DR_V_RMS0:  self->p_v_rms[0] = (float*) data; break;
	case DR_M_RMS0:  self->p_m_rms[0] = (float*) data; break;
	case DR_V_PEAK0: self->p_v_peak[0] = (float*) data; break;
	case DR_M_PEAK0: self->p_m_peak[0] = (float*) data; break;
	case DR_DR0:     self->p_dr[0] = (float*) data; break;
	case DR_TOTAL:   self->p_dr_total = (float*) data; break;
	case DR_INPUT1:  self->p_input[1] = (float*) data; break;
	case DR_OUTPUT1: self->p_output[1] = (float*) data; break;
	case DR_V_RMS1:  self->p_v_rms[1] = (float*) data; break;
	case DR_M_RMS1:  self->p_m_rms[1] = (float*) data; break;
	case DR_V_PEAK1: self->p_v_peak[1] = (float*) data; break;
	case DR_M_PEAK1: self->p_m_peak[1] = (float*) data; break;
	case DR_DR1:     self->p_dr[1] = (float*) data; break;
	default: break;
	}
}

void dr14_run (LV2_Handle h, uint32_t n_samples)             /* :354-486 */
{
	Dr14* self = (Dr14*) h;
	self->follow_host_transport = (*self->p_follow_host_transport != 0);

	if (self->control) {                                      /* reset from the GUI, transport from the host */
		FORGE_FOREACH_OBJECT (self->control, &self->u, obj) {
			if (obj->body.otype == self->u.time_Position) {   /* parse_time_position :262-281 */
				const LV2_Atom* speed = object_get (obj, self->u.time_speed);
				if (speed && speed->type == self->u.atom_Float) {
					const float ts = ((const LV2_Atom_Float*) speed)->body;
					if (ts != 0 && !self->tranport_rolling && self->follow_host_transport) reset_peaks (self);
					self->tranport_rolling = (ts != 0);
				}
			}
			if (obj->body.otype == self->mtr_dr14reset) reset_peaks (self);
			if (obj->body.otype == self->u.mtr_meters_on)  self->reinit_gui = 1;
			if (obj->body.otype == self->u.mtr_meters_off) self->reinit_gui = 0;
		}
	}
	if (*self->p_reset_button != 0) reset_peaks (self);

	/* RMS bar (host), true-peak ballistics (GPU): Kmeterdsp::process + TruePeakdsp::process, :385-388 */
	for (uint32_t c = 0; c < self->n_channels; ++c) km_process (&self->km[c], self->p_input[c], (int) n_samples);
	const float* in[2] = { self->p_input[0], self->p_input[1] };
	mtr_stream_result r;
	memset (&r, 0, sizeof (r));
	if (n_samples > 0) {
		mtr_engine_process_planar_host (self->amd, in, n_samples);
		mtr_engine_results (self->amd, 0, 1, &r);
	}

	/* 3 s non-overlapping windows, :394-410 */
	if (self->dr_operation_mode) {
		uint64_t scnt = self->sample_count;
		const uint64_t slmt = self->n_sample_cnt;
		for (uint32_t s = 0; s < n_samples; ++s) {
			for (uint32_t c = 0; c < self->n_channels; ++c) {
				const float v = self->p_input[c][s];
				self->rms_sum[c] += v * v;
				self->peak_cur[c] = MAXF (self->peak_cur[c], v);    /* the signed sample, as the reference */
			}
			if (++scnt > slmt) {
				calc_rms_score (self);
				scnt = 0;
			}
		}
		self->sample_count = scnt;
	}

	/* values to ports, :413-451 */
	float dr_total = 0;
	int dr_valid = 0;
	for (uint32_t c = 0; c < self->n_channels; ++c) {
		float rv, rp;
		const float pv = r.tpb_level[c], pp = r.tpb_peak[c];  /* TruePeakdsp::read (pv, pp) */
		km_read (&self->km[c], &rv, &rp);
		self->m_dbtp[c] = MAXF (self->m_dbtp[c], pp);
		*self->p_v_rms[c]  = coeff_to_db (rv);
		*self->p_v_peak[c] = coeff_to_db (pv);
		*self->p_m_peak[c] = coeff_to_db (self->m_dbtp[c]);
		if (self->dr_operation_mode) {
			const float rdb = self->m_rms[c];
			const float pdb = self->m_peak[c];
			const float dr = MINF (0, pdb) - rdb;
			if (rdb > -80 && pdb > -80) { dr_total += dr; dr_valid++; }
			*self->p_dr[c] = (rdb > -80 && pdb > -80) ? MAXF (1, MINF (20, dr)) : 21;
			*self->p_m_rms[c] = rdb;
		} else {
			*self->p_m_rms[c] = coeff_to_db (rp);
		}
	}
	if (self->n_channels > 1 && self->dr_operation_mode)
		*self->p_dr_total = dr_valid > 0 ? MAXF (1, MINF (20, dr_total / (float) dr_valid)) : 21;
	*self->p_block_count = 3.0 * self->num_fragments;

	if (self->reinit_gui) {                                   /* :455-466: markers that force a port change */
		if (self->n_channels > 1 && self->dr_operation_mode) *self->p_dr_total = 21;
		for (uint32_t c = 0; c < self->n_channels; ++c) {
			*self->p_m_peak[c] = -100;
			*self->p_m_rms[c] = -100;
			if (self->dr_operation_mode) *self->p_dr[c] = 21;
		}
		*self->p_block_count = -1 - (rand () & 0xffff);
	}
	for (uint32_t c = 0; c < self->n_channels; ++c)
		if (self->p_input[c] != self->p_output[c]) memcpy (self->p_output[c], self->p_input[c], sizeof (float) * n_samples);
}

void dr14_cleanup (LV2_Handle h)
{
	Dr14* self = (Dr14*) h;
	if (self->amd) mtr_engine_destroy (self->amd);
	free (self->hist[0]);
	free (self->hist[1]);
	free (self);
}
