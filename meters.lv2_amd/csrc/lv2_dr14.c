/* lv2_dr14.c — the dr14mono / dr14stereo and TPnRMSmono / TPnRMSstereo plugins of lib/meters_amd.so.
 *
 * A thin client of the batch engine: every number these plugins put on a port is computed on the GPU by
 * an n_streams = 1 engine that carries three meters,
 *
 *   MTR_METER_TPBALLIST  TruePeakdsp::process, the ballistic true-peak bar      (reference: src/dr14.c:388, 421)
 *   MTR_METER_KMETER     Kmeterdsp, the RMS bar and the held digital peak       (src/dr14.c:386, 422)
 *   MTR_METER_DR14       3 s window sums, the 0.01 dB histogram, the top-20 %   (src/dr14.c:283-352, 394-411;
 *                        score, the second-highest window peak, DR per channel   dr14 plugins only)
 *
 * and what is left here is the LV2 surface: the port map of lv2ttl/dr14*.ttl.in, the reset paths (GUI
 * message, control port, transport start — src/dr14.c:262-281, 368-383), the true-peak maximum the host
 * sees on the m_peak ports, and the "GUI attached" marker values (src/dr14.c:455-466).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lv2_min.h"
#include "mtr_engine.h"
#include "lv2_forge.h"
#include "lv2_plugins.h"

/* port indices, lv2ttl/dr14stereo.ttl.in (mono stops after the first channel's block) */
enum {
	PORT_ATOM_IN = 0, PORT_FOLLOW_TRANSPORT, PORT_RESET, PORT_BLOCKS,
	PORT_CH0 = 4,           /* per channel: in, out, peak bar, peak max, rms bar, rms max / score, dr */
	PORT_CH_STRIDE = 7,
	PORT_DR_TOTAL = 18,
	PORT_COUNT = 19
};
enum { CH_IN = 0, CH_OUT, CH_PEAK_BAR, CH_PEAK_MAX, CH_RMS_BAR, CH_RMS_MAX, CH_DR };

typedef struct {
	void*       port[PORT_COUNT];
	mtr_engine* engine;
	ForgeUrids  urid;
	LV2_URID    urid_reset;
	uint32_t    channels;
	int         with_dr;            /* dr14* (1) or TPnRMS* (0) */
	int         rolling;            /* host transport state, from time:Position */
	int         gui_attached;
	float       tp_max[2];          /* maximum of the raw true peak since the last reset (linear) */
} DrPlugin;

static float to_db (float lin) { return lin < .0001f ? -80.f : 20.f * log10f (lin); }     /* the reference's -80 dB floor, :235-238 */

static float* fport (DrPlugin* p, uint32_t ch, int which) { return (float*) p->port[PORT_CH0 + PORT_CH_STRIDE * ch + which]; }

/* everything reset_peaks touches (:245-260) lives in the engine, except the true-peak maximum */
static void clear_meters (DrPlugin* p)
{
	p->tp_max[0] = p->tp_max[1] = 0.f;
	mtr_engine_kmeter_reset (p->engine);
	if (p->with_dr) mtr_engine_dr14_reset (p->engine);
}

LV2_Handle dr14_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features)
{
	(void) path;
	static const struct { const char* name; uint32_t channels; int with_dr; } kinds[] = {
		{ MTR_URI "dr14mono", 1, 1 }, { MTR_URI "dr14stereo", 2, 1 }, { MTR_URI "TPnRMSmono", 1, 0 }, { MTR_URI "TPnRMSstereo", 2, 0 },
	};
	int kind = -1;
	for (int i = 0; i < 4; ++i) if (!strcmp (d->URI, kinds[i].name)) kind = i;
	if (kind < 0) return NULL;

	LV2_URID_Map* map = NULL;
	for (int i = 0; features && features[i]; ++i)
		if (!strcmp (features[i]->URI, LV2_URID__map)) map = (LV2_URID_Map*) features[i]->data;
	if (!map) {
		fprintf (stderr, "DR14LV2 error: Host does not support urid:map\n");
		return NULL;
	}
	DrPlugin* p = (DrPlugin*) calloc (1, sizeof (DrPlugin));
	if (!p) return NULL;
	p->channels = kinds[kind].channels;
	p->with_dr = kinds[kind].with_dr;
	forge_map_urids (map, &p->urid);
	p->urid_reset = map->map (map->handle, MTR_URI "dr14reset");

	mtr_config cfg;
	memset (&cfg, 0, sizeof (cfg));
	cfg.struct_size = sizeof (cfg);
	cfg.meters = MTR_METER_TPBALLIST | MTR_METER_KMETER | (p->with_dr ? MTR_METER_DR14 : 0u);
	cfg.n_streams = 1;
	cfg.n_channels = p->channels;
	cfg.sample_rate = (float) rate;
	if (lv2_engine_open (&cfg, features, &p->engine) != MTR_OK) {
		fprintf (stderr, "meters_amd: %s: %s\n", d->URI, mtr_last_error ());
		free (p);
		return NULL;
	}
	return p;
}

void dr14_connect_port (LV2_Handle h, uint32_t port, void* data)
{
	if (port < PORT_COUNT) ((DrPlugin*) h)->port[port] = data;
}

void dr14_run (LV2_Handle h, uint32_t n_samples)
{
	DrPlugin* p = (DrPlugin*) h;
	const int follow = *(const float*) p->port[PORT_FOLLOW_TRANSPORT] != 0.f;

	/* atom input: transport (a start clears the meters when the plugin follows the host), GUI reset, GUI on / off */
	const LV2_Atom_Sequence* in = (const LV2_Atom_Sequence*) p->port[PORT_ATOM_IN];
	if (in) {
		FORGE_FOREACH_OBJECT (in, &p->urid, obj) {
			const LV2_URID t = obj->body.otype;
			if (t == p->urid.time_Position) {
				const LV2_Atom* speed = object_get (obj, p->urid.time_speed);
				if (speed && speed->type == p->urid.atom_Float) {
					const int now = ((const LV2_Atom_Float*) speed)->body != 0.f;
					if (now && !p->rolling && follow) clear_meters (p);
					p->rolling = now;
				}
			} else if (t == p->urid_reset)           clear_meters (p);
			else if (t == p->urid.mtr_meters_on)     p->gui_attached = 1;
			else if (t == p->urid.mtr_meters_off)    p->gui_attached = 0;
		}
	}
	if (*(const float*) p->port[PORT_RESET] != 0.f) clear_meters (p);

	/* the block through the engine, then the three meters' read-outs */
	const float* chan[2] = { fport (p, 0, CH_IN), p->channels > 1 ? fport (p, 1, CH_IN) : NULL };
	mtr_stream_result tp;
	mtr_dr14_result dr;
	float rms[2] = { 0.f, 0.f }, held[2] = { 0.f, 0.f };
	memset (&tp, 0, sizeof (tp));
	memset (&dr, 0, sizeof (dr));
	int ok = 1;
	if (n_samples > 0) ok = mtr_engine_process_planar_host (p->engine, chan, n_samples) == MTR_OK;
	ok = ok && mtr_engine_results (p->engine, 0, 1, &tp) == MTR_OK
	        && mtr_engine_kmeter_read (p->engine, 0, 1, rms, held) == MTR_OK
	        && (!p->with_dr || mtr_engine_dr14_results (p->engine, 0, 1, &dr) == MTR_OK);
	if (!ok) fprintf (stderr, "meters_amd: dr14: %s\n", mtr_last_error ());

	for (uint32_t c = 0; c < p->channels && ok; ++c) {
		if (tp.tpb_peak[c] > p->tp_max[c]) p->tp_max[c] = tp.tpb_peak[c];
		*fport (p, c, CH_RMS_BAR)  = to_db (rms[c]);
		*fport (p, c, CH_PEAK_BAR) = to_db (tp.tpb_level[c]);
		*fport (p, c, CH_PEAK_MAX) = to_db (p->tp_max[c]);
		if (p->with_dr) {
			*fport (p, c, CH_RMS_MAX) = dr.m_rms[c];          /* the top-20 % RMS score, already in dB */
			*fport (p, c, CH_DR)      = dr.dr[c];
		} else {
			*fport (p, c, CH_RMS_MAX) = to_db (held[c]);      /* TPnRMS: the K-meter's held peak */
		}
	}
	if (ok) {
		if (p->with_dr && p->channels > 1) *(float*) p->port[PORT_DR_TOTAL] = dr.dr_total;
		*(float*) p->port[PORT_BLOCKS] = p->with_dr ? dr.block_count : 0.f;
	}

	if (p->gui_attached) {
		/* a GUI that has just attached must see every port change once: out-of-range markers and a block count
		 * that never repeats (src/dr14.c:455-466) */
		for (uint32_t c = 0; c < p->channels; ++c) {
			*fport (p, c, CH_PEAK_MAX) = -100.f;
			*fport (p, c, CH_RMS_MAX)  = -100.f;
			if (p->with_dr) *fport (p, c, CH_DR) = 21.f;
		}
		if (p->with_dr && p->channels > 1) *(float*) p->port[PORT_DR_TOTAL] = 21.f;
		*(float*) p->port[PORT_BLOCKS] = (float) (-1 - (rand () & 0xffff));
	}
	for (uint32_t c = 0; c < p->channels; ++c) {
		float* out = fport (p, c, CH_OUT);
		if (out != chan[c]) memcpy (out, chan[c], sizeof (float) * n_samples);
	}
}

void dr14_cleanup (LV2_Handle h)
{
	DrPlugin* p = (DrPlugin*) h;
	if (p->engine) mtr_engine_destroy (p->engine);
	free (p);
}
