/* lv2_plugins.h — the callbacks of the plugins that live in their own translation unit; all hidden
 * (only lv2_descriptor is exported from lib/meters_amd.so, as in the reference: Makefile:199). */
#ifndef MTR_LV2_PLUGINS_H
#define MTR_LV2_PLUGINS_H

#include <math.h>
#include <stdio.h>
#include <string.h>

#include "lv2_min.h"
#include "mtr_engine.h"

/* An LV2 run() has no error channel, and the reference never fails inside one.  The GPU plugins can (a device that
 * went away mid-session): a failing engine call is reported ONCE per failure streak on stderr, and the plugin then
 * writes MTR_LV2_NO_DATA — a quiet NaN, which no meter of the bundle ever produces — to its meter outputs instead of
 * leaving the last good values there, so a host or GUI can tell "no measurement" from "unchanged level".  Audio is
 * still passed through. */
#define MTR_LV2_NO_DATA ((float) NAN)
/* block size the engine is warmed up for at instantiate (mtr_engine_prepare_host) when the host does not say: larger blocks
 * still work, the first one of a larger size pays for its staging buffer */
#define MTR_LV2_MAX_BLOCK 8192u
/* A host that passes options:options with buf-size:maxBlockLength (an atom:Int; both need urid:map) has promised the largest
 * block run () will see: that is what gets warmed up.  (The reference does not read the option — its DSP needs no buffers
 * sized by it, TruePeakdsp::process asserts n <= 8192 instead, jmeters/truepeakdsp.cc:44.) */
static inline uint32_t lv2_max_block (const LV2_Feature* const* features)
{
	const LV2_URID_Map* map = NULL;
	const LV2_Options_Option* opt = NULL;
	for (int i = 0; features && features[i]; ++i) {
		if (!strcmp (features[i]->URI, LV2_URID__map)) map = (const LV2_URID_Map*) features[i]->data;
		else if (!strcmp (features[i]->URI, LV2_OPTIONS__options)) opt = (const LV2_Options_Option*) features[i]->data;
	}
	if (!map || !opt) return MTR_LV2_MAX_BLOCK;
	const LV2_URID key = map->map (map->handle, LV2_BUF_SIZE__maxBlockLength), t_int = map->map (map->handle, LV2_ATOM__Int);
	for (; opt->key; ++opt)
		if (opt->context == LV2_OPTIONS_INSTANCE && opt->key == key && opt->type == t_int && opt->size == sizeof (int32_t) && opt->value) {
			const int32_t n = *(const int32_t*) opt->value;
			if (n >= 1 && n <= (1 << 22)) return (uint32_t) n;
		}
	return MTR_LV2_MAX_BLOCK;
}
/* an engine for one plugin instance, warmed up: what run () does first must not be the allocation and module loading */
static inline int lv2_engine_open (const mtr_config* cfg, const LV2_Feature* const* features, mtr_engine** out)
{
	int rc = mtr_engine_create (cfg, out);
	if (rc != MTR_OK) return rc;
	rc = mtr_engine_prepare_host (*out, lv2_max_block (features));
	if (rc != MTR_OK) { mtr_engine_destroy (*out); *out = NULL; }
	return rc;
}
static inline int lv2_engine_ok (int rc, int* failing, const char* who)
{
	if (rc == MTR_OK) { *failing = 0; return 1; }
	if (!*failing) fprintf (stderr, "meters_amd: %s: %s — meter outputs carry NaN until the engine answers again\n", who, mtr_last_error ());
	*failing = 1;
	return 0;
}

/* lv2_ebur128.c — src/ebulv2.cc */
LV2_Handle  ebur128_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features);
void        ebur128_connect_port (LV2_Handle h, uint32_t port, void* data);
void        ebur128_run (LV2_Handle h, uint32_t n_samples);
void        ebur128_cleanup (LV2_Handle h);
const void* ebur128_extension_data (const char* uri);

/* lv2_intstat.c — src/bitmeter.c, src/sigdistlv2.c */
LV2_Handle  bim_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features);
void        bim_run (LV2_Handle h, uint32_t n_samples);
const void* bim_extension_data (const char* uri);
LV2_Handle  sdh_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features);
void        sdh_run (LV2_Handle h, uint32_t n_samples);
const void* sdh_extension_data (const char* uri);
void        intstat_connect_port (LV2_Handle h, uint32_t port, void* data);
void        intstat_cleanup (LV2_Handle h);

/* lv2_needle.c — the needle meters of src/meters.cc on jmeters/{iec1ppm,iec2ppm,msppm,stcorr,kmeter}dsp.cc (CPU) */
LV2_Handle  needle_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features);
void        needle_connect_port (LV2_Handle h, uint32_t port, void* data);
void        needle_run (LV2_Handle h, uint32_t n_samples);
void        cor_run (LV2_Handle h, uint32_t n_samples);
void        bbcm_run (LV2_Handle h, uint32_t n_samples);
void        kmeter_run (LV2_Handle h, uint32_t n_samples);
void        needle_cleanup (LV2_Handle h);
LV2_Handle  sur_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features);
void        sur_connect_port (LV2_Handle h, uint32_t port, void* data);
void        sur_run (LV2_Handle h, uint32_t n_samples);

/* lv2_dr14.c — src/dr14.c: DR-14 and true-peak + RMS */
LV2_Handle  dr14_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features);
void        dr14_connect_port (LV2_Handle h, uint32_t port, void* data);
void        dr14_run (LV2_Handle h, uint32_t n_samples);
void        dr14_cleanup (LV2_Handle h);

#endif
