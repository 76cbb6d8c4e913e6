/* lv2_plugins.h — the callbacks of the plugins that live in their own translation unit; all hidden
 * (only lv2_descriptor is exported from lib/meters_amd.so, as in the reference: Makefile:199). */
#ifndef MTR_LV2_PLUGINS_H
#define MTR_LV2_PLUGINS_H

#include "lv2_min.h"

/* lv2_ebur128.c — src/ebulv2.cc */
LV2_Handle  ebur128_instantiate (const LV2_Descriptor* d, double rate, const char* path, const LV2_Feature* const* features);
void        ebur128_connect_port (LV2_Handle h, uint32_t port, void* data);
void        ebur128_run (LV2_Handle h, uint32_t n_samples);
void        ebur128_cleanup (LV2_Handle h);
const void* ebur128_extension_data (const char* uri);

#endif
