/* lv2_min.h — the part of the LV2 core / atom / urid C ABI the meters plugin surface uses.
 *
 * Authored from the public LV2 specification (lv2plug.in); the LV2 SDK headers are not installed
 * in this image and nothing here is taken from the reference tree (which does not vendor them
 * either: src/meters.cc:24-28 includes <lv2/lv2plug.in/ns/lv2core/lv2.h> from the system).
 * Layouts are ABI: a stock LV2 host loads lib/meters_amd.so through exactly these structs.
 */
#ifndef MTR_LV2_MIN_H
#define MTR_LV2_MIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- lv2core ---------------------------------------------------------------------------- */
typedef void* LV2_Handle;

typedef struct {
	const char* URI;
	void*       data;
} LV2_Feature;

typedef struct LV2_Descriptor {
	const char* URI;
	LV2_Handle (*instantiate) (const struct LV2_Descriptor* descriptor, double sample_rate,
	                           const char* bundle_path, const LV2_Feature* const* features);
	void (*connect_port) (LV2_Handle instance, uint32_t port, void* data_location);
	void (*activate) (LV2_Handle instance);
	void (*run) (LV2_Handle instance, uint32_t sample_count);
	void (*deactivate) (LV2_Handle instance);
	void (*cleanup) (LV2_Handle instance);
	const void* (*extension_data) (const char* uri);
} LV2_Descriptor;

#define LV2_SYMBOL_EXPORT __attribute__ ((visibility ("default")))
LV2_SYMBOL_EXPORT const LV2_Descriptor* lv2_descriptor (uint32_t index);

/* ---- urid ------------------------------------------------------------------------------- */
#define LV2_URID__map "http://lv2plug.in/ns/ext/urid#map"
typedef uint32_t LV2_URID;
typedef void*    LV2_URID_Map_Handle;
typedef struct {
	LV2_URID_Map_Handle handle;
	LV2_URID (*map) (LV2_URID_Map_Handle handle, const char* uri);
} LV2_URID_Map;

/* ---- atom (binary layout; all bodies padded to 8 bytes inside containers) ------------------ */
#define LV2_ATOM_URI        "http://lv2plug.in/ns/ext/atom"
#define LV2_ATOM__Blank     LV2_ATOM_URI "#Blank"
#define LV2_ATOM__Object    LV2_ATOM_URI "#Object"
#define LV2_ATOM__Int       LV2_ATOM_URI "#Int"
#define LV2_ATOM__Long      LV2_ATOM_URI "#Long"
#define LV2_ATOM__Float     LV2_ATOM_URI "#Float"
#define LV2_ATOM__Double    LV2_ATOM_URI "#Double"
#define LV2_ATOM__Bool      LV2_ATOM_URI "#Bool"
#define LV2_ATOM__Sequence  LV2_ATOM_URI "#Sequence"
#define LV2_TIME__Position  "http://lv2plug.in/ns/ext/time#Position"
#define LV2_TIME__speed     "http://lv2plug.in/ns/ext/time#speed"

typedef struct { uint32_t size; uint32_t type; } LV2_Atom;
typedef struct { LV2_Atom atom; int32_t body; } LV2_Atom_Int;
typedef struct { LV2_Atom atom; float body; } LV2_Atom_Float;
typedef struct { uint32_t unit; uint32_t pad; } LV2_Atom_Sequence_Body;
typedef struct { LV2_Atom atom; LV2_Atom_Sequence_Body body; } LV2_Atom_Sequence;
typedef struct { int64_t frames; LV2_Atom body; } LV2_Atom_Event;
typedef struct { uint32_t id; uint32_t otype; } LV2_Atom_Object_Body;
typedef struct { LV2_Atom atom; LV2_Atom_Object_Body body; } LV2_Atom_Object;
typedef struct { uint32_t key; uint32_t context; LV2_Atom value; } LV2_Atom_Property_Body;

#ifdef __cplusplus
}
#endif
#endif
