/* lv2_min.h — the part of the LV2 core / atom / urid C ABI the meters plugin surface uses.
 *
 * Authored from the public LV2 specification (lv2plug.in); the LV2 SDK headers are not installed
 * in this image and nothing here is taken from the reference tree (which does not vendor them
 * either: src/meters.cc:24-28 includes <lv2/lv2plug.in/ns/lv2core/lv2.h> from the system).
 * Layouts are ABI: a stock LV2 host loads lib/meters_amd.so through exactly these structs.
 */
#ifndef MTR_LV2_MIN_H
#define MTR_LV2_MIN_H

#include <stddef.h>
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

/* ---- options + buf-size: the host's promise about block lengths (lv2plug.in/ns/ext/options, .../buf-size) ---- */
#define LV2_OPTIONS__options         "http://lv2plug.in/ns/ext/options#options"
#define LV2_BUF_SIZE__maxBlockLength "http://lv2plug.in/ns/ext/buf-size#maxBlockLength"
typedef enum { LV2_OPTIONS_INSTANCE, LV2_OPTIONS_RESOURCE, LV2_OPTIONS_BLANK, LV2_OPTIONS_PORT } LV2_Options_Context;
typedef struct {
	LV2_Options_Context context;
	uint32_t            subject;
	LV2_URID            key;      /* the array ends with key == 0 */
	uint32_t            size;
	LV2_URID            type;
	const void*         value;
} LV2_Options_Option;

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
typedef struct { uint32_t child_size; uint32_t child_type; } LV2_Atom_Vector_Body;
#define LV2_ATOM__Vector    LV2_ATOM_URI "#Vector"
#define LV2_TIME__frame     "http://lv2plug.in/ns/ext/time#frame"

/* ---- state extension (LV2 "state" spec): what ebur128_save / ebur128_restore use, src/ebulv2.cc:514-566 ---- */
#define LV2_STATE__interface "http://lv2plug.in/ns/ext/state#interface"
typedef void* LV2_State_Handle;
typedef enum { LV2_STATE_IS_POD = 1, LV2_STATE_IS_PORTABLE = 1 << 1, LV2_STATE_IS_NATIVE = 1 << 2 } LV2_State_Flags;
typedef enum {
	LV2_STATE_SUCCESS = 0, LV2_STATE_ERR_UNKNOWN = 1, LV2_STATE_ERR_BAD_TYPE = 2, LV2_STATE_ERR_BAD_FLAGS = 3,
	LV2_STATE_ERR_NO_FEATURE = 4, LV2_STATE_ERR_NO_PROPERTY = 5, LV2_STATE_ERR_NO_SPACE = 6
} LV2_State_Status;
typedef LV2_State_Status (*LV2_State_Store_Function) (LV2_State_Handle handle, uint32_t key, const void* value,
                                                      size_t size, uint32_t type, uint32_t flags);
typedef const void* (*LV2_State_Retrieve_Function) (LV2_State_Handle handle, uint32_t key, size_t* size,
                                                    uint32_t* type, uint32_t* flags);
typedef struct {
	LV2_State_Status (*save) (LV2_Handle instance, LV2_State_Store_Function store, LV2_State_Handle handle,
	                          uint32_t flags, const LV2_Feature* const* features);
	LV2_State_Status (*restore) (LV2_Handle instance, LV2_State_Retrieve_Function retrieve, LV2_State_Handle handle,
	                             uint32_t flags, const LV2_Feature* const* features);
} LV2_State_Interface;

#ifdef __cplusplus
}
#endif
#endif
