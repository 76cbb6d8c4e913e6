/* lv2_forge.h — a small atom writer / reader for the plugins that talk to a UI (EBUr128, bitmeter,
 * SigDistHist).  No LV2 SDK in this image: layouts come from include/lv2_min.h, and what is written
 * is byte for byte what lv2_atom_forge_* produces for the same calls (8-byte padded bodies; an object
 * is {id 1, otype} followed by properties {key, context 0, value atom}; every message is an event at
 * frame 0 appended to the notify sequence, whose atom.size grows by the padded event size). */
#ifndef MTR_LV2_FORGE_H
#define MTR_LV2_FORGE_H

#include <stdio.h>
#include <string.h>

#include "lv2_min.h"

#define MTR_URI "http://gareus.org/oss/lv2/meters#"

/* numeric keys of the control messages, src/uris.h:187-203 */
enum { KEY_INVALID = 0, CTL_START, CTL_PAUSE, CTL_RESET, CTL_TRANSPORTSYNC, CTL_AUTORESET, CTL_RADARTIME, CTL_UISETTINGS,
       CTL_LV2_RADARTIME, CTL_LV2_FTM, CTL_LV2_RESETRADAR, CTL_LV2_RESYNCDONE, CTL_SAMPLERATE, CTL_WINDOWED, CTL_AVERAGE };

typedef struct {
	LV2_URID atom_Blank, atom_Object, atom_Int, atom_Long, atom_Float, atom_Double, atom_Bool, atom_Vector, atom_Sequence;
	LV2_URID time_Position, time_speed;
	LV2_URID mtr_cckey, mtr_ccval, mtr_control, mtr_meters_on, mtr_meters_off, mtr_meters_cfg;
} ForgeUrids;

static inline void forge_map_urids (LV2_URID_Map* map, ForgeUrids* u)
{
#define FMAP(field, uri) u->field = map->map (map->handle, uri)
	FMAP (atom_Blank, LV2_ATOM__Blank); FMAP (atom_Object, LV2_ATOM__Object); FMAP (atom_Int, LV2_ATOM__Int);
	FMAP (atom_Long, LV2_ATOM__Long); FMAP (atom_Float, LV2_ATOM__Float); FMAP (atom_Double, LV2_ATOM__Double);
	FMAP (atom_Bool, LV2_ATOM__Bool); FMAP (atom_Vector, LV2_ATOM__Vector); FMAP (atom_Sequence, LV2_ATOM__Sequence);
	FMAP (time_Position, LV2_TIME__Position); FMAP (time_speed, LV2_TIME__speed);
	FMAP (mtr_cckey, MTR_URI "controlkey"); FMAP (mtr_ccval, MTR_URI "controlval"); FMAP (mtr_control, MTR_URI "control");
	FMAP (mtr_meters_on, MTR_URI "meteron"); FMAP (mtr_meters_off, MTR_URI "meteroff"); FMAP (mtr_meters_cfg, MTR_URI "metercfg");
#undef FMAP
}

typedef struct { uint8_t* buf; uint32_t cap, pos; LV2_Atom_Sequence* seq; const ForgeUrids* u; } Forge;
typedef struct { LV2_Atom_Event* ev; uint32_t ev_pos, body0; } ObjFrame;

static inline uint32_t pad8 (uint32_t n) { return (n + 7u) & ~7u; }

static inline void* forge_raw (Forge* f, uint32_t n)
{
	if (f->pos + pad8 (n) > f->cap) return NULL;
	void* p = f->buf + f->pos;
	memset (p, 0, pad8 (n));
	f->pos += pad8 (n);
	return p;
}

/* lv2_atom_forge_set_buffer + sequence_head: the host presets notify->atom.size to the capacity of the
 * buffer behind the port (src/ebulv2.cc:244) */
static inline void forge_begin (Forge* f, LV2_Atom_Sequence* notify, const ForgeUrids* u)
{
	f->buf = (uint8_t*) notify;
	f->cap = notify->atom.size + (uint32_t) sizeof (LV2_Atom);
	f->pos = 0;
	f->u = u;
	f->seq = (LV2_Atom_Sequence*) forge_raw (f, sizeof (LV2_Atom_Sequence));
	if (f->seq) { f->seq->atom.type = u->atom_Sequence; f->seq->atom.size = sizeof (LV2_Atom_Sequence_Body); }
}
/* bytes of the notify atom in use: what the reference reads back as self->notify->atom.size */
static inline uint32_t forge_used (const Forge* f) { return f->seq ? f->seq->atom.size : 0; }

/* lv2_atom_forge_frame_time (0) + lv2_atom_forge_object (id 1, otype) */
static inline int obj_begin (Forge* f, ObjFrame* fr, LV2_URID otype)
{
	if (!f->seq) return 0;
	fr->ev_pos = f->pos;
	fr->ev = (LV2_Atom_Event*) forge_raw (f, sizeof (LV2_Atom_Event) + sizeof (LV2_Atom_Object_Body));
	if (!fr->ev) return 0;
	fr->ev->frames = 0;
	fr->ev->body.type = f->u->atom_Object;
	LV2_Atom_Object_Body* ob = (LV2_Atom_Object_Body*) (fr->ev + 1);
	ob->id = 1; ob->otype = otype;
	fr->body0 = f->pos - (uint32_t) sizeof (LV2_Atom_Object_Body);
	return 1;
}
static inline void obj_end (Forge* f, ObjFrame* fr)
{
	fr->ev->body.size = f->pos - fr->body0;
	f->seq->atom.size += f->pos - fr->ev_pos;
}

/* property head + a value atom of `size` bytes */
static inline void prop_raw (Forge* f, LV2_URID key, LV2_URID type, const void* v, uint32_t size)
{
	LV2_Atom_Property_Body* p = (LV2_Atom_Property_Body*) forge_raw (f, (uint32_t) sizeof (LV2_Atom_Property_Body) + size);
	if (!p) return;
	p->key = key; p->context = 0; p->value.size = size; p->value.type = type;
	memcpy (p + 1, v, size);
}
static inline void prop_f (Forge* f, LV2_URID key, float v)   { prop_raw (f, key, f->u->atom_Float, &v, 4); }
static inline void prop_i (Forge* f, LV2_URID key, int32_t v) { prop_raw (f, key, f->u->atom_Int, &v, 4); }
static inline void prop_b (Forge* f, LV2_URID key, int32_t v) { prop_raw (f, key, f->u->atom_Bool, &v, 4); }
static inline void prop_l (Forge* f, LV2_URID key, int64_t v) { prop_raw (f, key, f->u->atom_Long, &v, 8); }
static inline void prop_d (Forge* f, LV2_URID key, double v)  { prop_raw (f, key, f->u->atom_Double, &v, 8); }
/* lv2_atom_forge_vector (sizeof (int32_t), atom:Int, n, data) */
static inline void prop_vec_i32 (Forge* f, LV2_URID key, const int32_t* data, uint32_t n)
{
	const uint32_t size = (uint32_t) sizeof (LV2_Atom_Vector_Body) + 4u * n;
	LV2_Atom_Property_Body* p = (LV2_Atom_Property_Body*) forge_raw (f, (uint32_t) sizeof (LV2_Atom_Property_Body) + size);
	if (!p) return;
	p->key = key; p->context = 0; p->value.size = size; p->value.type = f->u->atom_Vector;
	LV2_Atom_Vector_Body* vb = (LV2_Atom_Vector_Body*) (p + 1);
	vb->child_size = 4; vb->child_type = f->u->atom_Int;
	memcpy (vb + 1, data, 4u * n);
}

/* forge_kvcontrolmessage, src/uris.h:280-296 */
static inline void kv_message (Forge* f, int key, float value)
{
	ObjFrame fr;
	if (!obj_begin (f, &fr, f->u->mtr_control)) return;
	prop_i (f, f->u->mtr_cckey, key);
	prop_f (f, f->u->mtr_ccval, value);
	obj_end (f, &fr);
}

/* value of property `key` inside an object body, or NULL (lv2_atom_object_get for one key) */
static inline const LV2_Atom* object_get (const LV2_Atom_Object* obj, LV2_URID key)
{
	const uint8_t* p = (const uint8_t*) (&obj->body + 1);
	const uint8_t* end = (const uint8_t*) &obj->body + obj->atom.size;
	while (p + sizeof (LV2_Atom_Property_Body) <= end) {
		const LV2_Atom_Property_Body* pb = (const LV2_Atom_Property_Body*) p;
		if (pb->key == key) return &pb->value;
		p += pad8 ((uint32_t) sizeof (LV2_Atom_Property_Body) + pb->value.size);
	}
	return NULL;
}

/* get_cc_key_value, src/uris.h:298-318: 0 on success */
static inline int get_cc_key_value (const ForgeUrids* u, const LV2_Atom_Object* obj, int* k, float* v)
{
	*k = 0; *v = 0.f;
	const LV2_Atom* key = object_get (obj, u->mtr_cckey);
	const LV2_Atom* val = object_get (obj, u->mtr_ccval);
	if (!key || !val) {
		fprintf (stderr, "MTRlv2: Malformed ctrl message has no key or value.\n");
		return -1;
	}
	*k = ((const LV2_Atom_Int*) key)->body;
	*v = ((const LV2_Atom_Float*) val)->body;
	return 0;
}

/* iterate the object events of a control sequence */
#define FORGE_FOREACH_OBJECT(seq, u, obj)                                                                     \
	for (const uint8_t *p_ = (const uint8_t*) (&(seq)->body + 1),                                             \
	                   *end_ = (const uint8_t*) &(seq)->body + (seq)->atom.size;                              \
	     p_ + sizeof (LV2_Atom_Event) <= end_;                                                                \
	     p_ += pad8 ((uint32_t) sizeof (LV2_Atom_Event) + ((const LV2_Atom_Event*) p_)->body.size))           \
		for (const LV2_Atom_Object* obj = (const LV2_Atom_Object*) &((const LV2_Atom_Event*) p_)->body;       \
		     obj && (obj->atom.type == (u)->atom_Blank || obj->atom.type == (u)->atom_Object); obj = NULL)

#endif
