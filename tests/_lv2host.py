"""A minimal LV2 host in ctypes for lib/meters_amd.so: descriptor enumeration, port wiring, a URID
map, and just enough atom forging / parsing to drive the EBUr128 plugin headlessly.
Layouts follow the public LV2 C ABI (include/lv2_min.h). Test infrastructure."""
import ctypes as C
import os
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# MTR_PLUGIN_SO: an instrumented build of the same plugin (make check-asan)
PLUGIN_SO = os.environ.get("MTR_PLUGIN_SO") or os.path.join(ROOT, "meters.lv2_amd", "lib", "meters_amd.so")
MTR_URI = "http://gareus.org/oss/lv2/meters#"
ATOM = "http://lv2plug.in/ns/ext/atom#"


class Feature(C.Structure):
    _fields_ = [("URI", C.c_char_p), ("data", C.c_void_p)]


class Descriptor(C.Structure):
    pass


_INST = C.CFUNCTYPE(C.c_void_p, C.POINTER(Descriptor), C.c_double, C.c_char_p, C.POINTER(C.POINTER(Feature)))
_CONN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_void_p)
_VOIDH = C.CFUNCTYPE(None, C.c_void_p)
_RUN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32)
_EXT = C.CFUNCTYPE(C.c_void_p, C.c_char_p)
Descriptor._fields_ = [("URI", C.c_char_p), ("instantiate", _INST), ("connect_port", _CONN), ("activate", _VOIDH),
                       ("run", _RUN), ("deactivate", _VOIDH), ("cleanup", _VOIDH), ("extension_data", _EXT)]

_MAPFN = C.CFUNCTYPE(C.c_uint32, C.c_void_p, C.c_char_p)


class UridMap(C.Structure):
    _fields_ = [("handle", C.c_void_p), ("map", _MAPFN)]


class Host:
    def __init__(self):
        from meters.lv2_amd import engine  # noqa: F401  (preloads the HIP runtime the way the engine wants it)
        self.lib = C.CDLL(PLUGIN_SO)
        self.lib.lv2_descriptor.restype = C.POINTER(Descriptor)
        self.lib.lv2_descriptor.argtypes = [C.c_uint32]
        self.urids = {}
        self._mapfn = _MAPFN(self._map)
        self.urid_map = UridMap(None, self._mapfn)

    def _map(self, handle, uri):
        u = uri.decode()
        return self.urids.setdefault(u, len(self.urids) + 1)

    def urid(self, uri):
        return self.urids.setdefault(uri, len(self.urids) + 1)

    def descriptors(self):
        out, i = [], 0
        while True:
            d = self.lib.lv2_descriptor(i)
            if not d:
                return out
            out.append(d.contents)
            i += 1

    def find(self, name):
        for d in self.descriptors():
            if d.URI.decode() == MTR_URI + name:
                return d
        raise KeyError(name)


class OptionsOption(C.Structure):
    """LV2_Options_Option (lv2plug.in/ns/ext/options)."""
    _fields_ = [("context", C.c_int), ("subject", C.c_uint32), ("key", C.c_uint32), ("size", C.c_uint32), ("type", C.c_uint32),
                ("value", C.c_void_p)]


class Instance:
    def __init__(self, host, name, rate=48000.0, with_urid_map=True, max_block_length=None):
        """max_block_length: passed as options:options { buf-size:maxBlockLength (atom:Int) } like a host that states it."""
        self.host, self.desc = host, host.find(name)
        feats = []
        if with_urid_map:
            self._f = Feature(b"http://lv2plug.in/ns/ext/urid#map", C.cast(C.pointer(host.urid_map), C.c_void_p))
            feats.append(C.pointer(self._f))
        if max_block_length is not None:
            self._mbl = C.c_int32(max_block_length)
            self._opts = (OptionsOption * 2)(
                OptionsOption(0, 0, host.urid("http://lv2plug.in/ns/ext/buf-size#maxBlockLength"), 4,
                              host.urid("http://lv2plug.in/ns/ext/atom#Int"), C.cast(C.pointer(self._mbl), C.c_void_p)),
                OptionsOption(0, 0, 0, 0, 0, None))
            self._fo = Feature(b"http://lv2plug.in/ns/ext/options#options", C.cast(self._opts, C.c_void_p))
            feats.append(C.pointer(self._fo))
        arr = (C.POINTER(Feature) * (len(feats) + 1))(*feats, None)
        self._arr = arr
        self.handle = self.desc.instantiate(C.pointer(self.desc), rate, b"/tmp/", arr)
        self.ports = {}

    def ok(self):
        return bool(self.handle)

    def connect(self, port, array):
        """array = None disconnects the port (LV2 allows connect_port (instance, port, NULL))."""
        self.ports[port] = array
        self.desc.connect_port(self.handle, port, array.ctypes.data_as(C.c_void_p) if array is not None else None)

    def run(self, n):
        self.desc.run(self.handle, n)

    def cleanup(self):
        if self.handle:
            self.desc.cleanup(self.handle)
            self.handle = None

    # ---- LV2 State through extension_data (state#interface: {save, restore}) ----
    def _state_iface(self):
        p = self.desc.extension_data(b"http://lv2plug.in/ns/ext/state#interface")
        return C.cast(p, C.POINTER(_StateIface)).contents if p else None

    def state_save(self):
        """-> {key_urid: (bytes, type_urid, flags)} as the plugin hands it to the host's store()."""
        iface, kept = self._state_iface(), {}

        def store(handle, key, value, size, typ, flags):
            kept[key] = (C.string_at(value, size), typ, flags)
            return 0
        cb = _STORE(store)
        assert iface.save(self.handle, cb, None, 0, None) == 0
        return kept

    def state_restore(self, kept):
        iface, hold = self._state_iface(), []

        def retrieve(handle, key, size, typ, flags):
            if key not in kept:
                return None
            raw, t, fl = kept[key]
            buf = C.create_string_buffer(raw, len(raw))
            hold.append(buf)
            size[0], typ[0], flags[0] = len(raw), t, fl
            return C.addressof(buf)
        cb = _RETRIEVE(retrieve)
        assert iface.restore(self.handle, cb, None, 0, None) == 0


_STORE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32)
_RETRIEVE = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32),
                        C.POINTER(C.c_uint32))
_SAVE = C.CFUNCTYPE(C.c_int, C.c_void_p, _STORE, C.c_void_p, C.c_uint32, C.c_void_p)
_RESTORE = C.CFUNCTYPE(C.c_int, C.c_void_p, _RETRIEVE, C.c_void_p, C.c_uint32, C.c_void_p)


class _StateIface(C.Structure):
    _fields_ = [("save", _SAVE), ("restore", _RESTORE)]


# ---- atoms ------------------------------------------------------------------------------------

def _pad8(n):
    return (n + 7) & ~7


def forge_object(host, otype_uri, props):
    """props: list of (key_uri, 'i'|'f', value). Returns bytes of a complete atom:Object."""
    body = struct.pack("<II", 1, host.urid(otype_uri))
    for key, kind, val in props:
        t = host.urid(ATOM + ("Int" if kind == "i" else "Float"))
        v = struct.pack("<i" if kind == "i" else "<f", val)
        body += struct.pack("<IIII", host.urid(key), 0, 4, t) + v + b"\0" * 4
    return struct.pack("<II", len(body), host.urid(ATOM + "Object")) + body


def forge_sequence(host, objects):
    """A control-port buffer: atom:Sequence of events at frame 0."""
    body = struct.pack("<II", 0, 0)
    for ob in objects:
        ev = struct.pack("<q", 0) + ob
        body += ev + b"\0" * (_pad8(len(ev)) - len(ev))
    raw = struct.pack("<II", len(body), host.urid(ATOM + "Sequence")) + body
    return np.frombuffer(raw + b"\0" * 64, np.uint8).copy()


def notify_buffer(capacity=4096):
    buf = np.zeros(capacity + 8, np.uint8)
    return buf


def arm_notify(buf):
    """Hosts preset atom.size to the capacity before every run()."""
    struct.pack_into("<II", buf, 0, buf.size - 8, 0)


def parse_sequence(host, buf):
    """-> list of (otype_uri, {key_uri: value}) for every object event in an atom:Sequence."""
    rev = {v: k for k, v in host.urids.items()}
    size, typ = struct.unpack_from("<II", buf, 0)
    out, p, end = [], 16, 8 + size
    raw = buf.tobytes()
    while p + 16 <= end:
        _frames, asize, atype = struct.unpack_from("<qII", raw, p)
        if rev.get(atype, "").endswith(("#Object", "#Blank")):
            _oid, otype = struct.unpack_from("<II", raw, p + 16)
            q, qend, props = p + 24, p + 16 + asize, {}
            while q + 16 <= qend:
                key, _ctx, vsize, vtype = struct.unpack_from("<IIII", raw, q)
                vt = rev.get(vtype, "")
                if vt.endswith("#Float"):
                    val = struct.unpack_from("<f", raw, q + 16)[0]
                elif vt.endswith("#Double"):
                    val = struct.unpack_from("<d", raw, q + 16)[0]
                elif vt.endswith("#Long"):
                    val = struct.unpack_from("<q", raw, q + 16)[0]
                elif vt.endswith("#Vector"):
                    csize, _ctype = struct.unpack_from("<II", raw, q + 16)
                    val = np.frombuffer(raw, np.int32, (vsize - 8) // csize, q + 24).copy()
                else:
                    val = struct.unpack_from("<i", raw, q + 16)[0]
                props[rev.get(key, key)] = val
                q += _pad8(16 + vsize)
            out.append((rev.get(otype, otype), props))
        p += _pad8(16 + asize)
    return out
