"""The EBUr128 plugin end to end through the LV2 ABI, message by message.

`Model` below restates the bookkeeping of the reference's ebur128_run (src/ebulv2.cc:239-498: state-to-UI
messages, control keys, transport follow, radar ring + resync batches, histogram diffs, `ebulevels`) on top
of the oracle's DSP values for the audio seen so far; every cycle the plugin's notify sequence must be the
model's, object by object (floats: 1e-3 dB on M / S / max, +-0.01 dB on the gated values; histogram counts
exact).  LV2 State round-trips the packed settings word."""
import struct

import numpy as np
import pytest

from _lv2host import Host, Instance, MTR_URI, arm_notify, forge_object, forge_sequence, notify_buffer, parse_sequence

pytestmark = pytest.mark.gpu

K = MTR_URI
CTL = dict(START=1, PAUSE=2, RESET=3, TRANSPORTSYNC=4, AUTORESET=5, RADARTIME=6, UISETTINGS=7,
           LV2_RADARTIME=8, LV2_FTM=9, LV2_RESETRADAR=10, LV2_RESYNCDONE=11)
NINF = float("-inf")


@pytest.fixture(scope="module")
def host():
    return Host()


def cfg(host, key, val):
    return forge_object(host, K + "metercfg", [(K + "controlkey", "i", CTL[key]), (K + "controlval", "f", float(val))])


class Model:
    """ebur128_run's bookkeeping; DSP values are handed in (src/ebulv2.cc, line numbers in the comments)."""

    def __init__(self, rate, capacity):
        self.rate, self.capacity = rate, capacity
        self.ui_active = self.send_state = False
        self.ftm, self.rolling, self.integrating, self.dbtp = 0, False, False, False
        self.ui_settings = 8
        self.pos_max, self.pos_cur, self.spd_cur = 360, 0, 0
        self.radarM = [NINF] * 360
        self.radarS = [NINF] * 360
        self.MC = self.SC = NINF
        self.resync = -1
        self.set_radarspeed(120.0)                          # :171
        self.histM = np.zeros(751, np.int64)
        self.histS = np.zeros(751, np.int64)
        self.hist_maxM = self.hist_maxS = 0
        self.integration_time = 0
        self.tp_max = NINF
        self.out = []
        self.reset_audio_from = 0                           # the test re-runs the oracle from here after a RESET

    def set_radarspeed(self, seconds):                      # :76-79
        self.spd_max = max(int(np.rint(seconds * self.rate / self.pos_max)), 4096)

    def kv(self, key, val):
        self.out.append((K + "control", {K + "controlkey": CTL[key], K + "controlval": float(val)}))

    def reset(self, cycle_start_frame):                     # ebu_reset :44-63
        self.kv("LV2_RESETRADAR", 0)
        self.radarM = [NINF] * 360
        self.radarS = [NINF] * 360
        self.histM[:] = 0
        self.histS[:] = 0
        self.pos_cur = 0
        self.integration_time = 0
        self.hist_maxM = self.hist_maxS = 0
        self.tp_max = NINF
        self.reset_audio_from = cycle_start_frame
        self.was_reset = True

    def integrate(self, on, frame):                         # :65-74
        if self.integrating == on:
            return
        if on and (self.ftm & 2):
            self.reset(frame)
        self.integrating = on
        self.integr_events.append(on)

    def begin(self):
        self.out = []
        self.integr_events = []
        self.was_reset = False
        if self.send_state and self.ui_active:              # :248-255
            self.send_state = False
            self.kv("LV2_FTM", self.ftm)
            self.kv("LV2_RADARTIME", self.pos_max * self.spd_max / self.rate)
            self.kv("UISETTINGS", self.ui_settings)

    def control(self, msgs, n, frame):                      # :257-331
        for m in msgs:
            if m[0] == "speed":
                ts = m[1]
                if ts != 0 and not self.rolling and (self.ftm & 1):
                    self.integrate(True, frame)
                if ts == 0 and self.rolling and (self.ftm & 1):
                    self.integrate(False, frame)
                self.rolling = ts != 0
            elif m[0] == "meteron":
                self.ui_active, self.send_state, self.resync = True, True, 0
                self.histM[:] = 0
                self.histS[:] = 0
                self.hist_maxM = self.hist_maxS = 0
            elif m[0] == "meteroff":
                self.ui_active = False
            else:
                key, v = m
                if key == "START":
                    self.integrate(True, frame)
                elif key == "PAUSE":
                    self.integrate(False, frame)
                elif key == "RESET":
                    self.reset(frame)
                elif key == "TRANSPORTSYNC":
                    if v == 1:
                        self.ftm |= 1
                        if self.rolling != self.integrating:
                            self.integrate(self.rolling, frame)
                    else:
                        self.ftm &= ~1
                elif key == "AUTORESET":
                    self.ftm = (self.ftm | 2) if v == 1 else (self.ftm & ~2)
                elif key == "RADARTIME":
                    if 30 <= v <= 600:
                        self.set_radarspeed(v)
                        self.spd_max = max(self.spd_max, 2 * n)
                    self.kv("LV2_RADARTIME", self.pos_max * self.spd_max / self.rate)
                elif key == "UISETTINGS":
                    self.ui_settings = int(v)
                    self.dbtp = bool(self.ui_settings & 64)

    def radar_point(self, m, s, pos):
        self.out.append((K + "rdr_radarpoint", {K + "ebu_loudnessM": m, K + "ebu_loudnessS": s, K + "rdr_pointpos": pos,
                                               K + "rdr_pos_cur": self.pos_cur, K + "rdr_pos_max": self.pos_max}))

    def after_audio(self, n, o9, hm, hs, counts, tp_block_db):
        lm, mm, ls, ms, il, rn, rx = o9[0], o9[1], o9[2], o9[3], o9[4], o9[6], o9[7]
        if self.dbtp:                                       # :360-367
            self.tp_max = max(self.tp_max, tp_block_db)
        else:
            self.tp_max = NINF
        if self.resync >= 0:                                # :369-388
            batch = min((self.capacity - 512) // 192, 16)
            for _ in range(batch):
                if self.resync >= self.pos_max:
                    self.resync = -1
                    self.kv("LV2_RESYNCDONE", 0)
                    break
                self.radar_point(self.radarM[self.resync], self.radarS[self.resync], self.resync)
                self.resync += 1
        if lm > self.MC:                                    # :390-392
            self.MC = lm
        if lm > self.SC:
            self.SC = ls
        if self.integrating:
            self.integration_time += n
        self.spd_cur += n
        if self.spd_cur > self.spd_max:                     # :398-423
            if self.ui_active:
                self.radar_point(self.MC, self.SC, self.pos_cur)
            self.radarM[self.pos_cur], self.radarS[self.pos_cur] = self.MC, self.SC
            self.spd_cur %= self.spd_max
            self.pos_cur = (self.pos_cur + 1) % self.pos_max
            self.MC = self.SC = NINF
        if self.ui_active and counts[0] > 10 and counts[1] > 10:   # :425-462
            msgtx, changed = 0, False
            for i in range(110, 650):
                vm, vs = int(hm[i]), int(hs[i])
                if self.histM[i] != vm or self.histS[i] != vs:
                    msgtx += 1
                    if msgtx - 1 > 16:
                        break
                    self.histM[i], self.histS[i] = vm, vs
                    self.out.append((K + "rdr_histpoint", {K + "ebu_loudnessM": vm, K + "ebu_loudnessS": vs, K + "rdr_pointpos": i}))
                if vm > self.hist_maxM:
                    self.hist_maxM, changed = vm, True
                if vs > self.hist_maxS:
                    self.hist_maxS, changed = vs, True
            if changed:
                self.out.append((K + "rdr_histogram", {K + "ebu_loudnessM": self.hist_maxM, K + "ebu_loudnessS": self.hist_maxS}))
        if self.ui_active:                                  # :464-482
            self.out.append((K + "ebulevels", {
                K + "ebu_loudnessM": lm, K + "ebu_maxloudnM": mm, K + "ebu_loudnessS": ls, K + "ebu_maxloudnS": ms,
                K + "ebu_integrated": il, K + "ebu_range_min": rn, K + "ebu_range_max": rx, K + "truepeak": self.tp_max,
                K + "ebu_integrating": int(self.integrating), K + "ebu_integr_time": self.integration_time / self.rate}))


TOL = {K + "ebu_loudnessM": 1e-3, K + "ebu_maxloudnM": 1e-3, K + "ebu_loudnessS": 1e-3, K + "ebu_maxloudnS": 1e-3,
       K + "ebu_integrated": 0.01, K + "ebu_range_min": 0.1001, K + "ebu_range_max": 0.1001, K + "truepeak": 1e-3,
       K + "ebu_integr_time": 1e-4, K + "controlval": 1e-4}


def same(got, want, where):
    assert [g[0] for g in got] == [w[0] for w in want], (where, [g[0] for g in got], [w[0] for w in want])
    for (gt, gp), (_, wp) in zip(got, want):
        assert set(gp) == set(wp), (where, gt)
        for k, w in wp.items():
            g = gp[k]
            if isinstance(w, float) or k in TOL:
                if np.isinf(w) or np.isinf(g):
                    assert g == w, (where, gt, k, g, w)
                else:
                    assert abs(g - w) <= TOL.get(k, 1e-3), (where, gt, k, g, w)
            else:
                assert g == w, (where, gt, k, g, w)


def speed_msg(host, ts):
    body = struct.pack("<II", 1, host.urid("http://lv2plug.in/ns/ext/time#Position"))
    body += struct.pack("<IIII", host.urid("http://lv2plug.in/ns/ext/time#speed"), 0, 4,
                        host.urid("http://lv2plug.in/ns/ext/atom#Float")) + struct.pack("<f", ts) + b"\0" * 4
    return struct.pack("<II", len(body), host.urid("http://lv2plug.in/ns/ext/atom#Object")) + body


def test_ebur128_protocol_message_by_message(host, oracle):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden import tri_noise
    fs, B = 48000.0, 1024
    x = tri_noise(48000 * 8, 4711, 0.25, period=36000)
    ncyc = x.shape[0] // B
    CAP = 16384
    inst = Instance(host, "EBUr128")
    assert inst.ok()
    notify = notify_buffer(CAP)
    inst.connect(1, notify)
    model = Model(fs, CAP)

    # what the host sends, by cycle
    script = {
        0: [("meteron",), ("UISETTINGS", 8 + 64), ("RADARTIME", 30.0), ("START", 0)],
        40: [("RADARTIME", 10.0)],                            # out of range: only the reply
        100: [("PAUSE", 0)],
        110: [("START", 0)],
        150: [("meteroff",)],
        160: [("meteron",)],                                   # radar + histogram resync
        230: [("AUTORESET", 1), ("TRANSPORTSYNC", 1)],         # not rolling -> integration follows: pause
        240: [("speed", 1.0)],                                 # rolling: start, and auto-reset
        300: [("speed", 0.0)],
        310: [("TRANSPORTSYNC", 0), ("RESET", 0), ("START", 0)],
    }

    def wire(msgs):
        obs = []
        for m in msgs:
            if m[0] == "speed":
                obs.append(speed_msg(host, m[1]))
            elif m[0] in ("meteron", "meteroff"):
                obs.append(forge_object(host, K + m[0], []))
            else:
                obs.append(cfg(host, m[0], m[1]))
        return forge_sequence(host, obs)

    # The DSP side of the model: the oracle's Ebu_r128_proc + 2 x TruePeakdsp, fed the same blocks and the
    # same integration controls the model decides on.
    dsp = oracle.ebu_stream(fs)
    for c in range(ncyc):
        msgs = script.get(c, [])
        frame = c * B
        model.begin()
        model.control(msgs, B, frame)
        if model.was_reset and not model.integr_events:
            dsp.reset()
        for on in model.integr_events:                       # (an auto-reset precedes its start)
            if on and model.was_reset:
                dsp.reset()
            dsp.start() if on else dsp.pause()
        bl, br = x[frame:frame + B, 0].copy(), x[frame:frame + B, 1].copy()
        inst.connect(0, wire(msgs))
        for port, arr in ((2, bl), (3, bl), (4, br), (5, br)):
            inst.connect(port, arr)
        arm_notify(notify)
        inst.run(B)
        got = parse_sequence(host, notify)
        o9, hm, hs, counts, tp = dsp.process(bl, br)
        tpdb = 20 * np.log10(max(tp)) if max(tp) > 0 else NINF
        model.after_audio(B, o9, hm, hs, counts, tpdb)
        same(got, model.out, c)
    assert model.pos_cur > 10                                  # the radar did advance (30.72 s ring, 8 s of audio)

    # LV2 State: ui_settings | follow_transport_mode << 8 | radar_spd_max << 16 as one atom:Int (:514-553)
    kept = inst.state_save()
    key = host.urid(K + "ebu_state")
    raw, typ, flags = kept[key]
    word = struct.unpack("<I", raw)[0]
    assert typ == host.urid("http://lv2plug.in/ns/ext/atom#Int") and flags == 3
    assert word & 0xff == 8 + 64 and (word >> 8) & 3 == 2 and word >> 16 == 4096
    inst.cleanup()

    inst = Instance(host, "EBUr128")
    inst.state_restore({key: (struct.pack("<I", (5000 << 16) | (3 << 8) | 64), typ, flags)})
    notify = notify_buffer(CAP)
    inst.connect(1, notify)
    z = np.zeros(64, np.float32)
    for port in (2, 3, 4, 5):
        inst.connect(port, z)
    for c, msgs in enumerate(([("meteron",)], [])):
        inst.connect(0, wire(msgs))
        arm_notify(notify)
        inst.run(64)
        got = parse_sequence(host, notify)
    kv = [p for t, p in got if t == K + "control"]
    assert [(p[K + "controlkey"], round(p[K + "controlval"], 3)) for p in kv[:3]] == \
        [(CTL["LV2_FTM"], 3.0), (CTL["LV2_RADARTIME"], round(360 * 5000 / 48000.0, 3)), (CTL["UISETTINGS"], 64.0)]
    inst.cleanup()
