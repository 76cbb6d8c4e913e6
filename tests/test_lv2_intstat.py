"""The bitmeter and SigDistHist plugins end to end through the LV2 ABI: control keys in, the reference's
message cadence and payloads out (src/bitmeter.c:181-340, src/sigdistlv2.c:203-384), integer tables
bit-exact against the oracle on the samples of the current window, LV2 State round trip."""
import struct

import numpy as np
import pytest

import _signals as sig
from _lv2host import Host, Instance, MTR_URI, arm_notify, forge_object, forge_sequence, notify_buffer, parse_sequence

pytestmark = pytest.mark.gpu
K = MTR_URI
CTL = dict(START=1, PAUSE=2, RESET=3, TRANSPORTSYNC=4, AUTORESET=5, UISETTINGS=7, LV2_FTM=9, LV2_RESETRADAR=10,
           SAMPLERATE=12, WINDOWED=13, AVERAGE=14)


@pytest.fixture(scope="module")
def host():
    return Host()


def cfg(host, key, val=0.0):
    return forge_object(host, K + "metercfg", [(K + "controlkey", "i", CTL[key]), (K + "controlval", "f", float(val))])


def drive(host, inst, notify, block, objs):
    inst.connect(0, forge_sequence(host, objs))
    inst.connect(2, block)
    inst.connect(3, block)
    arm_notify(notify)
    inst.run(block.size)
    return parse_sequence(host, notify)


class BimModel:
    """bim_run's bookkeeping (src/bitmeter.c:181-340); the table of a window comes from the oracle."""

    def __init__(self, oracle, rate):
        self.oracle, self.rate = oracle, rate
        self.integrating, self.average, self.ui, self.send_state = True, False, False, False
        self.resync, self.itime = 0, 0
        self.window, self.base = [], [0, 0, 0]

    def clear(self, counters):                                   # bim_clear :47-55
        if counters is not None:
            self.base = [b + int(c) for b, c in zip(self.base, counters[2:5])]
        self.window, self.itime = [], 0

    def cycle(self, msgs, blk):
        out = []
        if self.send_state and self.ui:                          # :190-193
            self.send_state = False
            out.append(("control", {K + "controlkey": CTL["SAMPLERATE"], K + "controlval": self.rate}))
        for m in msgs:                                           # :195-235
            if m == "meteron":
                self.ui, self.send_state = True, True
            elif m == "meteroff":
                self.ui = False
            elif m == "START":
                self.integrating = True
            elif m == "PAUSE":
                self.integrating = False
            elif m == "RESET":                                   # bim_reset :57-60
                self.clear(None)
                self.base = [0, 0, 0]
                self.send_state = True
            elif m == "AVERAGE":
                self.average = True
            elif m == "WINDOWED":
                self.average = False
        n = blk.size
        if self.integrating:                                     # :246-259
            self.window.append(blk)
            self.itime += n
        fps_limit = int(n * np.ceil(self.rate / (5.0 * n)))      # :261
        self.resync += n
        if self.resync >= fps_limit or self.send_state:
            w = self.oracle.bitstats(np.concatenate(self.window) if self.window else np.zeros(0, np.float32))
            if self.ui and (self.integrating or self.send_state):
                out.append(("bim_stats", dict(itime=self.itime, w=w, base=list(self.base))))
            if self.resync >= fps_limit:
                self.resync %= fps_limit
                if self.ui:
                    out.append(("bim_information", {K + "ebu_integrating": int(self.integrating),
                                                    K + "bim_averaging": int(self.average)}))
                if not self.average:
                    self.clear(w["counters"])
        return out


def test_bitmeter_windows_average_reset_and_state(host, oracle):
    B, fs = 1024, 48000.0
    x = sig.lcg_noise(B * 64, 21, 0.7)[:, 0].copy()
    x[5::997] = np.nan
    x[7::1009] = np.inf
    x[11::499] = np.float32(1e-41)
    x[13::257] = 0.0
    x[17::263] = -0.0
    inst = Instance(host, "bitmeter")
    assert inst.ok()
    notify = notify_buffer(16384)
    inst.connect(1, notify)
    model = BimModel(oracle, fs)
    script = {0: ["meteron"], 13: ["AVERAGE"], 27: ["RESET"], 31: ["WINDOWED"], 40: ["PAUSE"], 44: ["START"],
              52: ["meteroff"], 55: ["meteron"]}
    n_stats = 0
    for c in range(64):
        msgs = script.get(c, [])
        blk = x[c * B:(c + 1) * B].copy()
        objs = [forge_object(host, K + m, []) if m.startswith("meter") else cfg(host, m) for m in msgs]
        got = drive(host, inst, notify, blk, objs)
        want = model.cycle(msgs, blk)
        assert [t.split("#")[1] for t, _ in got] == [t for t, _ in want], (c, [t for t, _ in got], [t for t, _ in want])
        for (gt, gp), (wt, wp) in zip(got, want):
            if wt != "bim_stats":
                assert gp == wp, (c, gt, gp, wp)
                continue
            n_stats += 1
            w, base = wp["w"], wp["base"]
            assert gp[K + "ebu_integr_time"] == wp["itime"], c
            assert np.array_equal(gp[K + "bim_data"], w["hist"]), (c, np.flatnonzero(gp[K + "bim_data"] != w["hist"])[:8])
            assert gp[K + "bim_zero"] == w["counters"][0] and gp[K + "bim_pos"] == w["counters"][1], c
            assert gp[K + "bim_nan"] == base[0] + w["counters"][2], c
            assert gp[K + "bim_inf"] == base[1] + w["counters"][3], c
            assert gp[K + "bim_den"] == base[2] + w["counters"][4], c
            assert np.float32(gp[K + "bim_min"]) == w["vmin"] and np.float32(gp[K + "bim_max"]) == w["vmax"], c
    assert n_stats >= 6
    kept = inst.state_save()
    key = host.urid(K + "bim_state")
    assert struct.unpack("<I", kept[key][0])[0] == 0                    # windowed at the end
    inst.cleanup()
    inst = Instance(host, "bitmeter")
    inst.state_restore({key: (struct.pack("<I", 1), kept[key][1], kept[key][2])})
    notify = notify_buffer(16384)
    inst.connect(1, notify)
    z = np.zeros(B * 10, np.float32)
    got = drive(host, inst, notify, z, [forge_object(host, K + "meteron", [])])
    info = [p for t, p in got if t == K + "bim_information"]
    assert info and info[0][K + "bim_averaging"] == 1
    inst.cleanup()


def test_sigdisthist_cadence_payload_and_transport(host, oracle):
    B, fs = 1024, 48000.0
    fps_limit = max(int(fs / 25.0), B)                                  # 1920
    x = (sig.lcg_noise(B * 40, 33, 0.9)[:, 0] * np.float32(1.3)).copy()  # some samples beyond +-1.2 are dropped
    inst = Instance(host, "SigDistHist")
    assert inst.ok()
    notify = notify_buffer(16384)
    inst.connect(1, notify)
    script = {0: [forge_object(host, K + "meteron", []), cfg(host, "UISETTINGS", 5.0)],
              3: [cfg(host, "START")], 20: [cfg(host, "PAUSE")], 24: [cfg(host, "RESET"), cfg(host, "START")]}
    integrating, seen, resync, send_state = False, [], 0, False
    n_hist = 0
    for c in range(40):
        blk = x[c * B:(c + 1) * B].copy()
        got = drive(host, inst, notify, blk, script.get(c, []))
        want = []
        if send_state and c > 0:
            want += ["control"] * 3
            send_state = False
        if c == 0:
            send_state = True
        if c == 3:
            integrating = True
        if c == 20:
            integrating = False
        if c == 24:
            want.append("control")                                      # LV2_RESETRADAR
            seen, resync, integrating = [], 0, True
        if integrating:
            seen.append(blk)
        resync += B
        if resync >= fps_limit or send_state:
            resync %= fps_limit
            if integrating or send_state:
                want.append("sdh_histogram")
            want.append("sdh_information")
        assert [t.split("#")[1] for t, _ in got] == want, (c, [t for t, _ in got], want)
        ctl = [p for t, p in got if t == K + "control"]
        if c == 1:
            assert [(p[K + "controlkey"], p[K + "controlval"]) for p in ctl] == \
                [(CTL["LV2_FTM"], 0.0), (CTL["SAMPLERATE"], 48000.0), (CTL["UISETTINGS"], 5.0)]
        for t, p in got:
            if t == K + "sdh_information":
                assert p[K + "ebu_integrating"] == int(integrating)
                assert p[K + "ebu_integr_time"] == sum(b.size for b in seen)
            if t == K + "sdh_histogram" and seen:
                n_hist += 1
                w = oracle.sigdist(np.concatenate(seen))
                assert np.array_equal(p[K + "sdh_hist_data"], w["bins"]), c
                assert p[K + "sdh_hist_max"] == w["peak_cnt"] and p[K + "sdh_hist_peak"] == w["peak_bin"], c
                assert abs(p[K + "sdh_hist_avg"] - w["avg"]) <= 1e-9 * max(1.0, abs(w["avg"])), c
    assert n_hist >= 10
    # transport follow with auto-reset: rolling starts integration and clears the histogram (:63-101)
    speed = struct.pack("<II", 1, host.urid("http://lv2plug.in/ns/ext/time#Position"))
    speed += struct.pack("<IIII", host.urid("http://lv2plug.in/ns/ext/time#speed"), 0, 4,
                         host.urid("http://lv2plug.in/ns/ext/atom#Float")) + struct.pack("<f", 1.0) + b"\0" * 4
    speed = struct.pack("<II", len(speed), host.urid("http://lv2plug.in/ns/ext/atom#Object")) + speed
    blk = x[:B].copy()
    drive(host, inst, notify, blk, [cfg(host, "PAUSE"), cfg(host, "AUTORESET", 1), cfg(host, "TRANSPORTSYNC", 1)])
    got = drive(host, inst, notify, blk, [speed])
    assert (K + "control", {K + "controlkey": CTL["LV2_RESETRADAR"], K + "controlval": 0.0}) in got
    for t, p in got:
        if t == K + "sdh_information":
            assert p[K + "ebu_integrating"] == 1 and p[K + "ebu_integr_time"] == B
        if t == K + "sdh_histogram":
            assert np.array_equal(p[K + "sdh_hist_data"], oracle.sigdist(blk)["bins"])
    kept = inst.state_save()
    key = host.urid(K + "sdh_state")
    assert struct.unpack("<I", kept[key][0])[0] == (5 | (3 << 8))
    inst.cleanup()
