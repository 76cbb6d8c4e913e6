#!/usr/bin/env python3
"""tools/gen_ttl.py <bundle_dir> — write manifest.ttl + meters_amd.ttl for the in-scope plugins so a
stock LV2 host can load meters.lv2_amd/lib/meters_amd.so as a bundle.

Port indices, symbols and ranges are the reference's (lv2ttl/meters.lv2.ttl.in: VU :13-43 etc.,
dBTP :1906-1975, EBUr128 :614-660, spectr30 :1292-1841), generated here from tables rather than
copied.  The plugins keep the reference's URIs, so host sessions that reference them keep working;
no UI is declared (the GUIs are out of scope)."""
import os
import sys

MTR = "http://gareus.org/oss/lv2/meters#"
BANDS = [25, 31, 40, 50, 63, 80, 100, 125, 160, 200, 250, 315, 400, 500, 630, 800, 1000, 1250, 1600, 2000,
         2500, 3150, 4000, 5000, 6300, 8000, 10000, 12500, 16000, 20000]

PREFIX = """@prefix atom: <http://lv2plug.in/ns/ext/atom#> .
@prefix bufsz: <http://lv2plug.in/ns/ext/buf-size#> .
@prefix opts: <http://lv2plug.in/ns/ext/options#> .
@prefix doap: <http://usefulinc.com/ns/doap#> .
@prefix lv2:  <http://lv2plug.in/ns/lv2core#> .
@prefix mtr:  <%s> .
@prefix pg:   <http://lv2plug.in/ns/ext/port-groups#> .
@prefix rdfs: <http://www.w3.org/2000/01/rdf-schema#> .
@prefix rsz:  <http://lv2plug.in/ns/ext/resize-port#> .
@prefix time: <http://lv2plug.in/ns/ext/time#> .
@prefix urid: <http://lv2plug.in/ns/ext/urid#> .

""" % MTR


def ctl(idx, sym, name, direction, lo=None, hi=None, default=None):
    s = "\t\ta lv2:ControlPort , lv2:%sPort ;\n\t\tlv2:index %d ;\n\t\tlv2:symbol \"%s\" ;\n\t\tlv2:name \"%s\" ;\n" % (
        direction, idx, sym, name)
    if default is not None:
        s += "\t\tlv2:default %s ;\n" % default
    if lo is not None:
        s += "\t\tlv2:minimum %s ;\n\t\tlv2:maximum %s ;\n" % (lo, hi)
    return s


def audio(idx, sym, name, direction):
    return "\t\ta lv2:AudioPort , lv2:%sPort ;\n\t\tlv2:index %d ;\n\t\tlv2:symbol \"%s\" ;\n\t\tlv2:name \"%s\" ;\n" % (
        direction, idx, sym, name)


def atom_port(idx, sym, name, direction, extra=""):
    return ("\t\ta atom:AtomPort , lv2:%sPort ;\n\t\tatom:bufferType atom:Sequence ;\n\t\tlv2:designation lv2:control ;\n"
            "%s\t\tlv2:index %d ;\n\t\tlv2:symbol \"%s\" ;\n\t\tlv2:name \"%s\" ;\n" % (direction, extra, idx, sym, name))


def plugin(uri, name, comment, ports, extra=""):
    """The reference's plugins all carry lv2:hardRTCapable (lv2ttl/meters.lv2.ttl.in:609).  Here only the ones whose run()
    stays on the host CPU do: a run() that copies to the GPU, launches kernels and waits for them takes driver locks,
    however short it is (26 - 132 us per 1024-frame block)."""
    body = " ] , [\n".join(ports)
    # the GPU plugins warm their engine up for the host's largest block when it says what that is (csrc/lv2_plugins.h: lv2_max_block)
    rt = ("\tlv2:optionalFeature opts:options ;\n\topts:supportedOption bufsz:maxBlockLength ;\n" if "GPU" in comment
          else "\tlv2:optionalFeature lv2:hardRTCapable ;\n")
    return ("mtr:%s\n\ta lv2:Plugin , lv2:AnalyserPlugin , doap:Project ;\n\tdoap:license <http://usefulinc.com/doap/licenses/gpl> ;\n"
            "\tdoap:name \"%s\" ;\n\tlv2:project <http://gareus.org/oss/lv2/meters> ;\n%s%s"
            "\tlv2:port [\n%s\t] ;\n\trdfs:comment \"%s\"\n\t.\n\n" % (uri, name, rt, extra, body, comment))


def spectr(stereo):
    p = [ctl(i, "band%d" % b, "%d Hz" % b, "Output", -100.0, 6.0, -100.0 if (i == 0 and not stereo) else None) for i, b in enumerate(BANDS)]   # (the mono plugin's band25 alone states a default)
    p += [ctl(30 + i, "max%d" % b, "%d Hz peak" % b, "Output", -100.0, 6.0) for i, b in enumerate(BANDS)]
    p += [ctl(60, "UIspeed", "Integration speed", "Input", 0.02, 15.0, 1.0),
          ctl(61, "UIreset", "Peak hold reset", "Input", -4.0, 4.0, -4.0),
          ctl(62, "UIgain", "UI gain", "Input", -12.0, 32.0, 0.0),
          ctl(63, "UImiscstate", "UI state", "Input", 0, 256, 1)]
    if stereo:
        p += [audio(64, "inL", "InL", "Input"), audio(65, "outL", "OutL", "Output"),
              audio(66, "inR", "InR", "Input"), audio(67, "outR", "OutR", "Output")]
    else:
        p += [audio(64, "in", "In", "Input"), audio(65, "out", "Out", "Output")]
    return p


NEEDLES = ["BBCmono", "BBCstereo", "EBUmono", "EBUstereo", "DINmono", "DINstereo", "NORmono", "NORstereo", "COR", "BBCM6",
           "K12mono", "K14mono", "K20mono", "K12stereo", "K14stereo", "K20stereo"]
NEEDLE_NAMES = {"BBC": "BBC PPM", "EBU": "EBU PPM", "DIN": "DIN PPM", "NOR": "Nordic PPM",
                "K12": "K12/RMS Meter", "K14": "K14/RMS Meter", "K20": "K20/RMS Meter"}


REF_LEVEL = {"BBC": -18.0, "EBU": -18.0, "DIN": -15.0, "NOR": -18.0}     # default of port 0 per scale (lv2ttl/meters.lv2.ttl.in:112-546)


def needles():
    """The needle meters share VU's port map (lv2ttl/meters.lv2.ttl.in:112-602; port indices of src/meters.cc:59-70): the
    reference level, audio through, the level in [0, 1]; the K-meters (:1976-2400) use port 0 as the peak-hold reset
    handshake (-4 .. 4) and add peak and hold outputs (mono: on the otherwise unused indices 4 and 5)."""
    t = ""
    for n in NEEDLES:
        if n == "COR":
            ports = [ctl(0, "unused", "unused", "Input", 0.0, 1.0, 0.0), audio(1, "inL", "InL", "Input"),
                     audio(2, "outL", "OutL", "Output"), ctl(3, "correlation", "Correlation", "Output", -1.0, 1.0),
                     audio(4, "inR", "InR", "Input"), audio(5, "outR", "OutR", "Output")]
            t += plugin(n, "Stereo Phase-Correlation Meter (MI355X build)", "Stereo phase correlation; host CPU.", ports)
            continue
        if n == "BBCM6":
            ports = [ctl(0, "ref", "Reference level", "Input", -30.0, 0.0, -18.0), audio(1, "inL", "InL", "Input"),
                     audio(2, "outL", "OutL", "Output"), ctl(3, "levelM6", "Level M", "Output", 0.0, 1.0),
                     audio(4, "inR", "InR", "Input"), audio(5, "outR", "OutR", "Output"),
                     ctl(6, "levelS6", "Level S", "Output", 0.0, 1.0), ctl(7, "s20", "S +20 dB", "Input", 0, 1, 0)]
            t += plugin(n, "BBC M-6 PPM (MI355X build)", "Mid / side peak programme meter; host CPU.", ports)
            continue
        kind, stereo = n[:3], n.endswith("stereo")
        label = "%s (%s, MI355X build)" % (NEEDLE_NAMES[kind], "Stereo" if stereo else "Mono")
        k = kind[0] == "K"
        ports = [ctl(0, "ref", "Peak hold reset" if k else "Reference level", "Input", -4.0, 4.0, -4.0) if k
                 else ctl(0, "ref", "Reference level", "Input", -30.0, 0.0, REF_LEVEL[kind])]
        if not stereo:
            ports += [audio(1, "in", "In", "Input"), audio(2, "out", "Out", "Output"), ctl(3, "level1", "Level", "Output", 0.0, 1.0)]
            if k:
                ports += [ctl(4, "peak", "Peak", "Output", 0.0, 1.0), ctl(5, "hold", "Peak hold", "Output", 0.0, 1.0)]
        else:
            ports += [audio(1, "inL", "InL", "Input"), audio(2, "outL", "OutL", "Output"), ctl(3, "levelL", "Level L", "Output", 0.0, 1.0),
                      audio(4, "inR", "InR", "Input"), audio(5, "outR", "OutR", "Output"), ctl(6, "levelR", "Level R", "Output", 0.0, 1.0)]
            if k:
                ports += [ctl(7, "peakL", "Peak L", "Output", 0.0, 1.0), ctl(8, "peakR", "Peak R", "Output", 0.0, 1.0),
                          ctl(9, "hold", "Peak hold", "Output", 0.0, 1.0)]
        t += plugin(n, label, "Needle meter ballistics on the host CPU.", ports)
    return t


DR14S = ["dr14mono", "dr14stereo", "TPnRMSmono", "TPnRMSstereo"]
# the reference ships metadata for 8, 5, 4 and 3 channels (lv2ttl/manifest.lv2.ttl.in:163-181; src/surmeter.c also answers
# surround7 / surround6, which no bundle declares) — and these default channel pairs for the four correlation meters
SURS = {"surround8": [(0, 1), (2, 3), (4, 5), (6, 7)], "surround5": [(1, 4), (2, 3), (0, 4), (0, 1)],
        "surround4": [(0, 2), (1, 3), (0, 3), (0, 1)], "surround3": [(0, 1), (0, 2), (1, 2), (0, 0)]}


def surrounds():
    """src/surmeter.c:74-113, lv2ttl/meters.lv2.ttl.in (surround8 ff.): port 0 the RMS gain, 1..12 = four x (channel A,
    channel B, correlation out), then per channel in, out, rms, peak — everything counted from 1 in the symbols."""
    t = ""
    for n, pairs in SURS.items():
        chn = int(n[-1])
        ports = [ctl(0, "rmsgain", "RMS gain", "Input", -20.0, 20.0, 0.0)]
        for c, (pa, pb) in enumerate(pairs):
            ports += [ctl(1 + 3 * c, "cor%dA" % (c + 1), "Correlation %d channel A" % (c + 1), "Input", 0, chn - 1, pa),
                      ctl(2 + 3 * c, "cor%dB" % (c + 1), "Correlation %d channel B" % (c + 1), "Input", 0, chn - 1, pb),
                      ctl(3 + 3 * c, "cor%d" % (c + 1), "Correlation %d" % (c + 1), "Output", -1.0, 1.0)]
        for c in range(chn):
            b = 13 + 4 * c
            ports += [audio(b, "in%d" % (c + 1), "In %d" % (c + 1), "Input"), audio(b + 1, "out%d" % (c + 1), "Out %d" % (c + 1), "Output"),
                      ctl(b + 2, "rms%d" % (c + 1), "RMS %d" % (c + 1), "Output", 0.0, 1.0),
                      ctl(b + 3, "peak%d" % (c + 1), "Peak %d" % (c + 1), "Output", 0.0, 1.0)]
        t += plugin(n, "Surround Meter (%d channels, MI355X build)" % chn, "K-meter per channel and pairwise correlation; host CPU.", ports)
    return t


def dr14s():
    """src/dr14.c:27-43, lv2ttl/meters.lv2.ttl.in (dr14mono ff.): control atom port, three control ports, then per channel audio in /
    out and the bar values.  dr14*: five per channel (true peak, its maximum, RMS, the score, DR), stereo adds the averaged DR;
    TPnRMS*: four per channel — the stereo variant keeps index 10 (the first channel's DR slot) as an unused output so that the
    second channel starts at 11 as in dr14stereo, and neither declares a DR port."""
    t = ""
    for n in DR14S:
        stereo, dr = n.endswith("stereo"), n.startswith("dr14")
        ports = [atom_port(0, "control", "UI to plugin communication", "Input", "\t\tatom:supports time:Position ;\n"),
                 ctl(1, "host_transport" if dr else "unused1", "Reset when the transport starts" if dr else "unused", "Input", 0, 1, 1),
                 ctl(2, "reset", "Reset", "Input", 0, 1, 0),
                 ctl(3, "blkcnt", "Integration time [s]", "Output", 0.0, 3600.0 if (dr or stereo) else 1.0)]
        for c in range(2 if stereo else 1):
            b = 4 + 7 * c
            suf = str(c + 1) if stereo else ""
            ports += [audio(b, "in%d" % (c + 1), "In %d" % (c + 1), "Input"), audio(b + 1, "out%d" % (c + 1), "Out %d" % (c + 1), "Output"),
                      ctl(b + 2, "dBTP_m" + suf, "True peak " + suf, "Output", -80.0, 6.0),
                      ctl(b + 3, "dBTP_p" + suf, "True peak max " + suf, "Output", -80.0, 6.0),
                      ctl(b + 4, "dBRMS_m" + suf, "RMS " + suf, "Output", -80.0, 0.0)]
            if dr:
                ports += [ctl(b + 5, "dBRMS_p" + suf, "RMS score " + suf, "Output", -80.0, 0.0),
                          ctl(b + 6, "dr" + suf, "DR " + suf, "Output", 0.0, 20.0)]
            else:
                # (the reference names the held K-meter peak dBRMS_p1 on the first stereo channel and dBFS_p elsewhere)
                ports += [ctl(b + 5, "dBRMS_p1" if (stereo and c == 0) else "dBFS_p" + (suf if stereo else ""), "Peak hold " + suf, "Output", -80.0, 0.0)]
                if stereo and c == 0:
                    ports += [ctl(b + 6, "unused2", "unused", "Output", 0.0, 1.0)]
        if stereo and dr:
            ports.append(ctl(18, "dr", "DR", "Output", 0.0, 20.0))
        label = ("DR-14 Crest-Factor Meter" if dr else "True-Peak and RMS Meter") + \
            (" (Stereo, MI355X build)" if stereo else " (Mono, MI355X build)")
        t += plugin(n, label, "True-peak ballistics, K-meter detector and the 3 s window statistics on the GPU.", ports,
                    extra="\tlv2:requiredFeature urid:map ;\n")
    return t


def main(out):
    os.makedirs(out, exist_ok=True)
    plugs = ["VUmono", "VUstereo", "EBUr128", "spectr30mono", "dBTPmono", "dBTPstereo", "spectr30stereo",
             "SigDistHist", "bitmeter"] + NEEDLES + DR14S + list(SURS)
    man = PREFIX + "".join("mtr:%s\n\ta lv2:Plugin ;\n\tlv2:binary <meters_amd.so> ;\n\trdfs:seeAlso <meters_amd.ttl> .\n\n" % p
                           for p in plugs)
    open(os.path.join(out, "manifest.ttl"), "w").write(man)

    t = PREFIX
    t += plugin("VUmono", "VU Meter (Mono, MI355X build)", "Volume unit meter; needle ballistics on the host CPU.",
                [ctl(0, "ref", "Reference level", "Input", -30.0, 0.0, -22.0), audio(1, "in", "In", "Input"),
                 audio(2, "out", "Out", "Output"), ctl(3, "level1", "Level", "Output", 0.0, 1.0)])
    t += plugin("VUstereo", "VU Meter (Stereo, MI355X build)", "Volume unit meter; needle ballistics on the host CPU.",
                [ctl(0, "ref", "Reference level", "Input", -30.0, 0.0, -22.0), audio(1, "inL", "InL", "Input"),
                 audio(2, "outL", "OutL", "Output"), ctl(3, "levelL", "Level L", "Output", 0.0, 1.0),
                 audio(4, "inR", "InR", "Input"), audio(5, "outR", "OutR", "Output"),
                 ctl(6, "levelR", "Level R", "Output", 0.0, 1.0)])
    t += plugin("EBUr128", "EBU R128 Meter (MI355X build)",
                "Stereo loudness meter according to EBU R128 with 4x true peak; DSP on the GPU.",
                [atom_port(0, "control", "UI to plugin communication", "Input", "\t\tatom:supports time:Position ;\n"),
                 atom_port(1, "notify", "plugin to UI communication", "Output", "\t\trsz:minimumSize 4096 ;\n"),
                 audio(2, "inL", "InL", "Input"), audio(3, "outL", "OutL", "Output"),
                 audio(4, "inR", "InR", "Input"), audio(5, "outR", "OutR", "Output")],
                extra="\tlv2:requiredFeature urid:map ;\n")
    t += plugin("spectr30mono", "1/3 Octave Spectrum (Mono, MI355X build)", "30-band 1/3-octave analyser; DSP on the GPU.", spectr(False))
    t += plugin("dBTPmono", "True Peak Meter (Mono, MI355X build)", "4x oversampled true-peak meter; DSP on the GPU.",
                [ctl(0, "ref", "Reset / notify", "Input", -4.0, 4.0, -4.0), audio(1, "in", "In", "Input"),
                 audio(2, "out", "Out", "Output"), ctl(3, "level1", "Level", "Output", 0.0, 1.0),
                 ctl(4, "peak", "Peak", "Output", 0.0, 1.0)])
    t += plugin("dBTPstereo", "True Peak Meter (Stereo, MI355X build)", "4x oversampled true-peak meter; DSP on the GPU.",
                [ctl(0, "ref", "Reset / notify", "Input", -4.0, 4.0, -4.0), audio(1, "inL", "InL", "Input"),
                 audio(2, "outL", "OutL", "Output"), ctl(3, "levelL", "Level L", "Output", 0.0, 1.0),
                 audio(4, "inR", "InR", "Input"), audio(5, "outR", "OutR", "Output"),
                 ctl(6, "levelR", "Level R", "Output", 0.0, 1.0), ctl(7, "peakL", "Peak L", "Output", 0.0, 1.0),
                 ctl(8, "peakR", "Peak R", "Output", 0.0, 1.0)])
    t += plugin("spectr30stereo", "1/3 Octave Spectrum (Stereo, MI355X build)", "30-band 1/3-octave analyser; DSP on the GPU.", spectr(True))
    # SigDistHist lv2ttl/meters.lv2.ttl.in:3255-3300, bitmeter :3376-3420: control / notify atom ports + one audio pair
    for uri, name, comment in (("SigDistHist", "Signal Distribution Histogram (MI355X build)",
                                "Histogram of sample values with mean / variance; counting on the GPU."),
                               ("bitmeter", "Bitmeter (MI355X build)",
                                "IEEE-754 bit-usage statistics of a float stream; counting on the GPU.")):
        t += plugin(uri, name, comment,
                    [atom_port(0, "control", "UI to plugin communication", "Input", "\t\tatom:supports time:Position ;\n"),
                     atom_port(1, "notify", "plugin to UI communication", "Output", "\t\trsz:minimumSize 8192 ;\n"),
                     audio(2, "in", "In", "Input"), audio(3, "out", "Out", "Output")],
                    extra="\tlv2:requiredFeature urid:map ;\n")
    t += needles()
    t += dr14s()
    t += surrounds()
    open(os.path.join(out, "meters_amd.ttl"), "w").write(t)
    print("wrote", out)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "meters_amd.lv2")
