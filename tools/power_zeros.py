"""tools/power_zeros.py — socket power and shader clock while k_kwtp16 runs on zeros and on the bench programme (5 s each):
is the zero-data launch (no operand toggling) still at the power cap, or is it the kernel's issue-bound floor at the top clock?"""
import subprocess, sys, threading, time
import torch
import meters.lv2_amd as M
S, T, fs = 8192, 480000, 48000.0
buf = torch.zeros((S, T, 2), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def sample(out, stop):
    while not stop.is_set():
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        p = [l.split(":")[-1].strip() for l in r.splitlines() if "Socket Graphics Package Power" in l]
        c = [l.split("(")[-1].split(")")[0] for l in r.splitlines() if "sclk" in l]
        if p and c:
            out.append((float(p[0]), c[0]))


for what in ("zeros", "programme"):
    if what == "programme":
        M.synth_fill_device(buf.data_ptr(), S, T, T, 777, fs, 1, st)
    torch.cuda.synchronize()
    with M.Engine(S, fs, M.METER_EBU | M.METER_TRUEPEAK) as e:
        e.integr_start()
        e.process_device(buf.data_ptr(), T, T, st)
        torch.cuda.synchronize()
        e.timing_enable(True)
        out, stop = [], threading.Event()
        th = threading.Thread(target=sample, args=(out, stop)); th.start()
        t0 = time.time()
        n = 0
        while time.time() - t0 < 5.0:
            for _ in range(20):
                e.process_device(buf.data_ptr(), T, T, st)
            torch.cuda.synchronize(); n += 20
        stop.set(); th.join()
        q = e.timing_query()
        mid = out[len(out) // 4: -len(out) // 4 or None]
        print("%-10s kernel %.3f ms   power %s W   sclk %s" % (what, q["ms_fused"] / q["calls"],
              sorted(p for p, _ in mid)[len(mid) // 2] if mid else "?", mid[len(mid) // 2][1] if mid else "?"), flush=True)
