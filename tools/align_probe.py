"""tools/align_probe.py — what does a lane's misaligned 128-byte read cost k_seg?  The bench shape at 48 kHz (every segment
starts on a cache line) against the same audio with the buffer's base moved by 1, 8 and 13 frames (8, 64, 104 bytes), and the
44.1 kHz shape (2205-frame fragments: a lane's start is 8 x (13 p mod 16) bytes into a line).  Kernel ms by HIP events.  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import meters.lv2_amd as M

def run(fs, off, meters, S=8192, steps=6):
    T = int(fs) * 10
    stride = T + 16
    flat = torch.empty(S * stride * 2 + 64, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ptr = flat.data_ptr() + 8 * off
    M.synth_fill_device(ptr, S, T, stride, 777, fs, 1, st)
    with M.Engine(S, fs, meters) as e:
        if meters & M.METER_EBU: e.integr_start()
        e.process_device(ptr, T, stride, st); torch.cuda.synchronize()
        e.timing_enable(True)
        for _ in range(steps): e.process_device(ptr, T, stride, st)
        torch.cuda.synchronize()
        pc = e.timing_calls()
        ms = float(sorted(pc[:, 0])[len(pc) // 2])
        print("fs %6.0f  base + %2d frames  %-7s kernel median %.3f ms (min %.3f)  %.1f %% of 8 TB/s  seg %s" % (
            fs, off, "ebu+tp" if meters & M.METER_EBU else "tp", ms, pc[:, 0].min(), 100 * S * T * 8 / (ms * 1e-3) / 8e12, e.seg_stats()), flush=True)

if __name__ == "__main__":
    both = M.METER_EBU | M.METER_TRUEPEAK
    for rep in range(2):
        for off in (0, 1, 8, 13):
            run(48000.0, off, both)
        run(44100.0, 0, both)
    run(48000.0, 0, M.METER_TRUEPEAK); run(48000.0, 13, M.METER_TRUEPEAK); run(44100.0, 0, M.METER_TRUEPEAK)
