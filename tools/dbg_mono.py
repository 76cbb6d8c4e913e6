import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import meters.lv2_amd as M
import _signals as sig
x = np.stack([sig.lcg_noise(20000, 500 + s, 0.5) for s in range(2)])
xm = np.ascontiguousarray(x[:2, :, 0])
print(xm.shape, xm.flags['C_CONTIGUOUS'], np.abs(xm).max())
for S in (2, 1):
    with M.Engine(S, 48000.0, M.METER_SPECTR30, n_channels=1) as e:
        e.process(xm[:S])
        r = e.spectrum()
        print(S, r['val'][0][:4], r['val'][0][-3:])
with M.Engine(2, 48000.0, M.METER_SPECTR30, n_channels=2) as e:
    e.process(x)
    print('stereo', e.spectrum()['val'][0][-3:])
