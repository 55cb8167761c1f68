"""Latency of the drop-in host-pointer path (H2D + kernel + D2H per call), PCIe included."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import picha_amd
rng = np.random.default_rng(1)
for (w, h, dw, dh, pix) in [(640, 480, 320, 240, 'rgba'), (3840, 2160, 1920, 1080, 'rgba'), (76, 50, 32, 24, 'rgb')]:
    bpp = {'rgba': 4, 'rgb': 3}[pix]
    img = picha_amd.Image(width=w, height=h, pixel=pix, data=rng.integers(0, 256, h * ((w * bpp + 3) & ~3), dtype=np.uint8))
    for _ in range(3): picha_amd.resizeSync(img, {'width': dw, 'height': dh})
    n = 50
    t0 = time.perf_counter()
    for _ in range(n): picha_amd.resizeSync(img, {'width': dw, 'height': dh})
    dt = (time.perf_counter() - t0) / n
    print('resizeSync %dx%d %s -> %dx%d : %.3f ms/call  %.1f Mpix/s (src), PCIe-inclusive' % (w, h, pix, dw, dh, dt * 1e3, w * h / dt / 1e6))
    for _ in range(3): picha_amd.colorConvertSync(img, {'pixel': 'grey'})
    t0 = time.perf_counter()
    for _ in range(n): picha_amd.colorConvertSync(img, {'pixel': 'grey'})
    dt = (time.perf_counter() - t0) / n
    print('colorConvertSync %dx%d %s -> grey : %.3f ms/call' % (w, h, pix, dt * 1e3))

# cfg3's pair through the host API: two calls (the grey 4K intermediate crosses PCIe twice) vs one chained call
img = picha_amd.Image(width=3840, height=2160, pixel='rgb', data=rng.integers(0, 256, 2160 * 3840 * 3, dtype=np.uint8))
o = {'pixel': 'grey', 'width': 512, 'height': 512, 'filter': 'lanczos'}
for _ in range(3):
    picha_amd.convertResizeSync(img, o); picha_amd.resizeSync(picha_amd.colorConvertSync(img, o), o)
n = 30
t0 = time.perf_counter()
for _ in range(n): two = picha_amd.resizeSync(picha_amd.colorConvertSync(img, o), o)
dt2 = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n): one = picha_amd.convertResizeSync(img, o)
dt1 = (time.perf_counter() - t0) / n
assert one.equalPixels(two)
print('4K rgb -> grey -> 512x512 lanczos: colorConvertSync + resizeSync %.3f ms, convertResizeSync %.3f ms (identical pixels)' % (dt2 * 1e3, dt1 * 1e3))
