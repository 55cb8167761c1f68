"""Times the CPU oracle (oracle/picha_oracle.c, one thread) on BASELINE.md's reference cases, on whatever host this runs on.
VERDICT r02 item 2: the port must run at the reference's speed (src/resize.cc compiled unmodified, 91.3 ms per
4K rgba -> 1080p image on the build container)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import oracle
from oracle.binding import BYTES, PIXELS

def cpu():
    for l in open('/proc/cpuinfo'):
        if l.startswith('model name'):
            return l.split(':', 1)[1].strip()
    return '?'

print('# host CPU:', cpu())
rng = np.random.default_rng(1)
cases = [('rgba', 3840, 2160, 1920, 1080, 'cubic', 0.70, 91.3), ('rgba', 3840, 2160, 1920, 1080, 'cubic', 1.0, 115.7),
         ('grey', 3840, 2160, 512, 512, 'lanczos', 1.0, 41.8), ('r16g16b16a16', 7680, 4320, 1920, 1080, 'mitchel', 1.0, 440.0),
         ('rgba', 640, 480, 320, 240, 'cubic', 0.70, 2.69)]
for px, sw, sh, dw, dh, f, fs, ref in cases:
    bpp = BYTES[PIXELS.index(px)]
    src = rng.integers(0, 256, size=sh * ((sw * bpp + 3) & ~3), dtype=np.uint8)
    oracle.resize(src, sw, sh, px, dw, dh, f, fs)
    reps = 3 if sw < 7000 else 1
    t = time.perf_counter()
    for _ in range(reps):
        oracle.resize(src, sw, sh, px, dw, dh, f, fs)
    dt = (time.perf_counter() - t) / reps
    print('%-13s %dx%d -> %dx%d %-8s scale %.2f: %8.2f ms/image  %7.1f Mpix/s (src)   reference (BASELINE.md, survey container): %6.1f ms' % (
        px, sw, sh, dw, dh, f, fs, dt * 1e3, sw * sh / dt / 1e6, ref))
src = rng.integers(0, 256, size=2160 * 3840 * 3, dtype=np.uint8)
oracle.color_convert(src, 3840, 2160, 'rgb', 'grey')
t = time.perf_counter(); oracle.color_convert(src, 3840, 2160, 'rgb', 'grey'); dt = time.perf_counter() - t
print('rgb -> grey   3840x2160 colorConvert: %8.2f ms/image  %7.1f Mpix/s   reference: 24.4 ms' % (dt * 1e3, 3840 * 2160 / dt / 1e6))
