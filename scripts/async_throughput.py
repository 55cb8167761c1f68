"""Drop-in host-pointer throughput: synchronous calls vs the batching async dispatcher (PCIe included)."""
import sys, time, threading, ctypes
sys.path.insert(0, '.')
import numpy as np
import picha_amd
from picha_amd import _native as N
rng = np.random.default_rng(1)
for (w, h, dw, dh, n) in [(640, 480, 320, 240, 2000), (1920, 1080, 960, 540, 400), (3840, 2160, 1920, 1080, 100)]:
    imgs = [picha_amd.Image(width=w, height=h, pixel='rgba', data=rng.integers(0, 256, h * w * 4, dtype=np.uint8)) for _ in range(8)]
    opts = {'width': dw, 'height': dh}
    for _ in range(3): picha_amd.resizeSync(imgs[0], opts)
    t0 = time.perf_counter()
    for i in range(n): picha_amd.resizeSync(imgs[i % 8], opts)
    ts = time.perf_counter() - t0
    sem = threading.Semaphore(0)
    cb = lambda err, out: sem.release()
    b0, r0 = ctypes.c_uint64(), ctypes.c_uint64(); N.lib().picha_hip_async_stats(ctypes.byref(b0), ctypes.byref(r0))
    t0 = time.perf_counter()
    for i in range(n): picha_amd.resize(imgs[i % 8], opts, cb)
    for i in range(n): sem.acquire()
    ta = time.perf_counter() - t0
    b1, r1 = ctypes.c_uint64(), ctypes.c_uint64(); N.lib().picha_hip_async_stats(ctypes.byref(b1), ctypes.byref(r1))
    print('%dx%d rgba -> %dx%d x%d: sync %.3f ms/img (%.0f Mpix/s) | async %.3f ms/img (%.0f Mpix/s), %d batches (avg %.1f img/batch)' % (
        w, h, dw, dh, n, ts / n * 1e3, w * h * n / ts / 1e6, ta / n * 1e3, w * h * n / ta / 1e6, b1.value - b0.value, n / max(1, b1.value - b0.value)))
