"""Timing probe of on-device refinement (developer tool): M synthetic patterns
simulated from a random 401 x 401 master pattern, starts 1 degree off."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kikuchipy_amd as ka  # noqa: E402
from kikuchipy_amd.indexing._refinement import rotation_from_euler  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=4096)
ap.add_argument("--s", type=int, default=60)
ap.add_argument("--mode", default="ori")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()

rng = np.random.default_rng(3)
# smooth random master pattern: low-pass filtered noise
f = np.fft.rfft2(rng.standard_normal((401, 401)))
ky, kx = np.meshgrid(np.fft.fftfreq(401), np.fft.rfftfreq(401), indexing="ij")
mpd = np.fft.irfft2(f * np.exp(-(kx**2 + ky**2) / (2 * 0.03**2)), s=(401, 401)).astype(np.float32)
mp = ka.EBSDMasterPattern(mpd)
det = ka.EBSDDetector(shape=(a.s, a.s), pc=(0.42, 0.78, 0.5))
eu = np.column_stack([rng.uniform(0.3, 6, a.m), rng.uniform(0.3, 2.8, a.m), rng.uniform(0.3, 6, a.m)])
t0 = time.time()
sim = mp.get_patterns(rotation_from_euler(eu), det, compute=True).data
noisy = sim + 0.3 * sim.std() * rng.standard_normal(sim.shape).astype(np.float32)
lo, hi = noisy.min(), noisy.max()
pats = ((noisy - lo) / (hi - lo) * 255).astype(np.uint8)
print(f"simulated {a.m} patterns in {time.time() - t0:.2f} s", flush=True)
rot0 = rotation_from_euler(eu + np.deg2rad(rng.uniform(-1, 1, eu.shape)))
s = ka.EBSD(pats)
for rep in range(a.reps):
    t0 = time.perf_counter()
    if a.mode == "ori":
        res = s.refine_orientation(rot0, det, mp, verbose=False)
    elif a.mode == "pc":
        sc, nd, ne = s.refine_projection_center(rot0, det, mp, verbose=False)
        res = ka.RefinementResult(sc, ne, None, None, (a.m,))
    else:
        res, nd = s.refine_orientation_projection_center(rot0, det, mp, verbose=False)
    dt = time.perf_counter() - t0
    c = s.context.counters()
    print(f"rep {rep}: wall {dt*1e3:.1f} ms  ({a.m/dt:.0f} patterns/s)  kernel total {c['refine_ms']:.1f} ms  "
          f"mean evals {res.num_evals.mean():.1f}  mean score {res.scores.mean():.4f}", flush=True)
if a.mode == "ori":
    err = np.abs(res.euler - eu)
    print("median |euler error| (deg):", np.rad2deg(np.median(err, axis=0)))
