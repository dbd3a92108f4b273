"""A 200 x 200 map through the Python API (developer tool): wall time of every stage of the
tutorial pipeline, and the engine counters of the indexing call."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kikuchipy_amd as ka
rng = np.random.default_rng(0)
f = np.fft.rfft2(rng.standard_normal((2, 401, 401)))
ky, kx = np.meshgrid(np.fft.fftfreq(401), np.fft.rfftfreq(401), indexing="ij")
mpd = np.fft.irfft2(f * np.exp(-(kx**2 + ky**2) / (2 * 0.03**2)), s=(401, 401)).astype(np.float32)
mp = ka.EBSDMasterPattern(mpd, hemisphere="both", phase_name="x")
det = ka.EBSDDetector(shape=(60, 60), pc=(0.42, 0.78, 0.5), sample_tilt=70)
q = rng.standard_normal((100000, 4)); q /= np.linalg.norm(q, axis=1)[:, None]
sim = mp.get_patterns(q, det)
# a 200 x 200 map: noisy projections of 40 000 of the dictionary orientations
pick = rng.integers(0, 100000, 40000)
t0 = time.perf_counter()
pats = mp.get_patterns(q[pick], det, compute=True).data
print(f"get_patterns(compute=True) 40000: {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
noisy = pats + 0.2 * pats.std() * rng.standard_normal(pats.shape).astype(np.float32)
exp = ((noisy - noisy.min()) / (noisy.max() - noisy.min()) * 255).astype(np.uint8).reshape(200, 200, 60, 60)
s = ka.EBSD(exp, static_background=np.full((60, 60), 100, np.uint8))
for rep in range(3):
    s.data = exp.copy()
    t = [time.perf_counter()]
    s.remove_static_background(); t.append(time.perf_counter())
    s.remove_dynamic_background(); t.append(time.perf_counter())
    res = s.dictionary_indexing(sim, keep_n=20, verbose=False); t.append(time.perf_counter())
    osm = ka.orientation_similarity_map(res); t.append(time.perf_counter())
    ref = s.refine_orientation(res, det, mp, verbose=False); t.append(time.perf_counter())
    d = np.diff(t) * 1e3
    print(f"rep {rep}: static {d[0]:.1f}  dynamic {d[1]:.1f}  DI {d[2]:.1f}  OSM {d[3]:.1f}  refine {d[4]:.1f} ms   hit rate {np.mean(res.simulation_indices[:,0]==pick):.3f} refined score {ref.scores.mean():.3f}", flush=True)
metric = ka.NormalizedCrossCorrelationMetric()
metric.context.set_profiling(True)
for rep in range(3):
    metric.context.reset_counters()
    t0 = time.perf_counter()
    res = s.dictionary_indexing(sim, metric=metric, keep_n=20, verbose=False)
    dt = time.perf_counter() - t0
    c = metric.context.counters()
    print(f"DI {dt*1e3:.1f} ms: match {c['match_ms']:.1f} prep {c['prep_ms']:.2f} merge {c['merge_ms']:.2f} project {c['project_ms']:.2f} launches {c['match_launches']} nsplit {c['match_nsplit']} grid {c['match_grid']} flops {c['match_flops']:.3e}", flush=True)
