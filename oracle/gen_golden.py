"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (container only).

TEST INFRASTRUCTURE.  Run with:

    /opt/conda/bin/python3.9 -W ignore oracle/gen_golden.py
    python -W ignore oracle/gen_golden.py refinement     # SciPy 1.15 solver results, see gen_refinement

It loads the reference's modules unmodified through `oracle/ref_shim.py` and
stores inputs + the reference's outputs as small fixtures.  The fixtures are
data (inputs and expected outputs); nothing of the reference's source travels.
Large synthetic inputs are regenerated from a seed at test time and verified
against the SHA-256 stored beside the expected outputs.
"""

import contextlib
import hashlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
ref = ref_shim.load_reference()
NCC = ref["ncc"].NormalizedCrossCorrelationMetric
NDP = ref["ndp"].NormalizedDotProductMetric
di = ref["di"]
pat = ref["pattern"]
Window = ref["window"].Window


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_di(exp, dic, metric="ncc", keep_n=20, n_per_iteration=None,
           navigation_mask=None, signal_mask=None, dtype=np.float32):
    """What EBSD.dictionary_indexing does (signals/ebsd.py:1921-1981) around
    the reference's `_dictionary_indexing`, which is what runs here."""
    nav_shape = exp.shape[:-2]
    n_dict = dic.shape[0]
    if n_per_iteration is None:
        n_per_iteration = n_dict
    cls = {"ncc": NCC, "ndp": NDP}[metric]
    m = cls()
    m.n_experimental_patterns = max(int(np.prod(nav_shape)), 1)
    m.n_dictionary_patterns = max(n_dict, 1)
    if navigation_mask is not None:
        m.navigation_mask = navigation_mask
    if signal_mask is not None:
        m.signal_mask = signal_mask
    m.dtype = dtype
    m.raise_error_if_invalid()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        xmap = di._dictionary_indexing(
            experimental=exp, experimental_nav_shape=nav_shape, dictionary=dic,
            step_sizes=(1,) * len(nav_shape), dictionary_xmap=ref_shim.FakeDictionaryXmap(),
            metric=m, keep_n=keep_n, n_per_iteration=n_per_iteration,
        )
    prop = xmap.kw["prop"]
    return (np.asarray(prop["scores"]), np.asarray(prop["simulation_indices"]),
            buf.getvalue(), repr(m))


# ------------------------------------------------------------------ dummy data
# tests' `dummy_signal` / `dummy_background` (conftest.py:166-229): data only
DUMMY = np.array(
    [5, 6, 5, 7, 6, 5, 6, 1, 0, 9, 7, 8, 7, 0, 8, 8, 7, 6, 0, 3, 3, 5, 2,
     9, 3, 3, 9, 8, 1, 7, 6, 4, 8, 8, 2, 2, 4, 0, 9, 0, 1, 0, 2, 2, 5, 8,
     6, 0, 4, 7, 7, 7, 6, 0, 4, 1, 6, 3, 4, 0, 1, 1, 0, 5, 9, 8, 4, 6, 0,
     2, 9, 2, 9, 4, 3, 6, 5, 6, 2, 5, 9], dtype=np.uint8).reshape((3, 3, 3, 3))
DUMMY_BG = np.array([5, 4, 5, 4, 3, 4, 4, 4, 3], dtype=np.uint8).reshape((3, 3))


def gen_dummy_di():
    out = {"dummy": DUMMY, "dummy_bg": DUMMY_BG}
    dic = DUMMY.reshape(-1, 3, 3)
    sig_mask = np.array([[0, 0, 0], [0, 1, 0], [0, 0, 0]], dtype=bool)
    nav_mask = np.array([[0, 0, 0], [0, 1, 0], [0, 0, 0]], dtype=bool)
    cases = {
        "ndp_all": dict(metric="ndp"),
        "ncc_all": dict(metric="ncc"),
        "ncc_sigmask_f64_it2": dict(metric="ncc", dtype=np.float64, n_per_iteration=2,
                                    signal_mask=sig_mask),
        "ndp_sigmask": dict(metric="ndp", signal_mask=sig_mask),
        "ndp_it2": dict(metric="ndp", n_per_iteration=2),
        "ncc_navmask_k1": dict(metric="ncc", keep_n=1, navigation_mask=nav_mask),
        "ndp_navmask_inv": dict(metric="ndp", navigation_mask=~nav_mask),
        "ncc_it4_k3": dict(metric="ncc", keep_n=3, n_per_iteration=4),
    }
    for name, kw in cases.items():
        s, i, msg, rep = run_di(DUMMY, dic, **kw)
        out[f"{name}__scores"] = s
        out[f"{name}__indices"] = i
        out[f"{name}__msg"] = np.array(msg)
        out[f"{name}__repr"] = np.array(rep)
    out["sig_mask"] = sig_mask
    out["nav_mask"] = nav_mask
    np.savez_compressed(os.path.join(OUT, "di_dummy.npz"), **out)
    print("di_dummy", len(out))


# ------------------------------------------------------------------ synthetic DI
def synth(seed, m, n, sy=60, sx=60):
    rng = np.random.default_rng(seed)
    exp = rng.integers(0, 256, (m, sy, sx)).astype(np.uint8)
    dic = rng.random((n, sy, sx)).astype(np.float32)
    return exp, dic


def gen_synth_di():
    out = {}
    seed, m, n = 2024, 48, 3000
    exp, dic = synth(seed, m, n)
    out["seed"], out["m"], out["n"] = seed, m, n
    out["exp_sha"], out["dic_sha"] = np.array(sha(exp)), np.array(sha(dic))
    circ = ~Window("circular", (60, 60)).astype(bool)
    out["circular_mask"] = np.asarray(circ)
    nav = np.zeros((6, 8), dtype=bool)
    nav[1, 2] = nav[4, 7] = nav[0, 0] = True
    out["nav_mask"] = nav
    cases = {
        "ncc_k20": dict(metric="ncc", keep_n=20),
        "ncc_k1": dict(metric="ncc", keep_n=1),
        "ncc_k5_it700": dict(metric="ncc", keep_n=5, n_per_iteration=700),
        "ndp_k20": dict(metric="ndp", keep_n=20),
        "ndp_k5_it1000": dict(metric="ndp", keep_n=5, n_per_iteration=1000),
        "ncc_k20_circ": dict(metric="ncc", keep_n=20, signal_mask=np.asarray(circ)),
        "ncc_k20_circ_it999": dict(metric="ncc", keep_n=20, signal_mask=np.asarray(circ),
                                   n_per_iteration=999),
        "ncc_k10_f64": dict(metric="ncc", keep_n=10, dtype=np.float64),
        "ndp_k50": dict(metric="ndp", keep_n=50),
    }
    for name, kw in cases.items():
        s, i, msg, rep = run_di(exp, dic, **kw)
        out[f"{name}__scores"] = s
        out[f"{name}__indices"] = i
    # with navigation mask on a 2-D map
    s, i, msg, rep = run_di(exp.reshape(6, 8, 60, 60), dic, metric="ncc", keep_n=7,
                            navigation_mask=nav, n_per_iteration=1500)
    out["ncc_k7_nav__scores"], out["ncc_k7_nav__indices"] = s, i
    out["ncc_k7_nav__msg"] = np.array(msg)
    # the survey's anchor
    rng = np.random.default_rng(0)
    e0 = rng.integers(0, 256, (3, 3, 60, 60)).astype(np.uint8)
    d0 = rng.random((1000, 60, 60)).astype(np.float32)
    for met in ("ncc", "ndp"):
        s, i, _, _ = run_di(e0, d0, metric=met, keep_n=5)
        out[f"anchor_{met}__scores"], out[f"anchor_{met}__indices"] = s, i
    out["anchor_exp_sha"], out["anchor_dic_sha"] = np.array(sha(e0)), np.array(sha(d0))
    np.savez_compressed(os.path.join(OUT, "di_synth.npz"), **out)
    print("di_synth", len(out))


# ------------------------------------------------------------------ Ni small (config 1)
def load_ni():
    import h5py

    p = os.path.join(ref_shim.SRC, "data", "kikuchipy_h5ebsd", "patterns.h5")
    with h5py.File(p, "r") as f:
        pats = f["Scan 1/EBSD/Data/patterns"][()]
        bg = f["Scan 1/EBSD/Header/static_background"][()]
    return np.asarray(pats).reshape(3, 3, 60, 60), np.asarray(bg)


def static_bg_ref(patterns, bg, operation, scale_bg):
    """signals/ebsd.py:518-573 around the reference's per-pattern kernel."""
    dtype_out = patterns.dtype.type
    omin, omax = pat.dtype_range[dtype_out]
    f = (pat._remove_static_background_subtract if operation == "subtract"
         else pat._remove_static_background_divide)
    bgf = bg.astype(np.float32)
    flat = patterns.reshape((-1,) + patterns.shape[-2:])
    out = np.stack([f(p, bgf, dtype_out, omin, omax, scale_bg) for p in flat])
    return out.reshape(patterns.shape)


def dynamic_bg_ref(patterns, operation, filter_domain, std, truncate):
    """signals/ebsd.py:645-696 around the reference's per-pattern kernel."""
    from scipy.ndimage import gaussian_filter

    sy, sx = patterns.shape[-2:]
    if std is None:
        std = sx / 8
    dtype_out = patterns.dtype.type
    omin, omax = pat.dtype_range[dtype_out]
    kw = {}
    if filter_domain == "frequency":
        func = ref["fft_barnes"]._fft_filter
        (kw["fft_shape"], kw["window_shape"], kw["transfer_function"],
         kw["offset_before_fft"], kw["offset_after_ifft"]) = \
            pat._dynamic_background_frequency_space_setup((sy, sx), std, truncate)
    else:
        func = gaussian_filter
        kw = {"sigma": std, "truncate": truncate}
    flat = patterns.reshape((-1, sy, sx))
    out = np.stack([
        pat._remove_dynamic_background(p, func, operation, dtype_out, omin, omax, **kw)
        for p in flat])
    return out.reshape(patterns.shape)


def gen_preproc():
    ni, ni_bg = load_ni()
    out = {"ni": ni, "ni_bg": ni_bg, "dummy": DUMMY, "dummy_bg": DUMMY_BG}
    for name, data, bg in (("ni", ni, ni_bg), ("dummy", DUMMY, DUMMY_BG)):
        for op in ("subtract", "divide"):
            for sc in (False, True):
                out[f"{name}__static_{op}_{int(sc)}"] = static_bg_ref(data, bg, op, sc)
    # dynamic: default (frequency, std=w/8, truncate 4), then variants
    dyn_cases = {
        "freq_sub_default": ("subtract", "frequency", None, 4.0),
        "freq_div_default": ("divide", "frequency", None, 4.0),
        "freq_sub_std5": ("subtract", "frequency", 5, 4.0),
        "freq_sub_std3_t3": ("subtract", "frequency", 3, 3.0),
        "spat_sub_default": ("subtract", "spatial", None, 4.0),
        "spat_div_std5": ("divide", "spatial", 5, 4.0),
    }
    for cname, (op, dom, std, tr) in dyn_cases.items():
        out[f"ni__dyn_{cname}"] = dynamic_bg_ref(ni, op, dom, std, tr)
    # dummy-signal dynamic cases pinned by tests/test_signals/test_ebsd.py:533-985
    for cname, (op, dom, std, tr) in {
        "spat_sub_std2": ("subtract", "spatial", 2, 4.0),
        "freq_sub_std2": ("subtract", "frequency", 2, 4.0),
        "freq_div_std2": ("divide", "frequency", 2, 4.0),
        "freq_sub_std1_t3": ("subtract", "frequency", 1, 3.0),
    }.items():
        for dt in (np.uint8, np.uint16, np.float32):
            d = DUMMY.astype(dt)
            out[f"dummy__dyn_{cname}_{np.dtype(dt).name}"] = dynamic_bg_ref(d, op, dom, std, tr)
    # the canonical pipeline (pattern_matching.ipynb cells 6/27/29): static -> dynamic
    st = static_bg_ref(ni, ni_bg, "subtract", False)
    dy = dynamic_bg_ref(st, "subtract", "frequency", None, 4.0)
    out["ni__static_then_dynamic"] = dy
    # uint16 Ni (scaled) to pin the wide integer path
    ni16 = ni.astype(np.uint16) * 257
    out["ni16"] = ni16
    out["ni16__static_subtract_0"] = static_bg_ref(ni16, ni_bg.astype(np.uint16) * 257,
                                                   "subtract", False)
    out["ni16__dyn_freq_sub_default"] = dynamic_bg_ref(ni16, "subtract", "frequency", None, 4.0)
    # windows / masks
    out["circular_60"] = np.asarray(Window("circular", (60, 60)))
    out["circular_5x7"] = np.asarray(Window("circular", (5, 7)))
    w = Window("gaussian", std=7.5, shape=(30, 30))
    out["gauss_30_std7p5"] = np.asarray(w)
    (fs, ws, tf, ob, oa) = pat._dynamic_background_frequency_space_setup((60, 60), 7.5, 4.0)
    out["dynsetup_60"] = np.array([fs[0], fs[1], ws[0], ws[1], ob[0], ob[1], oa[0], oa[1]])
    # raw background estimate (float32) of Ni pattern 0, to bound FFT-vs-direct
    out["ni0__fft_bg"] = ref["fft_barnes"]._fft_filter(
        ni[0, 0].astype(np.float32), tf, fs, ws, ob, oa).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "preproc.npz"), **out)
    print("preproc", len(out))

    # ---- config 1: Ni small vs a ~1k dictionary (plumbing): pre-processed Ni
    # patterns against a synthetic dictionary built from them (perturbed +
    # random), ncc, keep_n=5.  Dictionary is regenerated from the seed.
    rng = np.random.default_rng(7)
    base = dy.reshape(9, 60, 60).astype(np.float32) / 255.0
    # element-wise float32 only (no BLAS), so the dictionary regenerates bit-for-bit
    # under any NumPy: pattern j%9 blended with noise at a weight that varies with j
    noise = rng.random((1000, 60, 60)).astype(np.float32)
    wgt = (np.float32(0.35) + np.float32(0.5) * rng.random(1000).astype(np.float32))
    dic = base[np.arange(1000) % 9] * wgt[:, None, None] + noise * (np.float32(1) - wgt)[:, None, None]
    dic[0:999:111] = base  # exact copies: 9 dictionary entries match perfectly
    dic = dic.astype(np.float32)
    o1 = {"exp": dy, "dic_sha": np.array(sha(dic)), "seed": 7}
    circ = np.asarray(~Window("circular", (60, 60)).astype(bool))
    for name, kw in {
        "ncc_k5": dict(metric="ncc", keep_n=5),
        "ncc_k5_circ_it300": dict(metric="ncc", keep_n=5, signal_mask=circ, n_per_iteration=300),
    }.items():
        s, i, msg, rep = run_di(dy, dic, **kw)
        o1[f"{name}__scores"], o1[f"{name}__indices"] = s, i
        o1[f"{name}__msg"] = np.array(msg)
    np.savez_compressed(os.path.join(OUT, "config1_ni.npz"), **o1)
    print("config1_ni", len(o1))


# ------------------------------------------------------------------ master-pattern projection (8(f1))
def load_ni_master_pattern():
    """The Lambert master pattern the reference ships for its own tests
    (data/emsoft_ebsd_master_pattern/, uint8, 401 x 401 per hemisphere)."""
    try:
        import h5py
    except ImportError:  # system interpreter: the same arrays, as stored by gen_projection()
        g = np.load(os.path.join(OUT, "projection.npz"))
        return g["mp_upper"], g["mp_lower"]

    p = os.path.join(ref_shim.SRC, "data", "emsoft_ebsd_master_pattern",
                     "ni_mc_mp_20kv_uint8_gzip_opts9.h5")
    with h5py.File(p, "r") as f:
        up = np.asarray(f["EMData/EBSDmaster/mLPNH"][()]).reshape(401, 401)
        lo = np.asarray(f["EMData/EBSDmaster/mLPSH"][()]).reshape(401, 401)
    return up, lo


def random_quaternions(rng, n):
    q = rng.standard_normal((n, 4))
    q /= np.sqrt(np.sum(q**2, axis=1))[:, None]
    q[q[:, 0] < 0] *= -1
    return q


DETECTORS = {
    # name: shape, Bruker PC, sample_tilt, tilt, azimuthal, twist (degrees)
    "det60": ((60, 60), (0.4210, 0.7794, 0.5049), 70.0, 0.0, 0.0, 0.0),
    "det48x60": ((48, 60), (0.52, 0.71, 0.63), 69.5, 5.0, 3.0, 1.5),
}


def gen_projection():
    r = ref_shim.load_reference_projection()
    mpm = r["master_pattern"]
    s2d = ref_shim.load_function_source("detectors/_ebsd_detector.py", "_sample_to_detector_matrix")
    up, lo = load_ni_master_pattern()
    out = {"mp_upper": up, "mp_lower": lo}
    rng = np.random.default_rng(11)

    # reference tests' known answers for the Lambert projection
    # (tests/test_signals/test_ebsd_master_pattern.py:746-775) + awkward vectors
    vec = np.array([[0, 0, 1], [0, 1, 0], [2, 0, 0], [0, 0, -3], [0, 0, -1], [0, -1, 0], [-2, 0, 0],
                    [0, 0, 3], [1, 1, 0], [1, -1, 0.5], [-1, 1, -0.5], [-3, -3, 1e-9], [1e-12, 2e-12, 1],
                    [0.3, -0.7, 0.2], [-0.6, 0.1, -0.9], [1, 0, 1e-17], [0, 1e-300, -1]], dtype=np.float64)
    out["vec"] = vec
    out["vec__lambert"] = mpm._vector2lambert(vec)
    for name, val in zip(("nii", "nij", "niip", "nijp", "di", "dj", "dim", "djm"),
                         mpm._get_lambert_interpolation_parameters(v=vec, npx=401, npy=401, scale=200.0)):
        out[f"vec__{name}"] = val

    dcs = {}
    for name, (shape, pc, sigma, theta, omega, gamma) in DETECTORS.items():
        m = s2d(*np.deg2rad([sigma, theta, omega, gamma]))
        out[f"{name}__s2d"] = m
        aspect = shape[1] / shape[0]
        # EBSDDetector.gnomonic_bounds (detectors/_ebsd_detector.py:731-818), Bruker PC
        bounds = np.array([-aspect * (pc[0] / pc[2]), aspect * (1 - pc[0]) / pc[2],
                           -(1 - pc[1]) / pc[2], pc[1] / pc[2]], dtype=np.float64)
        # (~Rotation.from_matrix(m)).to_matrix() == m.T for a rotation matrix
        dc = mpm._get_direction_cosines_for_fixed_pc(
            gnomonic_bounds=bounds, pcz=np.float64(pc[2]), nrows=shape[0], ncols=shape[1],
            om_detector_to_sample=np.ascontiguousarray(m.T), signal_mask=np.ones(shape[0] * shape[1], bool))
        out[f"{name}__dc"] = dc
        dcs[name] = dc
    circ_keep = np.asarray(Window("circular", (60, 60)).astype(bool))
    out["det60__dc_circ"] = mpm._get_direction_cosines_for_fixed_pc(
        gnomonic_bounds=np.array([-(0.4210 / 0.5049), (1 - 0.4210) / 0.5049, -(1 - 0.7794) / 0.5049,
                                  0.7794 / 0.5049]),
        pcz=np.float64(0.5049), nrows=60, ncols=60,
        om_detector_to_sample=np.ascontiguousarray(out["det60__s2d"].T), signal_mask=circ_keep.ravel())

    def project(rot, dc, mu, ml, rescale, omin, omax, dtype_out):
        return mpm._project_patterns_from_master_pattern_with_fixed_pc(
            rotations=rot, direction_cosines=dc, master_upper=mu, master_lower=ml, npx=401, npy=401,
            scale=200.0, rescale=rescale, out_min=omin, out_max=omax, dtype_out=dtype_out)

    rot = random_quaternions(rng, 8)
    rot[0] = [1, 0, 0, 0]
    rot[1] = np.array([1, 1, 0, 0]) / np.sqrt(2)
    out["rot8"] = rot
    upf, lof = up.astype(np.float32), lo.astype(np.float32)
    inv = (255 - up).astype(np.uint8)
    # get_patterns' rescale rule (signals/ebsd_master_pattern.py:225-233)
    out["u8mp_f32__patterns"] = project(rot, dcs["det60"], up, lo, True, -1, 1, np.float32)
    out["f32mp_f32__patterns"] = project(rot, dcs["det60"], upf, lof, False, 1, 2, np.float32)
    out["f32mp_u8__patterns"] = project(rot, dcs["det60"], upf, lof, True, 0, 255, np.uint8)
    out["u8mp_u8__patterns"] = project(rot, dcs["det60"], up, lo, False, 1, 2, np.uint8)
    out["hemis_f32__patterns"] = project(rot, dcs["det60"], upf, inv.astype(np.float32), False, 1, 2,
                                         np.float32)
    out["det48x60_f32__patterns"] = project(rot[:4], dcs["det48x60"], upf, lof, False, 1, 2, np.float32)

    # ---- one PC per pattern (_get_direction_cosines_for_varying_pc + ..._with_varying_pc)
    pcs = np.array([[0.40, 0.50, 0.40], [0.60, 0.50, 0.40], [0.40, 0.50, 0.60], [0.4210, 0.7794, 0.5049]])
    gb = np.stack([-(pcs[:, 0] / pcs[:, 2]), (1 - pcs[:, 0]) / pcs[:, 2], -(1 - pcs[:, 1]) / pcs[:, 2],
                   pcs[:, 1] / pcs[:, 2]], axis=1)
    om60 = np.ascontiguousarray(out["det60__s2d"].T)
    dcv = mpm._get_direction_cosines_for_varying_pc(
        gnomonic_bounds=gb, pcz=np.ascontiguousarray(pcs[:, 2]), nrows=60, ncols=60, om_detector_to_sample=om60,
        signal_mask=np.ones(3600, bool))
    out["varpc__pcs"] = pcs
    out["varpc__dc_sample"] = dcv[:, ::97]
    kw = dict(rotations=rot[:4], direction_cosines=dcv, master_upper=upf, master_lower=lof, npx=401, npy=401,
              scale=200.0)
    out["varpc_f32__patterns"] = mpm._project_patterns_from_master_pattern_with_varying_pc(
        rescale=False, out_min=1, out_max=2, dtype_out=np.float32, **kw)
    kw.update(master_upper=up, master_lower=lo)
    out["varpc_u8mp_f32__patterns"] = mpm._project_patterns_from_master_pattern_with_varying_pc(
        rescale=True, out_min=-1, out_max=1, dtype_out=np.float32, **kw)

    # ---- end to end: dictionary of 1200 projected patterns -> reference DI
    n_dict = 1200
    rot_d = random_quaternions(rng, n_dict)
    dic = project(rot_d, dcs["det60"], up, lo, True, -1, 1, np.float32).reshape(n_dict, 60, 60)
    picks = rng.choice(n_dict, 24, replace=False)
    noisy = dic[picks] + 0.25 * rng.standard_normal((24, 60, 60)).astype(np.float32)
    exp = np.clip((noisy + 1.5) * 80, 0, 255).astype(np.uint8)
    out["di_rot"] = rot_d
    out["di_exp"] = exp
    out["di_picks"] = picks
    out["di_dic_sample"] = dic[::100]
    out["di_dic_sha"] = np.array(sha(dic))
    for name, kw in {
        "di_ncc_k10": dict(metric="ncc", keep_n=10, n_per_iteration=500),
        "di_ndp_k10_circ": dict(metric="ndp", keep_n=10, signal_mask=~circ_keep),
    }.items():
        s, i, msg, rep = run_di(exp, dic, **kw)
        out[f"{name}__scores"], out[f"{name}__indices"] = s, i
    np.savez_compressed(os.path.join(OUT, "projection.npz"), **out)
    print("projection", len(out))


# ------------------------------------------------------------------ refinement (8(f2))
def gen_refinement():
    """Objective functions and SciPy Nelder-Mead solvers of the reference
    (indexing/_refinement/_objective_functions.py, _solvers.py) on synthetic
    experiments: Ni master pattern, 60 x 60 detector, one PC per pattern.

    Run under BOTH interpreters: /opt/conda/bin/python3.9 (SciPy 1.7.1) writes
    refinement.npz; the system python (SciPy 1.15.3, the version on the GPU box)
    writes refinement_scipy115.npz with the solver results only - Nelder-Mead's
    handling of bounds changed between these SciPy versions."""
    import scipy
    import scipy.optimize

    r = ref_shim.load_reference_refinement()
    mpm, sol, obj, nbu = r["master_pattern"], r["solvers"], r["objective_functions"], r["numba_utils"]
    s2d = ref_shim.load_function_source("detectors/_ebsd_detector.py", "_sample_to_detector_matrix")
    modern = tuple(int(v) for v in scipy.__version__.split(".")[:2]) >= (1, 11)
    up, lo = load_ni_master_pattern()
    # _get_master_pattern_data (indexing/_refinement/_refinement.py:1288-1320)
    mpu = pat.rescale_intensity(up, dtype_out=np.float32)
    mpl = pat.rescale_intensity(lo, dtype_out=np.float32)
    fixed_mp = (mpu, mpl, 401, 401, 200.0)
    nrows = ncols = 60
    om = np.ascontiguousarray(s2d(*np.deg2rad([70.0, 0.0, 0.0, 0.0])).T)
    rng = np.random.default_rng(21)
    n = 4
    eu_true = np.column_stack([rng.uniform(0.3, 6.0, n), rng.uniform(0.3, 2.8, n), rng.uniform(0.3, 6.0, n)])
    pc_true = np.array([0.42, 0.78, 0.50]) + rng.uniform(-0.01, 0.01, (n, 3))
    keep = np.asarray(Window("circular", (60, 60)).astype(bool)).ravel()
    all_px = np.ones(3600, dtype=bool)

    def dc_for(pc, mask):
        return mpm._get_direction_cosines_for_fixed_pc(
            gnomonic_bounds=r["gnomonic_bounds"].get_gnomonic_bounds(nrows, ncols, *pc), pcz=pc[2],
            nrows=nrows, ncols=ncols, om_detector_to_sample=om, signal_mask=mask)

    pats = np.empty((n, 3600), dtype=np.uint8)
    for i in range(n):
        sim = mpm._project_single_pattern_from_master_pattern(
            rotation=nbu.rotation_from_euler(*eu_true[i]), direction_cosines=dc_for(pc_true[i], all_px),
            master_upper=mpu, master_lower=mpl, npx=401, npy=401, scale=200.0, rescale=False, out_min=0,
            out_max=1, dtype_out=np.float32)
        noisy = sim + 0.15 * rng.standard_normal(3600).astype(np.float32)
        pats[i] = np.clip((noisy + 1.3) * 98, 0, 255).astype(np.uint8)
    eu0 = eu_true + np.deg2rad(rng.uniform(-1.0, 1.0, (n, 3)))
    pc0 = pc_true + rng.uniform(-0.004, 0.004, (n, 3))
    out = {"patterns": pats, "eu_true": eu_true, "pc_true": pc_true, "eu0": eu0, "pc0": pc0,
           "om_detector_to_sample": om, "scipy_version": np.array(scipy.__version__)}

    nm = {"method": "Nelder-Mead"}
    if not modern:
        # ---- _prepare_pattern, uint8 (no rescale) and float32 (rescale), masked / unmasked
        p8, sq8 = sol._prepare_pattern(pats[0], False)
        pf, sqf = sol._prepare_pattern(pats[1][keep].astype(np.float32) * 0.37, True)
        out.update(prep_u8=p8, prep_u8_sqnorm=np.float64(sq8), prep_f32_masked=pf, prep_f32_masked_sqnorm=np.float64(sqf))
        out["mp_f32_upper_sample"] = mpu[::40, ::40]
        # ---- objective values at the starts and at a few offsets
        offs = np.array([[0, 0, 0, 0, 0, 0], [0.01, -0.02, 0.005, 0, 0, 0], [0, 0, 0, 0.003, -0.002, 0.004],
                         [-0.03, 0.01, 0.02, -0.005, 0.005, 0.002]])
        vals = np.empty((n, len(offs), 3))
        for i in range(n):
            for mask_name, mask in (("all", all_px),):
                p, sq = sol._prepare_pattern(pats[i][mask], False)
                for j, o in enumerate(offs):
                    x = np.concatenate([eu0[i], pc0[i]]) + o
                    vals[i, j, 0] = obj._refine_orientation_objective_function(
                        x[:3], p, dc_for(pc0[i], mask), *fixed_mp, sq)
                    vals[i, j, 1] = obj._refine_pc_objective_function(
                        x[3:], p, nbu.rotation_from_euler(*eu0[i]), *fixed_mp, mask, nrows, ncols, om, sq)
                    vals[i, j, 2] = obj._refine_orientation_pc_objective_function(
                        x, p, *fixed_mp, mask, nrows, ncols, om, sq)
        out["objective_offsets"] = offs
        out["objective_values"] = vals
        p, sq = sol._prepare_pattern(pats[2][keep], False)
        out["objective_masked"] = np.float64(obj._refine_orientation_pc_objective_function(
            np.concatenate([eu0[2], pc0[2]]), p, *fixed_mp, keep, nrows, ncols, om, sq))

    # ---- solvers (scipy.optimize.minimize, Nelder-Mead)
    def ori(i, bounds=None, starts=None, mask=all_px, mk=None):
        rot = eu0[i][None] if starts is None else starts
        kw = dict(mk or nm)
        b = np.zeros((len(rot), 3, 2)) if bounds is None else bounds
        return np.array(sol._refine_orientation_solver_scipy(
            pattern=pats[i][mask], rotation=rot, bounds=b, signal_mask=mask, rescale=False,
            method=scipy.optimize.minimize, method_kwargs=kw, trust_region_passed=bounds is not None,
            fixed_parameters=fixed_mp, pcx=pc0[i][0], pcy=pc0[i][1], pcz=pc0[i][2], nrows=nrows, ncols=ncols,
            om_detector_to_sample=om, n_pseudo_symmetry_ops=len(rot) - 1), dtype=np.float64)

    tr = np.deg2rad(2.0)
    out["ori_nm"] = np.stack([ori(i) for i in range(n)])
    out["ori_nm_masked"] = np.stack([ori(i, mask=keep) for i in range(2)])
    out["ori_nm_maxfev30"] = np.stack([ori(i, mk=dict(method="Nelder-Mead", options=dict(maxfev=30))) for i in range(2)])
    bnd = np.stack([np.stack([eu0[i] - tr, eu0[i] + tr], axis=1)[None] for i in range(n)])
    out["ori_nm_bounds"] = np.stack([ori(i, bounds=bnd[i]) for i in range(n)])
    # pseudo-symmetry: start 1 = a copy displaced by 3 degrees about phi1 (only the bookkeeping matters)
    starts = np.stack([np.stack([eu0[i] + [np.deg2rad(3), 0, 0], eu0[i]]) for i in range(2)])
    out["ori_nm_ps_starts"] = starts
    out["ori_nm_ps"] = np.stack([ori(i, starts=starts[i]) for i in range(2)])

    fixed_pc = fixed_mp + (all_px, nrows, ncols, om)
    res = []
    for i in range(2):
        res.append(sol._refine_pc_solver_scipy(
            pattern=pats[i], rotation=nbu.rotation_from_euler(*eu0[i]), pc=pc0[i], bounds=np.zeros((3, 2)),
            rescale=False, method=scipy.optimize.minimize, method_kwargs=dict(nm), fixed_parameters=fixed_pc,
            trust_region_passed=False))
    out["pc_nm"] = np.array(res, dtype=np.float64)
    res = []
    for i in range(2):
        x0 = np.concatenate([eu0[i], pc0[i]])[None]
        tr6 = np.array([tr, tr, tr, 0.02, 0.02, 0.02])
        b = np.stack([x0[0] - tr6, x0[0] + tr6], axis=1)[None]
        res.append(sol._refine_orientation_pc_solver_scipy(
            pattern=pats[i], rot_pc=x0, bounds=b, rescale=False, method=scipy.optimize.minimize,
            method_kwargs=dict(nm), fixed_parameters=fixed_pc, trust_region_passed=True))
    out["ori_pc_nm_bounds"] = np.array(res, dtype=np.float64)
    name = "refinement_scipy115.npz" if modern else "refinement.npz"
    if modern:
        out = {k: v for k, v in out.items() if k.startswith(("ori_", "pc_nm", "scipy_version"))}
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, sorted(out))


# ------------------------------------------------------------------ result consumers (8(f3))
class FakeXmap:
    """What `orientation_similarity_map` reads from a CrystalMap
    (indexing/_orientation_similarity_map.py:118-121): `.prop[...]`, `.shape`."""

    def __init__(self, prop, shape):
        self.prop, self.shape = prop, shape


def correlated_indices(rng, ny, nx, keep_n, n_dict=500):
    """Top-k index lists that overlap between neighbours like a real map: every
    point draws from a pool centred on a slowly varying grain label."""
    grain = (np.add.outer(np.arange(ny) // 4, np.arange(nx) // 5) * 37) % n_dict
    out = np.empty((ny * nx, keep_n), dtype=np.int64)
    for p, g0 in enumerate(grain.ravel()):
        pool = (g0 + np.arange(3 * keep_n)) % n_dict
        out[p] = rng.choice(pool, keep_n, replace=False)
    return out


def gen_consumers():
    import importlib

    osm_mod = ref_shim._load("kikuchipy.indexing._orientation_similarity_map",
                             "indexing/_orientation_similarity_map.py")
    osm = osm_mod.orientation_similarity_map
    rng = np.random.default_rng(31)
    out = {}
    idx = correlated_indices(rng, 9, 13, 20)
    out["idx_9x13_k20"] = idx
    x = FakeXmap({"simulation_indices": idx}, (9, 13))
    out["osm_default"] = osm(x)
    out["osm_normalized"] = osm(x, normalize=True)
    out["osm_nbest7"] = osm(x, n_best=7)
    out["osm_from5_to8"] = osm(x, n_best=8, from_n_best=5, normalize=True)
    square = np.ones((3, 3), dtype=int)
    out["osm_square_fp"] = osm(x, footprint=square, center_index=4)
    wide = np.array([[1, 1, 1, 1, 1]])
    out["osm_row_fp"] = osm(x, n_best=10, footprint=wide, center_index=2)
    # duplicate indices inside a list (zero-filled rows of masked points): unique-set semantics
    dup = idx.copy()
    dup[::7] = 0
    dup[3::11, 5:] = dup[3::11, :1]
    out["idx_dup"] = dup
    out["osm_dup"] = osm(FakeXmap({"simulation_indices": dup}, (9, 13)), n_best=12)
    # the reference tests' cases (tests/test_indexing/test_orientation_similarity_map.py:27-64)
    x = FakeXmap({"simulation_indices": np.tile(np.arange(5), (100, 1))}, (10, 10))
    out["osm_reftest_tile"] = osm(x)
    out["osm_reftest_tile_norm"] = osm(x, normalize=True)
    out["osm_reftest_from2_shape"] = np.array(
        osm(FakeXmap({"simulated_indices": np.ones((100, 5))}, (10, 10)), simulation_indices_prop="simulated_indices",
            from_n_best=2).shape)
    # ---- merge_crystal_maps with stand-in CrystalMaps (ref_shim.FakeCrystalMap)
    _ns_ok = ref_shim._ns("kikuchipy.signals", os.path.join(ref_shim.SRC, "signals")) if "kikuchipy.signals" not in sys.modules else None
    if "kikuchipy.signals.util" not in sys.modules:
        ref_shim._ns("kikuchipy.signals.util", os.path.join(ref_shim.SRC, "signals", "util"))
    ref_shim._load("kikuchipy.signals.util._crystal_map", "signals/util/_crystal_map.py")
    merge = ref_shim._load("kikuchipy.indexing._merge_crystal_maps", "indexing/_merge_crystal_maps.py").merge_crystal_maps

    def make_map(seed, shape, n, name, mask=None, lower_better=False, not_indexed=None):
        r = np.random.default_rng(seed)
        size = int(np.prod(shape))
        m = size if mask is None else int((~mask).sum())
        sc = np.sort(r.random((m, n)).astype(np.float32), axis=1)
        sc = sc if lower_better else sc[:, ::-1].copy()
        si = r.integers(0, 1000, (m, n)).astype(np.int64)
        q = r.standard_normal((m, n, 4))
        q /= np.linalg.norm(q, axis=-1, keepdims=True)
        pid = np.zeros(m, dtype=int)
        if not_indexed is not None:
            pid[not_indexed] = -1
        return dict(scores=sc, simulation_indices=si, rotations=q, phase_id=pid, mask=mask, name=name, shape=shape)

    def to_fake(d):
        return ref_shim.FakeCrystalMap(d["shape"], d["rotations"], {"scores": d["scores"],
                                       "simulation_indices": d["simulation_indices"]}, d["name"],
                                       None if d["mask"] is None else ~d["mask"].ravel(), d["phase_id"])

    def record(tag, maps, **kw):
        res = merge([to_fake(d) for d in maps], **kw)
        k = res.kw
        out[f"{tag}__phase_id"] = np.asarray(k["phase_id"])
        out[f"{tag}__rotations"] = np.asarray(k["rotations"].args[0])
        out[f"{tag}__phase_names"] = np.array(k["phase_list"].names)
        for name, val in k["prop"].items():
            out[f"{tag}__{name}"] = np.asarray(val)
        for j, d in enumerate(maps):
            for key in ("scores", "simulation_indices", "rotations", "phase_id"):
                out[f"{tag}__in{j}_{key}"] = d[key]
            if d["mask"] is not None:
                out[f"{tag}__in{j}_mask"] = d["mask"]

    shape = (4, 3)
    two = [make_map(1, shape, 5, "a"), make_map(2, shape, 5, "b")]
    record("merge2", two, simulation_indices_prop="simulation_indices")
    record("merge2_mean3", two, mean_n_best=3, simulation_indices_prop="simulation_indices")
    low = [make_map(3, shape, 5, "a", lower_better=True), make_map(4, shape, 5, "b", lower_better=True)]
    record("merge2_lower", low, greater_is_better=False, simulation_indices_prop="simulation_indices")
    record("merge2_negmean", low, mean_n_best=-2, simulation_indices_prop="simulation_indices")
    m0 = np.zeros(shape, dtype=bool); m0[0, 0] = m0[2, 1] = True
    m1 = np.zeros(shape, dtype=bool); m1[0, 0] = m1[3, 2] = m1[1, 1] = True
    three = [make_map(5, shape, 4, "a", mask=m0, not_indexed=[1]), make_map(6, shape, 4, "b", mask=m1),
             make_map(7, shape, 4, "c", not_indexed=[0, 1])]
    three[2]["scores"][0] = 0  # all maps: point (0, 0) masked out or not indexed
    record("merge3_masks", three, simulation_indices_prop="simulation_indices",
           navigation_masks=[m0, m1, None])
    same = [make_map(8, shape, 3, "a"), make_map(9, shape, 3, "a"), make_map(10, shape, 3, "b")]
    record("merge3_same_name", same, simulation_indices_prop="simulation_indices")
    record("merge2_no_indices", two)
    ni = [make_map(11, shape, 4, "a", not_indexed=[3, 7]), make_map(12, shape, 4, "b", not_indexed=[3, 5])]
    record("merge2_not_indexed", ni, simulation_indices_prop="simulation_indices")
    np.savez_compressed(os.path.join(OUT, "consumers.npz"), **out)
    print("consumers", sorted(out))


# ------------------------------------------------------------------ h5ebsd fixture (8(f4))
def gen_h5ebsd():
    """tests/golden/h5ebsd_ni.h5: a kikuchipy-h5ebsd file (layout of
    io/plugins/kikuchipy_h5ebsd/_api.py) written with h5py from the Ni patterns
    the reference ships: Scan 1 contiguous uint8; Scan 2 chunked + gzip +
    shuffle, one PC; Scan 3 float32, only 7 of the 9 patterns stored (the
    reader zero pads, io/plugins/_h5ebsd.py:367-378), no static background.
    Expected arrays go to h5ebsd_expected.npz."""
    import h5py

    pats, bg = load_ni()
    src = os.path.join(ref_shim.SRC, "data", "kikuchipy_h5ebsd", "patterns.h5")
    with h5py.File(src, "r") as f:
        pcs = [f[f"Scan 1/EBSD/Header/pc{a}"][()] for a in "xyz"]
    path = os.path.join(OUT, "h5ebsd_ni.h5")
    flat = pats.reshape(9, 60, 60)
    with h5py.File(path, "w") as f:
        f.create_dataset("manufacturer", data=np.array([b"kikuchipy"]))
        f.create_dataset("version", data=np.array([b"0.8.dev0"]))

        def header(scan, **extra):
            h = f.create_group(f"{scan}/EBSD/Header")
            vals = dict(n_rows=3, n_columns=3, pattern_height=60, pattern_width=60, binning=8,
                        detector_pixel_size=70.0, sample_tilt=70, azimuth_angle=0, elevation_angle=1.5,
                        step_x=1.5, step_y=1.5)
            vals.update(extra)
            for k, v in vals.items():
                h.create_dataset(k, data=np.atleast_1d(v))
            f.create_group(f"{scan}/SEM/Header").create_dataset("beam_energy", data=np.array([20.0]))
            return h

        h = header("Scan 1")
        h.create_dataset("static_background", data=bg)
        for a, v in zip("xyz", pcs):
            h.create_dataset(f"pc{a}", data=v)
        f.create_dataset("Scan 1/EBSD/Data/patterns", data=flat)
        h = header("Scan 2", step_x=0.5, step_y=0.25)
        h.create_dataset("static_background", data=bg)
        for a, v in zip("xyz", (0.42, 0.21, 0.5)):
            h.create_dataset(f"pc{a}", data=np.array([v]))
        f.create_dataset("Scan 2/EBSD/Data/patterns", data=flat[::-1], chunks=(2, 60, 60), compression="gzip",
                         compression_opts=4, shuffle=True)
        h = header("Scan 3", n_rows=1, n_columns=9)
        f.create_dataset("Scan 3/EBSD/Data/patterns", data=flat[:7].astype(np.float32) / 3)
    padded = np.zeros((9, 60, 60), dtype=np.float32)
    padded[:7] = flat[:7].astype(np.float32) / 3
    np.savez_compressed(os.path.join(OUT, "h5ebsd_expected.npz"), scan1=pats, scan2=flat[::-1].reshape(3, 3, 60, 60),
                        scan3=padded, static_background=bg, pc1=np.stack([p.ravel() for p in pcs], axis=1))
    print("h5ebsd", os.path.getsize(path))


# ------------------------------------------------------------------ the reference tests' own known answers
def gen_refknown():
    """Extract the hard-coded known-answer ARRAYS (data, not code) that the
    reference's test-suite holds for the pre-processing path:
    tests/test_signals/test_ebsd.py:245-443 (static), :534-916 (dynamic
    spatial), :924-985 (dynamic frequency), :476-487 (scale_bg)."""
    import ast

    path = os.path.join(ref_shim.REF_ROOT, "tests", "test_signals", "test_ebsd.py")
    tree = ast.parse(open(path).read())
    wanted = {
        "test_remove_static_background": "static",
        "test_remove_dynamic_background_spatial": "dyn_spatial",
        "test_remove_dynamic_background_frequency": "dyn_frequency",
    }
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in wanted:
            for dec in node.decorator_list:
                if not (isinstance(dec, ast.Call) and getattr(dec.func, "attr", "") == "parametrize"):
                    continue
                names = ast.literal_eval(dec.args[0]).replace(" ", "").split(",")
                cases = eval(compile(ast.Expression(dec.args[1]), path, "eval"), {"np": np})
                for ci, case in enumerate(cases):
                    tag = f"{wanted[node.name]}__{ci}"
                    for nm, val in zip(names, case):
                        if nm == "answer":
                            out[f"{tag}__answer"] = np.asarray(val)
                        else:
                            out[f"{tag}__{nm}"] = np.array(val)
    out["static_scalebg__answer"] = np.array([[15, 150, 15], [180, 255, 120], [150, 0, 75]])
    np.savez_compressed(os.path.join(OUT, "refknown.npz"), **out)
    print("refknown", sorted(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:  # e.g. `gen_golden.py projection`: only the named fixtures
        for name in sys.argv[1:]:
            globals()[f"gen_{name}"]()
        sys.exit(0)
    gen_dummy_di()
    gen_synth_di()
    gen_preproc()
    gen_refknown()
    gen_projection()
    gen_refinement()
    gen_consumers()
    gen_h5ebsd()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
