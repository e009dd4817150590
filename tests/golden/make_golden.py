#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (build container only).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        PYTHONPATH=/root/reference python3 /root/repo/tests/golden/make_golden.py

The reference cannot travel to the GPU box, so its outputs are committed here as
data (inputs + expected outputs only).  Every case is seeded; re-running this
script reproduces the files bit-for-bit on the same numpy/scipy/OpenBLAS.

Files written next to this script:
  cov_cases.npz    covariance.compute(): K, dK, K(X,X*), diag for every kernel
  core_cases.npz   GP.__compute_nlZ (nlZ, dnlZ), Posterior fields, GP.predict
  prior_cases.npz  GP.__compute_log_priors, normalization constants, f_min_fill designs
  full_cases.npz   GP.predict_full (with and without noise) and GP.quad
  fit_cases.npz    GP.fit end to end under a fixed global seed (hyp samples, predictions);
                   f2 = the reference's own examples/example_1.py model and data
  fullsize_cases.npz   nlZ and dnlZ of BASELINE cfg2 (sample 0) and cfg3 (samples 0, 1, 15)
                   at full size (N = 2048 / 4096), a few dozen doubles
  fullsize45_cases.npz the same at cfg4 (N = 16384, RQ: nlZ only -- the reference's (N, N, 22) gradient
                   tensor would need 47 GB) and cfg5 (N = 8192, S = 64: samples 0 and 63 with gradient,
                   7 and 8 nlZ only); run with the target `fullsize45`.  Target `cfg4grad` adds `cfg4_dnlZ` to that
                   file: ORACLE-DERIVED (oracle.core_streamed, one gradient plane at a time, ~20 GB, ~10 min) because
                   the reference needs 47 GB here; the script asserts that the streamed oracle's nlZ is the
                   reference's stored cfg4 nlZ bit for bit before it writes the gradient
  big20k_case.npz  N = 20480 (beyond the device library's former 16384 ceiling), D = 5, SE: the reference's nlZ and the
                   streamed oracle's gradient (oracle-derived, labelled; written only if its nlZ is the reference's bit for
                   bit); target `big20k`
  rank1_cases.npz  GP.update with ONE new point (the reference's rank-one path,
                   gaussian_process.py:750-844), high- and low-noise parametrisation
  api_sweep_reference.txt  (not written by this script) the output of tools/api_sweep.py run against the reference:
                   cd /tmp && GPYREG_MODULE=gpyreg PYTHONPATH=/root/reference:/root/repo python3 -W ignore \
                       /root/repo/tools/api_sweep.py > /root/repo/tests/golden/api_sweep_reference.txt
  plugin_sweep_reference.txt  likewise, tools/plugin_sweep.py (host-side plugin protocol; CPU test)
  numerics_sweep_reference.txt  likewise, tools/numerics_sweep.py (189 models through nlZ, predict, lpd, quad, ...);
                   numerics_sweep_n300_reference.txt: the same with SWEEP_N=300
  update_sweep_reference.txt  likewise, tools/update_sweep.py (GP.update call sequences, a prediction after each)
  sampler_sweep_reference.txt  likewise, tools/sampler_sweep.py (SliceSampler alone, host only; CPU test)
  fit_sweep_reference.txt  likewise, tools/fit_sweep.py (seeded fits over the option edge cases) against the reference
  draw_cases.npz   GP.random_function under a fixed global seed (posterior and prior draws, with and without
                   noise) and the reference's factor of rank-deficient / indefinite covariance matrices
                   (``__robust_cholesky``); target `draw`
"""

import os
import sys

import numpy as np

import gpyreg as gpr  # the reference (PYTHONPATH=/root/reference)

assert "/root/reference" in os.path.abspath(gpr.__file__), gpr.__file__

HERE = os.path.dirname(os.path.abspath(__file__))

KERNELS = {
    "se": lambda: gpr.covariance_functions.SquaredExponential(),
    "matern1": lambda: gpr.covariance_functions.Matern(1),
    "matern3": lambda: gpr.covariance_functions.Matern(3),
    "matern5": lambda: gpr.covariance_functions.Matern(5),
    "rq": lambda: gpr.covariance_functions.RationalQuadraticARD(),
    "se_iso": lambda: gpr.isotropic_covariance_functions.SquaredExponentialIsotropic(),
    "matern_iso1": lambda: gpr.isotropic_covariance_functions.MaternIsotropic(1),
    "matern_iso3": lambda: gpr.isotropic_covariance_functions.MaternIsotropic(3),
    "matern_iso5": lambda: gpr.isotropic_covariance_functions.MaternIsotropic(5),
}
MEANS = {
    "zero": lambda: gpr.mean_functions.ZeroMean(),
    "const": lambda: gpr.mean_functions.ConstantMean(),
    "negquad": lambda: gpr.mean_functions.NegativeQuadratic(),
}


def make_noise(p):
    return gpr.noise_functions.GaussianNoise(
        constant_add=p[0] == 1,
        user_provided_add=p[1] >= 1,
        scale_user_provided=p[1] == 2,
        rectified_linear_output_dependent_add=p[2] == 1,
    )


def cov_cases():
    out = {}
    names = []
    idx = 0
    for kname, mk in KERNELS.items():
        for (N, D, M) in [(7, 1, 3), (33, 3, 5), (20, 2, 130)]:
            rng = np.random.default_rng(7000 + idx)
            cov = mk()
            cov_N = cov.hyperparameter_count(D)
            X = rng.uniform(-2, 2, (N, D))
            Xs = rng.uniform(-2.5, 2.5, (M, D))
            hyp = 0.5 * rng.standard_normal(cov_N)
            K, dK = cov.compute(hyp, X, compute_grad=True)
            Ks = cov.compute(hyp, X, Xs)
            kd = cov.compute(hyp, Xs, compute_diag=True)
            tag = f"c{idx:03d}"
            names.append(f"{tag}|{kname}|{N}|{D}|{M}")
            out[tag + "_X"] = X
            out[tag + "_Xs"] = Xs
            out[tag + "_hyp"] = hyp
            out[tag + "_K"] = K
            out[tag + "_dK"] = np.ascontiguousarray(dK)
            out[tag + "_Ks"] = Ks
            out[tag + "_kd"] = kd
            idx += 1
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "cov_cases.npz"), **out)
    print("cov cases:", len(names))


# (kernel, mean, noise params, N, D, flavour)
CORE_CASES = [
    ("se", "const", (1, 0, 0), 7, 1, "plain"),
    ("se", "const", (1, 0, 0), 33, 2, "plain"),
    ("se", "const", (1, 0, 0), 130, 3, "plain"),
    ("se", "const", (1, 0, 0), 257, 5, "plain"),
    ("se", "zero", (1, 0, 0), 33, 2, "plain"),
    ("se", "negquad", (1, 0, 0), 33, 2, "plain"),
    ("se", "negquad", (1, 1, 0), 130, 3, "plain"),
    ("se", "const", (0, 0, 0), 33, 2, "lownoise"),
    ("se", "const", (0, 0, 0), 130, 2, "lownoise"),
    ("se", "zero", (0, 1, 0), 33, 2, "plain"),
    ("se", "const", (1, 1, 0), 33, 2, "plain"),
    ("se", "const", (1, 2, 0), 33, 2, "plain"),
    ("se", "const", (1, 0, 1), 33, 2, "plain"),
    ("se", "negquad", (1, 2, 1), 130, 3, "plain"),
    ("se", "const", (0, 1, 0), 33, 2, "tiny_s2"),
    ("matern1", "const", (1, 0, 0), 33, 2, "plain"),
    ("matern3", "const", (1, 0, 0), 33, 2, "plain"),
    ("matern3", "negquad", (1, 1, 0), 130, 4, "plain"),
    ("matern5", "const", (1, 0, 0), 33, 2, "plain"),
    ("matern5", "const", (1, 0, 0), 257, 10, "plain"),
    ("matern5", "zero", (0, 0, 0), 33, 3, "lownoise"),
    ("rq", "const", (1, 0, 0), 33, 2, "plain"),
    ("rq", "negquad", (1, 2, 0), 130, 3, "plain"),
    ("rq", "const", (1, 0, 0), 257, 6, "plain"),
    ("se_iso", "const", (1, 0, 0), 33, 3, "plain"),
    ("se_iso", "const", (1, 1, 0), 130, 3, "plain"),
    ("matern_iso1", "const", (1, 0, 0), 33, 2, "plain"),
    ("matern_iso3", "const", (1, 0, 0), 33, 3, "plain"),
    ("matern_iso5", "negquad", (1, 0, 0), 130, 3, "plain"),
    ("se", "const", (1, 0, 0), 33, 2, "jitter_high"),
    ("se", "const", (0, 0, 0), 33, 2, "jitter_low"),
    ("matern5", "const", (1, 0, 0), 130, 2, "jitter_high"),
    ("se", "const", (1, 0, 0), 128, 2, "plain"),
    ("se", "const", (1, 0, 0), 129, 2, "plain"),
    ("matern5", "const", (1, 0, 0), 384, 4, "plain"),
]


def core_cases():
    out = {}
    names = []
    for idx, (kname, mname, npar, N, D, flavour) in enumerate(CORE_CASES):
        rng = np.random.default_rng(9000 + idx)
        cov, mean, noise = KERNELS[kname](), MEANS[mname](), make_noise(npar)
        gp = gpr.GP(D=D, covariance=cov, mean=mean, noise=noise)
        cov_N = cov.hyperparameter_count(D)
        mean_N = mean.hyperparameter_count(D)
        noise_N = noise.hyperparameter_count()
        S = 2
        X = rng.uniform(-3, 3, (N, D))
        y = np.sin(np.sum(X, 1, keepdims=True) / np.sqrt(D)) + 0.1 * rng.standard_normal((N, 1))
        s2 = None
        if npar[1] >= 1:
            s2 = 0.01 + 0.05 * rng.uniform(size=(N, 1))
            if flavour == "tiny_s2":
                s2 = 1e-8 * (1 + rng.uniform(size=(N, 1)))  # vector noise, L_chol False
        hyp = np.zeros((S, cov_N + noise_N + mean_N))
        for s in range(S):
            h_cov = 0.3 * rng.standard_normal(cov_N)
            if not kname.endswith(("iso", "iso1", "iso3", "iso5")):
                h_cov[:D] += np.log(1.2 * np.sqrt(D))
            else:
                h_cov[0] += np.log(1.2 * np.sqrt(D))
            h_noise = []
            if npar[0] == 1:
                h_noise.append(np.log(0.1) + 0.2 * rng.standard_normal())
            if npar[1] == 2:
                h_noise.append(0.3 * rng.standard_normal())
            if npar[2] == 1:
                h_noise += [0.2 * rng.standard_normal(), np.log(0.05) + 0.1 * rng.standard_normal()]
            if mname == "zero":
                h_mean = []
            elif mname == "const":
                h_mean = [0.2 * rng.standard_normal()]
            else:
                h_mean = (
                    [0.2 * rng.standard_normal()]
                    + list(0.5 * rng.standard_normal(D))
                    + list(np.log(4.0) + 0.2 * rng.standard_normal(D))
                )
            hyp[s] = np.concatenate([h_cov, h_noise, h_mean])
        if flavour.startswith("jitter"):
            # duplicate points + huge output scale -> Cholesky needs the x10 escalation
            X[N // 2 :] = X[: N - N // 2]
            if flavour == "jitter_high":
                hyp[:, D] = 12.0  # log sigma_f
                hyp[:, cov_N] = np.log(1.1e-3)  # sn2 ~ 1.2e-6 >= 1e-6 -> L_chol True
            else:
                hyp[:, D] = 3.0
        gp.update(X_new=X, y_new=y, s2_new=s2, hyp=hyp)
        tag = f"g{idx:03d}"
        names.append(f"{tag}|{kname}|{mname}|{npar[0]}{npar[1]}{npar[2]}|{N}|{D}|{flavour}")
        out[tag + "_X"] = X
        out[tag + "_y"] = y
        if s2 is not None:
            out[tag + "_s2"] = s2
        out[tag + "_hyp"] = hyp
        nlZ = np.zeros(S)
        dnlZ = np.zeros((S, hyp.shape[1]))
        nlZ_only = np.zeros(S)
        for s in range(S):
            nlZ[s], dnlZ[s] = gp._GP__compute_nlZ(hyp[s], True, False)
            nlZ_only[s] = gp._GP__compute_nlZ(hyp[s], False, False)
        out[tag + "_nlZ"] = nlZ
        out[tag + "_nlZ_only"] = nlZ_only
        out[tag + "_dnlZ"] = dnlZ
        out[tag + "_alpha"] = np.stack([p.alpha[:, 0] for p in gp.posteriors])
        out[tag + "_sW"] = np.stack([p.sW[:, 0] for p in gp.posteriors])
        out[tag + "_sn2_mult"] = np.array([float(p.sn2_mult) for p in gp.posteriors])
        out[tag + "_L_chol"] = np.array([bool(p.L_chol) for p in gp.posteriors])
        if N <= 40:
            out[tag + "_L"] = np.stack([np.asarray(p.L) for p in gp.posteriors])
        else:  # keep the fixture small: diagonal, first/last rows, Frobenius norm
            out[tag + "_Ldiag"] = np.stack([np.diag(p.L) for p in gp.posteriors])
            out[tag + "_Lrow0"] = np.stack([np.asarray(p.L)[0] for p in gp.posteriors])
            out[tag + "_Lcol_last"] = np.stack([np.asarray(p.L)[:, -1] for p in gp.posteriors])
            out[tag + "_Lfro"] = np.array([np.linalg.norm(p.L) for p in gp.posteriors])
        # predictions
        M = 11
        xs = rng.uniform(-3.5, 3.5, (M, D))
        ys = np.sin(np.sum(xs, 1, keepdims=True) / np.sqrt(D))
        s2s = 0.02 * np.ones((M, 1)) if s2 is not None else None
        out[tag + "_xs"] = xs
        out[tag + "_ys"] = ys
        mu_sep, s2_sep = gp.predict(xs, ys, s2s, add_noise=False, separate_samples=True)
        out[tag + "_mu_sep"], out[tag + "_s2_sep"] = mu_sep, s2_sep
        mu_a, s2_a, lpd_a = gp.predict(xs, ys, s2s, add_noise=True, return_lpd=True)
        out[tag + "_mu_avg"], out[tag + "_s2n_avg"], out[tag + "_lpd_avg"] = mu_a, s2_a, lpd_a
        mu_b, s2_b, lpd_b = gp.predict(
            xs, ys, s2s, add_noise=False, separate_samples=True, return_lpd=True
        )
        out[tag + "_lpd_sep"] = lpd_b
        print(
            names[-1],
            "nlZ", nlZ,
            "mult", out[tag + "_sn2_mult"],
            "Lchol", out[tag + "_L_chol"],
            "nan-grad", int(np.isnan(dnlZ).sum()),
        )
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "core_cases.npz"), **out)
    print("core cases:", len(names))


def prior_cases():
    """GP.__compute_log_priors / normalization constants / f_min_fill design from the reference."""
    from gpyreg.f_min_fill import f_min_fill

    out = {}
    names = []
    kinds = ["gaussian", "student_t", "smoothbox", "smoothbox_student_t", None]
    for idx in range(12):
        rng = np.random.default_rng(12000 + idx)
        D = 1 + idx % 3
        cov = [KERNELS["se"], KERNELS["rq"], KERNELS["matern_iso5"]][idx % 3]()
        mean = [MEANS["const"], MEANS["negquad"], MEANS["zero"]][(idx // 3) % 3]()
        noise = make_noise([(1, 0, 0), (1, 2, 0), (1, 0, 1)][idx % 3])
        gp = gpr.GP(D=D, covariance=cov, mean=mean, noise=noise)
        info = (cov.hyperparameter_info(D) + noise.hyperparameter_info() + mean.hyperparameter_info(D))
        hyp_N = sum(c for _, c in info)
        pri, bnd = {}, {}
        for k, (name, cnt) in enumerate(info):
            kind = kinds[(idx + k) % 5]
            if kind == "gaussian":
                pri[name] = (kind, (rng.standard_normal(), np.exp(0.3 * rng.standard_normal())))
            elif kind == "student_t":
                pri[name] = (kind, (rng.standard_normal(), np.exp(0.3 * rng.standard_normal()), 3 + 4 * rng.uniform()))
            elif kind == "smoothbox":
                pri[name] = (kind, (-1.0 - rng.uniform(), 1.0 + rng.uniform(), 0.5 + rng.uniform()))
            elif kind == "smoothbox_student_t":
                pri[name] = (kind, (-1.0 - rng.uniform(), 1.0 + rng.uniform(), 0.5 + rng.uniform(), 3 + 4 * rng.uniform()))
            else:
                pri[name] = None
            if (idx + k) % 4 == 0:
                bnd[name] = None
            elif (idx + k) % 7 == 3:
                bnd[name] = (np.full(cnt, 0.25), np.full(cnt, 0.25))  # fixed dimension
            else:
                bnd[name] = (np.full(cnt, -4.0 - rng.uniform()), np.full(cnt, 4.0 + rng.uniform()))
        gp.set_priors(pri)
        gp.set_bounds(bnd)
        gp.hyper_priors["df"][np.isnan(gp.hyper_priors["df"])] = 7  # what fit() does (:1029)
        gp._GP__recompute_normalization_constants()
        tag = f"p{idx:03d}"
        H = 3.0 * rng.standard_normal((6, hyp_N))
        H[:, gp.lower_bounds == gp.upper_bounds] = 0.25
        H[5, :] = np.where(np.isnan(gp.lower_bounds), H[5], 0.25)
        lp = np.zeros(6)
        dlp = np.zeros((6, hyp_N))
        for r in range(6):
            lp[r], dlp[r] = gp._GP__compute_log_priors(H[r], True)
        for k in ("mu", "sigma", "df", "a", "b"):
            out[tag + "_" + k] = gp.hyper_priors[k]
        out[tag + "_lb"], out[tag + "_ub"] = gp.lower_bounds, gp.upper_bounds
        out[tag + "_norm"] = gp.normalization_constants
        out[tag + "_H"], out[tag + "_lp"], out[tag + "_dlp"] = H, lp, dlp
        # design: reference f_min_fill with an analytic objective under a fixed global seed
        LB = np.where(np.isnan(gp.lower_bounds), -5.0, gp.lower_bounds)
        UB = np.where(np.isnan(gp.upper_bounds), 5.0, gp.upper_bounds)
        PLB, PUB = np.maximum(LB, -2.0), np.minimum(UB, 2.0)
        PLB, PUB = np.minimum(PLB, PUB), np.maximum(PLB, PUB)
        x0 = np.clip(0.5 * rng.standard_normal((2, hyp_N)), LB, UB)
        np.random.seed(4321 + idx)
        fobj = lambda h: float(np.sum((h - 0.3) ** 2) + np.sin(3 * h[0]))
        Xd, yd = f_min_fill(fobj, x0, LB, UB, PLB, PUB, gp.hyper_priors, 40, "sobol")
        out[tag + "_dLB"], out[tag + "_dUB"], out[tag + "_dPLB"], out[tag + "_dPUB"] = LB, UB, PLB, PUB
        out[tag + "_dx0"], out[tag + "_dX"], out[tag + "_dy"] = x0, Xd, yd
        names.append(f"{tag}|{hyp_N}")
    # smooth-box priors on SEVERAL dimensions in different regions: the reference broadcasts the
    # per-class normaliser against the subset outside [a, b] (:1346-1356): one dimension outside ->
    # its term is counted once per class member; two of three outside -> numpy's broadcasting error
    for kind, tag, Dq in (("smoothbox", "q0", 3), ("smoothbox_student_t", "q1", 3), ("smoothbox", "q2", 2),
                          ("smoothbox_student_t", "q3", 2)):
        gp = gpr.GP(D=Dq, covariance=KERNELS["se"](), mean=MEANS["const"](), noise=make_noise((1, 0, 0)))
        par = (np.array([-1.0, -1.5, -0.5])[:Dq], np.array([1.0, 0.5, 1.5])[:Dq], np.array([0.7, 0.9, 1.1])[:Dq])
        pri = {name: None for name, _ in (gp.covariance.hyperparameter_info(Dq) + gp.noise.hyperparameter_info()
                                          + gp.mean.hyperparameter_info(Dq))}
        pri["covariance_log_lengthscale"] = (kind, par if kind == "smoothbox" else par + (np.array([3.0, 4.0, 5.0])[:Dq],))
        gp.set_priors(pri)
        gp.hyper_priors["df"][np.isnan(gp.hyper_priors["df"])] = 7
        gp._GP__recompute_normalization_constants()
        if Dq == 3:
            H = np.array([[0.0, 0.0, 0.0, 0.1, 0.2, 0.3],     # all inside
                          [2.0, 0.0, 0.0, 0.1, 0.2, 0.3],     # one above, two inside: broadcasting error
                          [-3.0, -3.0, 2.5, 0.1, 0.2, 0.3],   # all outside
                          [2.0, -3.0, 0.0, 0.1, 0.2, 0.3]])   # two outside, one inside: broadcasting error
        else:
            H = np.array([[0.0, 0.0, 0.1, 0.2, 0.3],          # both inside
                          [2.0, 0.0, 0.1, 0.2, 0.3],          # one above, one inside: terms counted twice
                          [-3.0, 2.5, 0.1, 0.2, 0.3],         # both outside
                          [0.5, -4.0, 0.1, 0.2, 0.3]])        # one inside, one below: terms counted twice
        lp, dlp, raised = np.full(4, np.nan), np.full((4, H.shape[1]), np.nan), np.zeros(4, bool)
        for r in range(4):
            try:
                lp[r], dlp[r] = gp._GP__compute_log_priors(H[r], True)
            except ValueError:
                raised[r] = True
        for k in ("mu", "sigma", "df", "a", "b"):
            out[tag + "_" + k] = gp.hyper_priors[k]
        out[tag + "_lb"], out[tag + "_ub"], out[tag + "_norm"] = gp.lower_bounds, gp.upper_bounds, gp.normalization_constants
        out[tag + "_H"], out[tag + "_lp"], out[tag + "_dlp"], out[tag + "_raised"] = H, lp, dlp, raised
        print(tag, kind, "lp", lp, "raised", raised)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "prior_cases.npz"), **out)
    print("prior cases:", len(names))


def fit_cases():
    """GP.fit end to end (examples/example_2.py shape: seeded data, priors, design, L-BFGS-B
    multi-start, slice sampling, update, predict) from the reference."""
    out = {}
    for idx, (kname, N, D) in enumerate([("se", 20, 2), ("matern5", 40, 1)]):
        np.random.seed(1235 + idx)
        X = np.random.uniform(low=-3, high=3, size=(N, D))
        y = np.reshape(np.sin(np.sum(X, 1)) + np.random.normal(scale=0.1, size=N), (-1, 1))
        gp = gpr.GP(D=D, covariance=KERNELS[kname](), mean=MEANS["const"](), noise=make_noise((1, 0, 0)))
        gp.set_priors({
            "covariance_log_outputscale": ("student_t", (0, np.log(10), 3)),
            "covariance_log_lengthscale": ("gaussian", (np.log(np.std(X, ddof=1)), np.log(10))),
            "noise_log_scale": ("gaussian", (np.log(1e-3), 1.0)),
            "mean_const": ("smoothbox", (np.min(y), np.max(y), 1.0)),
        })
        opts = {"n_samples": 6, "init_N": 128, "thin": 2, "burn": 12, "opts_N": 3}
        hyp, opt_res, samp = gp.fit(X=X, y=y, options=opts)
        xs = np.random.uniform(-3, 3, size=(15, D))
        mu, s2 = gp.predict(xs, add_noise=False)
        tag = f"f{idx}"
        out[tag + "_X"], out[tag + "_y"], out[tag + "_hyp"] = X, y, hyp
        out[tag + "_opt_x"], out[tag + "_opt_fun"] = opt_res.x, np.array(opt_res.fun)
        out[tag + "_xs"], out[tag + "_mu"], out[tag + "_s2"] = xs, mu, s2
        out[tag + "_lb"], out[tag + "_ub"] = gp.lower_bounds, gp.upper_bounds
        print(tag, kname, "opt fun", opt_res.fun, "hyp[0]", hyp[0])
    # f2: the reference's own examples/example_1.py -- same seed, data, model, prior and options
    # (Matern-3 + NegativeQuadratic + constant and user-provided noise, Student-t prior, N=31, D=1)
    from scipy.stats import norm

    np.random.seed(1234)
    N, D = 31, 1
    X = -5 + np.random.rand(N, 1) * 10
    s2 = 0.05 * np.exp(0.5 * X)
    y = np.sin(X) + np.sqrt(s2) * norm.ppf(np.random.random_sample(X.shape))
    y[y < 0] = -np.abs(3 * y[y < 0]) ** 2
    gp = gpr.GP(D=D, covariance=gpr.covariance_functions.Matern(degree=3),
                mean=gpr.mean_functions.NegativeQuadratic(),
                noise=gpr.noise_functions.GaussianNoise(constant_add=True, user_provided_add=True))
    gp.set_priors({"covariance_log_lengthscale": None, "covariance_log_outputscale": None, "mean_const": None,
                   "mean_location": None, "mean_log_scale": None,
                   "noise_log_scale": ("student_t", (np.log(1e-3), 1.0, 7))})
    hyp, opt_res, samp = gp.fit(X=X, y=y, s2=s2, options={"n_samples": 10})
    x_star = np.reshape(np.linspace(-15, 15, 200), (-1, 1))
    fmu, fs2 = gp.predict(x_star, add_noise=False)
    out["f2_X"], out["f2_y"], out["f2_s2"], out["f2_hyp"] = X, y, s2, hyp
    out["f2_opt_x"], out["f2_opt_fun"] = opt_res.x, np.array(opt_res.fun)
    out["f2_mu"], out["f2_fs2"] = fmu, fs2
    out["f2_lb"], out["f2_ub"] = gp.lower_bounds, gp.upper_bounds
    print("f2 example_1 opt fun", opt_res.fun, "hyp[0]", hyp[0])
    np.savez_compressed(os.path.join(HERE, "fit_cases.npz"), **out)


def full_cases():
    """GP.predict_full and GP.quad from the reference."""
    out = {}
    names = []
    cases = [("se", "const", (1, 0, 0), 33, 2), ("se", "negquad", (1, 0, 0), 130, 3),
             ("se", "zero", (0, 0, 0), 33, 2), ("matern5", "const", (1, 1, 0), 40, 2),
             ("se", "const", (1, 0, 0), 200, 1)]
    for idx, (kname, mname, npar, N, D) in enumerate(cases):
        rng = np.random.default_rng(15000 + idx)
        cov, mean, noise = KERNELS[kname](), MEANS[mname](), make_noise(npar)
        gp = gpr.GP(D=D, covariance=cov, mean=mean, noise=noise)
        cov_N, mean_N, noise_N = cov.hyperparameter_count(D), mean.hyperparameter_count(D), noise.hyperparameter_count()
        X = rng.uniform(-3, 3, (N, D))
        y = np.sin(np.sum(X, 1, keepdims=True) / np.sqrt(D)) + 0.1 * rng.standard_normal((N, 1))
        s2 = 0.01 + 0.05 * rng.uniform(size=(N, 1)) if npar[1] else None
        S = 3
        hyp = np.zeros((S, cov_N + noise_N + mean_N))
        for s in range(S):
            h_cov = 0.2 * rng.standard_normal(cov_N)
            h_cov[:D] += np.log(1.2 * np.sqrt(D))
            h_noise = [np.log(0.1) + 0.2 * rng.standard_normal()] if npar[0] else []
            h_mean = {"zero": [], "const": [0.2 * rng.standard_normal()]}.get(
                mname, [0.2 * rng.standard_normal()] + list(0.5 * rng.standard_normal(D))
                + list(np.log(4.0) + 0.2 * rng.standard_normal(D)))
            hyp[s] = np.concatenate([h_cov, h_noise, h_mean])
        gp.update(X_new=X, y_new=y, s2_new=s2, hyp=hyp)
        M = 9
        xs = rng.uniform(-3.5, 3.5, (M, D))
        tag = f"u{idx:03d}"
        names.append(f"{tag}|{kname}|{mname}|{npar[0]}{npar[1]}{npar[2]}|{N}|{D}")
        out[tag + "_X"], out[tag + "_y"], out[tag + "_hyp"], out[tag + "_xs"] = X, y, hyp, xs
        if s2 is not None:
            out[tag + "_s2"] = s2
        s2s = 0.02 * np.ones((M, 1)) if s2 is not None else None
        mu, C = gp.predict_full(xs, None, s2s, add_noise=False)
        out[tag + "_pf_mu"], out[tag + "_pf_cov"] = mu, C
        mu, C = gp.predict_full(xs, None, s2s, add_noise=True)
        out[tag + "_pf_cov_noise"] = C
        if kname == "se" and npar[0] == 1:  # the reference's quad indexes the constant-noise entry
            qm = rng.uniform(-1, 1, (5, D))
            qs = 0.3 + rng.uniform(size=(5, D))
            F, Fv = gp.quad(qm, qs, compute_var=True, separate_samples=True)
            Fa, Fva = gp.quad(qm, qs, compute_var=True)
            out[tag + "_qm"], out[tag + "_qs"] = qm, qs
            out[tag + "_F"], out[tag + "_Fv"], out[tag + "_Fa"], out[tag + "_Fva"] = F, Fv, Fa, Fva
            out[tag + "_F1"] = gp.quad(0.1, 0.5)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "full_cases.npz"), **out)
    print("full cases:", len(names))


def _bench_problem(cfg_idx, N, D, kernel, S):
    """The synthetic workload of SURVEY.md 8(d) (same draws as bench.py / the oracle)."""
    rng = np.random.default_rng(1000 + cfg_idx)
    X = rng.uniform(-3, 3, (N, D))
    y = np.sin(np.sum(X, 1, keepdims=True) / np.sqrt(D)) + 0.1 * rng.standard_normal((N, 1))
    base = [np.log(1.5 * np.sqrt(D) * (1 + 0.1 * d / D)) for d in range(D)] + [0.0]
    if kernel == "rq":
        base.append(0.0)
    base = np.asarray(base + [np.log(0.1), 0.0])
    hyp = base + 0.1 * rng.standard_normal((S, base.size))
    return X, y, hyp


def fullsize_cases():
    """nlZ, dnlZ of the reference at the BASELINE configurations' full sizes.  Inputs are
    NOT stored: they are regenerated from the seed (bench.synthetic_problem)."""
    import time

    out = {}
    for cfg_idx, N, D, kname, S, rows in [(2, 2048, 5, "se", 1, [0]), (3, 4096, 10, "matern5", 16, [0, 1, 15])]:
        X, y, hyp = _bench_problem(cfg_idx, N, D, kname, S)
        gp = gpr.GP(D=D, covariance=KERNELS[kname](), mean=MEANS["const"](), noise=make_noise((1, 0, 0)))
        gp.X, gp.y = X, y
        nl, dn = [], []
        for s in rows:
            t0 = time.time()
            a, b = gp._GP__compute_nlZ(hyp[s], True, False)
            nl.append(a)
            dn.append(b)
            print(f"cfg{cfg_idx} s={s}: nlZ={a!r} ({time.time() - t0:.1f} s)", flush=True)
        out[f"cfg{cfg_idx}_rows"] = np.array(rows)
        out[f"cfg{cfg_idx}_hyp"] = hyp[rows]
        out[f"cfg{cfg_idx}_nlZ"] = np.array(nl)
        out[f"cfg{cfg_idx}_dnlZ"] = np.stack(dn)
        out[f"cfg{cfg_idx}_Xsum"] = np.array([X.sum(), y.sum()])  # guards the regenerated inputs
    np.savez_compressed(os.path.join(HERE, "fullsize_cases.npz"), **out)


def fullsize45_cases():
    """Reference values at the two largest BASELINE configurations (inputs regenerated from the seed)."""
    import time

    out = {}
    for cfg_idx, N, D, kname, S, rows_grad, rows_nll in [(5, 8192, 8, "se", 64, [0, 63], [7, 8]),
                                                         (4, 16384, 20, "rq", 1, [], [0])]:
        X, y, hyp = _bench_problem(cfg_idx, N, D, kname, S)
        gp = gpr.GP(D=D, covariance=KERNELS[kname](), mean=MEANS["const"](), noise=make_noise((1, 0, 0)))
        gp.X, gp.y = X, y
        nl, dn = [], []
        for s in rows_grad:
            t0 = time.time()
            a, b = gp._GP__compute_nlZ(hyp[s], True, False)
            nl.append(a)
            dn.append(b)
            print(f"cfg{cfg_idx} s={s}: nlZ={a!r} ({time.time() - t0:.1f} s)", flush=True)
        for s in rows_nll:
            t0 = time.time()
            a = gp._GP__compute_nlZ(hyp[s], False, False)
            nl.append(a)
            print(f"cfg{cfg_idx} s={s}: nlZ={a!r} (no gradient, {time.time() - t0:.1f} s)", flush=True)
        out[f"cfg{cfg_idx}_rows"] = np.array(rows_grad + rows_nll)
        out[f"cfg{cfg_idx}_rows_with_grad"] = np.array(rows_grad, dtype=int)
        out[f"cfg{cfg_idx}_hyp"] = hyp[rows_grad + rows_nll]
        out[f"cfg{cfg_idx}_nlZ"] = np.array(nl)
        out[f"cfg{cfg_idx}_dnlZ"] = np.stack(dn) if dn else np.zeros((0, hyp.shape[1]))
        out[f"cfg{cfg_idx}_Xsum"] = np.array([X.sum(), y.sum()])
    np.savez_compressed(os.path.join(HERE, "fullsize45_cases.npz"), **out)


def cfg4_gradient():
    """cfg4's gradient, which the reference cannot form (covariance_functions.py:349-363 materialises a
    (22, 16384, 16384) tensor = 47 GB before gaussian_process.py:2487-2488 contracts it): the pinned oracle's
    streamed restatement (bit-identical to the reference on every core fixture and on cfg3 sample 0, see
    tests/test_oracle_golden.py), cross-checked here on the one number the reference CAN produce at this size."""
    import time

    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import gp_oracle as orc

    path = os.path.join(HERE, "fullsize45_cases.npz")
    out = dict(np.load(path, allow_pickle=False))
    X, y, hyp = _bench_problem(4, 16384, 20, "rq", 1)
    assert np.array_equal(hyp[0], out["cfg4_hyp"][0])
    model = dict(kernel="rq", degree=0, mean="const", noise=(1, 0, 0))
    t0 = time.time()
    nlZ, dnlZ = orc.core_streamed(model, hyp[0], X, y, None)
    print(f"cfg4 streamed oracle: nlZ={nlZ!r} ({time.time() - t0:.0f} s); reference nlZ={out['cfg4_nlZ'][0]!r}", flush=True)
    print("dnlZ =", np.array2string(dnlZ, precision=17))
    assert nlZ == out["cfg4_nlZ"][0], "the streamed oracle's nlZ is not the reference's"
    out["cfg4_dnlZ"] = dnlZ[None, :]
    out["cfg4_dnlZ_source"] = np.array("oracle.core_streamed (oracle-derived: the reference needs 47 GB here)")
    np.savez_compressed(path, **out)


def big20k_case():
    """Beyond the former N <= 16384 ceiling of the device library: N = 20480, D = 5, squared exponential, one sample
    (the inputs are bench.py's `--config 6` draw at that N, regenerated from the seed).  The reference computes nlZ
    (no gradient: its (N, N, 6) tensor would need 20 GB on top of K, L and the inverse); the gradient is ORACLE-DERIVED
    (oracle.core_streamed, one plane at a time), written only after its nlZ has been found equal to the reference's
    bit for bit, and labelled as such."""
    import time

    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import gp_oracle as orc

    N, D = 20480, 5
    X, y, hyp = _bench_problem(6, N, D, "se", 2)
    gp = gpr.GP(D=D, covariance=KERNELS["se"](), mean=MEANS["const"](), noise=make_noise((1, 0, 0)))
    gp.X, gp.y = X, y
    t0 = time.time()
    ref_nlZ = gp._GP__compute_nlZ(hyp[0], False, False)
    print(f"big20k reference: nlZ={ref_nlZ!r} (no gradient, {time.time() - t0:.0f} s)", flush=True)
    model = dict(kernel="se", degree=0, mean="const", noise=(1, 0, 0))
    t0 = time.time()
    nlZ, dnlZ = orc.core_streamed(model, hyp[0], X, y, None)
    print(f"big20k streamed oracle: nlZ={nlZ!r} ({time.time() - t0:.0f} s)", flush=True)
    print("dnlZ =", np.array2string(dnlZ, precision=17))
    assert nlZ == ref_nlZ, "the streamed oracle's nlZ is not the reference's"
    out = dict(N=np.array(N), D=np.array(D), hyp=hyp[:1], nlZ=np.array([ref_nlZ]), dnlZ=dnlZ[None, :],
               dnlZ_source=np.array("oracle.core_streamed (oracle-derived; its nlZ equals the reference's bit for bit)"),
               Xsum=np.array([X.sum(), y.sum()]))
    np.savez_compressed(os.path.join(HERE, "big20k_case.npz"), **out)


def rank1_cases():
    """GP.update(X_new=1 point, y_new) through the reference's rank-one path (:750-844):
    three consecutive appends; posterior fields after the last one, predictions after each."""
    out = {}
    names = []
    # "low": constant noise below the 1e-6 switch (L_chol False, Posterior.L = -inv) with a rough,
    # short-lengthscale kernel so that K + sn2 I stays well conditioned and the fixture is not
    # rounding dependent
    cases = [("se", "const", (1, 0, 0), 33, 2, "high"), ("matern5", "const", (1, 0, 0), 126, 3, "high"),
             ("se", "negquad", (1, 0, 0), 128, 2, "high"), ("matern1", "const", (1, 0, 0), 33, 2, "low"),
             ("matern3", "zero", (1, 0, 0), 127, 2, "low"), ("rq", "const", (1, 0, 0), 40, 2, "high"),
             ("matern3", "const", (1, 0, 0), 128, 3, "low")]
    for idx, (kname, mname, npar, N, D, flav) in enumerate(cases):
        rng = np.random.default_rng(17000 + idx)
        cov, mean, noise = KERNELS[kname](), MEANS[mname](), make_noise(npar)
        gp = gpr.GP(D=D, covariance=cov, mean=mean, noise=noise)
        cov_N, mean_N, noise_N = cov.hyperparameter_count(D), mean.hyperparameter_count(D), noise.hyperparameter_count()
        X = rng.uniform(-3, 3, (N, D))
        y = np.sin(np.sum(X, 1, keepdims=True) / np.sqrt(D)) + 0.1 * rng.standard_normal((N, 1))
        S = 3
        hyp = np.zeros((S, cov_N + noise_N + mean_N))
        for s in range(S):
            h_cov = 0.2 * rng.standard_normal(cov_N)
            h_cov[:D] += np.log(1.2 * np.sqrt(D)) if flav == "high" else np.log(0.4)
            h_noise = [(np.log(0.1) if flav == "high" else np.log(3e-4)) + 0.2 * rng.standard_normal()]
            h_mean = {"zero": [], "const": [0.2 * rng.standard_normal()]}.get(
                mname, [0.2 * rng.standard_normal()] + list(0.5 * rng.standard_normal(D))
                + list(np.log(4.0) + 0.2 * rng.standard_normal(D)))
            hyp[s] = np.concatenate([h_cov, h_noise, h_mean])
        gp.update(X_new=X, y_new=y, hyp=hyp)
        assert all(p.sn2_mult == 1 for p in gp.posteriors)
        assert all(bool(p.L_chol) == (flav == "high") for p in gp.posteriors)
        tag = f"r{idx:03d}"
        names.append(f"{tag}|{kname}|{mname}|{npar[0]}{npar[1]}{npar[2]}|{N}|{D}|{flav}")
        out[tag + "_X"], out[tag + "_y"], out[tag + "_hyp"] = X, y, hyp
        xs = rng.uniform(-3.5, 3.5, (7, D))
        out[tag + "_xs"] = xs
        Xn = rng.uniform(-3, 3, (3, D))
        yn = np.sin(np.sum(Xn, 1, keepdims=True) / np.sqrt(D)) + 0.1 * rng.standard_normal((3, 1))
        out[tag + "_Xn"], out[tag + "_yn"] = Xn, yn
        for k in range(3):
            import warnings
            with warnings.catch_warnings(record=True) as wlist:
                warnings.simplefilter("always")
                gp.update(X_new=Xn[k:k + 1], y_new=yn[k:k + 1])
            assert not wlist, "an unstable rank-one update would make the fixture rounding dependent"
            mu, s2 = gp.predict(xs, separate_samples=True)
            out[tag + f"_mu{k}"], out[tag + f"_s2{k}"] = mu, s2
        out[tag + "_alpha"] = np.stack([p.alpha[:, 0] for p in gp.posteriors])
        out[tag + "_sW"] = np.stack([p.sW[:, 0] for p in gp.posteriors])
        out[tag + "_L_chol"] = np.array([bool(p.L_chol) for p in gp.posteriors])
        out[tag + "_Ldiag"] = np.stack([np.diag(p.L) for p in gp.posteriors])
        out[tag + "_Llast_col"] = np.stack([np.asarray(p.L)[:, -1] for p in gp.posteriors])
        out[tag + "_Llast_row"] = np.stack([np.asarray(p.L)[-1, :] for p in gp.posteriors])
        out[tag + "_Lfro"] = np.array([np.linalg.norm(p.L) for p in gp.posteriors])
        conds = [np.linalg.cond(np.asarray(p.L)) for p in gp.posteriors]
        print(names[-1], "L_chol", out[tag + "_L_chol"], "alpha[-1]", out[tag + "_alpha"][:, -1], "cond(L) %.1e" % max(conds))
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "rank1_cases.npz"), **out)


def draw_cases():
    """GP.random_function (:2241-2329) with np.random seeded before every call, and __robust_cholesky (:2331-2355)
    on matrices that LAPACK's Cholesky rejects."""
    out = {}
    names = []
    cases = [("se", "const", (1, 0, 0), 33, 2, True), ("matern5", "const", (1, 1, 0), 40, 2, True),
             ("se", "negquad", (1, 0, 0), 130, 3, True), ("se", "const", (1, 0, 0), 0, 2, False)]
    for idx, (kname, mname, npar, N, D, data) in enumerate(cases):
        rng = np.random.default_rng(17000 + idx)
        cov, mean, noise = KERNELS[kname](), MEANS[mname](), make_noise(npar)
        gp = gpr.GP(D=D, covariance=cov, mean=mean, noise=noise)
        cov_N, mean_N, noise_N = cov.hyperparameter_count(D), mean.hyperparameter_count(D), noise.hyperparameter_count()
        S = 3
        hyp = np.zeros((S, cov_N + noise_N + mean_N))
        for s in range(S):
            h_cov = 0.2 * rng.standard_normal(cov_N)
            h_cov[:D] += np.log(1.2 * np.sqrt(D))
            h_noise = [np.log(0.1) + 0.2 * rng.standard_normal()] if npar[0] else []
            h_mean = {"zero": [], "const": [0.2 * rng.standard_normal()]}.get(
                mname, [0.2 * rng.standard_normal()] + list(0.5 * rng.standard_normal(D))
                + list(np.log(4.0) + 0.2 * rng.standard_normal(D)))
            hyp[s] = np.concatenate([h_cov, h_noise, h_mean])
        tag = f"d{idx:03d}"
        names.append(f"{tag}|{kname}|{mname}|{npar[0]}{npar[1]}{npar[2]}|{N}|{D}|{int(data)}")
        out[tag + "_hyp"] = hyp
        if data:
            X = rng.uniform(-3, 3, (N, D))
            y = np.sin(np.sum(X, 1, keepdims=True) / np.sqrt(D)) + 0.1 * rng.standard_normal((N, 1))
            s2 = 0.01 + 0.05 * rng.uniform(size=(N, 1)) if npar[1] else None
            out[tag + "_X"], out[tag + "_y"] = X, y
            if s2 is not None:
                out[tag + "_s2"] = s2
            gp.update(X_new=X, y_new=y, s2_new=s2, hyp=hyp)
        else:
            gp.update(hyp=hyp)
        xs = rng.uniform(-3.5, 3.5, (7, D))
        out[tag + "_xs"] = xs
        for k in range(4):  # several seeds: different hyperparameter samples get picked
            np.random.seed(900 + 10 * idx + k)
            out[tag + f"_f{k}"] = gp.random_function(xs)
            np.random.seed(900 + 10 * idx + k)
            out[tag + f"_y{k}"] = gp.random_function(xs, add_noise=True)
    # the factor of matrices that are not numerically positive definite
    rng = np.random.default_rng(17100)
    A = rng.standard_normal((6, 3))
    B = rng.standard_normal((5, 5))
    mats = [A @ A.T,                                   # rank 3 of 6: directions dropped
            np.ones((4, 4)),                           # rank 1
            (B + B.T) / 2,                             # indefinite: nothing to draw from
            np.diag([2.0, 1.0, 0.0, 0.5])]             # an exact zero on the diagonal
    for k, Cm in enumerate(mats):
        out[f"rc{k}_C"] = Cm
        out[f"rc{k}_T"] = gp._GP__robust_cholesky(Cm.copy())
    out["n_rc"] = np.array(len(mats))
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "draw_cases.npz"), **out)
    print("draw cases:", len(names), "+", len(mats), "factor cases")


if __name__ == "__main__":
    which = sys.argv[1:] or ["cov", "core", "prior", "fit", "full", "fullsize", "rank1", "draw"]
    if "draw" in which:
        draw_cases()
    if "fullsize" in which:
        fullsize_cases()
    if "fullsize45" in which:  # not in the default list: ~10 minutes and ~15 GB of host memory
        fullsize45_cases()
    if "cfg4grad" in which:  # ~10 minutes, ~20 GB; oracle-derived (see cfg4_gradient)
        cfg4_gradient()
    if "big20k" in which:  # ~15 minutes, ~25 GB; gradient oracle-derived (see big20k_case)
        big20k_case()
    if "rank1" in which:
        rank1_cases()
    if "cov" in which:
        cov_cases()
    if "core" in which:
        core_cases()
    if "prior" in which:
        prior_cases()
    if "fit" in which:
        fit_cases()
    if "full" in which:
        full_cases()
    sys.exit(0)
