#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the *reference* (PyBMF @ 2024_10_08).

Runs only in the build container, where the reference is mounted read-only at /root/reference.
Nothing of the reference is written here: the outputs are inputs/expected-output vectors (.npz/.json).

    python tests/golden/make_golden.py

The loader follows SURVEY.md Appendix A: three optional third-party packages that the reference imports
at module scope but that are absent from this image (IPython, p_tqdm, mlxtend) are registered as empty
in-memory modules; they are never called on the hot path.
"""
import contextlib
import hashlib
import io
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    ip = stub("IPython")
    ip.display = stub("IPython.display", display=lambda *a, **k: None)
    stub("p_tqdm", p_map=lambda f, *its, **k: list(map(f, *its)))
    ml = stub("mlxtend")
    ml.frequent_patterns = stub("mlxtend.frequent_patterns", apriori=None)
    import matplotlib
    matplotlib.use("Agg")
    import PyBMF  # noqa: F401
    return PyBMF


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        yield


FIT_KW = dict(task="reconstruction", show_logs=False, show_result=False, save_model=False)


def df_rows(df, skip_time=True):
    cols = [tuple(str(x) for x in c) for c in df.columns]
    rows = []
    for _, r in df.iterrows():
        rows.append([None if isinstance(v, str) else float(v) for v in r.tolist()])
    if skip_time:
        rows = [r[1:] for r in rows]
        cols = cols[1:]
    return {"columns": cols, "rows": rows}


def staged_fit(model, X):
    """fit() split at the point where the initial state exists (models/BaseModel.py:44-66)."""
    model.check_params(**FIT_KW)
    model.load_dataset(X_train=X, X_val=None, X_test=None)
    model.init_model()
    init = (model.U.copy(), model.V.copy())
    return init


def counts_of(PyBMF, X_csr, X_pd):
    from PyBMF.utils import TP, FP, TN, FN
    return [int(TP(X_csr, X_pd)), int(FP(X_csr, X_pd)), int(FN(X_csr, X_pd)), int(TN(X_csr, X_pd))]


def g1_penalty_trajectory(PyBMF):
    from PyBMF.generators import SyntheticMatrixGenerator
    from PyBMF.models import BinaryMFPenalty
    from PyBMF.utils import get_prediction_with_threshold
    mod = sys.modules["PyBMF.models.BinaryMFPenalty"]   # the module, not the class of the same name
    with quiet():
        gen = SyntheticMatrixGenerator(m=1000, n=500, k=8, density=[0.2, 0.2])
        gen.generate(seed=1000)
        sum_clean = int(gen.X.sum())
        gen.add_noise(noise=[0.05, 0.01], seed=2000)
        X = gen.X.toarray().astype(np.uint8)
        model = BinaryMFPenalty(k=8, U=None, V=None, W="full", reg=1, reg_growth=1.02, init_method="normal",
                                normalize_method="balance", max_iter=20, seed=2024)
        U0, V0 = staged_fit(model, gen.X)
        # one literal update from the initial state, for per-step parity
        V1 = mod.update_V(X=model.X_train, W=model.W, U=model.U, V=model.V, reg=model.reg)
        U1 = mod.update_U(X=model.X_train, W=model.W, U=model.U, V=V1, reg=model.reg)
        model._fit()
        X_pd = get_prediction_with_threshold(U=model.U, V=model.V, u=0.5, v=0.5)
        cnt = counts_of(PyBMF, gen.X, X_pd)
    np.savez_compressed(os.path.join(HERE, "g1_penalty_c1.npz"),
                        X_bits=np.packbits(X, axis=1, bitorder="little"), shape=np.array(X.shape),
                        U0=U0, V0=V0, U1=U1, V1=V1, U_final=model.U, V_final=model.V)
    meta = {"sum_clean": sum_clean, "sum_noisy": int(X.sum()),
            "row_sums_head": X.sum(1)[:5].tolist(), "col_sums_head": X.sum(0)[:5].tolist(),
            "params": dict(k=8, W="full", reg=1, reg_growth=1.02, init_method="normal",
                           normalize_method="balance", max_iter=20, seed=2024),
            "generator": dict(m=1000, n=500, k=8, density=[0.2, 0.2], seed=1000, noise=[0.05, 0.01], noise_seed=2000),
            "final_reg": float(model.reg), "final_counts_TP_FP_FN_TN": cnt,
            "updates": df_rows(model.logs["updates"]), "boolean": df_rows(model.logs["boolean"]),
            "attrs": sorted(k for k in model.__dict__.keys())}
    json.dump(meta, open(os.path.join(HERE, "g1_penalty_c1.json"), "w"), indent=1)
    return X, U0, V0


def g2_single_steps(PyBMF):
    mod = sys.modules["PyBMF.models.BinaryMFPenalty"]
    out = {}
    rs = np.random.RandomState(77)
    case = 0
    for (m, n, k) in [(67, 45, 5), (128, 96, 16), (33, 200, 8)]:
        X = (rs.rand(m, n) < 0.3).astype(np.float64)
        U = np.abs(rs.standard_normal((m, k))) * 0.4
        V = np.abs(rs.standard_normal((n, k))) * 0.4
        if case == 0:
            U[:, 2] = 0.0          # zero column -> denom==0 -> eps and F==0 -> eps paths in update_V / update_U
            X[:, 7] = 0.0          # empty data column
        W = np.ones((m, n))
        for reg in (0.0, 1.0, 1e3):
            with quiet():
                V1 = mod.update_V(X=X, W=W, U=U, V=V, reg=np.float64(reg))
                U1 = mod.update_U(X=X, W=W, U=U, V=V1, reg=np.float64(reg))
                err = mod.error(X_gt=X, X_pd=U1 @ V1.T, W=W, U=U1, V=V1, reg=np.float64(reg))
            tag = f"c{case}_r{reg:g}"
            out[tag + "_V1"], out[tag + "_U1"] = V1, U1
            out[tag + "_err"] = np.array(err, dtype=np.float64)
        out[f"c{case}_X"], out[f"c{case}_U"], out[f"c{case}_V"] = X.astype(np.uint8), U, V
        case += 1
    np.savez_compressed(os.path.join(HERE, "g2_penalty_steps.npz"), **out)


def g3_wnmf(PyBMF):
    from PyBMF.models import WNMF
    rs = np.random.RandomState(5)
    m, n, k = 180, 130, 12
    A, B = rs.rand(m, k), rs.rand(k, n)
    X = ((A @ B) / k + 0.01 * rs.rand(m, n)).astype(np.float32).astype(np.float64)
    X[rs.rand(m, n) < 0.1] = 0.0     # exact zeros: exercise the in-place eps quirk and the 'mask' pattern
    res = {}
    for W in ("full", "mask"):
        with quiet():
            model = WNMF(k=k, W=W, init_method="normal", max_iter=10, seed=2024)
            U0, V0 = staged_fit(model, X.copy())
            model._fit()
        res[W] = dict(U0=U0, V0=V0, U=model.U, V=model.V, X_after=np.asarray(model.X_train),
                      rows=df_rows(model.logs["updates"]))
    np.savez_compressed(os.path.join(HERE, "g3_wnmf.npz"), X=X,
                        **{f"{w}_{key}": res[w][key] for w in res for key in ("U0", "V0", "U", "V", "X_after")})
    json.dump({w: res[w]["rows"] for w in res} | {"params": dict(k=k, init_method="normal", max_iter=10, seed=2024)},
              open(os.path.join(HERE, "g3_wnmf.json"), "w"), indent=1)


def g4_threshold(PyBMF):
    from PyBMF.generators import SyntheticMatrixGenerator
    from PyBMF.models import WNMF, BinaryMFThreshold
    with quiet():
        gen = SyntheticMatrixGenerator(m=600, n=400, k=16, density=[0.15, 0.15])
        gen.generate(seed=31)
        gen.add_noise(noise=[0.05, 0.01], seed=32)
        X = gen.X.toarray().astype(np.float64)
        w = WNMF(k=16, W="full", init_method="normal", max_iter=60, seed=9)
        w.fit(X.copy(), **FIT_KW)
    U, V = w.U.copy(), w.V.copy()
    grid_u = [0.1, 0.25, 0.4, 0.55, 0.7]
    grid_v = [0.1, 0.25, 0.4, 0.55, 0.7]
    out = {"U": U, "V": V, "X_bits": np.packbits(X.astype(np.uint8), axis=1, bitorder="little"), "shape": np.array(X.shape)}
    meta = {"grid_u": grid_u, "grid_v": grid_v}
    for lam in (10, 100):
        with quiet():
            model = BinaryMFThreshold(k=16, U=U.copy(), V=V.copy(), W="full", u=0.5, v=0.5, lamda=lam,
                                      min_diff=1e-3, max_iter=100)
            staged_fit(model, X.copy())
            Fg = np.array([[model.F([u, v]) for v in grid_v] for u in grid_u])
            dFg = np.array([[model.dF([u, v]) for v in grid_v] for u in grid_u])
            calls = {"F": 0, "dF": 0}
            F0, dF0 = model.F, model.dF

            def F(x, _f=F0):
                calls["F"] += 1
                return _f(x)

            def dF(x, _g=dF0):
                calls["dF"] += 1
                return _g(x)
            model.F, model.dF = F, dF
            model._fit()
        out[f"F_grid_lam{lam}"], out[f"dF_grid_lam{lam}"] = Fg, dFg
        meta[f"lam{lam}"] = {"rows": df_rows(model.logs["updates"]), "u": float(model.u), "v": float(model.v),
                             "calls": calls}
    np.savez_compressed(os.path.join(HERE, "g4_threshold.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "g4_threshold.json"), "w"), indent=1)


def g5_metrics(PyBMF):
    from PyBMF.utils import get_metrics, to_sparse
    rs = np.random.RandomState(3)
    cases = []
    names = ["TP", "FP", "TN", "FN", "Recall", "Precision", "Accuracy", "F1", "RMSE", "MAE"]
    for i, (m, n, pg, pp) in enumerate([(40, 30, 0.3, 0.3), (17, 65, 0.5, 0.1), (25, 25, 0.2, 0.0), (25, 25, 0.0, 0.2),
                                        (64, 64, 0.9, 0.9)]):
        gt = (rs.rand(m, n) < pg).astype(np.int64)
        pd = (rs.rand(m, n) < pp).astype(np.int64)
        with quiet():
            r = get_metrics(gt=to_sparse(gt, "csr"), pd=to_sparse(pd, "csr"), metrics=names)
        cases.append({"gt_bits": np.packbits(gt.astype(np.uint8), axis=1, bitorder="little").tolist(),
                      "pd_bits": np.packbits(pd.astype(np.uint8), axis=1, bitorder="little").tolist(),
                      "shape": [m, n], "metrics": {k: float(v) for k, v in zip(names, r)}})
    json.dump(cases, open(os.path.join(HERE, "g5_metrics.json"), "w"))


def g6_generator(PyBMF):
    from PyBMF.generators import SyntheticMatrixGenerator
    from PyBMF.models import BinaryMFPenalty
    out = []
    for (m, n, k, dens, seed, noise, nseed) in [(1000, 500, 8, [0.2, 0.2], 1000, [0.05, 0.01], 2000),
                                                  (300, 450, 5, [0.1, 0.3], 7, [0.0, 0.0], 8),
                                                  (257, 129, 3, [0.25, 0.15], 123, [0.1, 0.02], 456)]:
        with quiet():
            g = SyntheticMatrixGenerator(m=m, n=n, k=k, density=dens)
            g.generate(seed=seed)
            clean = g.X.toarray().astype(np.uint8)
            g.add_noise(noise=noise, seed=nseed)
            noisy = g.X.toarray().astype(np.uint8)
        out.append(dict(m=m, n=n, k=k, density=dens, seed=seed, noise=noise, noise_seed=nseed,
                        sum_clean=int(clean.sum()), sum_noisy=int(noisy.sum()),
                        sha_clean=hashlib.sha256(np.packbits(clean, axis=1, bitorder="little").tobytes()).hexdigest(),
                        sha_noisy=hashlib.sha256(np.packbits(noisy, axis=1, bitorder="little").tobytes()).hexdigest()))
    inits = {}
    X = (np.random.RandomState(0).rand(60, 40) < 0.3).astype(np.float64)
    for method in ("normal", "uniform"):
        with quiet():
            mdl = BinaryMFPenalty(k=4, W="full", init_method=method, normalize_method=None, seed=2024)
            U0, V0 = staged_fit(mdl, X)
        inits[method] = {"U_head": U0.ravel()[:8].tolist(), "V_head": V0.ravel()[:8].tolist()}
    json.dump({"generator": out, "init": inits}, open(os.path.join(HERE, "g6_generator.json"), "w"), indent=1)


def main():
    PyBMF = load_reference()
    g1_penalty_trajectory(PyBMF)
    g2_single_steps(PyBMF)
    g3_wnmf(PyBMF)
    g4_threshold(PyBMF)
    g5_metrics(PyBMF)
    g6_generator(PyBMF)
    g7_masked(PyBMF)
    g8_threshold_masked(PyBMF)
    g9_prediction(PyBMF)
    g10_link_models(PyBMF)
    g11_cover_scores(PyBMF)
    g12_normalize(PyBMF)
    g13_kl_mask(PyBMF)
    g14_palm(PyBMF)
    g15_elbmf_masked(PyBMF)
    g16_pnlpf_masked(PyBMF)
    g17_val_test_sets(PyBMF)
    g18_kl_weights(PyBMF)
    g19_real_valued(PyBMF)
    g20_wide_rank(PyBMF)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))




def g7_masked(PyBMF):
    """Masked updates (W='mask'): X given as csr whose STORED entries (ones and explicit zeros, as after negative sampling)
    are the observed cells.  BinaryMFPenalty and WNMF, 8 iterations each."""
    from scipy.sparse import csr_matrix
    from PyBMF.models import BinaryMFPenalty, WNMF
    rs = np.random.RandomState(17)
    m, n, k = 150, 110, 6
    A = (rs.rand(m, k) < 0.25).astype(int)
    B = (rs.rand(n, k) < 0.25).astype(int)
    Xfull = np.minimum(A @ B.T, 1)
    obs = rs.rand(m, n) < 0.3
    obs[5, :] = False            # a row without any observed cell
    obs[:, 9] = False            # a column without any observed cell
    r, c = np.nonzero(obs)
    X = csr_matrix((Xfull[r, c].astype(np.float64), (r, c)), shape=(m, n))   # explicit zeros are kept as stored entries
    assert X.nnz == obs.sum()
    out = {"rows": r.astype(np.int32), "cols": c.astype(np.int32), "vals": Xfull[r, c].astype(np.uint8), "shape": np.array([m, n])}
    meta = {}
    with quiet():
        mdl = BinaryMFPenalty(k=k, W="mask", reg=1.0, reg_growth=1.3, init_method="normal", normalize_method="balance",
                              max_iter=7, seed=4)
        U0, V0 = staged_fit(mdl, X.copy())
        mdl._fit()
    out.update(p_U0=U0, p_V0=V0, p_U=mdl.U, p_V=mdl.V)
    meta["penalty"] = {"updates": df_rows(mdl.logs["updates"]), "boolean": df_rows(mdl.logs["boolean"]), "final_reg": float(mdl.reg)}
    with quiet():
        w = WNMF(k=k, W="mask", init_method="normal", max_iter=7, seed=4)
        U0, V0 = staged_fit(w, X.copy())
        w._fit()
    out.update(w_U0=U0, w_V0=V0, w_U=w.U, w_V=w.V)
    meta["wnmf"] = {"updates": df_rows(w.logs["updates"])}
    np.savez_compressed(os.path.join(HERE, "g7_masked.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "g7_masked.json"), "w"), indent=1)


def g8_threshold_masked(PyBMF):
    """BinaryMFThreshold with its DEFAULT W='mask' on a csr with explicit zeros: objective on a grid + a full line search."""
    from scipy.sparse import csr_matrix
    from PyBMF.models import BinaryMFThreshold
    z = np.load(os.path.join(HERE, "g7_masked.npz"))
    m, n = z["shape"]
    X = csr_matrix((z["vals"].astype(np.float64), (z["rows"], z["cols"])), shape=(m, n))
    U, V = z["w_U"], z["w_V"]       # factors of the masked WNMF run
    grid = [0.05, 0.2, 0.35, 0.5]
    with quiet():
        mdl = BinaryMFThreshold(k=6, U=U.copy(), V=V.copy(), u=0.3, v=0.3, lamda=10, min_diff=1e-3, max_iter=40)
        staged_fit(mdl, X.copy())
        Fg = np.array([[mdl.F([a, b]) for b in grid] for a in grid])
        dFg = np.array([[mdl.dF([a, b]) for b in grid] for a in grid])
        mdl._fit()
    np.savez_compressed(os.path.join(HERE, "g8_threshold_masked.npz"), F_grid=Fg, dF_grid=dFg, grid=np.array(grid))
    json.dump({"rows": df_rows(mdl.logs["updates"]), "u": float(mdl.u), "v": float(mdl.v)},
              open(os.path.join(HERE, "g8_threshold_masked.json"), "w"), indent=1)


def g9_prediction(PyBMF):
    """fit(X_train, X_val, X_test) with task='prediction' (scores over the stored entries of each set) for the three
    models, and task='reconstruction' with val / test sets (whole-matrix scores) for BinaryMFPenalty."""
    from scipy.sparse import csr_matrix
    from PyBMF.models import BinaryMFPenalty, WNMF, BinaryMFThreshold
    rs = np.random.RandomState(23)
    m, n, k = 130, 90, 5
    A = (rs.rand(m, k) < 0.25).astype(int)
    B = (rs.rand(n, k) < 0.25).astype(int)
    Xfull = np.minimum(A @ B.T, 1)
    part = rs.rand(m, n)
    sets = {}
    out = {"shape": np.array([m, n])}
    for name, lo, hi in (("train", 0.0, 0.30), ("val", 0.30, 0.38), ("test", 0.38, 0.47)):
        r, c = np.nonzero((part >= lo) & (part < hi))
        sets[name] = csr_matrix((Xfull[r, c].astype(np.float64), (r, c)), shape=(m, n))  # ones and explicit zeros
        assert sets[name].nnz == r.size
        out.update({name + "_rows": r.astype(np.int32), name + "_cols": c.astype(np.int32), name + "_vals": Xfull[r, c].astype(np.uint8)})
    meta = {}

    def staged(model, task):
        kw = dict(FIT_KW)
        kw["task"] = task
        model.check_params(**kw)
        model.load_dataset(X_train=sets["train"].copy(), X_val=sets["val"].copy(), X_test=sets["test"].copy())
        model.init_model()
        return model.U.copy(), model.V.copy()

    with quiet():
        mdl = BinaryMFPenalty(k=k, W="mask", reg=1.0, reg_growth=1.3, init_method="normal", normalize_method="balance",
                              max_iter=6, seed=8)
        U0, V0 = staged(mdl, "prediction")
        mdl._fit()
    out.update(p_U0=U0, p_V0=V0, p_U=mdl.U, p_V=mdl.V)
    meta["penalty_prediction"] = {"updates": df_rows(mdl.logs["updates"]), "boolean": df_rows(mdl.logs["boolean"])}
    with quiet():
        w = WNMF(k=k, W="mask", init_method="normal", max_iter=6, seed=8)
        U0, V0 = staged(w, "prediction")
        w._fit()
    out.update(w_U0=U0, w_V0=V0, w_U=w.U, w_V=w.V)
    meta["wnmf_prediction"] = {"updates": df_rows(w.logs["updates"])}
    with quiet():
        t = BinaryMFThreshold(k=k, U=w.U.copy(), V=w.V.copy(), u=0.3, v=0.3, lamda=10, min_diff=1e-3, max_iter=30)
        staged(t, "prediction")
        t._fit()
    meta["threshold_prediction"] = {"updates": df_rows(t.logs["updates"]), "u": float(t.u), "v": float(t.v)}
    with quiet():
        mdl = BinaryMFPenalty(k=k, W="full", reg=1.0, reg_growth=1.3, init_method="normal", normalize_method="balance",
                              max_iter=5, seed=8)
        U0, V0 = staged(mdl, "reconstruction")
        mdl._fit()
    out.update(r_U0=U0, r_V0=V0, r_U=mdl.U, r_V=mdl.V)
    meta["penalty_reconstruction"] = {"updates": df_rows(mdl.logs["updates"]), "boolean": df_rows(mdl.logs["boolean"])}
    np.savez_compressed(os.path.join(HERE, "g9_prediction.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "g9_prediction.json"), "w"), indent=1)


def g10_link_models(PyBMF):
    """PNLPF (sigmoid link on the product) and WNMF with the Kullback-Leibler loss, W='full', small planted Boolean X."""
    from PyBMF.models import PNLPF, WNMF
    rs = np.random.RandomState(31)
    m, n, k = 210, 150, 6
    A = (rs.rand(m, k) < 0.25).astype(int)
    B = (rs.rand(n, k) < 0.25).astype(int)
    X = np.minimum(A @ B.T, 1).astype(np.float64)
    flip = rs.rand(m, n) < 0.03
    X[flip] = 1 - X[flip]
    out = {"X": np.packbits(X.astype(np.uint8), axis=1), "shape": np.array([m, n])}
    meta = {}
    with quiet():
        p = PNLPF(k=k, W="full", reg=1.0, reg_growth=1.2, link_lamda=10, init_method="normal", normalize_method="balance",
                  max_iter=9, seed=5)
        U0, V0 = staged_fit(p, X.copy())
        p._fit()
    out.update(p_U0=U0, p_V0=V0, p_U=p.U, p_V=p.V)
    meta["pnlpf"] = {"updates": df_rows(p.logs["updates"]), "boolean": df_rows(p.logs["boolean"]), "final_reg": float(p.reg),
                     "params": {"k": k, "reg": 1.0, "reg_growth": 1.2, "link_lamda": 10, "max_iter": 9}}
    with quiet():
        w = WNMF(k=k, W="full", beta_loss="kullback-leibler", init_method="normal", max_iter=9, seed=5)
        U0, V0 = staged_fit(w, X.copy())
        w._fit()
    out.update(w_U0=U0, w_V0=V0, w_U=w.U, w_V=w.V)
    meta["wnmf_kl"] = {"updates": df_rows(w.logs["updates"])}
    np.savez_compressed(os.path.join(HERE, "g10_link_models.npz"), **out)
    with open(os.path.join(HERE, "g10_link_models.json"), "w") as f:
        json.dump(meta, f, indent=1)


def g11_cover_scores(PyBMF):
    """Confusion counts with axis, coverage_score / weighted_error / description_length (utils/metrics.py:56-201)."""
    from PyBMF.utils import TP, FP, TN, FN, ACC, coverage_score, weighted_error, description_length, to_sparse
    rs = np.random.RandomState(9)
    cases = []
    for (m, n, k, pg) in [(37, 70, 4, 0.3), (130, 45, 9, 0.15), (64, 64, 33, 0.5), (5, 200, 2, 0.05)]:
        gt = (rs.rand(m, n) < pg).astype(np.int64)
        U = (rs.rand(m, k) < 0.25).astype(np.int64)
        V = (rs.rand(n, k) < 0.25).astype(np.int64)
        pd = np.minimum(U @ V.T, 1)
        G, P = to_sparse(gt, "csr"), to_sparse(pd, "csr")
        c = {"shape": [m, n, k], "gt": np.packbits(gt.astype(np.uint8), axis=1).tolist(), "U": U.tolist(), "V": V.tolist()}
        for ax in (None, 0, 1):
            key = "all" if ax is None else f"axis{ax}"
            c[key] = {nm: np.asarray(fn(G, P, axis=ax)).astype(float).tolist() for nm, fn in (("TP", TP), ("FP", FP), ("TN", TN), ("FN", FN), ("ACC", ACC))}
            c[key]["coverage_score_0.5"] = np.asarray(coverage_score(G, P, axis=ax)).astype(float).tolist()
            c[key]["coverage_score_0.3"] = np.asarray(coverage_score(G, P, w_fp=0.3, axis=ax)).astype(float).tolist()
            c[key]["weighted_error_0.2_0.7"] = np.asarray(weighted_error(G, P, w_fp=0.2, w_fn=0.7, axis=ax)).astype(float).tolist()
        c["description_length"] = float(description_length(G, to_sparse(U, "csr"), to_sparse(V, "csr")))
        c["description_length_w"] = float(description_length(G, to_sparse(U, "csr"), to_sparse(V, "csr"), pd=P, w_model=0.5, w_fp=2.0, w_fn=3.0))
        cases.append(c)
    with open(os.path.join(HERE, "g11_cover_scores.json"), "w") as f:
        json.dump(cases, f)


def g12_normalize(PyBMF):
    """normalize_UV for every normalize_method (models/ContinuousModel.py:87-148, unique_values_mapping :225-231) through
    BinaryMFThreshold's init_model: factors after normalisation and the thresholds the line search ends on."""
    from PyBMF.models import BinaryMFThreshold
    z = np.load(os.path.join(HERE, "g4_threshold.npz"))
    m, n = z["shape"]
    X = np.unpackbits(z["X_bits"], axis=1, bitorder="little")[:, :n].astype(np.float64)[:120, :90]
    # two decimals: repeated values, so that the unique-value mappings are not plain rank transforms
    U, V = np.round(z["U"][:120, :6], 2), np.round(z["V"][:90, :6], 2)
    out, meta = {"X": X.astype(np.uint8), "U0": U, "V0": V}, {}
    for method in ("balance", "matrixwise-normalize", "columnwise-normalize", "matrixwise-mapping", "columnwise-mapping"):
        with quiet():
            mdl = BinaryMFThreshold(k=6, U=U.copy(), V=V.copy(), W="full", u=0.4, v=0.4, lamda=10, min_diff=1e-3, max_iter=8,
                                    normalize_method=method)
            staged_fit(mdl, X.copy())
            out[f"U_{method}"], out[f"V_{method}"] = np.asarray(mdl.U, dtype=np.float64), np.asarray(mdl.V, dtype=np.float64)
            mdl._fit()
        meta[method] = {"rows": df_rows(mdl.logs["updates"]), "u": float(mdl.u), "v": float(mdl.v)}
    np.savez_compressed(os.path.join(HERE, "g12_normalize.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "g12_normalize.json"), "w"), indent=1)


def g13_kl_mask(PyBMF):
    """WNMF, Kullback-Leibler loss with the reference's DEFAULT mask W='mask': on a dense Boolean matrix (pattern = its
    non-zeros) and on a csr with explicit zeros (pattern = stored cells).  X and the initial factors are those of g10."""
    from scipy.sparse import csr_matrix
    from PyBMF.models import WNMF
    z = np.load(os.path.join(HERE, "g10_link_models.npz"))
    m, n = z["shape"]
    X = np.unpackbits(z["X"], axis=1)[:, :n].astype(np.float64)
    # the reference divides 0 / 0 on an all-zero row or column under W='mask' (its zero cells are masked out before the
    # in-place eps can help) and stops with "NaN is found in prediction": give every row and column a one
    for i in np.flatnonzero(X.sum(1) == 0):
        X[i, i % n] = 1.0
    for j in np.flatnonzero(X.sum(0) == 0):
        X[j % m, j] = 1.0
    rs = np.random.RandomState(77)
    keep = rs.rand(m, n) < 0.6          # observed cells of the csr case; every non-zero stays observed
    keep |= X != 0
    r, c = np.nonzero(keep)
    Xs = csr_matrix((X[r, c], (r, c)), shape=(m, n))
    assert Xs.nnz == len(r)
    out, meta = {"rows": r.astype(np.int32), "cols": c.astype(np.int32), "X": np.packbits(X.astype(np.uint8), axis=1)}, {}
    for tag, data in (("dense", X), ("csr", Xs)):
        with quiet():
            w = WNMF(k=6, U=z["w_U0"].copy(), V=z["w_V0"].copy(), W="mask", beta_loss="kullback-leibler", init_method="custom", max_iter=6)
            staged_fit(w, data.copy())
            w._fit()
        out[tag + "_U"], out[tag + "_V"] = w.U, w.V
        meta[tag] = {"updates": df_rows(w.logs["updates"])}
    np.savez_compressed(os.path.join(HERE, "g13_kl_mask.npz"), **out)
    with open(os.path.join(HERE, "g13_kl_mask.json"), "w") as f:
        json.dump(meta, f, indent=1)


def g14_palm(PyBMF):
    """ELBMF (models/ELBMF.py:110-210) and PRIMP (models/PRIMP.py:51-160).  Both classes fail as shipped
    (ELBMF.init_model: normalize_UV(method=...) -> TypeError; PRIMP._fit: .toarray() on an ndarray -> AttributeError; recorded
    in the json), and both modules are commented out of PyBMF/models/__init__.py, so they are imported by module path and the
    module-level functions are driven from fixed initial factors (no dependence on torch.rand except the one primp() call):
      * ELBMF.prox / get_integrality_gap on a grid, update_U single steps (beta = 0 and beta > 0), and the class's own iPALM
        loop, reached by doing init_model's steps by hand without the crashing normalize_UV call;
      * PRIMP.elbmf_step_ipalm single steps, elbmf_ipalm runs (fp64 and fp32 tensors), the rounding of primp(), primp(seed)."""
    import importlib
    import torch
    E = importlib.import_module("PyBMF.models.ELBMF")
    P = importlib.import_module("PyBMF.models.PRIMP")
    from PyBMF.models.ContinuousModel import ContinuousModel
    rs = np.random.RandomState(5)
    m, n, k = 150, 100, 6
    A, B = rs.rand(m, k) < 0.15, rs.rand(n, k) < 0.15
    X = ((A.astype(int) @ B.T.astype(int)) > 0).astype(np.float64)
    X = np.where(rs.rand(m, n) < 0.02, 1 - X, X)
    U0, V0 = rs.rand(m, k), rs.rand(n, k)
    W = np.ones((m, n))
    out = {"X": np.packbits(X.astype(np.uint8), axis=1), "shape": np.array([m, n, k]), "U0": U0, "V0": V0}
    meta = {"crashes_as_shipped": {}}
    for name, make in (("ELBMF", lambda: E.ELBMF(k=k, U=U0.copy(), V=V0.copy(), W="full", init_method="custom", max_iter=3)),
                       ("PRIMP", lambda: P.PRIMP(k=k, max_iter=3, seed=3))):
        try:
            with quiet():
                make().fit(X.copy(), **FIT_KW)
            meta["crashes_as_shipped"][name] = None
        except Exception as e:  # noqa: BLE001
            meta["crashes_as_shipped"][name] = f"{type(e).__name__}: {e}"
    # ---- ELBMF element-wise pieces
    grid = np.concatenate([np.linspace(-0.3, 1.6, 39), [0.0, 0.5, 1.0, 0.5 + 1e-12, 0.5 - 1e-12]]).reshape(-1, 4)
    out["prox_in"] = grid
    for i, (kai, lam) in enumerate(((0.0, 0.0), (0.01, 0.0), (0.02, 0.3), (0.4, 2.0))):
        out[f"prox_out_{i}"] = E.prox(grid.copy(), kai, lam)
    meta["prox_params"] = [(0.0, 0.0), (0.01, 0.0), (0.02, 0.3), (0.4, 2.0)]
    meta["gap"] = [{"reg_l1": a, "reg_l2": b, "value": float(E.get_integrality_gap(grid, a, b))} for a, b in ((0.01, 0.02), (0.3, 1.7))]
    # ---- ELBMF single steps
    U_prev = U0 + 0.05 * rs.standard_normal((m, k))
    out["U_prev"] = U_prev
    steps = []
    for i, (l1, l2, beta) in enumerate(((0.01, 0.02, 0.0), (0.01, 0.5, 0.0), (0.05, 0.02, 0.1), (0.0, 0.0, 0.3))):
        Un, Ul = E.update_U(X, U0.copy(), V0.copy(), W, l1, l2, beta, U_prev.copy())
        assert np.array_equal(Ul, U0)
        Vn, _ = E.update_U(X.T, V0.copy(), Un, W.T, l1, l2, beta, V0.copy())
        out[f"step{i}_U"], out[f"step{i}_V"] = Un, Vn
        steps.append({"reg_l1": l1, "reg_l2": l2, "beta": beta})
    meta["steps"] = steps
    # ---- the class's own loop (iPALM), beta = 0 and beta > 0
    for tag, beta, iters in (("palm", 0.0, 12), ("ipalm", 0.2, 12)):
        mdl = E.ELBMF(k=k, U=U0.copy(), V=V0.copy(), W="full", init_method="custom", reg_l1=0.01, reg_l2=0.02, reg_growth=1.05,
                      beta=beta, max_iter=iters, min_diff=1e-8, tol=0.0)
        with quiet():
            mdl.check_params(**FIT_KW)
            mdl.load_dataset(X_train=X.copy(), X_val=None, X_test=None)
            ContinuousModel.init_model(mdl)
            mdl.init_UV()
            mdl._to_dense()
            mdl.U[mdl.U == 0] = np.finfo(float).eps
            mdl.V[mdl.V == 0] = np.finfo(float).eps
            mdl.iPALM()
        out[f"{tag}_U"], out[f"{tag}_V"] = np.asarray(mdl.U), np.asarray(mdl.V)
        meta[tag] = {"beta": beta, "max_iter": iters, "updates": df_rows(mdl.logs["updates"]),
                     "counts": counts_of(PyBMF, mdl.X_train if hasattr(mdl.X_train, "tocsr") else __import__("scipy.sparse").sparse.csr_matrix(mdl.X_train), mdl.X_pd)}
    # ---- PRIMP
    Xt = torch.from_numpy(X.copy())
    Ua = torch.from_numpy(U_prev.copy())
    psteps = []
    for i, (l1, l2, tau, beta) in enumerate(((0.01, 0.0, 1.0, 0.0), (0.05, 0.0, 1.3, 1e-4), (0.02, 0.1, 2.0, 0.2))):
        Un = P.elbmf_step_ipalm(Xt, torch.from_numpy(U0.copy()), torch.from_numpy(V0.T.copy()), Ua.clone(), l1, l2, tau, beta)
        out[f"pstep{i}_U"] = Un.numpy()
        psteps.append({"l1reg": l1, "l2reg": l2, "tau": tau, "beta": beta})
    meta["primp_steps"] = psteps
    for tag, dt, beta in (("primp64", torch.float64, 1e-4), ("primp32", torch.float32, 1e-4), ("primp64_b0", torch.float64, 0.0)):
        with quiet():
            U, Vt = P.elbmf_ipalm(Xt.to(dt), torch.from_numpy(U0.copy()).to(dt), torch.from_numpy(V0.T.copy()).to(dt), 0.01, 0,
                                  lambda t: 1.02 ** t, 25, 1e-8, beta, None)
        out[f"{tag}_U"], out[f"{tag}_Vt"] = U.numpy(), Vt.numpy()
        Ur, Vr = P.proxelbmfnn(U, 0.5, 0 * 1e12).round(), P.proxelbmfnn(Vt, 0.5, 0 * 1e12).round()
        out[f"{tag}_Ur"], out[f"{tag}_Vtr"] = Ur.numpy().astype(np.uint8), Vr.numpy().astype(np.uint8)
        meta[tag] = {"beta": beta, "maxiter": 25, "fn_final": float((Xt.to(dt) - U @ Vt).norm() ** 2)}
    with quiet():
        Ur, Vr = P.primp(Xt.float(), k, l1reg=0.01, maxiter=25, beta=1e-4, seed=3)
    torch.manual_seed(3)
    out["primp_seed3_U0"], out["primp_seed3_Vt0"] = torch.rand(m, k).numpy(), torch.rand(k, n).numpy()
    out["primp_seed3_Ur"], out["primp_seed3_Vtr"] = Ur.numpy().astype(np.uint8), Vr.numpy().astype(np.uint8)
    meta["torch_version"] = torch.__version__
    np.savez_compressed(os.path.join(HERE, "g14_palm.npz"), **out)
    with open(os.path.join(HERE, "g14_palm.json"), "w") as f:
        json.dump(meta, f, indent=1)


def g15_elbmf_masked(PyBMF):
    """ELBMF under a mask / weight matrix (PyBMF/models/ELBMF.py:177-196 with W != all ones): single steps of the module-level
    update_U with a 0/1 mask and with real weights, beta = 0 and > 0, and the class's own iPALM loop with W = a 0/1 matrix
    (init_model's working steps done by hand, as in g14: the class does not run as shipped)."""
    import importlib
    E = importlib.import_module("PyBMF.models.ELBMF")
    from PyBMF.models.ContinuousModel import ContinuousModel
    rs = np.random.RandomState(15)
    m, n, k = 140, 90, 5
    A, B = rs.rand(m, k) < 0.18, rs.rand(n, k) < 0.18
    X = ((A.astype(int) @ B.T.astype(int)) > 0).astype(np.float64)
    X = np.where(rs.rand(m, n) < 0.02, 1 - X, X)
    U0, V0 = rs.rand(m, k) * 0.6, rs.rand(n, k) * 0.6
    W01 = (rs.rand(m, n) < 0.6).astype(np.float64)
    W01[7, :] = 0.0        # an unobserved row and column
    W01[:, 11] = 0.0
    Wr = W01 * rs.choice([0.5, 1.0, 2.0], size=(m, n))
    U_prev = U0 + 0.05 * rs.standard_normal((m, k))
    out = {"X": np.packbits(X.astype(np.uint8), axis=1), "shape": np.array([m, n, k]), "U0": U0, "V0": V0, "W01": np.packbits(W01.astype(np.uint8), axis=1),
           "Wr": Wr, "U_prev": U_prev}
    steps = []
    for i, (Wm, l1, l2, beta) in enumerate(((W01, 0.01, 0.02, 0.0), (W01, 0.02, 0.3, 0.2), (Wr, 0.01, 0.02, 0.0), (Wr, 0.0, 0.1, 0.15))):
        Un, Ul = E.update_U(X, U0.copy(), V0.copy(), Wm, l1, l2, beta, U_prev.copy())
        Vn, _ = E.update_U(X.T, V0.copy(), U0.copy(), Wm.T, l1, l2, beta, V0.copy())
        out[f"mstep{i}_U"], out[f"mstep{i}_V"] = Un, Vn
        steps.append({"W": "W01" if Wm is W01 else "Wr", "reg_l1": l1, "reg_l2": l2, "beta": beta})
    meta = {"steps": steps}
    for tag, beta, iters in (("mpalm", 0.0, 10), ("mipalm", 0.2, 10)):
        # (the reference's init_W cannot take a matrix W under this NumPy / SciPy -- `self.W in ['mask', 'full']` raises for arrays --
        # so the class runs with W='mask' on a csr whose STORED entries, explicit zeros included, are the observed cells)
        from scipy.sparse import csr_matrix
        r, c = np.nonzero(W01)
        Xs = csr_matrix((X[r, c], (r, c)), shape=X.shape)
        mdl = E.ELBMF(k=k, U=U0.copy(), V=V0.copy(), W="mask", init_method="custom", reg_l1=0.01, reg_l2=0.02, reg_growth=1.05,
                      beta=beta, max_iter=iters, min_diff=1e-8, tol=0.0)
        with quiet():
            mdl.check_params(**FIT_KW)
            mdl.load_dataset(X_train=Xs.copy(), X_val=None, X_test=None)
            ContinuousModel.init_model(mdl)
            mdl.init_UV()
            mdl._to_dense()
            mdl.U[mdl.U == 0] = np.finfo(float).eps
            mdl.V[mdl.V == 0] = np.finfo(float).eps
            mdl.iPALM()
        out[f"{tag}_U"], out[f"{tag}_V"] = np.asarray(mdl.U), np.asarray(mdl.V)
        meta[tag] = {"beta": beta, "max_iter": iters, "updates": df_rows(mdl.logs["updates"]),
                     "counts": counts_of(PyBMF, mdl.X_train if hasattr(mdl.X_train, "tocsr") else __import__("scipy.sparse").sparse.csr_matrix(mdl.X_train), mdl.X_pd)}
    np.savez_compressed(os.path.join(HERE, "g15_elbmf_masked.npz"), **out)
    with open(os.path.join(HERE, "g15_elbmf_masked.json"), "w") as f:
        json.dump(meta, f, indent=1)


def g16_pnlpf_masked(PyBMF):
    """PNLPF under a mask (PyBMF/models/PNLPF.py:61-91 with W != all ones): the class with W='mask' on a csr whose stored entries
    (ones and explicit zeros) are the observed cells, and the module-level update_U / update_V with a real weight matrix."""
    from scipy.sparse import csr_matrix
    from PyBMF.models import PNLPF
    mod = sys.modules["PyBMF.models.PNLPF"]
    rs = np.random.RandomState(23)
    m, n, k = 160, 120, 6
    A = (rs.rand(m, k) < 0.25).astype(int)
    B = (rs.rand(n, k) < 0.25).astype(int)
    Xfull = np.minimum(A @ B.T, 1)
    obs = rs.rand(m, n) < 0.35
    obs[3, :] = False
    obs[:, 8] = False
    r, c = np.nonzero(obs)
    X = csr_matrix((Xfull[r, c].astype(np.float64), (r, c)), shape=(m, n))
    out = {"rows": r.astype(np.int32), "cols": c.astype(np.int32), "vals": Xfull[r, c].astype(np.uint8), "shape": np.array([m, n])}
    meta = {}
    with quiet():
        p = PNLPF(k=k, W="mask", reg=1.0, reg_growth=1.2, link_lamda=10, init_method="normal", normalize_method="balance", max_iter=7, seed=6)
        U0, V0 = staged_fit(p, X.copy())
        p._fit()
    out.update(p_U0=U0, p_V0=V0, p_U=p.U, p_V=p.V)
    meta["pnlpf"] = {"updates": df_rows(p.logs["updates"]), "boolean": df_rows(p.logs["boolean"]), "final_reg": float(p.reg),
                     "params": {"k": k, "reg": 1.0, "reg_growth": 1.2, "link_lamda": 10, "max_iter": 7}}
    # module-level steps with real weights on the dense matrix
    Xd = Xfull.astype(np.float64)
    Wr = obs * rs.choice([0.5, 1.0, 2.0], size=(m, n))
    U1, V1 = np.abs(rs.standard_normal((m, k))) * 0.4 + 1e-3, np.abs(rs.standard_normal((n, k))) * 0.4 + 1e-3
    out.update(Xd=np.packbits(Xd.astype(np.uint8), axis=1), Wr=Wr, s_U=U1, s_V=V1)
    for i, (reg, lam) in enumerate(((0.0, 10.0), (1.5, 10.0), (0.7, 4.0))):
        Vn = mod.update_V(X=Xd, W=Wr, U=U1, V=V1, reg=np.float64(reg), link_lamda=lam)
        Un = mod.update_U(X=Xd, W=Wr, U=U1, V=Vn, reg=np.float64(reg), link_lamda=lam)
        out[f"step{i}_V"], out[f"step{i}_U"] = Vn, Un
    meta["steps"] = [{"reg": 0.0, "link_lamda": 10.0}, {"reg": 1.5, "link_lamda": 10.0}, {"reg": 0.7, "link_lamda": 4.0}]
    np.savez_compressed(os.path.join(HERE, "g16_pnlpf_masked.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "g16_pnlpf_masked.json"), "w"), indent=1)


def g17_val_test_sets(PyBMF):
    """fit(X_train, X_val, X_test) for the models of SURVEY 8f: PNLPF (the inherited loop scores every set per iteration,
    BinaryMFPenalty.py:71,97 -> BaseModel.evaluate :209-257, RMSE / MAE against the sigmoid-link prediction) under task='prediction'
    (W='mask') and task='reconstruction' (W='full'); ELBMF's iPALM loop (ELBMF.py:143: ERR, Accuracy, Recall, Precision, F1 per set)
    under both tasks, init_model's working steps done by hand as in g14 / g15 (the class does not run as shipped).
    Data: the train / val / test split of g9 (csr matrices whose stored entries are ones AND explicit zeros)."""
    import importlib
    from scipy.sparse import csr_matrix
    from PyBMF.models import PNLPF
    from PyBMF.models.ContinuousModel import ContinuousModel
    E = importlib.import_module("PyBMF.models.ELBMF")
    z = np.load(os.path.join(HERE, "g9_prediction.npz"))
    m, n = (int(v) for v in z["shape"])
    sets = {nm: csr_matrix((z[nm + "_vals"].astype(np.float64), (z[nm + "_rows"], z[nm + "_cols"])), shape=(m, n)) for nm in ("train", "val", "test")}
    for nm in sets:
        assert sets[nm].nnz == len(z[nm + "_rows"])
    k = 5
    out, meta = {}, {}
    for task, W in (("prediction", "mask"), ("reconstruction", "full")):
        kw = dict(FIT_KW)
        kw["task"] = task
        with quiet():
            p = PNLPF(k=k, W=W, reg=1.0, reg_growth=1.2, link_lamda=10, init_method="normal", normalize_method="balance", max_iter=6, seed=8)
            p.check_params(**kw)
            p.load_dataset(X_train=sets["train"].copy(), X_val=sets["val"].copy(), X_test=sets["test"].copy())
            p.init_model()
            U0, V0 = p.U.copy(), p.V.copy()
            p._fit()
        out.update({f"pnlpf_{task}_U0": U0, f"pnlpf_{task}_V0": V0, f"pnlpf_{task}_U": p.U, f"pnlpf_{task}_V": p.V})
        meta[f"pnlpf_{task}"] = {"updates": df_rows(p.logs["updates"]), "boolean": df_rows(p.logs["boolean"]), "W": W,
                                 "params": {"k": k, "reg": 1.0, "reg_growth": 1.2, "link_lamda": 10, "max_iter": 6, "seed": 8}}
    rs = np.random.RandomState(31)
    EU0, EV0 = rs.rand(m, k) * 0.6, rs.rand(n, k) * 0.6
    out.update(elbmf_U0=EU0, elbmf_V0=EV0)
    for task, W in (("prediction", "mask"), ("reconstruction", "full")):
        kw = dict(FIT_KW)
        kw["task"] = task
        mdl = E.ELBMF(k=k, U=EU0.copy(), V=EV0.copy(), W=W, init_method="custom", reg_l1=0.01, reg_l2=0.02, reg_growth=1.05, beta=0.0,
                      max_iter=8, min_diff=1e-8, tol=0.0)
        with quiet():
            mdl.check_params(**kw)
            mdl.load_dataset(X_train=sets["train"].copy(), X_val=sets["val"].copy(), X_test=sets["test"].copy())
            ContinuousModel.init_model(mdl)
            mdl.init_UV()
            mdl._to_dense()
            mdl.U[mdl.U == 0] = np.finfo(float).eps
            mdl.V[mdl.V == 0] = np.finfo(float).eps
            mdl.iPALM()
        out.update({f"elbmf_{task}_U": np.asarray(mdl.U), f"elbmf_{task}_V": np.asarray(mdl.V)})
        meta[f"elbmf_{task}"] = {"updates": df_rows(mdl.logs["updates"]), "W": W,
                                 "params": {"k": k, "reg_l1": 0.01, "reg_l2": 0.02, "reg_growth": 1.05, "beta": 0.0, "max_iter": 8, "min_diff": 1e-8}}
    np.savez_compressed(os.path.join(HERE, "g17_val_test_sets.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "g17_val_test_sets.json"), "w"), indent=1)


def g18_kl_weights(PyBMF):
    """WNMF, Kullback-Leibler loss under a REAL weight matrix (PyBMF/models/WNMF.py:111-129, error :143-145).  The reference's
    init_W cannot take a matrix under this NumPy / SciPy (`self.W in ['mask', 'full']` raises for arrays), so the model is staged
    with W='full' and W is replaced by the weight matrix before the loop runs.  X, U0, V0: those of g10."""
    from PyBMF.models import WNMF
    z = np.load(os.path.join(HERE, "g10_link_models.npz"))
    m, n = (int(v) for v in z["shape"])
    X = np.unpackbits(z["X"], axis=1)[:, :n].astype(np.float64)
    rs = np.random.RandomState(18)
    obs = rs.rand(m, n) < 0.7
    obs |= X != 0
    Wr = obs * rs.choice([0.5, 1.0, 2.0], size=(m, n))
    with quiet():
        w = WNMF(k=6, U=z["w_U0"].copy(), V=z["w_V0"].copy(), W="full", beta_loss="kullback-leibler", init_method="custom", max_iter=6)
        staged_fit(w, X.copy())
        w.W = Wr.copy()
        w._fit()
    np.savez_compressed(os.path.join(HERE, "g18_kl_weights.npz"), Wr=Wr, U=np.asarray(w.U), V=np.asarray(w.V))
    json.dump({"updates": df_rows(w.logs["updates"])}, open(os.path.join(HERE, "g18_kl_weights.json"), "w"), indent=1)


def g19_real_valued(PyBMF):
    """Real-valued (not 0 / 1) training data on the models whose GPU path was Boolean-only until round 5: the reference casts whatever
    it is given to float64 and runs (PyBMF/models/ContinuousModel.py:188-203).  BinaryMFPenalty under W='full', under W='mask' on a
    csr and under a weight matrix; PNLPF under W='full'; WNMF with the Kullback-Leibler loss under W='full' (models/WNMF.py:111-129);
    BinaryMFThreshold under W='full' from the penalty fit's factors.  Two data sets: values in [0, 1], and values up to ~3 (where
    the reference's arithmetic "confusion" metrics, utils/metrics.py:56-77 on csr matrices, clamp differently from counts)."""
    from scipy.sparse import csr_matrix
    from PyBMF.models import BinaryMFPenalty, BinaryMFThreshold, PNLPF, WNMF
    rs = np.random.RandomState(19)
    m, n, k = 70, 50, 5
    A = rs.rand(m, k) * (rs.rand(m, k) < 0.45)
    B = rs.rand(n, k) * (rs.rand(n, k) < 0.45)
    X01 = np.minimum(A @ B.T, 1.0)
    X01[X01 < 0.05] = 0.0
    X3 = 3.0 * (A @ B.T)
    X3[X3 < 0.15] = 0.0
    out, meta = {"X01": X01, "X3": X3}, {}
    pen = dict(k=k, reg=1.0, reg_growth=1.1, init_method="normal", normalize_method="balance", max_iter=8, seed=7)
    for tag, X in (("x01", X01), ("x3", X3)):
        with quiet():
            p = BinaryMFPenalty(W="full", **pen)
            U0, V0 = staged_fit(p, X.copy())
            p._fit()
        out.update({f"pen_{tag}_U0": U0, f"pen_{tag}_V0": V0, f"pen_{tag}_U": p.U, f"pen_{tag}_V": p.V})
        meta[f"pen_{tag}"] = {"updates": df_rows(p.logs["updates"]), "boolean": df_rows(p.logs["boolean"]), "final_reg": float(p.reg)}
        with quiet():
            q = PNLPF(W="full", link_lamda=10, **pen)
            U0, V0 = staged_fit(q, X.copy())
            q._fit()
        out.update({f"pnlpf_{tag}_U0": U0, f"pnlpf_{tag}_V0": V0, f"pnlpf_{tag}_U": q.U, f"pnlpf_{tag}_V": q.V})
        meta[f"pnlpf_{tag}"] = {"updates": df_rows(q.logs["updates"]), "boolean": df_rows(q.logs["boolean"])}
        with quiet():
            w = WNMF(k=k, W="full", beta_loss="kullback-leibler", init_method="normal", max_iter=8, seed=7)
            U0, V0 = staged_fit(w, X.copy())
            w._fit()
        out.update({f"kl_{tag}_U0": U0, f"kl_{tag}_V0": V0, f"kl_{tag}_U": np.asarray(w.U), f"kl_{tag}_V": np.asarray(w.V)})
        meta[f"kl_{tag}"] = {"updates": df_rows(w.logs["updates"])}
        with quiet():
            t = BinaryMFThreshold(k=k, U=p.U.copy(), V=p.V.copy(), W="full", u=0.4, v=0.4, lamda=10, min_diff=1e-3, max_iter=6)
            staged_fit(t, X.copy())
            F0, dF0 = float(t.F([0.4, 0.4])), np.asarray(t.dF([0.4, 0.4]), dtype=np.float64)
            t._fit()
        out[f"thr_{tag}_dF0"] = dF0
        meta[f"thr_{tag}"] = {"rows": df_rows(t.logs["updates"]), "u": float(t.u), "v": float(t.v), "F0": F0}
    # masks on the [0, 1] data: the stored pattern of a csr (zeros included), and a weight matrix
    obs = rs.rand(m, n) < 0.5
    obs[2, :] = False
    obs[:, 5] = False
    r, c = np.nonzero(obs)
    Xm = csr_matrix((X01[r, c], (r, c)), shape=(m, n))
    out.update(mask_rows=r.astype(np.int32), mask_cols=c.astype(np.int32))
    with quiet():
        p = BinaryMFPenalty(W="mask", **pen)
        U0, V0 = staged_fit(p, Xm.copy())
        p._fit()
    out.update(pen_mask_U0=U0, pen_mask_V0=V0, pen_mask_U=p.U, pen_mask_V=p.V)
    meta["pen_mask"] = {"updates": df_rows(p.logs["updates"]), "boolean": df_rows(p.logs["boolean"])}
    Wr = obs * rs.choice([0.5, 1.0, 2.0], size=(m, n))
    with quiet():
        p = BinaryMFPenalty(W="full", **pen)
        U0, V0 = staged_fit(p, X01.copy())
        p.W = Wr.copy()
        p._fit()
    out.update(Wr=Wr, pen_wgt_U0=U0, pen_wgt_V0=V0, pen_wgt_U=p.U, pen_wgt_V=p.V)
    meta["pen_wgt"] = {"updates": df_rows(p.logs["updates"]), "boolean": df_rows(p.logs["boolean"])}
    meta["params"] = {"penalty": {kk: vv for kk, vv in pen.items()}, "link_lamda": 10, "threshold": {"u": 0.4, "v": 0.4, "lamda": 10, "min_diff": 1e-3, "max_iter": 6}}
    np.savez_compressed(os.path.join(HERE, "g19_real_valued.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "g19_real_valued.json"), "w"), indent=1)


def g20_wide_rank(PyBMF):
    """Rank 64 < k <= 128 (the reference has no rank limit, BinaryMFPenalty.py:32) beyond the all-ones mask on the training matrix
    alone: W='mask' on a csr with unstored cells, and X_val / X_test under both tasks.  k = 72 (a second block of 8 columns on the
    GPU side).  Data: a planted Boolean matrix split into train / val / test by cell (stored entries are ones AND explicit zeros)."""
    from scipy.sparse import csr_matrix
    from PyBMF.models import BinaryMFPenalty, WNMF
    rs = np.random.RandomState(41)
    m, n, kt, k = 130, 100, 9, 72
    A = (rs.rand(m, kt) < 0.2).astype(int)
    B = (rs.rand(n, kt) < 0.2).astype(int)
    Xfull = np.minimum(A @ B.T, 1)
    part = rs.rand(m, n)
    sets = {}
    out = {"shape": np.array([m, n])}
    for name, lo, hi in (("train", 0.0, 0.35), ("val", 0.35, 0.43), ("test", 0.43, 0.52)):
        r, c = np.nonzero((part >= lo) & (part < hi))
        sets[name] = csr_matrix((Xfull[r, c].astype(np.float64), (r, c)), shape=(m, n))
        assert sets[name].nnz == r.size
        out.update({name + "_rows": r.astype(np.int32), name + "_cols": c.astype(np.int32), name + "_vals": Xfull[r, c].astype(np.uint8)})
    meta = {"k": k}

    def staged(model, task, with_sets=True):
        kw = dict(FIT_KW)
        kw["task"] = task
        model.check_params(**kw)
        model.load_dataset(X_train=sets["train"].copy(), X_val=sets["val"].copy() if with_sets else None,
                           X_test=sets["test"].copy() if with_sets else None)
        model.init_model()
        return model.U.copy(), model.V.copy()

    pen = dict(k=k, reg=1.0, reg_growth=1.3, init_method="normal", normalize_method="balance", max_iter=5, seed=6)
    for name, W, task, with_sets in (("penalty_prediction", "mask", "prediction", True), ("penalty_reconstruction", "full", "reconstruction", True),
                                     ("penalty_mask_reconstruction", "mask", "reconstruction", False)):
        with quiet():
            mdl = BinaryMFPenalty(W=W, **pen)
            U0, V0 = staged(mdl, task, with_sets)
            mdl._fit()
        if "pen_U0" in out:   # (the initial factors depend on the seed and on X_train only)
            assert np.array_equal(out["pen_U0"], U0) and np.array_equal(out["pen_V0"], V0)
        out.update({"pen_U0": U0, "pen_V0": V0, name + "_U": mdl.U.astype(np.float32), name + "_V": mdl.V.astype(np.float32)})
        meta[name] = {"updates": df_rows(mdl.logs["updates"]), "boolean": df_rows(mdl.logs["boolean"]), "W": W, "task": task, "sets": with_sets,
                      "final_reg": float(mdl.reg)}
    for name, W, task, with_sets in (("wnmf_prediction", "mask", "prediction", True), ("wnmf_mask_reconstruction", "mask", "reconstruction", False)):
        with quiet():
            w = WNMF(k=k, W=W, init_method="normal", max_iter=5, seed=6)
            U0, V0 = staged(w, task, with_sets)
            w._fit()
        if "wnmf_U0" in out:
            assert np.array_equal(out["wnmf_U0"], U0) and np.array_equal(out["wnmf_V0"], V0)
        out.update({"wnmf_U0": U0, "wnmf_V0": V0, name + "_U": w.U.astype(np.float32), name + "_V": w.V.astype(np.float32)})
        meta[name] = {"updates": df_rows(w.logs["updates"]), "W": W, "task": task, "sets": with_sets}
    meta["params"] = {"penalty": pen, "wnmf": {"k": k, "init_method": "normal", "max_iter": 5, "seed": 6}}
    np.savez_compressed(os.path.join(HERE, "g20_wide_rank.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "g20_wide_rank.json"), "w"), indent=1)


if __name__ == "__main__":
    if os.environ.get("GOLDEN_ONLY") == "g20":
        g20_wide_rank(load_reference())
    elif os.environ.get("GOLDEN_ONLY") == "g19":
        g19_real_valued(load_reference())
    elif os.environ.get("GOLDEN_ONLY") == "g18":
        g18_kl_weights(load_reference())
    elif os.environ.get("GOLDEN_ONLY") == "g17":
        g17_val_test_sets(load_reference())
    elif os.environ.get("GOLDEN_ONLY") == "g16":
        g16_pnlpf_masked(load_reference())
    elif os.environ.get("GOLDEN_ONLY") == "g15":
        g15_elbmf_masked(load_reference())
    elif os.environ.get("GOLDEN_ONLY") == "g14":
        g14_palm(load_reference())
    elif os.environ.get("GOLDEN_ONLY") == "g13":
        g13_kl_mask(load_reference())
    elif os.environ.get("GOLDEN_ONLY") == "g12":
        g12_normalize(load_reference())
    elif os.environ.get("GOLDEN_ONLY") == "g11":
        g11_cover_scores(load_reference())
    elif os.environ.get("GOLDEN_ONLY") == "g10":
        g10_link_models(load_reference())
    elif os.environ.get("GOLDEN_ONLY") == "g9":
        g9_prediction(load_reference())
    elif os.environ.get("GOLDEN_ONLY") == "g7":
        g7_masked(load_reference())
    elif os.environ.get("GOLDEN_ONLY") == "g8":
        g8_threshold_masked(load_reference())
    else:
        main()
