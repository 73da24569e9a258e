"""Parameter / configuration plumbing, timer, naming, saving and the stopping rule shared by all models.

Same method names, keyword sets, attribute side effects and console messages as
``PyBMF/models/BaseModelTools.py`` (set_params :16-71, set_config :74-175, timer :178-214, _save_model :239-259,
early_stop :299-343), written for this package: nothing here touches the matrices.
"""
from __future__ import annotations

import pickle
import time

import numpy as np
import pandas as pd

from ..utils import _make_name, get_cache_path, ismat

CONFIG_KEYS = ("task", "seed", "display", "verbose", "scaling", "pixels", "show_logs", "save_model", "show_result")


def _say(key, value):
    print("[I] {:<12} : {}".format(key, value))


class BaseModelTools:
    def __init__(self):
        raise NotImplementedError("This is a helper class.")

    # ---- parameters -------------------------------------------------------------------------------------------
    def set_params(self, **kwargs):
        """Every keyword that is not a configuration key becomes an attribute (and is echoed)."""
        for name, value in kwargs.items():
            if name in CONFIG_KEYS:
                continue
            setattr(self, name, value)
            shown = len(value) if isinstance(value, list) else (value.shape if ismat(value) else value)
            _say(name, shown)

    def set_config(self, **kwargs):
        """task / seed / verbose / display / scaling / pixels / show_logs / save_model / show_result.

        Runs at construction and again at fit(**kwargs); `task` and `seed` only change when mentioned, the three
        finish() switches fall back to True whenever they are not mentioned in *this* call."""
        if "task" in kwargs:
            task = kwargs["task"]
            assert task in ["prediction", "reconstruction"], "Eval task must be 'prediction' or 'reconstruction'."
            self.task = task
            print("[I] task         :", self.task)
        if "seed" in kwargs:
            seed = kwargs["seed"]
            if seed is None and not hasattr(self, "seed"):
                seed = int(time.time())
            if seed is not None:
                self.seed = seed
                self.rng = np.random.RandomState(seed)
                print("[I] seed         :", self.seed)
        for flag in ("verbose", "display"):
            if not hasattr(self, flag):
                setattr(self, flag, False)
                _say(flag, False)
            if flag in kwargs and kwargs[flag] != getattr(self, flag):
                setattr(self, flag, kwargs[flag])
                _say(flag, kwargs[flag])
        self.scaling = kwargs["scaling"] if ("scaling" in kwargs and self.display) else 1.0
        self.pixels = kwargs["pixels"] if ("pixels" in kwargs and self.display) else 2
        for flag in ("show_logs", "save_model", "show_result"):
            if flag in kwargs:
                setattr(self, flag, kwargs[flag])
                _say(flag, kwargs[flag])
            else:
                setattr(self, flag, True)

    # ---- bookkeeping ------------------------------------------------------------------------------------------
    def _start_timer(self):
        self.time = time.time()

    def _make_name(self):
        if not hasattr(self, "name"):
            self.name = _make_name(model=self)
            print("[I] name         :", self.name)

    def _stop_timer(self):
        if not hasattr(self, "time"):
            print("[W] Timer not started.")
            return
        elapsed = time.time() - self.time
        h, rem = divmod(elapsed, 3600)
        mnt, sec = divmod(rem, 60)
        text = (f"{int(h)}h" if h > 0 else "") + (f"{int(mnt)}m" if mnt > 0 else "") + f"{int(sec)}s"
        print("[I] time elapsed : ", text)
        self.time = text

    def _init_logs(self):
        if not hasattr(self, "logs"):
            self.logs = {}

    def _init_factors(self):
        if hasattr(self, "U") or hasattr(self, "V"):
            print("[I] U, V existed. Skipping initialization.")
            return
        k = self.k if getattr(self, "k", None) is not None else 1
        self.U, self.V = np.zeros((self.m, k)), np.zeros((self.n, k))

    def _show_logs(self):
        for log in self.logs.values():
            if isinstance(log, pd.DataFrame):
                with pd.option_context("display.max_rows", None, "display.max_columns", None):
                    print(log)

    def _show_result(self):
        print("[W] show_result: plotting is outside this build (matplotlib display of gt/pd is not reproduced).")

    def _save_model(self, path=None, name=None):
        """Pickle the model state to ~/.pybmf/saved_models/<name>.pickle.  Unlike the reference (which pickles the whole
        __dict__, training matrix and mask included) device handles and the m x n data stay out: factors, logs and
        parameters are the checkpoint."""
        name = self.name
        skip = {"X_train", "X_val", "X_test", "W", "rng"}
        data = {k: v for k, v in self.__dict__.items() if k not in skip and not k.startswith("_")}
        if path is None:
            path, _ = get_cache_path(relative_path="saved_models/" + name + ".pickle")
        self.pickle_path = path
        with open(path, "wb") as fh:
            pickle.dump(data, fh, protocol=pickle.HIGHEST_PROTOCOL)
        print("[I] model saved as: {}.pickle".format(name))

    def import_model(self, **kwargs):
        for attr, value in kwargs.items():
            action = "Overwrote" if hasattr(self, attr) else "Imported"
            setattr(self, attr, value)
            self.print_msg("{} model parameter: {}".format(action, attr))

    def print_msg(self, msg, type="I"):
        if self.verbose:
            print("[{}] {}".format(type, msg))

    # ---- stopping rule ----------------------------------------------------------------------------------------
    def early_stop(self, error=None, diff=None, n_iter=None, n_factor=None, msg=None, k=None, verbose=True):
        """True while fitting should go on.  Stops on error <= tol, n_iter > max_iter (so max_iter + 1 updates run),
        diff < min_diff, n_factor >= k, or a forced `msg`."""
        reasons = []
        if error is not None and hasattr(self, "tol") and error <= self.tol:
            reasons.append(("Error <= tolerance", k))
        if n_iter is not None and hasattr(self, "max_iter") and n_iter > self.max_iter:
            reasons.append(("Reach maximum iteration", k))
        if diff is not None and hasattr(self, "min_diff") and diff < self.min_diff:
            reasons.append(("Difference lower than threshold", k))
        if n_factor is not None and getattr(self, "k", None) is not None and n_factor >= self.k:
            reasons.append(("Reach requested factor", None))
        if msg is not None:
            reasons.append((msg, k))
        for text, kk in reasons:
            self._early_stop(msg=text, verbose=verbose, k=kk)
        return not reasons

    def _early_stop(self, msg, verbose=True, k=None):
        if verbose:
            print("[W] Stopped in advance: " + msg)
        if k is not None:
            if verbose:
                print("[W] Obtained {} factor(s).".format(k))
            self.truncate_factors(k)

    def truncate_factors(self, k):
        self.U, self.V = self.U[:, :k], self.V[:, :k]
