"""The fit() template and evaluate() of every model (``PyBMF/models/BaseModel.py``: check_params :19-41, fit :44-66,
finish :104-119, load_dataset :123-150, evaluate :209-257, _evaluate :260-277).

``evaluate()`` keeps the reference's signature and log schema; its numbers come from the GPU (integer cover counts
for the Boolean scores, the tile-fused residual pass for RMSE / MAE) instead of scipy.sparse arithmetic.
"""
from __future__ import annotations

from itertools import product

import numpy as np

from ..utils import header, record, to_sparse
from .BaseModelTools import BaseModelTools

BOOLEAN_METRICS = ("TP", "FP", "TN", "FN", "Recall", "Precision", "Accuracy", "F1", "TPR", "PPV", "ACC")
REAL_METRICS = ("RMSE", "MAE")


class BaseModel(BaseModelTools):
    def __init__(self, **kwargs):
        raise NotImplementedError("This is a template class.")

    def check_params(self, **kwargs):
        self.set_params(**kwargs)
        self.set_config(**kwargs)

    def fit(self, X_train, X_val=None, X_test=None, **kwargs):
        self.check_params(**kwargs)
        self.load_dataset(X_train=X_train, X_val=X_val, X_test=X_test)
        self.init_model()

    def init_model(self):
        self._init_factors()
        self._init_logs()
        self._start_timer()
        self._make_name()

    def _fit(self):
        raise NotImplementedError("This is a template method.")

    def finish(self, show_logs=True, save_model=True, show_result=True):
        self._stop_timer()
        if getattr(self, "_sharded", False):
            import torch.distributed as dist
            if dist.get_rank() != 0:   # a sharded fit ends with the same model on every rank: rank 0 saves and shows it
                return
        if save_model:
            self._save_model()
        if show_result:
            self._show_result()
        if show_logs:
            self._show_logs()

    def load_dataset(self, X_train, X_val=None, X_test=None):
        if X_train is None:
            raise TypeError("Missing training data.")
        if X_val is None:
            print("[I] Missing validation data.")
        if X_test is None:
            print("[W] Missing testing data.")
        self._X_input = X_train            # may be ndarray, scipy sparse or a torch tensor (host or device)
        from scipy.sparse import issparse
        if issparse(X_train):
            self.X_train = to_sparse(X_train, "csr")
            if X_train.format != "csr":
                self._X_input = self.X_train    # coo / lil / dok ... : the device packer slices rows
        elif isinstance(X_train, np.ndarray):
            # the csr copy the reference keeps as `X_train` is made on first access: a dense 100k x 20k input would spend tens of
            # seconds and several GB on it, and the all-ones-mask fit never looks at it (the bits in HBM are packed from the array)
            self._X_train, self._X_train_src = None, X_train
        else:
            self.X_train = X_train          # device tensors / lazy row sources stay as they are
        self.X_val = None if X_val is None else to_sparse(X_val, "csr")
        self.X_test = None if X_test is None else to_sparse(X_test, "csr")
        self.m, self.n = X_train.shape
        for X in (self.X_val, self.X_test):
            if X is not None and X.shape != (self.m, self.n):
                raise ValueError("X_val / X_test must have the shape of X_train")

    @property
    def X_train(self):
        """The training matrix as the reference holds it (scipy csr for host inputs); built lazily from a dense array."""
        d = self.__dict__
        if d.get("_X_train") is None and d.get("_X_train_src") is not None:
            d["_X_train"] = to_sparse(d["_X_train_src"], "csr")
        return d.get("_X_train")

    @X_train.setter
    def X_train(self, value):
        self.__dict__["_X_train"], self.__dict__["_X_train_src"] = value, None

    # ---- prediction -------------------------------------------------------------------------------------------
    @property
    def X_pd(self):
        """The prediction matrix.  Materialised on first access (at 100k x 20k it is 2e9 cells; the loop never needs it)."""
        if self.__dict__.get("_X_pd") is None and hasattr(self, "_make_X_pd"):
            self.__dict__["_X_pd"] = self._make_X_pd()
        return self.__dict__.get("_X_pd")

    @X_pd.setter
    def X_pd(self, value):
        self.__dict__["_X_pd"] = value

    def predict_X(self, U=None, V=None, u=None, v=None, us=None, vs=None, boolean=True):
        from ..device_ops import boolean_product_csr, product_csr
        U = self.U if U is None else U
        V = self.V if V is None else V
        if boolean:
            self.X_pd = boolean_product_csr(U, V, u=u, v=v, us=us, vs=vs)
        else:
            self.X_pd = product_csr(U, V, boolean=False)

    def show_matrix(self, *args, **kwargs):
        print("[W] show_matrix: plotting is outside this build.")

    # ---- evaluation -------------------------------------------------------------------------------------------
    def evaluate(self, df_name, head_info={}, train_info={}, val_info={}, test_info={},
                 metrics=["Recall", "Precision", "Accuracy", "F1"],
                 train_metrics=None, val_metrics=None, test_metrics=None, verbose=False):
        """Score the current prediction on the train / val / test sets and append one row to logs[df_name]."""
        train_metrics = metrics if train_metrics is None else train_metrics
        val_metrics = metrics if val_metrics is None else val_metrics
        test_metrics = metrics if test_metrics is None else test_metrics
        columns = header(list(head_info.keys()), levels=3)
        results = list(head_info.values())
        sets = [("train", train_info, train_metrics)]
        if self.X_val is not None:
            sets.append(("val", val_info, val_metrics))
        if self.X_test is not None:
            sets.append(("test", test_info, test_metrics))
        for name, info, mts in sets:
            c, r = self._evaluate(name, info, mts)
            columns += c
            results += r
        buf = getattr(self, "_log_buffer", None)
        if buf is not None and not verbose:
            # inside a fit that logs one row per outer iteration (BinaryMFThreshold): a pandas append copies the table every time
            # (45 of the 190 ms of a config-#5 fit), so the rows are kept and become one append when the fit ends (_flush_logs)
            cols, rows = buf.setdefault(df_name, (columns, []))
            if cols != columns:       # a different set of columns: fall back to the immediate append
                self._flush_logs()
                record(df_dict=self.logs, df_name=df_name, columns=columns, records=results, verbose=verbose)
            else:
                rows.append(list(results))
            return
        record(df_dict=self.logs, df_name=df_name, columns=columns, records=results, verbose=verbose)

    def _flush_logs(self):
        from ..utils import record_many
        buf = getattr(self, "_log_buffer", None)
        if buf:
            for df_name, (columns, rows) in list(buf.items()):
                record_many(self.logs, df_name, columns, rows)
            buf.clear()

    def _evaluate(self, name, info, metrics):
        if getattr(self, "task", None) is None:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute 'task'")
        assert self.task in ("prediction", "reconstruction"), "[E] Task should be either 'prediction' or 'reconstruction'."
        values = self._score(name, list(metrics))
        columns = list(product([name], [0], list(info.keys()) + list(metrics)))
        return columns, list(info.values()) + values

    def _score(self, name, metrics):
        """Metric values of data set `name` for the current state; subclasses answer from the device."""
        raise NotImplementedError
