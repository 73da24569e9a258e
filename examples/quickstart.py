"""Quick start: the reference's config #1 (1000 x 500 planted Boolean matrix, k = 8) through the drop-in classes.

    python examples/quickstart.py            # needs an MI355X (gfx950) and the built library (see README)

Every class keeps the constructor, fit() keywords, attributes and log tables of its PyBMF counterpart."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pybmf_amd.generators import SyntheticMatrixGenerator
from pybmf_amd.models import BinaryMFPenalty, BinaryMFThreshold, PNLPF, WNMF

gen = SyntheticMatrixGenerator(m=1000, n=500, k=8, density=[0.2, 0.2])
gen.generate(seed=1000)
gen.add_noise(noise=[0.05, 0.01], seed=2000)
X = gen.X
quiet = dict(task="reconstruction", show_logs=False, show_result=False, save_model=False)

# penalty-function BMF by multiplicative updates; the Boolean scores of every iteration are in logs['boolean']
bmf = BinaryMFPenalty(k=8, W="full", reg=1, reg_growth=1.02, init_method="normal", normalize_method="balance", max_iter=20, seed=2024)
bmf.fit(X, **quiet)
print(bmf.logs["updates"].tail(2).to_string())
print(bmf.logs["boolean"].tail(1).to_string())

# real-valued factors with WNMF, then learn the two thresholds that binarise them
nmf = WNMF(k=8, W="full", init_method="normal", max_iter=30, seed=2024)
nmf.fit(X, **quiet)
thr = BinaryMFThreshold(k=8, U=nmf.U, V=nmf.V, W="full", u=0.3, v=0.3, lamda=10, max_iter=30)
thr.fit(X, **quiet)
print(f"thresholds u = {thr.u:.4f}, v = {thr.v:.4f}")
print(thr.logs["updates"].tail(1).to_string())

# the post-nonlinear variant (sigmoid link on the product)
pn = PNLPF(k=8, W="full", reg=1, reg_growth=1.2, link_lamda=10, init_method="normal", normalize_method="balance", max_iter=10, seed=2024)
pn.fit(X, **quiet)
print(pn.logs["boolean"].tail(1).to_string())
