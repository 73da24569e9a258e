"""cProfile of BinaryMFPenalty.fit at config #1 (1000 x 500, k = 8, 21 updates): where do the 4.7 ms go?"""
import sys, os, cProfile, pstats, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pybmf_amd.generators import SyntheticMatrixGenerator
from pybmf_amd.models import BinaryMFPenalty
gen = SyntheticMatrixGenerator(m=1000, n=500, k=8, density=[0.2, 0.2])
gen.generate(seed=1000)
gen.add_noise(noise=[0.05, 0.01], seed=2000)
X = gen.X
kw = dict(task="reconstruction", show_logs=False, show_result=False, save_model=False)
pr = cProfile.Profile()
for rep in range(6):
    with contextlib.redirect_stdout(io.StringIO()):
        m = BinaryMFPenalty(k=8, W="full", reg=1, reg_growth=1.02, init_method="normal", normalize_method="balance", max_iter=20, seed=2024)
        if rep >= 2:
            pr.enable()
        m.fit(X, **kw)
        torch.cuda.synchronize()
        pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumulative").print_stats(32)
