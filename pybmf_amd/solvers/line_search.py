"""Wolfe line search on a small parameter vector, host side (``PyBMF/solvers/line_search.py:4-104``).

The control flow is the reference's bracketing / bisection scheme; `f` and `myfprime` are callables -- in
BinaryMFThreshold they are one tile-fused GPU pass each (csrc/residual.hip), so the search itself stays in Python.
"""
import numpy as np


def line_search(f, myfprime, xk, pk, args=(), kwargs={}, maxiter=1000, c1=0.1, c2=0.4, prefetch=None, chain=8):
    """Returns (alpha, fc, gc, new_fval, old_fval, new_slope), signature-compatible with scipy.optimize.line_search.

    Start at alpha = 2 inside the bracket [0, 10].  Armijo fails -> shrink the upper end and bisect; Armijo holds but
    the curvature condition fails -> raise the lower end and bisect once an upper end was found, else grow alpha by 1.2.

    `prefetch(points)`: optional.  From a state (lo, alpha) the trial steps that follow while the Armijo test keeps failing are known
    before any F is evaluated -- alpha, (lo + alpha) / 2, ... (line_search.py:47-62 of the reference: no interpolation) -- so the
    search hands that chain, `chain` points long, to `prefetch` whenever its next step is not in the chain it announced last; an `f`
    that answers from what `prefetch` computed (one batched launch) makes the same decisions in a fraction of the launches.  The
    search itself, its counts and its return values do not depend on it."""
    lo, hi = 0, 10
    alpha = 2
    f0 = f(xk, *args, **kwargs)
    g0 = myfprime(xk, *args, **kwargs)
    n_f, n_g = 1, 1
    slope0 = np.dot(g0, pk)
    x = xk
    trips = 0
    announced = ()
    while trips <= maxiter:
        trips += 1
        if prefetch is not None and alpha not in announced:
            steps, a = [], alpha
            for _ in range(max(int(chain), 1)):
                steps.append(a)
                a = (lo + a) / 2
            announced = tuple(steps)
            prefetch([xk + a * pk for a in steps])
        x = xk + alpha * pk
        decrease_ok = f(x, *args, **kwargs) - f0 <= alpha * c1 * slope0
        n_f += 1
        if not decrease_ok:
            hi = alpha
            alpha = (lo + hi) / 2
            continue
        curvature_ok = np.dot(myfprime(x, *args, **kwargs), pk) >= c2 * slope0
        n_g += 1
        if curvature_ok:
            break
        if hi < 10:
            lo = alpha
            alpha = (lo + hi) / 2
        else:
            alpha = alpha * 1.2
    new_fval = f(x, *args, **kwargs)
    new_slope = myfprime(x, *args, **kwargs)
    return alpha, n_f + 1, n_g + 1, new_fval, f0, new_slope


def limit_step_size(x_min, x_max, x_last, xk, alpha, pk):
    """Shorten the step so that xk + alpha * pk stays inside the box [x_min, x_max]; returns (x_last, alpha)."""
    inside = (x_last <= x_max).all() and (x_last >= x_min).all()
    if inside:
        return x_last, alpha
    best = alpha
    for i in range(len(x_last)):
        cand = best
        if x_last[i] > x_max[i]:
            cand = (x_max[i] - xk[i]) / pk[i]
        if x_last[i] < x_min[i]:
            cand = (x_min[i] - xk[i]) / pk[i]
        if cand < best:
            best = cand
    return xk + best * pk, best
