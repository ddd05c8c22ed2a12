"""Drop-in for the reference's empirical-CDF helper and the `mip` column built from it.

    ecdf(x)            src/giremi/stat.py:7-29
    mean_mi_to_mip     src/giremi/script/giremi.py:415-429

The GLM scoring of stat.py (:32-143) is out of scope."""
from __future__ import annotations

from typing import Optional

import numpy as np

from .engine import Engine, default_engine


def ecdf(x, engine: Optional[Engine] = None):
    """f = ecdf(x); f(v) = fraction of x strictly below v (scalar or array v), as stat.ecdf"""
    ref = np.array(x, dtype=np.float64).ravel()
    if ref.size == 0:
        raise ZeroDivisionError('division by zero')

    def f(sample):
        eng = engine or default_engine()
        out = eng.ecdf(ref, np.atleast_1d(np.asarray(sample, np.float64)))
        return out if np.ndim(sample) else out[0]
    return f


def mean_mi_to_mip(mean_mi, site_type, engine: Optional[Engine] = None) -> np.ndarray:
    """the `mip` column: ECDF of the het_snp sites' mean_mi evaluated at every site's mean_mi; NaN where
    a site has no mean_mi; all NaN when no site has one (script/giremi.py:417-429)"""
    m = np.asarray(mean_mi, np.float64)
    t = np.asarray(site_type)
    out = np.full(m.shape, np.nan)
    if np.isnan(m).all():
        return out
    ref = m[~np.isnan(m) & (t == 'het_snp')]
    eng = engine or default_engine()
    return eng.ecdf(ref, m)
