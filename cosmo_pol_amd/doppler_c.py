"""Drop-in for the reference's SWIG module doppler_c (doppler/doppler_c.i:16-28)."""
import numpy as np

from . import _lib


def get_refl(n, Da, Db, D, rcs, N0, mu, lambdas, nu, step_D, D_min):
    lib = _lib.load()
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    Da, Db, D, rcs, N0, mu, lambdas, nu, step_D, D_min = map(f, (Da, Db, D, rcs, N0, mu, lambdas, nu, step_D, D_min))
    out = np.zeros(int(n), dtype=np.float32)
    p = _lib.fptr
    r = lib.get_refl(p(out), int(n), p(Da), Da.shape[0], 3, p(Db), Db.shape[0], 3, p(D), D.shape[0], 3,
                     p(rcs), rcs.shape[0], 3, p(N0), 3, p(mu), 3, p(lambdas), 3, p(nu), 3, p(step_D), 3, p(D_min), 3)
    if not r:
        raise TypeError("get_refl: bad arguments or no GPU")
    return (None, out)
