"""Minimal `tensorflow` shim used ONLY by tests/golden/make_golden.py to import the reference's
diffpiso/piso_helpers.py on PhiFlow's numpy backend in the build container (TensorFlow 1.14 is not installed).

It contains NO arithmetic: dtype aliases, `is_tensor` and a `custom_gradient` decorator that calls the wrapped
function, returns its forward value and stashes the returned gradient closure in LAST_GRAD so the generator
can evaluate the reference's own custom-gradient formulas.  It is never imported by the product, the oracle
or the test-suite.
"""
import numpy as _np

float32, float64, int32, bool = _np.float32, _np.float64, _np.int32, _np.bool_
LAST_GRAD = {}


def is_tensor(x):
    return False


def custom_gradient(f):
    def wrapped(*args, **kwargs):
        out, grad = f(*args, **kwargs)
        LAST_GRAD[f.__name__] = grad
        return out
    wrapped.__name__ = f.__name__
    return wrapped
