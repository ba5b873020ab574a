"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import importlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libavm_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
    return _LIB


def _pkg():
    return importlib.import_module("anticipated-vins-mono_amd")


def window_solve(opt, win, prior_out=None, summary=None, n_threads=1):
    """In-place Estimator::optimization() on host WindowArrays."""
    from_buffers = importlib.import_module("anticipated-vins-mono_amd.buffers")
    s = win.struct()
    po = prior_out.struct() if prior_out is not None else None
    sp = from_buffers.summary_ptr(summary) if summary is not None else None
    rc = lib().avmo_window_solve_batch(C.byref(opt), C.byref(s), C.byref(po) if po is not None else None, sp, int(n_threads))
    assert rc == 0
    return rc


def fsel_select(fsel, out, n_threads=1):
    s, o = fsel.struct(), out.struct()
    n = C.c_int64(0)
    rc = lib().avmo_fsel_select_batch(C.byref(s), C.byref(o), int(n_threads), C.byref(n))
    assert rc == 0
    return n.value
