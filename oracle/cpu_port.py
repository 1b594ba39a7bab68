"""ctypes loader for oracle/_build/libcpu_port.so (test infrastructure only; see cpu_port.c)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libcpu_port.so")
ANSATZ_IDS = {"hardware_efficient": 0, "all_to_all": 1, "basic": 2}
_lib = None


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_PATH)
        dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        L.port_circuit_probs.restype = C.c_int
        L.port_circuit_probs.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp]
        L.port_paramshift_probs.restype = C.c_int
        L.port_paramshift_probs.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, C.c_int, C.c_int, C.c_int, dp]
        L.port_gram_rows.restype = None
        L.port_gram_rows.argtypes = [C.c_int, C.c_double, dp, C.c_long, C.c_long, dp]
        L.port_gemv_rows.restype = C.c_double
        L.port_gemv_rows.argtypes = [C.c_int, dp, C.c_long, C.c_long, dp, dp]
        L.port_kron_matvec.restype = C.c_double
        L.port_kron_matvec.argtypes = [C.c_int, C.c_double, dp, dp, dp, dp]
        L.port_max_threads.restype = C.c_int
        _lib = L
    return _lib


def circuit_probs(ansatz, n, layers, thetas):
    thetas = np.ascontiguousarray(np.atleast_2d(thetas), dtype=np.float64)
    B, P = thetas.shape
    out = np.empty((B, 1 << n))
    lib().port_circuit_probs(ANSATZ_IDS[ansatz], n, layers, B, P, thetas.reshape(-1) if P else np.zeros(1), out.reshape(-1))
    return out


def paramshift_probs(ansatz, n, layers, theta, p_begin, p_end, include_base=True):
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    B = (1 if include_base else 0) + 2 * (p_end - p_begin)
    out = np.empty((B, 1 << n))
    used = lib().port_paramshift_probs(ANSATZ_IDS[ansatz], n, layers, theta.size, theta, p_begin, p_end,
                                       1 if include_base else 0, out.reshape(-1))
    return out, used


def gram_rows(S, n, length_scale, r0, r1):
    S = np.ascontiguousarray(S, dtype=np.float64)
    K = np.empty((r1 - r0, 1 << n))
    lib().port_gram_rows(n, float(length_scale), S.reshape(-1), r0, r1, K.reshape(-1))
    return K


def gemv_rows(K_rows, n, r0, r1, q):
    y = np.empty(r1 - r0)
    part = lib().port_gemv_rows(n, np.ascontiguousarray(K_rows).reshape(-1), r0, r1, np.ascontiguousarray(q), y)
    return y, part


def kron_matvec(S, q, n, length_scale=1.0):
    """(y = K_p q, q . y) matrix-free."""
    S = np.ascontiguousarray(S, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    y = np.empty(1 << n)
    work = np.empty(2 << n)
    k2 = lib().port_kron_matvec(n, float(length_scale), S.reshape(-1), q, y, work)
    return y, k2


def max_threads():
    return lib().port_max_threads()
