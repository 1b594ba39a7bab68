"""ctypes binding of libbornvi_hip.so (C ABI declared in include/bornvi.h).

The HIP library is the only compute backend of this package: there is no CPU fallback.  If the
shared object is missing or a GPU entry point is called without a GPU, an exception is raised.

`import torch` happens before the library is loaded so that the process uses ONE HIP runtime
(libamdhip64.so.7 is resolved to the copy PyTorch already mapped); device memory, streams and
collectives all come from PyTorch.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede loading the HIP library, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libbornvi_hip.so"
LIB_PATH = os.environ.get("BORNVI_LIB") or os.path.join(_HERE, LIB_NAME)   # BORNVI_LIB: a tuning build of the same library

BORNVI_OK = 0
ANSATZ_IDS = {"hardware_efficient": 0, "all_to_all": 1, "basic": 2}

_lib = None


class BornviError(RuntimeError):
    pass


class BnDesc(C.Structure):
    _fields_ = [("num_nodes", C.c_int32), ("max_parents", C.c_int32),
                ("role", C.c_void_p), ("n_parents", C.c_void_p), ("parents", C.c_void_p),
                ("cpt_off", C.c_void_p), ("cpt", C.c_void_p)]


_PROTOS = {
    # name: (restype, argtypes)
    "bornvi_version": (C.c_int, []),
    "bornvi_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "bornvi_destroy": (None, [C.c_void_p]),
    "bornvi_last_error": (C.c_char_p, [C.c_void_p]),
    "bornvi_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_longlong]),
    "bornvi_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong)]),
    "bornvi_num_params": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "bornvi_num_gates": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "bornvi_circuit_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "bornvi_circuit_probs": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    "bornvi_paramshift_probs": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bornvi_paramshift_probs_strided": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                  C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bornvi_paramshift_grad_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "bornvi_paramshift_grad": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bornvi_paramshift_dot_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "bornvi_paramshift_dot_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bornvi_paramshift_dot_finish": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bornvi_gate1q_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_int,
                                      C.POINTER(C.c_double), C.c_void_p]),
    "bornvi_cnot_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "bornvi_born_probs": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bornvi_score_from_cpts": (C.c_int, [C.c_void_p, C.POINTER(BnDesc), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bornvi_stein_gram_build": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bornvi_stein_gram_build_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_longlong, C.c_longlong,
                                               C.c_void_p, C.c_void_p]),
    "bornvi_stein_gram_ld": (C.c_longlong, [C.c_int]),
    "bornvi_stein_gram_build_rows_ld": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_longlong, C.c_longlong,
                                                  C.c_void_p, C.c_longlong, C.c_void_p]),
    "bornvi_stein_quadform_sym_ld": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bornvi_stein_quadform_sym_pairs_ld": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong,
                                                     C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                                     C.c_void_p]),
    "bornvi_stein_quadform_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_longlong, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bornvi_stein_kp_pairs": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_longlong, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bornvi_stein_quadform_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "bornvi_stein_quadform": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bornvi_stein_quadform_ld": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bornvi_stein_quadform_sym_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "bornvi_stein_quadform_sym": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_size_t, C.c_void_p]),
    "bornvi_stein_sym_strip_rows": (C.c_int, []),
    "bornvi_stein_quadform_sym_pairs": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bornvi_stein_matvec_kron_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "bornvi_stein_matvec_kron": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bornvi_ksd_grad_finish": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bornvi_clip_cast_grad": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bornvi_clip_adam_step": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                        C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bornvi_clip_cast_grad_guard": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "bornvi_adjoint_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "bornvi_adjoint_state": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_void_p]),
    "bornvi_adjoint_vjp": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "bornvi_debug_circuit_stamps": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong)]),
    "bornvi_plan_describe": (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_size_t]),
    "bornvi_stream_create_cu_range": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "bornvi_stream_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bornvi_plan_param_first_pass": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "bornvi_plan_fast_describe": (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_size_t,
                                                 C.POINTER(C.c_uint32), C.c_int]),
    "bornvi_plan_compact_describe": (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_size_t,
                                                    C.POINTER(C.c_uint32), C.c_int]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)


def lib():
    """The loaded library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BornviError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def plan_words(ansatz_id, n, layers, tile_bits=0):
    """Serialised execution plan (host only; no GPU needed)."""
    import numpy as np
    L = lib()
    need = L.bornvi_plan_describe(ansatz_id, n, layers, tile_bits, None, 0)
    if need < 0:
        raise BornviError("bornvi_plan_describe: unsupported configuration")
    buf = (C.c_uint32 * need)()
    L.bornvi_plan_describe(ansatz_id, n, layers, tile_bits, buf, need)
    return np.frombuffer(buf, dtype=np.uint32).copy()


def plan_param_first_pass(words):
    """First pass whose matrices depend on parameter p, read off a serialised plan (plan.hpp: PW_MATS lists the fused
    gates of every stage of a pass, a fused gate lists its parameters) -- the same rule the library applies when it
    orders a parameter-shift batch for prefix sharing (Plan::param_first_pass)."""
    import numpy as np
    W = words
    npass, nparams, off_fused, off_passtab = int(W[3]), int(W[5]), int(W[6]), int(W[7])
    first = np.full(nparams, -1, dtype=np.int64)
    for i in range(npass):
        base = int(W[off_passtab + i])
        for sm in range(int(W[base + 3]) * 4):                    # PW_NSTAGES stages x 4 register bits
            w = int(W[base + 128 + (sm >> 1)])                    # PW_MATS
            f = (w >> 16) if (sm & 1) else (w & 0xffff)
            if f == 0xffff:
                continue
            fw = W[off_fused + f * 10: off_fused + (f + 1) * 10]  # FUSED_WORDS
            for e in range(int(fw[1])):
                par = int(fw[3 + 2 * e])
                if par != 0xffffffff and par < nparams and first[par] < 0:
                    first[par] = i
    first[first < 0] = 0
    return first


def plan_fast_words(ansatz_id, n, layers, tile_bits=0):
    """Fast-path tables of a plan: (words, pass offsets), or (None, None) when the plan is not eligible."""
    import numpy as np
    L = lib()
    need = L.bornvi_plan_fast_describe(ansatz_id, n, layers, tile_bits, None, 0, None, 0)
    if need < 0:
        raise BornviError("bornvi_plan_fast_describe: unsupported configuration")
    if need == 0:
        return None, None
    npass = int(plan_words(ansatz_id, n, layers, tile_bits)[3])
    buf = (C.c_uint32 * need)()
    offs = (C.c_uint32 * npass)()
    L.bornvi_plan_fast_describe(ansatz_id, n, layers, tile_bits, buf, need, offs, npass)
    return np.frombuffer(buf, dtype=np.uint32).copy(), np.frombuffer(offs, dtype=np.uint32).copy()


R3 = 0x200     # flag in `tile_bits` of the plan_* helpers: the 3-register-wire plan (8 amplitudes per thread)


def plan_compact_words(ansatz_id, n, layers, tile_bits=0):
    """Compact tables of the 3-register-wire plan (plan.hpp: CompactTables): (words, pass offsets) or (None, None) when the
    plan is not eligible for circuit_pass_r3_kernel.  The library checks every word against its point evaluation."""
    import numpy as np
    L = lib()
    need = L.bornvi_plan_compact_describe(ansatz_id, n, layers, tile_bits, None, 0, None, 0)
    if need < 0:
        raise BornviError("bornvi_plan_compact_describe: unsupported configuration")
    if need == 0:
        return None, None
    npass = int(plan_words(ansatz_id, n, layers, tile_bits | R3)[3])
    buf = (C.c_uint32 * need)()
    offs = (C.c_uint32 * npass)()
    L.bornvi_plan_compact_describe(ansatz_id, n, layers, tile_bits, buf, need, offs, npass)
    return np.frombuffer(buf, dtype=np.uint32).copy(), np.frombuffer(offs, dtype=np.uint32).copy()


class Handle:
    """One bornvi handle per device; thin checked wrappers around the C ABI."""

    def __init__(self, device_index=0):
        self._lib = lib()
        h = C.c_void_p()
        rc = self._lib.bornvi_create(int(device_index), C.byref(h))
        if rc != BORNVI_OK:
            raise BornviError(f"bornvi_create failed ({rc}): {self._lib.bornvi_last_error(None).decode()}")
        self.h = h
        self.device_index = int(device_index)

    def close(self):
        if getattr(self, "h", None):
            self._lib.bornvi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, what):
        if rc != BORNVI_OK:
            raise BornviError(f"{what} failed ({rc}): {self._lib.bornvi_last_error(self.h).decode()}")

    def call(self, name, *args):
        self.check(getattr(self._lib, name)(self.h, *args), name)

    def size(self, name, *args):
        v = getattr(self._lib, name)(self.h, *args)
        if v == 0:
            raise BornviError(f"{name} returned 0: {self._lib.bornvi_last_error(self.h).decode()}")
        return int(v)


_handles = {}


def handle_for(device):
    """Cached handle for a torch.device (must be a 'cuda' device: PyTorch-ROCm's spelling)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise BornviError(f"the bornvi backend runs on MI355X only; got device '{dev}' (no CPU fallback)")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _handles:
        _handles[idx] = Handle(idx)
    return _handles[idx]


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
