"""Tensor-level wrappers over the bornvi C ABI (include/bornvi.h).

PyTorch owns every buffer (inputs, outputs, workspaces) and the stream; this module only checks
shapes / dtypes / devices and passes raw pointers.  Everything here needs an MI355X: tensors must
live on a 'cuda' (PyTorch-ROCm) device.  There is no CPU path.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _ext
from ._ext import ANSATZ_IDS, BornviError

_workspaces = {}
_ws_windows = {}     # key -> (byte offset, bytes): tools/probes place a workspace inside another buffer through this

# Cap for the circuit workspace (bytes); larger batches are processed in chunks by the library.
WORKSPACE_CAP = int(os.environ.get("BORNVI_WORKSPACE_CAP", str(48 << 30)))


def compute_device(preferred=None):
    """The GPU this process computes on: `preferred` if it is a cuda device, else cuda:LOCAL_RANK / current."""
    if preferred is not None:
        d = torch.device(preferred)
        if d.type == "cuda":
            return torch.device("cuda", d.index if d.index is not None else torch.cuda.current_device())
    if not torch.cuda.is_available():
        raise BornviError("no MI355X visible: the bornvi backend has no CPU fallback "
                          "(torch.cuda.is_available() is False)")
    return torch.device("cuda", torch.cuda.current_device())


def ansatz_id(ansatz_type):
    """quantum_born_machine.py:31-38/:113: any string other than the two named ones selects 'basic'."""
    return ANSATZ_IDS.get(ansatz_type, ANSATZ_IDS["basic"])


def num_params(ansatz_type, n, layers):
    return _ext.lib().bornvi_num_params(ansatz_id(ansatz_type), n, layers)


def _ws_key(dev, tag):
    return (dev.index, tag, int(torch.cuda.current_stream(dev).cuda_stream))


def _ws(dev, nbytes, tag="main"):
    """Cached workspace, one per (device, purpose, STREAM): a buffer is only ever used on the stream it was allocated
    on, so the caching allocator's stream-ordered reuse stays valid when a workspace is re-grown (the overlap modes
    launch on auxiliary / CU-masked streams)."""
    key = _ws_key(dev, tag)
    buf = _workspaces.get(key)
    win = _ws_windows.get(key)
    if buf is not None and win is not None:        # (probes only: a chosen window of a larger buffer)
        if win[1] >= nbytes:
            return buf[win[0]: win[0] + win[1]]
        del _ws_windows[key]
        buf = None
    if buf is None or buf.numel() < nbytes:
        _workspaces[key] = None
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        _workspaces[key] = buf
    return buf


def fresh_workspace(dev, nbytes):
    """A new device buffer that can be installed as a workspace (set_workspace)."""
    return torch.empty(int(nbytes), dtype=torch.uint8, device=dev)


def set_workspace(dev, tag, buf):
    """Makes `buf` the cached workspace `tag` of the current stream (the trainer places the symmetric contraction's
    workspace by measurement: KSDVariationalInference._place_gram)."""
    key = _ws_key(dev, tag)
    _ws_windows.pop(key, None)
    _workspaces[key] = buf


def release_workspaces():
    _workspaces.clear()
    _ws_windows.clear()


def _chk(t, dtype, dev, name, numel=None):
    """dtype / device / contiguity and (the C ABI sees raw pointers only) the element count the kernels will index."""
    if t.dtype != dtype or t.device != dev or not t.is_contiguous():
        raise BornviError(f"{name}: expected contiguous {dtype} on {dev}, got {t.dtype} on {t.device}")
    if numel is not None and t.numel() != int(numel):
        raise BornviError(f"{name}: expected {int(numel)} elements, got {t.numel()} (shape {tuple(t.shape)})")


def _chk_n(n, lo=1, hi=30):
    if not (isinstance(n, (int, np.integer)) and lo <= int(n) <= hi):
        raise BornviError(f"number of qubits / latent variables out of range: {n!r}")


def _ptr(t):
    return C.c_void_p(t.data_ptr())


_size_cache = {}


def _cached_size(h, name, *args):
    key = (h.device_index, name) + args
    v = _size_cache.get(key)
    if v is None:
        v = _size_cache[key] = h.size(name, *args)
    return v


def clip_cast_grad(grad64, max_norm):
    """float64 gradient -> (float32 gradient clipped like clip_grad_norm_, its float32 total norm [0-dim])."""
    dev = grad64.device
    h = _ext.handle_for(dev)
    _chk(grad64, torch.float64, dev, "grad64")
    g32 = torch.empty(grad64.shape, dtype=torch.float32, device=dev)
    norm = torch.empty((), dtype=torch.float32, device=dev)
    h.call("bornvi_clip_cast_grad", grad64.numel(), _ptr(grad64), float(max_norm), _ptr(g32), _ptr(norm),
           _ext.stream_ptr(dev))
    return g32, norm


def clip_cast_grad_guard(grad64, max_norm, loss, out=None, found_out=None):
    """clip_cast_grad plus the NaN/Inf guard of the loss as a device flag: -> (g32, norm [0-dim], found_inf [0-dim]
    float32, 1.0 when loss [1] float64 is NaN or +-Inf), one launch.  out / found_out: destinations to write into
    (theta.grad and the optimiser's found_inf tensor in a captured step: two copy nodes fewer per replay)."""
    dev = grad64.device
    h = _ext.handle_for(dev)
    _chk(grad64, torch.float64, dev, "grad64")
    _chk(loss, torch.float64, dev, "loss")
    if out is not None:
        _chk(out, torch.float32, dev, "out", grad64.numel())
    if found_out is not None:
        _chk(found_out, torch.float32, dev, "found_out", 1)
    g32 = out if out is not None else torch.empty(grad64.shape, dtype=torch.float32, device=dev)
    norm = torch.empty((), dtype=torch.float32, device=dev)
    found = found_out if found_out is not None else torch.empty((), dtype=torch.float32, device=dev)
    h.call("bornvi_clip_cast_grad_guard", grad64.numel(), _ptr(grad64), float(max_norm), _ptr(loss), _ptr(g32), _ptr(norm),
           _ptr(found), _ext.stream_ptr(dev))
    return g32, norm, found


def clip_adam_step(grad64, max_norm, loss, theta, grad32, theta64, exp_avg, exp_avg_sq, counters, lr_table, beta1, beta2,
                   eps, norm_out=None, loss_history=None, norm_history=None):
    """clip_cast_grad_guard + the Adam update of float32 theta + the float64 copy of the new theta, one launch
    (bornvi_clip_adam_step).  All tensors on the GPU and updated in place; -> the gradient's float32 norm [0-dim]."""
    dev = grad64.device
    h = _ext.handle_for(dev)
    P = grad64.numel()
    _chk(grad64, torch.float64, dev, "grad64")
    _chk(loss, torch.float64, dev, "loss")
    for t, dt, nm in ((theta, torch.float32, "theta"), (grad32, torch.float32, "grad32"), (theta64, torch.float64, "theta64"),
                      (exp_avg, torch.float32, "exp_avg"), (exp_avg_sq, torch.float32, "exp_avg_sq")):
        _chk(t, dt, dev, nm, P)
    _chk(counters, torch.int32, dev, "counters", 2)
    _chk(lr_table, torch.float64, dev, "lr_table")
    norm = norm_out if norm_out is not None else torch.empty((), dtype=torch.float32, device=dev)
    _chk(norm, torch.float32, dev, "norm_out", 1)
    if loss_history is not None:
        _chk(loss_history, torch.float64, dev, "loss_history", lr_table.numel())
    if norm_history is not None:
        _chk(norm_history, torch.float32, dev, "norm_history", lr_table.numel())
    h.call("bornvi_clip_adam_step", P, _ptr(grad64), float(max_norm), _ptr(loss), _ptr(theta), _ptr(grad32), _ptr(theta64),
           _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(counters), _ptr(lr_table), lr_table.numel(), float(beta1), float(beta2),
           float(eps), _ptr(norm), _ptr(loss_history) if loss_history is not None else None,
           _ptr(norm_history) if norm_history is not None else None, _ext.stream_ptr(dev))
    return norm


def set_option(dev, name, value):
    _size_cache.clear()
    h = _ext.handle_for(dev)
    h.call("bornvi_set_option", name.encode(), int(value))


def get_option(dev, name):
    """Current value of a planner / engine option of this device's handle (bornvi_get_option)."""
    v = C.c_longlong(0)
    _ext.handle_for(dev).call("bornvi_get_option", name.encode(), C.byref(v))
    return int(v.value)


def set_engine_option(dev, name, value):
    """Options that do not change plans or workspace sizes (e.g. "circuit_cus"): no cache invalidation."""
    _ext.handle_for(dev).call("bornvi_set_option", name.encode(), int(value))


_cu_streams = {}


def cu_range_stream(dev, first_cu, num_cus):
    """A stream restricted to the CUs [first_cu, first_cu + num_cus) (bornvi_stream_create_cu_range), as a
    torch.cuda.ExternalStream so that torch events / allocator bookkeeping work with it.  Cached for the process."""
    dev = torch.device(dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(first_cu), int(num_cus))
    if key not in _cu_streams:
        import ctypes as C
        h = _ext.handle_for(dev)
        st = C.c_void_p()
        h.call("bornvi_stream_create_cu_range", int(first_cu), int(num_cus), C.byref(st))
        _cu_streams[key] = torch.cuda.ExternalStream(st.value, device=dev)
    return _cu_streams[key]


# ---- circuits -------------------------------------------------------------------------------------
def circuit_probs(ansatz_type, n, layers, thetas):
    """thetas float64 [B, P] on a cuda device -> probs float64 [B, 2^n]."""
    dev = thetas.device
    h = _ext.handle_for(dev)
    aid = ansatz_id(ansatz_type)
    _chk_n(n)
    P = num_params(ansatz_type, n, layers)
    if thetas.dim() != 2 or thetas.shape[1] != P:
        raise BornviError(f"thetas must be [batch, {P}]")
    _chk(thetas, torch.float64, dev, "thetas")
    B = thetas.shape[0]
    probs = torch.empty((B, 1 << n), dtype=torch.float64, device=dev)
    need = h.size("bornvi_circuit_workspace_bytes", aid, n, layers, B)
    ws = _ws(dev, min(need, max(WORKSPACE_CAP, h.size("bornvi_circuit_workspace_bytes", aid, n, layers, 1))))
    h.call("bornvi_circuit_probs", aid, n, layers, B, _ptr(thetas), _ptr(probs), _ptr(ws), ws.numel(),
           _ext.stream_ptr(dev))
    return probs


def paramshift_probs(ansatz_type, n, layers, theta, p_begin, p_end, include_base=True, out=None, ws_tag="main",
                     p_stride=1):
    """theta float64 [P] -> probs [(1 if include_base) + 2 count, 2^n]: optional base row, then (+p, -p) rows for
    p = p_begin, p_begin + p_stride, ... < p_end (count of them; p_stride = 1: the range [p_begin, p_end))."""
    dev = theta.device
    h = _ext.handle_for(dev)
    aid = ansatz_id(ansatz_type)
    _chk_n(n)
    _chk(theta, torch.float64, dev, "theta", num_params(ansatz_type, n, layers))
    count = len(range(p_begin, p_end, p_stride))
    B = (1 if include_base else 0) + 2 * count
    if out is None:
        out = torch.empty((B, 1 << n), dtype=torch.float64, device=dev)
    else:
        _chk(out, torch.float64, dev, "out")
        if out.numel() != B << n:
            raise BornviError("out has the wrong size")
    if B == 0:
        return out
    need = _cached_size(h, "bornvi_circuit_workspace_bytes", aid, n, layers, B)
    ws = _ws(dev, min(need, max(WORKSPACE_CAP, _cached_size(h, "bornvi_circuit_workspace_bytes", aid, n, layers, 1))), ws_tag)
    h.call("bornvi_paramshift_probs_strided", aid, n, layers, _ptr(theta), int(p_begin), int(count), int(p_stride),
           1 if include_base else 0, _ptr(out), _ptr(ws), ws.numel(), _ext.stream_ptr(dev))
    return out


def paramshift_dot_supported(ansatz_type, n, layers, dev, count):
    """True when the fused path exists for this plan (multi-pass plan of the 8-amplitude kernel, no prefix sharing)."""
    h = _ext.handle_for(dev)
    key = (id(h), "dot_supported", ansatz_id(ansatz_type), int(n), int(layers), int(count))
    if key not in _size_cache:       # (a size of 0 is the library's "not available": no error)
        _size_cache[key] = int(_ext.lib().bornvi_paramshift_dot_workspace_bytes(h.h, ansatz_id(ansatz_type), int(n), int(layers), int(count)))
    return _size_cache[key] > 0


def paramshift_dot_begin(ansatz_type, n, layers, theta, p_begin, p_end, p_stride=1, ws_tag="dot"):
    """First half of a parameter-shift step with the dot product fused into the last circuit pass: runs the base circuit
    and the shifted circuits of p = p_begin, p_begin + p_stride, ... < p_end up to their last pass, and the base circuit
    to the end.  Returns (q [2^n], token); give the token to paramshift_dot_finish once y = K_p q is known."""
    dev = theta.device
    h = _ext.handle_for(dev)
    aid = ansatz_id(ansatz_type)
    _chk_n(n)
    _chk(theta, torch.float64, dev, "theta", num_params(ansatz_type, n, layers))
    count = len(range(p_begin, p_end, p_stride))
    need = int(_cached_size(h, "bornvi_paramshift_dot_workspace_bytes", aid, n, layers, count))
    if need == 0:
        raise BornviError("the fused parameter-shift dot is not available for this plan (see paramshift_dot_supported)")
    ws = _ws(dev, need, ws_tag)
    q = torch.empty(1 << n, dtype=torch.float64, device=dev)
    h.call("bornvi_paramshift_dot_begin", aid, n, layers, _ptr(theta), int(p_begin), int(count), int(p_stride), _ptr(q), _ptr(ws),
           ws.numel(), _ext.stream_ptr(dev))
    return q, (aid, int(n), int(layers), count, ws)


def paramshift_dot_finish(token, w, ksd2=None):
    """Second half: the shifted circuits' last pass with the weights w [2^n] -> (loss [1] or None, grad [count]);
    ksd2 [1] given: grad carries the factor 1 / (2 sqrt(max(ksd2, 1e-12))) and loss = sqrt(max(ksd2, 1e-12)); else 1/2."""
    aid, n, layers, count, ws = token
    dev = w.device
    h = _ext.handle_for(dev)
    _chk(w, torch.float64, dev, "w", 1 << n)
    if ksd2 is not None:
        _chk(ksd2, torch.float64, dev, "ksd2", 1)
    grad = torch.empty(count, dtype=torch.float64, device=dev)
    loss = torch.empty(1, dtype=torch.float64, device=dev) if ksd2 is not None else None
    h.call("bornvi_paramshift_dot_finish", aid, n, layers, count, _ptr(w), _ptr(ksd2) if ksd2 is not None else None,
           _ptr(grad) if count else None, _ptr(loss) if loss is not None else None, _ptr(ws), ws.numel(), _ext.stream_ptr(dev))
    return loss, grad


def paramshift_grad(ansatz_type, n, layers, theta, dLdq, p_begin, p_end, p_stride=1):
    """grad[i] = 1/2 dLdq . (q(theta + pi/2 e_p) - q(theta - pi/2 e_p)) for p = p_begin + i p_stride < p_end, float64."""
    dev = theta.device
    h = _ext.handle_for(dev)
    aid = ansatz_id(ansatz_type)
    _chk_n(n)
    _chk(theta, torch.float64, dev, "theta", num_params(ansatz_type, n, layers))
    _chk(dLdq, torch.float64, dev, "dLdq", 1 << n)
    ns = len(range(p_begin, p_end, p_stride))
    grad = torch.empty(ns, dtype=torch.float64, device=dev)
    if ns == 0:
        return grad
    shifted = paramshift_probs(ansatz_type, n, layers, theta, p_begin, p_end, include_base=False, p_stride=p_stride)
    # dot products on the device: reuse the finishing kernel with ksd2 = 1 (loss = 1, scale = 1/2)
    one = torch.ones(1, dtype=torch.float64, device=dev)
    h.call("bornvi_ksd_grad_finish", n, _ptr(shifted), ns, _ptr(dLdq), _ptr(one), None, None, _ptr(grad),
           _ext.stream_ptr(dev))
    return grad


def adjoint_state(ansatz_type, n, layers, theta, want_probs=True):
    """OPT-IN adjoint engine, forward walk: theta float64 [P] -> (state complex128 [2^n], probs float64 [2^n] or None)."""
    dev = theta.device
    h = _ext.handle_for(dev)
    aid = ansatz_id(ansatz_type)
    _chk_n(n)
    _chk(theta, torch.float64, dev, "theta", num_params(ansatz_type, n, layers))
    state = torch.empty(1 << n, dtype=torch.complex128, device=dev)
    probs = torch.empty(1 << n, dtype=torch.float64, device=dev) if want_probs else None
    ws = _ws(dev, _cached_size(h, "bornvi_adjoint_workspace_bytes", aid, n, layers), "adjoint")
    h.call("bornvi_adjoint_state", aid, n, layers, _ptr(theta), _ptr(state), _ptr(probs) if want_probs else None, _ptr(ws),
           ws.numel(), _ext.stream_ptr(dev))
    return state, probs


def adjoint_vjp(ansatz_type, n, layers, theta, state, dLdq):
    """OPT-IN adjoint engine, backward walk: grad[p] = d/dtheta_p sum_z dLdq[z] q_z(theta), float64 [P] -- the quantity
    paramshift_grad computes with 2P circuits, from one forward state and one backward walk."""
    dev = theta.device
    h = _ext.handle_for(dev)
    aid = ansatz_id(ansatz_type)
    _chk_n(n)
    P = num_params(ansatz_type, n, layers)
    _chk(theta, torch.float64, dev, "theta", P)
    _chk(state, torch.complex128, dev, "state", 1 << n)
    _chk(dLdq, torch.float64, dev, "dLdq", 1 << n)
    grad = torch.zeros(P, dtype=torch.float64, device=dev)
    ws = _ws(dev, _cached_size(h, "bornvi_adjoint_workspace_bytes", aid, n, layers), "adjoint")
    h.call("bornvi_adjoint_vjp", aid, n, layers, _ptr(theta), _ptr(state), _ptr(dLdq), _ptr(grad), _ptr(ws), ws.numel(),
           _ext.stream_ptr(dev))
    return grad


def gate1q_apply(state, n, wire, U):
    """In-place one-qubit gate on state complex128 [B, 2^n] (one HBM round trip)."""
    dev = state.device
    h = _ext.handle_for(dev)
    _chk_n(n, 1, 40)
    _chk(state, torch.complex128, dev, "state")
    if state.numel() % (1 << n):
        raise BornviError("state: element count is not a multiple of 2^n")
    Uh = np.ascontiguousarray(np.asarray(U, dtype=np.complex128).reshape(4)).view(np.float64)
    arr = (C.c_double * 8)(*Uh.tolist())
    h.call("bornvi_gate1q_apply", n, state.numel() >> n, _ptr(state), int(wire), arr, _ext.stream_ptr(dev))
    return state


def cnot_apply(state, n, control, target):
    dev = state.device
    h = _ext.handle_for(dev)
    _chk_n(n, 2, 40)
    _chk(state, torch.complex128, dev, "state")
    if state.numel() % (1 << n):
        raise BornviError("state: element count is not a multiple of 2^n")
    h.call("bornvi_cnot_apply", n, state.numel() >> n, _ptr(state), int(control), int(target), _ext.stream_ptr(dev))
    return state


def born_probs(state, n):
    dev = state.device
    h = _ext.handle_for(dev)
    _chk_n(n, 0, 40)
    _chk(state, torch.complex128, dev, "state")
    if state.numel() % (1 << n):
        raise BornviError("state: element count is not a multiple of 2^n")
    probs = torch.empty(state.shape, dtype=torch.float64, device=dev)
    h.call("bornvi_born_probs", n, state.numel() >> n, _ptr(state), _ptr(probs), _ext.stream_ptr(dev))
    return probs


# ---- Stein ---------------------------------------------------------------------------------------
def score_from_packed(packed, n, dev):
    """packed: dict from bayesian_network.pack_network -> (S [2^n, n], pxz [2^n]) float64 on dev."""
    h = _ext.handle_for(dev)
    t = {k: torch.as_tensor(np.ascontiguousarray(v)).to(dev) for k, v in packed.items()}
    desc = _ext.BnDesc(int(t["role"].numel()), int(t["parents"].shape[1]), t["role"].data_ptr(),
                       t["n_parents"].data_ptr(), t["parents"].data_ptr(), t["cpt_off"].data_ptr(),
                       t["cpt"].data_ptr())
    S = torch.empty((1 << n, n), dtype=torch.float64, device=dev)
    pxz = torch.empty(1 << n, dtype=torch.float64, device=dev)
    h.call("bornvi_score_from_cpts", C.byref(desc), n, _ptr(S), _ptr(pxz), _ext.stream_ptr(dev))
    torch.cuda.current_stream(dev).synchronize()   # `t` (descriptor arrays) must outlive the kernel
    return S, pxz


def gram_ld(n):
    """Row pitch (doubles) the library recommends for a dense K_p that the symmetric contraction will stream:
    2^n + 32 for n >= 14 (a power-of-two pitch makes the row streams of a band collide on one HBM channel: n = 16
    2.56 ms padded against 2.62 ... 2.83 ms dense, by allocation), else 2^n."""
    return int(_ext.lib().bornvi_stein_gram_ld(int(n)))


def _chk_matrix(K, rows, N, dev, name):
    """K must be a float64 [rows, N] matrix on dev with unit column stride; returns its row pitch in doubles.  A padded
    matrix is the [:, :N] view of a [rows, ld] buffer (what stein_gram(ld=...) returns)."""
    if K.dtype != torch.float64 or K.device != dev or K.dim() != 2 or tuple(K.shape) != (int(rows), int(N)):
        raise BornviError(f"{name}: expected float64 [{int(rows)}, {int(N)}] on {dev}, got {K.dtype} {tuple(K.shape)} on {K.device}")
    if rows == 0:
        return int(N)
    ld = int(K.stride(0)) if rows > 1 else max(int(K.stride(0)), int(N))
    if K.stride(1) != 1 or ld < N or (ld & 1 and N > 1):
        raise BornviError(f"{name}: rows must be contiguous with an even pitch >= {int(N)} (strides {tuple(K.stride())})")
    return ld


def stein_gram(S, n, length_scale=1.0, rows=None, out=None, ld=None):
    """Dense K_p [2^n, 2^n], or only its rows [rows[0], rows[1]) (one rank's block of a row shard).
    ld: row pitch in doubles (default 2^n: a contiguous matrix; gram_ld(n) for the padded layout the symmetric
    contraction streams best) -- the result is then the [:, :2^n] view of a [rows, ld] buffer.
    out: a [rows, 2^n] float64 destination with unit column stride (e.g. rows of a larger, possibly padded, buffer)."""
    dev = S.device
    h = _ext.handle_for(dev)
    _chk_n(n, 1, 17)
    N = 1 << n
    _chk(S, torch.float64, dev, "S", n << n)
    r0, r1 = (0, N) if rows is None else (int(rows[0]), int(rows[1]))
    if not (0 <= r0 <= r1 <= N):
        raise BornviError("stein_gram: row range out of bounds")
    if out is None:
        ld = N if ld is None else int(ld)
        if ld < N or ld & 1 and N > 1:
            raise BornviError("stein_gram: ld must be even and >= 2^n")
        K = torch.empty((r1 - r0, ld), dtype=torch.float64, device=dev)[:, :N]
    else:
        K = out
    pitch = _chk_matrix(K, r1 - r0, N, dev, "out" if out is not None else "K")
    h.call("bornvi_stein_gram_build_rows_ld", n, float(length_scale), _ptr(S), r0, r1, _ptr(K), pitch, _ext.stream_ptr(dev))
    return K


def sym_pair_shard(n, rank, world_size):
    """Strip pairs [pa, pb) of the symmetric contraction owned by `rank`, and the two row ranges they cover:
    ((pa, pb), (rows_lo_begin, rows_lo_end), (rows_hi_begin, rows_hi_end)).  None when 2^n is too small to cut
    into whole strip pairs for every rank (the row shard is used then)."""
    R = int(_ext.lib().bornvi_stein_sym_strip_rows())
    N = 1 << n
    if N % (2 * R) != 0:
        return None
    ns = N // R
    npairs = ns // 2
    chunk = -(-npairs // world_size)
    pa = min(npairs, rank * chunk)
    pb = min(npairs, pa + chunk)
    return (pa, pb), (pa * R, pb * R), ((ns - pb) * R, (ns - pa) * R)


def stein_quadform_sym_pairs(K_lo, K_hi, pa, pb, q, n, out=None):
    """This GPU's additive share of (K q, q^T K q) from its strip pairs [pa, pb) of the upper triangle:
    returns a [2^n + 1] vector (y_partial followed by ksd2_partial) -- the message of the all-reduce."""
    dev = q.device
    h = _ext.handle_for(dev)
    _chk_n(n, 1, 17)
    N = 1 << n
    _chk(q, torch.float64, dev, "q", N)
    ld = N
    if pb > pa:
        R = int(_ext.lib().bornvi_stein_sym_strip_rows())
        ld = _chk_matrix(K_lo, (pb - pa) * R, N, dev, "K_lo")
        if _chk_matrix(K_hi, (pb - pa) * R, N, dev, "K_hi") != ld:
            raise BornviError("K_lo and K_hi must have the same row pitch")
    if out is None:
        out = torch.empty(N + 1, dtype=torch.float64, device=dev)
    else:
        _chk(out, torch.float64, dev, "out", N + 1)
    ws = _ws(dev, _cached_size(h, "bornvi_stein_quadform_sym_workspace_bytes", n), "qfsym")
    h.call("bornvi_stein_quadform_sym_pairs_ld", n, _ptr(K_lo) if pb > pa else None, _ptr(K_hi) if pb > pa else None,
           ld, int(pa), int(pb), _ptr(q), C.c_void_p(out.data_ptr() + 8 * N), _ptr(out), _ptr(ws), ws.numel(),
           _ext.stream_ptr(dev))
    return out


def stein_quadform_rows(K_rows, r0, r1, q, n, out=None):
    """K_rows = rows [r0, r1) of K_p; returns a [r1 - r0 + 1] vector: those rows of K q followed by
    the partial sum over them of q_i y_i (the message one rank contributes to the all-gather)."""
    dev = K_rows.device
    h = _ext.handle_for(dev)
    _chk_n(n, 1, 17)
    nr = r1 - r0
    if not (0 <= r0 <= r1 <= (1 << n)):
        raise BornviError("stein_quadform_rows: row range out of bounds")
    _chk(K_rows, torch.float64, dev, "K_rows", nr << n)
    _chk(q, torch.float64, dev, "q", 1 << n)
    if out is None:
        out = torch.empty(nr + 1, dtype=torch.float64, device=dev)
    else:
        _chk(out, torch.float64, dev, "out", nr + 1)
    ws = _ws(dev, _cached_size(h, "bornvi_stein_quadform_workspace_bytes", n, 1), "qf")
    h.call("bornvi_stein_quadform_rows", n, _ptr(K_rows), int(r0), int(r1), _ptr(q), _ptr(out),
           C.c_void_p(out.data_ptr() + 8 * nr), _ptr(ws), ws.numel(), _ext.stream_ptr(dev))
    return out


def stein_kp_pairs(n, length_scale, zi, zj, si, sj):
    dev = si.device
    h = _ext.handle_for(dev)
    for t, dt, nm in ((zi, torch.int64, "zi"), (zj, torch.int64, "zj"), (si, torch.float64, "si"), (sj, torch.float64, "sj")):
        _chk(t, dt, dev, nm)
    M = zi.numel()
    _chk_n(n)
    if zj.numel() != M or si.numel() != M * n or sj.numel() != M * n:
        raise BornviError("stein_kp_pairs: zi, zj must be [M] and si, sj [M, n]")
    out = torch.empty(M, dtype=torch.float64, device=dev)
    h.call("bornvi_stein_kp_pairs", n, float(length_scale), M, _ptr(zi), _ptr(zj), _ptr(si), _ptr(sj), _ptr(out),
           _ext.stream_ptr(dev))
    return out


def stein_quadform(K, Q, n, want_y=True):
    """Q [B, 2^n] (or [2^n]) -> (ksd2 [B], Y [B, 2^n] or None)."""
    dev = K.device
    h = _ext.handle_for(dev)
    _chk_n(n, 1, 17)
    # K: dense [2^n, 2^n], or the [:, :2^n] view of a padded [2^n, ld] buffer (the trainer's K_p, stein_gram(ld=...))
    ld = _chk_matrix(K, 1 << n, 1 << n, dev, "K") if K.dim() == 2 else (_chk(K, torch.float64, dev, "K", 1 << (2 * n)) or (1 << n))
    if Q.numel() % (1 << n):
        raise BornviError("Q: element count is not a multiple of 2^n")
    Q2 = Q.reshape(-1, 1 << n)
    _chk(Q2, torch.float64, dev, "Q")
    B = Q2.shape[0]
    ksd2 = torch.empty(B, dtype=torch.float64, device=dev)
    Y = torch.empty_like(Q2) if want_y else None
    ws = _ws(dev, h.size("bornvi_stein_quadform_workspace_bytes", n, B), "qf")
    h.call("bornvi_stein_quadform_ld", n, _ptr(K), ld, _ptr(Q2), B, _ptr(ksd2), _ptr(Y) if want_y else None, _ptr(ws),
           ws.numel(), _ext.stream_ptr(dev))
    return ksd2, Y


def stein_sym_workspace_bytes(dev, n):
    """Workspace bytes of the symmetric contraction (stein_quadform_sym / _pairs) at this n."""
    return int(_cached_size(_ext.handle_for(dev), "bornvi_stein_quadform_sym_workspace_bytes", int(n)))


def stein_quadform_sym(K, q, n):
    """(ksd2 [1], y = K q [2^n]) for a symmetric K (as built by stein_gram): reads the upper triangle only."""
    dev = K.device
    h = _ext.handle_for(dev)
    _chk_n(n, 1, 17)
    ld = _chk_matrix(K, 1 << n, 1 << n, dev, "K")
    _chk(q, torch.float64, dev, "q", 1 << n)
    y = torch.empty(1 << n, dtype=torch.float64, device=dev)
    ksd2 = torch.empty(1, dtype=torch.float64, device=dev)
    ws = _ws(dev, _cached_size(h, "bornvi_stein_quadform_sym_workspace_bytes", n), "qfsym")
    h.call("bornvi_stein_quadform_sym_ld", n, _ptr(K), ld, _ptr(q), _ptr(ksd2), _ptr(y), _ptr(ws), ws.numel(),
           _ext.stream_ptr(dev))
    return ksd2, y


def stein_matvec_kron(S, q, n, length_scale=1.0):
    """Matrix-free (ksd2 [1], y = K_p q [2^n])."""
    dev = S.device
    h = _ext.handle_for(dev)
    _chk_n(n)
    _chk(S, torch.float64, dev, "S", n << n)
    _chk(q, torch.float64, dev, "q", 1 << n)
    y = torch.empty(1 << n, dtype=torch.float64, device=dev)
    ksd2 = torch.empty(1, dtype=torch.float64, device=dev)
    ws = _ws(dev, _cached_size(h, "bornvi_stein_matvec_kron_workspace_bytes", n), "kron")
    h.call("bornvi_stein_matvec_kron", n, float(length_scale), _ptr(S), _ptr(q), _ptr(y), _ptr(ksd2), _ptr(ws),
           ws.numel(), _ext.stream_ptr(dev))
    return ksd2, y


def ksd_grad_finish(n, shifted, n_shift, y, ksd2, want_dldq=False):
    """-> (loss [1], grad [n_shift], dLdq [2^n] or None); see bornvi_ksd_grad_finish."""
    dev = y.device
    h = _ext.handle_for(dev)
    _chk_n(n)
    _chk(y, torch.float64, dev, "y", 1 << n)
    _chk(ksd2, torch.float64, dev, "ksd2", 1)
    if n_shift:
        _chk(shifted, torch.float64, dev, "shifted", (2 * n_shift) << n)
    loss = torch.empty(1, dtype=torch.float64, device=dev)
    grad = torch.empty(n_shift, dtype=torch.float64, device=dev)
    dldq = torch.empty(1 << n, dtype=torch.float64, device=dev) if want_dldq else None
    h.call("bornvi_ksd_grad_finish", n, _ptr(shifted) if n_shift else None, int(n_shift), _ptr(y), _ptr(ksd2),
           _ptr(loss), _ptr(dldq) if want_dldq else None, _ptr(grad) if n_shift else None, _ext.stream_ptr(dev))
    return loss, grad, dldq
