"""Same-process, same-allocation A/B of symmetric-contraction variants: every library given on the command line gets
its own handle and runs bornvi_stein_quadform_sym on the SAME K_p (placement of the 32 GiB changes the rate by 15 %)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworks_amd import backend, _ext
from tensornetworks_amd.bayesian_network import synthetic_network
from tensornetworks_amd.stein_utils import score_matrix
n = 16
N = 1 << n
dev = torch.device("cuda:0")
bn, lat, obs, x = synthetic_network(n, seed=0)
S = score_matrix(bn, x, lat, device=dev)
K = backend.stein_gram(S, n, 1.0)
q = torch.rand(N, dtype=torch.float64, device=dev); q /= q.sum()
libs = [os.path.join(os.path.dirname(_ext.__file__), "libbornvi_hip.so")] + sys.argv[1:]
for rnd in range(2):
    for path in libs:
        L = C.CDLL(path)
        L.bornvi_stein_quadform_sym_workspace_bytes.restype = C.c_size_t
        L.bornvi_stein_quadform_sym_workspace_bytes.argtypes = [C.c_void_p, C.c_int]
        L.bornvi_stein_quadform_sym.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        h = C.c_void_p()
        assert L.bornvi_create(0, C.byref(h)) == 0
        wsb = L.bornvi_stein_quadform_sym_workspace_bytes(h, n)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        y = torch.empty(N, dtype=torch.float64, device=dev)
        k2 = torch.empty(1, dtype=torch.float64, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        run = lambda: L.bornvi_stein_quadform_sym(h, n, K.data_ptr(), q.data_ptr(), k2.data_ptr(), y.data_ptr(), ws.data_ptr(), wsb, st)
        for _ in range(3):
            assert run() == 0
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
        for a, b in ev:
            a.record(); run(); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
        print(f"{os.path.basename(path):28s} {ms:.3f} ms  ksd2 {k2.item():.15e}", flush=True)
        del ws
