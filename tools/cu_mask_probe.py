"""Can the issue-bound circuit kernel and the HBM-bound contraction share the chip by CU masking?
Two streams made with hipExtStreamCreateWithCUMask: x/8 of the CUs of every XCD for the circuits, the rest for the
contraction.  Reports each alone on its share and both together (n = 16, L = 6: one training step's work)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensornetworks_amd import backend as be
from tensornetworks_amd.bayesian_network import synthetic_network
from tensornetworks_amd.stein_utils import score_matrix

hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.cuda.init(); torch.zeros(1, device=dev)
NCU = torch.cuda.get_device_properties(dev).multi_processor_count


def masked_stream(select):
    bits = [1 if select(i) else 0 for i in range(NCU)]
    words = (C.c_uint32 * ((NCU + 31) // 32))()
    for i, b in enumerate(bits):
        if b:
            words[i // 32] |= 1 << (i % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), len(words), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev), sum(bits)


n = 16
bn, lat, obs, x = synthetic_network(n, seed=0)
S = score_matrix(bn, x, lat, device=dev)
K = be.stein_gram(S, n, 1.0)
q = torch.rand(1 << n, dtype=torch.float64, device=dev); q /= q.sum()
th16 = torch.rand(3 * 16 * 6, dtype=torch.float64, device=dev)


def A():
    be.stein_quadform_sym(K, q, n)


def Cc():
    be.paramshift_probs("hardware_efficient", 16, 6, th16, 0, 288, include_base=True)


def timed(fns, reps=6):
    for f, s in fns:
        with torch.cuda.stream(s): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for f, s in fns:
            with torch.cuda.stream(s): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


full = torch.cuda.Stream(dev)
a0, c0 = timed([(A, full)]), timed([(Cc, full)])
print(f"all {NCU} CUs: contraction {a0:.2f} ms  circuits {c0:.2f} ms  sum {a0 + c0:.2f}", flush=True)
patterns = {
    "5/8 of every 8 (skewed)": lambda i: ((i + i // 8) % 8) < 5,
    "4/8 of every 8 (skewed)": lambda i: ((i + i // 8) % 8) < 4,
    "even bits": lambda i: i % 2 == 0,
    "low half": lambda i: i < NCU // 2,
    "alternate groups of 8": lambda i: (i // 8) % 2 == 0,
    "alternate groups of 32": lambda i: (i // 32) % 2 == 0,
    "alternate pairs": lambda i: (i // 2) % 2 == 0,
    "3/8 circuits (skewed)": lambda i: ((i + i // 8) % 8) < 3,
}
for name, sel in patterns.items():
    sc, nc = masked_stream(sel)
    sa, na = masked_stream(lambda i: not sel(i))
    be.set_option(dev, "circuit_cus", nc)
    a, c = timed([(A, sa)]), timed([(Cc, sc)])
    both = timed([(A, sa), (Cc, sc)])
    print(f"{name:28s} circuits on {nc} CUs: {c:.2f} ms alone   contraction on {na} CUs: {a:.2f} ms alone   together {both:.2f} ms "
          f"(sequential on the whole chip {a0 + c0:.2f})", flush=True)
be.set_option(dev, "circuit_cus", 0)
