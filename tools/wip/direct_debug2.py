import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from tensornetworks_amd import backend as be, _ext
import plan_emulator as pe
from oracle import circuit as oc
dev = torch.device("cuda:0")
for ans, n, L, kb in [("hardware_efficient", 16, 2, 11), ("basic", 15, 2, 12)]:
    aid = _ext.ANSATZ_IDS[ans]
    W = _ext.plan_words(aid, n, L, kb); F, offs = _ext.plan_fast_words(aid, n, L, kb)
    be.set_option(dev, "tile_bits", kb)
    th = np.random.default_rng(3).uniform(-np.pi, np.pi, (2, oc.num_params(ans, n, L)))
    ref = np.stack([oc.probs(ans, n, L, t) for t in th])
    for pi in range(int(W[3])):
        P = W[int(W[int(W[7]) + pi]):]
        nst = int(P[3]); S = P[pe.PW_STAGES:]; fl = []
        for si in range(nst):
            hdr = int(S[0]); fl.append((hdr >> 8) & 0xff); S = S[hdr >> 16:]
        be.set_option(dev, "direct_stages", 2 | ((pi + 1) << 2))
        errs = []
        for rep in range(2):
            q = be.circuit_probs(ans, n, L, torch.as_tensor(th, device=dev)).cpu().numpy()
            errs.append(float(np.abs(q - ref).max()))
        print(ans, n, "pass", pi, "pflags", int(P[0]), "lo_out", int(P[5]), "stage flags", fl, "out_tab", int(F[int(offs[pi]) + 6]) != 0, "err", errs)
