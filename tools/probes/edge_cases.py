"""Probe: edge cases of the entry points (empty batches and ranges, strides beyond P, extreme tile sizes, the smallest n):
every line must print ok or a BornviError -- never crash."""
import sys, itertools, traceback
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tensornetworks_amd import backend as be
from tensornetworks_amd.bayesian_network import pack_network, synthetic_network
dev = torch.device('cuda', 0)
def tryit(name, fn):
    try:
        r = fn()
        print('ok  ', name, '' if r is None else r, flush=True)
    except be.BornviError as e:
        print('err ', name, str(e)[:100], flush=True)
    except Exception as e:
        print('EXC ', name, type(e).__name__, str(e)[:100], flush=True)
for ansatz in ("hardware_efficient", "all_to_all", "basic"):
    for n, L in [(1, 0), (1, 1), (2, 1), (13, 1), (14, 1)]:
        P = be.num_params(ansatz, n, L)
        th = torch.zeros(P, dtype=torch.float64, device=dev)
        tryit(f'{ansatz} n{n} L{L} batch0', lambda: tuple(be.circuit_probs(ansatz, n, L, torch.zeros((0, P), dtype=torch.float64, device=dev)).shape))
        tryit(f'{ansatz} n{n} L{L} shift empty range', lambda: tuple(be.paramshift_probs(ansatz, n, L, th, P, P, include_base=False).shape))
        tryit(f'{ansatz} n{n} L{L} shift stride>P', lambda: tuple(be.paramshift_probs(ansatz, n, L, th, 0, P, include_base=True, p_stride=P + 3).shape))
        tryit(f'{ansatz} n{n} L{L} shift first>=P', lambda: tuple(be.paramshift_probs(ansatz, n, L, th, P, P, include_base=True, p_stride=4).shape))
        for kb in (4, 5, 13):
            be.set_option(dev, 'tile_bits', kb)
            tryit(f'{ansatz} n{n} L{L} tile_bits {kb}', lambda: float(be.circuit_probs(ansatz, n, L, th[None]).sum()))
        be.set_option(dev, 'tile_bits', 13)
        for opt in ('prefix_share',):
            be.set_option(dev, opt, 1)
            tryit(f'{ansatz} n{n} L{L} {opt}', lambda: float(be.paramshift_probs(ansatz, n, L, th, 0, P, include_base=True).sum()))
            be.set_option(dev, opt, 0)
        tryit(f'{ansatz} n{n} L{L} adjoint', lambda: float(be.adjoint_vjp(ansatz, n, L, th, be.adjoint_state(ansatz, n, L, th)[0], torch.ones(1 << n, dtype=torch.float64, device=dev)).sum()))
for n in (1, 2, 9, 10):
    bn, lat, obs, x = synthetic_network(n, seed=0)
    S, pxz = be.score_from_packed(pack_network(bn, lat, x), n, dev)
    q = pxz / pxz.sum()
    K = be.stein_gram(S, n, 1.0)
    tryit(f'n{n} gram empty rows', lambda: tuple(be.stein_gram(S, n, 1.0, rows=(1, 1)).shape))
    tryit(f'n{n} rows empty', lambda: tuple(be.stein_quadform_rows(K[:0], 0, 0, q, n).shape))
    tryit(f'n{n} rows last', lambda: tuple(be.stein_quadform_rows(K[-1:], (1 << n) - 1, 1 << n, q, n).shape))
    tryit(f'n{n} quadform B=3', lambda: tuple(be.stein_quadform(K, torch.stack([q, q, q]), n, want_y=True)[1].shape))
    k2_, y_ = be.stein_quadform_sym(K, q, n)
    tryit(f'n{n} finish n_shift 0', lambda: tuple(t.shape for t in be.ksd_grad_finish(n, torch.zeros((0, 1 << n), dtype=torch.float64, device=dev), 0, y_, k2_)[:2]))
    sp = be.sym_pair_shard(n, 0, 2)
    tryit(f'n{n} pair shard', lambda: sp)
    tryit(f'n{n} kron', lambda: float(be.stein_matvec_kron(S, q, n, 1.0)[0]))
print('done')
