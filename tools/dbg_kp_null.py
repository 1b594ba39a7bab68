import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from tensornetworks_amd import backend as be
from tensornetworks_amd.bayesian_network import synthetic_network, pack_network
from oracle import stein as os_
dev = torch.device('cuda',0)
for n in (10, 13, 16):
    bn, lat, obs, x = synthetic_network(n, 0)
    S, pxz = be.score_from_packed(pack_network(bn, lat, x), n, dev)
    K = be.stein_gram(S, n, 1.0)
    post = (pxz/pxz.sum()).contiguous()
    k2, yp = be.stein_quadform(K, post, n)
    rows = torch.arange(0, 2**n, max(1,2**n//2048), device=dev)
    sc = K[rows].abs() @ post
    ratio = (yp[0][rows].abs()/sc)
    k2k, yk = be.stein_matvec_kron(S, post, n, 1.0)
    yo = os_.stein_matvec_kron(S.cpu().numpy(), post.cpu().numpy(), n)
    print(n, "ratio max %.3e"%ratio.max().item(), "|y|max %.3e"%yp.abs().max().item(), "sc max %.3e"%sc.max().item(),
          "kron |y|max %.3e"%yk.abs().max().item(), "oracle kron |y|max %.3e"%np.abs(yo).max(), "Kmax %.3e"%K[rows].abs().max().item())
    i = int(ratio.argmax()); r = int(rows[i])
    print("   worst row", r, "y", yp[0][r].item(), "sc", sc[i].item(), "torch K[r]@post", (K[r]@post).item())
