"""cProfile of the host side of training_step (n=8: the device work is ~0.1 ms, the rest is host)."""
import cProfile, pstats, sys, io, contextlib, time
sys.path.insert(0, '/root/repo')
import torch
from tensornetworks_amd.bayesian_network import synthetic_network
from tensornetworks_amd.ksd_vi_quantum import KSDVariationalInference
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 4
bn, lat, obs, x = synthetic_network(n, 0)
vi = KSDVariationalInference(bn, lat, obs, n, L, pytorch_device='cuda:0')
vi._prepare_stein(x)
params, opt, sched = vi.make_optimizer(0.005, 1000)
for _ in range(20):
    vi.training_step(params, opt, sched, 10.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    vi.training_step(params, opt, sched, 10.0)
pr.disable(); torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 200 * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28); print(s.getvalue()[:6000])
