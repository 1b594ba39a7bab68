"""Adversarial-VI path (SURVEY.md section 8(f) row 1 / BASELINE config 5) on the MI355X backend: the tables the
reference computes per sample (adversarial_vi.py:37-47, :60-102) against golden vectors captured from the
reference, the trainer's API / history, and one run at config-5 shape (n = 12 qubits, batch 65,536)."""
import contextlib
import io

import numpy as np
import pytest
import torch

from tensornetworks_amd.bayesian_network import get_sprinkler_network, synthetic_network
from conftest import golden

pytestmark = pytest.mark.gpu


def make(bn, lat, obs, device="cuda:0", layers=2, seed=0, **clf):
    from tensornetworks_amd.adversarial_vi import AdversarialVariationalInference
    torch.manual_seed(seed)
    return AdversarialVariationalInference(bn, lat, obs, born_machine_config={'ansatz_layers': layers, 'conditioning_dim': 0},
                                           classifier_config=clf, device=device)


@pytest.mark.parametrize("tag", ["sprinkler", "synthetic_n5_s0"])
def test_prior_and_log_likelihood_tables_match_reference(tag):
    g = golden("adversarial_tables.npz")
    if tag == "sprinkler":
        bn, lat, obs = get_sprinkler_network(False), ['C', 'S', 'R'], ['W']
    else:
        bn, lat, obs, _ = synthetic_network(5, 0)
    adv = make(bn, lat, obs)
    n = len(lat)
    np.testing.assert_allclose(np.array([adv.prior_z_dist_dict[z] for z in adv.prior_z_outcomes]), g[f"{tag}_prior"], rtol=1e-14)
    np.testing.assert_allclose(adv.prior_z_probs.cpu().numpy(), g[f"{tag}_prior_f32"], rtol=2e-7)
    Z = torch.tensor(adv.prior_z_outcomes, dtype=torch.float32, device="cuda:0")
    for xv in (0, 1):
        lp = adv._get_log_p_x_given_z(torch.tensor([float(xv)]), Z)
        assert lp.dtype == torch.float32 and lp.shape == (2 ** n,)
        np.testing.assert_allclose(lp.cpu().numpy(), g[f"{tag}_logp_x{xv}"], rtol=3e-6, atol=3e-7)
    shapes = [list(p.shape) + [0] * (2 - p.dim()) for p in adv.classifier.parameters()]
    assert shapes == g[f"{tag}_clf_shapes"].tolist()          # same default architecture (classifier_pytorch.py:27-41)
    s = adv._sample_from_prior_z(4000)
    assert s.shape == (4000, n) and s.dtype == torch.float32
    emp = torch.bincount(adv._index(s), minlength=2 ** n).double().cpu().numpy() / 4000
    assert np.abs(emp - g[f"{tag}_prior"]).max() < 0.05


@pytest.mark.parametrize("device", ["cpu", "cuda:0"])
def test_training_moves_towards_the_posterior(device):
    bn = get_sprinkler_network(False)
    lat, obs, x = ['C', 'S', 'R'], ['W'], {'W': 1}
    adv = make(bn, lat, obs, device=device, layers=3, seed=1)
    post, _ = bn.get_true_posterior(lat, x)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        hist = adv.train(x, num_epochs=60, batch_size=512, lr_born_machine=0.03, lr_classifier=0.01,
                         k_classifier_steps=3, k_born_steps=1, verbose=True, true_posterior_for_tvd=post)
    assert set(hist) == {'loss_classifier', 'loss_born_machine', 'tvd', 'grad_norm_born', 'grad_norm_classifier'}
    assert all(len(v) == 60 for v in hist.values()) and np.all(np.isfinite(hist['loss_born_machine']))
    assert "Loss D:" in buf.getvalue() and "Loss G:" in buf.getvalue() and "LR_G:" in buf.getvalue()
    assert min(hist['tvd']) < 0.75 * hist['tvd'][0]           # the variational family moves towards p(z|x)
    with pytest.raises(ValueError, match="Keys in x_observation_dict"):
        adv.train({'Q': 1}, 1, 8, 0.01, 0.01, verbose=False)


def test_config5_shape_runs():
    """n = 12 qubits, REINFORCE batch 65,536, classifier forward/backward -- two epochs."""
    n = 12
    # milder CPTs than the KSD benchmarks: with U(0.01, 0.99) tables some of the 4096 states have
    # p(z) < 1e-9, for which the reference's rule (adversarial_vi.py:91-96) makes the reward infinite
    bn, lat, obs, x = synthetic_network(n, 0, p_low=0.25, p_high=0.75)
    adv = make(bn, lat, obs, layers=4)
    with contextlib.redirect_stdout(io.StringIO()):
        hist = adv.train(x, num_epochs=2, batch_size=65536, lr_born_machine=0.003, lr_classifier=0.03,
                         k_classifier_steps=5, k_born_steps=1, verbose=False, adam_betas=(0.5, 0.999))
    assert np.all(np.isfinite(hist['loss_classifier'])) and np.all(np.isfinite(hist['loss_born_machine']))
    assert hist['grad_norm_born'][-1] > 0
    g = adv.born_machine.theta.grad
    assert g is not None and g.shape == (144,) and torch.isfinite(g).all()


def test_graphed_epochs_run_the_same_training():
    """train(graph_epochs=True): after two eager epochs the epoch body is captured into one HIP graph and replayed.  The
    sampling uses the same generator stream, so with the same seed the graphed and the eager run see the same samples:
    histories equal to rounding, parameters equal to rounding; and the graph really was used."""
    import torch
    from tensornetworks_amd.adversarial_vi import AdversarialVariationalInference
    from tensornetworks_amd.bayesian_network import synthetic_network
    n = 6
    dev = torch.device("cuda", 0)
    bn, lat, obs, x = synthetic_network(n, 3, p_low=0.25, p_high=0.75)
    runs = {}
    for mode in (False, True):
        torch.manual_seed(11)
        adv = AdversarialVariationalInference(bn, lat, obs, born_machine_config={'ansatz_layers': 2, 'conditioning_dim': 0},
                                              classifier_config={}, device=str(dev))
        h = adv.train(x, num_epochs=8, batch_size=4096, lr_born_machine=0.01, lr_classifier=0.02, verbose=False,
                      graph_epochs=mode)
        runs[mode] = (h, adv.born_machine.theta.detach().cpu().numpy().copy(), adv.graphed_epochs, adv.graph_error)
    assert runs[False][2] == 0
    assert runs[True][3] is None and runs[True][2] == 6, runs[True][2:]
    for key in ("loss_classifier", "loss_born_machine", "grad_norm_born", "grad_norm_classifier"):
        assert len(runs[True][0][key]) == 8 and np.all(np.isfinite(runs[True][0][key]))
    # the first two epochs are the same eager code with the same random numbers
    np.testing.assert_allclose(runs[True][0]["loss_classifier"][:2], runs[False][0]["loss_classifier"][:2], rtol=1e-6)
    np.testing.assert_allclose(runs[True][0]["loss_born_machine"][:2], runs[False][0]["loss_born_machine"][:2], rtol=1e-5, atol=1e-7)


def test_classifier_batch_split_gradient_equals_the_plain_layers():
    """BinaryClassifierMLP on a large GPU batch routes its Linear layers through a weight gradient reduced over the batch
    in chunks (one library GEMM with K = 131,072 used one or two workgroups): same logits, same gradients to fp32
    rounding as the plain nn.Sequential, same parameters / state_dict."""
    from tensornetworks_amd.classifier_pytorch import BinaryClassifierMLP
    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    clf = BinaryClassifierMLP(12).to(dev)
    x = torch.randn(16384, 12, device=dev)
    y = (torch.rand(16384, 1, device=dev) > 0.5).float()
    crit = torch.nn.BCEWithLogitsLoss()
    out_a = clf(x)                                         # chunked path (batch >= 8192, grad enabled)
    crit(out_a, y).backward()
    ga = [p.grad.clone() for p in clf.parameters()]
    clf.zero_grad()
    out_b = clf.network(x)                                 # the plain layers
    crit(out_b, y).backward()
    gb = [p.grad.clone() for p in clf.parameters()]
    assert torch.allclose(out_a, out_b, rtol=1e-5, atol=1e-6)
    for a, b in zip(ga, gb):
        assert torch.allclose(a, b, rtol=2e-4, atol=1e-6), float((a - b).abs().max())
    assert list(clf.state_dict().keys()) == [f"network.{i}.{k}" for i in (0, 2, 4) for k in ("weight", "bias")]


def test_skipped_born_updates_keep_the_last_applied_norm():
    """A NaN / Inf Born loss skips that update (reference adversarial_vi.py:224-231): history['loss_born_machine'] holds
    NaN for the epoch and history['grad_norm_born'] the norm of the last update that WAS applied -- 0.0 before the first.
    The loss of two epochs is made NaN (the running baseline stays finite)."""
    import torch
    bn, lat, obs, x = synthetic_network(5, 3, p_low=0.25, p_high=0.75)
    torch.manual_seed(3)
    adv = make(bn, lat, obs, layers=2)
    orig = type(adv)._reinforce_loss
    calls = {"n": 0}

    def poisoned(log_q, reward):            # epochs 0 and 3: a NaN loss (the baseline itself stays finite)
        bad = calls["n"] in (0, 3)
        calls["n"] += 1
        out = orig(log_q, reward)
        return out * float('nan') if bad else out

    type(adv)._reinforce_loss = staticmethod(poisoned)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            hist = adv.train(x, num_epochs=5, batch_size=256, lr_born_machine=0.01, lr_classifier=0.01, verbose=False,
                             graph_epochs=False)
    finally:
        type(adv)._reinforce_loss = staticmethod(orig)
    lb, gn = hist['loss_born_machine'], hist['grad_norm_born']
    assert np.isnan(lb[0]) and np.isnan(lb[3]) and np.all(np.isfinite([lb[1], lb[2], lb[4]]))
    assert gn[0] == 0.0 and gn[1] > 0 and gn[2] > 0 and gn[3] == gn[2] and np.isfinite(gn[4]) and gn[4] > 0
    assert torch.isfinite(adv.born_machine.theta).all()
