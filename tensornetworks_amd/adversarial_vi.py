"""Adversarial (KL) variational inference with the quantum Born machine -- SURVEY.md section 8(f) row 1,
BASELINE config 5 (n = 12 qubits, REINFORCE batch 65,536, classifier forward/backward on one MI355X).

Mirror of the reference trainer (adversarial_vi.py:12-270): same constructor / `train` signatures, history
keys, update rules and messages.  The reference wires it to the classical Born machine only
(adversarial_vi.py:28); here the variational family is `QuantumBornMachine`, whose `sample` /
`get_log_q_z_x` surface is the one the reference exposes for this purpose (quantum_born_machine.py:153-201).

What moved to the device:
  * p(z) and p(x, z) for all 2^n states come from the score kernel (`bornvi_score_from_cpts`) instead of
    2^n * 2^m Python enumerations (adversarial_vi.py:37-47) and one enumeration PER SAMPLE per step (:60-102);
    log p(x|z) becomes a [2^n] table gathered by state index;
  * sampling = `torch.multinomial` on the GPU + shift/mask bit unpacking (no per-sample tuple lookups);
  * log q(z) = index gather from q_theta (no O(2^n) `list.index` per sample, born_machine / :194);
  * d loss / d theta goes through the parameter-shift circuits of the HIP engine (autograd Function of
    quantum_born_machine.py), i.e. 2P batched circuits per Born step;
  * the classifier is stock torch.nn (classifier_pytorch.py).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.utils as nn_utils
import torch.optim as optim

from . import backend
from .bayesian_network import pack_network
from .classifier_pytorch import BinaryClassifierMLP
from .quantum_born_machine import QuantumBornMachine
from .utils import calculate_tvd, generate_all_binary_outcomes


class AdversarialVariationalInference:
    def __init__(self, bayesian_network, latent_vars_names, observed_vars_names,
                 born_machine_config, classifier_config, device='cpu'):
        """born_machine_config: keyword arguments of QuantumBornMachine (ansatz_layers, ansatz_type,
        conditioning_dim, device_name); `init_method` is forced to 'small_random' as the reference does
        (adversarial_vi.py:27).  classifier_config: keyword arguments of BinaryClassifierMLP."""
        self.bn = bayesian_network
        self.latent_vars_names = latent_vars_names
        self.observed_vars_names = observed_vars_names
        self.num_latent_vars = len(latent_vars_names)
        self.num_observed_vars = len(observed_vars_names)
        self.device = device

        born_machine_config = {**born_machine_config, 'init_method': 'small_random'}
        self.born_machine = QuantumBornMachine(num_latent_vars=self.num_latent_vars,
                                               **born_machine_config).to(device)

        classifier_input_dim = self.num_latent_vars
        if born_machine_config.get('conditioning_dim', 0) > 0:
            classifier_input_dim += born_machine_config['conditioning_dim']
        self.classifier = BinaryClassifierMLP(input_dim=classifier_input_dim, **classifier_config).to(device)

        # prior p(z): every non-latent node summed out (bayesian_network.py:255-306), all z in one launch
        self._cdev = backend.compute_device(device)
        n = self.num_latent_vars
        _, prior64 = backend.score_from_packed(pack_network(self.bn, list(latent_vars_names), {}), n, self._cdev)
        self._prior64 = prior64
        self.prior_z_probs = prior64.to(torch.float32).to(device)                      # reference :44-47
        if self.prior_z_probs.numel() > 0 and not torch.isclose(self.prior_z_probs.sum(), torch.tensor(1.0, device=device)):
            self.prior_z_probs = self.prior_z_probs / self.prior_z_probs.sum()
        self._shifts = torch.arange(n - 1, -1, -1, device=device)
        self._log_p_x_given_z = None
        self._log_p_key = None
        self._label_cache = None
        self._baseline = None
        self.timers = None      # optional {phase: [(start, end) events]} filled by the steps (bench.py)

    # reference attributes kept lazily (2^n Python objects)
    @property
    def prior_z_outcomes(self):
        return generate_all_binary_outcomes(self.num_latent_vars)

    @property
    def prior_z_dist_dict(self):
        return dict(zip(self.prior_z_outcomes, self._prior64.cpu().tolist()))

    def _bits(self, idx):
        return ((idx.unsqueeze(1) >> self._shifts) & 1).to(torch.float32)

    def _index(self, z):
        return (z.long() << self._shifts).sum(dim=1)

    def _sample_from_prior_z(self, num_samples):
        """reference :49-58"""
        if self.num_latent_vars == 0:
            return torch.empty(num_samples, 0, device=self.device)
        if self.prior_z_probs.numel() == 0:
            raise ValueError("Prior distribution p(z) is not properly initialized.")
        return self._bits(torch.multinomial(self.prior_z_probs, num_samples, replacement=True))

    def _log_p_table(self, x_obs_tensor):
        """log p(x_obs | z) for every z, float32, with the reference's edge rules (:91-99)."""
        key = tuple(x_obs_tensor.cpu().long().tolist())
        if self._log_p_key != key:
            x_dict = {nm: int(v) for nm, v in zip(self.observed_vars_names, key)}
            _, pxz = backend.score_from_packed(pack_network(self.bn, list(self.latent_vars_names), x_dict),
                                               self.num_latent_vars, self._cdev)
            pz = self._prior64
            ratio = (pxz / pz).to(torch.float32)
            table = torch.log(ratio + 1e-9)
            tiny = pz < 1e-9
            table = torch.where(tiny & (pxz > 1e-9), torch.full_like(table, float('inf')), table)
            table = torch.where(tiny & ~(pxz > 1e-9), torch.full_like(table, float('-inf')), table)
            self._log_p_x_given_z = table.to(self.device)
            self._log_p_key = key
        return self._log_p_x_given_z

    def _get_log_p_x_given_z(self, x_obs_tensor, z_samples_tensor):
        """reference :60-102 -- a table gather instead of one network enumeration per sample."""
        return self._log_p_table(x_obs_tensor)[self._index(z_samples_tensor)]

    # ---- one classifier step / one Born-machine step, device-resident (no host read-back inside) ----------------------
    def _spans(self, name):
        from .ksd_vi_quantum import _EventSpan
        return _EventSpan(self.timers if torch.device(self.device).type == "cuda" else None, name)

    def _clf_inputs(self, z, x_obs_tensor, with_x):
        if with_x:
            return torch.cat((z, x_obs_tensor.unsqueeze(0).expand(z.shape[0], -1)), dim=1)
        return z

    def _classifier_step(self, batch_size, x_obs_tensor, with_x, optimizer_classifier, criterion, clip):
        """reference :151-181: BCE on Born samples (label 1) against prior samples (label 0); one q_theta evaluation."""
        optimizer_classifier.zero_grad()
        with self._spans("sample"):
            with torch.no_grad():
                q = self.born_machine.get_probabilities().detach()
                z_from_born = self._bits(torch.multinomial(q / q.sum(), batch_size, replacement=True).to(self.device))
            z_from_prior = self._sample_from_prior_z(batch_size)
        with self._spans("classifier"):
            all_inputs = torch.cat((self._clf_inputs(z_from_born, x_obs_tensor, with_x),
                                    self._clf_inputs(z_from_prior, x_obs_tensor, with_x)), dim=0)
            labels = self._labels(batch_size)
            loss_d = criterion(self.classifier(all_inputs), labels)
            loss_d.backward()
            grad_norm_d = nn_utils.clip_grad_norm_(self.classifier.parameters(), clip)
            optimizer_classifier.step()
        return loss_d.detach(), grad_norm_d

    def _labels(self, batch_size):
        if self._label_cache is None or self._label_cache.shape[0] != 2 * batch_size:
            self._label_cache = torch.cat((torch.ones(batch_size, 1, device=self.device),
                                           torch.zeros(batch_size, 1, device=self.device)), dim=0)
        return self._label_cache

    @staticmethod
    def _reinforce_reward(logit_d, log_p, baseline, first, baseline_decay):
        """reference :202-215: raw reward = classifier logit - log p(x|z); the running baseline (a 0-dim tensor, updated
        IN PLACE: the tensor a captured epoch reads and writes) is the first epoch's mean reward, then an exponential
        average; returns the reward with the updated baseline applied."""
        raw_reward = logit_d - log_p
        mean_reward = raw_reward.mean()
        if first:
            baseline.copy_(mean_reward)
        else:
            baseline.mul_(baseline_decay).add_(mean_reward, alpha=1 - baseline_decay)
        return raw_reward - baseline

    @staticmethod
    def _reinforce_loss(log_q, reinforce_reward):
        """reference :217-222: loss_q = mean(log q(z) * reward - entropy bonus), entropy bonus = -0.01 log q(z)."""
        return (log_q * reinforce_reward.detach() - (-0.01 * log_q)).mean()

    def _born_step(self, batch_size, x_obs_tensor, with_x, optimizer_born, clip, baseline_decay, first):
        """reference :184-231: REINFORCE with a running baseline and an entropy bonus.  ONE differentiable q_theta
        evaluation serves the sampling and log q (the reference runs the circuit twice); the baseline is a device scalar;
        a NaN / Inf loss skips the update on the device (the fused optimiser's found_inf) -- nothing is read back.
        Returns (loss [0-dim], grad norm [0-dim], finite flag [0-dim bool])."""
        optimizer_born.zero_grad()
        theta = self.born_machine.theta
        with self._spans("born_forward"):
            q = self.born_machine.get_probabilities()                 # float64 [2^n], differentiable (parameter shift)
            with torch.no_grad():
                qd = q.detach()
                idx = torch.multinomial(qd / qd.sum(), batch_size, replacement=True)
                z_q = self._bits(idx.to(self.device))
                logit_d = self.classifier(self._clf_inputs(z_q, x_obs_tensor, with_x)).squeeze()
                log_p = self._log_p_active[idx.to(self.device)]       # (the table of this observation, built by train())
                reinforce_reward = self._reinforce_reward(logit_d, log_p, self._baseline, first, baseline_decay)
            log_q = torch.log(q.clamp(min=1e-9))[idx].to(self.device)
            loss_q = self._reinforce_loss(log_q, reinforce_reward)
        with self._spans("born_backward"):
            finite = torch.isfinite(loss_q.detach())
            fused_ok = theta.is_cuda and optimizer_born.defaults.get("fused")
            if fused_ok:
                loss_q.backward()                                     # 2P parameter-shift circuits on the HIP engine
                grad_norm_q = nn_utils.clip_grad_norm_(self.born_machine.parameters(), clip)
                self._found_inf.copy_((~finite).to(torch.float32))
                optimizer_born.found_inf = self._found_inf                 # skip the update on the device
                try:
                    optimizer_born.step()
                finally:
                    del optimizer_born.found_inf
            elif bool(finite):
                loss_q.backward()
                grad_norm_q = nn_utils.clip_grad_norm_(self.born_machine.parameters(), clip)
                optimizer_born.step()
            else:
                grad_norm_q = None
        return loss_q.detach(), grad_norm_q, finite

    def train(self, x_observation_dict, num_epochs, batch_size, lr_born_machine, lr_classifier,
              k_classifier_steps=1, k_born_steps=1, verbose=True, true_posterior_for_tvd=None,
              use_lr_scheduler=True, gradient_clip_norm=10.0, baseline_decay=0.99,
              optimizer_type="adam", adam_betas=(0.9, 0.999), *, graph_epochs=None):
        """Same signature, history keys and messages as the reference (adversarial_vi.py:104-270).  The epoch runs
        without host read-backs: losses and norms stay on the device and are fetched at the points where the reference
        prints (every num_epochs // 20 epochs) and once at the end; the "NaN or Inf" warning is therefore printed at
        the next such point instead of inside the step.
        graph_epochs (keyword-only extra; None = automatic: on the GPU with Adam, no event timers): after two eager
        epochs the whole epoch body -- sampling, classifier forward / backward / Adam, the Born step with its 2P
        parameter-shift circuits, the guard and its Adam step -- is captured ONCE into a HIP graph and replayed per epoch
        (an epoch is ~150 small launches; at n = 12 it is bound by their host work).  Same operations in the same order;
        the learning-rate schedulers advance on the host and fill the rate tensors the captured Adam kernels read.  If
        the capture fails the epochs simply go on eagerly."""
        if self.num_observed_vars > 0 and set(x_observation_dict.keys()) != set(self.observed_vars_names):
            raise ValueError("Keys in x_observation_dict must match self.observed_vars_names.")

        x_obs_list = [x_observation_dict[name] for name in self.observed_vars_names] if self.num_observed_vars > 0 else []
        x_obs_tensor = torch.tensor(x_obs_list, dtype=torch.float32, device=self.device)

        if self.born_machine.conditioning_dim > 0:
            if self.num_observed_vars == 0:
                raise ValueError("Born machine is conditional but no observed vars specified.")
            if self.born_machine.conditioning_dim != self.num_observed_vars:
                raise ValueError("Born machine conditioning_dim must match num_observed_vars if used.")

        on_gpu = torch.device(self.device).type == "cuda"
        fused = {"fused": True} if on_gpu else {}
        want_graph = (graph_epochs if graph_epochs is not None else True) and on_gpu and optimizer_type == "adam" \
            and self.timers is None and num_epochs > 3
        if optimizer_type == "adam":
            lr_b, lr_c = lr_born_machine, lr_classifier
            if want_graph:            # capturable Adam: step counter and learning rate live on the device
                fused = {**fused, "capturable": True}
                lr_b = torch.tensor(float(lr_born_machine), dtype=torch.float32, device=self.device)
                lr_c = torch.tensor(float(lr_classifier), dtype=torch.float32, device=self.device)
            optimizer_born = optim.Adam(self.born_machine.parameters(), lr=lr_b, betas=adam_betas, **fused)
            optimizer_classifier = optim.Adam(self.classifier.parameters(), lr=lr_c, betas=adam_betas, **fused)
        else:
            optimizer_born = optim.SGD(self.born_machine.parameters(), lr=lr_born_machine, momentum=0.9, **fused)
            optimizer_classifier = optim.SGD(self.classifier.parameters(), lr=lr_classifier, momentum=0.9, **fused)

        scheduler_born = scheduler_classifier = None
        if use_lr_scheduler:
            scheduler_born = optim.lr_scheduler.CosineAnnealingLR(optimizer_born, T_max=num_epochs, eta_min=lr_born_machine / 10)
            scheduler_classifier = optim.lr_scheduler.CosineAnnealingLR(optimizer_classifier, T_max=num_epochs, eta_min=lr_classifier / 10)

        criterion_classifier = nn.BCEWithLogitsLoss()
        self._baseline = torch.zeros((), device=self.device)
        self._found_inf = torch.zeros((), dtype=torch.float32, device=self.device)
        self._last_good_norm = torch.zeros((), dtype=torch.float32, device=self.device)
        self._label_cache = None
        self._log_p_active = self._log_p_table(x_obs_tensor)   # built once, outside the epochs (its key is a host read-back)
        self.graph_error = None
        dev_hist = {'loss_classifier': [], 'loss_born_machine': [], 'grad_norm_born': [], 'grad_norm_classifier': []}
        tvds = []
        best_tvd = float('inf')
        best_born_params = best_classifier_params = None
        in_features = self.classifier.network[0].in_features
        with_x = in_features == self.num_latent_vars + self.num_observed_vars and self.num_observed_vars > 0
        if not with_x and in_features != self.num_latent_vars:
            raise ValueError("Classifier input dimension mismatch.")
        nan_t = torch.full((), float('nan'), device=self.device)
        zero_t = torch.zeros((), device=self.device)
        loss_d = grad_norm_d = None
        skipped_seen = 0
        skipped = torch.zeros((), dtype=torch.int64, device=self.device)
        tvd_on_device = torch.is_tensor(true_posterior_for_tvd)

        def epoch_body(first):
            """(loss_d, grad_norm_d, loss_q [NaN where the update was skipped], grad_norm_q, number of skipped updates)"""
            loss_d = grad_norm_d = None
            for _ in range(k_classifier_steps):
                loss_d, grad_norm_d = self._classifier_step(batch_size, x_obs_tensor, with_x, optimizer_classifier,
                                                            criterion_classifier, gradient_clip_norm)
            loss_q = finite = None
            n_skip = torch.zeros((), dtype=torch.int64, device=self.device)
            for _ in range(k_born_steps):
                loss_q, gn, finite = self._born_step(batch_size, x_obs_tensor, with_x, optimizer_born, gradient_clip_norm,
                                                     baseline_decay, first=first)
                # history['grad_norm_born'] = the norm of the last update that WAS applied (the reference's grad_norm_q
                # survives skipped steps and epochs, 0.0 before the first good one, adversarial_vi.py:224-231): kept in a
                # device scalar, updated in place (a captured epoch reads and writes the same tensor)
                if gn is not None:
                    self._last_good_norm.copy_(torch.where(finite, gn.detach().to(self._last_good_norm.dtype), self._last_good_norm))
                n_skip = n_skip + (~finite).to(torch.int64)
            return (loss_d if loss_d is not None else nan_t,
                    grad_norm_d.detach() if grad_norm_d is not None else zero_t,
                    torch.where(finite, loss_q, nan_t.to(loss_q.dtype)) if loss_q is not None else nan_t,
                    self._last_good_norm.clone(), n_skip)

        graph = graph_out = side = None
        self.graphed_epochs = 0             # epochs of this call replayed from the graph
        for epoch in range(num_epochs):
            if graph is not None:
                graph.replay()
                out = tuple(t.clone() for t in graph_out)           # the graph owns its outputs: keep copies
                self.graphed_epochs += 1
            elif want_graph and epoch in (1, 2):
                try:
                    dev_t = torch.device(self.device)
                    if epoch == 1:          # an eager epoch on the capture stream: its workspaces exist before the capture
                        side = torch.cuda.Stream(device=dev_t)
                        side.wait_stream(torch.cuda.current_stream(dev_t))
                        with torch.cuda.stream(side):
                            out = epoch_body(False)
                        torch.cuda.current_stream(dev_t).wait_stream(side)
                    else:                   # capture the epoch body once, then run this epoch from the graph
                        torch.cuda.synchronize(dev_t)
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=side):
                            graph_out = epoch_body(False)
                        g.replay()
                        out = tuple(t.clone() for t in graph_out)
                        graph = g
                        self.graphed_epochs += 1
                except Exception as e:      # not capturable on this set-up: the epochs go on eagerly
                    want_graph = False
                    graph = None
                    self.graph_error = f"{type(e).__name__}: {e}"
                    torch.cuda.synchronize()
                    out = epoch_body(False)
            else:
                out = epoch_body(first=(epoch == 0))
            dev_hist['loss_classifier'].append(out[0])
            dev_hist['grad_norm_classifier'].append(out[1])
            dev_hist['loss_born_machine'].append(out[2])
            dev_hist['grad_norm_born'].append(out[3])
            skipped = skipped + out[4]

            if scheduler_born is not None:
                scheduler_born.step()
            if scheduler_classifier is not None:
                scheduler_classifier.step()

            if true_posterior_for_tvd is not None:
                if tvd_on_device:                     # array form (stein_utils.true_posterior_table): stays on the device
                    from .stein_utils import tvd_table
                    q_now = self.born_machine.get_probabilities().detach()
                    tvd = float(tvd_table(true_posterior_for_tvd.to(q_now.device), q_now))
                else:
                    tvd = calculate_tvd(true_posterior_for_tvd, self.born_machine.get_prob_dict())
                tvds.append(tvd)
                if tvd < best_tvd:
                    best_tvd = tvd
                    best_born_params = self.born_machine.state_dict()
                    best_classifier_params = self.classifier.state_dict()
            else:
                tvds.append(np.nan)

            if verbose and (epoch % max(1, num_epochs // 20) == 0 or epoch == num_epochs - 1):
                n_skipped = int(skipped)                                   # (this epoch's one read-back)
                for _ in range(n_skipped - skipped_seen):
                    print(f"Warning: NaN or Inf encountered in Born machine loss. Skipping update.")
                skipped_seen = n_skipped
                log_msg = f"Epoch {epoch+1}/{num_epochs} | Loss D: {float(dev_hist['loss_classifier'][-1]):.4f} | Loss G: {float(dev_hist['loss_born_machine'][-1]):.4f}"
                if scheduler_born is not None:
                    log_msg += f" | LR_G: {scheduler_born.get_last_lr()[0]:.6f}"
                if true_posterior_for_tvd is not None and len(true_posterior_for_tvd) and not np.isnan(tvds[-1]):
                    log_msg += f" | TVD: {tvds[-1]:.4f}"
                print(log_msg)

        history = {k: (torch.stack([t.to(torch.float64).reshape(()) for t in v]).cpu().tolist() if v else []) for k, v in dev_hist.items()}
        history['tvd'] = tvds
        for _ in range(int(skipped) - skipped_seen):
            print(f"Warning: NaN or Inf encountered in Born machine loss. Skipping update.")
        if best_born_params is not None and verbose:
            print(f"\nRestoring best parameters (TVD: {best_tvd:.6f})")
            self.born_machine.load_state_dict(best_born_params)
            self.classifier.load_state_dict(best_classifier_params)
        return history
