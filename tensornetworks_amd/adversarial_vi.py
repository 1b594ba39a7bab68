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

    def train(self, x_observation_dict, num_epochs, batch_size, lr_born_machine, lr_classifier,
              k_classifier_steps=1, k_born_steps=1, verbose=True, true_posterior_for_tvd=None,
              use_lr_scheduler=True, gradient_clip_norm=10.0, baseline_decay=0.99,
              optimizer_type="adam", adam_betas=(0.9, 0.999)):

        if self.num_observed_vars > 0 and set(x_observation_dict.keys()) != set(self.observed_vars_names):
            raise ValueError("Keys in x_observation_dict must match self.observed_vars_names.")

        x_obs_list = [x_observation_dict[name] for name in self.observed_vars_names] if self.num_observed_vars > 0 else []
        x_obs_tensor = torch.tensor(x_obs_list, dtype=torch.float32, device=self.device)

        born_machine_x_condition = None
        if self.born_machine.conditioning_dim > 0:
            if self.num_observed_vars == 0:
                raise ValueError("Born machine is conditional but no observed vars specified.")
            if self.born_machine.conditioning_dim != self.num_observed_vars:
                raise ValueError("Born machine conditioning_dim must match num_observed_vars if used.")
            born_machine_x_condition = x_obs_tensor

        if optimizer_type == "adam":
            optimizer_born = optim.Adam(self.born_machine.parameters(), lr=lr_born_machine, betas=adam_betas)
            optimizer_classifier = optim.Adam(self.classifier.parameters(), lr=lr_classifier, betas=adam_betas)
        else:
            optimizer_born = optim.SGD(self.born_machine.parameters(), lr=lr_born_machine, momentum=0.9)
            optimizer_classifier = optim.SGD(self.classifier.parameters(), lr=lr_classifier, momentum=0.9)

        scheduler_born = scheduler_classifier = None
        if use_lr_scheduler:
            scheduler_born = optim.lr_scheduler.CosineAnnealingLR(optimizer_born, T_max=num_epochs, eta_min=lr_born_machine / 10)
            scheduler_classifier = optim.lr_scheduler.CosineAnnealingLR(optimizer_classifier, T_max=num_epochs, eta_min=lr_classifier / 10)

        criterion_classifier = nn.BCEWithLogitsLoss()
        running_baseline = 0.0
        history = {'loss_classifier': [], 'loss_born_machine': [], 'tvd': [], 'grad_norm_born': [], 'grad_norm_classifier': []}
        best_tvd = float('inf')
        best_born_params = best_classifier_params = None
        in_features = self.classifier.network[0].in_features
        with_x = in_features == self.num_latent_vars + self.num_observed_vars and self.num_observed_vars > 0
        if not with_x and in_features != self.num_latent_vars:
            raise ValueError("Classifier input dimension mismatch.")
        loss_d = grad_norm_d = loss_q = grad_norm_q = None

        for epoch in range(num_epochs):
            # --- classifier steps (reference :151-181)
            for _ in range(k_classifier_steps):
                optimizer_classifier.zero_grad()
                z_from_born = self.born_machine.sample(batch_size, x_condition=born_machine_x_condition)
                z_from_prior = self._sample_from_prior_z(batch_size)
                if with_x:
                    x_rep = x_obs_tensor.unsqueeze(0).repeat(batch_size, 1)
                    inputs_born = torch.cat((z_from_born, x_rep), dim=1)
                    inputs_prior = torch.cat((z_from_prior, x_rep), dim=1)
                else:
                    inputs_born, inputs_prior = z_from_born, z_from_prior
                all_inputs = torch.cat((inputs_born, inputs_prior), dim=0)
                all_labels = torch.cat((torch.ones(batch_size, 1, device=self.device),
                                        torch.zeros(batch_size, 1, device=self.device)), dim=0)
                loss_d = criterion_classifier(self.classifier(all_inputs), all_labels)
                loss_d.backward()
                grad_norm_d = nn_utils.clip_grad_norm_(self.classifier.parameters(), gradient_clip_norm)
                optimizer_classifier.step()
            history['loss_classifier'].append(loss_d.item())
            history['grad_norm_classifier'].append(grad_norm_d.item())

            # --- Born-machine steps: REINFORCE with running baseline and entropy bonus (reference :184-231)
            for _ in range(k_born_steps):
                optimizer_born.zero_grad()
                z_q = self.born_machine.sample(batch_size, x_condition=born_machine_x_condition)
                if with_x:
                    clf_in = torch.cat((z_q, x_obs_tensor.unsqueeze(0).repeat(batch_size, 1)), dim=1)
                else:
                    clf_in = z_q
                logit_d = self.classifier(clf_in).squeeze()
                log_p = self._get_log_p_x_given_z(x_obs_tensor, z_q)
                raw_reward = logit_d - log_p
                mean_reward = raw_reward.detach().mean().item()
                running_baseline = mean_reward if epoch == 0 else baseline_decay * running_baseline + (1 - baseline_decay) * mean_reward
                reinforce_reward = raw_reward - running_baseline
                log_q = self.born_machine.get_log_q_z_x(z_q, born_machine_x_condition)
                entropy_bonus = -0.01 * log_q
                loss_q = (log_q * reinforce_reward.detach() - entropy_bonus).mean()
                if torch.isnan(loss_q) or torch.isinf(loss_q):
                    print(f"Warning: NaN or Inf encountered in Born machine loss. Skipping update.")
                else:
                    loss_q.backward()
                    grad_norm_q = nn_utils.clip_grad_norm_(self.born_machine.parameters(), gradient_clip_norm)
                    optimizer_born.step()
            ok = loss_q is not None and not (torch.isnan(loss_q) or torch.isinf(loss_q))
            history['loss_born_machine'].append(loss_q.item() if ok else np.nan)
            history['grad_norm_born'].append(grad_norm_q.item() if grad_norm_q is not None else 0.0)

            if scheduler_born is not None:
                scheduler_born.step()
            if scheduler_classifier is not None:
                scheduler_classifier.step()

            if true_posterior_for_tvd is not None:
                tvd = calculate_tvd(true_posterior_for_tvd, self.born_machine.get_prob_dict(x_condition=born_machine_x_condition))
                history['tvd'].append(tvd)
                if tvd < best_tvd:
                    best_tvd = tvd
                    best_born_params = self.born_machine.state_dict()
                    best_classifier_params = self.classifier.state_dict()
            else:
                history['tvd'].append(np.nan)

            if verbose and (epoch % max(1, num_epochs // 20) == 0 or epoch == num_epochs - 1):
                log_msg = f"Epoch {epoch+1}/{num_epochs} | Loss D: {loss_d.item():.4f} | Loss G: {history['loss_born_machine'][-1]:.4f}"
                if scheduler_born is not None:
                    log_msg += f" | LR_G: {scheduler_born.get_last_lr()[0]:.6f}"
                if true_posterior_for_tvd and not np.isnan(history['tvd'][-1]):
                    log_msg += f" | TVD: {history['tvd'][-1]:.4f}"
                print(log_msg)

        if best_born_params is not None and verbose:
            print(f"\nRestoring best parameters (TVD: {best_tvd:.6f})")
            self.born_machine.load_state_dict(best_born_params)
            self.classifier.load_state_dict(best_classifier_params)
        return history
