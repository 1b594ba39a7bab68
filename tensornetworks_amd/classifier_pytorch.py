"""MLP discriminator of the adversarial-VI path (mirror of the reference's classifier_pytorch.py:6-57).

Stock `torch.nn` layers (rocBLAS / hipBLASLt GEMMs): nothing here is hand-written -- SURVEY.md section 8(f)
keeps the classifier on the library path.  Same constructor, `network` attribute, `forward` and `get_probs`.
"""
import torch
import torch.nn as nn


class BinaryClassifierMLP(nn.Module):
    def __init__(self, input_dim, hidden_dims=None, use_batch_norm=False):
        super().__init__()
        if hidden_dims is None:
            hidden_dims = [max(input_dim * 2, 32), max(input_dim, 16)]        # reference :27-28
        layers, width = [], input_dim
        for h in hidden_dims:
            layers.append(nn.Linear(width, h))
            if use_batch_norm:
                layers.append(nn.BatchNorm1d(h))
            layers.append(nn.ReLU())
            width = h
        layers.append(nn.Linear(width, 1))                                     # one logit
        self.network = nn.Sequential(*layers)

    def forward(self, x):
        return self.network(x)

    def get_probs(self, x):
        return torch.sigmoid(self.forward(x))
