"""MLP discriminator of the adversarial-VI path (mirror of the reference's classifier_pytorch.py:6-57).

Stock `torch.nn` layers (rocBLAS / hipBLASLt GEMMs): nothing here is hand-written -- SURVEY.md section 8(f)
keeps the classifier on the library path.  Same constructor, `network` attribute, `forward` and `get_probs`.
"""
import torch
import torch.nn as nn


class _BatchSplitLinear(torch.autograd.Function):
    """y = x W^T + b like nn.Linear, with the weight gradient reduced over the batch in chunks.  At the adversarial-VI
    batch (2 x 65,536 samples, layers 12 -> 32 -> 16 -> 1) the library's weight-gradient GEMM has M x N of a few hundred
    elements and K = 131,072: one or two workgroups walk the whole batch (335 + 180 + 110 us per epoch in the rocprofv3
    trace of BASELINE config 5, 40 % of the epoch).  The same product as a batched GEMM over 256 chunks of the batch plus
    a sum uses the whole chip.  Same library (rocBLAS through torch.bmm), fp32, another summation order."""

    CHUNKS = 256

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        B = x.shape[0]
        C = _BatchSplitLinear.CHUNKS
        grad_x = grad_out @ weight if ctx.needs_input_grad[0] else None
        grad_w = None
        if ctx.needs_input_grad[1]:
            if B % C == 0:
                go = grad_out.reshape(C, B // C, -1)
                grad_w = torch.bmm(go.transpose(1, 2), x.reshape(C, B // C, -1)).sum(dim=0)
            else:
                grad_w = grad_out.t() @ x
        grad_b = grad_out.sum(dim=0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return grad_x, grad_w, grad_b


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
        # (large batches on the GPU: the same layers with the batch-chunked weight gradient above)
        if x.is_cuda and x.dim() == 2 and x.shape[0] >= 8192 and torch.is_grad_enabled():
            for layer in self.network:
                x = _BatchSplitLinear.apply(x, layer.weight, layer.bias) if isinstance(layer, nn.Linear) else layer(x)
            return x
        return self.network(x)

    def get_probs(self, x):
        return torch.sigmoid(self.forward(x))
