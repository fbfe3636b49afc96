"""Global embedding head, counterpart of the used part of the reference's model/fc.py
(split_batch :23-29, conv1_chamfer :60-75, conv1_max_embedding :114-128): 1x1 sparse conv 256 -> C,
per-sample max over voxels, Linear -> BatchNorm1d -> ReLU -> Linear.  The per-sample max is one
segmented-max kernel instead of a Python loop of boolean masks; the two Linear layers are dense torch
modules exactly as in the reference (the fused engine runs them through cs_conv_fwd instead)."""
import torch.nn as nn

from .. import backend as B
from .. import minkowski as ME


def split_batch(sp):
    """Per-sample feature blocks (kept for API parity; rows of one sample are contiguous)."""
    batch = sp.C[:, 0]
    n = int(batch.max().item()) + 1
    return [sp.F[batch == i, :] for i in range(n)]


class conv1_chamfer(nn.Module):
    def __init__(self, out_channels):
        super().__init__()
        self.final = ME.MinkowskiConvolution(in_channels=256, out_channels=out_channels, kernel_size=1,
                                             stride=1, dilation=1, bias=True, dimension=3)

    def forward(self, input):
        return self.final(input)


class conv1_max_embedding(nn.Module):
    def __init__(self, conv_channels, linear1_dim, linear2_dim):
        super().__init__()
        self.final = conv1_chamfer(conv_channels)
        self.fc1 = nn.Linear(conv_channels, linear1_dim)
        self.fc2 = nn.Linear(linear1_dim, linear2_dim)
        self.bn1 = nn.BatchNorm1d(linear1_dim)
        self.relu = nn.ReLU()

    def forward(self, input):
        y = self.final(input)
        n_batch = int(y.C[:, 0].max().item()) + 1
        pooled = B.segmented_max(y.F, y.C, n_batch)
        return self.fc2(self.relu(self.bn1(self.fc1(pooled))))
