"""Sparse ResUNet (FCGF-style), counterpart of the reference's model/resunet.py:25-333.

Same layers, names and data flow (so `load_state_dict` of a reference checkpoint works), written as a
table-driven module against the MinkowskiEngine-compatible surface in corsair_amd.minkowski:
  encoder : conv1 -> norm1 -> block1 | conv{2,3,4} (stride 2) -> norm -> block          (4 scales)
  decoder : conv{4,3,2}_tr (stride-2 transpose) -> norm -> block -> cat with the encoder skip
  head    : conv1_tr (1x1) -> ReLU -> final (1x1, bias) -> per-row L2 normalisation
forward returns (per-voxel feature SparseTensor [N,out], coarsest-scale SparseTensor [N8,256]).
"""
import torch

from .. import backend as B
from .. import minkowski as ME
from .common import get_norm
from .residual_block import get_block

MEF = ME.MinkowskiFunctional


class ResUNet2(ME.MinkowskiNetwork):
    NORM_TYPE = None
    BLOCK_NORM_TYPE = "BN"
    CHANNELS = [None, 32, 64, 128, 256]
    TR_CHANNELS = [None, 32, 64, 64, 128]

    def __init__(self, in_channels=3, out_channels=32, bn_momentum=0.1, normalize_feature=None,
                 conv1_kernel_size=None, D=3):
        super().__init__(D)
        C, T = self.CHANNELS, self.TR_CHANNELS
        self.normalize_feature = normalize_feature

        def conv(cin, cout, k, s, transpose=False, bias=False):
            cls = ME.MinkowskiConvolutionTranspose if transpose else ME.MinkowskiConvolution
            return cls(in_channels=cin, out_channels=cout, kernel_size=k, stride=s, dilation=1, bias=bias,
                       dimension=D)

        def stage(tag, cin, cout, k, s, transpose=False):
            setattr(self, "conv" + tag, conv(cin, cout, k, s, transpose))
            setattr(self, "norm" + tag, get_norm(self.NORM_TYPE, cout, bn_momentum=bn_momentum, D=D))
            setattr(self, "block" + tag, get_block(self.BLOCK_NORM_TYPE, cout, cout, bn_momentum=bn_momentum, D=D))

        stage("1", in_channels, C[1], conv1_kernel_size, 1)
        stage("2", C[1], C[2], 3, 2)
        stage("3", C[2], C[3], 3, 2)
        stage("4", C[3], C[4], 3, 2)
        stage("4_tr", C[4], T[4], 3, 2, transpose=True)
        stage("3_tr", C[3] + T[4], T[3], 3, 2, transpose=True)
        stage("2_tr", C[2] + T[3], T[2], 3, 2, transpose=True)
        self.conv1_tr = conv(C[1] + T[2], T[1], 1, 1)
        self.final = conv(T[1], out_channels, 1, 1, bias=True)

    def _stage(self, tag, x):
        return getattr(self, "block" + tag)(getattr(self, "norm" + tag)(getattr(self, "conv" + tag)(x)))

    def forward(self, x):
        s1 = self._stage("1", x)
        s2 = self._stage("2", MEF.relu(s1))
        s4 = self._stage("3", MEF.relu(s2))
        s8 = self._stage("4", MEF.relu(s4))
        feat = s8  # coarsest scale, input of the global embedding head
        y = ME.cat(MEF.relu(self._stage("4_tr", MEF.relu(s8))), s4)
        y = ME.cat(MEF.relu(self._stage("3_tr", y)), s2)
        y = ME.cat(MEF.relu(self._stage("2_tr", y)), s1)
        y = self.final(MEF.relu(self.conv1_tr(y)))
        if not self.normalize_feature:
            return y, feat
        unit = B.row_l2_normalize(y.F, 0.0)  # y.F / ||y.F||_2 per row, no epsilon (reference :260-262)
        wrap = lambda f, t: ME.SparseTensor(f, coordinate_map_key=t.coordinate_map_key,
                                            coordinate_manager=t.coordinate_manager)
        return wrap(unit, y), wrap(feat.F, feat)


class ResUNetBN2(ResUNet2):
    NORM_TYPE = "BN"


class ResUNetBN2B(ResUNet2):
    NORM_TYPE = "BN"
    TR_CHANNELS = [None, 64, 64, 64, 64]


class ResUNetBN2C(ResUNet2):
    NORM_TYPE = "BN"
    TR_CHANNELS = [None, 64, 64, 64, 128]


class ResUNetBN2D(ResUNet2):
    NORM_TYPE = "BN"
    TR_CHANNELS = [None, 64, 64, 128, 128]


class ResUNetBN2E(ResUNet2):
    NORM_TYPE = "BN"
    CHANNELS = [None, 128, 128, 128, 256]
    TR_CHANNELS = [None, 64, 128, 128, 128]


# instance-norm residual blocks, batch-norm elsewhere (model/resunet.py:311-333 of the reference)
class ResUNetIN2(ResUNet2):
    NORM_TYPE = "BN"
    BLOCK_NORM_TYPE = "IN"


class ResUNetIN2B(ResUNetBN2B):
    NORM_TYPE = "BN"
    BLOCK_NORM_TYPE = "IN"


class ResUNetIN2C(ResUNetBN2C):
    NORM_TYPE = "BN"
    BLOCK_NORM_TYPE = "IN"


class ResUNetIN2D(ResUNetBN2D):
    NORM_TYPE = "BN"
    BLOCK_NORM_TYPE = "IN"


class ResUNetIN2E(ResUNetBN2E):
    NORM_TYPE = "BN"
    BLOCK_NORM_TYPE = "IN"
