"""Conv → BatchNorm → ReLU stacks used by the feature pyramid and the cost-volume U-Nets.  The
Sequential indices (0 conv, 1 norm, 2 activation) are part of the checkpoint key names
(reference networks/gdb_nerf/modules.py:5-57)."""
import torch.nn as nn

_CONV = {(2, False): nn.Conv2d, (2, True): nn.ConvTranspose2d, (3, False): nn.Conv3d, (3, True): nn.ConvTranspose3d}
_NORM = {2: nn.BatchNorm2d, 3: nn.BatchNorm3d}


def _block(dims: int, transposed: bool, cin: int, cout: int, kernel_size, **conv_kw) -> nn.Sequential:
    return nn.Sequential(_CONV[(dims, transposed)](cin, cout, kernel_size, bias=False, **conv_kw),
                         _NORM[dims](cout), nn.ReLU(inplace=True))


def conv_block2d(in_channels, out_channels, kernel_size, stride=1, padding=0, groups=1):
    return _block(2, False, in_channels, out_channels, kernel_size, stride=stride, padding=padding, groups=groups)


def deconv_block2d(in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0):
    return _block(2, True, in_channels, out_channels, kernel_size, stride=stride, padding=padding, output_padding=output_padding)


def conv_block3d(in_channels, out_channels, kernel_size, stride=1, padding=0):
    return _block(3, False, in_channels, out_channels, kernel_size, stride=stride, padding=padding)


def deconv_block3d(in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0):
    return _block(3, True, in_channels, out_channels, kernel_size, stride=stride, padding=padding, output_padding=output_padding)
