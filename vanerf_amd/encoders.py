"""Image encoders of the source view (run once per source frame on PyTorch-ROCm / MIOpen; not part of the hot path).

Module and parameter names follow the reference (src/utils.py:331-391 ResBlk / ResBlkEncoder, 393-453 HourGlass / DeconvReLUGroup,
455-547 HGFilterV2, 549-607 ConvBlock) so that reference checkpoints load unchanged (`state_dict` key parity is checked in
tests/test_model_interface.py).  Outputs: geo encoder -> [(B,64,32,32), (B,8,128,128)], tex encoder -> (B,8,64,64) for a 128x128
input (the 256x256 source image is average-pooled once before both encoders, configs/vanerf.json:40-41).
"""
import functools

import torch
import torch.nn as nn
import torch.nn.functional as F


def _group(ch):
    return nn.GroupNorm(min(32, ch), ch)


class ConvBlock(nn.Module):
    """Pre-activation residual block with a 1/2 + 1/4 + 1/4 channel split (src/utils.py:549-607)."""

    def __init__(self, in_planes, out_planes, norm="group"):
        super().__init__()
        mk = _group if norm == "group" else nn.BatchNorm2d
        half, quarter = out_planes // 2, out_planes // 4
        self.bn1, self.bn2, self.bn3, self.bn4 = mk(in_planes), mk(half), mk(quarter), mk(in_planes)
        # bn4 is registered both as an attribute and as downsample.0 in the reference: both key sets exist in its state_dict
        self.downsample = nn.Sequential(self.bn4, nn.ReLU(True), nn.Conv2d(in_planes, out_planes, 1, bias=False)) if in_planes != out_planes else None
        self.conv1 = nn.Conv2d(in_planes, half, 3, padding=1, bias=False)
        self.conv2 = nn.Conv2d(half, quarter, 3, padding=1, bias=False)
        self.conv3 = nn.Conv2d(quarter, quarter, 3, padding=1, bias=False)

    def forward(self, x):
        o1 = self.conv1(F.relu(self.bn1(x)))
        o2 = self.conv2(F.relu(self.bn2(o1)))
        o3 = self.conv3(F.relu(self.bn3(o2)))
        res = x if self.downsample is None else self.downsample(x)
        return torch.cat((o1, o2, o3), 1) + res


_UP2 = {}


def _up2_matrix(n, device, dtype):
    """(2n, n) matrix of F.interpolate(scale_factor=2, mode="bicubic", align_corners=True) along one axis, read off the operator itself
    (a map that is constant along the other axis stays constant: the cubic weights sum to one)."""
    key = (n, device, dtype)
    if key not in _UP2:
        with torch.no_grad():
            basis = torch.eye(n, device=device, dtype=dtype).view(1, n, n, 1).expand(1, n, n, 2)
            _UP2[key] = F.interpolate(basis, scale_factor=2, mode="bicubic", align_corners=True)[0, :, :, 0].t().contiguous()
    return _UP2[key]


def _bicubic_up2(x):
    """Bicubic x2 up-sampling (src/utils.py:433).  Under autograd the separable operator is applied as two matrix products, U_h x U_w^T: the
    backward of F.interpolate scatters with atomics (3.4 ms per call at these sizes, 12 % of a training step), the backward of a product
    is a product.  Without autograd the library call itself runs, so inference is untouched."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return F.interpolate(x, scale_factor=2, mode="bicubic", align_corners=True)
    uh = _up2_matrix(x.shape[-2], x.device, x.dtype)
    uw = _up2_matrix(x.shape[-1], x.device, x.dtype)
    return torch.matmul(torch.matmul(uh, x), uw.t())


class HourGlass(nn.Module):
    """Recursive hourglass (src/utils.py:393-441): b1_k (skip), b2_k (down), b2_plus_1 (bottom), b3_k (up), bicubic x2 up-sampling."""

    def __init__(self, depth, num_features, norm="group"):
        super().__init__()
        self.depth = depth
        for level in range(depth, 0, -1):
            self.add_module(f"b1_{level}", ConvBlock(num_features, num_features, norm))
            self.add_module(f"b2_{level}", ConvBlock(num_features, num_features, norm))
            if level == 1:
                self.add_module("b2_plus_1", ConvBlock(num_features, num_features, norm))
        for level in range(1, depth + 1):
            self.add_module(f"b3_{level}", ConvBlock(num_features, num_features, norm))

    def _run(self, level, x):
        up = self._modules[f"b1_{level}"](x)
        low = self._modules[f"b2_{level}"](F.avg_pool2d(x, 2, stride=2))
        low = self._run(level - 1, low) if level > 1 else self._modules["b2_plus_1"](low)
        low = self._modules[f"b3_{level}"](low)
        return up + _bicubic_up2(low)

    def forward(self, x):
        return self._run(self.depth, x)


class DeconvReLUGroup(nn.Module):
    def __init__(self, in_ch, out_ch, bias=False):
        super().__init__()
        self.conv = nn.ConvTranspose2d(in_ch, out_ch, kernel_size=3, stride=2, padding=1, output_padding=1, bias=bias)
        self.nl = nn.ReLU(inplace=True)
        self.norm = nn.GroupNorm(min(32, out_ch), out_ch)

    def forward(self, x):
        return self.nl(self.norm(self.conv(x)))


class HGFilterV2(nn.Module):
    """Geometry encoder (src/utils.py:455-547): stem, n_stack hourglasses, a coarse (out_ch) and a fine (8-channel, 2x up) map."""

    def __init__(self, in_ch=3, out_ch=128, n_stack=2, n_downsample=4, norm="group", hd=False, **kwargs):
        super().__init__()
        self.n_stack, self.hd = n_stack, hd
        self.nl = nn.ReLU(True)
        self.unpack1 = DeconvReLUGroup(128, 32)
        self.conv_out = nn.Conv2d(32, kwargs.get("out_ch_hd", 8), kernel_size=5, padding=2)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3)
        self.bn1 = nn.GroupNorm(32, 64) if norm == "group" else nn.BatchNorm2d(64)
        self.conv2, self.conv3, self.conv4 = ConvBlock(64, 128, norm), ConvBlock(128, 128, norm), ConvBlock(128, 256, norm)
        for i in range(n_stack):
            self.add_module(f"m{i}", HourGlass(n_downsample, 256, norm))
            self.add_module(f"top_m_{i}", ConvBlock(256, 256, norm))
            self.add_module(f"conv_last{i}", nn.Conv2d(256, 256, 1))
            self.add_module(f"bn_end{i}", nn.GroupNorm(32, 256) if norm == "group" else nn.BatchNorm2d(256))
            self.add_module(f"l{i}", nn.Conv2d(256, out_ch, 1))
            if i < n_stack - 1:
                self.add_module(f"bl{i}", nn.Conv2d(256, 256, 1))
                self.add_module(f"al{i}", nn.Conv2d(out_ch, 256, 1))

    def forward(self, x):
        x = self.conv2(self.nl(self.bn1(self.conv1(x))))
        x_hd = self.conv_out(self.unpack1(x))
        if not self.hd:
            x = F.avg_pool2d(x, 2, stride=2)
        prev = self.conv4(self.conv3(x))
        out = None
        for i in range(self.n_stack):
            ll = self._modules[f"top_m_{i}"](self._modules[f"m{i}"](prev))
            ll = F.relu(self._modules[f"bn_end{i}"](self._modules[f"conv_last{i}"](ll)), True)
            out = self._modules[f"l{i}"](ll)
            if i < self.n_stack - 1:
                prev = prev + self._modules[f"bl{i}"](ll) + self._modules[f"al{i}"](out)
        return [out, x_hd]


class ResBlk(nn.Module):
    def __init__(self, ch, norm_layer):
        super().__init__()
        self.layers = nn.Sequential(nn.ReplicationPad2d(1), nn.Conv2d(ch, ch, 3), norm_layer(ch), nn.ReLU(True),
                                    nn.ReplicationPad2d(1), nn.Conv2d(ch, ch, 3), norm_layer(ch))

    def forward(self, x):
        return x + self.layers(x)


class ResBlkEncoder(nn.Module):
    """Texture encoder (src/utils.py:348-391): 7x7 stem, strided down-sampling, residual blocks, transposed-conv up-sampling, 7x7 head."""

    def __init__(self, in_ch=3, out_ch=8, ngf=16, n_downsample=3, n_blocks=4, n_upsample=3, norm="instance"):
        super().__init__()
        norm_layer = {"instance": functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=False),
                      "batch": functools.partial(nn.BatchNorm2d, affine=True, track_running_stats=True),
                      "group": functools.partial(nn.GroupNorm, 16)}[norm]
        layers = [nn.ReplicationPad2d(3), nn.Conv2d(in_ch, ngf, 7), norm_layer(ngf), nn.ReLU(True)]
        for i in range(n_downsample):
            c = ngf * 2 ** i
            layers += [nn.Conv2d(c, 2 * c, 3, stride=2, padding=1), norm_layer(2 * c), nn.ReLU(True)]
        c = ngf * 2 ** n_downsample
        layers += [ResBlk(c, norm_layer) for _ in range(n_blocks)]
        for i in range(n_upsample):
            c = ngf * 2 ** (n_downsample - i)
            layers += [nn.ConvTranspose2d(c, c // 2, 3, stride=2, padding=1, output_padding=1), norm_layer(c // 2), nn.ReLU(True)]
        if n_upsample > 0:
            layers += [nn.ReplicationPad2d(3), nn.Conv2d(c // 2, out_ch, 7)]
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        return self.layers(x)
