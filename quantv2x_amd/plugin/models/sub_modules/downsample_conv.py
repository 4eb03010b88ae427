"""The "shrinker": Conv k x k + ReLU + Conv3x3 + ReLU (bias, no BN).
Mirror of ``opencood/models/sub_modules/downsample_conv.py:7-51`` (keys ``layers.{i}.double_conv.{0,2}.*``)."""
import torch.nn as nn


class DoubleConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding):
        super().__init__()
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1),
            nn.ReLU(inplace=True))

    def forward(self, x):
        return self.double_conv(x)


class DownsampleConv(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layers = nn.ModuleList()
        c_in = config['input_dim']
        # 'kernal_size' is the reference's yaml spelling
        for k, c_out, s, p in zip(config['kernal_size'], config['dim'], config['stride'], config['padding']):
            self.layers.append(DoubleConv(c_in, c_out, kernel_size=k, stride=s, padding=p))
            c_in = c_out

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x
