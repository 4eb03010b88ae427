"""3x3-conv channel auto-encoder baseline compressor (``opencood/models/sub_modules/naive_compress.py:5-32``)."""
import torch.nn as nn


def _cbr(cin, cout):
    return [nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1),
            nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01), nn.ReLU()]


class NaiveCompressor(nn.Module):
    def __init__(self, input_dim, compress_raito):
        super().__init__()
        mid = input_dim // compress_raito
        self.encoder = nn.Sequential(*_cbr(input_dim, mid))
        self.decoder = nn.Sequential(*_cbr(mid, input_dim), *_cbr(input_dim, input_dim))

    def forward(self, x):
        return self.decoder(self.encoder(x))
