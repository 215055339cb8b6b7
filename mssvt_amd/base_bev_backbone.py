"""``BaseBEVBackbone`` -- the 2-D BEV backbone behind HeightCompression.

Drop-in for pcdet/models/backbones_2d/base_bev_backbone.py:6-114: constructor ``(model_cfg, input_channels)``, the
config keys LAYER_NUMS / LAYER_STRIDES / NUM_FILTERS / UPSAMPLE_STRIDES / NUM_UPSAMPLE_FILTERS, ``num_bev_features``,
``forward(data_dict)`` reading ``spatial_features`` and writing ``spatial_features_2d`` (+ ``spatial_features_{s}x``),
and the state-dict keys ``blocks.{i}.{k}.*`` / ``deblocks.{i}.{k}.*`` (a reference checkpoint loads by key).  Dense
convolutions run through the library (MIOpen) -- they are not on the sparse hot path."""
import torch
from torch import nn


def _get(cfg, key, default=None):
    return cfg.get(key, default) if hasattr(cfg, "get") else getattr(cfg, key, default)


class BaseBEVBackbone(nn.Module):
    def __init__(self, model_cfg, input_channels):
        super().__init__()
        self.model_cfg = model_cfg
        layer_nums = list(_get(model_cfg, "LAYER_NUMS", None) or [])
        layer_strides = list(_get(model_cfg, "LAYER_STRIDES", None) or [])
        num_filters = list(_get(model_cfg, "NUM_FILTERS", None) or [])
        assert len(layer_nums) == len(layer_strides) == len(num_filters)
        up_strides = list(_get(model_cfg, "UPSAMPLE_STRIDES", None) or [])
        up_filters = list(_get(model_cfg, "NUM_UPSAMPLE_FILTERS", None) or [])
        assert len(up_strides) == len(up_filters)
        bn = lambda c: nn.BatchNorm2d(c, eps=1e-3, momentum=0.01)  # noqa: E731
        c_in = [input_channels] + num_filters[:-1]
        self.blocks, self.deblocks = nn.ModuleList(), nn.ModuleList()
        for i, (n_layers, stride, width) in enumerate(zip(layer_nums, layer_strides, num_filters)):
            layers = [nn.ZeroPad2d(1), nn.Conv2d(c_in[i], width, kernel_size=3, stride=stride, padding=0, bias=False),
                      bn(width), nn.ReLU()]
            for _ in range(n_layers):
                layers += [nn.Conv2d(width, width, kernel_size=3, padding=1, bias=False), bn(width), nn.ReLU()]
            self.blocks.append(nn.Sequential(*layers))
            if up_strides:
                s = up_strides[i]
                if s >= 1:
                    up = nn.ConvTranspose2d(width, up_filters[i], s, stride=s, bias=False)
                else:  # a fractional stride downsamples (ref :59-69)
                    k = int(round(1.0 / s))
                    up = nn.Conv2d(width, up_filters[i], k, stride=k, bias=False)
                self.deblocks.append(nn.Sequential(up, bn(up_filters[i]), nn.ReLU()))
        c_out = sum(up_filters)
        if len(up_strides) > len(layer_nums):
            self.deblocks.append(nn.Sequential(
                nn.ConvTranspose2d(c_out, c_out, up_strides[-1], stride=up_strides[-1], bias=False), bn(c_out), nn.ReLU()))
        self.num_bev_features = c_out
        self.use_amp = bool(_get(model_cfg, "AMP", False))

    def forward(self, data_dict):
        feats = data_dict["spatial_features"]
        with torch.autocast(device_type=feats.device.type, enabled=self.use_amp and feats.is_cuda):
            x, ups = feats, []
            for i, blk in enumerate(self.blocks):
                x = blk(x)
                data_dict["spatial_features_%dx" % int(feats.shape[2] / x.shape[2])] = x
                ups.append(self.deblocks[i](x) if len(self.deblocks) > 0 else x)
            if len(ups) > 1:
                x = torch.cat(ups, dim=1)
            elif len(ups) == 1:
                x = ups[0]
            if len(self.deblocks) > len(self.blocks):
                x = self.deblocks[-1](x)
        data_dict["spatial_features_2d"] = x.float()
        return data_dict
