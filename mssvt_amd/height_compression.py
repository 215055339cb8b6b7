"""``HeightCompression`` (MAP_TO_BEV) on MI355X: the consumer right behind the MsSVT backbone.

Drop-in for the reference module (ref: pcdet/models/backbones_2d/map_to_bev/height_compression.py:5-50;
SURVEY.md section 8f rank 2): same constructor config keys (``NUM_BEV_FEATURES``, ``COMPRESS_LAYER_NUMS``,
``LAYER_STRIDES``, ``LAYER_DIALATIONS``, ``LAYER_PADDINGS``, ``AMP``), same ``batch_dict`` keys and the
same state-dict names (``compress_layers.{3i}`` conv, ``.{3i+1}`` batch norm), so reference checkpoints load.

What differs is where the dense grid comes from: ``SparseTensor.dense()`` of this package is ONE gather
kernel (``k_dense_bev``, csrc/dense_bev.hip) that writes ``(B, C, Z, Y, X)`` directly -- no zero fill, no
scatter, no permute copy -- and ``(B, C*Z, Y, X)`` is a view of it.  The optional 3x3 compression convs are
plain library convolutions (MIOpen through torch): they are dense, regular work outside the sparse path.
"""
import torch
import torch.nn as nn


def _cfg_get(cfg, key, default=None):
    if hasattr(cfg, "get"):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class HeightCompression(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = _cfg_get(model_cfg, "NUM_BEV_FEATURES")
        n_layers = _cfg_get(model_cfg, "COMPRESS_LAYER_NUMS", 3)
        strides = _cfg_get(model_cfg, "LAYER_STRIDES", [1, 1, 1])
        dilations = _cfg_get(model_cfg, "LAYER_DIALATIONS", [1, 1, 2])  # (sic) the reference's key
        paddings = _cfg_get(model_cfg, "LAYER_PADDINGS", [1, 1, 2])
        self.use_amp = bool(_cfg_get(model_cfg, "AMP", False))
        self.compress_layers = None
        if n_layers:
            c = self.num_bev_features
            layers = []
            for s, d, p in zip(strides[:n_layers], dilations[:n_layers], paddings[:n_layers]):
                layers += [nn.Conv2d(c, c, kernel_size=3, stride=s, padding=p, dilation=d, bias=False),
                           nn.BatchNorm2d(c), nn.ReLU(inplace=True)]
            self.compress_layers = nn.ModuleList(layers)

    def forward(self, batch_dict):
        sp = batch_dict["encoded_spconv_tensor"]
        with torch.autocast("cuda", enabled=self.use_amp and sp.features.is_cuda):
            dense = sp.dense()  # (B, C, Z, Y, X), gathered in one pass
            b, c, d, h, w = dense.shape
            bev = dense.view(b, c * d, h, w)
            if self.compress_layers is not None:
                for layer in self.compress_layers:
                    bev = layer(bev)
        batch_dict["spatial_features"] = bev.float()
        batch_dict["spatial_features_stride"] = batch_dict["encoded_spconv_tensor_stride"]
        return batch_dict
