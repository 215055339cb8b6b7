"""yaml -> attribute-dict config, compatible with what the reference's
``pcdet/config.py`` (:16-80) hands to ``MixedScaleSparseTransformer.__init__``:
attribute access, ``.get``, ``_BASE_CONFIG_`` includes.  (``easydict`` is not a
dependency.)"""
import os

import yaml


class Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(o):
        if isinstance(o, dict):
            return Config({k: Config.wrap(v) for k, v in o.items()})
        if isinstance(o, (list, tuple)):
            return [Config.wrap(v) for v in o]
        return o


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


def load_yaml(path):
    with open(path) as f:
        raw = yaml.safe_load(f)
    if '_BASE_CONFIG_' in raw:
        base_path = raw.pop('_BASE_CONFIG_')
        if not os.path.isabs(base_path):
            base_path = os.path.join(os.path.dirname(path), base_path)
        with open(base_path) as f:
            base = yaml.safe_load(f)
        _merge(base, raw)
        raw = base
    return Config.wrap(raw)


DEFAULT_CFG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfgs", "mssvt.yaml")


def build_backbone_from_cfg(cfg=None):
    """Construct the backbone the way ``Detector3DTemplate.build_backbone_3d`` does
    (ref: pcdet/models/detectors/detector3d_template.py:68-83)."""
    import numpy as np
    from .mssvt_backbone import MixedScaleSparseTransformer
    cfg = cfg if cfg is not None else load_yaml(DEFAULT_CFG)
    pcr = np.array(cfg.DATA_CONFIG.POINT_CLOUD_RANGE, dtype=np.float64)
    vs = None
    for proc in cfg.DATA_CONFIG.DATA_PROCESSOR:
        if 'VOXEL_SIZE' in proc:
            vs = proc['VOXEL_SIZE']
    grid = np.round((pcr[3:6] - pcr[0:3]) / np.array(vs)).astype(np.int64)  # ref data_processor.py:66-68
    bb = cfg.MODEL.BACKBONE_3D
    assert bb.NAME == 'MixedScaleSparseTransformer'
    return MixedScaleSparseTransformer(model_cfg=bb, input_channels=cfg.MODEL.VFE.NUM_FILTERS[-1],
                                       grid_size=grid.tolist(), voxel_size=vs,
                                       point_cloud_range=pcr.tolist())
