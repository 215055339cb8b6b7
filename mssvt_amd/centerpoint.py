"""Detector shell: ``CenterPoint`` over the MsSVT backbone, the way mssvt.yaml wires it.

Mirrors the reference's ``Detector3DTemplate`` (pcdet/models/detectors/detector3d_template.py:14-173: module topology
vfe -> backbone_3d -> map_to_bev_module -> backbone_2d -> dense_head, each built from its config node and registered
under that attribute name, so state-dict keys read ``vfe.*``, ``backbone_3d.backbone.{i}.*``, ``map_to_bev_module.*``,
``backbone_2d.*``, ``dense_head.*``), ``CenterPoint.forward`` / ``post_processing`` (detectors/centerpoint.py:4-50) in
eval mode, and the checkpoint loaders ``load_params_from_file`` / ``_load_state_dict`` (:330-372: update by key where
the shapes agree, report what stayed).  Training losses, the dataset classes, recall bookkeeping and the spconv weight
layout adaptation (no spconv module exists here) are out of scope."""
import contextlib
import os
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

from .base_bev_backbone import BaseBEVBackbone
from .center_head import CenterHead
from .dynamic_vfe import DynamicVFE
from .height_compression import HeightCompression
from .mssvt_backbone import MixedScaleSparseTransformer

VFE = {"DynamicVFE": DynamicVFE}
BACKBONES_3D = {"MixedScaleSparseTransformer": MixedScaleSparseTransformer}
MAP_TO_BEV = {"HeightCompression": HeightCompression}
BACKBONES_2D = {"BaseBEVBackbone": BaseBEVBackbone}
DENSE_HEADS = {"CenterHead": CenterHead}


def dataset_info(cfg, num_point_features=5):
    """What the reference's detector reads from its dataset object (detector3d_template.py:36-44), from the yaml."""
    pcr = np.array(cfg.DATA_CONFIG.POINT_CLOUD_RANGE, dtype=np.float64)
    vs = next(p["VOXEL_SIZE"] for p in cfg.DATA_CONFIG.DATA_PROCESSOR if "VOXEL_SIZE" in p)
    grid = np.round((pcr[3:6] - pcr[0:3]) / np.array(vs)).astype(np.int64)  # ref data_processor.py:66-68
    return SimpleNamespace(class_names=list(cfg.CLASS_NAMES), grid_size=grid, point_cloud_range=pcr, voxel_size=list(vs),
                           point_feature_encoder=SimpleNamespace(num_point_features=num_point_features),
                           depth_downsample_factor=None)


class CenterPoint(nn.Module):
    module_topology = ["vfe", "backbone_3d", "map_to_bev_module", "backbone_2d", "dense_head"]

    def __init__(self, model_cfg, num_class, dataset):
        super().__init__()
        self.model_cfg, self.num_class, self.dataset = model_cfg, num_class, dataset
        self.class_names = dataset.class_names
        self.register_buffer("global_step", torch.LongTensor(1).zero_())
        self.module_list = self.build_networks()

    @property
    def mode(self):
        return "TRAIN" if self.training else "TEST"

    def build_networks(self):
        ds = self.dataset
        info = dict(module_list=[], num_rawpoint_features=ds.point_feature_encoder.num_point_features,
                    num_point_features=ds.point_feature_encoder.num_point_features, grid_size=ds.grid_size,
                    point_cloud_range=ds.point_cloud_range, voxel_size=ds.voxel_size)
        for name in self.module_topology:
            module = getattr(self, "build_%s" % name)(info)
            self.add_module(name, module)
            if module is not None:
                info["module_list"].append(module)
        return info["module_list"]

    def build_vfe(self, info):
        cfg = self.model_cfg.get("VFE", None)
        if cfg is None:
            return None
        m = VFE[cfg.NAME](model_cfg=cfg, num_point_features=info["num_rawpoint_features"],
                          point_cloud_range=info["point_cloud_range"], voxel_size=info["voxel_size"],
                          grid_size=info["grid_size"])
        info["num_point_features"] = m.get_output_feature_dim()
        return m

    def build_backbone_3d(self, info):
        cfg = self.model_cfg.get("BACKBONE_3D", None)
        if cfg is None:
            return None
        m = BACKBONES_3D[cfg.NAME](model_cfg=cfg, input_channels=info["num_point_features"], grid_size=info["grid_size"],
                                   voxel_size=info["voxel_size"], point_cloud_range=info["point_cloud_range"])
        info["num_point_features"] = m.num_point_features
        return m

    def build_map_to_bev_module(self, info):
        cfg = self.model_cfg.get("MAP_TO_BEV", None)
        if cfg is None:
            return None
        m = MAP_TO_BEV[cfg.NAME](model_cfg=cfg, grid_size=info["grid_size"])
        info["num_bev_features"] = m.num_bev_features
        return m

    def build_backbone_2d(self, info):
        cfg = self.model_cfg.get("BACKBONE_2D", None)
        if cfg is None:
            return None
        m = BACKBONES_2D[cfg.NAME](model_cfg=cfg, input_channels=info["num_bev_features"])
        info["num_bev_features"] = m.num_bev_features
        return m

    def build_dense_head(self, info):
        cfg = self.model_cfg.get("DENSE_HEAD", None)
        if cfg is None:
            return None
        return DENSE_HEADS[cfg.NAME](
            model_cfg=cfg, input_channels=info["num_bev_features"],
            num_class=self.num_class if not cfg.get("CLASS_AGNOSTIC", False) else 1, class_names=self.class_names,
            grid_size=info["grid_size"], point_cloud_range=info["point_cloud_range"],
            predict_boxes_when_training=bool(self.model_cfg.get("ROI_HEAD", False)), voxel_size=info["voxel_size"])

    def forward(self, batch_dict):
        for m in self.module_list:
            batch_dict = m(batch_dict)
        if self.training:  # ref centerpoint.py:13-32: ({'loss': ...}, tb_dict, disp_dict)
            loss, tb_dict = self.dense_head.get_loss()
            return dict(loss=loss), dict(loss_rpn=loss.item(), **tb_dict), {}
        return self.post_processing(batch_dict)

    def post_processing(self, batch_dict):
        return batch_dict["final_box_dicts"], {}  # recall bookkeeping (ref :34-50) needs ground truth: not kept

    # ---- checkpoints (ref detector3d_template.py:330-372) ----------------------------------------------------
    def _load_state_dict(self, model_state_disk, *, strict=True):
        state = self.state_dict()
        update = {k: v for k, v in model_state_disk.items() if k in state and state[k].shape == v.shape}
        if strict:
            self.load_state_dict(update)
        else:
            state.update(update)
            self.load_state_dict(state)
        return state, update

    def load_params_from_file(self, filename, logger=None, to_cpu=False):
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        loc = torch.device("cpu") if to_cpu else None
        import pickle
        try:
            # tensors / plain containers only: loading a checkpoint cannot run code.  The numpy scalar types OpenPCDet
            # stores beside the weights (epoch, it, version strings) are plain data: allowed explicitly.
            safe = []
            try:
                import numpy as np
                core = getattr(np, "_core", None) or getattr(np, "core")
                safe = [core.multiarray.scalar, np.dtype]
            except Exception:  # noqa: BLE001
                safe = []
            with (torch.serialization.safe_globals(safe) if safe and hasattr(torch.serialization, "safe_globals")
                  else contextlib.nullcontext()):
                ckpt = torch.load(filename, map_location=loc, weights_only=True)
        except pickle.UnpicklingError as err:  # a global weights_only does not allow; corrupt files / I/O errors propagate as they are
            if os.environ.get("MSSVT_TRUST_CHECKPOINTS", "0") != "1":
                raise RuntimeError(
                    "%s holds pickled objects beyond tensors (%s); unpickling them can execute arbitrary code. Set "
                    "MSSVT_TRUST_CHECKPOINTS=1 to load a checkpoint you trust." % (filename, type(err).__name__)) from err
            ckpt = torch.load(filename, map_location=loc, weights_only=False)
        state, update = self._load_state_dict(ckpt["model_state"], strict=False)
        missed = [k for k in state if k not in update]
        if logger is not None:
            for k in missed:
                logger.info("Not updated weight %s: %s" % (k, str(state[k].shape)))
            logger.info("==> Done (loaded %d/%d)" % (len(update), len(state)))
        return len(update), len(state), missed


def build_detector(cfg=None):
    """The detector of a yaml config (default: mssvt_amd/cfgs/mssvt.yaml), built as the reference's build_network does
    (pcdet/models/__init__.py:10-14, detectors/__init__.py)."""
    from . import config
    cfg = cfg if cfg is not None else config.load_yaml(config.DEFAULT_CFG)
    assert cfg.MODEL.NAME == "CenterPoint"
    return CenterPoint(cfg.MODEL, len(cfg.CLASS_NAMES), dataset_info(cfg))
