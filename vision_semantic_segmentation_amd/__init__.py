"""MI355X-native per-frame hot path of AutonomousVehicleLaboratory/vision_semantic_segmentation.

Public names mirror the reference: ``SemanticSegmentation`` (src/semantic_segmentation.py),
``SemanticMapping`` (src/mapping.py), ``VisionSemanticSegmentationNode``
(src/vision_semantic_segmentation_node.py), ``get_cfg_defaults`` (src/config/base_cfg.py).
Heavy imports are deferred so that ``import vision_semantic_segmentation_amd`` works without a GPU.
"""
__version__ = "0.1.0"

_LAZY = {
    "SemanticMapping": ("mapping", "SemanticMapping"),
    "SemanticSegmentation": ("semantic_segmentation", "SemanticSegmentation"),
    "VisionSemanticSegmentationNode": ("vision_semantic_segmentation_node", "VisionSemanticSegmentationNode"),
    "get_cfg_defaults": ("config", "get_cfg_defaults"),
    "camera_setup_1": ("camera", "camera_setup_1"),
    "camera_setup_6": ("camera", "camera_setup_6"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module("." + mod, __name__), attr)
    raise AttributeError(name)
