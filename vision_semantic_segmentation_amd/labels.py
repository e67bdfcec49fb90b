"""Label tables of the reference (data, not code).

PALETTE_19: reference config/config_19.json, labels[k]["color"], k = network class id -- the
table get_labels()/apply_color_map() read (mapillary_visualization.py:9-18,70-89).
LABELS / LABELS_NAMES / LABEL_COLORS: src/config/base_cfg.py:47-57 defaults.
"""
import numpy as np

PALETTE_19 = [
    [196, 196, 196], [140, 140, 200], [128, 64, 128], [244, 35, 232], [70, 70, 70],
    [220, 20, 60], [255, 0, 0], [255, 0, 100], [255, 255, 255], [70, 130, 180],
    [107, 142, 35], [100, 128, 160], [153, 153, 153], [220, 220, 0], [119, 11, 32],
    [0, 60, 100], [0, 0, 142], [0, 0, 230], [0, 0, 70],
]

PALETTE_19_NAMES = [
    "construction--barrier--curb", "construction--flat--crosswalk-plain", "construction--flat--road",
    "construction--flat--sidewalk", "construction--structure--building", "human--person",
    "human--rider--bicyclist", "human--rider--motorcyclist", "marking--general", "nature--sky",
    "nature--vegetation", "object--manhole", "object--support--pole", "object--traffic-sign--front",
    "object--vehicle--bicycle", "object--vehicle--bus", "object--vehicle--car",
    "object--vehicle--motorcycle", "object--vehicle--truck",
]

LABELS = [2, 1, 8, 10, 3]
LABELS_NAMES = ["road", "crosswalk", "lane", "vegetation", "sidewalk"]
LABEL_COLORS = [[128, 64, 128], [140, 140, 200], [255, 255, 255], [107, 142, 35], [244, 35, 232]]


def get_labels(config_json_path=None):
    """mapillary_visualization.py:9-18: list of {"name","color"} dicts.  With no path (the
    reference's DATASET_CONFIG lives on the authors' NAS) the built-in 19-class table is used."""
    if config_json_path:
        import json
        with open(config_json_path) as f:
            return json.load(f)["labels"]
    return [{"name": n, "color": list(c)} for n, c in zip(PALETTE_19_NAMES, PALETTE_19)]


def vote_lut(palette, label_colors):
    """uint32[256]: network class id -> bitmask of map classes i whose LABEL_COLORS[i] matches the
    palette colour in R and G (blue is ignored by the reference: mapping.py:419)."""
    lut = np.zeros(256, dtype=np.uint32)
    for k, col in enumerate(palette):
        for i, lc in enumerate(label_colors):
            if int(col[0]) == int(lc[0]) and int(col[1]) == int(lc[1]):
                lut[k] |= np.uint32(1 << i)
    return lut
