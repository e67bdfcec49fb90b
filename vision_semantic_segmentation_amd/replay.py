"""Offline replay driver -- counterpart of src/mapping_replay.py:146-211.

The reference loads a hickle list of ``{"pcd", "pcd_frame_id", "semantic_image", "pose"}`` dicts
(written by mapping.py:309-326) and replays project_pcd + update_map over it.  hickle is not available
offline, so frames live in one ``.npz`` each with the same field names (``SemanticMapping.save_inputs``
writes them; the pose is 7 numbers tx,ty,tz,qx,qy,qz,qw).  The per-frame work is the fused GPU path.

    python -m vision_semantic_segmentation_amd.replay FRAMES_DIR [--cfg override.yaml] [--camera 1|6] [--out map.npy]
"""
import argparse
import glob
import os

import numpy as np

from .config import get_cfg_defaults
from .mapping import SemanticMapping
from .utils.utils_ros import Pose


def load_frames(directory):
    """-> list of frame dicts, in file-name order."""
    frames = []
    for path in sorted(glob.glob(os.path.join(directory, "frame_*.npz"))):
        z = np.load(path)
        pose = z["pose"]
        frames.append({"pcd": z["pcd"], "pcd_frame_id": str(z["pcd_frame_id"]), "semantic_image": z["semantic_image"],
                       "pose": Pose.from_array(pose) if pose.size == 7 else None})
    return frames


def mapping_replay(sm, input_list, camera_calibration=None):
    """mapping_replay.py:175-192: zero the grid, then project_pcd + update_map per frame (fused here).
    Returns the grid as a NumPy array [map_height, map_width, map_depth]."""
    cam = sm.cam1 if camera_calibration is None else camera_calibration
    sm.map = np.zeros((sm.map_height, sm.map_width, sm.map_depth))
    for frame in input_list:
        sm.pcd, sm.pcd_frame_id = frame["pcd"], frame["pcd_frame_id"]
        sm.mapping(frame["semantic_image"], frame["pose"], cam)
    return sm.map


def mapping_replay_dir(directory, cfg=None, camera=1, device=None):
    """mapping_replay.py:146-172 for a directory of frame_*.npz."""
    cfg = get_cfg_defaults() if cfg is None else cfg
    sm = SemanticMapping(cfg, device=device)
    cam = sm.cam1 if camera == 1 else sm.cam6
    return sm, mapping_replay(sm, load_frames(directory), cam)


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("frames_dir")
    ap.add_argument("--cfg", default="")
    ap.add_argument("--camera", type=int, default=1, choices=[1, 6])
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    cfg = get_cfg_defaults()
    if a.cfg:
        cfg.merge_from_file(a.cfg)
    sm, grid = mapping_replay_dir(a.frames_dir, cfg, a.camera)
    print("replayed %d frames, %d cells touched" % (sm.frames_mapped, int(np.any(grid != 0, axis=2).sum())))
    if a.out:
        np.save(a.out, grid)


if __name__ == "__main__":
    main()
