"""End-of-run rendering on the GPU -- counterpart of src/renderer.py (render_bev_map :32-59,
render_bev_map_with_thresholds :131-172, apply_filter :175-189), which the reference runs once at shutdown
(src/mapping.py:332-334).  Same function names and argument meaning; inputs may be NumPy arrays (results come
back as NumPy) or CUDA tensors (results stay on the GPU), so a live map can be rendered every frame.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _prep(map_):
    is_np = not isinstance(map_, torch.Tensor)
    t = torch.from_numpy(np.ascontiguousarray(map_)).cuda() if is_np else map_.contiguous()
    if t.dtype not in (torch.float64, torch.float32):
        t = t.to(torch.float64)
    assert t.dim() == 3, "map must be [W, H, C]"
    return t, is_np, (_lib.AVL_F64 if t.dtype == torch.float64 else _lib.AVL_F32)


def _colors(label_colors, c):
    lc = np.ascontiguousarray(np.asarray(label_colors), dtype=np.uint8)
    for col in lc:
        if len(col) != 3:
            raise ValueError("Color should be an RGB value.")
    if len(lc) != c:
        raise ValueError("Each channel should have a color!")
    return (C.c_uint8 * lc.size)(*lc.ravel().tolist())


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def render_bev_map(map, label_colors):
    """renderer.py:32-59: colour of each cell = colour of its arg-max channel; all-zero cells stay black."""
    t, is_np, dt = _prep(map)
    h, w, c = t.shape
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=t.device)
    _lib.check(_lib.lib().avl_render_bev_map(C.c_void_p(t.data_ptr()), dt, h, w, c, _colors(label_colors, c),
                                             C.c_void_p(out.data_ptr()), _stream(t)), "avl_render_bev_map")
    return out.cpu().numpy() if is_np else out


def render_bev_map_with_thresholds(map, label_colors, priority=None, thresholds=(0.01, 0.01, 0.01, 0.01, 0.01)):
    """renderer.py:131-172: a label is drawn where its normalised share reaches its threshold; `priority` lists the
    labels from low to high, higher ones overwrite lower ones."""
    t, is_np, dt = _prep(map)
    h, w, c = t.shape
    if priority is not None and len(priority) != c:
        raise ValueError("Each channel should have a priority.")
    pr = (C.c_int32 * c)(*[int(p) for p in (priority if priority is not None else range(c))])
    th = (C.c_double * c)(*[float(x) for x in list(thresholds)[:c]])
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=t.device)
    _lib.check(_lib.lib().avl_render_bev_map_thresholds(C.c_void_p(t.data_ptr()), dt, h, w, c, _colors(label_colors, c), pr, th,
                                                        C.c_void_p(out.data_ptr()), _stream(t)), "avl_render_bev_map_thresholds")
    return out.cpu().numpy() if is_np else out


def apply_filter(src):
    """renderer.py:175-189: 3x3 mean filter of every channel (cv2.filter2D, reflect-101 border)."""
    t, is_np, dt = _prep(src)
    h, w, c = t.shape
    dst = torch.empty_like(t)
    _lib.check(_lib.lib().avl_grid_box_filter(C.c_void_p(t.data_ptr()), C.c_void_p(dst.data_ptr()), dt, h, w, c, _stream(t)),
               "avl_grid_box_filter")
    return dst.cpu().numpy() if is_np else dst
