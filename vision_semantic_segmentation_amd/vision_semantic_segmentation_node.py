"""VisionSemanticSegmentationNode -- the reference's segmentation node
(src/vision_semantic_segmentation_node.py:41-136) around the HIP segmentation stack.

image_callback keeps the reference's sequence (:74-136): BGR->RGB, optional INTER_AREA downscale,
``SemanticSegmentation.segmentation``, uint8 cast, INTER_NEAREST upscale to the input size, palette
colouring, publish.  The upscale + colouring run as one HIP kernel (avl_colorize_labels); camera
undistortion (:84-87, cv2.undistort) is listed as the next row of SURVEY section 8f and is a hook.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .camera import camera_setup_1, camera_setup_6
from .labels import get_labels


def _palette_host(labels):
    pal = np.zeros((256, 3), dtype=np.uint8)
    for k, lab in enumerate(labels[:256]):
        pal[k] = lab["color"]
    return (C.c_uint8 * 768)(*pal.ravel().tolist())


def colorize_labels_device(labels_small, out_h, out_w, labels=None, stream=None):
    """uint8 CUDA tensor [h,w] of class ids -> uint8 CUDA tensor [out_h,out_w,3]:
    cv2.resize(INTER_NEAREST) (:109-110) followed by apply_color_map
    (mapillary_visualization.py:70-89), fused."""
    assert labels_small.is_cuda and labels_small.dtype == torch.uint8 and labels_small.dim() == 2
    labels_small = labels_small.contiguous()
    out = torch.empty((out_h, out_w, 3), dtype=torch.uint8, device=labels_small.device)
    pal = _palette_host(labels if labels is not None else get_labels())
    s = torch.cuda.current_stream(labels_small.device).cuda_stream if stream is None else stream
    rc = _lib.lib().avl_colorize_labels(C.c_void_p(labels_small.data_ptr()), int(labels_small.shape[1]),
                                        int(labels_small.shape[0]), pal, C.c_void_p(out.data_ptr()), int(out_w), int(out_h),
                                        C.c_void_p(s))
    _lib.check(rc, "avl_colorize_labels")
    return out


class VisionSemanticSegmentationNode(object):
    """Reference class: src/vision_semantic_segmentation_node.py:41."""

    def __init__(self, cfg, seg=None, use_ros=False, publish=None, undistort=None):
        if cfg.VISION_SEM_SEG.IMAGE_SCALE < 0 or cfg.VISION_SEM_SEG.IMAGE_SCALE > 1:
            raise ValueError("image scale should be in the range of [0, 1]")       # :43-44
        network_cfg = cfg.VISION_SEM_SEG.SEM_SEG_NETWORK
        if seg is None:
            from .semantic_segmentation import SemanticSegmentation
            seg = SemanticSegmentation(network_cfg)
        self.seg = seg
        self.seg_color_ref = get_labels(network_cfg.DATASET_CONFIG)                # :63
        self.cam6 = camera_setup_6()
        self.cam1 = camera_setup_1()
        self.image_scale = cfg.VISION_SEM_SEG.IMAGE_SCALE
        self.publish = publish          # callable(frame_id, colour image, header) or None
        self.undistort = undistort      # callable(image, camera) or None (cv2.undistort slot, :84-87)
        self.last_labels = None         # CUDA uint8 [h', w'] of the last frame (feeds the fused mapping path)
        if use_ros:
            self._setup_ros()

    def _setup_ros(self):
        import rospy
        from sensor_msgs.msg import Image
        self.image_sub_cam1 = rospy.Subscriber("/camera1/image_raw", Image, self.image_callback)
        self.image_sub_cam6 = rospy.Subscriber("/camera6/image_raw", Image, self.image_callback)
        self.image_pub_cam1 = rospy.Publisher("/camera1/semantic", Image, queue_size=1)
        self.image_pub_cam6 = rospy.Publisher("/camera6/semantic", Image, queue_size=1)

    def image_callback(self, msg):
        """:74-136.  msg.data: uint8[H,W,3] BGR (as the camera driver publishes it).  Returns the
        colourised uint8[H,W,3] image (also handed to ``publish``)."""
        image_in = msg.data if isinstance(msg.data, np.ndarray) else np.asarray(msg.data)
        image_in = image_in[:, :, ::-1]                                            # :83 BGR2RGB
        cam = {"camera1": self.cam1, "camera6": self.cam6}.get(msg.header.frame_id)
        if self.undistort is not None and cam is not None:
            image_in = self.undistort(image_in, cam)                               # :84-87
        h, w = image_in.shape[0], image_in.shape[1]
        if self.image_scale < 1:                                                   # :92-98
            rw, rh = int(w * self.image_scale), int(h * self.image_scale)
            image_in_resized = self.seg.resize_area(image_in, rh, rw)
        else:
            image_in_resized = image_in
        labels = self.seg.segmentation_device(np.ascontiguousarray(image_in_resized))   # :101-102 (uint8 on the GPU)
        self.last_labels = labels
        colored = colorize_labels_device(labels, h, w, self.seg_color_ref)         # :109-116
        out = colored.cpu().numpy()
        if self.publish is not None:
            self.publish(msg.header.frame_id, out, msg.header)                     # :129-134
        return out
