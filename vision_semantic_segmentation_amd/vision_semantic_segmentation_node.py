"""VisionSemanticSegmentationNode -- the reference's segmentation node
(src/vision_semantic_segmentation_node.py:41-136) around the HIP segmentation stack.

image_callback keeps the reference's sequence (:74-136): BGR->RGB, cv2.undistort, optional INTER_AREA downscale,
``SemanticSegmentation.segmentation``, uint8 cast, INTER_NEAREST upscale to the input size, palette
colouring, publish.  Pre-processing runs inside the network's first kernel (the stem's loader applies BGR->RGB, undistort and
INTER_AREA per pixel while it fills its LDS tile; avl_preprocess_image is the same function as a stand-alone kernel), the
upscale + colouring is one more kernel (avl_colorize_labels); the frame is uploaded once and only the colour image comes back.
"""
import ctypes as C
import threading

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


def preprocess_device(bgr, camera=None, factor=1, stream=None):
    """vision_semantic_segmentation_node.py:83-98 on the GPU (avl_preprocess_image): BGR->RGB, cv2.undistort with the
    camera's K / dist (skipped when camera is None), INTER_AREA downscale by the integer `factor`.
    bgr: uint8 [H,W,3] ndarray or CUDA tensor -> uint8 CUDA tensor [H/factor, W/factor, 3] (RGB)."""
    t = bgr if isinstance(bgr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(bgr))
    t = t.cuda().contiguous() if not t.is_cuda else t.contiguous()
    assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3
    h, w = int(t.shape[0]), int(t.shape[1])
    out = torch.empty((h // factor, w // factor, 3), dtype=torch.uint8, device=t.device)
    K = dist = None
    if camera is not None:
        K = (C.c_double * 9)(*np.asarray(camera.K, dtype=np.float64).ravel().tolist())
        dist = (C.c_double * 5)(*np.asarray(camera.dist, dtype=np.float64).ravel()[:5].tolist())
    s = torch.cuda.current_stream(t.device).cuda_stream if stream is None else stream
    _lib.check(_lib.lib().avl_preprocess_image(C.c_void_p(t.data_ptr()), h, w, K, dist, int(factor), C.c_void_p(out.data_ptr()),
                                               C.c_void_p(s)), "avl_preprocess_image")
    return out


def preprocess_area_device(bgr, camera, out_h, out_w, stream=None):
    """The same chain with cv2.resize(INTER_AREA) to ANY smaller size (avl_preprocess_image_area): what the reference does for an
    IMAGE_SCALE that is not 1 / integer (:92-98: width = int(W * scale), height = int(H * scale))."""
    t = bgr if isinstance(bgr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(bgr))
    t = t.cuda().contiguous() if not t.is_cuda else t.contiguous()
    assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3
    h, w = int(t.shape[0]), int(t.shape[1])
    out = torch.empty((int(out_h), int(out_w), 3), dtype=torch.uint8, device=t.device)
    K = dist = None
    if camera is not None:
        K = (C.c_double * 9)(*np.asarray(camera.K, dtype=np.float64).ravel().tolist())
        dist = (C.c_double * 5)(*np.asarray(camera.dist, dtype=np.float64).ravel()[:5].tolist())
    s = torch.cuda.current_stream(t.device).cuda_stream if stream is None else stream
    _lib.check(_lib.lib().avl_preprocess_image_area(C.c_void_p(t.data_ptr()), h, w, K, dist, int(out_h), int(out_w), C.c_void_p(out.data_ptr()),
                                                    C.c_void_p(s)), "avl_preprocess_image_area")
    return out


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

    def __init__(self, cfg, seg=None, use_ros=False, publish=None, undistort=True):
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
        self.undistort = undistort      # the reference always undistorts camera1 / camera6 frames (:84-87)
        self.last_labels = None         # CUDA uint8 [h', w'] of the last frame (feeds the fused mapping path)
        # rospy runs each subscription's callback on its own thread, and both cameras share one compiled plan (fixed
        # input / activation / label buffers on one stream): one frame at a time through upload -> net -> colour -> download
        self._lock = threading.Lock()
        self._ros_image_cls = None
        self.image_pub_cam1 = self.image_pub_cam6 = None
        if use_ros:
            self._setup_ros()

    def _setup_ros(self):
        import rospy
        from sensor_msgs.msg import Image
        self.image_sub_cam1 = rospy.Subscriber("/camera1/image_raw", Image, self.image_callback)
        self.image_sub_cam6 = rospy.Subscriber("/camera6/image_raw", Image, self.image_callback)
        self.image_pub_cam1 = rospy.Publisher("/camera1/semantic", Image, queue_size=1)
        self.image_pub_cam6 = rospy.Publisher("/camera6/semantic", Image, queue_size=1)
        self._ros_image_cls = Image

    def _publish_ros(self, colored, header):
        """:118-134 -- the colour image as an 8UC3 sensor_msgs/Image (cv_bridge's "passthrough" of a uint8 H x W x 3 array)
        with the INPUT message's stamp and frame_id, on the publisher of that camera."""
        pub = {"camera1": self.image_pub_cam1, "camera6": self.image_pub_cam6}.get(header.frame_id)
        if pub is None:
            return None                                           # :135-136: the reference only warns
        out = self._ros_image_cls()
        out.height, out.width = int(colored.shape[0]), int(colored.shape[1])
        out.encoding, out.is_bigendian, out.step = "8UC3", 0, int(colored.shape[1]) * 3
        out.data = np.ascontiguousarray(colored).tobytes()
        out.header.stamp = header.stamp
        out.header.frame_id = header.frame_id
        pub.publish(out)
        return out

    def _downscale_factor(self, h=None, w=None):
        """IMAGE_SCALE -> integer INTER_AREA factor (:92-98), or None when the frame does not shrink by an integer ratio (the reference's
        configs use 1.0 and 0.5): the general area resize (preprocess_area_device) then makes the (int(h * scale), int(w * scale)) input."""
        if self.image_scale >= 1:
            return 1
        f = int(round(1.0 / self.image_scale))
        if abs(f * self.image_scale - 1.0) > 1e-9 or (h is not None and (h % f or w % f)):
            return None
        return f

    def image_callback(self, msg):
        """:74-136.  msg.data: uint8[H,W,3] BGR (as the camera driver publishes it).  The whole chain runs on the GPU:
        pre-processing (:83-98), segmentation (:101-102), nearest upscale + palette (:109-116); only the published
        colour image comes back to the host.  Returns that uint8[H,W,3] image (also handed to ``publish``)."""
        if isinstance(msg.data, (np.ndarray, torch.Tensor)):
            bgr = msg.data
        else:                                         # a sensor_msgs/Image: bytes + height / width / step / encoding (:76-80)
            from .mapping import _imgmsg_to_array
            bgr = _imgmsg_to_array(msg)
            if bgr.ndim != 3:
                raise ValueError("image_callback needs a 3-channel image, got encoding %r" % (msg.encoding,))
        h, w = int(bgr.shape[0]), int(bgr.shape[1])
        cam = {"camera1": self.cam1, "camera6": self.cam6}.get(msg.header.frame_id)   # unknown frame ids: no undistortion (:88-89)
        with self._lock:
            cam = cam if self.undistort else None
            factor = self._downscale_factor(h, w)
            if factor is None:                                   # any other IMAGE_SCALE: OpenCV's general area resize as a stand-alone kernel
                labels = self.seg.segmentation_device(preprocess_area_device(bgr, cam, int(h * self.image_scale), int(w * self.image_scale)))
            elif self.seg.precision == "f32":                    # no 16-bit stem: stand-alone pre-processing kernel, then the network
                labels = self.seg.segmentation_device(preprocess_device(bgr, cam, factor))
            else:                                                # pre-processing inside the stem's loader (no RGB frame in between)
                labels = self.seg.segmentation_device_raw(bgr, None if cam is None else cam.K, None if cam is None else cam.dist, factor)
            self.last_labels = labels
            colored = colorize_labels_device(labels, h, w, self.seg_color_ref)
            out = colored.cpu().numpy()
        if self.publish is not None:
            self.publish(msg.header.frame_id, out, msg.header)                     # :129-134
        if self._ros_image_cls is not None:
            self._publish_ros(out, msg.header)
        return out
