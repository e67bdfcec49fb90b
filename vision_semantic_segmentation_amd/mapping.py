"""SemanticMapping -- the reference's mapping node (src/mapping.py:39-541) with its per-frame
arithmetic on the MI355X.

Same class name, method names, argument layouts and attributes as the reference, so a ROS shell or
the replay driver can call it unchanged:

    sm = SemanticMapping(cfg)
    sm.pcd_callback(msg); sm.pose_callback(msg); sm.image_callback(msg)      # mapping.py:172,221,261
    pcd_in_range, label = sm.project_pcd(pcd, frame_id, image, pose, cam)    # mapping.py:357
    grid = sm.update_map(grid, pcd_in_range, label)                           # mapping.py:391
    sm.mapping(semantic_image, pose, cam)                                     # mapping.py:292

What differs from the reference, by design:
  * project_pcd / update_map / mapping run as HIP kernels through libavl_hip.so (csrc/mapping.hip);
    there is no NumPy fallback -- a missing library or GPU raises.
  * the grid lives in HBM (``self.map_dev``, float64 like the reference unless MAPPING.GRID_DTYPE is
    "f32"); ``self.map`` downloads it on access.
  * ``mapping()`` is fused: points are projected, labelled, voted and applied without the
    intermediate masked point list; ``frame_device()`` does the same for inputs already in HBM
    (e.g. the argmax label map straight from SemanticSegmentation).
  * rospy subscribers/publishers are only created when ``use_ros=True`` and rospy imports; a lock
    serialises the three callbacks (the reference mutates its queues from three threads unlocked).
"""
import ctypes as C
import os
import os.path as osp
import threading

import numpy as np
import torch

from . import _lib
from .camera import camera_setup_1, camera_setup_6
from .config import get_cfg_defaults
from .data.confusion_matrix import ConfusionMatrix
from .labels import PALETTE_19, vote_lut
from .utils.utils_ros import euler_matrix, get_transform_from_pose
from .utils.logger import MyLogger

# src/mapping.py:404 (= minus the global_map pose of :232-233)
PCD_ORIGIN_OFFSET = (1369.0496826171875, 562.84814453125, 0.0)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


_DBL_CACHE = {}


def _dbl(a):
    """float64 ctypes array of a small host matrix (P, T, CM); memoised on the bytes -- the same few matrices are
    passed every frame and building a ctypes array element by element costs more than the kernel launch."""
    a = np.ascontiguousarray(a, dtype=np.float64).ravel()
    key = a.tobytes()
    hit = _DBL_CACHE.get(key)
    if hit is None:
        if len(_DBL_CACHE) > 256:
            _DBL_CACHE.clear()
        hit = (C.c_double * a.size).from_buffer_copy(key)
        _DBL_CACHE[key] = hit
    return hit


class DeviceGrid(object):
    """The BEV grid and its per-frame scratch in HBM (struct avl_grid of include/avl_hip.h)."""

    def __init__(self, Hm, Wm, C_, boundary, resolution, dtype, device):
        self.Hm, self.Wm, self.C = int(Hm), int(Wm), int(C_)
        self.boundary = boundary
        self.resolution = float(resolution)
        self.device = device
        self.torch_dtype = torch.float64 if dtype == "f64" else torch.float32
        self.map = torch.zeros((self.Hm, self.Wm, self.C), dtype=self.torch_dtype, device=device)
        self.cell_mask = torch.zeros(self.Hm * self.Wm, dtype=torch.int32, device=device)
        self.counter = torch.zeros(_lib.AVL_COUNTER_INTS, dtype=torch.int32, device=device)
        self.touched = None
        self.touched_cap = 0
        self.ensure_capacity(1 << 17)

    def ensure_capacity(self, n):
        need = min(int(n), self.Hm * self.Wm) + 64 * 256      # + the rounding slack of the partitioned lists (mapping.hip, kLists)
        if need > self.touched_cap:
            self.touched_cap = max(need, 2 * self.touched_cap)
            self.touched = torch.empty(self.touched_cap, dtype=torch.int32, device=self.device)

    def struct(self, map_tensor=None):
        m = self.map if map_tensor is None else map_tensor
        g = _lib.AvlGrid()
        g.map = m.data_ptr()
        g.map_dtype = _lib.AVL_F64 if m.dtype == torch.float64 else _lib.AVL_F32
        g.Hm, g.Wm, g.C = self.Hm, self.Wm, self.C
        g.off_x, g.off_y = PCD_ORIGIN_OFFSET[0], PCD_ORIGIN_OFFSET[1]
        g.b00, g.b10 = float(self.boundary[0][0]), float(self.boundary[1][0])
        g.resolution = self.resolution
        g.cell_mask = self.cell_mask.data_ptr()
        g.touched = self.touched.data_ptr()
        g.touched_cap = self.touched_cap
        g.counter = self.counter.data_ptr()
        g.counter_len = int(self.counter.numel())
        return g


class SemanticMapping(object):
    """Create a semantic bird's eye view map from LiDAR points and 2-D semantic segmentation images
    (reference class: src/mapping.py:39)."""

    def __init__(self, cfg=None, device=None, use_ros=False, logger=None):
        cfg = get_cfg_defaults() if cfg is None else cfg
        assert len(cfg.LABELS) == len(cfg.LABELS_NAMES) == len(cfg.LABEL_COLORS)     # mapping.py:53
        _lib.lib()  # fail loudly, now, if the HIP library is missing
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("SemanticMapping needs a GPU (no CPU fallback)")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.cfg = cfg
        self._lock = threading.RLock()

        self.depth_method = cfg.MAPPING.DEPTH_METHOD
        self._ros = None
        if use_ros:
            self._setup_ros()

        output_dir = cfg.OUTPUT_DIR
        if "@" in output_dir:
            output_dir = output_dir.replace("@", osp.join(osp.dirname(__file__), "../"))
            output_dir = osp.abspath(osp.join(output_dir, cfg.TASK_NAME))
        self.logger = logger if logger is not None else MyLogger("mapping", save_dir=None)
        self.output_dir = output_dir

        self.pose = None
        self.pose_queue = []
        self.pose_time = None
        self.cam1 = camera_setup_1()
        self.cam6 = camera_setup_6()

        self.pcd = None
        self.pcd_frame_id = None
        self.pcd_queue = []
        self.pcd_header_queue = []
        self.pcd_time = None
        self.pcd_range_max = cfg.MAPPING.PCD.RANGE_MAX
        self.use_pcd_intensity = cfg.MAPPING.PCD.USE_INTENSITY
        # build-specific: unpack PointCloud2 payloads on the GPU (self.pcd is then a CUDA float32 [N,4] tensor, not ndarray [4,N])
        self.unpack_on_device = bool(getattr(cfg.MAPPING.PCD, "UNPACK_ON_DEVICE", False))
        self._pc2_stage = None

        # map[x, y] (mapping.py:106-117)
        self._grid = None
        self._map_host = None
        self.map_pose = None
        self.save_map_to_file = False
        self.map_boundary = cfg.MAPPING.BOUNDARY
        self.resolution = cfg.MAPPING.RESOLUTION
        self.label_names = cfg.LABELS_NAMES
        self.label_colors = np.array(cfg.LABEL_COLORS)
        self.map_height = int((self.map_boundary[0][1] - self.map_boundary[0][0]) / self.resolution)
        self.map_width = int((self.map_boundary[1][1] - self.map_boundary[1][0]) / self.resolution)
        self.map_depth = len(self.label_names)
        if self.map_depth > _lib.AVL_MAX_MAP_CLASSES:
            raise ValueError("at most %d map classes are supported" % _lib.AVL_MAX_MAP_CLASSES)
        self.grid_dtype = getattr(cfg.MAPPING, "GRID_DTYPE", "f64")
        self.planar_match = getattr(cfg.MAPPING, "PLANAR_MATCH", "reference")
        self.local_to_base_lookup = None       # planar mode: callable(pose_time) -> 4x4 /local_map -> /base_link (the reference's TF lookup)

        self.position_rel = np.array([[0, 0, 0]]).T
        self.yaw_rel = 0
        self.preprocessing()

        self.test_cut_time = cfg.TEST_END_TIME
        if cfg.MAPPING.CONFUSION_MTX.LOAD_PATH != "":                                  # mapping.py:127-132
            confusion_matrix = ConfusionMatrix(load_path=cfg.MAPPING.CONFUSION_MTX.LOAD_PATH)
            self.confusion_matrix = confusion_matrix.get_submatrix(cfg.LABELS, to_probability=True, use_log=True)
        else:
            self.confusion_matrix = np.eye(len(self.label_names))

        self.ground_truth_dir = cfg.GROUND_TRUTH_DIR
        self.input_list = []
        self.input_dir = cfg.MAPPING.INPUT_DIR
        self.record_inputs = cfg.MAPPING.INPUT_DIR != ""
        self.frames_mapped = 0

        # device-side caches
        self._scratch = None
        self._pcd_dev = None
        self._out_pcd = None
        self._out_label = None
        self._out_count = torch.zeros(4, dtype=torch.int32, device=self.device)

    # ------------------------------------------------------------------ ROS plumbing (optional)
    def _setup_ros(self):
        import rospy  # noqa: F401  (only when asked for)
        from geometry_msgs.msg import PoseStamped
        from sensor_msgs.msg import Image, PointCloud2
        self._ros = rospy
        self.sub_pose = rospy.Subscriber("/current_pose", PoseStamped, self.pose_callback)
        self.image_sub_cam1 = rospy.Subscriber("/camera1/semantic", Image, self.image_callback)
        self.image_sub_cam6 = rospy.Subscriber("/camera6/semantic", Image, self.image_callback)
        if self.depth_method == "points_map":
            self.sub_pcd = rospy.Subscriber("/reduced_map", PointCloud2, self.pcd_callback)
        elif self.depth_method == "points_raw":
            self.sub_pcd = rospy.Subscriber("/points_raw", PointCloud2, self.pcd_callback)

    # ------------------------------------------------------------------ constants (mapping.py:142-170)
    def preprocessing(self):
        """Setup constant matrices (mapping.py:142-163; only the ones the LiDAR path uses)."""
        self.T_velodyne_to_basklink = self.set_velodyne_to_baselink()
        self.T_cam1_to_base = np.matmul(self.T_velodyne_to_basklink, self.cam1.T)
        self.T_cam6_to_base = np.matmul(self.T_velodyne_to_basklink, self.cam6.T)
        self.discretize_matrix_inv = np.array([
            [self.resolution, 0, self.map_boundary[0][0]],
            [0, self.resolution, self.map_boundary[1][1]],
            [0, 0, 1],
        ]).astype(np.float64)
        self.discretize_matrix = np.linalg.inv(self.discretize_matrix_inv)
        self.anchor_points_2 = np.array([                                          # mapping.py:160-163
            [self.map_width, self.map_width / 2, self.map_width / 2, self.map_width],
            [self.map_height / 4, self.map_height / 4, self.map_height * 3 / 4, self.map_height * 3 / 4],
        ], dtype=np.float64)

    def set_velodyne_to_baselink(self):
        """mapping.py:165-170"""
        T = euler_matrix(0., 0.140, 0.)
        t = np.array([[2.64, 0, 1.98]]).T
        T[0:3, -1::] = t
        return T

    # ------------------------------------------------------------------ the grid
    @property
    def grid(self):
        if self._grid is None:
            self._grid = DeviceGrid(self.map_height, self.map_width, self.map_depth, self.map_boundary,
                                    self.resolution, self.grid_dtype, self.device)
        return self._grid

    @property
    def map_dev(self):
        """The grid in HBM: torch tensor [map_height, map_width, map_depth]."""
        return self.grid.map

    @property
    def map(self):
        """The grid as a NumPy array (downloaded on access), or None before the first frame
        (mapping.py:107,303-304)."""
        if self._grid is None:
            return None
        if self._map_host is None:
            self._map_host = self._grid.map.cpu().numpy()
        return self._map_host

    @map.setter
    def map(self, value):
        self._map_host = None
        if value is None:
            self._grid = None
            return
        t = torch.as_tensor(np.ascontiguousarray(value))
        g = self.grid
        g.map.copy_(t.to(self.device, dtype=g.torch_dtype))

    # ------------------------------------------------------------------ callbacks
    def pcd_callback(self, msg):
        """mapping.py:172-183.  Accepts (a) any message with ``.points`` = array [4,N] or [N,4]
        (x,y,z,intensity), or (b) a sensor_msgs/PointCloud2, unpacked with one vectorised
        ``np.frombuffer`` instead of the reference's per-point Python loop."""
        if hasattr(msg, "points"):
            pts = np.asarray(msg.points)
            pcd = pts if pts.shape[0] == 4 else pts.T
            pcd = np.ascontiguousarray(pcd, dtype=np.float64)
        elif self.unpack_on_device:
            pcd = self.unpack_pointcloud2_device(msg)[0]       # float32 [N,4] in HBM; NaN points keep their slot, x = NaN
        else:
            pcd = unpack_pointcloud2(msg)
        with self._lock:
            self.pcd_queue.append(pcd)
            self.pcd_header_queue.append(msg.header)
            self.pcd_frame_id = msg.header.frame_id

    def unpack_pointcloud2_device(self, msg, stream=None):
        """pcd_callback's unpacking (mapping.py:172-183) on the GPU: the message payload is uploaded once (through a reused
        pinned staging buffer, asynchronously on `stream`) and avl_unpack_pointcloud2 writes float32 [N,4] (x,y,z,intensity)
        straight into the layout the fused frame reads.  Returns (points CUDA float32 [N,4], n_valid CUDA int32 [1]) --
        n_valid is what read_points(skip_nans=True) would have yielded; points that it would have skipped keep their slot
        with x = NaN and are rejected by the projection kernel.
        With an explicit `stream` the upload and the kernel overlap whatever the current stream is doing (they wait for it only when the
        staging buffer has just been (re)allocated).  The returned tensors then belong to THAT stream: a caller that consumes them on
        another stream must make it wait for `stream` first and call ``tensor.record_stream(consumer_stream)`` before dropping them."""
        offs = {f.name: f.offset for f in msg.fields}
        n = int(msg.width) * int(msg.height)
        step = int(msg.point_step)
        nbytes = n * step
        # Everything this call enqueues -- the zero-fill of `count`, the upload, the kernel -- goes on ONE stream `st` (the caller's,
        # or torch's current one), and the tensors it returns are allocated under that stream so that the caching allocator
        # knows who uses them (with an explicit stream the zero-fill used to run on torch's current stream and could
        # race with the kernel's atomic increments).
        st = torch.cuda.current_stream(self.device) if stream is None else torch.cuda.ExternalStream(int(stream), device=self.device)
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(st):
            points = torch.empty((max(n, 1), 4), dtype=torch.float32, device=self.device)[:n]
            count = torch.zeros(1, dtype=torch.int32, device=self.device)
        if n == 0:
            return points, count
        payload = _pointcloud2_payload(msg)                   # uint8 [n * point_step]: rows without their row_step padding
        if self._pc2_stage is None or self._pc2_stage[0].numel() < nbytes:
            cap = max(nbytes, 1 << 22)
            self._pc2_stage = (torch.empty(cap, dtype=torch.uint8).pin_memory(), torch.empty(cap, dtype=torch.uint8, device=self.device),
                               torch.cuda.Event(), torch.cuda.Event())
            if st.cuda_stream != cur.cuda_stream:
                st.wait_stream(cur)        # the fresh device buffer comes from the CURRENT stream's allocator pool: work queued there may still own the block
                                           # (only here -- reuse is ordered by the two events, and an unconditional wait removed the overlap a caller passes a stream for)
        host, dev, staged, unpacked = self._pc2_stage
        # The kernel reads what the copy wrote (same stream).  The pinned buffer is reused only after the previous message's copy
        # has executed (`staged`), the shared device staging buffer only after the previous message's KERNEL has (`unpacked`:
        # this stream waits for it, so consecutive calls may use different streams).
        staged.synchronize()                                   # no-op for a never-recorded event
        host.numpy()[:nbytes] = payload                        # one memcpy into pinned memory
        with torch.cuda.stream(st):
            st.wait_event(unpacked)
            dev[:nbytes].copy_(host[:nbytes], non_blocking=True)
            staged.record(st)
        rc = _lib.lib().avl_unpack_pointcloud2(_ptr(dev), n, step, offs["x"], offs["y"], offs["z"], offs["intensity"], _ptr(points),
                                               _ptr(count), C.c_void_p(st.cuda_stream))
        unpacked.record(st)
        if st.cuda_stream != cur.cuda_stream:
            dev.record_stream(st)                              # (the staging buffer lives in the current stream's allocator pool)
        _lib.check(rc, "avl_unpack_pointcloud2")
        return points, count

    @staticmethod
    def _nearest_in_queue(stamps, target_stamp):
        """Index selection shared by update_pcd / update_pose (mapping.py:185-219, :238-259): walk the queue for the first
        pair that brackets `target_stamp` strictly and keep the closer of the two (the earlier one on a tie); the
        queue is then trimmed from the earlier element of that pair.  Without a bracketing pair: the latest element,
        and the queue shrinks to it.  Returns (index of the pick, index to trim from)."""
        for i in range(len(stamps) - 1):
            if stamps[i + 1] > target_stamp and stamps[i] < target_stamp:
                later_is_closer = (target_stamp - stamps[i]) > (stamps[i + 1] - target_stamp)
                return (i + 1 if later_is_closer else i), i
        return len(stamps) - 1, len(stamps) - 1

    def update_pcd(self, target_stamp):
        """Closest point cloud w.r.t. target_stamp (mapping.py:185-219)."""
        pick, keep = self._nearest_in_queue([h.stamp for h in self.pcd_header_queue], target_stamp)
        pcd, stamp = self.pcd_queue[pick], self.pcd_header_queue[pick].stamp
        self.pcd_header_queue = self.pcd_header_queue[keep:]
        self.pcd_queue = self.pcd_queue[keep:]
        return pcd, stamp

    def pose_callback(self, msg):
        """mapping.py:221-226"""
        with self._lock:
            self.pose_queue.append(msg)
            if msg.header.stamp.secs >= self.test_cut_time:
                self.save_map_to_file = True

    def update_pose(self, target_stamp):
        """Closest pose w.r.t. target_stamp (mapping.py:238-259)."""
        pick, keep = self._nearest_in_queue([m.header.stamp for m in self.pose_queue], target_stamp)
        msg = self.pose_queue[pick]
        self.pose_queue = self.pose_queue[keep:]
        return msg.pose, msg.header.stamp

    def image_callback(self, msg):
        """mapping.py:261-290: semantic image arrives -> pick cloud and pose by stamp -> mapping()."""
        self.logger.log("Mapping image at: {}.{:09d}s".format(msg.header.stamp.secs, msg.header.stamp.nsecs))
        image_in = msg.data if isinstance(getattr(msg, "data", None), (np.ndarray, torch.Tensor)) else _imgmsg_to_array(msg)
        if msg.header.frame_id == "camera1":
            camera_calibration = self.cam1
        elif msg.header.frame_id == "camera6":
            camera_calibration = self.cam6
        else:
            # the reference only warns and then fails on an unbound variable (mapping.py:277-278)
            raise ValueError("cannot find camera for frame_id %s" % msg.header.frame_id)
        with self._lock:
            if self.depth_method in ["points_map", "points_raw"]:
                if len(self.pcd_header_queue) == 0:
                    return
                self.pcd, self.pcd_time = self.update_pcd(msg.header.stamp)
            if len(self.pose_queue) == 0:
                return
            self.pose, self.pose_time = self.update_pose(msg.header.stamp)
            self.mapping(image_in, self.pose, camera_calibration)

    # ------------------------------------------------------------------ per-frame hot path
    def mapping(self, semantic_image, pose, camera_calibration):
        """mapping.py:292-321 for the LiDAR depth methods: one fused project+vote+apply on the GPU.
        semantic_image: uint8[H,W,3] colourised labels (NumPy or a CUDA tensor)."""
        if self.depth_method not in ["points_map", "points_raw"]:                 # mapping.py:320-321
            img = self._as_device_u8(semantic_image)
            self.update_map_planar(self.map_dev, img, camera_calibration)
            self.frames_mapped += 1
        else:
            if self.pcd is None:
                return
            if self.record_inputs:                                               # mapping.py:309-313
                self.input_list.append({"pcd": np.array(_to_numpy(self.pcd)), "pcd_frame_id": self.pcd_frame_id,
                                        "semantic_image": np.array(_to_numpy(semantic_image)), "pose": pose})
            img = self._as_device_u8(semantic_image)
            self.frame_device(self.pcd, self.pcd_frame_id, img, pose, camera_calibration, src_kind="rgb")
        if self.save_map_to_file:                                                # mapping.py:323-345, both depth modes
            self.save_map_to_file = False
            self.finish_run()

    def finish_run(self, output_dir=None):
        """The shutdown branch of mapping() (mapping.py:323-345): dump the recorded inputs, smooth the grid
        (apply_filter), render it (render_bev_map), write global_map.png and -- with GROUND_TRUTH_DIR set --
        print IoU / accuracy / missing rate.  Filter, renderer and evaluation are GPU kernels; only the PNG and
        the printed numbers leave the device.  Returns the colour map (uint8 [Hm,Wm,3] NumPy)."""
        from .evaluation import Test
        from .renderer import apply_filter, render_bev_map
        if self.record_inputs and self.input_list:
            self.save_inputs()
        output_dir = output_dir or self.output_dir
        os.makedirs(output_dir, exist_ok=True)
        smooth = apply_filter(self.map_dev)                                      # :332, the filtered grid replaces the map
        self.map_dev.copy_(smooth.to(self.map_dev.dtype))
        self._map_host = None
        color_dev = render_bev_map(self.map_dev, self.label_colors)              # :334
        color_map = color_dev.cpu().numpy()
        output_file = osp.join(output_dir, "global_map.png")
        try:
            from PIL import Image
            # cv2.imwrite (:340) stores channel 0 of the array as BLUE; write the same file
            Image.fromarray(np.ascontiguousarray(color_map[:, :, ::-1])).save(output_file)
            self.logger.log("Saving image to %s" % output_file)
        except ImportError:
            np.save(output_file[:-4] + ".npy", color_map)
        if self.ground_truth_dir != "":                                          # :342-345
            Test(ground_truth_dir=self.ground_truth_dir, logger=self.logger).test_single_map(color_dev)
        return color_map

    def frame_device(self, pcd, pcd_frame_id, semantic, pose, camera_calibration, src_kind="rgb",
                     image_size=None, net_palette=PALETTE_19, stream=None):
        """Fused frame (avl_fused_frame).  ``pcd``: NumPy/torch [4,N] float64|float32 (SoA, the
        reference layout) or [N,4] float32 (AoS); ``semantic``: CUDA uint8 tensor, either the colour
        image [H,W,3] (src_kind="rgb") or a class-id map [h,w] (src_kind="classmap", sampled as the
        nearest-upscaled image of size ``image_size`` = (H, W))."""
        pts, n, dtype, pstride, cstride = self._points_view(pcd)
        T = self._origin_to_velodyne(pose) if pcd_frame_id != "velodyne" else None
        g = self.grid
        g.ensure_capacity(n)
        gs = g.struct()
        P = _dbl(camera_calibration.P)
        Tc = _dbl(T) if T is not None else None
        cm = _dbl(self.confusion_matrix)
        colors = self._colors_host()
        bonus = self._bonus_classes()
        if src_kind == "rgb":
            assert semantic.dim() == 3 and semantic.shape[2] == 3 and semantic.dtype == torch.uint8
            sh, sw = int(semantic.shape[0]), int(semantic.shape[1])
            ih, iw = sh, sw
            kind, lut = _lib.AVL_SRC_RGB, None
        else:
            assert semantic.dim() == 2 and semantic.dtype == torch.uint8
            sh, sw = int(semantic.shape[0]), int(semantic.shape[1])
            ih, iw = (sh, sw) if image_size is None else (int(image_size[0]), int(image_size[1]))
            kind = _lib.AVL_SRC_CLASSMAP
            lut = self._lut_host(net_palette)
        semantic = semantic.contiguous()
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        rc = _lib.lib().avl_fused_frame(C.byref(gs), pts, n, dtype, pstride, cstride, P, Tc, float(self.pcd_range_max),
                                        kind, _ptr(semantic), sw, sh, iw, ih, lut, colors, cm, bonus, C.c_void_p(s))
        _lib.check(rc, "avl_fused_frame")
        self._map_host = None
        self.frames_mapped += 1

    def project_pcd(self, pcd, pcd_frame_id, image, pose, camera_calibration):
        """mapping.py:357-389.  NumPy in, NumPy out: (pcd[:, mask] float64[4,M], label uint8[3,M]).
        Pass ``as_device=True`` through project_pcd_device for CUDA tensors instead."""
        if pcd is None:
            return
        out_pcd, out_label, m = self.project_pcd_device(pcd, pcd_frame_id, image, pose, camera_calibration)
        m = int(m.item())
        return out_pcd[:, :m].cpu().numpy(), out_label[:, :m].cpu().numpy()

    def project_pcd_device(self, pcd, pcd_frame_id, image, pose, camera_calibration):
        """avl_project_pcd: returns CUDA tensors (out_pcd f64[4,N], out_label u8[3,N], count int32[1]);
        only the first `count` columns are meaningful."""
        pts, n, dtype, pstride, cstride = self._points_view(pcd)
        img = self._as_device_u8(image)
        assert img.dim() == 3 and img.shape[2] == 3
        T = self._origin_to_velodyne(pose) if pcd_frame_id != "velodyne" else None
        ld = max(n, 1)
        out_pcd = torch.empty((4, ld), dtype=torch.float64, device=self.device)
        out_label = torch.empty((3, ld), dtype=torch.uint8, device=self.device)
        count = torch.zeros(4, dtype=torch.int32, device=self.device)
        need = int(_lib.lib().avl_project_pcd_scratch_bytes(n))
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=self.device)
        s = torch.cuda.current_stream(self.device).cuda_stream
        rc = _lib.lib().avl_project_pcd(pts, n, dtype, pstride, cstride, _dbl(camera_calibration.P),
                                        _dbl(T) if T is not None else None, float(self.pcd_range_max),
                                        _ptr(img), int(img.shape[1]), int(img.shape[0]),
                                        _ptr(out_pcd), _ptr(out_label), ld, _ptr(count), _ptr(self._scratch), C.c_void_p(s))
        _lib.check(rc, "avl_project_pcd")
        return out_pcd, out_label, count[:1]

    def semantic_cloud_device(self, pcd, pcd_frame_id, image, pose, camera_calibration):
        """The semantic point cloud mapping() publishes (mapping.py:314-317): project_pcd followed by
        create_point_cloud (utils_ros.py:31-59), both on the GPU.  Returns (records uint8[M,16] CUDA tensor in
        PointCloud2 layout x,y,z float32 + rgba uint32, M)."""
        out_pcd, out_label, count = self.project_pcd_device(pcd, pcd_frame_id, image, pose, camera_calibration)
        n = int(out_pcd.shape[1])
        rec = torch.empty((max(n, 1), 16), dtype=torch.uint8, device=self.device)
        s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(_lib.lib().avl_pack_semantic_cloud(_ptr(out_pcd), _ptr(out_label), n, n, _ptr(count), _ptr(rec), s),
                   "avl_pack_semantic_cloud")
        m = int(count.item())
        return rec[:m], m

    def update_map(self, map, pcd, label):
        """mapping.py:391-444.  ``map`` may be
          * a CUDA tensor [Hm,Wm,C] (float64/float32): updated in place on the GPU and returned;
          * a NumPy array (the reference's type): the vote kernel finds the touched cells, their rows
            are moved to the GPU, updated there by the same apply kernel, and written back -- the host
            only moves bytes, it does no arithmetic.  The array is mutated in place and returned."""
        m = int(pcd.shape[1])
        if m == 0:
            return map
        pcd_d = self._as_device(pcd, torch.float64)
        label_d = self._as_device_u8(label)
        assert pcd_d.shape[0] == 4 and label_d.shape[0] == 3 and label_d.shape[1] == m
        pcd_d, label_d = pcd_d.contiguous(), label_d.contiguous()
        g = self.grid
        assert tuple(map.shape) == (g.Hm, g.Wm, g.C), "map shape %s != grid %s" % (tuple(map.shape), (g.Hm, g.Wm, g.C))
        g.ensure_capacity(m)
        s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        colors, cm, bonus = self._colors_host(), _dbl(self.confusion_matrix), self._bonus_classes()
        L = _lib.lib()
        if isinstance(map, torch.Tensor):
            assert map.is_cuda and map.is_contiguous()
            gs = g.struct(map)
            _lib.check(L.avl_update_map(C.byref(gs), _ptr(pcd_d), _ptr(label_d), m, m, None, colors, cm, bonus, s),
                       "avl_update_map")
            if map.data_ptr() == g.map.data_ptr():
                self._map_host = None
            return map
        # NumPy grid: vote on the GPU, move only the touched rows
        gs = g.struct()
        _lib.check(L.avl_vote_points(C.byref(gs), _ptr(pcd_d), _ptr(label_d), m, m, None, colors, bonus, s), "avl_vote_points")
        u = int(g.counter[0].item())
        if u == 0:
            return map
        cells = g.touched[:u].cpu().numpy().astype(np.int64)
        cx, cy = cells // g.Wm, cells % g.Wm
        np_dtype = np.float64 if map.dtype != np.float32 else np.float32
        rows = torch.from_numpy(np.ascontiguousarray(map[cx, cy, :], dtype=np_dtype)).to(self.device)
        row_dtype = _lib.AVL_F64 if np_dtype == np.float64 else _lib.AVL_F32
        _lib.check(L.avl_grid_apply(C.byref(gs), cm, _ptr(rows), row_dtype, s), "avl_grid_apply")
        map[cx, cy, :] = rows.cpu().numpy()
        return map

    def update_map_planar(self, map_local, image, cam, T_local_to_base=None):
        """mapping.py:446-488: project the semantic image onto the map plane through a 4-point homography and update the grid
        (avl_planar_update: warp + class test + clamp in one pass over the grid).  ``map_local``: CUDA tensor [Hm,Wm,C] (updated
        in place) or NumPy array (uploaded, updated, written back); ``image``: uint8 [H,W,3].  The reference looks the
        /local_map -> /base_link transform up in a live ROS TF tree (:454-457); here it is ``T_local_to_base`` (4x4) or, when
        None, ``self.local_to_base_lookup(self.pose_time)``.  ``MAPPING.PLANAR_MATCH`` picks the class test (see config.py):
        with "reference" the warp cannot influence the result and is skipped."""
        match = 1 if self.planar_match == "colour" else 0
        is_np = not isinstance(map_local, torch.Tensor)
        grid_t = torch.from_numpy(np.ascontiguousarray(map_local)).to(self.device) if is_np else map_local
        assert grid_t.is_cuda and grid_t.is_contiguous() and tuple(grid_t.shape) == (self.map_height, self.map_width, self.map_depth)
        img = self._as_device_u8(image)
        hinv = None
        if match:
            if T_local_to_base is None:
                if getattr(self, "local_to_base_lookup", None) is None:
                    raise RuntimeError("update_map_planar needs T_local_to_base (the reference reads it from ROS TF): pass it or set "
                                       "SemanticMapping.local_to_base_lookup")
                T_local_to_base = self.local_to_base_lookup(self.pose_time)
            pts_img = planar_points_image(self.anchor_points_2, self.discretize_matrix_inv, np.asarray(T_local_to_base, dtype=np.float64),
                                          self.T_velodyne_to_basklink, cam.P)
            hinv = _dbl(np.linalg.inv(find_homography(pts_img.T, self.anchor_points_2.T)))
        # :468-470 `mask[:, 0:sep] = 0` with Python's slice rules: a negative sep (map_boundary[0][0] > 8, as in the reference's
        # own base_cfg) masks columns [0, Wm + sep), a sep beyond the width masks every column
        sep = int((8 - self.map_boundary[0][0]) / self.resolution)
        sep = len(range(*slice(0, sep).indices(self.map_width)))
        dt = _lib.AVL_F64 if grid_t.dtype == torch.float64 else _lib.AVL_F32
        s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = _lib.lib().avl_planar_update(_ptr(grid_t), dt, self.map_height, self.map_width, self.map_depth, _ptr(img), int(img.shape[0]),
                                          int(img.shape[1]), hinv, sep, self._colors_host(), match, s)
        _lib.check(rc, "avl_planar_update")
        if is_np:
            map_local[...] = grid_t.cpu().numpy()
            return map_local
        if self._grid is not None and grid_t.data_ptr() == self._grid.map.data_ptr():
            self._map_host = None
        return map_local

    # ------------------------------------------------------------------ multi-GPU: shared global grid
    def global_map(self, group=None, dst=None, exchange_dtype=None, mode="dense"):
        """Sum of every rank's private grid: each rank maps its own camera stream into its
        own grid; since a frame's contribution never depends on the grid's content, the shared grid
        is the element-wise sum.  mode "dense": all_reduce (or reduce to ``dst``) over RCCL on a copy, so the private
        grid keeps accumulating; ``exchange_dtype=torch.float32`` halves the payload (see distributed.reduce_grids).
        mode "sparse": an all-gather of every rank's (cell int32, delta float32 [C]) records instead (distributed.reduce_grids_sparse:
        ~0.6 MB per GPU at config C against 80 MB); "auto": sparse while every rank touched fewer than 5 % of the cells.
        Returns the CUDA tensor (valid on every rank, or on dst only); ``.last_exchange`` = (mode used, bytes this rank sent)."""
        from . import distributed as D
        if mode == "dense":
            total = D.reduce_grids(self.grid.map, group=group, dst=dst, exchange_dtype=exchange_dtype)
            self.last_exchange = ("dense", int(total.numel() * total.element_size()))
            return total
        assert dst is None, "the record exchange is an all-gather: every rank gets the sum"
        vdt = torch.float32 if exchange_dtype is None else exchange_dtype
        if mode == "sparse":
            total, sent = D.reduce_grids_sparse(self.grid.map, group=group, value_dtype=vdt)
            self.last_exchange = ("sparse", sent)
            return total
        if mode != "auto":
            raise ValueError("global_map mode must be 'dense', 'sparse' or 'auto', not %r" % mode)
        total, used, sent = D.reduce_grids_auto(self.grid.map, group=group, exchange_dtype=vdt)
        self.last_exchange = (used, sent)
        return total

    def save_inputs(self, path=None):
        """The reference dumps input_list with hickle (mapping.py:324-326); hickle is optional here,
        so frames go to one .npz each with the same field names (pose as 7 numbers)."""
        path = path or self.input_dir
        os.makedirs(path, exist_ok=True)
        for k, fr in enumerate(self.input_list):
            pose = fr["pose"]
            pose7 = pose.to_array() if hasattr(pose, "to_array") else _pose_to_array(pose)
            np.savez_compressed(osp.join(path, "frame_%06d.npz" % k), pcd=fr["pcd"], pcd_frame_id=np.array(fr["pcd_frame_id"]),
                                semantic_image=fr["semantic_image"], pose=pose7)

    def get_extrinsics(self, pose, camera_id):
        """mapping.py:528-540"""
        T_base_to_origin = get_transform_from_pose(pose)
        if camera_id == "camera1":
            T_cam_to_origin = np.matmul(T_base_to_origin, self.T_cam1_to_base)
        elif camera_id == "camera6":
            T_cam_to_origin = np.matmul(T_base_to_origin, self.T_cam6_to_base)
        else:
            raise ValueError("unable to find camera to base for camera_id %s" % camera_id)
        return np.linalg.inv(T_cam_to_origin)[0:3]

    # ------------------------------------------------------------------ helpers
    def _origin_to_velodyne(self, pose):
        """mapping.py:368-369 (host, float64 4x4)."""
        T_base_to_origin = get_transform_from_pose(pose)
        return np.linalg.inv(np.matmul(T_base_to_origin, self.T_velodyne_to_basklink))

    def _colors_host(self):
        c = np.ascontiguousarray(self.label_colors, dtype=np.uint8).ravel()
        key = c.tobytes()
        if getattr(self, "_colors_key", None) != key:
            self._colors_key, self._colors_c = key, (C.c_uint8 * c.size).from_buffer_copy(key)
        return self._colors_c

    def _lut_host(self, net_palette):
        key = (np.asarray(net_palette, dtype=np.uint8).tobytes(), np.ascontiguousarray(self.label_colors, dtype=np.uint8).tobytes())
        if getattr(self, "_lut_key", None) != key:
            lut_np = vote_lut(net_palette, self.label_colors)
            self._lut_key, self._lut_c = key, (C.c_uint32 * 256).from_buffer_copy(lut_np.astype(np.uint32).tobytes())
        return self._lut_c

    def _bonus_classes(self):
        """bit i set when class i is a "lane" class and USE_INTENSITY is on (mapping.py:427-431)."""
        if not self.use_pcd_intensity:
            return 0
        return sum(1 << i for i, name in enumerate(self.label_names) if name == "lane")

    def _as_device(self, a, dtype):
        if isinstance(a, torch.Tensor):
            return a.to(self.device, dtype=dtype)
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device, dtype=dtype)

    def _as_device_u8(self, a):
        if isinstance(a, torch.Tensor):
            assert a.dtype == torch.uint8
            return a.to(self.device).contiguous()
        a = np.ascontiguousarray(a)
        assert a.dtype == np.uint8, "semantic images / labels must be uint8"
        return torch.from_numpy(a).to(self.device)

    def _points_view(self, pcd):
        """-> (ctypes ptr, n, avl dtype, point_stride, comp_stride) and keeps the tensor alive."""
        if not isinstance(pcd, torch.Tensor):
            a = np.asarray(pcd)
            if a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float64)
            pcd = torch.from_numpy(np.ascontiguousarray(a))
        if pcd.dtype not in (torch.float32, torch.float64):
            pcd = pcd.to(torch.float64)
        t = pcd.to(self.device).contiguous()
        self._pcd_dev = t
        es = t.element_size()
        dtype = _lib.AVL_F64 if t.dtype == torch.float64 else _lib.AVL_F32
        if t.dim() == 2 and t.shape[0] == 4:            # SoA [4,N]: the reference layout
            n = int(t.shape[1])
            return _ptr(t), n, dtype, es, es * max(n, 1)
        if t.dim() == 2 and t.shape[1] == 4:            # AoS [N,4]
            return _ptr(t), int(t.shape[0]), dtype, 4 * es, es
        raise ValueError("point cloud must be [4,N] or [N,4], got %s" % (tuple(t.shape),))


def _to_numpy(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def _pose_to_array(pose):
    p, o = pose.position, pose.orientation
    return np.array([p.x, p.y, p.z, o.x, o.y, o.z, o.w], dtype=np.float64)


def _imgmsg_to_array(msg):
    """sensor_msgs/Image (rgb8/bgr8/8UC3 or mono8) -> ndarray, without cv_bridge."""
    ch = 3 if msg.encoding in ("rgb8", "bgr8", "8UC3") else 1
    a = np.frombuffer(msg.data, dtype=np.uint8).reshape(msg.height, msg.step)[:, :msg.width * ch]
    return a.reshape(msg.height, msg.width, ch) if ch == 3 else a.reshape(msg.height, msg.width)


def find_homography(pts_src, pts_dst):
    """cv2.findHomography(pts_src, pts_dst) for the four anchor correspondences of the planar mode (homography.py:37): the exact
    projective map by the normalised DLT, h33 = 1.  Host float64 (a 8 x 9 SVD)."""
    src, dst = np.asarray(pts_src, dtype=np.float64), np.asarray(pts_dst, dtype=np.float64)

    def normalise(p):
        c = p.mean(axis=0)
        s = np.sqrt(2.0) / max(np.sqrt(((p - c) ** 2).sum(axis=1)).mean(), 1e-300)
        return (p - c) * s, np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
    a, Ta = normalise(src)
    b, Tb = normalise(dst)
    A = []
    for (x, y), (u, v) in zip(a, b):
        A.append([-x, -y, -1, 0, 0, 0, u * x, u * y, u])
        A.append([0, 0, 0, -x, -y, -1, v * x, v * y, v])
    Hn = np.linalg.svd(np.array(A))[2][-1].reshape(3, 3)
    H = np.linalg.inv(Tb) @ Hn @ Ta
    return H / H[2, 2]


def planar_points_image(anchor, discretize_matrix_inv, T_local_to_base, T_velodyne_to_baselink, P):
    """mapping.py:449-463: anchor cells -> local metres (z = 0) -> velodyne -> image pixels ([2][4]); host float64."""
    pm = np.vstack([anchor, np.ones((1, anchor.shape[1]))])
    pl = np.matmul(discretize_matrix_inv, pm)
    pl[2, :] = 0
    pl = np.vstack([pl, np.ones((1, pl.shape[1]))])
    pv = np.matmul(np.matmul(np.linalg.inv(T_velodyne_to_baselink), T_local_to_base), pl)
    pi = np.matmul(P, pv)
    return pi[0:2] / pi[2]


def _pointcloud2_payload(msg):
    """The points of a sensor_msgs/PointCloud2 as one uint8 array [width * height * point_step]: organised clouds
    (height > 1) may pad each row to row_step, which is dropped here; big-endian payloads are refused (the reference's
    read_points would byte-swap them; nothing on this path produces one)."""
    if getattr(msg, "is_bigendian", False):
        raise NotImplementedError("big-endian PointCloud2 payloads are not supported")
    width, height, step = int(msg.width), int(msg.height), int(msg.point_step)
    row_step = int(getattr(msg, "row_step", 0) or width * step)
    if row_step < width * step:
        raise ValueError("PointCloud2 row_step %d < width * point_step %d" % (row_step, width * step))
    if height <= 1 or row_step == width * step:
        return np.frombuffer(msg.data, dtype=np.uint8, count=width * height * step)
    rows = np.frombuffer(msg.data, dtype=np.uint8, count=height * row_step).reshape(height, row_step)
    return np.ascontiguousarray(rows[:, :width * step]).reshape(-1)


def unpack_pointcloud2(msg):
    """sensor_msgs/PointCloud2 -> float64[4,N] (x,y,z,intensity), NaN points dropped
    (what mapping.py:178-180 does with a per-point loop; SURVEY Q8: the buffer is sized to the
    number of valid points instead of leaving garbage columns)."""
    offs = {f.name: f.offset for f in msg.fields}
    n = msg.width * msg.height
    raw = _pointcloud2_payload(msg).reshape(n, msg.point_step)
    cols = []
    for name in ("x", "y", "z", "intensity"):
        o = offs[name]
        cols.append(raw[:, o:o + 4].copy().view(np.float32).reshape(n))
    pts = np.stack(cols).astype(np.float64)
    keep = ~np.isnan(pts).any(axis=0)
    return np.ascontiguousarray(pts[:, keep])
