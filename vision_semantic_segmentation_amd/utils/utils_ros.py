"""ROS-free stand-ins for the pieces of src/utils/utils_ros.py the hot path needs.

``get_transform_from_pose`` (utils_ros.py:104-111) wraps ROS ``TransformerROS.fromTranslationRotation``;
ROS is optional here, so the 4x4 is built directly from the quaternion with the published
tf.transformations formulas.  Light message types (``Pose``, ``Header`` ...) let callers without
ROS feed the same callbacks; real ROS messages work too (only attribute access is used).
"""
import math

import numpy as np


class _Vec3(object):
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = x, y, z


class _Quat(object):
    def __init__(self, x=0.0, y=0.0, z=0.0, w=1.0):
        self.x, self.y, self.z, self.w = x, y, z, w


class Pose(object):
    """geometry_msgs/Pose look-alike."""

    def __init__(self, position=(0.0, 0.0, 0.0), orientation=(0.0, 0.0, 0.0, 1.0)):
        self.position = _Vec3(*position)
        self.orientation = _Quat(*orientation)

    @classmethod
    def from_array(cls, p7):
        return cls(tuple(float(v) for v in p7[:3]), tuple(float(v) for v in p7[3:7]))

    def to_array(self):
        p, o = self.position, self.orientation
        return np.array([p.x, p.y, p.z, o.x, o.y, o.z, o.w], dtype=np.float64)


class Stamp(object):
    """rospy.Time look-alike: ordered, subtractable (differences are floats in seconds)."""

    def __init__(self, secs=0, nsecs=0):
        self.secs, self.nsecs = int(secs), int(nsecs)

    def to_sec(self):
        return self.secs + 1e-9 * self.nsecs

    def __lt__(self, o): return (self.secs, self.nsecs) < (o.secs, o.nsecs)
    def __gt__(self, o): return (self.secs, self.nsecs) > (o.secs, o.nsecs)
    def __eq__(self, o): return (self.secs, self.nsecs) == (o.secs, o.nsecs)
    def __sub__(self, o): return self.to_sec() - o.to_sec()


class Header(object):
    def __init__(self, stamp=None, frame_id=""):
        self.stamp = stamp if stamp is not None else Stamp()
        self.frame_id = frame_id


class Message(object):
    """Generic message: header + arbitrary payload fields (data / pose / points)."""

    def __init__(self, header=None, **fields):
        self.header = header if header is not None else Header()
        for k, v in fields.items():
            setattr(self, k, v)


def quaternion_matrix(quaternion):
    """tf.transformations.quaternion_matrix: (x, y, z, w) -> 4x4 homogeneous rotation."""
    q = np.array(quaternion[:4], dtype=np.float64, copy=True)
    nq = np.dot(q, q)
    if nq < np.finfo(float).eps * 4.0:
        return np.identity(4)
    q *= math.sqrt(2.0 / nq)
    q = np.outer(q, q)
    return np.array((
        (1.0 - q[1, 1] - q[2, 2], q[0, 1] - q[2, 3], q[0, 2] + q[1, 3], 0.0),
        (q[0, 1] + q[2, 3], 1.0 - q[0, 0] - q[2, 2], q[1, 2] - q[0, 3], 0.0),
        (q[0, 2] - q[1, 3], q[1, 2] + q[0, 3], 1.0 - q[0, 0] - q[1, 1], 0.0),
        (0.0, 0.0, 0.0, 1.0)), dtype=np.float64)


def euler_matrix(ai, aj, ak):
    """tf.transformations.euler_matrix(ai, aj, ak, 'sxyz') -- static x, y, z rotations."""
    si, sj, sk = math.sin(ai), math.sin(aj), math.sin(ak)
    ci, cj, ck = math.cos(ai), math.cos(aj), math.cos(ak)
    cc, cs = ci * ck, ci * sk
    sc, ss = si * ck, si * sk
    M = np.identity(4)
    M[0, 0] = cj * ck
    M[0, 1] = sj * sc - cs
    M[0, 2] = sj * cc + ss
    M[1, 0] = cj * sk
    M[1, 1] = sj * ss + cc
    M[1, 2] = sj * cs - sc
    M[2, 0] = -sj
    M[2, 1] = cj * si
    M[2, 2] = cj * ci
    return M


def get_transform_from_pose(pose, tf_ros=None):
    """utils_ros.py:104-111: pose -> T_pose_to_origin = translation_matrix(t) . quaternion_matrix(q).
    `pose` is a geometry_msgs/Pose-like object or a 7-sequence (tx,ty,tz,qx,qy,qz,qw)."""
    if tf_ros is not None:
        translation = (pose.position.x, pose.position.y, pose.position.z)
        rotation = (pose.orientation.x, pose.orientation.y, pose.orientation.z, pose.orientation.w)
        return tf_ros.fromTranslationRotation(translation, rotation)
    if hasattr(pose, "position"):
        t = (pose.position.x, pose.position.y, pose.position.z)
        q = (pose.orientation.x, pose.orientation.y, pose.orientation.z, pose.orientation.w)
    else:
        t, q = tuple(pose[:3]), tuple(pose[3:7])
    M = np.identity(4)
    M[:3, 3] = t
    return np.dot(M, quaternion_matrix(q))


def create_point_cloud(xyz, rgb=None, frame_id="world"):
    """utils_ros.py:31-59 without ROS: returns a dict shaped like sensor_msgs/PointCloud2 (header.frame_id, height,
    width, fields, point_step, row_step, data bytes).  xyz: n x 3, rgb: n x 3 (or None).  Vectorised host version;
    SemanticMapping.semantic_cloud_device() is the GPU one (avl_pack_semantic_cloud)."""
    xyz = np.asarray(xyz)
    n = xyz.shape[0]
    if rgb is None:
        rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
        fields = [("x", 0, "FLOAT32"), ("y", 4, "FLOAT32"), ("z", 8, "FLOAT32")]
    else:
        rgb = np.asarray(rgb).astype(np.int64)
        rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgba", "<u4")])
        rec["rgba"] = (rgb[:, 0] & 255) | ((rgb[:, 1] & 255) << 8) | ((rgb[:, 2] & 255) << 16) | (255 << 24)
        fields = [("x", 0, "FLOAT32"), ("y", 4, "FLOAT32"), ("z", 8, "FLOAT32"), ("rgba", 12, "UINT32")]
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    return {"header": Header(frame_id=frame_id), "height": 1, "width": n, "fields": fields, "is_bigendian": False,
            "point_step": rec.dtype.itemsize, "row_step": rec.dtype.itemsize * n, "data": rec.tobytes(), "is_dense": True}
