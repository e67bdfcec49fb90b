from .utils import homogenize, dehomogenize  # noqa: F401
from .utils_ros import get_transform_from_pose, euler_matrix, quaternion_matrix, Pose, Header, Stamp, Message, create_point_cloud  # noqa: F401
