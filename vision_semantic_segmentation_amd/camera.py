"""Pinhole camera and the two hard-coded calibrations of the reference (src/camera.py:21-35,102-135).

Host-side float64 setup only: the kernels receive ``Camera.P`` (3x4, row-major) by value.
"""
import numpy as np


class Camera(object):
    """src/camera.py:21-35: P = K [R | t]; T = [[R t],[0 1]] (velodyne -> camera)."""

    def __init__(self, K, R, t, imSize=None, id=0, dist=None):
        self.id = id
        self.K = K
        self.R = R
        self.t = t
        self.P_norm = np.concatenate([R, t], axis=1)
        self.P = np.matmul(K, self.P_norm)
        self.T = np.vstack([self.P_norm, np.zeros((1, self.P_norm.shape[1]))])
        self.T[-1, -1] = 1
        self.K_inv = np.linalg.inv(self.K)
        self.C_world_inhomo = np.matmul(-R.T, t)
        self.imSize = imSize
        self.dist = dist

    def get_image_coordinate(self, X):
        """src/camera.py:87-91: image coordinates of 3xN world points."""
        x_homo = np.matmul(self.P, np.vstack((X, np.ones((1, X.shape[1])))))
        return x_homo[:-1] / x_homo[-1]

    def scaled(self, sx, sy, imSize=None):
        """A camera whose image is resized by (sx, sy): K rows scaled (used with IMAGE_SCALE / synthetic sizes)."""
        K = self.K.copy()
        K[0] *= sx
        K[1] *= sy
        return Camera(K, self.R, self.t, imSize=imSize, id=self.id, dist=self.dist)


def _from_autoware(K, Rt, dist, cam_id):
    # src/camera.py:111-112 / :129-130: the calibration file stores camera -> velodyne
    R = Rt[0:3, 0:3].T
    t = -np.matmul(R, Rt[0:3, 3:4])
    return Camera(K, R, t, imSize=[1920, 1440], id=cam_id, dist=dist)


def camera_setup_1():
    """src/camera.py:102-117"""
    K = np.array([[1826.998004, 0.000000, 1174.548672],
                  [0.000000, 1802.603136, 776.028597],
                  [0.000000, 0.000000, 1.000000]])
    Rt = np.array([[1.5426360183850896e-01, -6.8597082105982421e-02, 9.8564556584725482e-01, 4.7539938241243362e-02],
                   [-9.8802970661938061e-01, -1.0912135033489312e-02, 1.5387730224640517e-01, 3.1389930844306946e-01],
                   [1.9996357324159053e-04, -9.9758476614047986e-01, -6.9459300162133530e-02, -5.5608768016099930e-02],
                   [0., 0., 0., 1.]])
    dist = np.array([-0.136981, 0.043159, 0.006235, 0.018954, 0.000000])
    return _from_autoware(K, Rt, dist, 1)


def camera_setup_6():
    """src/camera.py:120-135"""
    K = np.array([[1790.634474, 0., 973.099292],
                  [0., 1785.950534, 803.294457],
                  [0., 0., 1.]])
    Rt = np.array([[-2.1022535018250471e-01, -9.2112145235168197e-02, 9.7330398891652492e-01, -1.4076865278184414e-02],
                   [-9.7735897207277012e-01, -4.6117027185500481e-03, -2.1153763709301088e-01, -3.1732881069183350e-01],
                   [2.3973774202277975e-02, -9.9573795995643932e-01, -8.9057134763516621e-02, -7.2184838354587555e-02],
                   [0., 0., 0., 1.]])
    dist = np.array([-0.191070, 0.100324, 0.004250, -0.003317, 0.000000])
    return _from_autoware(K, Rt, dist, 6)
