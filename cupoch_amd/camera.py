"""cupoch.camera.PinholeCameraIntrinsic mirror (src/cupoch/camera/pinhole_camera_intrinsic.h:40-120,
.cpp:40-92) -- the part the depth-image factories and the KinFu pose estimation use."""
import numpy as np


class PinholeCameraIntrinsic:
    def __init__(self, width=-1, height=-1, fx=0.0, fy=0.0, cx=0.0, cy=0.0):
        self.width, self.height = int(width), int(height)
        self.intrinsic_matrix = np.eye(3, dtype=np.float32)
        self.set_intrinsics(width, height, fx, fy, cx, cy)

    def set_intrinsics(self, width, height, fx, fy, cx, cy):
        self.width, self.height = int(width), int(height)
        m = np.eye(3, dtype=np.float32)
        m[0, 0], m[1, 1], m[0, 2], m[1, 2] = fx, fy, cx, cy
        self.intrinsic_matrix = m

    def get_focal_length(self):
        return float(self.intrinsic_matrix[0, 0]), float(self.intrinsic_matrix[1, 1])

    def get_principal_point(self):
        return float(self.intrinsic_matrix[0, 2]), float(self.intrinsic_matrix[1, 2])

    def is_valid(self):
        return self.width > 0 and self.height > 0

    def as4(self):
        """fx, fy, cx, cy -- what mi_icp_create_from_depth takes"""
        m = self.intrinsic_matrix
        return [float(m[0, 0]), float(m[1, 1]), float(m[0, 2]), float(m[1, 2])]

    def create_pyramid_level(self, level):
        """pinhole_camera_intrinsic.cpp:82-92 (fp32 arithmetic as there)"""
        level = int(level)
        if level == 0 or self.width <= 0 or self.height <= 0:
            return PinholeCameraIntrinsic(self.width, self.height, *self.as4())
        s = np.float32(np.float32(0.5) ** np.float32(level))
        h = np.float32(0.5)
        m = self.intrinsic_matrix
        return PinholeCameraIntrinsic(self.width >> level, self.height >> level, m[0, 0] * s, m[1, 1] * s,
                                      (m[0, 2] + h) * s - h, (m[1, 2] + h) * s - h)
