"""Output side of the node: what the reference's `utils.cpp` / `mono_vo.cpp` turn the tracker's
results into before they reach ROS topics, as plain Python data (no rclpy dependency):

* `pose_cv_to_ros`      — `affine3d_to_odometry_msg` / `affine3d_to_transform_stamped_msg` (src/utils.cpp:85-188):
                           OpenCV camera frame (Z fwd, X right, Y down) -> REP-103 (X fwd, Y left, Z up) by
                           conjugation with M = [[0,0,1],[-1,0,0],[0,-1,0]], quaternion by tf2's
                           Matrix3x3::getRotation + normalize.
* `odometry`            — the nav_msgs/Odometry fields with the reference's fixed covariances (src/utils.cpp:131-146)
                           and `grow_covariance` = the LOST-state inflation of `publish_odom` (src/mono_vo.cpp:176-190).
* `pointcloud2`         — `points3d_to_pointcloud_msg` (src/utils.cpp:190-243): unordered x,y,z float32 cloud,
                           point_step 12, each point mapped (z, -x, -y).
* `PathAccumulator`     — the growing nav_msgs/Path of `image_callback` (src/mono_vo.cpp:133-152).

A ROS 2 wrapper fills its message objects from these dicts field by field.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

CV_TO_ROS = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])


def rotation_to_quaternion(m) -> np.ndarray:
    """tf2::Matrix3x3::getRotation: (x, y, z, w), branch on the trace / largest diagonal element, then normalised
    as the reference does (`q_ros.normalize()`)."""
    m = np.asarray(m, np.float64).reshape(3, 3)
    trace = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4)
    if trace > 0.0:
        s = math.sqrt(trace + 1.0)
        q[3] = s * 0.5
        s = 0.5 / s
        q[0] = (m[2, 1] - m[1, 2]) * s
        q[1] = (m[0, 2] - m[2, 0]) * s
        q[2] = (m[1, 0] - m[0, 1]) * s
    else:
        # tf2: i = m[0][0] < m[1][1] ? (m[1][1] < m[2][2] ? 2 : 1) : (m[0][0] < m[2][2] ? 2 : 0)
        i = (2 if m[1, 1] < m[2, 2] else 1) if m[0, 0] < m[1, 1] else (2 if m[0, 0] < m[2, 2] else 0)
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = s * 0.5
        s = 0.5 / s
        q[3] = (m[k, j] - m[j, k]) * s
        q[j] = (m[j, i] + m[i, j]) * s
        q[k] = (m[k, i] + m[i, k]) * s
    return q / np.linalg.norm(q)


def pose_cv_to_ros(R_wc, t_wc):
    """Camera pose in the OpenCV world -> (position xyz, orientation quaternion xyzw) in the ROS world."""
    R = np.asarray(R_wc, np.float64).reshape(3, 3)
    t = np.asarray(t_wc, np.float64).reshape(3)
    R_ros = CV_TO_ROS @ R @ CV_TO_ROS.T
    return CV_TO_ROS @ t, rotation_to_quaternion(R_ros)


def odometry(R_wc, t_wc, stamp, frame_id="odom", child_frame_id="base_link") -> dict:
    pos, quat = pose_cv_to_ros(R_wc, t_wc)
    pose_cov = np.zeros(36)
    pose_cov[[0, 7, 14]] = 0.1
    pose_cov[[21, 28, 35]] = 0.05
    twist_cov = np.zeros(36)
    twist_cov[[0, 7, 35]] = 1e-3
    return {"header": {"stamp": stamp, "frame_id": frame_id}, "child_frame_id": child_frame_id,
            "position": pos, "orientation": quat, "pose_covariance": pose_cov, "twist_covariance": twist_cov}


def grow_covariance(odom: dict, seconds_since_valid: float, growth_rate: float) -> dict:
    """publish_odom while LOST: position variances += rate * dt, rotation variances += 0.1 * rate * dt."""
    inc = growth_rate * seconds_since_valid
    out = dict(odom)
    cov = odom["pose_covariance"].copy()
    cov[[0, 7, 14]] += inc
    cov[[21, 28, 35]] += inc * 0.1
    out["pose_covariance"] = cov
    return out


def transform_stamped(R_wc, t_wc, stamp, frame_id="odom", child_frame_id="base_link") -> dict:
    pos, quat = pose_cv_to_ros(R_wc, t_wc)
    return {"header": {"stamp": stamp, "frame_id": frame_id}, "child_frame_id": child_frame_id,
            "translation": pos, "rotation": quat}


def pointcloud2(points_cv, stamp, frame_id="odom") -> dict:
    p = np.asarray(points_cv, np.float32).reshape(-1, 3)
    ros = np.stack([p[:, 2], -p[:, 0], -p[:, 1]], 1).astype("<f4")
    n = len(p)
    return {"header": {"stamp": stamp, "frame_id": frame_id}, "height": 1, "width": n, "is_dense": True,
            "is_bigendian": False,
            "fields": [{"name": "x", "offset": 0, "datatype": 7, "count": 1}, {"name": "y", "offset": 4, "datatype": 7, "count": 1},
                       {"name": "z", "offset": 8, "datatype": 7, "count": 1}],        # 7 = PointField.FLOAT32
            "point_step": 12, "row_step": 12 * n, "data": ros.tobytes()}


@dataclass
class PathAccumulator:
    """nav_msgs/Path as image_callback grows it: one PoseStamped per frame with a valid pose, header = the latest."""
    frame_id: str = "odom"
    poses: list = field(default_factory=list)
    header: dict = field(default_factory=dict)

    def push(self, R_wc, t_wc, stamp) -> dict:
        pos, quat = pose_cv_to_ros(R_wc, t_wc)
        self.header = {"stamp": stamp, "frame_id": self.frame_id}
        self.poses.append({"header": dict(self.header), "position": pos, "orientation": quat})
        return {"header": self.header, "poses": self.poses}
