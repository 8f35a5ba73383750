"""Output-side mirror: REP-103 conversion, PointCloud2 packing, path accumulation."""
import struct

import numpy as np

from ros2_mono_vo_amd import ros_io, synth


def _quat_to_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_pose_cv_to_ros_axes_and_round_trip():
    # camera 2 m forward (cv +Z), 1 m right (cv +X), 0.5 m down (cv +Y): ROS x = 2, y = -1, z = -0.5
    pos, quat = ros_io.pose_cv_to_ros(np.eye(3), [1.0, 0.5, 2.0])
    assert np.allclose(pos, [2.0, -1.0, -0.5]) and np.allclose(quat, [0, 0, 0, 1])
    # a yaw to the right in the camera frame (about cv +Y, down) is a negative rotation about ROS +Z (up)
    R = synth.rot_y(10.0)
    pos, quat = ros_io.pose_cv_to_ros(R, [0, 0, 0])
    Rr = _quat_to_mat(quat)
    assert np.allclose(Rr, ros_io.CV_TO_ROS @ R @ ros_io.CV_TO_ROS.T, atol=1e-12)
    assert abs(np.linalg.norm(quat) - 1) < 1e-15 and quat[2] < 0 and abs(quat[0]) < 1e-12 and abs(quat[1]) < 1e-12
    # all four branches of Matrix3x3::getRotation
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = rng.normal(size=3); a /= np.linalg.norm(a)
        th = rng.uniform(-np.pi, np.pi)
        Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        Rm = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        q = ros_io.rotation_to_quaternion(Rm)
        assert np.allclose(_quat_to_mat(q), Rm, atol=1e-12)


def test_odometry_covariances_and_lost_growth():
    od = ros_io.odometry(np.eye(3), [0, 0, 1], stamp=12.5)
    assert od["position"][0] == 1.0 and od["child_frame_id"] == "base_link"
    assert list(od["pose_covariance"][[0, 7, 14, 21, 28, 35]]) == [0.1, 0.1, 0.1, 0.05, 0.05, 0.05]
    assert list(od["twist_covariance"][[0, 7, 35]]) == [1e-3, 1e-3, 1e-3] and od["twist_covariance"].sum() == 3e-3
    g = ros_io.grow_covariance(od, seconds_since_valid=2.0, growth_rate=0.5)
    assert np.allclose(g["pose_covariance"][[0, 7, 14]], 1.1) and np.allclose(g["pose_covariance"][[21, 28, 35]], 0.15)
    assert od["pose_covariance"][0] == 0.1       # the original message is not modified


def test_pointcloud2_layout():
    pts = np.array([[1, 2, 3], [-4, 5, 6.5]], np.float32)
    pc = ros_io.pointcloud2(pts, stamp=0)
    assert (pc["height"], pc["width"], pc["point_step"], pc["row_step"]) == (1, 2, 12, 24) and pc["is_dense"] and not pc["is_bigendian"]
    assert [f["offset"] for f in pc["fields"]] == [0, 4, 8] and [f["name"] for f in pc["fields"]] == ["x", "y", "z"]
    assert struct.unpack("<6f", pc["data"]) == (3.0, -1.0, -2.0, 6.5, 4.0, -5.0)
    assert ros_io.pointcloud2(np.zeros((0, 3)), 0)["data"] == b""


def test_path_accumulates():
    pa = ros_io.PathAccumulator()
    pa.push(np.eye(3), [0, 0, 0], 1.0)
    msg = pa.push(np.eye(3), [0, 0, 0.3], 2.0)
    assert len(msg["poses"]) == 2 and msg["header"]["stamp"] == 2.0 and msg["poses"][1]["position"][0] == 0.3
