"""ros2_mono_vo_amd — MI355X-native per-frame VO front-end (ORB, LK, Hamming matcher, RANSAC H/F, PnP,
recoverPose) behind the Tracker / Initializer / FeatureProcessor surface of Tatsuya-2/ros2_mono_vo.

The compute lives in libmvo_hip.so (hand-written gfx950 HIP kernels behind the C ABI of include/mvo.h).
This package is the Python host mirror; importing `Context` without the built library raises."""
from ._lib import MvoError, default_config, KP_DTYPE, MATCH_DTYPE  # noqa: F401
from .context import Context  # noqa: F401
