"""mvo_run, the ROS-free harness: frame sources and argument handling on the CPU; the full run needs the GPU."""
import numpy as np
import pytest

from ros2_mono_vo_amd import mvo_run


def test_raw_reader_and_encodings(tmp_path):
    rng = np.random.default_rng(1)
    fr = rng.integers(0, 256, (3, 6, 8, 4), dtype=np.uint8)
    p = tmp_path / "f.raw"
    p.write_bytes(fr.tobytes())
    got = list(mvo_run.raw_frames(str(p), 8, 6, "rgba8"))
    assert len(got) == 3 and all(np.array_equal(a, b) for a, b in zip(got, fr))
    bgr = mvo_run.to_bgr8(got[0], "rgba8")
    assert bgr.shape == (6, 8, 3) and np.array_equal(bgr[..., 0], fr[0][..., 2]) and np.array_equal(bgr[..., 2], fr[0][..., 0])
    assert np.array_equal(mvo_run.to_bgr8(got[0], "bgra8"), fr[0][..., :3])
    assert np.array_equal(mvo_run.to_bgr8(fr[0][..., :3], "rgb8"), fr[0][..., 2::-1])
    mono = list(mvo_run.raw_frames(str(p), 8, 6 * 4, "mono8"))
    assert len(mono) == 3 and mono[0].shape == (24, 8)
    p.write_bytes(fr.tobytes()[:-5])
    with pytest.raises(ValueError):
        list(mvo_run.raw_frames(str(p), 8, 6, "rgba8"))


def test_arguments_and_intrinsics():
    a = mvo_run.parse_args(["--synthetic", "parallax", "--frames", "5"])
    K = mvo_run.intrinsics(a)
    assert K[0, 0] == 0.9 * 640 and K[0, 2] == 320 and K[1, 2] == 240
    a = mvo_run.parse_args(["--raw", "x", "--width", "1241", "--height", "376", "--intrinsics", "718.856", "718.856", "607.1928", "185.2157"])
    assert mvo_run.intrinsics(a)[1, 2] == 185.2157
    with pytest.raises(SystemExit):
        mvo_run.parse_args([])   # a frame source is required
    frames = list(mvo_run.synthetic_frames("plane", 64, 48, 2, 3))
    assert len(frames) == 2 and frames[0].shape == (48, 64) and frames[0].dtype == np.uint8


@pytest.mark.gpu
def test_run_parallax_sequence(tmp_path, capsys):
    out = tmp_path / "traj.txt"
    assert mvo_run.main(["--synthetic", "parallax", "--frames", "8", "--tum", str(out)]) == 0
    text = capsys.readouterr().out
    assert "tracker=TRACKING" in text and "key-frames" in text
    rows = np.loadtxt(out, ndmin=2)
    assert rows.shape[1] == 8 and len(rows) >= 3
    assert np.allclose(np.linalg.norm(rows[:, 4:], axis=1), 1.0, atol=1e-6)
