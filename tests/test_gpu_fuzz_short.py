"""Short seeds of the two randomized campaigns (tests/fuzz_device_vs_oracle.py, tests/fuzz_tracker_vs_reference.py) under `-m gpu`,
so that every driver run of the GPU suite includes a few hundred random problems against the oracle: LK groups with border /
outside points, findHomography / findFundamentalMat / solvePnPRansac (n = 4, 5, 6, ...) masks and inlier lists bit for bit,
ORB on random crops, the matcher with planted ties, findEssentialMat + recoverPose + triangulatePoints, and two trials of six
random rendered streams through the whole device tracker.  The long campaigns stay a by-hand tool."""
import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_stages_short_seed():
    import fuzz_device_vs_oracle as F
    counts, bad = F.run(seed=3, seconds_a=10, seconds_b=6, verbose=False)
    assert bad == 0, counts
    assert counts["LK"] > 50 and counts["ORB"] > 5 and counts["E"] > 5, counts


def test_fuzz_tracker_short_seed():
    import fuzz_tracker_vs_reference as F
    frames, bad, worst = F.run(seed=4, trials=1, B=4, N=16, verbose=False)
    assert bad == 0 and frames > 30 and worst < 1e-4, (frames, bad, worst)
