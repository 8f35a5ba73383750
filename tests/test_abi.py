"""The C-ABI library loads and exports every symbol include/mvo.h declares (no compute calls: runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mvo.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mvo_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("mvo_create", "mvo_orb_detect_and_compute", "mvo_match_knn2_ratio", "mvo_lk_track",
                 "mvo_find_homography_ransac", "mvo_find_fundamental_ransac", "mvo_solve_pnp_ransac",
                 "mvo_find_essential_ransac", "mvo_recover_pose", "mvo_triangulate", "mvo_batch_track"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from ros2_mono_vo_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_lib.ABI_SYMBOLS) == declared_symbols()


def test_config_mirror_matches_header_layout():
    """ctypes Config must have the size the C struct has (mvo_config_default writes the whole struct)."""
    from ros2_mono_vo_amd import _lib
    c = _lib.default_config()
    assert (c.max_width, c.max_height, c.batch, c.nfeatures, c.fast_threshold) == (1280, 720, 1, 1000, 20)
    assert (c.lk_channels, c.lk_win, c.lk_max_level, c.lk_max_count) == (3, 21, 3, 30)
    assert abs(c.lk_epsilon - 0.01) < 1e-15 and abs(c.lk_min_eig - 1e-4) < 1e-18
    # reference defaults: include/mono_vo/tracker.hpp:137-147, include/mono_vo/initializer.hpp:109-115
    assert c.tracking_error_thresh == 30.0 and c.min_observations_before_triangulation == 100
    assert c.min_tracked_points == 10 and c.max_tracking_after_keyframe == 10
    assert abs(c.max_rotation_from_keyframe - 15.0 * 3.141592653589793 / 180.0) < 1e-15
    assert c.max_translation_from_keyframe == 1.0 and c.ransac_reproj_thresh == 1.0
    assert c.model_score_thresh == 0.85 and c.f_inlier_thresh == 0.5 and c.lowes_distance_ratio == 0.7
    assert c.occupancy_grid_div == 50 and c.kp_distribution_thresh == 0.5 and c.min_matches_for_init == 100
    assert c.init_model_score_thresh == 0.56 and c.device == -1 and c.ring_frames == 0


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: creating a context without a HIP device raises instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ros2_mono_vo_amd import Context, MvoError
    with pytest.raises(MvoError):
        Context()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "ros2_mono_vo_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_py" not in txt and "liborc" not in txt and "mvo_oracle.h" not in txt, f


def test_header_is_plain_c_and_cxx(tmp_path):
    """include/mvo.h is the FFI contract: it must compile as C11 and as C++17 on its own (no HIP, no torch, POD only),
    and a C caller must be able to reference every declared entry point."""
    import re
    import subprocess
    names = declared_symbols()
    assert len(names) >= 25
    body = "#include \"mvo.h\"\n#include <stddef.h>\nvoid* use_all(void) {\n  void* p = NULL;\n" + \
           "".join(f"  p = (void*)&{n};\n" for n in names) + "  return p;\n}\n"
    for name, cc, std in (("t.c", "gcc", "-std=c11"), ("t.cpp", "g++", "-std=c++17")):
        src = tmp_path / name
        src.write_text(body)
        subprocess.check_call([cc, std, "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])
