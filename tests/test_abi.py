"""The C ABI: every function include/avl_hip.h declares is exported by libavl_hip.so and bound by the
ctypes shim (no compute calls here -- there is no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared():
    text = open(os.path.join(ROOT, "include", "avl_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(avl_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    names = declared()
    for must in ("avl_project_pcd", "avl_update_map", "avl_fused_frame", "avl_seg_plan_create", "avl_seg_plan_run", "avl_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    so = os.path.join(ROOT, "vision_semantic_segmentation_amd", "libavl_hip.so")
    if not os.path.exists(so):
        pytest.fail("libavl_hip.so is not built: run __graft_entry__.build()")
    lib = ctypes.CDLL(so)
    for name in declared():
        assert hasattr(lib, name), "libavl_hip.so does not export %s" % name


def test_ctypes_shim_binds_every_declared_symbol():
    from vision_semantic_segmentation_amd import _lib, network  # noqa: F401  (network registers the seg entry points)
    assert sorted(_lib.exported_symbols()) == declared()


def test_argument_errors_do_not_need_a_gpu():
    from vision_semantic_segmentation_amd import _lib
    L = _lib.lib()
    assert L.avl_version().startswith(b"avl_hip")
    rc = L.avl_project_points(None, 5, 7, 16, 4, None, None, 100.0, 10, 10, None, None, None)
    assert rc == -1 and "pts is NULL" in _lib.last_error()
    assert L.avl_project_pcd_scratch_bytes(1000) >= 4000


def test_missing_library_fails_loudly(monkeypatch):
    from vision_semantic_segmentation_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libavl_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()
