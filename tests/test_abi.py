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


def test_plan_validation_refuses_this_rounds_removed_and_restricted_forms():
    """avl_seg_plan_create validates every op on the host (no GPU call): the fused depthwise op no longer takes f16 weight pairs
    (w_split 2), block tiles (w_layout 1) exist for a split input only, and the classifier's fused arg-max (out_f32 + out_mx = labels)
    needs N <= 32 and neither residual nor ReLU."""
    import ctypes as C
    import torch
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import OP_DWPW, OP_GEMM, AvlSegOp
    buf = torch.zeros(1 << 16, dtype=torch.uint8)
    ptr = (buf.data_ptr() + 255) // 256 * 256

    def create(op):
        plan = C.c_void_p()
        rc = _lib.lib().avl_seg_plan_create((AvlSegOp * 1)(op), 1, C.byref(plan))
        if rc == 0:
            _lib.lib().avl_seg_plan_destroy(plan)
        return rc, _lib.last_error()

    op = AvlSegOp()
    op.kind, op.dtype = OP_DWPW, _lib.AVL_F16
    op.in_ = op.in2 = op.out = op.weight = op.bias = ptr
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = 8, 8, 64, 64, 64
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = 8, 8, 256, 256, 256
    op.relu, op.w_rows, op.ksize, op.stride, op.pad, op.dil, op.groups, op.w_split = 1, 256, 3, 1, 1, 1, 64, 2
    rc, msg = create(op)
    assert rc == -1 and "w_split 2" in msg and "w_split 3" in msg
    op.w_split, op.w_layout = 3, 1
    rc, msg = create(op)
    assert rc == -1 and "w_layout 1" in msg
    op.w_layout = 0
    assert create(op)[0] == 0
    op.in_lo, op.w_layout = ptr, 1
    assert create(op)[0] == 0

    g = AvlSegOp()
    g.kind, g.dtype = OP_GEMM, _lib.AVL_F16
    g.in_ = g.out = g.weight = g.bias = g.out_mx = ptr
    g.in_h, g.in_w, g.in_c, g.in_ld, g.in_rows = 1, 256, 256, 256, 256
    g.out_h, g.out_w, g.out_c, g.out_ld, g.out_rows = 1, 256, 19, 19, 256
    g.out_f32, g.w_rows, g.ksize, g.stride, g.dil, g.groups = 1, 64, 1, 1, 1, 1
    assert create(g)[0] == 0
    g.relu = 1
    rc, msg = create(g)
    assert rc == -1 and "arg-max" in msg
    g.relu, g.out_c, g.out_ld = 0, 48, 48
    rc, msg = create(g)
    assert rc == -1 and "arg-max" in msg


def test_missing_library_fails_loudly(monkeypatch):
    from vision_semantic_segmentation_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libavl_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


SO = os.path.join(ROOT, "vision_semantic_segmentation_amd", "libavl_hip.so")
LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def _notes():
    import sys
    if not os.path.exists(SO):
        pytest.fail("libavl_hip.so is not built: run __graft_entry__.build()")
    if not os.path.exists(os.path.join(LLVM_BIN, "llvm-readelf")):
        pytest.skip("ROCm LLVM tools not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_notes
    return kernel_notes


def test_no_kernel_of_the_built_library_spills_or_owns_scratch():
    """The MX / ring GEMMs, the grouped conv and the fused depthwise kernels wait for their LDS-DMA with hand-counted
    `s_waitcnt vmcnt(N)`.  A VGPR spill puts scratch loads and stores on the same in-order counter (and hipcc drains `vmcnt(0)`
    behind a reload, i.e. every DMA burst in flight), so the SHIPPED code objects must be scratch-free: read from the notes of the
    built libavl_hip.so (`llvm-readelf --notes` on its unbundled gfx950 code objects), not from a recompilation with other flags.
    VERDICT r3: k_gemm_mx_pipe<1,8,0,0> spilled 6 VGPRs (28 bytes of scratch) and the old test never looked at it."""
    kn = _notes()
    notes = kn.kernel_notes(SO)
    assert len(notes) > 50
    for family in ("k_gemm_mx_pipe", "k_gemm_ring_mx", "k_gemm_ring", "k_gconv_mfma", "k_dwpw", "k_dwpw_x", "k_stem_mfma", "k_fused_vote", "k_bottleneck"):
        assert any(family in n for n in notes), "no %s kernel in the built library" % family
    dirty = {n: v for n, v in notes.items() if v["vgpr_spill_count"] or v["private_segment_fixed_size"]}
    assert not dirty, "kernels with VGPR spills / scratch in libavl_hip.so: %s" % dirty
    for n, v in notes.items():
        assert v["vgpr_count"] + 0 <= 512 and v["agpr_count"] <= 256
        # SGPR spills are v_readlane / v_writelane inside the instruction stream: 99 .. 121 in the shipped MX pipe kernels,
        # none in the fused bottleneck; a jump past these bounds is a scheduling regression worth a look
        if "k_gemm_mx_pipe" in n:
            assert v["sgpr_spill_count"] <= 130, (n, v["sgpr_spill_count"])
        if "k_bottleneck" in n:
            assert v["sgpr_spill_count"] == 0, (n, v["sgpr_spill_count"])


def test_mx_pipe_stream_holds_no_scalar_loads_and_no_scratch():
    """Inside the software-pipelined stream of k_gemm_mx_pipe (first to last MFMA of the kernel: K loops, tile switches and
    epilogues) there must be no kernarg / scalar load (SMEM completes out of order on lgkmcnt: hipcc would put `lgkmcnt(0)` in
    front of LDS fragment uses) and no scratch access (vmcnt).  Checked on the disassembly of the built library."""
    kn = _notes()
    dis = kn.disassembly(SO, "k_gemm_mx_pipe")
    assert len(dis) >= 4
    for sym, ins in dis.items():
        mf = [i for i, l in enumerate(ins) if l.startswith("v_mfma")]
        assert len(mf) >= 64, sym
        stream = ins[mf[0]:mf[-1]]
        bad = [l for l in stream if l.startswith(("s_load", "s_buffer_load", "scratch_"))]
        assert not bad, "%s: %s" % (sym, bad[:4])
        assert not any(l.startswith("scratch_") for l in ins), sym
        # the two hand-counted events are there (a compiler change that drops or rewrites the asm would show here)
        assert any(l.startswith("s_waitcnt vmcnt(4)") for l in stream) or "Li4ELi" in sym, sym


def test_release_library_reads_no_environment_and_holds_no_experiment_kernels():
    """VERDICT r3 / ADVICE r3: timing probes (results are garbage by design), A/B switches and the k_gemm_w4 experiment live in
    the experiments build only (`make experiments`, -DAVL_EXPERIMENTS -> libavl_hip_exp.so): the release library must not even
    import getenv, and must not contain a probe instantiation of the MX GEMM or the experiment kernel."""
    import subprocess
    kn = _notes()
    undefined = subprocess.run(["nm", "-D", "--undefined-only", SO], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined
    strings = subprocess.run(["strings", SO], capture_output=True, text=True, check=True).stdout
    for name in ("AVL_MX_PROBE", "AVL_GEMM_PROBE", "AVL_SWEEP_EXP", "AVL_GC_PROBE", "AVL_MX_PIPE", "AVL_MX_LATE", "AVL_MX_TILE",
                 "AVL_MX_STAGGER", "AVL_APPLY_MODE", "AVL_MASK_MODE", "AVL_GCONV_TH", "AVL_GC_DEPHASE", "AVL_GC_WALK", "AVL_DW_NCHUNK", "AVL_GEMM_DEEP",
                 "AVL_MX_SPREAD", "AVL_MX_PP", "AVL_MX_GRID", "AVL_SWEEP_VEC", "AVL_MX_SAMETILE", "AVL_BN_PROBE"):
        assert name not in strings, name
    names = list(kn.kernel_notes(SO))
    assert not any("k_gemm_w4" in n for n in names)
    assert not any("k_gemm_mx_pp" in n for n in names)
    import re as _re
    pipe = [_re.search(r"k_gemm_mx_pipe<(\d+), (\d+), (\d+), (\d+), (\d+)>", n) for n in names if "k_gemm_mx_pipe<" in n]
    assert len(pipe) >= 4 and all(pipe), "k_gemm_mx_pipe's template list changed: update this test (%s)" % [n for n in names if "k_gemm_mx_pipe<" in n][:2]
    for m in pipe:          # <IO, MI, LATE, PROBE, SPREAD>
        assert m.group(4) == "0" and m.group(5) == "0", "probe / experiment-schedule instantiation in the release library: %s" % m.group(0)
