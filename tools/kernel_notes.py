"""Per-kernel resource notes of the BUILT libavl_hip.so (what the GPU actually runs).

Unbundles every gfx950 code object of the shared library into a scratch directory
(`clang-offload-bundler --unbundle`; nothing is written next to the library) and reads the
AMDGPU metadata notes (`llvm-readelf --notes`): VGPR / AGPR / SGPR counts, spill counts and the
private (scratch) segment size of every kernel.

    python tools/kernel_notes.py [--all] [path/to/libavl_hip.so]

prints the kernels that spill or own scratch (``--all``: every kernel).  `tests/test_abi.py`
imports `kernel_notes()` to assert that the hand-scheduled product kernels are scratch-free.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
_FIELDS = ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
           "private_segment_fixed_size", "group_segment_fixed_size")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names),
                         capture_output=True, text=True, check=True).stdout.split("\n")
    return dict(zip(names, out))


def code_objects(lib, scratch):
    """Write every gfx950 code object bundled in `lib` into `scratch`; return their paths."""
    with open(lib, "rb") as f:
        blob = f.read()
    # a fat binary section holds one bundle per translation unit; each starts with this magic
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(magic, blob)]
    outs = []
    for i, s in enumerate(starts):
        e = starts[i + 1] if i + 1 < len(starts) else len(blob)
        src = os.path.join(scratch, "bundle%d" % i)
        with open(src, "wb") as f:
            f.write(blob[s:e])
        out = os.path.join(scratch, "gfx950_%d.co" % i)
        r = subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o",
                            "--input=" + src, "--output=" + out,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
        if r.returncode == 0 and os.path.getsize(out) > 0:
            outs.append(out)
    return outs


def kernel_notes(lib):
    """{demangled kernel name: {field: int}} over every code object of `lib`."""
    res = {}
    with tempfile.TemporaryDirectory() as scratch:
        for co in code_objects(lib, scratch):
            txt = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", co],
                                 capture_output=True, text=True, check=True).stdout
            for blk in txt.split("- .agpr_count:")[1:]:
                blk = ".agpr_count:" + blk
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                res[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in _FIELDS}
    dm = demangle(list(res))
    return {dm[k]: v for k, v in res.items()}


def disassembly(lib, needle):
    """{mangled symbol: [instruction text, ...]} for every kernel of `lib` whose symbol contains `needle`
    (`llvm-objdump -d` of the unbundled gfx950 code objects)."""
    out = {}
    with tempfile.TemporaryDirectory() as scratch:
        for co in code_objects(lib, scratch):
            txt = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", co],
                                 capture_output=True, text=True, check=True).stdout
            cur = None
            for line in txt.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1) if needle in m.group(1) and not m.group(1).endswith(".kd") else None
                    if cur:
                        out[cur] = []
                elif cur and line.strip():
                    out[cur].append(line.split("//")[0].strip())
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                            "vision_semantic_segmentation_amd", "libavl_hip.so")
    show_all = "--all" in sys.argv
    for name, n in sorted(kernel_notes(lib).items()):
        dirty = n["vgpr_spill_count"] or n["sgpr_spill_count"] or n["private_segment_fixed_size"]
        if show_all or dirty:
            short = name.replace("avl::(anonymous namespace)::", "")
            print("%-90s vgpr %3d agpr %3d sgpr %3d spill v%d s%d scratch %d lds %d" % (
                short[:90], n["vgpr_count"], n["agpr_count"], n["sgpr_count"], n["vgpr_spill_count"],
                n["sgpr_spill_count"], n["private_segment_fixed_size"], n["group_segment_fixed_size"]))


if __name__ == "__main__":
    main()
