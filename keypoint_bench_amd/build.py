"""Builds keypoint_bench_amd/libkpb.so (HIP, gfx950 only) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the GPU
box with the repository snapshot.  -ffp-contract=off: the parity contract needs un-fused float
arithmetic where the reference's CPU code has none (kernels that want FMAs call fmaf explicitly).

Each translation unit goes through FOUR steps (only the stale ones, in parallel), then the objects are linked:
  1. hipcc -S --cuda-device-only      -> _obj/X.s        the device code as assembly
  2. isa_fixup.fix_text               -> _obj/X.fixed.s  the packed-fp32 operand selection gfx950 executes wrongly beside MFMAs is
                                                          rewritten (operands swapped: same arithmetic, clean encoding) -- isa_fixup.py
  3. clang -x assembler, lld, clang-offload-bundler      -> the code object, bundled as hipcc would
  4. hipcc --cuda-host-only -fcuda-include-gpubinary     -> _obj/X.o  the host half with that code object embedded
(r05: steps 1 / 3 / 4 are what `hipcc -c` does internally -- `hipcc -### -c` lists them; running them apart is what lets step 2 in.)
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
SO = os.path.join(HERE, "libkpb.so")
SOURCES = ["api.hip", "detect.hip", "match.hip", "net_api.hip", "alike.hip", "convnet.hip", "lightglue.hip", "covis.hip", "lk.hip",
           "preprocess.hip", "geometry.hip"]
# -amdgpu-mfma-vgpr-form: MFMA accumulators in ordinary vector registers.  gfx950's register file is unified, so accumulation
# registers buy no occupancy, and every value that crosses between them and the vector ALUs costs a v_accvgpr_read / _write: 240 of
# them in lg_flash_h (192 -> 145 registers without), 24 in alike_block2, 8 in alike_block1_h (r03, found in the ISA).
# -ffile-prefix-map: the objects record the sources as ./keypoint_bench_amd/csrc/..., not by absolute path -- the library's hash (config.build.lib_sha256
# of every bench line) is then the same wherever the commit is checked out and built (r05)
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-Wall", "-Wno-unused-result", "-fvisibility=hidden", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
         "-ffile-prefix-map=%s=." % os.path.dirname(HERE)]
ARCH = "gfx950"


def llvm_bin():
    """Directory of the ROCm LLVM tools (clang, lld, clang-offload-bundler) that belong to the hipcc on PATH."""
    cands = []
    hipcc = shutil.which("hipcc")
    if hipcc:
        cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin"))
    cands += [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin"), "/opt/rocm/lib/llvm/bin"]
    for c in cands:
        if os.path.exists(os.path.join(c, "clang-offload-bundler")):
            return c
    raise RuntimeError("ROCm LLVM tools not found (looked in %s)" % ", ".join(cands))


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "kpb.h"),
                                                                                     os.path.abspath(__file__), os.path.join(HERE, "isa_fixup.py")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    return _stale(SO, srcs + _headers())


def compile_unit(src, obj, flags=FLAGS, verbose=False, fixup=True):
    """One translation unit through the four steps above; returns the number of instructions the fix-up pass rewrote."""
    try:
        from . import isa_fixup
    except ImportError:                 # run as a script (python keypoint_bench_amd/build.py)
        sys.path.insert(0, HERE)
        import isa_fixup
    tools = llvm_bin()
    stem = obj[:-2] if obj.endswith(".o") else obj

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    # -cuid: the compilation-unit id clang derives from the ABSOLUTE source path by default (it names the module's registration symbols) -- a fixed
    # one per translation unit keeps the library's hash independent of where the tree is checked out; device and host halves must agree on it
    flags = list(flags) + ["-cuid=kpb_" + os.path.splitext(os.path.basename(src))[0]]
    run(["hipcc"] + flags + ["-S", "--cuda-device-only", "-Wno-unused-command-line-argument", "-o", stem + ".s", src])
    with open(stem + ".s") as f:
        text = f.read()
    fixed, n = isa_fixup.fix_text(text) if fixup else (text, 0)
    with open(stem + ".fixed.s", "w") as f:
        f.write(fixed)
    run([os.path.join(tools, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=" + ARCH, "-c", stem + ".fixed.s", "-o", stem + ".dev.o"])
    run([os.path.join(tools, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", stem + ".hsaco", stem + ".dev.o"])
    run([os.path.join(tools, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
         "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--" + ARCH, "-input=/dev/null", "-input=" + stem + ".hsaco",
         "-output=" + stem + ".hipfb"])
    run(["hipcc"] + flags + ["--cuda-host-only", "-Wno-unused-command-line-argument", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", stem + ".hipfb",
                             "-c", src, "-o", obj])
    for ext in (".s", ".dev.o", ".hsaco", ".hipfb"):          # _obj/ travels to the GPU box: keep the object and the assembly that was assembled
        os.remove(stem + ext)
    if verbose or n:
        print("%s: %d packed-fp32 instruction(s) rewritten by isa_fixup" % (os.path.basename(src), n), flush=True)
    return n


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(lambda j: compile_unit(j[0], j[1], verbose=verbose), jobs))
    cmd = ["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", SO] + [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    _write_build_info()
    return SO


def source_hash():
    """sha256 over the bytes that decide what the library is: every file of csrc/, include/kpb.h, this file and isa_fixup.py, in name order
    (names hashed too).  Independent of git and of the checkout path: the record a library carries says what it was built FROM even when it was
    linked a minute before its commit (r05's record read `dirty, parent commit` for ever -- VERDICT r05 weak 12)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files += [os.path.join(HERE, "..", "include", "kpb.h"), os.path.abspath(__file__), os.path.join(HERE, "isa_fixup.py")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def _write_build_info():
    """_obj/build_info.json: what this library was linked from -- `src_sha256` (source_hash(): the sources themselves), plus the commit and whether
    the tree was dirty at link time (informational).  The GPU box receives the tree without .git, so bench.py reads the record from here
    (config.build in its JSON line) and checks src_sha256 against the sources it sees."""
    import json
    import time
    info = {"src_sha256": source_hash(), "git": None, "dirty": None, "time": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())}
    try:
        root = os.path.dirname(HERE)
        info["git"] = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                     timeout=10).stdout.strip() or None
        info["dirty"] = bool(subprocess.run(["git", "-C", root, "status", "--porcelain", "--untracked-files=no"], stdout=subprocess.PIPE,
                                            stderr=subprocess.DEVNULL, text=True, timeout=10).stdout.strip())
    except Exception:
        pass
    with open(os.path.join(OBJ, "build_info.json"), "w") as f:
        json.dump(info, f)


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(SO)
