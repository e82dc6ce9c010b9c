#!/usr/bin/env python3
"""Static checks of the gfx950 code that libkpb.so ships (r05; DESIGN.md section 3, Run-to-run determinism).

    python scripts/isa_lint.py [library.so | code-object | file.s ...]        (default: keypoint_bench_amd/libkpb.so)

Every kernel of every code object in the library is disassembled (llvm-objdump) and checked for

  E1  the packed-fp32 operand selection gfx950 executes wrongly beside f16 MFMAs: v_pk_{mul,add,fma}_f32 with op_sel[0] = 0, op_sel[1] = 1
      (keypoint_bench_amd/isa_fixup.py says what happens and rewrites it at build time; this is the check that nothing slipped through);
  E2  wait states the hardware needs and does not interlock (counted as LLVM's hazard recogniser counts them: one per instruction, n + 1 for
      `s_nop n`), which hipcc pads in its own code but NOT inside an inline-asm string:
        vector write of a VGPR       -> MFMA reading it as A / B / C              2   (measured r04: 1 behind a plain, 2 behind a packed result)
        vector write of a VGPR       -> v_permlane{16,32}_swap reading it         2
        vector write of a VGPR       -> DPP instruction reading it                 2
        vector write of an SGPR/VCC  -> vector-memory instruction reading it       5   (v_readfirstlane -> global_load_lds base, ...)
        vector write of an SGPR      -> v_readlane / v_writelane lane select       4
        scalar write of M0           -> LDS-DMA (global_load_lds_*, buffer_* lds)  1
        MFMA write of a VGPR         -> vector / LDS / memory read or vector write of it    passes + 3 (f16 / bf16 / i8 inputs), passes + 2 (fp32 / fp64)
      the look-back follows fall-through only (it stops at s_branch / s_endpgm / s_setpc): a hazard across a TAKEN branch is not
      seen -- the branch itself costs more than any of these distances;
and the sources under csrc/ for

  E3  a vector, matrix, LDS-return or memory-return instruction inside an inline-asm string (the compiler neither pads nor counts those);
      allowed in asm here: s_waitcnt, s_nop, s_mov_b32 to / from M0, global_load_lds_dwordx4 (no register result), s_memtime (stamps build).

  E4  (code-object metadata) a kernel that spills vector registers to scratch beyond the KNOWN_SPILLS list below: a spill inside a hot loop is the
      silent way a restructuring loses (r05: a block-1 variant at 96 registers spilled 39 dwords and ran 2x slower; ADVICE r04 asked for the check).
      Kernels with private ARRAYS but no spills (the RANSAC solvers) are not findings.

Informational (no failure): MFMAs whose 16-register destination overlaps their own A / B fragment (legal: A and B are read before the
first result is written; counted because VERDICT r04 asked), packed-fp32 instruction counts.
Exit status 1 on any E finding.  tests/test_isa_lint.py runs it on the built library in the CPU suite.
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keypoint_bench_amd import isa_fixup                    # noqa: E402
from keypoint_bench_amd.build import llvm_bin               # noqa: E402

_REG = re.compile(r"(?<![\w.])([vsa])(\d+)(?![\w\[])|(?<![\w.])([vsa])\[(\d+):(\d+)\]|(?<![\w.])(vcc|exec|m0|scc)(?:_lo|_hi)?(?![\w])")
MFMA_PASSES = [("v_mfma_f32_32x32x16", 8), ("v_mfma_f32_16x16x32", 4), ("v_mfma_f32_32x32x2", 16), ("v_mfma_f32_16x16x4", 8),
               ("v_mfma_f32_32x32x8", 8), ("v_mfma_f32_16x16x16", 4), ("v_mfma_f32_4x4", 2), ("v_mfma_f32_32x32x4", 16),
               ("v_mfma_f32_32x32x1", 16), ("v_mfma_f32_16x16x1", 8), ("v_mfma_f64", 8), ("v_mfma_i32_32x32", 8), ("v_mfma_i32_16x16", 4),
               ("v_mfma_scale", 8), ("v_mfma", 16), ("v_smfmac", 8)]
NO_DEST = ("s_waitcnt", "s_nop", "s_barrier", "s_branch", "s_cbranch", "s_endpgm", "s_cmp", "s_bitcmp", "global_store", "buffer_store", "ds_write",
           "flat_store", "scratch_store", "s_setprio", "s_sleep", "global_load_lds", "s_setreg", "s_sendmsg", "s_code_end", "s_setpc", "s_icache",
           "s_dcache", "buffer_wbl2", "buffer_inv", "s_trap", "s_waitcnt_", "ds_nop", "s_endpgm_saved", "s_set_gpr", "v_nop", "s_wakeup", "s_getpc_dummy")
TWO_DEST = ("v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co", "v_div_scale", "v_mad_u64", "v_mad_i64")
# kernels known to spill, with the most dwords they may: gemm_h's one-tile plain / residual forms at four waves per SIMD (three without spills measured
# slower, conv_mfma.h; the rotary form runs at three since r06), the rarely used radius-7 / 8 forms of the sparse NMS tail (1024 threads: 128 registers)
KNOWN_SPILLS = {"gemm_hILi2ELi1ELi0ELb1E": 3, "gemm_hILi2ELi1ELi1ELb0E": 1, "nms_tailILi7E": 6, "nms_tailILi8E": 215}
ASM_ALLOWED = re.compile(r"^(s_waitcnt|s_nop|s_mov_b32|s_mov_b64|global_load_lds_dwordx4|global_load_lds_dword|s_memtime|s_sleep|s_setprio)\b")


def regs(tok):
    out = set()
    for m in _REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        elif m.group(3):
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
        else:
            out.add((m.group(6), 0))
    return out


class Ins:
    __slots__ = ("op", "ops", "defs", "uses", "text", "valu", "mfma", "vmem", "dpp")


def parse_ins(text):
    t = text.split("//")[0].split(";")[0].strip()
    if not t or t.endswith(":") or t.startswith((".", "<")):
        return None
    op, _, rest = t.partition(" ")
    i = Ins()
    i.op, i.text = op, t
    i.mfma = op.startswith(("v_mfma", "v_smfmac"))
    i.valu = op.startswith("v_") and not i.mfma
    i.vmem = op.startswith(("global_", "buffer_", "flat_", "scratch_"))
    i.dpp = " row_" in t or "quad_perm" in t or "row_mask" in t or "_dpp" in op
    ops = isa_fixup._split_operands(rest)
    ops = [o.split(" ")[0] if not o.startswith("[") else o for o in ops]
    i.ops = ops
    if op.startswith(NO_DEST):
        i.defs, i.uses = set(), set().union(*[regs(o) for o in ops]) if ops else set()
        return i
    nd = 2 if op.startswith(TWO_DEST) else 1
    d, u = set(), set()
    for o in ops[:nd]:
        d |= regs(o)
    for o in ops[nd:]:
        u |= regs(o)
    if op.startswith("v_permlane") and "swap" in op:
        d = regs(ops[0]) | regs(ops[1])
        u = set(d)
    if op.startswith(("v_fmac", "v_mac", "v_dot2c", "v_pk_fmac")):
        u |= regs(ops[0])
    if op.startswith("v_cmp") and op.endswith("_e32"):
        u |= d
        d = {("vcc", 0)}
    if op.startswith(("v_cndmask", "v_addc_co", "v_subb_co", "v_subbrev_co")) and op.endswith("_e32"):
        u.add(("vcc", 0))
    if op.startswith("v_div_fmas"):
        u.add(("vcc", 0))
    if op.startswith(("ds_write", "ds_add", "ds_max", "ds_min")) and "rtn" not in op:
        u |= d
        d = set()
    i.defs, i.uses = d, u
    return i


def wait_states(i):
    return int(i.ops[0], 0) + 1 if i.op == "s_nop" else 1


def passes_of(op):
    for prefix, p in MFMA_PASSES:
        if op.startswith(prefix):
            return p
    return 16


def kernels_of_disassembly(text):
    """{kernel name: [Ins]} from llvm-objdump -d or hipcc -S text."""
    out, cur = {}, None
    for line in text.split("\n"):
        m = re.match(r"^(?:[0-9a-f]+ )?<([\w.$]+)>:\s*$", line) or re.match(r"^([A-Za-z_][\w.$]*):\s*(;.*)?$", line)
        if m and not m.group(1).startswith((".L", "L")):
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end") or "s_code_end" in line:
            cur = None
            continue
        i = parse_ins(line)
        if i:
            cur.append(i)
    return {k: v for k, v in out.items() if v}


def lint_kernel(name, ins):
    errors, info = [], {"mfma_dst_overlaps_ab": 0, "packed_fp32": 0}

    def back(k, limit):
        """(instruction, wait states between it and ins[k]) walking back over fall-through predecessors."""
        ws = 0
        for j in range(k - 1, -1, -1):
            x = ins[j]
            if x.op.startswith(("s_branch", "s_endpgm", "s_setpc")):
                return
            yield x, ws
            ws += wait_states(x)
            if ws >= limit:
                return

    for k, i in enumerate(ins):
        if i.op.startswith("v_pk_") and i.op.split("_e64")[0].endswith("_f32"):
            info["packed_fp32"] += 1
            if isa_fixup.is_affected("\t" + i.text):
                errors.append(("E1", "packed fp32 with op_sel [0,1,.]", i.text))
            elif isa_fixup.is_unparsed("\t" + i.text):       # a v_pk_{mul,add,fma}_f32 the parser cannot read is an error, not a pass
                errors.append(("E1", "packed fp32 instruction the lint cannot parse", i.text))
        vuse = {r for r in i.uses if r[0] == "v"}
        suse = {r for r in i.uses if r[0] in ("s", "vcc")}
        if i.mfma:
            d, a, b = regs(i.ops[0]), regs(i.ops[1]), regs(i.ops[2])
            if d & (a | b):
                info["mfma_dst_overlaps_ab"] += 1
            for x, ws in back(k, 2):
                if x.valu and ({r for r in x.defs if r[0] == "v"} & vuse):
                    errors.append(("E2", "vector write -> MFMA operand at %d wait state(s), 2 needed" % ws, x.text + "  ...  " + i.text))
        if i.op.startswith("v_permlane") and "swap" in i.op or i.dpp:
            for x, ws in back(k, 2):
                if x.valu and ({r for r in x.defs if r[0] == "v"} & vuse):
                    errors.append(("E2", "vector write -> %s at %d wait state(s), 2 needed" % ("DPP" if i.dpp else "permlane swap", ws), x.text + "  ...  " + i.text))
        if i.vmem and suse:
            for x, ws in back(k, 5):
                if x.valu and (x.defs & suse):
                    errors.append(("E2", "vector write of a scalar register -> vector memory at %d wait state(s), 5 needed" % ws, x.text + "  ...  " + i.text))
        if i.op.startswith(("v_readlane", "v_writelane")) and len(i.ops) == 3 and regs(i.ops[2]) & suse:       # the lane-select operand
            for x, ws in back(k, 4):
                if x.valu and (x.defs & regs(i.ops[2])):
                    errors.append(("E2", "vector write of a scalar register -> lane select at %d wait state(s), 4 needed" % ws, x.text + "  ...  " + i.text))
        if i.op.startswith("global_load_lds") or (i.op.startswith("buffer_load") and " lds" in i.text):
            for x, ws in back(k, 1):
                if x.op.startswith("s_") and ("m0", 0) in x.defs:
                    errors.append(("E2", "M0 write -> LDS-DMA at %d wait state(s), 1 needed" % ws, x.text + "  ...  " + i.text))
        if not i.mfma and (vuse or i.valu):
            touched = vuse | ({r for r in i.defs if r[0] == "v"} if i.valu else set())
            if touched:
                for x, ws in back(k, 19):
                    if x.mfma:
                        need = passes_of(x.op) + (2 if re.match(r"v_mfma_f(32|64)_\d+x\d+x\d+_?f(32|64)", x.op) else 3)      # fp32 / fp64 inputs: not on the XDL pipe
                        if ws < need and (regs(x.ops[0]) & touched):
                            errors.append(("E2", "MFMA result touched at %d wait state(s), %d needed" % (ws, need), x.text + "  ...  " + i.text))
    return errors, info


def lint_sources(csrc):
    errors = []
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        src = open(path).read()
        for m in re.finditer(r"\basm\s+(?:volatile\s*)?\(", src):
            # the string literals of the statement's first argument (up to the first ':' outside a literal)
            j, depth, lits, cur, in_s = m.end(), 1, [], "", False
            while j < len(src) and depth:
                ch = src[j]
                if in_s:
                    if ch == "\\":
                        cur += src[j:j + 2]
                        j += 2
                        continue
                    if ch == '"':
                        in_s = False
                        lits.append(cur)
                        cur = ""
                    else:
                        cur += ch
                elif ch == '"':
                    in_s = True
                elif ch == ":":
                    break
                elif ch == "(":
                    depth += 1
                elif ch == ")":
                    depth -= 1
                j += 1
            text = "".join(lits).replace("\\n", "\n").replace("\\t", " ")
            line_no = src.count("\n", 0, m.start()) + 1
            for ln in text.split("\n"):
                ln = ln.strip()
                if ln and not ASM_ALLOWED.match(ln):
                    errors.append(("E3", "%s:%d: instruction in inline asm outside the allowed list" % (os.path.relpath(path, ROOT), line_no), ln))
    return errors


def disassemble(path, workdir):
    """Disassembly texts of every gfx950 code object in `path` (a host shared library / object with offload bundles, a bare code object, or a .s file)."""
    if path.endswith(".s"):
        return [open(path).read()]
    tools = llvm_bin()
    local = os.path.join(workdir, os.path.basename(path))
    shutil.copy(path, local)
    subprocess.run([os.path.join(tools, "llvm-objdump"), "--offloading", local], cwd=workdir, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cos = sorted(glob.glob(local + ".*gfx950*")) or [local]
    texts = []
    for co in cos:
        r = subprocess.run([os.path.join(tools, "llvm-objdump"), "-d", "--no-show-raw-insn", co], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        texts.append(r.stdout)
    return texts


def spill_counts(path, workdir):
    """{kernel: vector registers spilled} from the code objects' metadata (llvm-readelf --notes) or a .s file's amdhsa.kernels section."""
    if path.endswith(".s"):
        metas = [open(path).read()]
    else:
        tools = llvm_bin()
        local = os.path.join(workdir, os.path.basename(path))
        cos = sorted(glob.glob(local + ".*gfx950*")) or [local]        # disassemble() has extracted them
        metas = [subprocess.run([os.path.join(tools, "llvm-readelf"), "--notes", co], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout for co in cos]
    out = {}
    for meta in metas:
        for m in re.finditer(r"\.name:\s+(\S+)(?:(?!\.name:).)*?\.vgpr_spill_count:\s+(\d+)", meta, re.S):
            out[m.group(1)] = max(out.get(m.group(1), 0), int(m.group(2)))
    return out


def lint(paths, check_sources=True, verbose=True):
    all_errors, totals, nk = [], {"mfma_dst_overlaps_ab": 0, "packed_fp32": 0}, 0
    with tempfile.TemporaryDirectory() as wd:
        for p in paths:
            for text in disassemble(p, wd):
                for name, ins in kernels_of_disassembly(text).items():
                    nk += 1
                    errs, info = lint_kernel(name, ins)
                    for k in totals:
                        totals[k] += info[k]
                    all_errors += [(e[0], name, e[1], e[2]) for e in errs]
            for name, n in spill_counts(p, wd).items():
                allowed = max([v for k, v in KNOWN_SPILLS.items() if k in name] or [0])
                if n > allowed:
                    all_errors.append(("E4", name, "%d vector registers spilled to scratch (allowed: %d)" % (n, allowed), os.path.basename(p)))
    if check_sources:
        all_errors += [(e[0], "-", e[1], e[2]) for e in lint_sources(os.path.join(ROOT, "keypoint_bench_amd", "csrc"))]
    if verbose:
        for code, kernel, what, where in all_errors:
            print("%s  %s\n      %s\n      %s" % (code, kernel, what, where))
        print("isa_lint: %d kernels, %d finding(s); info: %d packed-fp32 instructions, %d MFMAs whose destination overlaps their own A / B fragment"
              % (nk, len(all_errors), totals["packed_fp32"], totals["mfma_dst_overlaps_ab"]))
    return all_errors, totals, nk


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    errs, _, _ = lint(args or [os.path.join(ROOT, "keypoint_bench_amd", "libkpb.so")], check_sources="--no-sources" not in sys.argv)
    sys.exit(1 if errs else 0)
