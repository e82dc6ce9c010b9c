"""The assembly fix-up pass and the ISA lint (r05; DESIGN.md section 3, Run-to-run determinism).

gfx950 returns a wrong LOW result in lanes 48..63 for v_pk_{mul,add,fma}_f32 with op_sel[0] = 0, op_sel[1] = 1 while an f16 MFMA executes
on the SIMD (scripts/ubench/pk_opsel.hip).  keypoint_bench_amd/isa_fixup.py swaps the two commuting operands of every such instruction
before the translation unit is assembled; scripts/isa_lint.py checks the linked library.  No GPU needed: hipcc's assembler runs here."""
import importlib.util
import itertools
import os
import subprocess

import pytest

from keypoint_bench_amd import build, isa_fixup

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "scripts", "isa_lint.py"))
isa_lint = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(isa_lint)


def _lanes(line):
    """What the low and the high result lane of a packed instruction compute, as symbols (src0 / src1 commute)."""
    _, _, ops, m, _, _ = isa_fixup.parse_packed(line)
    n = len(ops) - 1
    lo = [("-" if m["neg_lo"][i] else "") + ops[1 + i] + (".hi" if m["op_sel"][i] else ".lo") for i in range(n)]
    hi = [("-" if m["neg_hi"][i] else "") + ops[1 + i] + (".hi" if m["op_sel_hi"][i] else ".lo") for i in range(n)]
    return (frozenset(lo[:2]), tuple(lo[2:])), (frozenset(hi[:2]), tuple(hi[2:]))


def test_fixup_rewrites_exactly_the_affected_selection_and_keeps_the_arithmetic():
    def mod(name, v, d):
        return "" if all(x == d for x in v) else " %s:[%s]" % (name, ",".join(map(str, v)))

    checked = 0
    for mnem, n in (("v_pk_mul_f32", 2), ("v_pk_add_f32", 2), ("v_pk_fma_f32", 3)):
        for sel, hi, nl, nh in itertools.product(*[list(itertools.product((0, 1), repeat=n))] * 4):
            line = "\t%s v[4:5], v[0:1], s[2:3]%s" % (mnem, ", v[6:7]" if n == 3 else "")
            line += mod("op_sel", sel, 0) + mod("op_sel_hi", hi, 1) + mod("neg_lo", nl, 0) + mod("neg_hi", nh, 0)
            fixed = isa_fixup.fix_line(line)
            assert _lanes(fixed) == _lanes(line), (line, fixed)
            assert not isa_fixup.is_affected(fixed), fixed
            assert (fixed != line) == (sel[0] == 0 and sel[1] == 1), line
            checked += 1
    assert checked == 2 * 256 + 4096


def test_fixup_leaves_everything_else_alone():
    for line in ("\tv_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]", "\tv_fma_f32 v0, v1, v2, v3", "\tv_pk_fma_f16 v0, v1, v2, v3 op_sel:[0,1,0]",
                 "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]", "; v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]", ""):
        assert isa_fixup.fix_line(line) == line
    text, n = isa_fixup.fix_text("\tv_pk_add_f32 v[12:13], v[12:13], v[12:13] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1] ; a comment\n\ts_endpgm")
    assert n == 1 and "op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]" in text and "; a comment" in text


def test_a_packed_fp32_line_the_parser_cannot_read_is_refused_not_passed():
    """the safety net must not depend on the assembler's text syntax staying what it is -- a v_pk_{mul,add,fma}_f32 line with an
    unknown modifier spelling or operand count is an error for the rewrite (it raises) and for the lint (is_unparsed), never a silent pass."""
    odd = ("\tv_pk_mul_f32 v[0:1], v[2:3]", "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[0,1,0]")
    for line in odd:
        assert isa_fixup.is_unparsed(line) and not isa_fixup.is_affected(line)
        with pytest.raises(ValueError):
            isa_fixup.fix_text("\ts_nop 0\n" + line + "\n")
    for line in ("\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]", "\tv_pk_mov_b32 v[0:1], v[2:3], v[4:5]", "; v_pk_mul_f32 v[0:1], v[2:3]", "\tv_pk_fma_f16 v0, v1"):
        assert not isa_fixup.is_unparsed(line)


def test_the_shipped_library_is_clean():
    so = build.build()
    errors, totals, kernels = isa_lint.lint([so], verbose=False)
    assert kernels > 100 and totals["packed_fp32"] > 1000, "the disassembly did not find the library's kernels"
    assert not errors, errors[:5]


# the head's failing shape in twelve lines: an f16 MFMA in flight, then the splat of a pair's second element as the compiler wrote it
_KERNEL = """
\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.text
\t.globl\tprobe
\t.p2align\t8
\t.type\tprobe,@function
probe:
\tv_mfma_f32_32x32x16_f16 v[16:31], v[0:3], v[4:7], v[16:31]
\tv_mul_f32_e32 v8, s0, v9
%s
\ts_nop 7
\ts_nop 3
\tv_add_f32_e32 v32, v16, v17
\ts_endpgm
.Lfunc_end0:
"""


def _assemble(tmp_path, body):
    s = tmp_path / "probe.s"
    s.write_text(_KERNEL % body)
    obj = tmp_path / "probe.o"
    subprocess.check_call([os.path.join(build.llvm_bin(), "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(s), "-o", str(obj)])
    return str(s), str(obj)


def test_lint_is_red_on_the_encoding_that_failed_and_green_once_fixed(tmp_path):
    bad = "\tv_pk_mul_f32 v[10:11], v[8:9], v[12:13] op_sel:[0,1] op_sel_hi:[0,1]"
    s, obj = _assemble(tmp_path, bad)
    for target in (s, obj):                                    # the text hipcc -S writes and the assembled code object alike
        errors, _, kernels = isa_lint.lint([target], check_sources=False, verbose=False)
        assert kernels == 1 and [e[0] for e in errors] == ["E1"], (target, errors)
    fixed, n = isa_fixup.fix_text(_KERNEL % bad)
    assert n == 1
    (tmp_path / "fixed.s").write_text(fixed)
    errors, _, _ = isa_lint.lint([str(tmp_path / "fixed.s")], check_sources=False, verbose=False)
    assert not errors, errors


def test_lint_counts_wait_states_in_front_of_matrix_operands_and_results(tmp_path):
    # a vector result used as an MFMA operand by the next instruction (what an inline-asm string can do and hipcc never pads)
    s, _ = _assemble(tmp_path, "\tv_cvt_pk_f16_f32 v0, v8, v9\n\tv_mfma_f32_32x32x16_f16 v[16:31], v[0:3], v[4:7], v[16:31]")
    errors, _, _ = isa_lint.lint([s], check_sources=False, verbose=False)
    assert any(e[0] == "E2" and "MFMA operand" in e[2] for e in errors), errors
    # an MFMA result read three instructions later
    s, _ = _assemble(tmp_path, "\tv_add_f32_e32 v33, v16, v16")
    errors, _, _ = isa_lint.lint([s], check_sources=False, verbose=False)
    assert any(e[0] == "E2" and "MFMA result" in e[2] for e in errors), errors
    s, _ = _assemble(tmp_path, "\tv_mov_b32_e32 v40, v41")
    assert not isa_lint.lint([s], check_sources=False, verbose=False)[0]


def test_no_vector_instruction_hides_in_an_inline_asm_string(tmp_path):
    assert not isa_lint.lint_sources(os.path.join(ROOT, "keypoint_bench_amd", "csrc"))
    (tmp_path / "x.hip").write_text('__device__ void f(float& a, unsigned h) { asm volatile("v_fma_mix_f32 %0, -%1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(a) : "v"(h)); }\n'
                                    '__device__ void g() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }\n')
    errors = isa_lint.lint_sources(str(tmp_path))
    assert len(errors) == 1 and errors[0][0] == "E3" and "v_fma_mix_f32" in errors[0][2]


def test_a_kernel_that_spills_vector_registers_is_a_finding(tmp_path):
    """E4 reads the code object's metadata: vector registers spilled to scratch beyond the short list of kernels known to.  The text is what
    hipcc -S appends to a translation unit (amdhsa.kernels); the shipped library's count is part of test_the_shipped_library_is_clean."""
    meta = ("\n\t.amdgpu_metadata\n---\namdhsa.kernels:\n  - .name:           probe\n    .private_segment_fixed_size: %d\n    .vgpr_count:     96\n"
            "    .vgpr_spill_count: %d\n...\n\t.end_amdgpu_metadata\n")
    s = tmp_path / "spill.s"
    s.write_text(_KERNEL % "\tv_mov_b32_e32 v40, v41" + meta % (160, 40))
    errors, _, _ = isa_lint.lint([str(s)], check_sources=False, verbose=False)
    assert [e[0] for e in errors] == ["E4"] and "40 vector registers" in errors[0][2], errors
    s.write_text(_KERNEL % "\tv_mov_b32_e32 v40, v41" + meta % (3872, 0))          # private arrays without spills (the RANSAC solvers) are fine
    assert not isa_lint.lint([str(s)], check_sources=False, verbose=False)[0]


def test_the_library_hash_does_not_depend_on_the_checkout_path():
    """config.build.lib_sha256 names a COMMIT: build.py maps the sources' absolute paths away and gives every translation unit a fixed compilation-unit
    id (clang derives one from the absolute path otherwise).  Building the tree twice at two paths is a minute of hipcc -- profiles/README.md records
    that it was done; here the two flags are pinned and the shipped library is checked to carry no absolute path of this checkout."""
    assert any(f.startswith("-ffile-prefix-map=") and f.endswith("=.") for f in build.FLAGS)
    import inspect
    assert '"-cuid=kpb_"' in inspect.getsource(build.compile_unit)
    so = build.build()
    blob = open(so, "rb").read()
    assert os.path.join(ROOT, "keypoint_bench_amd", "csrc").encode() not in blob
