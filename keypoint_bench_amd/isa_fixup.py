"""Assembly pass between hipcc's code generator and the assembler: rewrites the one packed-fp32 operand selection that MI355X
(gfx950) executes wrongly beside matrix instructions, and scans for it.

The defect (found in round 5 from a build of the ALIKE head that differed run to run in EVERY run; scripts/ubench/pk_opsel.hip and
gen_pk_opsel.py isolate it, profiles/r05_pk_opsel_*.txt are the records):

    v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32   with op_sel[0] = 0 and op_sel[1] = 1
        -- the LOW result lane takes the low dword of src0 and the HIGH dword of src1 --

    returns a wrong LOW result in lanes 48..63 of the wave (src1's high dword reads as 0.0: a product of 0, a sum of src0 alone, an fma
    of src2 alone) in about one execution in a thousand while an f16 MFMA (v_mfma_f32_32x32x16_f16, v_mfma_f32_16x16x32_f16) of ANY wave
    of the SIMD is executing.  Every other op_sel / op_sel_hi combination of the three instructions, v_pk_mov_b32, and all of them
    without MFMAs in flight (or beside v_mfma_f32_32x32x2_f32) gave 0 wrong results in 2.6e7 lane samples each.  hipcc (ROCm 7.2) emits
    the encoding when its SLP vectoriser folds a shuffle into a packed instruction (a splat of element 1, `a.x + a.y` as one packed
    add); it knows no hazard there.

The three instructions commute in src0 / src1, so the pass SWAPS the two operands together with their op_sel, op_sel_hi, neg_lo and
neg_hi bits: op_sel [0, 1] becomes [1, 0] -- the same arithmetic on an encoding that measured clean -- at no cost.
`python -m keypoint_bench_amd.isa_fixup file.s` prints what it would change; build.py applies it to every translation unit, and
scripts/isa_lint.py checks the linked library for leftovers.
"""
import re
import sys

_PK = re.compile(r"^(\s*)(v_pk_(?:mul|add|fma)_f32)(?:_e64)?\s+(.*?)\s*$")
_MOD = re.compile(r"\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,]+)\]")
_DEFAULT = {"op_sel": 0, "op_sel_hi": 1, "neg_lo": 0, "neg_hi": 0}


def _split_operands(text):
    out, cur, depth = [], "", 0
    for ch in text:
        depth += ch == "["
        depth -= ch == "]"
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def parse_packed(line):
    """(indent, mnemonic, operands [dst, src0, src1(, src2)], {modifier: [bits]}, trailing text) of a packed-fp32 line, else None."""
    code, sep, comment = line.partition(";")
    m = _PK.match(code)
    if not m:
        return None
    indent, mnem, rest = m.groups()
    first = _MOD.search(rest)
    ops_text, mod_text = (rest[:first.start()], rest[first.start():]) if first else (rest, "")
    ops = _split_operands(ops_text)
    nsrc = len(ops) - 1
    if nsrc not in (2, 3):
        return None
    mods = {}
    for name, bits in _MOD.findall(mod_text):
        v = [int(b) for b in bits.split(",")]
        mods[name] = v + [_DEFAULT[name]] * (nsrc - len(v))
    other = _MOD.sub("", mod_text).strip()          # e.g. clamp
    for name, d in _DEFAULT.items():
        mods.setdefault(name, [d] * nsrc)
    return indent, mnem, ops, mods, other, (sep + comment if sep else "")


_PK_LOOSE = re.compile(r"^\s*v_pk_(?:mul|add|fma)_f32\b")


def is_unparsed(line):
    """True for a line that IS one of the three instructions but that parse_packed cannot take apart (another assembler's modifier spelling,
    an operand count other than 2 or 3): neither the rewrite nor the lint may pass such a line silently -- the library's determinism must not
    hang on the disassembler's text syntax staying what it is."""
    code = line.partition(";")[0]
    return bool(_PK_LOOSE.match(code)) and parse_packed(line) is None


def is_affected(line):
    p = parse_packed(line)
    return bool(p) and p[3]["op_sel"][0] == 0 and p[3]["op_sel"][1] == 1


def fix_line(line):
    """The line with src0 / src1 (and their modifier bits) swapped if it carries the affected operand selection, else the line unchanged."""
    p = parse_packed(line)
    if not p:
        return line
    indent, mnem, ops, mods, other, comment = p
    if not (mods["op_sel"][0] == 0 and mods["op_sel"][1] == 1):
        return line
    ops[1], ops[2] = ops[2], ops[1]
    text = indent + mnem + " " + ", ".join(ops)
    for name in ("op_sel", "op_sel_hi", "neg_lo", "neg_hi"):
        v = mods[name]
        v[0], v[1] = v[1], v[0]
        if any(b != _DEFAULT[name] for b in v):
            text += " %s:[%s]" % (name, ",".join(str(b) for b in v))
    if other:
        text += " " + other
    return text + (" " + comment if comment else "")


def fix_text(asm_text):
    """(fixed text, number of instructions rewritten)."""
    out, n = [], 0
    for line in asm_text.split("\n"):
        if is_unparsed(line):
            raise ValueError("isa_fixup: cannot parse the packed-fp32 instruction %r -- refusing to pass it through unchecked" % line.strip())
        new = fix_line(line)
        n += new is not line and new != line
        out.append(new)
    return "\n".join(out), n


if __name__ == "__main__":
    for path in sys.argv[1:]:
        src = open(path).read()
        _, count = fix_text(src)
        print("%s: %d packed-fp32 instructions with op_sel [0, 1, .]" % (path, count))
        for ln in src.split("\n"):
            if is_affected(ln):
                print("   ", ln.strip(), " ->", fix_line(ln).strip())
