"""A small interpreter for the hand-written GCN assembly of k_pass_chain_planes (blance_amd/csrc/k_pass_chain.h,
planes_walk_w2): the inline-asm text is taken from the PREPROCESSED translation unit (hipcc -E, so exactly the strings
the compiler assembles, macro expansion included) and executed instruction by instruction on Python integers.  The SIMT
emulator cannot run inline assembly -- it takes the C++ twin of the loop -- so this is how the scalar loop itself is
checked on a machine without a GPU (tests/test_planes_asm.py).

Only what that loop uses is implemented: SALU moves / logic / shifts / compares / add with their SCC results, the two
cross-lane moves v_readlane_b32 / v_writelane_b32, s_branch / s_cbranch_scc0 / s_cbranch_scc1, GNU-as numeric local
labels ("20f": the next definition of 20 below, "0b": the nearest above) -- and v_min_u32_dpp with the DPP controls of
the lane minima of dev_common.h (wave_min_u32_bcast, row_min_u32), which the emulator likewise replaces by builtins.
Named operands %[x] are looked up in a dict: an int for an SGPR operand (32 or 64 bits wide, as the instruction says), a list of 64 ints for a VGPR."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M32 = (1 << 32) - 1
M64 = (1 << 64) - 1


def preprocessed_asm_templates(source=os.path.join("blance_amd", "csrc", "tu_chain.hip")):
    """Every `asm volatile("..." "..." : ...)` of the preprocessed translation unit as (template text, operand text)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-E", "--cuda-device-only", os.path.join(ROOT, source)],
                         capture_output=True, text=True, check=True).stdout
    res = []
    for m in re.finditer(r"asm volatile\(", out):
        i = m.end()
        parts = []
        while True:
            while out[i] in " \t\n":
                i += 1
            if out[i] != '"':
                break
            j = i + 1
            while out[j] != '"':
                j += 2 if out[j] == "\\" else 1
            parts.append(out[i + 1:j])
            i = j + 1
        # operands: up to the parenthesis that closes the statement
        depth, j = 1, i
        while depth:
            if out[j] == '"':
                j += 1
                while out[j] != '"':
                    j += 2 if out[j] == "\\" else 1
            elif out[j] == "(":
                depth += 1
            elif out[j] == ")":
                depth -= 1
            j += 1
        text = "".join(parts).replace("\\n", "\n").replace("\\t", "\t")
        res.append((text, out[i:j - 1]))
    return res


class Program:
    def __init__(self, text):
        self.ins = []          # (mnemonic, [operands]) or ("label", name)
        for raw in text.split("\n"):
            line = raw.strip()
            if not line:
                continue
            m = re.match(r"^(\d+):$", line)
            if m:
                self.ins.append(("label", m.group(1)))
                continue
            mn, _, rest = line.partition(" ")
            ops, depth, cur = [], 0, ""
            for ch in rest:                                   # commas inside quad_perm:[..] do not separate operands
                depth += ch == "["
                depth -= ch == "]"
                if ch == "," and depth == 0:
                    ops.append(cur.strip())
                    cur = ""
                else:
                    cur += ch
            if cur.strip():
                ops.append(cur.strip())
            self.ins.append((mn, ops))

    def target(self, pc, ref):
        name, direction = ref[:-1], ref[-1]
        rng = range(pc + 1, len(self.ins)) if direction == "f" else range(pc - 1, -1, -1)
        for i in rng:
            if self.ins[i] == ("label", name):
                return i
        raise KeyError("label %s from %d" % (ref, pc))


class Machine:
    """SGPRs by number, m0, scc; named operands in `ops` (ints / 64-entry lists)."""

    def __init__(self, ops):
        self.s = {}
        self.m0 = 0
        self.scc = 0
        self.ops = ops
        self.executed = 0

    # ---- operand access
    def rd(self, o, bits):
        mask = M64 if bits == 64 else M32
        if o == "m0":
            return self.m0 & mask
        m = re.match(r"^s\[(\d+):(\d+)\]$", o)
        if m:
            lo = int(m.group(1))
            return (self.s.get(lo, 0) | (self.s.get(lo + 1, 0) << 32)) & mask
        m = re.match(r"^s(\d+)$", o)
        if m:
            return self.s.get(int(m.group(1)), 0) & mask
        m = re.match(r"^%\[(\w+)\]$", o)
        if m:
            return self.ops[m.group(1)] & mask
        return int(o, 0) & mask

    def wr(self, o, v, bits):
        v &= M64 if bits == 64 else M32
        if o == "m0":
            self.m0 = v & M32
            return
        m = re.match(r"^s\[(\d+):(\d+)\]$", o)
        if m:
            lo = int(m.group(1))
            self.s[lo], self.s[lo + 1] = v & M32, v >> 32
            return
        m = re.match(r"^s(\d+)$", o)
        if m:
            self.s[int(m.group(1))] = v & M32
            return
        m = re.match(r"^%\[(\w+)\]$", o)
        self.ops[m.group(1)] = v

    def vreg(self, o):
        return self.ops[re.match(r"^%\[(\w+)\]$", o).group(1)]

    # ---- execution
    def run(self, prog, limit=10_000_000):
        pc = 0
        while pc < len(prog.ins):
            mn, a = prog.ins[pc]
            pc += 1
            if mn == "label":
                continue
            self.executed += 1
            if self.executed > limit:
                raise RuntimeError("instruction limit")
            if mn in ("s_mov_b32", "s_mov_b64"):
                bits = 64 if mn.endswith("64") else 32
                self.wr(a[0], self.rd(a[1], bits), bits)
            elif mn == "v_readlane_b32":
                self.wr(a[0], self.vreg(a[1])[self.rd(a[2], 32) & 63], 32)
            elif mn == "v_writelane_b32":
                self.vreg(a[0])[self.rd(a[2], 32) & 63] = self.rd(a[1], 32)
            elif mn in ("s_andn2_b64", "s_andn2_b32", "s_or_b64", "s_or_b32", "s_xor_b64", "s_and_b64", "s_and_b32"):
                bits = 64 if mn.endswith("64") else 32
                x, y = self.rd(a[1], bits), self.rd(a[2], bits)
                v = {"andn2": x & ~y, "or": x | y, "xor": x ^ y, "and": x & y}[mn.split("_")[1]] & (M64 if bits == 64 else M32)
                self.wr(a[0], v, bits)
                self.scc = 1 if v else 0
            elif mn == "s_lshl_b64":
                v = (self.rd(a[1], 64) << (self.rd(a[2], 32) & 63)) & M64
                self.wr(a[0], v, 64)
                self.scc = 1 if v else 0
            elif mn == "s_ff1_i32_b64":
                x = self.rd(a[1], 64)
                self.wr(a[0], ((x & -x).bit_length() - 1) if x else M32, 32)
            elif mn == "s_cmp_lg_u64":
                self.scc = 1 if self.rd(a[0], 64) != self.rd(a[1], 64) else 0
            elif mn == "s_cmp_lt_u32":
                self.scc = 1 if self.rd(a[0], 32) < self.rd(a[1], 32) else 0
            elif mn == "s_add_u32":
                v = self.rd(a[1], 32) + self.rd(a[2], 32)
                self.wr(a[0], v, 32)
                self.scc = 1 if v > M32 else 0
            elif mn == "s_branch":
                pc = prog.target(pc - 1, a[0])
            elif mn == "s_cbranch_scc0":
                if not self.scc:
                    pc = prog.target(pc - 1, a[0])
            elif mn == "s_cbranch_scc1":
                if self.scc:
                    pc = prog.target(pc - 1, a[0])
            elif mn == "s_nop":
                pass
            elif mn == "v_min_u32_dpp":
                self.dpp_min(a)
            else:
                raise NotImplementedError(mn)

    def dpp_min(self, a):
        """v_min_u32_dpp vdst, vsrc0, vsrc1 <ctrl> row_mask:m bank_mask:m -- vdst = min(vsrc0 of the lane the control names,
        vsrc1) in the lanes whose row and bank are enabled and whose source lane exists (bound_ctrl is not set: the others
        keep vdst).  Controls: quad_perm:[a,b,c,d], row_half_mirror, row_mirror, row_bcast:15, row_bcast:31."""
        tail = a[2].split()
        dst, src0, src1 = self.vreg_n(a[0]), list(self.vreg_n(a[1])), list(self.vreg_n(tail[0]))
        ctrl = " ".join(tail[1:] + a[3:]) if len(a) > 3 else " ".join(tail[1:])
        ctrl = ctrl.replace(" ,", ",")
        row_mask = int(re.search(r"row_mask:(0x[0-9a-f]+|\d+)", ctrl).group(1), 0)
        bank_mask = int(re.search(r"bank_mask:(0x[0-9a-f]+|\d+)", ctrl).group(1), 0)
        q = re.search(r"quad_perm:\[(\d),(\d),(\d),(\d)\]", ctrl.replace(" ", ""))
        for lane in range(64):
            if not (row_mask >> (lane >> 4)) & 1 or not (bank_mask >> ((lane & 15) >> 2)) & 1:
                continue
            if q:
                src = (lane & ~3) + int(q.group(1 + (lane & 3)))
            elif "row_half_mirror" in ctrl:
                src = (lane & ~7) + 7 - (lane & 7)
            elif "row_mirror" in ctrl:
                src = (lane & ~15) + 15 - (lane & 15)
            elif "row_bcast:15" in ctrl:
                src = (lane & ~15) - 1 if lane >= 16 else None
            elif "row_bcast:31" in ctrl:
                src = 31 if lane >= 32 else None
            else:
                raise NotImplementedError(ctrl)
            if src is None:
                continue
            dst[lane] = min(src0[src], src1[lane])

    def vreg_n(self, o):
        """a VGPR operand: %[name] or a positional %0 of the asm statement"""
        m = re.match(r"^%\[(\w+)\]$", o) or re.match(r"^%(\d+)$", o)
        return self.ops[m.group(1)]
