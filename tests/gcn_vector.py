"""An interpreter for the hand-written gfx950 assembly of the queue kernel's lean walk (blance_amd/csrc/k_queue_walk.h:
queue_walk_k2) -- the subset of the ISA that text uses, one wave64: scalar unit with SCC, vector unit under EXEC (VALU
compares into SGPR pairs / VCC, conditional moves, 32-bit integer ops, the few fp64 ops of the key computation), lane reads
and writes, DPP wave shifts, LDS loads and stores.  The text is taken from the PREPROCESSED translation unit (gcn_scalar.
preprocessed_asm_templates), i.e. exactly the strings the assembler gets, macro expansion included.  The SIMT emulator
cannot run inline assembly -- it takes the C++ twin of the loop -- so this is how the assembly itself is checked on a
machine without a GPU (tests/test_queue_walk_asm.py), next to the device tests that run both.

Registers are fixed in that text (v200.., s38..): the machine has plain register files.  Timing (s_nop, s_waitcnt) is ignored;
every load completes at once."""
import math
import re
import struct

import numpy as np

from gcn_scalar import Program, M32, M64

LANES = np.arange(64, dtype=np.int64)


def _f64(bits):
    return struct.unpack("<d", struct.pack("<Q", bits & M64))[0]


def _bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


class Wave:
    def __init__(self, lds_bytes=160 * 1024):
        self.s = [0] * 110
        self.v = np.zeros((256, 64), dtype=np.uint32)
        self.vcc = 0
        self.exec = M64
        self.m0 = 0
        self.scc = 0
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)
        self.executed = 0
        self.trace = None

    # ---- scalar operands
    def rs(self, o, bits=32):
        mask = M64 if bits == 64 else M32
        if o == "vcc":
            return self.vcc & mask
        if o == "exec":
            return self.exec & mask
        if o == "m0":
            return self.m0 & mask
        m = re.match(r"^s\[(\d+):(\d+)\]$", o)
        if m:
            lo = int(m.group(1))
            return (self.s[lo] | (self.s[lo + 1] << 32)) & mask
        m = re.match(r"^s(\d+)$", o)
        if m:
            if bits == 64:
                lo = int(m.group(1))
                return (self.s[lo] | (self.s[lo + 1] << 32)) & mask
            return self.s[int(m.group(1))] & mask
        v = int(o, 0)
        return v & mask                                    # (inline constants and literals are sign extended to the width)

    def ws(self, o, v, bits=32):
        v &= M64 if bits == 64 else M32
        if o == "vcc":
            self.vcc = v
            return
        if o == "exec":
            self.exec = v
            return
        if o == "m0":
            self.m0 = v & M32
            return
        m = re.match(r"^s\[(\d+):(\d+)\]$", o) or re.match(r"^s(\d+)$", o)
        lo = int(m.group(1))
        self.s[lo] = v & M32
        if bits == 64:
            self.s[lo + 1] = v >> 32

    # ---- vector operands: 32-bit values per lane as int64 arrays (unsigned), 64-bit as Python-int object arrays
    def rv(self, o):
        m = re.match(r"^v(\d+)$", o)
        if m:
            return self.v[int(m.group(1))].astype(np.int64)
        return np.full(64, self.rs(o, 32), dtype=np.int64)

    def rv64(self, o):
        m = re.match(r"^v\[(\d+):(\d+)\]$", o)
        if m:
            lo = int(m.group(1))
            return [int(self.v[lo][i]) | (int(self.v[lo + 1][i]) << 32) for i in range(64)]
        x = self.rs(o, 64)
        return [x] * 64

    def active(self):
        return np.array([(self.exec >> i) & 1 for i in range(64)], dtype=bool)

    def wv(self, o, vals, act=None):
        r = int(re.match(r"^v(\d+)$", o).group(1))
        act = self.active() if act is None else act
        vals = np.asarray(vals, dtype=np.int64) & M32
        self.v[r][act] = vals[act].astype(np.uint32)

    def wv64(self, o, vals):
        lo = int(re.match(r"^v\[(\d+):(\d+)\]$", o).group(1))
        act = self.active()
        for i in range(64):
            if act[i]:
                self.v[lo][i] = vals[i] & M32
                self.v[lo + 1][i] = (vals[i] >> 32) & M32

    def wmask(self, o, bools):
        act = self.active()
        m = 0
        for i in range(64):
            if act[i] and bools[i]:
                m |= 1 << i
        self.ws(o, m, 64)

    @staticmethod
    def s32(x):
        x = np.asarray(x, dtype=np.int64) & M32
        return np.where(x >= (1 << 31), x - (1 << 32), x)

    # ---- LDS
    def lds_rd(self, addr, nbytes):
        return int.from_bytes(bytes(self.lds[addr:addr + nbytes]), "little")

    def lds_wr(self, addr, val, nbytes):
        self.lds[addr:addr + nbytes] = np.frombuffer(int(val).to_bytes(nbytes, "little"), dtype=np.uint8)

    # ---- execution
    def run(self, prog, limit=5_000_000):
        pc = 0
        ins = prog.ins
        while pc < len(ins):
            mn, a = ins[pc]
            pc += 1
            if mn == "label":
                continue
            self.executed += 1
            if self.executed > limit:
                raise RuntimeError("instruction limit")
            if self.trace is not None:
                self.trace.append((pc - 1, mn, a))
            # ---------------- control
            if mn == "s_branch":
                pc = prog.target(pc - 1, a[0])
            elif mn == "s_cbranch_scc0":
                if not self.scc:
                    pc = prog.target(pc - 1, a[0])
            elif mn == "s_cbranch_scc1":
                if self.scc:
                    pc = prog.target(pc - 1, a[0])
            elif mn in ("s_nop", "s_waitcnt"):
                pass
            # ---------------- scalar
            elif mn in ("s_mov_b32", "s_mov_b64"):
                bits = 64 if mn.endswith("64") else 32
                self.ws(a[0], self.rs(a[1], bits), bits)
            elif mn in ("s_and_b32", "s_and_b64", "s_or_b64", "s_andn2_b64"):
                bits = 64 if mn.endswith("64") else 32
                x, y = self.rs(a[1], bits), self.rs(a[2], bits)
                v = {"and": x & y, "or": x | y, "andn2": x & ~y}[mn.split("_")[1]] & (M64 if bits == 64 else M32)
                self.ws(a[0], v, bits)
                self.scc = 1 if v else 0
            elif mn == "s_and_saveexec_b64":
                old = self.exec
                self.exec = old & self.rs(a[1], 64)
                self.ws(a[0], old, 64)
                self.scc = 1 if self.exec else 0
            elif mn in ("s_lshl_b32", "s_lshl_b64", "s_lshr_b32"):
                bits = 64 if mn.endswith("64") else 32
                x, n = self.rs(a[1], bits), self.rs(a[2], 32) & (bits - 1)
                v = ((x << n) if "lshl" in mn else (x >> n)) & (M64 if bits == 64 else M32)
                self.ws(a[0], v, bits)
                self.scc = 1 if v else 0
            elif mn == "s_ashr_i32":
                x = self.rs(a[1], 32)
                x = x - (1 << 32) if x >> 31 else x
                v = (x >> (self.rs(a[2], 32) & 31)) & M32
                self.ws(a[0], v)
                self.scc = 1 if v else 0
            elif mn == "s_sext_i32_i16":
                x = self.rs(a[1], 32) & 0xffff
                self.ws(a[0], (x - 0x10000 if x >> 15 else x) & M32)
            elif mn == "s_bfe_u32":
                x, c = self.rs(a[1], 32), self.rs(a[2], 32)
                off, wid = c & 31, (c >> 16) & 0x7f
                v = (x >> off) & ((1 << wid) - 1)
                self.ws(a[0], v)
                self.scc = 1 if v else 0
            elif mn == "s_mul_i32":
                self.ws(a[0], (self.rs(a[1], 32) * self.rs(a[2], 32)) & M32)
            elif mn == "s_add_u32":
                v = self.rs(a[1], 32) + self.rs(a[2], 32)
                self.ws(a[0], v & M32)
                self.scc = 1 if v > M32 else 0
            elif mn == "s_addc_u32":
                v = self.rs(a[1], 32) + self.rs(a[2], 32) + self.scc
                self.ws(a[0], v & M32)
                self.scc = 1 if v > M32 else 0
            elif mn == "s_sub_u32":
                x, y = self.rs(a[1], 32), self.rs(a[2], 32)
                self.ws(a[0], (x - y) & M32)
                self.scc = 1 if y > x else 0
            elif mn == "s_subb_u32":
                x, y = self.rs(a[1], 32), self.rs(a[2], 32) + self.scc
                self.ws(a[0], (x - y) & M32)
                self.scc = 1 if y > x else 0
            elif mn == "s_ff1_i32_b64":
                x = self.rs(a[1], 64)
                self.ws(a[0], ((x & -x).bit_length() - 1) if x else M32)
            elif mn == "s_bcnt1_i32_b64":
                v = bin(self.rs(a[1], 64)).count("1")
                self.ws(a[0], v)
                self.scc = 1 if v else 0
            elif mn in ("s_bitcmp1_b32", "s_bitcmp1_b64"):
                bits = 64 if mn.endswith("64") else 32
                self.scc = (self.rs(a[0], bits) >> (self.rs(a[1], 32) & (bits - 1))) & 1
            elif mn == "s_bitset1_b64":
                self.ws(a[0], self.rs(a[0], 64) | (1 << (self.rs(a[1], 32) & 63)), 64)
            elif mn in ("s_cselect_b32", "s_cselect_b64"):
                bits = 64 if mn.endswith("64") else 32
                self.ws(a[0], self.rs(a[1], bits) if self.scc else self.rs(a[2], bits), bits)
            elif mn.startswith("s_cmp_"):
                _, _, rel, ty = mn.split("_")
                bits = 64 if ty.endswith("64") else 32
                x, y = self.rs(a[0], bits), self.rs(a[1], bits)
                if ty.startswith("i"):
                    x = x - (1 << bits) if x >> (bits - 1) else x
                    y = y - (1 << bits) if y >> (bits - 1) else y
                self.scc = 1 if {"eq": x == y, "lg": x != y, "lt": x < y, "le": x <= y, "gt": x > y, "ge": x >= y}[rel] else 0
            # ---------------- lane access (ignores EXEC)
            elif mn == "v_readlane_b32":
                self.ws(a[0], int(self.v[int(a[1][1:])][self.rs(a[2], 32) & 63]))
            elif mn == "v_readfirstlane_b32":
                first = ((self.exec & -self.exec).bit_length() - 1) if self.exec else 0
                self.ws(a[0], int(self.v[int(a[1][1:])][first]))
            elif mn == "v_writelane_b32":
                self.v[int(a[0][1:])][self.rs(a[2], 32) & 63] = self.rs(a[1], 32)
            # ---------------- vector compares
            elif mn.startswith("v_cmp_"):
                parts = mn.split("_")              # v cmp rel type enc
                rel, ty = parts[2], parts[3]
                if mn.endswith("_e64"):
                    dst, x, y = a[0], a[1], a[2]
                else:
                    dst, x, y = "vcc", a[1], a[2]
                if ty == "f64":
                    xs = [_f64(v) for v in self.rv64(x)]
                    ys = [_f64(v) for v in self.rv64(y)]
                elif ty == "u64":
                    xs, ys = self.rv64(x), self.rv64(y)
                elif ty == "i32":
                    xs, ys = self.s32(self.rv(x)), self.s32(self.rv(y))
                else:
                    xs, ys = self.rv(x), self.rv(y)
                f = {"eq": lambda p, q: p == q, "ne": lambda p, q: p != q, "lt": lambda p, q: p < q, "le": lambda p, q: p <= q,
                     "gt": lambda p, q: p > q, "ge": lambda p, q: p >= q}[rel]
                self.wmask(dst, [bool(f(xs[i], ys[i])) for i in range(64)])
            # ---------------- vector ALU
            elif mn == "v_mov_b32_e32":
                self.wv(a[0], self.rv(a[1]))
            elif mn == "v_bfrev_b32_e32":
                x = self.rv(a[1])
                self.wv(a[0], [int("{:032b}".format(int(v) & M32)[::-1], 2) for v in x])
            elif mn in ("v_cndmask_b32_e32", "v_cndmask_b32_e64"):
                sel = self.vcc if mn.endswith("e32") else self.rs(a[3], 64)
                pick = np.array([(sel >> i) & 1 for i in range(64)], dtype=bool)
                self.wv(a[0], np.where(pick, self.rv(a[2]), self.rv(a[1])))
            elif mn == "v_add_u32_e32":
                self.wv(a[0], self.rv(a[1]) + self.rv(a[2]))
            elif mn == "v_sub_u32_e32":
                self.wv(a[0], self.rv(a[1]) - self.rv(a[2]))
            elif mn == "v_and_b32_e32":
                self.wv(a[0], self.rv(a[1]) & self.rv(a[2]))
            elif mn == "v_or_b32_e32":
                self.wv(a[0], self.rv(a[1]) | self.rv(a[2]))
            elif mn == "v_xor_b32_e32":
                self.wv(a[0], self.rv(a[1]) ^ self.rv(a[2]))
            elif mn == "v_lshlrev_b32_e32":
                self.wv(a[0], self.rv(a[2]) << (self.rv(a[1]) & 31))
            elif mn == "v_lshrrev_b32_e32":
                self.wv(a[0], (self.rv(a[2]) & M32) >> (self.rv(a[1]) & 31))
            elif mn == "v_ashrrev_i32_e32":
                self.wv(a[0], self.s32(self.rv(a[2])) >> (self.rv(a[1]) & 31))
            elif mn == "v_lshl_add_u32":
                self.wv(a[0], (self.rv(a[1]) << (self.rv(a[2]) & 31)) + self.rv(a[3]))
            elif mn == "v_cvt_f64_i32_e32":
                self.wv64(a[0], [_bits(float(int(v))) for v in self.s32(self.rv(a[1]))])
            elif mn == "v_add_f64":
                xs, ys = self.rv64(a[1]), self.rv64(a[2])
                self.wv64(a[0], [_bits(_f64(xs[i]) + _f64(ys[i])) for i in range(64)])
            elif mn == "v_ldexp_f64":
                xs, es = self.rv64(a[1]), self.s32(self.rv(a[2]))
                self.wv64(a[0], [_bits(math.ldexp(_f64(xs[i]), int(es[i]))) for i in range(64)])
            elif mn == "v_mov_b32_dpp":
                self.dpp_mov(a)
            # ---------------- LDS
            elif mn.startswith("ds_read_"):
                nbytes = {"b32": 4, "b64": 8, "u16": 2, "u8": 1}[mn.split("_")[2]]
                addr = self.rv(a[1])
                act = self.active()
                vals = [self.lds_rd(int(addr[i]), nbytes) if act[i] else 0 for i in range(64)]
                if nbytes == 8:
                    self.wv64(a[0], vals)
                else:
                    self.wv(a[0], vals)
            elif mn.startswith("ds_write_"):
                nbytes = {"b32": 4, "b64": 8, "b16": 2}[mn.split("_")[2]]
                addr = self.rv(a[0])
                act = self.active()
                vals = self.rv64(a[1]) if nbytes == 8 else self.rv(a[1])
                for i in range(64):
                    if act[i]:
                        self.lds_wr(int(addr[i]), int(vals[i]) & ((1 << (8 * nbytes)) - 1), nbytes)
            else:
                raise NotImplementedError(mn + " " + ", ".join(a))

    def dpp_mov(self, a):
        """v_mov_b32_dpp vdst, vsrc wave_shl:1 | wave_shr:1 row_mask:0xf bank_mask:0xf (bound_ctrl off): lane i takes lane i + 1 /
        i - 1 of vsrc; the lane without a source keeps vdst; EXEC applies to the destination."""
        tail = a[1].split()
        src = self.v[int(tail[0][1:])].copy()
        ctrl = " ".join(tail[1:])
        assert "row_mask:0xf" in ctrl and "bank_mask:0xf" in ctrl, ctrl
        dst = self.v[int(a[0][1:])]
        act = self.active()
        if "wave_shl:1" in ctrl:
            for i in range(63):
                if act[i]:
                    dst[i] = src[i + 1]
        elif "wave_shr:1" in ctrl:
            for i in range(1, 64):
                if act[i]:
                    dst[i] = src[i - 1]
        else:
            raise NotImplementedError(ctrl)


def queue_walk_program():
    """The text of queue_walk_k2 as the assembler gets it (the longest asm statement of tu_queue.hip)."""
    import os
    from gcn_scalar import preprocessed_asm_templates
    tpl = preprocessed_asm_templates(os.path.join("blance_amd", "csrc", "tu_queue.hip"))
    text, _ = max(tpl, key=lambda t: len(t[0]))
    assert "v[200:201]" in text and "wave_shl:1" in text
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return Program(text)
