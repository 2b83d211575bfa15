"""A small gfx950 assembly DSL with a functional emulator.

The hand-scheduled GEMM k-loop (tools/kgen/gemm_z192_gen.py -> sam_road_amd/csrc/gemm_z192_body.inc) is written against this
DSL: every instruction is appended to a Prog both as TEXT (what hipcc's assembler gets, verbatim, inside one asm volatile
block) and as an emulation closure.  The emulator runs a whole workgroup (4 waves x 64 lanes, shared LDS, numpy buffers as
global memory) and is deliberately ADVERSARIAL about the asynchronous parts, because those are what a GPU run cannot localise:

  * LDS-DMA (buffer_load ... lds) writes land either at issue ("eager") or only when a counted s_waitcnt vmcnt retires them
    ("lazy") — a ds_read that is not ordered behind the covering wait + barrier sees stale bytes in one of the two modes;
  * ds_read results are delivered either at issue or at the retiring s_waitcnt lgkmcnt (destination poisoned until then, LDS
    sampled at the LATE point in lazy mode: a DMA that overwrites a slab before its readers are done shows up);
  * between barriers the waves run in a configurable order (0123 / 3210 / round-robin).

A schedule is accepted when every mode combination produces the reference result.  A static pass (check_hazards) counts the
software wait states of the gfx940-family hazards this code can hit (VALU write -> MFMA operand, MFMA result -> VALU /
v_accvgpr_read / store, VALU write -> v_permlane32_swap, s_mov m0 -> LDS-DMA, VALU-written SGPR -> VMEM, store data overwrite).

Only the instructions the generators use are implemented.  Test infrastructure / build tooling: not part of the product path.
"""
import numpy as np

POISON = np.uint32(0x7FC0DEAD)


# ------------------------------------------------------------------------------------------------ registers
class Reg:
    __slots__ = ("file", "idx", "n")

    def __init__(self, file, idx, n=1):
        self.file, self.idx, self.n = file, idx, n
        if n > 1:       # gfx90a+: VGPR / AGPR tuples are 64-bit aligned; SGPR pairs even, quads and wider 4-aligned
            assert idx % (2 if file != "s" or n == 2 else 4) == 0, f"misaligned register tuple {file}[{idx}:{idx + n - 1}]"

    def __str__(self):
        p = {"v": "v", "a": "a", "s": "s"}[self.file]
        return f"{p}{self.idx}" if self.n == 1 else f"{p}[{self.idx}:{self.idx + self.n - 1}]"

    def __getitem__(self, i):           # sub-register
        assert 0 <= i < self.n
        return Reg(self.file, self.idx + i, 1)

    def sub(self, i, n):
        assert 0 <= i and i + n <= self.n
        return Reg(self.file, self.idx + i, n)

    def regs(self):
        return [(self.file, self.idx + i) for i in range(self.n)]


def V(i, n=1):
    assert 0 <= i and i + n <= 256
    return Reg("v", i, n)


def A(i, n=1):
    assert 0 <= i and i + n <= 256
    return Reg("a", i, n)


def S(i, n=1):
    assert 0 <= i and i + n <= 102
    return Reg("s", i, n)


class Special:
    def __init__(self, name):
        self.name = name

    def __str__(self):
        return self.name


M0, VCC, EXEC_LO, EXEC_HI = Special("m0"), Special("vcc"), Special("exec_lo"), Special("exec_hi")


class Lit:
    """32-bit literal / inline constant (int or float)."""

    def __init__(self, v):
        self.v = v

    def __str__(self):
        if isinstance(self.v, float):
            return "0x%08x" % int(np.float32(self.v).view(np.uint32))
        return str(self.v) if -16 <= self.v <= 64 else "0x%08x" % (self.v & 0xFFFFFFFF)

    def u32(self):
        if isinstance(self.v, float):
            return np.uint32(np.float32(self.v).view(np.uint32))
        return np.uint32(self.v & 0xFFFFFFFF)


class Mod:
    """Source operand with VOP3 neg / abs modifiers."""

    def __init__(self, r, neg=False, ab=False):
        self.r, self.neg, self.ab = r, neg, ab

    def __str__(self):
        s = f"abs({self.r})" if self.ab else str(self.r)       # not |v|: '|' is a dialect separator inside an inline-asm string
        return "-" + s if self.neg else s


def neg(r):
    return Mod(r, neg=True)


def nabs(r):
    return Mod(r, neg=True, ab=True)


def vabs(r):
    return Mod(r, ab=True)


# ------------------------------------------------------------------------------------------------ program
class Ins:
    __slots__ = ("text", "emu", "kind", "reads", "writes", "extra", "label")

    def __init__(self, text, emu, kind, reads=(), writes=(), extra=None):
        self.text, self.emu, self.kind, self.reads, self.writes, self.extra = text, emu, kind, list(reads), list(writes), extra
        self.label = None


def _regs_of(ops):
    out = []
    for o in ops:
        if isinstance(o, Mod):
            o = o.r
        if isinstance(o, Reg):
            out += o.regs()
        elif isinstance(o, Special):
            out.append(("x", o.name))
    return out


class Prog:
    def __init__(self):
        self.ins = []
        self.labels = {}
        self._uid = 0

    # -- plumbing
    def add(self, text, emu, kind, reads=(), writes=(), extra=None):
        self.ins.append(Ins(text, emu, kind, _regs_of(reads), _regs_of(writes), extra))
        return self.ins[-1]

    def label(self, name):
        self.labels[name] = len(self.ins)
        self.add(f"{name}:", lambda st: None, "label")
        self.ins[-1].label = name

    def newlabel(self, stem):
        self._uid += 1
        return f".L{stem}_{self._uid}_%="

    def comment(self, txt):
        self.add(f"; {txt}", lambda st: None, "comment")

    def text(self):
        return "\n".join(("    " if i.kind != "label" else "") + i.text for i in self.ins)

    def n_real(self):
        return sum(1 for i in self.ins if i.kind not in ("label", "comment"))

    # ---------------------------------------------------------------------------------------- SALU
    def _sval(self, st, o):
        if isinstance(o, Reg):
            assert o.file == "s" and o.n == 1
            return np.uint32(st.s[o.idx])
        if isinstance(o, Lit):
            return o.u32()
        if isinstance(o, Special):
            return np.uint32(st.special[o.name])
        if isinstance(o, int):
            return np.uint32(o & 0xFFFFFFFF)
        raise TypeError(o)

    def _swrite(self, st, d, val):
        val = int(val) & 0xFFFFFFFF
        if isinstance(d, Special):
            st.special[d.name] = val
        else:
            st.s[d.idx] = val

    def _lit(self, o):
        return Lit(o) if isinstance(o, (int, float)) else o

    def raw(self, text):
        """Text only (operand plumbing of the enclosing asm statement); the emulator's caller sets those registers up."""
        self.add(text, lambda st: None, "comment")

    def s_mov_b32(self, d, a):
        a = self._lit(a)

        def emu(st):
            self._swrite(st, d, self._sval(st, a))
            if isinstance(a, Reg) and isinstance(d, Reg) and a.idx in st.sobj:
                st.sobj[d.idx] = st.sobj[a.idx]
        self.add(f"s_mov_b32 {d}, {a}", emu, "salu", [a], [d])

    def _s_bin(self, name, fn, d, a, b, scc=None):
        a, b = self._lit(a), self._lit(b)

        def emu(st):
            x, y = int(self._sval(st, a)), int(self._sval(st, b))
            r = fn(x, y)
            if scc is not None:
                st.scc = int(scc(x, y, r))
            self._swrite(st, d, r)
        self.add(f"{name} {d}, {a}, {b}", emu, "salu", [a, b], [d])

    def s_add_u32(self, d, a, b):
        self._s_bin("s_add_u32", lambda x, y: x + y, d, a, b, scc=lambda x, y, r: r > 0xFFFFFFFF)

    def s_sub_u32(self, d, a, b):
        self._s_bin("s_sub_u32", lambda x, y: x - y, d, a, b, scc=lambda x, y, r: y > x)

    def s_min_u32(self, d, a, b):
        self._s_bin("s_min_u32", lambda x, y: min(int(x), int(y)), d, a, b)

    def s_addc_u32(self, d, a, b):
        """d = a + b + scc (carry in); scc = carry out"""
        a, b = self._lit(a), self._lit(b)

        def emu(st):
            v = int(self._sval(st, a)) + int(self._sval(st, b)) + int(st.scc)
            self._swrite(st, d, v)
            st.scc = 1 if v > 0xFFFFFFFF else 0
        self.add(f"s_addc_u32 {d}, {a}, {b}", emu, "salu", [a, b], [d])

    def s_mul_i32(self, d, a, b):
        self._s_bin("s_mul_i32", lambda x, y: x * y, d, a, b)

    def s_lshl_b32(self, d, a, b):
        self._s_bin("s_lshl_b32", lambda x, y: x << (y & 31), d, a, b, scc=lambda x, y, r: (r & 0xFFFFFFFF) != 0)

    def s_lshr_b32(self, d, a, b):
        self._s_bin("s_lshr_b32", lambda x, y: x >> (y & 31), d, a, b, scc=lambda x, y, r: (r & 0xFFFFFFFF) != 0)

    def s_and_b32(self, d, a, b):
        self._s_bin("s_and_b32", lambda x, y: x & y, d, a, b, scc=lambda x, y, r: r != 0)

    def s_or_b32(self, d, a, b):
        self._s_bin("s_or_b32", lambda x, y: x | y, d, a, b, scc=lambda x, y, r: r != 0)

    def _s_cmp(self, name, fn, a, b):
        a, b = self._lit(a), self._lit(b)

        def emu(st):
            st.scc = int(fn(int(self._sval(st, a)), int(self._sval(st, b))))
        self.add(f"{name} {a}, {b}", emu, "salu", [a, b], [])

    def s_cmp_eq_u32(self, a, b):
        self._s_cmp("s_cmp_eq_u32", lambda x, y: x == y, a, b)

    def s_cmp_lg_u32(self, a, b):
        self._s_cmp("s_cmp_lg_u32", lambda x, y: x != y, a, b)

    def s_cmp_lt_u32(self, a, b):
        self._s_cmp("s_cmp_lt_u32", lambda x, y: x < y, a, b)

    def s_cmp_ge_u32(self, a, b):
        self._s_cmp("s_cmp_ge_u32", lambda x, y: x >= y, a, b)

    def s_cmp_gt_u32(self, a, b):
        self._s_cmp("s_cmp_gt_u32", lambda x, y: x > y, a, b)

    def s_cselect_b32(self, d, a, b):
        a, b = self._lit(a), self._lit(b)
        self.add(f"s_cselect_b32 {d}, {a}, {b}", lambda st: self._swrite(st, d, self._sval(st, a) if st.scc else self._sval(st, b)), "salu", [a, b], [d])

    def s_branch(self, label):
        self.add(f"s_branch {label}", lambda st: label, "branch")

    def s_cbranch_scc1(self, label):
        self.add(f"s_cbranch_scc1 {label}", lambda st: label if st.scc else None, "branch")

    def s_cbranch_scc0(self, label):
        self.add(f"s_cbranch_scc0 {label}", lambda st: None if st.scc else label, "branch")

    def s_nop(self, n):
        assert 0 <= n <= 15
        self.add(f"s_nop {n}", lambda st: None, "nop", extra=n + 1)

    def s_setprio(self, n):
        self.add(f"s_setprio {n}", lambda st: None, "salu")

    def s_sleep(self, n):
        self.add(f"s_sleep {n}", lambda st: None, "salu")

    def s_barrier(self):
        self.add("s_barrier", lambda st: "BARRIER", "barrier")

    def s_waitcnt(self, vmcnt=None, lgkmcnt=None):
        parts = []
        if vmcnt is not None:
            assert 0 <= vmcnt <= 63
            parts.append(f"vmcnt({vmcnt})")
        if lgkmcnt is not None:
            assert 0 <= lgkmcnt <= 15
            parts.append(f"lgkmcnt({lgkmcnt})")

        def emu(st):
            if vmcnt is not None:
                st.wait_vm(vmcnt)
            if lgkmcnt is not None:
                st.wait_lgkm(lgkmcnt)
        self.add("s_waitcnt " + " ".join(parts), emu, "wait", extra=(vmcnt, lgkmcnt))

    def s_load_dwordx(self, n, d, base, offset):
        """s_load_dwordx{n} d, base(s pair: a tagged host buffer), byte offset (imm or SGPR)."""
        assert d.n == n and base.n == 2
        off_op = offset

        def emu(st):
            buf = st.sobj[base.idx]
            offset = int(st.s[off_op.idx]) if isinstance(off_op, Reg) else off_op
            assert offset % 4 == 0 and offset + 4 * n <= buf.size, "s_load out of range"
            vals = np.frombuffer(buf, dtype=np.uint32, count=n, offset=offset).copy()
            objs = st.mem_objs.get((id(buf), offset))

            def deliver(vals=vals, objs=objs):
                for i in range(n):
                    st.s[d.idx + i] = int(vals[i])
                if objs:
                    for rel, o in objs.items():       # pointer fields: tag the SGPR pair with the buffer object
                        if rel < n:
                            st.sobj[d.idx + rel] = o
            st.issue_lgkm(deliver, smem=True)
        name = {1: "s_load_dword", 2: "s_load_dwordx2", 4: "s_load_dwordx4", 8: "s_load_dwordx8"}[n]
        otxt = str(off_op) if isinstance(off_op, Reg) else f"0x{off_op:x}"
        self.add(f"{name} {d}, {base}, {otxt}", emu, "smem", [base] + ([off_op] if isinstance(off_op, Reg) else []), [d])

    # ---------------------------------------------------------------------------------------- VALU (integer)
    def _vsrc(self, st, o):
        if isinstance(o, Mod):
            x = self._vsrc(st, o.r).view(np.float32)
            if o.ab:
                x = np.abs(x)
            if o.neg:
                x = -x
            return x.view(np.uint32)
        if isinstance(o, Reg):
            if o.file == "v":
                return st.v[o.idx].copy()
            if o.file == "a":
                return st.a[o.idx].copy()
            return np.full(64, st.s[o.idx], dtype=np.uint32)
        if isinstance(o, Lit):
            return np.full(64, o.u32(), dtype=np.uint32)
        if isinstance(o, (int, float)):
            return np.full(64, Lit(o).u32(), dtype=np.uint32)
        raise TypeError(o)

    def _vwrite(self, st, d, val):
        val = np.asarray(val).astype(np.uint32) if np.asarray(val).dtype != np.uint32 else np.asarray(val)
        m = st.exec_mask()
        tgt = st.v if d.file == "v" else st.a
        tgt[d.idx][m] = val[m]

    def _v_op(self, name, fn, d, srcs, kind="valu", suffix=""):
        srcs = [self._lit(s) for s in srcs]

        def emu(st):
            self._vwrite(st, d, fn(*[self._vsrc(st, s) for s in srcs]))
        self.add(f"{name} {d}, " + ", ".join(str(s) for s in srcs) + suffix, emu, kind, srcs, [d])

    def v_mov_b32(self, d, a):
        self._v_op("v_mov_b32", lambda x: x, d, [a])

    def v_add_u32(self, d, a, b):
        self._v_op("v_add_u32", lambda x, y: x + y, d, [a, b])

    def v_sub_u32(self, d, a, b):
        self._v_op("v_sub_u32", lambda x, y: x - y, d, [a, b])

    def v_mul_lo_u32(self, d, a, b):
        self._v_op("v_mul_lo_u32", lambda x, y: (x.astype(np.uint64) * y.astype(np.uint64)).astype(np.uint32), d, [a, b])

    def v_lshlrev_b32(self, d, sh, a):
        self._v_op("v_lshlrev_b32", lambda s_, x: x << (s_ & 31), d, [sh, a])

    def v_lshrrev_b32(self, d, sh, a):
        self._v_op("v_lshrrev_b32", lambda s_, x: x >> (s_ & 31), d, [sh, a])

    def v_and_b32(self, d, a, b):
        self._v_op("v_and_b32", lambda x, y: x & y, d, [a, b])

    def v_or_b32(self, d, a, b):
        self._v_op("v_or_b32", lambda x, y: x | y, d, [a, b])

    def v_xor_b32(self, d, a, b):
        self._v_op("v_xor_b32", lambda x, y: x ^ y, d, [a, b])

    def v_lshl_add_u32(self, d, a, sh, c):
        self._v_op("v_lshl_add_u32", lambda x, s_, z: (x << (s_ & 31)) + z, d, [a, sh, c])

    def v_lshl_or_b32(self, d, a, sh, c):
        self._v_op("v_lshl_or_b32", lambda x, s_, z: (x << (s_ & 31)) | z, d, [a, sh, c])

    def v_and_or_b32(self, d, a, b, c):
        self._v_op("v_and_or_b32", lambda x, y, z: (x & y) | z, d, [a, b, c])

    def v_cmp_eq_u32(self, a, b):
        """vcc = (a == b) per lane"""
        a, b = self._lit(a), self._lit(b)

        def emu(st):
            m = self._vsrc(st, a) == self._vsrc(st, b)
            st.special["vcc"] = int(sum(1 << i for i in range(64) if m[i]))
        self.add(f"v_cmp_eq_u32 vcc, {a}, {b}", emu, "valu", [a, b], [VCC])

    def v_cndmask_b32(self, d, a, b):
        """d = vcc ? b : a"""
        a, b = self._lit(a), self._lit(b)

        def emu(st):
            vcc = st.special["vcc"]
            m = np.array([(vcc >> i) & 1 for i in range(64)], dtype=bool)
            self._vwrite(st, d, np.where(m, self._vsrc(st, b), self._vsrc(st, a)))
        self.add(f"v_cndmask_b32 {d}, {a}, {b}, vcc", emu, "valu", [a, b, VCC], [d])

    def v_mbcnt_lane_id(self, d, tmp_ok=True):
        """d = lane id (v_mbcnt_lo + v_mbcnt_hi with an all-ones mask)."""
        self.add(f"v_mbcnt_lo_u32_b32 {d}, -1, 0", lambda st: self._vwrite(st, d, np.minimum(np.arange(64), 32).astype(np.uint32)), "valu", [], [d])
        self.add(f"v_mbcnt_hi_u32_b32 {d}, -1, {d}", lambda st: self._vwrite(st, d, np.arange(64, dtype=np.uint32)), "valu", [d], [d])

    def v_readfirstlane_b32(self, d, a):
        self.add(f"v_readfirstlane_b32 {d}, {a}", lambda st: self._swrite(st, d, self._vsrc(st, a)[0]), "valu_sgpr", [a], [d])

    def v_accvgpr_read_b32(self, d, a):
        assert d.file == "v" and a.file == "a"
        self._v_op("v_accvgpr_read_b32", lambda x: x, d, [a], kind="accread")

    def v_accvgpr_write_b32(self, d, a):
        assert d.file == "a"
        self._v_op("v_accvgpr_write_b32", lambda x: x, d, [a], kind="valu")

    # ---------------------------------------------------------------------------------------- VALU (f32)
    @staticmethod
    def _f(x):
        return x.view(np.float32)

    @staticmethod
    def _u(x):
        return np.asarray(x, dtype=np.float32).view(np.uint32)

    def v_add_f32(self, d, a, b):
        self._v_op("v_add_f32", lambda x, y: self._u(self._f(x) + self._f(y)), d, [a, b])

    def v_mul_f32(self, d, a, b):
        self._v_op("v_mul_f32", lambda x, y: self._u(self._f(x) * self._f(y)), d, [a, b])

    def v_max_f32(self, d, a, b):
        self._v_op("v_max_f32", lambda x, y: self._u(np.fmax(self._f(x), self._f(y))), d, [a, b])

    def v_sub_f32(self, d, a, b):
        self._v_op("v_sub_f32", lambda x, y: self._u(self._f(x) - self._f(y)), d, [a, b])

    def v_max3_f32(self, d, a, b, c):
        self._v_op("v_max3_f32", lambda x, y, z: self._u(np.fmax(np.fmax(self._f(x), self._f(y)), self._f(z))), d, [a, b, c])

    def v_min_u32(self, d, a, b):
        self._v_op("v_min_u32", lambda x, y: np.minimum(x, y), d, [a, b])

    def v_rcp_f32(self, d, a):
        with np.errstate(divide="ignore"):
            self._v_op("v_rcp_f32", lambda x: self._u((1.0 / self._f(x).astype(np.float64)).astype(np.float32)), d, [a], kind="trans")

    def v_pk_add_f32(self, d, a, b):
        """d.lo = a.lo + b.lo, d.hi = a.hi + b.hi on 64-bit register pairs"""
        assert d.n == 2 and a.n == 2 and b.n == 2

        def emu(st):
            for h in range(2):
                self._vwrite(st, d[h], self._u(self._f(self._vsrc(st, a[h])) + self._f(self._vsrc(st, b[h]))))
        self.add(f"v_pk_add_f32 {d}, {a}, {b}", emu, "valu", [a, b], [d])

    def v_pk_mul_f32(self, d, a, b, bcast_a=False):
        """d.lo = a.lo * b.lo, d.hi = a.hi * b.hi; bcast_a: a's LOW dword feeds both halves (op_sel_hi:[0,1])"""
        assert d.n == 2 and a.n == 2 and b.n == 2

        def emu(st):
            lo = self._f(self._vsrc(st, a[0])) * self._f(self._vsrc(st, b[0]))
            hi = self._f(self._vsrc(st, a[0 if bcast_a else 1])) * self._f(self._vsrc(st, b[1]))
            self._vwrite(st, d[0], self._u(lo))
            self._vwrite(st, d[1], self._u(hi))
        self.add(f"v_pk_mul_f32 {d}, {a}, {b}" + (" op_sel_hi:[0,1]" if bcast_a else ""), emu, "valu", [a, b], [d])

    def v_cmp_neq_f32(self, a, b):
        """vcc = !(a == b) per lane (true for unordered)"""
        a, b = self._lit(a), self._lit(b)

        def emu(st):
            m = ~(self._f(self._vsrc(st, a)) == self._f(self._vsrc(st, b)))
            st.special["vcc"] = int(sum(1 << i for i in range(64) if m[i]))
        self.add(f"v_cmp_neq_f32 vcc, {a}, {b}", emu, "valu", [a, b], [VCC])

    def v_cmp_lt_f32(self, a, b):
        """vcc = a < b per lane (false for unordered); VOPC: a may be an SGPR / inline constant, b is a VGPR"""
        a, b = self._lit(a), self._lit(b)

        def emu(st):
            with np.errstate(invalid="ignore"):
                m = self._f(self._vsrc(st, a)) < self._f(self._vsrc(st, b))
            st.special["vcc"] = int(sum(1 << i for i in range(64) if m[i]))
        self.add(f"v_cmp_lt_f32 vcc, {a}, {b}", emu, "valu", [a, b], [VCC])

    def v_cmp_gt_f32(self, a, b):
        """vcc = a > b per lane (false for unordered)"""
        a, b = self._lit(a), self._lit(b)

        def emu(st):
            with np.errstate(invalid="ignore"):
                m = self._f(self._vsrc(st, a)) > self._f(self._vsrc(st, b))
            st.special["vcc"] = int(sum(1 << i for i in range(64) if m[i]))
        self.add(f"v_cmp_gt_f32 vcc, {a}, {b}", emu, "valu", [a, b], [VCC])

    def s_cbranch_vccz(self, label):
        self.add(f"s_cbranch_vccz {label}", lambda st: label if st.special["vcc"] == 0 else None, "branch", [VCC], [])

    def s_cbranch_vccnz(self, label):
        self.add(f"s_cbranch_vccnz {label}", lambda st: label if st.special["vcc"] != 0 else None, "branch", [VCC], [])

    def v_fma_f32(self, d, a, b, c):
        def fn(x, y, z):
            return self._u((self._f(x).astype(np.float64) * self._f(y).astype(np.float64) + self._f(z).astype(np.float64)).astype(np.float32))
        self._v_op("v_fma_f32", fn, d, [a, b, c])

    def v_fmac_f32(self, d, a, b):
        """d = a * b + d"""
        a, b = self._lit(a), self._lit(b)

        def emu(st):
            x, y, z = self._f(self._vsrc(st, a)), self._f(self._vsrc(st, b)), self._f(self._vsrc(st, d))
            self._vwrite(st, d, self._u((x.astype(np.float64) * y.astype(np.float64) + z.astype(np.float64)).astype(np.float32)))
        self.add(f"v_fmac_f32 {d}, {a}, {b}", emu, "valu", [a, b, d], [d])

    def v_fmaak_f32(self, d, a, b, k):
        """d = a * b + K (32-bit literal)"""
        kk = Lit(float(k))

        def fn(x, y):
            return self._u((self._f(x).astype(np.float64) * self._f(y).astype(np.float64) + np.float64(np.float32(k))).astype(np.float32))
        self.add(f"v_fmaak_f32 {d}, {a}, {b}, {kk}", lambda st: self._vwrite(st, d, fn(self._vsrc(st, a), self._vsrc(st, b))), "valu", [a, b], [d])

    def v_pk_fma_f32(self, d, a, b, c, bcast=(False, False, False), neg_a=False):
        """Packed 2 x f32: d.lo = a.lo*b.lo + c.lo, d.hi = a.hi*b.hi + c.hi on 64-bit register pairs.  bcast[i]: source i feeds its LOW
        dword to both halves (op_sel_hi = 0: a constant kept once); a float for c means an inline constant; neg_a negates a (both halves)."""
        assert d.n == 2 and a.n == 2 and (isinstance(b, float) or b.n == 2)

        def half(st, o, h, bc):
            if isinstance(o, float):
                return np.full(64, np.float32(o), np.float32).astype(np.float64)
            idx = o.idx + (0 if bc else h)
            if o.file == "s":
                return np.full(64, np.uint32(st.s[idx]), np.uint32).view(np.float32).astype(np.float64)
            return (st.v if o.file == "v" else st.a)[idx].view(np.float32).astype(np.float64)

        def emu(st):
            res = []
            for h in range(2):
                x = half(st, a, h, bcast[0]) * (-1.0 if neg_a else 1.0)
                res.append((x * half(st, b, h, bcast[1]) + half(st, c, h, bcast[2])).astype(np.float32).view(np.uint32))
            st.v[d.idx], st.v[d.idx + 1] = res[0], res[1]
        osh = ",".join("0" if bc else "1" for bc in bcast)
        txt = f"v_pk_fma_f32 {d}, {a}, {b}, {Lit(c) if isinstance(c, float) else c} op_sel_hi:[{osh}]"
        if neg_a:
            txt += " neg_lo:[1,0,0] neg_hi:[1,0,0]"
        self.add(txt, emu, "valu", [o for o in (a, b, c) if isinstance(o, Reg)], [d])

    def v_exp_f32(self, d, a):
        with np.errstate(over="ignore", under="ignore"):
            self._v_op("v_exp_f32", lambda x: self._u(np.exp2(self._f(x).astype(np.float64)).astype(np.float32)), d, [a], kind="trans")

    def v_cvt_pk_f16_f32(self, d, a, b):
        def fn(x, y):
            lo = self._f(x).astype(np.float16).view(np.uint16).astype(np.uint32)
            hi = self._f(y).astype(np.float16).view(np.uint16).astype(np.uint32)
            return lo | (hi << 16)
        self._v_op("v_cvt_pk_f16_f32", fn, d, [a, b])

    def v_permlane32_swap_b32(self, d, s_):
        """lanes 32..63 of d swap with lanes 0..31 of s_ (both registers are written)."""
        def emu(st):
            x, y = st.v[d.idx].copy(), st.v[s_.idx].copy()
            nx, ny = x.copy(), y.copy()
            nx[32:] = y[:32]
            ny[:32] = x[32:]
            st.v[d.idx], st.v[s_.idx] = nx, ny
        self.add(f"v_permlane32_swap_b32 {d}, {s_}", emu, "permlane", [d, s_], [d, s_])

    # ---------------------------------------------------------------------------------------- MFMA
    def v_mfma_f32_32x32x16_f16(self, d, a, b, c):
        """D[32x32] = A[32x16] B[16x32] + C.  Lane l: A[m = l%32][k = 8*(l/32)+e], B[k = 8*(l/32)+e][n = l%32];
        D register r of lane l: row (r&3) + 8*(r>>2) + 4*(l>>5), column l%32 (MI355X guide §3)."""
        assert d.n == 16 and a.n == 4 and b.n == 4
        czero = not isinstance(c, Reg)
        if czero:
            assert c == 0
        else:
            assert c.n == 16 and c.file == d.file, "C and D share the ACC_CD bit: same register file"

        def unpack(st, r):
            src = st.v if r.file == "v" else st.a
            w = np.stack([src[r.idx + i] for i in range(4)], axis=1)        # [64 lanes, 4 dwords]
            return w.view(np.float16).reshape(64, 8).astype(np.float32)      # 8 halves per lane

        def emu(st):
            fa, fb = unpack(st, a), unpack(st, b)
            Am = np.zeros((32, 16), np.float32)
            Bm = np.zeros((16, 32), np.float32)
            for h in range(2):
                Am[:, 8 * h:8 * h + 8] = fa[32 * h:32 * h + 32]
                Bm[8 * h:8 * h + 8, :] = fb[32 * h:32 * h + 32].T
            Dm = Am.astype(np.float64) @ Bm.astype(np.float64)
            tgt = st.v if d.file == "v" else st.a
            lane = np.arange(64)
            for r in range(16):
                row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                val = Dm[row, lane & 31]
                if not czero:
                    csrc = st.v if c.file == "v" else st.a
                    val = val + csrc[c.idx + r].view(np.float32).astype(np.float64)
                tgt[d.idx + r] = val.astype(np.float32).view(np.uint32)
        cs = "0" if czero else str(c)
        self.add(f"v_mfma_f32_32x32x16_f16 {d}, {a}, {b}, {cs}", emu, "mfma", [a, b] + ([] if czero else [c]), [d])

    # ---------------------------------------------------------------------------------------- LDS
    def ds_read_b128(self, d, addr, offset=0):
        assert d.n == 4 and 0 <= offset <= 65535 and offset % 16 == 0

        def emu(st):
            ad = st.v[addr.idx].astype(np.int64) + offset
            assert (ad % 16 == 0).all() and (ad >= 0).all() and (ad + 16 <= st.wg.lds.size).all(), "ds_read_b128 address"

            def sample():
                idx = ad[:, None] + np.arange(16)[None, :]
                return st.wg.lds[idx].reshape(64, 16).copy().view(np.uint32).reshape(64, 4)
            tgt = st.v if d.file == "v" else st.a
            if st.wg.ds_lazy:
                for i in range(4):
                    tgt[d.idx + i] = np.full(64, POISON, np.uint32)
                box = {"w": None}

                def sample_now():                 # called early by a later ds_write of the same wave (in-order LDS queue)
                    if box["w"] is None:
                        box["w"] = sample()

                def deliver():
                    sample_now()
                    for i in range(4):
                        tgt[d.idx + i] = box["w"][:, i].copy()
                deliver.sample_now = sample_now
                st.issue_lgkm(deliver)
            else:
                w = sample()
                for i in range(4):
                    tgt[d.idx + i] = w[:, i].copy()
                st.issue_lgkm(lambda: None)
        self.add(f"ds_read_b128 {d}, {addr}" + (f" offset:{offset}" if offset else ""), emu, "ds", [addr], [d])

    def ds_write_b64(self, addr, data, offset=0):
        """A wave's LDS instructions execute in issue order: its own earlier reads sample LDS before this write lands."""
        assert data.n == 2 and 0 <= offset <= 65535 and offset % 8 == 0

        def emu(st):
            st.flush_ds_reads()
            ad = st.v[addr.idx].astype(np.int64) + offset
            assert (ad % 8 == 0).all() and (ad >= 0).all() and (ad + 8 <= st.wg.lds.size).all(), "ds_write_b64 address"
            src = st.v if data.file == "v" else st.a
            w = np.stack([src[data.idx], src[data.idx + 1]], axis=1).copy().view(np.uint8).reshape(64, 8)
            m = st.exec_mask()
            for l in np.nonzero(m)[0]:          # lane order: a higher lane wins an address clash, as in hardware
                st.wg.lds[ad[l]:ad[l] + 8] = w[l]
            st.issue_lgkm(lambda: None)
        self.add(f"ds_write_b64 {addr}, {data}" + (f" offset:{offset}" if offset else ""), emu, "ds", [addr, data], [])

    def ds_write_b128(self, addr, data, offset=0):
        assert data.n == 4 and 0 <= offset <= 65535 and offset % 16 == 0

        def emu(st):
            st.flush_ds_reads()
            ad = st.v[addr.idx].astype(np.int64) + offset
            assert (ad % 16 == 0).all() and (ad >= 0).all() and (ad + 16 <= st.wg.lds.size).all(), "ds_write_b128 address"
            src = st.v if data.file == "v" else st.a
            w = np.stack([src[data.idx + k] for k in range(4)], axis=1).copy().view(np.uint8).reshape(64, 16)
            for l in np.nonzero(st.exec_mask())[0]:
                st.wg.lds[ad[l]:ad[l] + 16] = w[l]
            st.issue_lgkm(lambda: None)
        self.add(f"ds_write_b128 {addr}, {data}" + (f" offset:{offset}" if offset else ""), emu, "ds", [addr, data], [])

    def _ds_read_n(self, name, nbytes, d, addr, offset, post=None):
        """nbytes (4 / 8) per lane; post(bytes[64, nbytes]) -> dwords[64, nbytes / 4] (the transposed read permutes across lanes)"""
        ndw = nbytes // 4
        assert d.n == ndw and 0 <= offset <= 65535 and offset % nbytes == 0

        def emu(st):
            ad = st.v[addr.idx].astype(np.int64) + offset
            assert (ad % nbytes == 0).all() and (ad >= 0).all() and (ad + nbytes <= st.wg.lds.size).all(), name + " address"

            def sample():
                raw = st.wg.lds[ad[:, None] + np.arange(nbytes)[None, :]].reshape(64, nbytes).copy()
                if post is not None:
                    raw = post(raw)
                return raw.view(np.uint32).reshape(64, ndw)
            tgt = st.v if d.file == "v" else st.a
            if st.wg.ds_lazy:
                for i in range(ndw):
                    tgt[d.idx + i] = np.full(64, POISON, np.uint32)
                box = {"w": None}

                def sample_now():
                    if box["w"] is None:
                        box["w"] = sample()

                def deliver():
                    sample_now()
                    for i in range(ndw):
                        tgt[d.idx + i] = box["w"][:, i].copy()
                deliver.sample_now = sample_now
                st.issue_lgkm(deliver)
            else:
                w = sample()
                for i in range(ndw):
                    tgt[d.idx + i] = w[:, i].copy()
                st.issue_lgkm(lambda: None)
        self.add(f"{name} {d}, {addr}" + (f" offset:{offset}" if offset else ""), emu, "ds", [addr], [d])

    def ds_read_b32(self, d, addr, offset=0):
        self._ds_read_n("ds_read_b32", 4, d, addr, offset)

    def ds_read_b64(self, d, addr, offset=0):
        self._ds_read_n("ds_read_b64", 8, d, addr, offset)

    def ds_read_b64_tr_b16(self, d, addr, offset=0):
        """Transposed read: in every 16-lane group, lane i hands in the address of 4 consecutive 16-bit values D_i[0..3];
        lane c of the group receives element j = D_{4j + c/4}[c % 4] (a [4][16] block: lane c gets column c)."""
        def post(raw):
            h = raw.view(np.uint16).reshape(4, 16, 4)             # [group, lane i, element]
            out = np.empty_like(h)
            for c in range(16):
                for j in range(4):
                    out[:, c, j] = h[:, 4 * j + c // 4, c % 4]
            return out.reshape(64, 4).view(np.uint8).reshape(64, 8)
        self._ds_read_n("ds_read_b64_tr_b16", 8, d, addr, offset, post)

    def ds_write_b32(self, addr, data, offset=0):
        assert data.n == 1 and 0 <= offset <= 65535 and offset % 4 == 0

        def emu(st):
            st.flush_ds_reads()
            ad = st.v[addr.idx].astype(np.int64) + offset
            assert (ad % 4 == 0).all() and (ad >= 0).all() and (ad + 4 <= st.wg.lds.size).all(), "ds_write_b32 address"
            src = st.v if data.file == "v" else st.a
            w = src[data.idx].copy().view(np.uint8).reshape(64, 4)
            for l in np.nonzero(st.exec_mask())[0]:
                st.wg.lds[ad[l]:ad[l] + 4] = w[l]
            st.issue_lgkm(lambda: None)
        self.add(f"ds_write_b32 {addr}, {data}" + (f" offset:{offset}" if offset else ""), emu, "ds", [addr, data], [])

    # ---------------------------------------------------------------------------------------- VMEM
    @staticmethod
    def _rsrc_base(st, rsrc):
        """Kernarg pointers are 0 in the emulator (the SGPR pair carries the numpy buffer as a tag): the numeric base of a resource is
        whatever the program ADDED to the pointer, i.e. a byte offset into the tagged buffer."""
        return int(st.s[rsrc.idx]) | ((int(st.s[rsrc.idx + 1]) & 0xFFFF) << 32)

    def tag_copy(self, d, s_):
        """Emulator only: SGPR d now points into the buffer that SGPR s_ is tagged with (pointer arithmetic keeps the numeric offset)."""
        def emu(st):
            st.sobj[d.idx] = st.sobj[s_.idx]
        self.add(f"; {d} points into the buffer of {s_}", emu, "comment")

    def buffer_load_dwordx4(self, d, voff, rsrc, soff, offset=0):
        """buffer_load_dwordx4 d, voff, rsrc, soff offen [offset:imm]: 16 B per lane into VGPRs or AGPRs (delivered at the retiring vmcnt)."""
        assert d.n == 4 and rsrc.n == 4 and 0 <= offset < 4096

        def emu(st):
            buf = st.sobj[rsrc.idx]
            nrec = int(st.s[rsrc.idx + 2])
            base = self._rsrc_base(st, rsrc)
            off = st.v[voff.idx].astype(np.int64) + int(st.s[soff.idx]) + offset
            m = st.exec_mask()
            assert (off[m] >= 0).all() and (off[m] + 16 <= min(nrec, buf.size - base)).all(), "buffer_load out of range"
            data = np.zeros((64, 16), np.uint8)
            data[m] = buf[(base + off[m][:, None] + np.arange(16)[None, :])]
            w = data.view(np.uint32).reshape(64, 4)
            tgt = st.v if d.file == "v" else st.a
            for i in range(4):
                tgt[d.idx + i] = np.full(64, POISON, np.uint32)

            def deliver():
                for i in range(4):
                    tgt[d.idx + i] = w[:, i].copy()
            st.issue_vm(deliver)
        self.add(f"buffer_load_dwordx4 {d}, {voff}, {rsrc}, {soff} offen" + (f" offset:{offset}" if offset else ""), emu, "vmem_load", [voff, rsrc, soff], [d])

    def buffer_load_lds(self, nbytes, voff, rsrc, soff):
        """buffer_load_dword{,x4} voff, rsrc, soff offen lds: every active lane moves nbytes from
        base + voff + soff to LDS[M0 + lane * nbytes]."""
        assert nbytes in (4, 16) and rsrc.n == 4

        def emu(st):
            buf = st.sobj[rsrc.idx]
            nrec = int(st.s[rsrc.idx + 2])
            base = self._rsrc_base(st, rsrc)
            off = st.v[voff.idx].astype(np.int64) + int(st.s[soff.idx])
            m = st.exec_mask()
            assert (off[m] >= 0).all() and (off[m] + nbytes <= min(nrec, buf.size - base)).all(), \
                f"LDS-DMA source out of range: max {off[m].max()} + {nbytes} > {min(nrec, buf.size - base)}"
            assert (off[m] % nbytes == 0).all()
            lbase = int(st.special["m0"])
            dst = lbase + np.arange(64, dtype=np.int64) * nbytes
            assert lbase % nbytes == 0 and dst[m].max() + nbytes <= st.wg.lds.size
            data = buf[(base + off[m][:, None] + np.arange(nbytes)[None, :])].copy()
            dsel = dst[m]

            def land():
                st.wg.lds[(dsel[:, None] + np.arange(nbytes)[None, :])] = data
            if st.wg.dma_lazy:
                st.issue_vm(land)
            else:
                land()
                st.issue_vm(lambda: None)
        name = "buffer_load_dwordx4" if nbytes == 16 else "buffer_load_dword"
        self.add(f"{name} {voff}, {rsrc}, {soff} offen lds", emu, "vmem_lds", [voff, rsrc, soff, M0], [])

    def buffer_store_dwordx4(self, data, voff, rsrc, soff, nt=False, pol=""):
        assert data.n == 4

        def emu(st):
            buf = st.sobj[rsrc.idx]
            off = self._rsrc_base(st, rsrc) + st.v[voff.idx].astype(np.int64) + int(st.s[soff.idx])
            m = st.exec_mask()
            assert (off[m] >= 0).all() and (off[m] + 16 <= buf.size).all() and (off[m] % 16 == 0).all(), "store out of range"
            src = st.v if data.file == "v" else st.a
            w = np.stack([src[data.idx + i] for i in range(4)], axis=1)[m]
            buf[(off[m][:, None] + np.arange(16)[None, :])] = w.copy().view(np.uint8).reshape(-1, 16)
            st.wg.stores.append((id(buf), off[m].copy()))
            st.issue_vm(lambda: None)
        self.add(f"buffer_store_dwordx4 {data}, {voff}, {rsrc}, {soff} offen" + (" nt" if nt else "") + (" " + pol if pol else ""), emu, "vmem_store",
                 [data, voff, rsrc, soff], [])


# ------------------------------------------------------------------------------------------------ emulator
class WaveState:
    def __init__(self, wg, wave_id):
        self.wg, self.wave_id = wg, wave_id
        self.v = np.full((256, 64), POISON, np.uint32)
        self.a = np.full((256, 64), POISON, np.uint32)
        self.s = [0xDEAD0000] * 104
        self.sobj = {}
        self.mem_objs = wg.mem_objs
        self.special = {"m0": 0, "vcc": 0, "exec_lo": 0xFFFFFFFF, "exec_hi": 0xFFFFFFFF}
        self.scc = 0
        self.vm, self.lgkm = [], []
        self.pc = 0
        self.done = False
        self.at_barrier = False
        self.max_vm = 0
        self._exec_key, self._exec_arr = None, None

    def exec_mask(self):
        e = self.special["exec_lo"] | (self.special["exec_hi"] << 32)
        if e != self._exec_key:
            self._exec_key = e
            self._exec_arr = np.array([(e >> i) & 1 for i in range(64)], dtype=bool)
        return self._exec_arr

    def issue_vm(self, fn):
        self.vm.append(fn)
        self.max_vm = max(self.max_vm, len(self.vm))
        assert len(self.vm) <= 63, "more than 63 VMEM operations outstanding: vmcnt saturates"

    def issue_lgkm(self, fn, smem=False):
        self.lgkm.append((fn, smem))
        assert len(self.lgkm) <= 15 or True

    def flush_ds_reads(self):
        for fn, _ in self.lgkm:
            if hasattr(fn, "sample_now"):
                fn.sample_now()

    def wait_vm(self, n):
        while len(self.vm) > n:
            self.vm.pop(0)()

    def wait_lgkm(self, n):
        if any(sm for _, sm in self.lgkm):
            assert n == 0, "SMEM returns out of order: only lgkmcnt(0) is meaningful while an s_load is outstanding"
        while len(self.lgkm) > n:
            self.lgkm.pop(0)[0]()


class Workgroup:
    def __init__(self, prog, n_waves=4, lds_bytes=160 * 1024, dma_lazy=False, ds_lazy=False, order="0123"):
        self.prog, self.lds = prog, np.zeros(lds_bytes, np.uint8)
        self.lds[:] = 0xA5
        self.dma_lazy, self.ds_lazy, self.order = dma_lazy, ds_lazy, order
        self.mem_objs = {}
        self.stores = []
        self.waves = [WaveState(self, w) for w in range(n_waves)]
        self.executed = 0

    def run(self, max_ins=50_000_000):
        prog, labels = self.prog.ins, self.prog.labels
        n = len(self.waves)
        if self.order == "0123":
            seq = list(range(n))
        elif self.order == "3210":
            seq = list(range(n))[::-1]
        else:
            seq = None
        rr = 0
        while True:
            live = [w for w in self.waves if not w.done]
            if not live:
                break
            runnable = [w for w in live if not w.at_barrier]
            if not runnable:
                assert all(w.at_barrier for w in live) and len(live) == n, "barrier with exited waves"
                for w in live:
                    w.at_barrier = False
                continue
            if seq is not None:
                st = min(runnable, key=lambda w: seq.index(w.wave_id))
                burst = 1 << 30
            else:                       # round-robin, a handful of instructions at a time
                st = runnable[rr % len(runnable)]
                rr += 1
                burst = 7
            for _ in range(burst):
                if st.pc >= len(prog):
                    st.done = True
                    st.wait_vm(0)
                    st.wait_lgkm(0)
                    break
                ins = prog[st.pc]
                st.pc += 1
                r = ins.emu(st)
                self.executed += 1
                assert self.executed < max_ins, "emulation ran away"
                if r == "BARRIER":
                    st.at_barrier = True
                    break
                if isinstance(r, str):
                    st.pc = labels[r]
        return self


# ------------------------------------------------------------------------------------------------ static hazard check
def check_hazards(prog, verbose=False):
    """Wait states between producer and consumer for the software-managed hazards this code can hit (counts follow LLVM's
    GCNHazardRecognizer for gfx940/gfx950 and the MI355X guide §5.7).  Straight-line approximation: the instruction list is
    walked in program order, branches are ignored (loops are laid out so that the fall-through order is the hot order).
    Every instruction is one wait state, s_nop N is N+1."""
    RULES = [
        # (producer kind, consumer kind, register relation, states needed, description)
        ("valu", "mfma", "w->r", 2, "VALU write -> MFMA operand"),
        ("accread", "mfma", "w->r", 2, "VALU write -> MFMA operand"),
        ("trans", "mfma", "w->r", 2, "VALU write -> MFMA operand"),
        ("mfma", "valu", "w->rw", 12, "8-pass MFMA result -> VALU"),
        ("mfma", "trans", "w->rw", 12, "8-pass MFMA result -> VALU"),
        ("mfma", "accread", "w->rw", 12, "8-pass MFMA result -> v_accvgpr_read"),
        ("mfma", "vmem_store", "w->r", 12, "8-pass MFMA result -> store data"),
        ("mfma", "ds", "w->rw", 12, "8-pass MFMA result -> LDS op on the register"),
        ("valu", "permlane", "w->r", 2, "VALU write -> v_permlane32_swap"),
        ("trans", "permlane", "w->r", 2, "VALU write -> v_permlane32_swap"),
        ("accread", "permlane", "w->r", 2, "VALU write -> v_permlane32_swap"),
        ("valu_sgpr", "vmem_lds", "w->r", 5, "VALU-written SGPR -> VMEM"),
        ("valu_sgpr", "vmem_store", "w->r", 5, "VALU-written SGPR -> VMEM"),
    ]
    problems2 = []
    hist = []
    for ins in prog.ins:
        if ins.kind in ("label", "comment"):
            continue
        for between, p in hist:
            for pk, ck, rel, need, desc in RULES:
                if p.kind != pk or ins.kind != ck or between >= need:
                    continue
                pw = set(p.writes)
                touched = set(ins.reads) | (set(ins.writes) if rel == "w->rw" else set())
                if pw & touched:
                    problems2.append(f"{desc}: '{p.text}' -> '{ins.text}' has {between} wait states, needs {need}")
            if p.kind == "salu" and ("x", "m0") in p.writes and ins.kind == "vmem_lds" and between < 1:
                problems2.append(f"m0 write -> LDS-DMA back to back: '{p.text}' -> '{ins.text}'")
            if p.kind == "vmem_store" and ins.kind in ("valu", "trans", "accread", "permlane", "mfma", "ds") and between < 2:
                if {r for r in p.reads if r[0] in ("v", "a")} & set(ins.writes):
                    problems2.append(f"store data overwritten too early: '{p.text}' -> '{ins.text}' ({between} wait states)")
        states = ins.extra if ins.kind == "nop" else 1
        hist = [(b + states, p) for b, p in hist if b + states < 20]
        hist.append((0, ins))
    if verbose:
        for p_ in problems2:
            print("HAZARD", p_)
    return problems2


def check_footprint(prog, first_sgpr=36, first_vgpr=4):
    """The register contract of a generated one-statement body (sam_road_amd/csrc/asm_body.hpp): it writes no SGPR below `first_sgpr`
    and no VGPR below `first_vgpr` (the compiler keeps the statement's operands there), and every narrowing of exec is followed by the
    write that restores all ones — the AMDGPU backend requires exec to leave an asm statement as it entered and accepts neither exec nor
    m0 on a clobber list (both are reserved registers).  Returns a list of violations."""
    bad = []
    exec_state = {"exec_lo": True, "exec_hi": True}        # True = all ones
    for ins in prog.ins:
        for f, i in ins.writes:
            if f == "s" and i < first_sgpr:
                bad.append(f"writes s{i}: '{ins.text}'")
            if f == "v" and i < first_vgpr:
                bad.append(f"writes v{i}: '{ins.text}'")
            if f == "x" and i in exec_state:
                exec_state[i] = ins.text.replace(" ", "").endswith(",-1")
        if ins.kind in ("branch", "barrier", "label") and not all(exec_state.values()):
            bad.append(f"exec is narrowed across '{ins.text}'")
    if not all(exec_state.values()):
        bad.append("exec is not restored at the end of the body")
    return bad
