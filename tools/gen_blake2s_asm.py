#!/usr/bin/env python3
"""gen_blake2s_asm.py — writes the Blake2s compression F(0, m, 0, 0) (the Merkle shape of frieda's trees: stwo's
Blake2sMerkleHasher::hash_node from the all-zero state) as ONE hand-scheduled gfx950 inline-asm block per message shape.

Why generated text: the throughput form of frieda_amd/csrc/blake2s.h depends on the ORDER of the instruction stream (runs of one VALU
rate class, the wave's priority switched at the run boundaries).  Written as C++ with data-flow pins the order survives the compiler, but
every pin is an inline-asm statement and the compiler puts an `s_nop 0` behind each one (1556 among the 12127 instructions of
tree5r<LEAF4>).  One asm block per compression has no statement boundaries inside: the order is the text.

The generator tracks which state words are still compile-time constants (h = 0, IV) so that round 0 folds exactly as the compiler folds it
(a = 0 + 0 + m is a move, d = IV ^ a takes the literal, b = 0 ^ c is c), and turns `+ 0` message words of the leaf shape
(4 column words + 12 zeros) into two-operand adds.

usage: python tools/gen_blake2s_asm.py > frieda_amd/csrc/blake2s_asm.h          (the product header: variants `node`, `leaf`)
       python tools/gen_blake2s_asm.py --bench > tools/blake2s_asm_variants.h   (every variant of VARIANTS, for tools/blake2s_asm.hip)
"""
import sys

IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
SIGMA = [
    [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
    [11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4], [7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8],
    [9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13], [2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9],
    [12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11], [13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10],
    [6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5], [10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0],
]
M32 = 0xFFFFFFFF


class Gen:
    """Emits the instruction stream of one compression.  State word i lives in asm operand %i (an early-clobber output); message word
    j in operand %(16 + j).  `val[i]`: None = the register holds the value, an int = still a known constant (not yet in the register),
    ('alias', j) = equal to state word j's register."""

    def __init__(self, n_msg, prio_rot, prio_add3, prio_fast, rot16="alignbit", split_leaf_adds=False, prio_mode="runs"):
        self.n_msg = n_msg  # message words 0 .. n_msg-1 are registers, the rest are zeros
        self.prio = {"rot": prio_rot, "add3": prio_add3, "fast": prio_fast}
        self.rot16 = rot16
        self.split_leaf_adds = split_leaf_adds
        self.prio_mode = prio_mode
        self.lines = []
        self.val = [0] * 8 + list(IV)
        self.cur_prio = None
        self.count = {"slow": 0, "fast": 0, "s": 0}

    # -- helpers
    def reg(self, i):
        return "%%%d" % i

    def mreg(self, j):
        return "%%%d" % (16 + j)

    def setprio(self, kind):
        p = self.prio[kind]
        if p is None or p == self.cur_prio:
            return
        self.lines.append("s_setprio %d" % p)
        self.count["s"] += 1
        self.cur_prio = p

    def emit(self, cls, text):
        self.lines.append(text)
        self.count[cls] += 1

    def src(self, i):
        """source operand text for state word i (register, or literal while it is a constant)"""
        v = self.val[i]
        if v is None:
            return self.reg(i)
        if isinstance(v, tuple):
            return "%%%d" % v[1] if v[0] == "in" else self.reg(v[1])  # ('in', k): asm input operand %k; ('alias', j): state word j's register
        return lit(v)

    # -- the steps of G over one (a, b, c, d) quadruple.  Each returns a list of (class, text); constants fold, `x ^ 0` aliases.
    def written(self, i):
        assert not any(isinstance(v, tuple) and v[0] == "alias" and v[1] == i for v in self.val), "a register is overwritten while an alias still reads it"
        self.val[i] = None

    def add_abm(self, a, b, j):
        """a = a + b + m[j]"""
        k, regs = 0, []
        for i in (a, b):
            if isinstance(self.val[i], int):
                k = (k + self.val[i]) & M32
            else:
                regs.append(self.src(i))
        if j < self.n_msg:
            regs.append(self.mreg(j))
        if not regs:
            self.val[a] = k
            return []
        A = self.reg(a)
        if len(regs) == 1:
            ops = [("fast", "v_mov_b32 %s, %s" % (A, regs[0]) if k == 0 else "v_add_u32 %s, %s, %s" % (A, lit(k), regs[0]))]
        elif len(regs) == 2 and k == 0:
            ops = [("fast", "v_add_u32 %s, %s, %s" % (A, regs[0], regs[1]))]
        elif len(regs) == 2:  # (VOP3 takes no literal on gfx9: two two-operand adds)
            ops = [("fast", "v_add_u32 %s, %s, %s" % (A, lit(k), regs[0])), ("fast", "v_add_u32 %s, %s, %s" % (A, A, regs[1]))]
        else:
            ops = [("slow", "v_add3_u32 %s, %s, %s, %s" % (A, regs[0], regs[1], regs[2]))]
        self.written(a)
        return ops

    def xor_into(self, d, a):
        """d ^= a"""
        vd, va = self.val[d], self.val[a]
        if isinstance(vd, int) and isinstance(va, int):
            self.val[d] = vd ^ va
            return []
        if isinstance(va, int):
            if va == 0:
                return []
            ops = [("fast", "v_xor_b32 %s, %s, %s" % (self.reg(d), lit(va), self.src(d)))]
        elif isinstance(vd, int):
            if vd == 0:
                assert self.val[a] is None or self.val[a][0] == "alias"
                self.val[d] = ("alias", a if self.val[a] is None else self.val[a][1])
                return []
            ops = [("fast", "v_xor_b32 %s, %s, %s" % (self.reg(d), lit(vd), self.src(a)))]
        else:
            ops = [("fast", "v_xor_b32 %s, %s, %s" % (self.reg(d), self.src(d), self.src(a)))]
        self.written(d)
        return ops

    def rot(self, d, r):
        if isinstance(self.val[d], int):
            x = self.val[d]
            self.val[d] = ((x >> r) | (x << (32 - r))) & M32
            return []
        s = self.src(d)
        if r == 16 and self.rot16 == "pk":
            ops = [("fast", "v_pk_add_u16 %s, %s, 0 op_sel:[1,0] op_sel_hi:[0,0]" % (self.reg(d), s))]
        else:
            ops = [("slow", "v_alignbit_b32 %s, %s, %s, %d" % (self.reg(d), s, s, r))]
        self.written(d)
        return ops

    def add_cd(self, c, d):
        vc, vd = self.val[c], self.val[d]
        if isinstance(vc, int) and isinstance(vd, int):
            self.val[c] = (vc + vd) & M32
            return []
        if isinstance(vc, int):
            ops = [("fast", "v_add_u32 %s, %s, %s" % (self.reg(c), lit(vc), self.src(d)))]
        elif isinstance(vd, int):
            ops = [("fast", "v_add_u32 %s, %s, %s" % (self.reg(c), lit(vd), self.src(c)))]
        else:
            ops = [("fast", "v_add_u32 %s, %s, %s" % (self.reg(c), self.src(c), self.src(d)))]
        self.written(c)
        return ops

    def run(self, kind, ops):
        """one run: `ops` = the steps' instruction lists (empty where a step folded away)"""
        ops = [o for group in ops for o in group]
        if not ops:
            return
        if self.prio_mode == "runs":
            if kind == "add3" and self.split_leaf_adds:
                # a run of a += b + m in the leaf shape mixes v_add3 (slow) and v_add (fast): slow ones first at the raised priority
                slow = [o for o in ops if o[0] == "slow"]
                fast = [o for o in ops if o[0] == "fast"]
                if slow:
                    self.setprio("add3")
                    for o in slow:
                        self.emit(*o)
                if fast:
                    self.setprio("fast")
                    for o in fast:
                        self.emit(*o)
                return
            self.setprio(kind if any(o[0] == "slow" for o in ops) else "fast")
            for o in ops:
                self.emit(*o)
        else:  # "class": the priority follows every single instruction's class
            for o in ops:
                self.setprio("fast" if o[0] == "fast" else kind)
                self.emit(*o)

    def half_round(self, r, h, steps=None):
        """steps: None = the whole half-round; else {q: n}: only quadruples q, each up to its first n of the twelve operations of G
        (a += b + x, d ^= a, rot 16, c += d, b ^= c, rot 12, a += b + y, d ^= a, rot 8, c += d, b ^= c, rot 7)"""
        a = [0, 1, 2, 3]
        b = [5, 6, 7, 4] if h else [4, 5, 6, 7]
        c = [10, 11, 8, 9] if h else [8, 9, 10, 11]
        d = [15, 12, 13, 14] if h else [12, 13, 14, 15]
        o = 8 * h
        lim = {q: 12 for q in range(4)} if steps is None else steps

        def Q(k):  # the quadruples that still take operation k (1-based)
            return [q for q in sorted(lim) if lim[q] >= k]

        self.run("add3", [self.add_abm(a[q], b[q], SIGMA[r][o + 2 * q]) for q in Q(1)])
        self.run("fast", [self.xor_into(d[q], a[q]) for q in Q(2)])
        self.run("rot" if self.rot16 != "pk" else "fast", [self.rot(d[q], 16) for q in Q(3)])
        self.run("fast", [self.add_cd(c[q], d[q]) for q in Q(4)] + [self.xor_into(b[q], c[q]) for q in Q(5)])
        self.run("add3", [self.rot(b[q], 12) for q in Q(6)] + [self.add_abm(a[q], b[q], SIGMA[r][o + 2 * q + 1]) for q in Q(7)])
        self.run("fast", [self.xor_into(d[q], a[q]) for q in Q(8)])
        self.run("rot", [self.rot(d[q], 8) for q in Q(9)])
        self.run("fast", [self.add_cd(c[q], d[q]) for q in Q(10)] + [self.xor_into(b[q], c[q]) for q in Q(11)])
        self.run("rot", [self.rot(b[q], 7) for q in Q(12)])

    # ---- the proof-of-work shape (Blake2sChannel::mix_u64 with the nonce: F(digest, [nonce_lo, nonce_hi, 0 x 14], 0, 0, 0, 0)) ----
    # Only the low word of the output is produced (trailing zeros up to 32 bits are decided by it; the caller recomputes the rare survivor in
    # full).  Of round 0's column step only G(v0, v4, v8, v12, m0, m1) depends on the nonce: the other three quadruples are the same for every
    # nonce of a blob and come in as inputs (GRIND_IN below), computed once per claim.  The last half-round runs only what out[0] = h0 ^ v0 ^ v8
    # needs: G(v0, v5, v10, v15) up to its second a += b + y, G(v2, v7, v8, v13) up to its second c += d.
    GRIND_IN = [0, 4, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15]  # state words that arrive in input operands %18 .. %31 (m0 = %16, m1 = %17)

    def grind(self):
        assert self.n_msg == 2
        for k, i in enumerate(self.GRIND_IN):
            self.val[i] = ("in", 18 + k)
        self.val[8], self.val[12] = IV[0], IV[4]
        self.half_round(0, 0, {0: 12})
        self.half_round(0, 1)
        for r in range(1, 9):
            self.half_round(r, 0)
            self.half_round(r, 1)
        self.half_round(9, 0)
        self.half_round(9, 1, {0: 7, 2: 10})
        self.setprio("fast")
        self.emit("fast", "v_xor_b32 %s, %s, %s" % (self.reg(0), self.reg(0), self.reg(8)))
        self.emit("fast", "v_xor_b32 %s, %s, %s" % (self.reg(0), self.reg(0), "%18"))  # ^ h0
        if self.cur_prio not in (None, 0):
            self.lines.append("s_setprio 0")
            self.count["s"] += 1
        return self.lines

    def compression(self):
        for r in range(10):
            self.half_round(r, 0)
            self.half_round(r, 1)
        assert all(v is None for v in self.val)
        # out[i] = v[i] ^ v[8 + i] into the low eight state registers (fast class)
        self.setprio("fast")
        for i in range(8):
            self.emit("fast", "v_xor_b32 %s, %s, %s" % (self.reg(i), self.reg(i), self.reg(8 + i)))
        if self.cur_prio not in (None, 0):
            self.lines.append("s_setprio 0")
            self.count["s"] += 1
        return self.lines


def lit(k):
    return "0x%x" % k if k > 64 else "%d" % k


def reference_compress(m, h=None, only_column_quads=None):
    """plain Python F(h, m, 0, 0) (h = None: the zero state): the generator's own check of the emitted dataflow (tests/test_isa_structure.py
    runs it).  only_column_quads: stop after these quadruples of round 0's column step and return the 16 state words (the grind's inputs)."""
    h = [0] * 8 if h is None else list(h)
    v = list(h) + list(IV)
    rotr = lambda x, r: ((x >> r) | (x << (32 - r))) & M32

    def G(a, b, c, d, x, y):
        v[a] = (v[a] + v[b] + x) & M32
        v[d] = rotr(v[d] ^ v[a], 16)
        v[c] = (v[c] + v[d]) & M32
        v[b] = rotr(v[b] ^ v[c], 12)
        v[a] = (v[a] + v[b] + y) & M32
        v[d] = rotr(v[d] ^ v[a], 8)
        v[c] = (v[c] + v[d]) & M32
        v[b] = rotr(v[b] ^ v[c], 7)

    if only_column_quads is not None:
        s = SIGMA[0]
        for q in only_column_quads:
            G(q, 4 + q, 8 + q, 12 + q, m[s[2 * q]], m[s[2 * q + 1]])
        return v
    for r in range(10):
        s = SIGMA[r]
        G(0, 4, 8, 12, m[s[0]], m[s[1]])
        G(1, 5, 9, 13, m[s[2]], m[s[3]])
        G(2, 6, 10, 14, m[s[4]], m[s[5]])
        G(3, 7, 11, 15, m[s[6]], m[s[7]])
        G(0, 5, 10, 15, m[s[8]], m[s[9]])
        G(1, 6, 11, 12, m[s[10]], m[s[11]])
        G(2, 7, 8, 13, m[s[12]], m[s[13]])
        G(3, 4, 9, 14, m[s[14]], m[s[15]])
    return [h[i] ^ v[i] ^ v[8 + i] for i in range(8)]


def interpret(lines, m, n_out=8):
    """executes the emitted text on integers (operands %0..%15 state, %16.. the inputs `m`): the generator checks itself"""
    regs = {}

    def val(tok):
        tok = tok.strip()
        if tok.startswith("%"):
            i = int(tok[1:])
            return m[i - 16] if i >= 16 else regs[i]
        return int(tok, 0)

    rotr = lambda x, r: ((x >> r) | (x << (32 - r))) & M32
    for ln in lines:
        op, _, rest = ln.partition(" ")
        if op == "s_setprio":
            continue
        args = [t for t in rest.replace(" op_sel:[1,0] op_sel_hi:[0,0]", "").split(",")]
        dst = int(args[0].strip()[1:])
        s = [val(t) for t in args[1:]]
        if op == "v_mov_b32":
            regs[dst] = s[0]
        elif op == "v_add_u32":
            regs[dst] = (s[0] + s[1]) & M32
        elif op == "v_add3_u32":
            regs[dst] = (s[0] + s[1] + s[2]) & M32
        elif op == "v_xor_b32":
            regs[dst] = s[0] ^ s[1]
        elif op == "v_alignbit_b32":
            assert s[0] == s[1]
            regs[dst] = rotr(s[0], s[2])
        elif op == "v_pk_add_u16":
            regs[dst] = rotr(s[0], 16)
        else:
            raise ValueError(ln)
    return [regs[i] for i in range(n_out)]


# name -> (message registers, generator options).  `node` / `leaf` are the product's forms: the run order and priorities of
# blake2s.h's 0xB000 setting (rotate runs at 3, runs that hold a v_add3 at 2, fast-class runs at 0).
VARIANTS = {
    "node": (16, dict(prio_rot=3, prio_add3=2, prio_fast=0)),
    "leaf": (4, dict(prio_rot=3, prio_add3=2, prio_fast=0)),
    "grind": (2, dict(prio_rot=3, prio_add3=2, prio_fast=0)),  # the proof-of-work shape (Gen.grind): out[0] only
}
BENCH_VARIANTS = {
    "node_p2": (16, dict(prio_rot=2, prio_add3=2, prio_fast=0)),
    "leaf_p2": (4, dict(prio_rot=2, prio_add3=2, prio_fast=0)),
    "node_p31": (16, dict(prio_rot=3, prio_add3=1, prio_fast=0)),
    "leaf_p31": (4, dict(prio_rot=3, prio_add3=1, prio_fast=0)),
    "node_nop": (16, dict(prio_rot=None, prio_add3=None, prio_fast=None)),
    "leaf_nop": (4, dict(prio_rot=None, prio_add3=None, prio_fast=None)),
    "node_pk": (16, dict(prio_rot=3, prio_add3=2, prio_fast=0, rot16="pk")),
    "leaf_pk": (4, dict(prio_rot=3, prio_add3=2, prio_fast=0, rot16="pk")),
    "leaf_split": (4, dict(prio_rot=3, prio_add3=2, prio_fast=0, split_leaf_adds=True)),
    "node_cls": (16, dict(prio_rot=3, prio_add3=2, prio_fast=0, prio_mode="class")),
    "leaf_cls": (4, dict(prio_rot=3, prio_add3=2, prio_fast=0, prio_mode="class")),
}


def build(name, variants):
    n_msg, opt = variants[name]
    g = Gen(n_msg, **opt)
    return n_msg, (g.grind() if name.startswith("grind") else g.compression()), g.count


def emit_function(name, n_msg, lines, count, out):
    out.append("// %s: %d message registers; %d slow-class + %d fast-class VALU instructions, %d s_setprio"
               % (name, n_msg, count["slow"], count["fast"], count["s"]))
    out.append("__device__ __forceinline__ void b2_asm_%s(const uint32_t (&m)[16], uint32_t (&out)[8]) {" % name)
    out.append("    uint32_t " + ", ".join("v%d" % i for i in range(16)) + ";")
    out.append("    asm volatile(")
    for ln in lines:
        out.append('        "%s\\n\\t"' % ln)
    outs = ", ".join('"=&v"(v%d)' % i for i in range(16))
    ins = ", ".join('"v"(m[%d])' % j for j in range(n_msg))
    out.append("        : " + outs)
    out.append("        : " + ins + ");")
    for i in range(8):
        out.append("    out[%d] = v%d;" % (i, i))
    out.append("}")
    out.append("")


def emit_grind_function(name, lines, count, out):
    out.append("// %s: F(h, [m0, m1, 0 x 14], 0, 0, 0, 0) word 0 only; %d slow-class + %d fast-class VALU instructions, %d s_setprio.  pre[14] = the state words"
               % (name, count["slow"], count["fast"], count["s"]))
    out.append("// %s in this order: h0, h4, then the three quadruples of round 0's column step that do not see the nonce (b2_grind_prepare)"
               % ", ".join("v%d" % i for i in Gen.GRIND_IN))
    out.append("__device__ __forceinline__ uint32_t b2_asm_%s(uint32_t m0, uint32_t m1, const uint32_t (&pre)[14]) {" % name)
    out.append("    uint32_t " + ", ".join("v%d" % i for i in range(16)) + ";")
    out.append("    asm volatile(")
    for ln in lines:
        out.append('        "%s\\n\\t"' % ln)
    outs = ", ".join('"=&v"(v%d)' % i for i in range(16))
    ins = '"v"(m0), "v"(m1), ' + ", ".join('"v"(pre[%d])' % j for j in range(14))
    out.append("        : " + outs)
    out.append("        : " + ins + ");")
    out.append("    return v0;")
    out.append("}")
    out.append("")


def selfcheck(variants):
    import random

    rnd = random.Random(7)
    for name in variants:
        n_msg, lines, _ = build(name, variants)
        if name.startswith("grind"):
            for _ in range(5):
                h = [rnd.getrandbits(32) for _ in range(8)]
                m = [rnd.getrandbits(32), rnd.getrandbits(32)] + [0] * 14
                st = reference_compress(m, h, only_column_quads=[1, 2, 3])
                inputs = m[:2] + [st[i] for i in Gen.GRIND_IN]
                assert interpret(lines, inputs, 1)[0] == reference_compress(m, h)[0], name
            continue
        for _ in range(3):
            m = [rnd.getrandbits(32) if j < n_msg else 0 for j in range(16)]
            assert interpret(lines, m) == reference_compress(m), name


def main():
    bench = "--bench" in sys.argv
    variants = dict(VARIANTS)
    if bench:
        variants.update(BENCH_VARIANTS)
    selfcheck(variants)
    out = []
    out.append("// GENERATED by tools/gen_blake2s_asm.py%s — do not edit; regenerate instead." % (" --bench" if bench else ""))
    out.append("// Blake2s F(0, m, 0, 0) (Merkle shape, SURVEY.md A.3) as one hand-scheduled gfx950 asm block per message shape: the run")
    out.append("// structure and the s_setprio switches of blake2s.h's throughput form with no statement boundaries inside (see the generator).")
    out.append("#pragma once")
    out.append("#include <stdint.h>")
    out.append("")
    out.append("namespace frieda {")
    out.append("")
    for name in variants:
        n_msg, lines, count = build(name, variants)
        if name.startswith("grind"):
            emit_grind_function(name, lines, count, out)
        else:
            emit_function(name, n_msg, lines, count, out)
    out.append("}  // namespace frieda")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
