#!/usr/bin/env python3
"""Generates poseidon_sbox_asm.inc: the Goldilocks S-box x -> x^7 (glwasm.js:695-757) as gfx950 assembly text.

Why assembly: hipcc builds a Goldilocks product from 22 vector instructions (it re-derives every carry with a 64-bit
compare and pays a move for every zero-extended addend); the sequence below needs 13:

    t = a0*b0                      v_mad_u64_u32
    u = a0*b1 + (t >> 32)          v_mov (the addend pair's high word stays zero) + v_mad_u64_u32
    v = a1*b0 + u                  v_mad_u64_u32 with the WHOLE u as addend; the 65th bit comes out as the carry
    w = a1*b1 + (v >> 32) + cy<<32 v_mov + v_cndmask (carry -> high word of the addend pair) + v_mad_u64_u32
    lo = (t0, v0)                  v_mov
    z = w0*(2^32-1) + lo           v_mad_u64_u32, carry c        (2^64 = 2^32-1 mod p)
    z += c*(2^32-1)                v_cndmask + v_mad_u64_u32     (cannot wrap)
    r = z - w1                     v_sub_co + v_subbrev_co       (2^96 = -1 mod p); a borrow (probability ~2^-32) is
                                   OR-ed into the `bad` lane mask and the caller redoes that S-box the slow way
gfx950 needs two wait states between a vector instruction that writes vcc / an SGPR and a vector instruction that
reads it (hipcc pads its own carry chains with `s_nop 1`); inside an asm statement nobody pads, so the text carries them.

Temporaries whose halves are addressed (the multiply-add's register pairs) are fixed physical registers, listed as
clobbers; everything else is an operand.
"""
import sys

BASE = int(sys.argv[1]) if len(sys.argv) > 1 else 96          # first scratch VGPR (even)


class Chain:
    def __init__(self, base, carry, ops):
        b = base
        self.P, self.Q, self.R, self.E, self.F = [(b + 2 * i, b + 2 * i + 1) for i in range(5)]
        self.cy = carry                 # "vcc" or an operand like "%7" printing as s[n:n+1]
        self.vop3 = carry != "vcc"
        self.ops = ops                  # dict: x0,x1,y0,y1,a0,a1 (x2), b0,b1 (x3), bad

    @staticmethod
    def pair(p):
        return "v[%d:%d]" % p

    def mul(self, A0, A1, B0, B1, D0, D1):
        P, Q, R, E, F, cy = self.P, self.Q, self.R, self.E, self.F, self.cy
        pr, v = self.pair, lambda n: "v%d" % n
        e64 = "_e64" if self.vop3 else ""
        return [
            "v_mad_u64_u32 %s, %s, %s, %s, 0" % (pr(P), cy, A0, B0),
            "v_mov_b32 %s, %s" % (v(E[0]), v(P[1])),
            "v_mad_u64_u32 %s, %s, %s, %s, %s" % (pr(Q), cy, A0, B1, pr(E)),
            "v_mad_u64_u32 %s, %s, %s, %s, %s" % (pr(R), cy, A1, B0, pr(Q)),          # W cy
            "v_mov_b32 %s, %s" % (v(F[0]), v(R[1])),
            "v_mov_b32 %s, %s" % (v(P[1]), v(R[0])),
            "v_cndmask_b32_e64 %s, 0, 1, %s" % (v(F[1]), cy),                          # R cy
            "v_mad_u64_u32 %s, %s, %s, %s, %s" % (pr(Q), cy, A1, B1, pr(F)),
            "v_mad_u64_u32 %s, %s, %s, -1, %s" % (pr(R), cy, v(Q[0]), pr(P)),           # W cy
            "NOP2",
            "v_cndmask_b32_e64 %s, 0, 1, %s" % (v(E[0]), cy),                          # R cy
            "v_mad_u64_u32 %s, %s, %s, -1, %s" % (pr(R), cy, v(E[0]), pr(R)),
            "v_sub_co_u32%s %s, %s, %s, %s" % (e64, D0, cy, v(R[0]), v(Q[1])),          # W cy
            "NOP2",
            "v_subbrev_co_u32%s %s, %s, 0, %s, %s" % (e64, D1, cy, v(R[1]), cy),        # R/W cy
            "s_or_b64 %s, %s, %s" % (self.ops["bad"], self.ops["bad"], cy),
        ]

    def sbox(self):
        o = self.ops
        ins = ["v_mov_b32 v%d, 0" % self.E[1]]
        ins += self.mul(o["x0"], o["x1"], o["x0"], o["x1"], o["a0"], o["a1"])          # x2
        ins += self.mul(o["a0"], o["a1"], o["x0"], o["x1"], o["b0"], o["b1"])          # x3
        ins += self.mul(o["a0"], o["a1"], o["a0"], o["a1"], o["a0"], o["a1"])          # x4 (in place of x2)
        ins += self.mul(o["b0"], o["b1"], o["a0"], o["a1"], o["y0"], o["y1"])          # x7
        return ins

    def clobbers(self):
        return ["v%d" % r for p in (self.P, self.Q, self.R, self.E, self.F) for r in p]


def emit_single():
    ops = {"y0": "%0", "y1": "%1", "a0": "%2", "a1": "%3", "b0": "%4", "b1": "%5", "bad": "%6", "x0": "%7", "x1": "%8"}
    c = Chain(BASE, "vcc", ops)
    text = [("s_nop 1" if i == "NOP2" else i) for i in c.sbox()]
    return text, c.clobbers() + ["vcc"]


def emit_triple():
    """three independent S-boxes interleaved instruction by instruction, each with its own scratch and its own carry SGPR
    pair: between a carry's writer and its reader there are always two instructions of the other chains, so no s_nop"""
    chains = []
    # operands: per chain k: y0,y1 = %(2k), %(2k+1); a0,a1,b0,b1 = %(6+4k..); cy_k = %(18+k); bad = %21; x = %(22+2k), %(23+2k)
    for k in range(3):
        ops = {"y0": "%%%d" % (2 * k), "y1": "%%%d" % (2 * k + 1), "a0": "%%%d" % (6 + 4 * k), "a1": "%%%d" % (7 + 4 * k),
               "b0": "%%%d" % (8 + 4 * k), "b1": "%%%d" % (9 + 4 * k), "bad": "%21", "x0": "%%%d" % (22 + 2 * k), "x1": "%%%d" % (23 + 2 * k)}
        chains.append(Chain(BASE + 10 * k, "%%%d" % (18 + k), ops))
    lists = [c.sbox() for c in chains]
    text = []
    for row in zip(*lists):
        if row[0] == "NOP2":
            continue
        text += list(row)
    clob = [r for c in chains for r in c.clobbers()]
    return text, clob


def cstr(lines):
    return " \\\n    ".join('"%s\\n\\t"' % l for l in lines)


def main():
    s_text, s_clob = emit_single()
    t_text, t_clob = emit_triple()
    out = ["// generated by gen_sbox_asm.py %d -- do not edit (the generator documents the sequence)" % BASE,
           "#define GL_SBOX1_TEXT \\\n    " + cstr(s_text),
           "#define GL_SBOX1_CLOBBERS " + ", ".join('"%s"' % c for c in s_clob),
           "#define GL_SBOX3_TEXT \\\n    " + cstr(t_text),
           "#define GL_SBOX3_CLOBBERS " + ", ".join('"%s"' % c for c in t_clob), ""]
    sys.stdout.write("\n".join(out))


if __name__ == "__main__":
    main()
