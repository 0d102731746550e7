#!/usr/bin/env python
"""Tables and constants of the beam search's log-sum-exp (csrc/beam_kernels.hip: BEAM_TAB, LN2_32_*),
from 60-digit decimals: 2^(j/32), 1 / (1 + i/64), log(1 + i/64).  Writes a C header for
tools/beam_lse_check.c:

    python tools/beam_lse_tables.py > /tmp/beam_lse_tables.h
    gcc -O2 -ffp-contract=off -I/tmp -o /tmp/beam_lse_check tools/beam_lse_check.c -lm && /tmp/beam_lse_check
"""
import struct
from decimal import Decimal, getcontext

getcontext().prec = 60


def trunc(x, bits):
    u = struct.unpack("<Q", struct.pack("<d", x))[0] & ~((1 << bits) - 1)
    return struct.unpack("<d", struct.pack("<Q", u))[0]


def arr(name, a):
    out = "static const double %s[%d] = {\n" % (name, len(a))
    for k in range(0, len(a), 4):
        out += "    " + ", ".join(float.hex(v) for v in a[k:k + 4]) + ",\n"
    return out + "};\n"


def main():
    exp2 = [float(Decimal(2) ** (Decimal(j) / Decimal(32))) for j in range(32)]
    invc = [float(Decimal(1) / (Decimal(1) + Decimal(i) / Decimal(64))) for i in range(65)]
    logc = [float((Decimal(1) + Decimal(i) / Decimal(64)).ln()) for i in range(65)]
    ln2_32 = Decimal(2).ln() / 32
    hi = trunc(float(ln2_32), 12)           # n * hi is exact for |n| < 2^12
    lo = float(ln2_32 - Decimal(hi))
    inv = float(Decimal(32) / Decimal(2).ln())
    print(arr("EXP2_32", exp2) + arr("INVC", invc) + arr("LOGC", logc)
          + "static const double LN2_32_HI = %s, LN2_32_LO = %s, INV_LN2_32 = %s;"
          % (float.hex(hi), float.hex(lo), float.hex(inv)))


if __name__ == "__main__":
    main()
