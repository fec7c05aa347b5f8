#!/usr/bin/env python3
"""Writes include/avp_glibc_tab.h: the lookup tables of glibc 2.35's fp64 atan2 / asin / acos / tan / pow.

Why a table dump and not a computation: the reference's Reeds-Shepp words call CPython math.atan2 / asin / acos /
tan and float ** 2 (path_plan/rs_curve.py:176-664), i.e. the platform libm, and exact ties between mirror-image
words are decided by the last bit of those results. glibc's kernels evaluate polynomials around tabulated break
points that are NOT round numbers (they were picked so that atan(x_i), asin(x_i), ... are unusually close to a
double), so the tables cannot be regenerated from first principles the way the sin/cos table was
(csrc/gen_sincos_table.py). They are data of the GNU C Library (LGPL-2.1-or-later; sysdeps/ieee754/dbl-64/
uatan.tbl, asincos.tbl, root.tbl, powtwo.tbl, utan.tbl, branred.h, e_pow_log_data.c, e_exp_data.c) and are read here from
the distribution's own static archive /usr/lib/x86_64-linux-gnu/libm-2.35.a with binutils; the generated header
carries that notice. include/avp_glibc_libm.h holds the arithmetic that uses them.

Usage: python3 include/gen_glibc_tab.py [path/to/libm-2.35.a]     (needs ar, nm, objcopy)
"""
import os
import struct
import subprocess
import sys
import tempfile

ARCHIVE = sys.argv[1] if len(sys.argv) > 1 else "/usr/lib/x86_64-linux-gnu/libm-2.35.a"
HERE = os.path.dirname(os.path.abspath(__file__))


def section(obj, name, tmp):
    out = os.path.join(tmp, "sec.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=" + name, obj, out])
    with open(out, "rb") as f:
        return f.read()


def symbols(obj):
    res = {}
    for line in subprocess.check_output(["nm", obj], text=True).splitlines():
        p = line.split()
        if len(p) == 3:
            res[p[2]] = int(p[0], 16)
    return res


def doubles(blob, off, n):
    return [struct.unpack_from("<Q", blob, off + 8 * i)[0] for i in range(n)]


def emit(f, ctype, name, dims, words, per_line=4, as_double=True):
    f.write("AVP_GLIBC_TAB %s %s%s = {\n" % (ctype, name, dims))
    for i in range(0, len(words), per_line):
        row = words[i:i + per_line]
        if as_double:
            f.write("  " + " ".join("AVP_D(0x%016x)," % w for w in row) + "\n")
        else:
            f.write("  " + " ".join("0x%016xull," % w for w in row) + "\n")
    f.write("};\n")


def main():
    with tempfile.TemporaryDirectory() as tmp:
        members = ["e_atan2-fma.o", "e_asin-fma.o", "s_tan-fma.o", "e_pow_log_data.o", "e_exp_data.o", "branred.o"]
        subprocess.check_call(["ar", "x", ARCHIVE] + members, cwd=tmp)
        p = lambda m: os.path.join(tmp, m)
        atan2_ro = section(p("e_atan2-fma.o"), ".rodata", tmp)
        assert symbols(p("e_atan2-fma.o"))["cij"] == 0 and len(atan2_ro) == 241 * 7 * 8
        cij = doubles(atan2_ro, 0, 241 * 7)
        asin_ro = section(p("e_asin-fma.o"), ".rodata", tmp)
        s = symbols(p("e_asin-fma.o"))
        assert (s["powtwo"], s["inroot"], s["asncs"]) == (0, 0xe0, 0x4e0) and len(asin_ro) == 0x5520
        powtwo = doubles(asin_ro, 0, 28)
        inroot = doubles(asin_ro, 0xe0, 128)
        asncs = doubles(asin_ro, 0x4e0, (0x5520 - 0x4e0) // 8)
        tan_ro = section(p("s_tan-fma.o"), ".rodata", tmp)
        assert len(tan_ro) == 186 * 4 * 8
        xfg = doubles(tan_ro, 0, 186 * 4)
        powlog = section(p("e_pow_log_data.o"), ".rodata", tmp)
        assert len(powlog) >= 72 + 128 * 32
        powlog_head = doubles(powlog, 0, 9)            # ln2hi, ln2lo, poly[7]
        powlog_tab = doubles(powlog, 72, 128 * 4)      # {invc, pad, logc, logctail}
        expd = section(p("e_exp_data.o"), ".rodata", tmp)
        assert len(expd) >= 112 + 256 * 8
        exp_head = doubles(expd, 0, 14)                # invln2N, shift, negln2hiN, negln2loN, poly[4], exp2_shift, exp2_poly[5]
        exp_tab = doubles(expd, 112, 256)
        br_ro = section(p("branred.o"), ".rodata", tmp)
        assert symbols(p("branred.o"))["toverp"] == 0 and len(br_ro) == 75 * 8
        toverp = doubles(br_ro, 0, 75)                 # 2/pi in 24-bit pieces (branred.h)
    # sanity: the values the arithmetic relies on
    as_f = lambda w: struct.unpack("<d", struct.pack("<Q", w))[0]
    assert abs(as_f(cij[0]) - 1.0 / 16) < 1e-3 and abs(as_f(cij[240 * 7]) - 1.0) < 5e-3
    # the scalar constants of __pow_log_data / __exp_data are literals in avp_glibc_libm.h (avpg_pow2): checked here
    want_log = ["0x1.62e42fefa3800p-1", "0x1.ef35793c76730p-45", "-0x1p-1", "-0x1.555555555556p-1", "0x1.0000000000006p-1",
                "0x1.999999959554ep-1", "-0x1.555555529a47ap-1", "-0x1.2495b9b4845e9p+0", "0x1.0002b8b263fc3p+0"]
    want_exp = ["0x1.71547652b82fep+7", "0x1.8p+52", "-0x1.62e42fefa0000p-8", "-0x1.cf79abc9e3b3ap-47", "0x1.ffffffffffdbdp-2",
                "0x1.555555555543cp-3", "0x1.55555cf172b91p-5", "0x1.1111167a4d017p-7"]
    assert [as_f(w) for w in powlog_head] == [float.fromhex(h) for h in want_log]
    assert [as_f(w) for w in exp_head[:8]] == [float.fromhex(h) for h in want_exp]
    out = os.path.join(HERE, "avp_glibc_tab.h")
    with open(out, "w") as f:
        f.write("/* GENERATED by include/gen_glibc_tab.py from %s -- do not edit.\n" % os.path.basename(ARCHIVE))
        f.write(" *\n * Lookup tables of the GNU C Library 2.35 fp64 atan2 / asin / acos / tan / pow kernels\n")
        f.write(" * (sysdeps/ieee754/dbl-64: uatan.tbl, asincos.tbl, root.tbl, powtwo.tbl, utan.tbl, branred.h, e_pow_log_data.c,\n")
        f.write(" * e_exp_data.c). Copyright (C) Free Software Foundation, Inc.; the GNU C Library is free software,\n")
        f.write(" * distributed under the GNU Lesser General Public License, version 2.1 or (at your option) any later\n")
        f.write(" * version; these tables are data of that library, reproduced bit for bit. See avp_glibc_libm.h. */\n")
        f.write("#ifndef AVP_GLIBC_TAB_H\n#define AVP_GLIBC_TAB_H\n")
        emit(f, "uint64_t", "AVP_G_CIJ", "[241 * 7]", cij, 7, False)
        emit(f, "uint64_t", "AVP_G_POWTWO", "[28]", powtwo, 4, False)
        emit(f, "uint64_t", "AVP_G_INROOT", "[128]", inroot, 4, False)
        emit(f, "uint64_t", "AVP_G_ASNCS", "[%d]" % len(asncs), asncs, 4, False)
        emit(f, "uint64_t", "AVP_G_XFG", "[186 * 4]", xfg, 4, False)
        emit(f, "uint64_t", "AVP_G_POWLOG_TAB", "[128 * 4]", powlog_tab, 4, False)
        emit(f, "uint64_t", "AVP_G_EXP_TAB", "[256]", exp_tab, 4, False)
        emit(f, "uint64_t", "AVP_G_TOVERP", "[75]", toverp, 5, False)
        f.write("#endif\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
