#!/usr/bin/env python3
"""Fixtures for the branches of the netlib expander that none of the nine files the reference ships reaches: a
RANGES section, every kind of bound, numbers in exponent notation (both signs of the exponent, the two shortened
forms), mantissas of more than nine digits, the table of repeated numbers, "mystery" lines and a checksum line in the
middle of a section.

This script WRITES the compressed text itself (a small encoder of netlib's compressed form, written from the format
as reference tools/emps.c reads it: it only has to produce input that tool accepts) and then runs the reference's own
expander, compiled from where it lies into oracle/_ref/emps (oracle/Makefile), on it.  What the tool prints is the
fixture: tests/golden/netlib_extra/<name>.mps beside the input tests/golden/netlib_extra/<name>.  The product's
expander (csrc/host/netlib_text.cc) is then compared with the tool byte for byte (tests/test_host_layer.py).

  python tests/golden/make_netlib_extra.py      (needs oracle/_ref/emps, i.e. the reference checkout mounted)
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EMPS = os.path.join(ROOT, "oracle", "_ref", "emps")
OUT = os.path.join(HERE, "netlib_extra")

SYMBOLS = "!\"#$%&'()*+,-./0123456789;<=>?@" + "ABCDEFGHIJKLMNOPQRSTUVWXYZ[]^_`abcdefghijklmnopqrstuvwxyz{|}~"
assert len(SYMBOLS) == 92
VALUE = {c: i for i, c in enumerate(SYMBOLS)}


def check_symbol(line):
    x = 0
    for c in line:
        x = (x >> 1) + VALUE.get(c, 92) + (16384 if x & 1 else 0)
    return SYMBOLS[x % 92]


def index(x):
    """0..22 in one symbol; otherwise base 46, first digit below 23, last digit offset by 46."""
    if 0 <= x <= 22:
        return SYMBOLS[23 + x]
    digits = []
    while x:
        digits.append(x % 46)
        x //= 46
    digits.reverse()
    if digits[0] >= 23 or len(digits) == 1:
        digits.insert(0, 0)
    digits[-1] += 46
    return "".join(SYMBOLS[d] for d in digits)


def integer(x):
    """An integer value (printed as `x.`)."""
    neg, x = x < 0, abs(x)
    lead = 46 + (23 if neg else 0) + 11
    if x <= 5:
        return SYMBOLS[lead + 6 + x]
    digits = []
    while x:
        digits.append(x % 46)
        x //= 46
    digits.reverse()
    if digits[0] >= 6 or len(digits) == 1:
        digits.insert(0, 0)
    digits[-1] += 46
    return SYMBOLS[lead + digits[0]] + "".join(SYMBOLS[d] for d in digits[1:])


def base92(x, width=None):
    digits = []
    while x:
        digits.append(x % 92)
        x //= 92
    digits = digits[::-1] or [0]
    if width:
        digits = [0] * (width - len(digits)) + digits
    return digits


def mantissa(digits, ex, neg=False):
    """The decimal digits `digits` (a string without leading zeros) times 10**ex."""
    if len(digits) <= 8:
        body = base92(int(digits))
    else:   # two words: nine digits (>= 1e8 once all five symbols are in), then the rest behind a leading 1
        first, rest = digits[:9], digits[9:]
        assert first[0] != "0" and int(first) >= 10 ** 8
        body = base92(int(first), 5) + base92(int("1" + rest))
    k = len(body) - 1
    assert 0 <= k <= 10
    return SYMBOLS[46 + (23 if neg else 0) + k] + SYMBOLS[ex + 50] + "".join(SYMBOLS[d] for d in body)


class Writer:
    """Lines of a compressed problem with the check symbols and the checksum line every 71 lines."""

    def __init__(self):
        self.lines, self.check = [], ""

    def raw(self, line):
        assert len(line) <= 75, line
        self.lines.append(line)

    def data(self, line):
        self.raw(line)
        self.check += check_symbol(line)
        if len(self.check) >= 71:
            self.flush()

    def flush(self):
        if self.check:
            self.raw(" " + self.check)
            self.check = ""


def packed(w, fields, width=70):
    """Fields back to back, a new line whenever the next field would not fit."""
    line = ""
    for f in fields:
        if len(line) + len(f) > width:
            w.data(line)
            line = ""
        line += f
    if line:
        w.data(line)


def problem(name, rows, columns, rhs, ranges, bounds, table):
    """rows: [(type, name)]; columns: [(name, [(row index from 1, number field)])]; rhs / ranges: [(set name, [(row, field)])];
    bounds: [(set name, [(kind 1..6, column index from 1, field or None)])]; table: number fields referred to as index(k)."""
    w = Writer()
    nz = sum(len(c[1]) for c in columns)
    rhsnz = sum(len(s[1]) for s in rhs)
    ranz = sum(len(s[1]) for s in ranges)
    bdnz = sum(len(s[1]) for s in bounds)
    w.raw("NAME          %s" % name)
    w.raw("%d %d %d %d %d %d %d %d" % (len(rows), len(columns), max(len(c[1]) for c in columns), nz, len(rhs), rhsnz, len(ranges), ranz))
    w.raw("%d %d %d" % (len(bounds), bdnz, len(table)))
    packed(w, table)
    for t, n in rows:
        w.data(t + n)
    w.data(":* a line the expander passes through, in front of the columns")

    def section(sets, is_bounds=False):
        for set_name, entries in sets:
            w.data(index(0) + set_name)
            fields = []
            for e in entries:
                if is_bounds:
                    kind, col, field = e
                    fields.append(index(kind) + index(col) + (field or ""))
                else:
                    row, field = e
                    fields.append(index(row) + field)
            packed(w, fields)

    section(columns)
    section(rhs)
    section(ranges)
    section(bounds, True)
    w.flush()
    return "\n".join(w.lines) + "\n"


def build():
    n_rows, n_cols = 40, 30
    rows = [("N", "COST")] + [("LEG"[i % 3], "R%05d" % i) for i in range(1, n_rows)]
    table = [mantissa("25", -1), integer(-7), mantissa("123456789012", -6), mantissa("5", -12)]
    numbers = [
        integer(0), integer(3), integer(5), integer(6), integer(45), integer(46), integer(277), integer(12345678), integer(-1), integer(-300),
        mantissa("15", -1), mantissa("375", -3, True), mantissa("1", -5), mantissa("12345", 10), mantissa("123456", 6), mantissa("999", 2),
        mantissa("7", 11), mantissa("5", -12), mantissa("25", -11, True), mantissa("25", -11), mantissa("123456789", -4), mantissa("1234567890123", -9),
        mantissa("4", -9), mantissa("31415926", -7), mantissa("271828182845", -11, True), mantissa("6", 3), mantissa("6", 2), mantissa("12345678", 4),
        index(1), index(2), index(3), index(4),
    ]
    columns = []
    q = 0
    for j in range(n_cols):
        entries = []
        for t in range(1 + j % 4):
            entries.append((1 + (j * 7 + t * 11) % n_rows, numbers[q % len(numbers)]))
            q += 1
        columns.append(("X%06d" % (j * 37), entries))
    rhs = [("RHS", [(2 + 3 * k, numbers[(5 * k) % len(numbers)]) for k in range(9)]), ("RHS2", [(5, mantissa("5", -1)), (9, integer(2)), (12, index(2))])]
    ranges = [("RNG", [(2 + 2 * k, numbers[(3 * k + 1) % len(numbers)]) for k in range(7)]), ("RNG2", [(4, integer(10))])]
    bounds = [("BND", [(1, 1, mantissa("4", 0)), (2, 2, integer(-4)), (3, 3, mantissa("15", -1)), (4, 4, None), (5, 5, None), (6, 6, None),
                       (1, 24, mantissa("1", 12)), (2, 30, mantissa("1", -13, True))]),
              ("BND2", [(1, 7, index(1))])]
    return problem("EXTRA1", rows, columns, rhs, ranges, bounds, table)


def main():
    os.makedirs(OUT, exist_ok=True)
    text = build()
    path = os.path.join(OUT, "extra1")
    open(path, "w").write(text)
    res = subprocess.run([EMPS, path], capture_output=True)
    if res.returncode != 0 or not res.stdout.endswith(b"ENDATA\n"):
        raise SystemExit("the reference tool rejected the input:\n%s\n%s" % (res.stdout[-400:].decode(), res.stderr.decode()))
    open(path + ".mps", "wb").write(res.stdout)
    print("wrote %s (%d lines) and %s.mps (%d lines)" % (path, text.count("\n"), path, res.stdout.count(b"\n")))


if __name__ == "__main__":
    main()
