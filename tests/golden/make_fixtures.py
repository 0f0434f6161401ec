"""Regenerates tests/golden/ from a mounted reference checkout (run once, here; the GPU box
only sees the committed result).  Everything copied is DATA the reference's own tests hold:

  lp/test*.txt, lp/sol*.txt          <- tests/            (LP inputs and the CLI goldens)
  netlib_sol_max_cost/*.txt          <- datasets/netlib_simplex_sol/
  netlib_sol_max_reduction/*.txt     <- datasets/netlib_simplex_sol_max_reduction/
  netlib_sol_revised/*.txt           <- datasets/netlib_revised_simplex_sol/
  netlib_sol_revised_extra/*.txt     <- datasets/netlib_revised_simplex_sol/ (agg, agg2, agg3: files the reference
                                        ships but its ctest skips, "takes too long", CMakeLists.txt:138)
  netlib_mps/<name>.mps              <- datasets/netlib/<name> decoded with oracle/_ref/emps
                                        (the reference's tools/emps.c compiled by `make -C oracle ref`)
  netlib_compressed/<name>           <- datasets/netlib/<name> as shipped (netlib's compressed form; the input of
                                        the product's own expander, whose output must equal netlib_mps/<name>.mps)

usage: python tests/golden/make_fixtures.py [--out DIR] [--reference /root/reference]
With --out pointing elsewhere the committed fixtures can be checked with `diff -r`."""
import argparse
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

LP_INPUTS = [0, 1, 3, 4, 5, 8, 9, 12, 13, 14, 15, 16, 17, 20, 21]
LP_GOLDENS = ["sol12.txt", "sol13.txt", "sol14.txt"] + [
    "sol%d_%s.txt" % (c, s) for c in (15, 16, 17) for s in ("simplex_tableau", "simplex_tableau_max_reduction", "revised_simplex_tableau")]
NETLIB = {"netlib_sol_max_cost": ("netlib_simplex_sol", ["afiro", "adlittle", "kb2", "israel", "agg", "agg2", "agg3", "fit1d"]),
          "netlib_sol_max_reduction": ("netlib_simplex_sol_max_reduction", ["afiro", "adlittle", "kb2", "israel", "agg", "agg2", "agg3"]),
          "netlib_sol_revised": ("netlib_revised_simplex_sol", ["afiro", "adlittle", "kb2", "israel"]),
          "netlib_sol_revised_extra": ("netlib_revised_simplex_sol", ["agg", "agg2", "agg3"])}
MPS = ["afiro", "adlittle", "kb2", "israel", "agg", "agg2", "agg3", "fit1d"]
COMPRESSED = MPS + ["25fv47"]   # 25fv47 has no solution golden upstream; it is the largest expansion case


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--reference", default="/root/reference")
    a = ap.parse_args()
    ref = a.reference
    emps = os.path.join(ROOT, "oracle", "_ref", "emps")
    if not os.path.exists(emps):
        raise SystemExit("oracle/_ref/emps missing: run `make -C oracle ref` with the reference mounted")
    os.makedirs(os.path.join(a.out, "lp"), exist_ok=True)
    for c in LP_INPUTS:
        src = os.path.join(ref, "tests", "test%d.txt" % c)
        if os.path.exists(src):
            shutil.copyfile(src, os.path.join(a.out, "lp", "test%d.txt" % c))
    for g in LP_GOLDENS:
        shutil.copyfile(os.path.join(ref, "tests", g), os.path.join(a.out, "lp", g))
    for dst, (src_dir, names) in NETLIB.items():
        os.makedirs(os.path.join(a.out, dst), exist_ok=True)
        for n in names:
            shutil.copyfile(os.path.join(ref, "datasets", src_dir, n + ".txt"), os.path.join(a.out, dst, n + ".txt"))
    os.makedirs(os.path.join(a.out, "netlib_mps"), exist_ok=True)
    os.makedirs(os.path.join(a.out, "netlib_compressed"), exist_ok=True)
    for n in COMPRESSED:
        shutil.copyfile(os.path.join(ref, "datasets", "netlib", n), os.path.join(a.out, "netlib_compressed", n))
        with open(os.path.join(ref, "datasets", "netlib", n), "rb") as fin, open(os.path.join(a.out, "netlib_mps", n + ".mps"), "wb") as fout:
            subprocess.run([emps], stdin=fin, stdout=fout, check=True)


if __name__ == "__main__":
    main()
