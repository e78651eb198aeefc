"""Golden vectors for the recipe's `cmp` stage, produced by RUNNING the
reference's own Perl scripts here: data/scripts/window.pl on seeded float32 streams with the window
files of data/win/, and data/scripts/addhtkheader.pl on the composed file.  Only inputs' seeds and the
scripts' outputs are stored (tests/golden/cmp_windows.npz); nothing of the scripts themselves.

    python oracle/gen_golden_cmp.py        (needs /root/reference and perl)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("WORLD_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def stream(seed, T, dim, holes=False):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((T, dim)).astype(np.float32)
    if holes:                                  # unvoiced stretches carry the ignore value
        a[:3] = -1.0e10
        a[10:17] = -1.0e10
        a[T - 2:] = -1.0e10
        a[25, :] = -1.0e10
    return a


def main():
    scripts = os.path.join(REF, "data", "scripts")
    win = os.path.join(REF, "data", "win")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        cases = {"mgc": (101, 37, 5, False), "lf0": (102, 37, 1, True), "bap": (103, 37, 3, False)}
        blocks = []
        for name, (seed, T, dim, holes) in cases.items():
            a = stream(seed, T, dim, holes)
            path = os.path.join(tmp, name + ".in")
            a.tofile(path)
            wins = [os.path.join(win, "%s.win%d" % (name, i)) for i in (1, 2, 3)]
            res = subprocess.run(["perl", os.path.join(scripts, "window.pl"), str(dim), path] + wins,
                                 capture_output=True, check=True).stdout
            w = np.frombuffer(res, dtype=np.float32).reshape(T, 3 * dim)
            out[name + "_seed"] = seed
            out[name + "_shape"] = np.array([T, dim])
            out[name + "_holes"] = holes
            out[name + "_windowed"] = w
            blocks.append(w)
        cmp_ = np.concatenate(blocks, axis=1)                 # merge +f -s 0: [mgc | lf0 | bap]
        path = os.path.join(tmp, "tmp.cmp")
        cmp_.tofile(path)
        byte = 4 * cmp_.shape[1]
        res = subprocess.run(["perl", os.path.join(scripts, "addhtkheader.pl"), "16000", "80", str(byte), "9", path],
                             capture_output=True, check=True).stdout
        out["htk_header"] = np.frombuffer(res[:12], dtype=np.uint8)
        assert res[12:] == cmp_.tobytes()
        out["htk_args"] = np.array([16000, 80, byte, 9])
    np.savez_compressed(os.path.join(OUT, "cmp_windows.npz"), **out)
    print("cmp_windows ok", cmp_.shape, bytes(out["htk_header"]).hex())


if __name__ == "__main__":
    sys.exit(main())
