"""Golden vectors for the synth CLI's bap decode (test/synth.cpp:231-246 -> test/sptkfunctions.cpp mgc2sp),
produced by the CLI's SPTK port compiled as it is (make -C oracle ref -> _ref/libsptk_ref.so).

Inputs are the reference's own coded bap rows already stored in tests/golden/codec_*.npz (every 8th frame);
outputs are mgc2sp's log-spectrum x for the bins the CLI keeps, and one full freqt result per fixture.
Run in the container that has /root/reference:  python oracle/gen_golden_sptk.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.bindings import SptkReference  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    ref = SptkReference()
    out = {}
    for name in ("codec_16k", "codec_48k"):
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        F = int(g["fft_size"])
        bap = g["bap"].astype(np.float64)[::8]
        order = bap.shape[1] - 1 if bap.shape[1] % 2 else bap.shape[1]        # synth.cpp:233-235
        rows = bap[:, :order + 1].copy()
        rows[:, 0] += 9.210340                                                # synth.cpp:241
        x = np.stack([ref.mgc2sp(r, 0.55, F)[:order] for r in rows])
        out[name + "_rows"] = rows
        out[name + "_x"] = x
        out[name + "_freqt"] = ref.freqt(rows[len(rows) // 2], F // 2, -0.55)
        out[name + "_fft_size"] = F
        print(name, "rows", rows.shape, "order", order, "x range", x.min(), x.max())
    np.savez_compressed(os.path.join(GOLDEN, "sptk_mgc2sp.npz"), **out)


if __name__ == "__main__":
    main()
