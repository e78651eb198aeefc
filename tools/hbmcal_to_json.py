"""Calibration of FETCH_SIZE / WRITE_SIZE (tools/micro/hbmcal.hip) -> profiles/hbm_counter_calibration.json.

    python tools/hbmcal_to_json.py FETCH_DIR WRITE_DIR STDOUT_OF_HBMCAL [TAG]

FETCH_DIR / WRITE_DIR: output directories of the two `rocprofv3 --kernel-trace --pmc ...` passes over tools/micro/hbmcal;
STDOUT_OF_HBMCAL: the "cal <kernel> <bytes read> <bytes written>" lines the binary prints.  The factor of a pattern is
(bytes really moved) / (counter x 1024): what a counter reading has to be multiplied by for that access width.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ORDER = ["cal_read<double,8>", "cal_read<double,8>+8B", "cal_read<double2,4>", "cal_write<double,8>",
         "cal_write<double,8>+8B", "cal_write<double2,4>", "cal_frames"]


def dispatches(dirname, counter):
    """[(kernel name, KB)] in dispatch order."""
    rows = []
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == counter and "cal_" in row.get("Kernel_Name", ""):
                    rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"])))
    rows.sort()
    return [(n, v) for _, n, v in rows]


def main():
    fdir, wdir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    tag = sys.argv[4] if len(sys.argv) > 4 else "r03"
    actual = {}
    for ln in open(out):
        p = ln.split()
        if len(p) == 4 and p[0] == "cal":
            actual[p[1]] = (int(p[2]), int(p[3]))
    fetch, write = dispatches(fdir, "FETCH_SIZE"), dispatches(wdir, "WRITE_SIZE")
    n = len(ORDER)
    assert len(fetch) == 2 * n and len(write) == 2 * n, (len(fetch), len(write))
    recs = []
    for k, name in enumerate(ORDER):
        rd, wr = actual[name]
        f_kb = fetch[n + k][1]            # the second repetition
        w_kb = write[n + k][1]
        recs.append({"pattern": name, "bytes_read": rd, "bytes_written": wr, "FETCH_SIZE_kb": f_kb, "WRITE_SIZE_kb": w_kb,
                     "fetch_factor": round(rd / (f_kb * 1024.0), 4) if rd and f_kb else None,
                     "write_factor": round(wr / (w_kb * 1024.0), 4) if wr and w_kb else None})
    by = {r["pattern"]: r for r in recs}
    doc = {"tag": tag, "device": "MI355X (gfx950), ROCm 7.2, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes",
           "how": "tools/micro/hbmcal.hip streams 2 GiB once per kernel (8 B and 16 B per lane, aligned and skewed by 8 B) and "
                  "runs a frame-shaped kernel (overlapping 681-sample windows in, 513-double rows out); factor = bytes moved / "
                  "(counter x 1024)",
           "patterns": recs,
           "fetch_factor_8B_per_lane": by["cal_read<double,8>"]["fetch_factor"],
           "fetch_factor_16B_per_lane": by["cal_read<double2,4>"]["fetch_factor"],
           "write_factor_8B_per_lane": by["cal_write<double,8>"]["write_factor"],
           "write_factor_16B_per_lane": by["cal_write<double2,4>"]["write_factor"],
           "files": "profiles/%s_hbmcal_fetch.csv, profiles/%s_hbmcal_write.csv, profiles/%s_hbmcal_stdout.txt" % (tag, tag, tag)}
    path = os.path.join(ROOT, "profiles", "hbm_counter_calibration.json")
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
