"""Turn the two PMC passes of tools/profile_round.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, counter_collection CSVs)
into one record of profiles/pmc_traffic.json: HBM-side KB of the dominant kernel per pass, the frames of a pass, and a hash
of the kernel's sources (bench.py prints `roofline.traffic` only while that hash still matches).

    python tools/pmc_to_json.py FETCH_DIR WRITE_DIR FRAMES FS TAG [WORKLOAD]

WORKLOAD: analysis_synthesis (default; d4c_kernel), harvest (hv_band_fft_kernel), synthesis (synth_pulse_kernel).
The readings are stored as reported; bench.py applies the factors of profiles/hbm_counter_calibration.json.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_hash only; nothing touches the GPU)


def is_usual_d4c(name):
    """The usual instantiation of d4c_kernel (template argument RARE = false), demangled or mangled, or -- at fft 4096
    -- any of the three kernels it is split into (d4c_big.hpp); their traffic is added up per launch of the first."""
    if "d4cb_centroid_kernel" in name or "d4cb_spectrum_kernel" in name or "d4cb_band_kernel" in name:
        return True
    return ("d4c_kernel<" in name and ", false>" in name) or ("d4c_kernelILi" in name and "ELb0E" in name)


def counts_as_launch(name):
    return "d4cb_spectrum_kernel" not in name and "d4cb_band_kernel" not in name


MATCH = {"analysis_synthesis": is_usual_d4c,
         "harvest": lambda n: "hv_band_fft_kernel" in n,
         "synthesis": lambda n: "synth_pulse_kernel" in n}


# a kernel that runs exactly once per pass of the workload: its launches count the passes of a profiled run, so that a
# pass that launches the dominant kernel several times (Synthesis in pieces, D4C in chunks) is still one pass
PASS_MARK = {"analysis_synthesis": "stonemask_kernel", "synthesis": "synth_timebase_kernel",
             "harvest": "hv_contour_kernel"}


# launches of the mark per pass: Synthesis alone prepares a batch in two parts (synthesis.hip: launch_synthesis), each
# with its own time-base launch
MARKS_PER_PASS = {"synthesis": 2}


def total(dirname, counter, match=is_usual_d4c, mark=None):
    kb, launches, name, passes = 0.0, 0, None, 0
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                kn = row.get("Kernel_Name", "")
                if mark and mark in kn:
                    passes += 1
                if match(kn):
                    kb += float(row["Counter_Value"])
                    if counts_as_launch(kn):
                        launches += 1
                        name = kn
    return kb, launches, name, passes


def short_name(kn):
    """`void wm::hv_raw_kernel<...>(args)` -> `hv_raw_kernel`"""
    kn = kn.split("(")[0].split("<")[0].strip()
    return kn.split("::")[-1].split(" ")[-1]


def per_kernel(dirname, counter):
    """KB per kernel name over the whole profiled process (every wm:: kernel), and its launches."""
    out = {}
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                kn = row.get("Kernel_Name", "")
                if row.get("Counter_Name") != counter or "wm::" not in kn:
                    continue
                e = out.setdefault(short_name(kn), [0.0, 0])
                e[0] += float(row["Counter_Value"])
                e[1] += 1
    return out


def main():
    fdir, wdir, frames, fs, tag = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    workload = sys.argv[6] if len(sys.argv) > 6 else "analysis_synthesis"
    fkb, fl, name, fp = total(fdir, "FETCH_SIZE", MATCH[workload], PASS_MARK[workload])
    wkb, wl, _, wp = total(wdir, "WRITE_SIZE", MATCH[workload], PASS_MARK[workload])
    assert fl and fl == wl and fp and fp == wp, (fl, wl, fp, wp)
    per = MARKS_PER_PASS.get(workload, 1)
    assert fp % per == 0, (fp, per)
    fp //= per
    wp //= per
    kname = (name or "d4c_kernel").split("(")[0]
    if "d4cb_" in kname:
        kname = "d4cb_centroid_kernel + d4cb_spectrum_kernel + d4cb_band_kernel (the D4C scope at fft 4096)"
    # fetch_kb / write_kb: per PASS over `frames` frames (the sum of the kernel's launches of a pass)
    rec = {"workload": workload, "kernel": kname, "fs": fs, "frames": frames, "launches": fl, "passes": fp,
           "fetch_kb": fkb / fp, "write_kb": wkb / wp, "source_sha": bench.kernel_source_hash(workload), "tag": tag,
           "files": "profiles/%s_pmc_fetch.csv, profiles/%s_pmc_write.csv" % (tag, tag),
           "unit_note": "FETCH_SIZE / WRITE_SIZE as reported (KB); bench.py multiplies by the factors measured in "
                        "profiles/hbm_counter_calibration.json (reads x 2.0, writes x 1.0 on gfx950)"}
    # every kernel of the workload, per pass: what bench.py's per-kernel rooflines are made of (roofline.kernels)
    pf, pw = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    total_passes = fp * per
    rec["per_kernel"] = {k: {"fetch_kb": round(pf[k][0] / total_passes * per, 3),
                             "write_kb": round(pw.get(k, [0.0, 0])[0] / total_passes * per, 3),
                             "launches_per_pass": round(pf[k][1] / total_passes * per, 3)}
                         for k in sorted(pf)}
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    recs = []
    if os.path.exists(path):
        recs = [r for r in json.load(open(path))
                if not (r.get("fs") == fs and r.get("workload", "analysis_synthesis") == workload)]
    recs.append(rec)
    json.dump(recs, open(path, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
