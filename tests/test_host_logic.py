import importlib

import numpy as np

sd = importlib.import_module("hts-train-world_amd.synth_data")
sh = importlib.import_module("hts-train-world_amd.sharding")


def test_generator_is_deterministic_and_int16_exact():
    a = sd.make_utterance(3, 16000, duration=0.5)
    b = sd.make_utterance(3, 16000, duration=0.5)
    np.testing.assert_array_equal(a, b)
    assert np.all(a * 32768 == np.round(a * 32768))        # exactly what wavread would hand over
    assert 0.29 < np.abs(a).max() <= 0.3001
    # never digital silence: no run of 64 exact zeros
    z = (a == 0).astype(int)
    assert np.convolve(z, np.ones(64), "valid").max() < 64


def test_durations_follow_range():
    lens = [len(sd.make_utterance(i, 16000, (2.0, 8.0))) for i in range(8)]
    assert all(2.0 * 16000 <= n <= 8.0 * 16000 for n in lens)
    assert len(set(lens)) > 4


def test_frame_count_matches_reference_formula():
    assert sh.frame_count(53680, 16000, 5.0) == 672        # GetSamplesForDIO (dio.cpp:638-640), config 1
    assert sh.frame_count(192000, 48000, 1.0) == 4001


def test_lpt_shards_balance_and_cover():
    rng = np.random.default_rng(0)
    costs = rng.integers(400, 1600, 1000).tolist()
    shards = sh.lpt_shards(costs, 8)
    flat = sorted(i for s in shards for i in s)
    assert flat == list(range(1000))
    loads = [sum(costs[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(costs)
    assert sh.lpt_shards([5, 3], 4) == [[0], [1], [], []]


def test_lpt_shards_with_a_handicap_for_rank_0():
    """The sweep's rank 0 also moves everybody's features to the host: its frames count for more, so it gets fewer
    (sharding.rank0_handicap: 1.11 / 1.21 / 1.37 at 2 / 4 / 8 ranks); every utterance is still dealt exactly once and
    the WEIGHTED loads balance."""
    rng = np.random.default_rng(1)
    costs = rng.integers(400, 1600, 1000).tolist()
    assert sh.rank0_handicap(1) == [1.0]
    for world, f0 in ((2, 1.1073), (4, 1.2146), (8, 1.37)):
        h = sh.rank0_handicap(world)
        assert abs(h[0] - f0) < 1e-3 and h[1:] == [1.0] * (world - 1)
        shards = sh.lpt_shards(costs, world, h)
        assert sorted(i for s in shards for i in s) == list(range(1000))
        loads = [sum(costs[i] for i in s) for s in shards]
        weighted = [l * k for l, k in zip(loads, h)]
        assert max(weighted) - min(weighted) <= max(costs) * h[0]
        assert loads[0] < min(loads[1:])                      # rank 0 analyses less ...
        assert abs(loads[0] * h[0] - np.mean(loads[1:])) <= max(costs) * h[0]   # ... by its factor
    assert sh.lpt_shards(costs, 4, None) == sh.lpt_shards(costs, 4, [1.0] * 4)
    assert sh.rank0_handicap(8, factor=2.0)[0] == 2.0


def test_native_file_writer(tmp_path):
    """WorldMi355WriteFiles (host only): a list of arrays to a list of files with plain threads -- the fwrite loops at
    the end of the reference's analysis CLI (test/analysis.cpp:360-390) for a whole batch; errors name the file."""
    import pytest
    W = importlib.import_module("hts-train-world_amd.world")
    rng = np.random.default_rng(5)
    slab = rng.standard_normal((500, 7)).astype(np.float32)
    items, want = [], {}
    off = 0
    for k in range(40):                                    # row slices of one slab, ragged, one empty
        n = 0 if k == 11 else int(rng.integers(1, 25))
        a = slab[off:off + n]
        off += n
        p = tmp_path / ("u%03d.sp" % k)
        items.append((p, a))
        want[p] = a.tobytes()
    (tmp_path / "u005.sp").write_bytes(b"x" * 10000)       # an older, longer file is truncated
    W.write_files(items, threads=6)
    for p, b in want.items():
        assert p.read_bytes() == b
    with pytest.raises(RuntimeError, match="no_such_dir"):
        W.write_files([(tmp_path / "no_such_dir" / "a.f0", slab[:3])], threads=2)
    with pytest.raises(AssertionError):
        W.write_files([(tmp_path / "t.f0", slab[:, 0])])   # a column of a slab is not contiguous


def test_driver_line_fits_the_drivers_tail():
    """bench.py prints a compact line (the driver records the last 8 KB of stdout; round 4's 14 KB line lost its first
    half there): every workload's numbers survive, the contract's keys and objects are intact."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    with open(os.path.join(root, "profiles", "r04_g_bench_default.json")) as f:
        full = json.loads(f.read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 8192                         # the case that went wrong
    line = bench.driver_line(full, "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) <= 7000, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert line["roofline"][k] == full["roofline"][k], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert line["host_inclusive"]["value"] == full["host_inclusive"]["value"]
    assert line["host_inclusive_coded"]["value"] == full["host_inclusive_coded"]["value"]
    assert line["parity"]["vs"] == "reference"
    for name in ("harvest", "synthesis", "sweep"):
        side, ref = line["side_workloads"][name], full["side_workloads"][name]
        assert side["value"] == ref["value"] and side["ms_per_step"] == ref["ms_per_step"]
        assert side["roofline"]["frac"] == ref["roofline"]["frac"] and side["roofline"]["launch_ms"] == ref["roofline"]["launch_ms"]
        assert side["cpu_baseline"]["value"] == ref["cpu_baseline"]["value"] and side["parity"]
        assert line["summary"][name] == ref["value"]
    assert list(line)[-1] == "summary"                           # the numbers that matter are the line's last bytes
