"""The N > 1 path on the HIP kernels themselves: bench.py --gpus 2 starts its two ranks itself (gloo rendezvous, both ranks on
the one GPU of the test box), deals a fixed job of voice-clone utterances to them (strong scaling) and must produce, utterance by
utterance, the codes a single rank produces for the whole job."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(tmp_path, gpus, tag):
    dump = tmp_path / f"crc_{tag}.json"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--mode", "strong", "--utterances", "24", "--batch", "8",
           "--frames", "20", "--steps", "1", "--warmup", "1", "--shape", "T0", "--dtype", "f32", "--greedy", "--no-cpu-baseline",
           "--dump-codes", str(dump)]
    env = dict(os.environ, QTTS_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), json.load(open(dump))


@pytest.mark.timeout(600)
def test_two_ranks_reproduce_one_rank_on_the_device(tmp_path):
    one, crc1 = _bench(tmp_path, 1, "one")
    two, crc2 = _bench(tmp_path, 2, "two")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["scaling"] == two["scaling"] == "strong"
    assert one["config"]["utterances_total"] == two["config"]["utterances_total"] == 24
    assert one["value"] > 0 and two["value"] > 0
    assert len(crc1) == 24 and all(v is not None for v in crc2.values())
    assert crc1 == crc2          # f32 greedy: every utterance's codes, whichever rank and batch decoded it


@pytest.mark.timeout(600)
def test_benchmark_drives_continuous_batching(tmp_path):
    """bench.py with utterances of unequal length: the sessions work through their queues in row slots, finished utterances are
    decoded by the codec from inside the completion hook, and the line accounts for every utterance and frame.  The strong job
    with the same spread gives every utterance the codes of the weak job's plain decode of it (f32 greedy: rows never interact)."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--shape", "T0", "--dtype", "f32", "--greedy", "--no-cpu-baseline",
            "--frames", "24", "--batch", "8", "--warmup", "1"]
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)

    def run(extra):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=280, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        return json.loads(lines[0])

    spread = run(["--steps", "4", "--length-spread", "6:24", "--slots", "5", "--codec-group", "3"])
    assert spread["config"]["continuous_batching"] is True and spread["config"]["fixed_length"] is False
    assert spread["per_rank"][0][2] == 4 * 8                    # every utterance of the four steps was handed over
    frames = spread["per_rank"][0][1] / 0.08                    # audio seconds -> frames of all utterances
    assert 4 * 8 * 6 <= round(frames) <= 4 * 8 * 24 and spread["value"] > 0
    assert spread["roofline_codec"]["device_ms_isolated"] > 0   # the codec ran (inside the hook and for the isolated figure)
    dump_a, dump_b = tmp_path / "a.json", tmp_path / "b.json"
    run(["--mode", "strong", "--utterances", "20", "--steps", "1", "--length-spread", "6:24", "--slots", "4", "--dump-codes", str(dump_a)])
    run(["--mode", "strong", "--utterances", "20", "--steps", "1", "--length-spread", "6:24", "--slots", "16", "--dump-codes", str(dump_b)])
    a, b = json.load(open(dump_a)), json.load(open(dump_b))
    assert len(a) == 20 and a == b                              # the slot count does not change any utterance's codes
