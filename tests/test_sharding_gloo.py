"""world_size-2 `gloo` run of the multi-GPU path on CPU: utterances are dealt to ranks, each rank decodes its share
independently (here with the CPU oracle standing in for the device), results are gathered to rank 0 and must equal
the single-process results utterance by utterance."""
import os
import socket

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    import torch.distributed as dist
    from conftest import make_oracle
    from golden.make_golden import GREEDY
    from oracle import bind
    from qwen3_tts_rs_amd import synth
    from qwen3_tts_rs_amd.shard import gather_results, run_sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o, cfg, _, tmpl = make_oracle(bind, "T0", "f32", talker=True, codec=False)
    texts = [synth.synthetic_text_ids(tmpl, 9 + 2 * i, i) for i in range(5)]
    prefill = bind.resolve_codec_prefill(cfg, "english", "speaker1")
    spk = bind.resolve_speaker_token(cfg, "speaker1")

    def fn(items):
        return [o.generate(t, prefill, GREEDY, 4, speaker_token=spk)["codes"] for t in items]

    idx, res = run_sharded(texts, [len(t) for t in texts], fn, rank, world)
    dist.barrier()
    out = gather_results(idx, res, len(texts), dist)
    if rank == 0:
        q.put((idx, [np.asarray(x) for x in out]))
    else:
        assert out is None
        q.put((idx, None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharding_matches_single_process(bind):
    import torch.multiprocessing as mp
    from conftest import make_oracle
    from golden.make_golden import GREEDY
    from qwen3_tts_rs_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    shares = sorted(g[0] for g in got)
    assert sorted(i for s in shares for i in s) == list(range(5)) and all(len(s) >= 2 for s in shares)
    gathered = next(g[1] for g in got if g[1] is not None)
    o, cfg, _, tmpl = make_oracle(bind, "T0", "f32", talker=True, codec=False)
    prefill = bind.resolve_codec_prefill(cfg, "english", "speaker1")
    spk = bind.resolve_speaker_token(cfg, "speaker1")
    for i in range(5):
        ref = o.generate(synth.synthetic_text_ids(tmpl, 9 + 2 * i, i), prefill, GREEDY, 4, speaker_token=spk)["codes"]
        assert np.array_equal(ref, gathered[i])


def _reduce_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank, bench.reduce_job(1.0 + rank, 10.0 * (rank + 1), dist, "cpu")))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bench_aggregates_max_time_and_total_audio():
    """bench.py's only collective: value = audio seconds of ALL ranks / the SLOWEST rank's time."""
    import torch.multiprocessing as mp
    import bench
    assert bench.reduce_job(2.0, 5.0, None, "cpu") == (2.0, 5.0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert got[0] == got[1] == (2.0, 30.0)
