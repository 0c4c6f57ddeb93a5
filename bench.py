#!/usr/bin/env python3
"""Headline benchmark: audio-seconds generated per wall-second.

Workload (one "step"): batch of synthetic utterances per GPU through the whole hot path of the 1.7B-like shape set
(prompt -> talker decode -> code predictor -> sampling for `--frames` frames, then the codec decoder to 24 kHz
waveforms on the host).  Weights are random-init of that architecture, texts are synthetic token ids.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus N ...            (no launcher: the script starts its N ranks itself)

Every rank holds one full set of weights and decodes its own utterances (no collective on the data path); within a rank
`--pipeline` batches are in flight at once, each in a decode session with its own HIP stream, and a step is one batch through
the whole path.  `--mode strong` fixes the total work instead: `--utterances` voice-clone utterances (reference clip -> mel ->
speaker encoder -> x-vector prompt -> decode -> codec) are dealt to the ranks by shard.assign_utterances, a step is one pass
over all of them.  `--natural-eos` and `--length-spread LO:HI` (utterances of unequal length, in either mode) decode through
continuous batching (`--continuous` asks for it with fixed lengths too): every session works through its utterances in `--slots` rows, a finished row's slot goes to the next
queued utterance, and finished utterances are turned into audio in groups of `--codec-group` while the other rows keep
decoding (`--plain` forces whole batches instead).  Rank 0 prints one JSON line.  The CPU oracle is timed on a bounded sample as a reported baseline (rank 0,
N = 1 only)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 matrix
FRAME_SECONDS = 0.08          # 12.5 Hz codec frames
SEED = 20261111


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pipeline", type=int, default=3,
                    help="batches in flight per GPU: each runs on its own HIP stream (own context + weight replica); the decode "
                         "frame is latency-bound, so independent batches overlap almost for free")
    ap.add_argument("--shape", default="S17")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU per step")
    ap.add_argument("--frames", type=int, default=375, help="frames generated per utterance (30 s of audio)")
    ap.add_argument("--greedy", action="store_true", help="CLI --greedy semantics instead of seeded sampling")
    ap.add_argument("--natural-eos", action="store_true",
                    help="let rows end where the model says EOS (the reference's rule); the default keeps EOS suppressed for --frames "
                         "frames so that every row of every run produces the same fixed length")
    ap.add_argument("--no-codec", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the decode frame kernel by kernel (counter collection cannot follow hipGraphs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=96)
    ap.add_argument("--first-frames", type=int, default=25,
                    help="frames of the first audio chunk (2 s): decoded as soon as they exist, while generation continues")
    ap.add_argument("--mode", default="weak", choices=["weak", "strong"],
                    help="weak: --batch utterances per GPU per step; strong: --utterances voice-clone utterances in total per step, "
                         "dealt to the ranks")
    ap.add_argument("--utterances", type=int, default=256, help="strong mode: utterances per step over all ranks")
    ap.add_argument("--strong-rows", type=int, default=0,
                    help="strong mode: rows per decode launch (0 = by the rank's share: 128 from 256 utterances on, else 64 - on one MI355X "
                         "two sessions of 128 rows decode 2255 audio-s/s, three of 64 2049, two of 64 1667, one of 128 1467)")
    ap.add_argument("--continuous", action="store_true",
                    help="decode through continuous batching (implied by --natural-eos and --length-spread)")
    ap.add_argument("--plain", action="store_true", help="whole batches through qtts_generate_ex even where continuous batching is implied")
    ap.add_argument("--length-spread", default=None, metavar="LO:HI",
                    help="utterance u is LO + hash(u) %% (HI - LO + 1) frames long (EOS suppressed up to its own length) instead of "
                         "--frames for all: the spread of lengths a real job has")
    ap.add_argument("--slots", type=int, default=0,
                    help="continuous batching: rows per session (0 = 128, on two sessions unless --pipeline says otherwise)")
    ap.add_argument("--codec-group", type=int, default=32, help="continuous batching: finished utterances per codec decode")
    ap.add_argument("--queue-order", default="longest-first", choices=["longest-first", "arrival"],
                    help="continuous batching with --length-spread: a session's queue ordered by the requests' own frame limits, longest "
                         "first (the rows then drain together at the end of the job instead of idling behind the last long utterance), "
                         "or left in arrival order")
    ap.add_argument("--dump-codes", default=None, help="rank 0 writes {utterance index: crc32 of its codes} of the last step here (json)")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks as child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as
    torch.distributed.run would set them), pass rank 0's JSON line through and return the worst exit code.  The parent never
    touches the GPU."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    for line in out.splitlines():   # the contract line to stdout, anything a library printed there to stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr, flush=True)
    return max(abs(c) for c in codes)


def build_model(args, device):
    import torch
    from qwen3_tts_rs_amd import LoaderConfig, Model, synth
    clone = args.mode == "strong"
    cfg, ccfg, tmpl = synth.shape_set(args.shape, "base" if clone else "custom_voice")
    m = Model(cfg, None if args.no_codec else ccfg, LoaderConfig(dtype=args.dtype, device=device))
    keys = synth.talker_keys(cfg) + (synth.speaker_keys(cfg) if clone else []) + ([] if args.no_codec else synth.codec_keys(ccfg))
    for k, t in synth.iter_weights_torch(keys, SEED, f"cuda:{device}", args.dtype == "bf16"):
        m.load_tensor(k, t)
        del t
    m.finalize()
    torch.cuda.empty_cache()
    return m, cfg, ccfg, tmpl


def text_len(u):
    return 8 + (u * 37 + 11) % 57                      # text content length in [8, 64]


def own_frames(args, u):
    """Frame limit of utterance u under --length-spread (0 = the call's --frames)."""
    if not args.length_spread:
        return 0
    lo, hi = (int(v) for v in args.length_spread.split(":"))
    return lo + (u * 2654435761 % 4294967296) % (hi - lo + 1)


def make_requests(args, tmpl, rank, step=0):
    from qwen3_tts_rs_amd import GenerationRequest, synth
    reqs = []
    for b in range(args.batch):
        u = rank * args.batch + b                      # global utterance index
        reqs.append(GenerationRequest(text_ids=synth.synthetic_text_ids(tmpl, text_len(u), u), language="english", speaker="speaker1",
                                      seed=None if args.greedy else 42 + u, max_new_tokens=own_frames(args, u + 7919 * step)))
    return reqs


def reference_clip(u):
    """Synthetic reference audio of utterance u: 3.0 - 9.9 s at 24 kHz, a few sines under noise."""
    n = 72000 + 641 * (u % 257)
    rng = np.random.default_rng(1000 + u)
    t = np.arange(n, dtype=np.float32) / 24000.0
    x = sum(a * np.sin(2 * np.pi * f * t + p) for a, f, p in zip((0.3, 0.2, 0.1), (110.0 + 7 * (u % 13), 440.0, 1870.0), rng.uniform(0, 6.28, 3)))
    return (x + 0.02 * rng.standard_normal(n)).astype(np.float32)


def strong_share(args, rank, world):
    """This rank's utterances of the fixed job (ascending global indices): longest-first assignment on text length + clip length."""
    from qwen3_tts_rs_amd.shard import assign_utterances
    costs = [text_len(u) + (72000 + 641 * (u % 257)) / 1920.0 for u in range(args.utterances)]
    return assign_utterances(costs, world)[rank]


def cpu_baseline(args, cfg, ccfg, tmpl):
    """The CPU oracle (a port of the reference's algorithm, f32, OpenMP over the host cores) on a bounded sample of the
    same workload: one utterance, a few frames, plus the codec decode of those frames."""
    import torch
    from oracle import bind
    from qwen3_tts_rs_amd import synth
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    o = bind.Oracle(fast=True)
    keys = synth.talker_keys(cfg) + ([] if args.no_codec else synth.codec_keys(ccfg))
    for k, t in synth.iter_weights_torch(keys, SEED, "cuda:0", False):
        o.load_tensor(k, t.float().cpu().numpy())
        del t
    o.set_model_config(cfg)
    if not args.no_codec:
        o.set_codec_config(ccfg)
    text = synth.synthetic_text_ids(tmpl, 12, 0)
    samp = dict(do_sample=False, subtalker_do_sample=False, temperature=1.0, top_k=50, top_p=1.0, repetition_penalty=1.05)
    t0 = time.perf_counter()
    r = o.generate(text, bind.resolve_codec_prefill(cfg, "english", "speaker1"), samp, args.cpu_frames,
                   speaker_token=bind.resolve_speaker_token(cfg, "speaker1"))
    t1 = time.perf_counter()
    n = r["effective_len"]
    if not args.no_codec and n > 0:
        o.codec_decode(r["codes"][:n, :ccfg["decoder_config"]["num_quantizers"]])
    t2 = time.perf_counter()
    audio_s = n * FRAME_SECONDS
    return {"value": audio_s / (t2 - t0), "unit": "audio_s/s", "cores": threads, "kind": "port",
            "sample": f"1 utterance (12 text tokens), {n} frames greedy + codec decode of those frames, f32, "
                      f"generate {t1 - t0:.2f} s, codec {t2 - t1:.2f} s"}


def pin_near_gpu(local: int):
    """Keeps this rank's host threads (one per decode session) on the CPUs of its GPU's NUMA node when the machine has more than
    one node and the process is allowed to run there.  Returns the node, or None when nothing was changed."""
    try:
        import glob
        import torch
        if len(glob.glob("/sys/devices/system/node/node[0-9]*")) < 2:
            return None
        pr = torch.cuda.get_device_properties(local)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        allowed = cpus & os.sched_getaffinity(0)
        if len(allowed) < 4:          # a cpuset that leaves almost nothing of that node: stay where we are
            return None
        os.sched_setaffinity(0, allowed)
        return node
    except Exception:
        return None


def gather_ranks(values, dist, device: str):
    """values: this rank's [elapsed_s, audio_s, utterances, frames] -> the same for every rank (list of lists), so that a
    scaling result below N x can be told apart: imbalance (audio per rank) or host contention (time per rank)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [list(values)]
    import torch
    mine = torch.tensor(values, dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [[float(v) for v in t] for t in out]


def reduce_job(elapsed: float, audio_s: float, dist, device: str):
    """Whole-job aggregate over ranks: (slowest rank's time, audio seconds of all ranks).  The only collective of the path."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return elapsed, audio_s
    import torch
    mx = torch.tensor([elapsed], dtype=torch.float64, device=device)
    sm = torch.tensor([audio_s], dtype=torch.float64, device=device)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    return float(mx[0]), float(sm[0])


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    # QTTS_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("QTTS_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend != "nccl":
        local = local % max(ndev, 1)
    if local >= ndev:
        print(f"bench.py: rank {rank} has LOCAL_RANK={local} but this node shows {ndev} GPU(s): one rank per GPU is needed "
              f"(QTTS_BENCH_BACKEND=gloo lets ranks share devices for a rehearsal)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    pinned = pin_near_gpu(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import threading
    from qwen3_tts_rs_amd import SamplingConfig, synth
    strong = args.mode == "strong"
    # fixed-length jobs gain nothing from slot refill (all rows of a batch end together) and a lone last batch decodes faster than
    # its rows spread over three busy sessions, so whole batches stay the default there
    continuous = (args.continuous or args.natural_eos or bool(args.length_spread)) and not args.plain
    if args.length_spread and not continuous:
        sys.exit("bench.py: --length-spread needs continuous batching (a plain batch runs every row to the same length)")
    mine = strong_share(args, rank, world) if strong else []
    if strong:
        # the job is a fixed set of utterances, so the rows per launch are the scheduler's choice (the weak line keeps the named
        # batch of 64); a rank's share goes through as few sessions as hold it: a lone stream decodes a frame in about 60 % of
        # the time each of three concurrent streams takes, with kernels picked for running alone
        if args.strong_rows > 0:
            args.batch = args.strong_rows
        elif args.batch == 64:   # (an explicit --batch is kept)
            args.batch = 128 if len(mine) >= 256 else 64
        P = max(1, min(args.pipeline, -(-len(mine) // args.batch)))
    elif continuous:
        # slots are a serving setting, not the named batch: 128 rows on two sessions decode more per second than 64 on three
        # (lengths 100-375: 1630 against 1410 audio-s/s); explicit --slots / --pipeline are kept
        if args.slots == 0:
            args.slots = 128
        P = max(1, min(2 if args.pipeline == 3 and args.slots > 64 else args.pipeline, args.steps))
    else:
        P = max(1, min(args.pipeline, args.steps))
    # one set of weights per GPU; every batch in flight beyond the first runs in a session of the same model (its own HIP
    # stream, KV caches, workspaces and frame graph)
    m, cfg, ccfg, tmpl = build_model(args, local)
    models = [m] + [m.open_session() for _ in range(P - 1)]
    clips = [reference_clip(u) for u in mine]            # host audio, prepared outside the timed region like the texts
    reqs = make_requests(args, tmpl, rank)
    sampling = SamplingConfig.cli_greedy() if args.greedy else SamplingConfig()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    stats = dict(prefill_ms=0.0, frames_ms=0.0, frames=0, codec_ms=0.0, audio_s=0.0, step_ms=[], first_ms=[], launch_host_ms=0.0, utterances=0,
                 row_frames=0, calls=0)
    out_last = [None]
    lock = threading.Lock()

    csize = ccfg["decoder_config"]["codebook_size"]

    def step(mod, record, reqs=reqs):
        t_s = time.perf_counter()
        first_ms = []

        cb_ms = []
        cb_parts = []

        def on_first(codes):
            t_cb = time.perf_counter()
            # streaming decode session: the samples this push returns (1920 n - 555) are exactly those of the whole decode
            # a row that already produced EOS (a cb0 outside the codebook) has no audio from that frame on: feed silence codes
            done = np.maximum.accumulate(codes[:, :, 0] >= csize, axis=1)
            codes = np.where(done[:, :, None], 0, codes).astype(np.uint32)
            cs = mod.open_codec_stream(codes.shape[0], codes.shape[1])
            t_o = time.perf_counter()
            cs.push(codes)
            first_ms.append((time.perf_counter() - t_s) * 1e3)
            t_p = time.perf_counter()
            cs.close()
            cb_ms.append((time.perf_counter() - t_cb) * 1e3)
            cb_parts.append(((t_o - t_cb) * 1e3, (t_p - t_o) * 1e3, (time.perf_counter() - t_p) * 1e3, mod.codec_device_ms()))

        stream_first = not args.no_codec and 2 < args.first_frames < args.frames
        out = mod.generate(reqs, sampling, args.frames, use_graph=not args.no_graph, first_frames=args.first_frames if stream_first else 0,
                           on_first_chunk=on_first if stream_first else None, min_new_tokens=0 if args.natural_eos else args.frames)
        t_g = time.perf_counter()
        wavs = [] if args.no_codec else mod.decode_output(out)
        if record and os.environ.get("QTTS_BENCH_TIMING"):   # where a step's wall time goes beyond the device phases (stderr)
            t_d = time.perf_counter()
            parts = " / ".join("%.1f" % v for v in (cb_parts[0] if cb_parts else ()))
            print(f"step: generate wall {(t_g - t_s) * 1e3:.1f} ms (device prefill {mod.last_timing['prefill_ms']:.1f} + frames "
                  f"{mod.last_timing['frames_ms']:.1f}), first-chunk callback at {first_ms[0] if first_ms else 0:.1f} ms, "
                  f"(the callback took {cb_ms[0] if cb_ms else 0:.1f} ms: open / push / close / push on the device {parts}), decode wall {(t_d - t_g) * 1e3:.1f} ms (device {0.0 if args.no_codec else mod.codec_device_ms():.1f})", file=sys.stderr)
        if record:
            with lock:
                stats["prefill_ms"] += mod.last_timing["prefill_ms"]
                stats["frames_ms"] += mod.last_timing["frames_ms"]
                stats["frames"] += mod.last_timing["frames"]
                stats["launch_host_ms"] += mod.last_timing["graph_launch_host_ms"]
                stats["codec_ms"] += 0.0 if args.no_codec else mod.codec_device_ms()
                stats["audio_s"] += sum(out.effective_lengths) * FRAME_SECONDS
                stats["utterances"] += len(out.effective_lengths)
                stats["step_ms"].append((time.perf_counter() - t_s) * 1e3)
                stats["first_ms"].append(first_ms[0] if first_ms else stats["step_ms"][-1])
                out_last[0] = out
        return out, wavs

    G = cfg["talker_config"]["num_code_groups"]

    def session_slots(n_queued):
        if args.slots > 0:
            return min(args.slots, n_queued)
        return min(args.batch, n_queued)

    def run_continuous(mod, rq, record, on_codes=None):
        """One session's utterances through continuous batching; finished utterances go to the codec in groups, on this
        session's stream, between two frames of its other rows."""
        t_s = time.perf_counter()
        pending, tally = [], dict(codec_ms=0.0, audio_s=0.0, done=0)

        def flush():
            if pending and not args.no_codec:
                T = max(len(c) for c in pending)
                arr = np.zeros((len(pending), T, G), np.uint32)
                for i, c in enumerate(pending):
                    arr[i, :len(c)] = c
                mod.decode_codes(arr, [len(c) for c in pending], chunk=-2)
                tally["codec_ms"] += mod.codec_device_ms()
            pending.clear()

        def on_done(u, codes, eff):
            tally["audio_s"] += eff * FRAME_SECONDS
            tally["done"] += 1
            if on_codes is not None:
                on_codes(u, codes[:eff])
            if eff > 0:
                pending.append(codes[:eff])
            if len(pending) >= args.codec_group:
                flush()

        out = mod.generate(rq, sampling, args.frames, use_graph=not args.no_graph, slots=session_slots(len(rq)), on_done=on_done,
                           min_new_tokens=0 if args.natural_eos else args.frames)
        flush()
        if record:
            with lock:
                stats["prefill_ms"] += mod.last_timing["prefill_ms"]
                stats["frames_ms"] += mod.last_timing["frames_ms"]
                stats["frames"] += mod.last_timing["frames"]
                stats["row_frames"] += sum(out.steps_per_utterance)
                stats["launch_host_ms"] += mod.last_timing["graph_launch_host_ms"]
                stats["codec_ms"] += tally["codec_ms"]
                stats["audio_s"] += tally["audio_s"]
                stats["utterances"] += tally["done"]
                stats["step_ms"].append((time.perf_counter() - t_s) * 1e3)
                stats["calls"] += 1
                out_last[0] = out
        return out

    codes_crc = {}
    probe = [reqs]   # one batch of requests for the eager profiling pass after the timed region

    def strong_batches():
        """One pass over this rank's utterances: reference clips -> x-vectors on the device (mel + speaker encoder), then
        batches of --batch utterances."""
        from qwen3_tts_rs_amd import GenerationRequest
        import zlib
        emb = m.extract_speaker_embedding(clips) if mine else np.zeros((0, 1), np.float32)
        out = []
        for i in range(0, len(mine), args.batch):
            idx = mine[i:i + args.batch]
            out.append((idx, [GenerationRequest(text_ids=synth.synthetic_text_ids(tmpl, text_len(u), u), language="english",
                                                speaker_embed=emb[i + j], seed=None if args.greedy else 42 + u) for j, u in enumerate(idx)]))
        return out, zlib.crc32

    def strong_requests():
        from qwen3_tts_rs_amd import GenerationRequest
        emb = m.extract_speaker_embedding(clips) if mine else np.zeros((0, 1), np.float32)
        return [GenerationRequest(text_ids=synth.synthetic_text_ids(tmpl, text_len(u), u), language="english", speaker_embed=emb[j],
                                  seed=None if args.greedy else 42 + u, max_new_tokens=own_frames(args, u)) for j, u in enumerate(mine)]

    def run_steps_continuous(n_steps, record):
        """Every session gets one queue per call: in weak mode the batches of all its steps (rows of the next batch start as rows
        of the previous one finish), in strong mode its part of the rank's share, pass after pass."""
        import zlib
        errs = []
        queues = [[] for _ in range(P)]   # per session: list of (global utterance index or None, request)
        if strong:
            for _ in range(n_steps):
                rq = strong_requests()
                if rq:
                    probe[0] = rq[:args.batch]
                for w in range(P):
                    queues[w].append([(mine[j], rq[j]) for j in range(w, len(rq), P)])
        else:
            for k in range(n_steps):
                queues[k % P].append([(None, r) for r in (make_requests(args, tmpl, rank, k) if args.length_spread else reqs)])
            queues = [[sum(q, [])] if q else [] for q in queues]   # one call per session over all its batches

        def worker(w):
            try:
                for items in queues[w]:
                    if not items:
                        continue
                    if args.length_spread and args.queue_order == "longest-first":
                        items = sorted(items, key=lambda it: -(it[1].max_new_tokens or args.frames))
                    ids = [u for u, _ in items]

                    def keep(i, codes, ids=ids):
                        if ids[i] is not None and record:
                            with lock:
                                codes_crc[ids[i]] = int(zlib.crc32(np.ascontiguousarray(codes).tobytes()))
                    run_continuous(models[w], [r for _, r in items], record, keep)
            except Exception as ex:  # surfaced after the join
                errs.append(ex)

        th = [threading.Thread(target=worker, args=(w,)) for w in range(P)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]

    def run_steps(n_steps, record):
        # weak: steps are dealt round-robin to the in-flight slots; every slot runs its steps back to back on its own stream.
        # strong: every step is one pass over the rank's share, its batches dealt to the slots the same way.
        if continuous:
            return run_steps_continuous(n_steps, record)
        errs = []
        if strong:
            work = []
            for _ in range(n_steps):
                batches, crc = strong_batches()
                work += batches
            if work:
                probe[0] = work[0][1]
        else:
            work = [(None, reqs)] * n_steps

        def worker(w):
            try:
                for idx, rq in work[w::P]:
                    out, _ = step(models[w], record, rq)
                    if idx is not None and record:
                        with lock:
                            for j, u in enumerate(idx):
                                codes_crc[u] = int(crc(np.ascontiguousarray(out.codes[j, :out.effective_lengths[j]]).tobytes()))
            except Exception as ex:  # surfaced after the join
                errs.append(ex)

        th = [threading.Thread(target=worker, args=(w,)) for w in range(P)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]

    run_steps(args.warmup, False)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    barrier()
    elapsed = time.perf_counter() - t0

    rdev = f"cuda:{local}" if backend == "nccl" else "cpu"
    elapsed_max, audio_total = reduce_job(elapsed, stats["audio_s"], dist if world > 1 else None, rdev)
    per_rank = gather_ranks([elapsed, stats["audio_s"], stats["utterances"], stats["frames"]], dist if world > 1 else None, rdev)
    if strong and args.dump_codes:
        from qwen3_tts_rs_amd.shard import gather_results
        got = gather_results(sorted(codes_crc), [codes_crc[u] for u in sorted(codes_crc)], args.utterances, dist if world > 1 else None)
        if rank == 0:
            with open(args.dump_codes, "w") as f:
                json.dump({str(u): c for u, c in enumerate(got)}, f)

    result = None
    if rank == 0:
        es = 2 if args.dtype == "bf16" else 4
        if continuous:
            frames_run = max(stats["frames"], 1)                  # every frame is a launch; the loop's time includes the admissions
            rows_live = stats["row_frames"] / frames_run          # mean rows that produced a frame per launch
            avg_ctx = 40 + stats["row_frames"] / max(stats["utterances"], 1) / 2.0
        else:
            calls = len(stats["step_ms"])                         # generate calls of this rank in the timed region
            frames_run = max(stats["frames"] - calls, 1)          # the first frame of each call comes out of the prefill
            rows_live = float(args.batch)
            avg_ctx = 40 + args.frames / 2.0                      # mean prompt (~40 positions) + half the generated frames
        ms_per_frame = stats["frames_ms"] / frames_run            # per stream, HIP events around each call's frame loop
        bytes_frame = synth.decode_bytes_per_frame(cfg, rows_live, avg_ctx, es)
        # frame loops of the in-flight batches overlap (and overlap codec decodes): the bandwidth the decode frames achieve
        # together is bounded below by their algorithmic bytes over the whole timed region
        achieved = bytes_frame * frames_run / elapsed / 1e9
        # HBM-side bytes of one frame from the PMC counters: collected once per round with tools/dev/pmc_frame.sh (rocprofv3 cannot
        # follow a hipGraph, so the passes run the frame eagerly), committed under profiles/ and quoted here when this run is the
        # configuration they were taken on (bench_data/ travels with the tree to the GPU box, profiles/ does not)
        # The figure is quoted only for the kernel selection the counters were taken on ("alone": one stream; "shared": the
        # kernels picked for several streams in flight, which is what P > 1 runs), and says where it comes from.
        traffic, traffic_ref = None, None
        tpath = os.path.join(ROOT, "bench_data", "frame_traffic.json")   # a copy lives under profiles/
        if os.path.exists(tpath) and args.shape == "S17" and args.batch == 64 and args.dtype == "bf16" and not continuous:
            tj = json.load(open(tpath)).get("shared" if P > 1 else "alone")
            if tj:
                traffic = tj.get("traffic_bytes")
                traffic_ref = {"bytes": traffic, "kernel_selection": "shared" if P > 1 else "alone", "measured": tj.get("measured"),
                               "how": "rocprofv3 --pmc passes over eagerly launched frames of one stream (tools/dev/pmc_frame.sh)"}
        roofline = {"bound": "hbm", "kernel": "decode frame (one hipGraph launch: talker step + code-predictor passes + sampling)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_reference": traffic_ref, "bytes_per_launch": bytes_frame, "avg_launch_ms": round(ms_per_frame, 4),
                    "launches": frames_run, "streams_in_flight": P,
                    "note": "achieved = algorithmic bytes of every decode frame in the timed region / wall time of the region "
                            "(includes the overlapped prefill and codec phases); avg_launch_ms is per stream",
                    "weight_bytes_per_launch": synth.weight_bytes_per_frame(cfg, es)["total"],
                    # the same bytes over the duration of one frame launch of ONE stream (HIP events around each call's frame loop):
                    # the per-step fraction; with several streams in flight their frames overlap, so this understates the chip
                    "per_step": {"achieved": round(bytes_frame / (ms_per_frame * 1e-3) / 1e9, 1),
                                 "frac": round(bytes_frame / (ms_per_frame * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "avg_launch_ms": round(ms_per_frame, 4), "streams_sharing_the_gpu": P}}
        # the same frames priced against the matrix cores: at 64 rows a weight-streaming GEMM does 64 FLOP per byte, far below the
        # ridge (~310 FLOP/B), so this is a reported utilisation, not the bound
        fl_frame = synth.decode_flops_per_frame(cfg, rows_live, avg_ctx)
        mfma_peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
        mfma_ach = fl_frame * frames_run / elapsed / 1e12
        roofline["mfma_utilisation"] = {"achieved": round(mfma_ach, 1), "peak": mfma_peak, "unit": "TFLOP/s",
                                        "frac": round(mfma_ach / mfma_peak, 4), "flops_per_launch": fl_frame}
        # dominant weight-streaming GEMM, timed launch by launch with HIP events in an eager pass after the timed region
        m.generate(probe[0], sampling, 9, use_graph=False, profile_gemm=True)
        prof = [p for p in m.gemm_profile() if p["M"] == len(probe[0])]
        gemm = None
        if prof:
            top = max(prof, key=lambda p: p["total_us"])
            avg_us = top["total_us"] / top["launches"]
            b2b = m.bench_gemm(top["M"], top["N"], top["K"], 100)
            wbytes = top["N"] * top["K"] * es
            gemm = {"bound": "hbm", "kernel": f"skinny GEMM M={top['M']} N={top['N']} K={top['K']} ({args.dtype})",
                    "achieved": round(wbytes / (b2b * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(wbytes / (b2b * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_launch": wbytes,
                    "avg_launch_us_back_to_back": round(b2b, 2), "avg_launch_us_in_frame_events": round(avg_us, 2),
                    "share_of_gemm_time": round(top["total_us"] / sum(p["total_us"] for p in m.gemm_profile()), 3)}
        codec = None
        if not args.no_codec and stats["codec_ms"] > 0:
            # one decode of the last batch with nothing else on the GPU: device time of the codec kernels by themselves
            m.decode_output(out_last[0])
            iso_ms = m.codec_device_ms()
            if continuous:   # the isolated decode takes the first --batch utterances of the last call
                lo = out_last[0]
                out_last[0] = type(lo)(codes=lo.codes[:args.batch], num_steps=lo.num_steps, effective_lengths=lo.effective_lengths[:args.batch])
                m.decode_output(out_last[0])
                iso_ms = m.codec_device_ms()
            fl = synth.codec_flops_per_frame(ccfg) * sum(out_last[0].effective_lengths)
            peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
            ach = fl / (iso_ms * 1e-3) / 1e12
            # HBM-side bytes of the decode from the PMC counters (tools/dev/pmc_codec.sh: 16 x 375 frames, scaled by the frames
            # decoded here: every stage's traffic is proportional to them), quoted for the shapes the passes were taken on
            ctraffic = None
            cpath = os.path.join(ROOT, "bench_data", "codec_traffic.json")   # a copy lives under profiles/
            if os.path.exists(cpath) and args.shape == "S17" and args.dtype == "bf16":
                ct = json.load(open(cpath))
                ctraffic = round(ct["traffic_bytes"] * sum(out_last[0].effective_lengths) / float(ct["batch"] * ct["frames"]))
            codec = {"bound": "mfma", "kernel": "codec decoder, one batch alone on the GPU (conv / transposed-conv implicit GEMMs with fused epilogues)",
                     "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": ctraffic,
                     "device_ms_isolated": round(iso_ms, 2), "device_ms_per_step_overlapped": round(stats["codec_ms"] / args.steps, 2)}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline(args, cfg, ccfg, tmpl)
            except Exception as e:  # the baseline is a reported figure, not part of the product path
                cpu = {"value": None, "unit": "audio_s/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        ms_step = elapsed_max / args.steps * 1e3
        result = {
            "metric": "audio-sec generated/sec", "value": round(audio_total / elapsed_max, 2), "unit": "audio_s/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 2),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"{args.shape} Base-shaped random-init model, {args.utterances} voice-clone utterances in total per step "
                                    f"(3-10 s reference clips -> mel -> speaker encoder -> x-vector prompts), dealt to the ranks, decoded "
                                    f"in batches of {args.batch}, " if strong else
                                    f"{args.shape} CustomVoice-shaped random-init model, batch {args.batch} synthetic texts per GPU, ") +
                                   (f"{args.length_spread} frames each by utterance index" if args.length_spread else f"{args.frames} frames each") +
                                   f" ({'greedy' if args.greedy else 'seeded sampling'}, "
                                   f"{'EOS as sampled' if args.natural_eos else 'EOS suppressed up to that length'}), codec decode "
                                   f"{'off' if args.no_codec else 'to 24 kHz waveforms on the host'}" +
                                   (f"; continuous batching, {stats['calls']} calls, mean {rows_live:.1f} rows live per frame launch, "
                                    f"finished utterances decoded in groups of {args.codec_group}" +
                                    (f", queue order {args.queue_order}" if args.length_spread else "") if continuous else ""),
                       "batch_per_gpu": args.batch, "frames": args.frames, "fixed_length": not (args.natural_eos or args.length_spread),
                       "continuous_batching": continuous,
                       "utterances_total": args.utterances if strong else args.batch * world,
                       "parallelism": f"dp{world} (per-utterance)"},
            # first audio = the first `first_frames` frames through a streaming decode session (final samples), decoded and on the
            # host while generation continues; full utterance = the reference's own (non-streaming) definition
            "p50_first_audio_ms": round(float(np.median(stats["first_ms"])), 1) if stats["first_ms"] else None,
            "first_audio_frames": args.first_frames if stats["first_ms"] else None,
            "p50_full_utterance_ms": round(float(np.median(stats["step_ms"])), 1), "batches_in_flight": P,
            "phase_ms_per_step": {"prefill": round(stats["prefill_ms"] / args.steps, 2), "frames": round(stats["frames_ms"] / args.steps, 2),
                                  "codec_device": round(stats["codec_ms"] / args.steps, 2)},
            "ms_per_frame": round(ms_per_frame, 4),
            "graph_launch_host_ms_per_frame": round(stats["launch_host_ms"] / frames_run, 4),
            "roofline": roofline, "roofline_gemm": gemm, "roofline_codec": codec, "cpu_baseline": cpu,
            # per rank: [seconds in the timed region, audio seconds produced, utterances, decode frames]; numa_node: where rank 0 pinned itself
            "per_rank": [[round(v, 3) for v in r] for r in per_rank], "numa_node": pinned,
        }
    for mod in reversed(models):   # sessions before the model they borrow from
        mod.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
