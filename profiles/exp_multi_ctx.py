#!/usr/bin/env python3
"""Experiment (round 3): does the latency-bound chain of one sub-batch hide under the streaming passes of another
when the sub-batches are INDEPENDENT pipelines (own context, own arena, own stream, no join between steps)?

    python profiles/exp_multi_ctx.py --pairs 1024 --parts 2 --steps 6 [--prio] [--stagger_ms 30]

One JSON line: pairs/s with `parts` contexts on `parts` streams against the single-stream figure of the same run.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-vecalign_amd"))
sys.path.insert(0, ROOT)

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1024)
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prio", action="store_true", help="descending stream priorities")
    ap.add_argument("--stagger_ms", type=float, default=0.0, help="host sleep between the parts' first launches")
    ap.add_argument("--skip_single", action="store_true")
    args = ap.parse_args()
    import torch
    from bench import synth_pairs_device, alignment_types
    from svx import _lib
    from svx.vecalign import dp_utils
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    N = M = 4096
    K, d = 4, 1024
    types = alignment_types(K + 1)
    W = int(np.ceil(K / 2.0)) + 5
    P = args.pairs
    docs = []
    for i in range(0, P, 16):
        docs += synth_pairs_device(N, M, K, d, list(range(i, min(P, i + 16))), dev, torch.bfloat16)
    rngs = [np.random.RandomState(np.random.SeedSequence([2024, 0, i]).generate_state(4)) for i in range(P)]
    out = {"pairs": P, "parts": args.parts, "steps": args.steps, "prio": args.prio}

    if not args.skip_single:
        pb = dp_utils.PreparedBatch(docs, types, 0.2, W, 300, 20000, 100, rngs=rngs, device=0)
        for _ in range(args.warmup):
            pb.run()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(args.steps):
            pb.run()
        torch.cuda.synchronize()
        el = time.perf_counter() - t
        out["single"] = {"ms_per_step": 1e3 * el / args.steps, "pairs_per_s": P * args.steps / el}
        ref_info = pb.raw_results()[0].copy()
        del pb
        _lib._ctxs.clear()
        torch.cuda.empty_cache()

    S = args.parts
    lo_hi = [(P * s // S, P * (s + 1) // S) for s in range(S)]
    pbs, streams = [], []
    for s, (lo, hi) in enumerate(lo_hi):
        _lib._ctxs.clear()   # a fresh context (own arena, own side stream) per part
        pbs.append(dp_utils.PreparedBatch(docs[lo:hi], types, 0.2, W, 300, 20000, 100, rngs=rngs[lo:hi], device=0))
        streams.append(torch.cuda.Stream(device=dev, priority=(-s if args.prio else 0)))
    torch.cuda.synchronize()

    def step():
        for pb, st in zip(pbs, streams):
            with torch.cuda.stream(st):
                pb.run()

    for w in range(args.warmup):
        if w == 0 and args.stagger_ms > 0:
            for pb, st in zip(pbs, streams):
                with torch.cuda.stream(st):
                    pb.run()
                time.sleep(args.stagger_ms * 1e-3)
        else:
            step()
    if args.stagger_ms <= 0:
        torch.cuda.synchronize()
    starts, ends = [], []
    for st in streams:
        e = torch.cuda.Event(enable_timing=True)
        e.record(st)
        starts.append(e)
    t = time.perf_counter()
    for _ in range(args.steps):
        step()
    for st in streams:
        e = torch.cuda.Event(enable_timing=True)
        e.record(st)
        ends.append(e)
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    dev_ms = max(starts[0].elapsed_time(e) for e in ends)   # from stream 0's start to the last stream's end (includes the stagger: conservative)
    out["multi"] = {"ms_per_step": dev_ms / args.steps, "pairs_per_s": P * args.steps / (dev_ms * 1e-3), "host_wall_ms_per_step": 1e3 * el / args.steps,
                    "per_stream_ms_per_step": [starts[i].elapsed_time(ends[i]) / args.steps for i in range(S)]}
    info = np.concatenate([pb.raw_results()[0] for pb in pbs])
    out["alignments_total"] = int(info[:, 0].sum())
    if not args.skip_single:
        out["same_counts_as_single"] = bool((info == ref_info).all())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
