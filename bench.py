#!/usr/bin/env python3
"""Benchmark of the segment-alignment hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): synthetic document pairs, N = M = 4096 segments, d = 1024,
bf16 candidate embeddings with 4 overlap layers per side (alignment_max_size 5 -> 10 alignment
types, band 14), reference-faithful coarse-to-fine search (dp_utils.vecalign semantics).  A "step"
aligns one batch of `--pairs` document pairs per GPU, inputs and sampled indices already resident
in HBM.  Document pairs are independent, so ranks share nothing: weak scaling, no collective.

Prints ONE JSON line: metric aligned doc-pairs/s (whole job), plus `roofline` for the dominant
kernel (HIP-event time on the launch stream) and `cpu_baseline` (the CPU oracle, 1 thread, on a
bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "speech-vecalign_amd"))

import numpy as np
import torch


def synth_pair_device(N, M, K, d, seed, dev, dtype):
    """Seeded synthetic pair in the reference's candidate layout, generated on the device:
    layer k row i = sum of base rows i-k..i (rows i < k zero); target = noisy copy of the source."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    L = max(N, M)
    base = torch.randn((L, d), generator=g, device=dev, dtype=torch.float32)
    tgt = base + 0.5 * torch.randn((L, d), generator=g, device=dev, dtype=torch.float32)

    def layers(b, n):
        cs = torch.cat([torch.zeros((1, d), device=dev, dtype=torch.float64), torch.cumsum(b[:n].double(), 0)])
        out = torch.zeros((K, n, d), device=dev, dtype=torch.float32)
        for k in range(K):
            out[k, k:] = (cs[k + 1:n + 1] - cs[:n - k]).float()
        return out.to(dtype).contiguous()

    return layers(base, N), layers(tgt, M)


def dp_cells(N, M, W, max_full, types):
    """DP node evaluations per pair: band nodes of every refined level + coarse dense nodes."""
    sizes = []
    s0, s1 = N, M
    while s0 * s1 > max_full * max_full:
        sizes.append((s0, s1))
        s0, s1 = s0 // 2, s1 // 2
    cells = (s0 + 1) * (s1 + 1)
    if not sizes:
        sizes = [(N, M)]
        cells += (N + M + 1 + 2) * 2 * W
    else:
        cells += sum((a + b + 3 + 2) * 2 * W for a, b in sizes)
    return cells


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=1024, help="document pairs per GPU per step")
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--overlaps", type=int, default=4)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--cpu_pairs", type=int, default=4, help="pairs timed on the CPU oracle (rank 0, N=1 only); 0 = skip")
    ap.add_argument("--no_profile", action="store_true")
    ap.add_argument("--streams", type=int, default=1, help="internal streams a batch is split over (svx_set_streams)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=dev)  # RCCL; used for the timing barrier / max only

    from svx import _lib
    from svx.vecalign import dp_utils

    N, M, K, d = args.n, args.m, args.overlaps, args.d
    a = K + 1
    types = [(x, y) for x in range(1, a) for y in range(1, a) if x + y <= a]
    W = int(np.ceil(K / 2.0)) + 5
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    esz = 4 if args.dtype == "f32" else 2
    P = args.pairs
    # Keep the step inside this GPU's free memory: a pair needs its inputs plus the scratch arena (measured 95 MB for
    # the 67 MB of bf16 inputs of the default workload); every rank uses the smallest count any rank can hold.
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    per_pair = K * (N + M) * d * esz + int(1.6 * K * (N + M) * d * 2) + (8 << 20)
    fit = max(1, int(0.92 * free_b) // per_pair)
    if dist is not None:
        tf = torch.tensor([fit], device=dev, dtype=torch.int64)
        dist.all_reduce(tf, op=dist.ReduceOp.MIN)
        fit = int(tf.item())
    if fit < P:
        print("bench: %d pairs per step do not fit the free HBM (%.0f GB); using %d" % (P, free_b / 1e9, fit), file=sys.stderr)
        P = fit
    docs = [synth_pair_device(N, M, K, d, 1000 * rank + i, dev, tdt) for i in range(P)]
    rngs = [np.random.RandomState(np.random.SeedSequence([2024, rank, i]).generate_state(4)) for i in range(P)]
    pb = dp_utils.PreparedBatch(docs, types, 0.2, W, 300, 20000, 100, rngs=rngs, device=local)
    ctx = pb.ctx
    lib = ctx.lib

    lib.svx_set_streams(ctx.h, args.streams)
    for _ in range(args.warmup):
        pb.run()
    torch.cuda.synchronize()
    first = pb.results() if args.warmup > 0 else None
    if not args.no_profile:
        lib.svx_set_profiling(ctx.h, 1)
    stage_names = ["pyr0", "pyr1", "pyrN", "pyr_aux", "knob_sort", "knob_scores0", "knob_scoresN", "knob", "dense_costs", "dense_dp",
                   "path", "band_costs0", "band_costsN", "band_dp0", "band_dpN", "traceback", "setup", "total", "host_plan",
                   "host_launch"]
    stage_ms = {s: 0.0 for s in stage_names}
    stage_launch = {s: 0 for s in stage_names}

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pb.run()
        if not args.no_profile:  # run() synchronised the stream to read its events
            for s in stage_names:
                stage_ms[s] += lib.svx_stage_ms(ctx.h, s.encode())
                stage_launch[s] += lib.svx_stage_launches(ctx.h, s.encode())
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    lib.svx_set_profiling(ctx.h, 0)
    res = pb.results()

    total_pairs = P * args.steps * world
    value = total_pairs / elapsed
    cells = dp_cells(N, M, W, 300, types)

    out = {
        "metric": "aligned doc-pairs/sec", "value": value, "unit": "doc-pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "synthetic %dx%d d=%d %s embeddings, %d overlap layers/side, %d alignment types, band %d, "
                               "coarse-to-fine (max_size_full_dp=300), %d doc-pairs per GPU per step" %
                               (N, M, d, args.dtype, K, len(types), 2 * W, P),
                   "pairs_per_step_per_gpu": P, "pairs_per_step_requested": args.pairs, "streams": args.streams, "N": N, "M": M, "d": d, "overlaps": K, "parallelism": "dp%d (pairs sharded, no collective)" % world},
        "dp_cells_per_s": value * cells, "dp_cells_per_pair": cells,
        "hbm_bytes_resident": {"inputs": int(sum(a.numel() * a.element_size() + b.numel() * b.element_size() for a, b in docs)),
                               "scratch_arena": int(lib.svx_scratch_bytes(ctx.h))},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (HIP events on the launch stream, timed region).
        # Algorithmic bytes = what the kernel must move once given its inputs and outputs;
        # the whole-path figure of SURVEY.md 8(d) is K(N+M)d*e = every candidate embedding read once.
        sizes = [(N, M)]
        while sizes[-1][0] * sizes[-1][1] > 300 * 300:
            sizes.append((sizes[-1][0] // 2, sizes[-1][1] // 2))
        L = len(sizes) - 1
        T = len(types)
        Bw = 2 * W
        row0 = d * esz
        alg = {
            # level 0 only reads the inputs (norms, column sums); level 1 re-reads them and forms its rows on the fly
            "pyr0": K * (N + M) * row0,
            "pyr1": (K * (N + M) * row0 + (sizes[1][0] + sizes[1][1]) * d * 4 +
                     (K * (sizes[2][0] + sizes[2][1]) * d * 4 if L >= 2 else 0)) if L >= 1 else 0,
            "pyrN": sum(K * (a + b) * d * 4 + (a + b) * d * 4 + (K * (sizes[l + 1][0] + sizes[l + 1][1]) * d * 4 if l < L else 0)
                        for l, (a, b) in enumerate(sizes) if l >= 2),
            "knob_scores0": (20000 + N) * row0,
            "knob_scoresN": sum((20000 + a) * d * 4 for l, (a, b) in enumerate(sizes) if 1 <= l < L),
            "band_costs0": K * (N + M) * row0 + T * (N + M + 3) * Bw * 4,
            "band_costsN": sum((a + b) * d * 4 + (a + b + 3) * Bw * 4 for l, (a, b) in enumerate(sizes) if 1 <= l < L),
            "band_dp0": (T * 4 + 9) * (N + M + 5) * Bw,
            "band_dpN": sum(13 * (a + b + 5) * Bw for l, (a, b) in enumerate(sizes) if 1 <= l < L),
        }
        alg_bytes_pair = K * (N + M) * row0
        if not args.no_profile and stage_ms["total"] > 0:
            # stages that are launches of the same kernel (template) are one entry; the dominant KERNEL is the
            # entry with the largest share of the step
            groups = {"k_pyramid": ["pyr0", "pyr1", "pyrN"], "k_band_costs_batch": ["band_costs0", "band_costsN"],
                      "k_knob_scores": ["knob_scores0", "knob_scoresN"], "k_sparse_dp_fast_batch": ["band_dp0", "band_dpN"]}
            gms = {g: sum(stage_ms[k] for k in ks) for g, ks in groups.items()}
            dom = max(gms, key=gms.get)
            launches = max(1, sum(stage_launch[k] for k in groups[dom]))
            avg_ms = gms[dom] / launches
            per_launch_bytes = sum(alg[k] for k in groups[dom]) * P * args.steps / launches
            achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
            # HBM bytes per launch from rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction of
            # MI355X_MICROARCH.md), measured per pair and per launch of every instantiation (profiles/r01_hbm_traffic.json),
            # weighted by the launches each instantiation has in a step and averaged like avg_ms
            traffic = None
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))["hbm_bytes_per_pair_per_launch"]
                if args.dtype == "bf16" and (N, M, K, d) == (4096, 4096, 4, 1024):
                    per_step = {"k_pyramid": [("k_pyramid<ElemBF16, 2, 1>", 1), ("k_pyramid<ElemBF16, 2, 2>", 1), ("k_pyramid<ElemF32, 4, 0>", L - 1)],
                                "k_band_costs_batch": [("k_band_costs_batch<ElemBF16, true, 12, 2>", 1), ("k_band_costs_batch<ElemF32, false, 6, 4>", L - 1)],
                                "k_knob_scores": [("k_knob_scores<ElemBF16, 2, true>", 1), ("k_knob_scores<ElemF32, 4, false>", 1)],
                                "k_sparse_dp_fast_batch": [("k_sparse_dp_fast_batch<3, 4>", 1), ("k_sparse_dp_fast_batch<1, 1>", L - 1)]}[dom]
                    traffic = sum(tj[k] * c for k, c in per_step) * P / sum(c for _, c in per_step)
            except Exception:
                traffic = None
            out["roofline"] = {"bound": "hbm", "kernel": dom, "stages": groups[dom], "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                               "frac": achieved / 8000.0, "traffic": traffic,
                               "avg_launch_ms": avg_ms, "launches": launches,
                               "algorithmic_bytes_per_launch": per_launch_bytes,
                               "share_of_step": gms[dom] / stage_ms["total"],
                               "whole_path_input_bytes_per_pair": alg_bytes_pair,
                               "whole_path_GBps": alg_bytes_pair * value / max(1, world) / 1e9}
            out["stage_ms_per_step"] = {k: v / args.steps for k, v in stage_ms.items()}
            out["kernel_GBps"] = {k: alg[k] * P * args.steps / (stage_ms[k] * 1e-3) / 1e9 for k in alg if stage_ms[k] > 0}
        # ---- CPU baseline: the oracle (restatement of the reference, pinned bit-exact against it) on one thread
        if world == 1 and args.cpu_pairs > 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle
            try:
                from threadpoolctl import threadpool_limits
            except Exception:
                threadpool_limits = None
            ncpu = min(args.cpu_pairs, P)
            hosts = [(docs[i][0].float().cpu().numpy(), docs[i][1].float().cpu().numpy()) for i in range(ncpu)]
            rr = [np.random.RandomState(np.random.SeedSequence([2024, rank, i]).generate_state(4)) for i in range(ncpu)]
            oracle.lib()

            def run_cpu():
                t = time.perf_counter()
                outs = [oracle.vecalign(h0, h1, types, 0.2, W, 300, 20000, 100, rng=r) for (h0, h1), r in zip(hosts, rr)]
                return time.perf_counter() - t, outs
            if threadpool_limits is not None:
                with threadpool_limits(limits=1):
                    cpu_s, cpu_out = run_cpu()
            else:
                cpu_s, cpu_out = run_cpu()
            out["cpu_baseline"] = {"value": ncpu / cpu_s, "unit": "doc-pairs/s", "cores": 1, "kind": "port",
                                   "sample": "%d of the same %dx%d pairs, oracle/oracle.py vecalign() (C restatement of "
                                             "dp_core.pyx + numpy), 1 thread, %.1f s" % (ncpu, N, M, cpu_s)}
            same, worst = True, 0.0
            for i in range(ncpu):
                same = same and (cpu_out[i][0]['final_alignments'] == res[i][0])
                if len(cpu_out[i][0]['alignment_scores']) == len(res[i][1]):
                    worst = max(worst, float(np.abs(cpu_out[i][0]['alignment_scores'] - res[i][1]).max()))
            out["parity"] = {"pairs_checked": ncpu, "spans_identical": bool(same), "max_score_diff": worst}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
