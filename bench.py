#!/usr/bin/env python3
"""Benchmark of the segment-alignment hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: this script starts its own N ranks)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...     (also works)

Workload `c2` (default; BASELINE.json configs[1], the configuration the metric is quoted on): synthetic document
pairs, N = M = 4096 segments, d = 1024, bf16 candidate embeddings with 4 overlap layers per side
(alignment_max_size 5 -> 10 alignment types, band 14), reference-faithful coarse-to-fine search (dp_utils.vecalign
semantics).  A "step" aligns one batch of `--pairs` document pairs per GPU, inputs and sampled indices already
resident in HBM.  Workload `c3` (configs[2]): `--pairs` ragged pairs per GPU, N, M ~ U{512..8192}.  Document pairs
are independent, so ranks share nothing: weak scaling, no data-path collective (RCCL only for the timing barrier).

Prints ONE JSON line:
  value           aligned doc-pairs/s, whole job, inputs resident in HBM;
  roofline        the dominant kernel (by rocprof symbol = one stage timer) against HBM peak.  `achieved` follows
                  SURVEY.md 8(d): compulsory bytes K(N+M)d*e per pair x pairs per launch / the kernel's average
                  launch time (HIP events on the launch stream, timed region).  `own_pass` is the same kernel
                  priced with the bytes ITS pass must move once, `whole_path` prices the
                  whole step with the compulsory bytes, `traffic` is the PMC figure of the committed rocprofv3 run
                  at this very configuration (null when there is none);
  stages          per-stage time, bytes and rate; gathers served by L2 / Infinity Cache are labelled as such;
  end_to_end      fresh pairs and fresh indices every step with the inputs starting (a) in pinned host memory,
                  (b) in files on disk through the seg_align CLI path (PCIe-bound: 67 MB per pair);
  cpu_baseline    the CPU oracle on a bounded sample of the same workload: 1 thread (the reference's execution
                  model) and, under `all_cores`, P = os.cpu_count() independent processes.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "speech-vecalign_amd"))

import numpy as np

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md
CPU_CHILDREN = []        # Popen objects of the all-cores CPU leg (idle until the GPU part is over)


def alignment_types(a):
    return [(x, y) for x in range(1, a) for y in range(1, a) if x + y <= a]


# ------------------------------------------------------------------------------------------ CPU worker (no GPU)
def cpu_worker(argv):
    """One of the P processes of the all-cores CPU leg: builds its own synthetic pairs (tests/synth.py, the same
    generator family as the GPU side), waits for 'go', aligns them with the oracle, prints the seconds it took."""
    seed, count, N, M, K, d = (int(v) for v in argv)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass
    import oracle
    from synth import make_pair, round_bf16
    oracle.lib()
    assert sys.stdin.readline().strip() == "prep"
    types = alignment_types(K + 1)
    W = int(np.ceil(K / 2.0)) + 5
    docs = []
    for i in range(count):
        v0, v1 = make_pair(N, M, K, d, 100000 + 1000 * seed + i)
        docs.append((round_bf16(v0), round_bf16(v1)))
    print("ready", flush=True)
    assert sys.stdin.readline().strip() == "go"
    t = time.perf_counter()
    for i, (v0, v1) in enumerate(docs):
        oracle.vecalign(v0, v1, types, 0.2, W, 300, 20000, 100, rng=np.random.RandomState(seed * 100 + i))
    print("%.6f" % (time.perf_counter() - t), flush=True)


# ------------------------------------------------------------------------------------------ synthetic inputs
def synth_pairs_device(N, M, K, d, seeds, dev, dtype):
    """Seeded synthetic pairs in the reference's candidate layout, generated on the device, len(seeds) pairs per
    call (few large kernels instead of thousands of small ones): layer k row i = sum of base rows i-k..i
    (rows i < k zero); target = noisy copy of the source.  -> list of (vecs0 [K][N][d], vecs1 [K][M][d])."""
    import torch
    P, L = len(seeds), max(N, M)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seeds[0]) * 1000003 + P)
    base = torch.randn((P, L, d), generator=g, device=dev, dtype=torch.float32)
    tgt = base + 0.5 * torch.randn((P, L, d), generator=g, device=dev, dtype=torch.float32)

    def layers(b, n):
        cs = torch.cat([torch.zeros((P, 1, d), device=dev, dtype=torch.float64), torch.cumsum(b[:, :n].double(), 1)], dim=1)
        out = torch.zeros((P, K, n, d), device=dev, dtype=dtype)
        for k in range(K):
            out[:, k, k:] = (cs[:, k + 1:n + 1] - cs[:, :n - k]).float().to(dtype)
        return out

    a, b = layers(base, N), layers(tgt, M)
    return [(a[i], b[i]) for i in range(P)]


def synth_pair_device(N, M, K, d, seed, dev, dtype):
    return synth_pairs_device(N, M, K, d, [seed], dev, dtype)[0]


def level_sizes(N, M, max_full=300):
    sizes = [(N, M)]
    while sizes[-1][0] * sizes[-1][1] > max_full * max_full:
        sizes.append((sizes[-1][0] // 2, sizes[-1][1] // 2))
    return sizes


def dp_cells(N, M, W, max_full=300):
    """DP node evaluations per pair: band nodes of every refined level + coarse dense nodes."""
    sizes = level_sizes(N, M, max_full)
    s0, s1 = sizes[-1]
    cells = (s0 + 1) * (s1 + 1)
    if len(sizes) == 1:
        cells += (N + M + 1 + 2) * 2 * W
    else:
        cells += sum((a + b + 3 + 2) * 2 * W for a, b in sizes[:-1])
    return cells


def stage_bytes(N, M, K, d, esz, T, W):
    """Bytes each streaming stage must move once per pair, given its inputs and outputs,
    and what serves them.  The gathers of the sampled scores re-read rows that sit in L2 / Infinity Cache."""
    sizes = level_sizes(N, M)
    L = len(sizes) - 1
    Bw, row0 = 2 * W, d * esz
    lv = lambda l: sizes[l][0] + sizes[l][1]
    out = {
        "pyr0": (K * (N + M) * row0, "hbm"),
        "band_costs0": (K * (N + M) * row0 + T * (N + M + 3) * Bw * 4, "hbm"),
        "knob_scores0": ((20000 + N) * row0, "l2/infinity-cache gather"),
        "band_dp0": ((T * 4 + 9) * (N + M + 5) * Bw, "latency (serial chain)"),
    }
    if L >= 1:
        out["pyr1"] = (K * (N + M) * row0 + lv(1) * d * 4 + (K * lv(2) * d * 4 if L >= 2 else 0), "hbm")
        out["pyrN"] = (sum(K * lv(l) * d * 4 + lv(l) * d * 4 + (K * lv(l + 1) * d * 4 if l < L else 0) for l in range(2, L + 1)), "hbm")
        out["knob_scoresN"] = (sum((20000 + sizes[l][0]) * d * 4 for l in range(1, L)), "l2/infinity-cache gather")
        out["band_costsN"] = (sum(lv(l) * d * 4 + (lv(l) + 3) * Bw * 4 for l in range(1, L)), "hbm")
        out["band_dpN"] = (sum(13 * (lv(l) + 5) * Bw for l in range(1, L)), "latency (serial chain)")
    return out


# rocprof symbol of each timed stage at the default workload (one stage = launches of one kernel instantiation)
STAGE_KERNEL = {
    "pyr0": "k_pyramid<E,NCH,1,FULL> (level 0)", "pyr1": "k_pyramid<E,NCH,2,FULL> (level 1)", "pyrN": "k_pyramid<ElemF32,NCH,0,FULL> (levels >= 2)",
    "knob_scores0": "k_knob_scores<E,NCH,true>", "knob_scoresN": "k_knob_scores<ElemF32,NCH,false>",
    "band_costs0": "k_band_costs (level 0)", "band_costsN": "k_band_costs (levels >= 1)",
    "band_dp0": "k_sparse_dp_fast_batch (level 0)", "band_dpN": "k_sparse_dp_fast_batch (levels >= 1)",
    "traceback": "k_sparse_traceback_batch (levels >= 1)", "traceback0": "k_sparse_traceback_batch (level 0)", "path0": "k_search_path_batch (level 0)", "knob_sort": "k_knob_sort", "knob": "k_del_penalty_batch",
    "dense_costs": "k_dense_costs_batch", "dense_dp": "k_dense_stage_batch", "path": "k_search_path_batch (levels >= 1)", "pyr_aux": "k_colmean + k_sample_mean",
}


# ------------------------------------------------------------------------------------------ end-to-end legs
def e2e_host_memory(args, dev, types, W, tdt, K, N, M, d):
    """Fresh pairs and fresh sampled indices every step, inputs starting in pinned host memory: uploads of
    sub-batch i+1 (copy stream) overlap the alignment of sub-batch i; results come back to pinned memory."""
    import torch
    from svx.vecalign import dp_utils
    pool_n, sub = args.e2e_pairs, max(1, min(args.e2e_batch, args.e2e_pairs))
    host = []
    made = []
    for i in range(0, pool_n, 16):
        made += synth_pairs_device(N, M, K, d, [50000 + j for j in range(i, min(pool_n, i + 16))], dev, tdt)
    for a, b in made:
        ha = torch.empty(a.shape, dtype=a.dtype, pin_memory=True)
        hb = torch.empty(b.shape, dtype=b.dtype, pin_memory=True)
        ha.copy_(a)
        hb.copy_(b)
        host.append((ha, hb))
    torch.cuda.synchronize()
    del made
    copy_stream = torch.cuda.Stream(device=dev)
    compute = torch.cuda.current_stream(dev)
    bytes_pair = sum(x.numel() * x.element_size() for x in host[0])

    def one_pass(step):
        jobs = []
        prev = None
        for b0 in range(0, pool_n, sub):
            chunk = host[b0:b0 + sub]
            with torch.cuda.stream(copy_stream):
                devs = [(a.to(dev, non_blocking=True), b.to(dev, non_blocking=True)) for a, b in chunk]
                up = torch.cuda.Event()
                up.record(copy_stream)
            compute.wait_event(up)
            for a, b in devs:
                a.record_stream(compute)
                b.record_stream(compute)
            rngs = [np.random.RandomState(np.random.SeedSequence([777, step, b0 + i]).generate_state(4)) for i in range(len(chunk))]
            pb = dp_utils.PreparedBatch(devs, types, 0.2, W, 300, 20000, 100, rngs=rngs, device=dev.index)
            pb.run()
            ev = pb.fetch_async()
            if prev is not None:
                prev[1].synchronize()
                jobs.append(prev[0].raw_results()[0][:, 0].sum())
            prev = (pb, ev)
        prev[1].synchronize()
        jobs.append(prev[0].raw_results()[0][:, 0].sum())
        return int(sum(jobs))

    one_pass(0)  # warm-up (arena growth, pinned pools)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n_align = 0
    for s in range(args.e2e_steps):
        n_align += one_pass(1 + s)
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    pairs = pool_n * args.e2e_steps
    return {"value": pairs / el, "unit": "doc-pairs/s", "pairs": pairs, "sub_batch": sub, "seconds": el,
            "pcie_GBps": pairs * bytes_pair / el / 1e9, "alignments": n_align,
            "what": "inputs in pinned host memory, fresh indices drawn every step (native MT19937), H2D on a copy stream "
                    "overlapping compute, results to pinned memory; PCIe-bound (%.0f MB per pair)" % (bytes_pair / 1e6)}


def write_synthetic_files(root, n_pairs, N, M, K, d, dev):
    """Synthetic documents in the reference's file formats (README.md:174-260): segment files 'start end',
    candidate files (string-sorted like concat_segs.py:118), raw fp16 .embed files in candidate order."""
    import torch
    meta = []
    for sub in ("seg", "cat", "emb"):
        for lang in ("en", "de"):
            os.makedirs(os.path.join(root, sub, lang), exist_ok=True)
    for p in range(n_pairs):
        docs = synth_pair_device(N, M, K, d, 90000 + p, dev, torch.float16)
        for lang, v in zip(("en", "de"), docs):
            n = v.shape[1]
            starts = np.arange(n, dtype=np.int64) * 1000
            ends = starts + 800
            cand = [(k, i) for i in range(n) for k in range(K) if i - k >= 0]       # V[k][i] = segments i-k .. i
            keys = ["%d %d" % (starts[i - k], ends[i]) for k, i in cand]
            order = sorted(range(len(keys)), key=keys.__getitem__)
            ks = torch.tensor([cand[o][0] for o in order], device=dev)
            iis = torch.tensor([cand[o][1] for o in order], device=dev)
            rows = v[ks, iis].contiguous().cpu().numpy()
            stem = "doc%04d_%s" % (p, lang)
            with open(os.path.join(root, "seg", lang, stem + ".txt"), "w") as f:
                f.write("".join("%d %d\n" % (s, e) for s, e in zip(starts, ends)))
            with open(os.path.join(root, "cat", lang, stem + ".txt"), "w") as f:
                f.write("".join(keys[o] + "\n" for o in order))
            rows.tofile(os.path.join(root, "emb", lang, stem + ".embed"))
        meta.append("/audio/doc%04d_en.ogg\t/audio/doc%04d_de.ogg" % (p, p))
    with open(os.path.join(root, "metadata.tsv"), "w") as f:
        f.write("\n".join(meta) + "\n")


def e2e_files(args, dev, K, N, M, d):
    """Files on disk -> alignment files through svx.seg_align.align (the reference's CLI surface): native table
    building and pinned reads on host threads, device gather, svx_align_batch, native formatting, writer threads."""
    import shutil
    from svx.seg_align import align as A
    root = tempfile.mkdtemp(prefix="svx_bench_", dir=os.environ.get("SVX_TMPDIR", tempfile.gettempdir()))
    try:
        t = time.perf_counter()
        write_synthetic_files(root, args.e2e_files, N, M, K, d, dev)
        gen_s = time.perf_counter() - t
        base = [os.path.join(root, "metadata.tsv"), None, "--src_lang", "en", "--tgt_lang", "de", "--seg_dir", os.path.join(root, "seg"),
                "--concat_dir", os.path.join(root, "cat"), "--embed_dir", os.path.join(root, "emb"), "--fp16_embed",
                "-a", str(K + 1), "--seed", "5", "--batch_size", str(args.e2e_batch)]
        times = []
        for rep in range(1 + args.e2e_steps):
            argv = list(base)
            argv[1] = os.path.join(root, "out%d" % rep)
            t = time.perf_counter()
            A.main(argv)
            times.append(time.perf_counter() - t)
        nbytes = sum(os.path.getsize(os.path.join(root, "emb", l, f)) for l in ("en", "de") for f in os.listdir(os.path.join(root, "emb", l)))
        best = min(times[1:])
        out_ok = len(os.listdir(os.path.join(root, "out1", "en-de"))) == args.e2e_files
        return {"value": args.e2e_files / best, "unit": "doc-pairs/s", "pairs": args.e2e_files, "seconds_per_pass": times[1:],
                "first_pass_seconds": times[0], "file_GBps": nbytes / best / 1e9, "outputs_written": bool(out_ok),
                "generation_seconds": gen_s, "io_threads": min(32, os.cpu_count() or 4),
                "what": "svx.seg_align.align on %d synthetic %dx%d file sets (page cache warm after the first pass): "
                        "files -> alignment files" % (args.e2e_files, N, M)}
    finally:
        shutil.rmtree(root, ignore_errors=True)


# ------------------------------------------------------------------------------------------ straight-band workloads
def init_ranks(torch, local):
    """-> (device, dist or None, device the timing reductions run on).  One process per GPU over RCCL ("nccl").
    SVX_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: the ranks share the visible
    GPUs round-robin and the barrier / MAX / SUM of the timing run over gloo on the host (no RCCL between two
    ranks of one device); the alignment path itself has no collective either way."""
    world = int(os.environ.get("WORLD_SIZE", 1))
    backend = os.environ.get("SVX_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world <= 1:
        return dev, None, dev
    import torch.distributed as dist
    if backend == "gloo":
        dist.init_process_group("gloo")
        return dev, dist, torch.device("cpu")
    dist.init_process_group("nccl", device_id=dev)  # RCCL; used for the timing barrier / max only
    return dev, dist, dev


def bench_straight(args):
    """BASELINE configs[3] (`c4`: N = M = 32768, Sakoe-Chiba band 2048 around the straight diagonal, one pair per
    step) and the dense reading of configs[1] (`dense`: N = M = 4096, every cell of the lattice, a batch of pairs per
    step): SVX_SEARCH_STRAIGHT -- MFMA cost tiles feeding the DP as a wavefront of 32 x 32 tiles over all CUs.
    The dominant kernel is the tile sweep; its roofline is the dense 16-bit MFMA peak (T * nodes * 2d flops)."""
    import torch
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    dev, dist, rdev = init_ranks(torch, local)
    local = dev.index
    from svx import _lib
    from svx.vecalign import dp_utils
    _lib.context(local).set_pipeline(False)   # (the tile sweep never pipelines; narrow straight bands are not benchmarked here)
    K, d = args.overlaps, args.d
    types = alignment_types(K + 1)
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    if args.workload == "c4":
        N = M = 32768 if (args.n, args.m) == (4096, 4096) else args.n
        W, P = args.band // 2, (args.pairs or 1)
    else:
        N, M = args.n, args.m
        W, P = max(N, M) + 1, (args.pairs or 64)
    docs = []
    for i in range(0, P, 4):
        docs += synth_pairs_device(N, M, K, d, [300 + 1000 * rank + j for j in range(i, min(P, i + 4))], dev, tdt)
    rngs = [np.random.RandomState(np.random.SeedSequence([4242, rank, i]).generate_state(4)) for i in range(P)]
    pb = dp_utils.PreparedBatch(docs, types, 0.2, W, 300, 20000, 100, rngs=rngs, device=local, search="straight")
    ctx, lib = pb.ctx, pb.ctx.lib
    for _ in range(args.warmup):
        pb.run()
    torch.cuda.synchronize()
    lib.svx_set_profiling(ctx.h, 2)   # accumulate: events around every stage, read once after the timed steps
    names = ["pyr0", "pyr_aux", "knob_sort", "knob_scores0", "knob", "path0", "tiles", "traceback0", "setup", "total"]
    ms = {k: 0.0 for k in names}
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pb.run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    for k in names:
        ms[k] = lib.svx_stage_ms(ctx.h, k.encode())
    if dist is not None:
        tt = torch.tensor([elapsed], device=rdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    lib.svx_set_profiling(ctx.h, 0)
    res = pb.results()
    ok = all([x for al in r[0] for x in al[0]] == list(range(N)) and [y for al in r[0] for y in al[1]] == list(range(M)) for r in res)
    B = 2 * W
    # DP nodes inside the band and the lattice: sum over node diagonals of the overlap of [bo, bo + B) with the lattice
    nodes = 0
    for a in range(N + M + 1):
        yc = a * M // (N + M)
        ylo, yhi = max(yc - W, 0, a - N), min(yc + W - 1, M, a)
        nodes += max(0, yhi - ylo + 1)
    flops = float(len(types)) * nodes * 2 * d
    value = P * args.steps * world / elapsed
    tile_ms = ms["tiles"] / args.steps
    out = {"metric": "aligned doc-pairs/sec", "value": value, "unit": "doc-pairs/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": ("synthetic %dx%d d=%d %s, %d overlap layers/side, %d alignment types, " % (N, M, d, args.dtype, K, len(types))) +
                                  ("Sakoe-Chiba band of %d cells around the straight diagonal" % B if args.workload == "c4" else
                                   "dense search (band %d covers the whole lattice)" % B) + ", %d doc-pairs per GPU per step" % P,
                      "name": args.workload, "pairs_per_step_per_gpu": P, "N": N, "M": M, "d": d, "overlaps": K, "band": B,
                      "parallelism": "dp%d (pairs sharded, no collective)" % world},
           "seconds_per_pair": elapsed / (P * args.steps), "dp_cells_per_pair": nodes, "dp_cells_per_s": value * nodes,
           "alignments_cover_both_documents": bool(ok),
           "stage_ms_per_step": {k: v / args.steps for k, v in ms.items()},
           "roofline": {"bound": "mfma", "kernel": "k_band_tiles (cost tiles + DP wavefront)", "stage": "tiles",
                        "achieved": flops * P / (tile_ms * 1e-3) / 1e12 if tile_ms > 0 else None, "peak": 2500.0, "unit": "TFLOP/s",
                        "frac": (flops * P / (tile_ms * 1e-3) / 1e12 / 2500.0) if tile_ms > 0 else None, "traffic": None,
                        "traffic_source": "no counter pass committed for this mode (the tile sweep re-reads its rows out of L2: DESIGN.md section 6)",
                        "avg_launch_ms": tile_ms, "algorithmic_flops_per_launch": flops * P,
                        "note": "the sweep is bound by the serial chain of tile anti-diagonals (%d of them) and the float64 DP inside a tile, "
                                "not by the matrix cores" % ((N // 32 + 1) + (M // 32 + 1) - 1)}}
    del pb, docs
    torch.cuda.empty_cache()
    return out if rank == 0 else None


# ------------------------------------------------------------------------------------------ main
def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1, not under torchrun): start N ranks as children BEFORE this process touches
    the GPU, relay rank 0's JSON line, exit with the children's code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu_worker":
        return cpu_worker(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)   # (the pipeline fills in the first step and drains after the last: amortised over the run)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "dense"])
    ap.add_argument("--band", type=int, default=2048, help="c4: band width (cells per diagonal) around the straight diagonal")
    ap.add_argument("--pairs", type=int, default=None, help="document pairs per GPU per step (default 1024: BASELINE configs[1] and [2])")
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--overlaps", type=int, default=4)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--cpu_pairs", type=int, default=4, help="pairs timed on the CPU oracle, 1 thread (rank 0, N=1 only); 0 = skip")
    ap.add_argument("--cpu_procs", type=int, default=-1, help="processes of the all-cores CPU leg (-1 = os.cpu_count(), 0 = skip)")
    ap.add_argument("--cpu_pairs_per_proc", type=int, default=2)
    ap.add_argument("--e2e_pairs", type=int, default=128, help="pool of pairs in pinned host memory for the end-to-end leg; 0 = skip")
    ap.add_argument("--e2e_batch", type=int, default=32)
    ap.add_argument("--e2e_steps", type=int, default=2)
    ap.add_argument("--e2e_files", type=int, default=64, help="document pairs written to disk for the files leg; 0 = skip")
    ap.add_argument("--no_profile", action="store_true")
    ap.add_argument("--extra_workloads", type=int, default=1, help="default c2 run on one GPU: also time bounded legs of c3, c4 (1 / 4 / 8 pairs "
                    "per step) and the dense mode, reported under `workloads`; 0 = skip")
    ap.add_argument("--pipeline", type=int, default=1, help="software pipeline over consecutive steps (svx_set_pipeline): the latency-bound "
                    "refinement chain of one half-batch runs beside the streaming passes of the other; 0 = every step runs start to end on one stream")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    out = bench_straight(args) if args.workload in ("c4", "dense") else bench_ctf(args)
    if out is not None:
        if args.extra_workloads and args.workload == "c2" and int(os.environ.get("WORLD_SIZE", 1)) == 1:
            out["workloads"] = extra_workloads(args)
        print(json.dumps(out))
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def extra_workloads(args):
    """The other BASELINE configurations and the dense reading of configs[1], each as a bounded leg of the default run
    (same process, after the c2 legs have released their memory): value + roofline object per leg."""
    import copy
    legs = {}

    def leg(name, **kw):
        a = copy.copy(args)
        a.cpu_pairs, a.cpu_procs, a.e2e_pairs, a.e2e_files, a.extra_workloads = 0, 0, 0, 0, 0
        for k, v in kw.items():
            setattr(a, k, v)
        t = time.perf_counter()
        try:
            r = bench_straight(a) if a.workload in ("c4", "dense") else bench_ctf(a)
            keep = ("value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline", "seconds_per_pair", "dp_cells_per_s",
                    "stage_ms_per_step", "alignments_cover_both_documents")
            legs[name] = {k: r[k] for k in keep if k in r}
            legs[name]["leg_seconds"] = time.perf_counter() - t
        except Exception as e:  # never lose the headline line to a side leg
            legs[name] = {"error": repr(e)}
    leg("c3", workload="c3", pairs=1024, steps=5, warmup=1)
    leg("c4_1pair", workload="c4", pairs=1, steps=2, warmup=1)
    leg("c4_4pairs", workload="c4", pairs=4, steps=2, warmup=1)
    leg("c4_8pairs", workload="c4", pairs=8, steps=1, warmup=1)
    leg("dense", workload="dense", pairs=64, steps=1, warmup=1)
    return legs


def bench_ctf(args):
    """Workloads c2 / c3: the reference's coarse-to-fine search.  -> the result dict on rank 0, None elsewhere."""
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    N, M, K, d = args.n, args.m, args.overlaps, args.d

    # the all-cores CPU leg runs in child processes started before this process initialises the GPU; they idle
    # until the GPU part is over
    global CPU_CHILDREN
    cpu_children = CPU_CHILDREN
    try:
        host_cores = len(os.sched_getaffinity(0))
    except Exception:
        host_cores = os.cpu_count() or 1
    ncpu_procs = min(host_cores, 64) if args.cpu_procs < 0 else args.cpu_procs
    if rank == 0 and world == 1 and ncpu_procs > 0 and args.workload == "c2":
        for w in range(ncpu_procs):
            cpu_children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu_worker", str(w), str(args.cpu_pairs_per_proc),
                                                  str(N), str(M), str(K), str(d)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))

    import torch
    if not torch.cuda.is_available():
        for c in cpu_children:
            c.kill()
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    dev, dist, rdev = init_ranks(torch, local)
    local = dev.index

    from svx import _lib
    from svx.utils.mp_utils import balanced_shards
    from svx.vecalign import dp_utils

    a = K + 1
    types = alignment_types(a)
    W = int(np.ceil(K / 2.0)) + 5
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    esz = 4 if args.dtype == "f32" else 2
    P = args.pairs if args.pairs is not None else 1024
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    if args.workload == "c2":
        shapes = [(N, M)] * P
        # keep the step inside this GPU's free memory: inputs + scratch arena (measured 1.45x the bf16 inputs)
        per_pair = K * (N + M) * d * esz + int(1.6 * K * (N + M) * d * 2) + (8 << 20)
        fit = max(1, int(0.92 * free_b) // per_pair)
        if dist is not None:
            tf = torch.tensor([fit], device=rdev, dtype=torch.int64)
            dist.all_reduce(tf, op=dist.ReduceOp.MIN)
            fit = int(tf.item())
        if fit < P:
            print("bench: %d pairs per step do not fit the free HBM (%.0f GB); using %d" % (P, free_b / 1e9, fit), file=sys.stderr)
            P = fit
            shapes = [(N, M)] * P
        seeds = [1000 * rank + i for i in range(P)]
    else:
        # configs[2]: P * world ragged pairs, N, M ~ U{512..8192} (seed 1), dealt to the ranks by longest-processing-time on N + M
        rs = np.random.RandomState(1)
        allshapes = [(int(rs.randint(512, 8193)), int(rs.randint(512, 8193))) for _ in range(P * world)]
        mine = balanced_shards([n + m for n, m in allshapes], world)[rank]
        shapes = [allshapes[i] for i in mine]
        seeds = [5000 + i for i in mine]
    if args.workload == "c2":
        docs = []
        for i in range(0, len(seeds), 16):
            docs += synth_pairs_device(N, M, K, d, seeds[i:i + 16], dev, tdt)
    else:
        docs = [synth_pair_device(n, m, K, d, s, dev, tdt) for (n, m), s in zip(shapes, seeds)]
    rngs = [np.random.RandomState(np.random.SeedSequence([2024, rank, i]).generate_state(4)) for i in range(len(docs))]
    pb = dp_utils.PreparedBatch(docs, types, 0.2, W, 300, 20000, 100, rngs=rngs, device=local)
    ctx = pb.ctx
    lib = ctx.lib

    ctx.set_pipeline(bool(args.pipeline))
    for _ in range(args.warmup):
        pb.run()
    pb.flush()
    torch.cuda.synchronize()
    if not args.no_profile:
        # accumulating mode: HIP events around every stage of every timed step, read once after the loop -- the steps
        # queue behind one another as they do without profiling (mode 1 synchronises after every call to read them)
        lib.svx_set_profiling(ctx.h, 2)
    stage_names = ["pyr0", "pyr1", "pyrN", "pyr_aux", "knob_sort", "knob_scores0", "knob_scoresN", "knob", "dense_costs", "dense_dp",
                   "path", "path0", "band_costs0", "band_costsN", "band_dp0", "band_dpN", "traceback", "traceback0", "setup", "total", "host_plan",
                   "host_launch"]
    stage_ms = {s: 0.0 for s in stage_names}
    stage_launch = {s: 0 for s in stage_names}

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pb.run()
    pb.flush()   # (the chain the last step's pipeline held back: inside the timed region)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if not args.no_profile:  # totals over the timed steps
        for s in stage_names:
            stage_ms[s] = lib.svx_stage_ms(ctx.h, s.encode())
            stage_launch[s] = lib.svx_stage_launches(ctx.h, s.encode())
    npairs_local = len(docs)
    total_pairs = npairs_local * args.steps
    if dist is not None:
        tt = torch.tensor([elapsed], device=rdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tp = torch.tensor([total_pairs], device=rdev, dtype=torch.int64)
        dist.all_reduce(tp, op=dist.ReduceOp.SUM)
        total_pairs = int(tp.item())
    lib.svx_set_profiling(ctx.h, 0)
    ctx.set_pipeline(False)   # (the legs below -- CPU parity sample, end-to-end, other workloads -- start from the default state)
    res = None if os.environ.get("SVX_BENCH_NOCHECK") else pb.results()   # (NOCHECK: timing experiments with deliberately wrong kernels)

    value = total_pairs / elapsed
    cells = float(np.mean([dp_cells(n, m, W) for n, m in shapes]))
    wl = ("synthetic %dx%d d=%d %s embeddings, %d overlap layers/side, %d alignment types, band %d, coarse-to-fine "
          "(max_size_full_dp=300), %d doc-pairs per GPU per step" % (N, M, d, args.dtype, K, len(types), 2 * W, npairs_local)) if args.workload == "c2" else \
         ("%d ragged synthetic doc-pairs per GPU per step, N, M ~ U{512..8192} (seed 1), d=%d %s, %d overlap layers/side, "
          "band %d, pairs dealt to ranks by longest-processing-time on N+M" % (npairs_local, d, args.dtype, K, 2 * W))
    out = {
        "metric": "aligned doc-pairs/sec", "value": value, "unit": "doc-pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": wl, "name": args.workload, "pairs_per_step_per_gpu": npairs_local, "pipeline": int(bool(args.pipeline)),
                   "N": N, "M": M, "d": d, "overlaps": K, "parallelism": "dp%d (pairs sharded, no collective)" % world},
        "dp_cells_per_s": value * cells, "dp_cells_per_pair": cells,
        "hbm_bytes_resident": {"inputs": int(sum(x.numel() * x.element_size() + y.numel() * y.element_size() for x, y in docs)),
                               "scratch_arena": int(lib.svx_scratch_bytes(ctx.h))},
    }

    if rank == 0:
        comp_pair = float(np.mean([K * (n + m) * d * esz for n, m in shapes]))  # every candidate embedding read once
        if not args.no_profile and stage_ms["total"] > 0:
            own = stage_bytes(N, M, K, d, esz, len(types), W) if args.workload == "c2" else {}
            stages = {}
            for s in stage_names:
                if stage_launch[s] <= 0 or s in ("total", "host_plan", "host_launch", "setup"):
                    continue
                e = {"kernel": STAGE_KERNEL.get(s, s), "ms_per_step": stage_ms[s] / args.steps, "launches_per_step": stage_launch[s] / args.steps}
                if s in own:
                    e["own_pass_MB_per_pair"] = own[s][0] / 1e6
                    e["own_pass_GBps"] = own[s][0] * npairs_local * args.steps / (stage_ms[s] * 1e-3) / 1e9
                    e["served_by"] = own[s][1]
                stages[s] = e
            out["stages"] = stages
            out["stage_ms_per_step"] = {k: v / args.steps for k, v in stage_ms.items()}
            # c3 (ragged pairs: the number of pyramid levels differs from pair to pair) prices only the stages that make
            # exactly one pass over every pair
            # (with the pipeline on, the event-timed span of a latency-bound kernel includes what it waits for beside the
            #  streaming kernels: the dominant kernel is picked among the kernels that have the context's stream to themselves)
            streaming = ("pyr0", "pyr1", "pyrN", "knob_scores0", "knob_scoresN", "band_costs0", "band_costsN", "dense_costs")
            cand = {k: v for k, v in stages.items() if k in streaming} if args.workload == "c2" else \
                {k: v for k, v in stages.items() if k in ("pyr0", "pyr1", "band_costs0", "knob_scores0")}
            dom = max(cand, key=lambda s: cand[s]["ms_per_step"])
            launches = max(1, stage_launch[dom])
            avg_ms = stage_ms[dom] / launches
            # passes over the batch a stage makes per step (one per pyramid level it covers); with the software pipeline a
            # pass is cut into several launches (half-batches, and slices of them for the two big pyramid passes), so a
            # launch processes pairs * passes / launches-per-step document pairs
            L = len(level_sizes(N, M)) - 1
            passes = {"pyrN": max(1, L - 1), "band_costsN": max(1, L - 1), "band_dpN": max(1, L - 1), "traceback": max(1, L - 1), "path": max(1, L - 1)}.get(dom, 1)
            pairs_per_launch = npairs_local * passes * args.steps / launches
            comp_launch = comp_pair * pairs_per_launch
            achieved = comp_launch / (avg_ms * 1e-3) / 1e9
            traffic = traffic_note = None
            for tname in ("r03_hbm_traffic.json", "r02_hbm_traffic.json"):   # the newest committed counter run of this configuration
                try:
                    tj = json.load(open(os.path.join(ROOT, "profiles", tname)))
                    same = (tj.get("pairs_per_step") == npairs_local and tj.get("workload") == args.workload and tj.get("dtype") == args.dtype
                            and (tj.get("N"), tj.get("M"), tj.get("d"), tj.get("overlaps")) == (N, M, d, K))
                    if same and dom in tj.get("hbm_bytes_per_launch", {}):
                        # counters are per launch of the run they were taken in; a launch here may cover fewer pairs
                        per_pair = tj["hbm_bytes_per_launch"][dom] / float(tj.get("pairs_per_launch", {}).get(dom, tj["pairs_per_step"]))
                        traffic = per_pair * pairs_per_launch
                        traffic_note = "profiles/%s: %s" % (tname, tj.get("how"))
                        break
                except Exception:
                    pass
            if traffic is None:
                traffic_note = "no committed counter pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE) at this workload / pairs per step: traffic is null, not zero"
            rl = {"bound": "hbm", "kernel": STAGE_KERNEL.get(dom, dom), "stage": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                  "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_note,
                  "avg_launch_ms": avg_ms, "launches_per_step": launches / args.steps, "pairs_per_launch": pairs_per_launch,
                  "algorithmic_bytes_per_launch": comp_launch,
                  "definition": "SURVEY 8(d): K(N+M)d*e bytes per pair x pairs per launch / average launch time",
                  "share_of_step": stage_ms[dom] / stage_ms["total"],
                  "whole_path": {"bytes_per_pair": comp_pair, "GBps": comp_pair * value / max(1, world) / 1e9,
                                 "frac": comp_pair * value / max(1, world) / 1e9 / HBM_PEAK_GBPS}}
            if dom in own:
                g = own[dom][0] * npairs_local * args.steps / (stage_ms[dom] * 1e-3) / 1e9
                rl["own_pass"] = {"bytes_per_pair": own[dom][0], "GBps": g, "frac": g / HBM_PEAK_GBPS, "served_by": own[dom][1]}
            out["roofline"] = rl
        # ---- CPU baseline, 1 thread: the oracle (restatement of the reference, pinned bit-exact against it) on the same pairs
        if world == 1 and args.cpu_pairs > 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle
            try:
                from threadpoolctl import threadpool_limits
            except Exception:
                threadpool_limits = None
            ncpu = min(args.cpu_pairs, npairs_local)
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
                                   "sample": "%d of the same pairs, oracle/oracle.py vecalign() (C restatement of "
                                             "dp_core.pyx + numpy), 1 thread, %.1f s" % (ncpu, cpu_s)}
            same, worst = True, 0.0
            for i in range(ncpu):
                same = same and (cpu_out[i][0]['final_alignments'] == res[i][0])
                if len(cpu_out[i][0]['alignment_scores']) == len(res[i][1]):
                    worst = max(worst, float(np.abs(cpu_out[i][0]['alignment_scores'] - res[i][1]).max()))
            out["parity"] = {"pairs_checked": ncpu, "spans_identical": bool(same), "max_score_diff": worst}
            del hosts
        # ---- end-to-end legs (fresh pairs, fresh indices; inputs in host memory / on disk)
        del pb, docs
        torch.cuda.empty_cache()
        if world == 1 and args.workload == "c2":
            e2e = {}
            if args.e2e_pairs > 0:
                e2e["host_memory"] = e2e_host_memory(args, dev, types, W, tdt, K, N, M, d)
            if args.e2e_files > 0:
                e2e["files"] = e2e_files(args, dev, K, N, M, d)
            if e2e:
                out["end_to_end"] = e2e
        # ---- CPU baseline, all cores: P independent processes over disjoint pairs (leg ii)
        if cpu_children:
            try:
                for c in cpu_children:
                    c.stdin.write("prep\n")
                    c.stdin.flush()
                for c in cpu_children:
                    assert c.stdout.readline().strip() == "ready"
                for c in cpu_children:
                    c.stdin.write("go\n")
                    c.stdin.flush()
                secs = [float(c.stdout.readline().strip()) for c in cpu_children]
                for c in cpu_children:
                    c.wait(timeout=30)
                tot = len(cpu_children) * args.cpu_pairs_per_proc
                out.setdefault("cpu_baseline", {})["all_cores"] = {
                    "value": tot / max(secs), "unit": "doc-pairs/s", "cores": len(cpu_children), "kind": "port",
                    "sample": "%d processes x %d pairs of the same workload (own seeds), 1 thread each, started together; "
                              "slowest process %.1f s" % (len(cpu_children), args.cpu_pairs_per_proc, max(secs))}
            except Exception as e:  # never lose the GPU numbers to the CPU leg
                out.setdefault("cpu_baseline", {})["all_cores"] = {"error": repr(e)}
                for c in cpu_children:
                    c.kill()
        return out
    return None


if __name__ == "__main__":
    try:
        main()
    finally:
        for c in CPU_CHILDREN:  # the idle CPU-leg children must never outlive a failed run
            if c.poll() is None:
                c.kill()
