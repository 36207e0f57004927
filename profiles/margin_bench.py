"""Timing of the margin-scoring kernels (svx_knn_mean_sim) -- not the headline bench (bench.py).
python profiles/margin_bench.py [--n 100000] [--db 100000] [--d 1024] [--k 16] [--reps 3]
Prints one JSON line: ms per search, achieved TFLOP/s (2 n N d flops) against the dense fp16 MFMA peak."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-vecalign_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--db", type=int, default=100000)
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--storage", default="fp16")
    a = ap.parse_args()
    import torch
    from svx.postprocess.flat_index import FlatIndex
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(a.n, a.d, device="cuda", generator=g)
    idx = FlatIndex(a.d, a.storage)
    idx.add(torch.randn(a.db, a.d, device="cuda", generator=g))
    idx.mean_sim(q[:1024], a.k)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    idx.ctx.use_current_stream()
    ev[0].record()
    for _ in range(a.reps):
        out = idx.mean_sim(q, a.k)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / a.reps
    tf = 2.0 * a.n * a.db * a.d / (ms * 1e-3) / 1e12
    print(json.dumps({"op": "svx_knn_mean_sim", "n": a.n, "db": a.db, "d": a.d, "k": a.k, "storage": a.storage,
                      "ms": round(ms, 3), "tflops": round(tf, 1), "mfma_peak_tflops": 2500.0, "frac": round(tf / 2500.0, 4),
                      "checksum": float(out.double().sum().item())}))


if __name__ == "__main__":
    main()
