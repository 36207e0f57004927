import os, sys, ctypes, json
import numpy as np
sys.path.insert(0, "speech-vecalign_amd"); sys.path.insert(0, "tests")
import torch
from svx import _lib
from synth import alignment_types
ctx = _lib.context(0)
N = M = 4096; W = 7; B = 14
out = {}
for name, types in (("T10", alignment_types(5)), ("T1", [(1, 1)])):
    T = len(types)
    A = N + M + 3
    path_y = np.minimum(np.arange(A) // 2, M)
    boff = (path_y - W).astype(np.int32)
    costs = torch.rand((T, A, B), device="cuda", dtype=torch.float32)
    dboff = torch.from_numpy(boff).cuda()
    csum = torch.empty((A + 2, B), dtype=torch.float64, device="cuda")
    xp = torch.empty((A + 2, B), dtype=torch.int32, device="cuda"); yp = torch.empty_like(xp)
    bout = torch.empty(A + 2, dtype=torch.int32, device="cuda")
    flat = [v for xy in types for v in xy]
    ct = (ctypes.c_int32 * len(flat))(*flat)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    def run():
        ctx.check(ctx.lib.svx_sparse_dp(ctx.h, p(costs), p(dboff), A, B, ct, T, 0.3, N, M, p(csum), p(xp), p(yp), p(bout)))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.use_current_stream()
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    out[name] = round(e0.elapsed_time(e1) / 5, 3)
print(json.dumps({"dbg": os.environ.get("SVX_DP_DBG", "0"), "ms": out, "us_per_diag": {k: round(v * 1e3 / (N + M + 5), 3) for k, v in out.items()}}))
