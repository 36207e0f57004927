"""Driver of tests/test_host_sanitized.py: runs INSIDE a process whose LD_PRELOAD is the AddressSanitizer runtime and
exercises the native host helpers of the C ABI (speech-vecalign_amd/csrc/svx_host.hip: MT19937 choice stream, index
drawing, candidate table from user files, alignment formatter) out of a build of that ONE source file made with
`g++ -fsanitize=address,undefined -fno-sanitize-recover=all` (no GPU, no HIP).  Any finding of either sanitizer aborts
the process; every answer is also checked (numpy's stream, the Python mirror of the table, Python's formatting), so
the run is a functional test of the instrumented build too.  Prints "HOST_SAN_OK <checks>" at the end.

    python tests/native/host_san_driver.py <libsvx_host_san.so> <scratch dir>
"""
import ctypes
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "speech-vecalign_amd"))

c_i32, c_i64, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
checks = 0


def ok(cond, what):
    global checks
    if not cond:
        print("HOST_SAN_FAIL", what)
        sys.exit(3)
    checks += 1


def main():
    lib = ctypes.CDLL(sys.argv[1])
    tmp = sys.argv[2]
    lib.svx_mt19937_choice.restype = ctypes.c_int
    lib.svx_mt19937_choice.argtypes = [c_vp, ctypes.POINTER(c_i32), c_i64, c_i64, c_vp]
    lib.svx_draw_indices.restype = ctypes.c_int
    lib.svx_norm_index_count.restype = c_i64
    lib.svx_knob_index_count.restype = c_i64
    lib.svx_candidate_table.restype = ctypes.c_int
    lib.svx_candidate_table.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, c_vp, ctypes.c_int,
                                        ctypes.POINTER(c_i32), ctypes.POINTER(c_i64), ctypes.c_char_p, ctypes.c_int]
    lib.svx_format_alignments.restype = c_i64
    lib.svx_format_alignments.argtypes = [c_vp, c_vp, c_i64, c_vp, c_i64]
    lib.svx_num_levels.restype = ctypes.c_int
    lib.svx_knob_count.restype = c_i64

    # ---- MT19937 choice stream against numpy (incl. the regeneration boundary and n = 1, which draws nothing)
    for seed, n, size in [(7, 4096, 5000), (1, 1, 10), (2, 2, 1300), (3, 2 ** 31 - 1, 700), (4, 37, 0), (5, 0xffffffff, 64)]:
        rs = np.random.RandomState(seed)
        st = rs.get_state()
        key = np.array(st[1], dtype=np.uint32)
        pos = c_i32(int(st[2]))
        out = np.empty(max(size, 1), np.int32)
        rc = lib.svx_mt19937_choice(key.ctypes.data, ctypes.byref(pos), n, size, out.ctypes.data)
        ok(rc == 0, "choice rc")
        want = rs.choice(n, size=size, replace=True) if size else np.zeros(0, np.int64)
        ok(np.array_equal(out[:size].astype(np.int64) & 0xffffffff, np.asarray(want, np.int64) & 0xffffffff), "choice stream n=%d" % n)
        ok(np.array_equal(key, rs.get_state()[1]) and pos.value == rs.get_state()[2], "choice state n=%d" % n)
    bad = c_i32(700)
    ok(lib.svx_mt19937_choice(np.zeros(624, np.uint32).ctypes.data, ctypes.byref(bad), 5, 5, np.zeros(5, np.int32).ctypes.data) != 0, "pos > 624 refused")
    ok(lib.svx_mt19937_choice(None, ctypes.byref(bad), 5, 5, None) != 0, "null refused")

    # ---- svx_draw_indices: layout sizes and the stream, through the product's own Python wrapper logic restated here
    from svx.vecalign.dp_utils import level_sizes
    for (n, m, k0, k1, mf, cs, ns, h0, h1) in [(4096, 4096, 4, 4, 300, 20000, 100, 0, 0), (40, 37, 3, 3, 300, 20000, 100, 0, 0),
                                               (1, 1, 1, 1, 300, 20000, 100, 0, 0), (333, 1200, 4, 4, 300, 20000, 7, 1, 1),
                                               (65536, 3, 2, 1, 10, 999, 1, 0, 0), (700, 1, 2, 2, 300, 20000, 0, 0, 1)]:
        nn = lib.svx_norm_index_count(n, m, k0, k1, mf, ns, h0, h1)
        kn = lib.svx_knob_index_count(n, m, mf, cs)
        rs = np.random.RandomState(11)
        st = rs.get_state()
        key, pos = np.array(st[1], dtype=np.uint32), c_i32(int(st[2]))
        no, ko = np.empty(max(nn, 1), np.int32), np.empty(max(kn, 1), np.int32)
        rc = lib.svx_draw_indices(c_vp(key.ctypes.data), ctypes.byref(pos), n, m, k0, k1, mf, cs, ns, h0, h1,
                                  c_vp(no.ctypes.data if nn else 0), c_vp(ko.ctypes.data))
        ok(rc == 0, "draw rc %s" % ((n, m),))
        # numpy in the reference's call order (dp_utils.py:340-348 per depth, then :286-302 per depth)
        import math
        want_n, want_k = [], []
        sizes = level_sizes(n, m, mf)
        for depth, (a, b) in enumerate(sizes):
            for (size_other, k_other, skip) in ((b, k1, depth == 0 and h0), (a, k0, depth == 0 and h1)):
                spo = math.ceil(ns / k_other)
                if skip or spo == 0 or size_other == 0:  # (the product draws nothing for an empty level, like its Python mirror)
                    continue
                for _ in range(k_other):
                    want_n.append(rs.choice(size_other, size=spo, replace=True))
        for (a, b) in sizes:
            if a * b < cs:
                want_k += [np.repeat(np.arange(a), b), np.tile(np.arange(b), a)]
            else:
                want_k += [rs.choice(a, size=cs, replace=True), rs.choice(b, size=cs, replace=True)]
        wn = np.concatenate(want_n) if want_n else np.zeros(0, np.int64)
        wk = np.concatenate(want_k)
        ok(len(wn) == nn and len(wk) == kn, "index counts %s: %d/%d %d/%d" % ((n, m), len(wn), nn, len(wk), kn))
        ok(np.array_equal(no[:nn], wn) and np.array_equal(ko[:kn], wk), "index stream %s" % ((n, m),))
    ok(lib.svx_draw_indices(None, None, 1, 1, 1, 1, 1, 1, 1, 0, 0, None, None) != 0, "draw null refused")
    ok(lib.svx_num_levels(4096, 4096, 300) == 4 and lib.svx_num_levels(1, 1, 1) == 0 and lib.svx_num_levels(2 ** 30, 2 ** 30, 1) == 30, "levels")
    ok(lib.svx_knob_count(50000, 50000, 20000) == 20000 and lib.svx_knob_count(3, 4, 20000) == 12, "knob count")

    # ---- candidate table: the shipped example against the Python mirror, then hostile files
    from svx.utils import embedding_utils as E
    from svx.vecalign.vecalign import load_ignore_index_file

    def table(seg, cat, K, ign=None, cap=None):
        cap = os.path.getsize(seg) // 2 + 2 if cap is None else cap
        t = np.full((K, max(cap, 1)), -7, np.int32)
        nl, nc = c_i32(0), c_i64(0)
        err = ctypes.create_string_buffer(512)
        rc = lib.svx_candidate_table(seg.encode(), cat.encode(), None if ign is None else ign.encode(), K, t.ctypes.data, cap,
                                     ctypes.byref(nl), ctypes.byref(nc), err, 512)
        return rc, t[:, :nl.value].copy() if rc == 0 else None, nc.value, err.value.decode(errors="replace")

    for fixture in ("example_trim", "example_full"):
        D = os.path.join(ROOT, "tests", "golden", fixture)
        for lang, side in (("en", "src"), ("de", "tgt")):
            with open(f"{D}/cat_segs_{lang}.txt") as f:
                s2i = {}
                for i, line in enumerate(f):
                    s2i.setdefault(line.strip(), i)
            lines = open(f"{D}/segments_{lang}.txt").readlines()
            ign = load_ignore_index_file(f"{D}/ignore_{side}.txt")
            for K, ig, igf in ((5, ign, f"{D}/ignore_{side}.txt"), (3, None, None), (9, ign, f"{D}/ignore_{side}.txt")):
                want = E.candidate_index_table(s2i, lines, K, ig, overlap_segments=True)
                rc, got, ncand, msg = table(f"{D}/segments_{lang}.txt", f"{D}/cat_segs_{lang}.txt", K, igf)
                ok(rc == 0 and np.array_equal(want, got), "table %s %s K=%d: %s" % (fixture, lang, K, msg))
    w = lambda name, data: (open(os.path.join(tmp, name), "wb").write(data), os.path.join(tmp, name))[1]
    cat = w("cat.txt", b"10 25\n0 10\n0 25\n0 10\n")
    hostile = {
        "empty": b"", "newline_only": b"\n\n\n", "one_token": b"0 10\n17\n", "cr_only": b"0 10\r10 25\r25 31\r",
        "crlf": b"0 10\r\n10 25\r\n", "no_final_newline": b"0 10\n10 25", "tabs_and_spaces": b"  0\t10  \n\t10   25\n",
        "long_ids": b"%s %s\n" % (b"9" * 5000, b"8" * 70000), "nul_bytes": b"0 10\n\x00\x00 5\n", "three_tokens": b"0 10 junk\n10 25 more junk\n",
        "binary": bytes(range(256)) * 40, "very_long_line": b"0 " + b"1" * (1 << 20) + b"\n",
    }
    for name, data in hostile.items():
        seg = w("seg_%s.txt" % name, data)
        for K in (1, 2, 7):
            rc, got, ncand, msg = table(seg, cat, K)
            ok(rc == 0 or len(msg) > 0, "hostile %s: failure without a message" % name)
            if rc == 0:
                ok(got.shape[0] == K and (got >= -1).all() and (got < 4).all(), "hostile %s: rows out of range" % name)
            rc2, _, _, _ = table(seg, cat, K, cap=1)     # a table that is too small must be refused, not overrun
            ok(rc2 != 0 or got is None or got.shape[1] <= 1, "hostile %s: cap ignored" % name)
        for cname, cdata in hostile.items():            # the same bytes as the candidate file and as the ignore file
            rc, got, ncand, msg = table(w("seg_ok.txt", b"0 10\n10 25\n25 31\n"), w("cat_%s.txt" % cname, cdata), 3,
                                        ign=w("ign_%s.txt" % cname, cdata))
            ok(rc == 0 or len(msg) > 0, "hostile candidate/ignore %s" % cname)
    rc, got, ncand, msg = table(w("seg_ok.txt", b"0 10\r\n10 25\n25 31\n"), cat, 2)
    ok(rc == 0 and ncand == 4 and got.tolist() == [[1, 0, -1], [-1, 2, -1]], "known small table: %s %s" % (got, msg))
    rc, _, _, msg = table(os.path.join(tmp, "does_not_exist"), cat, 2, cap=4)
    ok(rc != 0 and "cannot read" in msg, "missing file: %s" % msg)
    nl, nc = c_i32(0), c_i64(0)
    ok(lib.svx_candidate_table(w("seg_ok.txt", b"0 10\n10 25\n").encode(), cat.encode(), None, 2, None, 0, ctypes.byref(nl), ctypes.byref(nc), None, 0) == 0
       and nl.value == 2 and nc.value == 4, "size query with no table and no message buffer")

    # ---- formatter: Python's own formatting, the size query, buffers that are too small by one byte
    from svx.vecalign.vecalign import print_alignments
    from svx.vecalign.dp_utils import alignments_to_rows
    al = [([0], [0]), ([1, 2, 3], [1]), ([], [2]), ([4], []), (list(range(5, 40)), list(range(3, 60)))]
    sc = np.array([0.123456789, 1e-9, 0.5, 12345.678901, 1e300])
    rows = np.ascontiguousarray(alignments_to_rows(al), np.int32)
    for scores in (sc, None):
        buf = io.StringIO()
        print_alignments(al, scores=scores, ofile=buf)
        want = buf.getvalue().encode()
        sp = c_vp(scores.ctypes.data) if scores is not None else None
        need = lib.svx_format_alignments(rows.ctypes.data, sp, len(al), None, 0)
        ok(need == len(want), "formatter size query %d != %d" % (need, len(want)))
        out = ctypes.create_string_buffer(need)
        ok(lib.svx_format_alignments(rows.ctypes.data, sp, len(al), out, need) == need and out.raw[:need] == want, "formatter text")
        for cap in (0, 1, need - 1):
            small = ctypes.create_string_buffer(max(cap, 1))
            ok(lib.svx_format_alignments(rows.ctypes.data, sp, len(al), small, cap) == need, "formatter with cap %d" % cap)
    ok(lib.svx_format_alignments(None, None, 0, None, 0) == 0, "formatter: nothing to print")
    print("HOST_SAN_OK", checks)


if __name__ == "__main__":
    main()
