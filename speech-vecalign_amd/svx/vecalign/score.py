"""Strict / lax precision, recall and F1 of alignments against gold alignments.

Same definitions as the reference's evaluation harness (svecalign/vecalign/score.py:35-134): a test
alignment is a strict hit if it occurs verbatim in the gold set; otherwise it is a lax hit if some
source id of it is linked in gold to some target id of it.  Recall is precision with the roles
swapped after dropping deletions (score.py:88-92).

    python -m svx.vecalign.score -t test.txt [...] -g gold.txt [...]
"""
import argparse
import sys
from collections import defaultdict

from ..utils.file_utils import read_alignments

KEYS = ("recall_strict", "recall_lax", "precision_strict", "precision_lax", "f1_strict", "f1_lax")


def _as_set(alignments):
    return {(tuple(x), tuple(y)) for x, y in alignments if len(x) or len(y)}


def _precision(goldalign, testalign):
    """-> [strict hits, strict misses, lax hits, lax misses] of `testalign` w.r.t. `goldalign`."""
    test, gold = _as_set(testalign), _as_set(goldalign)
    linked = defaultdict(set)  # gold: source id -> target ids it is aligned with
    for gs, gt in gold:
        for s in gs:
            linked[s].update(gt)
    strict_hit = strict_miss = lax_hit = lax_miss = 0
    for ts, tt in test:
        if (ts, tt) in gold:
            strict_hit += 1
            lax_hit += 1
            continue
        strict_miss += 1
        reachable = set()
        for s in ts:
            reachable |= linked.get(s, set())
        if reachable.intersection(tt):
            lax_hit += 1
        else:
            lax_miss += 1
    return [strict_hit, strict_miss, lax_hit, lax_miss]


def _ratio(hit, miss, default):
    return default if hit + miss == 0 else hit / float(hit + miss)


def _f1(p, r, default):
    return default if p + r == 0 else 2 * (p * r) / (p + r)


def score_multiple(gold_list, test_list, value_for_div_by_0=0.0):
    pc, rc = [0, 0, 0, 0], [0, 0, 0, 0]
    for gold, test in zip(gold_list, test_list):
        pc = [a + b for a, b in zip(pc, _precision(gold, test))]
        test_nd = [(x, y) for x, y in test if len(x) and len(y)]
        gold_nd = [(x, y) for x, y in gold if len(x) and len(y)]
        rc = [a + b for a, b in zip(rc, _precision(test_nd, gold_nd))]
    d = value_for_div_by_0
    ps, pl = _ratio(pc[0], pc[1], d), _ratio(pc[2], pc[3], d)
    rs, rl = _ratio(rc[0], rc[1], d), _ratio(rc[2], rc[3], d)
    return dict(recall_strict=rs, recall_lax=rl, precision_strict=ps, precision_lax=pl,
                f1_strict=_f1(ps, rs, d), f1_lax=_f1(pl, rl, d))


def log_final_scores(res, file=sys.stderr):
    rows = (("Precision", "precision"), ("Recall", "recall"), ("F1", "f1"))
    print(' ---------------------------------', file=file)
    print('|             |  Strict |    Lax  |', file=file)
    for label, key in rows:
        print('| %-11s |   %.3f |   %.3f |' % (label, res[key + '_strict'], res[key + '_lax']), file=file)
    print(' ---------------------------------', file=file)


def main(argv=None):
    ap = argparse.ArgumentParser('Compute strict/lax precision and recall for one or more pairs of gold/test alignments',
                                 formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument('-t', '--test', type=str, nargs='+', required=True, help='one or more test alignment files')
    ap.add_argument('-g', '--gold', type=str, nargs='+', required=True, help='one or more gold alignment files')
    args = ap.parse_args(argv)
    if len(args.test) != len(args.gold):
        raise Exception('number of gold/test files must be the same')
    res = score_multiple(gold_list=[read_alignments(x) for x in args.gold], test_list=[read_alignments(x) for x in args.test])
    log_final_scores(res)
    return res


if __name__ == '__main__':
    main()
