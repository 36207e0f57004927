#!/usr/bin/env bash
# Build the reference's one native module (svecalign/vecalign/dp_core.pyx) from where it
# lies under /root/reference into oracle/_ref/ (git-ignored).  This is the same Cython->C->gcc
# step the reference performs itself at import time through pyximport
# (/root/reference/svecalign/vecalign/dp_utils.py:23-27), run by hand because /root/reference is
# read-only and pyximport is configured inplace=True.  TEST INFRASTRUCTURE ONLY: nothing in the
# product imports this; it exists to pin oracle/ against the real reference in this container
# (tests/golden/make_golden.py, tests/test_oracle_vs_reference.py).
set -euo pipefail
REF=${SVX_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
if [ ! -f "$REF/svecalign/vecalign/dp_core.pyx" ]; then
  echo "reference not present at $REF; skipping _ref build"
  exit 0
fi
mkdir -p "$OUT"
SUF=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
if [ -f "$OUT/dp_core$SUF" ] && [ "$OUT/dp_core$SUF" -nt "$REF/svecalign/vecalign/dp_core.pyx" ]; then
  exit 0
fi
cython -3 "$REF/svecalign/vecalign/dp_core.pyx" -o "$OUT/dp_core.c"
# same flags distutils/pyximport would use (python's own CFLAGS: -O2, no fast-math)
CFLAGS=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('CFLAGS'))")
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
NPINC=$(python3 -c "import numpy; print(numpy.get_include())")
gcc -shared -fPIC $CFLAGS -w -I"$PYINC" -I"$NPINC" "$OUT/dp_core.c" -o "$OUT/dp_core$SUF"
rm -f "$OUT/dp_core.c"
echo "built $OUT/dp_core$SUF"
