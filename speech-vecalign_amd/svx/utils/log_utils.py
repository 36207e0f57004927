"""Logging set-up shared by the CLIs: same format and LOGLEVEL switch as the reference
(svecalign/utils/log_utils.py:7-13)."""
import logging
import os
from functools import partial

logging.basicConfig(
    format="%(asctime)s | %(levelname)s | %(name)s | %(message)s",
    datefmt="%Y-%m-%d %H:%M:%S",
    level=os.environ.get("LOGLEVEL", "INFO").upper(),
)

try:
    import tqdm
    my_tqdm = partial(tqdm.tqdm, mininterval=20, maxinterval=60)
except Exception:  # pragma: no cover
    def my_tqdm(x, **kw):
        return x
