"""Console output helpers with the reference's UX contract
(gsconverter/utils/utility_functions.py:12-37, config.py:9): ``status_print`` always
prints through ``tqdm.write`` so progress bars stay intact, ``debug_print`` only when
the DEBUG flag is set.  When the reference package is importable its own ``config``
module is the flag's source of truth so ``--debug`` keeps working after install()."""
from __future__ import annotations

DEBUG = False


def _debug_enabled() -> bool:
    try:
        from gsconverter.utils import config as _cfg  # type: ignore
        return bool(_cfg.DEBUG) or DEBUG
    except Exception:
        return DEBUG


def status_print(*args, **kwargs):
    try:
        from tqdm import tqdm
        tqdm.write(" ".join(map(str, args)), **kwargs)
    except ImportError:
        print(*args, **kwargs)


def debug_print(*args, **kwargs):
    if _debug_enabled():
        status_print(*args, **kwargs)
