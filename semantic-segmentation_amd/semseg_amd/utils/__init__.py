"""Device-side pieces of the reference's evaluation loop (utils/trnval_utils.py, utils/misc.py)."""
from .eval_tail import confusion_matrix, fast_hist   # noqa: F401
