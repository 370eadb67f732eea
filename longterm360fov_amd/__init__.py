"""longterm360fov_amd - MI355X-native seq2seq-LSTM hot path of ChengeLi/LongTerm360FoV.

Host side is Python (the reference is Python) over a C-ABI HIP library
(``include/fov360.h`` -> ``longterm360fov_amd/lib/libfov360_hip.so``).  The package mirrors the
reference's call surface for this path only: ``config.cfg`` knobs, the ``utility`` data helpers
and a Keras-style model object with ``compile / fit / predict / predict_on_batch``.
"""
from .config import cfg  # noqa: F401

__all__ = ["cfg"]
