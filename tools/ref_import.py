"""Import recipe for the read-only reference at /root/reference (build container only).

Only tools/capture_goldens.py (and ad-hoc validation runs) use this; nothing here
travels to the GPU box except the fixtures the capture script writes to tests/golden/.
Recipe per SURVEY.md section 8(c).
"""
import os
import sys
import types
from unittest.mock import MagicMock

REF = "/root/reference/MEVI"


def setup():
    os.environ.setdefault("PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION", "python")
    sys.dont_write_bytecode = True
    import collections
    import collections.abc
    for name in ("Sequence", "Mapping", "MutableMapping", "Iterable", "Callable"):
        if not hasattr(collections, name):
            setattr(collections, name, getattr(collections.abc, name))
    sys.modules.setdefault("sacremoses", MagicMock())
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import transformers  # vendored 3.4.0 fork (must precede the faiss stub)
    assert transformers.__version__.startswith("3.4"), transformers.__version__
    sys.modules.setdefault("faiss", MagicMock())
    import torch.nn as nn

    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        trainer = None

        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    pl.Trainer = MagicMock()
    pl.seed_everything = MagicMock()
    sys.modules.setdefault("pytorch_lightning", pl)
    for sub in ("callbacks", "loggers", "plugins", "utilities", "utilities.distributed"):
        sys.modules.setdefault("pytorch_lightning." + sub, MagicMock())
    sys.modules.setdefault("nltk", MagicMock())
    return transformers
