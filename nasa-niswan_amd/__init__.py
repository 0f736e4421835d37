"""MI355X-native ConvLSTM hot path for Smart-NINT (drop-in for the reference's model.py /
train.py batch step).  Import as ``nasa_niswan_amd`` (the directory name carries a hyphen;
``nasa_niswan_amd/`` at the repository root is the importable alias of this package)."""
from ._lib import NintError, load as load_library  # noqa: F401
from .model import ConvLSTM, ConvLSTMCell  # noqa: F401

__all__ = ["ConvLSTM", "ConvLSTMCell", "load_library", "NintError"]
