"""Counterpart of the reference's model/training package (Trainer, UNetLoss, get_optimizer)."""
from .checkpoint import create_filename, gen_prefix, load_checkpoint, save_checkpoint  # noqa: F401
from .cost import UNetLoss  # noqa: F401
from .optimizer import get_optimizer  # noqa: F401
from .trainer import Trainer  # noqa: F401
