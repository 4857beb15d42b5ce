"""Counterpart of the reference's model/training package (Trainer, UNetLoss, get_optimizer)."""
from .cost import UNetLoss  # noqa: F401
from .optimizer import get_optimizer  # noqa: F401
from .trainer import Trainer  # noqa: F401
