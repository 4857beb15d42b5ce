"""msau_amd -- MI355X-native training path of the Multi-Stage Attentional U-Net."""
from .model import MSAUWrapper, TrainEngine, param_shapes  # noqa: F401
from .model_box import BMSAUWrapper  # noqa: F401

__all__ = ["MSAUWrapper", "BMSAUWrapper", "TrainEngine", "param_shapes"]
