"""`BMSAUWrapper`: counterpart of the reference's box-convolution variant (model/model_box.py:360-406).

Same three coupled U-Net stages as `MSAUWrapper`, with every residual 3x3 block replaced by a `MultiBoxConvBlock`
(model_box.py:9-59): ReLU, then `num_box_convs` x [BoxConv2d(c -> num_box_per_channels * c, learnable real-valued boxes
up to `max_box_sizes`) -> conv 1x1 back to c channels], residual add, ReLU.  Constructor keywords and defaults are the
reference's (`num_box_convs` 3, `max_box_sizes` 28, `num_box_per_channels` 3, ...); `forward -> (pred, logits,
aux_logits)`; state_dict keys follow the reference's module tree
(`...conv_box_list.{l}.conv_list.{2i}.{x_min,x_max,y_min,y_max}` [c, F], `...conv_list.{2i+1}.custom_conv.{weight,bias}`).

**PARITY UNPINNED.**  `BoxConv2d` comes from the third-party package `box_convolution` (shrubb/box-convolutions), which is
neither vendored in the reference nor installed here, and the reference holds no fixtures for this path.  The box filter
implemented by csrc/boxconv.hip follows the published definition (Burkov & Lempitsky, NeurIPS 2018) with the conventions
written down in oracle/box_oracle.py, and is verified against that restatement only: self-consistent, not
reference-identical (box parameter initialisation and edge conventions of the package may differ).
"""
from __future__ import annotations

from .model import MSAUWrapper


class BMSAUWrapper(MSAUWrapper):
    def __init__(self, channels=1, n_class=2, model_kwargs={}):
        kw = dict(model_kwargs)
        self.num_box_convs = kw.get("num_box_convs", 3)
        self.max_box_sizes = kw.get("max_box_sizes", 28)
        self.num_box_per_channels = kw.get("num_box_per_channels", 3)
        if isinstance(self.max_box_sizes, (tuple, list)):
            if self.max_box_sizes[0] != self.max_box_sizes[1]:
                raise NotImplementedError("max_box_sizes: one value for both axes (the reference passes an int)")
            self.max_box_sizes = self.max_box_sizes[0]
        if self.num_box_convs < 1 or self.num_box_per_channels < 1 or self.max_box_sizes < 1:
            raise ValueError("num_box_convs, num_box_per_channels and max_box_sizes must be positive")
        if kw.get("activation_name", "relu") != "relu":
            raise NotImplementedError("the box variant's kernels implement activation_name='relu' only")
        super().__init__(channels, n_class, kw)

    def _variant_cfg(self, kw: dict) -> dict:
        return dict(variant="box", num_box_convs=int(self.num_box_convs), num_box_per_channels=int(self.num_box_per_channels),
                    max_box_sizes=float(self.max_box_sizes))

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """Loads like MSAUWrapper -- and says what cannot be promised.  The stored box borders `x_min / x_max / y_min / y_max`
        are read HERE as fractions of `max_box_sizes` (msau_box_params, oracle/box_oracle.py).  The `box_convolution` package
        the reference imports keeps its parameters in its own re-parametrised units (reportedly scaled by the maximal input
        size times a `reparametrization_factor`, default 8 -- the package is neither in the reference tree nor installed, so
        this could not be checked): a checkpoint trained with the reference loads without error, but its boxes may come out at
        a different scale.  PARITY UNPINNED; a warning is the honest interface until a fixture pins it."""
        import warnings
        if any(k.rsplit(".", 1)[-1] in ("x_min", "x_max", "y_min", "y_max") for k in state_dict):
            warnings.warn("BMSAUWrapper.load_state_dict: box-border parameters are interpreted as fractions of max_box_sizes; "
                          "the third-party BoxConv2d's own parameter units are unpinned here (no source, no fixture), so a "
                          "reference-trained box checkpoint may not reproduce its boxes", RuntimeWarning, stacklevel=2)
        return super().load_state_dict(state_dict, strict=strict, assign=assign)
