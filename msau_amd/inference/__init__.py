"""Inference side of the hot path: counterpart of the reference's `inference/` package.
`KVModel.predict` paints the character-id mask on the host, runs the forward-only HIP plan (one-hot input painted on
the device, no activations kept, softmax + argmax in the end conv's epilogue, NHWC output) and keeps the reference's
CPU post-processing (`_extract_value`, `post_process_kv`)."""
from .kv_model import KVModel  # noqa: F401
from .postprocess import CLASS_NAMES, post_process_kv  # noqa: F401
