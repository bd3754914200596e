"""`from smalltts.codec.onnx import Encoder, Decoder` (reference src/smalltts/codec/onnx.py:28-75)."""
from smalltts_amd.api import Decoder, Encoder  # noqa: F401
