"""`from smalltts.infer.onnx import SmallTTS, estimate_duration` (reference src/smalltts/infer/onnx.py:11-18,50-159)."""
from smalltts_amd.api import (CHARS_PER_SECOND, HOP_SIZE, NUM_STEPS, SAMPLE_RATE, SmallTTS,  # noqa: F401
                              estimate_duration)
