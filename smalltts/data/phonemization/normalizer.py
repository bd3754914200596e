"""`from smalltts.data.phonemization.normalizer import EnglishTextNormalizer` (reference normalizer.py:8-149)."""
from smalltts_amd.normalizer import EnglishTextNormalizer  # noqa: F401
