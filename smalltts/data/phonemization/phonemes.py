"""`from smalltts.data.phonemization.phonemes import get_token_ids` (reference src/smalltts/data/phonemization/phonemes.py:10-89)."""
from smalltts_amd.phonemes import (NV_REPEAT, decode_token_ids, get_token_ids, idx2p, p2idx, phoneme_len,  # noqa: F401
                                   symbols)
