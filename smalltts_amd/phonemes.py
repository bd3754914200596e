"""Phoneme symbol inventory and token ids for the text front-end (CPU side of the hot path).

The id space is the reference's (src/smalltts/data/phonemization/phonemes.py:10-55): 197 symbols
-> ids 1..197, id 0 = padding, `phoneme_len` = 198; non-verbal `[event]` tags expand to
NV_REPEAT copies of the event id (:39,77-89).  Text is first normalised like the reference does (numbers, currency, abbreviations spelled out:
`normalizer.py`).  Phonemisation itself needs espeak through the
`phonemizer` package (pinned 3.3.0 by the reference's uv.lock), which is not available offline;
when it is missing, pass token ids directly (CLI `--tokens`) or use backend="chars", a clearly
non-reference grapheme fallback for smoke runs.
"""
from __future__ import annotations

import re
from typing import Iterable, List, Optional

NV_REPEAT = 4

PUNCTUATION = ';:,.!?¡¿—…"«»"" '
LATIN = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"
IPA = ("ɑɐɒæɓʙβɔɕçɗɖðʤəɘɚɛɜɝɞɟʄɡɠɢʛɦɧħɥʜɨɪʝɭɬɫɮʟɱɯɰŋɳɲɴøɵɸθœɶʘɹɺɾɻʀʁɽʂʃʈʧʉʊʋⱱʌɣɤʍχʎʏʑʐʒʔʡʕʢǀǁǂǃˈˌːˑʼʴʰʱʲʷˠˤ˞↓↑→↗↘'̩'ᵻ")
EVENTS = ("babble boo burp chant cheer cough cry gargle gasp groan grunt hiccup hum laughter moan "
          "shout sigh sing sneeze sniff snore whisper whistle").split()


def _build_symbols() -> List[str]:
    ordered = dict.fromkeys(PUNCTUATION + LATIN + IPA)       # first occurrence wins, order kept
    ordered.update(dict.fromkeys(f"[{e}]" for e in EVENTS))
    return list(ordered)


symbols: List[str] = _build_symbols()
p2idx = {s: i for i, s in enumerate(symbols, start=1)}
idx2p = {i: s for s, i in p2idx.items()}
phoneme_len = len(symbols) + 1

_EVENT_RE = re.compile(r"\[(\w+)\]")
_WORDS_RE = re.compile(r"\w+|[^\w\s]")
_espeak = None


def event_id(label: str) -> Optional[int]:
    label = label.lower()
    return p2idx[f"[{label}]"] if label in EVENTS else None


_normalizer = None


def normalize_text(text: str) -> str:
    """Abbreviation / number expansion the reference applies before espeak (phonemes.py:67-70 -> normalizer.py)."""
    global _normalizer
    if _normalizer is None:
        from .normalizer import EnglishTextNormalizer
        _normalizer = EnglishTextNormalizer()
    return _normalizer.normalize(text)


def _espeak_phonemize(text: str) -> str:
    global _espeak
    if _espeak is None:
        try:
            from phonemizer.backend import EspeakBackend
            from phonemizer.logger import get_logger
        except Exception as e:  # pragma: no cover - depends on the host
            raise RuntimeError("phonemizer/espeak is not installed: pass token ids (--tokens) or "
                               "backend='chars'") from e
        _espeak = EspeakBackend(language="en-us", preserve_punctuation=True, with_stress=True,
                                words_mismatch="ignore", logger=get_logger(verbosity="quiet"))
    return " ".join(_WORDS_RE.findall(_espeak.phonemize([text])[0]))


def get_token_ids(text: str, backend: str = "espeak") -> List[int]:
    out: List[int] = []
    for i, part in enumerate(_EVENT_RE.split(text)):
        if i % 2:                                   # captured [event] label
            eid = event_id(part)
            if eid is not None:
                out += [eid] * NV_REPEAT
        elif part.strip():
            part = normalize_text(part)              # reference _phonemize: normalise, then phonemise
            s = _espeak_phonemize(part) if backend == "espeak" else part
            out += [p2idx[c] for c in s if c in p2idx]
    return out


def decode_token_ids(ids: Iterable[int]) -> str:
    return "".join(idx2p.get(int(t), "") for t in ids)


def parse_tokens_arg(arg: str) -> List[int]:
    """CLI helper: '1,2,3' / '1 2 3' / a JSON list -> ids."""
    return [int(t) for t in re.findall(r"-?\d+", arg)]
