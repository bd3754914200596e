"""English text normalisation in front of the phonemiser: abbreviation and number expansion.

The reference runs `EnglishTextNormalizer.normalize(text)` before espeak (reference
`src/smalltts/data/phonemization/phonemes.py:67-70`, class in `normalizer.py:8-149`, adapted there from ZipVoice /
espnet's tacotron cleaners), so "$1,250", "Dr.", "90th" or "3/4" reach the phonemiser spelled out.  Same rules, same order
here: abbreviations (normalizer.py:17-41,146-149), then commas in numbers, pounds, dollars, fractions, decimals, percent,
ordinals, cardinals (:134-144), with years 1001..2999 read in pairs (:113-132).

The reference spells numbers with the `inflect` package (`number_to_words`, `ordinal`; un-vendored third party, not installed
here).  `_Words` below restates the subset of inflect's documented behaviour those calls use — PARITY UNPINNED against inflect
itself: "one hundred and twenty-three" / andword="" -> "one hundred twenty-three", ", " between thousands groups,
hyphenated tens, group=2 pairs with zero="oh", "21st" -> "twenty-first".
"""
from __future__ import annotations

import re

_UNITS = ("zero one two three four five six seven eight nine ten eleven twelve thirteen fourteen fifteen sixteen seventeen "
          "eighteen nineteen").split()
_TENS = "_ _ twenty thirty forty fifty sixty seventy eighty ninety".split()
_MILL = ("", " thousand", " million", " billion", " trillion", " quadrillion", " quintillion", " sextillion", " septillion",
         " octillion", " nonillion", " decillion")
_ORD_IRREGULAR = {"one": "first", "two": "second", "three": "third", "five": "fifth", "eight": "eighth", "nine": "ninth",
                  "twelve": "twelfth"}


class _Words:
    """number -> words the way inflect.engine() spells them for the calls of the reference normaliser."""

    @staticmethod
    def _tens(t: int, u: int, zero: str = "zero") -> str:
        n = 10 * t + u
        if n < 20:
            return _UNITS[n] if n else ""
        return _TENS[t] + ("-" + _UNITS[u] if u else "")

    @classmethod
    def cardinal(cls, n: int, andword: str = "and") -> str:
        if n == 0:
            return "zero"
        groups = []
        i = 0
        while n > 0:
            n, g = divmod(n, 1000)
            if g:
                h, r = divmod(g, 100)
                parts = []
                if h:
                    parts.append(_UNITS[h] + " hundred")
                if r:
                    if h and andword:
                        parts.append(andword)
                    parts.append(cls._tens(r // 10, r % 10))
                if i >= len(_MILL):
                    raise ValueError("number too large to spell")
                groups.append((" ".join(parts) + _MILL[i], g))
            i += 1
        groups.reverse()
        words = [w for w, _ in groups]
        # inflect joins the groups with ", "; a final group below one hundred is attached with the and-word instead
        if len(words) > 1 and andword and groups[-1][1] < 100 and not groups[-1][0].endswith(tuple(m for m in _MILL if m)):
            return ", ".join(words[:-1]) + f" {andword} " + words[-1]
        return ", ".join(words)

    @classmethod
    def pairs(cls, n: int, zero: str = "oh") -> str:
        """inflect number_to_words(n, group=2, zero=zero): digits read in pairs from the left ("19|84", "19|05" -> "oh five")."""
        s = str(n)
        out = []
        for i in range(0, len(s), 2):
            ch = s[i:i + 2]
            if len(ch) == 1:
                out.append(_UNITS[int(ch)] if ch != "0" else zero)
            elif ch[0] != "0":
                out.append(cls._tens(int(ch[0]), int(ch[1])))
            elif ch[1] != "0":
                out.append(f"{zero} {_UNITS[int(ch[1])]}")
            else:
                out.append(f"{zero} {zero}")
        return ", ".join(out)

    @staticmethod
    def ordinal_of_words(words: str) -> str:
        """inflect ordinal("twenty-one") -> "twenty-first": only the last word changes."""
        m = re.search(r"([a-z]+)$", words)
        if not m:
            return words
        last = m.group(1)
        if last in _ORD_IRREGULAR:
            new = _ORD_IRREGULAR[last]
        elif last.endswith("y"):
            new = last[:-1] + "ieth"
        else:
            new = last + "th"
        return words[: m.start(1)] + new

    @classmethod
    def ordinal_token(cls, tok: str) -> str:
        """inflect number_to_words("21st") -> "twenty-first" (default and-word)."""
        return cls.ordinal_of_words(cls.cardinal(int(re.match(r"[0-9]+", tok).group(0))))


class EnglishTextNormalizer:
    _ABBREV = [("mrs", "misess"), ("mr", "mister"), ("dr", "doctor"), ("st", "saint"), ("co", "company"), ("jr", "junior"),
               ("maj", "major"), ("gen", "general"), ("drs", "doctors"), ("rev", "reverend"), ("lt", "lieutenant"),
               ("hon", "honorable"), ("sgt", "sergeant"), ("capt", "captain"), ("esq", "esquire"), ("ltd", "limited"),
               ("col", "colonel"), ("ft", "fort"), ("etc", "et cetera"), ("btw", "by the way")]   # normalizer.py:20-39

    def __init__(self) -> None:
        self._abbreviations = [(re.compile(r"\b%s\b" % a, re.IGNORECASE), b) for a, b in self._ABBREV]
        self._comma_number_re = re.compile(r"([0-9][0-9\,]+[0-9])")
        self._decimal_number_re = re.compile(r"([0-9]+\.[0-9]+)")
        self._percent_number_re = re.compile(r"([0-9\.\,]*[0-9]+%)")
        self._pounds_re = re.compile(r"£([0-9\,]*[0-9]+)")
        self._dollars_re = re.compile(r"\$([0-9\.\,]*[0-9]+)")
        self._fraction_re = re.compile(r"([0-9]+)/([0-9]+)")
        self._ordinal_re = re.compile(r"[0-9]+(st|nd|rd|th)")
        self._number_re = re.compile(r"[0-9]+")

    def normalize(self, text: str) -> str:
        return self.normalize_numbers(self.expand_abbreviations(text))

    def expand_abbreviations(self, text: str) -> str:
        for rx, rep in self._abbreviations:
            text = rx.sub(rep, text)
        return text

    # --- numbers (normalizer.py:61-144) ---
    def fraction_to_words(self, num: int, den: int) -> str:
        if num == 1 and den == 2:
            return " one half "
        if num == 1 and den == 4:
            return " one quarter "
        if den == 2:
            return " " + _Words.cardinal(num) + " halves "
        if den == 4:
            return " " + _Words.cardinal(num) + " quarters "
        return " " + _Words.cardinal(num) + " " + _Words.ordinal_of_words(_Words.cardinal(den)) + " "

    @staticmethod
    def _expand_dollars(m) -> str:
        match = m.group(1)
        parts = match.split(".")
        if len(parts) > 2:
            return " " + match + " dollars "
        dollars = int(parts[0]) if parts[0] else 0
        cents = int(parts[1]) if len(parts) > 1 and parts[1] else 0
        du, cu = ("dollar" if dollars == 1 else "dollars"), ("cent" if cents == 1 else "cents")
        if dollars and cents:
            return " %s %s, %s %s " % (dollars, du, cents, cu)
        if dollars:
            return " %s %s " % (dollars, du)
        if cents:
            return " %s %s " % (cents, cu)
        return " zero dollars "

    @staticmethod
    def _expand_number(m) -> str:
        num = int(m.group(0))
        if 1000 < num < 3000:
            if num == 2000:
                return " two thousand "
            if 2000 < num < 2010:
                return " two thousand " + _Words.cardinal(num % 100) + " "
            if num % 100 == 0:
                return " " + _Words.cardinal(num // 100) + " hundred "
            return " " + _Words.pairs(num, zero="oh").replace(", ", " ") + " "
        return " " + _Words.cardinal(num, andword="") + " "

    def normalize_numbers(self, text: str) -> str:
        text = self._comma_number_re.sub(lambda m: m.group(1).replace(",", ""), text)
        text = self._pounds_re.sub(r"\1 pounds", text)
        text = self._dollars_re.sub(self._expand_dollars, text)
        text = self._fraction_re.sub(lambda m: self.fraction_to_words(int(m.group(1)), int(m.group(2))), text)
        text = self._decimal_number_re.sub(lambda m: m.group(1).replace(".", " point "), text)
        text = self._percent_number_re.sub(lambda m: m.group(1).replace("%", " percent "), text)
        text = self._ordinal_re.sub(lambda m: " " + _Words.ordinal_token(m.group(0)) + " ", text)
        text = self._number_re.sub(self._expand_number, text)
        return text


_NEEDS = re.compile(r"[0-9$£%]|\b(?:" + "|".join(a for a, _ in EnglishTextNormalizer._ABBREV) + r")\b", re.IGNORECASE)


def needs_normalization(text: str) -> bool:
    """True when normalize() would rewrite the text (digits, currency, percent or one of the abbreviations)."""
    return _NEEDS.search(text) is not None
