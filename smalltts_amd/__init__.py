"""smalltts_amd — MI355X-native engine for the smalltts synthesis hot path.
`from smalltts_amd import SmallTTS` mirrors `from smalltts import SmallTTS` (reference
src/smalltts/__init__.py:1-6: lazy attribute so importing the package stays cheap)."""


def __getattr__(name):
    if name in ("SmallTTS", "estimate_duration", "Encoder", "Decoder"):
        from . import api
        return getattr(api, name)
    raise AttributeError(name)
