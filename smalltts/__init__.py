"""Drop-in import surface of the reference package (`src/smalltts/__init__.py:1-6`): `smalltts.SmallTTS` resolves lazily, and the
module paths the reference's scripts import — `smalltts.infer.onnx`, `smalltts.codec.onnx`, `smalltts.infer.utils`,
`smalltts.data.phonemization.phonemes`, `smalltts.assets.ensure` — re-export the MI355X implementation in `smalltts_amd`.
Re-exports only: this package is boundary, not a component."""


def __getattr__(name):
    if name == "SmallTTS":
        from .infer.onnx import SmallTTS

        return SmallTTS
    raise AttributeError(name)
