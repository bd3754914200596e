"""Import the reference's PyTorch modules from /root/reference IN THIS CONTAINER ONLY
(golden-vector generation). The reference's decorative import-time deps that are not
installed here (jaxtyping, beartype, phonemizer, inflect, onnxruntime) are replaced by
in-process stubs; none of them touches the math. Never used on the GPU box."""
import sys
import types

REF_SRC = "/root/reference/src"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Ann:
    def __class_getitem__(cls, item):
        return cls


def install_stubs():
    if "jaxtyping" not in sys.modules:
        def jaxtyped(fn=None, *, typechecker=None):
            if fn is None:
                return lambda f: f
            return fn
        _stub("jaxtyping", Float=_Ann, Bool=_Ann, Int=_Ann, Int64=_Ann, jaxtyped=jaxtyped,
              config=types.SimpleNamespace(update=lambda *a, **k: None))
    if "beartype" not in sys.modules:
        _stub("beartype", beartype=lambda f: f)
    if "phonemizer" not in sys.modules:
        class EspeakBackend:
            def __init__(self, *a, **k):
                pass

            def phonemize(self, texts):
                raise RuntimeError("espeak is not available in this container")
        _stub("phonemizer")
        _stub("phonemizer.backend", EspeakBackend=EspeakBackend)
        _stub("phonemizer.logger", get_logger=lambda **k: None)
    if "inflect" not in sys.modules:
        class _Eng:
            def __getattr__(self, n):
                return lambda *a, **k: ""
        _stub("inflect", engine=lambda: _Eng())
    if "onnxruntime" not in sys.modules:
        _stub("onnxruntime", InferenceSession=object, SessionOptions=object,
              GraphOptimizationLevel=types.SimpleNamespace(ORT_ENABLE_ALL=0),
              get_available_providers=lambda: ["CPUExecutionProvider"])
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)


def import_reference():
    install_stubs()
    import logging
    lvl = logging.getLogger().level
    from smalltts.models.backbone.model import DiTModel  # noqa
    import smalltts.infer.onnx as ref_infer  # noqa
    import smalltts.train.utils as ref_train_utils  # noqa
    import smalltts.data.phonemization.phonemes as ref_ph  # noqa
    logging.getLogger().setLevel(lvl)
    return DiTModel, ref_infer, ref_train_utils, ref_ph
