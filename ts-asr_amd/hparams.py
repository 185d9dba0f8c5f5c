"""Minimal HyperPyYAML reader: enough to load the reference's hparams/LibriSpeechMix/*.yaml UNCHANGED.

``hyperpyyaml`` is not installed on the GPU box, and the reference YAMLs use five of its
tags: ``!new:``, ``!name:``, ``!apply:``, ``!ref`` (with ``<key>`` interpolation and arithmetic such as
``<vocab_size> - 1``) and ``!PLACEHOLDER``, plus tuple-like strings ``(128, 128)``. This reader supports
exactly those. Class paths that point into the reference's Python packages are re-pointed to their MI355X
mirrors through ``ALIASES`` - so ``!new:speechbrain.lobes.features.Fbank`` builds ``ts-asr_amd.nnet.Fbank``.
Paths with no mirror on the hot path (checkpointing, WER statistics, ...) become ``Unavailable``
placeholders that raise only when used (out of scope).
"""
import ast
import functools
import importlib
import re

import yaml

_PKG = __name__.rsplit(".", 1)[0]  # "ts-asr_amd"

ALIASES = {
    "speechbrain.lobes.features.Fbank": _PKG + ".nnet.Fbank",
    "speechbrain.processing.features.InputNormalization": _PKG + ".nnet.InputNormalization",
    "speechbrain.lobes.models.convolution.ConvolutionFrontEnd": _PKG + ".nnet.ConvolutionFrontEnd",
    "speechbrain.lobes.augment.SpecAugment": _PKG + ".nnet.SpecAugment",
    "speechbrain.processing.speech_augmentation.SpeedPerturb": _PKG + ".nnet.SpeedPerturb",
    "speechbrain.processing.speech_augmentation.Resample": _PKG + ".nnet.Resample",
    "models.conformer.ConformerEncoder": _PKG + ".conformer.ConformerEncoder",
    "speechbrain.nnet.linear.Linear": _PKG + ".nnet.Linear",
    "speechbrain.nnet.embedding.Embedding": _PKG + ".nnet.Embedding",
    "speechbrain.nnet.RNN.LSTM": _PKG + ".nnet.LSTM",
    "speechbrain.nnet.transducer.transducer_joint.Transducer_joint": _PKG + ".rnnt.Transducer_joint",
    "speechbrain.nnet.losses.transducer_loss": _PKG + ".rnnt.transducer_loss",
    "speechbrain.nnet.schedulers.NoamScheduler": _PKG + ".core.NoamScheduler",
    "speechbrain.utils.epoch_loop.EpochCounter": _PKG + ".core.EpochCounter",
    "speechbrain.decoders.transducer.TransducerBeamSearcher": _PKG + ".decoders.TransducerBeamSearcher",
}


class Unavailable:
    """Stands for a reference component outside the hot path; raises on first real use."""

    def __init__(self, path, *args, **kwargs):
        self._path, self._args, self._kwargs = path, args, kwargs

    def _fail(self, *a, **k):
        raise NotImplementedError(f"{self._path} is outside the MI355X hot path (SURVEY.md section 2) and has no mirror")

    __call__ = _fail

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        self._fail()

    def __repr__(self):
        return f"Unavailable({self._path})"


def _import(path):
    path = ALIASES.get(path, path)
    mod, _, attr = path.rpartition(".")
    if path.startswith(("speechbrain.", "models.", "hyperpyyaml.")):
        return None
    return getattr(importlib.import_module(mod), attr)


_REF = re.compile(r"<([^<>]+)>")
_ARITH = re.compile(r"^[\s\d\.\+\-\*/\(\)eE]+$")


class _Resolver:
    def __init__(self, root, overrides):
        self.root = root  # yaml MappingNode
        self.top = {k.value: v for k, v in root.value}
        self.cache, self.overrides, self.busy = {}, dict(overrides or {}), set()

    # ---- top-level keys -----------------------------------------------------------------
    def key(self, name):
        if name in self.cache:
            return self.cache[name]
        if name in self.overrides:
            val = self.overrides[name]
        else:
            if name not in self.top:
                raise KeyError(f"!ref <{name}>: no such key")
            if name in self.busy:
                raise ValueError(f"circular !ref through <{name}>")
            self.busy.add(name)
            val = self.node(self.top[name])
            self.busy.discard(name)
        self.cache[name] = val
        return val

    def lookup(self, expr):
        """<a>, <a.b> or <a[b]> forms."""
        m = re.match(r"^([A-Za-z0-9_]+)(.*)$", expr)
        val = self.key(m.group(1))
        for part in re.findall(r"\.([A-Za-z0-9_]+)|\[([^\]]+)\]", m.group(2)):
            k = part[0] or part[1]
            val = val[k] if isinstance(val, dict) else (val[int(k)] if isinstance(val, (list, tuple)) else getattr(val, k))
        return val

    # ---- nodes -----------------------------------------------------------------------------
    def node(self, n):
        tag = n.tag
        if tag == "!PLACEHOLDER":
            raise ValueError("'!PLACEHOLDER' must be replaced via an override")
        if tag == "!ref":
            return self.ref(n.value)
        for prefix in ("!new:", "!name:", "!apply:"):
            if tag.startswith(prefix):
                return self.call(prefix, tag[len(prefix):], n)
        if isinstance(n, yaml.MappingNode):
            return {self.node(k): self.node(v) for k, v in n.value}
        if isinstance(n, yaml.SequenceNode):
            return [self.node(v) for v in n.value]
        val = yaml.SafeLoader.construct_object(_LOADER_FOR_SCALARS, n, deep=True)
        if isinstance(val, str) and re.match(r"^\(.*\)$", val.strip()):
            try:
                return ast.literal_eval(val.strip())
            except (ValueError, SyntaxError):
                pass
        return val

    def ref(self, text):
        text = text.strip()
        m = _REF.fullmatch(text)
        if m:
            return self.lookup(m.group(1))
        sub = _REF.sub(lambda mm: str(self.lookup(mm.group(1))), text)
        if _ARITH.match(sub):
            return eval(compile(ast.parse(sub, mode="eval"), "<ref>", "eval"), {"__builtins__": {}}, {})  # digits/operators only
        return sub

    def call(self, kind, path, n):
        args, kwargs = [], {}
        if isinstance(n, yaml.MappingNode):
            kwargs = {self.node(k): self.node(v) for k, v in n.value}
        elif isinstance(n, yaml.SequenceNode):
            args = [self.node(v) for v in n.value]
        elif n.value not in ("", None):
            args = [yaml.safe_load(n.value)]
        target = _import(path)
        if target is None:
            return Unavailable(path, *args, **kwargs)
        if kind == "!name:":
            return functools.partial(target, *args, **kwargs) if (args or kwargs) else target
        return target(*args, **kwargs)


class _ScalarLoader(yaml.SafeLoader):
    pass


# PyYAML (YAML 1.1) reads "1.e-8" as float only with a sign in the exponent; also accept "1e-8"/"1.e8" like ruamel does.
_ScalarLoader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"^[-+]?(?:[0-9][0-9_]*)(?:\.[0-9_]*)?[eE][-+]?[0-9]+$"), list("-+0123456789"))
_LOADER_FOR_SCALARS = _ScalarLoader("")


def load_hyperpyyaml(yaml_stream, overrides=None):
    """Returns the dict of live objects, like hyperpyyaml.load_hyperpyyaml. ``overrides``: dict or YAML string."""
    text = yaml_stream.read() if hasattr(yaml_stream, "read") else str(yaml_stream)
    if isinstance(overrides, str):
        overrides = yaml.safe_load(overrides) or {}
    loader = _ScalarLoader(text)
    try:
        root = loader.get_single_node()
    finally:
        loader.dispose()
    res = _Resolver(root, overrides)
    out = {}
    for k in res.top:
        if k in res.overrides:
            out[k] = res.key(k)
            continue
        n = res.top[k]
        if n.tag == "!PLACEHOLDER":
            raise ValueError(f"'{k}' is a !PLACEHOLDER and must be replaced via an override")
        out[k] = res.key(k)
    for k, v in res.overrides.items():
        out.setdefault(k, v)
    return out
