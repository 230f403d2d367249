"""Reads HuDiff checkpoints (the format is the reference's own: a torch.save'd dict).

Envelopes (reference file:line):
  antibody pre-train   {'config','model',...}                         antibody_scripts/antibody_train.py:439-445
  antibody fine-tune   {'fineconfig','pretrain_config','model',...}   antibody_scripts/antibody_finetune.py:348-355
                       (released hudiffab.pt; sample.py:451 reads 'pretrain_config')
  nanobody pre-train   {'config','model',...}                         nanobody_scripts/nanotrain.py:325-331
  nanobody fine-tune   {'config','model','abnativ_params','infilling_params',...}  nanofinetune.py:531-539
                       (released hudiffnb.pt; only the 'infilling_pretrain.*' weights are used for
                       sampling, nanosample.py:185-193, 268-270, 286-287)
The config objects inside are pickled ``easydict.EasyDict`` instances; when that package is absent a
minimal stand-in class is registered so the pickle loads.  torch is used here only as the file reader.
"""
from __future__ import annotations

import sys
import types
from typing import Mapping


class _EasyDict(dict):
    """Attribute-access dict, enough to unpickle easydict.EasyDict payloads."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, Mapping) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        super().__setitem__(k, v)
        super().__setattr__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _ensure_easydict():
    try:
        import easydict  # noqa: F401
    except ImportError:
        mod = types.ModuleType("easydict")
        mod.EasyDict = _EasyDict
        _EasyDict.__module__ = "easydict"
        _EasyDict.__qualname__ = _EasyDict.__name__ = "EasyDict"
        sys.modules["easydict"] = mod


def EasyDict(*a, **k):
    """EasyDict-compatible constructor (the real package's class if installed)."""
    _ensure_easydict()
    return sys.modules["easydict"].EasyDict(*a, **k)


_SAFE_BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray",
                  "complex", "slice", "range", "object"}
# EXACT (module, name) pairs only.  No whole module is admitted: torch.storage._load_from_bytes is torch.load(weights_only=False)
# on an embedded byte string -- an unrestricted nested unpickle -- and torch.serialization.load / torch._utils helpers that
# take a callable are similar gadgets.  What a state_dict + config pickle really names is this short list.
_SAFE_EXACT = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"), ("easydict", "EasyDict"), ("argparse", "Namespace"),
    ("_codecs", "encode"),
    # tensors: torch.save writes _rebuild_tensor_v2(storage, offset, size, stride, requires_grad, hooks) (+ _rebuild_parameter for
    # nn.Parameter values); the storages themselves arrive through persistent_load, which names only a torch.*Storage class
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch.nn.parameter", "Parameter"), ("torch.storage", "TypedStorage"), ("torch.storage", "UntypedStorage"),
    ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"),
    # numpy scalars / arrays inside configs (iteration counters, loss values)
    ("numpy", "dtype"), ("numpy", "ndarray"),
    ("numpy.core.multiarray", "scalar"), ("numpy.core.multiarray", "_reconstruct"),
    ("numpy._core.multiarray", "scalar"), ("numpy._core.multiarray", "_reconstruct"),
}
_TORCH_DTYPE_PREFIXES = ("float", "int", "uint", "bfloat", "complex", "bool", "double", "half", "long", "short")


def _allowed(module, name):
    if module == "builtins":
        return name in _SAFE_BUILTINS
    if (module, name) in _SAFE_EXACT:
        return True
    if module == "torch":          # torch.FloatStorage & co (legacy typed-storage classes), torch.float32 & co (dtype singletons)
        return (name.endswith("Storage") and name[0].isupper() and name.isidentifier()) or \
            (name.startswith(_TORCH_DTYPE_PREFIXES) and name.replace("_", "").isalnum() and name.islower())
    if module == "numpy.dtypes":   # numpy >= 2 pickles dtype classes by name (Float64DType, ...)
        return name.endswith("DType") and name.isidentifier()
    return False


def _restricted_pickle_module():
    """A ``pickle_module`` for torch.load whose Unpickler resolves only tensor-rebuilding helpers, plain containers
    and EasyDict -- an exact (module, name) allow-list: a checkpoint cannot name os.system / builtins.eval / arbitrary
    classes, nor torch.storage._load_from_bytes (a nested unrestricted unpickle).  (torch's own ``weights_only=True``
    unpickler cannot be used: it refuses to fill dict subclasses such as EasyDict.)"""
    import pickle

    class RestrictedUnpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if not _allowed(module, name):
                raise pickle.UnpicklingError(f"checkpoint names {module}.{name}, which is outside the allow-list")
            return super().find_class(module, name)

    mod = types.ModuleType("hudiff_amd._restricted_pickle")
    mod.Unpickler = RestrictedUnpickler
    mod.UnpicklingError = pickle.UnpicklingError
    mod.load = lambda f, **kw: RestrictedUnpickler(f, **kw).load()
    mod.loads = pickle.loads
    mod.dump, mod.dumps, mod.Pickler = pickle.dump, pickle.dumps, pickle.Pickler
    mod.__name__ = "pickle"
    return mod


def load_checkpoint(path, map_location="cpu", trust_pickle=None):
    """torch.load(path) tolerant of EasyDict configs.

    The file is read through a restricted unpickler (an exact allow-list of tensor-rebuilding helpers, plain containers,
    numpy scalars and ``easydict.EasyDict``), so that a ``--ckpt`` from an untrusted source cannot name a callable of its
    choosing (tests/test_host_logic.py drives the known gadgets -- os / builtins callables, torch.storage._load_from_bytes,
    torch.serialization.load -- through it).  A checkpoint that
    carries other pickled classes needs the unrestricted loader, which executes whatever the pickle says: opt in
    with ``trust_pickle=True`` or ``HUDIFF_TRUST_CKPT=1`` for files you trust (the reference always loads that way,
    antibody_scripts/sample.py:448)."""
    import os
    import pickle
    import torch
    _ensure_easydict()
    if trust_pickle is None:
        trust_pickle = os.environ.get("HUDIFF_TRUST_CKPT") == "1"
    kw = {}
    try:
        import inspect
        if "weights_only" in inspect.signature(torch.load).parameters:
            kw["weights_only"] = False          # the restriction is ours (pickle_module); torch < 1.13 has no such keyword
    except (TypeError, ValueError):
        pass
    if trust_pickle:
        return torch.load(path, map_location=map_location, **kw)
    try:
        return torch.load(path, map_location=map_location, pickle_module=_restricted_pickle_module(), **kw)
    except pickle.UnpicklingError as e:
        raise RuntimeError(f"{path}: {e}.  If the file comes from a trusted source, re-run with HUDIFF_TRUST_CKPT=1 "
                           "(unrestricted pickle).") from e


def _strip_module(sd):
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def split_nano_framework_state(model_state):
    """nanosample.py:185-193: {'eval_abnativ_model.*', 'infilling_pretrain.*', ...} -> infilling weights."""
    pre = "infilling_pretrain."
    return {k[len(pre):]: v for k, v in model_state.items() if k.startswith(pre)}


def antibody_model_from_checkpoint(ckpt, ckpt_version="finetune"):
    """sample.py:446-458 -> (config, state_dict, finetune_flag)."""
    if ckpt_version == "pretrain":
        config, finetune = ckpt["config"], False
    elif ckpt_version == "finetune":
        config, finetune = ckpt["pretrain_config"], True
    else:
        raise ValueError("ckpt version has not existed.")
    return config, _strip_module(ckpt["model"]), finetune


def nanobody_model_from_checkpoint(ckpt, model="finetune_vh"):
    """nanosample.py:252-288 -> (name, model_params, state_dict)."""
    if model == "pretrain":
        cfg = ckpt["config"]
        return cfg["name"] if "name" in cfg else "nano", dict(cfg["model"]), _strip_module(ckpt["model"])
    if model == "finetune_vh":
        return "nano", dict(ckpt["infilling_params"]), split_nano_framework_state(_strip_module(ckpt["model"]))
    raise ValueError(model)
