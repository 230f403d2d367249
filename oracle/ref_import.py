"""TEST INFRASTRUCTURE ONLY -- import harness for the read-only reference tree.

Used by ``oracle/make_golden.py`` (and by ``tests/test_oracle_vs_reference.py`` when
``/root/reference`` exists) to import the reference's own ``AntiTFNet`` /
``NanoAntiTFNet`` in THIS container.  Nothing here travels to the GPU box as
anything but a dead file: ``/root/reference`` does not exist there.

Two things are needed to make ``import model.encoder.model`` work:

1. permissive dummy modules for third-party packages that the reference imports at
   module-import time but never executes on the sampling path (pymol, abnumber, Bio,
   seaborn, anarci, lmdb, easydict, ...);
2. a restatement of the three classes HuDiff takes from the un-vendored, un-pinned
   PyPI package ``sequence-models`` (environment.yaml:23): ``PositionFeedForward``,
   ``MaskedConv1d`` and ``ByteNetBlock``.  Their semantics are restated from the
   published upstream package (microsoft/protein-sequence-models); the parameter
   names/shapes are pinned by the checkpoint contract in SURVEY.md App. B.
   **Parity at this boundary is "unpinned"**: no reference test or golden vector
   covers it, so the restatement below is the definition the oracle is held to.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import sys
import types

REFERENCE_ROOT = "/root/reference"

_DUMMY_ROOTS = (
    "pymol", "abnumber", "Bio", "seaborn", "anarci", "lmdb", "easydict",
    "pkg_resources", "matplotlib", "sklearn", "tensorboard", "esm", "apex",
)
_DUMMY_EXACT = ("dataset.abnativ_alignment.align_and_clean",)


class _AnythingMeta(type):
    """Class-level attribute sink (``from Bio.Align import substitution_matrices; substitution_matrices.load``)."""

    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()


class _Anything(metaclass=_AnythingMeta):
    """Attribute sink: any attribute / call / subscript yields another sink."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __getitem__(self, item):
        return _Anything()

    def __iter__(self):
        return iter(())


class _DummyModule(types.ModuleType):
    __path__: list = []

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything


class _DummyFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def __init__(self):
        self._real = {}

    def _really_importable(self, root):
        if root not in self._real:
            self._real[root] = any(
                f is not self and getattr(f, "find_spec", None) and f.find_spec(root, None)
                for f in sys.meta_path)
        return self._real[root]

    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if fullname in _DUMMY_EXACT or (root in _DUMMY_ROOTS and not self._really_importable(root)):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _DummyModule(spec.name)

    def exec_module(self, module):
        pass


def _install_sequence_models():
    """Restatement of sequence_models.{layers,convolutional} (see module docstring)."""
    import torch.nn as nn

    class PositionFeedForward(nn.Module):
        def __init__(self, d_in, d_out, rank=None):
            super().__init__()
            assert rank is None, "HuDiff always passes rank=None"
            self.conv = nn.Conv1d(d_in, d_out, 1)

        def forward(self, x):
            return self.conv(x.transpose(1, 2)).transpose(1, 2)

    class MaskedConv1d(nn.Conv1d):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1,
                     groups=1, bias=True):
            padding = dilation * (kernel_size - 1) // 2
            super().__init__(in_channels, out_channels, kernel_size, stride=stride,
                             dilation=dilation, groups=groups, bias=bias, padding=padding)

        def forward(self, x, input_mask=None):
            if input_mask is not None:
                x = x * input_mask
            return super().forward(x.transpose(1, 2)).transpose(1, 2)

    class ByteNetBlock(nn.Module):
        def __init__(self, d_in, d_h, d_out, kernel_size, dilation=1, groups=1, causal=False,
                     activation="relu", rank=None):
            super().__init__()
            assert not causal and rank is None and groups == 1
            self.conv = MaskedConv1d(d_h, d_h, kernel_size=kernel_size, dilation=dilation,
                                     groups=groups)
            act = {"relu": nn.ReLU, "gelu": nn.GELU}[activation]
            self.sequence1 = nn.Sequential(
                nn.LayerNorm(d_in), act(), PositionFeedForward(d_in, d_h, rank=rank),
                nn.LayerNorm(d_h), act())
            self.sequence2 = nn.Sequential(
                nn.LayerNorm(d_h), act(), PositionFeedForward(d_h, d_out, rank=rank))

        def forward(self, x, input_mask=None):
            return x + self.sequence2(self.conv(self.sequence1(x), input_mask=input_mask))

    class DoubleEmbedding(nn.Module):  # never instantiated by HuDiff (n_frozen_embs=None)
        def __init__(self, *a, **k):
            raise NotImplementedError

    pkg = types.ModuleType("sequence_models")
    pkg.__path__ = []
    layers = types.ModuleType("sequence_models.layers")
    layers.PositionFeedForward = PositionFeedForward
    layers.DoubleEmbedding = DoubleEmbedding
    conv = types.ModuleType("sequence_models.convolutional")
    conv.ByteNetBlock = ByteNetBlock
    conv.MaskedConv1d = MaskedConv1d
    sys.modules["sequence_models"] = pkg
    sys.modules["sequence_models.layers"] = layers
    sys.modules["sequence_models.convolutional"] = conv


_installed = False


def install():
    """Make ``from model.encoder.model import AntiTFNet`` importable. Idempotent."""
    global _installed
    if _installed:
        return
    sys.dont_write_bytecode = True  # /root/reference is read-only
    sys.meta_path.insert(0, _DummyFinder())
    _install_sequence_models()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def reference_models():
    """-> (AntiTFNet, NanoAntiTFNet) classes of the reference."""
    install()
    from model.encoder.model import AntiTFNet          # /root/reference/model/encoder/model.py:325
    from model.nanoencoder.model import NanoAntiTFNet  # /root/reference/model/nanoencoder/model.py:290
    return AntiTFNet, NanoAntiTFNet


def reference_tables():
    """Slot / mask / region tables read with ast (the modules themselves import lmdb/Bio).

    dataset/preprocess.py:195-362 and dataset/oas_pair_dataset_new.py:25-40.
    """
    import ast
    out = {}
    wanted = {
        "dataset/preprocess.py": [
            "HEAVY_POSITIONS_dict", "LIGHT_POSITIONS_dict", "HEAVY_CDR_INDEX", "LIGHT_CDR_INDEX",
            "HEAVY_CDR_KABAT_NO_VERNIER", "LIGHT_CDR_KABAT_NO_VERNIER", "INPAINT_HEAVY_CDR_INDEX"],
        "dataset/oas_pair_dataset_new.py": ["HEAVY_REGION_INDEX", "LIGHT_REGION_INDEX"],
    }
    for rel, names in wanted.items():
        tree = ast.parse(open(f"{REFERENCE_ROOT}/{rel}").read())
        for node in tree.body:
            if isinstance(node, ast.Assign) and len(node.targets) == 1 and \
                    isinstance(node.targets[0], ast.Name) and node.targets[0].id in names:
                out[node.targets[0].id] = ast.literal_eval(node.value)
    return out
