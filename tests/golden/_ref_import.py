"""Import shims that let the reference's kernel-level modules load in the build
container (no GPU, several optional deps absent).  Used ONLY by
``make_golden.py``; nothing here travels into the tests or the product.

Recipe from SURVEY.md §8c: stub absent third-party modules, give ``msgspec`` a
dataclass-backed ``Struct``, fake the two torch.cuda device queries that
``sglang.srt.utils.common`` makes at import time on a ROCm torch build, and run
Triton in interpreter mode on CPU tensors.
"""
import dataclasses
import importlib.abc
import importlib.machinery
import os
import sys
import types

REF_PY = "/root/reference/python"


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        val = _Stub(f"{self.__name__}.{name}")
        setattr(self, name, val)
        return val

    def __call__(self, *a, **k):
        return _Stub(self.__name__ + "()")


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("orjson", "pybase64", "IPython", "zmq", "torchvision", "pynvml",
             "setproctitle", "uvloop", "partial_json_parser", "interegular",
             "outlines", "xgrammar", "llguidance", "openai", "tiktoken", "PIL",
             "cv2", "decord", "soundfile", "scipy_stub")

    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _install_msgspec():
    if "msgspec" in sys.modules:
        return
    m = types.ModuleType("msgspec")

    class _Meta(type):
        def __new__(mcls, name, bases, ns, **kw):
            cls = super().__new__(mcls, name, bases, ns)
            if name != "Struct":
                cls = dataclasses.dataclass(
                    frozen=kw.get("frozen", False), kw_only=kw.get("kw_only", False)
                )(cls)
            return cls

        def __init__(cls, name, bases, ns, **kw):
            super().__init__(name, bases, ns)

    class Struct(metaclass=_Meta):
        pass

    def field(default=dataclasses.MISSING, default_factory=dataclasses.MISSING, name=None):
        if default_factory is not dataclasses.MISSING:
            return dataclasses.field(default_factory=default_factory)
        if default is not dataclasses.MISSING:
            return dataclasses.field(default=default)
        return dataclasses.field()

    m.Struct = Struct
    m.field = field
    m.json = _Stub("msgspec.json")
    m.msgpack = _Stub("msgspec.msgpack")
    m.structs = _Stub("msgspec.structs")
    m.UNSET = None
    m.UnsetType = type(None)
    sys.modules["msgspec"] = m


def install():
    os.environ.setdefault("TRITON_INTERPRET", "1")
    if REF_PY not in sys.path:
        sys.path.insert(0, REF_PY)
    sys.meta_path.append(_StubFinder())
    _install_msgspec()
    import torch

    class _Props:
        gcnArchName = "gfx950:sramecc+:xnack-"
        multi_processor_count = 256
        name = "AMD Instinct MI355X"
        total_memory = 288 * 2**30
        major, minor = 9, 5
        warp_size = 64

    torch.cuda.get_device_properties = lambda *a, **k: _Props()
    torch.cuda.get_device_capability = lambda *a, **k: (9, 5)
