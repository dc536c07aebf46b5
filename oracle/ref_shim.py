"""TEST INFRASTRUCTURE ONLY -- import shim for the upstream reference package.

Works ONLY in the build container (where /root/reference exists); it is used by
oracle/gen_golden.py to produce the fixtures under tests/golden/ and by the
`-m "not gpu"` test that pins oracle/restate.py against the live reference.
Nothing on the GPU box imports this module (the reference does not travel).

The reference (`/root/reference/ssdn/ssdn/__init__.py:1-3`) pulls in torchvision,
h5py, colorlog ... which are not installed; they are not needed by the hot path
(models/, denoiser.py), so empty stub modules are registered for them.
"""
import os
import sys
import types

REF_ROOT = "/root/reference/ssdn"


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "ssdn"))


class _AnyThing:
    """Callable/instantiable placeholder for symbols of packages that are not installed."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None


class _StubModule(types.ModuleType):
    def __getattr__(self, item):  # any missing symbol resolves to a placeholder class
        if item.startswith("__"):
            raise AttributeError(item)
        return _AnyThing


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = _StubModule(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave like a package so sub-imports resolve
    sys.modules[name] = m
    return m


def import_reference():
    """Return the reference's `ssdn` package (imported from REF_ROOT)."""
    if not available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    import numpy as np

    if not hasattr(np, "int"):
        np.int = int  # utils/n2v_ups.py:73 uses the removed alias
    sys.dont_write_bytecode = True  # the mount is read-only

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return None

    tv = _stub("torchvision")
    tv.utils = _stub("torchvision.utils", make_grid=lambda *a, **k: None)
    tv.transforms = _stub("torchvision.transforms", RandomCrop=_Any)
    tv.transforms.functional = _stub("torchvision.transforms.functional")
    tv.datasets = _stub("torchvision.datasets")
    tv.datasets.folder = _stub(
        "torchvision.datasets.folder",
        has_file_allowed_extension=lambda *a, **k: True,
        is_image_file=lambda *a, **k: True,
        IMG_EXTENSIONS=(),
    )
    _stub("h5py")
    _stub("imagesize")
    _stub("overrides", EnforceOverrides=object, overrides=lambda f: f)
    _stub("colorlog", ColoredFormatter=_Any)
    _stub("colored_traceback", Colorizer=_Any)
    import torch.utils  # noqa

    _stub("torch.utils.tensorboard", SummaryWriter=_Any)
    _stub("torch.utils.tensorboard.writer", SummaryWriter=_Any)
    _stub("nptyping", Array=_Any)

    # make sure OUR `ssdn` (same import name) is not the one that resolves -- and put it back afterwards, so that objects
    # already imported from it keep their identity (pickling checks `sys.modules[cls.__module__].<name> is cls`)
    ours = {k: v for k, v in sys.modules.items() if k == "ssdn" or k.startswith("ssdn.")}
    for k in ours:
        del sys.modules[k]
    sys.path.insert(0, REF_ROOT)
    try:
        import ssdn  # noqa
    finally:
        sys.path.remove(REF_ROOT)
    ref = sys.modules["ssdn"]
    # detach so our own package can be imported afterwards under the same name
    mods = {k: v for k, v in sys.modules.items() if k == "ssdn" or k.startswith("ssdn.")}
    for k in mods:
        del sys.modules[k]
    sys.modules.update(ours)
    ref._all_modules = mods
    return ref


class reference_modules:
    """Context manager: temporarily put the reference's modules back in sys.modules
    (needed while unpickling / running code that does `import ssdn` lazily)."""

    def __init__(self, ref):
        self.ref = ref

    def __enter__(self):
        self.saved = {k: v for k, v in sys.modules.items() if k == "ssdn" or k.startswith("ssdn.")}
        for k in self.saved:
            del sys.modules[k]
        sys.modules.update(self.ref._all_modules)
        return self.ref

    def __exit__(self, *exc):
        for k in list(sys.modules):
            if k == "ssdn" or k.startswith("ssdn."):
                del sys.modules[k]
        sys.modules.update(self.saved)
