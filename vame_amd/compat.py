"""`import vame` for code written against the reference package.

The reference's callers import the hot path by its dotted names -- `import vame` then `vame.train_model(config)` /
`vame.pose_segmentation(config)` (examples/demo.py:48,56), `from vame.model.rnn_vae import RNN_VAE` (vame/model/evaluate.py:20),
`from vame.model.rnn_model import RNN_VAE` (vame/analysis/pose_segmentation.py:24), `from vame.model import SEQUENCE_DATASET`
(vame/model/__init__.py:16).  `install_alias()` registers those names in `sys.modules`, backed by this package, so such code runs
unmodified on the MI355X path.  It is opt-in (call it, or install the one-line `vame/__init__.py` shim that
tools/install_vame_alias.py writes) and it refuses to shadow a reference package that is already imported.

Only what the build implements is aliased (train / embed path and its neighbouring drivers); the reference's other
entry points (init_new_project, motif_videos, community, umap visualisation, gif, csv_to_numpy, egocentric_alignment --
vame/__init__.py:14,19-28) raise an AttributeError that says so instead of pretending to exist.
"""
import importlib
import sys
import types

_MODULES = {
    "vame.model": "vame_amd.model",
    "vame.model.rnn_vae": "vame_amd.model.rnn_vae",
    "vame.model.rnn_model": "vame_amd.model.rnn_model",
    "vame.model.dataloader": "vame_amd.model.dataloader",
    "vame.model.create_training": "vame_amd.model.create_training",
    "vame.model.evaluate": "vame_amd.model.evaluate",
    "vame.analysis": "vame_amd.analysis",
    "vame.analysis.pose_segmentation": "vame_amd.analysis.pose_segmentation",
    "vame.analysis.generative_functions": "vame_amd.analysis.generative_functions",
    "vame.util": "vame_amd.util",
    "vame.util.auxiliary": "vame_amd.util.auxiliary",
}
_TOP_LEVEL = ("create_trainset", "train_model", "evaluate_model", "pose_segmentation", "generative_model")     # vame/__init__.py:15-18,23
_NOT_ON_THE_PATH = ("init_new_project", "motif_videos", "community", "community_videos", "visualization", "gif", "csv_to_numpy",
                    "egocentric_alignment", "update_config")


class _AliasModule(types.ModuleType):
    def __getattr__(self, name):
        if name in _NOT_ON_THE_PATH:
            raise AttributeError(f"vame.{name} is not part of the MI355X train / embed path (vame_amd implements "
                                 f"{', '.join(_TOP_LEVEL)}); call it from the reference package")
        raise AttributeError(f"module 'vame' (vame_amd alias) has no attribute {name!r}")


def is_alias(mod):
    return isinstance(mod, _AliasModule)


def install_alias(replace=False):
    """Make `import vame` (and the dotted hot-path modules) resolve to vame_amd.  replace=True also replaces a `vame` module that is
    already in sys.modules (the shim written by tools/install_vame_alias.py passes it for its own half-initialised entry)."""
    import vame_amd
    have = sys.modules.get("vame")
    if have is not None and not is_alias(have) and not replace:
        raise ImportError("a different `vame` package is already imported; vame_amd.compat.install_alias() will not shadow it "
                          "(import vame_amd before vame, or pass replace=True)")
    top = _AliasModule("vame", doc="vame -> vame_amd alias (vame_amd/compat.py)")
    top.__path__ = []                                  # a package: `import vame.model.rnn_vae` resolves through sys.modules
    top.__version__ = vame_amd.__version__
    top.__vame_amd_alias__ = True
    for name in _TOP_LEVEL:
        setattr(top, name, getattr(vame_amd, name))
    sys.modules["vame"] = top
    for alias, real in _MODULES.items():
        mod = importlib.import_module(real)
        sys.modules[alias] = mod
        parent, _, leaf = alias.rpartition(".")
        if parent == "vame":                           # deeper parents ARE vame_amd's packages: their attributes exist already (and
            setattr(top, leaf, mod)                    # `vame.analysis.pose_segmentation` must stay the function, as in the reference)
    top.auxiliary = sys.modules["vame.util.auxiliary"]    # vame/__init__.py:27
    return top


def uninstall_alias():
    for name in [n for n in sys.modules if n == "vame" or n.startswith("vame.")]:
        if name == "vame" and not is_alias(sys.modules[name]):
            return
        del sys.modules[name]
