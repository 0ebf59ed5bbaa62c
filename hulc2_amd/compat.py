"""Stand-ins for the third-party layers around the hot path (Hydra, OmegaConf, PyTorch-Lightning).

The reference drives `Hulc2` through Hydra `_target_` instantiation and Lightning's hook protocol
(SURVEY.md §8b).  When those packages are importable they are used as-is; in images without them (this
build container, the GPU box) the minimal equivalents below keep the same call signatures so the model,
the tests and bench.py run unchanged.  Nothing here does arithmetic.
"""
import importlib
import sys
from typing import Any, Dict

import torch
import torch.nn as nn

try:  # pragma: no cover - exercised only where hydra exists
    import hydra as _hydra
    from omegaconf import DictConfig, OmegaConf  # noqa: F401
    HAVE_HYDRA = True
except Exception:  # noqa: BLE001
    _hydra = None
    HAVE_HYDRA = False

try:  # pragma: no cover
    import pytorch_lightning as _pl
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    _pl = None
    HAVE_LIGHTNING = False


class Config(dict):
    """dict with attribute access — the subset of DictConfig behaviour the model constructors use
    (`cfg.key`, `cfg["key"]`, `"key" in cfg`, item/attribute assignment in setup_input_sizes)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(x):
        if isinstance(x, dict) and not isinstance(x, Config):
            return Config({k: Config.wrap(v) for k, v in x.items()})
        if isinstance(x, (list, tuple)):
            return type(x)(Config.wrap(v) for v in x)
        return x


# reference class paths -> this package (used when the name `hulc2` is not installed as an alias)
_ALIAS = {"hulc2.": "hulc2_amd."}


def _locate(path: str):
    mod, _, attr = path.rpartition(".")
    try:
        return getattr(importlib.import_module(mod), attr)
    except (ImportError, AttributeError):
        for src, dst in _ALIAS.items():
            if path.startswith(src):
                mod2 = dst + mod[len(src):]
                return getattr(importlib.import_module(mod2), attr)
        raise


def instantiate(cfg, *args, **kwargs):
    """hydra.utils.instantiate for `_target_` configs (non-recursive, like `_recursive_: false`)."""
    if cfg is None:
        return None
    if HAVE_HYDRA and not isinstance(cfg, Config) and type(cfg).__name__ in ("DictConfig",):
        tgt = cfg.get("_target_")
        if tgt and tgt.startswith("hulc2.") and not any(isinstance(f, _LeafAliasFinder) for f in sys.meta_path):
            install_as_hulc2()
        return _hydra.utils.instantiate(cfg, *args, **kwargs)
    if not isinstance(cfg, dict):
        raise TypeError(f"cannot instantiate from {type(cfg)}")
    if not cfg or "_target_" not in cfg:
        return None
    params = {k: v for k, v in cfg.items() if not k.startswith("_")}
    params.update(kwargs)
    return _locate(cfg["_target_"])(*args, **params)


class _MiniLightningModule(nn.Module):
    """The slice of pl.LightningModule's surface Hulc2 touches: log, device, save_hyperparameters, trainer."""

    def __init__(self):
        super().__init__()
        self.logged: Dict[str, Any] = {}
        self.trainer = None
        self.hparams = Config()

    @property
    def device(self) -> torch.device:
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    def log(self, name, value, **kwargs):
        self.logged[name] = value.detach() if isinstance(value, torch.Tensor) else value

    def save_hyperparameters(self, *a, **k):
        pass

    def print(self, *a, **k):
        print(*a, **k)


LightningModule = _pl.LightningModule if HAVE_LIGHTNING else _MiniLightningModule


# Reference leaf module -> module of this package.  ONLY leaves are aliased: the reference's own packages
# (`hulc2`, `hulc2.utils`, `hulc2.datasets`, `hulc2.models`, `hulc2.affordance`, ...) stay what they are, so everything
# else `hulc2/training.py:20-25,40,95` imports (utils.utils, kl_callbacks, the data modules, the rollout callbacks,
# the transforms) keeps coming from the reference tree.
LEAF_ALIASES = {
    "hulc2.models.hulc2": "hulc2_amd.models.hulc2",                                       # conf/model/calvin_hulc++.yaml:14
    "hulc2.models.perceptual_encoders.concat_encoders": "hulc2_amd.models.perceptual_encoders.concat_encoders",
    "hulc2.models.perceptual_encoders.vision_network": "hulc2_amd.models.perceptual_encoders.vision_network",
    "hulc2.models.perceptual_encoders.vision_network_gripper": "hulc2_amd.models.perceptual_encoders.vision_network_gripper",
    "hulc2.models.perceptual_encoders.vision_r3m": "hulc2_amd.models.perceptual_encoders.vision_r3m",
    "hulc2.models.encoders.goal_encoders": "hulc2_amd.models.encoders.goal_encoders",
    "hulc2.models.plan_encoders.plan_proposal_net": "hulc2_amd.models.plan_encoders.plan_proposal_net",
    "hulc2.models.plan_encoders.plan_recognition_net": "hulc2_amd.models.plan_encoders.plan_recognition_net",
    "hulc2.models.decoders.action_decoder": "hulc2_amd.models.decoders.action_decoder",
    "hulc2.models.decoders.logistic_decoder_rnn": "hulc2_amd.models.decoders.logistic_decoder_rnn",
    "hulc2.models.auxiliary_loss_networks.proj_vis_lang": "hulc2_amd.models.auxiliary_loss_networks.proj_vis_lang",
    "hulc2.utils.distributions": "hulc2_amd.utils.distributions",
    "hulc2.affordance.models.language_encoders.sbert_lang_encoder": "hulc2_amd.models.language_encoders.sbert_lang_encoder",
    "hulc2.affordance.pixel_aff_lang_detector": "hulc2_amd.affordance.pixel_aff_lang_detector",   # conf/affordance/aff_detection/r3m.yaml
}


def _alias_parents():
    out = set()
    for leaf in LEAF_ALIASES:
        parts = leaf.split(".")
        for n in range(1, len(parts)):
            out.add(".".join(parts[:n]))
    return out


class _AliasLoader:
    """Loader that hands out an already imported hulc2_amd module under the reference's leaf name."""

    def __init__(self, target):
        self.target = target
        self._spec = None

    def create_module(self, spec):
        mod = importlib.import_module(self.target)
        self._spec = getattr(mod, "__spec__", None)
        return mod

    def exec_module(self, module):
        # importlib stamped the alias spec on the module; put its own back (reload / pickling by its real name)
        if self._spec is not None:
            module.__spec__ = self._spec


class _LeafAliasFinder:
    """First on sys.meta_path: the aliased leaves resolve to this package whatever `hulc2` tree is importable.
    Parents are imported by the normal machinery first (the reference's packages when present), and the leaf is
    bound as an attribute of its parent like any other submodule."""

    def find_spec(self, fullname, path=None, target=None):
        tgt = LEAF_ALIASES.get(fullname)
        if tgt is None:
            return None
        from importlib.machinery import ModuleSpec
        return ModuleSpec(fullname, _AliasLoader(tgt), origin="alias:" + tgt)


class _EmptyParentFinder:
    """Last on sys.meta_path: when NO `hulc2` tree is importable (this image, the GPU box) the parents of the aliased
    leaves become empty packages, so `import hulc2.models.hulc2` still resolves."""

    _parents = None

    def find_spec(self, fullname, path=None, target=None):
        if _EmptyParentFinder._parents is None:
            _EmptyParentFinder._parents = _alias_parents()
        if fullname not in _EmptyParentFinder._parents:
            return None
        from importlib.machinery import ModuleSpec
        spec = ModuleSpec(fullname, _EmptyPackageLoader(), origin="hulc2_amd-empty-parent", is_package=True)
        spec.submodule_search_locations = []
        return spec


class _EmptyPackageLoader:
    def create_module(self, spec):
        return None

    def exec_module(self, module):
        pass


def install_as_hulc2() -> None:
    """Make the reference's class paths (yaml `_target_: hulc2.models...`) resolve to the MI355X classes.

    Works before or after the reference's `hulc2` package has been imported, and leaves that package alone: only the
    leaf modules of LEAF_ALIASES are replaced.  Call it once before Hydra instantiates the model."""
    if not any(isinstance(f, _LeafAliasFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _LeafAliasFinder())
    if not any(isinstance(f, _EmptyParentFinder) for f in sys.meta_path):
        sys.meta_path.append(_EmptyParentFinder())
    importlib.invalidate_caches()
    # leaves (or parents) that were imported before this call: swap the leaf in place
    for leaf, tgt in LEAF_ALIASES.items():
        parent, _, name = leaf.rpartition(".")
        have = sys.modules.get(leaf)
        if have is not None and getattr(have, "__name__", "") != tgt:
            sys.modules[leaf] = importlib.import_module(tgt)
        if leaf in sys.modules and parent in sys.modules:
            setattr(sys.modules[parent], name, sys.modules[leaf])
        # names the reference's packages re-export from an aliased leaf (hulc2/models/__init__.py:7 SBertLang)
    ref_models = sys.modules.get("hulc2.models")
    if ref_models is not None and hasattr(ref_models, "lang_encoders") and isinstance(ref_models.lang_encoders, dict):
        enc = importlib.import_module(LEAF_ALIASES["hulc2.affordance.models.language_encoders.sbert_lang_encoder"])
        ref_models.SBertLang = enc.SBertLang
        ref_models.lang_encoders["sbert"] = enc.SBertLang


def uninstall_as_hulc2() -> None:
    """Undo install_as_hulc2 (tests)."""
    sys.meta_path[:] = [f for f in sys.meta_path if not isinstance(f, (_LeafAliasFinder, _EmptyParentFinder))]
    for leaf in LEAF_ALIASES:
        m = sys.modules.get(leaf)
        if m is not None and getattr(m, "__name__", "").startswith("hulc2_amd"):
            del sys.modules[leaf]
    for parent in _alias_parents():
        m = sys.modules.get(parent)
        if m is not None and getattr(getattr(m, "__spec__", None), "origin", None) == "hulc2_amd-empty-parent":
            del sys.modules[parent]
