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
        if tgt and tgt.startswith("hulc2.") and "hulc2" not in sys.modules:
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


def install_as_hulc2() -> None:
    """Expose this package under the reference's import name so yaml `_target_: hulc2.models...` resolves
    to the MI355X classes.  Call before Hydra instantiates the model."""
    import hulc2_amd

    names = [
        "hulc2_amd", "hulc2_amd.models", "hulc2_amd.models.hulc2", "hulc2_amd.models.perceptual_encoders",
        "hulc2_amd.models.perceptual_encoders.concat_encoders", "hulc2_amd.models.perceptual_encoders.vision_network",
        "hulc2_amd.models.perceptual_encoders.vision_network_gripper", "hulc2_amd.models.perceptual_encoders.vision_r3m",
        "hulc2_amd.models.encoders",
        "hulc2_amd.models.encoders.goal_encoders", "hulc2_amd.models.plan_encoders",
        "hulc2_amd.models.plan_encoders.plan_proposal_net", "hulc2_amd.models.plan_encoders.plan_recognition_net",
        "hulc2_amd.models.decoders", "hulc2_amd.models.decoders.action_decoder", "hulc2_amd.models.decoders.logistic_decoder_rnn",
        "hulc2_amd.models.auxiliary_loss_networks", "hulc2_amd.models.auxiliary_loss_networks.proj_vis_lang",
        "hulc2_amd.utils", "hulc2_amd.utils.distributions",
    ]
    for n in names:
        m = importlib.import_module(n)
        sys.modules["hulc2" + n[len("hulc2_amd"):]] = m
    # conf/model/language_encoder/sbert.yaml:1 names the sentence encoder under the affordance package
    import types
    enc = importlib.import_module("hulc2_amd.models.language_encoders.sbert_lang_encoder")
    chain = ("hulc2", "hulc2.affordance", "hulc2.affordance.models", "hulc2.affordance.models.language_encoders")
    for parent, pkg in zip(chain, chain[1:]):
        if pkg not in sys.modules:
            mod = types.ModuleType(pkg)
            mod.__path__ = []
            sys.modules[pkg] = mod
        setattr(sys.modules[parent], pkg.rsplit(".", 1)[1], sys.modules[pkg])
    sys.modules[chain[-1] + ".sbert_lang_encoder"] = enc
    sys.modules[chain[-1]].sbert_lang_encoder = enc
    _ = hulc2_amd
