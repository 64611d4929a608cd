"""Configuration tree for the hot path without Hydra/OmegaConf.

The reference reads one flat YAML through ``@hydra.main`` and overrides it from the command line with ``a.b=c``
(main_h3wb.py:567, config/config.yaml).  ``load()`` gives the same attribute tree (``args.model.number_of_frames``,
``args.ft2d.scale`` ...) from that YAML file when it is given, or from the defaults of the keys this package
reads, and applies the same override syntax.
"""
from types import SimpleNamespace

# the keys D3DP / the harness read (common/diffusionpose.py:62-103,140-153; main_h3wb.py:306,685-688) with the
# values of the reference configuration
DEFAULTS = {
    "general": {"part_based_model": True, "evaluate": "best_epoch.bin", "checkpoint": "", "checkpoint_frequency": 20},
    "data": {"dataset": "h3wb", "num_kps": 134, "merge_hands": True, "subjects_train": "S1,S5,S6,S7",
             "subjects_test": "S8", "actions": "*"},
    "experiment": {"downsample": 1, "gpu": "0"},
    "model": {"diff_model": "MixSTE2", "number_of_frames": 27, "stride": 27, "batch_size": 1024,
              "test_time_augmentation": True, "input_size": 5, "dep": 8, "cs": 288, "epochs": 400,
              "data_augmentation": True, "learning_rate": 0.00006, "lr_decay": 0.993, "wb_loss": False,
              "mse_loss": False, "weighted_loss": False},
    "ft2d": {"scale": 1.0, "timestep": 1000, "sampling_timesteps": 5, "num_proposals": 10, "debug": False, "p2": False},
}


def _parse(text):
    low = text.strip().lower()
    if low in ("true", "false"):
        return low == "true"
    if low in ("null", "none", "~"):
        return None
    for cast in (int, float):
        try:
            return cast(text)
        except ValueError:
            pass
    return text.strip("'\"")


def _namespace(tree):
    return SimpleNamespace(**{k: _namespace(v) if isinstance(v, dict) else v for k, v in tree.items()})


def load(path=None, overrides=()):
    """YAML file (optional) + ``section.key=value`` overrides -> nested SimpleNamespace."""
    tree = {k: dict(v) for k, v in DEFAULTS.items()}
    if path is not None:
        import yaml
        with open(path) as f:
            for section, values in (yaml.safe_load(f) or {}).items():
                if isinstance(values, dict):
                    tree.setdefault(section, {}).update(values)
                else:
                    tree[section] = values
    for item in overrides:
        dotted, _, value = item.partition("=")
        if not _ or "." not in dotted:
            raise ValueError(f"override {item!r} is not of the form section.key=value")
        node = tree
        *parents, leaf = dotted.split(".")
        for part in parents:
            node = node.setdefault(part, {})
        node[leaf] = _parse(value)
    return _namespace(tree)
