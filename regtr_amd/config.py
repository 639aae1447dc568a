"""Config contract of the reference: YAML sections flattened into one dict (utils/misc.py:10-29) and wrapped in an
EasyDict (test.py:53, demo.py:159); model code reads cfg.x, cfg.get('x', d), cfg['x'] and 'x' in cfg."""
import yaml


class Config(dict):
    """Attribute-access dict (stand-in for easydict.EasyDict, which is not a dependency here)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def load_config(path):
    with open(path, 'r') as f:
        cfg = yaml.safe_load(f)
    config = Config()
    for _, section in cfg.items():
        for k, v in section.items():
            config[k] = v
    return config


def as_config(cfg):
    return cfg if isinstance(cfg, Config) else Config(cfg)
