"""YAML config surface (reference utils.py:6-9): attribute access, and a missing key evaluates to an
empty, falsy dict — the shipped YAMLs rely on that (e.g. toponet_vith_256.yaml has no NO_SAM)."""
import yaml


class Config(dict):
    def __getattr__(self, k):
        if k.startswith("__"):                 # copy / pickle / deepcopy protocol probes must see "no such attribute"
            raise AttributeError(k)
        try:
            return self[k]
        except KeyError:
            return Config()

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        return dict(self)


def load_config(path):
    with open(path) as f:
        return Config(yaml.safe_load(f))
