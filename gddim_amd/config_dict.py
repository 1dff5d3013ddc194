"""Minimal stand-in for `ml_collections.ConfigDict` (not installed in this image).

Only what the reference's config files and run drivers use: nested attribute access, assignment,
`lock()` (cld_jax/main.py:33-34 defines the flag with lock_config=True, so overrides may only touch
existing keys) and `--config.a.b=v` style overrides."""
import ast


class ConfigDict(object):
    def __init__(self, initial=None):
        object.__setattr__(self, "_fields", {})
        object.__setattr__(self, "_locked", False)
        if initial:
            for k, v in initial.items():
                setattr(self, k, ConfigDict(v) if isinstance(v, dict) else v)

    def __getattr__(self, name):
        try:
            return self._fields[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if self._locked and name not in self._fields:
            raise KeyError("config is locked; unknown key %r" % name)
        self._fields[name] = value

    __getitem__ = __getattr__
    __setitem__ = __setattr__

    def __contains__(self, name):
        return name in self._fields

    def get(self, name, default=None):
        """dict-style access with a default, as ml_collections.ConfigDict offers."""
        return self._fields.get(name, default)

    def keys(self):
        return self._fields.keys()

    def items(self):
        return self._fields.items()

    def lock(self):
        object.__setattr__(self, "_locked", True)
        for v in self._fields.values():
            if isinstance(v, ConfigDict):
                v.lock()
        return self

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, ConfigDict) else v) for k, v in self._fields.items()}

    def copy_and_resolve_references(self):
        return ConfigDict(self.to_dict())

    def override(self, dotted, value):
        """`--config.sampling.nfe=50` -> cfg.override('sampling.nfe', '50')."""
        parts = dotted.split(".")
        node = self
        for p in parts[:-1]:
            node = getattr(node, p)
        if parts[-1] not in node:
            raise KeyError("unknown config key %r" % dotted)
        if isinstance(value, str):
            try:
                value = ast.literal_eval(value)
            except (ValueError, SyntaxError):
                pass
        setattr(node, parts[-1], value)
        return self

    def __repr__(self):
        return "ConfigDict(%r)" % self.to_dict()
