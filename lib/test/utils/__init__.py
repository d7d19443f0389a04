"""`lib.test.utils.TrackerParams`: the attribute bag that lib/test/parameter/uvltrack.py:19-48 of the reference fills
(cfg, template / search factor and size, checkpoint path) and the tracker reads (reference lib/test/utils/params.py:5-27)."""

_MISSING = object()


class TrackerParams:
    def set_default_values(self, default_vals: dict):
        """Only fills attributes that are not set yet."""
        for key in default_vals:
            if not self.has(key):
                setattr(self, key, default_vals[key])

    def has(self, name: str) -> bool:
        return name in vars(self) or hasattr(type(self), name)

    def get(self, name: str, *default):
        if len(default) > 1:
            raise ValueError("Can only give one default value.")
        fallback = default[0] if default else _MISSING
        value = getattr(self, name, fallback)
        if value is _MISSING:
            raise AttributeError(name)
        return value
