"""`lib.test.utils` of the reference, reduced to TrackerParams (reference lib/test/utils/params.py:5-27)."""


class TrackerParams:
    """Attribute bag for tracker parameters."""

    def set_default_values(self, default_vals: dict):
        for name, val in default_vals.items():
            if not hasattr(self, name):
                setattr(self, name, val)

    def get(self, name: str, *default):
        if len(default) > 1:
            raise ValueError("Can only give one default value.")
        if not default:
            return getattr(self, name)
        return getattr(self, name, default[0])

    def has(self, name: str):
        return hasattr(self, name)
