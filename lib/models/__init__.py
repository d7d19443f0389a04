# same trick as the reference (lib/models/__init__.py:1): `lib.models.uvltrack` resolves to the MODULE
# lib/models/uvltrack/uvltrack.py, which is what tracking/profile_model.py:66-67 relies on.
from .uvltrack import uvltrack  # noqa: F401
