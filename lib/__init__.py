"""Drop-in `lib` package: the part of the reference's lib/ tree that the per-frame path is reached through
(lib.registry, lib.models, lib.config.uvltrack.config, lib.utils.misc), backed by uvltrack_amd."""
