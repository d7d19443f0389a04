"""Name -> builder registries with the reference's surface (lib/registry.py:6-49):
`Registry.register(name)` as a decorator, `Registry.register(name, obj)` as a call, dict access to look up."""


class Registry(dict):
    def register(self, module_name, module=None):
        def _add(obj):
            if module_name in self:
                raise AssertionError("%s is already registered" % module_name)
            self[module_name] = obj
            return obj
        if module is not None:
            _add(module)
            return None
        return _add


ACTORS = Registry()
MODELS = Registry()
BACKBONES = Registry()
HEADS = Registry()
LOSSES = Registry()
