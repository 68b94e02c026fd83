"""a cheap attribute path for modules that keep per-iteration state"""


class _FastAttr:
    """nn.Module.__setattr__ costs ~3 us per plain assignment (dict lookups and isinstance checks for parameters, buffers and
    sub-modules); the per-iteration state of the sampler and the network is ~20 assignments per training step, as much host time
    as three kernel launches.  Names listed in `_FAST_ATTRS` (plain per-step state: never a Parameter, a registered buffer or a
    sub-module) go straight to the instance dict."""
    _FAST_ATTRS = frozenset()

    def __setattr__(self, name, value):
        if name in self._FAST_ATTRS:
            self.__dict__[name] = value
        else:
            super().__setattr__(name, value)
