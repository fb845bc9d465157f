from . import lpip  # noqa
