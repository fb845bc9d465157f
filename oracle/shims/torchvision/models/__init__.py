from . import detection  # noqa
