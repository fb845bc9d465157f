from . import models, utils, transforms  # noqa
