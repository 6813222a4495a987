"""Import-name shim: ``custom_imports = dict(imports=["wedetect"])`` in the reference's config files
(config/wedetect_base.py:37) and ``from wedetect.models import ...`` in user code resolve to the MI355X-native
package ``wedetect_amd``.  Importing it fills wedetect_amd's registries and, when mmdet / mmengine are installed,
enters the same names into theirs so a stock ``mmdet.apis.init_detector`` builds the device detector."""
from wedetect_amd import __version__  # noqa: F401
from wedetect_amd import apis as _apis  # noqa: F401  (registers models and transforms)
from wedetect_amd.registry import MODELS, TRANSFORMS, register_with_mmengine

from .models import *  # noqa: F401,F403
from .datasets import *  # noqa: F401,F403

MMENGINE_REGISTERED = register_with_mmengine()
