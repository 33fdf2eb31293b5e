"""MI355X-native sparse bundle adjustment: drop-in for the solve step of
egirgin/bundle_adjustment's ``src/bundle_adjuster.py``."""
from .map_structures import Keyframe, KeyPoint, Map, MapPoint  # noqa: F401
from .problem import BAProblem  # noqa: F401
from .bundle_adjuster import BundleAdjuster  # noqa: F401
