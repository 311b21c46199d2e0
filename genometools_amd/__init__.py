"""MI355X-native enhanced-suffix-array construction: a drop-in for the hot
path of `gt suffixerator` (GenomeTools).  See DESIGN.md."""
from .esa import EsaEngine, EsaResult, suffixerator_tables  # noqa: F401
from . import synth  # noqa: F401
