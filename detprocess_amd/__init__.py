"""detprocess_amd -- MI355X-native engine for the detprocess of1x1 feature-extraction hot path."""

from .engine import OFPlan, SynthSource, synth_traces   # noqa: F401
from .filters import FilterTables, build_filter   # noqa: F401
from .algorithms import FeatureExtractors         # noqa: F401
from .config import YamlConfig                    # noqa: F401
from .filterdata import FilterData                # noqa: F401
from .ofbase import OFBase, search_range          # noqa: F401
from .process import FeatureProcessing            # noqa: F401
from .oftrigger import OptimumFilterTrigger       # noqa: F401
from .ofnxm import NxMPlan, build_nxm_filter      # noqa: F401

__version__ = "0.1.0"
