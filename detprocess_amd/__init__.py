"""detprocess_amd -- MI355X-native engine for the detprocess of1x1 feature-extraction hot path."""

from .engine import OFPlan, synth_traces          # noqa: F401
from .filters import FilterTables, build_filter   # noqa: F401

__version__ = "0.1.0"
