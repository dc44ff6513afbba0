"""regt-gcn_amd: MI355X-native implementation of the RegT-GCN forward/backward hot path.

The directory name is not a Python identifier; import it as ``regtgcn_amd`` (the repo-root shim
``regtgcn_amd.py`` registers this package under that name).
"""
from . import _lib
from ._lib import RegtError, load as load_library
from .graph import PreparedGraph, prepare_graph
from .functional import RegTGCNFunction, regt_gcn_forward, param_names
from . import ops, data, dist, etl, train, evaluate
from .nn import (A3TGCN, GAT, ConvStackedA3TGCN, ConvStackedTemporalGCN, GATTemporal, GraphSAGE, GraphSAGETemporalGCN, RegionalA3TGCN,
                 RegionalTemporalGCN, TemporalGCN, TGCN)

__all__ = ["RegtError", "load_library", "PreparedGraph", "prepare_graph", "RegTGCNFunction", "regt_gcn_forward",
           "param_names", "RegionalTemporalGCN", "RegionalA3TGCN", "TemporalGCN", "A3TGCN", "TGCN", "ConvStackedTemporalGCN",
           "ConvStackedA3TGCN", "GraphSAGETemporalGCN", "GraphSAGE", "GATTemporal", "GAT"]
