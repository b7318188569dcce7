"""cgat_amd -- MI355X (gfx950) implementation of the hyllios/CGAT edge-attention hot path.

`CGAtNet` and the layer classes keep the reference's constructor / forward API and
state_dict layout, so `--version cgat_amd` plugs into the reference LightningModule
(lightning_module.py:165-176).  Importing this package loads libcgat_hip.so and raises if
it is missing: there is no CPU or PyTorch-eager fallback.
"""
from . import _lib  # noqa: F401  (loads the shared library, fails loudly)
from .hypernet import H_Net, H_Net_0, HyperFC
from .mlp import ResidualNetwork, Rezero, SimpleNetwork
from .nets import CGAtNet, GATConvEdges, GATConvNodes, MHAttention, MultiHeadNetwork
from .roost import MessageLayer, Roost, WeightedAttention
from .graph import GraphBatch, synthetic_batch
from .collate import PackedDataset
from .optim import FusedAdamW, FusedLamb, RobustL1, RobustL2, cyclical_lr
from .ops import get_bilinear_mode, set_bilinear_mode, set_validate_indices, set_edge_storage, get_edge_storage
from .trainer import DataParallelTrainer, Normalizer
from .chunked import set_max_edges_per_pass
from .capture import GraphedStep
from . import debug

__all__ = ["CGAtNet", "GATConvNodes", "GATConvEdges", "MultiHeadNetwork", "MHAttention", "H_Net", "H_Net_0",
           "HyperFC", "SimpleNetwork", "ResidualNetwork", "Rezero", "Roost", "MessageLayer", "WeightedAttention",
           "GraphBatch", "synthetic_batch", "PackedDataset", "FusedAdamW", "FusedLamb", "RobustL1", "RobustL2", "cyclical_lr", "set_bilinear_mode", "get_bilinear_mode",
           "set_validate_indices", "DataParallelTrainer", "Normalizer", "set_max_edges_per_pass",
           "set_edge_storage", "get_edge_storage", "GraphedStep", "debug"]
