"""magic_amd -- MI355X-native hot path of MAGIC (masked graph auto-encoder + KNN detector).

Public surface mirrors the reference modules: ``magic_amd.model.gat`` (GAT, GATConv),
``magic_amd.model.autoencoder`` (GMAEModel, build_model), ``magic_amd.model.loss_func``
(sce_loss), ``magic_amd.model.eval`` (KNN evaluators), ``magic_amd.utils.*`` and the
DGL-subset graph container ``magic_amd.graph`` (``Graph``, ``graph``, ``batch``).
"""
from . import ops  # noqa: F401  (registers torch.ops.magic.*)
from .graph import Graph, batch, graph  # noqa: F401

__version__ = "0.1.0"
