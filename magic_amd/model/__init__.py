from .autoencoder import GMAEModel, build_model  # noqa: F401
from .gat import GAT, GATConv  # noqa: F401
from .loss_func import sce_loss  # noqa: F401
