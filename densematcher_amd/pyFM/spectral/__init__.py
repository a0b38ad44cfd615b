from .convert import FM_to_p2p, mesh_FM_to_p2p, mesh_FM_to_p2p_precise, p2p_to_FM, mesh_p2p_to_FM, MappedIndicator  # noqa: F401
from .nn_utils import knn_query  # noqa: F401
