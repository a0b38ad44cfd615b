from .icp import icp_iteration, icp_refine, mesh_icp_refine  # noqa: F401
from .zoomout import zoomout_iteration, zoomout_refine, mesh_zoomout_refine, mesh_zoomout_refine_p2p  # noqa: F401
