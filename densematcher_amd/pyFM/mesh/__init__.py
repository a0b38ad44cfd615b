from .trimesh import TriMesh  # noqa: F401
