"""
densematcher_amd: the matching hot path of DenseMatcher (per-vertex features + Laplace-Beltrami bases ->
functional map -> vertex-to-vertex maps -> ZoomOut / ICP), MI355X-native.

    engine.MatchEngine            batched device-tensor API over the C ABI (include/densematch.h)
    functional_map.compute_surface_map, pyFM.*   the reference's Python call surface, NumPy in / NumPy out
    shard                         pairs -> ranks (one process per GPU, no data-path collective)
    synth                         synthetic meshes / eigenbases / descriptors (inputs of the path)
"""
__version__ = "0.1.0"
