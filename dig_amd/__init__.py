"""dig_amd — MI355X-native (gfx950, HIP) message-passing engine for DIG's ``dig.threedgraph`` hot path.

    from dig_amd.threedgraph.method import SphereNet, DimeNetPP, SchNet, ComENet, run
    from dig_amd.threedgraph.evaluation import ThreeDEvaluator
    from dig_amd.threedgraph.utils import xyz_to_dat
    from dig_amd.ops import radius_graph, scatter, scatter_min

The package never falls back to a CPU implementation: without ``dig_amd/lib/libdig3d.so`` (built by
``python -m dig_amd.build``) or without a GPU every op raises ``dig_amd._hip.Dig3dError``.
"""
__version__ = '0.1.0'
