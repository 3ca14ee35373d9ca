from dig_amd.ggraph3D.spherenet import SphereNet, swish  # noqa: F401
