"""Drop-in alias for the one ggraph3D module on the threedgraph hot path: G-SphereNet's private SphereNet
(``dig.ggraph3D.method.G_SphereNet.model.spherenet.SphereNet``) resolves to the MI355X engine (dig_amd.ggraph3D)."""
