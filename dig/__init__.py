"""Drop-in alias: ``from dig.threedgraph.method import SphereNet, run`` resolves to the MI355X engine
(``dig_amd``).  Only the threedgraph hot path exists here; DIG's other sub-packages are
out of scope."""
