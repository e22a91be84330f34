"""Top-level `layers` package so the reference's `from layers.attention import *` /
`from layers.encoding import *` (models.py:4-5, train.py:23) resolve to the MI355X build."""
