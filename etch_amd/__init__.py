"""etch_amd -- MI355X-native (gfx950) implementation of the ETCH inference hot path.

Hot path = SURVEY.md section 8: GT_network_equiv.forward (EPN encoder + heads) and fit_smpl
(marker aggregation + LM SMPL fit), behind the reference's own model / operator API.
"""
__version__ = "0.1.0"
