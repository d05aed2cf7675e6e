"""bwgr_amd -- MI355X (gfx950) Gibbs sweep engine behind bWGR's wgr()/KMUP and Bayes* samplers.

The numerical path is libbwgr_hip.so (hand-written HIP, C ABI in include/bwgr.h); this package is the host-side
mirror of the reference's R interface.  Importing the package does not need a GPU; calling it does.
"""
from .api import (Panel, Chain, Group, KMUP, KMUP2, BayesA, BayesB, BayesC, BayesL, BayesRR, BayesCpi, BayesDpi, BayesA2, BayesB2, BayesRR2, mcmcCV, fit_many, sample_rows, wgr, MODELS, emRR, emBA, emBB, emBC, emBCpi, emDE, emBL, emEN, emML, lasso, em_order,
                  debug_variates)
from ._lib import BwgrError, device_count

__all__ = ["Panel", "Chain", "Group", "KMUP", "KMUP2", "BayesA", "BayesB", "BayesC", "BayesL", "BayesRR", "BayesCpi", "BayesDpi", "BayesA2", "BayesB2", "BayesRR2", "mcmcCV", "fit_many", "emRR", "emBA", "emBB", "emBC", "emBCpi", "emDE", "emBL", "emEN", "emML", "lasso", "em_order", "sample_rows", "wgr",
           "MODELS", "BwgrError", "device_count", "debug_variates"]
