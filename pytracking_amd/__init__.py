"""pytracking_amd -- MI355X (gfx950) implementation of PyTracking's online model-optimisation hot path.

  filter        apply_filter / apply_feat_transpose / filter_gradient        (ltr/models/layers/filter.py)
  optimizer     DiMPSteepestDescentGN / DiMPL2SteepestDescentGN / PrDiMPSteepestDescentNewton
                                                                              (ltr/models/target_classifier/optimizer.py)
  prroi_pool    PrRoIPool2D                                                   (ltr/external/PreciseRoIPooling)
  optimization  ConjugateGradient for ConvProblem                             (pytracking/libs/optimization.py)
  steepestdescent  GNSteepestDescent on LWTLResidual (LWL few-shot learner)  (ltr/models/meta/steepestdescent.py,
                                                                              ltr/models/lwl/loss_residual_modules.py)
  sequences     one sequence per GPU + end-of-batch throughput gather        (pytracking/evaluation/running.py)
  install       patch the symbols above into an importable reference tree
All compute goes through the C ABI of libpt_hot.so (include/pt_hot.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
